#!/bin/bash
# Round 5: the spread finish and the wave-fit rule of the bucket reduction, A/B in one process (tools/r05_ab_mid.py)
out=${1:-gpurun_out/r05_reduce_ab2.txt}
export R5_STAGES=1
export R5_CONFIGS="q:final_spread=0,reduce_fit=0;q_fit:final_spread=0;s1:final_spread=1;s2:final_spread=2;s3:final_spread=3;c15_q_r4:window_bits=15,final_spread=0,reduce_fit=0;c15_q:window_bits=15,final_spread=0;c15_s2:window_bits=15,final_spread=2;c15_s3:window_bits=15,final_spread=3;c14_s2:window_bits=14,final_spread=2;c13_s2:window_bits=13,final_spread=2"
timeout 1500 python tools/r05_ab_mid.py 16384 32768 65536 131072 311427 > "$out" 2>&1
R5_CONFIGS="q:final_spread=0,reduce_fit=0;s2:final_spread=2;s3:final_spread=3;c15_s2:window_bits=15,final_spread=2" timeout 600 python tools/r05_ab_mid.py 1048576 >> "$out" 2>&1
grep "^##" "$out" | grep -v stages
grep "stages" "$out" | grep "n=  65536" | cut -c1-250
