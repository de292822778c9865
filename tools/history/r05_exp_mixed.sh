#!/bin/bash
# Round 5: mixed window widths (256 // c windows, the last 256 - (256 // c) c of them c + 1 bits wide): which c at which size
out=${1:-gpurun_out/r05_mixed_sweep.txt}
export R5_CONFIGS="def:;c10:window_bits=10;c11:window_bits=11;c12:window_bits=12;c13:window_bits=13;c14:window_bits=14;c15:window_bits=15;c16:window_bits=16;c13u:window_bits=13,mixed_windows=0"
R5_STAGES=1 timeout 1700 python tools/r05_ab_mid.py 6144 9000 12288 16384 24576 32768 49152 65536 98304 131072 196608 262144 311427 > "$out" 2>&1
grep "^##" "$out" | grep -v stages
