#!/bin/bash
# Round 5: where c = 15 (unsigned last window, 12-lane sums) beats the neighbours, and the finish variants under the headline
out=${1:-gpurun_out/r05_reduce_ab3.txt}
export R5_CONFIGS="def_s2:final_spread=2;def_s3:final_spread=3;c15_s2:window_bits=15,final_spread=2;c15_s3:window_bits=15,final_spread=3;c16_s3:window_bits=16,final_spread=3;c12_s3:window_bits=12,final_spread=3"
timeout 1500 python tools/r05_ab_mid.py 6144 8192 10240 12288 16384 20480 24576 49152 98304 131072 196608 262144 311427 > "$out" 2>&1
grep "^##" "$out"
for k in 1 2 3; do for fs in 0 2 3; do
  echo "headline final_spread=$fs: $(timeout 300 python bench.py --steps 50 --warmup 10 --no-extra --no-cpu-baseline --opt final_spread=$fs 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["result_ok"])')" | tee -a "$out"
done; done
