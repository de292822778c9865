#!/bin/bash
# GPU clocks sampled while the headline loop runs plain and with an RCCL process group (no exchange): is the 6 % a clock difference?
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
sample() { for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|socclk\|mclk\|fclk\|Power (W)\|Socket" | tr '\n' ' '; echo; sleep 0.25; done; }
for mode in plain group plain group; do
  unset BENCH_FORCE_DIST BENCH_EXCHANGE
  [ $mode = group ] && export BENCH_FORCE_DIST=1 BENCH_EXCHANGE=0
  sample > /tmp/clk_$mode.txt &
  SP=$!
  python3 bench.py --gpus 1 --steps 4000 --warmup 6 --no-cpu-baseline --soak-seconds 0 --no-extra 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('%-6s ms_per_step %.4f' % ('$mode', d['ms_per_step']))"
  kill $SP 2>/dev/null; wait $SP 2>/dev/null
  sort /tmp/clk_$mode.txt | uniq -c | sort -rn | head -4
done
