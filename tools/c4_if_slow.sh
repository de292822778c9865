#!/bin/bash
# the cProfile of config C4's prover, printed only when this box runs it slowly (some boxes of the pool take 3x the wall time with the same GPU time)
timeout 200 python3 tools/profile_c4.py > gpurun_out/c4_slowcheck.txt 2>&1
ms=$(grep -m1 "prove ms" gpurun_out/c4_slowcheck.txt | awk '{print int($3)}')
echo "C4 prove: ${ms} ms on $(hostname)"
if [ "${ms:-0}" -gt 18 ]; then head -40 gpurun_out/c4_slowcheck.txt; python3 - <<'PY'
import os, time
print("load", os.getloadavg(), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
try: print(open("/sys/fs/cgroup/cpu.stat").read())
except Exception as e: print(e)
PY
fi
