#!/bin/bash
# A/B of two BUILDS of libbpmi.so on one box (BPMI_LIB): the headline bench in the driver's form without the extras, alternating,
# $2 rounds.   bash tools/r06_ab_builds.sh ab_build/libbpmi_before_format3.so 4
other=$1; rounds=${2:-4}
for r in $(seq 1 $rounds); do
  for lib in tree "$other"; do
    if [ "$lib" = tree ]; then unset BPMI_LIB; else export BPMI_LIB=$PWD/$lib; fi
    python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --soak-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('round $r  %-40s %.4f ms per step' % ('$lib', d['ms_per_step']))"
  done
done
