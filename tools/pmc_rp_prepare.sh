#!/bin/bash
# Hardware counters of k_rp_prepare (config C5, 2^14 proofs; run from the repo root ON THE GPU BOX):
#   bash tools/pmc_rp_prepare.sh > gpurun_out/pmc_rp_prepare.txt
# Separate --pmc passes with --kernel-trace only (no --stats / sys-trace next to --pmc).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_rp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C5_PINNED=1 C5_PREPARE=device
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_WAIT_IFETCH" "TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/profile_c5.py > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "k_rp_prepare" in row.get("Kernel_Name", ""):
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("%-32s per launch %.4g  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
done
rm -rf $OUT        # the raw counter traces are tens of MB; only the summary above is kept
