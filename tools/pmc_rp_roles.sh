#!/bin/bash
# Per-role hardware counters of k_rp_prepare (config C5, 2^14 proofs; run from the repo root ON THE GPU BOX):
#   bash tools/pmc_rp_roles.sh >> gpurun_out/pmc_rp_prepare.txt
# One profiling run per role (option rp_only_role): instructions, lifetime and wait cycles of one wave of that role.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_roles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C5_PINNED=1 C5_PREPARE=device
for r in 0 1 2 3; do
  C5_ONLY_ROLE=$r timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/r$r -- python3 $R/tools/profile_c5.py > $OUT/r$r.log 2>&1
  f=$(find $OUT/r$r -name "*counter_collection.csv" | head -1)
  echo "role $r"; python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "k_rp_prepare" in row.get("Kernel_Name", ""):
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("  %-20s per active wave %.4g" % (k, sum(v) / len(v) / 256))
PY
done
rm -rf $OUT        # the raw counter traces are tens of MB; only the summary above is kept
