#!/bin/bash
# which reduction tree of k_digit_sums: four builds, the same loop of synchronous 2^16-pair MSMs under the kernel trace.
# (Record of an experiment: the builds libbpmi_exp_{XOR,NOPRED}.so came from -DBPMI_EXP_TREE_XOR / _NOPRED switches in k_digit_sums that were
# removed once the winner -- butterfly for power-of-two groups, every lane adding otherwise -- was in; profiles/r05_reduction_tree_forms.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_exp_tree
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in OLD "" XOR NOPRED OLD ""; do
  lib=$R/python-bulletproofs_amd/libbpmi${v:+_exp_$v}.so
  export BPMI_LIB=$lib
  opts="reduce_fit=0 final_spread=0"; [ "$v" = OLD ] && opts=""
  timeout 300 rocprofv3 --kernel-trace -d $OUT/tl -o tl -- python3 $R/tools/r05_msm_loop.py 65536 60 $opts > $OUT/run.txt 2>&1
  DB=$(find $OUT/tl -name "*results.db" | head -1)
  echo "#### build ${v:-NEW}: $(tail -1 $OUT/run.txt)"
  python3 $R/tools/rocpd_timeline.py $DB stats | grep -E "k_digit|k_accum|k_segscan"
  rm -rf $OUT/tl
done
