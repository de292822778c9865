R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/r04big}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/tl -o tl -- python3 $R/tools/c3_round_times.py 20 "$@" > $OUT/rounds.txt 2> $OUT/tl.err
DB=$(find $OUT/tl -name "*results.db" | head -1)
# the last proof's second large round: accumulate launches per proof = 40 (20 rounds x 2) minus late rounds...; take the 3rd accumulate from the end of the multifold backwards
python3 $R/tools/rocpd_timeline.py $DB nth k_accum_l0 2 60 14 > $OUT/timeline_big_round.txt
rm -rf $OUT/tl
cat $OUT/timeline_big_round.txt
