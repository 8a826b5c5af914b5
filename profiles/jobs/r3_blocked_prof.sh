cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3blk
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o blk -- python3 tools/blockedbench.py 170 > $O/prof_run.log 2>&1 || { tail -5 $O/prof_run.log; exit 1; }
grep "^sb" $O/prof_run.log
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo $f; head -12 $f | cut -c1-160
