cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3blk
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof32 -o blk -- python3 tools/blockedbench.py 160 --f32 > $O/prof32_run.log 2>&1 || { tail -5 $O/prof32_run.log; exit 1; }
grep "^sb" $O/prof32_run.log
