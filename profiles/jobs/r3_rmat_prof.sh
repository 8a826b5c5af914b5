cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3rm; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o rm -- python3 tools/rmatbench.py 20 22 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
grep "^rmat" $O/run.log
