cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3hc; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o hc -- python3 tools/hotcoldbench.py 22 --k16384 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
grep "^rmat" $O/run.log
