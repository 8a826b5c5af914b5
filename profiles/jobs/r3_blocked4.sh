cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3blk
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked" > $O/pytest2.log 2>&1; rc=$?; tail -15 $O/pytest2.log; [ $rc = 0 ] || exit 1
timeout -k 10 1000 python3 tools/dispatch_audit.py > $O/audit.log 2>&1; rc=$?; tail -30 $O/audit.log; [ $rc = 0 ] || exit 1
