cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3blk
mkdir -p $O
timeout -k 10 300 python3 tools/blockedbench.py 170 130 100 2>&1 | tee $O/blockedbench2.log
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; rc=$?; tail -15 $O/pytest_all.log; [ $rc = 0 ] || exit 1
