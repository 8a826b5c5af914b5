cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3blk; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked" > $O/pytest4.log 2>&1; rc=$?; tail -5 $O/pytest4.log; [ $rc = 0 ] || exit 1
timeout -k 10 600 python3 tools/rmatbench.py 19 20 21 22 2>&1 | tee $O/rmatbench.log
timeout -k 10 200 python3 tools/blockedbench.py 170 130 2>&1 | tee -a $O/rmatbench.log
