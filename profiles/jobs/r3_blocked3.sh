cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3blk
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked_plan" > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log; [ $rc = 0 ] || exit 1
timeout -k 10 600 python3 tools/blockedbench.py 170 150 130 100 2>&1 | tee $O/blockedbench.log
