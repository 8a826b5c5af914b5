cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3runs; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked or all_plans" > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc = 0 ] || exit 1
timeout -k 10 200 python3 tools/blockedbench.py 170 130 100 2>&1 | tee $O/runs2.log
timeout -k 10 200 python3 tools/blockedbench.py 130 --plant600000 2>&1 | tee -a $O/runs2.log
timeout -k 10 200 python3 tools/blockedbench.py 170 --f32 2>&1 | tee -a $O/runs2.log
timeout -k 10 400 python3 tools/rmatbench.py 20 22 2>&1 | tee -a $O/runs2.log
timeout -k 10 600 python3 tools/hotcoldbench.py 22 --k16384 2>&1 | tee -a $O/runs2.log
