cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3i
mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked_plan" > $O/pytest_blocked.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_blocked.log
tail -15 $O/pytest_blocked.log
grep -q "pytest rc=0" $O/pytest_blocked.log && timeout -k 10 400 python3 tools/kbench.py --reps 30 --cases sb170,sb100,rmat20,grid2000 --plans stream,blocked --tunings xcd1nt0 > $O/kbench_blocked.log 2>&1
cat $O/kbench_blocked.log
