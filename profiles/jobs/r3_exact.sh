cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3p
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "documented_order" > $O/pytest_exact.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_exact.log; tail -12 $O/pytest_exact.log
timeout -k 10 500 python3 tools/shapebench.py --cases C,D/8,B --configs 1:0:0 --rounds 8 > $O/shapebench_contract_off.log 2>&1; cat $O/shapebench_contract_off.log
