cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3blk; mkdir -p $O
echo "f32: $(timeout -k 10 120 python3 tools/blockedbench.py 160 170 --f32 | cut -d';' -f1-2 | cut -c1-160)" | tee $O/f32_after.log
echo "f64: $(timeout -k 10 120 python3 tools/blockedbench.py 160 170 | cut -d';' -f1-2| cut -c1-160)" | tee -a $O/f32_after.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked or all_plans" > $O/pytest3.log 2>&1; rc=$?; tail -5 $O/pytest3.log; [ $rc = 0 ] || exit 1
