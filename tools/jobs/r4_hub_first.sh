# round 4: the HUB plan's first run -- parity (edge cases under every plan, bit identity with tests/hub_order.py), then time
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r4d
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "hub_is_bit or all_plans_on_edge or all_plans_on_benchmark" > $OUT/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -15 $OUT/tests.log
[ $rc -eq 0 ] && timeout -k 10 500 python3 tools/kbench.py --reps 30 --cases rmat20,rmat22,rmat18 --plans hub,stream,auto --tunings xcd1nt0 > $OUT/kbench_hub.log 2>&1; echo "kbench rc=$?"; cat $OUT/kbench_hub.log
