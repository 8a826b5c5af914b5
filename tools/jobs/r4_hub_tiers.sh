# round 4: the HUB plan's second tier (a dense table in memory behind the LDS table): parity first, then time by table size
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r4f
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "hub or all_plans_on_edge or all_plans_on_benchmark or fortran" > $OUT/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -15 $OUT/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python3 tools/hubbench.py --cases rmat20,rmat22,rmat18 --tables 0,262144,1048576,2097152,4194304,8388608 > $OUT/hubbench.log 2>&1; echo "hubbench rc=$?"; cat $OUT/hubbench.log
