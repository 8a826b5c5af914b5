# round 4, last check: the whole GPU suite and smoke() on the final tree, bench.py once more (defaults of the driver)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4final2
mkdir -p $OUT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/tests_all.log 2>&1; rc=$?; echo "all gpu tests rc=$rc"; tail -4 $OUT/tests_all.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
