# round 4: new tests first (multi-partition device CG, bench keys, Fortran caller, hub), then the whole GPU suite, then the class C investigation
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4h
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "row_partitions or parts_refuse or bench_line or fortran or auto_takes_blocked" > $OUT/tests_new.log 2>&1; rc=$?; echo "new tests rc=$rc"; tail -25 $OUT/tests_new.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/kbench.py --cases C,B --plans striped --tunings xcd1nt0 --ab-rounds 6 --abl 16,28,32 > $OUT/abl.log 2>&1; echo "abl rc=$?"; cat $OUT/abl.log
for tk in 3 4; do
  timeout -k 10 200 python3 tools/kbench.py --cases C --plans striped --tunings xcd1nt0 --tk $tk --ab-rounds 6 --abl 256 > $OUT/ring$tk.log 2>&1; echo "ring $tk rc=$?"; grep "A/B" $OUT/ring$tk.log
done
timeout -k 10 200 python3 tools/kbench.py --cases C --plans striped --tunings xcd1nt0 --ab-rounds 6 --abl 256 > $OUT/ring2.log 2>&1; grep "A/B" $OUT/ring2.log
timeout -k 10 300 python3 tools/kbench.py --cases C --plans striped --tunings xcd1nt0 --stamps --stamps-launches 40 > $OUT/stamps.log 2>&1; echo "stamps rc=$?"; cat $OUT/stamps.log
