# round 3, first call: where the round-2 kernels stand on this round's boxes, and the PMC evidence VERDICT r2 asked for
# (k_stream on SparseBench 170^3 and R-MAT 2^20, k_striped on the class D 1/8 row block)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3a
mkdir -p $OUT
timeout -k 10 500 python3 tools/kbench.py --reps 60 --cases D/8,D/4,C --plans striped --tunings xcd1nt0 --abl 16,4 --ab-rounds 6 --stamps > $OUT/kbench_base.log 2>&1; echo "rc=$?" >> $OUT/kbench_base.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python3 tools/kbench.py --reps 5 --cases sb170,rmat20,D/8 --plans stream,striped --tunings xcd1nt0 > $OUT/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
rm -rf $OUT/pmc_*/*.db 2>/dev/null
ls -la $OUT $OUT/pmc_FETCH_SIZE | head -30
tail -40 $OUT/kbench_base.log
