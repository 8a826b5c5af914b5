# round 4, first call: evidence before code
#  - SQ counters of k_stream on R-MAT 2^20 / 2^22 (two --pmc passes, kernel trace only)
#  - timing-only ablations of k_stream (debug library): phase 2 or the gather wait?
#  - read ceilings with k_striped's launch and load shapes, ordinary and non-temporal, at 366 MB / 868 MB / 6.9 GB
#  - the new fragment-edit test
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4a
mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "fragment or cache_and_write" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout -k 10 300 tools/bin/ntstream 30 > $OUT/ntstream.log 2>&1; echo "ntstream rc=$?"; cat $OUT/ntstream.log
timeout -k 10 600 python3 tools/streamabl.py --cases rmat20,rmat22 > $OUT/streamabl.log 2>&1; echo "streamabl rc=$?"; cat $OUT/streamabl.log
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" ; do
  p=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$p -o pmc -- \
      python3 tools/kbench.py --reps 10 --cases rmat20,rmat22 --plans stream --tunings xcd1nt0 > $OUT/$p.log 2>&1
  echo "pass $p rc=$?"
done
python3 tools/sq_summary.py $OUT k_stream,k_long > $OUT/stream_sq_counters.json; cat $OUT/stream_sq_counters.json
rm -rf $OUT/*/*.db $OUT/*/*/*.db 2>/dev/null
echo done
