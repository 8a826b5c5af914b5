# round 4, call 1 (re-run after the container was replaced): evidence for VERDICT r3 items 1a, 2a + the HUB plan's numbers
#  - HUB / fragment tests, kbench hub vs stream, read ceilings (ntstream), timing-only ablations of k_stream
#  - SQ counters of k_stream and k_hub on R-MAT 2^20 / 2^22 (two --pmc passes, kernel trace only)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4e1
mkdir -p $OUT
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "hub or fragment or cache_and_write or all_plans_on_edge" > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $OUT/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 tools/bin/ntstream 30 > $OUT/ntstream.log 2>&1; echo "ntstream rc=$?"; cat $OUT/ntstream.log
timeout -k 10 400 python3 tools/kbench.py --reps 30 --cases rmat20,rmat22,rmat18 --plans hub,stream,auto --tunings xcd1nt0 > $OUT/kbench_hub.log 2>&1; echo "kbench rc=$?"; cat $OUT/kbench_hub.log
timeout -k 10 400 python3 tools/streamabl.py --cases rmat20,rmat22 --rounds 4 > $OUT/streamabl.log 2>&1; echo "streamabl rc=$?"; cat $OUT/streamabl.log
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" ; do
  p=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$p -o pmc -- \
      python3 tools/kbench.py --reps 10 --cases rmat20,rmat22 --plans stream,hub --tunings xcd1nt0 > $OUT/$p.log 2>&1
  echo "pass $p rc=$?"
done
python3 tools/sq_summary.py $OUT k_stream,k_long,k_hub > $OUT/stream_sq_counters.json; cat $OUT/stream_sq_counters.json
rm -rf $OUT/*/*.db $OUT/*/*/*.db 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +20M -delete
echo done
