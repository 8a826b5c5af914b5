# round 4: AUTO against every plan by name (HUB among them) on the audit's grid, f64 and f32; HBM traffic of the HUB plan from two --pmc passes
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4audit
mkdir -p $OUT
timeout -k 10 900 python3 tools/dispatch_audit.py > $OUT/audit.log 2>&1; echo "audit rc=$?"; tail -25 $OUT/audit.log
timeout -k 10 600 python3 tools/dispatch_audit.py --f32 > $OUT/audit_f32.log 2>&1; echo "audit f32 rc=$?"; tail -3 $OUT/audit_f32.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python3 tools/kbench.py --reps 5 --cases rmat20,rmat22 --plans hub,stream --tunings xcd1nt0 > $OUT/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
python3 tools/sq_summary.py $OUT k_hub,k_hub_fix,k_hub_pack,k_stream,k_long,k_long_finish rmat20,rmat22 > $OUT/hub_traffic_counters.json; head -c 3000 $OUT/hub_traffic_counters.json
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +30M -delete
