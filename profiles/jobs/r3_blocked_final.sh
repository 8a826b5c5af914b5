cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3bf
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 tools/dispatch_audit.py > $O/audit_f64.log 2>&1; echo "audit f64 rc=$?"; tail -1 $O/audit_f64.log
timeout -k 10 600 python3 tools/dispatch_audit.py --f32 > $O/audit_f32.log 2>&1; echo "audit f32 rc=$?"; tail -1 $O/audit_f32.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o kb -- python3 tools/kbench.py --reps 20 --cases sb170 --plans blocked,stream --tunings xcd1nt0 > $O/kbench_trace.log 2>&1; echo "trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python3 tools/kbench.py --reps 5 --cases sb170 --plans blocked,stream --tunings xcd1nt0 > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
rm -rf $O/*/*.db 2>/dev/null
python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE/pmc_counter_collection.csv $O/pmc_WRITE_SIZE/pmc_counter_collection.csv > $O/pmc_traffic.json; echo "traffic rc=$?"
timeout -k 10 600 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"
python3 -c "
import json;d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1]);print(d['value'],d['roofline']['frac'],d['sparsebench_170'],d['graphs_rmat20'])"
