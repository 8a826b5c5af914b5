cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3x
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 600 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"
python3 -c "
import json;d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1]);print(d['value'],d['roofline']['frac'],d['roofline']['launch_us'],d['cold']['frac'],d['npb_cg']['mops'],d['npb_cg_device']['mops'],d['scale_ref']['roofline_frac'],d['scale_ref']['ms_per_step']);print(d['sparsebench_170']);print(d['graphs_rmat20'])"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "trace rc=$?"
python3 tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/bench_kernel_by_grid.csv; head -8 $O/bench_kernel_by_grid.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python3 bench.py --no-cpu --no-npb --no-configs --steps 20 --warmup 2 > $O/pmc_$c.json 2> $O/pmc_$c.err; echo "pmc $c rc=$?"
done
python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE/pmc_counter_collection.csv $O/pmc_WRITE_SIZE/pmc_counter_collection.csv > $O/pmc_traffic.json; grep -A3 "hbm_bytes_per_spmv" $O/pmc_traffic.json
rm -rf $O/trace/*.db $O/pmc_*/*.db 2>/dev/null
