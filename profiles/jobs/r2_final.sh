# round 2, final code: GPU suite + smoke, the bench line, rocprofv3 trace of the bench command, PMC traffic passes, N = 2 rehearsal
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2c
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
timeout -k 10 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "trace rc=$?"
python3 tools/trace_summary.py $OUT/trace/bench_kernel_trace.csv > $OUT/bench_kernel_by_grid.csv; head -8 $OUT/bench_kernel_by_grid.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python3 bench.py --no-cpu --no-npb --steps 20 --warmup 2 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err; echo "pmc $c rc=$?"
done
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv > $OUT/pmc_traffic.json
rm -rf $OUT/trace/*.db $OUT/pmc_*/*.db $OUT/trace/bench_kernel_trace.csv 2>/dev/null
BENCH_DIST_BACKEND=gloo timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 2 --steps 20 --warmup 3 > $OUT/bench_n2_rehearsal.json 2> $OUT/bench_n2_rehearsal.err; echo "bench n2 rc=$?"
python3 -c "
import json
d=json.load(open('$OUT/bench_n1.json'))
print('N1 value', d['value'], 'ms/step', d['ms_per_step'], 'launch_us', d['roofline']['launch_us'], 'frac', d['roofline']['frac'], 'untimed', d['config']['untimed_steps_before_timing'])
print('npb_cg', d['npb_cg']['mops'], d['npb_cg']['time_s'], d['npb_cg']['us_per_abi_call'], '| device cg', d['npb_cg_device']['mops'])
print('f32', d['spmv_f32']['us_per_spmv'], d['spmv_f32']['roofline_frac'], '| scale_ref', d['scale_ref']['ms_per_step'], d['scale_ref']['roofline_frac'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['ms_per_spmv'])
d=json.loads([l for l in open('$OUT/bench_n2_rehearsal.json') if l.startswith('{')][-1])  # gloo prints to stdout too
print('N2 rehearsal value', d['value'], d['ms_per_step'], 'gather', d['gather_inclusive']['value'], d['gather_inclusive']['ms_per_step'])
t=json.load(open('$OUT/pmc_traffic.json')); print(t['hbm_bytes_per_spmv'])
"
