# everything the driver runs at round end, plus the bench line
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2b
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r2b/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2b/pytest_gpu.log
tail -6 gpurun_out/r2b/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2b/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r2b/smoke.log
tail -2 gpurun_out/r2b/smoke.log
timeout -k 10 900 python bench.py > gpurun_out/r2b/bench_n1.json 2> gpurun_out/r2b/bench_n1.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r2b/bench_n1.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'launch_us', d['roofline']['launch_us'], 'frac', d['roofline']['frac'])
print('npb_cg', d['npb_cg']['mops'], d['npb_cg']['time_s'], d['npb_cg']['us_per_abi_call'], '| device cg', d['npb_cg_device']['mops'])
print('f32', d['spmv_f32']['us_per_spmv'], d['spmv_f32']['roofline_frac'], '| scale_ref', d['scale_ref']['ms_per_step'], d['scale_ref']['roofline_frac'], d['scale_ref'].get('workgroups'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['ms_per_spmv'])
"
timeout -k 10 600 python tools/kbench.py --reps 50 --cases C,D/8 --plans striped --tunings xcd1nt0,xcd0nt0 > gpurun_out/r2b/kbench_xcd.log 2>&1; tail -8 gpurun_out/r2b/kbench_xcd.log
timeout -k 10 600 python tools/kbench.py --reps 50 --cases C,D/8 --plans striped --tunings xcd1nt0 --tk 3 > gpurun_out/r2b/kbench_r3.log 2>&1; tail -4 gpurun_out/r2b/kbench_r3.log
