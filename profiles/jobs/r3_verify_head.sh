cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3z
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 600 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"
python3 -c "
import json;d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1]);print(d['value'],d['roofline']['frac'],d['cold']['frac'],d['npb_cg']['mops'],d['npb_cg']['upload_s'],d['scale_ref']['roofline_frac'],d['parboil'])"
