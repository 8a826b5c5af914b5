# the default bench line once more on whatever box this call gets (box-to-box spread of the headline)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/spread
n=$(ls gpurun_out/spread 2>/dev/null | wc -l)
rocm-smi --showproductname 2>/dev/null | grep -i -m2 "card series\|GFX" > gpurun_out/spread/box_$$.txt
timeout -k 10 900 python bench.py > gpurun_out/spread/bench_$$.json 2> gpurun_out/spread/bench_$$.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/spread/bench_$$.json") if l.startswith("{")][-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "launch_us", d["roofline"]["launch_us"],
      "D", d["scale_ref"]["ms_per_step"], d["scale_ref"]["roofline_frac"], "npb", d["npb_cg"]["mops"], d["npb_cg_device"]["mops"])
PY
