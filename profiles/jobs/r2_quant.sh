cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/stamps
timeout -k 10 800 python tools/quant_probe.py > gpurun_out/stamps/quant.log 2>&1; echo "rc=$?" >> gpurun_out/stamps/quant.log
cat gpurun_out/stamps/quant.log
