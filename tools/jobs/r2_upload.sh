cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/upload
timeout -k 10 900 python tools/upload_breakdown.py C,D > gpurun_out/upload/breakdown.log 2>&1; echo "rc=$?" >> gpurun_out/upload/breakdown.log
cat gpurun_out/upload/breakdown.log
