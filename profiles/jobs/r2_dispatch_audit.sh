cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 1100 python tools/dispatch_audit.py > gpurun_out/r2a/dispatch_audit.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/dispatch_audit.log
cat gpurun_out/r2a/dispatch_audit.log
