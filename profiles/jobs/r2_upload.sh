cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/upload
cat /sys/fs/cgroup/cpu.max > gpurun_out/upload/default_threads.log 2>&1
timeout -k 10 900 python tools/upload_breakdown.py C,D >> gpurun_out/upload/default_threads.log 2>&1
cat gpurun_out/upload/default_threads.log
