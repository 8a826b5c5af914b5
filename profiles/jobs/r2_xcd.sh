cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2b
timeout -k 10 1100 python tools/kbench.py --reps 30 --cases B,C,D/8,D/4,D/2,D --plans striped --tunings xcd1nt0,xcd0nt0,xcd1nt0,xcd0nt0 > gpurun_out/r2b/kbench_xcd_all.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/kbench_xcd_all.log
grep -v "^#" gpurun_out/r2b/kbench_xcd_all.log
