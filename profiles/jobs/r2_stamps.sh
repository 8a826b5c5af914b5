cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2b
timeout -k 10 900 python tools/kbench.py --reps 80 --cases C,B,D/8 --plans striped --tunings xcd1nt0 --abl 32 --ab-rounds 16 --stamps > gpurun_out/r2b/kbench_stamps.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/kbench_stamps.log
cat gpurun_out/r2b/kbench_stamps.log
