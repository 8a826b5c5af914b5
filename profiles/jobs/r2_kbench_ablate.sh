# round 2: where the time of the grouped STRIPED kernel goes (ablations are timing-only)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 900 python tools/kbench.py --reps 50 --cases C,B,D/8 --plans striped --tunings xcd1nt0 --abl 4 > gpurun_out/r2a/kbench_ablate.log 2>&1
echo "rc=$?" >> gpurun_out/r2a/kbench_ablate.log
cat gpurun_out/r2a/kbench_ablate.log
