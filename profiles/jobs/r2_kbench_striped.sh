# round 2: the grouped STRIPED kernel on the NPB shapes, stripe-group counts swept (tools/kbench.py)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 900 python tools/kbench.py --reps 50 --cases C,B,D/8 --plans striped --tunings xcd1nt0,xcd0nt0 --sgroups 0,5,2,1,1000 > gpurun_out/r2a/kbench_striped.log 2>&1
echo "rc=$?" >> gpurun_out/r2a/kbench_striped.log
cat gpurun_out/r2a/kbench_striped.log
