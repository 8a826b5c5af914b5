# round 2: all of NPB class D on one GPU (bench.py's scale_ref workload), stripe-group counts swept
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 1100 python tools/kbench.py --reps 20 --cases D --plans striped --tunings xcd1nt0 --sgroups 0,1,3,4 --abl 4 > gpurun_out/r2a/kbench_D.log 2>&1
echo "rc=$?" >> gpurun_out/r2a/kbench_D.log
cat gpurun_out/r2a/kbench_D.log
