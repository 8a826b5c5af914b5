# is a workgroup's lateness at the end of a k_striped launch a matter of its place on the chip?
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/stamps
for i in 1 2; do
timeout -k 10 600 python tools/kbench.py --reps 30 --cases C --plans striped --tunings xcd1nt0,xcd0nt0 --stamps --stamps-launches 30 >> gpurun_out/stamps/places2.log 2>&1; echo "rc=$?" >> gpurun_out/stamps/places2.log
done
cat gpurun_out/stamps/places2.log
