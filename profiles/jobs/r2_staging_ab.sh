# round 2: arrival-ordered staging of the later stripes (tuning key 6 bit 256 = the static scheme), interleaved A/B;
# parity of the STRIPED tests first
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/stg
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "striped or class_d or class_c" > gpurun_out/stg/tests.log 2>&1; rc=$?; tail -5 gpurun_out/stg/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python tools/kbench.py --reps 30 --cases D/8,D/4,C,B,D --plans striped --tunings xcd1nt0 --abl 256 --ab-rounds 10 > gpurun_out/stg/kbench_ab.log 2>&1
echo "rc=$?" >> gpurun_out/stg/kbench_ab.log
cat gpurun_out/stg/kbench_ab.log
