cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3e
timeout -k 10 600 python3 bench.py > gpurun_out/r3e/bench_n1.json 2> gpurun_out/r3e/bench_n1.err; echo "n1 rc=$?"
tail -c 3000 gpurun_out/r3e/bench_n1.json; tail -5 gpurun_out/r3e/bench_n1.err
timeout -k 10 600 python3 bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/r3e/bench_n2_inproc.json 2> gpurun_out/r3e/bench_n2_inproc.err; echo "n2 rc=$?"
tail -c 2500 gpurun_out/r3e/bench_n2_inproc.json; tail -5 gpurun_out/r3e/bench_n2_inproc.err
