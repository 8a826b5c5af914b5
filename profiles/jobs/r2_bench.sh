# round 2: the driver's N = 1 bench line, then the N = 2 code path rehearsed with two ranks sharing the one GPU (gloo)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2a
timeout -k 10 900 python bench.py > gpurun_out/r2a/bench_n1.json 2> gpurun_out/r2a/bench_n1.err; echo "bench n1 rc=$?"
cat gpurun_out/r2a/bench_n1.json
BENCH_DIST_BACKEND=gloo timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/r2a/bench_n2_rehearsal.json 2> gpurun_out/r2a/bench_n2_rehearsal.err; echo "bench n2 rc=$?"
cat gpurun_out/r2a/bench_n2_rehearsal.json; tail -5 gpurun_out/r2a/bench_n2_rehearsal.err
