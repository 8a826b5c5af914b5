# round 4: the N = 2 code paths of bench.py rehearsed on the one GPU (two row blocks of class D wrapped onto the card): one process
# driving both, and two gloo ranks under the launcher; each line carries npb_cg AND npb_cg_device (spmv_hip_cg_create_multi)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r4n2
mkdir -p $O
timeout -k 10 900 python3 bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_n2_inproc.json 2> $O/bench_n2_inproc.err; echo "inproc n2 rc=$?"; tail -c 1500 $O/bench_n2_inproc.json
BENCH_DIST_BACKEND=gloo timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_n2_launcher.json 2> $O/bench_n2_launcher.err; echo "launcher n2 rc=$?"; tail -c 600 $O/bench_n2_launcher.json
timeout -k 10 900 python3 bench.py --gpus 8 --steps 10 --warmup 2 --no-npb > $O/bench_n8_inproc_kernels_only.json 2> $O/bench_n8.err; echo "inproc n8 (kernels only) rc=$?"; cut -c1-400 $O/bench_n8_inproc_kernels_only.json
