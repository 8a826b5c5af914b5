cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3h
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
BENCH_DIST_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_n2_launcher.json 2> $O/bench_n2_launcher.err; echo "launcher n2 rc=$?"
tail -c 1800 $O/bench_n2_launcher.json; tail -3 $O/bench_n2_launcher.err
timeout -k 10 500 python3 tools/dispatch_audit.py > $O/dispatch_audit.log 2>&1; echo "audit rc=$?"; grep -c . $O/dispatch_audit.log; grep "behind" $O/dispatch_audit.log
( for pin in auto register; do for ov in 1 0; do echo "== spmvdrv sb 170, SPMV_HIP_PIN=$pin SPMV_HIP_YOVERLAP=$ov"; SPMV_HIP_PIN=$pin SPMV_HIP_YOVERLAP=$ov lilac-benchmarks_amd/lib/spmvdrv sb lilac-benchmarks_amd/lib/libhip-spmv.so 170 | grep "Matrix multiply"; done; done ) > $O/sb170_driver.log 2>&1
cat $O/sb170_driver.log
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/cold -o cold -- python3 tools/coldstart.py > $O/coldstart.log 2>&1; echo "cold rc=$?"
python3 tools/coldstart.py --summarise $O/cold/cold_kernel_trace.csv | tee $O/coldstart_summary.log
rm -rf $O/cold/*.db $O/cold/cold_kernel_trace.csv 2>/dev/null
