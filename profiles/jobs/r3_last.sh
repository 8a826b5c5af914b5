cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3o
mkdir -p $O
( echo "# spmvdrv sb 170 with SPMV_HIP_STATS=2 (a synchronisation between the phases of every call attributes the time)"
  SPMV_HIP_STATS=2 lilac-benchmarks_amd/lib/spmvdrv sb lilac-benchmarks_amd/lib/libhip-spmv.so 170 ) > $O/sb170_attribution.log 2>&1; cat $O/sb170_attribution.log
BENCH_DIST_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_n2_launcher.json 2> $O/bench_n2_launcher.err; echo "launcher n2 rc=$?"
python3 -c "
import json;d=json.loads(open('$O/bench_n2_launcher.json').read().strip().splitlines()[-1]);print(d['n_gpus'],d['value'],d['npb_cg']['mops'],d['npb_cg']['verified'],d['npb_cg']['upload_s'],d['gather_inclusive']['value'])"
timeout -k 10 600 python3 bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_n2_inproc.json 2> $O/bench_n2_inproc.err; echo "inproc n2 rc=$?"
python3 -c "
import json;d=json.loads(open('$O/bench_n2_inproc.json').read().strip().splitlines()[-1]);print(d['n_gpus'],d['value'],d['npb_cg']['mops'],d['npb_cg']['verified'],d['npb_cg']['upload_s'],d['gather_inclusive']['value'])"
