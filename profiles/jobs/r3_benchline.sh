cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3m
mkdir -p $O
S=$(date +%s.%N)
timeout -k 10 600 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"
E=$(date +%s.%N); echo "bench wall seconds: $(echo "$E - $S" | bc)" | tee $O/bench_wall.txt
tail -c 600 $O/bench_line.json
