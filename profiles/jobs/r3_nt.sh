cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3y
mkdir -p $O
timeout -k 10 400 python3 tools/kbench.py --reps 30 --cases sb170,rmat20,D/8 --plans stream --tunings xcd1nt0,xcd1nt1 > $O/kbench_stream_nt.log 2>&1; cat $O/kbench_stream_nt.log
timeout -k 10 400 python3 tools/shapebench.py --cases D,C,D/4 --configs 1:0:0 --rounds 6 > $O/shapebench_nt.log 2>&1; cat $O/shapebench_nt.log
