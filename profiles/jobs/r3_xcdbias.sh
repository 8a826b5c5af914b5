cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3x
mkdir -p $O
timeout -k 10 700 python3 tools/shapebench.py --cases C,D/8,B --configs 1:0:0:0,1:0:0:10,1:0:0:20,1:0:0:30,1:0:0:-20 --rounds 10 > $O/xcdbias_$(hostname)_$$.log 2>&1; cat $O/xcdbias_*.log | tail -20
