cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3bf
mkdir -p $O
timeout -k 10 500 python3 tools/shapebench.py --cases npb:128000x16 --configs 1:0:0,1:8:0,1:4:0,1:2:0,1:1:0 --rounds 6 2>&1 | tee $O/shape_128000x16.log
timeout -k 10 500 python3 tools/shapebench.py --cases npb:128000x64 --configs 1:0:0,1:8:0,1:4:0,1:2:0,1:1:0 --rounds 6 2>&1 | tee $O/shape_128000x64.log
timeout -k 10 500 python3 tools/shapebench.py --cases npb:512000x64 --configs 1:0:0,1:8:0,1:4:0,1:2:0,1:16:0 --rounds 6 2>&1 | tee $O/shape_512000x64.log
