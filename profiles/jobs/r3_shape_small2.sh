cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3bf
mkdir -p $O
for c in "npb:128000x16 1:0:0,1:8:0" "npb:128000x64 1:0:0,1:8:0" "npb:512000x64 1:0:0,1:8:0" "C 1:0:0,1:5:0" "D/8 1:0:0,1:16:0" "B 1:0:0,1:5:0"; do set -- $c
timeout -k 10 500 python3 tools/shapebench.py --cases $1 --configs $2 --rounds 6 2>&1 | grep -v "^#" | cut -c1-150 | tee -a $O/shape_after.log
done
