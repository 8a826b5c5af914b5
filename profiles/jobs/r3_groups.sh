cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3v
mkdir -p $O
timeout -k 5 900 python3 tools/shapebench.py --cases C,B,D/8,D/4,D/2 --configs 1:0:0 --rounds 6 --allocs 3 > $O/shapebench_new_model.log 2>&1; grep -E "medians of|^#" $O/shapebench_new_model.log; grep "workgroups=" $O/shapebench_new_model.log | awk '{print $1, $2, $3, $4, $7, $8}' | sort -u
