# round 4: what would keeping the hub columns of x in LDS leave of k_stream's time on R-MAT graphs?  (timing-only ablation)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r4b
mkdir -p $OUT
timeout -k 10 600 python3 tools/streamabl.py --cases rmat20,rmat22 --rounds 4 --masks 0,2,1032,1288,1544,1800 > $OUT/streamabl_hub.log 2>&1; echo "rc=$?"; cat $OUT/streamabl_hub.log
