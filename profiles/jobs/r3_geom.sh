cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3d
timeout -k 10 800 python3 tools/shapebench.py --cases D/8,D/4,D/2,C,B --configs 1:0:0,1:20:0,1:10:0 --rounds 8 > gpurun_out/r3d/shapebench.log 2>&1; echo "rc=$?" >> gpurun_out/r3d/shapebench.log
cat gpurun_out/r3d/shapebench.log
timeout -k 10 300 python3 tools/wgtimes.py --case D/8 --launches 40 > gpurun_out/r3d/wgtimes_D8.log 2>&1; echo "rc=$?"
cat gpurun_out/r3d/wgtimes_D8.log
