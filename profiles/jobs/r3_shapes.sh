cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3b
timeout -k 10 800 python3 tools/shapebench.py --cases ${CASES:-D/8,D/4,C,B} --configs ${CONFIGS:-1:0:0,2:0:0,2:0:768,2:10:0} --rounds 8 > gpurun_out/r3b/shapebench.log 2>&1; echo "rc=$?" >> gpurun_out/r3b/shapebench.log
cat gpurun_out/r3b/shapebench.log
