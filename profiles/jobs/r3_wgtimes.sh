cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3c
for c in D/8 C; do
timeout -k 10 300 python3 tools/wgtimes.py --case $c --launches 40 --dump > gpurun_out/r3c/wgtimes_$(echo $c | tr -d /).log 2>&1; echo "rc=$?"
done
grep "^#" gpurun_out/r3c/*.log
