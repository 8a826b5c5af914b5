cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3bin
mkdir -p $O
for cfg in "170 2048 50 0 1" "170 2048 50 0 0" "170 2048 50 1 1" "170 2048 50 2 1" "170 1024 50 1 1" "170 1024 50 2 1"; do
  echo "== $cfg" | tee -a $O/binbench2.log
  timeout -k 10 200 ./tools/bin/binbench $cfg 2>&1 | tail -1 | tee -a $O/binbench2.log || exit 1
done
