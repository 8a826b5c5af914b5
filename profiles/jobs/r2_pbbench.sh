# feasibility of the column-BLOCKED product (the pbbench prototype (removed in round 4)) on SparseBench's shape
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/pb
hipcc --offload-arch=gfx950 -O3 the pbbench prototype (removed in round 4) -o /tmp/pbbench
for args in "170 16 512 8 4 1" "170 16 512 4 4 1" "170 16 512 2 4 1"; do
  echo "== $args" | tee -a gpurun_out/pb/pbbench6.log
  timeout -k 10 200 /tmp/pbbench $args 2>&1 | tee -a gpurun_out/pb/pbbench6.log
done
