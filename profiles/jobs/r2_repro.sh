cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/repro
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reproducible or striped_geometries" 2>&1 | tail -5
timeout -k 10 600 python tools/soak_determinism.py > gpurun_out/repro/soak.log 2>&1; echo "soak rc=$?"; cat gpurun_out/repro/soak.log
