cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "no_room or striped_geometries or tiled" 2>&1 | tail -15
