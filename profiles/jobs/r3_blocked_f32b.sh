cd "${GRAFT_REPO_ROOT:-.}"
for m in 0 1 2; do
echo "f32 mode $m: $(BLK_MODE=$m timeout -k 10 120 python3 tools/blockedbench.py 160 --f32 | cut -d';' -f2)"
echo "f64 mode $m: $(BLK_MODE=$m timeout -k 10 120 python3 tools/blockedbench.py 160 | cut -d';' -f2)"
done
