cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
echo "pad4: $(timeout -k 10 120 python3 tools/blockedbench.py 170 130 | cut -d';' -f2)"
echo "pad8: $(BLK_PAD8=1 timeout -k 10 120 python3 tools/blockedbench.py 170 130 | cut -d';' -f2)"
done
