# does an equal number of row blocks per stripe group (250 = 10 x 25 workgroups, 6 CUs idle) beat 256 (25 / 26 per group)?
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/stamps
for w in 256 250 256 250; do
echo "== --swgs $w" >> gpurun_out/stamps/swgs.log
timeout -k 10 600 python tools/kbench.py --reps 60 --cases C --plans striped --tunings xcd1nt0 --swgs $w --stamps --stamps-launches 20 2>&1 | grep -v "^#   per-workgroup" >> gpurun_out/stamps/swgs.log
done
echo "== D/8: 256 vs 240 (12 per group)" >> gpurun_out/stamps/swgs.log
for w in 256 240 260; do
echo "== --swgs $w" >> gpurun_out/stamps/swgs.log
timeout -k 10 600 python tools/kbench.py --reps 40 --cases D/8 --plans striped --tunings xcd1nt0 --swgs $w 2>&1 >> gpurun_out/stamps/swgs.log
done
cat gpurun_out/stamps/swgs.log
