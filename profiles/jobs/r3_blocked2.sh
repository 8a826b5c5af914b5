cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3i
mkdir -p $O
timeout -k 10 300 python3 tools/blockedbench.py > $O/blockedbench.log 2>&1; cat $O/blockedbench.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bb -- python3 tools/blockedbench.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r3i/prof/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_blk' in r['Name'] or 'k_stream' in r['Name'] or 'k_bk' in r['Name']:
            print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'])
PY
rm -rf $O/prof/*.db
