cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3blk
mkdir -p $O
for rep in 1 2; do
for u in "4 4" "2 2" "2 3" "1 2" "2 1" "1 1" "3 3" "3 2"; do set -- $u
  echo "U1=$1 U2=$2: $(BLK_U1=$1 BLK_U2=$2 timeout -k 10 120 python3 tools/blockedbench.py 170 | cut -d';' -f2 | cut -d'(' -f1)" | tee -a $O/u_sweep2.log
done; done
