# number of concurrent frame parts (SR_SPLIT = 1, 2, 4): single-GPU frame time and the per-rank cost of an N-way strip split
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in "$@"; do
  echo "SR_SPLIT=$m"
  SR_SPLIT=$m python scripts/gpu_strip_balance.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:(round(v['max_ms'],2), round(v['speedup_bound'],2)) for k,v in d.items()})"
done
