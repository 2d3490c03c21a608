# A/B of two library builds on the C3 batch: bash tools/ab_lib.sh variants/a.so variants/b.so (paths relative to the repo root)
cd /root/repo
for v in $1 $2 $1 $2; do
  echo "== $v"
  AASM_LIB_OVERRIDE=/root/repo/$v timeout -k 10 100 python tools/phase_probe.py --contigs 5000 --recs 1000 --k 4 --reps 5 2>&1 | tail -3 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['total_ms'], 'heap_prep', d['phases'].get('heap_prep'), 'heap', d['phases'].get('heap'))"
done
