# A/B of two library builds (e.g. copies of alignasm_amd/libalignasm_amd.so before and after a change, kept under variants/):
#   bash tools/ab_lib.sh variants/a.so variants/b.so [phase_probe arguments, default: the C3 batch]
# Each build runs twice, interleaved; per repetition: step ms and the K7 phases.
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
ARGS=${*:-"--contigs 5000 --recs 1000 --k 4"}
for v in $A $B $A $B; do
  echo "== $v"
  AASM_LIB_OVERRIDE=$PWD/$v timeout -k 10 200 python tools/phase_probe.py $ARGS --reps 5 2>&1 | tail -3 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['total_ms'], 'heap_prep', d['phases'].get('heap_prep'), 'heap', d['phases'].get('heap'), 'select', d['phases'].get('select'))"
done
