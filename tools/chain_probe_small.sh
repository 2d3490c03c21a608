#!/bin/bash
for spec in "1 1000" "250 1000" "625 1000" "1000 1000"; do
  set -- $spec
  for ch in none all; do
    echo -n "contigs $1 chain $ch: "
    python tools/phase_probe.py --contigs $1 --recs $2 --reps 5 --chain $ch | grep '^{' | python -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin][1:]
b = min(rows, key=lambda r: r['total_ms'])
print(b['total_ms'], {k: b['phases'][k] for k in ('sptree', 'heap_prep', 'heap', 'chain', 'fwd', 'select') if k in b['phases']})"
  done
done
for ch in none auto; do
  echo -n "heavy-tail 5000 chain $ch: "
  python tools/phase_probe.py --contigs 5000 --recs 1000 --reps 4 --heavy 1 --chain $ch | grep '^{' | python -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin][1:]
b = min(rows, key=lambda r: r['total_ms'])
print(b['total_ms'], {k: b['phases'][k] for k in ('sptree', 'heap_prep', 'heap', 'chain', 'fwd', 'select') if k in b['phases']})"
done
