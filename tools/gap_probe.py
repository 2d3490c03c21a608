"""Diagnostic: from a rocprofv3 kernel trace of bench.py, the last step's span, the union of kernel busy time inside it and the
largest gaps (host waits between the pipeline's stages show up as idle GPU)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
# steps: a step starts with aasm_k1_sort (first kernel of the pipeline without K0)
starts = [i for i, e in enumerate(ev) if "aasm_k1_sortE" in e[2] or e[2].startswith("aasm::aasm_k1_sort(") or "aasm_k1_sort(" in e[2] and "rank" not in e[2] and "fix" not in e[2]]
starts = [i for i in starts if "rank" not in ev[i][2] and "fix" not in ev[i][2]]
if len(starts) < 3: print("steps not found", len(starts)); sys.exit(0)
a, b = starts[-2], starts[-1]
seg = ev[a:b]
t0, t1 = seg[0][0], max(e[1] for e in seg)
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
gaps = []
for s, e, n in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("step span %.3f ms, GPU busy (union) %.3f ms, idle %.3f ms in %d gaps, %d dispatches" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), len(seg)))
for g, n in sorted(gaps, reverse=True)[:12]: print("  gap %.1f us before %s" % (g / 1e3, n[:60]))
