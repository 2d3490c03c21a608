"""Kernel timeline of the last solve in a rocprofv3 --kernel-trace csv: start / end (ms from the first kernel of that solve) per kernel."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last occurrence of aasm_k1_sort( marks the start of the last solve
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("aasm::aasm_k1_sort(")]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d >= 0.0:
        print("%8.3f %8.3f  %7.3f ms  q%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, d, r.get("Queue_Id", "?"), r["Kernel_Name"][:50]))
