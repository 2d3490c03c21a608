"""A/B of library builds on ONE box: python tools/ab.py [--rounds 2] [--args "..."] variants/a.so variants/b.so[:ENV=VAL,...] ...
Each build runs `tools/phase_probe.py <args> --reps 6` per round, the builds interleaved; prints the per-phase MINIMUM over the
repetitions of all rounds (HIP-event phase times) and the best step."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--args", default="--contigs 5000 --recs 1000 --k 4")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
best = {}
for r in range(a.rounds):
    for lib in a.libs:
        path, _, kv = lib.partition(":")                              # "variants/x.so:KEY=VAL,KEY2=VAL2": environment of that variant
        env = dict(os.environ, AASM_LIB_OVERRIDE=os.path.join(ROOT, path))
        for item in filter(None, kv.split(",")):
            k_, _, v_ = item.partition("=")
            env[k_] = v_
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "phase_probe.py")] + a.args.split() + ["--reps", "6"], env=env, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            print(lib, "FAILED", p.stderr[-500:]); continue
        for ln in p.stdout.splitlines():
            if not ln.startswith("{"):
                continue
            d = json.loads(ln)
            if d["rep"] == 0:
                continue
            b = best.setdefault(lib, {"total": 1e9})
            b["total"] = min(b["total"], d["total_ms"])
            for k, v in d["phases"].items():
                b[k] = min(b.get(k, 1e9), v)
for lib in a.libs:
    if lib in best:
        b = best[lib]
        print("%-28s step %.3f  " % (lib, b["total"]) + " ".join("%s %.3f" % (k, b[k]) for k in ("sptree", "fwd", "heap_prep", "heap", "select", "final", "revcsr", "edges", "pairs", "sort", "topo", "enum", "gather", "misc") if k in b), flush=True)
