"""End-to-end timing of the `alignasm <input.paf>` command line on a synthetic PAF with cs tags:
file read + cs parse, solve on the GPU, three output files.  """
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alignasm_amd as A

ap = argparse.ArgumentParser()
ap.add_argument("--contigs", type=int, default=500); ap.add_argument("--recs", type=int, default=1000)
ap.add_argument("--k", type=int, default=4); ap.add_argument("--seed", type=int, default=21)
ap.add_argument("--dir", default="/tmp/aasm_e2e"); ap.add_argument("--threads", default="0", help="host threads (-t); a comma list runs the command once per value")
ap.add_argument("--quiet", type=int, default=0, help="1: only the timing line of each run")
a = ap.parse_args()
os.makedirs(a.dir, exist_ok=True)
path = os.path.join(a.dir, "synth.paf")
t = time.time()
paf = A.Paf.synth(a.contigs, a.recs, a.seed)
paf.save(path)
print(f"generated {path}: {os.path.getsize(path) / 2**20:.0f} MiB in {time.time() - t:.1f}s", flush=True)
paf.close()
outs = [path[:-4] + s for s in (".aln.paf", ".aln.alt.paf", ".aln.all.paf")]
for thr in [int(x) for x in str(a.threads).split(",")]:
    cmd = [os.path.join(ROOT, "alignasm_amd", "alignasm"), path, "--max-paths", str(a.k), "--timing"]
    if thr:
        cmd += ["-t", str(thr)]
    for rep in range(2):                   # second run: file in the page cache, GPU context warm-up is per process
        for o in outs:                     # to NEW files (replacing a 3 GB file frees its page cache inside rename())
            if os.path.exists(o): os.remove(o)
        t = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=(dict(os.environ) if a.quiet else dict(os.environ, AASM_IO_TIMING="1")))
        lines = r.stderr.strip().splitlines()
        if a.quiet: lines = [l for l in lines if l.startswith("alignasm timing")]
        print(f"threads {thr} run {rep}: rc={r.returncode} wall {time.time() - t:.2f}s | {' / '.join(lines)}", flush=True)
sizes = {s: os.path.getsize(path[:-4] + s) for s in (".aln.paf", ".aln.alt.paf", ".aln.all.paf")}
print("outputs MiB:", {k: round(v / 2**20, 1) for k, v in sizes.items()})
