"""Scratch profiling helper: per-phase HIP-event times of one resident batch."""
import argparse, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alignasm_amd as A

ap = argparse.ArgumentParser()
ap.add_argument("--contigs", type=int, default=500); ap.add_argument("--recs", type=int, default=1000)
ap.add_argument("--k", type=int, default=4); ap.add_argument("--dense", type=int, default=0)
ap.add_argument("--seed", type=int, default=21); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--heavy", type=int, default=0); ap.add_argument("--dup", type=int, default=0); ap.add_argument("--shuffle", type=int, default=0)
ap.add_argument("--input-order", type=int, default=0, help="1: K7's several-waves class in input order instead of largest node bound first")
ap.add_argument("--grid-order", type=int, default=0, help="1: blocks take their work items in grid order (default: XCD by XCD)")
ap.add_argument("--chain", default="auto", help="the chain class (aasm_k67_chain): auto / all / none")
ap.add_argument("--graph-launches", type=int, default=0, help="1: rows, reversed CSR and sweep headers by the separate launches (default: one workgroup per contig where the contigs are small)")
ap.add_argument("--chain-own-queue", type=int, default=0, help="1: the chain class's heap wave keeps its own BFS queue (default: the order comes from a wave of its own)")
ap.add_argument("--cs", type=int, default=0, help="1: the batch carries cs tags and the device derives the match ranges (K0)")
a = ap.parse_args()
t = time.time()
if a.cs:
    src = A.Paf.synth(a.contigs, a.recs, a.seed, dense=bool(a.dense), heavy_tail=bool(a.heavy))
    os.makedirs("/tmp/aasm_probe", exist_ok=True); src.save("/tmp/aasm_probe/p.paf"); src.close()
    hb = A.Paf.read("/tmp/aasm_probe/p.paf", device_ranges=True)
    v = hb.view(); print("gen+read %.2fs records=%d ranges=%d cs_bytes=%d" % (time.time() - t, v.n_records, v.n_ranges, os.path.getsize("/tmp/aasm_probe/p.paf")), flush=True)
else:
    paf = A.Paf.synth(a.contigs, a.recs, a.seed, dense=bool(a.dense), heavy_tail=bool(a.heavy), dup_every=a.dup, shuffle=bool(a.shuffle), no_cs=True); hb = paf.batch(); paf.close()
    print("gen %.2fs records=%d ranges=%d" % (time.time() - t, hb.view.n_records, hb.view.n_ranges), flush=True)
t = time.time(); db = A.DeviceBatch(hb); print("upload %.2fs" % (time.time() - t), flush=True)
for r in range(a.reps):
    t = time.time(); res = db.solve(max_paths=a.k, timing=True, heap_input_order=bool(a.input_order), grid_order=bool(a.grid_order), chain=a.chain, graph_launches=bool(a.graph_launches), chain_own_queue=bool(a.chain_own_queue)); wall = time.time() - t
    st = res.stats(); res.close()
    ph = {k: round(v, 3) for k, v in st["phase_ms"].items() if v > 0}
    print(json.dumps({"rep": r, "wall_ms": round(wall * 1e3, 2), "total_ms": round(st["total_ms"], 3), "phases": ph,
                      "V": st["n_vertices"], "E": st["n_edges"], "dev_MB": st["device_bytes"] >> 20}), flush=True)
