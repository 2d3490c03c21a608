"""K7's several-waves class on dense batches, launched largest first, by waves per contig (the library's rule: 16 while
NMW * 16 <= 6144 wave slots, else 8 while NMW * 8 <= 6144, else 4)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alignasm_amd as A
shapes = [tuple(int(y) for y in x.split("x")) for x in (sys.argv[1] if len(sys.argv) > 1 else "1250x1000,700x1000,400x1500,5000x250,200x2000").split(",")]
for nc, n in shapes:
    paf = A.Paf.synth(nc, n, 31, dense=True, no_cs=True)
    db = A.DeviceBatch(paf)
    for waves in (0, 4, 8, 16):
        for _ in range(3):
            res = db.solve(max_paths=16, timing=True, heap_block_waves=waves)
            st = res.stats(); res.close()
        print(json.dumps({"contigs": nc, "records": n, "waves": waves or "rule", "heap_ms": round(st["phase_ms"]["heap"], 2), "total_ms": round(st["total_ms"], 2)}), flush=True)
    db.close(); paf.close()
