"""K7 on ONE giant dense contig by waves per contig (the library gives a lone contig 16): how far from chain-bound it is."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alignasm_amd as A
for n in (5000, 20000):
    paf = A.Paf.synth(1, n, 77, dense=True, no_cs=True)
    db = A.DeviceBatch(paf)
    for waves in (16, 8, 4):
        for _ in range(2):
            res = db.solve(max_paths=4, timing=True, heap_block_waves=waves)
            st = res.stats(); res.close()
        print(n, "records, waves", waves, "heap_ms", round(st["phase_ms"]["heap"], 2), flush=True)
    db.close(); paf.close()
