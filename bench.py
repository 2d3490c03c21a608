#!/usr/bin/env python3
"""bench.py -- contigs/sec (+ edges-relaxed/sec) of the per-contig path inference on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
through torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

A "step" is one pass of the whole hot path (K1..K9 + output compaction) over one batch of
synthetic contigs that is ALREADY RESIDENT IN HBM (uploaded before the timed region).

N = 1: BASELINE.json configs[2] "human whole-genome-scale PAF (~5M records, k=4 alt paths),
1 MI355X", concretised by SURVEY.md 8(d) as C3: 5 000 contigs x 1 000 records, sparse graph
mode, K = 4, seed 21.  (configs[1], 50 k records at K=1, is far too small to occupy the chip and
is a parity-test case instead.)

N > 1: BASELINE.json configs[3] = C4, "the same file as C3, contig-sharded across the GPUs
(static per-contig partition)": every rank builds the SAME C3 file (seed 21), cuts it with the
library's cost-balanced contiguous partition (aasm_partition_contigs, csrc/aasm_shard.cpp) and
solves only its own block.  Contigs are independent, so no collective runs on the data path
(only the barrier + MAX-reduce of the wall time).  Total work is fixed as N grows, hence
"scaling": "strong"; value = 5 000 contigs x steps / max-over-ranks time.  The weak-scaling
number (every rank solves its own C3-sized file, seed 21 + 1000*rank) is reported beside it
under "weak".  `--workload c5` (BASELINE configs[4]: 10 000 dense contigs, K=16) is ONE file too:
aasm_partition_contigs cuts it into 8 cost-balanced pieces (a piece is what fits one GPU's HBM with
its ~59 GB workspace), rank r of N generates and solves pieces [r*8/N, (r+1)*8/N) one after the
other, a step = every piece once; strong scaling, value = 10 000 contigs x steps / time.

Extra keys of the N = 1 line: `step_with_fetch_ms` (the step plus the D2H + ragged pack of the
three result lists); `k10000` (the same hot path at the reference's shipped MAX_PATH_COUNT = 10000,
paf_data.cpp:729, on the WHOLE batch, and on its first 1 000 contigs as in earlier rounds);
`c3_heavy_tail` (SURVEY 8(d)'s variant: contig sizes log-normal(ln 600, 1) clipped to [1, 8000] and
rescaled to the same 5 M records, seed 21) and `c3_dup3` (every third record duplicated on another
chromosome: equal (qry_str, qry_end) keys and tied path scores, the multi-mapping regime), each with
its phase times and its ratio to the uniform step; `e2e` (the `alignasm` command line on the C3 file
with cs tags: stage seconds and file bytes; skipped with --no-e2e).

roofline: the dominant kernel of the timed region (largest average HIP-event time on the
library's stream), its ALGORITHMIC bytes per launch (byte model: DESIGN.md "Byte model")
and the HBM peak of /opt/skills/guides/MI355X_MICROARCH.md (8 TB/s).  `traffic` (PMC
FETCH/WRITE bytes) is collected offline with rocprofv3 --pmc; see profiles/.
cpu_baseline: the oracle (oracle/liboracle.so, a CPU restatement: kind "port") timed on
this box's host cores over a bounded sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (contigs, recs, dense, K, seed, description)
    "c2": (50, 1000, False, 1, 11, "C2 single-chromosome synthetic PAF: 50 contigs x 1000 records, sparse, K=1, seed 11"),
    "c3": (5000, 1000, False, 4, 21, "C3 human WGS-scale synthetic PAF: 5000 contigs x 1000 records (5M records), sparse, K=4, seed 21"),
    # C5 (BASELINE configs[4]): ONE 10000-contig dense file.  aasm_partition_contigs cuts it into 8 pieces (what one GPU holds:
    # ~59 GB of workspace per piece); rank r of N solves pieces [r*8/N, (r+1)*8/N) one after the other - the same file at every N
    "c5": (10000, 1000, True, 16, 31, "C5 cancer-karyotype synthetic PAF: 10000 contigs x 1000 records (10M records), dense, K=16, seed 31"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def kernel_bytes(st):
    """Algorithmic bytes per launch for the five heavy kernels (DESIGN.md 'Byte model')."""
    V, E, H = st["n_vertices"], st["n_edges"], st["n_heap_nodes"]
    dense = E > 6 * V                                               # (the pipeline's own rule: wide SP trees get the several-waves heap kernel)
    return {
        # the chain class (aasm_k67_chain: sweep + pre-pass + heaps of a contig in one workgroup; small batches and the long tail): the
        # bytes of the two kernels it stands for.  On a batch where only the tail is in the class the figure is an upper bound
        "chain": ("aasm_k67_chain", 24 * E + 2 * 40 * V + 24 * E + 40 * V + 24 * H),
        "sptree": ("aasm_k6_rev_sweep", 24 * E + 2 * 40 * V),
        "fwd": ("aasm_k5_fwd_sweep", 24 * E + 2 * 8 * V),
        "heap": ("aasm_k7_heap_mw" if dense else "aasm_k7_heap", 24 * E + 40 * V + 24 * H),
        "enum": ("aasm_k8_enum", 64 * st["pq_pushes"] + 40 * st["n_paths_found"]),
        "select": ("aasm_k9_sel_convert", 24 * st["ispr_edges"] + 40 * st["ispr_vertices"] + 8 * st["path_edges"] + 40 * st["out_elems"]),
    }


def total_bytes(st, n_records):
    """B = 48 N + 24 R_touched + 4*24 E + 3*40 V + 24 H + 3*64 K_found."""
    return (48 * n_records + 24 * st["range_steps"] + 96 * st["n_edges"] + 120 * st["n_vertices"]
            + 24 * st["n_heap_nodes"] + 192 * st["n_paths_found"])


def merge_stats(acc, st):
    """Sum of the per-piece statistics of one step (a rank may hold its block of the file as several resident pieces)."""
    if acc is None:
        acc = {k: (dict(v) if isinstance(v, dict) else v) for k, v in st.items()}
        return acc
    for k, v in st.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                acc[k][kk] = acc[k].get(kk, 0.0) + vv
        elif k == "device_bytes":
            acc[k] = max(acc[k], v)
        elif isinstance(v, (int, float)):
            acc[k] = acc.get(k, 0) + v
    return acc


def timed_steps(dbs, K, steps, barrier):
    """steps x (every resident piece of this rank once); returns (elapsed seconds, summed phase times, stats of the last step)."""
    if not isinstance(dbs, (list, tuple)):
        dbs = [dbs]
    phase_acc, stats = {}, None
    barrier()
    t_begin = time.perf_counter()
    for _ in range(steps):
        stats = None
        for db in dbs:
            res = db.solve(max_paths=K, timing=True)      # enqueues the pipeline and syncs its stream
            st = res.stats()
            res.close()
            for k, v in st["phase_ms"].items():
                phase_acc[k] = phase_acc.get(k, 0.0) + v
            stats = merge_stats(stats, st)
    elapsed = time.perf_counter() - t_begin
    barrier()
    return elapsed, phase_acc, stats


def library_identity(A):
    """Which shared object this process measured: path + sha256 (the line did not say)."""
    import hashlib
    h = hashlib.sha256()
    with open(A.api.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return {"path": os.path.relpath(A.api.LIB_PATH, ROOT), "sha256": h.hexdigest(), "override": bool(os.environ.get("AASM_LIB_OVERRIDE"))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--contigs", type=int, default=0, help="override contigs (exploration only)")
    ap.add_argument("--recs", type=int, default=0)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip step_with_fetch / k10000 / weak (profiling runs)")
    ap.add_argument("--cpu-sample", type=int, default=5000, help="contigs of the workload timed on the CPU oracle")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end command-line run")
    ap.add_argument("--allow-override", action="store_true", help="accept AASM_LIB_OVERRIDE (a diagnostic build of the library; the line names what it loaded)")
    args = ap.parse_args()

    if os.environ.get("AASM_LIB_OVERRIDE") and not args.allow_override:
        raise SystemExit("bench.py: AASM_LIB_OVERRIDE is set (%s); the bench measures the in-tree library - unset it, or pass "
                         "--allow-override to measure that build on purpose" % os.environ["AASM_LIB_OVERRIDE"])

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (nothing has touched the GPU yet) and
        # hand their exit code on; rank 0's JSON line goes through the child's stdout unchanged.
        raise SystemExit(self_launch(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    dist = None
    torch = None
    # AASM_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share
    # devices round-robin, the timing reduction runs over gloo on the CPU); the driver's runs use nccl (= RCCL)
    backend = os.environ.get("AASM_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import alignasm_amd as A
    from alignasm_amd import shard
    from alignasm_amd._abi import HostBatch

    if A.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: alignasm_amd has no CPU fallback")
    if backend == "nccl" and A.device_count() < world:
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible; one rank per GPU is the contract "
                         "(AASM_BENCH_BACKEND=gloo rehearses the sharded path with ranks sharing devices)" % (world, A.device_count()))
    if backend != "nccl":
        local_rank = local_rank % A.device_count()
    nc, nr, dense, K, seed, desc = WORKLOADS[args.workload]
    custom = bool(args.contigs or args.recs or args.k)
    nc = args.contigs or nc
    nr = args.recs or nr
    K = args.k or K
    if custom:
        desc = f"CUSTOM {nc} contigs x {nr} records, {'dense' if dense else 'sparse'}, K={K}, seed {seed} (not the BASELINE workload)"
    pieces_file = args.workload == "c5"                    # C5: one file in 8 (or a multiple of N) pieces, a rank solves its pieces in turn
    strong = world > 1 or pieces_file                      # C4 / C5: one file, sharded
    if world > 1 and not pieces_file and not custom:
        desc = ("C4 = the C3 file (5000 contigs x 1000 records, sparse, K=4, seed 21) contig-sharded over %d GPUs, "
                "static cost-balanced contiguous partition" % world)

    def barrier():
        if dist is not None:
            dist.barrier()

    def allreduce(x, op):
        if dist is None:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return float(t.item())

    # ---- synthetic batch of this rank, uploaded before the timed region
    t0 = time.time()
    pafs = []                                              # library handles to release at the end
    if pieces_file:
        n_pieces = 8 if 8 % world == 0 else world * -(-8 // world)
        n_pieces = min(n_pieces, nc)
        whole = A.Paf.synth(nc, nr, seed, dense=dense, records_only=True)    # every rank cuts the SAME file (records only: what the cost model reads)
        cuts = shard.partition_contigs(whole, n_pieces)
        whole.close()
        per = n_pieces // world
        mine = []
        for pc in range(rank * per, (rank + 1) * per):
            pp = A.Paf.synth(nc, nr, seed, dense=dense, no_cs=True, first=cuts[pc], count=cuts[pc + 1] - cuts[pc])   # = those contigs of the whole file
            pafs.append(pp)
            mine.append(pp)
        paf = mine[0]
        my_contigs = sum(cuts[pc + 1] - cuts[pc] for pc in range(rank * per, (rank + 1) * per))
        n_records = sum(int(m.view().n_records) for m in mine)
        contigs_total = nc
        desc += "; cut into %d pieces by aasm_partition_contigs, %d piece(s) per rank solved one after the other" % (n_pieces, per)
    elif world > 1:
        paf = A.Paf.synth(nc, nr, seed, dense=dense, no_cs=True)            # the SAME file on every rank
        cuts = shard.partition_contigs(paf, world)
        mine = [HostBatch.from_view_range(paf.view(), cuts[rank], cuts[rank + 1])]
        paf.close()
        paf = None
        my_contigs, n_records = mine[0].n_contigs, int(mine[0].view.n_records)
        contigs_total = nc
    else:
        paf = A.Paf.synth(nc, nr, seed, dense=dense, no_cs=True)
        pafs.append(paf)
        mine = [paf]
        my_contigs, n_records = nc, int(paf.view().n_records)
        contigs_total = nc
        cuts = None
    gen_s = time.time() - t0
    t0 = time.time()
    dbs = [A.DeviceBatch(m, device=local_rank) for m in mine]
    db = dbs[0]
    upload_s = time.time() - t0

    for _ in range(args.warmup):
        for d_ in dbs:
            d_.solve(max_paths=K, timing=True).close()
    elapsed, phase_acc, stats = timed_steps(dbs, K, args.steps, barrier)
    elapsed = allreduce(elapsed, dist.ReduceOp.MAX if dist else None)
    total_edges = allreduce(stats["n_edges"], dist.ReduceOp.SUM if dist else None)
    max_contigs = allreduce(my_contigs, dist.ReduceOp.MAX if dist else None)

    extras = {}
    if not args.no_extras:
        # the step plus the result fetch (D2H of the three ragged lists + host-side pack)
        barrier()
        t = time.perf_counter()
        for _ in range(args.steps):
            for d_ in dbs:
                res = d_.solve(max_paths=K)
                A.api.free_out(res.fetch_raw())                      # aasm_result_fetch: D2H + ragged pack, as a C caller receives it
                res.close()
        extras["step_with_fetch_ms"] = round(allreduce((time.perf_counter() - t) * 1e3 / max(args.steps, 1), dist.ReduceOp.MAX if dist else None), 3)
        if world == 1 and len(dbs) == 1:
            # the reference's shipped MAX_PATH_COUNT (paf_data.cpp:729): the whole batch, and a 1000-contig slice of it
            def at_k10000(batch_dev, n):
                batch_dev.solve(max_paths=10000).close()
                t = time.perf_counter()
                for _ in range(2):
                    r = batch_dev.solve(max_paths=10000, timing=True)
                    stk = r.stats()
                    r.close()
                msk = (time.perf_counter() - t) * 1e3 / 2
                return {"ms_per_step": round(msk, 3), "contigs": n, "contigs_per_sec": round(n / msk * 1e3, 1), "paths_found": stk["n_paths_found"],
                        "pq_pushes": stk["pq_pushes"], "enum_ms": round(stk["phase_ms"].get("enum", 0.0), 3), "select_ms": round(stk["phase_ms"].get("select", 0.0), 3)}
            full = at_k10000(db, my_contigs)
            nk = min(1000, my_contigs)
            hbk = HostBatch.from_view_range(mine[0].view() if not isinstance(mine[0], HostBatch) else mine[0].view, 0, nk)
            dbk = A.DeviceBatch(hbk, device=local_rank)
            part = at_k10000(dbk, nk)
            dbk.close()
            extras["k10000"] = dict(full, max_paths=10000, first_1000=part,
                                    note="the same batch at the reference's shipped MAX_PATH_COUNT; first_1000 = its first %d contigs alone" % nk)
            if args.workload == "c3" and not custom:
                # the shapes real PAFs have (alignasm.cpp:346-361 takes contigs of any size): same record count, same K
                uniform_ms = elapsed * 1e3 / max(args.steps, 1)
                for key, kw, what in (("c3_heavy_tail", {"heavy_tail": True}, "contig sizes log-normal(ln 600, 1) clipped to [1, 8000], rescaled to 5M records"),
                                      ("c3_dup3", {"dup_every": 3}, "every third record duplicated on another chromosome (equal sort keys, tied scores)"),
                                      ("c3_shuffled", {"shuffle": True}, "the SAME C3 file with the records of every contig in random order (the generator writes contigs in query order, "
                                       "which K1's sorted-input shortcut detects: aligner output has no such order - `phase_ms.sort` here is the sort phase that does not depend on it)")):
                    pv = A.Paf.synth(nc, nr, seed, dense=dense, no_cs=True, **kw)
                    sizes = pv.batch().arrays["ctg_rec_off"]
                    longest = int((sizes[1:] - sizes[:-1]).max())
                    dbv = A.DeviceBatch(pv, device=local_rank)
                    dbv.solve(max_paths=K).close()
                    ev, pacc, stv = timed_steps(dbv, K, args.steps, barrier)
                    msv = ev * 1e3 / max(args.steps, 1)
                    extras[key] = {"ms_per_step": round(msv, 3), "contigs_per_sec": round(nc / msv * 1e3, 1), "ratio_to_uniform": round(msv / uniform_ms, 3),
                                   "records": int(sizes[-1]), "longest_contig": longest, "V": stv["n_vertices"], "E": stv["n_edges"], "H": stv["n_heap_nodes"],
                                   "paths_converted": stv["n_paths_converted"], "phase_ms": {k: round(v / max(args.steps, 1), 3) for k, v in pacc.items() if v > 0},
                                   "shape": what}
                    dbv.close()
                    pv.close()
            if not args.no_e2e and not custom:
                extras["e2e"] = e2e_cli(nc, nr, seed, dense, K)
        if world > 1 and not pieces_file:
            # weak scaling beside it: every rank solves its own C3-sized file
            db.close()
            pw = A.Paf.synth(nc, nr, seed + 1000 * rank, dense=dense, no_cs=True)
            dbw = A.DeviceBatch(pw, device=local_rank)
            dbw.solve(max_paths=K).close()
            ew, _, _ = timed_steps(dbw, K, args.steps, barrier)
            ew = allreduce(ew, dist.ReduceOp.MAX)
            extras["weak"] = {"value": round(nc * world * args.steps / ew, 2), "unit": "contigs/s", "ms_per_step": round(ew * 1e3 / max(args.steps, 1), 3),
                              "contigs_per_gpu": nc, "note": "every rank solves its own C3-sized file (seed 21 + 1000*rank)"}
            dbw.close()
            pw.close()
            dbs = []

    if rank == 0:
        steps = max(args.steps, 1)
        ms_per_step = elapsed * 1e3 / steps
        value = contigs_total * steps / elapsed
        edges_relaxed = 2.0 * total_edges * steps / elapsed            # 2*E per contig
        avg = {k: v / steps for k, v in phase_acc.items()}
        kb = kernel_bytes(stats)
        dom = max(kb, key=lambda k: avg.get(k, 0.0))
        dom_name, dom_bytes = kb[dom]
        dom_ms = avg.get(dom, 0.0)
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        Btot = total_bytes(stats, n_records)
        pipe_ms = stats["total_ms"]
        out = {
            "metric": "contigs_per_sec", "value": round(value, 2), "unit": "contigs/s",
            "edges_relaxed_per_sec": round(edges_relaxed, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": desc, "contigs_total": contigs_total, "contigs_on_rank0": my_contigs, "contigs_on_fullest_rank": int(max_contigs),
                       "records_per_contig": nr, "max_paths": K, "graph_mode": "dense" if dense else "sparse",
                       "parallelism": f"contig-shard x{world}", "records_on_rank0": n_records,
                       "partition": ("aasm_partition_contigs cuts %s" % cuts) if cuts else "none"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel_ms": round(dom_ms, 3), "kernel_bytes": int(dom_bytes),
                         "pipeline_bytes": int(Btot), "pipeline_ms": round(pipe_ms, 3),
                         "pipeline_achieved": round(Btot / (pipe_ms * 1e-3) / 1e9, 2) if pipe_ms > 0 else 0.0,
                         "scope": "rank 0" if world > 1 else "the batch"},
            "phase_ms": {k: round(v, 3) for k, v in avg.items() if v > 0},
            "graph": {"V": stats["n_vertices"], "E": stats["n_edges"], "P": stats["n_pairs"], "H": stats["n_heap_nodes"],
                      "paths_found": stats["n_paths_found"], "paths_converted": stats["n_paths_converted"],
                      "device_MB": stats["device_bytes"] >> 20},
            "setup": {"gen_s": round(gen_s, 2), "upload_s": round(upload_s, 2)},
            "library": library_identity(A),
        }
        out.update(extras)
        pmc = pmc_traffic(args.workload, custom or world > 1, dom_name)
        if pmc:
            out["roofline"].update(pmc)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(paf, min(nc, paf.n_contigs), K, args.cpu_sample)
        print(json.dumps(out), flush=True)
    for d_ in dbs:
        d_.close()
    for p_ in pafs:
        p_.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n):
    """One rank per GPU through torch.distributed.run, as the driver launches N > 1 itself; returns the child's exit code."""
    import subprocess
    # --standalone: torchrun picks a free rendezvous port itself (binding port 0, closing it and handing the number on left a window
    # for another process to take it); --local-addr: the container's hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("bench.py: --gpus %d without a launcher, starting: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def pmc_traffic(workload, not_profiled_shape, kernel):
    """HBM traffic of the dominant kernel from the committed OFFLINE rocprofv3 --pmc passes
    (profiles/rNN_<workload>_pmc_fetch_write.json, newest round: one pass per counter, --kernel-trace only).
    FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads half
    the bytes of a wide (16 B/lane) coalesced stream and is uncalibrated for other widths; these
    kernels gather 4-48 B per access, so the raw value is reported and the x2 figure is given
    as the upper bound.  Only attached when the run IS the profiled workload."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_%s_pmc_fetch_write.json" % workload)))
    if not_profiled_shape or not files:
        return None
    path = files[-1]
    data = json.load(open(path))
    for name, v in data.items():
        if (name.startswith("aasm::" + kernel + "(") or name.startswith(kernel + "(") or name == kernel) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            f, w_ = v["FETCH_SIZE"]["mean_per_launch"] * 1024, v["WRITE_SIZE"]["mean_per_launch"] * 1024
            return {"traffic": int(f + w_), "traffic_fetch_raw": int(f), "traffic_write": int(w_),
                    "traffic_upper_fetch_x2": int(2 * f + w_),
                    "traffic_source": "offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/" + os.path.basename(path)}
    return None


def cpu_baseline(paf, nc, K, sample):
    """Oracle (CPU restatement, kind 'port') on the first `sample` contigs, all host cores; beside it the same port on 1 thread
    and on a ladder of thread counts (bounded samples), so the all-cores figure can be read: `value_1t`, `scaling`,
    `stops_scaling_at` = the smallest thread count that reaches 90 % of the best rate of the ladder."""
    import ctypes as C
    import aasm_testlib as T
    from alignasm_amd._abi import BatchOut, HostBatch, Opts
    cores = max(1, os.cpu_count() or 1)
    lib = T.oracle()
    o = Opts(int(K), 0, 0, 0, 0)

    def run(n, threads):
        hb = HostBatch.from_view_range(paf.view(), 0, n)
        out = BatchOut()
        t0 = time.perf_counter()
        rc = lib.oracle_solve_batch(C.byref(hb.view), C.byref(o), threads, C.byref(out))
        dt = time.perf_counter() - t0
        lib.oracle_free_out(C.byref(out))
        assert rc == 0
        return n / dt, dt

    n = min(sample, nc)
    usable = usable_cpus()
    r1, dt1 = run(min(40, nc), 1)
    ladder = []
    for t in sorted({2, 4, 8, 16, 32, 64, 128, cores} - {1}):
        if t > cores:
            continue
        rt, _ = run(min(nc, max(40, 25 * t)), t)
        ladder.append((t, rt))
    ladder_best = max(ladder, key=lambda x: x[1])[0] if ladder else 1
    full = []                       # the whole sample at: every hardware thread, the CPUs the process may use, twice that, the ladder's best count
    for t in sorted({cores, usable, min(cores, 2 * usable), ladder_best}):
        rt, dtt = run(n, t)
        full.append((t, rt, dtt))
    threads, rate, dt = max(full, key=lambda x: x[1])
    best = max([rate] + [r for _, r in ladder])
    stops = next((t for t, r in ladder if r >= 0.9 * best), cores)
    ref = reference_prefix(paf, nc, usable)
    return {"reference_k1_k8": ref, "value": round(rate, 2), "unit": "contigs/s", "cores": threads, "kind": "port",
            "sample": f"first {n} contigs of the same workload, K={K}, one contig per task over {threads} threads, {dt:.2f} s wall "
                      f"(the best of {[t for t, _, _ in full]} threads on this sample; the host shows {cores} hardware threads, the process may use {usable} CPUs)",
            "full_sample": [{"threads": t, "contigs_per_sec": round(r, 1)} for t, r, _ in full],
            "value_1t": round(r1, 2), "sample_1t": f"first {min(40, nc)} contigs on 1 thread, {dt1:.2f} s",
            "scaling": [{"threads": t, "contigs_per_sec": round(r, 1)} for t, r in ladder], "stops_scaling_at": stops}


def reference_prefix(paf, nc, usable):
    """The REAL reference's own K1 ... K8 (oracle/_ref/libaasm_ref_prefix.so = paf_data.cpp:223-738 compiled in the build container; it
    travels with the snapshot) beside the port stopped at the same line, same contigs, same threads, on THIS box's cores: how the
    `port` figure above reads in reference time.  The reference's MAX_PATH_COUNT is fixed at 10 000, so the port runs that K here too.
    None when the library is not there."""
    import ctypes as C
    import aasm_testlib as T
    from alignasm_amd._abi import HostBatch
    R = T.ref_prefix(mono=False)
    if R is None:
        return None
    O = T.oracle()
    O.oracle_time_prefix.restype = C.c_double
    n = min(96, nc)
    hb = HostBatch.from_view_range(paf.view(), 0, n)
    out = {"sample": f"first {n} contigs of the workload, K1-K8 only, one contig per task",
           "note": "timed at K = 10000 on BOTH sides: MAX_PATH_COUNT is a constant of the reference (paf_data.cpp:729), so its K1-K8 cannot run at the "
                   "headline's K; the `port` figures of this block (`value`, `value_1t`, `scaling`) are at the headline's K",
           "max_paths": 10000, "rows": []}
    for th in sorted({1, min(8, usable), usable}):
        m = n if th > 1 else min(n, 24)
        tr = R.refp_time_batch(C.byref(hb.view), C.c_int64(0), C.c_int64(m), th, 0)
        tp = O.oracle_time_prefix(C.byref(hb.view), C.c_int64(0), C.c_int64(m), th, 0, C.c_int64(10000))
        if not (tr > 0 and tp > 0):                                  # (an error return of either timer: no row, the rest of cpu_baseline stands)
            out["rows"].append({"threads": th, "contigs": m, "error": "timer returned reference %r, port %r" % (tr, tp)})
            continue
        out["rows"].append({"threads": th, "contigs": m, "reference_ms_per_contig": round(1e3 * tr / m, 3), "port_ms_per_contig": round(1e3 * tp / m, 3),
                            "reference_over_port": round(tr / tp, 2)})
    return out


def usable_cpus():
    """CPUs this process may really use: hardware threads cut to the affinity mask and the cgroup CPU quota."""
    n = max(1, os.cpu_count() or 1)
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                    # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
        if q != "max" and int(per) > 0:
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def e2e_cli(nc, nr, seed, dense, K):
    """`alignasm <file.paf> --max-paths K` on the workload's file WITH cs tags (solve-only AND end to end): the
    process's own stage clock (--timing) of the second of two runs (input in the page cache), file sizes, wall time."""
    import shutil
    import subprocess
    import tempfile
    import alignasm_amd as A
    d = tempfile.mkdtemp(prefix="aasm_e2e_")
    try:
        path = os.path.join(d, "synth.paf")
        t0 = time.perf_counter()
        paf = A.Paf.synth(nc, nr, seed, dense=dense)
        paf.save(path)
        paf.close()
        gen_s = time.perf_counter() - t0
        exe = os.path.join(ROOT, "alignasm_amd", "alignasm")
        runs = []
        for _ in range(2):
            for sfx in (".aln.paf", ".aln.alt.paf", ".aln.all.paf"):     # to NEW files both times (replacing a 3 GB file frees its page cache inside rename())
                if os.path.exists(path[:-4] + sfx):
                    os.remove(path[:-4] + sfx)
            t0 = time.perf_counter()
            r = subprocess.run([exe, path, "--max-paths", str(K), "--timing"], capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": "alignasm exited with %d: %s" % (r.returncode, r.stderr[-300:])}
            line = [ln for ln in r.stderr.splitlines() if ln.startswith("alignasm timing:")]
            stage = {}
            if line:
                tok = line[-1].replace("(", " ").replace(")", " ").split()
                for i, t in enumerate(tok):
                    if t in ("read_s", "solve_s", "upload", "device", "fetch", "write_s", "total_s", "records", "contigs", "overlap_s"):
                        try:
                            stage[t] = float(tok[i + 1])
                        except (ValueError, IndexError):
                            pass
            runs.append({"wall_s": round(wall, 3), "stages": stage})
        out_bytes = sum(os.path.getsize(path[:-4] + sfx) for sfx in (".aln.paf", ".aln.alt.paf", ".aln.all.paf"))
        best = min(runs, key=lambda x: x["stages"].get("total_s", x["wall_s"]))
        return {"total_s": best["stages"].get("total_s"), "process_wall_s": best["wall_s"], "stages": best["stages"], "runs": runs,
                "input_bytes": os.path.getsize(path), "output_bytes": out_bytes, "gen_s": round(gen_s, 2),
                "cmd": "alignasm synth.paf --max-paths %d --timing" % K,
                "note": "total_s = inside the process, file open to last byte written (outputs into the page cache); process_wall_s adds exec + HIP start-up + exit"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
