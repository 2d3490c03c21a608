"""N > 1 path on CPU: two gloo ranks each solve their contiguous shard of contigs (with the
host emulation of the kernels standing in for the GPU) and the concatenation must equal
the unsharded oracle result.  Also checks the barrier + MAX-over-ranks timing reduction that
bench.py performs, and the partition's invariants."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import aasm_testlib as T
    from alignasm_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hb = T.synth(9, 60, 17, heavy_tail=True, dup_every=5)     # every rank builds the same file
    cuts = shard.partition_contigs(hb.arrays["ctg_rec_off"], world)
    mine = hb.subset(list(range(cuts[rank], cuts[rank + 1])))
    part = T.emul_solve(mine, 64)
    dist.barrier()
    t = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, {k: part[k] for k in T.OUT_KEYS} | {"n_contigs": part["n_contigs"]})
    if rank == 0:
        whole = shard.concat_outputs(gathered)
        want = T.oracle_solve(hb, 64)
        q.put((T.diff_outputs(want, whole, stats=False), float(t.item()), cuts))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_contig_sharding_matches_unsharded():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    bad, tmax, cuts = q.get(timeout=180)
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert bad == [] and abs(tmax - 0.020) < 1e-12
    assert cuts[0] == 0 and cuts[-1] == 9 and all(a < b for a, b in zip(cuts, cuts[1:]))


def test_partition_invariants():
    from alignasm_amd import shard
    rng = np.random.default_rng(0)
    for C_, n in ((1, 4), (7, 8), (100, 8), (5000, 8), (33, 2)):
        sizes = rng.integers(1, 4000, C_)
        off = np.concatenate([[0], np.cumsum(sizes)])
        cuts = shard.partition_contigs(off, n)
        assert cuts[0] == 0 and cuts[-1] == C_ and all(a < b for a, b in zip(cuts, cuts[1:])) and len(cuts) - 1 == min(n, C_)
        if C_ >= 50 * n:
            loads = [shard.contig_costs(off)[a:b].sum() for a, b in zip(cuts, cuts[1:])]
            assert max(loads) < 1.25 * (sum(loads) / len(loads))
