"""N > 1 path on CPU: two processes (gloo) each hold one prefix-bucket shard of a suffix array, exchange
{first, last, count} with all_gather and stitch the boundary LCP exactly as bench.py does over RCCL.
The shards come from the oracle's output cut at a first-character boundary, which is how the device
pipeline splits (contiguous top-digit ranges)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sufr_amd
        from sufr_amd import shards, synth
        from oracle_helper import Oracle
        o = Oracle()
        x, _ = synth.syn_human(60_000, seed=3)
        raw = x.numpy()
        norm = o.normalize(raw, True)
        sa, lcp, _ = o.build(norm, is_dna=True, threads=1)
        # contiguous ranges of first characters: rank r takes [cuts[r], cuts[r+1])
        firsts = norm[sa.astype(np.int64)]
        edges = [0] + [int(np.searchsorted(firsts, ord(c), side="left")) for c in "CT"[:world - 1]] + [sa.size]
        if world == 3:
            edges = [0, int(np.searchsorted(firsts, ord("C"))), int(np.searchsorted(firsts, ord("T"))), sa.size]
        lo, hi = edges[rank], edges[rank + 1]
        my_sa = sa[lo:hi].copy(); my_lcp = lcp[lo:hi].copy()
        if rank > 0 and my_lcp.size:
            my_lcp[0] = 0xFFFFFFFF          # what a shard cannot know by itself
        bounds = shards.exchange_boundaries(int(my_sa[0]) if my_sa.size else 0, int(my_sa[-1]) if my_sa.size else 0,
                                            int(my_sa.size), torch.device("cpu"), dist)
        assert [b[2] for b in bounds] == [edges[i + 1] - edges[i] for i in range(world)]
        assert shards.output_offset(bounds, rank) == lo
        k = shards.stitched_first_lcp(bounds, rank, lambda st, ln: norm[st:st + ln], norm.size)
        if rank == 0:
            assert k is None
        else:
            assert k == int(lcp[lo]) == sufr_amd.lcp_pair(norm, int(sa[lo - 1]), int(sa[lo]))
            my_lcp[0] = k
        # gather the stitched shards on rank 0 and compare with the unsharded arrays
        gathered = [None] * world
        dist.all_gather_object(gathered, (my_sa, my_lcp))
        if rank == 0:
            assert np.array_equal(np.concatenate([g[0] for g in gathered]), sa)
            assert np.array_equal(np.concatenate([g[1] for g in gathered]), lcp)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_boundary_exchange_and_stitch_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results
