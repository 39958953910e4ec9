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


def _worker(rank, world, port, q, out_path):
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
        # the form bench.py's timed N > 1 step uses: the triple is cut out of the shard's own array (no .item()) and
        # gathered into ONE [world, 3] tensor (on a GPU it stays in HBM and feeds sufr_hip_stitch_device_u32)
        dev_bounds = shards.gather_boundaries_device(torch.from_numpy(my_sa.astype(np.int64)), int(my_sa.size), dist)
        assert dev_bounds.shape == (world, 3) and dev_bounds.dtype == torch.int64
        assert [tuple(int(v) for v in row) for row in dev_bounds.tolist()] == list(bounds)
        i32 = torch.from_numpy(my_sa.astype(np.uint32).view(np.int32))      # what the u32 ABI fills: int32-typed
        assert torch.equal(shards.gather_boundaries_device(i32, int(my_sa.size), dist), dev_bounds)
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
        # ---- the multi-writer file (sufr_builder.rs:875-906 with one writer per shard): rank 0 lays the file out
        # through the C ABI (sufr_write_frame: header + name table, host code only), every rank writes its slice
        # at  sa_pos + 4 * (suffixes of the ranks before it);  the result must be the single-writer file
        import ctypes as C
        import struct
        from sufr_amd import _lib
        from sufr_amd.cli import create_args
        offset, total, has_prev, prev_last = shards.write_plan(bounds, rank)
        assert total == sa.size and offset == lo
        assert has_prev == (rank > 0 and my_sa.size > 0) and (not has_prev or prev_last == int(sa[lo - 1]))
        names = (C.c_char_p * 2)(b"chr_one", b"2")
        starts = (C.c_uint64 * 2)(0, 40_000)
        sd = _lib.SequenceData(norm.ctypes.data_as(C.POINTER(C.c_uint8)), norm.size, starts, names, 2)
        args = create_args("in.fa", out_path, is_dna=True, ignore_softmask=True)
        if rank == 0:
            err = C.create_string_buffer(512)
            assert _lib.lib().sufr_write_frame(out_path.encode(), C.byref(sd), C.byref(args), total, err, len(err)) == 0
        dist.barrier()
        fd = os.open(out_path, os.O_WRONLY)
        with open(out_path, "rb") as fh:
            head = fh.read(36)
        text_pos, sa_pos, lcp_pos = struct.unpack_from("<QQQ", head, 12)
        if rank == 0:
            os.pwrite(fd, norm.tobytes(), text_pos)
        os.pwrite(fd, my_sa.astype("<u4").tobytes(), sa_pos + 4 * offset)
        os.pwrite(fd, my_lcp.astype("<u4").tobytes(), lcp_pos + 4 * offset)
        os.close(fd)
        dist.barrier()
        if rank == 0:
            o.write_file(out_path + ".ref", is_dna=True, allow_ambiguity=False, ignore_softmask=True, norm_text=norm,
                         sa=sa, lcp=lcp, sequence_starts=(0, 40_000), sequence_names=("chr_one", "2"))
            assert open(out_path, "rb").read() == open(out_path + ".ref", "rb").read()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_boundary_exchange_and_stitch_gloo(world, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, str(tmp_path / "sharded.sufr"))) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results
