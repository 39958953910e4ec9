#!/usr/bin/env python3
"""bench.py -- suffixes sorted per second (SA + LCP) on the MI355X construction path.

    python bench.py --gpus N --steps K --warmup W [--workload human|elegans|ecoli] [--bases B]

One *step* = one full pass of the hot path over one genome-sized batch that is already resident in
HBM: text normalisation + byte histogram, radix partition on packed k-char prefix keys, the remaining
MSD levels, the in-LDS leaf sort and every re-keying level, ending with the complete SA and LCP
arrays in HBM (libsufr_hip.so, sufr_hip_sort_device_u32).  Default workload: the configuration the
BASELINE.json metric is quoted on, a GRCh38-sized genome (3.1 Gb, --dna --ignore-softmask, 256
partitions); the real assembly is not available offline, so a seeded synthetic stand-in of the same size
and repeat/masking structure is generated directly in HBM (sufr_amd/synth.py).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  The genome is sharded by
prefix-bucket range: every rank holds the text, derives the same splitters from the on-device k-mer
histogram and builds only its bucket range; the only exchange is an all_gather of
{first suffix, last suffix, count} per rank for the boundary-LCP stitch (sufr_builder.rs:893-902).
Total work is fixed as N grows => "scaling": "strong".  value = suffixes of the whole genome / time.
`python bench.py --gpus N` outside torch.distributed.run starts its N ranks itself (a child `python -m
torch.distributed.run --nproc-per-node N bench.py ...`; the parent never touches the GPU).  On a one-GPU box the N > 1 leg
runs as `--gpus 2 --backend gloo --share-device` (two ranks, one device, host-side exchange) and, with
SUFR_BENCH_FORCE_DIST=1 at N = 1, through RCCL itself: nccl is initialised at world size 1, the rank builds TWO first-digit
shards one after the other and every step runs the all_gather on device tensors + sufr_hip_stitch_device_u32 on the
context's stream exactly as a rank of an N-GPU job does.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch
import torch.distributed as dist

import sufr_amd
from sufr_amd import synth

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (generator, default bases, sufr create flags, partitions, label)
    "human": (synth.syn_human, 3_100_000_000, dict(is_dna=True, ignore_softmask=True), 256,
              "syn_human 3.1 Gb stand-in for GRCh38, --dna --ignore-softmask -n 256 (BASELINE configs[3])"),
    "elegans": (synth.syn_elegans, 100_286_401, dict(is_dna=True), 64,
                "syn_elegans 100 Mb stand-in for C. elegans, --dna -n 64 (BASELINE configs[2])"),
    "ecoli": (synth.syn_ecoli, 4_641_652, dict(is_dna=True), 16,
              "syn_ecoli 4.6 Mb stand-in for E. coli K-12, --dna (BASELINE configs[1])"),
}


def pmc_traffic_bytes(workload: str):
    """HBM bytes per launch of the radix-partition kernel from the committed rocprofv3 --pmc summary of
    this same command (profiles/r*_pmc_<workload>.csv; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for 16-B/lane streaming reads on gfx950, WRITE_SIZE as read).  None when no profile exists."""
    import csv
    import glob
    import hashlib
    files = sorted(glob.glob(str(ROOT / "profiles" / f"r*_pmc_{workload}.csv")))
    if not files:
        return None, None
    # the counters belong to ONE version of the kernel: the summary's first line names the sha256 of sufr_part.inc it was
    # collected from (profiles/summarize_pmc.py); a different source means a stale figure -> null and a loud line
    lines = open(files[-1]).read().splitlines()
    have = hashlib.sha256((ROOT / "sufr_amd" / "csrc" / "sufr_part.inc").read_bytes()).hexdigest()
    tagged = lines[0].split("sha256=")[-1].strip() if lines and lines[0].startswith("#") else None
    if tagged != have:
        print(f"bench.py: roofline.traffic is null: {Path(files[-1]).name} was collected from "
              f"{'an untagged' if tagged is None else 'another'} version of sufr_part.inc (re-run profiles/pmc.sh)", file=sys.stderr)
        return None, None
    fetch = write = None
    for r in csv.DictReader(ln for ln in lines if not ln.startswith("#")):
        if "k_msd_part_text" in r["kernel"] or "k_scatter_text" in r["kernel"]:
            if r["counter"] == "FETCH_SIZE":
                fetch = float(r["largest_dispatch_value"])
            if r["counter"] == "WRITE_SIZE":
                write = float(r["largest_dispatch_value"])
    if fetch is None or write is None:
        return None, None
    return (2.0 * fetch + write) * 1024.0, "profiles/" + Path(files[-1]).name


def host_memory_gb() -> float:
    return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2**30


def cpu_baseline(text_cpu: np.ndarray, flags: dict, partitions: int, target_s: float = 15.0, full_text=None):
    """The oracle (C restatement of the reference algorithm, kind = "port") timed on this box's host
    cores: a bounded prefix of the workload swept over thread counts and -- when the host can hold it (`full_text`
    is a callable that returns the whole text; the caller checks memory and cores) -- ONE run on the whole workload at
    the sweep's best thread count, which then is `value` (the 6 % prefix flattered the CPU)."""
    sys.path.insert(0, str(ROOT / "tests"))
    from oracle_helper import Oracle
    try:
        o = Oracle(native=True)     # rebuilt here with -march=native
    except Exception:
        o = Oracle()
    cores = os.cpu_count() or 1

    def run(nb, threads=None):
        sample = np.concatenate([text_cpu[:nb], np.frombuffer(b"$", dtype=np.uint8)])
        norm = o.normalize(sample, flags.get("ignore_softmask", False))
        t0 = time.perf_counter()
        _, _, st = o.build(norm, is_dna=flags.get("is_dna", False), num_partitions=partitions, threads=threads or cores)
        return st.num_suffixes, time.perf_counter() - t0, st

    # one bounded sample, every thread count on the SAME sample, the best reported (the port used to
    # get slower with more threads -- per-thread bucket vectors and single-thread copies, now parallel in the oracle --
    # so a figure at "all cores" timed its threading, not the algorithm)
    nb = min(text_cpu.size, 4_000_000)
    s, dt, st = run(nb, min(cores, 16))
    want = int(min(text_cpu.size, max(nb, nb * (target_s / 4.0 / max(dt, 1e-3)) * 0.8), 200_000_000))
    if want > nb * 2:
        nb = want
    sweep = {}
    best = None
    for t in sorted({t for t in (16, 32, 64, 128, cores) if t <= cores} or {cores}):
        s, dt, st = run(nb, t)
        sweep[str(t)] = round(s / max(dt, 1e-9), 1)
        if best is None or s / dt > best[0]:
            best = (s / dt, t, s, dt, st)
    rate, bt, s, dt, st = best
    sweep_note = (f"first {nb} bases of the same text (+'$'), {s} suffixes, {partitions} partitions; best of the sweep: "
                  f"{bt} threads, {dt:.2f} s wall (partition {st.t_partition:.2f} s, sort {st.t_sort:.2f} s)")
    out = {"value": rate, "unit": "suffixes/s", "cores": bt, "best_threads": bt, "host_cores": cores, "kind": "port",
           "threads_sweep": sweep, "sample": sweep_note, "sample_only": True,
           "caveat": (f"a C port of the reference algorithm (oracle/), not the reference: its OpenMP partition + merge-sort loops peak at "
                      f"{bt} of this host's {cores} hardware threads and get slower beyond (threads_sweep); the reference's Rust / rayon "
                      "loops may use the cores better.  A stated baseline, not a target: no claim rests on the GPU / CPU ratio")}
    if full_text is not None:
        try:
            whole = full_text()
            norm = o.normalize(whole, flags.get("ignore_softmask", False))
            del whole
            t0 = time.perf_counter()
            _, _, fst = o.build(norm, is_dna=flags.get("is_dna", False), num_partitions=partitions, threads=bt)
            fdt = time.perf_counter() - t0
            out.update({"value": fst.num_suffixes / fdt, "sample_only": False, "sample_sweep": {"value": rate, "what": sweep_note},
                        "sample": f"the WHOLE workload: {norm.size} bytes, {fst.num_suffixes} suffixes, {partitions} partitions, "
                                  f"{bt} threads (the sweep's best on the prefix), {fdt:.1f} s wall (partition "
                                  f"{fst.t_partition:.1f} s, sort {fst.t_sort:.1f} s)"})
        except Exception as e:                   # the headline line must not depend on it
            out["full_run_error"] = repr(e)[:200]
    else:
        print("bench.py: cpu_baseline is timed on a PREFIX of the workload only (sample_only: true): the whole-workload run "
              "needs >= 200 GB of host memory and >= 32 cores, or was switched off", file=sys.stderr)
    return out


def e2e_create(text_cpu: np.ndarray, starts, flags: dict, partitions: int, s_total: int, want_hash: bool = False):
    """`sufr create` end to end (t_create): the synthetic text written as a FASTA file, then
    the native CLI from FASTA parse to the closed .sufr file.  Reported next to `value`, never as `value`."""
    import shutil
    import subprocess
    import tempfile
    need = int(text_cpu.size * 1.03) + 8 * s_total + text_cpu.size + (64 << 20)
    tmp = Path(tempfile.mkdtemp(prefix="sufr_e2e_", dir=os.environ.get("TMPDIR", "/tmp")))
    try:
        if shutil.disk_usage(tmp).free < need * 1.2:
            return {"skipped": f"needs {need >> 20} MiB of scratch space in {tmp.parent}"}
        fa, out = tmp / "in.fa", tmp / "out.sufr"
        write_fasta(fa, text_cpu, starts)
        cmd = [str(sufr_amd.CLI_PATH), "--log", "debug", "create", "-n", str(partitions), "-o", str(out), str(fa)]
        for flag, opt in (("is_dna", "--dna"), ("ignore_softmask", "--ignore-softmask"), ("allow_ambiguity", "-a")):
            if flags.get(flag):
                cmd.insert(cmd.index("-n"), opt)
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": r.stderr.strip()[-200:]}
        phases = [ln for ln in r.stdout.splitlines() if "host phases" in ln]
        ph = phases[-1].split("host phases: ")[-1] if phases else None
        out_d = {"seconds": dt, "suffixes_per_s": s_total / dt, "fasta_bytes": fa.stat().st_size,
                 "sufr_bytes": out.stat().st_size, "phases": ph,
                 "what": "native `sufr create`: FASTA parse, H2D, build, D2H, .sufr written (process start-up included)"}
        if want_hash:
            out_d["sufr_sha256"] = file_sha256(out)
        if ph:
            # the four phases of the wall time: what the process reports since main() + what the caller's clock sees around it
            import re
            m = re.search(r"start-up \+ read ([0-9.]+)s.*H2D \+ build ([0-9.]+)s, D2H \+ write ([0-9.]+)s, since main\(\) ([0-9.]+)s", ph)
            if m:
                ready, build, write, inside = (float(x) for x in m.groups())
                out_d["phases_s"] = {"start_up_and_read": round(ready, 3), "h2d_and_build": round(build, 3),
                                     "d2h_and_write": round(write, 3),
                                     "process_start_and_exit": round(dt - ready - build - write, 3),
                                     "sum": round(dt, 3), "inside_main": round(inside, 3)}
        return out_d
    except Exception as e:      # the headline number must not depend on scratch space or a subprocess
        return {"error": repr(e)[:200]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def host_abi(text_cpu: np.ndarray, flags: dict, partitions: int, reps: int = 3):
    """The host-buffer ABI (sufr_hip_build_u32: the call INTEGRATION.md section 3 puts inside SufrBuilder::new) on the same text:
    pageable host text in, normalised text + SA + LCP out into freshly allocated pageable arrays (their first-touch faults are
    part of the call).  PCIe-inclusive: reported next to `value`, never as `value`."""
    import ctypes as C
    from sufr_amd import _lib
    try:
        n = text_cpu.size
        ctx = sufr_amd.Context(0)
        L = _lib.lib()
        fl = _lib.FLAG_RAW_TEXT | (_lib.FLAG_DNA if flags.get("is_dna") else 0) | \
            (_lib.FLAG_IGNORE_SOFTMASK if flags.get("ignore_softmask") else 0) | (_lib.FLAG_ALLOW_AMBIGUITY if flags.get("allow_ambiguity") else 0)
        calls = []
        for _ in range(reps):
            norm = np.empty(n, dtype=np.uint8); sa = np.empty(n, dtype=np.uint32); lcp = np.empty(n, dtype=np.uint32)
            ns = C.c_uint64(0); st = _lib.Stats()
            t0 = time.perf_counter()
            rc = L.sufr_hip_build_u32(ctx.handle, text_cpu.ctypes.data, n, fl, 0, None, partitions, 42, norm.ctypes.data,
                                      sa.ctypes.data, lcp.ctypes.data, n, C.byref(ns), C.byref(st))
            dt = time.perf_counter() - t0
            ctx.check(rc)
            calls.append({"seconds": round(dt, 4), "h2d_s": round(st.host_read_s, 4), "build_s": round(st.host_build_s, 4),
                          "d2h_s": round(st.host_write_s, 4)})
            s = ns.value
            del norm, sa, lcp
        ctx.close()
        best = min(calls, key=lambda c: c["seconds"])
        gb = (2 * n + 8 * s) / 1e9
        return {"entry_point": "sufr_hip_build_u32", "seconds": best["seconds"], "suffixes_per_s": s / best["seconds"],
                "pcie_gb": round(gb, 2), "gb_per_s": round(gb / best["seconds"], 1), "calls": calls,
                "what": "pageable host buffers both ways: H2D of the text, device build, D2H of the normalised text + SA + LCP "
                        "(first call: pinned staging buffers and the workspace are allocated)"}
    except Exception as e:      # the headline number must not depend on host memory
        return {"error": repr(e)[:200]}


def store_floor(s_total: int, text_len: int, bins: int):
    """What the memory side takes of the partition kernel's STORE PATTERN alone (profiles/micro/scatter_write.hip --floor): the same
    number of 12-byte records in runs of (records per 65 536-position tile / first digits) over the same number of first digits,
    cursors claimed with returning atomics, from a resident grid of one workgroup per CU -- no text, no ranking.  The kernel cannot
    be faster than this with this record format and tile size; measured live, outside the timed region."""
    import subprocess
    exe = Path(__file__).resolve().parent / "profiles" / "micro" / "scatter_write"
    if not exe.exists() or bins <= 0:
        return None
    run = max(1, round(s_total / max(1, text_len) * 65536 / bins))
    try:
        r = subprocess.run([str(exe), "--floor", str(s_total), str(bins), str(run)], capture_output=True, text=True, timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if r.returncode == 0 and lines else None
    except Exception:
        return None


def write_fasta(fa: Path, text_cpu: np.ndarray, starts):
    body = text_cpu[:-1]
    cuts = list(starts) + [body.size + 1]
    with open(fa, "wb") as f:
        for i in range(len(starts)):
            seq = body[cuts[i]:cuts[i + 1] - 1]
            f.write(f">seq{i + 1} synthetic\n".encode())
            full = (seq.size // 60) * 60
            if full:
                f.write(np.concatenate([seq[:full].reshape(-1, 60),
                                        np.full((full // 60, 1), 10, dtype=np.uint8)], axis=1).tobytes())
            if seq.size > full:
                f.write(seq[full:].tobytes() + b"\n")


def e2e_create_sharded(text, starts, flags: dict, partitions: int, rank: int, world: int, local_rank: int, dev, backend,
                       want_hash: bool = False):
    """`sufr create` with one process per GPU (N > 1): rank 0 writes the FASTA file, every rank parses it, builds its
    first-digit range on its own GPU and streams its SA / LCP slice into its range of the ONE output file
    (sufr_amd.shards.create_sharded: sufr_hip_shard_build, all_gather of 24 bytes per rank, sufr_write_frame,
    sufr_hip_shard_write).  Contexts are up already: unlike the N = 1 figure no process start-up is included."""
    import ctypes as C
    import shutil
    import tempfile
    from sufr_amd import _lib, shards
    from sufr_amd.cli import create_args
    box = [None]
    if rank == 0:
        box[0] = tempfile.mkdtemp(prefix="sufr_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    dist.broadcast_object_list(box, src=0)
    tmp = Path(box[0])
    fa, out = tmp / "in.fa", tmp / "out.sufr"
    try:
        if rank == 0:
            write_fasta(fa, text.cpu().numpy(), starts)
        dist.barrier()
        ctx = _lib.Context(local_rank)
        args = create_args(str(fa), str(out), num_partitions=partitions, **flags)
        cdev = dev if backend == "nccl" else "cpu"
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        sd = _lib.SequenceData()
        err = C.create_string_buffer(512)
        rc = _lib.lib().sufr_read_sequence_file(os.fsencode(str(fa)), ord("%"), C.byref(sd), err, len(err))
        if rc != 0:
            raise RuntimeError(err.value.decode())
        t1 = time.perf_counter()
        bounds, st = shards.create_sharded(ctx, sd, args, str(out), rank, world, dist, cdev)
        dt = time.perf_counter() - t0
        _lib.lib().sufr_sequence_data_free(C.byref(sd))
        ctx.close()
        t = torch.tensor([dt, t1 - t0], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        s_total = sum(b[2] for b in bounds)
        dist.barrier()               # every rank's slice is in the file
        return {"seconds": float(t[0]), "read_seconds": float(t[1]), "suffixes_per_s": s_total / float(t[0]),
                "sufr_bytes": out.stat().st_size if rank == 0 else None, "shard_suffixes": [b[2] for b in bounds],
                "sufr_sha256": file_sha256(out) if want_hash and rank == 0 else None,
                "what": f"{world} ranks, one GPU each: FASTA parse (every rank), H2D, shard build, 24-byte all_gather, "
                        "D2H + every rank's slice written into the one .sufr file (contexts already up).  WRITE-BOUND: the "
                        "file is 8 bytes per suffix + the text and the page cache of one host takes ~11-15 GB/s whoever "
                        "writes; the builds are < 5 % of it, so this figure does not scale with the number of GPUs"}
    except Exception as e:
        return {"error": repr(e)[:300]}
    finally:
        dist.barrier()
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)


def search_rate(builder, norm, sa, dev, is_dna: bool, num_queries: int = 4_000_000, query_len: int = 32):
    """Query side, never part of `value`: the suffix array of the last timed step is wrapped in
    place and searched for substrings of the text (every tenth with one changed symbol)."""
    try:
        from sufr_amd import DeviceIndex
        n = norm.numel()
        g = torch.Generator(device=dev); g.manual_seed(1)
        at = torch.randint(0, max(1, n - query_len - 1), (num_queries,), generator=g, device=dev)
        qb = norm[(at[:, None] + torch.arange(query_len, device=dev)[None, :]).reshape(-1)].contiguous()
        flip = torch.arange(0, num_queries, 10, device=dev) * query_len
        qb[flip] = norm[torch.randint(0, n, (flip.numel(),), generator=g, device=dev)]
        off = (torch.arange(num_queries + 1, device=dev, dtype=torch.int64) * query_len).contiguous()
        ix = DeviceIndex.wrap(builder.ctx, norm, sa, is_dna=is_dna)
        ms = []
        for _ in range(3):
            t0 = time.perf_counter()
            lo, hi = ix.search_device(qb, off)
            ms.append((time.perf_counter() - t0) * 1e3)
        found = float((hi > lo).float().mean())
        # the answer against the text: both ends of every sampled range start with the query
        pick = torch.randint(0, num_queries, (100_000,), generator=g, device=dev)
        pick = pick[hi[pick] > lo[pick]]
        q = qb.view(num_queries, query_len)[pick]
        for ranks in (lo[pick], hi[pick] - 1):
            pos = (sa[ranks].to(torch.int64) & 0xFFFFFFFF)[:, None] + torch.arange(query_len, device=dev)[None, :]
            assert bool(((pos < n) & (norm[pos.clamp(max=n - 1)] == q)).all()), "device search: range end does not match"
        ix.close()
        med = sorted(ms)[1]
        return {"kernel": "k_search_batch", "queries": num_queries, "query_len": query_len, "ms": med,
                "queries_per_s": num_queries / (med * 1e-3), "found": found, "checked_ranges": int(pick.numel())}
    except Exception as e:                       # the headline line must not depend on this
        return {"error": f"{type(e).__name__}: {e}"}


def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` as a child process and return its exit code.  Called before any HIP / torch.cuda call (counting
    devices does not initialise the GPU on this image); the child's ranks do the GPU work."""
    import subprocess
    if args.share_device and args.backend != "gloo":
        # N RCCL ranks on one device die with a duplicate-GPU error: say so here instead (advisor r5)
        print("bench.py: --share-device puts every rank on cuda:0 and needs the host-side exchange: add  --backend gloo", file=sys.stderr)
        return 2
    # (torch.cuda.device_count() reads the device list without initialising the GPU on this image -- unlike
    # torch.cuda.is_available() or any HIP call --, and it honours the visibility masks a KFD topology walk would not)
    have = torch.cuda.device_count()
    if not args.share_device and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this box shows {have} GPU(s).  One rank per GPU over RCCL needs {args.gpus}; to "
              f"rehearse the N > 1 leg on one GPU use  --gpus {args.gpus} --backend gloo --share-device  (ranks share cuda:0, "
              "host-side exchange)  or  SUFR_BENCH_FORCE_DIST=1 --gpus 1  (RCCL at world size 1, two shards per step)",
              file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def file_sha256(path) -> str:
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                return h.hexdigest()
            h.update(b)


def main():
    if os.environ.get("SUFR_BENCH_STACKS_AFTER"):       # debugging aid: every rank dumps its Python stacks after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SUFR_BENCH_STACKS_AFTER"]), repeat=False, file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SUFR_BENCH_WORKLOAD", "human"), choices=sorted(WORKLOADS))
    ap.add_argument("--bases", type=int, default=int(os.environ.get("SUFR_BENCH_BASES", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end `sufr create` measurement (N=1 only)")
    ap.add_argument("--placement-trials", type=int, default=int(os.environ.get("SUFR_BENCH_PLACEMENT_TRIALS", "3")),
                    help="contexts (work-buffer placements) tried before timing; the fastest is kept (1 = off)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the check of the timed run's arrays (N = 1, outside the timed region: permutation of "
                         "the suffix starts, order and exact unbounded LCP on 1.1e6 sampled ranks)")
    ap.add_argument("--no-search", action="store_true",
                    help="skip the query-side figure (batched device search on the arrays of the last step; outside the timed region)")
    ap.add_argument("--backend", default=os.environ.get("SUFR_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend (nccl = RCCL; gloo only for smoke-testing N>1 on one GPU)")
    ap.add_argument("--share-device", action="store_true",
                    help="smoke test: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--force-dist", action="store_true", default=bool(int(os.environ.get("SUFR_BENCH_FORCE_DIST", "0") or 0)),
                    help="N = 1 only: initialise nccl (RCCL) at world size 1, build TWO first-digit shards per step and run the "
                         "N > 1 exchange (all_gather on device tensors + sufr_hip_stitch_device_u32) on this one GPU")
    ap.add_argument("--e2e-hash", action="store_true", help="add the sha256 of the .sufr file that e2e_create wrote to its record")
    ap.add_argument("--full-cpu-baseline", choices=("auto", "on", "off"), default=os.environ.get("SUFR_BENCH_FULL_CPU", "auto"),
                    help="time the CPU port on the WHOLE workload (auto: when the host has >= 200 GB and >= 32 cores)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: start the N ranks as a CHILD job (decided before anything touches the GPU; the
        # parent only counts devices, waits and passes the child's output and exit code on -- it never execs)
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.force_dist and world != 1:
        raise SystemExit("--force-dist / SUFR_BENCH_FORCE_DIST is the one-GPU rehearsal of the N > 1 step: use it with --gpus 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    elif args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    vshards = 2 if args.force_dist else 1          # first-digit shards this rank builds per step

    gen, default_bases, flags, partitions, label = WORKLOADS[args.workload]
    bases = args.bases or default_bases
    data = "synthetic"
    fasta = os.environ.get("SUFR_BENCH_FASTA", "")
    if fasta and os.path.exists(fasta):
        # a real assembly on the box ("use them only if present on the GPU box (path via env var)"):
        # read by the package's own reader (util.rs:51-89), built with the workload's flags
        seqs = sufr_amd.read_sequence_file(fasta)
        text = torch.from_numpy(np.frombuffer(seqs.seq, dtype=np.uint8).copy()).to(dev)
        starts = list(seqs.start_positions)
        label = f"{os.path.basename(fasta)} ({len(starts)} sequences) with the flags of: {label}"
        data = "real"
        bases = text.numel() - 1
    elif args.share_device and world > 1:
        # N generators time-slicing ONE device crawl (eight of them at 3.1 Gb were still in their first kernel launches after
        # 70 s; one alone takes ~5 s): the ranks generate one after the other
        text = starts = None
        for r in range(world):
            if r == rank:
                text, starts = gen(bases, device=dev)
                torch.cuda.synchronize()
            dist.barrier()
    else:
        text, starts = gen(bases, device=dev)   # identical on every rank (same seed, same device type)
    torch.cuda.synchronize()
    n = text.numel()
    t_prog = time.perf_counter()

    def progress(what: str):
        """one line per phase and rank on stderr when several ranks run (a rank that dies or starves is then visible in the log)"""
        if world > 1:
            print(f"[bench rank {rank}/{world}] {what} at {time.perf_counter() - t_prog:.1f} s", file=sys.stderr, flush=True)

    progress(f"text ready (n = {n})")
    builder = sufr_amd.DeviceBuilder(local_rank)
    placement_ms = []
    out_sa = [None] * vshards
    out_lcp = [None] * vshards
    stats_acc = []
    from sufr_amd import shards
    soft = flags.get("ignore_softmask", False)
    totals = {"s_total": 0}
    num_shards = world * vshards

    def step():
        """One pass of the hot path on this rank: its first-digit shard(s) built, boundaries exchanged, first LCP stitched."""
        outs, counts = [], []
        for v in range(vshards):
            sa, lcp = builder.sort(text, raw_text=True, shard_index=rank * vshards + v, num_shards=num_shards,
                                   out_sa=out_sa[v], out_lcp=out_lcp[v], num_partitions=partitions, **flags)
            outs.append((sa, lcp)); counts.append(builder.num_suffixes)
            if vshards > 1:
                stats_acc.append(builder.stats.as_dict())
        s_local = counts[0]
        sa, lcp = outs[0]
        if num_shards == 1:
            totals["s_total"] = s_local
        elif args.backend == "nccl" or args.force_dist:
            # the only exchange of the path: {first, last, count} per shard (24 bytes, assembled on the device and
            # gathered into device memory over RCCL), then the boundary-LCP stitch as a kernel on this rank's text
            rows = [shards.gather_boundaries_device(o[0], c, dist, always_collective=True) for o, c in zip(outs, counts)]
            bounds = torch.stack(rows, dim=1).reshape(num_shards, 3).contiguous()     # row = global shard index
            for v in range(vshards):
                shards.stitch_device(builder.ctx, n, bounds, rank * vshards + v, outs[v][1])
            totals["bounds"] = bounds               # (read once, after the timed region)
        else:
            # gloo smoke test (CPU tensors): the host form of the same exchange
            first = int(sa[0].item()) & 0xFFFFFFFF if s_local else 0
            last = int(sa[s_local - 1].item()) & 0xFFFFFFFF if s_local else 0
            bl = shards.exchange_boundaries(first, last, s_local, "cpu", dist)
            k = shards.stitched_first_lcp(
                bl, rank, lambda st, ln: sufr_amd.normalize(text[st:st + ln].cpu().numpy(), soft), n)
            if k is not None:
                lcp[0] = k
            totals["bounds"] = torch.tensor(bl, dtype=torch.int64)
        if vshards > 1:
            totals["outs"] = outs
        return sa, lcp

    # size the output arrays once.  One shard: the first call allocates n entries.  Shards: eight ranks that each take 8 n bytes
    # for the sizing call (24.8 GB at 3.1 Gb, beside text, generator scratch and workspace) do not fit ONE 288 GB device -- the
    # --share-device rehearsal of round 6 sat in the allocator for 50 minutes --: a shard's arrays are tried at 1.4 x its even
    # share first and at n only if the build reports that it needs more (SUFR_HIP_E_CAPACITY).
    if num_shards > 1:
        guess = int(n / num_shards * 1.4) + (1 << 20)
        if guess < n:
            out_sa = [torch.empty(guess, dtype=torch.int32, device=dev) for _ in range(vshards)]
            out_lcp = [torch.empty(guess, dtype=torch.int32, device=dev) for _ in range(vshards)]
    try:
        sa, lcp = step()
    except Exception as e:          # noqa: BLE001 -- a shard above the guess: the full-size arrays after all
        if num_shards == 1 or "capacity" not in str(e).lower():
            raise
        out_sa = [None] * vshards; out_lcp = [None] * vshards
        torch.cuda.empty_cache()
        sa, lcp = step()
    progress("first build done")
    if vshards == 1:
        cap = int(builder.num_suffixes * 1.02) + 1024
        out_sa = [torch.empty(cap, dtype=torch.int32, device=dev)]
        out_lcp = [torch.empty(cap, dtype=torch.int32, device=dev)]
    else:
        caps = [int(o[0].numel() * 1.02) + 1024 for o in totals["outs"]]
        totals.pop("outs")
        out_sa = [torch.empty(c, dtype=torch.int32, device=dev) for c in caps]
        out_lcp = [torch.empty(c, dtype=torch.int32, device=dev) for c in caps]
        stats_acc.clear()
    del sa, lcp
    torch.cuda.empty_cache()
    # Workspace placement: the device time of one build depends on where hipMalloc puts the work buffers
    # (profiles/README.md: a plain 12 GB copy runs at 4.4-5.2 TB/s depending on the allocation).  The bench
    # creates a few contexts, builds on each (placement_ms) and times the one with the MEDIAN build time: `value`
    # is what a user's single context typically gets; value_best is the same count over the fastest placement.
    if args.share_device and world > 1:
        args.placement_trials = 1       # (N ranks on ONE device: three contexts per rank are 24 workspaces of ~10 GB at N = 8)
    if args.placement_trials > 1:
        cands = [builder]
        for _ in range(args.placement_trials - 1):
            cands.append(sufr_amd.DeviceBuilder(local_rank))
        for b in cands:
            builder = b
            step()                      # allocates this context's workspace
            step()
            placement_ms.append(round(float(b.stats.ms_total), 2))
        order = sorted(range(len(cands)), key=lambda i: placement_ms[i])
        keep = order[len(order) // 2]           # the MEDIAN placement is the one that gets timed
        for i, b in enumerate(cands):
            if i != keep:
                b.close()
        builder = cands[keep]
        torch.cuda.empty_cache()
    for _ in range(max(0, args.warmup)):
        step()
    progress("warm-up done")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    stats_acc.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sa, lcp = step()
        if vshards == 1:
            stats_acc.append(builder.stats.as_dict())
    barrier()
    dt = time.perf_counter() - t0
    dt = allmax(dt)

    if num_shards > 1:
        totals["s_total"] = int(totals["bounds"][:, 2].sum().item())      # one read-back, outside the timed region
    s_total = totals["s_total"]
    verified = None
    search = None
    forced = None
    if args.force_dist:
        # outside the timed region: the two stitched shards of the last timed step, concatenated, against ONE unsharded
        # build of the same text on the same context -- whole arrays, every element
        outs = totals.pop("outs")
        cat_sa = torch.cat([o[0] for o in outs]); cat_lcp = torch.cat([o[1] for o in outs])
        one_sa, one_lcp = builder.sort(text, raw_text=True, num_partitions=partitions, **flags)
        forced = {"backend": "nccl (RCCL), world size 1", "virtual_shards": vshards,
                  "shard_suffixes": [int(o[0].numel()) for o in outs],
                  "equal_to_single_build": bool(torch.equal(cat_sa, one_sa) and torch.equal(cat_lcp, one_lcp)),
                  "stitched_first_lcp": int(outs[1][1][0].item()) if outs[1][1].numel() else None,
                  "what": "every timed step = 2 shard builds + all_gather_into_tensor of the {first, last, count} triples on "
                          "device tensors (RCCL) + sufr_hip_stitch_device_u32 on the context's stream; ms_per_step is the "
                          "time of BOTH shards on one GPU, not an N = 2 figure"}
        assert forced["equal_to_single_build"], "forced-dist: stitched shards differ from the single build"
        del cat_sa, cat_lcp, one_sa, one_lcp
    if num_shards == 1 and not args.no_verify:
        # outside the timed region: the arrays of the last timed step against the text (sufr_amd/verify.py)
        sys.path.insert(0, str(ROOT / "tests"))
        import gpu_verify as verify          # the GPU-side property checker: test infrastructure, like oracle/
        verify.check_permutation(text, sa, is_dna=flags.get("is_dna", False), allow_ambiguity=False,
                                 ignore_softmask=soft)
        lut = verify.normalize_lut(dev, soft)
        norm = torch.empty_like(text)
        for lo in range(0, n, 1 << 28):
            norm[lo:lo + (1 << 28)] = lut[text[lo:lo + (1 << 28)].long()]
        verified = verify.check_sampled_ranks(norm, sa, lcp, samples=1_000_000, deep_samples=100_000)
        verified["what"] = ("SA = permutation of the suffix starts (count, sum, weighted sum, xor of hashes); order and "
                            "exact unbounded LCP on sampled adjacent ranks, deep_ranks of them with LCP >= 64")
        if not args.no_search:
            search = search_rate(builder, norm, sa, dev, bool(flags.get("is_dna", False)))
        del norm

    per_rank = None
    if world > 1:
        keys = ["ms_total", "ms_normalize", "ms_hist_text", "ms_partition", "ms_passes", "ms_finish", "ms_deep"]
        mine = {k: round(float(np.mean([st_[k] for st_ in stats_acc])), 3) for k in keys}
        mine["num_suffixes"] = int(stats_acc[-1]["num_suffixes"])
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    e2e_multi = None
    if world > 1 and not args.no_e2e:
        builder.close()              # every rank's bench context makes room for its create context
        e2e_multi = e2e_create_sharded(text, starts, flags, partitions, rank, world, local_rank, dev, args.backend,
                                       want_hash=args.e2e_hash)

    if rank == 0:
        keys = ["ms_total", "ms_normalize", "ms_hist_text", "ms_partition", "ms_passes", "ms_finish", "ms_deep"]
        avg = {k: float(np.mean([s[k] for s in stats_acc])) for k in keys}      # (forced-dist: per SHARD build)
        st = stats_acc[-1]
        # radix-partition kernel (k_scatter_text): algorithmic bytes per launch = n (text) + 4 s (indices),
        # SURVEY.md 8(d) / BASELINE.md section 4; s = suffixes this rank keeps
        alg_bytes = n + 4 * st["num_suffixes"]
        achieved = alg_bytes / (avg["ms_partition"] * 1e-3) / 1e9 if avg["ms_partition"] > 0 else 0.0
        # HBM bytes of that launch from the PMC counters: read from the committed summary of a profiled run of this same
        # command (the counters need rocprofv3 passes of their own), and labelled with the file they come from
        traffic, traffic_src = pmc_traffic_bytes(args.workload) if num_shards == 1 and not args.bases else (None, None)
        out = {
            "metric": "suffixes sorted/sec (SA+LCP)",
            "value": s_total * args.steps / dt,
            "unit": "suffixes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "value_best": (s_total / (min(placement_ms) * 1e-3)) if placement_ms and num_shards == 1 else None,
            "dtype": "u8 text / u32 indices / u64 packed keys",
            "data": data,
            "config": {"workload": label, "text_len": n, "num_suffixes": s_total,
                       "parallelism": f"prefix-bucket shards x{num_shards}" + (" (two per step on ONE GPU: forced-dist rehearsal)" if args.force_dist else ""), "bits_per_char": st["bits_per_char"],
                       "radix_passes": st["num_passes"], "digit_bits": st["digit_bits"],
                       "levels": st["num_levels"], "deep_records": st["deep_records"],
                       "placement_trials": max(1, args.placement_trials), "placement_ms": placement_ms},
            "roofline": {"kernel": ("k_msd_part_text" if st.get("partition_variant") else "k_scatter_text")
                         + " (radix partition: text -> (key, index) records in first-digit buckets)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": alg_bytes, "ms": avg["ms_partition"]},
            "device_ms": avg,
            "verified": verified,
        }
        if search is not None:
            out["search"] = search
        if forced is not None:
            out["forced_dist"] = forced
        if world > 1:
            # what a rank cannot shed: every rank streams the WHOLE text through the text pass
            # (ms_normalize + ms_hist_text) and through the scan side of the partition kernel; only the rest shrinks with N
            fixed = [round(r["ms_normalize"] + r["ms_hist_text"], 3) for r in per_rank]
            out["per_rank"] = {"device_ms": per_rank, "text_pass_ms": fixed,
                               "note": "strong scaling with a per-rank fixed cost: every rank reads the whole text in the text "
                                       "pass (text_pass_ms) and in the scan of the partition kernel (~2.5 ms of ms_partition at "
                                       "3.1 Gb); sorting work (ms_passes, ms_deep, the rest of ms_partition) is ~1/N.  One-GPU "
                                       "shard probe (profiles/r06_shard_probe.txt): 51.0 / 29.3 / 19.7 / 14.1 ms per rank at "
                                       "N = 1 / 2 / 4 / 8 => at most 87 / 65 / 45 % efficiency before any communication"}
            if args.share_device:
                out["per_rank"]["share_device"] = ("all ranks ran on cuda:0 (smoke test of the N > 1 leg on a one-GPU box): "
                                                   "value is NOT an N-GPU figure")
        if world == 1 and not args.no_cpu_baseline:
            sample_bases = min(bases, 400_000_000)
            full = None
            big_host = host_memory_gb() >= 200 and (os.cpu_count() or 1) >= 32
            if args.full_cpu_baseline == "on" or (args.full_cpu_baseline == "auto" and big_host and n > sample_bases + 1):
                full = lambda: text.cpu().numpy()       # noqa: E731 -- the oracle needs ~24 bytes per base of host memory
            out["cpu_baseline"] = cpu_baseline(text[:sample_bases].cpu().numpy(), flags, partitions, full_text=full)
        if world == 1 and not args.no_e2e:
            builder.close()          # the CLI is its own process with its own context: free this one's HBM first
            text_cpu = text.cpu().numpy()
            if st.get("partition_variant"):
                del text
                torch.cuda.empty_cache()
                fl = store_floor(s_total, n, int(st.get("top_hi", 0)) - int(st.get("top_lo", 0)))
                if fl:
                    out["roofline"]["store_floor_ms"] = fl["store_floor_ms"]
                    out["roofline"]["store_floor"] = {k: fl[k] for k in ("records", "bins", "run", "grid", "bytes")}
                    out["roofline"]["store_floor"]["what"] = ("the kernel's store pattern alone (12-byte records, runs of `run` records over "
                                                              "`bins` first digits, resident grid): profiles/micro/scatter_write.hip --floor")
            if host_memory_gb() >= 10 * n / 2**30 + 8:
                out["host_abi"] = host_abi(text_cpu, flags, partitions)
            out["e2e_create"] = e2e_create(text_cpu, starts, flags, partitions, s_total, want_hash=args.e2e_hash)
        if e2e_multi is not None:
            out["e2e_create"] = e2e_multi
        print(json.dumps(out), flush=True)
    builder.close()
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
