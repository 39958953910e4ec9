"""Soak of the per-group sort of the re-keying levels against the CPU checker:   python profiles/soak_families.py [minutes] [seed]
Texts of 2 - 6 Mb with high-copy repeat families (tie groups of every size class: windows of small groups, groups above
4 096 records, groups above 16 384) plus homopolymer / tandem stretches and N runs, built through the host ABI with random
flags and, in turn, plainly, in forced windows, with --max-query-len, or over 2 - 5 shards through the device ABI with the
first LCPs stitched on the device.  (Test-side tooling, like soak.py.)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import sufr_amd
from oracle_helper import Oracle
from test_gpu_parity import _sharded_arrays

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 100000
rng = np.random.default_rng(seed)
oracle = Oracle()
ctx = sufr_amd.Context(0)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
fails = 0
cases = 0
t_end = time.time() + minutes * 60
while time.time() < t_end:
    cases += 1
    n = int(rng.integers(2_000_000, 6_000_000))
    t = acgt[rng.integers(0, 4, n)]
    desc = []
    for _ in range(int(rng.integers(1, 4))):
        unit = int(rng.integers(30, 400)); copies = int(rng.integers(2_000, 60_000)); mut = float(rng.choice([0.0, 0.002, 0.01, 0.03]))
        copies = min(copies, n // (2 * unit))
        fam = acgt[rng.integers(0, 4, unit)]
        for a in rng.integers(0, n - unit - 1, copies):
            c = fam.copy()
            if mut:
                hit = rng.random(unit) < mut
                c[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
            t[a:a + unit] = c
        desc.append((unit, copies, mut))
    for _ in range(int(rng.integers(0, 4))):
        ln = int(rng.integers(1_000, 60_000)); at = int(rng.integers(0, n - ln))
        t[at:at + ln] = np.resize(acgt[rng.integers(0, 4, int(rng.integers(1, 6)))], ln)
    for _ in range(int(rng.integers(0, 3))):
        ln = int(rng.integers(100, 900)); at = int(rng.integers(0, n - ln)); t[at:at + ln] = ord("N")
    raw = np.concatenate([t, np.frombuffer(b"$", dtype=np.uint8)])
    amb = bool(rng.random() < 0.3)
    ctxt = f"seed {seed} case {cases} n={raw.size} families={desc} amb={amb}"
    try:
        want_sa, want_lcp, _ = oracle.build(raw, is_dna=True, allow_ambiguity=amb, threads=8)
    except RuntimeError:
        continue
    mode = cases % 4
    try:
        if mode == 3:                                           # shards of the device ABI, concatenated
            shards = int(rng.integers(2, 6))
            db = sufr_amd.DeviceBuilder(0)
            x = torch.from_numpy(raw).cuda()
            Lq = int(rng.choice([8, 16, 40])) if rng.random() < 0.3 else None     # capped: against the one-GPU build of the same cap
            gsa, glcp, _ = _sharded_arrays(db, x, x.numel(), shards, is_dna=True, allow_ambiguity=amb, max_query_len=Lq)
            if Lq is not None:
                one_sa, one_lcp = db.sort(x, is_dna=True, allow_ambiguity=amb, max_query_len=Lq)
                want_sa = one_sa.cpu().numpy().view(np.uint32); want_lcp = one_lcp.cpu().numpy().view(np.uint32)
            db.close()
            ok = np.array_equal(gsa.astype(np.int64), want_sa.astype(np.int64)) and np.array_equal(glcp.astype(np.int64), want_lcp.astype(np.int64))
            if not ok:
                fails += 1
                print(f"FAIL {ctxt} mode={mode} shards={shards}", flush=True)
            print(f"case {cases} mode {mode} shards={shards} {'ok' if ok else 'FAIL'}", flush=True)
            continue
        if mode == 1:
            ctx.set_window(int(rng.integers(raw.size // 6, raw.size)), int(rng.choice([64, 5000, 200000])))
        L = int(rng.choice([8, 16, 40])) if mode == 2 else None
        b = sufr_amd.SufrBuilder(sufr_amd.SufrBuilderArgs(text=raw, is_dna=True, allow_ambiguity=amb, max_query_len=L), index_width=4, ctx=ctx, write=False)
        ctx.set_window(0, 0)
        sa = b.suffix_array.astype(np.int64); lcp = b.lcp.astype(np.int64)
        if L is None:
            ok = np.array_equal(sa, want_sa.astype(np.int64)) and np.array_equal(lcp, want_lcp.astype(np.int64))
        else:
            tie = lcp >= L
            ok = (np.array_equal(lcp, np.minimum(want_lcp.astype(np.int64), L)) and np.array_equal(np.sort(sa), np.sort(want_sa.astype(np.int64)))
                  and bool(np.all(sa[1:][tie[1:]] < sa[:-1][tie[1:]])))
        if not ok:
            fails += 1
            print(f"FAIL {ctxt} mode={mode}", flush=True)
        print(f"case {cases} mode {mode} levels={b.stats.num_levels} deep={b.stats.deep_records} {'ok' if ok else 'FAIL'}", flush=True)
    except Exception as e:
        fails += 1
        print(f"FAIL {ctxt} mode={mode}: {e!r}", flush=True)
        ctx.set_window(0, 0)
print(f"soak_families: seed {seed}, {cases} texts, {fails} failures")
sys.exit(1 if fails else 0)
