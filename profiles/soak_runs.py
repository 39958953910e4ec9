"""Soak of the closed-form run buckets (sufr_runs.inc) against the oracle:   python profiles/soak_runs.py [minutes] [seed]
Texts of 2.5 - 8 Mb whose bucket of one repeated symbol holds 1 - 4 M suffixes: random symbol (N under --allow-ambiguity, with or
without --ignore-softmask on lower-case runs; G / T / A / C in a plain build), random run counts and lengths below 1 000 (the
byte-walking reference is the checker), random or templated tails, 1 - 3 shards with the device stitch, sometimes in two windows.
Whole SA and LCP must equal the oracle's.  (Test-side tooling: imports the checker and helpers from tests/.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import sufr_amd
from oracle_helper import Oracle
from test_gpu_parity import _sharded_arrays, _runs_text, _templated_runs_text

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 100000
rng = np.random.default_rng(seed)
o = Oracle()
t_end = time.time() + minutes * 60
cases = fails = 0
while time.time() < t_end:
    kind = int(rng.integers(0, 4))
    kw = dict(is_dna=True)
    soft = False
    hi = int(rng.integers(60, 990))
    if kind == 0:
        raw = _templated_runs_text(int(rng.integers(1, 1 << 30)), int(rng.integers(3000, 9000)), int(rng.integers(1, 30)), hi,
                                   ntemplates=int(rng.integers(2, 80)), tlen=int(rng.integers(30, 400)))
        kw["allow_ambiguity"] = True
    elif kind == 1:
        n = int(rng.integers(3_000_000, 8_000_000)); nr = int(n * rng.uniform(0.25, 0.6) / ((21 + hi) / 2))
        raw = _runs_text(int(rng.integers(1, 1 << 30)), n, ord("N"), nr, int(rng.integers(5, 40)), hi); kw["allow_ambiguity"] = True
    elif kind == 2:
        n = int(rng.integers(3_000_000, 8_000_000)); nr = int(n * rng.uniform(0.25, 0.6) / ((21 + hi) / 2))
        raw = _runs_text(int(rng.integers(1, 1 << 30)), n, int(rng.choice(list(b"acgtn"))), nr, int(rng.integers(5, 40)), hi)
        kw["allow_ambiguity"] = True; soft = True
    else:
        n = int(rng.integers(3_000_000, 8_000_000)); nr = int(n * rng.uniform(0.3, 0.6) / ((21 + hi) / 2))
        raw = _runs_text(int(rng.integers(1, 1 << 30)), n, int(rng.choice(list(b"ACGT"))), nr, int(rng.integers(5, 40)), hi)
    norm = o.normalize(raw, soft)
    osa, olcp, _ = o.build(norm, threads=min(32, os.cpu_count() or 1), **kw)
    x = torch.from_numpy(raw).cuda()
    db = sufr_amd.DeviceBuilder(0)
    mode = int(rng.integers(0, 4))
    if mode == 0:
        shards = int(rng.integers(2, 4))
        gsa, glcp, _ = _sharded_arrays(db, x, x.numel(), shards, raw_text=True, ignore_softmask=soft, **kw)
    else:
        if mode == 1:
            db.ctx.set_window(int(x.numel() * rng.uniform(0.4, 0.7)), 1 << 16)
        sa, lcp = db.sort(x, raw_text=True, ignore_softmask=soft, index_width=8 if mode == 1 else 4, **kw)
        gsa = sa.cpu().numpy(); glcp = lcp.cpu().numpy()
        if mode != 1:
            gsa = gsa.view(np.uint32); glcp = glcp.view(np.uint32)
    db.close()
    ok = np.array_equal(gsa.astype(np.uint64), osa.astype(np.uint64)) and np.array_equal(glcp.astype(np.uint64), olcp.astype(np.uint64))
    cases += 1
    if not ok:
        fails += 1
        print(f"FAIL kind={kind} mode={mode} n={raw.size} hi={hi} {kw} soft={soft} seed={seed}", flush=True)
print(f"soak_runs: seed {seed}, {minutes} min: {cases} texts with a run bucket of 1-4 M suffixes against the oracle, {fails} failures")
sys.exit(1 if fails else 0)
