"""Soak of the round-5 windowed-build code against the oracle with fresh seeds:   python profiles/soak_wide.py [minutes] [seed]
Random structured texts of 20 k - 250 k symbols (repeat families, tandem arrays, N runs, homopolymers, exact duplicates, protein,
two-letter texts) in forced windows with random window / margin, a random CAP on the re-build margin (so that repeats longer than
it take the whole-text repair, sufr_wide.inc repair_window), 1 - 6 shards stitched on the device, both index widths, plain /
--allow-ambiguity / non-DNA / -m L builds.  Plain builds must equal the oracle's arrays bit for bit, -m builds the one-window
build.  Prints one line per failure and a summary; exit code 1 when anything differed.  (Test-side tooling, like profiles/soak.py.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import sufr_amd
from sufr_amd import shards
from oracle_helper import Oracle
from test_gpu_parity import _break_long_n_runs

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 100000
rng = np.random.default_rng(seed)
oracle = Oracle()
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)


def text(rng):
    n = int(rng.integers(20_000, 250_000))
    kind = int(rng.integers(0, 8))
    t = acgt[rng.integers(0, 4, n)].copy()
    if kind == 1:
        fam = acgt[rng.integers(0, 4, int(rng.integers(200, 4000)))]
        for _ in range(int(rng.integers(5, 120))):
            at = int(rng.integers(0, n - fam.size)); c = fam.copy()
            hit = rng.random(fam.size) < rng.choice([0.0, 0.0, 0.001, 0.02])
            c[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
            t[at:at + fam.size] = c
    elif kind == 2:
        for _ in range(30):
            u = acgt[rng.integers(0, 4, int(rng.integers(1, 12)))]
            ln = int(rng.integers(50, 6000)); at = int(rng.integers(0, n - ln)); t[at:at + ln] = np.resize(u, ln)
    elif kind == 3:
        t = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)[rng.integers(0, 9, n)].copy()
        for _ in range(40):
            ln = int(rng.integers(1, 950)); at = int(rng.integers(0, n - ln)); t[at:at + ln] = ord("N")
    elif kind == 4:
        for _ in range(10):
            ln = int(rng.integers(100, 20000)); at = int(rng.integers(0, n - ln)); t[at:at + ln] = acgt[rng.integers(0, 4)]
    elif kind == 5:
        t = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)[rng.integers(0, 20, n)].copy()
    elif kind == 6:
        for _ in range(6):
            ln = int(rng.integers(500, 15000)); a = int(rng.integers(0, n - ln)); b = int(rng.integers(0, n - ln))
            t[b:b + ln] = t[a:a + ln].copy()
    elif kind == 7:
        t = acgt[rng.integers(0, 2, n)].copy()
    t = _break_long_n_runs(t)            # (runs of >= 1000 N: the reference's own --allow-ambiguity output is approximate there)
    t[-1] = ord("$")
    return t, kind


def build(x, n, window, margin, retry, nsh, width, **flags):
    db = sufr_amd.DeviceBuilder(0)
    db.ctx.set_window(window, margin); db.ctx.set_window_retry(retry)
    parts, rows, rep = [], [], 0
    for r in range(nsh):
        sa, lcp = db.sort(x, index_width=width, shard_index=r, num_shards=nsh, **flags)
        rep += db.ctx.window_repairs
        parts.append((sa.clone(), lcp.clone())); rows.append(shards.gather_boundaries_device(sa, sa.numel()))
    bounds = torch.cat(rows).contiguous()
    for r in range(nsh):
        shards.stitch_device(db.ctx, n, bounds, r, parts[r][1])
    db.ctx.synchronize()
    sa = torch.cat([p[0] for p in parts]).cpu().numpy(); lcp = torch.cat([p[1] for p in parts]).cpu().numpy()
    db.close()
    if width == 4:
        sa, lcp = sa.view(np.uint32), lcp.view(np.uint32)
    return sa.astype(np.uint64), lcp.astype(np.uint64), rep


t_end = time.time() + minutes * 60
it = fails = repaired_cases = sharded_cases = 0
while time.time() < t_end:
    t, kind = text(rng)
    n = t.size
    x = torch.from_numpy(t).cuda()
    dna = kind != 5
    flags = dict(is_dna=True) if dna and rng.random() < 0.6 else (dict(is_dna=True, allow_ambiguity=True) if dna and rng.random() < 0.5 else dict())
    window = int(rng.integers(max(2_000, n // 14), n))
    margin = int(rng.choice([16, 64, 300, 2000]))
    retry = int(rng.choice([0, 16, 100, 700, 5000]))
    nsh = int(rng.choice([1, 1, 2, 3, 6]))
    width = int(rng.choice([4, 8]))
    mql = int(rng.choice([0, 0, 0, 9, 40, 900]))
    what = f"seed={seed} it={it} kind={kind} n={n} window={window} margin={margin} retry={retry} shards={nsh} width={width} flags={flags} mql={mql}"
    try:
        if mql:
            db = sufr_amd.DeviceBuilder(0)
            want_sa, want_lcp = (a.cpu().numpy().view(np.uint32).astype(np.uint64) for a in db.sort(x, max_query_len=mql, **flags))
            db.close()
            sa, lcp, rep = build(x, n, window, margin, retry, nsh, width, max_query_len=mql, **flags)
        else:
            osa, olcp, _ = oracle.build(t, threads=8, **flags)
            want_sa, want_lcp = osa.astype(np.uint64), olcp.astype(np.uint64)
            sa, lcp, rep = build(x, n, window, margin, retry, nsh, width, **flags)
        ok = np.array_equal(sa, want_sa) and np.array_equal(lcp, want_lcp)
    except Exception as e:                       # a refusal is a failure here: every one of these inputs must build
        ok = False; rep = 0
        what += f"  EXCEPTION {e}"
    if not ok:
        fails += 1
        print("FAIL", what, flush=True)
    repaired_cases += rep > 0; sharded_cases += nsh > 1
    it += 1
print(f"soak_wide: seed {seed}, {it} builds in {minutes} min ({sharded_cases} sharded, {repaired_cases} with whole-text repairs), {fails} failures")
sys.exit(1 if fails else 0)
