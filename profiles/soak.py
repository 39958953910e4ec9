"""A soak of the whole library against the CPU checker with fresh seeds:   python profiles/soak.py [minutes] [seed]
A third argument "big" restricts the loop to structured texts of 1 - 6 Mb.  Loops until the time is up over (1) small random texts (tests/test_gpu_parity.py::_fuzz_text) and structured 20 k - 400 k
texts through the host ABI, (2) the same texts in forced windows, (3) the device search of the result against the host
search of the written file, (4) --seed-mask builds against the checker and --max-query-len builds against min(LCP, L) and
the descending-position rule, each in one window and in forced windows.  Prints one line per failure and a summary; exit code 1 when anything differed.
(Test-side tooling: it imports the checker from tests/, like the tests do.)"""
import os
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import sufr_amd
from oracle_helper import Oracle, naive_sa_lcp
from test_gpu_parity import _fuzz_text, _break_long_n_runs
from test_query import random_queries

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 100000
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"          # structured texts of 1 - 6 Mb only
rng = np.random.default_rng(seed)
oracle = Oracle()
ctx = sufr_amd.Context(0)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
fails = 0
cases = {"host": 0, "windows": 0, "search": 0, "options": 0}


def structured(rng):
    n = int(rng.integers(1_000_000, 6_000_000)) if BIG else int(rng.integers(40_000, 300_000))
    kind = int(rng.integers(0, 8))
    t = acgt[rng.integers(0, 4, n)]
    if kind == 1:
        fam = acgt[rng.integers(0, 4, int(rng.integers(200, 3000)))]
        for _ in range(int(rng.integers(5, 200))):
            at = int(rng.integers(0, n - fam.size)); c = fam.copy()
            hit = rng.random(fam.size) < rng.choice([0.0, 0.001, 0.02, 0.1])
            c[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
            t[at:at + fam.size] = c
    elif kind == 2:
        for _ in range(30):
            u = acgt[rng.integers(0, 4, int(rng.integers(1, 12)))]
            ln = int(rng.integers(50, 5000)); at = int(rng.integers(0, n - ln)); t[at:at + ln] = np.resize(u, ln)
    elif kind == 3:
        t = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)[rng.integers(0, 9, n)]
        for _ in range(40):
            ln = int(rng.integers(1, 900)); at = int(rng.integers(0, n - ln)); t[at:at + ln] = ord("N")
    elif kind == 4:
        for _ in range(10):
            ln = int(rng.integers(100, 30000)); at = int(rng.integers(0, n - ln)); t[at:at + ln] = acgt[rng.integers(0, 4)]
    elif kind == 5:
        t = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)[rng.integers(0, 20, n)]
    elif kind == 6:
        for _ in range(6):
            ln = int(rng.integers(500, 20000)); a = int(rng.integers(0, n - ln)); b = int(rng.integers(0, n - ln))
            t[b:b + ln] = t[a:a + ln].copy()
    elif kind == 7:
        t = acgt[rng.integers(0, 2, n)]
    t = t.copy()
    for c in rng.integers(1, n - 1, size=int(rng.integers(0, 5))):
        t[c] = ord("%")
    return np.concatenate([t, np.frombuffer(b"$", dtype=np.uint8)]), kind


def report(what, ctxt, detail):
    global fails
    fails += 1
    print(f"FAIL {what}: {ctxt}: {detail}", flush=True)


t_end = time.time() + minutes * 60
it = 0
while time.time() < t_end:
    it += 1
    if it % 3 == 0 or BIG:
        raw, kind = structured(rng); is_dna = kind != 5
    else:
        raw = _fuzz_text(rng); kind = -1; is_dna = bool(rng.random() < 0.6)
    soft = bool(rng.random() < 0.5)
    amb = bool(is_dna and rng.random() < 0.25)
    if amb:
        raw = _break_long_n_runs(raw, soft)
    ctxt = f"seed {seed} it {it} kind {kind} n={raw.size} dna={is_dna} soft={soft} amb={amb}"
    norm = oracle.normalize(np.ascontiguousarray(raw), soft)
    if BIG and kind in (2, 4) and amb:      # (megabase texts with planted runs and --allow-ambiguity: minutes in the CPU checker)
        continue
    try:
        if norm.size < 4:
            raise RuntimeError("too short for the reference")
        want_sa, want_lcp, _ = oracle.build(norm, is_dna=is_dna, allow_ambiguity=amb, threads=8)
    except RuntimeError:                    # inputs the reference itself cannot build (pivot hazards): the naive witness
        if norm.size > 20_000:
            continue
        want_sa, want_lcp = naive_sa_lcp(norm, is_dna, amb)
    want_sa = want_sa.astype(np.uint64); want_lcp = want_lcp.astype(np.uint64)
    # (1) host ABI, one window
    try:
        ctx.set_window(0, 0)
        b = sufr_amd.SufrBuilder(sufr_amd.SufrBuilderArgs(text=raw, is_dna=is_dna, allow_ambiguity=amb, ignore_softmask=soft),
                                 index_width=4, ctx=ctx, write=False)
        cases["host"] += 1
        if not (np.array_equal(b.suffix_array.astype(np.uint64), want_sa) and np.array_equal(b.lcp.astype(np.uint64), want_lcp)):
            report("host build", ctxt, f"arrays differ: text {bytes(raw)[:40]!r} sa {b.suffix_array[:12].tolist()} want {want_sa[:12].tolist()} "
                                       f"lcp {b.lcp[:12].tolist()} want {want_lcp[:12].tolist()}")
    except Exception as e:
        report("host build", ctxt, repr(e))
    # (2) forced windows, both widths
    if raw.size > 64:
        window = int(rng.integers(max(16, raw.size // 15 + 1), raw.size))
        margin = int(rng.choice([16, 64, 1000, 100000]))
        width = int(rng.choice([4, 8]))
        try:
            ctx.set_window(window, margin)
            b = sufr_amd.SufrBuilder(sufr_amd.SufrBuilderArgs(text=raw, is_dna=is_dna, allow_ambiguity=amb, ignore_softmask=soft),
                                     index_width=width, ctx=ctx, write=False)
            cases["windows"] += 1
            if not (np.array_equal(b.suffix_array.astype(np.uint64), want_sa) and np.array_equal(b.lcp.astype(np.uint64), want_lcp)
                    and np.array_equal(b.text, norm)):
                report("windowed build", ctxt + f" window={window} margin={margin} width={width}", "arrays differ")
        except Exception as e:
            report("windowed build", ctxt + f" window={window} margin={margin} width={width}", repr(e))
        ctx.set_window(0, 0)
    # (4) --seed-mask / --max-query-len, one window and forced windows
    if it % 5 == 0 and raw.size > 64 and norm.size >= 4:
        try:
            window = int(rng.integers(max(16, raw.size // 15 + 1), raw.size)); margin = int(rng.choice([16, 64, 1000]))
            width = int(rng.choice([4, 8]))
            if rng.random() < 0.5:
                mask = "1" + "".join(rng.choice(["0", "1"], size=int(rng.integers(1, 20)))) + "1"
                if "0" not in mask:
                    mask = "10" + mask                    # (the reference rejects masks without a 0, types.rs:36-200)
                try:
                    msa, mlcp, _ = oracle.build(norm, is_dna=is_dna, allow_ambiguity=amb, seed_mask=mask, threads=8)
                except RuntimeError:                      # inputs the reference itself cannot build (pivot hazards)
                    continue
                for w in ((0, 0), (window, margin)):
                    ctx.set_window(*w)
                    b = sufr_amd.SufrBuilder(sufr_amd.SufrBuilderArgs(text=raw, is_dna=is_dna, allow_ambiguity=amb, ignore_softmask=soft,
                                                                      seed_mask=mask), index_width=width, ctx=ctx, write=False)
                    if not (np.array_equal(b.suffix_array.astype(np.uint64), msa.astype(np.uint64)) and
                            np.array_equal(b.lcp.astype(np.uint64), mlcp.astype(np.uint64))):
                        report("seed-mask build", ctxt + f" mask={mask} window={w} width={width}", "arrays differ")
            else:
                L = int(rng.choice([1, 2, 5, 12, 21, 22, 40, 300]))
                got = []
                for w in ((0, 0), (window, margin)):
                    ctx.set_window(*w)
                    b = sufr_amd.SufrBuilder(sufr_amd.SufrBuilderArgs(text=raw, is_dna=is_dna, allow_ambiguity=amb, ignore_softmask=soft,
                                                                      max_query_len=L), index_width=width, ctx=ctx, write=False)
                    got.append((b.suffix_array.astype(np.int64), b.lcp.astype(np.uint64)))
                sa1, lcp1 = got[0]
                tie = lcp1 >= L
                if not (np.array_equal(lcp1, np.minimum(want_lcp, L)) and np.array_equal(np.sort(sa1), np.sort(want_sa.astype(np.int64)))
                        and bool(np.all(sa1[1:][tie[1:]] < sa1[:-1][tie[1:]]))):
                    report("max-query-len build", ctxt + f" L={L}", "not the canonical form")
                if not (np.array_equal(got[1][0], sa1) and np.array_equal(got[1][1], lcp1)):
                    report("max-query-len build", ctxt + f" L={L} window={window} margin={margin} width={width}", "windows differ from one window")
            cases["options"] += 1
        except Exception as e:
            report("option build", ctxt, repr(e))
        ctx.set_window(0, 0)
    # (3) device search against the host search of the written file
    if it % 4 == 0 and want_sa.size > 8:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "x.sufr")
            try:
                mql_build = int(rng.choice([0, 0, 5, 11]))
                args = sufr_amd.SufrBuilderArgs(text=raw, path=path, is_dna=is_dna, allow_ambiguity=amb, ignore_softmask=soft,
                                                max_query_len=mql_build or None)
                sufr_amd.SufrBuilder(args, index_width=4, ctx=ctx, write=True)
                f = sufr_amd.SufrFile(path)
                ix = sufr_amd.DeviceIndex.load(ctx, f)
                qs = random_queries(rng, f, 400, 24)
                mql = [None, 3, 8][int(rng.integers(0, 3))]
                lo, hi = ix.search(qs, mql)
                cases["search"] += 1
                for i, q in enumerate(qs):
                    r = f.search(q, mql)
                    if (int(lo[i]), int(hi[i])) != (r if r else (0, 0)):
                        report("device search", ctxt + f" built_mql={mql_build} mql={mql} query={q!r}", f"device {(int(lo[i]), int(hi[i]))} host {r}")
                        break
                ix.close(); f.close()
            except Exception as e:
                report("device search", ctxt, repr(e))
print(f"soak: seed {seed}, {it} iterations, {cases}, {fails} failures", flush=True)
ctx.close()
sys.exit(1 if fails else 0)
