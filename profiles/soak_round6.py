"""A soak of round 6's two new paths against the CPU checker with fresh seeds:   python profiles/soak_round6.py [minutes] [seed]
(1) DNA texts with random bytes outside {$ % A C G N T} planted (IUPAC letters, '#', bytes below '$' and above 'T'; single bytes,
clusters, runs, inside N runs and repeats), random flags, 1-5 shards: the shards concatenated = the oracle's arrays.
(2) --max-query-len L for random L in 11..21 (the capped build built directly), half of them with such bytes planted too, 1-4
shards: = the canonical form computed from the oracle's exact arrays.  Prints one line per failure and a summary; exit code 1 when anything differed.
(Test-side tooling: it imports the checker from tests/, like the tests do.)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import sufr_amd
from oracle_helper import Oracle
from test_gpu_mql_fast import canonical

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 100000
rng = np.random.default_rng(seed)
oracle = Oracle()
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
FOREIGN = np.frombuffer(b"RYKMSWBDHVU#!*-Z[~", dtype=np.uint8)


def text(rng):
    n = int(rng.choice([300, 5_000, 60_000, 400_000, 2_500_000]))
    t = acgt[rng.choice(4, n, p=rng.choice([[.25, .25, .25, .25], [.55, .15, .15, .15], [.4, .1, .1, .4]]))].copy()
    kind = int(rng.integers(0, 6))
    if kind == 1 and n > 5000:                                # a family of near-identical copies
        fam = acgt[rng.integers(0, 4, int(rng.integers(100, 2000)))]
        for _ in range(int(rng.integers(5, 400))):
            at = int(rng.integers(0, n - fam.size)); c = fam.copy()
            hit = rng.random(fam.size) < rng.choice([0.0, 0.002, 0.03])
            c[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
            t[at:at + fam.size] = c
    elif kind == 2:                                           # tandem arrays and homopolymers
        for _ in range(12):
            u = acgt[rng.integers(0, 4, int(rng.integers(1, 7)))]
            ln = int(rng.integers(20, max(21, n // 8))); at = int(rng.integers(0, n - ln)); t[at:at + ln] = np.resize(u, ln)
    elif kind == 3:                                           # soft-masked stretches and N runs (below 1000: the oracle stays exact under -a)
        for _ in range(20):
            ln = int(rng.integers(1, min(900, n // 4))); at = int(rng.integers(0, n - ln))
            t[at:at + ln] = ord("N") if rng.random() < 0.5 else (t[at:at + ln] | 0x20)
    elif kind == 4 and n > 2000:                              # exact duplicates
        for _ in range(4):
            ln = int(rng.integers(50, n // 6)); a = int(rng.integers(0, n - ln)); b = int(rng.integers(0, n - ln)); t[b:b + ln] = t[a:a + ln].copy()
    for c in rng.integers(1, n - 1, size=int(rng.integers(0, 4))):
        t[c] = ord("%")
    t[-1] = ord("$")
    return t


def long_n_runs(norm):
    isn = np.concatenate([[0], (norm == ord("N")).astype(np.int8), [0]]); d = np.diff(isn)
    return int(((np.nonzero(d == -1)[0] - np.nonzero(d == 1)[0]) >= 1000).sum())


def expected(norm, amb):
    """the oracle's arrays; with --allow-ambiguity and two or more runs of >= 1000 'N' (stretches of kind 3 can touch) the reference's
    own order is approximate (its N-run shortcut), and the witness is the exact order: the oracle's byte-wise
    build over all positions (the suffix set of --allow-ambiguity), which has no N-run table"""
    if amb and long_n_runs(norm) >= 2:
        cases["exact_witness"] = cases.get("exact_witness", 0) + 1
        osa, olcp, _ = oracle.build(norm, is_dna=False, num_partitions=16, threads=8)
    else:
        osa, olcp, _ = oracle.build(norm, is_dna=True, allow_ambiguity=amb, num_partitions=2 if norm.size < 5000 else 16, threads=8)
    return osa, olcp


def plant(rng, t):
    n = t.size
    k = int(rng.integers(1, max(2, min(40, n // 70))))
    for _ in range(k):
        at = int(rng.integers(0, n - 1)); b = FOREIGN[rng.integers(0, FOREIGN.size)]
        shape = rng.random()
        if shape < 0.7: t[at] = b
        elif shape < 0.85: t[at:min(n - 1, at + int(rng.integers(2, 6)))] = b
        else:                                                 # a copy of the surrounding stretch elsewhere, with and without the byte
            ln = int(rng.integers(10, min(400, n // 4))); src = max(0, at - ln // 2); dst = int(rng.integers(0, n - 1 - ln))
            if src + ln < n - 1:
                t[dst:dst + ln] = t[src:src + ln]; t[at] = b
    if rng.random() < 0.3 and n > 100:
        lower = rng.integers(0, n - 1, 5); t[lower] = t[lower] | 0x20
    return t


def shards_of(db, x, shards, **kw):
    sas, lcps = [], []
    for k in range(shards):
        sa, lcp = db.sort(x, raw_text=True, shard_index=k, num_shards=shards, **kw)
        sas.append(sa.cpu().numpy().view(np.uint32).copy()); lcps.append(lcp.cpu().numpy().view(np.uint32).copy())
    starts = np.cumsum([0] + [p.size for p in sas[:-1]])
    return np.concatenate(sas), np.concatenate(lcps), starts[1:]


db = sufr_amd.DeviceBuilder(0)
fails = 0
cases = {"exceptions": 0, "exceptions_taken": 0, "capped": 0}
t_end = time.time() + minutes * 60
while time.time() < t_end:
    raw = text(rng)
    soft = bool(rng.random() < 0.3); amb = bool(rng.random() < 0.4)
    x = None
    if rng.random() < 0.6:
        raw = plant(rng, raw); raw[-1] = ord("$")
        shards = int(rng.choice([1, 1, 2, 3, 5]))
        x = torch.from_numpy(raw).cuda()
        norm = oracle.normalize(raw, soft)
        try:
            osa, olcp = expected(norm, amb)
        except RuntimeError:           # (the reference cannot draw its pivots from a handful of eligible suffixes: not a case)
            continue
        try:
            gsa, glcp, cuts = shards_of(db, x, shards, is_dna=True, allow_ambiguity=amb, ignore_softmask=soft)
        except Exception as e:
            fails += 1; print(f"FAIL exceptions: n={raw.size} shards={shards} amb={amb} soft={soft}: {e!r}", flush=True); continue
        cases["exceptions"] += 1
        cases["exceptions_taken"] += int(db.stats.num_exceptions > 0)
        keep = np.ones(osa.size, dtype=bool); keep[cuts[cuts < osa.size]] = False
        if gsa.size != osa.size or not np.array_equal(gsa, osa) or not np.array_equal(glcp[keep], olcp[keep]):
            fails += 1
            print(f"FAIL exceptions: n={raw.size} shards={shards} amb={amb} soft={soft} seed={seed}: arrays differ "
                  f"({gsa.size} vs {osa.size} suffixes)", flush=True)
    else:
        L = int(rng.integers(11, 22)); shards = int(rng.choice([1, 1, 2, 4]))
        if rng.random() < 0.5:                                 # capped builds with listed bytes: re-placed under the capped order
            raw = plant(rng, raw); raw[-1] = ord("$")
        x = torch.from_numpy(raw).cuda()
        norm = oracle.normalize(raw, soft)
        try:
            osa, olcp = expected(norm, amb)
        except RuntimeError:
            continue
        wsa, wlcp = canonical(osa, olcp, L)
        try:
            gsa, glcp, cuts = shards_of(db, x, shards, is_dna=True, allow_ambiguity=amb, ignore_softmask=soft, max_query_len=L)
        except Exception as e:
            fails += 1; print(f"FAIL capped: n={raw.size} L={L} shards={shards}: {e!r}", flush=True); continue
        cases["capped"] += 1
        keep = np.ones(wsa.size, dtype=bool); keep[cuts[cuts < wsa.size]] = False
        if gsa.size != wsa.size or not np.array_equal(gsa, wsa) or not np.array_equal(glcp[keep], wlcp[keep]):
            fails += 1
            print(f"FAIL capped: n={raw.size} L={L} shards={shards} amb={amb} soft={soft} seed={seed}: arrays differ", flush=True)
db.close()
print(f"soak_round6: seed {seed}, {minutes} min: {cases}, {fails} failures", flush=True)
sys.exit(1 if fails else 0)
