"""Soak of the large-tile path (n >= 2^25: the 1024-thread instantiations of k_msd_part_text) against the oracle:
   python profiles/soak_part.py [minutes] [seed]
Genome-shaped texts of 34 - 70 Mb (sufr_amd/synth.py: soft-masked runs, N runs, repeat families) under random flags -- which
decide how dense the suffix-start bitmap is: half-empty words, whole tiles above the staging capacity (two units per tile) --,
random alphabets of 2 / 3 / 4 bits per code, and 1 / 2 / 3 / 5 / 8 shards with the device stitch.  Whole SA and LCP must equal the
oracle's.  (Test-side tooling: imports the checker and helpers from tests/.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import sufr_amd
from sufr_amd import synth
from oracle_helper import Oracle
from test_gpu_parity import _sharded_arrays

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 100000
rng = np.random.default_rng(seed)
o = Oracle()
db = sufr_amd.DeviceBuilder(0)
t_end = time.time() + minutes * 60
cases = fails = 0
while time.time() < t_end:
    n = int(rng.integers(33_600_000, 70_000_000))
    kind = int(rng.integers(0, 5))
    if kind <= 2:
        x, _ = synth.syn_human(n, seed=int(rng.integers(1, 1 << 30)), device="cuda")
        amb = bool(rng.random() < 0.4); soft = bool(rng.random() < 0.5)
        if amb:                                            # N runs broken below 1000 at random places: the reference stays deterministic
            g = torch.Generator(device="cuda"); g.manual_seed(int(rng.integers(1, 1 << 30)))
            isn = (x == ord("N")) | (x == ord("n")) | (torch.tensor(soft, device="cuda") & (x >= 97) & (x <= 122))
            brk = isn & ((torch.rand(x.numel(), generator=g, device="cuda") < 1 / 300) | (torch.arange(x.numel(), device="cuda") % 900 == 0))
            x[brk] = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[torch.randint(0, 4, (x.numel(),), generator=g, device="cuda")][brk]
            x[-1] = ord("$")
        kw = dict(is_dna=True, allow_ambiguity=amb, ignore_softmask=soft)
    else:
        sigma = int(rng.choice([2, 3, 5, 9, 14]))
        alpha = np.frombuffer(b"ACDEFGHIKLMNPQ", dtype=np.uint8)[:sigma]
        raw = alpha[rng.integers(0, sigma, n)].copy(); raw[-1] = ord("$")
        x = torch.from_numpy(raw).cuda(); kw = dict(is_dna=False, ignore_softmask=False)
    shards = int(rng.choice([1, 1, 2, 3, 5, 8]))
    raw = x.cpu().numpy()
    norm = o.normalize(raw, kw.get("ignore_softmask", False))
    okw = {k: v for k, v in kw.items() if k != "ignore_softmask"}
    if shards == 1:
        sa, lcp = db.sort(x, raw_text=True, **kw)
        gsa = sa.cpu().numpy().view(np.uint32); glcp = lcp.cpu().numpy().view(np.uint32)
    else:
        gsa, glcp, _ = _sharded_arrays(db, x, x.numel(), shards, raw_text=True, **kw)
    osa, olcp, _ = o.build(norm, threads=min(32, os.cpu_count() or 1), **okw)
    ok = np.array_equal(gsa, osa) and np.array_equal(glcp, olcp)
    cases += 1
    if not ok:
        fails += 1
        print(f"FAIL n={n} kind={kind} shards={shards} {kw} seed={seed}", flush=True)
    del x
print(f"soak_part: seed {seed}, {minutes} min: {cases} texts of 34-70 Mb against the oracle, {fails} failures")
db.close()
sys.exit(1 if fails else 0)
