"""Whole-array evidence at the headline size: the FULL 3.1 Gb stand-in (BASELINE configs[3], --dna
--ignore-softmask -n 256; second argument "dna": the same text with its soft-masked repeats indexed) built on the GPU and by the oracle (the C restatement of the reference algorithm, pinned to its golden
files) on the host cores; xxh64 of the whole SA and of the whole LCP array compared.   python profiles/c4_oracle_hash.py [bases]
Not part of the test suite: ~50 GB of host memory and a minute or two of 32 host threads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import xxhash
import sufr_amd
from sufr_amd import synth
from oracle_helper import Oracle

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "ignore_softmask"       # or "dna": the soft-masked repeats indexed (Makefile:79);
soft = mode in ("ignore_softmask", "iupac")                          # "iupac": the headline flags on the text with 50 IUPAC letters planted
mem_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2**30
need_gb = bases * (24 if soft else 48) / 2**30
print(f"host memory {mem_gb:.0f} GB, this run needs ~{need_gb:.0f} GB", flush=True)
if mem_gb < need_gb * 1.5:
    sys.exit("not enough host memory for the oracle at this size")
x, _ = synth.syn_human(bases, seed=4, device="cuda")
if mode == "iupac":          # (round 6: bytes outside {$ % A C G N T} keep the 3-bit table, sufr_exc.inc)
    g = torch.Generator(device="cuda"); g.manual_seed(17)
    at = torch.randint(1000, x.numel() - 1000, (50,), generator=g, device="cuda")
    x[at] = torch.tensor(list(b"RYKMSWBDHV" * 5), dtype=torch.uint8, device="cuda")
db = sufr_amd.DeviceBuilder(0)
sa, lcp = db.sort(x, is_dna=True, ignore_softmask=soft, raw_text=True, num_partitions=256)
st = db.stats
print(f"GPU build ({mode}): n={x.numel()} s={st.num_suffixes} device {st.ms_total:.1f} ms, {st.num_levels} levels, "
      f"{st.num_exceptions} bytes outside the DNA table, {st.num_reinserted} suffixes re-placed", flush=True)
gsa = sa.cpu().numpy().view(np.uint32); glcp = lcp.cpu().numpy().view(np.uint32)
raw = x.cpu().numpy()
del x, sa, lcp
db.close(); torch.cuda.empty_cache()
o = Oracle(native=True)
norm = o.normalize(raw, soft)
del raw
t0 = time.perf_counter()
osa, olcp, ost = o.build(norm, is_dna=True, num_partitions=256, threads=min(32, os.cpu_count() or 1))
dt = time.perf_counter() - t0
print(f"oracle: {ost.num_suffixes} suffixes in {dt:.1f} s on {min(32, os.cpu_count() or 1)} threads (partition {ost.t_partition:.1f} s, sort {ost.t_sort:.1f} s)", flush=True)
assert gsa.size == osa.size, (gsa.size, osa.size)
h = [xxhash.xxh64(a.tobytes()).hexdigest() for a in (gsa, osa, glcp, olcp)]
print(f"SA  xxh64 gpu {h[0]} oracle {h[1]} {'EQUAL' if h[0] == h[1] else 'DIFFERENT'}")
print(f"LCP xxh64 gpu {h[2]} oracle {h[3]} {'EQUAL' if h[2] == h[3] else 'DIFFERENT'}  (max LCP {int(glcp.max())})")
sys.exit(0 if h[0] == h[1] and h[2] == h[3] else 1)
