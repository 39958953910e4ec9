"""Device time of the reference authors' other large builds (reference Makefile:67-83): `--dna -s 111010010100110111`
on the 3.1 Gb human stand-in, `--dna -m 12` on the 100 Mb C. elegans stand-in, `--dna -m 16` on the 4.6 Mb E. coli
stand-in (second build on a warm context):   python profiles/modes_bench.py [substring of the case names to run]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sufr_amd
from sufr_amd import synth

CASES = [
    ("human -s 111010010100110111", synth.syn_human, 3_100_000_000, 4, dict(is_dna=True, seed_mask="111010010100110111", num_partitions=800)),
    ("human --dna -n 800 (Makefile:79)", synth.syn_human, 3_100_000_000, 4, dict(is_dna=True, num_partitions=800)),
    ("human --dna -m 16", synth.syn_human, 3_100_000_000, 4, dict(is_dna=True, max_query_len=16, num_partitions=800)),
    ("elegans -m 12", synth.syn_elegans, 100_286_401, 2, dict(is_dna=True, max_query_len=12, num_partitions=64)),
    ("elegans --dna", synth.syn_elegans, 100_286_401, 2, dict(is_dna=True, num_partitions=64)),
    ("ecoli -m 16", synth.syn_ecoli, 4_641_652, 1, dict(is_dna=True, max_query_len=16)),
    ("ecoli --dna", synth.syn_ecoli, 4_641_652, 1, dict(is_dna=True)),
]
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""
for name, gen, bases, seed, flags in CASES:
    if ONLY not in name:
        continue
    x, _ = gen(bases, seed=seed, device="cuda")
    db = sufr_amd.DeviceBuilder(0)
    out_sa = torch.empty(x.numel(), dtype=torch.int32, device="cuda")
    out_lcp = torch.empty_like(out_sa)
    for rep in range(2):
        sa, lcp = db.sort(x, raw_text=True, out_sa=out_sa, out_lcp=out_lcp, **flags)
    st = db.stats
    print(f"{name:36s} n={x.numel():>11d} s={sa.numel():>11d} levels={st.num_levels:3d} passes={st.num_passes:3d}  "
          f"total {st.ms_total:9.2f} ms = {sa.numel() / st.ms_total / 1e6:7.2f} G suffixes/s  (text {st.ms_normalize:.2f} part {st.ms_partition:.2f} "
          f"passes {st.ms_passes:.2f} finish {st.ms_finish:.2f} deep {st.ms_deep:.2f})", flush=True)
    db.close()
    del x, out_sa, out_lcp
    torch.cuda.empty_cache()
