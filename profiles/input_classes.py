"""Device time of one build for the input classes of DESIGN.md section 4 ("other input classes"), second build on a
warm context:   python profiles/input_classes.py [class ...]
Classes: human, human_dna (soft-masked repeats indexed), human_amb, human_amb_soft, protein, all_a, dup100k, copies300,
human_iupac (the headline text with 50 IUPAC ambiguity codes planted, as an NCBI GRCh38 carries them), human_hash (the headline
text with '#' between the sequences: any byte outside {$ % A C G N T} takes a DNA build off the fixed 3-bit code table)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sufr_amd
from sufr_amd import synth


def acgt(g, n, dev):
    return torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n,), generator=g, device=dev)]


def make(name):
    dev = "cuda"
    g = torch.Generator(device=dev); g.manual_seed(17)
    if name.startswith("human"):
        x, _ = synth.syn_human(3_100_000_000, seed=4, device=dev)
        flags = {"human": dict(ignore_softmask=True), "human_dna": {}, "human_amb": dict(allow_ambiguity=True),
                 "human_amb_soft": dict(allow_ambiguity=True, ignore_softmask=True),
                 "human_iupac": dict(ignore_softmask=True), "human_hash": dict(ignore_softmask=True)}[name]
        if name == "human_iupac":
            at = torch.randint(1000, x.numel() - 1000, (50,), generator=g, device=dev)
            x[at] = torch.tensor(list(b"RYKMSWBDHV" * 5), dtype=torch.uint8, device=dev)
        if name == "human_hash":
            x[x == ord("%")] = ord("#")
        return x, dict(is_dna=True, **flags)
    if name == "protein":
        aa = torch.tensor(list(b"ACDEFGHIKLMNPQRSTVWY"), dtype=torch.uint8, device=dev)
        x = aa[torch.randint(0, 20, (1_000_000_000,), generator=g, device=dev)]
        x[torch.arange(500, x.numel(), 500, device=dev)] = ord("%")
        x[-1] = ord("$")
        return x, dict(is_dna=False)
    if name == "all_a":
        x = torch.full((1_000_000_001,), ord("A"), dtype=torch.uint8, device=dev); x[-1] = ord("$")
        return x, dict(is_dna=True)
    if name == "dup100k":                     # 20 Mb random DNA with one exact 100 kb duplicate
        x = acgt(g, 20_000_001, dev); x[15_000_000:15_100_000] = x[3_000_000:3_100_000]; x[-1] = ord("$")
        return x, dict(is_dna=True)
    if name == "copies300":                   # 300 exact copies of a 50 kb segment in 50 Mb
        x = acgt(g, 50_000_001, dev)
        seg = x[1_000_000:1_050_000].clone()
        for k in range(300):
            at = 2_000_000 + k * 150_000
            x[at:at + 50_000] = seg
        x[-1] = ord("$")
        return x, dict(is_dna=True)
    raise SystemExit(f"unknown class {name}")


names = sys.argv[1:] or ["human", "human_dna", "human_amb", "human_amb_soft", "protein", "all_a", "dup100k", "copies300"]
for name in names:
    x, flags = make(name)
    db = sufr_amd.DeviceBuilder(0)
    out_sa = torch.empty(x.numel(), dtype=torch.int32, device="cuda")
    out_lcp = torch.empty_like(out_sa)
    for rep in range(2):
        sa, lcp = db.sort(x, raw_text=True, out_sa=out_sa, out_lcp=out_lcp, **flags)
    st = db.stats
    print(f"{name:16s} n={x.numel():>11d} s={sa.numel():>11d} levels={st.num_levels:3d} msd_passes={st.num_passes:2d} "
          f"deep_records={st.deep_records:>11d}  total {st.ms_total:8.2f} ms  (norm {st.ms_normalize:.2f} hist {st.ms_hist_text:.2f} "
          f"part {st.ms_partition:.2f} passes {st.ms_passes:.2f} deep {st.ms_deep:.2f})", flush=True)
    db.close()
    del x, out_sa, out_lcp, sa, lcp
    torch.cuda.empty_cache()
