"""Run-to-run variance probe: the same build repeated inside one process with the workspace re-allocated
between rounds, optionally cycling environment settings per round
(python profiles/mode_probe.py [n_bases] ["K=V,K=V;K=V;..."])."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sufr_amd
from sufr_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
cfgs = sys.argv[2].split(";") if len(sys.argv) > 2 else [""]
text, _ = synth.syn_human(n, seed=4, device="cuda")
out_sa = torch.empty(text.numel(), dtype=torch.int32, device="cuda")
out_lcp = torch.empty(text.numel(), dtype=torch.int32, device="cuda")
for round_ in range(2 * len(cfgs) if len(cfgs) > 1 else 8):
    cfg = cfgs[(round_ // 2) % len(cfgs)]
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    db = sufr_amd.DeviceBuilder(0)
    for rep in range(2):
        db.sort(text, is_dna=True, ignore_softmask=True, raw_text=True, out_sa=out_sa, out_lcp=out_lcp)
        st = db.stats
        if rep:
            print(f"round {round_} [{cfg}]: total {st.ms_total:.1f} hist {st.ms_hist_text:.2f} part {st.ms_partition:.2f} "
                  f"passes {st.ms_passes:.2f} finish {st.ms_finish:.2f} deep {st.ms_deep:.2f}", flush=True)
    db.close()
