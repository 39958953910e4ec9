"""Run-to-run variance probe.  Device time of one build depends on where the work buffers land in HBM
(profiles/README.md), so settings are compared INSIDE one allocation: every round creates a context
(fresh buffers) and runs each setting on it in turn
(python profiles/mode_probe.py [n_bases] ["K=V,K=V;K=V;..."] [rounds])."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sufr_amd
from sufr_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
cfgs = sys.argv[2].split(";") if len(sys.argv) > 2 else [""]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
text, _ = synth.syn_human(n, seed=4, device="cuda")
out_sa = torch.empty(text.numel(), dtype=torch.int32, device="cuda")
out_lcp = torch.empty(text.numel(), dtype=torch.int32, device="cuda")
for round_ in range(rounds):
    db = sufr_amd.DeviceBuilder(0)
    db.sort(text, is_dna=True, ignore_softmask=True, raw_text=True, out_sa=out_sa, out_lcp=out_lcp)   # allocate
    for cfg in cfgs:
        for kv in filter(None, cfg.split(",")):
            k, v = kv.split("=")
            os.environ[k] = v
        db.sort(text, is_dna=True, ignore_softmask=True, raw_text=True, out_sa=out_sa, out_lcp=out_lcp)
        st = db.stats
        print(f"round {round_} [{cfg}]: total {st.ms_total:.1f} norm {st.ms_normalize:.2f} hist {st.ms_hist_text:.2f} "
              f"part {st.ms_partition:.2f} passes {st.ms_passes:.2f} finish {st.ms_finish:.2f} deep {st.ms_deep:.2f}",
              flush=True)
    db.close()
