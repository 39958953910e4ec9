"""Per-shard device time on ONE GPU: what a rank of an N-GPU job spends on its prefix-bucket range
(python profiles/shard_probe.py [n_bases]).  The text passes (splitter histogram, pass-0 histogram and the
partition kernel) stream the whole text on every rank; everything after them scales with s / N."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sufr_amd
from sufr_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
text, _ = synth.syn_human(n, seed=4, device="cuda")
out_sa = torch.empty(text.numel(), dtype=torch.int32, device="cuda")
out_lcp = torch.empty(text.numel(), dtype=torch.int32, device="cuda")
db = sufr_amd.DeviceBuilder(0)
SHARDS = tuple(int(x) for x in os.environ.get("SUFR_SHARDS", "1,2,4,8").split(","))
ENDS_ONLY = bool(os.environ.get("SUFR_SHARD_ENDS"))     # time only the first and the last rank of every split
for shards in SHARDS:
    worst = 0.0
    for r in (sorted({0, shards - 1}) if ENDS_ONLY else range(shards)):
        for rep in range(int(os.environ.get("SUFR_SHARD_REPS", "2"))):
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); ev0.record()
            db.sort(text, is_dna=True, ignore_softmask=True, raw_text=True, out_sa=out_sa, out_lcp=out_lcp,
                    shard_index=r, num_shards=shards)
            ev1.record(); torch.cuda.synchronize()
        st = db.stats
        wall = ev0.elapsed_time(ev1)
        worst = max(worst, wall)
        if r in (0, shards - 1):
            print(f"shards {shards} rank {r}: s={db.num_suffixes} wall {wall:.1f} ms | device total {st.ms_total:.1f} norm "
                  f"{st.ms_normalize:.2f} hist {st.ms_hist_text:.2f} part {st.ms_partition:.2f} passes {st.ms_passes:.2f} "
                  f"finish {st.ms_finish:.2f} deep {st.ms_deep:.2f}", flush=True)
    print(f"shards {shards}: slowest rank {worst:.1f} ms", flush=True)
db.close()
