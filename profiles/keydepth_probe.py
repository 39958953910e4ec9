"""What a shorter key costs end to end ("why not 8-byte records"): the same build with the 64-bit key
cut to K characters (probes build, SUFR_PROBE_KEY_CHARS; K = 21 is the shipped key).  An 8-byte record after the first
digit would hold a 32-bit remainder = 10 2/3 characters, i.e. sort to 15 2/3 characters.
    SUFR_AMD_PROBES_LIB=1 SUFR_PROBE_KEY_CHARS=16 python profiles/keydepth_probe.py [check]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sufr_amd
from sufr_amd import synth

K = os.environ.get("SUFR_PROBE_KEY_CHARS", "21")
if len(sys.argv) > 1 and sys.argv[1] == "check":      # the knob must not change the arrays
    x, _ = synth.syn_human(5_000_000, seed=9, device="cuda")
    db = sufr_amd.DeviceBuilder(0)
    sa, lcp = (t.clone() for t in db.sort(x, is_dna=True, ignore_softmask=True, raw_text=True))
    import hashlib
    print(f"K={K} check: sa {hashlib.sha1(sa.cpu().numpy().tobytes()).hexdigest()[:12]} lcp {hashlib.sha1(lcp.cpu().numpy().tobytes()).hexdigest()[:12]}")
    db.close()
    sys.exit(0)
x, _ = synth.syn_human(3_100_000_000, seed=4, device="cuda")
db = sufr_amd.DeviceBuilder(0)
out_sa = torch.empty(x.numel() // 2 + (1 << 20), dtype=torch.int32, device="cuda")
out_lcp = torch.empty_like(out_sa)
for rep in range(2):
    sa, lcp = db.sort(x, is_dna=True, ignore_softmask=True, raw_text=True, out_sa=out_sa, out_lcp=out_lcp, num_partitions=256)
st = db.stats
print(f"key of {K:>2s} characters: levels={st.num_levels:3d} deep_records={st.deep_records:>11d}  total {st.ms_total:8.2f} ms  "
      f"(text {st.ms_normalize:.2f} part {st.ms_partition:.2f} passes {st.ms_passes:.2f} deep {st.ms_deep:.2f})", flush=True)
db.close()
