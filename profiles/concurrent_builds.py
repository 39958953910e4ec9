"""How much do two builds overlap when they run at the same time on ONE GPU (two contexts, two host threads, the contexts'
own streams)?  The build is a chain of kernels that are each bound by something else (text pass and leaf sort: vector ALU;
partition and level-2 scatter: HBM stores; deep levels: latency) -- if two chains in different phases share the chip well, a
build that pipelines its own level-2 scatter against its leaf sort (chunks of first-digit buckets on two streams) can expect
a similar gain.  python profiles/concurrent_builds.py [bases]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sufr_amd
from sufr_amd import synth

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
x, _ = synth.syn_human(bases, seed=4, device="cuda")
flags = dict(is_dna=True, ignore_softmask=True, raw_text=True, num_partitions=256)
B = [sufr_amd.DeviceBuilder(0) for _ in range(2)]
outs = []
for b in B:
    sa, lcp = b.sort(x, **flags)
    cap = int(b.num_suffixes * 1.02) + 1024
    del sa, lcp
    torch.cuda.empty_cache()
    outs.append((torch.empty(cap, dtype=torch.int32, device="cuda"), torch.empty(cap, dtype=torch.int32, device="cuda")))
for b, o in zip(B, outs):
    for _ in range(2):
        b.sort(x, out_sa=o[0], out_lcp=o[1], **flags)
K = 6
def run(b, o, k, delay=0.0):
    time.sleep(delay)
    for _ in range(k):
        b.sort(x, out_sa=o[0], out_lcp=o[1], **flags)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(B[0], outs[0], K); t1 = time.perf_counter()
print(f"one build at a time : {(t1 - t0) / K * 1e3:7.2f} ms per build (device {B[0].stats.ms_total:.2f})", flush=True)
for delay in (0.0, 0.013, 0.027):
    th = [threading.Thread(target=run, args=(B[i], outs[i], K, delay * i)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0 - delay
    print(f"two at a time (second one {delay * 1e3:4.0f} ms behind): {dt / (2 * K) * 1e3:7.2f} ms per build  ({dt * 1e3:.1f} ms for {2 * K} builds; device {B[0].stats.ms_total:.2f} / {B[1].stats.ms_total:.2f} per build)", flush=True)
