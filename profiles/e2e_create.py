#!/usr/bin/env python3
"""End-to-end numbers that complement bench.py (which times the device-resident hot path only):
  * `sufr create` wall time of the native CLI on a synthetic FASTA (FASTA parse + H2D + build + D2H +
    .sufr write), with the phase lines it logs;
  * the host-buffer C ABI (sufr_hip_build_u32: pageable host text in, SA/LCP/normalised text out), i.e. the
    PCIe-inclusive rate of the drop-in boundary.
Usage: python profiles/e2e_create.py [elegans|human] > gpurun_out/e2e_<workload>.json"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import sufr_amd
from sufr_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "elegans"
gen, bases, flags, parts = {
    "elegans": (synth.syn_elegans, 100_286_401, dict(is_dna=True), 64),
    "human": (synth.syn_human, 3_100_000_000, dict(is_dna=True, ignore_softmask=True), 256),
}[wl]
text, starts = gen(bases, device="cuda")
raw = text.cpu().numpy()
del text
torch.cuda.empty_cache()
n = raw.size
out = {"workload": wl, "text_len": int(n)}

# ---- host-buffer ABI ---------------------------------------------------------------------------
ctx = sufr_amd.Context(0)
args = sufr_amd.SufrBuilderArgs(text=raw, num_partitions=parts, **flags)
for rep in range(2):
    t0 = time.perf_counter()
    b = sufr_amd.SufrBuilder(args, ctx=ctx, write=False)
    dt = time.perf_counter() - t0
out["host_abi"] = {"seconds": dt, "num_suffixes": int(b.num_suffixes), "suffixes_per_s": b.num_suffixes / dt,
                   "device_ms": b.stats.ms_total,
                   "note": "second call (workspace warm); pageable host buffers: H2D n bytes, D2H n + 8 s bytes"}
s = int(b.num_suffixes)
del b
ctx.close()

# ---- sufr create through the native CLI ----------------------------------------------------------
if wl == "elegans" or os.environ.get("SUFR_E2E_CLI_HUMAN"):
    tmp = Path(os.environ.get("TMPDIR", "/tmp"))
    fa = tmp / f"syn_{wl}.fa"
    seqs = []
    body = raw[:-1]
    cuts = list(starts) + [body.size + 1]
    with open(fa, "wb") as f:
        for i in range(len(starts)):
            seq = body[cuts[i]:cuts[i + 1] - 1]
            f.write(f">seq{i + 1} synthetic\n".encode())
            full = (seq.size // 60) * 60
            if full:
                lines = np.concatenate([seq[:full].reshape(-1, 60), np.full((full // 60, 1), 10, dtype=np.uint8)], axis=1)
                f.write(lines.tobytes())
            if seq.size > full:
                f.write(seq[full:].tobytes() + b"\n")
    cmd = [str(sufr_amd.CLI_PATH), "--log", "debug", "create", "--dna", "-n", str(parts), "-o", str(tmp / f"syn_{wl}.sufr"), str(fa)]
    if flags.get("ignore_softmask"):
        cmd.insert(5, "--ignore-softmask")
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    out["sufr_create"] = {"seconds": dt, "suffixes_per_s": s / dt, "returncode": r.returncode,
                          "fasta_bytes": fa.stat().st_size,
                          "sufr_bytes": (tmp / f"syn_{wl}.sufr").stat().st_size if r.returncode == 0 else 0,
                          "log": r.stdout.strip().splitlines()[-5:], "stderr": r.stderr[-300:]}
    for p in (fa, tmp / f"syn_{wl}.sufr"):
        try:
            p.unlink()
        except OSError:
            pass
print(json.dumps(out))
