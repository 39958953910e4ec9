#!/bin/bash
# gpurun -- 'bash profiles/e2e_threads.sh': `sufr create` of the 3.1 Gb stand-in with different numbers of writer threads
# (the probes build of the library under the binary's name in a scratch directory)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
D=/tmp/e2e_bin; rm -rf $D; mkdir -p $D
cp sufr_amd/csrc/_build/sufr $D/sufr; cp sufr_amd/csrc/_build/libsufr_hip_probes.so $D/libsufr_hip.so
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from sufr_amd import synth
import bench
x, starts = synth.syn_human(3_100_000_000, seed=4, device="cuda")
from pathlib import Path; bench.write_fasta(Path("/tmp/e2e_human.fa"), x.cpu().numpy(), starts)
print("fasta written", os.path.getsize("/tmp/e2e_human.fa"))
PY
python3 - <<'PY2'
import os, subprocess, time
for t in (12, 4, 8, 24, 48):
    for rep in range(2):
        env = dict(os.environ, SUFR_PROBE_WRITE_THREADS=str(t))
        t0 = time.perf_counter()
        r = subprocess.run(["/tmp/e2e_bin/sufr", "--log", "debug", "create", "--dna", "--ignore-softmask", "-n", "256", "-o", "/tmp/e2e_out.sufr",
                            "/tmp/e2e_human.fa"], env=env, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        ph = [l for l in (r.stdout + r.stderr).splitlines() if "host phases" in l]
        print(f"writer threads {t}: {dt:.2f} s wall, rc {r.returncode}; {ph[-1] if ph else r.stderr[-200:]}", flush=True)
        os.unlink("/tmp/e2e_out.sufr")
PY2
