"""`sufr create` on the 3.1 Gb stand-in, several runs, with the CLI's own phase line and the caller's clock around the process
(python profiles/e2e_repeat.py [runs] [bases]): where the wall time goes, start and exit included."""
import os, re, subprocess, sys, tempfile, time, shutil
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import sufr_amd
from sufr_amd import synth
import bench

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bases = int(sys.argv[2]) if len(sys.argv) > 2 else 3_100_000_000
text, starts = synth.syn_human(bases, seed=4, device="cuda")
raw = text.cpu().numpy(); del text
tmp = Path(tempfile.mkdtemp(prefix="sufr_e2e_", dir=os.environ.get("TMPDIR", "/tmp")))
fa, out = tmp / "in.fa", tmp / "out.sufr"
bench.write_fasta(fa, raw, starts)
del raw
os.sync()
cmd = [str(sufr_amd.CLI_PATH), "--log", "debug", "create", "--dna", "--ignore-softmask", "-n", "256", "-o", str(out), str(fa)]
for r in range(runs):
    if out.exists(): out.unlink()
    t0e = time.time(); t0 = time.perf_counter()
    p = subprocess.run(cmd, capture_output=True, text=True)
    dt = time.perf_counter() - t0; t1e = time.time()
    ph = [l for l in p.stdout.splitlines() if "host phases" in l]
    ep = [l for l in p.stdout.splitlines() if "epoch:" in l]
    extra = ""
    if ep:
        m = re.search(r"main ([0-9.]+) exit ([0-9.]+)", ep[-1])
        extra = f" | exec->main {float(m.group(1)) - t0e:.3f}s, exit->reaped {t1e - float(m.group(2)):.3f}s"
    print(f"run {r}: wall {dt:.3f}s rc {p.returncode} | {ph[-1].split('host phases: ')[-1] if ph else p.stderr[-200:]}{extra}", flush=True)
shutil.rmtree(tmp, ignore_errors=True)
