#!/bin/bash
# Where the wall time of `sufr create` outside main() goes: process start (loading the HIP runtime), HIP initialisation and
# exit, on an input where the build itself is nothing.   bash profiles/e2e_startup.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 - "$R" <<'PY'
import subprocess, sys, time
R = sys.argv[1]
S = R + "/sufr_amd/csrc/_build/sufr"
def t(cmd):
    a = time.perf_counter(); r = subprocess.run(cmd, capture_output=True, text=True); b = time.perf_counter()
    return b - a, r.stdout
for i in range(3):
    print("%.3f s  sufr --help" % t([S, "--help"])[0])
for i in range(3):
    dt, out = t([S, "--log", "debug", "create", "--dna", "-o", "/tmp/e2e_tiny.sufr", R + "/tests/golden/inputs/2.fa"])
    print("%.3f s  sufr create 2.fa | %s" % (dt, out.strip().splitlines()[-1]))
PY
