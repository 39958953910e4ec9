#!/bin/bash
# gpurun -- 'bash profiles/wide.sh': kernel totals of the windowed build of a 4.4e9-byte text
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/wide
rm -rf /tmp/prof_w; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -- python3 $R/profiles/wide_bench.py > $R/gpurun_out/wide/bench.txt 2>&1
grep "^n=" $R/gpurun_out/wide/bench.txt
python3 - <<'PY' | tee $R/gpurun_out/wide/kernels.txt
import csv, glob
f = glob.glob("/tmp/prof_w/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print(f'{float(r["TotalDurationNs"]) / 2e6:9.2f} ms/build {int(r["Calls"]) // 2:6d} calls  {r["Name"][:100]}')
PY
