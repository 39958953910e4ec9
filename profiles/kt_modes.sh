#!/bin/bash
# gpurun -- 'bash profiles/kt_modes.sh "<case substring>"': per-kernel totals of one case of profiles/modes_bench.py
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; C=${1:-elegans -m}
mkdir -p $R/gpurun_out
rm -rf /tmp/prof_km; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_km -- python3 $R/profiles/modes_bench.py "$C" > /tmp/prof_km.log 2>&1
grep -v amdgpu /tmp/prof_km.log | tail -2
python3 - <<'PY' | tee $R/gpurun_out/kt_modes.txt
import csv, glob
f = glob.glob("/tmp/prof_km/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sufr::" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    print(f'{float(r["TotalDurationNs"]) / 2e6:9.3f} ms/build {int(r["Calls"]) // 2:6d} calls  {r["Name"].split("(")[0][:90]}')
PY
