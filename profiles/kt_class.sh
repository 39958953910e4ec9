#!/bin/bash
# gpurun -- 'bash profiles/kt_class.sh <class>': per-kernel totals of one input class (second build included)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; C=${1:-human_dna}
mkdir -p $R/gpurun_out
rm -rf /tmp/prof_kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $R/profiles/input_classes.py $C > /tmp/prof_kt.log 2>&1
grep "^$C" /tmp/prof_kt.log
python3 - <<'PY' | tee $R/gpurun_out/kt_class.txt
import csv, glob
f = glob.glob("/tmp/prof_kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"]) / 2e6:9.2f} ms/build {int(r["Calls"]) // 2:6d} calls  {r["Name"][:110]}')
PY
