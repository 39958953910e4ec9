#!/bin/bash
# Kernel-time table only (no counters): bash profiles/quick.sh <tag> [human|elegans|ecoli] ["SUFR_PROBE_X=.. .."]
# (a third argument runs the probes build with those knobs)
set -u
TAG=${1:-q}; WL=${2:-human}; KNOBS=${3:-}
if [ -n "$KNOBS" ]; then export SUFR_AMD_PROBES_LIB=1; for kv in $KNOBS; do export "$kv"; done; fi
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/quick_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify --placement-trials 1"
$CMD > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
rm -rf /tmp/prof_kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- $CMD > /tmp/prof_kt.log 2>&1
cp $(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$WL.csv
python3 - "$OUT/kernel_stats_$WL.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sufr::" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"sufr kernels: {tot / 4e6:.2f} ms per build (4 builds in the trace)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print(f'{float(r["TotalDurationNs"]) / 4e6:8.3f} ms/build  {int(r["Calls"]) // 4:5d} calls  {float(r["AverageNs"]) / 1e3:10.1f} us avg  {r["Name"].split("(")[0][:90]}')
PY
python3 -c "
import json,sys
d=json.load(open('$OUT/bench_$WL.json')); print(json.dumps({k:d[k] for k in ('value','ms_per_step','device_ms')})); print(d['roofline'])"
