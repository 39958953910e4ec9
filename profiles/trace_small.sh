#!/bin/bash
# gpurun -- 'bash profiles/trace_small.sh [workload]': the launch sequence of ONE build of a small workload with start times and gaps
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; W=${1:-ecoli}
mkdir -p $R/gpurun_out
rm -rf /tmp/prof_ts; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ts -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-search --no-verify --placement-trials 1 > /tmp/prof_ts.log 2>&1
tail -1 /tmp/prof_ts.log | cut -c1-300
python3 - <<'PY' | tee $R/gpurun_out/trace_small.txt
import csv, glob
f = glob.glob("/tmp/prof_ts/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sufr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_text_pass_dna" in r["Kernel_Name"] or "k_normalize" in r["Kernel_Name"]]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
print(f"{b - a} launches in one build; columns: start us, gap before us, duration us, kernel")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f} {(e - s) / 1e3:7.1f}  {r['Kernel_Name'].split('(')[0][:70]}  grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))}")
    prev_end = e
print(f"build span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
PY
