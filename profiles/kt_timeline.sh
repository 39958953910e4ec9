#!/bin/bash
# gpurun -- 'bash profiles/kt_timeline.sh <class> [min_us]': the dispatches of one build of an input class in launch
# order (kernels above min_us microseconds), to see which level of the deep path the time goes to
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; C=${1:-human_dna}; MIN=${2:-500}
mkdir -p $R/gpurun_out
# (class "bench:<workload>" traces bench.py --workload <workload> instead)
if [ "${C#bench:}" != "$C" ]; then
  CMD="python3 $R/bench.py --workload ${C#bench:} --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify --placement-trials 1"
else
  CMD="python3 $R/profiles/input_classes.py $C"
fi
rm -rf /tmp/prof_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- $CMD > /tmp/prof_tl.log 2>&1
grep "^$C" /tmp/prof_tl.log
python3 - $MIN <<'PY' | tee $R/gpurun_out/kt_timeline.txt
import csv, glob, sys
mn = float(sys.argv[1])
f = glob.glob("/tmp/prof_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sufr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last build: from the last k_text_pass / first kernel of a build on
starts = [i for i, r in enumerate(rows) if "k_text_pass" in r["Kernel_Name"] or "k_normalize" in r["Kernel_Name"]]
rows = rows[starts[-1]:] if starts else rows
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d >= mn:
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "")
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:9.2f} ms  {d / 1e3:8.3f} ms  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?")):>10}  {nm[:70]}')
PY
