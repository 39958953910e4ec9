cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_ov; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ov -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-search --no-verify --placement-trials 1 > /tmp/prof_ov.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_ov/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sufr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_text_pass_dna" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 300000:
        print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms  {(e - s) / 1e6:7.3f}  {r['Kernel_Name'].split('(')[0][:60]}  queue {r.get('Queue_Id','')}")
PY
