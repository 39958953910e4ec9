cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_kt -- python3 $R/profiles/input_classes.py human_amb > /tmp/prof_kt.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_finish" in r["Kernel_Name"]]
rows.sort(key=lambda r: -(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for r in rows[:12]:
    print(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2), "ms", r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")))
PY
