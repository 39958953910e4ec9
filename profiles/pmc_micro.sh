#!/bin/bash
# Counters of the stand-alone store-pattern probe (profiles/micro/scatter_write): bash profiles/pmc_micro.sh
# WRITE_SIZE / FETCH_SIZE (KB) and L2 hits / misses per dispatch of k_scatter (direct pattern) and k_combine (XCD-local combining)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_micro; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for C in "WRITE_SIZE FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1)); rm -rf /tmp/prof_mic$i
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_mic$i -- $R/profiles/micro/scatter_write 300000000 > /tmp/prof_mic$i.log 2>&1   # (a fifth of the records: counter collection serialises every dispatch)
done
python3 - <<'PY' | tee $OUT/pmc_scatter_write.txt
import csv, glob
from collections import defaultdict
rows = defaultdict(dict)
for f in glob.glob("/tmp/prof_mic*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))][r["Counter_Name"]] = float(r["Counter_Value"])
print("# per dispatch of profiles/micro/scatter_write (300 M records of 12 bytes = 3.6 GB per dispatch; sizes in GB = counter KB * 1024 / 1e9; FETCH_SIZE not doubled)")
for (d, k), c in sorted(rows.items()):
    g = lambda n: c.get(n, float("nan"))
    print(f"{d:4d} {k:24s} WRITE {g('WRITE_SIZE') * 1024 / 1e9:7.2f} GB  FETCH {g('FETCH_SIZE') * 1024 / 1e9:7.2f} GB  L2 hit {g('TCC_HIT_sum') / 1e6:9.1f} M  miss {g('TCC_MISS_sum') / 1e6:9.1f} M  "
          f"EA write requests {g('TCC_EA0_WRREQ_sum') / 1e6:8.1f} M ({g('TCC_EA0_WRREQ_64B_sum') / 1e6:8.1f} M of 64 bytes)")
PY
