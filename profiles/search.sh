#!/bin/bash
# gpurun -- 'bash profiles/search.sh': query tests, the search rate (100 Mb and 3.1 Gb indexes), kernel trace and
# counters of k_search_batch on the 3.1 Gb index (counter passes apart from the trace, as the guide prescribes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
OUT=$R/gpurun_out/search; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_query.py -x -q 2>&1 | tail -5 > $OUT/tests.txt
timeout 600 python profiles/search_bench.py 1e8 1e7 32 2>&1 | grep -v amdgpu.ids > $OUT/bench_100m.txt
SEARCH_NO_TABLE=1 timeout 900 python profiles/search_bench.py 3.1e9 1.6e7 32 2>&1 | grep "^device" > $OUT/bench_3g_no_table.txt
cd /tmp
rm -rf /tmp/prof_s; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python3 $R/profiles/search_bench.py 3.1e9 1.6e7 32 > $OUT/bench_3g.txt 2>&1
grep "^device\|^host" $OUT/bench_3g.txt > $OUT/bench_3g.tmp; mv $OUT/bench_3g.tmp $OUT/bench_3g.txt
python3 - <<PY > $OUT/kernel_stats_search.txt
import csv, glob
f = glob.glob("/tmp/prof_s/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_search_batch" in r["Name"]:
        print(r["Name"][:60], "calls", r["Calls"], "avg ms", float(r["AverageNs"]) / 1e6, "min ms", float(r["MinNs"]) / 1e6, "max ms", float(r["MaxNs"]) / 1e6)
PY
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
         "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1)); rm -rf /tmp/prof_spmc$i
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_spmc$i -- python3 $R/profiles/search_bench.py 3.1e9 1.6e7 32 > /tmp/prof_spmc$i.log 2>&1
done
python3 $R/profiles/summarize_pmc.py $OUT/pmc_search.csv /tmp/prof_spmc*/
grep k_search_batch $OUT/pmc_search.csv > $OUT/pmc_search_kernel.csv
cat $OUT/tests.txt $OUT/bench_100m.txt $OUT/bench_3g_no_table.txt $OUT/bench_3g.txt $OUT/kernel_stats_search.txt $OUT/pmc_search_kernel.csv
