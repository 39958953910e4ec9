#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box and keeps only compact summaries
# (the raw traces exceed what gpurun copies back).  Usage (from the repo root, via gpurun):
#     bash profiles/collect.sh r01 [human|elegans]
# Outputs gpurun_out/profiles_<round>/: kernel_stats_<workload>.csv (from --kernel-trace --stats),
# pmc_<workload>.csv (per-kernel counter means, one --pmc pass per counter group, no trace domains mixed
# in) and bench_<workload>.json (the same command un-profiled).
set -u
ROUND=${1:-r01}; WL=${2:-human}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$ROUND; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify --placement-trials 1"
$CMD > $OUT/bench_$WL.json 2> /dev/null
rm -rf /tmp/prof_kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- $CMD > /tmp/prof_kt.log 2>&1
cp $(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$WL.csv
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/prof_pmc$i
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_pmc$i -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-verify --placement-trials 1 > /tmp/prof_pmc$i.log 2>&1
done
python3 $R/profiles/summarize_pmc.py $OUT/pmc_$WL.csv /tmp/prof_pmc*/
ls -la $OUT
