#!/bin/bash
# Counter passes only (kernel-trace + --pmc, nothing else): bash profiles/pmc.sh <tag> [human|elegans]
set -u
TAG=${1:-p}; WL=${2:-human}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/prof_pmc$i
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_pmc$i -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-verify --placement-trials 1 > /tmp/prof_pmc$i.log 2>&1
done
python3 $R/profiles/summarize_pmc.py $OUT/pmc_$WL.csv /tmp/prof_pmc*/
