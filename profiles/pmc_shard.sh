#!/bin/bash
# Counter passes over the partition kernel of a SHARDED build (what one rank of an N-GPU job runs):
#   bash profiles/pmc_shard.sh <tag> <shards> ["CTR1 CTR2 .." ...]      (knobs / SUFR_AMD_PROBES_LIB from the environment)
set -u
TAG=$1; SH=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export SUFR_SHARDS=$SH SUFR_SHARD_ENDS=1 SUFR_SHARD_REPS=1
i=0; rm -rf /tmp/prof_pmcs*
if [ $# -eq 0 ]; then
  set -- "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" \
         "GRBM_GUI_ACTIVE WRITE_SIZE"
fi
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_pmcs$i -- python3 $R/profiles/shard_probe.py > /tmp/prof_pmcs$i.log 2>&1
  tail -2 /tmp/prof_pmcs$i.log
done
python3 $R/profiles/summarize_pmc.py $OUT/pmc_shard$SH.csv /tmp/prof_pmcs*/
grep "part_text" $OUT/pmc_shard$SH.csv
