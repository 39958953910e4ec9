#!/bin/bash
# One counter pass with the counters given: bash profiles/pmc_custom.sh <tag> <workload> "CTR1 CTR2 ..." ["CTR .." ...]
set -u
TAG=$1; WL=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0; rm -rf /tmp/prof_pmcc*
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_pmcc$i -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-verify --placement-trials 1 > /tmp/prof_pmcc$i.log 2>&1
done
python3 $R/profiles/summarize_pmc.py $OUT/pmc_$WL.csv /tmp/prof_pmcc*/
