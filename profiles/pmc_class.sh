#!/bin/bash
# HBM counters of one input class (python profiles/input_classes.py <class>: two builds), separate --pmc passes as
# MI355X_MICROARCH.md prescribes:   bash profiles/pmc_class.sh human_dna   -> gpurun_out/pmc_class/pmc_<class>.csv
set -u
C=${1:-human_dna}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_class; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0; rm -rf /tmp/prof_pmcd*
for CTR in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/prof_pmcd$i -- python3 $R/profiles/input_classes.py $C > /tmp/prof_pmcd$i.log 2>&1
done
python3 $R/profiles/summarize_pmc.py $OUT/pmc_$C.csv /tmp/prof_pmcd*/
grep -E "k_gather_keys|k_finish|k_group_sort" $OUT/pmc_$C.csv | grep -E "FETCH_SIZE|WRITE_SIZE|TCC_MISS" | cut -c1-200
