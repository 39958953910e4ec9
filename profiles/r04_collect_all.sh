#!/bin/bash
# Everything round 4 quotes, collected in one gpurun call on the FINAL code (summaries only; copy what is kept into profiles/):
#   bash profiles/r04_collect_all.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_final; mkdir -p $O
cd $R
bash profiles/collect.sh r04 human   > $O/collect_human.log 2>&1
bash profiles/collect.sh r04 elegans > $O/collect_elegans.log 2>&1
bash profiles/collect.sh r04 ecoli   > $O/collect_ecoli.log 2>&1
# the write side of the two scatter kernels, two passes of the same counters (do they agree?) + the request counters
bash profiles/pmc_custom.sh r04ea human "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" > $O/pmc_ea1.log 2>&1
cp $R/gpurun_out/pmc_r04ea/pmc_human.csv $O/pmc_ea_pass1_human.csv
bash profiles/pmc_custom.sh r04ea human "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" > $O/pmc_ea2.log 2>&1
cp $R/gpurun_out/pmc_r04ea/pmc_human.csv $O/pmc_ea_pass2_human.csv
python3 profiles/shard_probe.py > $O/shard_probe.txt 2>&1
bash profiles/pmc_shard.sh r04s8 8 > $O/pmc_shard8.log 2>&1; cp $R/gpurun_out/pmc_r04s8/pmc_shard8.csv $O/ 2>/dev/null
SUFR_AMD_PROBES_LIB=1 SUFR_HIP_DEBUG=1 SUFR_SHARDS=1,8 SUFR_SHARD_ENDS=1 python3 profiles/shard_probe.py > $O/shard_stamps.txt 2>&1
python3 profiles/input_classes.py > $O/input_classes.txt 2>&1
python3 profiles/modes_bench.py > $O/modes.txt 2>&1
bash profiles/kt_class.sh human_dna > $O/kt_human_dna.txt 2>&1
python3 bench.py > $O/bench_default_human.json 2> $O/bench_default_human.err
ls -la $O
