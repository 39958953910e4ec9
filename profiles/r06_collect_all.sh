#!/bin/bash
# Everything round 6 quotes, collected in one gpurun call on the FINAL code (summaries only; copy what is kept into profiles/):
#   bash profiles/r06_collect_all.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_final; mkdir -p $O
cd $R
timeout 900 bash profiles/collect.sh r06 human   > $O/collect_human.log 2>&1
timeout 600 bash profiles/collect.sh r06 elegans > $O/collect_elegans.log 2>&1
timeout 600 bash profiles/collect.sh r06 ecoli   > $O/collect_ecoli.log 2>&1
cp $R/gpurun_out/profiles_r06/* $O/ 2>/dev/null
timeout 600 python3 profiles/shard_probe.py > $O/shard_probe.txt 2>&1
timeout 900 python3 profiles/input_classes.py human human_iupac human_hash human_dna human_amb human_amb_soft protein all_a dup100k copies300 > $O/input_classes.txt 2>&1
timeout 900 python3 profiles/modes_bench.py > $O/modes.txt 2>&1
timeout 600 bash profiles/kt_class.sh human_dna > $O/kernel_table_human_dna.txt 2>&1
timeout 600 bash profiles/kt_class.sh human_iupac > $O/kernel_table_human_iupac.txt 2>&1
timeout 300 python3 profiles/host_abi_probe.py 3100000000 3 > $O/host_abi_final.txt 2>&1
# the N > 1 leg on the one GPU: eight gloo ranks sharing the device (C5 at its real shape), and RCCL at world size 1 with two shards per step
timeout 600 python3 bench.py --gpus 8 --backend gloo --share-device --e2e-hash > $O/bench_n8_share_human.json 2> $O/bench_n8_share_human.err
SUFR_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench_forced_dist_human.json 2> $O/bench_forced_dist_human.err
# the driver's line (median of three placements, verified, cpu_baseline on the WHOLE workload when the host allows, host_abi, e2e_create)
timeout 900 python3 bench.py --e2e-hash > $O/bench_default_human.json 2> $O/bench_default_human.err
ls -la $O
