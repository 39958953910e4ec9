#!/bin/bash
# Everything round 5 quotes, collected in one gpurun call on the FINAL code (summaries only; copy what is kept into profiles/):
#   bash profiles/r05_collect_all.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_final; mkdir -p $O
cd $R
bash profiles/collect.sh r05 human   > $O/collect_human.log 2>&1
bash profiles/collect.sh r05 elegans > $O/collect_elegans.log 2>&1
bash profiles/collect.sh r05 ecoli   > $O/collect_ecoli.log 2>&1
cp $R/gpurun_out/profiles_r05/* $O/ 2>/dev/null
python3 profiles/shard_probe.py > $O/shard_probe.txt 2>&1
python3 profiles/input_classes.py > $O/input_classes.txt 2>&1
python3 profiles/modes_bench.py > $O/modes.txt 2>&1
python3 profiles/wide_bench.py > $O/wide_bench.txt 2>&1
./profiles/micro/scatter_write > $O/scatter_write_combine.txt 2>&1
./profiles/micro/level2_pattern > $O/level2_pattern.txt 2>&1
# the N > 1 leg on the one GPU: two gloo ranks sharing the device, and RCCL at world size 1 with two shards per step
python3 bench.py --gpus 2 --backend gloo --share-device --steps 5 --warmup 2 > $O/bench_n2_share_human.json 2> $O/bench_n2_share_human.err
SUFR_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench_forced_dist_human.json 2> $O/bench_forced_dist_human.err
# the driver's line (median of three placements, verified, cpu_baseline on the WHOLE workload when the host allows, e2e_create)
python3 bench.py > $O/bench_default_human.json 2> $O/bench_default_human.err
ls -la $O
