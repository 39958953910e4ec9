#!/bin/bash
# ms per build of the three workloads (no CPU baseline, no end-to-end leg): gpurun -- 'bash profiles/quick3.sh [steps]'
S=${1:-20}
for wl in ecoli elegans human; do
  python3 bench.py --workload $wl --steps $S --warmup 3 --no-cpu-baseline --no-e2e --no-search --placement-trials 1 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$wl', round(d['ms_per_step'],4), 'ms', round(d['value']/1e9,3), 'G/s', {k: round(v,3) for k,v in d['device_ms'].items()})"
done
