#!/bin/bash
# Device time of the bench workload under different SUFR_PROBE_* settings (probes build), one line per setting:
#   bash profiles/knobs.sh human "SUFR_PROBE_T1=11" "SUFR_PROBE_BMAX=256" ...
WL=${1:-human}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  env SUFR_AMD_PROBES_LIB=1 "$@" python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-verify --placement-trials 1 2>/dev/null |
    python3 -c "import json,sys; d=json.load(sys.stdin); print('%-60s' % sys.argv[1], {k: round(v,2) for k,v in d['device_ms'].items()})" "$*"
}
run SUFR_PROBE_NONE=1
for s in "$@"; do run $s; done
