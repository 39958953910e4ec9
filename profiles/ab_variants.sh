#!/bin/bash
# gpurun -- 'bash profiles/ab_variants.sh "<classes>" <variant.so> ...': the same input classes under several builds of the
# library in ONE session (each variant is copied over the probes library, which SUFR_AMD_PROBES_LIB=1 makes the binding load;
# "base" = the shipped library).  Variants are plain production builds with one compile-time difference.
R=${GRAFT_REPO_ROOT:-/root/repo}; B=$R/sufr_amd/csrc/_build
CLASSES=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then cp $B/libsufr_hip.so $B/libsufr_hip_probes.so; else cp $B/var/$v $B/libsufr_hip_probes.so; fi
  for rep in 1 2; do
    echo "== $v (run $rep)"
    SUFR_AMD_PROBES_LIB=1 python3 $R/profiles/input_classes.py $CLASSES 2>&1 | grep -v "PROBES\|amdgpu.ids"
  done
done
