#!/bin/bash
# A/B of library builds on the cfg2 probe: tools/mrhs_ab.sh libA.so libB.so ...  (alternating, 2 rounds)
R=${GRAFT_REPO_ROOT:-/root/repo}
for rnd in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    VARPRO_HIP_LIBRARY=$R/$lib python $R/tools/mrhs_probe.py 2>&1 | grep -E "evaluate|global fit"
  done
done
