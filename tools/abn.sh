#!/bin/bash
# Same-box comparison of several builds of the library: bash tools/abn.sh <lib.so> [<lib.so> ...]
# two rounds of the rollout kernel at K = 4096 (tools/ksweep.py) per library.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
  for L in "$@"; do
    printf "%s " "$(basename $L)"; MPPI_HIP_LIBRARY=$L timeout -k 10 90 python3 "$R/tools/ksweep.py" ${KS:-4096} 2> /dev/null | grep -o '"rollout_ms": [0-9.]*' || exit 1
  done
done
