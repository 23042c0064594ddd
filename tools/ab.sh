#!/bin/bash
# Same-box comparison of two builds of the library: bash tools/ab.sh <old.so> [K ...]
# (NEW=<path> selects another library than the built one as the new side)
# alternates old / new runs of the rollout kernel (tools/ksweep.py) and one bench.py each.
set -e -o pipefail
OLD=$1; shift
KS=${@:-4096}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
  echo "new:"; MPPI_HIP_LIBRARY=${NEW:-$R/trajopt_amd/libmppi_hip.so} timeout -k 10 90 python3 "$R/tools/ksweep.py" $KS 2> /dev/null | cut -c1-60
  echo "old:"; MPPI_HIP_LIBRARY=$OLD timeout -k 10 90 python3 "$R/tools/ksweep.py" $KS 2> /dev/null | cut -c1-60
done
echo "bench new:"; MPPI_HIP_LIBRARY=${NEW:-$R/trajopt_amd/libmppi_hip.so} timeout -k 10 120 python3 "$R/bench.py" --no-cpu-baseline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*'
echo "bench old:"; MPPI_HIP_LIBRARY=$OLD timeout -k 10 120 python3 "$R/bench.py" --no-cpu-baseline 2> /dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*'
