#!/bin/bash
# HBM traffic of the update kernels (FETCH_SIZE and WRITE_SIZE passes of their own) for three shapes:
#   bash tools/update_traffic.sh <tag>        -> gpurun_out/<tag>/update_traffic.jsonl
set -e -o pipefail
TAG=${1:-upd}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
: > "$O/update_traffic.jsonl"
run() {   # name, kernel name part, algorithmic bytes, step_breakdown arguments
  timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/f_$1" -- python3 "$R/tools/step_breakdown.py" $4 $5 > "$O/f_$1.log" 2>&1
  timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/w_$1" -- python3 "$R/tools/step_breakdown.py" $4 $5 > "$O/w_$1.log" 2>&1
  python3 "$R/tools/summarize_pmc.py" hbm_of "$2" "$3" "$O/f_$1" "$O/w_$1" >> "$O/update_traffic.jsonl"
  rm -rf "$O/f_$1" "$O/w_$1"
  echo "$1 done"
}
# noise H*7*K*4 (+ weights 8 K per step from L2 / returns 4 K)
run k4096   "update_rows_kernel<float, 7, 2>"      7602176     4096
run k65536  "update_chunk_sums_kernel<float, 7>"   117964800   65536
run p1024   "update_steps_kernel<float, 7>"        943718400   512 1024
