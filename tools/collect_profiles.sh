#!/bin/bash
# Collects the measurements kept under profiles/ on a one-GPU MI355X box:
#   bash tools/collect_profiles.sh <tag>        -> gpurun_out/<tag>/...
# Every GPU step has its own time limit and the script stops at the first failure.
set -e -o pipefail
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp

timeout -k 10 300 python3 "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err"
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$R/bench.py" --no-cpu-baseline \
    > "$O/bench_under_rocprof.json" 2> "$O/trace.err"
cp "$(find "$O/trace" -name '*kernel_stats.csv' | head -n 1)" "$O/kernel_stats.csv"
echo "kernel trace done"

# counters: one pass each, nothing else traced
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
    SQ_WAIT_ANY SQ_INSTS_BRANCH --output-format csv -d "$O/pmc_sq" -- python3 "$R/tools/ksweep.py" 4096 > "$O/pmc_sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/tools/ksweep.py" 4096 > "$O/pmc_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/tools/ksweep.py" 4096 > "$O/pmc_write.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 \
    --output-format csv -d "$O/pmc_flop" -- python3 "$R/tools/ksweep.py" 4096 > "$O/pmc_flop.log" 2>&1
python3 "$R/tools/summarize_pmc.py" sq "$O/pmc_sq" > "$O/rollout_pmc_sq.json"
python3 "$R/tools/summarize_pmc.py" hbm "$O/pmc_fetch" "$O/pmc_write" > "$O/rollout_traffic.json"
# (the listing is made beforehand on the build machine: make -C trajopt_amd/csrc asm && cp /tmp/mppi_Reacher7Dof.s build/)
if [ -f "$R/build/mppi_Reacher7Dof.s" ]; then
  (cd "$R/tools" && python3 summarize_flops.py "$O/pmc_flop" "$R/build/mppi_Reacher7Dof.s") > "$O/rollout_flops.json"
fi
echo "counters done"

MPPI_BENCH_SHARDED=1 timeout -k 10 300 python3 "$R/bench.py" --no-cpu-baseline > "$O/bench_sharded_one_rank.json" 2> /dev/null
timeout -k 10 120 python3 "$R/tools/ksweep.py" 4096 8192 16384 32768 65536 131072 262144 524288 > "$O/ksweep.jsonl" 2> /dev/null
timeout -k 10 120 python3 "$R/tools/step_breakdown.py" 4096 > "$O/step_breakdown.txt" 2> /dev/null
timeout -k 10 120 python3 "$R/tools/step_breakdown.py" 8192 >> "$O/step_breakdown.txt" 2> /dev/null
timeout -k 10 120 python3 "$R/tools/step_breakdown.py" 65536 >> "$O/step_breakdown.txt" 2> /dev/null
timeout -k 10 120 python3 "$R/tools/step_breakdown.py" 512 1024 >> "$O/step_breakdown.txt" 2> /dev/null
timeout -k 10 180 python3 "$R/tools/bench_planners.py" > "$O/planners.jsonl" 2> /dev/null
timeout -k 10 180 python3 "$R/tools/time_train_step.py" > "$O/train_step.jsonl" 2> /dev/null
timeout -k 10 180 python3 "$R/tools/host_noise_pieces.py" > "$O/host_noise_pieces.json" 2> /dev/null
timeout -k 10 240 python3 "$R/tools/tree_study.py" "$O/tree_quat_study.json" tree_18dof_quat > /dev/null 2> "$O/tree_quat.err"
timeout -k 10 240 python3 "$R/tools/tree_study.py" "$O/tree_study.json" tree_18dof > /dev/null 2> "$O/tree.err"
timeout -k 10 240 python3 "$R/tools/tree_study.py" "$O/tree_rk4_study.json" tree_18dof_rk4 > /dev/null 2> "$O/tree_rk4.err"
rm -rf "$O/trace" "$O/pmc_sq" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_flop"
echo "all done"
