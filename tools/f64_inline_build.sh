#!/bin/bash
# Study builds of a one-model library, to look for the cause of wrong fp64 values when the fp64 substep is inlined
# into the rollout loop: plain, with automatic variables zero- / pattern-initialised, and without the
# value-assumption flags.  Usage: tools/f64_inline_build.sh [MODEL struct] [extra -D flags]
# Outputs build/libmppi_f64_<variant>.so (run tools/f64_inline_check.py on the GPU).
set -e
MODEL=${1:-Tree18DofQuat}
EXTRA=${2:-}
cd "$(dirname "$0")/../trajopt_amd/csrc"
mkdir -p ../../build obj/study
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-variable -Wno-unused-function $EXTRA"
FM="-fno-signed-zeros -ffinite-math-only"
build() {   # name, extra flags
    ( hipcc $BASE $2 -DMPPI_MODEL=$MODEL -c -o obj/study/ops_$1.o model_ops.hip &&
      hipcc $BASE $2 -DMPPI_ONLY_MODEL=$MODEL -c -o obj/study/abi_$1.o mppi_hip.hip &&
      hipcc --offload-arch=gfx950 -shared -o ../../build/libmppi_f64_$1.so obj/study/abi_$1.o obj/study/ops_$1.o ) &
}
build plain   "$FM"
build zero    "$FM -ftrivial-auto-var-init=zero"
build pattern "$FM -ftrivial-auto-var-init=pattern"
build nofm    ""
build nsz     "-fno-signed-zeros"
build finite  "-ffinite-math-only"
wait
ls -la ../../build/libmppi_f64_*.so
