#!/bin/bash
# Bisection of "-ftrivial-auto-var-init=zero makes the inlined fp64 kernel right": builds the model's kernels with
# zero-initialisation stopped after N automatic variables, for every N given.  The first N whose build is right
# names the variable (see tools/f64_inline_check.py).  Usage: tools/f64_bisect_build.sh MODEL N1 N2 ...
set -e
MODEL=$1; shift
cd "$(dirname "$0")/../trajopt_amd/csrc"
mkdir -p ../../build obj/study
FE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-variable -Wno-unused-function"
[ -f obj/study/abi_bisect_$MODEL.o ] || hipcc $FE -DMPPI_ONLY_MODEL=$MODEL -c -o obj/study/abi_bisect_$MODEL.o mppi_hip.hip
rm -f ../../build/libmppi_f64_*.so
for N in "$@"; do
    ( hipcc $FE -ftrivial-auto-var-init=zero -ftrivial-auto-var-init-stop-after=$N -DMPPI_MODEL=$MODEL -c -o obj/study/ops_n$N.o model_ops.hip 2>/dev/null &&
      hipcc --offload-arch=gfx950 -shared -o ../../build/libmppi_f64_n$N.so obj/study/abi_bisect_$MODEL.o obj/study/ops_n$N.o ) &
done
wait
ls ../../build/libmppi_f64_*.so
