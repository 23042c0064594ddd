#!/bin/bash
# LLVM pass bisection of the wrong fp64 values: builds the model's kernels with -opt-bisect-limit=N for every N
# given (all optimisation passes after the N-th are skipped).  Usage: tools/f64_optbisect_build.sh MODEL N1 N2 ...
set -e
MODEL=$1; shift
cd "$(dirname "$0")/../trajopt_amd/csrc"
mkdir -p ../../build obj/study
FE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-variable -Wno-unused-function"
[ -f obj/study/abi_bisect_$MODEL.o ] || hipcc $FE -DMPPI_ONLY_MODEL=$MODEL -c -o obj/study/abi_bisect_$MODEL.o mppi_hip.hip
rm -f ../../build/libmppi_f64_*.so
for N in "$@"; do
    ( hipcc $FE -mllvm -opt-bisect-limit=$N -DMPPI_MODEL=$MODEL -c -o obj/study/ops_o$N.o model_ops.hip 2>/dev/null &&
      hipcc --offload-arch=gfx950 -shared -o ../../build/libmppi_f64_o$N.so obj/study/abi_bisect_$MODEL.o obj/study/ops_o$N.o ) &
done
wait
ls ../../build/libmppi_f64_*.so
