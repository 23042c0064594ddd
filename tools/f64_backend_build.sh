#!/bin/bash
# Companion of f64_inline_build.sh: the same source and front-end flags (so the same LLVM IR) compiled with different
# BACKEND options only.  If correctness differs between these, the wrong fp64 values are made by code generation
# (scheduling / register spilling), not by the source.  Usage: tools/f64_backend_build.sh [MODEL struct]
set -e
MODEL=${1:-Tree18DofQuat}
cd "$(dirname "$0")/../trajopt_amd/csrc"
mkdir -p ../../build obj/study
FE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -Wno-unused-variable -Wno-unused-function"
build() {   # name, backend flags
    ( hipcc $FE $2 -DMPPI_MODEL=$MODEL -c -o obj/study/ops_$1.o model_ops.hip &&
      hipcc $FE $2 -DMPPI_ONLY_MODEL=$MODEL -c -o obj/study/abi_$1.o mppi_hip.hip &&
      hipcc --offload-arch=gfx950 -shared -o ../../build/libmppi_f64_$1.so obj/study/abi_$1.o obj/study/ops_$1.o ) &
}
rm -f ../../build/libmppi_f64_*.so
build be_maxilp     "-mllvm -amdgpu-sched-strategy=max-ilp"
build be_default    ""
build be_nosgpr2vgpr "-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-spill-sgpr-to-vgpr=false"
build be_noagprspill "-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-spill-vgpr-to-agpr=false"
build be_maxocc     "-mllvm -amdgpu-sched-strategy=max-occupancy"
wait
ls -la ../../build/libmppi_f64_*.so
