#!/bin/bash
# runs tools/f64_inline_check.py for every build/libmppi_f64_*.so (on the GPU box)
for f in build/libmppi_f64_*.so; do
    MPPI_CHECK_F64_ONLY=1 MPPI_HIP_LIBRARY=$PWD/$f timeout -k 10 100 python tools/f64_inline_check.py "$@" 2>&1 | grep "f64:" || exit 1
done
