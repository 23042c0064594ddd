#!/bin/bash
# Per-sample step on the host under MemorySanitizer (halves of packed pairs left undefined as on the device) and
# under UBSan + ASan, with the device build's value-assumption flags and pattern-initialised automatic variables.
set -e
cd "$(dirname "$0")/.."
CXX=/opt/rocm/lib/llvm/bin/clang++
FM="-fno-signed-zeros -ffinite-math-only"
mkdir -p build
$CXX -O1 -g -std=c++17 $FM -DMPPI_UNDEF_HALVES -fsanitize=memory -fsanitize-memory-track-origins -fno-omit-frame-pointer -o build/host_msan tools/host_sanitize.cpp
$CXX -O2 -g -std=c++17 $FM -DMPPI_UNDEF_HALVES -fsanitize=undefined,address -fno-sanitize-recover=all -ftrivial-auto-var-init=pattern -Wuninitialized -o build/host_ubsan tools/host_sanitize.cpp
$CXX -O2 -std=c++17 $FM -o build/host_plain tools/host_sanitize.cpp
./build/host_msan > build/host_msan.out
./build/host_ubsan > build/host_ubsan.out
./build/host_plain > build/host_plain.out
echo "msan vs plain:"; diff build/host_msan.out build/host_plain.out | head -5 && echo same
echo "ubsan(pattern) vs plain:"; diff build/host_ubsan.out build/host_plain.out | head -5 && echo same
