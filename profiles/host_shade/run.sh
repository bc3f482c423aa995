#!/bin/bash
# bash profiles/host_shade/run.sh  — the shading headers as host C++ under MSan, then under ASan + UBSan (CPU only)
set -e
H=$(cd "$(dirname "$0")" && pwd); K=$H/../../crust-render_amd/csrc/kernels; T=${TMPDIR:-/tmp}/crt_host_shade; mkdir -p $T
CXX=/opt/rocm/lib/llvm/bin/clang++
F="-std=c++17 -O1 -g -ffp-contract=off -fno-fast-math -I$H -I$K -Wno-unused-function -Wno-unknown-attributes"
$CXX $F -fsanitize=memory -fsanitize-memory-track-origins=2 -fno-omit-frame-pointer $H/shade_host.cpp -o $T/shade_msan
$T/shade_msan
$CXX $F -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer $H/shade_host.cpp -o $T/shade_ubsan
$T/shade_ubsan
