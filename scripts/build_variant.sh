#!/bin/bash
# Build an experimental variant of the library into build/<name>.so (travels to the GPU box, git-ignored):
#   scripts/build_variant.sh name [-DFLAG ...]        then run with MMC_HIP_LIB=$PWD/build/name.so
cd "$(dirname "$0")/.."
mkdir -p build
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude "$@" \
    -o build/$NAME.so metropolismontecarlo_amd/csrc/mmc_hip.hip -ldl
