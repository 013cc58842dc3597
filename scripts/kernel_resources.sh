#!/bin/bash
# Register / LDS / scratch / spill figures of every kernel whose name matches $1 (default: the move
# kernels), from the compiler's own remarks.  Usage: scripts/kernel_resources.sh [pattern] [extra flags]
PAT=${1:-k_move_eval_wave}
shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude "$@" \
    -Rpass-analysis=kernel-resource-usage -c metropolismontecarlo_amd/csrc/mmc_hip.hip -o /tmp/mmc_res.o 2>&1 |
  grep -A12 "Function Name: .*$PAT" | grep -E "Function Name|VGPRs:|SGPRs:|Spill|ScratchSize|Occupancy|LDS Size" |
  sed -e 's/.*remark: [^ ]* *//'
