#!/bin/bash
# Kernel-tuning aid: builds variants of librl8_amd.so with one phase of the tower
# forward kernel compiled out (RL8_DIAG_SKIP bits, see mlp_kernels.hip) into
# build_diag/, to be timed with
#   RL8_AMD_LIBRARY=build_diag/librl8_amd_skip<bits>.so python tools/kernel_microbench.py --only mlp_tower_forward
set -e
cd "$(dirname "$0")/.."
mkdir -p build_diag
for bits in "$@"; do
  make -s -C rl8_amd/csrc BUILD="$PWD/build_diag/obj$bits" OUT="$PWD/build_diag/librl8_amd_skip$bits.so" \
       FLAGS_EXTRA="-DRL8_DIAG_SKIP=$bits"
done
