#!/bin/bash
# Kernel-tuning aid: experimental builds of librl8_amd.so (same ABI) into build_diag/.
#   tools/diag_mlp.sh 64 128 256   one memory stream of a tower kernel compiled out
#                                  (RL8_DIAG_SKIP bits, see mlp_kernels.hip; mlp_split_kernels.hip: 8 h2 stores,
#                                  32 h1 stores, 64 h2 loads, 128 dZ2 stores, 2048 no matrix work, 4096 h2 stores kept in L2), timed with
#       RL8_AMD_LIBRARY=build_diag/librl8_amd_skip<bits>.so python tools/kernel_microbench.py --only mlp_tower_backward_fused
#   tools/diag_mlp.sh trace        phase timestamps compiled in, read with tools/phase_trace.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build_diag
for what in "$@"; do
  if [ "$what" = stamp ]; then
    make -s -C rl8_amd/csrc BUILD="$PWD/build_diag/objstamp" OUT="$PWD/build_diag/librl8_amd_stamp.so" \
         FLAGS_EXTRA="-DRL8_ROWS_STAMP"
  elif [ "${what#lr}" != "$what" ]; then  # lr<bits>: RL8_LR_DIAG of lstm_rows_kernels.hip; lrsafe: vmcnt(0) at every barrier
    if [ "$what" = lrsafe ]; then extra="-DRL8_LR_SAFE_WAITS=1"
    elif [ "$what" = lrstamp ]; then extra="-DRL8_LR_STAMP"
    else extra="-DRL8_LR_DIAG=${what#lr}"; fi
    make -s -C rl8_amd/csrc BUILD="$PWD/build_diag/obj$what" OUT="$PWD/build_diag/librl8_amd_$what.so" FLAGS_EXTRA="$extra"
  elif [ "${what#ls}" != "$what" ]; then  # ls<bits>: RL8_LS_DIAG of lstm_split_kernels.hip (1 no h/c/gate stores, 2 no transcendentals)
    make -s -C rl8_amd/csrc BUILD="$PWD/build_diag/obj$what" OUT="$PWD/build_diag/librl8_amd_$what.so" FLAGS_EXTRA="-DRL8_LS_DIAG=${what#ls}"
  elif [ "$what" = trace ]; then
    make -s -C rl8_amd/csrc BUILD="$PWD/build_diag/objtrace" OUT="$PWD/build_diag/librl8_amd_trace.so" \
         FLAGS_EXTRA="-DRL8_PHASE_TRACE"
  else
    make -s -C rl8_amd/csrc BUILD="$PWD/build_diag/obj$what" OUT="$PWD/build_diag/librl8_amd_skip$what.so" \
         FLAGS_EXTRA="-DRL8_DIAG_SKIP=$what"
  fi
done
