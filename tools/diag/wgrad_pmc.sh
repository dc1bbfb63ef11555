# SQ counters of the rank-one weight-gradient kernel (kernel_microbench --only mlp_wgrad_gate_bits), one set per pass.
R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/wg_pmc/p$i -o p -- python3 $R/tools/kernel_microbench.py --rounds 1 --only mlp_wgrad_gate_bits,mlp_tower_forward_gate_only_f16,mlp_tower_backward_gate_f16 > $R/gpurun_out/wg_pmc_$i.log 2>&1
  echo "set $i ($set) rc=$?"
done
