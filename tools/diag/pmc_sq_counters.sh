#!/bin/bash
R=${GRAFT_REPO_ROOT}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$R/gpurun_out/pmc_sq1" -o mb --output-format csv -- python3 "$R/tools/kernel_microbench.py" --rounds 1 > "$R/gpurun_out/pmc_sq1.log" 2>&1 || { echo "pass 1 failed"; tail -5 "$R/gpurun_out/pmc_sq1.log"; exit 1; }
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d "$R/gpurun_out/pmc_sq2" -o mb --output-format csv -- python3 "$R/tools/kernel_microbench.py" --rounds 1 > "$R/gpurun_out/pmc_sq2.log" 2>&1 || { echo "pass 2 failed"; tail -5 "$R/gpurun_out/pmc_sq2.log"; exit 1; }
timeout -k 10 500 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE --kernel-trace -d "$R/gpurun_out/pmc_sq3" -o mb --output-format csv -- python3 "$R/tools/kernel_microbench.py" --rounds 1 > "$R/gpurun_out/pmc_sq3.log" 2>&1 || { echo "pass 3 failed"; tail -5 "$R/gpurun_out/pmc_sq3.log"; exit 1; }
find "$R/gpurun_out/pmc_sq1" "$R/gpurun_out/pmc_sq2" "$R/gpurun_out/pmc_sq3" -name "*counter_collection.csv" | head
echo done
