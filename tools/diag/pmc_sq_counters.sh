#!/bin/bash
# SQ issue counters of a script's kernels (default: tools/kernel_microbench.py --rounds 1), three rocprofv3 --pmc passes
# (kernel trace only), summarised by tools/diag/summarize_sq_counters.py.
#   gpurun -- 'bash tools/diag/pmc_sq_counters.sh [tag script args...]'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-mb}
[ $# -gt 0 ] && shift
mkdir -p "$R/gpurun_out"
if [ $# -eq 0 ]; then set -- "$R/tools/kernel_microbench.py" --rounds 1; fi
cd /tmp && export TMPDIR=/tmp
pass() {
  n=$1; shift
  timeout -k 10 500 rocprofv3 --pmc $COUNTERS --kernel-trace -d "$R/gpurun_out/pmc_sq${n}_${TAG}" -o mb --output-format csv -- python3 "$@" > "$R/gpurun_out/pmc_sq${n}_${TAG}.log" 2>&1 || { echo "pass $n failed"; tail -5 "$R/gpurun_out/pmc_sq${n}_${TAG}.log"; exit 1; }
}
COUNTERS="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" pass 1 "$@"
COUNTERS="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" pass 2 "$@"
COUNTERS="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE" pass 3 "$@"
cd "$R" && python3 tools/diag/summarize_sq_counters.py gpurun_out/pmc_sq1_${TAG}/mb_counter_collection.csv gpurun_out/pmc_sq2_${TAG}/mb_counter_collection.csv gpurun_out/pmc_sq3_${TAG}/mb_counter_collection.csv | tee gpurun_out/sq_issue_accounting_${TAG}.txt
