# SQ counters of the fused rollout step (tools/diag/rollout_step_time.py launches it ~100 times per variant), kernel trace only.
R=$PWD; cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d "$R/gpurun_out/pmc_step_$tag" -o s --output-format csv -- python3 "$R/tools/diag/rollout_step_time.py" > "$R/gpurun_out/pmc_step_$tag.log" 2>&1 || echo "pass failed: $set"
done
cd "$R"
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_step_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "rollout_step" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
