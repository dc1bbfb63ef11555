"""Per-kernel issue accounting from the three SQ counter passes of tools/diag/pmc_sq_counters.sh.

    python tools/diag/summarize_sq_counters.py gpurun_out/pmc_sq1/mb_counter_collection.csv gpurun_out/pmc_sq2/... gpurun_out/pmc_sq3/...

Per dispatch (averaged over the dispatches of a kernel): T = GRBM_GUI_ACTIVE / 8 XCDs (cycles the kernel ran); per SIMD
(1024 of them): cycles the matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES), cycles the vector ALU was busy
(SQ_ACTIVE_INST_VALU counts quad-cycles: x 4), cycles both were (SQ_VALU_MFMA_COEXEC_CYCLES); LDS bank-conflict cycles per
CU (256).  busy = mfma + valu - both.
"""
import collections
import csv
import sys

WANT = ("mlp_rows_forward_kernel<1, 2, 2", "mlp_rows_forward_kernel<1, 2, 0", "mlp_rows_forward_kernel<1, 2, 1", "mlp_rows_backward_gate_kernel<1, 1",
        "mlp_rows_backward_general_kernel<1, 2", "mlp_wgrad_gate16_kernel<1, false>", "mlp_wgrad_fused16_kernel<1, 2>",
        "lstm_step_split_kernel", "lstm_rows_backward", "mlp_wgrad_loadh16")


def load(path):
    total, seen = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not any(w in k for w in WANT):
            continue
        total[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(seen[k]) for c, v in cs.items()} for k, cs in total.items()}


merged = collections.defaultdict(dict)
for path in sys.argv[1:]:
    for k, cs in load(path).items():
        merged[k].update(cs)
print(f"{'kernel':58} {'T kcyc':>8} {'mfma':>6} {'valu':>6} {'both':>6} {'busy':>6} {'valu/mfma insts':>16} {'lds confl/CU':>13}")
for k, c in sorted(merged.items()):
    if "GRBM_GUI_ACTIVE" not in c:
        continue
    t = c["GRBM_GUI_ACTIVE"] / 8
    mfma = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / t
    valu = 4 * c["SQ_ACTIVE_INST_VALU"] / 1024 / t
    both = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / 1024 / t
    ratio = c.get("SQ_INSTS_VALU", 0) / max(1.0, c.get("SQ_INSTS_MFMA", 0))
    confl = c.get("SQ_LDS_BANK_CONFLICT", 0) / 256 / t
    name = k.replace("void rl8::", "").split("(")[0][:58]
    print(f"{name:58} {t / 1e3:8.0f} {mfma:6.2f} {valu:6.2f} {both:6.2f} {mfma + valu - both:6.2f} {ratio:16.2f} {confl:13.3f}")
