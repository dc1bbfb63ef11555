"""Kernel-tuning aid / safety check for the hand-scheduled kernels whose LDS reads
are inline asm (invisible to the compiler): a register spill or reload issued
while such reads are in flight may save / restore a destination register before
its data has arrived.  Reads device assembly (hipcc -S --cuda-device-only) and
reports, per kernel, the scratch instructions and how many of them sit between a
ds_read_b128 and the next lgkmcnt(0) wait.
    python tools/check_async_spills.py build_diag/split/split.s
"""
import re
import sys


def analyse(text: str) -> dict[str, tuple[int, int]]:
    out = {}
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        inflight, bad, total = False, 0, 0
        for line in m.group(2).split("\n"):
            if "ds_read_b128" in line or "ds_read2st64" in line:
                inflight = True
            elif "lgkmcnt(0)" in line:
                inflight = False
            if "scratch_" in line:
                total += 1
                bad += inflight
        out[m.group(1)] = (total, bad)
    return out


if __name__ == "__main__":
    for name, (total, bad) in analyse(open(sys.argv[1]).read()).items():
        if total:
            print(f"{name[:100]}: {total} scratch instructions, {bad} while asm LDS reads are in flight")
