"""Instruction mix of the longest loop of a kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only):
how many VALU / MFMA / SALU / LDS / VMEM instructions one trip issues -- the first thing to look at when SQ counters
say a matrix kernel is bound by operand production.  Usage: python tools/loop_mix.py file.s 'kernel-name-regex'"""

import re
import sys
from collections import Counter


def kernel_body(text: str, pattern: str) -> tuple[str, list[str]]:
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and re.search(pattern, l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start].split(":")[0], lines[start + 1:end]


def longest_loop(body: list[str]) -> tuple[int, int]:
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = [(labels[m.group(1)], i) for i, l in enumerate(body)
             if (m := re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)) and labels.get(m.group(1), 1 << 30) < i]
    # the innermost loop that holds matrix instructions (else the longest one)
    with_mfma = [t for t in loops if any("v_mfma" in l for l in body[t[0]:t[1]])]
    return min(with_mfma, key=lambda t: t[1] - t[0]) if with_mfma else max(loops, key=lambda t: t[1] - t[0])


def main() -> None:
    name, body = kernel_body(open(sys.argv[1]).read(), sys.argv[2])
    lo, hi = longest_loop(body)
    mix = Counter()
    for l in body[lo:hi + 1]:
        l = l.strip()
        if l and not l.startswith((".", ";")):
            mix[l.split()[0]] += 1
    kinds = {"mfma": 0, "valu": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "other": 0}
    for op, n in mix.items():
        k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "smem" if op.startswith(("s_load", "s_buffer"))
             else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "flat_"))
             else "other")
        kinds[k] += n
    print(name)
    print(f"loop lines {lo}..{hi}: {sum(mix.values())} instructions", kinds)
    for op, n in mix.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
        print(f"  {n:5d} {op}")


if __name__ == "__main__":
    main()
