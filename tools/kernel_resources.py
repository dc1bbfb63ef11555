"""Per-kernel registers / scratch / static LDS of one kernel file, from the compiler's assembly (no GPU needed).

    python tools/kernel_resources.py mlp_rows_kernels.hip [name-filter]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rl8_amd", "csrc")


def listing(source: str) -> str:
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        flags = ["-fno-slp-vectorize"] if not source.startswith(("mlp_kernels", "gae", "ppo", "rollout", "stats", "classic", "piecewise", "lstm_kernels")) else []
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        f"-I{ROOT}/include", f"-I{CSRC}", *flags, "-S", "--cuda-device-only", "-o", asm,
                        os.path.join(CSRC, source)], check=True)
        return open(asm).read()


def main() -> None:
    text = listing(sys.argv[1])
    needle = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, body in re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        if needle not in name:
            continue
        get = lambda key: int(re.search(rf"\.amdhsa_{key} (\d+)", body).group(1))  # noqa: E731
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print(f"{get('next_free_vgpr'):4d} vgpr {get('next_free_sgpr'):4d} sgpr {get('private_segment_fixed_size'):5d} scratch"
              f" {get('group_segment_fixed_size'):6d} lds  {demangled[:110]}")


if __name__ == "__main__":
    main()
