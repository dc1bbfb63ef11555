"""The shipped tower kernels must not spill: a register spill in these MFMA loops
costs 30-50 % (it happened twice during tuning and only showed up as a slower
bench line). Compiles mlp_kernels.hip to assembly (hipcc cross-compiles without a
GPU) and reads the kernel descriptors."""

import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_compiled_tower_kernels_do_not_use_scratch(tmp_path):
    csrc = os.path.join(ROOT, "rl8_amd", "csrc")
    asm = tmp_path / "mlp.s"
    subprocess.run(
        [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{csrc}",
         "-S", "--cuda-device-only", "-o", str(asm), os.path.join(csrc, "mlp_kernels.hip")],
        check=True, capture_output=True, timeout=600,
    )
    text = asm.read_text()
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)
    assert len(kernels) >= 40
    checked = 0
    for name, body in kernels:
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        # widths compiled in (d_in 1,2,3,5 x n_out 1,2,3): two workgroups per CU => <= 256 registers, no scratch
        m = re.search(r"mlp_tower_(forward|backward)_kernelILi(\d+)ELi(\d+)E", name)
        if m and m.group(2) != "0" and m.group(3) != "0":
            assert scratch == 0, (name, scratch)
            assert vgprs <= 256, (name, vgprs)
            checked += 1
        elif "mlp_wgrad_kernel" in name:
            assert scratch == 0 and vgprs <= 256, (name, scratch, vgprs)
            checked += 1
        else:
            assert scratch == 0, (name, scratch)  # run-time widths get one workgroup per CU instead
    assert checked >= 36 + 1


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_compiled_split_kernels_resources(tmp_path):
    """The bf16-plane tower kernels: every variant that is compiled (and so can be
    dispatched: rl8_mlp_*_split_supports) must be free of scratch and fit two
    workgroups per CU."""
    csrc = os.path.join(ROOT, "rl8_amd", "csrc")
    asm = tmp_path / "split.s"
    subprocess.run(
        [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{csrc}",
         "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", str(asm), os.path.join(csrc, "mlp_split_kernels.hip")],
        check=True, capture_output=True, timeout=900,
    )
    text = asm.read_text()
    waits = re.findall(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)\n\ts_barrier", text)
    assert len(waits) >= 14 * 6 and set(waits) == {"0"}  # the hand-written step barriers
    # no packed fp32 arithmetic in these kernels (a hazard beside the bf16 MFMAs: see split_pair())
    assert not re.search(r"\bv_pk_(fma|add|mul)_f32\b", text)
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)
    checked = 0
    for name, body in kernels:
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        if re.search(r"mlp_tower_(forward|backward)_split_kernel|mlp_wgrad_split_kernel", name):
            # every compiled (= dispatched) variant: no scratch at all -- these kernels read
            # LDS through inline asm, so a spill between a read and its wait is a hazard,
            # not just a slowdown -- and two workgroups per CU
            assert scratch == 0, (name, scratch)
            assert vgprs <= 256, (name, vgprs)
            checked += 1
    assert checked >= 24 + 12 + 5 + 12
    _check_wgrad_scalar_windows(text)


def _sgprs(operand: str) -> set[int]:
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", operand)
    return {int(m.group(1))} if m else set()


def _check_wgrad_scalar_windows(text: str) -> None:
    """The weight-gradient kernel issues s_buffer_load by hand, without a wait of its
    own, and counts ds_reads in order (s_waitcnt lgkmcnt(N > 0)).  Both lean on the
    instruction stream, so check it:
      * from each hand-issued scalar load to the next `lgkmcnt(0)` + s_barrier nothing may
        touch its destination registers (a copy or spill there would save stale data);
      * no scalar-cache load may be in flight at a counted wait (it returns out of order).
    """
    bodies = re.findall(r"\n(_ZN3rl822mlp_wgrad_split_kernel\S+):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S)
    assert len(bodies) >= 17
    loads = counted = 0
    for name, body in bodies:
        lines = [ln.strip() for ln in body.split("\n")]
        lines = [ln for ln in lines if ln and not ln.startswith((";", "."))]
        pending: set[int] = set()   # destinations of hand-issued loads not yet behind a barrier
        smem_in_flight = False
        for i, ln in enumerate(lines):
            op, _, rest = ln.partition(" ")
            operands = [o.strip() for o in rest.split(";")[0].split(",")]
            if op.startswith("s_buffer_load"):
                pending |= _sgprs(operands[0])
                assert not any(pending & _sgprs(o) for o in operands[1:]), (name, ln)  # incl. its own descriptor
                smem_in_flight = True
                loads += 1
                continue
            if op.startswith("s_load"):
                smem_in_flight = True
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", rest)
                if m and int(m.group(1)) == 0:
                    smem_in_flight = False
                    if i + 1 < len(lines) and lines[i + 1].startswith("s_barrier"):
                        pending = set()
                elif m:
                    counted += 1
                    assert not smem_in_flight, (name, i, ln)
                continue
            for o in operands:
                assert not (pending & _sgprs(o)), (name, i, ln)
    assert loads >= 16 * 2 * 2 and counted >= 16 * 2

