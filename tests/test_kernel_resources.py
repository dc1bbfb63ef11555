"""The shipped tower kernels must not spill: a register spill in these MFMA loops
costs 30-50 % (it happened twice during tuning and only showed up as a slower
bench line). Compiles mlp_kernels.hip to assembly (hipcc cross-compiles without a
GPU) and reads the kernel descriptors."""

import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_inflight_regs as inflight  # noqa: E402


def assert_no_inflight_register_access(text: str, pattern: str = "", min_hand_loads: int = 0) -> int:
    """No kernel of this listing touches the destination of a hand-issued (inline asm)
    load before a wait that covers it (tools/check_inflight_regs.py)."""
    hand = 0
    for name, body in inflight.kernels_of(text):
        if pattern and not re.search(pattern, name):
            continue
        violations, stats = inflight.check_kernel(name, body)
        assert not violations, (name, [(v.index, v.line, v.load) for v in violations[:3]])
        hand += stats.get("hand_loads", 0)
    assert hand >= min_hand_loads, hand
    return hand


def test_inflight_checker_flags_what_it_should():
    """Positive controls: the walker must see a copy between an asm load and its wait,
    honour counted waits and branch arms, and find the packed-fp32 WAR window."""
    def kernel(lines):
        return "\n".join("\t" + ln for ln in lines)

    bad = kernel([";;#ASMSTART", "ds_read_b128 v[4:7], v1 offset:0", ";;#ASMEND", "v_mov_b32_e32 v9, v5",
                  ";;#ASMSTART", "s_waitcnt lgkmcnt(0)", ";;#ASMEND", "s_endpgm"])
    violations, stats = inflight.check_kernel("k", bad)
    assert len(violations) == 1 and violations[0].line.startswith("v_mov_b32") and stats["hand_loads"] == 1
    good = kernel([";;#ASMSTART", "ds_read_b128 v[4:7], v1 offset:0", ";;#ASMEND",
                   ";;#ASMSTART", "ds_read_b128 v[8:11], v1 offset:16", ";;#ASMEND",
                   ";;#ASMSTART", "s_waitcnt lgkmcnt(1)", ";;#ASMEND", "v_mov_b32_e32 v20, v5", "s_waitcnt lgkmcnt(0)",
                   "v_mov_b32_e32 v21, v9", "s_endpgm"])
    assert inflight.check_kernel("k", good)[0] == []
    # a scalar-cache load in flight returns out of order: a counted wait retires nothing
    smem = kernel([";;#ASMSTART", "ds_read_b128 v[4:7], v1 offset:0", ";;#ASMEND", "s_load_dword s4, s[0:1], 0x0",
                   "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v20, v5", "s_endpgm"])
    assert len(inflight.check_kernel("k", smem)[0]) == 1
    # the wait sits on one arm of a branch only
    arm = kernel([";;#ASMSTART", "ds_read_b128 v[4:7], v1 offset:0", ";;#ASMEND", "s_cbranch_scc1 .LBB0_2",
                  "s_waitcnt lgkmcnt(0)", ".LBB0_2:", "v_mov_b32_e32 v20, v5", "s_endpgm"])
    assert len(inflight.check_kernel("k", arm)[0]) == 1
    # compiler-visible loads are the compiler's business (no false alarm on its schedules)
    visible = kernel(["ds_read_b128 v[4:7], v1 offset:0", "v_mov_b32_e32 v9, v5", "s_endpgm"])
    assert inflight.check_kernel("k", visible)[0] == []
    war = kernel(["v_pk_fma_f32 v[196:197], v[224:225], v[194:195], v[196:197] op_sel_hi:[1,0,1]",
                  "v_pk_fma_f32 v[226:227], v[226:227], v[206:207], 0 op_sel_hi:[1,1,0]",
                  "v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]", "v_mov_b32_e32 v224, v225"])
    hits = inflight.packed_war(war)
    assert len(hits) == 1 and hits[0][2] == 3 and hits[0][4] == [("v", 224)]


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
    # the fp32 weight-gradient kernel issues its LDS fragment reads by hand
    assert_no_inflight_register_access(text, min_hand_loads=100)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_compiled_split_kernels_resources(tmp_path):
    """The plane weight-gradient kernels of mlp_split_kernels.hip (bf16 and fp16 planes; the forward and data-gradient
    kernels of this file were removed in round 3): every compiled variant free of scratch, 256 registers at most."""
    csrc = os.path.join(ROOT, "rl8_amd", "csrc")
    asm = tmp_path / "split.s"
    subprocess.run(
        [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{csrc}",
         "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", str(asm), os.path.join(csrc, "mlp_split_kernels.hip")],
        check=True, capture_output=True, timeout=900,
    )
    text = asm.read_text()
    assert len(re.findall(r"s_waitcnt lgkmcnt\(0\)\n\ts_barrier", text)) >= 40  # the hand-written step barriers
    # no packed fp32 arithmetic in these kernels (a hazard beside the bf16 MFMAs: see split_pair())
    assert not re.search(r"\bv_pk_(fma|add|mul)_f32\b", text)
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)
    checked = 0
    for name, body in kernels:
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        assert not re.search(r"mlp_tower_(forward|backward)_split_kernel", name)
        if re.search(r"mlp_wgrad_(gate16|fused16|loadh16)_kernel", name):  # (round 4) sixteen waves: four per SIMD, 128 registers each
            # (round 6: d_in 6 and 7.  The widest variants -- gate16 at d_in 7, fused16 at 6 x 4 -- keep two or three
            # registers of the prologue / epilogue in scratch; none of it inside the chunk loop, checked below)
            wide16 = re.search(r"gate16_kernelILi[67]E|fused16_kernelILi[67]ELi[34]E", name) is not None
            assert (scratch == 0 or (wide16 and scratch <= 16)) and vgprs <= 128, (name, scratch, vgprs)
            if scratch:
                code = text[text.index("\n" + name + ":"):]
                code = code[:code.index(".Lfunc_end")]
                loop = re.search(r"\.LBB\d+_\d+:\s*; =>This Inner Loop Header.*?s_cbranch_\w+ \.LBB\d+_\d+", code, re.S)
                assert loop is not None and "v_mfma" in loop.group(0) and "scratch_" not in loop.group(0), name
            checked += 1
        if re.search(r"mlp_wgrad_split_kernel|mlp_wgrad_gate_kernel", name):
            # every compiled (= dispatched) variant: no scratch at all -- these kernels read
            # LDS through inline asm, so a spill between a read and its wait is a hazard,
            # not just a slowdown -- and two workgroups per CU
            # (round 5: the widest fused variant on the exact bf16 planes, d_in 5 x n_out 4 -- 72 scalar registers of
            # observations and dOut per step -- keeps three registers in scratch; the walkers below hold for it as for
            # the others: no hand-issued load's destination, vector or scalar, is touched before its wait)
            # (round 6: d_in 6 and 7 x n_out 3 and 4 -- up to 88 scalar registers per step -- keep up to ten)
            widest = re.search(r"mlp_wgrad_split_kernelILi[567]ELi[34]ELb0EE", name) is not None
            assert scratch == 0 or (widest and scratch <= 48), (name, scratch)
            assert vgprs <= 256, (name, vgprs)
            checked += 1
    # general (5 run-time/compiled widths from memory + 12 fused, bf16 and fp16 planes), two-operand (4 + 4), gate-plane kernels
    assert checked >= 5 + 12 + 12 + 4 + 4 + 8 + 8 + 8 + 12 + 4, checked
    _check_wgrad_scalar_windows(text)
    # Every hand-issued load (ds_read_b128, s_buffer_load_dwordx8, ...) of every kernel: nothing
    # reads or overwrites its destination before a wait that covers it, on any path.
    assert_no_inflight_register_access(text, min_hand_loads=50 * 100)
    # and the round-1 event's instruction pair cannot form: no packed fp32 op has a source
    # overwritten within the next two VALU slots (there are no packed fp32 ops at all)
    for name, body in inflight.kernels_of(text):
        assert inflight.packed_war(body) == [], name


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



@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_compiled_lstm_split_kernels_resources(tmp_path):
    """The fp16-plane LSTM step kernel (lstm_split_kernels.hip): every compiled input width
    free of scratch, two workgroups per CU, no packed fp32 arithmetic beside the MFMAs,
    and no hand-issued load's destination touched before its wait."""
    csrc = os.path.join(ROOT, "rl8_amd", "csrc")
    asm = tmp_path / "lstm_split.s"
    subprocess.run(
        [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{csrc}",
         "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", str(asm), os.path.join(csrc, "lstm_split_kernels.hip")],
        check=True, capture_output=True, timeout=900,
    )
    text = asm.read_text()
    assert not re.search(r"\bv_pk_(fma|add|mul)_f32\b", text)
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)
    checked = 0
    for name, body in kernels:
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        assert scratch == 0, (name, scratch)
        if "lstm_step_split_kernel" in name:
            assert vgprs <= 256, (name, vgprs)
            checked += 1
    assert checked == 14  # d_in 1..7 x {rollout, training} (round 6; {1, 2, 3, 5} until then)
    assert_no_inflight_register_access(text, "lstm_step_split_kernel", min_hand_loads=8 * 100)
    for name, body in inflight.kernels_of(text):
        assert inflight.packed_war(body) == [], name


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_compiled_f16_kernels_resources(tmp_path):
    """The fp16 two-plane data-gradient kernels (mlp_f16_kernels.hip): no scratch,
    two workgroups per CU, fp16 MFMAs only, no packed fp32 arithmetic and no hand-issued
    load's destination touched before its wait."""
    csrc = os.path.join(ROOT, "rl8_amd", "csrc")
    asm = tmp_path / "mlp_f16.s"
    subprocess.run(
        [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{csrc}",
         "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", str(asm), os.path.join(csrc, "mlp_f16_kernels.hip")],
        check=True, capture_output=True, timeout=900,
    )
    text = asm.read_text()
    assert not re.search(r"\bv_pk_(fma|add|mul)_f32\b", text)
    assert "v_mfma_f32_32x32x16_bf16" not in text and "v_mfma_f32_32x32x16_f16" in text
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)
    checked = 0
    for name, body in kernels:
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
        assert scratch == 0, (name, scratch)
        if "mlp_tower_backward_f16_kernel" in name:
            assert vgprs <= 256 and lds <= 80 * 1024, (name, vgprs, lds)
            checked += 1
    # d_in in 1..5 x n_out in 1..4 data-gradient kernels (general mode: the reference of the rows-shape kernels; their own
    # gate mode went in round 5)
    assert checked == 20
    assert_no_inflight_register_access(text, "mlp_tower_backward_f16_kernel", min_hand_loads=20 * 40)
    for name, body in inflight.kernels_of(text):
        assert inflight.packed_war(body) == [], name


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_compiled_rows_forward_kernels_resources(tmp_path):
    """The rows-per-wave forward (mlp_rows_kernels.hip, round 3; width classes since round 5): every compiled variant
    (d_in class in {1, 2, 3, 8} x n_out class in {1, 2, 4, 8}, class 16 x {1, 2, 4}, x {inference, training with h2,
    training with the gate bits alone}) free of scratch -- classes 8 and 16 excepted, see below -- (its fragment and
    record reads are hand-issued with counted waits: a spill between a read and its wait would save stale data), two
    workgroups per CU, 16x16x32 fp16 MFMAs only, no packed fp32 arithmetic beside them, no hand-issued load's
    destination touched before a covering wait, no scalar load in flight at a counted wait, and every mid-step
    barrier behind a counted vmcnt."""
    csrc = os.path.join(ROOT, "rl8_amd", "csrc")
    asm = tmp_path / "mlp_rows.s"
    subprocess.run(
        [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{csrc}",
         "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", str(asm), os.path.join(csrc, "mlp_rows_kernels.hip")],
        check=True, capture_output=True, timeout=900,
    )
    text = asm.read_text()
    assert not re.search(r"\bv_pk_(fma|add|mul)_f32\b", text)
    assert "v_mfma_f32_16x16x32_f16" in text and "v_mfma_f32_32x32x16" not in text
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)
    checked = 0
    for name, body in kernels:
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        assert ("mlp_rows_forward_kernel" in name or "mlp_rows_backward_gate_kernel" in name
                or "mlp_rows_backward_general_kernel" in name)
        # width class 8 (d_in 4..8: layer 1 on the matrix pipe -- the rows' fragments, their factors and the next tile's
        # input pairs are fourteen more registers per lane than class 1 carries) does not quite fit 256 registers: up to
        # twenty live in scratch, none inside the inner loop of eight half-steps and none a hand-issued load's destination
        # (the walker below holds for these variants too); every other class: no scratch at all
        wide = re.search(r"mlp_rows_forward_kernelILi(8|16)E", name) is not None
        wider = re.search(r"mlp_rows_forward_kernelILi16E", name) is not None  # (two passes: twice the fragments and pairs)
        # the general data gradient's class 8 parks the rows' z1 fragments and the tile's scales (8 + 3 registers, used at
        # the opening and in the epilogue only) in scratch across the matrix loop: stored at the opening, reloaded behind
        # the last half-step
        parked = re.search(r"mlp_rows_backward_general_kernelILi8E", name) is not None
        assert vgprs <= 256 and (scratch <= (160 if wider else 96) if wide or parked else scratch == 0), (name, scratch, vgprs)
        checked += "mlp_rows_forward_kernel" in name
    # width classes {1, 2, 3, 8} x output classes {1, 2, 4, 8} x {inference, h2 stored, gate bits only}
    # + the gate-mode data gradient, d_in class in {1, 2, 3, 8} x n_out in {1, 2}, + the general one, the same classes x
    # KOUT in {2, 4}
    assert checked == 48 + 9 and len(kernels) == 48 + 9 + 8 + 8
    for name, body in inflight.kernels_of(text):
        # the spilling class: the inner loop of the rollout's (SAVE 0) and the gate-bits (SAVE 2) variants stays free of
        # scratch; the h2-storing one (SAVE 1, which also carries the optional h1 store) reloads inside it
        if re.search(r"mlp_rows_forward_kernelILi(8|16)ELi\dELi[02]E", name):
            loop = re.search(r"Inner Loop Header.*?s_cbranch_scc0", body, re.S)
            assert loop is not None and "scratch_" not in loop.group(0), name
    # ring discipline: the barrier inside a half-step waits for "all but the pieces of one younger chunk" (four-chunk
    # rings: inference, gate bits only) or for everything (three-chunk rings: h2 stored)
    forward = "".join(body for name, body in inflight.kernels_of(text) if "mlp_rows_forward_kernel" in name)
    waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)\n\ts_barrier", forward)
    assert len(waits) >= 57 * 12 and set(waits) <= {"0", "4", "8"} and "4" in waits
    assert_no_inflight_register_access(text, "mlp_rows_forward_kernel", min_hand_loads=57 * 100)
    # the rows-per-wave data gradient: same rules (its barriers also count the next tile's row loads and the wave's
    # gate block, so their vmcnt values are not a fixed set)
    assert_no_inflight_register_access(text, "mlp_rows_backward_gate_kernel", min_hand_loads=8 * 50)
    # ... and exactly these: per tile twelve instances of the half-step (0, 1, 2, 3, the loop body's four, 12, 13, 14, 15).
    # A barrier publishes the chunk whose pieces went out kAhead - 1 half-steps earlier; what may be in flight behind those
    # pieces: kAhead - 2 chunks, the row loads of the next tile (half-step 0; half-step 1 only with a four-chunk ring), the
    # gate block (half-step 14; 15 only with a four-chunk ring).  (The three-chunk ring once counted both twice: an
    # intermittent 5e-6 in dW1 at d_in = 3.)
    for name, body in inflight.kernels_of(text):
        m = re.search(r"mlp_rows_backward_gate_kernelILi(\d)ELi(\d)ELi(\d)E", name)
        if not m:
            continue
        d_in, ring = int(m.group(1)), int(m.group(3))
        ahead, rows = ring - 1, 2 * (1 + (2 if d_in == 8 else d_in))  # (class 8: a lane loads an input PAIR per row)
        base = 4 * (ahead - 2)
        want = sorted([base + rows, base + (rows if ahead >= 3 else 0)] + [base] * 8 + [base + 1, base + (1 if ahead >= 3 else 0)]
                      + [4 * (ahead - 1)])  # (+ the prologue's)
        got = [int(v) for v in re.findall(r"s_waitcnt vmcnt\((\d+)\)\n\ts_barrier", body)]
        if d_in == 8 and len(got) == len(want) + 1:  # (class 8: the compiler's own wait in front of the FINAL barrier)
            assert got[-1] == 0
            got = got[:-1]
        assert sorted(got) == want, (name, got, want)
    # the general-head data gradient (round 5): the same ring discipline with 2 (KOUT + d_in) row loads per tile and a
    # three-chunk ring
    assert_no_inflight_register_access(text, "mlp_rows_backward_general_kernel", min_hand_loads=8 * 100)
    seen = 0
    for name, body in inflight.kernels_of(text):
        m = re.search(r"mlp_rows_backward_general_kernelILi(\d)ELi(\d)ELi(\d)E", name)
        if not m:
            continue
        d_in, k_out, ring = int(m.group(1)), int(m.group(2)), int(m.group(3))
        ahead, rows = ring - 1, 2 * (k_out + (2 if d_in == 8 else d_in))
        base = 4 * (ahead - 2)
        # (+ the prologue's; and, in some variants, the compiler's own wait in front of the final __syncthreads, which lets
        # the last tile's never-used row loads -- requested in its epilogue for a tile past the end -- stay in flight
        # behind the kernel's hand-written vmcnt(0))
        want = sorted([base + rows, base + (rows if ahead >= 3 else 0)] + [base] * 8 + [base + 1, base + (1 if ahead >= 3 else 0)]
                      + [4 * (ahead - 1)])
        got = [int(v) for v in re.findall(r"s_waitcnt vmcnt\((\d+)\)\n\ts_barrier", body)]
        # (class 8: ... or one of its parked registers' reloads, by the compiler's count; the hand-written vmcnt(0) stands in front)
        slack = rows + (1 if d_in == 8 else 0)
        assert sorted(got[:13]) == want and len(got) <= 14 and all(v <= slack for v in got[13:]), (name, got, want)
        assert "v_mfma_f32_16x16x32_f16" in body
        seen += 1
    assert seen == 8
    for name, body in inflight.kernels_of(text):
        assert inflight.packed_war(body) == [], name


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_compiled_lstm_rows_backward_kernel_resources(tmp_path):
    """lstm_rows_kernels.hip: the barriers of the backward-through-time kernel count the wave's vector-memory operations
    (s_waitcnt vmcnt(N) with N from a table of loads / stores / requests per gate-step), so the compiled stream must hold
    exactly those: no scratch (a spill is a vector-memory operation), and between consecutive barriers of the chunk loop
    the numbers of loads, direct-to-LDS loads and stores of the table; bf16 32x32x16 MFMAs only, no packed fp32."""
    csrc = os.path.join(ROOT, "rl8_amd", "csrc")
    asm = tmp_path / "lstm_rows.s"
    subprocess.run(
        [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{csrc}",
         "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", str(asm), os.path.join(csrc, "lstm_rows_kernels.hip")],
        check=True, capture_output=True, timeout=900,
    )
    text = asm.read_text()
    assert not re.search(r"\bv_pk_(fma|add|mul)_f32\b", text)
    # both forms: dL/dh_t read as an array, or formed from the heads' gradient and weights (_heads_kernel: phase A of
    # gate-step 6 is fourteen direct-to-LDS loads instead of sixteen, and the barriers that look back over it count two less)
    # (round 6) ... and the HEADS form on fp16 planes: FOUR direct-to-LDS pieces of W_hh^T per gate-step instead of six
    forms = {
        "lstm_rows_backward_kernel": (["B40", "B24", "B32", "B28", "B28", "B12", "B20", "B36"], 22, 6),
        "lstm_rows_backward_heads_kernel": (["B38", "B24", "B32", "B28", "B28", "B12", "B20", "B34"], 20, 6),
        "lstm_rows_backward_heads16_kernel": (["B26", "B20", "B36", "B32", "B24", "B8", "B16", "B22"], 10, 4),
    }
    bodies = dict(re.findall(r"\.amdhsa_kernel (\S*lstm_rows_backward\w*_kernel\S*)(.*?)\.end_amdhsa_kernel", text, re.S))
    assert len(bodies) == 3
    for form, (want, step6, dma) in forms.items():
        body = next(b for n, b in bodies.items() if form + "E" in n or n.endswith(form))
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)) == 0
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)) <= 512
        code = next(b for n, b in inflight.kernels_of(text) if form + "E" in n)
        assert ("v_mfma_f32_32x32x16_f16" if dma == 4 else "v_mfma_f32_32x32x16_bf16") in code and "scratch_" not in code
        assert ("v_mfma_f32_32x32x16_bf16" in code) == (dma == 6)
        # the chunk loop: the eight counted barriers in order, and what the wave issues between them
        ops = re.findall(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)\n\ts_barrier|(buffer_load_dwordx4[^\n]* lds)|(buffer_load_dwordx4)|(buffer_store_dwordx4)", code)
        seq = ["B" + w if w else "D" if d else "L" if ld else "S" for w, d, ld, st in ops]
        bars = [i for i, x in enumerate(seq) if x.startswith("B")]
        first = next(j for j in range(len(bars) - 7) if [seq[i] for i in bars[j:j + 8]] == want)
        ends = bars[first + 1:first + 9] if first + 8 < len(bars) else bars[first + 1:first + 8] + [len(seq)]
        groups = [seq[lo + 1:hi] for lo, hi in zip(bars[first:first + 8], ends)]
        counts = [(g.count("L"), g.count("D"), g.count("S")) for g in groups]
        # [row loads, direct-to-LDS loads (parked row loads + 6 of W_hh^T), stores] per gate-step; the four stores of the
        # next chunk's dG_o follow step 7 on the loop's back edge
        # (every row load is parked: direct-to-LDS, eight per two-array phase, sixteen / fourteen for phase A)
        # (the fp16 form's three-park schedule requests {f, c_prev} in step 0, the next chunk's {o, c_t} in step 1 and its
        # {i, g} in step 2 -- eight parked loads each -- and the rest of phase A, six loads, in step 6)
        step1 = (0, 8 + dma, 8) if dma == 4 else (0, dma, 8)
        assert counts[:7] == [(0, 8 + dma, 0), step1, (0, 8 + dma, 0), (0, dma, 0), (0, dma, 0), (0, dma, 8), (0, step6, 0)], (form, counts)
        assert counts[7][:2] == (0, dma), (form, counts)
