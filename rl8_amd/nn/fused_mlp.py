"""The default policy / value tower as fused gfx950 kernels.

``Linear(d_in, 256) -> ReLU -> Linear(256, 256) -> ReLU -> Linear(256, n_out)``
(``src/rl8/models/_feedforward.py:336-362`` of the reference) runs as one forward
kernel and two backward kernels, in fp32 on the matrix cores, instead of ~8
(forward) / ~20 (backward) eager launches whose 256-wide activations each make
an HBM round trip. Parameters stay ordinary
``torch.nn.Linear`` weights; this module only changes how the tower is evaluated.

``tower_forward`` falls back to the module's own eager path whenever the tower
is not exactly that shape (other widths, activations, norm layers, non-HIP or
non-fp32 inputs), so custom models are unaffected.

Under ``enable_amp`` (torch autocast) the fused towers still run, in fp32: that is
at least the precision autocast asks for, and on MI355X it is also ~5x faster than
the autocast path of the same modules (37 vs 8 M transitions/s on the headline
config), whose casts and elementwise launches dominate at this width.

"""

from __future__ import annotations

import os
from typing import Sequence

import torch
import torch.nn as nn

from .. import hip
from . import piecewise_mlp

#: Set to False to evaluate towers with eager PyTorch (A/B comparisons).
ENABLED = True

#: How the 256x256 product of the forward kernel is formed: "f16" = both fp32
#: operands scaled by powers of two (per activation row / per weight matrix) and
#: split into two fp16 planes, three plane products per 16 k on the fp16 matrix
#: pipe, fp32 accumulate (fp32 accuracy: same fp64 bars as the others);
#: "f32" = v_mfma_f32_32x32x2_f32 (2.7x slower; also what widths without a plane
#: kernel run).  (The six-product bf16-plane forward / data-gradient kernels of rounds 1-2,
#: "split", were removed in round 3.)  ``RL8_AMD_TOWER_GEMM`` /
#: ``RL8_AMD_TOWER_FORWARD_GEMM`` / ``RL8_AMD_TOWER_BACKWARD_GEMM`` override.
FORWARD_GEMM = os.environ.get("RL8_AMD_TOWER_FORWARD_GEMM", os.environ.get("RL8_AMD_TOWER_GEMM", "f16"))
#: Same choice for the data-gradient product of the backward pass (the weight
#: gradient runs on fp16 planes scaled per output column, ``hip`` / mlp_split_kernels.hip).
BACKWARD_GEMM = os.environ.get("RL8_AMD_TOWER_BACKWARD_GEMM", os.environ.get("RL8_AMD_TOWER_GEMM", "f16"))
for _name, _value in (("forward", FORWARD_GEMM), ("backward", BACKWARD_GEMM)):
    if _value not in ("f16", "f32"):
        # ("split", the bf16-plane forward / data-gradient generation, was removed in round 3: say so instead of
        # silently running the fp32-MFMA kernels -- ADVICE r3)
        raise ValueError(f"RL8_AMD_TOWER_GEMM / RL8_AMD_TOWER_{_name.upper()}_GEMM = {_value!r}: the towers' {_name} product"
                         " runs as 'f16' (fp16 planes, default) or 'f32' (fp32 MFMA); 'split' no longer exists")


def _packed(layer: nn.Linear, transposed: bool, split: bool | str = False) -> torch.Tensor:
    """MFMA-fragment-ordered copy of ``layer.weight``, cached ON the layer and
    re-made when the optimizer has changed the weight (version counter) or the
    weight tensor has been replaced / moved."""
    w2 = layer.weight
    cache = layer.__dict__.setdefault("_rl8_w2_packs", {})
    hit = cache.get((transposed, split))
    if hit is not None and hit[0] == w2._version and hit[1] == w2.data_ptr():
        return hit[2]
    pack = hip.mlp_pack_w2_f16 if split == "f16" else hip.mlp_pack_w2
    packed = pack(w2, transposed=transposed)
    cache[(transposed, split)] = (w2._version, w2.data_ptr(), packed)
    return packed


def _packed_gate(layer: nn.Linear, w3: torch.Tensor, w3_key: tuple) -> torch.Tensor:
    """``hip.mlp_pack_w2_f16_gate(layer.weight, w3)``, cached on the layer like ``_packed`` and re-made when
    either weight has changed. ``w3_key`` identifies the head PARAMETERS (version counters and addresses):
    ``w3`` itself is a fresh ``torch.cat`` temporary for multi-head towers, whose version is always 0 and whose
    address the caching allocator may hand out again."""
    w2 = layer.weight
    cache = layer.__dict__.setdefault("_rl8_w2_packs", {})
    hit = cache.get("gate")
    key = (w2._version, w2.data_ptr(), w3_key, tuple(w3.shape))
    if hit is not None and hit[0] == key:
        return hit[1]
    packed = hip.mlp_pack_w2_f16_gate(w2, w3)
    cache["gate"] = (key, packed)
    return packed


def _gates_off() -> bool:
    return bool(int(os.environ.get("RL8_WGRAD_GATE_OFF", "0") or 0))


_PAIR_HINT = False


class expect_pair_gradients:
    """Context manager for callers that KNOW the gradients of a two-output tower evaluated inside it will be exact
    negatives of each other (``Algorithm``: a two-way ``Categorical`` policy under the fused PPO loss, whose kernel
    emits antisymmetric logit gradients by construction). Towers that ask (``pair_gradients=None`` resolves to
    this hint) then keep only the gate bits of h2 from the first iteration on. The promise is checked on the
    device in the backward (``rl8_mlp_dout_pair_check``); a broken one costs a re-run of the forward, never a
    wrong gradient. One exception: a gradient tensor registered by ``trust_pair_gradient`` -- the two-class loss
    kernel's own output, a pair by construction -- is taken on trust while its address, element count and version
    counter are the registered ones (any in-place change, copy or rescaling of it is checked again)."""

    def __enter__(self):
        global _PAIR_HINT
        self._before = _PAIR_HINT
        _PAIR_HINT = True
        return self

    def __exit__(self, *exc):
        global _PAIR_HINT
        _PAIR_HINT = self._before
        return False


def pair_hint() -> bool:
    return _PAIR_HINT


#: Gradient tensors known to be exact pairs BY CONSTRUCTION: the output of ``rl8_ppo_loss_categorical_fwd_bwd_f32`` for
#: two classes (the kernel stores ``g`` and ``-g``), registered by ``nn.functional.fused_ppo_loss`` under (address,
#: element count, version).  A backward that receives exactly such a tensor skips ``rl8_mlp_dout_pair_check`` -- a pass
#: over dOut and a host read per backward: with 8 shuffled minibatches 64 host round trips per ``step()`` -- and anything
#: else (a scaled copy under AMP, a custom loss) is still checked on the device.
_TRUSTED_PAIRS: dict[int, tuple[torch.Tensor, int]] = {}


def trust_pair_gradient(g: torch.Tensor) -> None:
    """Registers ``g`` (kept alive here until a backward consumes it or four newer ones arrive, so that its address
    cannot be handed to another tensor in between)."""
    while len(_TRUSTED_PAIRS) >= 4:
        _TRUSTED_PAIRS.pop(next(iter(_TRUSTED_PAIRS)))
    _TRUSTED_PAIRS[g.data_ptr()] = (g, g._version)


def _trusted_pair(g: torch.Tensor) -> bool:
    hit = _TRUSTED_PAIRS.pop(g.data_ptr(), None)
    return (hit is not None and hit[0].numel() == g.numel() and hit[0]._version == hit[1]
            and hit[0].untyped_storage().data_ptr() == g.untyped_storage().data_ptr())


class _FusedTower(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, layer2, grad_mode, pair, w3_key):  # type: ignore[override]
        # needs_input_grad reflects the parameters' requires_grad even when the
        # caller runs under no_grad (rollouts), and inside forward() grad mode is
        # always off: the caller's grad mode comes in as an argument, so that
        # activations are kept only when a backward can follow.
        need_grad = grad_mode and any(ctx.needs_input_grad[1:7])
        if FORWARD_GEMM == "f16" and hip.mlp_forward_f16_supports(x.shape[1], w3.shape[0]):
            # h1 is stored only if a backward kernel will read it (the plane kernels recompute it)
            keep_h1 = not (BACKWARD_GEMM == "f16" and hip.mlp_backward_f16_supports(x.shape[1], w3.shape[0]))
            f16 = True
            # Rank-one heads (one output; two outputs with exactly opposite gradients, as the last backward of
            # this tower found them) need only the gate bits of h2 in the backward pass: no h2 store, no h2
            # read.  Should a two-output head stop being rank-one, its backward re-runs this forward for h2.
            n_out = w3.shape[0]
            gate_only = (need_grad and f16 and not keep_h1 and BACKWARD_GEMM == "f16"
                         and hip.mlp_backward_f16_supports(x.shape[1], n_out) and not _gates_off()
                         and (n_out == 1 or (n_out == 2 and pair)))
            out, h1, h2, gate = hip.mlp_tower_forward_split(x, w1, b1, _packed(layer2, False, "f16"),
                                                            b2, w3, b3, save=need_grad, save_h1=keep_h1, save_gate=True,
                                                            save_h2=not gate_only)
        else:
            out, h1, h2 = hip.mlp_tower_forward(x, w1, b1, _packed(layer2, False), b2, w3, b3, save=need_grad)
            gate = None
        if need_grad:
            ctx.layer2 = layer2
            ctx.w3_key = w3_key
            # (w2 is saved although the kernels read its packed copies: autograd's version check then refuses a
            # backward after an in-place change of the weight, as it would for the eager modules)
            ctx.save_for_backward(x, h1, h2, w3, w1, b1, gate, b2, b3, w2)
        return out

    @staticmethod
    def backward(ctx, dout):  # type: ignore[override]
        return _tower_backward(ctx, dout)


def _tower_backward(ctx, dout):
    """Backward of ``_FusedTower`` and of ``_ReplayedTower`` (same saved tensors, same kernels)."""
    x, h1, h2, w3, w1, b1, gate, b2, b3, w2 = ctx.saved_tensors
    split: bool | str = False
    if BACKWARD_GEMM == "f16" and gate is not None and hip.mlp_backward_f16_supports(x.shape[1], w3.shape[0]):
        split = "f16"
    layer2 = ctx.layer2
    w3_key = ctx.w3_key
    gate_pack = (lambda: _packed_gate(layer2, w3, w3_key)) if split == "f16" and w3.shape[0] <= 2 else None

    def h2_again():  # (a two-output head that is not rank-one after all: the forward once more, with h2)
        return hip.mlp_tower_forward_split(x, w1, b1, _packed(layer2, False, "f16"), b2, w3, b3, save=True,
                                           save_h1=False, save_gate=True)[2]

    info: dict = {}
    # (a two-output head an earlier backward found NOT to be a pair -- mean / log_std of a normal -- is not asked again)
    known_general = w3.shape[0] == 2 and layer2.__dict__.get("_rl8_rank_one") is False
    known_pair = w3.shape[0] == 2 and dout.dtype == torch.float32 and dout.is_contiguous() and _trusted_pair(dout)
    g = hip.mlp_tower_backward(x, h1, h2, dout.contiguous().float(), _packed(layer2, True, split), w3,
                               w1, b1, wgrad_split=BACKWARD_GEMM == "f16", gate2=gate if split else None,
                               gate_pack=gate_pack, w2=w2, b2=b2, h2_fn=h2_again, info=info, assume_general=known_general,
                               assume_pair=known_pair)
    if w3.shape[0] == 2:  # what this backward found, for callers that give no hint (see tower_forward)
        layer2.__dict__["_rl8_rank_one"] = bool(info.get("rank_one", False))
    return None, g["w1"], g["b1"], g["w2"], g["b2"], g["w3"], g["b3"], None, None, None, None


# --------------------------------------------------------------------------- #
# The rollout's forward passes, kept for the first SGD pass.
#
# ``Algorithm.collect()`` evaluates both towers on every observation of the buffer
# (reference ``algorithms/_feedforward.py:362-367``); SGD iteration 0 of ``step()``
# then evaluates them AGAIN on the same observations with the same weights
# (``:519-524``).  The forward kernel is deterministic per row, so that second pass
# reproduces numbers that already exist.  A ``RolloutRecord`` makes the rollout's
# launches run in the training forward's save mode (gate bits of h2, 32 B per row;
# h2 too for heads that are not rank-one) into ``[H][N]`` slabs -- the row order of
# the time-major buffer's flat full batch -- and ``replay`` hands those to autograd
# in place of a forward launch.  Used only while every parameter of the tower is
# what the rollout saw (version counters and addresses).
# --------------------------------------------------------------------------- #
#: Absolute cap on the h2 slabs a rollout record may hold (1 KiB per row and general-head tower: 34 GB at 2^25 rows),
#: besides "a third of what is free when the slab is made" (ADVICE r4).  ``RL8_AMD_RECORD_H2_GIB`` overrides; 0: never.
def _record_h2_budget_bytes() -> int:
    try:  # (a malformed value must not abort importing the package: ADVICE r5)
        gib = float(os.environ.get("RL8_AMD_RECORD_H2_GIB", "48"))
    except ValueError:
        gib = 48.0
    return int(max(gib, 0.0) * (1 << 30))


RECORD_H2_BUDGET_BYTES = _record_h2_budget_bytes()
#: ... and the slabs are given back when, at the end of a step(), less than this is free on the device.
RECORD_H2_KEEP_FREE_BYTES = 8 << 30


class _TowerRecord:
    def __init__(self, params: list[torch.Tensor], n_out: int, gate_only: bool, rows: int, device) -> None:
        self.params = params
        self.n_out, self.gate_only, self.rows = n_out, gate_only, rows
        self.out = torch.empty(rows, n_out, dtype=torch.float32, device=device)
        self.gate = torch.empty(rows, 8, dtype=torch.int32, device=device)
        self.h2 = None if gate_only else torch.empty(rows, hip.MLP_HIDDEN, dtype=torch.float32, device=device)
        self.key: None | tuple = None
        self.seen: set[int] = set()
        #: address of the input each recorded timestep was evaluated on; a second evaluation of the same tower inside
        #: one ``record.at(t)`` (a model that runs it on obs AND next_obs) spoils the timestep: never replayed
        self.inputs: dict[int, int] = {}
        self.spoiled = False

    def current_key(self) -> tuple:
        return tuple((p._version, p.data_ptr()) for p in self.params)

    def leaves(self) -> list[torch.Tensor]:
        return [t for t in (self.out, self.gate, self.h2) if t is not None]


class RolloutRecord:
    """Outputs, gate bits (and h2 where needed) of the towers for the ``steps x rows_per_step`` observations of one
    rollout, in time-major row order. ``at(t)`` is the context ``collect()`` evaluates timestep ``t`` in."""

    def __init__(self, steps: int, rows_per_step: int, *, keep_general: bool) -> None:
        self.steps, self.rows_per_step = steps, rows_per_step
        self.keep_general = keep_general
        self.towers: dict[int, _TowerRecord] = {}
        self.t = -1
        self.refused: set[int] = set()

    def begin(self) -> None:
        for tr in self.towers.values():
            tr.seen.clear()
            tr.inputs.clear()
            tr.spoiled = False
            tr.key = None

    def at(self, t: int) -> "_Recording":
        return _Recording(self, t)

    def bytes(self) -> int:
        return sum(t.numel() * t.element_size() for tr in self.towers.values() for t in tr.leaves())

    def _tower(self, layer2: nn.Linear, params: list[torch.Tensor], n_out: int, gate_only: bool, device) -> None | _TowerRecord:
        tr = self.towers.get(id(layer2))
        rows = self.steps * self.rows_per_step
        if tr is not None and (tr.n_out, tr.gate_only, tr.rows) == (n_out, gate_only, rows) and tr.out.device == device:
            tr.params = params
            return tr
        if id(layer2) in self.refused:
            return None
        if not gate_only:
            # h2 is 1 KiB per row (34 GB per tower at 2^25 rows): only where every row is read back and it fits easily
            need = rows * (hip.MLP_HIDDEN * 4 + 32 + 4 * n_out)
            free, _ = torch.cuda.mem_get_info(device)
            held = sum(t.h2.numel() * 4 for t in self.towers.values() if t.h2 is not None)
            if not self.keep_general or need > free // 3 or held + need > RECORD_H2_BUDGET_BYTES:
                self.refused.add(id(layer2))
                return None
        self.towers.pop(id(layer2), None)
        try:
            tr = self.towers[id(layer2)] = _TowerRecord(params, n_out, gate_only, rows, device)
        except torch.cuda.OutOfMemoryError:  # (the tower is then evaluated without a record: one more forward per step())
            self.refused.add(id(layer2))
            return None
        return tr

    def release_if_tight(self, device) -> int:
        """End of a ``step()``: give the h2 slabs back (and refuse new ones) when the device is short of memory --
        activations and workspaces allocated by ``step()`` come after the slabs and must not be what runs out.  Returns
        the bytes released."""
        if not any(tr.h2 is not None for tr in self.towers.values()):
            return 0  # (nothing to give back: no driver query -- hipMemGetInfo costs ~100 ms per call when several
            #            processes share the device, as the multi-rank rehearsals on one GPU do)
        free, _ = torch.cuda.mem_get_info(device)
        if free >= RECORD_H2_KEEP_FREE_BYTES:
            return 0
        released = 0
        for key in [k for k, tr in self.towers.items() if tr.h2 is not None]:
            released += self.towers[key].h2.numel() * 4
            del self.towers[key]
            self.refused.add(key)
        return released

    def _usable(self, tr: _TowerRecord) -> bool:
        """The tower saw all the timesteps, once each, with the parameters it has now."""
        return (tr.key is not None and not tr.spoiled and len(tr.seen) == self.steps
                and tr.current_key() == tr.key)

    def require_inputs(self, base_ptr: int, step_bytes: int) -> None:
        """The caller replays the record against the rows of ONE dense array (the time-major observations, timestep t
        at ``base_ptr + t * step_bytes``): a tower recorded on anything else -- a model that feeds it a transformed or
        shifted view -- is never replayed."""
        for tr in self.towers.values():
            if any(ptr != base_ptr + t * step_bytes for t, ptr in tr.inputs.items()):
                tr.spoiled = True

    def valid(self) -> bool:
        """Some recorded tower can still be replayed."""
        return any(self._usable(tr) for tr in self.towers.values())

    def rows(self, start: int, stop: int) -> dict[int, tuple]:
        """Record rows ``[start, stop)`` (time-major order) of every usable tower."""
        return {k: (tr.key, tr.out[start:stop], tr.gate[start:stop], tr.h2[start:stop] if tr.h2 is not None else None)
                for k, tr in self.towers.items() if self._usable(tr)}

    def gather(self, index: torch.Tensor) -> dict[int, tuple]:
        """The same for the samples ``index`` names (reference sample ids ``env * H + t``): one strided gather per
        tower (``rl8_gather_minibatch``) out of the ``[H][N]`` slabs seen as ``[N, H, ...]`` leaves."""
        H, N = self.steps, self.rows_per_step
        got: dict[int, tuple] = {}
        for k, tr in self.towers.items():
            if not self._usable(tr):
                continue
            leaves = [t.view(H, N, t.shape[1]).transpose(0, 1) for t in tr.leaves()]
            dense = hip.gather_minibatch(index, H, leaves)
            got[k] = (tr.key, dense[0], dense[1], dense[2] if tr.h2 is not None else None)
        return got


_RECORDING: None | RolloutRecord = None
_REPLAY: None | tuple[dict[int, tuple], torch.Tensor] = None


class _Recording:
    def __init__(self, record: RolloutRecord, t: int) -> None:
        self.record, self.t = record, t

    def __enter__(self):
        global _RECORDING
        self._before = _RECORDING
        self.record.t = self.t
        _RECORDING = self.record
        return self.record

    def __exit__(self, *exc):
        global _RECORDING
        _RECORDING = self._before
        return False


class replay:
    """Context for a grad-enabled pass over ``x`` (a dense ``[m, d]`` tensor): towers found in ``rows`` (from
    ``RolloutRecord.rows`` / ``.gather``) whose parameters are still the recorded ones skip their forward launch."""

    def __init__(self, rows: dict[int, tuple], x: torch.Tensor) -> None:
        self.rows, self.x = rows, x

    def __enter__(self):
        global _REPLAY
        self._before = _REPLAY
        _REPLAY = (self.rows, self.x)
        return self

    def __exit__(self, *exc):
        global _REPLAY
        _REPLAY = self._before
        return False


#: Counts for tests / the bench line: towers evaluated from the record, rows recorded.
replay_stats = {"replayed_towers": 0, "replayed_rows": 0, "recorded_rows": 0}


class _ReplayedTower(torch.autograd.Function):
    """``_FusedTower`` whose forward already ran (during the rollout): returns the recorded output and saves the
    recorded gate bits / h2 for the same backward kernels."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, layer2, w3_key, out, gate, h2):  # type: ignore[override]
        ctx.layer2 = layer2
        ctx.w3_key = w3_key
        ctx.save_for_backward(x, None, h2, w3, w1, b1, gate, b2, b3, w2)
        return out.detach()

    @staticmethod
    def backward(ctx, dout):  # type: ignore[override]
        return _tower_backward(ctx, dout) + (None,)


def _tower_params(l1: nn.Linear, l2: nn.Linear, heads: Sequence[nn.Linear]) -> list[torch.Tensor]:
    return [l1.weight, l1.bias, l2.weight, l2.bias, *[h.weight for h in heads], *[h.bias for h in heads]]


def _match(trunk: nn.Module, heads: Sequence[nn.Linear]) -> None | tuple[nn.Linear, nn.Linear]:
    """(layer1, layer2) if ``trunk`` is ``Sequential(MLP(Linear, ReLU, Linear), ReLU)``
    (optionally followed by the single head) with 256-wide biased layers."""
    if not isinstance(trunk, nn.Sequential) or len(trunk) < 2:
        return None
    mlp, act = trunk[0], trunk[1]
    if not isinstance(mlp, nn.Sequential) or len(mlp) != 3 or not isinstance(act, nn.ReLU):
        return None
    l1, a1, l2 = mlp[0], mlp[1], mlp[2]
    if not (isinstance(l1, nn.Linear) and isinstance(a1, nn.ReLU) and isinstance(l2, nn.Linear)):
        return None
    if l1.out_features != hip.MLP_HIDDEN or l2.out_features != hip.MLP_HIDDEN or l2.in_features != hip.MLP_HIDDEN:
        return None
    if l1.in_features > hip.MLP_MAX_IN or l1.bias is None or l2.bias is None:
        return None
    if sum(h.out_features for h in heads) > hip.MLP_MAX_OUT or any(h.bias is None for h in heads):
        return None
    if any(h.in_features != hip.MLP_HIDDEN for h in heads):
        return None
    return l1, l2


def tower_forward(trunk: nn.Sequential, heads: Sequence[nn.Linear], x: torch.Tensor, *,
                  pair_gradients: None | bool = None) -> None | torch.Tensor:
    """``cat([head(trunk(x)) for head in heads], -1)`` through the fused kernels,
    or ``None`` when this tower / input is not eligible (caller then runs the
    modules eagerly).

    ``pair_gradients`` (two-output towers only): ``True`` -- the caller expects the two outputs' gradients to be
    exact negatives (a two-way categorical under the fused loss), so the forward keeps the gate bits of h2 alone;
    ``False`` -- h2 is stored; ``None`` -- whatever this tower's previous backward found (a tower used outside
    ``Algorithm``). Checked on the device in the backward either way."""
    if not ENABLED or not x.is_cuda or x.dtype != torch.float32 or x.ndim != 2:
        return None
    layers = _match(trunk, heads)
    if layers is None or x.shape[1] != layers[0].in_features:
        return None
    l1, l2 = layers
    if piecewise_mlp.ENABLED and x.shape[1] == 1:  # opt-in prototype: scalar observations from an exact piecewise-linear table
        out = piecewise_mlp.tower_forward(l1, l2, heads, x)
        if out is not None:
            return out
    if len(heads) == 1:
        w3, b3 = heads[0].weight, heads[0].bias
    else:
        w3 = torch.cat([h.weight for h in heads], 0)
        b3 = torch.cat([h.bias for h in heads], 0)
    if pair_gradients is None:
        pair_gradients = bool(l2.__dict__.get("_rl8_rank_one", False))
    w3_key = tuple((h.weight._version, h.weight.data_ptr()) for h in heads)
    n_out, d_in = w3.shape[0], x.shape[1]
    plane_path = (FORWARD_GEMM == "f16" and BACKWARD_GEMM == "f16" and hip.mlp_forward_f16_supports(d_in, n_out)
                  and hip.mlp_backward_f16_supports(d_in, n_out))
    if _RECORDING is not None and not torch.is_grad_enabled() and plane_path and x.shape[0] == _RECORDING.rows_per_step:
        rec = _RECORDING
        gate_only = not _gates_off() and (n_out == 1 or (n_out == 2 and bool(pair_gradients)))
        params = _tower_params(l1, l2, heads)
        tr = rec._tower(l2, params, n_out, gate_only, x.device)
        if tr is not None:
            key = tr.current_key()
            if tr.key != key:
                tr.key, tr.seen = key, set()
                tr.inputs.clear()
                tr.spoiled = False
            if rec.t in tr.seen:  # second evaluation in this timestep's context: the slab rows of t would be overwritten
                tr.spoiled = True
                return _FusedTower.apply(x.contiguous(), l1.weight, l1.bias, l2.weight, l2.bias, w3, b3, l2,
                                         torch.is_grad_enabled(), bool(pair_gradients), w3_key)
            tr.inputs[rec.t] = x.data_ptr()
            lo = rec.t * rec.rows_per_step
            sl = slice(lo, lo + rec.rows_per_step)
            out = hip.mlp_tower_forward_split(
                x.contiguous(), l1.weight, l1.bias, _packed(l2, False, "f16"), l2.bias, w3, b3, save=True,
                save_h1=False, save_gate=True, save_h2=not gate_only, out=tr.out[sl], gate_out=tr.gate[sl],
                h2_out=None if gate_only else tr.h2[sl], timer_name="mlp_tower_forward_record")[0]
            tr.seen.add(rec.t)
            replay_stats["recorded_rows"] += rec.rows_per_step
            return out
    if _REPLAY is not None and torch.is_grad_enabled() and plane_path:
        rows, expect = _REPLAY
        hit = rows.get(id(l2))
        if (hit is not None and x.data_ptr() == expect.data_ptr() and x.shape == expect.shape and x.is_contiguous()
                and hit[0] == tuple((p._version, p.data_ptr()) for p in _tower_params(l1, l2, heads))
                and hit[1].shape == (x.shape[0], n_out)
                and (hit[3] is not None or n_out == 1 or (n_out == 2 and bool(pair_gradients)))):
            replay_stats["replayed_towers"] += 1
            replay_stats["replayed_rows"] += x.shape[0]
            return _ReplayedTower.apply(x, l1.weight, l1.bias, l2.weight, l2.bias, w3, b3, l2, w3_key,
                                        hit[1], hit[2], hit[3])
    return _FusedTower.apply(x.contiguous(), l1.weight, l1.bias, l2.weight, l2.bias, w3, b3, l2,
                             torch.is_grad_enabled(), bool(pair_gradients), w3_key)
