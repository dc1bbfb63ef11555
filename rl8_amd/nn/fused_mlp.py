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
    this hint) then keep only the gate bits of h2 from the first iteration on. The promise is still checked on the
    device in every backward (``rl8_mlp_dout_pair_check``); a broken one costs a re-run of the forward, never a
    wrong gradient."""

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
        g = hip.mlp_tower_backward(x, h1, h2, dout.contiguous().float(), _packed(layer2, True, split), w3,
                                   w1, b1, wgrad_split=BACKWARD_GEMM == "f16", gate2=gate if split else None,
                                   gate_pack=gate_pack, w2=w2, b2=b2, h2_fn=h2_again, info=info)
        if w3.shape[0] == 2:  # what this backward found, for callers that give no hint (see tower_forward)
            layer2.__dict__["_rl8_rank_one"] = bool(info.get("rank_one", False))
        return None, g["w1"], g["b1"], g["w2"], g["b2"], g["w3"], g["b3"], None, None, None, None


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
    if len(heads) == 1:
        w3, b3 = heads[0].weight, heads[0].bias
    else:
        w3 = torch.cat([h.weight for h in heads], 0)
        b3 = torch.cat([h.bias for h in heads], 0)
    if pair_gradients is None:
        pair_gradients = bool(l2.__dict__.get("_rl8_rank_one", False))
    w3_key = tuple((h.weight._version, h.weight.data_ptr()) for h in heads)
    return _FusedTower.apply(x.contiguous(), l1.weight, l1.bias, l2.weight, l2.bias, w3, b3, l2,
                             torch.is_grad_enabled(), bool(pair_gradients), w3_key)
