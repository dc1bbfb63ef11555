"""The default recurrent models' LSTM as fused gfx950 kernels.

``torch.nn.LSTM(d_in, 256, num_layers=1, batch_first=True)`` -- what
``DefaultContinuousRecurrentModel`` / ``DefaultDiscreteRecurrentModel`` are built
around (``src/rl8/models/_recurrent.py:201-321`` of the reference) -- runs as one
forward kernel (time loop inside, gates never leave the chip) and, for training,
one backward-through-time kernel plus the weight-gradient kernels, instead of two
GEMMs and a pointwise kernel per timestep. Parameters stay the module's own;
``lstm_forward`` returns ``None`` for any other LSTM (more layers, other widths,
projections, bidirectional, non-HIP / non-fp32 inputs) and the caller runs the
module itself.

"""

from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import hip

#: Set to False to evaluate LSTMs with PyTorch (A/B comparisons).
ENABLED = True
#: "split": the forward step on the bf16 matrix pipe (fp32-accurate bf16-plane products,
#: lstm_split_kernels.hip) where the input width has a compiled variant; "f32": the
#: fp32-MFMA kernel with the time loop inside (lstm_kernels.hip).
FORWARD_GEMM = os.environ.get("RL8_AMD_LSTM_GEMM", "split")
#: The backward through time on bf16 planes, a wave per 32 sequences (lstm_rows_kernels.hip); 0: the fp32-MFMA kernel.
BACKWARD_ROWS = os.environ.get("RL8_AMD_LSTM_BACKWARD_ROWS", "1") != "0"
#: Training passes through LSTM + heads as one autograd node whose backward forms the heads' data gradient inside the
#: backward-through-time kernel (lstm_heads_forward); 0: two nodes, dL/dh through HBM.
FUSE_HEADS = os.environ.get("RL8_AMD_LSTM_FUSE_HEADS", "1") != "0"


def _eligible(lstm: nn.LSTM, x: torch.Tensor) -> bool:
    return (
        ENABLED
        and x.is_cuda
        and x.dtype == torch.float32
        and x.ndim == 3
        and lstm.num_layers == 1
        and lstm.hidden_size == hip.LSTM_HIDDEN
        and lstm.batch_first
        and lstm.bias
        and not lstm.bidirectional
        and lstm.proj_size == 0
        and x.shape[2] == lstm.input_size
        and hip.lstm_supports(lstm.input_size)
    )


def _packs(lstm: nn.LSTM, transposed: bool) -> torch.Tensor:
    """Fragment-ordered copies of the weights, cached ON the module and re-made
    when the optimizer has changed a parameter (version counters) or a parameter
    tensor has been replaced / moved."""
    params = (lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
    stamp = tuple((p._version, p.data_ptr()) for p in params)
    cache = lstm.__dict__.setdefault("_rl8_lstm_packs", {})
    hit = cache.get(transposed)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    if transposed == "split":
        packed = hip.lstm_pack_split(*params)
    elif transposed == "rows":
        packed = hip.lstm_rows_backward_pack(params[1])
    else:
        packed = hip.lstm_pack_transposed(params[1]) if transposed else hip.lstm_pack(*params)
    cache[transposed] = (stamp, packed)
    return packed


def use_split(lstm: nn.LSTM) -> bool:
    return FORWARD_GEMM == "split" and hip.lstm_split_supports(lstm.input_size)


#: The fp16 planes of the last training pass's initial hidden states and max |h0|: the SGD iterations of one step() read
#: the same rows of the buffer (the sequence-major copy of a full-buffer minibatch), whose split is 0.5 GB in, 0.5 GB out.
_h0_cache: dict[str, tuple] = {}
#: Set by ``RecurrentAlgorithm.step()`` while it reads its sequence-major copy of the buffer; off, every pass splits its
#: own h0 (a version counter does not see writes made through raw pointers, e.g. the rollout's into the buffer).
SHARE_H0_PLANES = False


def _h0_planes(h0: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """(planes, max |h0|) of ``h0``; with ``SHARE_H0_PLANES`` re-made only unless this very memory, unmodified since,
    was split last time (the entry keeps ``h0`` alive, so the address cannot have been handed to another tensor)."""
    key = (h0.data_ptr(), tuple(h0.shape), tuple(h0.stride()), h0._version, h0.device)
    hit = _h0_cache.get("entry") if SHARE_H0_PLANES else None
    if hit is not None and hit[0] == key:
        return hit[2], hit[3]
    _h0_cache.pop("entry", None)
    bound = torch.empty(1, dtype=torch.float32, device=h0.device)
    planes = hip.lstm_split_state(h0, bound_out=bound)
    if SHARE_H0_PLANES:
        _h0_cache["entry"] = (key, h0, planes, bound)
    return planes, bound


def clear_state_cache() -> None:
    """Drops the cached planes (and the reference to the rows they were made from) and stops sharing."""
    global SHARE_H0_PLANES
    SHARE_H0_PLANES = False
    _h0_cache.clear()


class _FusedLSTM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h0, c0, w_ih, w_hh, b_ih, b_hh, lstm, grad_mode):  # type: ignore[override]
        need_grad = grad_mode and any(ctx.needs_input_grad[3:7])
        if use_split(lstm):
            packed, wb = _packs(lstm, "split")
            # max |h0| for the backward's fp16-plane weight gradient comes out of the state split
            planes0, bound = _h0_planes(h0) if need_grad else (None, None)
            hs, _, cn, gates, cs = hip.lstm_forward_split(x, h0, c0, packed, wb, save=need_grad, h0_planes=planes0)
        else:
            bound = None
            hs, _, cn, gates, cs = hip.lstm_forward(x, h0, c0, _packs(lstm, False), save=need_grad)
        ctx.set_materialize_grads(False)
        if need_grad:
            ctx.lstm = lstm
            ctx.h0_bound = bound
            ctx.save_for_backward(x, h0, c0, hs, gates, cs)
        ctx.mark_non_differentiable(cn)
        # (c_n may be the last column of the saved cell states, strided: a training pass, which drops the final
        # states, must not pay for a dense copy of 2^19 rows -- 1.6 ms per iteration of the recurrent bench)
        return hs, cn

    @staticmethod
    def backward(ctx, dhs, dcn):  # type: ignore[override]
        x, h0, c0, hs, gates, cs = ctx.saved_tensors
        if dhs is None:
            dhs = torch.zeros_like(hs)
        dhs = dhs.contiguous().float()
        if use_split(ctx.lstm) and BACKWARD_ROWS:
            g = hip.lstm_backward(x, h0, c0, hs, gates, cs, dhs, None, split=True, rows_packed=_packs(ctx.lstm, "rows"),
                                  h0_bound=ctx.h0_bound, hs_bound=1.0)  # (hs: this LSTM's own outputs, |o tanh c| < 1)
        else:
            g = hip.lstm_backward(x, h0, c0, hs, gates, cs, dhs, _packs(ctx.lstm, True), split=use_split(ctx.lstm))
        return None, None, None, g["w_ih"], g["w_hh"], g["b"], g["b"], None, None


def lstm_forward(lstm: nn.LSTM, x: torch.Tensor, h0: torch.Tensor, c0: torch.Tensor):
    """``lstm(x, (h0[None], c0[None]))`` for ``x`` [B, L, d], ``h0`` / ``c0`` [B, 256]
    through the fused kernels: ``(hs [B, L, 256], h_n [B, 256], c_n [B, 256])``, or
    ``None`` when this LSTM / input is not eligible. No gradient flows to ``x``,
    ``h0``, ``c0`` (rollout-buffer data) nor out of ``c_n``."""
    if not _eligible(lstm, x):
        return None
    hs, cn = _FusedLSTM.apply(
        x.contiguous(), h0.contiguous().float(), c0.contiguous().float(), lstm.weight_ih_l0, lstm.weight_hh_l0,
        lstm.bias_ih_l0, lstm.bias_hh_l0, lstm, torch.is_grad_enabled(),
    )
    return hs, hs[:, -1], cn  # h_n is h_{L-1}: a view, so a gradient into it reaches dhs by itself


class _FusedHeads(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, w, b):  # type: ignore[override]
        out = hip.linear_heads_forward(h, w, b)
        ctx.save_for_backward(h, w)
        return out

    @staticmethod
    def backward(ctx, dout):  # type: ignore[override]
        h, w = ctx.saved_tensors
        dh, dw, db = hip.linear_heads_backward(h, dout.contiguous().float(), w)
        return dh, dw, db


class _FusedLSTMHeads(torch.autograd.Function):
    """LSTM + output heads of a training pass as one node: the heads' data gradient dL/dh_t = dOut x W (a rank-n
    product, n <= 4) is formed inside the backward-through-time kernel from the 16 bytes per row-step it is made
    of, instead of being written as [B, L, 256] by the heads' backward and read back (4 KiB of traffic per
    row-step of the recurrent bench less)."""

    @staticmethod
    def forward(ctx, x, h0, c0, w_ih, w_hh, b_ih, b_hh, w_heads, b_heads, lstm):  # type: ignore[override]
        packed, wb = _packs(lstm, "split")
        planes0, bound = _h0_planes(h0)
        hs, _, cn, gates, cs = hip.lstm_forward_split(x, h0, c0, packed, wb, save=True, h0_planes=planes0)
        out = hip.linear_heads_forward(hs.view(-1, hip.LSTM_HIDDEN), w_heads, b_heads)
        ctx.set_materialize_grads(False)
        ctx.lstm, ctx.h0_bound = lstm, bound
        ctx.save_for_backward(x, h0, c0, hs, gates, cs, w_heads)
        ctx.mark_non_differentiable(cn)
        return out, hs, cn

    @staticmethod
    def backward(ctx, dout, dhs, dcn):  # type: ignore[override]
        x, h0, c0, hs, gates, cs, w_heads = ctx.saved_tensors
        flat = hs.view(-1, hip.LSTM_HIDDEN)
        if dout is None:
            dout = torch.zeros(flat.shape[0], w_heads.shape[0], dtype=torch.float32, device=flat.device)
        dout = dout.contiguous().float()
        common = dict(split=True, rows_packed=_packs(ctx.lstm, "rows"), h0_bound=ctx.h0_bound, hs_bound=1.0)
        if dhs is None:  # nothing but the heads reads the latents: the usual case
            _, dw, db = hip.linear_heads_backward(flat, dout, w_heads, need_dh=False)
            g = hip.lstm_backward(x, h0, c0, hs, gates, cs, None, None, heads=(dout, w_heads), **common)
        else:
            dh, dw, db = hip.linear_heads_backward(flat, dout, w_heads)
            g = hip.lstm_backward(x, h0, c0, hs, gates, cs, dh.view_as(hs) + dhs.float(), None, **common)
        return None, None, None, g["w_ih"], g["w_hh"], g["b"], g["b"], dw, db, None


def _heads_eligible(heads: list[nn.Linear], max_out: int) -> bool:
    if any(h.in_features != hip.LSTM_HIDDEN or h.bias is None or h.weight.dtype != torch.float32 for h in heads):
        return False
    return sum(h.out_features for h in heads) <= max_out


def lstm_heads_forward(lstm: nn.LSTM, heads: list[nn.Linear], x: torch.Tensor, h0: torch.Tensor, c0: torch.Tensor):
    """A training pass through ``lstm`` and ``Linear(256, n_i)`` heads on its outputs as one autograd node
    (:class:`_FusedLSTMHeads`): ``([head_i(hs) as [B * L, n_i]], hs [B, L, 256], h_n, c_n)``, or ``None`` when this
    combination is not eligible (no gradient wanted, more than four head outputs, an LSTM the fp16-plane forward or
    the plane-product backward does not take): the caller then runs :func:`lstm_forward` and :func:`heads_forward`."""
    if not (FUSE_HEADS and torch.is_grad_enabled() and _eligible(lstm, x) and use_split(lstm) and BACKWARD_ROWS):
        return None
    if not _heads_eligible(heads, hip.ROWS_BACKWARD_HEADS):
        return None
    if not any(p.requires_grad for p in lstm.parameters()):
        return None
    widths = [h.out_features for h in heads]
    w = torch.cat([h.weight for h in heads], 0) if len(heads) > 1 else heads[0].weight
    b = torch.cat([h.bias for h in heads], 0) if len(heads) > 1 else heads[0].bias
    out, hs, cn = _FusedLSTMHeads.apply(
        x.contiguous(), h0.contiguous().float(), c0.contiguous().float(), lstm.weight_ih_l0, lstm.weight_hh_l0,
        lstm.bias_ih_l0, lstm.bias_hh_l0, w, b, lstm,
    )
    return list(out.split(widths, dim=1)), hs, hs[:, -1], cn


def heads_forward(heads: list[nn.Linear], latents: torch.Tensor) -> None | list[torch.Tensor]:
    """``[head(latents) for head in heads]`` for ``Linear(256, n_i)`` heads on
    ``latents`` [..., 256] in one pass over ``latents`` (and one for the backward),
    or ``None`` when not eligible."""
    if not ENABLED or not latents.is_cuda or latents.dtype != torch.float32 or latents.shape[-1] != hip.LSTM_HIDDEN:
        return None
    if any(h.in_features != hip.LSTM_HIDDEN or h.bias is None for h in heads):
        return None
    widths = [h.out_features for h in heads]
    if sum(widths) > 8:
        return None
    w = torch.cat([h.weight for h in heads], 0) if len(heads) > 1 else heads[0].weight
    b = torch.cat([h.bias for h in heads], 0) if len(heads) > 1 else heads[0].bias
    flat = latents.reshape(-1, hip.LSTM_HIDDEN)
    out = _FusedHeads.apply(flat.contiguous(), w, b)
    return list(out.split(widths, dim=1))
