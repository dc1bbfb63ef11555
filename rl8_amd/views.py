"""View requirements: how a ``[B, T, ...]`` buffer leaf becomes model input.

Mirrors the reference's ``src/rl8/views.py`` (``pad_last_sequence`` :54-90,
``pad_whole_sequence`` :93-123, ``rolling_window`` :126-152, ``RollingWindow``
:155-237, ``PaddedRollingWindow`` :240-321, ``ViewRequirement`` :324-453).

``shift = 0`` is the PPO hot path: a flatten for training, the last timestep for
sampling (no copy on the time-major buffer). ``shift > 0`` gives sequence models
the last ``shift + 1`` observations of every sample, either as a plain rolling
window (a strided view; the first ``shift`` samples of each row are dropped) or
padded with zeros in front plus a mask (no samples dropped).

Windows are built with ``Tensor.unfold`` -- a stride trick, no data movement
until a consumer asks for a dense tensor -- on any device.

"""

from __future__ import annotations

from typing import Callable, Literal

import torch

from .data import DataKeys
from .tensordict import TensorDict

ViewKind = Literal["last", "all"]
ViewMethod = Literal["rolling_window", "padded_rolling_window"]


def _padded(x: torch.Tensor, pad: int) -> tuple[torch.Tensor, torch.Tensor]:
    """``pad`` zero steps in front of ``x`` along time, and the mask marking them."""
    b, t = x.shape[:2]
    front = x.new_zeros(b, pad, *x.shape[2:])
    mask = torch.zeros(b, t + pad, dtype=torch.bool, device=x.device)
    mask[:, :pad] = True
    return torch.cat([front, x], dim=1), mask


def _inputs_and_mask(inputs: torch.Tensor, mask: torch.Tensor) -> TensorDict:
    out = TensorDict({}, batch_size=list(mask.shape), device=inputs.device)
    out[DataKeys.INPUTS] = inputs
    out[DataKeys.PADDING_MASK] = mask
    return out


def pad_last_sequence(x: torch.Tensor, size: int, /) -> TensorDict:
    """The last ``size`` steps of ``x`` ``[B, T, ...]``, zero-padded in front when
    ``T < size``: ``{"inputs": [B, size, ...], "padding_mask": [B, size]}``."""
    t = x.shape[1]
    if t < size:
        return _inputs_and_mask(*_padded(x, size - t))
    mask = torch.zeros(x.shape[0], size, dtype=torch.bool, device=x.device)
    return _inputs_and_mask(x[:, -size:, ...], mask)


def pad_whole_sequence(x: torch.Tensor, size: int, /) -> TensorDict:
    """``size - 1`` zero steps in front of ``x`` ``[B, T, ...]`` so that a rolling
    window of ``size`` over the result has ``T`` positions."""
    return _inputs_and_mask(*_padded(x, size - 1))


def rolling_window(x: torch.Tensor, size: int, /, *, step: int = 1) -> torch.Tensor:
    """``[B, T, ...] -> [B, (T - size) / step + 1, size, ...]``: every run of
    ``size`` consecutive steps, as a strided view."""
    windows = x.unfold(1, size, step)  # the window lands in the LAST dimension
    return windows.movedim(-1, 2)


def _map(x: torch.Tensor | TensorDict, fn: Callable[[torch.Tensor], torch.Tensor | TensorDict],
         batch_size: list[int]) -> torch.Tensor | TensorDict:
    """``fn`` on a tensor, or on every leaf of a (nested) tensordict."""
    if isinstance(x, torch.Tensor):
        return fn(x)
    return x.apply(fn, batch_size=batch_size)


class RollingWindow:
    """Plain rolling window: no padding, no mask; the first ``size - 1`` steps of
    each row cannot start a window and are dropped."""

    @staticmethod
    def apply_all(x: torch.Tensor | TensorDict, size: int, /) -> torch.Tensor | TensorDict:
        """``[B, T, ...] -> [B * (T - size + 1), size, ...]``."""
        if isinstance(x, torch.Tensor):
            return rolling_window(x, size).reshape(-1, size, *x.shape[2:])
        b, t = x.shape[:2]
        return x.apply(lambda leaf: rolling_window(leaf, size), batch_size=[b, t - size + 1]).reshape(-1)

    @staticmethod
    def apply_last(x: torch.Tensor | TensorDict, size: int, /) -> torch.Tensor | TensorDict:
        """``[B, T, ...] -> [B, min(T, size), ...]``: the most recent steps."""
        b, t = x.shape[:2]
        return _map(x, lambda leaf: leaf[:, -size:, ...], [b, min(t, size)])

    @staticmethod
    def drop_size(size: int, /) -> int:
        return size - 1


class PaddedRollingWindow:
    """Rolling window over a front-padded sequence: every step keeps a window;
    the padding is flagged in ``"padding_mask"``."""

    @staticmethod
    def apply_all(x: torch.Tensor | TensorDict, size: int, /) -> TensorDict:
        """``[B, T, ...] -> [B * T]`` batch of ``{"inputs": [size, ...],
        "padding_mask": [size]}`` (per leaf for a tensordict)."""
        b, t = x.shape[:2]
        padded = _map(x, lambda leaf: pad_whole_sequence(leaf, size), [b, t + size - 1])
        return RollingWindow.apply_all(padded, size)  # type: ignore[return-value]

    @staticmethod
    def apply_last(x: torch.Tensor | TensorDict, size: int, /) -> TensorDict:
        """``[B, T, ...] -> [B, size]`` batch, zero-padded in front while ``T < size``."""
        return _map(x, lambda leaf: pad_last_sequence(leaf, size), [x.shape[0], size])  # type: ignore[return-value]

    @staticmethod
    def drop_size(size: int, /) -> int:
        return 0


_METHODS = {"rolling_window": RollingWindow, "padded_rolling_window": PaddedRollingWindow}


class ViewRequirement:
    """Preprocessing of one buffer key before it reaches the model.

    Args:
        shift: Number of additional previous timesteps each sample sees.
        method: ``"rolling_window"`` (cheap, drops the first ``shift`` samples
            of each row) or ``"padded_rolling_window"`` (keeps every sample,
            adds a padding mask).

    """

    #: :class:`RollingWindow` or :class:`PaddedRollingWindow`.
    method: type[RollingWindow] | type[PaddedRollingWindow]

    shift: int

    def __init__(self, *, shift: int = 0, method: ViewMethod = "padded_rolling_window") -> None:
        if shift < 0:
            raise ValueError(f"{self.__class__.__name__} `shift` must be non-negative.")
        if method not in _METHODS:
            raise ValueError(f"Unknown view method {method!r}; expected one of {sorted(_METHODS)}.")
        self.shift = shift
        self.method = _METHODS[method]

    @property
    def is_identity(self) -> bool:
        """No windowing: training flattens, sampling takes the last step."""
        return self.shift == 0

    def apply_all(self, key: str | tuple[str, ...], batch: TensorDict, /) -> torch.Tensor | TensorDict:
        """``[B, T, ...] -> [B*T, ...]`` for ``shift = 0``, else ``[B_NEW, shift + 1, ...]``
        with ``B_NEW <= B * T`` depending on the method."""
        item = batch[key]
        with torch.no_grad():
            if not self.shift:
                if isinstance(item, torch.Tensor):
                    return item.flatten(end_dim=1)
                return item.reshape(-1)
            return self.method.apply_all(item, self.shift + 1)

    def apply_last(self, key: str | tuple[str, ...], batch: TensorDict, /) -> torch.Tensor | TensorDict:
        """``[B, T, ...] -> [B, ...]`` (most recent timestep) for ``shift = 0``,
        else ``[B, shift + 1, ...]``."""
        item = batch[key]
        with torch.no_grad():
            if not self.shift:
                return item[:, -1, ...]
            return self.method.apply_last(item, self.shift + 1)

    @property
    def drop_size(self) -> int:
        """Samples lost at the start of every row by the method."""
        return self.method.drop_size(self.shift + 1)
