"""View requirements: how a ``[B, T, ...]`` buffer leaf becomes model input.

Only the ``shift=0`` path is on the PPO hot path (``flatten`` for training,
last timestep for sampling; reference ``src/rl8/views.py:408-412, 444-445``).
Rolling / padded windows (``shift > 0``, ``views.py:151-309``) are out of scope
for this build and raise.

"""

from __future__ import annotations

from typing import Literal

import torch

from .tensordict import TensorDict

ViewKind = Literal["last", "all"]
ViewMethod = Literal["rolling_window", "padded_rolling_window"]


class ViewRequirement:
    """Preprocessing of one buffer key before it reaches the model.

    Args:
        shift: Number of additional previous timesteps each sample sees. Only
            ``0`` is supported here.
        method: Windowing method for ``shift > 0`` (accepted for signature
            compatibility).

    """

    def __init__(self, *, shift: int = 0, method: ViewMethod = "padded_rolling_window") -> None:
        if shift < 0:
            raise ValueError(f"{self.__class__.__name__} `shift` must be non-negative.")
        if shift:
            raise NotImplementedError(
                "rl8_amd implements the shift=0 view only (rolling-window views are"
                " outside the accelerated PPO path)."
            )
        self.shift = shift
        self.method = method

    def apply_all(self, key: str | tuple[str, ...], batch: TensorDict, /) -> torch.Tensor | TensorDict:
        """``[B, T, ...] -> [B*T, ...]``."""
        item = batch[key]
        with torch.no_grad():
            if isinstance(item, torch.Tensor):
                return item.flatten(end_dim=1)
            return item.reshape(-1)

    def apply_last(self, key: str | tuple[str, ...], batch: TensorDict, /) -> torch.Tensor | TensorDict:
        """``[B, T, ...] -> [B, ...]`` (most recent timestep)."""
        item = batch[key]
        with torch.no_grad():
            return item[:, -1, ...]

    @property
    def drop_size(self) -> int:
        return 0
