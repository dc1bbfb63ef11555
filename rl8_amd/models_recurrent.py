"""Recurrent policy / value networks (LSTM on PyTorch-ROCm / MIOpen).

Interfaces follow the reference's ``src/rl8/models/_recurrent.py``:
``RecurrentModel`` :19-138 (``forward(batch, states) -> (features, states)``,
``state_spec``, ``init_states``), ``DefaultContinuousRecurrentModel`` :169-256,
``DefaultDiscreteRecurrentModel`` :259-341. Module names match, so reference
``state_dict``s load unchanged.

"""

from __future__ import annotations

from abc import abstractmethod
from typing import Any, Protocol

import torch
import torch.nn as nn

from ._utils import assert_1d_spec
from .data import DataKeys, Device
from .specs import Categorical, Composite, TensorSpec, Unbounded
from .tensordict import TensorDict


class RecurrentModel(nn.Module):
    """Observations + recurrent states -> distribution features, a value
    estimate (via :meth:`value_function`) and updated recurrent states.

    Both inputs of :meth:`forward` have batch shape ``[B, T, ...]``; features come
    back flattened to ``[B*T, ...]``, states as ``[B, ...]`` (no time dim).

    """

    #: Spec of the recurrent states (one ``[num_layers, hidden]`` block each).
    state_spec: Composite

    def __init__(self, observation_spec: TensorSpec, action_spec: TensorSpec, /, **config: Any) -> None:
        super().__init__()
        self.observation_spec = observation_spec
        self.action_spec = action_spec
        self.config = config

    @property
    def device(self) -> Device:
        return next(self.parameters()).device

    @staticmethod
    def default_model_cls(observation_spec: TensorSpec, action_spec: TensorSpec, /) -> type["RecurrentModel"]:
        if not isinstance(observation_spec, Unbounded):
            raise TypeError(f"Observation spec {observation_spec} has no default model support.")
        assert_1d_spec(observation_spec)
        assert_1d_spec(action_spec)
        if isinstance(action_spec, Unbounded):
            return DefaultContinuousRecurrentModel
        if isinstance(action_spec, Categorical):
            return DefaultDiscreteRecurrentModel
        raise TypeError(f"Action spec {action_spec} has no default model support.")

    @abstractmethod
    def forward(self, batch: TensorDict, states: TensorDict, /) -> tuple[TensorDict, TensorDict]:
        """``batch["obs"]`` ``[B, T, ...]`` and ``states`` ``[B, T, ...]`` (only
        ``[:, 0]`` is read) -> (features ``[B*T, ...]``, new states ``[B, ...]``)."""

    def init_states(self, n: int, /) -> TensorDict:
        """Initial recurrent states for ``n`` sequences (zeros)."""
        return self.state_spec.zero([n])

    def to(self, device: Device) -> "RecurrentModel":  # type: ignore[override]
        self.observation_spec = self.observation_spec.to(device)
        self.action_spec = self.action_spec.to(device)
        self.state_spec = self.state_spec.to(device)
        return nn.Module.to(self, device)

    @abstractmethod
    def value_function(self) -> torch.Tensor:
        """Value estimate ``[B*T, 1]`` of the most recent forward pass."""


class RecurrentModelFactory(Protocol):
    def __call__(self, observation_spec: TensorSpec, action_spec: TensorSpec, /, **config: Any) -> RecurrentModel:
        ...


GenericRecurrentModel = RecurrentModel


def _lstm_state_spec(num_layers: int, hidden_size: int, device: Any) -> Composite:
    return Composite(
        {
            DataKeys.HIDDEN_STATES: Unbounded(shape=torch.Size([num_layers, hidden_size]), device=device),
            DataKeys.CELL_STATES: Unbounded(shape=torch.Size([num_layers, hidden_size]), device=device),
        }
    )


def _small_head(in_dim: int, out_dim: int) -> nn.Linear:
    head = nn.Linear(in_dim, out_dim, bias=True)
    nn.init.uniform_(head.weight, a=-1e-3, b=1e-3)
    nn.init.zeros_(head.bias)
    return head


def _run_lstm(lstm: nn.LSTM, obs: torch.Tensor, states: TensorDict) -> tuple[torch.Tensor, TensorDict, int]:
    from .nn import fused_lstm

    h_first = states[DataKeys.HIDDEN_STATES][:, 0, ...]  # [B, layers, hidden]
    c_first = states[DataKeys.CELL_STATES][:, 0, ...]
    fused = fused_lstm.lstm_forward(lstm, obs, h_first[:, 0], c_first[:, 0]) if lstm.num_layers == 1 else None
    if fused is not None:
        latents, h_last, c_last = fused
        h_n, c_n = h_last.unsqueeze(0), c_last.unsqueeze(0)
    else:
        h_0 = h_first.permute(1, 0, 2).contiguous()
        c_0 = c_first.permute(1, 0, 2).contiguous()
        # PyTorch's own fused LSTM cell (GEMM + one pointwise kernel per step), not the
        # MIOpen RNN: on gfx950 the latter is 2x slower at these shapes and its
        # 32-bit-indexed workspace faults beyond 2^18 rows per call.
        with torch.backends.cudnn.flags(enabled=False):
            latents, (h_n, c_n) = lstm(obs, (h_0, c_0))
    new_states = TensorDict(
        {DataKeys.HIDDEN_STATES: h_n.permute(1, 0, 2), DataKeys.CELL_STATES: c_n.permute(1, 0, 2)},
        batch_size=obs.size(0),
    )
    return latents, new_states, obs.size(0)


def _run_lstm_heads(lstm: nn.LSTM, heads: list[nn.Linear], obs: torch.Tensor,
                    states: TensorDict) -> tuple[list[torch.Tensor], TensorDict]:
    """``[head(lstm(obs)) for head in heads]`` and the new states: a training pass as ONE fused autograd node where
    that applies (``fused_lstm.lstm_heads_forward``), else the LSTM (fused or the module) and then the heads."""
    from .nn import fused_lstm

    if lstm.num_layers == 1 and torch.is_grad_enabled():
        h_first = states[DataKeys.HIDDEN_STATES][:, 0, 0]
        c_first = states[DataKeys.CELL_STATES][:, 0, 0]
        fused = fused_lstm.lstm_heads_forward(lstm, heads, obs, h_first, c_first)
        if fused is not None:
            outs, _, h_last, c_last = fused
            new_states = TensorDict(
                {DataKeys.HIDDEN_STATES: h_last.unsqueeze(1), DataKeys.CELL_STATES: c_last.unsqueeze(1)},
                batch_size=obs.size(0),
            )
            return outs, new_states
    latents, new_states, _ = _run_lstm(lstm, obs, states)
    outs = fused_lstm.heads_forward(heads, latents)
    if outs is None:
        outs = [head(latents) for head in heads]
    return outs, new_states


class DefaultContinuousRecurrentModel(RecurrentModel):
    """LSTM + ``mean`` / ``log_std`` heads + a value head."""

    def __init__(
        self,
        observation_spec: Unbounded,
        action_spec: Unbounded,
        /,
        *,
        hidden_size: int = 256,
        num_layers: int = 1,
        bias: bool = True,
    ) -> None:
        super().__init__(observation_spec, action_spec)
        self.state_spec = _lstm_state_spec(num_layers, hidden_size, action_spec.device)
        self.lstm = nn.LSTM(observation_spec.shape[0], hidden_size, num_layers=num_layers, bias=bias, batch_first=True)
        self.action_mean = _small_head(hidden_size, action_spec.shape[0])
        self.action_log_std = _small_head(hidden_size, action_spec.shape[0])
        self.vf_model = nn.Linear(hidden_size, 1, bias=bias)
        self._value: None | torch.Tensor = None

    def forward(self, batch: TensorDict, states: TensorDict, /) -> tuple[TensorDict, TensorDict]:
        obs = batch[DataKeys.OBS]
        outs, new_states = _run_lstm_heads(self.lstm, [self.action_mean, self.action_log_std, self.vf_model], obs, states)
        action_mean = outs[0].reshape(-1, self.action_spec.shape[0])
        action_log_std = outs[1].reshape(-1, self.action_spec.shape[0])
        self._value = outs[2].reshape(-1, 1)
        return (
            TensorDict(
                {"mean": action_mean, "log_std": torch.tanh(action_log_std)},
                batch_size=action_mean.size(0),
                device=obs.device,
            ),
            new_states,
        )

    def to(self, device: Device) -> "DefaultContinuousRecurrentModel":  # type: ignore[override]
        self._value = None
        return super().to(device)  # type: ignore[return-value]

    def value_function(self) -> torch.Tensor:
        assert self._value is not None
        return self._value


class DefaultDiscreteRecurrentModel(RecurrentModel):
    """LSTM + a logits head + a value head."""

    def __init__(
        self,
        observation_spec: Unbounded,
        action_spec: Categorical,
        /,
        *,
        hidden_size: int = 256,
        num_layers: int = 1,
        bias: bool = True,
    ) -> None:
        super().__init__(observation_spec, action_spec)
        self.state_spec = _lstm_state_spec(num_layers, hidden_size, action_spec.device)
        self.lstm = nn.LSTM(observation_spec.shape[0], hidden_size, num_layers=num_layers, bias=bias, batch_first=True)
        self.feature_head = _small_head(hidden_size, action_spec.shape[0] * action_spec.space.n)
        self.vf_head = nn.Linear(hidden_size, 1, bias=bias)
        self._value: None | torch.Tensor = None

    def forward(self, batch: TensorDict, states: TensorDict, /) -> tuple[TensorDict, TensorDict]:
        obs = batch[DataKeys.OBS]
        outs, new_states = _run_lstm_heads(self.lstm, [self.feature_head, self.vf_head], obs, states)
        logits = outs[0].reshape(-1, self.action_spec.shape[0], self.action_spec.space.n)
        self._value = outs[1].reshape(-1, 1)
        return TensorDict({"logits": logits}, batch_size=logits.size(0), device=obs.device), new_states

    def to(self, device: Device) -> "DefaultDiscreteRecurrentModel":  # type: ignore[override]
        self._value = None
        return super().to(device)  # type: ignore[return-value]

    def value_function(self) -> torch.Tensor:
        assert self._value is not None
        return self._value
