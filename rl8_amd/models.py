"""Policy / value networks. Interfaces follow the reference's
``src/rl8/models/_feedforward.py`` (``Model`` :20-203, ``DefaultContinuousModel``
:234-310, ``DefaultDiscreteModel`` :313-383) and ``src/rl8/nn/modules/mlp.py``.

The arithmetic stays on PyTorch-ROCm (rocBLAS / hipBLASLt GEMMs), as the
north_star specifies: this build's hand-written kernels are the memory-bound
ops around the networks, not the networks. Module structure (and therefore
``state_dict`` keys) matches the reference so its checkpoints load unchanged.

"""

from __future__ import annotations

from abc import abstractmethod
from typing import Any, Protocol, Sequence

import torch
import torch.nn as nn

from ._utils import assert_1d_spec
from .data import DataKeys, Device
from .specs import Categorical, TensorSpec, Unbounded
from .tensordict import TensorDict
from .views import ViewKind, ViewRequirement

ACTIVATIONS: dict[str, type[nn.Module]] = {
    "elu": nn.ELU,
    "gelu": nn.GELU,
    "identity": nn.Identity,
    "leaky_relu": nn.LeakyReLU,
    "relu": nn.ReLU,
    "relu6": nn.ReLU6,
    "selu": nn.SELU,
    "sigmoid": nn.Sigmoid,
    "swish": nn.SiLU,
    "tanh": nn.Tanh,
}


def get_activation(name: str, /, **params: Any) -> nn.Module:
    """Activation module by name."""
    return ACTIVATIONS[name](**params)


class MLP(nn.Sequential):
    """``Linear -> act -> ... -> Linear`` (no trailing activation)."""

    def __init__(
        self,
        input_dim: int,
        hiddens: Sequence[int],
        /,
        *,
        activation_fn: str = "relu",
        norm_layer: None | type[nn.Module] = None,
        bias: bool = True,
        dropout: float = 0.0,
        inplace: bool = False,
    ) -> None:
        params = {"inplace": inplace} if inplace else {}
        layers: list[nn.Module] = []
        in_dim = input_dim
        for hidden_dim in hiddens[:-1]:
            layers.append(nn.Linear(in_dim, hidden_dim, bias=bias))
            if norm_layer is not None:
                layers.append(norm_layer(hidden_dim))
            layers.append(get_activation(activation_fn, **params))
            if dropout:
                layers.append(nn.Dropout(p=dropout))
            in_dim = hidden_dim
        layers.append(nn.Linear(in_dim, hiddens[-1], bias=bias))
        super().__init__(*layers)


class Model(nn.Module):
    """Feed-forward policy component: observations -> distribution features,
    plus a value estimate readable through :meth:`value_function` after each
    forward pass.

    Args:
        observation_spec: Spec of the forward-pass input.
        action_spec: Spec of the action distribution's output.
        config: Model-specific configuration.

    """

    #: How buffer leaves are turned into forward-pass inputs.
    view_requirements: dict[str, ViewRequirement]

    def __init__(self, observation_spec: TensorSpec, action_spec: TensorSpec, /, **config: Any) -> None:
        super().__init__()
        self.observation_spec = observation_spec
        self.action_spec = action_spec
        self.config = config
        self.view_requirements = {DataKeys.OBS: ViewRequirement(shift=0)}

    @property
    def device(self) -> Device:
        return next(self.parameters()).device

    def apply_view_requirements(self, batch: TensorDict, /, *, kind: ViewKind = "last") -> TensorDict:
        """``[B, T, ...]`` batch -> model input (``"last"``: ``[B, ...]``,
        ``"all"``: ``[B*T, ...]``)."""
        out = {}
        batch_size = None
        for key, view_requirement in self.view_requirements.items():
            if kind == "all":
                item = view_requirement.apply_all(key, batch)
            elif kind == "last":
                item = view_requirement.apply_last(key, batch)
            else:
                raise ValueError(f"Unknown view kind {kind!r}.")
            out[key] = item
            if batch_size is None:
                batch_size = item.size(0)
        return TensorDict(out, batch_size=batch_size, device=batch.device)

    @staticmethod
    def default_model_cls(observation_spec: TensorSpec, action_spec: TensorSpec, /) -> type["Model"]:
        if not isinstance(observation_spec, Unbounded):
            raise TypeError(f"Observation spec {observation_spec} has no default model support.")
        assert_1d_spec(observation_spec)
        assert_1d_spec(action_spec)
        if isinstance(action_spec, Unbounded):
            return DefaultContinuousModel
        if isinstance(action_spec, Categorical):
            return DefaultDiscreteModel
        raise TypeError(f"Action spec {action_spec} has no default model support.")

    @property
    def drop_size(self) -> int:
        return next(iter(v.drop_size for v in self.view_requirements.values()))

    @abstractmethod
    def forward(self, batch: TensorDict, /) -> TensorDict:
        """``batch["obs"]`` ``[B, ...]`` -> features for the action distribution."""

    def to(self, device: Device) -> "Model":  # type: ignore[override]
        self.observation_spec = self.observation_spec.to(device)
        self.action_spec = self.action_spec.to(device)
        return nn.Module.to(self, device)

    @abstractmethod
    def value_function(self) -> torch.Tensor:
        """Value estimate ``[B, 1]`` of the most recent forward pass."""

    def validate_view_requirements(self) -> None:
        drop_sizes = {k: v.drop_size for k, v in self.view_requirements.items()}
        if len(set(drop_sizes.values())) > 1:
            raise RuntimeError(
                f"{self} view requirements with drop sizes {drop_sizes} result in an"
                " ambiguous batch size."
            )


class ModelFactory(Protocol):
    def __call__(self, observation_spec: TensorSpec, action_spec: TensorSpec, /, **config: Any) -> Model:
        ...


GenericModel = Model


def _tower(input_dim: int, hiddens: Sequence[int], activation_fn: str, bias: bool) -> list[nn.Module]:
    return [
        MLP(input_dim, hiddens, activation_fn=activation_fn, bias=bias, inplace=False),
        get_activation(activation_fn),
    ]


def _small_head(in_dim: int, out_dim: int) -> nn.Linear:
    head = nn.Linear(in_dim, out_dim, bias=True)
    nn.init.uniform_(head.weight, a=-1e-3, b=1e-3)
    nn.init.zeros_(head.bias)
    return head


class _MeanAndLogStd(torch.autograd.Function):
    """``(out[:, :a], tanh(out[:, a:]))`` of a fused tower's ``[M, 2a]`` output as two DENSE tensors (what the loss
    kernel reads), with a backward that hands the tower ONE dense ``[M, 2a]`` gradient: ``cat([g_mean, g_ls (1 -
    ls^2)])`` -- tanh's own backward formula, bit for bit.  Plain slicing costs, per SGD pass at 2^25 rows, two
    slice-backwards (a zero fill + a strided copy of [M, 2a] each), their add, a tanh backward and two ``contiguous()``
    clones: 1.2 ms of elementwise launches (4.8 ms of config 4's 254 per step; tools/diag/feedforward_aten_ops.py)."""

    @staticmethod
    def forward(ctx, out: torch.Tensor, a: int):  # type: ignore[override]
        mean = out[:, :a].contiguous()
        log_std = torch.tanh(out[:, a:])
        ctx.save_for_backward(log_std)
        return mean, log_std

    @staticmethod
    def backward(ctx, g_mean, g_ls):  # type: ignore[override]
        (log_std,) = ctx.saved_tensors
        if g_mean is None:
            g_mean = torch.zeros_like(log_std)
        g_raw = torch.zeros_like(log_std) if g_ls is None else g_ls * (1 - log_std * log_std)
        return torch.cat([g_mean, g_raw], 1), None


class DefaultContinuousModel(Model):
    """1D observations -> ``mean`` / ``log_std`` of a normal, and a separate
    value tower."""

    def __init__(
        self,
        observation_spec: Unbounded,
        action_spec: Unbounded,
        /,
        *,
        hiddens: Sequence[int] = (256, 256),
        activation_fn: str = "relu",
        bias: bool = True,
    ) -> None:
        super().__init__(observation_spec, action_spec)
        obs_dim = observation_spec.shape[0]
        self.latent_model = nn.Sequential(*_tower(obs_dim, hiddens, activation_fn, bias))
        self.action_mean = _small_head(hiddens[-1], action_spec.shape[0])
        self.action_log_std = _small_head(hiddens[-1], action_spec.shape[0])
        self.vf_model = nn.Sequential(
            *_tower(obs_dim, hiddens, activation_fn, bias), nn.Linear(hiddens[-1], 1)
        )
        self._value: None | torch.Tensor = None

    def forward(self, batch: TensorDict, /) -> TensorDict:
        from .nn import fused_mlp

        obs = batch[DataKeys.OBS]
        fused = fused_mlp.tower_forward(self.latent_model, [self.action_mean, self.action_log_std], obs)
        if fused is not None:
            action_mean, log_std = _MeanAndLogStd.apply(fused, self.action_mean.out_features)
        else:
            latents = self.latent_model(obs)
            action_mean = self.action_mean(latents)
            log_std = torch.tanh(self.action_log_std(latents))
        value = fused_mlp.tower_forward(self.vf_model[:2], [self.vf_model[2]], obs)
        self._value = value if value is not None else self.vf_model(obs)
        return TensorDict(
            {"mean": action_mean, "log_std": log_std},
            batch_size=batch.batch_size,
            device=obs.device,
        )

    def to(self, device: Device) -> "DefaultContinuousModel":  # type: ignore[override]
        self._value = None
        return super().to(device)  # type: ignore[return-value]

    def value_function(self) -> torch.Tensor:
        assert self._value is not None
        return self._value


class DefaultDiscreteModel(Model):
    """1D observations -> ``logits`` ``[B, A, K]``, and a separate value tower."""

    def __init__(
        self,
        observation_spec: Unbounded,
        action_spec: Categorical,
        /,
        *,
        hiddens: Sequence[int] = (256, 256),
        activation_fn: str = "relu",
        bias: bool = True,
    ) -> None:
        super().__init__(observation_spec, action_spec)
        obs_dim = observation_spec.shape[0]
        self.feature_model = nn.Sequential(*_tower(obs_dim, hiddens, activation_fn, bias))
        self.feature_model.append(
            _small_head(hiddens[-1], action_spec.shape[0] * action_spec.space.n)
        )
        self.vf_model = nn.Sequential(
            *_tower(obs_dim, hiddens, activation_fn, bias), nn.Linear(hiddens[-1], 1)
        )
        self._value: None | torch.Tensor = None

    def forward(self, batch: TensorDict, /) -> TensorDict:
        from .nn import fused_mlp

        obs = batch[DataKeys.OBS]
        # (two actions: under the fused PPO loss the two logit gradients are exact negatives, and Algorithm says so)
        two_way = self.action_spec.shape[0] == 1 and self.action_spec.space.n == 2
        logits = fused_mlp.tower_forward(self.feature_model[:2], [self.feature_model[2]], obs,
                                         pair_gradients=True if two_way and fused_mlp.pair_hint() else None)
        if logits is None:
            logits = self.feature_model(obs)
        logits = logits.reshape(-1, self.action_spec.shape[0], self.action_spec.space.n)
        value = fused_mlp.tower_forward(self.vf_model[:2], [self.vf_model[2]], obs)
        self._value = value if value is not None else self.vf_model(obs)
        return TensorDict({"logits": logits}, batch_size=batch.batch_size, device=obs.device)

    def to(self, device: Device) -> "DefaultDiscreteModel":  # type: ignore[override]
        self._value = None
        return super().to(device)  # type: ignore[return-value]

    def value_function(self) -> torch.Tensor:
        assert self._value is not None
        return self._value
