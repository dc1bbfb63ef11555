"""``RecurrentPolicy`` = recurrent model + action distribution; ``sample``
follows the reference's ``src/rl8/policies/_recurrent.py:68-164``."""

from __future__ import annotations

import os
from typing import Any

import cloudpickle
import torch

from .data import DataKeys, Device
from .distributions import Distribution, NoiseStream
from .models_recurrent import RecurrentModel, RecurrentModelFactory
from .specs import Composite, TensorSpec
from .tensordict import TensorDict


class RecurrentPolicy:
    """The union of a recurrent model and an action distribution (constructor
    arguments as :class:`rl8_amd.policies.Policy`)."""

    def __init__(
        self,
        observation_spec: TensorSpec,
        action_spec: TensorSpec,
        /,
        *,
        model: None | RecurrentModel = None,
        model_cls: None | RecurrentModelFactory = None,
        model_config: None | dict[str, Any] = None,
        distribution_cls: None | type[Distribution] = None,
        device: Device = "cpu",
    ) -> None:
        self.model_config = model_config or {}
        if model and model_cls:
            raise ValueError(
                "`model` and `model_cls` args are mutually exclusive."
                "Provide one or the other, but not both."
            )
        if model is None:
            model_cls = model_cls or RecurrentModel.default_model_cls(observation_spec, action_spec)
            self.model = model_cls(observation_spec, action_spec, **self.model_config)
        else:
            self.model = model
        self.model = self.model.to(device)
        self.distribution_cls = distribution_cls or Distribution.default_dist_cls(action_spec)
        self.noise_stream: None | NoiseStream = None
        self.injected_noise: None | torch.Tensor = None

    @property
    def action_spec(self) -> TensorSpec:
        return self.model.action_spec

    @property
    def device(self) -> Device:
        return self.model.device

    @property
    def observation_spec(self) -> TensorSpec:
        return self.model.observation_spec

    @property
    def state_spec(self) -> Composite:
        return self.model.state_spec

    def init_states(self, n: int, /) -> TensorDict:
        """New recurrent states for ``n`` sequences."""
        return self.model.init_states(n)

    def to(self, device: Device, /) -> "RecurrentPolicy":
        self.model = self.model.to(device)
        return self

    def sample(
        self,
        batch: TensorDict,
        /,
        states: None | TensorDict = None,
        *,
        deterministic: bool = False,
        inplace: bool = False,
        requires_grad: bool = False,
        return_actions: bool = True,
        return_logp: bool = False,
        return_values: bool = False,
    ) -> tuple[TensorDict, TensorDict]:
        """Run the model on ``batch`` / ``states`` (both ``[B, T, ...]``) and,
        optionally, sample actions / log-probabilities / values. Returns outputs of
        batch size ``[B*T]`` and the updated states of batch size ``[B]``."""
        training = self.model.training
        if deterministic == training:
            self.model.train(not training)
        prev = torch.is_grad_enabled()
        torch.set_grad_enabled(requires_grad)
        try:
            B, T = batch.batch_size
            states = self.model.init_states(B).reshape(B, 1) if states is None else states
            features, out_states = self.model(batch, states)
            out = batch.reshape(B * T) if inplace else TensorDict({}, batch_size=B * T, device=batch.device)
            out[DataKeys.FEATURES] = features
            if return_actions:
                dist = self.distribution_cls(features, self.model)
                dist.noise_stream = self.noise_stream
                dist.noise, self.injected_noise = self.injected_noise, None
                actions, logp = dist.sample_with_logp(deterministic=deterministic)
                out[DataKeys.ACTIONS] = actions
                if return_logp:
                    out[DataKeys.LOGP] = logp
            if return_values:
                out[DataKeys.VALUES] = self.model.value_function()
        finally:
            torch.set_grad_enabled(prev)
            if deterministic == training:
                self.model.train(training)
        return out, out_states

    def save(self, path: str | os.PathLike[str], /) -> None:
        with open(path, "wb") as f:
            cloudpickle.dump(self, f)
