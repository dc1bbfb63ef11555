"""``Policy`` = model + action distribution. Signature and semantics of
``sample`` follow the reference's ``src/rl8/policies/_feedforward.py:66-176``;
MLflow deployment wrappers are out of scope.

"""

from __future__ import annotations

import os
from typing import Any

import cloudpickle
import torch

from .data import DataKeys, Device
from .distributions import Distribution, NoiseStream
from .models import Model, ModelFactory
from .specs import TensorSpec
from .tensordict import TensorDict
from .views import ViewKind


class Policy:
    """The union of a feed-forward model and an action distribution.

    Args:
        observation_spec: Spec of environment observations / model inputs.
        action_spec: Spec of distribution outputs / environment inputs.
        model: Model instance (mutually exclusive with ``model_cls``).
        model_cls: Model class or factory.
        model_config: Keyword arguments of ``model_cls``.
        distribution_cls: Action distribution class.
        device: Device the policy lives on.

    """

    def __init__(
        self,
        observation_spec: TensorSpec,
        action_spec: TensorSpec,
        /,
        *,
        model: None | Model = None,
        model_cls: None | ModelFactory = None,
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
            model_cls = model_cls or Model.default_model_cls(observation_spec, action_spec)
            self.model = model_cls(observation_spec, action_spec, **self.model_config)
        else:
            self.model = model
        self.model = self.model.to(device)
        self.distribution_cls = distribution_cls or Distribution.default_dist_cls(action_spec)
        #: Philox address of this policy's draws; ``None`` -> process default.
        self.noise_stream: None | NoiseStream = None
        #: Noise injected into the next ``sample`` call (parity tests).
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

    def to(self, device: Device, /) -> "Policy":
        self.model = self.model.to(device)
        return self

    def sample(
        self,
        batch: TensorDict,
        /,
        *,
        kind: ViewKind = "last",
        deterministic: bool = False,
        inplace: bool = False,
        requires_grad: bool = False,
        return_actions: bool = True,
        return_logp: bool = False,
        return_values: bool = False,
        return_views: bool = False,
    ) -> TensorDict:
        """Run the model on ``batch`` (``[B, T, ...]``) and, optionally, sample
        actions / log-probabilities / values.

        ``kind="last"`` uses only the most recent timestep of ``batch``;
        ``kind="all"`` flattens ``B`` and ``T``. ``deterministic`` puts the model
        in eval mode and takes the distribution's mode. Outputs go into a new
        tensordict of batch size ``[B]`` or ``[B*T]`` unless ``inplace``.

        """
        if DataKeys.VIEWS in batch.keys():
            in_batch = batch[DataKeys.VIEWS]
        else:
            in_batch = self.model.apply_view_requirements(batch, kind=kind)

        training = self.model.training
        if deterministic == training:
            self.model.train(not training)
        prev = torch.is_grad_enabled()
        torch.set_grad_enabled(requires_grad)
        try:
            features = self.model(in_batch)
            out = (
                batch.reshape(-1)
                if inplace
                else TensorDict({}, batch_size=in_batch.batch_size, device=batch.device)
            )
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
            if return_views:
                out[DataKeys.VIEWS] = in_batch
        finally:
            torch.set_grad_enabled(prev)
            if deterministic == training:
                self.model.train(training)
        return out

    def save(self, path: str | os.PathLike[str], /) -> None:
        """Cloud-pickle the policy to ``path``."""
        with open(path, "wb") as f:
            cloudpickle.dump(self, f)
