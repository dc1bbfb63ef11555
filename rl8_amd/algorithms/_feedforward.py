"""Feed-forward PPO: ``AlgorithmConfig`` / ``Algorithm.collect()`` / ``.step()``.

The API -- config fields and defaults, ``collect(*, env_config, deterministic)
-> CollectStats``, ``step() -> StepStats``, the ``RuntimeError`` when stepping
unbuffered, the stat keys -- follows the reference's
``src/rl8/algorithms/_feedforward.py`` (``AlgorithmConfig`` :29-179,
``Algorithm.__init__`` :206-299, ``collect`` :301-441, ``step`` :443-615,
``validate`` :617-697). The machinery underneath is MI355X-first:

* the rollout buffer is stored TIME-MAJOR (``[H+1, N, d]``; exposed as the
  reference's ``[N, H+1, d]`` through transposed views), so every per-timestep
  column is one contiguous slab: all rollout writes, the GAE scan, the stats
  pass and the flattening for training are coalesced and copy-free;
* per timestep, ``collect()`` issues the policy network (PyTorch-ROCm) and ONE
  fused HIP launch (sample + env.step + bookkeeping) for built-in envs;
* ``collect()`` synchronises with the host once (the reference: 9 times),
  ``step()`` once per call unless KL early stopping needs the value sooner;
* GAE / normalisation / PPO loss forward+backward / minibatch gather are single
  HIP launches; the loss kernel hands finished gradients to autograd;
* a full-buffer minibatch is never shuffled or gathered (its mean does not
  depend on order) and is streamed through the network in row chunks that
  accumulate into one optimizer step, so 2^20 x 32 samples fit 288 GB of HBM;
* with ``torch.distributed`` initialised, environments are sharded across
  ranks (``rl8_amd.parallel.EnvShards``).

"""

from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass
from typing import Any, Literal

import torch
import torch.amp as amp
import torch.nn as nn
import torch.optim as optim
from torch.amp.grad_scaler import GradScaler

from .. import hip
from .._utils import StatTracker, assert_nd_spec, memory_stats, profile_ms
from ..data import (
    AlgorithmHparams,
    AlgorithmState,
    CollectStats,
    DataKeys,
    Device,
    MemoryStats,
    StepStats,
)
from ..distributions import Categorical, Distribution, NoiseStream, Normal, SquashedNormal
from ..env import Env, EnvFactory
from ..models import Model, ModelFactory
from ..nn import fused_mlp
from ..nn.functional import fused_ppo_loss, gae_launch, has_fused_loss, losses_from_sums
from ..parallel import EnvShards
from ..policies import Policy
from ..schedulers import EntropyScheduler, LRScheduler, ScheduleKind
from ..specs import Composite, Unbounded
from ..tensordict import TensorDict

#: Rows pushed through the policy network per forward/backward pass inside one
#: minibatch.  The bf16-plane towers keep 1 KiB (h2) + 32 B (gate bits) per row and
#: tower between forward and backward: 2^25 rows -- BASELINE's whole 33.5 M-sample
#: batch in one pass -- peak at 68 GiB of the 288 GB, and fewer, larger launches
#: measured 2.4 % faster than 2^23-row passes (1.2 % at 2^24).
DEFAULT_MAX_ROWS_PER_PASS = 1 << 25


@dataclass
class AlgorithmConfig:
    """Configuration of a feed-forward PPO algorithm (fields and defaults as in
    the reference)."""

    #: Model instance (mutually exclusive with ``model_cls``).
    model: None | Model = None
    #: Model class / factory; inferred from the env specs when omitted.
    model_cls: None | ModelFactory = None
    #: Keyword arguments of ``model_cls``.
    model_config: None | dict[str, Any] = None
    #: Action distribution class; inferred from the action spec when omitted.
    distribution_cls: None | type[Distribution] = None
    #: Transitions per env collected by each ``collect()``.
    horizon: int = 32
    #: ``collect()`` calls between env resets (negative: reset only once).
    horizons_per_env_reset: int = 1
    #: Number of parallel environments.
    num_envs: int = 8192
    #: Optimizer class.
    optimizer_cls: type[optim.Optimizer] = optim.Adam
    #: Optimizer keyword arguments (default ``{"lr": 1e-3}``).
    optimizer_config: None | dict[str, Any] = None
    #: Accumulate gradients over the minibatches of an SGD iteration.
    accumulate_grads: bool = False
    #: Automatic mixed precision for the network passes.
    enable_amp: bool = False
    #: Learning-rate schedule over env transitions.
    lr_schedule: None | list[tuple[int, float]] = None
    lr_schedule_kind: ScheduleKind = "step"
    #: Entropy bonus weight (ignored when a schedule is given).
    entropy_coeff: float = 0.0
    entropy_coeff_schedule: None | list[tuple[int, float]] = None
    entropy_coeff_schedule_kind: ScheduleKind = "step"
    #: GAE lambda.
    gae_lambda: float = 0.95
    #: Discount factor.
    gamma: float = 0.95
    #: Minibatch size (``None``: the whole buffer).
    sgd_minibatch_size: None | int = None
    #: SGD passes over the buffer per ``step()``.
    num_sgd_iters: int = 4
    #: Shuffle samples into minibatches.
    shuffle_minibatches: bool = True
    #: PPO ratio clip.
    clip_param: float = 0.2
    #: Value-loss clip.
    vf_clip_param: float = 5.0
    #: Dual clip for negative advantages (``None``: off).
    dual_clip_param: None | float = None
    #: Value-loss weight.
    vf_coeff: float = 1.0
    #: Stop updating when approximate KL exceeds 1.5x this (``None``: never).
    target_kl_div: None | float = None
    #: Gradient-norm clip.
    max_grad_norm: float = 5.0
    #: Standardise advantages.
    normalize_advantages: bool = True
    #: Scale rewards by the std of reversed discounted returns.
    normalize_rewards: bool = True
    #: Device; ``"auto"`` picks the HIP device.
    device: Device | Literal["auto"] = "auto"

    def build(self, env_cls: EnvFactory) -> "Algorithm":
        """Build and validate an :class:`Algorithm`."""
        algo = Algorithm(env_cls, config=self)
        algo.validate()
        return algo


def _resolve_device(requested: Any) -> str:
    """The reference picks ``"cuda"`` whenever one is visible
    (``_feedforward.py:210-216``). This build runs on a HIP device only."""
    if not torch.cuda.is_available():
        raise hip.HipExtensionError(
            "rl8_amd needs a HIP device (torch.cuda.is_available() is False); it has"
            " no CPU path. The CPU restatement used for parity checks lives in oracle/."
        )
    hip.load()
    if requested in ("auto", "cuda", None) or str(requested) == "cpu":
        if str(requested) == "cpu":
            raise hip.HipExtensionError(
                "device='cpu' requested, but rl8_amd has no CPU path (see oracle/ for"
                " the CPU restatement used as the parity checker)."
            )
        return f"cuda:{torch.cuda.current_device()}"
    return str(requested)


class Algorithm:
    """Feed-forward PPO with the usual stabilising tricks.

    Args:
        env_cls: Parallel environment class / factory; stepped ``horizon``
            times per :meth:`collect`.
        config: Algorithm configuration.

    Examples:
        >>> from rl8_amd import AlgorithmConfig
        >>> from rl8_amd.env import DiscreteDummyEnv
        >>> algo = AlgorithmConfig().build(DiscreteDummyEnv)  # doctest: +SKIP
        >>> algo.collect()  # doctest: +SKIP
        >>> algo.step()  # doctest: +SKIP

    """

    buffer: TensorDict
    buffer_spec: Composite
    entropy_scheduler: EntropyScheduler
    env: Env
    grad_scaler: GradScaler
    hparams: AlgorithmHparams
    lr_scheduler: LRScheduler
    optimizer: optim.Optimizer
    policy: Policy
    state: AlgorithmState

    def __init__(self, env_cls: EnvFactory, /, config: None | AlgorithmConfig = None) -> None:
        config = config or AlgorithmConfig()
        device = _resolve_device(config.device)
        self.shards = EnvShards()
        max_num_envs = env_cls.max_num_envs if hasattr(env_cls, "max_num_envs") else config.num_envs
        num_envs = min(config.num_envs, max_num_envs)
        max_horizon = env_cls.max_horizon if hasattr(env_cls, "max_horizon") else 1_000_000
        horizon = min(config.horizon, max_horizon)
        if self.shards.active:
            if num_envs % self.shards.world_size:
                raise ValueError("`num_envs` must be divisible by the number of ranks.")
            local_envs = num_envs // self.shards.world_size
        else:
            local_envs = num_envs
        self.local_num_envs = local_envs
        self.env = env_cls(local_envs, horizon, device=device)
        self.env.env_offset = self.shards.env_offset(local_envs)
        assert_nd_spec(self.env.observation_spec)
        assert_nd_spec(self.env.action_spec)
        self.policy = self._make_policy(config, device)
        self.shards.broadcast_parameters_(self.policy.model)
        self.noise = NoiseStream()
        self.noise.row_offset = self.env.env_offset
        self.policy.noise_stream = self.noise
        self.buffer_spec = Composite(
            {
                DataKeys.OBS: self.env.observation_spec,
                **self._extra_buffer_specs(),
                DataKeys.REWARDS: Unbounded(1, device=device),
                DataKeys.ACTIONS: self.env.action_spec,
                DataKeys.LOGP: Unbounded(1, device=device),
                DataKeys.VALUES: Unbounded(1, device=device),
                DataKeys.ADVANTAGES: Unbounded(1, device=device),
                DataKeys.RETURNS: Unbounded(1, device=device),
            },
        )
        if config.normalize_rewards:
            self.buffer_spec.set(DataKeys.REVERSED_DISCOUNTED_RETURNS, Unbounded(1, device=device))
        self.buffer_spec = self.buffer_spec.to(device)
        self._allocate_buffer(local_envs, horizon)
        optimizer_config = config.optimizer_config or {"lr": 1e-3}
        optimizer = config.optimizer_cls(self.policy.model.parameters(), **optimizer_config)
        self.lr_scheduler = LRScheduler(
            optimizer, schedule=config.lr_schedule, kind=config.lr_schedule_kind
        )
        self.entropy_scheduler = EntropyScheduler(
            config.entropy_coeff,
            schedule=config.entropy_coeff_schedule,
            kind=config.entropy_coeff_schedule_kind,
        )
        sgd_minibatch_size = (
            config.sgd_minibatch_size
            if config.sgd_minibatch_size
            else self._default_minibatch_size(config, num_envs, horizon)
        )
        self.hparams = self._make_hparams(
            accumulate_grads=config.accumulate_grads,
            clip_param=config.clip_param,
            device=device,
            dual_clip_param=config.dual_clip_param,
            enable_amp=config.enable_amp,
            gae_lambda=config.gae_lambda,
            gamma=config.gamma,
            horizon=horizon,
            horizons_per_env_reset=config.horizons_per_env_reset,
            max_grad_norm=config.max_grad_norm,
            normalize_advantages=config.normalize_advantages,
            normalize_rewards=config.normalize_rewards,
            num_envs=num_envs,
            num_sgd_iters=config.num_sgd_iters,
            sgd_minibatch_size=sgd_minibatch_size,
            shuffle_minibatches=config.shuffle_minibatches,
            target_kl_div=config.target_kl_div,
            vf_clip_param=config.vf_clip_param,
            vf_coeff=config.vf_coeff,
        ).validate()
        if self.shards.active and sgd_minibatch_size % self.shards.world_size:
            raise ValueError("`sgd_minibatch_size` must be divisible by the number of ranks.")
        self.state = self._make_state()
        self.optimizer = optimizer
        self.grad_scaler = GradScaler(device="cuda", enabled=config.enable_amp)
        self.max_rows_per_pass = DEFAULT_MAX_ROWS_PER_PASS
        #: Keep what the towers compute during ``collect()`` (outputs + gate bits) and let the SGD passes that
        #: still see the rollout's weights -- iteration 0 up to the first optimizer step -- start from it instead
        #: of running the forward kernels again on the same rows (``fused_mlp.RolloutRecord``; bit-identical).
        self.reuse_rollout_forward = os.environ.get("RL8_AMD_REUSE_ROLLOUT", "1") != "0"
        self._record: None | fused_mlp.RolloutRecord = None
        self._record_obs_version = -1
        self._batch_rows: None | tuple[str, None | torch.Tensor] = None
        #: Parity hooks: noise / permutations recorded from a reference run.
        self.injected_noise: None | torch.Tensor = None  # [H, N, ...]
        self.injected_permutations: None | list[torch.Tensor] = None

    # ------------------------------------------------------------------ #
    # Construction hooks (overridden by the recurrent algorithm).
    # ------------------------------------------------------------------ #
    def _make_policy(self, config: Any, device: str) -> Any:
        return Policy(
            self.env.observation_spec,
            self.env.action_spec,
            model=config.model,
            model_cls=config.model_cls,
            model_config=config.model_config,
            distribution_cls=config.distribution_cls,
            device=device,
        )

    def _extra_buffer_specs(self) -> dict[str, Any]:
        return {}

    def _default_minibatch_size(self, config: Any, num_envs: int, horizon: int) -> int:
        return num_envs * horizon

    def _make_hparams(self, **common: Any) -> AlgorithmHparams:
        return AlgorithmHparams(**common)

    def _make_state(self) -> AlgorithmState:
        return AlgorithmState()

    # ------------------------------------------------------------------ #
    # Buffer: time-major storage, env-major views.
    # ------------------------------------------------------------------ #
    def _allocate_buffer(self, num_envs: int, horizon: int) -> None:
        obs_spec = self.env.observation_spec
        if isinstance(obs_spec, Composite) or isinstance(self.env.action_spec, Composite):
            raise NotImplementedError(
                "rl8_amd's rollout buffer holds tensor observation / action specs"
                " (composite specs are outside the accelerated path)."
            )
        #: key -> ``[H+1, N, ...]`` storage of the flat leaves.
        self._tm: dict[str, torch.Tensor] = {}
        #: recurrent-state leaves (``buffer["states"][k]``), same layout.
        self._tm_states: dict[str, torch.Tensor] = {}

        def slab(spec: Any) -> torch.Tensor:
            return torch.zeros(horizon + 1, num_envs, *spec.shape, dtype=spec.dtype, device=spec.device)

        views: dict[str, Any] = {}
        device = None
        for key in self.buffer_spec:
            spec = self.buffer_spec[key]
            if isinstance(spec, Composite):
                if key != DataKeys.STATES:
                    raise NotImplementedError(f"composite buffer leaf {key!r} is not supported")
                nested = {}
                for sub in spec:
                    storage = slab(spec[sub])
                    self._tm_states[sub] = storage
                    nested[sub] = storage.transpose(0, 1)
                views[key] = TensorDict(nested, batch_size=[num_envs, horizon + 1])
            else:
                storage = slab(spec)
                self._tm[key] = storage
                views[key] = storage.transpose(0, 1)
                device = storage.device
        self.buffer = TensorDict(views, batch_size=[num_envs, horizon + 1], device=device)

    def _release_step_caches(self) -> None:
        """What was kept for the SGD iterations of the step() that is ending (subclasses add theirs)."""
        from ..nn import fused_mlp

        fused_mlp._TRUSTED_PAIRS.clear()  # (a loss gradient nobody ran a backward on)
        record = getattr(self, "_record", None)
        if record is not None and record.towers:  # (the 1-KiB-per-row slabs go first when memory is short)
            record.release_if_tight(self._tm[DataKeys.LOGP].device)

    def _reset_buffer(self) -> None:
        """``buffer_spec.zero(...)`` then ``obs[:, -1] = final_obs`` (and the final
        recurrent states) of the reference (:603-609), in place: storage is
        reused, not re-allocated."""
        h = self.hparams.horizon
        for key, storage in self._tm.items():
            if key == DataKeys.OBS:
                storage[:h].zero_()
            else:
                storage.zero_()
        for storage in self._tm_states.values():
            storage[:h].zero_()

    # ------------------------------------------------------------------ #
    # Small accessors shared with the reference's base class.
    # ------------------------------------------------------------------ #
    @property
    def horizons_per_env_reset(self) -> int:
        return self.hparams.horizons_per_env_reset

    def memory_stats(self) -> MemoryStats:
        return memory_stats(self.hparams.device_type)

    @property
    def params(self) -> dict[str, Any]:
        from dataclasses import asdict

        return {
            "env_cls": self.env.__class__.__name__,
            "model_cls": self.policy.model.__class__.__name__,
            "distribution_cls": self.policy.distribution_cls.__name__,
            "optimizer_cls": self.optimizer.__class__.__name__,
            "entropy_coeff": self.entropy_scheduler.coeff,
            **asdict(self.hparams),
        }

    # ------------------------------------------------------------------ #
    # collect()
    # ------------------------------------------------------------------ #
    def _fusable(self) -> bool:
        """One-launch-per-timestep path: a built-in env exposing
        ``fused_rollout_step`` with the distribution its kernel implements."""
        if not hasattr(self.env, "fused_rollout_step") or not self._identity_views():
            return False
        dist_cls = self.policy.distribution_cls
        supported = getattr(self.env, "fused_distributions", (Categorical, Normal, SquashedNormal))
        return dist_cls in supported and has_fused_loss(dist_cls)

    def _identity_views(self) -> bool:
        """The model reads ``obs`` as is (no rolling windows): observations can be
        handed to it as ``[N, ...]`` slabs without going through the view
        requirements."""
        views = getattr(self.policy.model, "view_requirements", None)
        if views is None:  # recurrent models take observations as they are
            return True
        return set(views) == {DataKeys.OBS} and all(v.is_identity for v in views.values())

    def _forward(self, obs: torch.Tensor, *, deterministic: bool) -> tuple[TensorDict, torch.Tensor]:
        """Policy network on a ``[N, obs...]`` slab -> (features, values)."""
        sample = self.policy.sample(
            TensorDict({DataKeys.VIEWS: TensorDict({DataKeys.OBS: obs}, batch_size=obs.shape[0])},
                       batch_size=obs.shape[0]),
            kind="last",
            deterministic=deterministic,
            inplace=False,
            requires_grad=False,
            return_actions=False,
            return_logp=False,
            return_values=True,
            return_views=False,
        )
        return sample[DataKeys.FEATURES], sample[DataKeys.VALUES]

    def collect(
        self,
        *,
        env_config: None | dict[str, Any] = None,
        deterministic: bool = False,
    ) -> CollectStats:
        """Roll the policy out for ``horizon`` steps in every environment,
        filling the buffer; returns summary statistics and sets the
        ``buffered`` flag :meth:`step` requires.

        The environment is reset first according to ``horizons_per_env_reset``;
        otherwise the last observation carries over.

        """
        hp = self.hparams
        H = hp.horizon
        tm = self._tm
        rdr = tm.get(DataKeys.REVERSED_DISCOUNTED_RETURNS)
        with profile_ms() as collect_timer:
            env_was_reset = False
            carry = (self.state.horizons and hp.horizons_per_env_reset < 0) or (
                self.state.horizons % hp.horizons_per_env_reset
            )
            if carry:
                tm[DataKeys.OBS][0].copy_(tm[DataKeys.OBS][H])
                if rdr is not None:
                    rdr[0].copy_(rdr[H])
            else:
                tm[DataKeys.OBS][0].copy_(self.env.reset(config=env_config))
                env_was_reset = True
                if rdr is not None:
                    rdr[0].zero_()

            fused = self._fusable()
            gamma = float(torch.tensor(hp.gamma, dtype=torch.float32))
            record = self._rollout_record() if fused else None
            self._record_obs_version = -1
            dist_cls = self.policy.distribution_cls
            pair = (fused_mlp.expect_pair_gradients() if record is not None and issubclass(dist_cls, Categorical)
                    else contextlib.nullcontext())
            for t in range(H):
                obs_t = tm[DataKeys.OBS][t]
                noise_t = self.injected_noise[t] if self.injected_noise is not None else None
                step_id = self.noise.next_step()
                if fused:
                    with (record.at(t) if record is not None else contextlib.nullcontext()), pair:
                        features, values = self._forward(obs_t, deterministic=deterministic)
                    self._fused_step(features, values, noise_t, t, gamma, step_id, deterministic)
                else:
                    self._generic_step(obs_t, noise_t, t, gamma, step_id, deterministic)

            # Bootstrap value at the last observation (:396-408).
            if self._identity_views():
                _, values = self._forward(tm[DataKeys.OBS][H], deterministic=deterministic)
            else:
                values = self.policy.sample(
                    self.buffer, kind="last", deterministic=deterministic, inplace=False, requires_grad=False,
                    return_actions=False, return_logp=False, return_values=True, return_views=False,
                )[DataKeys.VALUES]
            tm[DataKeys.VALUES][H].copy_(values)

            # One pass, one host sync (:411-436 takes 8 reductions and 9 syncs).
            raw = hip.rollout_stats(
                self.buffer[DataKeys.REWARDS],
                self.buffer[DataKeys.REVERSED_DISCOUNTED_RETURNS] if rdr is not None else None,
            )
            collect_stats, reward_scale = _collect_stats_from_raw(self.shards.combine_rollout_stats(raw))
            self.state.horizons += 1
            self.state.buffered = True
            self.state.reward_scale = reward_scale if hp.normalize_rewards else 1.0
            self.injected_noise = None
            if record is not None:  # (a Python-level write to the observations before step() voids the record)
                self._record_obs_version = tm[DataKeys.OBS]._version

        collect_stats["env/resets"] = hp.num_envs * int(env_was_reset)
        collect_stats["env/steps"] = hp.num_envs * hp.horizon
        collect_stats["profiling/collect_ms"] = collect_timer()
        return collect_stats

    def _rollout_record(self) -> None | fused_mlp.RolloutRecord:
        """The record this ``collect()`` fills (made once, its slabs reused by every rollout), or ``None``."""
        if not self.reuse_rollout_forward or not has_fused_loss(self.policy.distribution_cls):
            return None
        hp = self.hparams
        rec = self._record
        if rec is None or (rec.steps, rec.rows_per_step) != (hp.horizon, self.local_num_envs):
            rec = self._record = fused_mlp.RolloutRecord(hp.horizon, self.local_num_envs, keep_general=False)
        # heads that need h2 (1 KiB per row) are recorded only when every row is read back: one optimizer step per
        # SGD iteration.  Gate-bit towers (32 B per row) always.
        rec.keep_general = hp.num_minibatches == 1 or hp.accumulate_grads
        rec.begin()
        return rec

    def _fused_step(
        self, features: TensorDict, values: torch.Tensor, noise: None | torch.Tensor, t: int,
        gamma: float, step_id: int, deterministic: bool,
    ) -> None:
        tm = self._tm
        rdr = tm.get(DataKeys.REVERSED_DISCOUNTED_RETURNS)
        dist_cls = self.policy.distribution_cls
        if issubclass(dist_cls, Categorical):
            f1, f2 = features["logits"].contiguous(), None
        else:
            f1, f2 = features["mean"].contiguous(), features["log_std"].contiguous()
        self.env.fused_rollout_step(
            squashed=issubclass(dist_cls, SquashedNormal),
            features=f1,
            features2=f2,
            value=values.contiguous(),
            noise=noise.contiguous() if noise is not None else None,
            action_col=tm[DataKeys.ACTIONS][t],
            logp_col=tm[DataKeys.LOGP][t],
            value_col=tm[DataKeys.VALUES][t],
            reward_col=tm[DataKeys.REWARDS][t],
            obs_col_next=tm[DataKeys.OBS][t + 1],
            rdr_t=rdr[t] if rdr is not None else None,
            rdr_t1=rdr[t + 1] if rdr is not None else None,
            gamma=gamma,
            seed=self.noise.seed,
            step=step_id,
            env_offset=self.env.env_offset,
            deterministic=deterministic,
        )

    def _generic_step(
        self, obs_t: torch.Tensor, noise: None | torch.Tensor, t: int, gamma: float, step_id: int,
        deterministic: bool,
    ) -> None:
        """Any ``Env`` / ``Distribution``: policy.sample -> env.step -> one
        bookkeeping launch."""
        tm = self._tm
        rdr = tm.get(DataKeys.REVERSED_DISCOUNTED_RETURNS)
        self.policy.injected_noise = noise
        self.noise.step = step_id  # policy.sample advances it again by one
        sample = self.policy.sample(
            self.buffer[:, : (t + 1), ...],
            kind="last",
            deterministic=deterministic,
            inplace=False,
            requires_grad=False,
            return_actions=True,
            return_logp=True,
            return_values=True,
            return_views=False,
        )
        out = self.env.step(sample[DataKeys.ACTIONS])
        hip.rollout_scatter(
            sample[DataKeys.ACTIONS].contiguous(),
            sample[DataKeys.LOGP].contiguous(),
            sample[DataKeys.VALUES].contiguous(),
            out[DataKeys.REWARDS].contiguous(),
            out[DataKeys.OBS].contiguous(),
            tm[DataKeys.ACTIONS][t],
            tm[DataKeys.LOGP][t],
            tm[DataKeys.VALUES][t],
            tm[DataKeys.REWARDS][t],
            tm[DataKeys.OBS][t + 1],
            rdr[t] if rdr is not None else None,
            rdr[t + 1] if rdr is not None else None,
            gamma,
        )

    # ------------------------------------------------------------------ #
    # step()
    # ------------------------------------------------------------------ #
    def step(self) -> StepStats:
        """Update the policy from the collected buffer; returns losses,
        coefficients and the approximate KL."""
        if not self.state.buffered:
            raise RuntimeError(
                f"{self.__class__.__name__} is not buffered. "
                "Call `collect` once prior to `step`."
            )
        try:
            return self._step()
        except BaseException:
            # whatever the SGD passes shared between them (the recurrent algorithm: the fp16 planes of the sequences'
            # initial hidden states, a module-level switch in nn.fused_lstm) must not outlive a step() that failed
            self._release_step_caches()
            raise

    def _step(self) -> StepStats:
        hp = self.hparams
        H, tm = hp.horizon, self._tm
        world = self.shards.world_size
        with profile_ms() as step_timer:
            gae_launch(
                self.buffer[DataKeys.REWARDS],
                self.buffer[DataKeys.VALUES],
                self.buffer[DataKeys.ADVANTAGES],
                self.buffer[DataKeys.RETURNS],
                gae_lambda=hp.gae_lambda,
                gamma=hp.gamma,
                reward_scale=self.state.reward_scale,
                normalize_advantages=hp.normalize_advantages,
                write_scaled_rewards=False,
                moment_reduce=self.shards.sum_ if self.shards.active else None,
            )

            num_minibatches = hp.num_minibatches
            gas = num_minibatches if hp.accumulate_grads else 1
            grad_scale = 1.0 / (self._samples_per_minibatch() * gas)

            stat_tracker = StatTracker(
                ["coefficients/entropy", "coefficients/vf", "losses/entropy", "losses/policy",
                 "losses/vf", "losses/total", "monitors/kl_div"],
                sum_keys=["losses/entropy", "losses/policy", "losses/vf", "losses/total",
                          "monitors/kl_div"],
            )
            entropy_coeff = self.entropy_scheduler.coeff
            sync_each = hp.target_kl_div is not None
            pending: list[tuple[torch.Tensor, bool]] = []

            def record(loss_values: dict[str, float], step_this_batch: bool) -> None:
                stat_tracker.update(
                    {
                        "coefficients/entropy": entropy_coeff,
                        "coefficients/vf": hp.vf_coeff,
                        "losses/entropy": loss_values["entropy"],
                        "losses/policy": loss_values["policy"],
                        "losses/vf": loss_values["vf"],
                        "losses/total": loss_values["total"],
                        "monitors/kl_div": loss_values["kl"] / gas,
                    },
                    reduce=step_this_batch,
                )

            stop_early = False
            window: list[torch.Tensor] = []   # loss sums of the open accumulation window
            for sgd_iter in range(hp.num_sgd_iters):
                for i, batch in enumerate(self._iter_minibatches(sgd_iter)):
                    step_this_batch = (i + 1) % gas == 0
                    sums = self._minibatch_forward_backward(batch, entropy_coeff, grad_scale)
                    window.append(sums)
                    if not sync_each:
                        pending.append((sums, step_this_batch))
                    if not step_this_batch:
                        continue
                    # shards: gradient + this window's loss sums in one all-reduce
                    self.shards.sum_gradients_(self.policy.model.parameters(), window)
                    window = []
                    if sync_each:
                        loss_values = losses_from_sums(
                            *sums.tolist(), entropy_coeff=entropy_coeff, vf_coeff=hp.vf_coeff,
                            grad_accumulation_steps=gas,
                        )
                        record(loss_values, step_this_batch)
                        if loss_values["kl"] > 1.5 * hp.target_kl_div:
                            # The reference breaks before backward() (:577-582); the
                            # fused loss has already run it, so its gradients are
                            # dropped. `target_kl_div` excludes `accumulate_grads`
                            # (AlgorithmHparams, reference data.py:227-231), hence
                            # gas == 1 here and `.grad` held nothing else: both end
                            # with every `.grad` None (tests/golden/early_stop.npz).
                            assert gas == 1
                            self.optimizer.zero_grad()
                            stop_early = True
                            break
                    self.grad_scaler.unscale_(self.optimizer)
                    nn.utils.clip_grad_norm_(self.policy.model.parameters(), hp.max_grad_norm)
                    self.grad_scaler.step(self.optimizer)
                    self.grad_scaler.update()
                    self.optimizer.zero_grad()
                if stop_early:
                    break

            if pending:
                # Single device->host copy for the whole step().
                stacked = torch.stack([s for s, _ in pending]).tolist()
                for row, (_, step_this_batch) in zip(stacked, pending):
                    record(
                        losses_from_sums(*row, entropy_coeff=entropy_coeff, vf_coeff=hp.vf_coeff,
                                         grad_accumulation_steps=gas),
                        step_this_batch,
                    )

            self.lr_scheduler.step(hp.num_envs * self.state.horizons)
            self.entropy_scheduler.step(hp.num_envs * self.state.horizons)
            self._reset_buffer()
            self.state.buffered = False
            self.injected_permutations = None
            self._flat_full = None
            self._views_all = None
            self._packed = None
            self._release_step_caches()
            step_stats = stat_tracker.items()
        step_stats["profiling/step_ms"] = step_timer()
        return step_stats  # type: ignore[return-value]

    #: Leaves a training minibatch is made of.
    TRAIN_KEYS = (DataKeys.OBS, DataKeys.ACTIONS, DataKeys.LOGP, DataKeys.ADVANTAGES, DataKeys.RETURNS)

    def _samples_per_minibatch(self) -> int:
        """Global number of samples the loss of one minibatch averages over."""
        return self.hparams.sgd_minibatch_size

    def _permutation(self, sgd_iter: int, size: int) -> torch.Tensor:
        """Order in which this SGD iteration visits the local units (samples, or
        sequences for the recurrent algorithm): a fresh permutation per iteration
        as the reference's ``Batcher`` draws (``_utils.py:211-218``), an injected
        one (parity runs), or ``arange`` without shuffling."""
        device = self._tm[DataKeys.LOGP].device
        if not self.hparams.shuffle_minibatches:
            return torch.arange(size, device=device)
        if self.injected_permutations is not None:
            return self.injected_permutations[sgd_iter].to(device)
        return torch.randperm(size, device=device)

    def _iter_minibatches(self, sgd_iter: int):
        """Yields the minibatches of one SGD iteration as dicts of dense tensors."""
        hp = self.hparams
        H, tm = hp.horizon, self._tm
        local_samples = self.local_num_envs * H
        self._batch_rows = None
        if not self._identity_views():
            yield from self._iter_view_minibatches(sgd_iter)
            return
        if hp.num_minibatches == 1:
            # Time-major storage: dropping the last column and flattening is the
            # contiguous prefix -- no copy, no shuffle (a mean over the whole
            # buffer does not depend on order).
            if getattr(self, "_flat_full", None) is None:
                self._flat_full = {
                    k: tm[k][:H].reshape(local_samples, *tm[k].shape[2:]) for k in self.TRAIN_KEYS
                }
            self._batch_rows = ("all", None)
            yield self._flat_full
            return
        local_mb = hp.sgd_minibatch_size // self.shards.world_size
        perm = self._permutation(sgd_iter, local_samples)
        if getattr(self, "_packed", None) is None:
            # the buffer is shuffled num_sgd_iters x num_minibatches times: lay each
            # sample's fields side by side once, then every gather reads one row
            self._packed = hip.PackedSamples(H, [self.buffer[k] for k in self.TRAIN_KEYS])
        for index in torch.split(perm, local_mb):
            index = index.contiguous()
            self._batch_rows = ("index", index)
            yield dict(zip(self.TRAIN_KEYS, self._packed.gather(index)))

    def _iter_view_minibatches(self, sgd_iter: int):
        """Minibatches for models with rolling-window view requirements
        (``src/rl8/algorithms/_feedforward.py:471-482``): the windows of the whole
        buffer are built once per ``step()`` (env-major sample order, like the
        flattened buffer) and indexed per minibatch."""
        hp = self.hparams
        H = hp.horizon
        local_samples = self.local_num_envs * H
        if getattr(self, "_views_all", None) is None:
            views = self.policy.model.apply_view_requirements(self.buffer[:, :-1, ...], kind="all")
            if views.batch_size[0] != local_samples:
                raise ValueError(
                    f"The model's view requirements keep {views.batch_size[0]} of {local_samples} samples;"
                    " the buffer needs one window per sample (use `padded_rolling_window`)."
                )
            self._views_all = views
        keys = [k for k in self.TRAIN_KEYS if k != DataKeys.OBS]
        local_mb = hp.sgd_minibatch_size // self.shards.world_size
        if hp.num_minibatches == 1:
            perm = torch.arange(local_samples, device=self._tm[DataKeys.LOGP].device)
        else:
            perm = self._permutation(sgd_iter, local_samples)
        for index in torch.split(perm, local_mb):
            gathered = hip.gather_minibatch(index.contiguous(), H, [self.buffer[k] for k in keys])
            batch: dict[str, Any] = dict(zip(keys, gathered))
            batch[DataKeys.VIEWS] = self._views_all[index]
            yield batch

    def _minibatch_forward_backward(
        self, batch: dict[str, torch.Tensor], entropy_coeff: float, grad_scale: float
    ) -> torch.Tensor:
        """Network forward, fused loss forward+backward, network backward for
        one minibatch, streamed in row chunks that accumulate into ``.grad``.
        Returns the minibatch's raw loss sums (device, fp64)."""
        hp = self.hparams
        rows = batch[DataKeys.LOGP].shape[0]
        dist_cls = self.policy.distribution_cls
        fused = has_fused_loss(dist_cls)
        total_sums: None | torch.Tensor = None
        scale = None
        if hp.enable_amp:
            scale = self.grad_scaler.scale(torch.ones((), device=batch[DataKeys.LOGP].device))
        # Rows the rollout already took through the towers with the weights they still have (iteration 0, before the
        # first optimizer step): start from the record instead of launching the forward again.
        recorded: None | dict[int, tuple] = None
        rec, batch_rows = getattr(self, "_record", None), getattr(self, "_batch_rows", None)
        if (rec is not None and batch_rows is not None and DataKeys.OBS in batch
                and self._record_obs_version == self._tm[DataKeys.OBS]._version and rec.valid()):
            obs_tm = self._tm[DataKeys.OBS]
            rec.require_inputs(obs_tm.data_ptr(), obs_tm.stride(0) * obs_tm.element_size())
        if (rec is not None and batch_rows is not None and DataKeys.OBS in batch
                and self._record_obs_version == self._tm[DataKeys.OBS]._version and rec.valid()):
            recorded = rec.rows(0, rows) if batch_rows[0] == "all" else rec.gather(batch_rows[1])
        for start in range(0, rows, self.max_rows_per_pass):
            stop = min(rows, start + self.max_rows_per_pass)
            chunk = {k: v[start:stop] for k, v in batch.items()}
            n = stop - start
            replayed = contextlib.nullcontext() if not recorded else fused_mlp.replay(
                {k: (key, *[None if t is None else t[start:stop] for t in rest]) for k, (key, *rest) in recorded.items()},
                chunk[DataKeys.OBS])
            # A two-way Categorical under the fused loss: rl8_ppo_loss_categorical_fwd_bwd_f32 emits logit gradients
            # that are exact negatives of each other, so the policy tower may keep only the gate bits of h2 from the
            # first SGD iteration on (fused_mlp.expect_pair_gradients; still checked on the device in the backward).
            pair = fused_mlp.expect_pair_gradients() if fused and issubclass(dist_cls, Categorical) else contextlib.nullcontext()
            with amp.autocast("cuda", enabled=hp.enable_amp), pair, replayed:
                views = chunk[DataKeys.VIEWS] if DataKeys.VIEWS in chunk else TensorDict(
                    {DataKeys.OBS: chunk[DataKeys.OBS]}, batch_size=n)
                sample = self.policy.sample(
                    TensorDict({DataKeys.VIEWS: views}, batch_size=n),
                    kind="all",
                    deterministic=False,
                    inplace=False,
                    requires_grad=True,
                    return_actions=False,
                    return_logp=False,
                    return_values=True,
                    return_views=False,
                )
            features, values = sample[DataKeys.FEATURES], sample[DataKeys.VALUES]
            if fused:
                sums, inputs, grads = fused_ppo_loss(
                    dist_cls, features, values, chunk[DataKeys.ACTIONS], chunk[DataKeys.LOGP],
                    chunk[DataKeys.ADVANTAGES], chunk[DataKeys.RETURNS],
                    clip_param=hp.clip_param, dual_clip_param=hp.dual_clip_param,
                    entropy_coeff=entropy_coeff, vf_clip_param=hp.vf_clip_param,
                    vf_coeff=hp.vf_coeff, grad_scale=grad_scale,
                )
                if scale is not None:
                    grads = [g * scale.to(g.dtype) for g in grads]
                torch.autograd.backward(inputs, grads)
            else:
                sums = self._composed_loss_backward(features, values, chunk, entropy_coeff,
                                                    grad_scale, scale)
            total_sums = sums if total_sums is None else total_sums + sums
        assert total_sums is not None
        return total_sums

    def _composed_loss_backward(
        self, features: TensorDict, values: torch.Tensor, chunk: dict[str, torch.Tensor],
        entropy_coeff: float, grad_scale: float, scale: None | torch.Tensor,
    ) -> torch.Tensor:
        """Custom distributions: loss composed from the distribution's own
        ``logp`` / ``entropy`` (tensor ops on the device), SUM-reduced so chunks
        and shards add up exactly like the fused kernel's sums."""
        import torch.nn.functional as F

        hp = self.hparams
        dist = self.policy.distribution_cls(features, self.policy.model)
        adv, ret = chunk[DataKeys.ADVANTAGES], chunk[DataKeys.RETURNS]
        log_ratio = dist.logp(chunk[DataKeys.ACTIONS]) - chunk[DataKeys.LOGP]
        ratio = torch.exp(log_ratio)
        vf = torch.clamp(F.smooth_l1_loss(values, ret, reduction="none"), 0.0, hp.vf_clip_param).sum()
        surr1 = adv * ratio
        surr2 = adv * torch.clamp(ratio, 1 - hp.clip_param, 1 + hp.clip_param)
        clip1 = torch.min(surr1, surr2)
        if hp.dual_clip_param:
            policy = torch.where(adv < 0, torch.max(clip1, hp.dual_clip_param * adv), clip1).sum()
        else:
            policy = clip1.sum()
        total = hp.vf_coeff * vf - policy
        entropy = torch.zeros((), device=values.device)
        if entropy_coeff != 0:
            entropy = dist.entropy().sum()
            total = total - entropy_coeff * entropy
        loss = total * grad_scale
        if scale is not None:
            loss = loss * scale
        loss.backward()
        with torch.no_grad():
            kl = ((ratio - 1) - log_ratio).sum()
            count = torch.tensor(float(values.shape[0]), device=values.device)
            return torch.stack([entropy, policy, vf, count, kl]).double()

    # ------------------------------------------------------------------ #
    # validate()
    # ------------------------------------------------------------------ #
    def validate(self) -> None:
        """Shape checks on one reset / sample / step round trip, with the
        reference's assertion messages (:617-697)."""
        n = self.local_num_envs
        obs = self.env.reset()
        self.env.observation_spec.assert_is_in(obs)
        try:
            self.buffer[DataKeys.OBS][:, 0, ...] = obs
        except RuntimeError as e:
            raise AssertionError(
                f"The observation from {self.env.reset.__qualname__} doesn't match the"
                " observation spec shape."
            ) from e
        sample_batch = self.policy.sample(
            self.buffer[:, :1, ...],
            kind="last",
            deterministic=False,
            inplace=False,
            requires_grad=False,
            return_actions=True,
            return_logp=True,
            return_values=True,
        )
        actions = sample_batch[DataKeys.ACTIONS]
        assert actions.ndim >= 2, (
            "Actions must be at least 2D and have shape ``[N, ...]`` (where ``N`` is"
            " the number of independent elements or environment instances, and ``...``"
            " is any number of additional dimensions)."
        )
        self.env.action_spec.assert_is_in(actions)
        try:
            self.buffer[DataKeys.ACTIONS][:, 0, ...] = actions
        except RuntimeError as e:
            raise AssertionError(
                "The action sampled from the policy doesn't match the action spec."
            ) from e
        assert sample_batch[DataKeys.LOGP].shape == torch.Size([n, 1]), (
            "Action log probabilities must be 2D and have shape ``[N, 1]`` (where ``N``"
            " is the number of independent elements or environment instances)."
        )
        assert sample_batch[DataKeys.VALUES].shape == torch.Size([n, 1]), (
            "Expected value estimates must be 2D and have shape ``[N, 1]`` (where ``N``"
            " is the number of independent elements or environment instances)."
        )
        out_batch = self.env.step(actions)
        obs = out_batch[DataKeys.OBS]
        self.env.observation_spec.assert_is_in(obs)
        try:
            self.buffer[DataKeys.OBS][:, 1, ...] = obs
        except RuntimeError as e:
            raise AssertionError(
                f"The observation from {self.env.step.__qualname__} doesn't match the"
                " observation spec shape."
            ) from e
        assert out_batch[DataKeys.REWARDS].shape == torch.Size([n, 1]), (
            "Rewards must be 2D and have shape ``[N, 1]`` (where ``N`` is the number of"
            " independent elements or environment instances)."
        )


def _collect_stats_from_raw(raw: list[float]) -> tuple[CollectStats, float]:
    """Means and unbiased stds from the 12 raw moments of
    ``rl8_rollout_stats_f32``."""
    n, s1, s2, mn, mx, nh, r1, r2, rmn, rmx, d1, d2 = raw

    def std(count: float, total: float, total_sq: float) -> float:
        if count < 2:
            return float("nan")
        return max((total_sq - total * total / count) / (count - 1.0), 0.0) ** 0.5

    stats: CollectStats = {
        "returns/min": mn,
        "returns/max": mx,
        "returns/mean": s1 / n,
        "returns/std": std(n, s1, s2),
        "rewards/min": rmn,
        "rewards/max": rmx,
        "rewards/mean": r1 / nh,
        "rewards/std": std(nh, r1, r2),
    }
    return stats, std(nh, d1, d2)
