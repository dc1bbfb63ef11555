"""GAE and the PPO loss, backed by the gfx950 kernels.

Signatures and semantics follow the reference's ``src/rl8/nn/functional.py``:
``generalized_advantage_estimate`` :50-123 and ``ppo_losses`` :259-363. Both take
and return tensordicts exactly as the reference does; what runs underneath is

* ``rl8_gae_scan_f32`` + ``rl8_advantage_normalise_f32`` -- one launch for the
  whole reverse scan (the reference issues 5H+6 strided ops), for both the
  reference's env-major ``[B, T+1, 1]`` layout (LDS-staged tiles) and this
  build's time-major rollout buffer (register scan);
* ``rl8_ppo_loss_*_fwd_bwd_f32`` -- loss terms, approximate KL and the gradients
  w.r.t. the distribution features and the value estimates in one launch, hooked
  into autograd so ``losses["total"].backward()`` keeps working.

Tensors must live on a HIP device: there is no CPU fallback.

"""

from __future__ import annotations

from typing import Any, Callable

import numpy as np
import torch

from .. import hip
from ..data import DataKeys
from ..distributions import Categorical, Distribution, Normal, SquashedNormal
from ..tensordict import TensorDict

MomentReduce = Callable[[torch.Tensor], torch.Tensor]


def _f32(x: float) -> float:
    """The value ``x`` takes when torch casts a Python scalar to float32."""
    return float(np.float32(x))


def _require_hip(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise hip.HipExtensionError(
            f"{what} is on {t.device}: rl8_amd runs this op on a HIP device only"
            " (no CPU fallback)."
        )


def gae_launch(
    rewards: torch.Tensor,
    values: torch.Tensor,
    advantages: torch.Tensor,
    returns: torch.Tensor,
    *,
    gae_lambda: float,
    gamma: float,
    reward_scale: float,
    normalize_advantages: bool,
    write_scaled_rewards: bool,
    moment_reduce: None | MomentReduce = None,
) -> torch.Tensor:
    """Scan + optional normalisation over ``[N, H+1, 1]`` leaves that share one
    layout (env-major or time-major). Returns the device moments
    ``(count, sum, sum_sq)`` of ``advantages[:, :H]`` (after ``moment_reduce``,
    which an env-sharded run uses to all-reduce them)."""
    layout, es, ts = hip.buffer_layout(rewards)
    if layout < 0:
        raise ValueError("GAE leaves must be env-major or time-major dense tensors")
    for name, t in (("values", values), ("advantages", advantages), ("returns", returns)):
        if t.shape != rewards.shape or t.stride() != rewards.stride() or t.dtype != torch.float32:
            raise ValueError(f"{name} must share shape, strides and dtype (float32) with rewards")
    n, h = rewards.shape[0], rewards.shape[1] - 1
    if h < 1:
        raise ValueError("GAE needs at least two columns ([B, T + 1, ...])")
    moments = hip.gae_scan(
        rewards,
        values,
        advantages,
        returns,
        layout=layout,
        n=n,
        h=h,
        gamma=_f32(gamma),
        gamma_lambda=_f32(gamma * gae_lambda),
        reward_denominator=_f32(reward_scale + 1e-8),
        write_scaled_rewards=write_scaled_rewards,
    )
    if normalize_advantages:
        if moment_reduce is not None:
            moments = moment_reduce(moments)
        hip.advantage_normalise(advantages, layout=layout, n=n, h=h, moments=moments)
    return moments


def generalized_advantage_estimate(
    batch: TensorDict,
    /,
    *,
    gae_lambda: float = 0.95,
    gamma: float = 0.95,
    inplace: bool = False,
    normalize_advantages: bool = True,
    return_returns: bool = True,
    reward_scale: float = 1.0,
) -> TensorDict:
    """Generalized Advantage Estimate (and returns) from rewards and values.

    Args:
        batch: Tensordict of batch size ``[B, T + 1, ...]`` with ``"rewards"``
            and ``"values"``. As in the reference, the rewards it holds are
            divided by ``reward_scale + 1e-8``.
        gae_lambda: GAE bias/variance trade-off.
        gamma: Discount factor.
        inplace: Store outputs in ``batch`` instead of a new tensordict.
        normalize_advantages: Standardise ``advantages[:, :-1]`` with their
            batch mean and (unbiased) standard deviation.
        return_returns: Also store ``"returns" = advantages + values``.
        reward_scale: Reward normaliser.

    Returns:
        Tensordict with ``"advantages"`` and, optionally, ``"returns"``.

    """
    rewards = batch[DataKeys.REWARDS]
    values = batch[DataKeys.VALUES]
    _require_hip(rewards, "rewards")
    _require_hip(values, "values")
    out = batch if inplace else TensorDict({}, batch_size=batch.batch_size, device=batch.device)

    work_rewards = rewards
    if rewards.dtype != torch.float32 or hip.buffer_layout(rewards)[0] < 0:
        work_rewards = rewards.to(torch.float32).contiguous()
    work_values = values
    if (
        values.dtype != torch.float32
        or values.stride() != work_rewards.stride()
        or values.shape != work_rewards.shape
    ):
        work_values = torch.empty_like(work_rewards)
        work_values.copy_(values)

    advantages = out[DataKeys.ADVANTAGES] if DataKeys.ADVANTAGES in out.keys() else None
    if (
        advantages is None
        or advantages.stride() != work_rewards.stride()
        or advantages.dtype != torch.float32
        or advantages.shape != work_rewards.shape
    ):
        advantages = torch.empty_like(work_rewards)
    returns = torch.empty_like(work_rewards)

    gae_launch(
        work_rewards,
        work_values,
        advantages,
        returns,
        gae_lambda=gae_lambda,
        gamma=gamma,
        reward_scale=reward_scale,
        normalize_advantages=normalize_advantages,
        write_scaled_rewards=True,
    )
    batch[DataKeys.REWARDS] = work_rewards
    out[DataKeys.ADVANTAGES] = advantages
    if return_returns:
        out[DataKeys.RETURNS] = returns
    return out


# --------------------------------------------------------------------------- #
# PPO loss.
# --------------------------------------------------------------------------- #
class PPOLossSums:
    """Raw fp64 sums from one fused loss launch: entropy, policy, vf, count, kl.
    Env shards add these with one all-reduce before anything is divided."""

    KEYS = ("entropy", "policy", "vf", "count", "kl")

    def __init__(self, sums: torch.Tensor) -> None:
        self.sums = sums

    def losses(
        self, *, entropy_coeff: float, vf_coeff: float, grad_accumulation_steps: int = 1
    ) -> dict[str, float]:
        """One device->host copy; returns entropy / policy / vf / total (each
        divided by ``grad_accumulation_steps``) and the unscaled kl."""
        ent_s, pol_s, vf_s, cnt, kl_s = self.sums.tolist()
        return losses_from_sums(
            ent_s, pol_s, vf_s, cnt, kl_s,
            entropy_coeff=entropy_coeff, vf_coeff=vf_coeff,
            grad_accumulation_steps=grad_accumulation_steps,
        )


def losses_from_sums(
    ent_s: float, pol_s: float, vf_s: float, cnt: float, kl_s: float, *,
    entropy_coeff: float, vf_coeff: float, grad_accumulation_steps: int = 1,
) -> dict[str, float]:
    entropy = ent_s / cnt if entropy_coeff != 0 else 0.0
    policy, vf = pol_s / cnt, vf_s / cnt
    total = vf_coeff * vf - policy
    if entropy_coeff != 0:
        total -= entropy_coeff * entropy
    g = grad_accumulation_steps
    return {
        "entropy": entropy / g,
        "policy": policy / g,
        "vf": vf / g,
        "total": total / g,
        "kl": kl_s / cnt,
    }


def fused_ppo_loss(
    distribution_cls: type[Distribution],
    features: TensorDict,
    values: torch.Tensor,
    actions: torch.Tensor,
    logp_old: torch.Tensor,
    advantages: torch.Tensor,
    returns: torch.Tensor,
    *,
    clip_param: float,
    dual_clip_param: None | float,
    entropy_coeff: float,
    vf_clip_param: float,
    vf_coeff: float,
    grad_scale: float,
    with_grad: bool = True,
) -> tuple[torch.Tensor, list[torch.Tensor], list[torch.Tensor]]:
    """One fused launch. Returns ``(sums, inputs, grads)`` where ``inputs`` are
    the differentiable tensors (features then values) and ``grads`` the matching
    ``d total / d input * grad_scale``; feed both to ``torch.autograd.backward``.

    ``grad_scale`` is ``1 / (global minibatch size * grad_accumulation_steps)``.

    """
    hp = hip.ppo_hparams(
        clip_param=clip_param, dual_clip_param=dual_clip_param, entropy_coeff=entropy_coeff,
        vf_clip_param=vf_clip_param, vf_coeff=vf_coeff, grad_scale=grad_scale,
    )
    flat = lambda t: t.detach().contiguous()  # noqa: E731
    if issubclass(distribution_cls, Categorical):
        logits = features["logits"]
        sums, g_logits, g_values = hip.ppo_loss_categorical(
            flat(logits).float(), flat(values).float(), flat(actions), flat(logp_old),
            flat(advantages), flat(returns), hp, with_grad=with_grad,
        )
        if not with_grad:
            return sums, [], []
        g_logits = g_logits.to(logits.dtype)
        if logits.shape[-1] == 2 and logits.numel() == 2 * values.numel() and g_logits.dtype == torch.float32:
            from . import fused_mlp

            fused_mlp.trust_pair_gradient(g_logits)  # (one action dimension, two classes: the kernel stored g and -g)
        return sums, [logits, values], [g_logits, g_values.to(values.dtype)]
    if issubclass(distribution_cls, Normal):
        if issubclass(distribution_cls, SquashedNormal) and entropy_coeff != 0:
            raise NotImplementedError(
                f"Entropy isn't defined for {distribution_cls.__name__}. Set the"
                " entropy coefficient to `0` to avoid this error during training."
            )
        mean, log_std = features["mean"], features["log_std"]
        sums, g_mean, g_ls, g_values = hip.ppo_loss_normal(
            flat(mean).float(), flat(log_std).float(), flat(values).float(), flat(actions),
            flat(logp_old), flat(advantages), flat(returns), hp,
            squashed=issubclass(distribution_cls, SquashedNormal), with_grad=with_grad,
        )
        if not with_grad:
            return sums, [], []
        return (
            sums,
            [mean, log_std, values],
            [g_mean.to(mean.dtype), g_ls.to(log_std.dtype), g_values.to(values.dtype)],
        )
    raise TypeError(f"{distribution_cls.__name__} has no fused loss kernel")


def has_fused_loss(distribution_cls: type[Distribution]) -> bool:
    """True when ``distribution_cls`` computes ``logp`` / ``entropy`` exactly as
    one of the built-in distributions (so the fused kernel restates it)."""
    for base in (Categorical, SquashedNormal, Normal):
        if issubclass(distribution_cls, base):
            return (
                distribution_cls.logp is base.logp and distribution_cls.entropy is base.entropy
            )
    return False


class _AttachPrecomputedGrads(torch.autograd.Function):
    """Scalar whose gradient w.r.t. ``inputs`` was already computed by the fused
    kernel; backward scales those gradients by the incoming scalar."""

    @staticmethod
    def forward(ctx: Any, total: torch.Tensor, n_inputs: int, *tensors: torch.Tensor) -> torch.Tensor:
        ctx.grads = tensors[n_inputs:]
        return total.clone()

    @staticmethod
    def backward(ctx: Any, grad_total: torch.Tensor) -> tuple[Any, ...]:
        scaled = tuple(g * grad_total.to(g.dtype) for g in ctx.grads)
        return (None, None, *scaled, *([None] * len(scaled)))


def ppo_losses(
    buffer_batch: TensorDict,
    sample_batch: TensorDict,
    sample_distribution: Distribution,
    /,
    *,
    clip_param: float = 0.2,
    dual_clip_param: None | float = 5.0,
    entropy_coeff: float = 0.0,
    vf_clip_param: float = 1.0,
    vf_coeff: float = 1.0,
) -> TensorDict:
    """Proximal Policy Optimization loss (dual-clipped surrogate, clamped Huber
    value loss, optional entropy bonus), mean-reduced.

    Args:
        buffer_batch: ``[B, ...]`` tensordict with ``"actions"``,
            ``"advantages"``, ``"logp"``, ``"returns"``.
        sample_batch: ``[B, ...]`` tensordict with ``"values"`` from the current
            policy.
        sample_distribution: Distribution built from the current policy's
            features.
        clip_param, dual_clip_param, entropy_coeff, vf_clip_param, vf_coeff:
            As in the reference (``dual_clip_param=None`` disables dual clip).

    Returns:
        Tensordict with scalar ``"entropy"``, ``"policy"``, ``"vf"``, ``"total"``;
        ``total.backward()`` propagates into the features and the values.

    """
    values = sample_batch[DataKeys.VALUES]
    _require_hip(values, "values")
    dist_cls = type(sample_distribution)
    device = values.device
    if has_fused_loss(dist_cls):
        m = values.shape[0]
        need_grad = torch.is_grad_enabled() and (
            values.requires_grad
            or any(f.requires_grad for f in sample_distribution.features.values())
        )
        sums, inputs, grads = fused_ppo_loss(
            dist_cls,
            sample_distribution.features,
            values,
            buffer_batch[DataKeys.ACTIONS],
            buffer_batch[DataKeys.LOGP],
            buffer_batch[DataKeys.ADVANTAGES],
            buffer_batch[DataKeys.RETURNS],
            clip_param=clip_param,
            dual_clip_param=dual_clip_param,
            entropy_coeff=entropy_coeff,
            vf_clip_param=vf_clip_param,
            vf_coeff=vf_coeff,
            grad_scale=1.0 / m,
            with_grad=need_grad,
        )
        means = sums / sums[3]
        policy_loss = means[1].float()
        vf_loss = means[2].float()
        total = vf_coeff * means[2] - means[1]
        if entropy_coeff != 0:
            entropy_loss = means[0].float()
            total = total - entropy_coeff * means[0]
        else:
            entropy_loss = torch.tensor([0.0])
        total = total.float()
        if need_grad:
            total = _AttachPrecomputedGrads.apply(total, len(inputs), *inputs, *grads)
        return TensorDict(
            {"entropy": entropy_loss, "policy": policy_loss, "vf": vf_loss, "total": total},
            batch_size=[],
        )

    # Custom distribution: compose the loss from its own logp / entropy with
    # tensor ops on the device (same operation order as the reference).
    import torch.nn.functional as F

    advantages = buffer_batch[DataKeys.ADVANTAGES]
    p_ratio = torch.exp(
        sample_distribution.logp(buffer_batch[DataKeys.ACTIONS]) - buffer_batch[DataKeys.LOGP]
    )
    vf_loss = torch.mean(
        torch.clamp(
            F.smooth_l1_loss(values, buffer_batch[DataKeys.RETURNS], reduction="none"),
            0.0,
            vf_clip_param,
        )
    )
    surr1 = advantages * p_ratio
    surr2 = advantages * torch.clamp(p_ratio, 1 - clip_param, 1 + clip_param)
    clip1 = torch.min(surr1, surr2)
    if dual_clip_param:
        clip2 = torch.max(clip1, dual_clip_param * advantages)
        policy_loss = torch.where(advantages < 0, clip2, clip1).mean()
    else:
        policy_loss = clip1.mean()
    total_loss = vf_coeff * vf_loss - policy_loss
    if entropy_coeff != 0:
        entropy_loss = sample_distribution.entropy().mean()
        total_loss = total_loss - entropy_coeff * entropy_loss
    else:
        entropy_loss = torch.tensor([0.0])
    del device
    return TensorDict(
        {"entropy": entropy_loss, "policy": policy_loss, "vf": vf_loss, "total": total_loss},
        batch_size=[],
    )
