"""CPU restatement of ``Algorithm.collect()`` / ``Algorithm.step()``.
TEST INFRASTRUCTURE (see ``rl8_oracle.c``): the checker for the GPU path's
end-to-end behaviour and the ``cpu_baseline`` leg of ``bench.py``. Never imported
by ``rl8_amd``.

Follows the reference's ``src/rl8/algorithms/_feedforward.py`` -- ``collect``
:301-441, ``step`` :443-615 -- with the same op order: env-major ``[N, H+1, d]``
buffer, per-timestep forward -> sample -> env.step -> RDR -> column writes,
bootstrap value, stats; then GAE, flatten to ``n*H + t``, per-iteration
permutation, minibatch gather, loss, clip, Adam. The policy / value MLPs run on
torch CPU (as in the reference); everything else calls the C restatement.

Pinned by tests/test_oracle_traces.py against the reference's recorded traces.

"""

from __future__ import annotations

import time
from typing import Any

import numpy as np
import torch
import torch.nn as nn

from . import oracle


def _tower(obs_dim: int, hidden: int = 256) -> list[nn.Module]:
    return [nn.Sequential(nn.Linear(obs_dim, hidden), nn.ReLU(), nn.Linear(hidden, hidden)), nn.ReLU()]


def _head(in_dim: int, out_dim: int) -> nn.Linear:
    head = nn.Linear(in_dim, out_dim)
    nn.init.uniform_(head.weight, a=-1e-3, b=1e-3)
    nn.init.zeros_(head.bias)
    return head


class DiscreteModel(nn.Module):
    """Default discrete model of the reference (models/_feedforward.py:313-383);
    same module tree, so its state_dict keys match."""

    def __init__(self, obs_dim: int, action_dims: int, classes: int) -> None:
        super().__init__()
        self.a, self.k = action_dims, classes
        self.feature_model = nn.Sequential(*_tower(obs_dim), _head(256, action_dims * classes))
        self.vf_model = nn.Sequential(*_tower(obs_dim), nn.Linear(256, 1))

    def forward(self, obs: torch.Tensor) -> tuple[dict[str, torch.Tensor], torch.Tensor]:
        return {"logits": self.feature_model(obs).reshape(-1, self.a, self.k)}, self.vf_model(obs)


class ContinuousModel(nn.Module):
    """Default continuous model of the reference (models/_feedforward.py:234-310)."""

    def __init__(self, obs_dim: int, action_dims: int) -> None:
        super().__init__()
        self.latent_model = nn.Sequential(*_tower(obs_dim))
        self.action_mean = _head(256, action_dims)
        self.action_log_std = _head(256, action_dims)
        self.vf_model = nn.Sequential(*_tower(obs_dim), nn.Linear(256, 1))

    def forward(self, obs: torch.Tensor) -> tuple[dict[str, torch.Tensor], torch.Tensor]:
        latents = self.latent_model(obs)
        feats = {"mean": self.action_mean(latents), "log_std": torch.tanh(self.action_log_std(latents))}
        return feats, self.vf_model(obs)


class OraclePPO:
    """Feed-forward PPO on the CPU for the dummy envs and CartPole."""

    def __init__(
        self,
        env: str = "discrete",  # "discrete" | "continuous" | "cartpole"
        *,
        num_envs: int = 8192,
        horizon: int = 32,
        distribution: None | str = None,  # "categorical" | "normal" | "squashed"
        gamma: float = 0.95,
        gae_lambda: float = 0.95,
        num_sgd_iters: int = 4,
        sgd_minibatch_size: None | int = None,
        accumulate_grads: bool = False,
        shuffle_minibatches: bool = True,
        clip_param: float = 0.2,
        vf_clip_param: float = 5.0,
        dual_clip_param: None | float = None,
        vf_coeff: float = 1.0,
        entropy_coeff: float = 0.0,
        max_grad_norm: float = 5.0,
        normalize_advantages: bool = True,
        normalize_rewards: bool = True,
        horizons_per_env_reset: int = 1,
        lr: float = 1e-3,
        seed: int = 0,
        cartpole_config: None | dict[str, Any] = None,
    ) -> None:
        self.env_kind = env
        self.n, self.h = num_envs, horizon
        self.distribution = distribution or ("normal" if env == "continuous" else "categorical")
        self.gamma, self.gae_lambda = gamma, gae_lambda
        self.num_sgd_iters = num_sgd_iters
        self.mb = sgd_minibatch_size or num_envs * horizon
        self.num_minibatches = (num_envs * horizon) // self.mb
        self.gas = self.num_minibatches if accumulate_grads else 1
        self.shuffle = shuffle_minibatches
        self.hp_kw = dict(clip_param=clip_param, dual_clip_param=dual_clip_param, entropy_coeff=entropy_coeff,
                          vf_clip_param=vf_clip_param, vf_coeff=vf_coeff)
        self.max_grad_norm = max_grad_norm
        self.normalize_advantages, self.normalize_rewards = normalize_advantages, normalize_rewards
        self.horizons_per_env_reset = horizons_per_env_reset
        self.seed = seed
        self.noise_step = 0
        self.reset_count = 0
        self.horizons = 0
        self.reward_scale = 1.0
        if env == "cartpole":
            self.obs_dim, self.k = 5, 3
            self.cp_cfg = oracle.cartpole_cfg(**(cartpole_config or {}))
            self.model: nn.Module = DiscreteModel(5, 1, 3)
        elif env == "discrete":
            self.obs_dim, self.k = 1, 2
            self.model = DiscreteModel(1, 1, 2)
        else:
            self.obs_dim = 1
            self.model = ContinuousModel(1, 1)
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr)
        adt = np.float32 if env == "continuous" else np.int64
        n, h1 = num_envs, horizon + 1
        self.buf = {
            "obs": np.zeros((n, h1, self.obs_dim), np.float32),
            "rewards": np.zeros((n, h1, 1), np.float32),
            "actions": np.zeros((n, h1, 1), adt),
            "logp": np.zeros((n, h1, 1), np.float32),
            "values": np.zeros((n, h1, 1), np.float32),
            "advantages": np.zeros((n, h1, 1), np.float32),
            "returns": np.zeros((n, h1, 1), np.float32),
            "reversed_discounted_returns": np.zeros((n, h1, 1), np.float32),
        }
        self.state: None | np.ndarray = None

    # -- env ---------------------------------------------------------------
    def _reset(self, reset_state: None | np.ndarray) -> np.ndarray:
        if self.env_kind == "cartpole":
            self.state = (oracle.cartpole_reset(self.n, 0.01, self.seed, self.reset_count)
                          if reset_state is None else np.array(reset_state, np.float32))
            x, xd, th, thd = self.state
            obs = np.stack([x, xd, np.cos(th), np.sin(th), thd], axis=1).astype(np.float32)
        else:
            self.state = (oracle.dummy_env_reset(self.n, 100.0, self.seed, self.reset_count)
                          if reset_state is None else np.array(reset_state, np.float32))
            obs = self.state
        self.reset_count += 1
        return obs

    def _env_step(self, actions: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        if self.env_kind == "cartpole":
            self.state, obs, reward = oracle.cartpole_step(self.state, actions, self.cp_cfg)
            return obs, reward
        self.state, reward = oracle.dummy_env_step(self.state, actions)
        return self.state, reward

    def _sample(self, feats: dict[str, torch.Tensor], noise: None | np.ndarray, step: int):
        if self.distribution == "categorical":
            return oracle.categorical_sample(feats["logits"].numpy(), noise, seed=self.seed, step=step)
        return oracle.normal_sample(feats["mean"].numpy(), feats["log_std"].numpy(), noise,
                                    squashed=self.distribution == "squashed", seed=self.seed, step=step)

    # -- collect -----------------------------------------------------------
    def collect(self, *, noise: None | np.ndarray = None, reset_state: None | np.ndarray = None) -> dict[str, float]:
        t0 = time.perf_counter()
        b, h = self.buf, self.h
        env_was_reset = False
        carry = (self.horizons and self.horizons_per_env_reset < 0) or (self.horizons % self.horizons_per_env_reset)
        if carry:
            b["obs"][:, 0] = b["obs"][:, -1]
            b["reversed_discounted_returns"][:, 0] = b["reversed_discounted_returns"][:, -1]
        else:
            b["obs"][:, 0] = self._reset(reset_state)
            env_was_reset = True
            b["reversed_discounted_returns"][:, 0] = 0.0
        with torch.no_grad():
            for t in range(h):
                feats, values = self.model(torch.from_numpy(np.ascontiguousarray(b["obs"][:, t])))
                actions, logp = self._sample(feats, noise[t] if noise is not None else None, self.noise_step)
                self.noise_step += 1
                obs, reward = self._env_step(actions)
                if self.normalize_rewards:
                    b["reversed_discounted_returns"][:, t + 1] = oracle.rdr_step(
                        np.ascontiguousarray(b["reversed_discounted_returns"][:, t]), reward, self.gamma)
                b["actions"][:, t] = actions
                b["logp"][:, t] = logp
                b["values"][:, t] = values.numpy()
                b["rewards"][:, t] = reward
                b["obs"][:, t + 1] = obs
            _, values = self.model(torch.from_numpy(np.ascontiguousarray(b["obs"][:, h])))
            b["values"][:, h] = values.numpy()
        stats = oracle.rollout_stats(b["rewards"], b["reversed_discounted_returns"] if self.normalize_rewards else None)
        self.reward_scale = stats.pop("reward_scale") if self.normalize_rewards else 1.0
        stats.pop("reward_scale", None)
        self.horizons += 1
        stats["env/resets"] = self.n * int(env_was_reset)
        stats["env/steps"] = self.n * h
        stats["profiling/collect_ms"] = (time.perf_counter() - t0) * 1e3
        return stats

    # -- step --------------------------------------------------------------
    def _loss_backward(self, idx: None | np.ndarray, flat: dict[str, np.ndarray], entropy_coeff: float):
        mb = flat if idx is None else {k: oracle.gather_rows(idx, v) for k, v in flat.items()}
        obs = torch.from_numpy(mb["obs"])
        feats, values = self.model(obs)
        hp = oracle.ppo_hparams(grad_accumulation_steps=self.gas, **{**self.hp_kw, "entropy_coeff": entropy_coeff})
        common = (values.detach().numpy(), mb["actions"], mb["logp"], mb["advantages"], mb["returns"])
        if self.distribution == "categorical":
            losses, g_logits, g_values = oracle.ppo_loss_categorical(feats["logits"].detach().numpy(), *common, hp)
            torch.autograd.backward([feats["logits"], values], [torch.from_numpy(g_logits), torch.from_numpy(g_values)])
        else:
            losses, g_mean, g_ls, g_values = oracle.ppo_loss_normal(
                feats["mean"].detach().numpy(), feats["log_std"].detach().numpy(), *common, hp,
                squashed=self.distribution == "squashed")
            torch.autograd.backward(
                [feats["mean"], feats["log_std"], values],
                [torch.from_numpy(g_mean), torch.from_numpy(g_ls), torch.from_numpy(g_values)])
        return losses

    def step(self, *, perms: None | list[np.ndarray] = None) -> dict[str, float]:
        t0 = time.perf_counter()
        b, h, n = self.buf, self.h, self.n
        out = oracle.gae(b["rewards"], b["values"], gamma=self.gamma, gae_lambda=self.gae_lambda,
                         reward_scale=self.reward_scale, normalize_advantages=self.normalize_advantages)
        b["advantages"], b["returns"] = out["advantages"], out["returns"]
        final_obs = b["obs"][:, -1].copy()
        flat = {k: np.ascontiguousarray(b[k][:, :h]).reshape(n * h, -1)
                for k in ("obs", "actions", "logp", "advantages", "returns")}
        entropy_coeff = self.hp_kw["entropy_coeff"]
        sums = {k: 0.0 for k in ("entropy", "policy", "vf", "total", "kl")}
        avgs = {k: [0.0, 0] for k in sums}
        for it in range(self.num_sgd_iters):
            if self.shuffle:
                perm = perms[it] if perms is not None else oracle.permutation(n * h, self.seed, self.horizons * 1000 + it)
            else:
                perm = np.arange(n * h)
            for i in range(self.num_minibatches):
                idx = None if (self.num_minibatches == 1 and perms is None) else perm[i * self.mb:(i + 1) * self.mb]
                step_this_batch = (i + 1) % self.gas == 0
                losses = self._loss_backward(idx, flat, entropy_coeff)
                losses["kl"] = losses["kl"] / self.gas
                for k in sums:
                    sums[k] += losses[k]
                if step_this_batch:
                    for k in sums:
                        avg, cnt = avgs[k]
                        avgs[k] = [(sums[k] + cnt * avg) / (cnt + 1), cnt + 1]
                        sums[k] = 0.0
                    nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
                    self.optimizer.step()
                    self.optimizer.zero_grad()
        for v in b.values():
            v[...] = 0
        b["obs"][:, -1] = final_obs
        return {
            "coefficients/entropy": entropy_coeff,
            "coefficients/vf": self.hp_kw["vf_coeff"],
            "losses/entropy": avgs["entropy"][0],
            "losses/policy": avgs["policy"][0],
            "losses/vf": avgs["vf"][0],
            "losses/total": avgs["total"][0],
            "monitors/kl_div": avgs["kl"][0],
            "profiling/step_ms": (time.perf_counter() - t0) * 1e3,
        }


def load_reference_weights(model: nn.Module, g: dict[str, np.ndarray], prefix: str = "init_") -> None:
    model.load_state_dict({k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)})


def first_iteration_losses(g: dict[str, np.ndarray], it: int, _oracle: Any = oracle) -> dict[str, float]:
    """Losses of the FIRST SGD iteration implied by a trace's post-collect buffer
    and initial weights (full-batch traces; weights unchanged before it)."""
    n, h1 = g[f"it{it}_collect_rewards"].shape[:2]
    h = h1 - 1
    out = _oracle.gae(g[f"it{it}_collect_rewards"], g[f"it{it}_collect_values"], gamma=0.95, gae_lambda=0.95,
                      reward_scale=float(g[f"it{it}_reward_scale"]), normalize_advantages=True)
    model = DiscreteModel(1, 1, 2)
    load_reference_weights(model, g)
    flat = lambda a: np.ascontiguousarray(a[:, :h]).reshape(n * h, -1)  # noqa: E731
    with torch.no_grad():
        feats, values = model(torch.from_numpy(flat(g[f"it{it}_collect_obs"])))
    losses, _, _ = _oracle.ppo_loss_categorical(
        feats["logits"].numpy(), values.numpy(), flat(g[f"it{it}_collect_actions"]), flat(g[f"it{it}_collect_logp"]),
        flat(out["advantages"]), flat(out["returns"]), _oracle.ppo_hparams(vf_clip_param=5.0), grads=False)
    return {"losses/policy": losses["policy"], "losses/vf": losses["vf"], "losses/total": losses["total"],
            "monitors/kl_div": losses["kl"], "losses/entropy": losses["entropy"]}
