"""Golden-vector generator: runs the REAL reference (imported unmodified from
``/root/reference``) on small seeded inputs and stores inputs + outputs as
``.npz`` fixtures next to this file.

Runs only in the build container (``/root/reference`` does not exist on the GPU
box). The three third-party packages the reference imports but this image lacks
(``tensordict``, ``torchrl``, ``mlflow``) are provided by ``_stubs/`` -- containers
and no-op tracking only, no arithmetic (see ``_stubs/README.md``).

Usage::

    python tests/golden/generate_fixtures.py

Every fixture is data (inputs and expected outputs); no reference source text is
stored. Reference call sites are cited per fixture.

"""

from __future__ import annotations

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path[:0] = [
    os.path.join(HERE, "_stubs"),
    REPO,
    os.path.join(REFERENCE, "src"),
    REFERENCE,
]

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(8)

from tensordict import TensorDict  # noqa: E402  (stub -> rl8_amd.tensordict)

import rl8  # noqa: E402,F401
from rl8 import AlgorithmConfig, RecurrentAlgorithmConfig  # noqa: E402
from rl8.data import DataKeys  # noqa: E402
from rl8.distributions import Categorical, Normal, SquashedNormal  # noqa: E402
from rl8.env import ContinuousDummyEnv, DiscreteDummyEnv  # noqa: E402
from rl8.nn.functional import generalized_advantage_estimate, ppo_losses  # noqa: E402


def save(name: str, **arrays) -> None:
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path)} bytes, {len(out)} arrays")


# --------------------------------------------------------------------------- #
# F1: generalized_advantage_estimate  (src/rl8/nn/functional.py:50-123)
# --------------------------------------------------------------------------- #
def gen_gae() -> None:
    arrays = {}
    cases = []
    g = torch.Generator().manual_seed(1)
    # (name, N, H, gamma, lambda, reward_scale, normalize)
    specs = [
        ("a", 64, 32, 0.95, 0.95, 1.0, False),
        ("b", 64, 32, 0.95, 0.95, 1.0, True),
        ("c", 64, 32, 0.95, 0.95, 271.828, True),
        ("d", 70, 15, 0.99, 0.9, 3.5, True),  # ragged N, even H+1
        ("e", 3, 1, 0.95, 0.95, 2.0, True),  # horizon 1
        ("f", 257, 7, 1.0, 1.0, 1.0, False),
        ("g", 33, 128, 0.95, 0.95, 17.0, True),  # CartPole-like horizon
    ]
    for name, n, h, gamma, lam, scale, norm in specs:
        state = torch.empty(n, 1).uniform_(-100, 100, generator=g)
        rewards = -(state + torch.randn(n, h + 1, generator=g).cumsum(1)).abs()
        rewards = rewards.unsqueeze(-1).contiguous()
        values = torch.randn(n, h + 1, 1, generator=g)
        batch = TensorDict(
            {DataKeys.REWARDS: rewards.clone(), DataKeys.VALUES: values.clone()},
            batch_size=[n, h + 1],
        )
        out = generalized_advantage_estimate(
            batch,
            gae_lambda=lam,
            gamma=gamma,
            inplace=False,
            normalize_advantages=norm,
            return_returns=True,
            reward_scale=scale,
        )
        arrays[f"{name}_rewards"] = rewards
        arrays[f"{name}_values"] = values
        arrays[f"{name}_scaled_rewards"] = batch[DataKeys.REWARDS]
        arrays[f"{name}_advantages"] = out[DataKeys.ADVANTAGES]
        arrays[f"{name}_returns"] = out[DataKeys.RETURNS]
        arrays[f"{name}_params"] = np.array([gamma, lam, scale, float(norm)], np.float64)
        cases.append(name)
    # Reference known-answer test (tests/test_nn/test_functional.py:14-49).
    n, h = 10, 5
    batch = TensorDict(
        {
            DataKeys.REWARDS: torch.ones(n, h + 1, 1),
            DataKeys.VALUES: torch.ones(n, h + 1, 1),
        },
        batch_size=[n, h + 1],
    )
    out = generalized_advantage_estimate(
        batch, gae_lambda=1, gamma=1, inplace=False, normalize_advantages=False
    )
    undiscounted = torch.flip(torch.cumsum(torch.ones(n, h + 1, 1), dim=1), dims=(1,))
    assert (out[DataKeys.ADVANTAGES] == undiscounted - 1).all()
    assert (out[DataKeys.RETURNS] == undiscounted).all()
    arrays["kat_advantages"] = out[DataKeys.ADVANTAGES]
    arrays["kat_returns"] = out[DataKeys.RETURNS]
    arrays["cases"] = np.array(cases)
    save("gae.npz", **arrays)


# --------------------------------------------------------------------------- #
# F2: ppo_losses + approximate KL + autograd grads
#     (src/rl8/nn/functional.py:259-363, algorithms/_feedforward.py:552-559,
#      distributions.py:113-170)
# --------------------------------------------------------------------------- #
def gen_ppo_losses() -> None:
    arrays = {}
    cases = []
    g = torch.Generator().manual_seed(2)
    m = 384

    def common():
        adv = torch.randn(m, 1, generator=g)
        returns = torch.randn(m, 1, generator=g) * 3
        values = (returns + torch.randn(m, 1, generator=g) * 2).requires_grad_(True)
        return adv, returns, values

    def finish(name, dist, actions, logp_old, adv, returns, values, feats, hp):
        buffer_batch = TensorDict(
            {
                DataKeys.ACTIONS: actions,
                DataKeys.LOGP: logp_old,
                DataKeys.ADVANTAGES: adv,
                DataKeys.RETURNS: returns,
            },
            batch_size=[m],
        )
        sample_batch = TensorDict({DataKeys.VALUES: values}, batch_size=[m])
        losses = ppo_losses(buffer_batch, sample_batch, dist, **hp)
        losses["total"].backward()
        with torch.no_grad():
            lr = dist.logp(actions) - logp_old
            kl = torch.mean((torch.exp(lr) - 1) - lr)
        arrays[f"{name}_actions"] = actions
        arrays[f"{name}_logp_old"] = logp_old
        arrays[f"{name}_advantages"] = adv
        arrays[f"{name}_returns"] = returns
        arrays[f"{name}_values"] = values
        arrays[f"{name}_grad_values"] = values.grad
        for k, f in feats.items():
            arrays[f"{name}_feat_{k}"] = f
            arrays[f"{name}_grad_{k}"] = f.grad
        arrays[f"{name}_losses"] = np.array(
            [
                float(losses["entropy"].reshape(-1)[0]),
                float(losses["policy"]),
                float(losses["vf"]),
                float(losses["total"].reshape(-1)[0]),
                float(kl),
            ],
            np.float64,
        )
        arrays[f"{name}_hparams"] = np.array(
            [
                hp["clip_param"],
                hp["dual_clip_param"] if hp["dual_clip_param"] else 0.0,
                hp["entropy_coeff"],
                hp["vf_clip_param"],
                hp["vf_coeff"],
            ],
            np.float64,
        )
        cases.append(name)

    hps = {
        "p0": dict(clip_param=0.2, dual_clip_param=None, entropy_coeff=0.0, vf_clip_param=5.0, vf_coeff=1.0),
        "p1": dict(clip_param=0.2, dual_clip_param=5.0, entropy_coeff=0.0, vf_clip_param=5.0, vf_coeff=1.0),
        "p2": dict(clip_param=0.3, dual_clip_param=None, entropy_coeff=1e-2, vf_clip_param=1.0, vf_coeff=0.5),
        "p3": dict(clip_param=0.1, dual_clip_param=3.0, entropy_coeff=1e-2, vf_clip_param=2.0, vf_coeff=2.0),
    }
    for ncls in (2, 3, 5):
        for hname, hp in hps.items():
            adv, returns, values = common()
            logits = (torch.randn(m, 1, ncls, generator=g) * 1.5).requires_grad_(True)
            actions = torch.randint(0, ncls, (m, 1), generator=g)
            dist = Categorical(TensorDict({"logits": logits}, batch_size=[m]), None)
            with torch.no_grad():
                logp_old = dist.logp(actions) + torch.randn(m, 1, generator=g) * 0.4
            finish(f"cat{ncls}_{hname}", dist, actions, logp_old, adv, returns, values, {"logits": logits}, hp)
    for dname, dcls in (("normal", Normal), ("squashed", SquashedNormal)):
        for hname, hp in hps.items():
            if dname == "squashed" and hp["entropy_coeff"] != 0:
                continue
            for adim in (1, 2):
                adv, returns, values = common()
                mean = torch.randn(m, adim, generator=g).requires_grad_(True)
                log_std = torch.tanh(torch.randn(m, adim, generator=g)).requires_grad_(True)
                feats = TensorDict({"mean": mean, "log_std": log_std}, batch_size=[m])
                dist = dcls(feats, None)
                with torch.no_grad():
                    actions = dcls(feats, None).sample()
                    if dname == "squashed":
                        # exercise the clamp branch too
                        actions[:4] = torch.tensor([[1.0] * adim, [-1.0] * adim, [0.0] * adim, [0.9999999] * adim])
                    logp_old = dist.logp(actions) + torch.randn(m, 1, generator=g) * 0.4
                finish(
                    f"{dname}{adim}_{hname}", dist, actions, logp_old, adv, returns,
                    values, {"mean": mean, "log_std": log_std}, hp,
                )
    arrays["cases"] = np.array(cases)
    save("ppo_losses.npz", **arrays)


# --------------------------------------------------------------------------- #
# F3: env steps (src/rl8/env.py:224-230,253-259; examples/cartpole/env.py:12-64)
# --------------------------------------------------------------------------- #
def gen_env_steps() -> None:
    arrays = {}
    g = torch.Generator().manual_seed(3)
    n, steps = 257, 6
    # Discrete dummy.
    env = DiscreteDummyEnv(n, 32)
    env.state = torch.empty(n, 1).uniform_(-100, 100, generator=g)
    arrays["disc_state0"] = env.state.clone()
    acts, states, rewards = [], [], []
    for _ in range(steps):
        a = torch.randint(0, 2, (n, 1), generator=g)
        out = env.step(a)
        acts.append(a)
        states.append(out[DataKeys.OBS].clone())
        rewards.append(out[DataKeys.REWARDS].clone())
    arrays["disc_actions"] = torch.stack(acts)
    arrays["disc_states"] = torch.stack(states)
    arrays["disc_rewards"] = torch.stack(rewards)
    # Continuous dummy.
    env = ContinuousDummyEnv(n, 32)
    env.state = torch.empty(n, 1).uniform_(-100, 100, generator=g)
    arrays["cont_state0"] = env.state.clone()
    acts, states, rewards = [], [], []
    for _ in range(steps):
        a = torch.tanh(torch.randn(n, 1, generator=g))
        out = env.step(a)
        acts.append(a)
        states.append(out[DataKeys.OBS].clone())
        rewards.append(out[DataKeys.REWARDS].clone())
    arrays["cont_actions"] = torch.stack(acts)
    arrays["cont_states"] = torch.stack(states)
    arrays["cont_rewards"] = torch.stack(rewards)

    # CartPole, eager body of the compiled step (the function torch.compile wraps).
    from examples.cartpole import env as cartpole_env

    eager_step = cartpole_env.step
    for attr in ("_torchdynamo_orig_callable", "__wrapped__"):
        if hasattr(eager_step, attr):
            eager_step = getattr(eager_step, attr)
            break
    from dataclasses import asdict

    for integ in ("euler", "semi-implicit"):
        cfg = cartpole_env.CartPoleConfig(kinematics_integrator=integ)
        state = torch.normal(0, 0.5, size=(4, n), generator=g)
        arrays[f"cp_{integ}_state0"] = state.clone()
        acts, states, obss, rewards = [], [], [], []
        for _ in range(steps):
            a = torch.randint(0, 3, (n, 1), generator=g)
            x, x_dot, theta, theta_dot = state
            state, obs, reward = eager_step(x, x_dot, theta, theta_dot, a, **asdict(cfg))
            acts.append(a)
            states.append(state.clone())
            obss.append(obs.contiguous().clone())
            rewards.append(reward.clone())
        arrays[f"cp_{integ}_actions"] = torch.stack(acts)
        arrays[f"cp_{integ}_states"] = torch.stack(states)
        arrays[f"cp_{integ}_obs"] = torch.stack(obss)
        arrays[f"cp_{integ}_rewards"] = torch.stack(rewards)
    # Non-default physics.
    cfg = cartpole_env.CartPoleConfig(force_mag=7.5, gravity=3.7, length=0.8, pole_mass=0.25, cart_mass=2.0, tau=0.01)
    state = torch.normal(0, 0.5, size=(4, n), generator=g)
    a = torch.randint(0, 3, (n, 1), generator=g)
    x, x_dot, theta, theta_dot = state
    s2, obs, reward = eager_step(x, x_dot, theta, theta_dot, a, **asdict(cfg))
    arrays["cp_custom_cfg"] = np.array(
        [cfg.force_mag, cfg.gravity, cfg.length, cfg.pole_mass, cfg.pole_mass_length, cfg.total_mass, cfg.tau],
        np.float64,
    )
    arrays["cp_custom_state0"] = state
    arrays["cp_custom_actions"] = a
    arrays["cp_custom_state1"] = s2
    arrays["cp_custom_obs"] = obs.contiguous()
    arrays["cp_custom_rewards"] = reward
    save("env_steps.npz", **arrays)


# --------------------------------------------------------------------------- #
# F3b: the other example envs (N2): examples/mountain_car/env.py:12-38,
#      examples/pendulum/env.py:12-39 -- eager bodies of the compiled steps
# --------------------------------------------------------------------------- #
def _eager(fn):
    for attr in ("_torchdynamo_orig_callable", "__wrapped__"):
        if hasattr(fn, attr):
            return getattr(fn, attr)
    return fn


def gen_classic_env_steps() -> None:
    from dataclasses import asdict

    from examples.mountain_car import env as mc_env
    from examples.pendulum import env as pd_env

    arrays = {}
    g = torch.Generator().manual_seed(5)
    n, steps = 257, 8
    mc_step, pd_step = _eager(mc_env.step), _eager(pd_env.step)
    for tag, cfg in (
        ("mc_default", mc_env.MountainCarConfig()),
        ("mc_custom", mc_env.MountainCarConfig(force_mag=0.0015, goal_position=0.45, goal_velocity=0.01,
                                               gravity=0.002, max_position=0.55, max_speed=0.06, min_position=-1.1)),
    ):
        # spread over the whole track so the wall, the clips and the goal all occur
        position = torch.empty(n).uniform_(cfg.min_position - 0.02, cfg.max_position + 0.02, generator=g)
        position = position.clip(cfg.min_position, cfg.max_position)
        velocity = torch.empty(n).uniform_(-1.2 * cfg.max_speed, 1.2 * cfg.max_speed, generator=g)
        state = torch.vstack((position, velocity))
        arrays[f"{tag}_cfg"] = np.array(list(asdict(cfg).values()), np.float64)
        arrays[f"{tag}_cfg_keys"] = np.array(list(asdict(cfg).keys()))
        arrays[f"{tag}_state0"] = state.clone()
        acts, states, obss, rewards = [], [], [], []
        for _ in range(steps):
            a = torch.randint(0, 3, (n, 1), generator=g)
            p, v = state
            state, obs, reward = mc_step(p, v, a, **asdict(cfg))
            acts.append(a)
            states.append(state.clone())
            obss.append(obs.contiguous().clone())
            rewards.append(reward.clone())
        arrays[f"{tag}_actions"] = torch.stack(acts)
        arrays[f"{tag}_states"] = torch.stack(states)
        arrays[f"{tag}_obs"] = torch.stack(obss)
        arrays[f"{tag}_rewards"] = torch.stack(rewards)
    for tag, cfg in (
        ("pd_default", pd_env.PendulumConfig()),
        ("pd_custom", pd_env.PendulumConfig(dt=0.02, g=9.81, l=0.7, m=1.3, max_speed=6.0, max_torque=1.5)),
    ):
        th = torch.empty(n).uniform_(-3 * torch.pi, 3 * torch.pi, generator=g)
        thdot = torch.empty(n).uniform_(-1.1 * cfg.max_speed, 1.1 * cfg.max_speed, generator=g)
        state = torch.vstack((th, thdot))
        arrays[f"{tag}_cfg"] = np.array(list(asdict(cfg).values()), np.float64)
        arrays[f"{tag}_cfg_keys"] = np.array(list(asdict(cfg).keys()))
        arrays[f"{tag}_state0"] = state.clone()
        acts, states, obss, rewards = [], [], [], []
        for _ in range(steps):
            a = torch.randn(n, 1, generator=g) * 1.5  # beyond +-max_torque now and then
            t, td = state
            state, obs, reward = pd_step(t, td, a, **asdict(cfg))
            acts.append(a)
            states.append(state.clone())
            obss.append(obs.contiguous().clone())
            rewards.append(reward.clone())
        arrays[f"{tag}_actions"] = torch.stack(acts)
        arrays[f"{tag}_states"] = torch.stack(states)
        arrays[f"{tag}_obs"] = torch.stack(obss)
        arrays[f"{tag}_rewards"] = torch.stack(rewards)
    save("classic_env_steps.npz", **arrays)


# --------------------------------------------------------------------------- #
# F3c: view requirements (N4): src/rl8/views.py:54-453
# --------------------------------------------------------------------------- #
def _flatten_td(prefix, item, arrays):
    if torch.is_tensor(item):
        arrays[prefix] = item.contiguous().clone()
        return
    arrays[prefix + "__batch"] = np.array(list(item.batch_size), np.int64)
    for k in item.keys():
        _flatten_td(f"{prefix}__{k}", item[k], arrays)


def gen_views() -> None:
    from rl8.views import (PaddedRollingWindow, RollingWindow, ViewRequirement, pad_last_sequence,
                           pad_whole_sequence, rolling_window)

    arrays = {}
    g = torch.Generator().manual_seed(9)
    cases = {
        "a": torch.randn(3, 7, 2, generator=g),
        "b": torch.randn(2, 1, 4, generator=g),      # shorter than the window
        "c": torch.randn(4, 5, generator=g),         # no feature dimension
        "d": torch.randn(2, 6, 2, 3, generator=g),   # two feature dimensions
    }
    names = []
    for name, x in cases.items():
        arrays[f"{name}_x"] = x
        for size in (1, 2, 3, 5):
            tag = f"{name}_s{size}"
            names.append(tag)
            _flatten_td(f"{tag}_pad_last", pad_last_sequence(x, size), arrays)
            _flatten_td(f"{tag}_pad_whole", pad_whole_sequence(x, size), arrays)
            _flatten_td(f"{tag}_padded_all", PaddedRollingWindow.apply_all(x, size), arrays)
            _flatten_td(f"{tag}_padded_last", PaddedRollingWindow.apply_last(x, size), arrays)
            _flatten_td(f"{tag}_rolling_last", RollingWindow.apply_last(x, size), arrays)
            if size <= x.shape[1]:
                arrays[f"{tag}_window"] = rolling_window(x, size).contiguous()
                arrays[f"{tag}_window_step2"] = rolling_window(x, size, step=2).contiguous()
                arrays[f"{tag}_rolling_all"] = RollingWindow.apply_all(x, size).contiguous()
    arrays["tensor_cases"] = np.array(names)
    # nested tensordict through ViewRequirement (keys as a model would use them)
    td = TensorDict(
        {"obs": TensorDict({"prices": torch.randn(3, 6, 2, generator=g), "volume": torch.randn(3, 6, generator=g)},
                           batch_size=[3, 6])},
        batch_size=[3, 6],
    )
    arrays["td_prices"] = td["obs"]["prices"]
    arrays["td_volume"] = td["obs"]["volume"]
    for method in ("rolling_window", "padded_rolling_window"):
        for shift in (0, 2):
            vr = ViewRequirement(shift=shift, method=method)
            tag = f"td_{method}_shift{shift}"
            _flatten_td(f"{tag}_all", vr.apply_all("obs", td), arrays)
            _flatten_td(f"{tag}_last", vr.apply_last("obs", td), arrays)
            _flatten_td(f"{tag}_all_leaf", vr.apply_all(("obs", "prices"), td), arrays)
            _flatten_td(f"{tag}_last_leaf", vr.apply_last(("obs", "prices"), td), arrays)
            arrays[f"{tag}_drop_size"] = np.array(vr.drop_size)
    save("views.npz", **arrays)


# --------------------------------------------------------------------------- #
# F4: samplers (src/rl8/distributions.py:113-170 -> torch.distributions)
# --------------------------------------------------------------------------- #
def draw_exponential_like(probs: torch.Tensor) -> torch.Tensor:
    """The noise ``torch.multinomial`` will draw next (num_samples=1 path:
    ``q = empty_like(p).exponential_(1)``, result ``argmax(p / q)``), read without
    advancing the default generator."""
    state = torch.get_rng_state()
    q = torch.empty_like(probs).exponential_(1)
    torch.set_rng_state(state)
    return q


def gen_samplers() -> None:
    arrays = {}
    torch.manual_seed(4)
    m = 4096
    for ncls in (2, 3, 5):
        scale = 1e-3 if ncls == 2 else 2.0
        logits = torch.randn(m, 1, ncls) * scale
        dist = Categorical(TensorDict({"logits": logits}, batch_size=[m]), None)
        q = draw_exponential_like(dist.dist.probs.reshape(-1, ncls))
        actions = dist.sample()
        check = torch.argmax(dist.dist.probs.reshape(-1, ncls) / q, dim=-1, keepdim=True)
        assert (check == actions).all(), "multinomial != argmax(p/q): torch changed"
        arrays[f"cat{ncls}_logits"] = logits
        arrays[f"cat{ncls}_q"] = q.reshape(m, 1, ncls)
        arrays[f"cat{ncls}_actions"] = actions
        arrays[f"cat{ncls}_logp"] = dist.logp(actions)
        arrays[f"cat{ncls}_entropy"] = dist.entropy()
        arrays[f"cat{ncls}_mode"] = dist.deterministic_sample()
    for adim in (1, 3):
        mean = torch.randn(m, adim)
        log_std = torch.tanh(torch.randn(m, adim))
        feats = TensorDict({"mean": mean, "log_std": log_std}, batch_size=[m])
        for dname, dcls in (("normal", Normal), ("squashed", SquashedNormal)):
            dist = dcls(feats, None)
            state = torch.get_rng_state()
            eps = torch.randn(m, adim)
            torch.set_rng_state(state)
            actions = dist.sample()
            ref = mean + torch.exp(log_std) * eps
            if dname == "squashed":
                ref = ref.tanh()
            assert (ref == actions).all(), "Normal.sample != loc + scale*randn"
            arrays[f"{dname}{adim}_mean"] = mean
            arrays[f"{dname}{adim}_log_std"] = log_std
            arrays[f"{dname}{adim}_eps"] = eps
            arrays[f"{dname}{adim}_actions"] = actions
            arrays[f"{dname}{adim}_logp"] = dist.logp(actions)
            arrays[f"{dname}{adim}_mode"] = dist.deterministic_sample()
            if dname == "normal":
                arrays[f"{dname}{adim}_entropy"] = dist.entropy()
    save("samplers.npz", **arrays)


# --------------------------------------------------------------------------- #
# F5/F6: end-to-end traces of Algorithm.collect()/step()
#        (src/rl8/algorithms/_feedforward.py:301-615)
# --------------------------------------------------------------------------- #
class Recorder:
    """Records the noise torch draws inside the reference's collect()/step()
    by wrapping torch entry points (never reference code)."""

    def __init__(self) -> None:
        self.cat_q: list[torch.Tensor] = []
        self.normal_eps: list[torch.Tensor] = []
        self.perms: list[torch.Tensor] = []
        self.resets: list[torch.Tensor] = []

    def __enter__(self):
        rec = self
        self._cat_sample = torch.distributions.Categorical.sample
        self._normal_sample = torch.distributions.Normal.sample
        self._randperm = torch.randperm

        def cat_sample(dist, sample_shape=torch.Size()):
            probs_2d = dist.probs.reshape(-1, dist._num_events)
            rec.cat_q.append(draw_exponential_like(probs_2d).reshape(dist.probs.shape))
            return rec._cat_sample(dist, sample_shape)

        def normal_sample(dist, sample_shape=torch.Size()):
            state = torch.get_rng_state()
            rec.normal_eps.append(torch.randn(dist.loc.shape))
            torch.set_rng_state(state)
            out = rec._normal_sample(dist, sample_shape)
            assert (out == dist.loc + dist.scale * rec.normal_eps[-1]).all()
            return out

        def randperm(*args, **kwargs):
            out = rec._randperm(*args, **kwargs)
            rec.perms.append(out.clone())
            return out

        torch.distributions.Categorical.sample = cat_sample
        torch.distributions.Normal.sample = normal_sample
        torch.randperm = randperm
        return self

    def __exit__(self, *exc):
        torch.distributions.Categorical.sample = self._cat_sample
        torch.distributions.Normal.sample = self._normal_sample
        torch.randperm = self._randperm


def snapshot_buffer(buffer, prefix, arrays):
    for k, v in buffer.items():
        if torch.is_tensor(v):
            arrays[f"{prefix}_{k}"] = v.clone()
        else:  # recurrent states: keep the last column and one mid-sequence column
            for sk, sv in v.items():
                arrays[f"{prefix}_{k}_{sk}_last"] = sv[:, -1].clone()
                arrays[f"{prefix}_{k}_{sk}_col6"] = sv[:, 6].clone()


def gen_trace(name, env_cls, config_kwargs, iterations=2, recurrent=False) -> None:
    arrays = {}
    torch.manual_seed(42)
    cfg_cls = RecurrentAlgorithmConfig if recurrent else AlgorithmConfig
    algo = cfg_cls(num_envs=64, horizon=32, device="cpu", **config_kwargs).build(env_cls)
    for k, v in algo.policy.model.state_dict().items():
        arrays[f"init_{k}"] = v.clone()
    collect_keys = None
    step_keys = None
    for it in range(iterations):
        with Recorder() as rec:
            wrapped_reset = algo.env.reset

            def reset(*, config=None, _r=wrapped_reset):
                out = _r(config=config)
                rec.resets.append(out.clone())
                return out

            algo.env.reset = reset
            collect_stats = algo.collect()
            algo.env.reset = wrapped_reset
            snapshot_buffer(algo.buffer, f"it{it}_collect", arrays)
            arrays[f"it{it}_reward_scale"] = np.float64(algo.state.reward_scale)
            step_stats = algo.step()
        if rec.resets:
            arrays[f"it{it}_reset_state"] = rec.resets[0]
        if rec.cat_q:
            arrays[f"it{it}_cat_q"] = torch.stack(rec.cat_q)
        if rec.normal_eps:
            arrays[f"it{it}_normal_eps"] = torch.stack(rec.normal_eps)
        arrays[f"it{it}_perms"] = torch.stack(rec.perms)
        collect_keys = sorted(k for k in collect_stats if not k.startswith("profiling"))
        step_keys = sorted(k for k in step_stats if not k.startswith("profiling"))
        arrays[f"it{it}_collect_stats"] = np.array([collect_stats[k] for k in collect_keys], np.float64)
        arrays[f"it{it}_step_stats"] = np.array([step_stats[k] for k in step_keys], np.float64)
        for k, v in algo.policy.model.state_dict().items():
            arrays[f"it{it}_final_{k}"] = v.clone()
        arrays[f"it{it}_final_obs"] = algo.buffer[DataKeys.OBS][:, -1].clone()
    arrays["collect_stat_keys"] = np.array(collect_keys)
    arrays["step_stat_keys"] = np.array(step_keys)
    arrays["config"] = np.array([f"{k}={v}" for k, v in config_kwargs.items()] or ["default"])
    save(name, **arrays)


# --------------------------------------------------------------------------- #
# F7: first-update pins of the assembled path (reference-held numbers for the
#     1e-5 end-to-end bar): per-minibatch StatTracker updates of the traced
#     configs (src/rl8/algorithms/_feedforward.py:562-574, _recurrent.py twin)
#     and a num_sgd_iters=1 / one-minibatch run of every traced variant with the
#     gradient the reference hands to its first optimizer step (:585-590).
# --------------------------------------------------------------------------- #
STAT_KEYS = (
    "coefficients/entropy", "coefficients/vf", "losses/entropy", "losses/policy",
    "losses/vf", "losses/total", "monitors/kl_div",
)


class UpdateRecorder:
    """Records every ``StatTracker.update`` call (data + reduce flag) and the
    gradients present at the first ``optimizer.step`` by wrapping the methods at
    run time (the reference source stays untouched)."""

    def __init__(self, algo) -> None:
        self.algo = algo
        self.updates: list[list[float]] = []
        self.first_grads: None | dict[str, torch.Tensor] = None

    def __enter__(self):
        import rl8._utils as ref_utils

        rec = self
        self._cls = ref_utils.StatTracker
        self._update = ref_utils.StatTracker.update
        self._opt_step = self.algo.optimizer.step

        def update(tracker, data, /, *, reduce=False):
            rec.updates.append([float(data[k]) for k in STAT_KEYS] + [float(reduce)])
            return rec._update(tracker, data, reduce=reduce)

        def opt_step(*args, **kwargs):
            if rec.first_grads is None:
                rec.first_grads = {
                    k: p.grad.detach().clone() for k, p in rec.algo.policy.model.named_parameters()
                    if p.grad is not None
                }
            return rec._opt_step(*args, **kwargs)

        ref_utils.StatTracker.update = update
        self.algo.optimizer.step = opt_step
        return self

    def __exit__(self, *exc):
        self._cls.update = self._update
        self.algo.optimizer.step = self._opt_step


def gen_first_update(name, trace_name, env_cls, config_kwargs, recurrent=False) -> None:
    trace = dict(np.load(os.path.join(HERE, trace_name)))
    cfg_cls = RecurrentAlgorithmConfig if recurrent else AlgorithmConfig
    arrays = {}

    def build(**overrides):
        torch.manual_seed(42)
        kwargs = {**config_kwargs, **overrides}
        algo = cfg_cls(num_envs=64, horizon=32, device="cpu", **kwargs).build(env_cls)
        for k, v in algo.policy.model.state_dict().items():
            assert np.array_equal(v.numpy(), trace[f"init_{k}"]), f"init weights differ from {trace_name}: {k}"
        return algo

    def check_collect(algo):
        # the rollout must be the one the committed trace holds: same inputs for the tests
        for k, v in algo.buffer.items():
            if torch.is_tensor(v):
                assert np.array_equal(v.numpy(), trace[f"it0_collect_{k}"]), f"collect differs from {trace_name}: {k}"

    # (a) the traced config itself: every per-minibatch update of iteration 0
    algo = build()
    algo.collect()
    check_collect(algo)
    with UpdateRecorder(algo) as rec:
        step_stats = algo.step()
    keys = [str(k) for k in trace["step_stat_keys"]]
    for k, w in zip(keys, trace["it0_step_stats"]):
        assert step_stats[k] == w, f"{trace_name}: re-run differs at {k}: {step_stats[k]} vs {w}"
    arrays["traced_updates"] = np.array(rec.updates, np.float64)

    # (b) one SGD iteration over one full-buffer minibatch
    algo = build(num_sgd_iters=1, sgd_minibatch_size=None)
    algo.collect()
    check_collect(algo)
    with UpdateRecorder(algo) as rec:
        step_stats = algo.step()
    assert len(rec.updates) == 1
    arrays["sgd1_updates"] = np.array(rec.updates, np.float64)
    arrays["sgd1_step_stats"] = np.array([step_stats[k] for k in keys], np.float64)
    total_sq = 0.0
    for k, gval in rec.first_grads.items():
        arrays[f"sgd1_grad_{k}"] = gval
        total_sq += float((gval.double() ** 2).sum())
    arrays["sgd1_clipped_grad_norm"] = np.float64(total_sq ** 0.5)
    for k, v in algo.policy.model.state_dict().items():
        arrays[f"sgd1_final_{k}"] = v.clone()
    arrays["stat_keys"] = np.array(STAT_KEYS + ("reduce",))
    arrays["step_stat_keys"] = np.array(keys)
    save(name, **arrays)


TRACED_VARIANTS = [
    # (first-update fixture, trace it extends, env, config, recurrent)
    ("first_update_ff_discrete.npz", "trace_ff_discrete.npz", "discrete", {}, False),
    ("first_update_ff_discrete_minibatch.npz", "trace_ff_discrete_minibatch.npz", "discrete",
     dict(sgd_minibatch_size=256, entropy_coeff=1e-2, dual_clip_param=5.0, horizons_per_env_reset=2), False),
    ("first_update_ff_continuous_squashed.npz", "trace_ff_continuous_squashed.npz", "continuous",
     dict(distribution_cls=SquashedNormal), False),
    ("first_update_ff_continuous_normal.npz", "trace_ff_continuous_normal.npz", "continuous",
     dict(entropy_coeff=1e-2), False),
    ("first_update_rec_discrete.npz", "trace_rec_discrete.npz", "discrete", {}, True),
    ("first_update_rec_continuous_minibatch.npz", "trace_rec_continuous_minibatch.npz", "continuous",
     dict(sgd_minibatch_size=128, entropy_coeff=1e-2, seq_len=8, seqs_per_state_reset=2, horizons_per_env_reset=2),
     True),
]


def gen_first_updates() -> None:
    envs = {"discrete": DiscreteDummyEnv, "continuous": ContinuousDummyEnv}
    for name, trace_name, env, cfg, recurrent in TRACED_VARIANTS:
        gen_first_update(name, trace_name, envs[env], cfg, recurrent=recurrent)


# --------------------------------------------------------------------------- #
# F7b: the SECOND iteration teacher-forced: the traces' iteration 1 is compared
#      free-running, on weights that drifted through iteration 0's Adam steps, at a wide band.
#      Here the reference's own iteration-1 numbers are recorded per minibatch, with the gradient
#      of its first optimizer step, so that a test can load the reference's it0_final weights
#      (already in the trace), replay iteration 1 and hold the first update to 1e-5 again:
#      a second rollout (carried observations, non-zero reward scale, used Adam state untouched
#      by the first update's statistics) through the shuffled / packed path.
# --------------------------------------------------------------------------- #
def gen_second_iteration(name, trace_name, env_cls, config_kwargs, recurrent=False) -> None:
    trace = dict(np.load(os.path.join(HERE, trace_name)))
    torch.manual_seed(42)
    cfg_cls = RecurrentAlgorithmConfig if recurrent else AlgorithmConfig
    algo = cfg_cls(num_envs=64, horizon=32, device="cpu", **config_kwargs).build(env_cls)
    algo.collect()
    algo.step()
    for k, v in algo.policy.model.state_dict().items():
        assert np.array_equal(v.numpy(), trace[f"it0_final_{k}"]), f"{trace_name}: it0 weights differ: {k}"
    algo.collect()
    for k, v in algo.buffer.items():
        if torch.is_tensor(v):
            assert np.array_equal(v.numpy(), trace[f"it1_collect_{k}"]), f"{trace_name}: it1 collect differs: {k}"
    with UpdateRecorder(algo) as rec:
        step_stats = algo.step()
    keys = [str(k) for k in trace["step_stat_keys"]]
    for k, w in zip(keys, trace["it1_step_stats"]):
        assert step_stats[k] == w, f"{trace_name}: it1 re-run differs at {k}"
    arrays = {"it1_updates": np.array(rec.updates, np.float64), "stat_keys": np.array(STAT_KEYS + ("reduce",))}
    total_sq = 0.0
    for k, gval in rec.first_grads.items():
        arrays[f"it1_first_grad_{k}"] = gval
        total_sq += float((gval.double() ** 2).sum())
    arrays["it1_first_clipped_grad_norm"] = np.float64(total_sq ** 0.5)
    save(name, **arrays)


def gen_second_iterations() -> None:
    envs = {"discrete": DiscreteDummyEnv, "continuous": ContinuousDummyEnv}
    for name, trace_name, env, cfg, recurrent in TRACED_VARIANTS:
        gen_second_iteration(name.replace("first_update_", "second_iteration_"), trace_name, envs[env], cfg,
                             recurrent=recurrent)


# --------------------------------------------------------------------------- #
# F7c: self-contained two-iteration fixtures for what the traces do not reach:
#      * a recurrent config whose LSTM states ARE carried across collect() calls
#        (src/rl8/algorithms/_recurrent.py:380-392: last column -> first column; :384-392: re-initialised only
#        where state.seqs hits seqs_per_state_reset) -- both traced recurrent configs re-initialise at t = 0 of
#        every collect(), so their second iteration never reads a carried state;
#      * CartPole with horizons_per_env_reset = 2: the [4, N] physics state (examples/cartpole/env.py:128-150)
#        and the last observation / reversed discounted return carried into a used buffer.
#      Everything a teacher-forced replay of iteration 1 needs is recorded: iteration 0's inputs, the reference's
#      weights / env state / carried columns after iteration 0, iteration 1's noise and permutations, its rollout,
#      every StatTracker.update and the gradient at its first optimizer step.
# --------------------------------------------------------------------------- #
def gen_two_iterations(name, env_cls, config_kwargs, *, recurrent=False, env_state=None) -> None:
    arrays = {}
    torch.manual_seed(42)
    cfg_cls = RecurrentAlgorithmConfig if recurrent else AlgorithmConfig
    algo = cfg_cls(num_envs=64, horizon=32, device="cpu", **config_kwargs).build(env_cls)
    env_state = env_state or (lambda env: env.state)
    for k, v in algo.policy.model.state_dict().items():
        arrays[f"init_{k}"] = v.clone()
    keys = step_keys = None
    for it in range(2):
        with Recorder() as rec:
            real_reset = algo.env.reset
            resets = []

            def reset(*, config=None, _r=real_reset):
                out = _r(config=config)
                resets.append(env_state(algo.env).clone())
                return out

            algo.env.reset = reset
            collect_stats = algo.collect()
            algo.env.reset = real_reset
            snapshot_buffer(algo.buffer, f"it{it}_collect", arrays)
            arrays[f"it{it}_reward_scale"] = np.float64(algo.state.reward_scale)
            arrays[f"it{it}_env_state"] = env_state(algo.env).clone()
            with UpdateRecorder(algo) as urec:
                step_stats = algo.step()
        assert len(resets) == (1 if it == 0 else 0), "the point of this fixture is a carried environment"
        if resets:
            arrays[f"it{it}_reset_state"] = resets[0]
        if rec.cat_q:
            arrays[f"it{it}_cat_q"] = torch.stack(rec.cat_q)
        if rec.normal_eps:
            arrays[f"it{it}_normal_eps"] = torch.stack(rec.normal_eps)
        arrays[f"it{it}_perms"] = torch.stack(rec.perms)
        keys = sorted(k for k in collect_stats if not k.startswith("profiling"))
        step_keys = sorted(k for k in step_stats if not k.startswith("profiling"))
        arrays[f"it{it}_collect_stats"] = np.array([collect_stats[k] for k in keys], np.float64)
        arrays[f"it{it}_step_stats"] = np.array([step_stats[k] for k in step_keys], np.float64)
        arrays[f"it{it}_updates"] = np.array(urec.updates, np.float64)
        arrays[f"it{it}_final_obs"] = algo.buffer[DataKeys.OBS][:, -1].clone()
        arrays[f"it{it}_final_rdr"] = algo.buffer[DataKeys.REVERSED_DISCOUNTED_RETURNS][:, -1].clone()
        if it == 0:  # what iteration 1 starts from
            for k, v in algo.policy.model.state_dict().items():
                arrays[f"it0_final_{k}"] = v.clone()
        else:
            total_sq = 0.0
            for k, gval in urec.first_grads.items():
                arrays[f"it1_first_grad_{k}"] = gval
                total_sq += float((gval.double() ** 2).sum())
            arrays["it1_first_clipped_grad_norm"] = np.float64(total_sq ** 0.5)
    arrays["collect_stat_keys"] = np.array(keys)
    arrays["step_stat_keys"] = np.array(step_keys)
    arrays["stat_keys"] = np.array(STAT_KEYS + ("reduce",))
    arrays["config"] = np.array([f"{k}={v}" for k, v in config_kwargs.items()])
    save(name, **arrays)


def gen_carried_second_iterations() -> None:
    from examples.cartpole import env as cp_env

    # states re-initialised every 16 sequences of 4 steps = 64 steps = with the environment (data.py:296-309 wants
    # seq_len * seqs_per_state_reset to divide horizon * horizons_per_env_reset): all of iteration 1 runs on the
    # states, observations and returns carried over from iteration 0
    gen_two_iterations("second_iteration_rec_carry.npz", DiscreteDummyEnv,
                       dict(seq_len=4, seqs_per_state_reset=16, horizons_per_env_reset=2), recurrent=True)
    cp_env.step = _eager(cp_env.step)
    gen_two_iterations("second_iteration_ff_cartpole.npz", cp_env.CartPole, dict(horizons_per_env_reset=2))


# --------------------------------------------------------------------------- #
# F8: config 3 end to end: CartPole (examples/cartpole/env.py:12-64,101-150)
#     through Algorithm.collect() / .step() (src/rl8/algorithms/_feedforward.py:301-615): the only
#     built-in path through a three-way Categorical head on a five-wide observation. Self-contained
#     fixture: initial weights, reset STATE (the reference's reset returns observations only), the
#     multinomial noise per timestep, buffer + CollectStats, every StatTracker.update of the default
#     4-iteration step, and a one-iteration run with its first clipped gradient and weights.
#     The eager body of the @torch.compile'd step is used (<= 1 ulp from the compiled one).
# --------------------------------------------------------------------------- #
def gen_cartpole_first_update() -> None:
    from examples.cartpole import env as cp_env

    cp_env.step = _eager(cp_env.step)
    gen_env_first_update("first_update_ff_cartpole.npz", cp_env.CartPole)


def walk_env(d: int, a: int):
    """The tests' own small environment (tests/_envs.py: same arithmetic against this build's ``Env``) written
    against the REFERENCE's ``Env`` (src/rl8/env.py:16-128): a point in ``d`` dimensions, discrete action ``k`` of
    ``a`` pushes coordinate ``k % d``; every operation is one fp32 rounding per element."""
    from rl8.env import Env
    from torchrl.data import Categorical as CategoricalSpec
    from torchrl.data import Unbounded

    class Walk(Env):
        def __init__(self, num_envs, /, horizon=None, *, device="cpu"):
            super().__init__(num_envs, horizon, device=device)
            self.observation_spec = Unbounded(d, device=device)
            self.action_spec = CategoricalSpec(a, shape=torch.Size([1]), device=device)

        def reset(self, *, config=None):
            self.state = torch.empty(self.num_envs, d, device=self.device).uniform_(-1.0, 1.0)
            return self.state

        def step(self, action):
            push = torch.zeros_like(self.state)
            push.scatter_(1, action.reshape(-1, 1) % d, 1.0)
            self.state = 0.75 * self.state + push - 0.25
            rewards = -self.state.abs().sum(-1, keepdim=True)
            return TensorDict({DataKeys.OBS: self.state, DataKeys.REWARDS: rewards}, batch_size=self.num_envs,
                              device=self.device)

    return Walk


def gen_walk_first_update() -> None:
    """F10 (round 5): the reference's DefaultDiscreteModel (src/rl8/models/_feedforward.py:313-383)
    on a 4-wide observation with a 4-way head -- widths none of the built-in environments has -- through collect() /
    step(): rollout, the traced 4-iteration step's updates, and a one-iteration run's first gradient and weights."""
    gen_env_first_update("first_update_ff_walk4.npz", walk_env(4, 4))


def gen_env_first_update(fixture: str, env_cls) -> None:
    arrays = {}

    def run(tag, **overrides):
        torch.manual_seed(42)
        algo = AlgorithmConfig(num_envs=64, horizon=32, device="cpu", **overrides).build(env_cls)
        init = {k: v.clone() for k, v in algo.policy.model.state_dict().items()}
        states = []
        with Recorder() as rec:
            real_reset = algo.env.reset

            def reset(*, config=None):
                out = real_reset(config=config)
                states.append(algo.env.state.clone())
                return out

            algo.env.reset = reset
            collect_stats = algo.collect()
            algo.env.reset = real_reset
            buffer = {}
            snapshot_buffer(algo.buffer, "it0_collect", buffer)
            reward_scale = np.float64(algo.state.reward_scale)
            with UpdateRecorder(algo) as urec:
                step_stats = algo.step()
        return algo, init, states, rec, collect_stats, buffer, reward_scale, urec, step_stats

    algo, init, states, rec, collect_stats, buffer, reward_scale, urec, step_stats = run("traced")
    for k, v in init.items():
        arrays[f"init_{k}"] = v
    arrays["it0_reset_state"] = states[0]                      # [4, N]: x, x_dot, theta, theta_dot
    arrays["it0_cat_q"] = torch.stack(rec.cat_q)
    arrays["it0_perms"] = torch.stack(rec.perms)
    arrays.update(buffer)
    arrays["it0_reward_scale"] = reward_scale
    collect_keys = sorted(k for k in collect_stats if not k.startswith("profiling"))
    step_keys = sorted(k for k in step_stats if not k.startswith("profiling"))
    arrays["it0_collect_stats"] = np.array([collect_stats[k] for k in collect_keys], np.float64)
    arrays["it0_step_stats"] = np.array([step_stats[k] for k in step_keys], np.float64)
    arrays["collect_stat_keys"] = np.array(collect_keys)
    arrays["step_stat_keys"] = np.array(step_keys)
    arrays["traced_updates"] = np.array(urec.updates, np.float64)

    algo1, init1, states1, rec1, _, buffer1, _, urec1, step_stats1 = run("sgd1", num_sgd_iters=1)
    for k, v in init1.items():
        assert np.array_equal(v.numpy(), init[k].numpy()), k
    for k, v in buffer1.items():
        assert np.array_equal(v.numpy(), buffer[k].numpy()), f"the two rollouts differ: {k}"
    assert len(urec1.updates) == 1
    arrays["sgd1_updates"] = np.array(urec1.updates, np.float64)
    arrays["sgd1_step_stats"] = np.array([step_stats1[k] for k in step_keys], np.float64)
    total_sq = 0.0
    for k, gval in urec1.first_grads.items():
        arrays[f"sgd1_grad_{k}"] = gval
        total_sq += float((gval.double() ** 2).sum())
    arrays["sgd1_clipped_grad_norm"] = np.float64(total_sq ** 0.5)
    for k, v in algo1.policy.model.state_dict().items():
        arrays[f"sgd1_final_{k}"] = v.clone()
    arrays["stat_keys"] = np.array(STAT_KEYS + ("reduce",))
    save(fixture, **arrays)


def gen_early_stop() -> None:
    """KL early stop (src/rl8/algorithms/_feedforward.py:577-582) on the rollout of
    trace_ff_discrete.npz with four minibatches per SGD iteration. The reference
    rejects ``target_kl_div`` together with ``accumulate_grads``
    (src/rl8/data.py:227-231), so a stop always finds ``.grad`` freshly cleared."""
    trace = dict(np.load(os.path.join(HERE, "trace_ff_discrete.npz")))
    try:
        AlgorithmConfig(num_envs=64, horizon=32, device="cpu", accumulate_grads=True, sgd_minibatch_size=512,
                        target_kl_div=0.1).build(DiscreteDummyEnv)
        raise AssertionError("reference accepted target_kl_div with accumulate_grads")
    except ValueError as e:
        assert "not compatible with gradient" in str(e)

    def run(**kw):
        torch.manual_seed(42)
        algo = AlgorithmConfig(num_envs=64, horizon=32, device="cpu", sgd_minibatch_size=512, **kw).build(
            DiscreteDummyEnv)
        with Recorder() as noise:
            algo.collect()
            for k, v in algo.buffer.items():
                assert np.array_equal(v.numpy(), trace[f"it0_collect_{k}"]), k
            with UpdateRecorder(algo) as rec:
                stats = algo.step()
        return algo, rec, stats, noise

    _, rec, _, _ = run()
    kls = np.array(rec.updates)[:, STAT_KEYS.index("monitors/kl_div")]
    # update 0 sees the rollout's own weights (kl == 0); one Adam step later the kl is
    # far above any later one: stop there, with a target well clear of both
    j, target = 1, 0.1
    assert kls[0] == 0.0 and kls[1] > 3 * 1.5 * target, kls
    algo, rec, stats, noise = run(target_kl_div=target)
    assert len(rec.updates) == j + 1, len(rec.updates)
    arrays = {
        "target_kl_div": np.float64(target),
        "stopped_at_update": np.int64(j),
        "kl_without_stop": kls,
        "updates": np.array(rec.updates, np.float64),
        "stat_keys": np.array(STAT_KEYS + ("reduce",)),
        "perms": torch.stack(noise.perms),
        "step_stat_keys": np.array(sorted(k for k in stats if not k.startswith("profiling"))),
    }
    arrays["step_stats"] = np.array([stats[k] for k in arrays["step_stat_keys"]], np.float64)
    arrays["grads_are_none_after_stop"] = np.array(all(p_.grad is None for p_ in algo.policy.model.parameters()))
    for k, p_ in algo.policy.model.named_parameters():
        arrays[f"final_{k}"] = p_.detach().clone()
    save("early_stop.npz", **arrays)


def main() -> None:
    if len(sys.argv) > 1 and sys.argv[1] == "first_updates":
        gen_first_updates()
        gen_early_stop()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "second_iterations":
        gen_second_iterations()
        gen_carried_second_iterations()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "carried":
        gen_carried_second_iterations()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cartpole":
        gen_cartpole_first_update()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "walk":
        gen_walk_first_update()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "views":
        gen_views()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "classic":
        gen_classic_env_steps()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "recurrent":
        gen_trace("trace_rec_discrete.npz", DiscreteDummyEnv, {}, recurrent=True)
        gen_trace(
            "trace_rec_continuous_minibatch.npz",
            ContinuousDummyEnv,
            dict(sgd_minibatch_size=128, entropy_coeff=1e-2, seq_len=8, seqs_per_state_reset=2,
                 horizons_per_env_reset=2),
            recurrent=True,
        )
        return
    gen_gae()
    gen_ppo_losses()
    gen_env_steps()
    gen_classic_env_steps()
    gen_views()
    gen_samplers()
    gen_trace("trace_ff_discrete.npz", DiscreteDummyEnv, {})
    gen_trace(
        "trace_ff_discrete_minibatch.npz",
        DiscreteDummyEnv,
        dict(sgd_minibatch_size=256, entropy_coeff=1e-2, dual_clip_param=5.0, horizons_per_env_reset=2),
    )
    gen_trace(
        "trace_ff_continuous_squashed.npz",
        ContinuousDummyEnv,
        dict(distribution_cls=SquashedNormal),
    )
    gen_trace("trace_ff_continuous_normal.npz", ContinuousDummyEnv, dict(entropy_coeff=1e-2))
    gen_trace("trace_rec_discrete.npz", DiscreteDummyEnv, {}, recurrent=True)
    gen_trace(
        "trace_rec_continuous_minibatch.npz",
        ContinuousDummyEnv,
        dict(sgd_minibatch_size=128, entropy_coeff=1e-2, seq_len=8, seqs_per_state_reset=2, horizons_per_env_reset=2),
        recurrent=True,
    )
    gen_first_updates()
    gen_early_stop()
    gen_second_iterations()
    gen_carried_second_iterations()
    gen_cartpole_first_update()
    gen_walk_first_update()


if __name__ == "__main__":
    main()
