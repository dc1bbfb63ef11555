"""Small environments defined by the TESTS (not the product): widths the built-in environments lack.

``walk_env(d, a)``: a point in ``d`` dimensions; discrete action ``k`` of ``a`` pushes coordinate ``k % d``:
``state <- 0.75 * state + onehot(k % d) - 0.25``, reward ``-sum |state|``.  Every operation is one IEEE fp32
rounding per element, so the CPU reference and a GPU run agree bit for bit on the state.  The SAME arithmetic is
written against the reference's ``Env`` base class in ``tests/golden/generate_fixtures.py`` (``walk_env`` there), which
produced ``first_update_ff_walk4.npz`` with ``d = a = 4``: the default models' run-time widths beside the built-in
environments' (observations of 1 and 5 floats, heads of 2 and 3 logits;
``/root/reference/src/rl8/models/_feedforward.py:313-383`` accepts any).
"""

from __future__ import annotations

from typing import Any

import torch

from rl8_amd.data import DataKeys
from rl8_amd.env import Env
from rl8_amd.specs import Categorical, Unbounded
from rl8_amd.tensordict import TensorDict


def walk_step(state: torch.Tensor, action: torch.Tensor, d: int) -> tuple[torch.Tensor, torch.Tensor]:
    push = torch.zeros_like(state)
    push.scatter_(1, action.reshape(-1, 1) % d, 1.0)
    state = 0.75 * state + push - 0.25
    return state, -state.abs().sum(-1, keepdim=True)


def walk_env(d: int, a: int) -> type[Env]:
    class Walk(Env):
        def __init__(self, num_envs: int, /, horizon: None | int = None, *, device: Any = "cpu") -> None:
            super().__init__(num_envs, horizon, device=device)
            self.observation_spec = Unbounded(d, device=device)
            self.action_spec = Categorical(a, shape=torch.Size([1]), device=device)

        def reset(self, *, config: None | dict[str, Any] = None) -> torch.Tensor:
            self.state = torch.empty(self.num_envs, d, device=self.device).uniform_(-1.0, 1.0)
            return self.state

        def step(self, action: torch.Tensor) -> TensorDict:
            self.state, rewards = walk_step(self.state, action, d)
            return TensorDict({DataKeys.OBS: self.state, DataKeys.REWARDS: rewards}, batch_size=self.num_envs,
                              device=self.device)

    Walk.__name__ = f"Walk{d}x{a}"
    return Walk
