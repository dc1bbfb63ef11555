"""Recurrent PPO: ``RecurrentAlgorithmConfig`` / ``RecurrentAlgorithm``.

API follows the reference's ``src/rl8/algorithms/_recurrent.py``
(``RecurrentAlgorithmConfig`` :29-192, ``collect`` :325-479, ``step`` :481-652):
truncated BPTT over ``seq_len`` steps, recurrent states carried in the rollout
buffer, re-initialised every ``seqs_per_state_reset`` sequences, sequences (not
samples) shuffled into minibatches.

Built on :class:`Algorithm`: same time-major buffer (the ``states`` leaves are
``[H+1][N][layers][hidden]`` slabs, so the per-timestep state carry is a dense
copy, not the reference's strided one), same fused per-timestep launch, same GAE
and fused loss kernels. A minibatch of sequences is assembled by
``rl8_gather_minibatch``: per-sample leaves through the narrow path, the initial
states of each sequence (1 KiB rows) through the wave-per-row path. As in the
feed-forward algorithm, a minibatch that is the whole buffer (the default) is not
shuffled -- its mean does not depend on the order of the sequences: they are laid
sequence-major once per ``step()`` in buffer order and every SGD iteration reads
that copy (the reference draws ``randperm`` and gathers per iteration,
``src/rl8/_utils.py:211-225``; ``bench.py --recurrent --minibatches 4`` times the
shuffled path). A training pass through the default models is ONE autograd node
(``nn/fused_lstm.py:_FusedLSTMHeads``): the heads' data gradient is formed inside
the backward-through-time kernel.

"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any

import torch
import torch.amp as amp

from .. import hip
from .._utils import profile_ms
from ..data import (
    CollectStats,
    DataKeys,
    RecurrentAlgorithmHparams,
    RecurrentAlgorithmState,
)
from ..distributions import Categorical
from ..env import EnvFactory
from ..models_recurrent import RecurrentModel, RecurrentModelFactory
from ..nn.functional import fused_ppo_loss, has_fused_loss
from ..policies_recurrent import RecurrentPolicy
from ..specs import Composite
from ..tensordict import TensorDict
from ._feedforward import Algorithm, AlgorithmConfig, _collect_stats_from_raw

#: Rows (sequences x seq_len) pushed through the LSTM per forward/backward pass.
#: Rows (sequences x steps) of one forward / backward pass through the LSTM;
#: larger minibatches accumulate over several passes. 2^21 rows of the default
#: 256-wide LSTM keep the pass's activations around 20 GB.
RECURRENT_MAX_ROWS_PER_PASS = 1 << 21


class _LeanRollout:
    """The per-timestep launches of ``collect()`` for the default discrete recurrent
    model on a dummy env, issued straight through the C ABI with addresses computed
    once: LSTM step (new states written into the buffer's next column -- no copies),
    the logits head, the value head, and the fused sampler + ``env.step`` +
    bookkeeping kernel. Same kernels, same arguments, same order as
    ``policy.sample()`` + ``_fused_step`` (tests/test_algorithm_gpu.py:
    ``test_recurrent_lean_and_plumbed_rollouts_agree``); what goes is ~150 us of
    tensordict / autograd-function / ``torch.cat`` host work per timestep, which at
    8192 environments per GPU (BASELINE configs[4]) was twice the kernels' time."""

    TIMER_EVERY = 16  # with hip.timer enabled, only every 16th timestep is bracketed by events

    def __init__(self, algo: "RecurrentAlgorithm", deterministic: bool) -> None:
        from ..nn import fused_lstm

        self.lib = hip.load()
        model = algo.policy.model
        tm, stm = algo._tm, algo._tm_states
        self.n = n = algo.local_num_envs
        self.d_in = int(tm[DataKeys.OBS].shape[-1])
        self.k = int(model.feature_head.out_features)
        dev = tm[DataKeys.OBS].device
        self.split = fused_lstm.use_split(model.lstm)
        if self.split:  # bf16-plane step kernel: W_hh planes, [w_ih | bias] rows, planes of h_{t-1}
            self.packed, self.wb = fused_lstm._packs(model.lstm, "split")
        else:
            self.packed, self.wb = fused_lstm._packs(model.lstm, False), None
        # parameters are leaf tensors: .detach() shares storage (kept alive on self)
        self.params = [p.detach().contiguous() for p in (model.feature_head.weight, model.feature_head.bias,
                                                         model.vf_head.weight, model.vf_head.bias)]
        cache = algo.__dict__.setdefault("_lean_scratch", {})
        key = (n, self.k, str(dev))
        if cache.get("key") != key:
            cache.update(key=key, hs=torch.empty(n, hip.LSTM_HIDDEN, device=dev),
                         logits=torch.empty(n, self.k, device=dev), value=torch.empty(n, 1, device=dev),
                         planes=hip.lstm_state_planes(n, dev, copies=2))
        self.hs, self.logits, self.value, self.planes = cache["hs"], cache["logits"], cache["value"], cache["planes"]
        rdr = tm.get(DataKeys.REVERSED_DISCOUNTED_RETURNS)

        def column(t: torch.Tensor) -> tuple[int, int]:
            return t.data_ptr(), t.stride(0) * t.element_size()

        self.obs, self.act, self.logp = column(tm[DataKeys.OBS]), column(tm[DataKeys.ACTIONS]), column(tm[DataKeys.LOGP])
        self.val, self.rew = column(tm[DataKeys.VALUES]), column(tm[DataKeys.REWARDS])
        self.rdr = column(rdr) if rdr is not None else None
        self.h, self.c = column(stm[DataKeys.HIDDEN_STATES]), column(stm[DataKeys.CELL_STATES])
        self.state_ptr = algo.env.state.data_ptr()
        self.gamma = float(torch.tensor(algo.hparams.gamma, dtype=torch.float32))
        self.seed, self.env_offset = algo.noise.seed, algo.env.env_offset
        self.deterministic = int(deterministic)
        self.ptrs = [t.data_ptr() for t in (self.packed, self.hs, self.logits, self.value, *self.params)]
        self.split_ptrs = (self.planes.data_ptr(), self.wb.data_ptr() if self.wb is not None else None)
        self.planes_half = self.planes.numel() // 2
        self.planes_of = -1   # timestep whose hidden state the planes buffer (t & 1) holds (-1: none)
        # two-way categorical + value head: evaluated inside the timestep's last kernel (RL8_AMD_ROLLOUT_FUSE_HEADS=0: two launches)
        self.fuse_heads = self.k == 2 and os.environ.get("RL8_AMD_ROLLOUT_FUSE_HEADS", "1") != "0"

    @staticmethod
    def available(algo: "RecurrentAlgorithm") -> bool:
        from ..env import DummyEnv
        from ..models_recurrent import DefaultDiscreteRecurrentModel
        from ..nn import fused_lstm

        model = algo.policy.model
        if type(model) is not DefaultDiscreteRecurrentModel or not isinstance(algo.env, DummyEnv):
            return False
        if algo.policy.distribution_cls is not Categorical or model.action_spec.shape[0] != 1:
            return False
        lstm = model.lstm
        obs = algo._tm[DataKeys.OBS]
        return (fused_lstm.ENABLED and lstm.num_layers == 1 and lstm.hidden_size == hip.LSTM_HIDDEN and lstm.bias
                and lstm.proj_size == 0 and not lstm.bidirectional and hip.lstm_supports(lstm.input_size)
                and obs.dtype == torch.float32 and model.vf_head.bias is not None
                and all(p.dtype == torch.float32 for p in model.parameters()))

    def step(self, t: int, noise: None | torch.Tensor, step_id: int) -> None:
        lib, n, stream = self.lib, self.n, hip._stream()
        packed, hs, logits, value, w_pol, b_pol, w_vf, b_vf = self.ptrs
        at = lambda col, i: col[0] + i * col[1]  # noqa: E731
        timed = hip.timer.enabled and t % self.TIMER_EVERY == 0
        if self.split:
            planes, wb = self.split_ptrs
            H = hip.LSTM_HIDDEN
            p_in, p_out = planes + (t & 1) * self.planes_half, planes + ((t + 1) & 1) * self.planes_half
            if self.planes_of != t:
                # the first timestep, or one whose states were just re-initialised: planes of h_t
                # from the buffer; otherwise the previous step's kernel has left them
                hip._check(lib.rl8_lstm_split_state(at(self.h, t), H, n, p_in, stream), "rl8_lstm_split_state")
            with hip._timed("lstm_step", n) if timed else _NO_TIMER:
                hip._check(lib.rl8_lstm_step_split_f32(at(self.obs, t), self.d_in, self.d_in, p_in, at(self.c, t), H,
                                                       packed, wb, n, at(self.h, t + 1), H, at(self.c, t + 1), H, None,
                                                       0, p_out, stream), "rl8_lstm_step_split_f32")
            self.planes_of = t + 1
            hs = at(self.h, t + 1)  # the heads read h_t where the buffer keeps it
        else:
            with hip._timed("lstm_forward", n) if timed else _NO_TIMER:
                hip._check(lib.rl8_lstm_forward_f32(at(self.obs, t), n, 1, self.d_in, at(self.h, t), at(self.c, t), packed,
                                                    hs, at(self.h, t + 1), at(self.c, t + 1), None, None, stream),
                           "rl8_lstm_forward_f32")
        if self.fuse_heads:
            # the two heads inside the sampler + env.step + bookkeeping kernel: one launch instead of two
            with hip._timed("rollout_step_dummy", n) if timed else _NO_TIMER:
                hip._check(lib.rl8_rollout_step_dummy_heads_f32(
                    hs, w_pol, b_pol, w_vf, b_vf, noise.data_ptr() if noise is not None else None, self.state_ptr,
                    at(self.act, t), at(self.logp, t), at(self.val, t), at(self.rew, t), at(self.obs, t + 1),
                    at(self.rdr, t) if self.rdr else None, at(self.rdr, t + 1) if self.rdr else None, self.gamma, n,
                    self.seed, step_id, self.env_offset, self.deterministic, stream), "rl8_rollout_step_dummy_heads_f32")
            return
        with hip._timed("linear_heads_forward", n) if timed else _NO_TIMER:
            if self.k + 1 <= hip.HEADS_MAX_OUT:  # both heads in one pass over h_t
                hip._check(lib.rl8_linear_heads_forward_pair_f32(hs, n, w_pol, b_pol, self.k, logits, w_vf, b_vf, 1, value,
                                                                 stream), "rl8_linear_heads_forward_pair_f32")
            else:
                hip._check(lib.rl8_linear_heads_forward_f32(hs, n, w_pol, b_pol, self.k, logits, stream),
                           "rl8_linear_heads_forward_f32")
                hip._check(lib.rl8_linear_heads_forward_f32(hs, n, w_vf, b_vf, 1, value, stream),
                           "rl8_linear_heads_forward_f32")
        with hip._timed("rollout_step_dummy", n) if timed else _NO_TIMER:
            hip._check(lib.rl8_rollout_step_dummy_f32(
                1, 0, logits, None, value, noise.data_ptr() if noise is not None else None, self.state_ptr,
                at(self.act, t), at(self.logp, t), at(self.val, t), at(self.rew, t), at(self.obs, t + 1),
                at(self.rdr, t) if self.rdr else None, at(self.rdr, t + 1) if self.rdr else None, self.gamma, n,
                self.seed, step_id, self.env_offset, self.deterministic, stream), "rl8_rollout_step_dummy_f32")


class _NoTimer:
    def __enter__(self) -> None:
        return None

    def __exit__(self, *exc: Any) -> None:
        return None


_NO_TIMER = _NoTimer()


@dataclass
class RecurrentAlgorithmConfig(AlgorithmConfig):
    """Configuration of a recurrent PPO algorithm: :class:`AlgorithmConfig` plus
    the truncated-BPTT settings."""

    model: None | RecurrentModel = None  # type: ignore[assignment]
    model_cls: None | RecurrentModelFactory = None  # type: ignore[assignment]
    #: Truncated backpropagation-through-time sequence length (a factor of
    #: ``horizon``).
    seq_len: int = 4
    #: Sequences collected between re-initialisations of the recurrent states
    #: (negative: never after the first).
    seqs_per_state_reset: int = 8

    def build(self, env_cls: EnvFactory) -> "RecurrentAlgorithm":  # type: ignore[override]
        algo = RecurrentAlgorithm(env_cls, config=self)
        algo.validate()
        return algo


class RecurrentAlgorithm(Algorithm):
    """Recurrent PPO (LSTM policies) with the usual stabilising tricks."""

    hparams: RecurrentAlgorithmHparams
    policy: RecurrentPolicy  # type: ignore[assignment]
    state: RecurrentAlgorithmState  # type: ignore[assignment]
    #: Issue the rollout's per-timestep launches through :class:`_LeanRollout` when the
    #: model / env pair allows it (False: always through ``policy.sample()``).
    lean_rollout: bool = True

    def __init__(self, env_cls: EnvFactory, /, config: None | RecurrentAlgorithmConfig = None) -> None:
        config = config or RecurrentAlgorithmConfig()
        self._recurrent_config = config
        super().__init__(env_cls, config=config)

    # -- construction hooks ---------------------------------------------------
    def _make_policy(self, config: Any, device: str) -> RecurrentPolicy:
        return RecurrentPolicy(
            self.env.observation_spec,
            self.env.action_spec,
            model=config.model,
            model_cls=config.model_cls,
            model_config=config.model_config,
            distribution_cls=config.distribution_cls,
            device=device,
        )

    def _extra_buffer_specs(self) -> dict[str, Any]:
        return {DataKeys.STATES: self.policy.state_spec}

    def _default_minibatch_size(self, config: Any, num_envs: int, horizon: int) -> int:
        return num_envs * (horizon // config.seq_len)

    def _make_hparams(self, **common: Any) -> RecurrentAlgorithmHparams:
        return RecurrentAlgorithmHparams(
            seq_len=self._recurrent_config.seq_len,
            seqs_per_state_reset=self._recurrent_config.seqs_per_state_reset,
            **common,
        )

    def _make_state(self) -> RecurrentAlgorithmState:
        return RecurrentAlgorithmState()

    # -- helpers ----------------------------------------------------------------
    def _state_slabs(self, t: int) -> TensorDict:
        """States at column ``t`` as a ``[N, 1, layers, hidden]`` tensordict."""
        n = self.local_num_envs
        return TensorDict(
            {k: v[t].unsqueeze(1) for k, v in self._tm_states.items()}, batch_size=[n, 1]
        )

    def _samples_per_minibatch(self) -> int:
        return self.hparams.sgd_minibatch_size * self.hparams.seq_len

    # -- collect ------------------------------------------------------------------
    def collect(
        self,
        *,
        env_config: None | dict[str, Any] = None,
        deterministic: bool = False,
    ) -> CollectStats:
        """As :meth:`Algorithm.collect`, additionally carrying the recurrent
        states from column to column and re-initialising them at sequence
        boundaries according to ``seqs_per_state_reset``."""
        hp = self.hparams
        H, N = hp.horizon, self.local_num_envs
        tm, stm = self._tm, self._tm_states
        rdr = tm.get(DataKeys.REVERSED_DISCOUNTED_RETURNS)
        self._flat_full = None  # (a sequence-major copy of the buffer this collect() overwrites)
        self._release_step_caches()
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
            for v in stm.values():  # :380-382
                v[0].copy_(v[H])

            fused = self._fusable()
            lean = None
            if fused and self.lean_rollout and _LeanRollout.available(self):
                lean = _LeanRollout(self, deterministic)
            gamma = float(torch.tensor(hp.gamma, dtype=torch.float32))
            for t in range(H):
                if self.state.seqs and hp.seqs_per_state_reset < 0:
                    pass
                elif not (t % hp.seq_len) and not (self.state.seqs % hp.seqs_per_state_reset):
                    init = self.policy.init_states(N)  # :385-392
                    for k, v in stm.items():
                        v[t].copy_(init[k])
                    if lean is not None:
                        lean.planes_of = -1  # h_t was replaced: its planes must be re-made
                noise_t = self.injected_noise[t] if self.injected_noise is not None else None
                step_id = self.noise.next_step()
                if lean is not None:
                    lean.step(t, noise_t.contiguous() if noise_t is not None else None, step_id)
                    if not ((t + 1) % hp.seq_len):
                        self.state.seqs += 1
                    continue
                in_batch = TensorDict({DataKeys.OBS: tm[DataKeys.OBS][t].unsqueeze(1)}, batch_size=[N, 1])
                if fused:
                    sample, new_states = self.policy.sample(
                        in_batch, self._state_slabs(t), deterministic=deterministic, inplace=False,
                        requires_grad=False, return_actions=False, return_logp=False, return_values=True,
                    )
                    self._fused_step(sample[DataKeys.FEATURES], sample[DataKeys.VALUES], noise_t, t, gamma,
                                     step_id, deterministic)
                else:
                    self.policy.injected_noise = noise_t
                    self.noise.step = step_id
                    sample, new_states = self.policy.sample(
                        in_batch, self._state_slabs(t), deterministic=deterministic, inplace=False,
                        requires_grad=False, return_actions=True, return_logp=True, return_values=True,
                    )
                    out = self.env.step(sample[DataKeys.ACTIONS])
                    hip.rollout_scatter(
                        sample[DataKeys.ACTIONS].contiguous(), sample[DataKeys.LOGP].contiguous(),
                        sample[DataKeys.VALUES].contiguous(), out[DataKeys.REWARDS].contiguous(),
                        out[DataKeys.OBS].contiguous(), tm[DataKeys.ACTIONS][t], tm[DataKeys.LOGP][t],
                        tm[DataKeys.VALUES][t], tm[DataKeys.REWARDS][t], tm[DataKeys.OBS][t + 1],
                        rdr[t] if rdr is not None else None, rdr[t + 1] if rdr is not None else None, gamma,
                    )
                for k, v in stm.items():  # :428
                    v[t + 1].copy_(new_states[k])
                if not ((t + 1) % hp.seq_len):
                    self.state.seqs += 1

            # Bootstrap value at the last observation and state (:433-446).
            in_batch = TensorDict({DataKeys.OBS: tm[DataKeys.OBS][H].unsqueeze(1)}, batch_size=[N, 1])
            sample, _ = self.policy.sample(
                in_batch, self._state_slabs(H), deterministic=deterministic, inplace=False,
                requires_grad=False, return_actions=False, return_logp=False, return_values=True,
            )
            tm[DataKeys.VALUES][H].copy_(sample[DataKeys.VALUES])

            # Stats.  The reference's recurrent variant slices rewards [:, 1:-1]
            # (:449), unlike the feed-forward [:, :-1]; reward_scale still uses
            # rdr[:, 1:].  Two passes of the stats kernel reproduce both.
            rewards = self.buffer[DataKeys.REWARDS]
            if H >= 2:
                raw = hip.rollout_stats(rewards[:, 1:], None)
            else:
                raw = torch.full((12,), float("nan"), dtype=torch.float64, device=rewards.device)
            if rdr is not None:
                raw_rdr = hip.rollout_stats(rewards, self.buffer[DataKeys.REVERSED_DISCOUNTED_RETURNS])
                raw = torch.cat([raw[:10], raw_rdr[10:]])
                rdr_count = float(N * H)
            values = self.shards.combine_rollout_stats(raw)
            collect_stats, _ = _collect_stats_from_raw(values)
            if rdr is not None:
                world = self.shards.world_size
                _, reward_scale = _collect_stats_from_raw(
                    values[:5] + [rdr_count * world] + values[6:]
                )
            else:
                reward_scale = 1.0
            self.state.horizons += 1
            self.state.buffered = True
            self.state.reward_scale = reward_scale if hp.normalize_rewards else 1.0
            self.injected_noise = None

        collect_stats["env/resets"] = hp.num_envs * int(env_was_reset)
        collect_stats["env/steps"] = hp.num_envs * hp.horizon
        collect_stats["profiling/collect_ms"] = collect_timer()
        return collect_stats

    # -- step ---------------------------------------------------------------------
    def _release_step_caches(self) -> None:
        from ..nn import fused_lstm

        super()._release_step_caches()

        fused_lstm.clear_state_cache()  # the planes of the initial hidden states, shared by the SGD iterations

    def _iter_minibatches(self, sgd_iter: int):
        """Minibatches of SEQUENCES: per-sample leaves as ``[B*L, ...]`` in
        (sequence, step) order, observations as ``[B, L, ...]``, and each
        sequence's initial recurrent states ``[B, layers, hidden]``."""
        hp = self.hparams
        H, L = hp.horizon, hp.seq_len
        local_seqs = self.local_num_envs * (H // L)
        local_mb = hp.sgd_minibatch_size // self.shards.world_size
        state_keys = list(self._tm_states)
        whole = local_mb >= local_seqs and self.injected_permutations is None
        if whole:
            from ..nn import fused_lstm

            fused_lstm.SHARE_H0_PLANES = True  # until _release_step_caches(): every pass reads the same initial states
        if whole and getattr(self, "_flat_full", None) is not None:
            yield self._flat_full
            return
        # One minibatch that is the whole buffer: its mean does not depend on the order of the sequences (the
        # feed-forward algorithm's full-buffer rule), so they are laid sequence-major ONCE per step() in buffer order
        # and every SGD iteration reads that copy -- no permutation, one gather instead of num_sgd_iters.
        perm = (torch.arange(local_seqs, device=self._tm[DataKeys.LOGP].device) if whole
                else self._permutation(sgd_iter, local_seqs))
        steps = torch.arange(L, device=perm.device)
        for seq_index in torch.split(perm, local_mb):
            # reference sequence id q = env * (H/L) + s  ->  sample ids q*L + j
            sample_ids = (seq_index[:, None] * L + steps[None, :]).reshape(-1).contiguous()
            # (the whole buffer in order: a transposition through LDS tiles, every byte moved once -- the indexed
            # gather reads a 64-byte sector per 4-byte cell of a time-major leaf)
            gathered = hip.gather_minibatch(None if whole else sample_ids, H, [self.buffer[k] for k in self.TRAIN_KEYS])
            batch = dict(zip(self.TRAIN_KEYS, gathered))
            first_ids = (seq_index * L).contiguous()
            states = hip.gather_minibatch(
                first_ids, H, [self.buffer[DataKeys.STATES][k] for k in state_keys]
            )
            batch["_states"] = dict(zip(state_keys, states))
            batch["_num_seqs"] = seq_index.numel()
            if whole:
                self._flat_full = batch  # (dropped at the end of step(), with the buffer it was read from)
            yield batch

    def _minibatch_forward_backward(
        self, batch: dict[str, Any], entropy_coeff: float, grad_scale: float
    ) -> torch.Tensor:
        hp = self.hparams
        L = hp.seq_len
        num_seqs = batch["_num_seqs"]
        dist_cls = self.policy.distribution_cls
        # A user's Distribution: the loss is composed from its own logp / entropy with device tensor ops, as the
        # reference does with whatever class it was given (src/rl8/algorithms/_recurrent.py:560-585) -- the
        # feed-forward algorithm's rule (Algorithm._composed_loss_backward), here per pass of sequences.
        fused = has_fused_loss(dist_cls)
        seqs_per_pass = max(1, min(self.max_rows_per_pass, RECURRENT_MAX_ROWS_PER_PASS) // L)
        total_sums: None | torch.Tensor = None
        scale = None
        if hp.enable_amp:
            scale = self.grad_scaler.scale(torch.ones((), device=batch[DataKeys.LOGP].device))
        for start in range(0, num_seqs, seqs_per_pass):
            stop = min(num_seqs, start + seqs_per_pass)
            b = stop - start
            rows = slice(start * L, stop * L)
            obs = batch[DataKeys.OBS][rows].reshape(b, L, *batch[DataKeys.OBS].shape[1:])
            states = TensorDict(
                {k: v[start:stop].unsqueeze(1) for k, v in batch["_states"].items()}, batch_size=[b, 1]
            )
            with amp.autocast("cuda", enabled=hp.enable_amp):
                sample, _ = self.policy.sample(
                    TensorDict({DataKeys.OBS: obs}, batch_size=[b, L]), states, deterministic=False,
                    inplace=False, requires_grad=True, return_actions=False, return_logp=False,
                    return_values=True,
                )
            if fused:
                sums, inputs, grads = fused_ppo_loss(
                    dist_cls, sample[DataKeys.FEATURES], sample[DataKeys.VALUES], batch[DataKeys.ACTIONS][rows],
                    batch[DataKeys.LOGP][rows], batch[DataKeys.ADVANTAGES][rows], batch[DataKeys.RETURNS][rows],
                    clip_param=hp.clip_param, dual_clip_param=hp.dual_clip_param, entropy_coeff=entropy_coeff,
                    vf_clip_param=hp.vf_clip_param, vf_coeff=hp.vf_coeff, grad_scale=grad_scale,
                )
                if scale is not None:
                    grads = [g * scale.to(g.dtype) for g in grads]
                torch.autograd.backward(inputs, grads)
            else:
                chunk = {k: batch[k][rows] for k in (DataKeys.ACTIONS, DataKeys.LOGP, DataKeys.ADVANTAGES,
                                                     DataKeys.RETURNS)}
                sums = self._composed_loss_backward(sample[DataKeys.FEATURES], sample[DataKeys.VALUES], chunk,
                                                    entropy_coeff, grad_scale, scale)
            total_sums = sums if total_sums is None else total_sums + sums
        assert total_sums is not None
        return total_sums

    # -- validate -------------------------------------------------------------------
    def validate(self) -> None:
        """Shape checks on one reset / sample / step round trip (:654-756)."""
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
        states = self.policy.init_states(n)
        try:
            self.buffer[DataKeys.STATES][:, 0, ...] = states
        except RuntimeError as e:
            raise AssertionError(
                "The recurrent states from the policy don't match the state spec."
            ) from e
        sample_batch, out_states = self.policy.sample(
            self.buffer[:, :1, ...].select(DataKeys.OBS),
            self.buffer[DataKeys.STATES][:, :1, ...],
            deterministic=False, inplace=False, requires_grad=False, return_actions=True,
            return_logp=True, return_values=True,
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
            self.buffer[DataKeys.STATES][:, 1, ...] = out_states
        except RuntimeError as e:
            raise AssertionError(
                "The action or states sampled from the policy don't match their specs."
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
        self.env.observation_spec.assert_is_in(out_batch[DataKeys.OBS])
        assert out_batch[DataKeys.REWARDS].shape == torch.Size([n, 1]), (
            "Rewards must be 2D and have shape ``[N, 1]`` (where ``N`` is the number of"
            " independent elements or environment instances)."
        )
        # validate() must not leave anything behind in the buffer
        self._reset_buffer()
        self._tm[DataKeys.OBS][self.hparams.horizon].zero_()
        for v in self._tm_states.values():
            v.zero_()


__all__ = ["RecurrentAlgorithm", "RecurrentAlgorithmConfig", "Composite"]
