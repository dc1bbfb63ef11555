"""Tensor specs: the slice of ``torchrl.data`` the rl8 PPO path uses.

The reference describes observation / action / buffer layouts with
``torchrl.data.{Unbounded, Categorical, Composite}`` (``src/rl8/env.py:194,222,251``,
``src/rl8/algorithms/_feedforward.py:239-256``). ``torchrl`` cannot be installed
on the MI355X boxes, so the drop-in carries specs of its own that answer the
same questions: ``shape``, ``ndim``, ``dtype``, ``device``, ``space.n``,
``zero(batch)``, ``rand(batch)``, ``to(device)``, ``assert_is_in(value)``.

"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Iterator, Mapping

import torch

from .tensordict import TensorDict


def _as_shape(shape: Any) -> torch.Size:
    if shape is None:
        return torch.Size([])
    if isinstance(shape, int):
        return torch.Size([shape])
    return torch.Size(list(shape))


def _as_batch(batch: Any) -> list[int]:
    if batch is None:
        return []
    if isinstance(batch, int):
        return [batch]
    return list(batch)


class TensorSpec:
    """Common base: a leaf (or tree of leaves) description of tensors."""

    shape: torch.Size
    dtype: torch.dtype
    device: None | torch.device

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def zero(self, shape: Any = None) -> Any:
        raise NotImplementedError

    def rand(self, shape: Any = None) -> Any:
        raise NotImplementedError

    def to(self, device: Any) -> "TensorSpec":
        raise NotImplementedError

    def is_in(self, value: Any) -> bool:
        raise NotImplementedError

    def assert_is_in(self, value: Any) -> None:
        if not self.is_in(value):
            raise AssertionError(f"{value} is not in spec {self}.")

    def encode(self, value: Any) -> Any:
        raise NotImplementedError


class Unbounded(TensorSpec):
    """Real-valued tensor of a fixed trailing shape with no bounds."""

    def __init__(
        self,
        shape: Any = None,
        *,
        device: Any = None,
        dtype: torch.dtype = torch.float32,
    ) -> None:
        self.shape = _as_shape(shape)
        self.device = torch.device(device) if device is not None else None
        self.dtype = dtype

    def __repr__(self) -> str:
        return (
            f"Unbounded(shape={tuple(self.shape)}, dtype={self.dtype},"
            f" device={self.device})"
        )

    def zero(self, shape: Any = None) -> torch.Tensor:
        return torch.zeros(
            *_as_batch(shape), *self.shape, dtype=self.dtype, device=self.device
        )

    def rand(self, shape: Any = None) -> torch.Tensor:
        return torch.randn(
            *_as_batch(shape), *self.shape, dtype=self.dtype, device=self.device
        )

    def to(self, device: Any) -> "Unbounded":
        return Unbounded(self.shape, device=device, dtype=self.dtype)

    def is_in(self, value: Any) -> bool:
        if not torch.is_tensor(value):
            return False
        n = len(self.shape)
        trailing = value.shape[value.ndim - n :] if n else torch.Size([])
        return tuple(trailing) == tuple(self.shape) and value.dtype == self.dtype

    def encode(self, value: Any) -> torch.Tensor:
        return torch.as_tensor(value, dtype=self.dtype, device=self.device)


@dataclass(frozen=True)
class _CategoricalSpace:
    #: Number of classes.
    n: int


class Categorical(TensorSpec):
    """Integer tensor whose elements are class indices in ``[0, n)``."""

    def __init__(
        self,
        n: int,
        shape: Any = None,
        *,
        device: Any = None,
        dtype: torch.dtype = torch.int64,
    ) -> None:
        self.space = _CategoricalSpace(int(n))
        self.shape = _as_shape(shape)
        self.device = torch.device(device) if device is not None else None
        self.dtype = dtype

    @property
    def n(self) -> int:
        return self.space.n

    def __repr__(self) -> str:
        return (
            f"Categorical(n={self.space.n}, shape={tuple(self.shape)},"
            f" dtype={self.dtype}, device={self.device})"
        )

    def zero(self, shape: Any = None) -> torch.Tensor:
        return torch.zeros(
            *_as_batch(shape), *self.shape, dtype=self.dtype, device=self.device
        )

    def rand(self, shape: Any = None) -> torch.Tensor:
        return torch.randint(
            0,
            self.space.n,
            (*_as_batch(shape), *self.shape),
            dtype=self.dtype,
            device=self.device,
        )

    def to(self, device: Any) -> "Categorical":
        return Categorical(self.space.n, self.shape, device=device, dtype=self.dtype)

    def is_in(self, value: Any) -> bool:
        if not torch.is_tensor(value):
            return False
        n = len(self.shape)
        trailing = value.shape[value.ndim - n :] if n else torch.Size([])
        if tuple(trailing) != tuple(self.shape) or value.dtype != self.dtype:
            return False
        if value.numel() == 0:
            return True
        return bool(((value >= 0) & (value < self.space.n)).all())

    def encode(self, value: Any) -> torch.Tensor:
        return torch.as_tensor(value, dtype=self.dtype, device=self.device)


class Composite(TensorSpec):
    """A keyed tree of specs; materialises as a :class:`TensorDict`."""

    def __init__(
        self,
        specs: None | Mapping[str, Any] = None,
        *,
        device: Any = None,
        **kwargs: Any,
    ) -> None:
        self._specs: dict[str, TensorSpec] = {}
        self.device = torch.device(device) if device is not None else None
        self.shape = torch.Size([])
        self.dtype = torch.float32
        for k, v in {**(dict(specs) if specs else {}), **kwargs}.items():
            self.set(k, v)

    def __repr__(self) -> str:
        inner = ", ".join(f"{k}: {v}" for k, v in self._specs.items())
        return f"Composite({inner})"

    def set(self, key: str, spec: Any) -> "Composite":
        if isinstance(spec, Mapping):
            spec = Composite(spec, device=self.device)
        self._specs[key] = spec
        return self

    def __setitem__(self, key: str, spec: Any) -> None:
        self.set(key, spec)

    def __getitem__(self, key: Any) -> TensorSpec:
        if isinstance(key, tuple):
            out: Any = self
            for k in key:
                out = out[k]
            return out
        return self._specs[key]

    def __delitem__(self, key: str) -> None:
        del self._specs[key]

    def __iter__(self) -> Iterator[str]:
        return iter(self._specs)

    def __len__(self) -> int:
        return len(self._specs)

    def __contains__(self, key: str) -> bool:
        return key in self._specs

    def keys(self) -> Any:
        return self._specs.keys()

    def values(self) -> Any:
        return self._specs.values()

    def items(self) -> Any:
        return self._specs.items()

    def zero(self, shape: Any = None) -> TensorDict:
        batch = _as_batch(shape)
        out = TensorDict({}, batch_size=batch, device=self.device)
        for k, v in self._specs.items():
            out[k] = v.zero(batch)
        return out

    def rand(self, shape: Any = None) -> TensorDict:
        batch = _as_batch(shape)
        out = TensorDict({}, batch_size=batch, device=self.device)
        for k, v in self._specs.items():
            out[k] = v.rand(batch)
        return out

    def to(self, device: Any) -> "Composite":
        out = Composite({}, device=device)
        for k, v in self._specs.items():
            out._specs[k] = v.to(device)
        return out

    def is_in(self, value: Any) -> bool:
        try:
            return all(v.is_in(value[k]) for k, v in self._specs.items())
        except KeyError:
            return False

    def encode(self, value: Mapping[str, Any]) -> TensorDict:
        out = TensorDict({}, batch_size=[], device=self.device)
        for k, v in self._specs.items():
            out._data[k] = v.encode(value[k])
        return out


__all__ = ["Categorical", "Composite", "TensorSpec", "Unbounded"]
