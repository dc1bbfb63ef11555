"""A small batched tensor container with the slice of the ``tensordict.TensorDict``
surface that the rl8 PPO path touches.

The reference passes ``tensordict.TensorDict`` objects across every boundary on
the hot path (``Env.step`` output ``src/rl8/env.py:226-230``, the rollout buffer
``src/rl8/algorithms/_feedforward.py:256``, ``Policy.sample`` output
``src/rl8/policies/_feedforward.py:154-158``, ``ppo_losses`` output
``src/rl8/nn/functional.py:355-363``). Neither ``tensordict`` nor ``torchrl`` is
installable on the MI355X boxes, so the drop-in keeps the *shape* of that API
with a container of its own: a mapping of keys to tensors (or nested
containers) that all share leading batch dimensions, where indexing the
container indexes every leaf along those batch dimensions and returns views.

Only behaviour the path relies on is provided; it is not a general
re-implementation of the ``tensordict`` package.

"""

from __future__ import annotations

from typing import Any, Callable, Iterator, Mapping

import torch

Key = "str | tuple[str, ...]"


def _as_size(batch_size: Any) -> torch.Size:
    if batch_size is None:
        return torch.Size([])
    if isinstance(batch_size, torch.Size):
        return batch_size
    if isinstance(batch_size, int):
        return torch.Size([batch_size])
    return torch.Size(list(batch_size))


def _is_index_key(item: Any) -> bool:
    """Strings and tuples made only of strings address leaves; anything else
    addresses batch elements."""
    if isinstance(item, str):
        return True
    return (
        isinstance(item, tuple)
        and len(item) > 0
        and all(isinstance(i, str) for i in item)
    )


def _strip_ellipsis(index: Any) -> Any:
    """``td[:, :t, ...]`` means "rest of the dims untouched"; leaves already
    get that for free so the trailing ellipsis can go."""
    if isinstance(index, tuple):
        index = tuple(i for i in index if i is not Ellipsis)
        if len(index) == 1:
            return index[0]
        if len(index) == 0:
            return slice(None)
    elif index is Ellipsis:
        return slice(None)
    return index


class TensorDict:
    """Mapping of string keys to tensors that share leading batch dimensions.

    Args:
        source: Mapping of keys to tensors, nested mappings, or other
            :class:`TensorDict` instances.
        batch_size: Leading dimensions common to every leaf.
        device: Optional device every leaf is moved to.

    """

    __slots__ = ("_data", "_batch_size", "_device")

    def __init__(
        self,
        source: None | Mapping[str, Any] = None,
        batch_size: Any = None,
        device: None | str | torch.device = None,
    ) -> None:
        self._data: dict[str, Any] = {}
        self._batch_size = _as_size(batch_size)
        self._device = torch.device(device) if device is not None else None
        if source:
            for k, v in source.items():
                self[k] = v

    # ------------------------------------------------------------------ #
    # Introspection.
    # ------------------------------------------------------------------ #
    @property
    def batch_size(self) -> torch.Size:
        return self._batch_size

    @property
    def shape(self) -> torch.Size:
        return self._batch_size

    @property
    def batch_dims(self) -> int:
        return len(self._batch_size)

    def dim(self) -> int:
        return len(self._batch_size)

    ndim = property(dim)

    @property
    def device(self) -> None | torch.device:
        if self._device is not None:
            return self._device
        for v in self._data.values():
            return v.device
        return None

    def size(self, dim: None | int = None) -> Any:
        if dim is None:
            return self._batch_size
        return self._batch_size[dim]

    def numel(self) -> int:
        return self._batch_size.numel()

    def keys(self) -> Any:
        return self._data.keys()

    def values(self) -> Any:
        return self._data.values()

    def items(self) -> Any:
        return self._data.items()

    def __iter__(self) -> Iterator[str]:
        return iter(self._data)

    def __len__(self) -> int:
        return self._batch_size[0] if len(self._batch_size) else 0

    def __contains__(self, key: str) -> bool:
        return key in self._data

    def __repr__(self) -> str:
        fields = ", ".join(
            f"{k}: {tuple(v.shape)}" + (f" {v.dtype}" if torch.is_tensor(v) else "")
            for k, v in self._data.items()
        )
        return (
            f"TensorDict({{{fields}}}, batch_size={list(self._batch_size)},"
            f" device={self.device})"
        )

    # ------------------------------------------------------------------ #
    # Leaf validation.
    # ------------------------------------------------------------------ #
    def _coerce(self, key: str, value: Any) -> Any:
        if isinstance(value, TensorDict):
            pass
        elif isinstance(value, Mapping):
            value = TensorDict(value, batch_size=self._batch_size, device=self._device)
        elif not torch.is_tensor(value):
            value = torch.as_tensor(value)
        nb = len(self._batch_size)
        if tuple(value.shape[:nb]) != tuple(self._batch_size):
            raise RuntimeError(
                f"Leaf {key!r} with shape {tuple(value.shape)} does not start"
                f" with batch size {tuple(self._batch_size)}."
            )
        if self._device is not None and value.device != self._device:
            value = value.to(self._device)
        return value

    # ------------------------------------------------------------------ #
    # Key / index access.
    # ------------------------------------------------------------------ #
    def get(self, key: Any, default: Any = None) -> Any:
        try:
            return self[key]
        except KeyError:
            return default

    def __getitem__(self, item: Any) -> Any:
        if isinstance(item, str):
            return self._data[item]
        if _is_index_key(item):
            out: Any = self
            for k in item:
                out = out[k]
            return out
        return self._index(item)

    def _index(self, index: Any) -> "TensorDict":
        index = _strip_ellipsis(index)
        nb = len(self._batch_size)
        out = TensorDict({}, batch_size=[], device=self._device)
        new_batch: None | torch.Size = None
        for k, v in self._data.items():
            if isinstance(v, TensorDict):
                sub = v._index(index)
                out._data[k] = sub
                if new_batch is None:
                    new_batch = sub.batch_size[: len(sub.batch_size) - (len(v.batch_size) - nb)]
            else:
                leaf = v[index]
                out._data[k] = leaf
                if new_batch is None:
                    new_batch = leaf.shape[: leaf.ndim - (v.ndim - nb)]
        if new_batch is None:
            probe_index = index
            if torch.is_tensor(index):
                probe_index = index.cpu()
            elif isinstance(index, tuple):
                probe_index = tuple(
                    i.cpu() if torch.is_tensor(i) else i for i in index
                )
            new_batch = torch.empty(()).expand(self._batch_size)[probe_index].shape
        out._batch_size = torch.Size(new_batch)
        return out

    def __setitem__(self, item: Any, value: Any) -> None:
        if isinstance(item, str):
            self._data[item] = self._coerce(item, value)
            return
        if _is_index_key(item):
            target: Any = self
            for k in item[:-1]:
                if k not in target._data:
                    target._data[k] = TensorDict(
                        {}, batch_size=target._batch_size, device=target._device
                    )
                target = target._data[k]
            target[item[-1]] = value
            return
        self._index_assign(item, value)

    def _index_assign(self, index: Any, value: Any) -> None:
        index = _strip_ellipsis(index)
        if isinstance(value, (TensorDict, Mapping)):
            for k, v in value.items():
                leaf = self._data[k]
                if isinstance(leaf, TensorDict):
                    leaf._index_assign(index, v)
                else:
                    leaf[index] = v
        else:
            for leaf in self._data.values():
                if isinstance(leaf, TensorDict):
                    leaf._index_assign(index, value)
                else:
                    leaf[index] = value

    def __delitem__(self, key: Any) -> None:
        if isinstance(key, str):
            del self._data[key]
            return
        target: Any = self
        for k in key[:-1]:
            target = target._data[k]
        del target._data[key[-1]]

    def set(self, key: Any, value: Any) -> "TensorDict":
        self[key] = value
        return self

    def pop(self, key: str, *default: Any) -> Any:
        return self._data.pop(key, *default)

    def update(self, other: Mapping[str, Any]) -> "TensorDict":
        for k, v in other.items():
            self[k] = v
        return self

    # ------------------------------------------------------------------ #
    # Whole-container transforms.
    # ------------------------------------------------------------------ #
    def apply(
        self,
        fn: Callable[[torch.Tensor], torch.Tensor],
        *,
        batch_size: Any = None,
    ) -> "TensorDict":
        new_batch = self._batch_size if batch_size is None else _as_size(batch_size)
        out = TensorDict({}, batch_size=new_batch, device=self._device)
        for k, v in self._data.items():
            if isinstance(v, TensorDict):
                out._data[k] = v.apply(fn, batch_size=batch_size)
            else:
                out._data[k] = fn(v)
        return out

    def reshape(self, *shape: Any) -> "TensorDict":
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        nb = len(self._batch_size)
        numel = self._batch_size.numel()
        dims = list(shape)
        if -1 in dims:
            known = 1
            for d in dims:
                if d != -1:
                    known *= d
            dims[dims.index(-1)] = numel // known if known else 0
        new_batch = torch.Size(dims)
        out = TensorDict({}, batch_size=new_batch, device=self._device)
        for k, v in self._data.items():
            if isinstance(v, TensorDict):
                out._data[k] = v.reshape(*dims, *v.batch_size[nb:])
            else:
                out._data[k] = v.reshape(*dims, *v.shape[nb:])
        return out

    def flatten(self, start_dim: int = 0, end_dim: int = -1) -> "TensorDict":
        nb = len(self._batch_size)
        if end_dim < 0:
            end_dim += nb
        dims = list(self._batch_size)
        merged = 1
        for d in dims[start_dim : end_dim + 1]:
            merged *= d
        return self.reshape(*dims[:start_dim], merged, *dims[end_dim + 1 :])

    def select(self, *keys: str, inplace: bool = False) -> "TensorDict":
        if inplace:
            for k in list(self._data.keys()):
                if k not in keys:
                    del self._data[k]
            return self
        out = TensorDict({}, batch_size=self._batch_size, device=self._device)
        for k in keys:
            out._data[k] = self._data[k]
        return out

    def exclude(self, *keys: str) -> "TensorDict":
        out = TensorDict({}, batch_size=self._batch_size, device=self._device)
        for k, v in self._data.items():
            if k not in keys:
                out._data[k] = v
        return out

    def to(self, device: Any) -> "TensorDict":
        device = torch.device(device)
        out = TensorDict({}, batch_size=self._batch_size, device=device)
        for k, v in self._data.items():
            out._data[k] = v.to(device)
        return out

    def cpu(self) -> "TensorDict":
        return self.to("cpu")

    def clone(self) -> "TensorDict":
        return self.apply(lambda x: x.clone())

    def detach(self) -> "TensorDict":
        return self.apply(lambda x: x.detach())

    def contiguous(self) -> "TensorDict":
        return self.apply(lambda x: x.contiguous())

    def to_dict(self) -> dict[str, Any]:
        return {
            k: (v.to_dict() if isinstance(v, TensorDict) else v)
            for k, v in self._data.items()
        }

    # ------------------------------------------------------------------ #
    # Comparisons (used by exact-tensor tests only).
    # ------------------------------------------------------------------ #
    def __eq__(self, other: Any) -> "TensorDict":  # type: ignore[override]
        out = TensorDict({}, batch_size=self._batch_size, device=self._device)
        for k, v in self._data.items():
            o = other[k] if isinstance(other, (TensorDict, Mapping)) else other
            out._data[k] = v == o
        return out

    __hash__ = None  # type: ignore[assignment]

    def all(self) -> bool:
        for v in self._data.values():
            if not bool(v.all()):
                return False
        return True

    def any(self) -> bool:
        for v in self._data.values():
            if bool(v.any()):
                return True
        return False


def is_tensordict(x: Any) -> bool:
    """True for this container and for duck-typed equivalents (the real
    ``tensordict.TensorDict`` if a caller hands one over)."""
    return isinstance(x, TensorDict) or (
        hasattr(x, "batch_size") and hasattr(x, "keys") and not torch.is_tensor(x)
    )


__all__ = ["TensorDict", "is_tensordict"]
