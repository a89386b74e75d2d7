"""Read reference ``torch.save`` files (checkpoints ``.pth.tar``, CrystalGraph ``.pt`` caches)
without executing code stored in them.

The reference loads both with plain ``torch.load`` (model.py:684-688, crystalgraph.py:138-167),
which unpickles arbitrary callables.  Here the pickle stream is read with an ``Unpickler`` whose
``find_class`` resolves only the names needed to rebuild tensors, numpy arrays and plain
containers; every other global in the stream becomes an inert placeholder that swallows its
constructor arguments (its attribute dict, data only, is kept), so objects the engine never looks at (trainer bookkeeping,
custom classes) load as harmless stand-ins instead of running.
"""

from __future__ import annotations

import pickle
import types

_ALLOWED = {
    ("collections", "OrderedDict"),
    ("collections", "defaultdict"),
    ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "slice"), ("builtins", "complex"),
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "int"),
    ("builtins", "float"), ("builtins", "str"), ("builtins", "bool"), ("builtins", "bytes"),
    ("builtins", "bytearray"), ("builtins", "object"),
    ("copyreg", "_reconstructor"),
    ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_tensor_v2"),
    ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
    ("torch.nn.parameter", "Parameter"),
    ("torch.serialization", "_get_layout"),
    ("numpy", "dtype"), ("numpy", "ndarray"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
}
_TORCH_DTYPES = {"float16", "float32", "float64", "bfloat16", "int8", "int16", "int32", "int64", "uint8", "bool",
                 "half", "float", "double", "short", "int", "long", "complex64", "complex128"}


class Inert:
    """Stand-in for any object whose class is not on the allow list."""

    def __init__(self, *args, **kwargs) -> None:
        pass

    def __call__(self, *args, **kwargs):
        return Inert()

    def __setstate__(self, state) -> None:
        if isinstance(state, dict):          # keep plain data attributes (a pickled CrystalGraph keeps its tensors)
            self.__dict__.update(state)

    def __reduce__(self):
        return (Inert, ())

    # containers restored through append / extend / __setitem__ opcodes
    def append(self, item) -> None:
        pass

    def extend(self, items) -> None:
        pass

    def __setitem__(self, key, value) -> None:
        pass

    def __repr__(self) -> str:
        return "<inert>"


def _inert_class(module: str, name: str) -> type:
    return type(name, (Inert,), {"__module__": f"inert.{module}"})


class RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        if module == "torch" and (name in _TORCH_DTYPES or name.endswith("Storage")):
            return super().find_class(module, name)
        return _inert_class(module, name)


# the ``pickle_module`` interface torch.load expects: .Unpickler, .load, .loads and the protocol constants
restricted_pickle = types.ModuleType("chgnet_amd.restricted_pickle")
restricted_pickle.__dict__.update({k: getattr(pickle, k) for k in dir(pickle) if not k.startswith("__")})
restricted_pickle.Unpickler = RestrictedUnpickler


def _load(file, **kwargs):
    return RestrictedUnpickler(file, **kwargs).load()


def _loads(data, **kwargs):
    import io

    return RestrictedUnpickler(io.BytesIO(data), **kwargs).load()


restricted_pickle.load = _load
restricted_pickle.loads = _loads


def load_torch_file(path):
    """``torch.load(path)`` on the CPU with the restricted unpickler (zip and legacy formats)."""
    import torch

    try:
        return torch.load(path, map_location="cpu", weights_only=True)   # plain tensor/dict files: torch's own safe reader
    except Exception:  # noqa: BLE001  files holding other objects: fall through to the restricted reader
        pass
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=restricted_pickle)
