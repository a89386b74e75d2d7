"""chgnet_amd -- MI355X-native engine for the CHGNet energy/force/stress/magmom path.

Host side (Python) mirrors the reference API for this path
(``CHGNet.predict_structure`` / ``predict_graph``, ``CrystalGraph``,
``CrystalGraphConverter``, ASE ``CHGNetCalculator``); all compute runs in the
hand-written gfx950 HIP kernels behind the C-ABI in ``include/chgnet_hip.h``.
There is no CPU fallback: if the HIP extension is missing, prediction raises.
"""

from __future__ import annotations

from typing import Literal

__version__ = "0.1.0"

PredTask = Literal["e", "ef", "em", "efs", "efsm"]  # reference chgnet/__init__.py:15
VALID_TASKS = ("e", "ef", "em", "efs", "efsm")

from chgnet_amd.graph import CrystalGraph, CrystalGraphConverter, Lattice, Structure  # noqa: E402

__all__ = ["CrystalGraph", "CrystalGraphConverter", "Lattice", "Structure", "PredTask", "VALID_TASKS"]


def __getattr__(name):  # lazy: keeps `import chgnet_amd` free of the HIP library
    if name == "CHGNet":
        from chgnet_amd.model import CHGNet

        return CHGNet
    if name == "CHGNetCalculator":
        from chgnet_amd.calculator import CHGNetCalculator

        return CHGNetCalculator
    raise AttributeError(name)
