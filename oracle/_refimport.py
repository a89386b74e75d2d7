"""Import the *unmodified* reference (``/root/reference/chgnet``) in this container.

TEST INFRASTRUCTURE.  Used only by ``tests/golden/make_golden.py`` (fixture
generation) and by CPU tests that re-validate the oracle against the live reference
when ``/root/reference`` exists.  It never runs on the GPU box (the reference does
not travel) and the product never imports it.

The reference needs pymatgen / ase / pynvml / monty which are absent here; its model
math does not use them, so they are replaced by inert stand-in modules
(import sites: chgnet/model/dynamics.py:12-31, model/model.py:10,
model/composition_model.py:8, utils/common_utils.py:7, utils/vasp_utils.py:8-10,
data/dataset.py:11).
"""

from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CHGNET_REFERENCE_ROOT", "/root/reference")

_STUBS = [
    "pynvml", "monty", "monty.io", "monty.os", "monty.os.path",
    "pymatgen", "pymatgen.core", "pymatgen.core.structure", "pymatgen.analysis",
    "pymatgen.analysis.eos", "pymatgen.io", "pymatgen.io.ase", "pymatgen.io.vasp",
    "pymatgen.io.vasp.outputs", "pymatgen.symmetry", "pymatgen.symmetry.analyzer",
    "ase", "ase.units", "ase.calculators", "ase.calculators.calculator",
    "ase.md", "ase.md.npt", "ase.md.nptberendsen", "ase.md.velocitydistribution",
    "ase.md.verlet", "ase.optimize", "ase.optimize.bfgs", "ase.optimize.bfgslinesearch",
    "ase.optimize.fire", "ase.optimize.lbfgs", "ase.optimize.mdmin",
    "ase.optimize.sciopt", "ase.filters", "ase.io", "ase.constraints",
    "wandb",
]


class _StubModule(types.ModuleType):
    def __getattr__(self, name):  # any attribute is a fresh dummy class
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {})
        setattr(self, name, cls)
        return cls


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "chgnet", "model"))


def install_stubs() -> None:
    for name in _STUBS:
        if name in sys.modules:
            continue
        try:  # keep a real module if it is importable
            __import__(name)
            continue
        except Exception:
            pass
        mod = _StubModule(name)
        mod.__path__ = []  # behave as a package
        sys.modules[name] = mod
    for name in _STUBS:  # `from pkg import sub` must yield the stub sub-module
        if "." in name and isinstance(sys.modules.get(name), _StubModule):
            parent, child = name.rsplit(".", 1)
            if isinstance(sys.modules.get(parent), _StubModule):
                setattr(sys.modules[parent], child, sys.modules[name])
    sys.modules["ase.units"].GPa = 1.0 / 160.21766208
    calc = sys.modules["ase.calculators.calculator"]
    calc.all_changes = []
    calc.all_properties = []


def load_reference():
    """Return the reference ``chgnet`` package (model + graph sub-modules imported)."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import chgnet  # noqa: PLC0415
    import chgnet.graph.converter  # noqa: F401, PLC0415
    import chgnet.graph.crystalgraph  # noqa: F401, PLC0415
    import chgnet.model.model  # noqa: F401, PLC0415

    return chgnet
