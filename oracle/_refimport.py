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


def build_cygraph(scratch: str | None = None) -> str:
    """make the reference's own Cython/C graph builder (``chgnet.graph.cygraph`` =
    cygraph.pyx + fast_converter_libraries/create_graph.c) importable, so that
    ``CrystalGraphConverter(algorithm="fast")`` really is the fast path.  /root/reference is read-only, hence a
    verbatim copy of the package goes to a scratch directory OUTSIDE the repository (default
    /tmp/chgnet_reference_build), where ``cythonize`` + ``build_ext --inplace`` run exactly like the reference's
    setup.py:6-10.  Returns the directory to put first on ``sys.path``."""
    import shutil  # noqa: PLC0415
    import subprocess  # noqa: PLC0415

    scratch = scratch or os.environ.get("CHGNET_REFERENCE_BUILD", "/tmp/chgnet_reference_build")
    pkg = os.path.join(scratch, "chgnet")
    have = [f for f in (os.listdir(os.path.join(pkg, "graph")) if os.path.isdir(os.path.join(pkg, "graph")) else [])
            if f.startswith("cygraph") and f.endswith(".so")]
    if not have:
        import tempfile  # noqa: PLC0415

        marker = ".chgnet_reference_build"      # only directories this function made are ever removed
        if os.path.isdir(scratch) and not os.path.exists(os.path.join(scratch, marker)):
            raise RuntimeError(f"{scratch} exists and was not created by build_cygraph (no {marker} file): refusing to replace it; "
                               "point CHGNET_REFERENCE_BUILD at a fresh path")
        parent = os.path.dirname(os.path.abspath(scratch)) or "."
        os.makedirs(parent, exist_ok=True)
        work = tempfile.mkdtemp(prefix=os.path.basename(scratch) + ".", dir=parent)   # private: parallel test processes do not collide
        open(os.path.join(work, marker), "w").close()
        shutil.copytree(os.path.join(REFERENCE_ROOT, "chgnet"), os.path.join(work, "chgnet"),
                        ignore=shutil.ignore_patterns("__pycache__", "pretrained"))
        setup_py = os.path.join(work, "setup_cygraph.py")
        with open(setup_py, "w") as fh:   # the reference's setup.py:1-10, minus the package metadata
            fh.write("import numpy as np\nfrom Cython.Build import cythonize\nfrom setuptools import Extension, setup\n"
                     "setup(name='cygraph_build', ext_modules=cythonize([Extension('chgnet.graph.cygraph', ['chgnet/graph/cygraph.pyx'],"
                     " include_dirs=[np.get_include()])], language_level=3), script_args=['build_ext', '--inplace'])\n")
        subprocess.run([sys.executable, setup_py], cwd=work, check=True, capture_output=True)
        def built(root: str) -> bool:
            graph_dir = os.path.join(root, "chgnet", "graph")
            return os.path.isdir(graph_dir) and any(f.startswith("cygraph") and f.endswith(".so") for f in os.listdir(graph_dir))

        try:
            if built(scratch):                    # another process finished while this one compiled: its build stays, ours goes
                shutil.rmtree(work, ignore_errors=True)
            else:
                if os.path.isdir(scratch):        # a marked directory WITHOUT a finished build (an interrupted earlier run): replace it
                    shutil.rmtree(scratch)
                os.rename(work, scratch)
        except OSError:                           # lost the rename race: use the winner's build
            shutil.rmtree(work, ignore_errors=True)
            if not built(scratch):
                raise
    return scratch


def load_reference(fast_graph: bool = False):
    """Return the reference ``chgnet`` package (model + graph sub-modules imported).  ``fast_graph=True``
    imports it from the scratch copy that carries the compiled ``cygraph`` (see ``build_cygraph``): the
    Python sources are byte-identical to /root/reference, only the extension module is added."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    install_stubs()
    root = build_cygraph() if fast_graph else REFERENCE_ROOT
    if "chgnet" in sys.modules and not getattr(sys.modules["chgnet"], "__file__", "").startswith(root):
        raise RuntimeError("the reference package is already imported from another location")
    if root not in sys.path:
        sys.path.insert(0, root)
    import chgnet  # noqa: PLC0415
    import chgnet.graph.converter  # noqa: F401, PLC0415
    import chgnet.graph.crystalgraph  # noqa: F401, PLC0415
    import chgnet.model.model  # noqa: F401, PLC0415

    return chgnet
