"""Compile the REFERENCE's Python model for the GPU box's CPU baseline.

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  ``python oracle/build_ref_model.py`` (this container: needs /root/reference)
byte-compiles the reference package ``/root/reference/chgnet/**/*.py`` FROM WHERE IT LIES (``py_compile``; nothing is
copied as source) into ONE archive of ``.pyc`` files, ``oracle/_ref/chgnet_ref_bytecode.zip`` -- a built artefact like
``oracle/_ref/libref_graph.so``: git-ignored, not gpurun-ignored, so it travels to the GPU box, where /root/reference
does not exist.  ``bench.py``'s ``cpu_baseline`` leg (and only it) imports the archive through ``zipimport`` with the
same stand-in modules as ``oracle/_refimport.py`` and times the UNMODIFIED ``CHGNet.predict_graph``
(chgnet/model/model.py:593-665) on the bench box's own host cores: ``cpu_baseline.kind = "reference"``.

The archive is tied to the interpreter that wrote it (CPython magic number in every .pyc); the build container and the
GPU box run the same image.  ``load()`` refuses an archive of another interpreter instead of failing inside zipimport.
"""

from __future__ import annotations

import importlib.util
import os
import py_compile
import sys
import tempfile
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
ARCHIVE = os.path.join(HERE, "_ref", "chgnet_ref_bytecode.zip")
REFERENCE_ROOT = os.environ.get("CHGNET_REFERENCE_ROOT", "/root/reference")


def build(archive: str = ARCHIVE) -> str | None:
    """(Re)build the archive when the reference is present; returns its path, or None without a reference."""
    pkg = os.path.join(REFERENCE_ROOT, "chgnet")
    if not os.path.isdir(pkg):
        return archive if os.path.exists(archive) else None
    sources = []
    for root, dirs, files in os.walk(pkg):
        dirs[:] = [d for d in dirs if d not in ("__pycache__", "pretrained")]
        sources += [os.path.join(root, f) for f in files if f.endswith(".py")]
    if os.path.exists(archive) and all(os.path.getmtime(s) <= os.path.getmtime(archive) for s in sources):
        return archive
    os.makedirs(os.path.dirname(archive), exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp, zipfile.ZipFile(archive + ".tmp", "w", zipfile.ZIP_DEFLATED) as z:
        for i, src in enumerate(sorted(sources)):
            rel = os.path.relpath(src, REFERENCE_ROOT)
            out = os.path.join(tmp, f"{i}.pyc")
            py_compile.compile(src, cfile=out, dfile=rel, doraise=True)       # dfile: tracebacks name chgnet/..., not /root/reference
            z.write(out, rel + "c")                                           # chgnet/model/model.pyc
        z.writestr("MAGIC", importlib.util.MAGIC_NUMBER.hex())
    os.replace(archive + ".tmp", archive)
    return archive


def load(archive: str = ARCHIVE):
    """Import the reference from the bytecode archive (stand-ins for pymatgen / ase / ... as in oracle/_refimport.py)."""
    from oracle._refimport import install_stubs

    if not os.path.exists(archive):
        raise RuntimeError(f"{archive} is missing: run `python oracle/build_ref_model.py` where /root/reference exists")
    with zipfile.ZipFile(archive) as z:
        if z.read("MAGIC").decode() != importlib.util.MAGIC_NUMBER.hex():
            raise RuntimeError("the reference bytecode archive was written by another Python version")
    install_stubs()
    if "chgnet" in sys.modules and archive not in (getattr(sys.modules["chgnet"], "__file__", "") or ""):
        raise RuntimeError("the reference package is already imported from another location")
    if archive not in sys.path:
        sys.path.insert(0, archive)
    import chgnet  # noqa: PLC0415
    import chgnet.graph.crystalgraph  # noqa: F401, PLC0415
    import chgnet.model.model  # noqa: F401, PLC0415

    return chgnet


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(HERE))
    print(build())
