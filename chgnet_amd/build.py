"""Compile the native libraries in-tree (gfx950 only).

  libchgnet_graph.so  host crystal-graph builder   (g++,   include/chgnet_graph.h)
  libchgnet_hip.so    CDNA4 kernels + engine C-ABI (hipcc, include/chgnet_hip.h)

``python -m chgnet_amd.build`` builds both; the artefacts land in ``chgnet_amd/lib``
(git-ignored, but they travel to the GPU box with the gpurun snapshot).
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
INCLUDE = os.path.join(REPO_DIR, "include")

GRAPH_LIB = os.path.join(LIB_DIR, "libchgnet_graph.so")
HIP_LIB = os.path.join(LIB_DIR, "libchgnet_hip.so")

# one translation unit per subsystem (csrc/engine_internal.h): a kernel edit recompiles the unit that launches it
HIP_SOURCES = ["engine.hip", "engine_predict.hip", "engine_predict_wide.hip", "engine_train.hip", "engine_train_wide.hip", "engine_graph.hip", "comm.hip"]
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-munsafe-fp-atomics",      # native global_atomic_add_f32, no CAS loops
    "-ffp-contract=fast",
    "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
]


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd: list[str]) -> None:
    print("[chgnet_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build_graph(force: bool = False) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    src = os.path.join(CSRC, "host_graph.cpp")
    deps = [src, os.path.join(INCLUDE, "chgnet_graph.h")]
    if force or not _newer(GRAPH_LIB, deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", f"-I{INCLUDE}", src, "-o", GRAPH_LIB])
    return GRAPH_LIB


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _compile_units(out: str, extra: list[str], obj_dir: str, force: bool) -> None:
    """Compile every unit to an object (in parallel; only those older than a source they depend on), then link."""
    from concurrent.futures import ThreadPoolExecutor

    import hashlib

    os.makedirs(obj_dir, exist_ok=True)
    # objects depend on the flags too: a stamp of the command line next to them forces a rebuild when flags / defines change
    stamp, flags_id = os.path.join(obj_dir, "flags.sha"), hashlib.sha256(" ".join([*HIP_FLAGS, *extra]).encode()).hexdigest()
    if not (os.path.exists(stamp) and open(stamp).read().strip() == flags_id):
        force = True
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))] + [os.path.join(INCLUDE, "chgnet_hip.h")]
    objs, todo = [], []
    for name in HIP_SOURCES:
        src, obj = os.path.join(CSRC, name), os.path.join(obj_dir, name.replace(".hip", ".o"))
        objs.append(obj)
        deps = [src, *headers] + ([os.path.join(CSRC, name.replace("_wide", ""))] if name.endswith("_wide.hip") else [])   # it includes that unit
        if force or not _newer(obj, deps):
            todo.append([hipcc_path(), *HIP_FLAGS, *extra, f"-I{INCLUDE}", f"-I{CSRC}", "-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as pool:
        list(pool.map(_run, todo))
    with open(stamp, "w") as f:
        f.write(flags_id)
    # relink when the library is missing OR older than any object (an interrupted link, objects restored from a snapshot)
    if todo or not _newer(out, objs):
        _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])


def build_hip(force: bool = False) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    _compile_units(HIP_LIB, [], os.path.join(LIB_DIR, "obj"), force)
    return HIP_LIB


def build_variant(name: str, defines: list[str], extra_flags: list[str] | None = None) -> str:
    """Experiment build (timing studies only): libchgnet_hip_<name>.so with extra -D / compiler flags."""
    out = os.path.join(LIB_DIR, f"libchgnet_hip_{name}.so")
    if any(d.startswith(("CHG_EXP_", "CHG_PHASE_TIMING")) for d in defines) and "CHG_EXPERIMENTS" not in defines:
        defines = [*defines, "CHG_EXPERIMENTS"]   # mfma_tile.h refuses experiment switches without it
    _compile_units(out, [*(extra_flags or []), *[f"-D{d}" for d in defines]], os.path.join(LIB_DIR, f"obj_{name}"), True)
    return out


def clean_variants() -> None:
    """Remove experiment libraries and their objects from chgnet_amd/lib: a snapshot pushed to the GPU box carries the product only."""
    for f in os.listdir(LIB_DIR):
        p = os.path.join(LIB_DIR, f)
        if (f.startswith("libchgnet_hip_") and f.endswith(".so")) or f == "split_lab" or (f.startswith("obj_") and os.path.isdir(p)):
            if os.path.isdir(p) and not os.path.islink(p):
                shutil.rmtree(p)
            else:
                os.remove(p)
    # ... and objects in lib/obj that no product unit produces (an experiment unit compiled by hand)
    obj_dir = os.path.join(LIB_DIR, "obj")
    keep = {name.replace(".hip", ".o") for name in HIP_SOURCES} | {"flags.sha"}
    for f in os.listdir(obj_dir) if os.path.isdir(obj_dir) else ():
        if f not in keep:
            os.remove(os.path.join(obj_dir, f))


def build_all(force: bool = False, keep_variants: bool = False) -> None:
    build_graph(force)
    build_hip(force)
    if not keep_variants:
        clean_variants()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, keep_variants="--keep-variants" in sys.argv)
