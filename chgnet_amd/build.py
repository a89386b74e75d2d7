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

HIP_SOURCES = ["engine.hip", "comm.hip"]
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
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


def build_hip(force: bool = False) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".hip"))]
    deps.append(os.path.join(INCLUDE, "chgnet_hip.h"))
    if force or not _newer(HIP_LIB, deps):
        _run([hipcc_path(), *HIP_FLAGS, f"-I{INCLUDE}", f"-I{CSRC}", *srcs, "-o", HIP_LIB])
    return HIP_LIB


def build_variant(name: str, defines: list[str], extra_flags: list[str] | None = None) -> str:
    """Experiment build (timing studies only): libchgnet_hip_<name>.so with extra -D / compiler flags."""
    out = os.path.join(LIB_DIR, f"libchgnet_hip_{name}.so")
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    if any(d.startswith(("CHG_EXP_", "CHG_PHASE_TIMING")) for d in defines) and "CHG_EXPERIMENTS" not in defines:
        defines = [*defines, "CHG_EXPERIMENTS"]   # mfma_tile.h refuses experiment switches without it
    _run([hipcc_path(), *HIP_FLAGS, *(extra_flags or []), *[f"-D{d}" for d in defines], f"-I{INCLUDE}", f"-I{CSRC}", *srcs, "-o", out])
    return out


def build_all(force: bool = False) -> None:
    build_graph(force)
    build_hip(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
