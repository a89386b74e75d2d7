"""ctypes view of oracle/_ref/libref_graph.so = the reference's compiled C graph builder
(create_graph.c, "fast" algorithm) + the flat-array shim.  TEST INFRASTRUCTURE ONLY."""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libref_graph.so")


def build(reference_root: str = "/root/reference") -> str | None:
    """Compile from the reference sources where they lie (this container only)."""
    if not os.path.isdir(os.path.join(reference_root, "chgnet", "graph", "fast_converter_libraries")):
        return LIB if os.path.exists(LIB) else None
    subprocess.run(["make", "-C", HERE, f"REF={reference_root}"], check=True, capture_output=True)
    return LIB


def available() -> bool:
    return os.path.exists(LIB)


def reference_graph(n_atoms, center, neighbor, image, distance, r_bond: float) -> dict:
    lib = ctypes.CDLL(LIB)
    center = np.ascontiguousarray(center, np.int64)
    neighbor = np.ascontiguousarray(neighbor, np.int64)
    image = np.ascontiguousarray(image, np.int64).reshape(-1, 3)
    distance = np.ascontiguousarray(distance, np.float64)
    E = len(center)
    ag = np.zeros((E, 2), np.int32)
    d2u = np.zeros(E, np.int32)
    u2d = np.zeros(max(E, 1), np.int32)
    counts = np.zeros(2, np.int64)
    cap = 64 * max(E, 1)
    bg = np.zeros((cap, 5), np.int32)
    ip, dp, i32 = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
    lib.ref_graph_flat.restype = ctypes.c_int
    st = lib.ref_graph_flat(ctypes.c_int64(n_atoms), ctypes.c_int64(E), center.ctypes.data_as(ip), neighbor.ctypes.data_as(ip),
                            image.ctypes.data_as(ip), distance.ctypes.data_as(dp), ctypes.c_double(r_bond),
                            ag.ctypes.data_as(i32), d2u.ctypes.data_as(i32), u2d.ctypes.data_as(i32), bg.ctypes.data_as(i32),
                            ctypes.c_int64(cap), counts.ctypes.data_as(ip))
    if st != 0:
        raise ValueError(f"reference create_graph: status {st}")
    return {"atom_graph": ag, "directed2undirected": d2u, "undirected2directed": u2d[: counts[0]].copy(),
            "bond_graph": bg[: counts[1]].copy()}
