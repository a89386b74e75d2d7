"""``CrystalGraphConverter`` -- structure -> ``CrystalGraph`` via the native builder.

Mirrors reference chgnet/graph/converter.py:27-291 (constructor arguments, isolated-atom
policy, ``as_dict``/``from_dict``), but the neighbour list, the directed/undirected
bookkeeping and the bond-graph enumeration all run in csrc/host_graph.cpp behind
include/chgnet_graph.h -- there is no Python ``legacy`` algorithm here.
"""

from __future__ import annotations

import ctypes
import os
import sys

import numpy as np

from chgnet_amd.graph.crystalgraph import CrystalGraph
from chgnet_amd.graph.structure import atomic_numbers_of

_LIB = None


class _CGraph(ctypes.Structure):
    _fields_ = [
        ("n_atoms", ctypes.c_int32), ("n_directed", ctypes.c_int32), ("n_undirected", ctypes.c_int32),
        ("n_angles", ctypes.c_int32), ("n_isolated", ctypes.c_int32),
        ("atom_graph", ctypes.POINTER(ctypes.c_int32)), ("image", ctypes.POINTER(ctypes.c_int32)),
        ("distance", ctypes.POINTER(ctypes.c_double)),
        ("directed2undirected", ctypes.POINTER(ctypes.c_int32)),
        ("undirected2directed", ctypes.POINTER(ctypes.c_int32)),
        ("bond_graph", ctypes.POINTER(ctypes.c_int32)),
    ]


def graph_lib() -> ctypes.CDLL:
    """Load libchgnet_graph.so (built on demand with g++; raises if that fails)."""
    global _LIB  # noqa: PLW0603
    if _LIB is None:
        from chgnet_amd.build import GRAPH_LIB, build_graph

        if not os.path.exists(GRAPH_LIB):
            build_graph()
        lib = ctypes.CDLL(GRAPH_LIB)
        pp = ctypes.POINTER(ctypes.POINTER(_CGraph))
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int64)
        lib.chg_graph_build.argtypes = [ctypes.c_int32, dp, dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, pp]
        lib.chg_graph_build.restype = ctypes.c_int
        lib.chg_graph_build_with.argtypes = [ctypes.c_int32, dp, dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, pp]
        lib.chg_graph_build_with.restype = ctypes.c_int
        lib.chg_graph_from_neighbors.argtypes = [ctypes.c_int32, ctypes.c_int64, ip, ip, ip, dp, ctypes.c_double, pp]
        lib.chg_graph_from_neighbors.restype = ctypes.c_int
        lib.chg_pack_batch.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
        lib.chg_pack_batch.restype = ctypes.c_int
        lib.chg_graph_free.argtypes = [ctypes.POINTER(_CGraph)]
        lib.chg_graph_free.restype = None
        lib.chg_graph_strerror.argtypes = [ctypes.c_int]
        lib.chg_graph_strerror.restype = ctypes.c_char_p
        _LIB = lib
    return _LIB


def _take(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def _unpack(gp) -> dict:
    g = gp.contents
    ed, eu, na = g.n_directed, g.n_undirected, g.n_angles
    return {
        "n_isolated": g.n_isolated,
        "atom_graph": _take(g.atom_graph, 2 * ed, np.int32).reshape(ed, 2),
        "image": _take(g.image, 3 * ed, np.int32).reshape(ed, 3),
        "distance": _take(g.distance, ed, np.float64),
        "directed2undirected": _take(g.directed2undirected, ed, np.int32),
        "undirected2directed": _take(g.undirected2directed, eu, np.int32),
        "bond_graph": _take(g.bond_graph, 5 * na, np.int32).reshape(na, 5),
    }


def _check(lib, status: int) -> None:
    if status != 0:
        msg = lib.chg_graph_strerror(status).decode()
        raise (ValueError if status in (-1, -3) else MemoryError)(f"graph builder: {msg}")


_SEARCH = {"auto": 0, "pairs": 1, "cells": 2}


def build_graph_arrays(frac: np.ndarray, lattice: np.ndarray, r_atom: float, r_bond: float,
                       numerical_tol: float = 1e-8, search: str = "auto") -> dict:
    """Neighbour list + graph for one structure -> dict of flat arrays.  ``search``: "auto" (all pairs below 1024
    atoms, cell list above), "pairs" or "cells" -- the result is the same, bit for bit."""
    lib = graph_lib()
    frac = np.ascontiguousarray(frac, dtype=np.float64)
    lattice = np.ascontiguousarray(lattice, dtype=np.float64)
    out = ctypes.POINTER(_CGraph)()
    dp = ctypes.POINTER(ctypes.c_double)
    st = lib.chg_graph_build_with(len(frac), frac.ctypes.data_as(dp), lattice.ctypes.data_as(dp),
                                  float(r_atom), float(r_bond), float(numerical_tol), _SEARCH[search], ctypes.byref(out))
    _check(lib, st)
    try:
        return _unpack(out)
    finally:
        lib.chg_graph_free(out)


def graph_arrays_from_neighbors(n_atoms: int, center, neighbor, image, distance, r_bond: float) -> dict:
    """Graph from a caller-supplied neighbour list (e.g. pymatgen's), rows in given order."""
    lib = graph_lib()
    center = np.ascontiguousarray(center, dtype=np.int64)
    neighbor = np.ascontiguousarray(neighbor, dtype=np.int64)
    image = np.ascontiguousarray(image, dtype=np.int64).reshape(-1, 3)
    distance = np.ascontiguousarray(distance, dtype=np.float64)
    out = ctypes.POINTER(_CGraph)()
    ip, dp = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)
    st = lib.chg_graph_from_neighbors(int(n_atoms), len(center), center.ctypes.data_as(ip),
                                      neighbor.ctypes.data_as(ip), image.ctypes.data_as(ip),
                                      distance.ctypes.data_as(dp), float(r_bond), ctypes.byref(out))
    _check(lib, st)
    try:
        return _unpack(out)
    finally:
        lib.chg_graph_free(out)


class CrystalGraphConverter:
    """Convert a structure to a ``CrystalGraph`` (reference converter.py:27-190)."""

    def __init__(
        self,
        *,
        atom_graph_cutoff: float = 6,
        bond_graph_cutoff: float = 3,
        algorithm: str = "fast",
        on_isolated_atoms: str = "error",
        verbose: bool = False,
    ) -> None:
        self.atom_graph_cutoff = atom_graph_cutoff
        self.bond_graph_cutoff = atom_graph_cutoff if bond_graph_cutoff is None else bond_graph_cutoff
        self.on_isolated_atoms = on_isolated_atoms
        # the reference offers "legacy" (Python) and "fast" (C); here both names select the
        # same native builder, kept so that saved model_args round-trip (converter.py:66-86)
        self.algorithm = "fast"
        graph_lib()
        if verbose:
            print(self)

    def __repr__(self) -> str:
        atom_graph_cutoff = self.atom_graph_cutoff
        bond_graph_cutoff = self.bond_graph_cutoff
        algorithm = self.algorithm
        return f"{type(self).__name__}({algorithm=}, {atom_graph_cutoff=}, {bond_graph_cutoff=})"

    def __call__(self, structure, graph_id=None, mp_id=None) -> CrystalGraph:
        return self.forward(structure, graph_id=graph_id, mp_id=mp_id)

    def forward(self, structure, graph_id=None, mp_id=None) -> CrystalGraph:
        """Structure (ours or pymatgen's) -> CrystalGraph (reference converter.py:102-190)."""
        n_atoms = len(structure)
        atomic_number = atomic_numbers_of(structure)
        frac = np.asarray(structure.frac_coords, dtype=np.float64).reshape(n_atoms, 3)
        lattice = np.asarray(structure.lattice.matrix, dtype=np.float64)
        arrays = build_graph_arrays(frac, lattice, self.atom_graph_cutoff, self.bond_graph_cutoff)
        n_isolated_atoms = arrays["n_isolated"]
        if n_isolated_atoms:
            atom_graph_cutoff = self.atom_graph_cutoff
            msg = (
                f"Structure {graph_id=} has {n_isolated_atoms} isolated atom(s) with "
                f"{atom_graph_cutoff=}. "
                f"CHGNet calculation will likely go wrong"
            )
            if self.on_isolated_atoms == "error":
                raise ValueError(msg)
            elif self.on_isolated_atoms == "warn":  # noqa: RET506
                print(msg, file=sys.stderr)
        return CrystalGraph(
            atomic_number=atomic_number,
            atom_frac_coord=frac.astype(np.float32),
            atom_graph=arrays["atom_graph"],
            neighbor_image=arrays["image"].astype(np.float32),
            directed2undirected=arrays["directed2undirected"],
            undirected2directed=arrays["undirected2directed"],
            bond_graph=arrays["bond_graph"],
            lattice=lattice.astype(np.float32),
            graph_id=graph_id,
            mp_id=mp_id,
            composition=structure.composition.formula,
            atom_graph_cutoff=self.atom_graph_cutoff,
            bond_graph_cutoff=self.bond_graph_cutoff,
        )

    def set_isolated_atom_response(self, on_isolated_atoms: str) -> None:
        self.on_isolated_atoms = on_isolated_atoms

    def as_dict(self) -> dict:
        return {
            "atom_graph_cutoff": self.atom_graph_cutoff,
            "bond_graph_cutoff": self.bond_graph_cutoff,
            "algorithm": self.algorithm,
        }

    @classmethod
    def from_dict(cls, dct: dict) -> "CrystalGraphConverter":
        return cls(**dct)
