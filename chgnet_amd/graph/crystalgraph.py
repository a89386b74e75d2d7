"""``CrystalGraph`` -- the input record of the hot path.

Field-for-field the record defined at reference chgnet/graph/crystalgraph.py:18-100
(same names, shapes and dtypes: int32 indices, float32 geometry), held as numpy arrays
because the engine consumes host buffers through a C-ABI.  ``predict_graph`` also
accepts the reference's torch-backed ``CrystalGraph`` (anything exposing these
attributes).
"""

from __future__ import annotations

from typing import Any

import numpy as np

_FIELDS = (
    "atomic_number", "atom_frac_coord", "atom_graph", "neighbor_image",
    "directed2undirected", "undirected2directed", "bond_graph", "lattice",
)


def as_numpy(x, dtype) -> np.ndarray:
    """numpy view/copy of a numpy array, list or (CPU/GPU, grad-tracking) torch tensor."""
    if hasattr(x, "detach"):  # torch.Tensor without importing torch
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


class CrystalGraph:
    """A data class for crystal graph (see reference crystalgraph.py:15-100)."""

    def __init__(
        self,
        atomic_number,
        atom_frac_coord,
        atom_graph,
        atom_graph_cutoff: float,
        neighbor_image,
        directed2undirected,
        undirected2directed,
        bond_graph,
        bond_graph_cutoff: float,
        lattice,
        graph_id: str | None = None,
        mp_id: str | None = None,
        composition: str | None = None,
    ) -> None:
        self.atomic_number = as_numpy(atomic_number, np.int32).reshape(-1)
        self.atom_frac_coord = as_numpy(atom_frac_coord, np.float32).reshape(-1, 3)
        self.atom_graph = as_numpy(atom_graph, np.int32).reshape(-1, 2)
        self.atom_graph_cutoff = atom_graph_cutoff
        self.neighbor_image = as_numpy(neighbor_image, np.float32).reshape(-1, 3)
        self.directed2undirected = as_numpy(directed2undirected, np.int32).reshape(-1)
        self.undirected2directed = as_numpy(undirected2directed, np.int32).reshape(-1)
        self.bond_graph = as_numpy(bond_graph, np.int32).reshape(-1, 5)
        self.bond_graph_cutoff = bond_graph_cutoff
        self.lattice = as_numpy(lattice, np.float32).reshape(3, 3)
        self.graph_id = graph_id
        self.mp_id = mp_id
        self.composition = composition
        if len(self.directed2undirected) != 2 * len(self.undirected2directed):
            raise ValueError(
                f"{graph_id} number of directed indices ({len(self.directed2undirected)}) !="
                f" 2 * number of undirected indices ({2 * len(self.undirected2directed)})!"
            )

    @classmethod
    def from_reference(cls, g) -> "CrystalGraph":
        """Adopt any object exposing the reference CrystalGraph attributes."""
        if isinstance(g, cls):
            return g
        return cls(
            atomic_number=g.atomic_number, atom_frac_coord=g.atom_frac_coord,
            atom_graph=g.atom_graph, atom_graph_cutoff=g.atom_graph_cutoff,
            neighbor_image=g.neighbor_image, directed2undirected=g.directed2undirected,
            undirected2directed=g.undirected2directed, bond_graph=g.bond_graph,
            bond_graph_cutoff=g.bond_graph_cutoff, lattice=g.lattice,
            graph_id=getattr(g, "graph_id", None), mp_id=getattr(g, "mp_id", None),
            composition=getattr(g, "composition", None),
        )

    def to(self, device: str = "cpu") -> "CrystalGraph":  # noqa: ARG002
        """Reference crystalgraph.py:102-118 moves tensors; host buffers stay put here
        (the engine uploads the packed batch itself)."""
        return self

    def to_dict(self) -> dict[str, Any]:
        out = {k: getattr(self, k) for k in _FIELDS}
        out.update(atom_graph_cutoff=self.atom_graph_cutoff, bond_graph_cutoff=self.bond_graph_cutoff,
                   graph_id=self.graph_id, mp_id=self.mp_id, composition=self.composition)
        return out

    @classmethod
    def from_dict(cls, dic: dict[str, Any]) -> "CrystalGraph":
        return cls(**dic)

    def save(self, fname: str | None = None, save_dir: str = ".") -> str:
        """Save in the reference's on-disk format (crystalgraph.py:138-155): ``torch.save`` of the
        ``to_dict()`` dictionary with tensor values, so files are interchangeable both ways
        (``GraphData`` caches, dataset.py:400-402).  torch is used as the (de)serialiser only."""
        import os

        import torch

        if fname is not None:
            save_name = os.path.join(save_dir, fname)
        elif self.graph_id is not None:
            save_name = os.path.join(save_dir, f"{self.graph_id}.pt")
        else:
            save_name = os.path.join(save_dir, f"{self.composition}.pt")
        dic = self.to_dict()
        for key in _FIELDS:
            dic[key] = torch.from_numpy(np.ascontiguousarray(dic[key]))
        torch.save(dic, f=save_name)
        return save_name

    @classmethod
    def from_file(cls, file_name: str) -> "CrystalGraph":
        """Load a graph written by ``save`` here or by the reference (a dict of tensors)."""
        from ..safe_load import load_torch_file

        obj = load_torch_file(file_name)
        if isinstance(obj, dict):
            return cls.from_dict(obj)
        return cls.from_reference(obj)       # a pickled reference CrystalGraph object

    @property
    def num_isolated_atoms(self) -> int:
        """Atoms without any bond inside the atom-graph cutoff (reference crystalgraph.py:188-198)."""
        return len(self.atomic_number) - len(np.unique(self.atom_graph[:, 0]))

    def __repr__(self) -> str:
        composition = self.composition
        atom_graph_cutoff = self.atom_graph_cutoff
        bond_graph_cutoff = self.bond_graph_cutoff
        n_atoms = len(self.atomic_number)
        atom_graph_len = len(self.atom_graph)
        bond_graph_len = len(self.bond_graph)
        return (
            f"CrystalGraph({composition=}, {atom_graph_cutoff=}, {bond_graph_cutoff=}, "
            f"{n_atoms=}, {atom_graph_len=}, {bond_graph_len=})"
        )
