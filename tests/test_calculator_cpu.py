"""CHGNetCalculator's ASE-facing conversion (reference chgnet/model/dynamics.py:150-181: AseAtomsAdaptor.get_structure(atoms) ->
graph converter -> predict) under an ``Atoms``-shaped object.  ASE is absent in the build container, so a stand-in with the three
accessors the calculator uses stands for ``ase.Atoms``; where ASE imports, the same checks run on the real class."""

from __future__ import annotations

import numpy as np
import pytest

from chgnet_amd.calculator import HAVE_ASE, atoms_to_structure
from chgnet_amd.graph.structure import Lattice, Structure


class CellDuck:
    """``ase.cell.Cell``: indexable, ``cell[:]`` is the 3x3 array."""

    def __init__(self, m) -> None:
        self._m = np.array(m, dtype=np.float64)

    def __getitem__(self, item):
        return self._m[item]


class AtomsDuck:
    """The slice of ``ase.Atoms`` the calculator touches: get_cell()[:], get_scaled_positions(wrap=False), get_atomic_numbers()."""

    def __init__(self, cell, numbers, positions) -> None:
        self._cell, self._z, self._pos = np.array(cell, np.float64), np.array(numbers), np.array(positions, np.float64)
        self.wrap_seen = []

    def get_cell(self):
        return CellDuck(self._cell)

    def get_atomic_numbers(self):
        return self._z.copy()

    def get_positions(self):
        return self._pos.copy()

    def get_scaled_positions(self, wrap=True):
        self.wrap_seen.append(wrap)
        frac = np.linalg.solve(self._cell.T, self._pos.T).T
        return frac % 1.0 if wrap else frac

    def copy(self):
        return AtomsDuck(self._cell, self._z, self._pos)

    def __len__(self) -> int:
        return len(self._z)


def triclinic_case():
    rng = np.random.default_rng(4)
    cell = np.array([[4.1, 0.0, 0.0], [1.3, 4.4, 0.0], [0.7, -0.9, 5.2]])
    frac = rng.random((9, 3)) * 1.6 - 0.3                 # unwrapped: some coordinates outside [0, 1)
    z = rng.choice([3, 8, 25, 27], size=9)
    return cell, z, frac


def test_atoms_duck_becomes_the_same_structure():
    cell, z, frac = triclinic_case()
    atoms = AtomsDuck(cell, z, frac @ cell)
    s = atoms_to_structure(atoms)
    assert atoms.wrap_seen == [False], "unwrapped fractional coordinates (images of the graph stay the reference's)"
    want = Structure(Lattice(cell), z, frac)
    assert np.allclose(np.asarray(s.lattice.matrix), cell, rtol=0, atol=0)
    assert np.array_equal(np.asarray(s.atomic_numbers), z)
    assert np.abs(np.asarray(s.frac_coords) - frac).max() < 1e-14
    assert np.abs(np.asarray(s.cart_coords) - np.asarray(want.cart_coords)).max() < 1e-13
    assert atoms_to_structure(want) is want              # structures pass through


def test_atoms_duck_and_structure_give_the_same_graph():
    from chgnet_amd import CrystalGraphConverter

    cell, z, frac = triclinic_case()
    conv = CrystalGraphConverter(atom_graph_cutoff=6, bond_graph_cutoff=3)
    conv.set_isolated_atom_response("ignore")
    g1 = conv(atoms_to_structure(AtomsDuck(cell, z, frac @ cell)))
    g2 = conv(Structure(Lattice(cell), z, frac))
    for k in ("atom_graph", "bond_graph", "neighbor_image", "directed2undirected", "undirected2directed"):
        assert np.array_equal(np.asarray(getattr(g1, k)), np.asarray(getattr(g2, k))), k


@pytest.mark.skipif(not HAVE_ASE, reason="ASE is not installed here: the stand-in above covers the accessor contract")
def test_real_ase_atoms():
    from ase import Atoms

    cell, z, frac = triclinic_case()
    atoms = Atoms(numbers=z, cell=cell, scaled_positions=frac, pbc=True)
    s = atoms_to_structure(atoms)
    assert np.abs(np.asarray(s.frac_coords) - frac).max() < 1e-12 and np.array_equal(np.asarray(s.atomic_numbers), z)
