from __future__ import annotations

import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with `-m gpu` on the GPU box")


def load_case(name: str):
    """Golden case -> (CrystalGraph, npz dict)."""
    from chgnet_amd.graph.crystalgraph import CrystalGraph

    d = np.load(os.path.join(GOLDEN, f"case_{name}.npz"))
    g = CrystalGraph(
        atomic_number=d["atomic_number"], atom_frac_coord=d["atom_frac_coord"], atom_graph=d["atom_graph"],
        atom_graph_cutoff=6, neighbor_image=d["neighbor_image"], directed2undirected=d["directed2undirected"],
        undirected2directed=d["undirected2directed"], bond_graph=d["bond_graph"], bond_graph_cutoff=3,
        lattice=d["lattice"], graph_id=name)
    return g, d


@pytest.fixture(scope="session")
def golden_weights():
    return dict(np.load(os.path.join(GOLDEN, "weights_seed0.npz")))


@pytest.fixture(scope="session")
def trained_like_weights():
    """Second golden weight set at trained-checkpoint magnitudes (tests/golden/make_golden.py)."""
    return dict(np.load(os.path.join(GOLDEN, "weights_trained_like.npz")))


@pytest.fixture(scope="session")
def packed_weights(golden_weights):
    from chgnet_amd.pack import pack_weights

    return pack_weights(golden_weights)


@pytest.fixture(scope="session")
def hip_engine(packed_weights):
    """The HIP engine on cuda:0 -- fails (not skips) if the extension or the GPU is missing."""
    from chgnet_amd.engine import Engine

    eng = Engine(packed_weights, 0)
    yield eng
    eng.close()
