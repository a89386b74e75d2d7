"""Checkpoint / graph-cache readers must not run code stored in the file."""
import os
import pickle

import numpy as np
import pytest
import torch

from chgnet_amd.graph import CrystalGraph
from chgnet_amd.model import random_state_dict
from chgnet_amd.safe_load import Inert, load_torch_file

from conftest import load_case


class _Hostile:
    """Unpickling this with the stock unpickler creates the marker file."""

    def __init__(self, marker):
        self.marker = marker

    def __reduce__(self):
        return (os.system, (f"touch {self.marker}",))


class _TrainerLike:
    def __init__(self):
        self.lr = 1e-3
        self.history = {"e": [1.0, 2.0]}


def test_checkpoint_with_foreign_objects(tmp_path):
    marker = tmp_path / "executed"
    sd = random_state_dict({}, seed=3)
    ckpt = {
        "model": {"state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "model_args": {"atom_fea_dim": 64}},
        "trainer": {"obj": _TrainerLike(), "hook": _Hostile(str(marker))},
    }
    path = tmp_path / "ckpt.pth.tar"
    torch.save(ckpt, path)
    state = load_torch_file(str(path))
    assert not marker.exists(), "code stored in the checkpoint was executed"
    assert isinstance(state["trainer"]["hook"], Inert)
    assert isinstance(state["trainer"]["obj"], Inert) and state["trainer"]["obj"].lr == 1e-3
    for k, v in sd.items():
        np.testing.assert_array_equal(state["model"]["state_dict"][k].numpy(), np.asarray(v))
    # sanity: the stock reader would have run it
    with pytest.raises(Exception):
        torch.load(path, weights_only=True)


def test_plain_checkpoint_roundtrip(tmp_path):
    sd = random_state_dict({}, seed=4)
    path = tmp_path / "plain.pth.tar"
    torch.save({"model": {"state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "model_args": {}}}, path)
    state = load_torch_file(str(path))
    assert set(state["model"]["state_dict"]) == set(sd)


class _RefGraphLike:
    """Attribute set of a pickled reference CrystalGraph (crystalgraph.py:19-100)."""


def test_pickled_graph_object(tmp_path):
    g, _ = load_case("limno2")
    obj = _RefGraphLike()
    for k, v in g.to_dict().items():
        setattr(obj, k, torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v)
    path = tmp_path / "g.pt"
    torch.save(obj, path)
    back = CrystalGraph.from_file(str(path))
    for k in ("atomic_number", "atom_graph", "bond_graph", "neighbor_image", "lattice"):
        np.testing.assert_array_equal(np.asarray(getattr(back, k)), np.asarray(getattr(g, k)))


def test_legacy_pickle_stream():
    data = pickle.dumps({"x": _Hostile("/tmp/never"), "y": [1, 2, 3]})
    from chgnet_amd.safe_load import restricted_pickle
    out = restricted_pickle.loads(data)
    assert out["y"] == [1, 2, 3] and isinstance(out["x"], Inert)
    assert not os.path.exists("/tmp/never")
