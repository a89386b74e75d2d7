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


# ---- files the REFERENCE itself wrote (tests/golden/make_ref_files.py imports the unmodified reference and calls its own
#      Trainer.save / CrystalGraph.save) ------------------------------------------------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reference_trainer_save_is_read_bit_for_bit():
    """trainer/trainer.py:614-623 -> CHGNet.from_file: every tensor of the state_dict equals the live module's at save time
    (ref_trainer_save_state.npz), model_args come back as plain data, and the bookkeeping the engine never looks at -- a REAL
    torch.optim.Adam state after one step, the scheduler, the history, the trainer args -- loads as data too."""
    from chgnet_amd.model import CHGNet

    path = os.path.join(GOLDEN, "ref_trainer_save.pth.tar")
    want = dict(np.load(os.path.join(GOLDEN, "ref_trainer_save_state.npz")))
    state = load_torch_file(path)
    assert set(state) == {"model", "optimizer", "scheduler", "training_history", "trainer_args"}
    sd = state["model"]["state_dict"]
    assert set(sd) == set(want)
    for k, v in want.items():
        got = sd[k].numpy()
        assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v), k
    args = state["model"]["model_args"]
    assert args["n_conv"] == 4 and args["mlp_hidden_dims"] == (64, 64, 64) and args["composition_model"] == "MPtrj" and args["version"] is None
    opt = state["optimizer"]
    n_params = len(opt["param_groups"][0]["params"])
    assert len(opt["state"]) == n_params - 1 and 0 not in opt["state"]    # parameter 0 is the frozen AtomRef: never stepped, no Adam state
    assert opt["param_groups"][0]["lr"] < 1e-3                             # one scheduler step below the initial rate
    first = opt["state"][1]
    assert float(first["step"]) == 1.0 and first["exp_avg"].shape == first["exp_avg_sq"].shape
    assert state["scheduler"]["T_max"] == 50 and state["scheduler"]["last_epoch"] == 1
    assert state["training_history"]["e"]["train"] == [0.0123] and state["training_history"]["f"]["val"] == [0.0456]
    assert state["trainer_args"]["targets"] == "efsm" and state["trainer_args"]["optimizer"] == "Adam"
    model = CHGNet.from_file(path)
    got = model.state_dict()
    assert sum(int(np.prod(v.shape)) for k, v in got.items() if not k.startswith("composition_model")) >= 412_525 - 200
    for k, v in want.items():
        assert np.array_equal(np.asarray(got[k]), v), k
    # the weights differ from the unstepped model (weights_seed0.npz) by exactly one Adam step: the file was really written after it
    seed0 = dict(np.load(os.path.join(GOLDEN, "weights_seed0.npz")))
    k = "atom_conv_layers.0.twoBody_atom.mlp_core.layers.0.weight"
    step = np.abs(want[k] - seed0[k])
    assert step.max() <= 1.001e-3 and step.mean() > 5e-4


@pytest.mark.parametrize("name", ["limno2", "s16tri"])
def test_reference_graph_save_is_read_bit_for_bit(name):
    """graph/crystalgraph.py:138-156 (torch.save(self.to_dict())) -> CrystalGraph.from_file."""
    g = CrystalGraph.from_file(os.path.join(GOLDEN, f"ref_graph_{name}.pt"))
    d = np.load(os.path.join(GOLDEN, f"case_{name}.npz"))
    for k in ("atomic_number", "atom_frac_coord", "atom_graph", "neighbor_image", "directed2undirected", "undirected2directed",
              "bond_graph", "lattice"):
        got, want = np.asarray(getattr(g, k)), d[k]
        assert got.shape == want.shape and np.array_equal(got, want), k
    assert g.atom_graph_cutoff == 6.0 and g.bond_graph_cutoff == 3.0
    assert g.graph_id == name and g.mp_id == f"mp-{name}" and g.composition == "ref"
    ours, _ = load_case(name)
    assert np.array_equal(np.asarray(ours.bond_graph), np.asarray(g.bond_graph))


def test_hostile_member_next_to_reference_content(tmp_path):
    """The reference-written checkpoint re-saved with a member whose unpickling would run a command: CHGNet.from_file gives the same
    weights and nothing runs."""
    from chgnet_amd.model import CHGNet

    marker = tmp_path / "executed"
    state = torch.load(os.path.join(GOLDEN, "ref_trainer_save.pth.tar"), map_location="cpu", weights_only=False)
    state["trainer_args"]["callback"] = _Hostile(str(marker))
    state["model"]["model_args"]["note"] = _Hostile(str(marker))
    path = tmp_path / "poisoned.pth.tar"
    torch.save(state, path)
    loaded = load_torch_file(str(path))
    assert not marker.exists() and isinstance(loaded["trainer_args"]["callback"], Inert)
    loaded["model"]["model_args"].pop("note")
    model = CHGNet.from_dict(loaded["model"])
    want = dict(np.load(os.path.join(GOLDEN, "ref_trainer_save_state.npz")))
    for k, v in want.items():
        assert np.array_equal(np.asarray(model.state_dict()[k]), v), k
    assert not marker.exists()
