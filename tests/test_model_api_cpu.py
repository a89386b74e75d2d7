"""Host-side API surface that needs no GPU (reference tests/test_model.py:223-283 for the model
bookkeeping: parameter count, init message, (de)serialisation, error messages)."""

from __future__ import annotations

import numpy as np
import pytest

from chgnet_amd.model import CHGNet, random_state_dict


def test_init_message_and_param_count(capsys, golden_weights):
    model = CHGNet(state_dict=golden_weights)
    out, _ = capsys.readouterr()
    assert "CHGNet initialized with 412,525 parameters" in out          # tests/test_model.py:240-248
    assert model.n_params == 412525 and model.version is None
    assert model.graph_converter.atom_graph_cutoff == 6 and model.graph_converter.bond_graph_cutoff == 3
    assert model.is_intensive and model.n_conv == 4 and model.composition_model == "AtomRef"
    v = CHGNet(state_dict=golden_weights, version="0.3.0")
    assert v.version == "0.3.0"
    assert "CHGNet v0.3.0 initialized with 412,525 parameters" in capsys.readouterr()[0]


def test_random_init_has_reference_state_dict_shapes(golden_weights):
    sd = random_state_dict({"n_conv": 4}, seed=1)
    assert set(sd) == set(golden_weights)
    for k, v in sd.items():
        assert v.shape == golden_weights[k].shape and v.dtype == np.float32, k
    sd02 = random_state_dict({"n_conv": 4, "mlp_out_bias": True})
    assert sum(v.size for v in sd02.values()) == 412525 + 7 * 64      # 0.2.0: mlp_out biases (7 conv layers)


def test_as_dict_from_dict_round_trip(golden_weights):
    model = CHGNet(state_dict=golden_weights)
    clone = CHGNet.from_dict(model.as_dict())
    assert clone.model_args == model.model_args
    for k, v in model.state_dict().items():
        assert np.array_equal(v, clone.state_dict()[k])
    assert model.todict() == {"model_name": "CHGNet", "model_args": model.model_args}


def test_checkpoint_file_round_trip(tmp_path, golden_weights):
    """The reference's checkpoint format: torch.save({"model": {"state_dict", "model_args"}, ...})."""
    import torch

    model = CHGNet(state_dict=golden_weights)
    path = tmp_path / "ckpt.pth.tar"
    torch.save({"model": {"state_dict": {k: torch.tensor(v) for k, v in golden_weights.items()},
                          "model_args": model.model_args}, "trainer_args": {}}, path)
    loaded = CHGNet.from_file(str(path))
    assert loaded.n_params == 412525
    assert np.array_equal(loaded.state_dict()["mlp.layers.7.weight"], golden_weights["mlp.layers.7.weight"])


def test_argument_errors(golden_weights):
    model = CHGNet(state_dict=golden_weights)
    with pytest.raises(TypeError, match="must be CrystalGraph or list of CrystalGraphs"):
        model.predict_graph(42)
    with pytest.raises(ValueError, match="Invalid task='abc'. Must be one of"):
        model.predict_graph([], task="abc")
    with pytest.raises(ValueError, match="Unknown model_name='nope'"):
        CHGNet.load(model_name="nope")
    with pytest.raises(FileNotFoundError, match="CHGNET_CHECKPOINT_DIR"):
        CHGNet.load(model_name="0.3.0", checkpoint_dir="/nonexistent")
    with pytest.raises(ValueError, match="MI355X GPUs only"):
        CHGNet(state_dict=golden_weights, use_device="cpu")
    with pytest.raises(NotImplementedError):
        CHGNet(state_dict=golden_weights, read_out="attn", mlp_first=False)


def test_chunk_planner():
    from chgnet_amd.model import _plan_chunks

    # reference behaviour when the atom floor is off
    assert _plan_chunks([8] * 5, 2, 0) == [(0, 2), (2, 4), (4, 5)]
    # chunks grow to the atom floor, never below batch_size structures
    assert _plan_chunks([10] * 10, 2, 45) == [(0, 5), (5, 10)]
    assert _plan_chunks([100, 1, 1, 1], 1, 50) == [(0, 1), (1, 4)]
    assert _plan_chunks([], 16, 100) == []
    import pytest
    with pytest.raises(ValueError):
        _plan_chunks([1], 0, 0)
