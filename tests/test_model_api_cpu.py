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


def test_split_and_retry_on_device_out_of_memory():
    """A chunk whose arena does not fit is halved and retried; a single structure that does not fit raises."""
    from chgnet_amd.engine import EngineOutOfMemory
    from chgnet_amd.model import _run_splitting

    calls = []

    def run(chunk):
        calls.append(len(chunk))
        if len(chunk) > 3:
            raise EngineOutOfMemory("too large")
        return [{"i": i} for i in chunk]

    out = _run_splitting(run, list(range(10)))
    assert [o["i"] for o in out] == list(range(10))          # order preserved
    assert calls[0] == 10 and max(calls[1:]) <= 5 and all(c <= 3 for c in calls if c not in (10, 5))

    def never(chunk):
        raise EngineOutOfMemory("single structure too large")

    with pytest.raises(EngineOutOfMemory):
        _run_splitting(never, [0])


def test_pack_batch_rejects_out_of_range_input():
    """Indices and atomic numbers are checked on the host: nothing out of range reaches the device
    (the reference raises IndexError from nn.Embedding(94) / index_select for the same inputs)."""
    import copy

    from chgnet_amd.pack import pack_batch
    from conftest import load_case

    g, _ = load_case("limno2")
    pack_batch([g, g])                      # sane input passes
    bad = copy.deepcopy(g)
    bad.atomic_number = bad.atomic_number.copy()
    bad.atomic_number[0] = 95               # Am: beyond the 94 embedding rows
    with pytest.raises(IndexError, match="atomic number 95"):
        pack_batch([bad])
    bad.atomic_number[0] = 0
    with pytest.raises(IndexError, match="atomic number 0"):
        pack_batch([bad])
    for field, col, val, what in (("atom_graph", 1, 8, r"atom_graph\[:, 1\]"), ("atom_graph", 0, -1, r"atom_graph\[:, 0\]"),
                                  ("bond_graph", 1, 10**6, r"bond_graph\[:, 1\]"), ("bond_graph", 4, -3, r"bond_graph\[:, 4\]")):
        bad = copy.deepcopy(g)
        arr = getattr(bad, field).copy()
        arr[5, col] = val
        setattr(bad, field, arr)
        with pytest.raises(IndexError, match=what):
            pack_batch([g, bad])            # second structure: checked against ITS OWN index range
    bad = copy.deepcopy(g)
    bad.directed2undirected = bad.directed2undirected.copy()
    bad.directed2undirected[0] = bad.directed2undirected[1] = bad.directed2undirected[2]
    with pytest.raises(ValueError, match="exactly two directed edges"):
        pack_batch([bad])


def test_cutoff_coeff_must_be_a_positive_integer(golden_weights):
    for p in (0, 2.5, -1):
        with pytest.raises(NotImplementedError, match="cutoff_coeff"):
            CHGNet(state_dict=golden_weights, cutoff_coeff=p)
    assert CHGNet(state_dict=golden_weights, cutoff_coeff=5.0).model_args["cutoff_coeff"] == 5.0


def test_forward_rejects_unknown_task(golden_weights):
    model = CHGNet(state_dict=golden_weights)
    with pytest.raises(ValueError, match="Invalid task='x'"):
        model.forward([], task="x")


def test_pipelined_chunk_loop_order_overlap_and_out_of_memory_fallback():
    """predict_*'s chunk loop: results in input order, the next chunk prepared while one is in flight, one batch alive
    at a time, a chunk that does not fit split through the synchronous path, and no batch left behind by an exception."""
    import numpy as np

    from chgnet_amd.engine import EngineOutOfMemory
    from chgnet_amd.model import _run_pipelined

    log, alive = [], []

    class Batch:
        def __init__(self, chunk):
            self.chunk = chunk
            alive.append(self)

        def free(self):
            alive.remove(self)

    def prepare(chunk):
        log.append(("prepare", chunk[0]))
        return ("prepared", tuple(chunk))

    def make_launch(too_big=(), fail_at=None):
        def launch(chunk, prepared):
            assert prepared == ("prepared", tuple(chunk))
            assert not alive                                # the previous batch was collected first
            if chunk[0] in too_big:
                raise EngineOutOfMemory("arena")
            if chunk[0] == fail_at:
                raise ValueError("isolated atoms")
            log.append(("launch", chunk[0]))
            return Batch(chunk)
        return launch

    def collect(batch):
        log.append(("collect", batch.chunk[0]))
        batch.free()
        n = len(batch.chunk)
        return {"e": np.asarray(batch.chunk, np.float32)}, np.arange(n + 1)

    def run_sync(chunk):
        log.append(("sync", chunk[0], len(chunk)))
        if len(chunk) > 1:
            raise EngineOutOfMemory("still too large")
        return [{"e": np.float32(chunk[0])}]

    chunks = [[0, 1], [2, 3], [4, 5], [6]]
    out = _run_pipelined(chunks, prepare, make_launch(), collect, run_sync)
    assert [float(o["e"]) for o in out] == [0, 1, 2, 3, 4, 5, 6] and not alive
    # chunk i+1 is prepared between launch(i) and collect(i)
    assert log.index(("prepare", 2)) > log.index(("launch", 0)) and log.index(("prepare", 2)) < log.index(("collect", 0))
    assert log.index(("launch", 2)) > log.index(("collect", 0))
    # chunks that do not fit go through the splitting path, order kept (first, middle and last position)
    for big in ({0}, {2}, {4}, {2, 4}):
        log.clear()
        out = _run_pipelined(chunks, prepare, make_launch(too_big=big), collect, run_sync)
        assert [float(o["e"]) for o in out] == [0, 1, 2, 3, 4, 5, 6] and not alive
        assert not [e for e in log if e[0] == "sync" and e[2] == 2]      # a refused chunk is not tried whole again
        assert {e[1] for e in log if e[0] == "sync"} == {x for b in big for x in (b, b + 1)}
    # a single structure that is refused raises (it would be refused again)
    with pytest.raises(EngineOutOfMemory):
        _run_pipelined(chunks, prepare, make_launch(too_big={6}), collect, run_sync)
    assert not alive
    # an exception while a sweep is in flight frees that batch
    with pytest.raises(ValueError):
        _run_pipelined(chunks, prepare, make_launch(fail_at=4), collect, run_sync)
    assert not alive
    assert _run_pipelined([], prepare, make_launch(), collect, run_sync) == []
