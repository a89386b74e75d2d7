"""The RELEASED 0.2.0 architecture (chgnet/pretrained/0.2.0/README.md:12-37: 9 radial / 9 angular functions, a 64-64
energy head, cutoffs 5 / 3, envelope exponent 5, ``mlp_out`` biases) -- what ``CHGNet.load(model_name="0.2.0")``
(model.py:718-736) instantiates.  Fixtures: tests/golden/make_golden_v020.py (the unmodified reference, run here).

CPU part: the oracle is pinned to the reference's outputs and to its ``loss.backward()`` for this architecture; the
float64 model of the kernel pipeline (zero-padded bases, two-layer head) equals the oracle; the Python surface
(``CHGNet(**args)``, ``from_file``, ``load``) accepts it.  GPU part: the engine against the reference's outputs.
"""

from __future__ import annotations

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

V020_ARGS = dict(num_radial=9, num_angular=9, mlp_hidden_dims=[64, 64], atom_graph_cutoff=5, bond_graph_cutoff=3,
                 cutoff_coeff=5, mlp_out_bias=True)
CASES = ["limno2", "noangle", "s16tri", "s40", "li9co7o16"]
# the same tables as tests/test_oracle_golden.py (oracle vs fixture) and tests/test_gpu_parity.py (engine vs fixture)
TOL_ORACLE = {"e": 2e-6, "f": 2e-6, "s": 5e-6, "m": 3e-6, "site_energies": 3e-6, "atom_fea": 1e-5, "crystal_fea": 5e-5}
TOL_ORACLE_TL = {"e": 4e-6, "f": 4e-5, "s": 4e-4, "m": 2e-5, "site_energies": 3e-5, "atom_fea": 1e-4, "crystal_fea": 5e-4}
TOL = {"e": 5e-6, "f": 1e-5, "s": 1e-4, "m": 1e-5, "site_energies": 1e-5, "atom_fea": 5e-5, "crystal_fea": 3e-4}
KW = dict(return_site_energies=True, return_atom_feas=True, return_crystal_feas=True)


def load_case_v020(name: str):
    from chgnet_amd.graph.crystalgraph import CrystalGraph

    d = np.load(os.path.join(GOLDEN, f"case_v020_{name}.npz"))
    g = CrystalGraph(
        atomic_number=d["atomic_number"], atom_frac_coord=d["atom_frac_coord"], atom_graph=d["atom_graph"],
        atom_graph_cutoff=5, neighbor_image=d["neighbor_image"], directed2undirected=d["directed2undirected"],
        undirected2directed=d["undirected2directed"], bond_graph=d["bond_graph"], bond_graph_cutoff=3,
        lattice=d["lattice"], graph_id=name)
    return g, d


@pytest.fixture(scope="module")
def weights():
    return dict(np.load(os.path.join(GOLDEN, "weights_v020.npz")))


@pytest.fixture(scope="module")
def weights_tl():
    return dict(np.load(os.path.join(GOLDEN, "weights_v020_trained_like.npz")))


def _oracle(w, dtype=torch.float32):
    from oracle.chgnet_oracle import OracleCHGNet

    return OracleCHGNet(w, atom_graph_cutoff=5.0, bond_graph_cutoff=3.0, cutoff_coeff=5, dtype=dtype)


# ---- CPU: the checker is pinned -------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_reference_on_the_released_020_architecture(weights, weights_tl, name):
    torch.set_num_threads(1)
    g, d = load_case_v020(name)
    for w, prefix, tol in ((weights, "out_", TOL_ORACLE), (weights_tl, "tl_out_", TOL_ORACLE_TL)):
        out = _oracle(w).predict_graph(g, "efsm", **KW)
        for key, t in tol.items():
            ref = d[prefix + key]
            assert out[key].shape == ref.shape
            err = float(np.abs(out[key] - ref).max()) if ref.size else 0.0
            assert err <= t, f"{name}:{prefix}{key} {err:.2e}"


def test_oracle_parameter_gradients_match_the_reference_backward_on_020(weights):
    """All 141 tensors (the 136 of 0.3.0 minus one head layer plus seven ``mlp_out`` biases) of the reference's own
    ``loss.backward()`` in train mode (trainer.py:399-411) on the five-case batch."""
    d = np.load(os.path.join(GOLDEN, "grad_v020_five.npz"))
    want = {k[len("grad/"):]: d[k] for k in d.files if k.startswith("grad/")}
    graphs = [load_case_v020(str(n))[0] for n in d["order"]]
    torch.set_num_threads(4)
    t = lambda a: torch.tensor(np.asarray(a, np.float64))  # noqa: E731
    ce, gm, gf, gs = d["cot_e"], d["cot_m"], d["cot_f"], d["cot_s"]
    got = _oracle(weights, torch.float64).parameter_gradients(
        graphs, lambda o: (o["e"] * t(ce)).sum() + (o["m"] * t(gm)).sum() + (o["f"] * t(gf)).sum() + (o["s"] * t(gs)).sum(), task="efsm")
    assert set(got) == set(want) and len(want) == 141
    bad = {}
    for k, ref in want.items():
        scale, err = float(np.abs(ref).max()), float(np.abs(got[k] - ref).max())
        if scale == 0.0:
            assert err == 0.0, k
        elif err / scale > 1e-4:
            bad[k] = err / scale
    assert not bad, "; ".join(f"{k} {v:.1e}" for k, v in sorted(bad.items(), key=lambda kv: -kv[1])[:8])


def test_pipeline_model_with_padded_bases_and_two_layer_head_equals_the_oracle_fp64(weights):
    """What the kernels compute for this architecture -- 31-wide bases with the tail zero-padded (``pack.pad_radial`` /
    ``pad_angular``), the head's third layer skipped, ``q_bias`` for the bonds outside the bond graph -- is the
    reference's function: float64 pipeline model == float64 autograd oracle."""
    from chgnet_amd.pack import pack_batch, pack_weights
    from oracle.staged_ref import StagedModel

    pw = pack_weights(weights, V020_ARGS)
    assert (pw.n_mlp_hidden, pw.num_radial, pw.num_angular, pw.cutoff_coeff, pw.atom_graph_cutoff) == (2, 9, 9, 5, 5.0)
    graphs = [load_case_v020(n)[0] for n in ("limno2", "noangle", "s16tri")]
    ref = _oracle(weights, torch.float64).forward(graphs, "efsm", **KW)
    out = StagedModel(pw).run(pack_batch(graphs))
    cat = lambda parts: np.concatenate([np.atleast_1d(p) for p in parts])  # noqa: E731
    assert np.abs(out["e"] - np.array(ref["e"])).max() < 1e-12
    assert np.abs(out["f"] - cat(ref["f"])).max() < 1e-12
    assert np.abs(out["s"] - np.stack(ref["s"])).max() < 1e-11
    assert np.abs(out["m"] - cat(ref["m"])).max() < 1e-12
    assert np.abs(out["site_energies"] - cat(ref["site_energies"])).max() < 1e-12
    assert np.abs(out["atom_fea"] - cat(ref["atom_fea"])).max() < 1e-12


def test_padding_round_trip_and_rejections():
    from chgnet_amd.pack import check_model_args, pad_angular, unpad_angular

    w = np.arange(64 * 9, dtype=np.float32).reshape(64, 9)
    p = pad_angular(w)
    assert p.shape == (64, 31) and np.array_equal(unpad_angular(p, 9), w)
    assert np.array_equal(p[:, 5:16], np.zeros((64, 11))) and np.array_equal(p[:, 16:20], w[:, 5:9]) and not p[:, 20:].any()
    check_model_args(V020_ARGS)
    for bad in (dict(num_radial=32), dict(num_angular=8), dict(num_angular=33), dict(mlp_hidden_dims=[64]), dict(mlp_hidden_dims=[32, 32]),
                dict(mlp_hidden_dims=64)):
        with pytest.raises(NotImplementedError):
            check_model_args(bad)


def test_python_surface_accepts_the_released_020_architecture(weights, tmp_path, capsys):
    """``CHGNet(**README arguments)`` has the reference's 403,126 parameters; a checkpoint in the reference's file
    format under the released file name loads through ``CHGNet.load(model_name="0.2.0")`` (model.py:718-736: it passes
    ``mlp_out_bias=True``, ``version="0.2.0"``)."""
    from chgnet_amd import CHGNet

    model = CHGNet(**V020_ARGS)
    assert model.n_params == 403126
    assert "CHGNet initialized with 403,126 parameters" in capsys.readouterr().out
    assert set(model.state_dict()) == set(weights) and all(model.state_dict()[k].shape == v.shape for k, v in weights.items())
    root = tmp_path / "0.2.0"
    root.mkdir()
    args = {k: v for k, v in V020_ARGS.items() if k != "mlp_out_bias"}      # the released file's model_args predate the switch
    torch.save({"model": {"state_dict": {k: torch.tensor(v) for k, v in weights.items()}, "model_args": args}},
               str(root / "chgnet_0.2.0_e30f77s348m32.pth.tar"))
    loaded = CHGNet.load(model_name="0.2.0", checkpoint_dir=str(tmp_path), verbose=False)
    assert loaded.version == "0.2.0" and loaded.n_params == 403126
    assert "CHGNet v0.2.0 initialized with 403,126 parameters" in capsys.readouterr().out
    assert loaded.graph_converter.atom_graph_cutoff == 5 and loaded._weights.n_mlp_hidden == 2
    assert all(np.array_equal(loaded.state_dict()[k], v) for k, v in weights.items())


# ---- GPU: the engine against the reference's outputs -------------------------------------------------------------------
@pytest.fixture(scope="module")
def model_v020(weights):
    from chgnet_amd import CHGNet

    m = CHGNet(state_dict=weights, **V020_ARGS)
    yield m
    if m._engine is not None:
        m._engine.close()


@pytest.fixture(scope="module")
def model_v020_tl(weights_tl):
    from chgnet_amd import CHGNet

    m = CHGNet(state_dict=weights_tl, **V020_ARGS)
    yield m
    if m._engine is not None:
        m._engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_engine_matches_the_reference_on_the_released_020_architecture(model_v020, name):
    g, d = load_case_v020(name)
    out = model_v020.predict_graph(g, task="efsm", **KW)
    for key, tol in TOL.items():
        ref = d["out_" + key]
        assert out[key].shape == ref.shape and out[key].dtype == np.float32
        assert np.isfinite(out[key]).all(), key
        err = float(np.abs(out[key] - ref).max()) if ref.size else 0.0
        assert err < tol, f"{name}:{key} max|d|={err:.3e} tol={tol:.1e}"


@pytest.mark.gpu
def test_engine_matches_the_reference_020_trained_like_weights_in_one_batch(model_v020_tl):
    """Trained-checkpoint magnitudes, all five cases in ONE device batch (the zero-angle cell rides with cells that have
    angles, so it receives the BondConv biases like in the reference: model.py:459 tests the batch, not the structure --
    the fixture values are single-structure calls, so the zero-angle case is compared on its own)."""
    names = [n for n in CASES if n != "noangle"]
    outs = model_v020_tl.predict_graph([load_case_v020(n)[0] for n in names], task="efsm", **KW)
    for n, out in zip(names, outs):
        d = load_case_v020(n)[1]
        scale = {"e": 1.0, "f": max(1.0, float(np.abs(d["tl_out_f"]).max())), "s": max(1.0, float(np.abs(d["tl_out_s"]).max()))}
        for key, tol in TOL.items():
            ref = d["tl_out_" + key]
            err = float(np.abs(out[key] - ref).max())
            assert err < 4 * tol * scale.get(key, 1.0), f"{n}:{key} max|d|={err:.3e}"
    g, d = load_case_v020("noangle")
    out = model_v020_tl.predict_graph(g, task="efsm", **KW)
    for key, tol in TOL.items():
        assert float(np.abs(out[key] - d["tl_out_" + key]).max()) < 4 * tol, key


@pytest.mark.gpu
def test_engine_020_predict_structure_builds_the_5A_graph_on_the_device(model_v020):
    """Structures in, the 5 A / 3 A graph built on the device (the converter's cutoffs come from the model arguments)."""
    from chgnet_amd.graph.structure import Lattice, Structure

    d = load_case_v020("s16tri")[1]
    s = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"])
    out = model_v020.predict_structure(s, task="efsm")
    for key in ("e", "f", "s", "m"):
        assert float(np.abs(out[key] - d["out_" + key]).max()) < TOL[key], key


@pytest.mark.gpu
def test_engine_020_full_size_batch_vs_oracle_and_singles(model_v020_tl, weights_tl):
    """BASELINE's full batch size on the released architecture: 1024 perturbed LiMnO2 5x1x1 cells (40 atoms) through
    ``predict_structure`` (graphs built on the device at the model's 5 A / 3 A cutoffs, one device batch), a sample of them against the
    fp32 oracle on host-built graphs, and against the same structures predicted one at a time (results do not depend on the batch)."""
    import torch

    import bench
    from chgnet_amd import CrystalGraphConverter
    from oracle.chgnet_oracle import OracleCHGNet

    torch.set_num_threads(8)
    structs = bench.workload_structures(1024, 9100)
    outs = model_v020_tl.predict_structure(structs, task="efs", batch_size=1024)
    assert len(outs) == 1024 and all(np.isfinite(o["e"]) and np.isfinite(o["f"]).all() and np.isfinite(o["s"]).all() for o in outs)
    pick = [0, 1, 257, 511, 768, 1023]
    conv = CrystalGraphConverter(atom_graph_cutoff=5, bond_graph_cutoff=3)
    ref = OracleCHGNet(weights_tl, atom_graph_cutoff=5.0, bond_graph_cutoff=3.0, cutoff_coeff=5).predict_graph([conv(structs[i]) for i in pick], "efs", batch_size=len(pick))
    for i, r in zip(pick, ref):
        fs, ss = max(1.0, float(np.abs(r["f"]).max())), max(1.0, float(np.abs(r["s"]).max()))
        assert abs(float(outs[i]["e"]) - float(r["e"])) < 4 * TOL["e"] * max(1.0, abs(float(r["e"]))), i
        assert float(np.abs(outs[i]["f"] - r["f"]).max()) < 4 * TOL["f"] * fs, i
        assert float(np.abs(outs[i]["s"] - r["s"]).max()) < 4 * TOL["s"] * ss, i
        one = model_v020_tl.predict_structure(structs[i], task="efs")
        assert abs(float(one["e"]) - float(outs[i]["e"])) < 2e-6 * max(1.0, abs(float(one["e"])))
        assert float(np.abs(one["f"] - outs[i]["f"]).max()) < 2e-5 * fs and float(np.abs(one["s"] - outs[i]["s"]).max()) < 2e-4 * ss


# ---- parameter gradients incl. the mlp_out biases (CPU: the float64 models of the two training sweeps) -----------------
def _blob_from(wg: dict, pw) -> np.ndarray:
    blob = np.zeros(pw.blob.size, np.float64)
    for name, g in wg.items():
        off, shape = pw.offsets[name]
        assert tuple(np.shape(g)) == tuple(shape), (name, np.shape(g), shape)
        blob[off:off + int(np.prod(shape))] = np.asarray(g, np.float64).reshape(-1)
    return blob


DEAD = ("angle_layers.2.", "composition_model")


def test_sweep_models_give_the_mlp_out_bias_gradients_fp64(weights):
    """The reference adds a BondConv's ``mlp_out`` bias to EVERY bond (layers.py:252-258 aggregates over all of them); the
    engine keeps bonds outside the bond graph at their embedding + a constant shift (``q_bias`` / ``q_shift``).  The
    gradient of such a bias is therefore the column sum of dL/d(bond features) over ALL bonds, and W_bond's gradient
    gains (sum over non-node bonds of dL/dQ) x shift.  First-order sweep (oracle/staged_ref.py) and the two-adjoint sweep
    for force / stress terms (oracle/staged_train.py) against torch autograd / double backward, float64, all 141 tensors."""
    from chgnet_amd.pack import pack_batch, pack_weights, unpack_weight_grads
    from oracle.staged_ref import StagedModel
    from oracle.staged_train import StagedTrainer

    pw = pack_weights(weights, V020_ARGS)
    assert pw.mlp_out_bias
    graphs = [load_case_v020(n)[0] for n in ("limno2", "noangle", "s16tri")]
    pb = pack_batch(graphs)
    oracle = _oracle(weights, torch.float64)
    rng = np.random.default_rng(41)
    gE, gF, gS = rng.normal(size=pb.n_struct), rng.normal(size=(pb.n_atoms, 3)), rng.normal(size=(pb.n_struct, 3, 3))
    tE, tF, tS = torch.tensor(gE), torch.tensor(gF), torch.tensor(gS)

    def check(got, want, tol, what):
        assert set(got) == set(want) == set(weights) and len(want) == 141
        for k, ref in want.items():
            if k.startswith(DEAD) or (what == "e" and k.startswith("site_wise")):
                assert not np.any(got[k]), k
                continue
            scale = np.abs(ref).max()
            if scale == 0:
                assert np.abs(got[k]).max() < 1e-12, (what, k)
                continue
            assert np.abs(got[k] - ref).max() < tol * scale, (what, k, np.abs(got[k] - ref).max(), scale)

    want = oracle.parameter_gradients(graphs, lambda o: (o["e"] * tE).sum())
    out = StagedModel(pw).run(pb, e_cot=gE)
    check(unpack_weight_grads(_blob_from(out["wgrad"], pw), pw), want, 1e-10, "e")
    assert np.abs(want["bond_conv_layers.0.mlp_out.layers.1.bias"]).max() > 0

    want = oracle.parameter_gradients(graphs, lambda o: (o["e"] * tE).sum() + (o["f"] * tF).sum() + (o["s"] * tS).sum(), task="efs")
    out = StagedTrainer(pw).run(pb, gE=gE, gF=gF, gS=gS)
    got = unpack_weight_grads(_blob_from(out["wgrad"], pw), pw)
    for k in list(want):
        if k.startswith("site_wise"):
            assert not np.any(got[k])
            want.pop(k), got.pop(k)
    assert len(want) == 139
    for k, ref in want.items():
        if k.startswith(DEAD):
            continue
        scale = np.abs(ref).max()
        assert scale > 0 and np.abs(got[k] - ref).max() < 2e-9 * scale, (k, np.abs(got[k] - ref).max(), scale)


@pytest.mark.gpu
def test_engine_loss_gradients_match_the_reference_backward_on_020(model_v020):
    """``chg_backward`` on the released 0.2.0 architecture: the full E + F + S + M loss against ``p.grad`` of the UNMODIFIED
    reference after ``loss.backward()`` in train mode (tests/golden/make_golden_v020.py; trainer.py:399-411) -- all 141
    tensors, the seven ``mlp_out`` biases and the padded-basis embeddings among them (VERDICT r04 'missing 5').  Tolerance
    as for the 0.3.0 fixtures (tests/test_gpu_train.py): the fixture is fp32, 3.2e-4 of each tensor's largest entry."""
    d = np.load(os.path.join(GOLDEN, "grad_v020_five.npz"))
    want = {k[len("grad/"):]: d[k] for k in d.files if k.startswith("grad/")}
    graphs = [load_case_v020(str(n))[0] for n in d["order"]]
    try:
        model_v020.forward(graphs, task="efsm")
        got = model_v020.backward(d["cot_e"], d["cot_m"], d["cot_f"], d["cot_s"])
    finally:
        model_v020.release_forward_state()
    assert set(got) == set(want) and len(want) == 141
    msgs = []
    for k, ref in want.items():
        assert got[k].shape == ref.shape and got[k].dtype == np.float32, k
        if k.startswith(("angle_layers.2.", "composition_model")) and not np.any(ref):
            assert not np.any(got[k]), k
            continue
        scale, err = float(np.abs(ref).max()), float(np.abs(got[k] - ref).max())
        if not np.isfinite(got[k]).all() or not err <= 3.2e-4 * scale:
            msgs.append(f"{k}: {err:.3e} / {scale:.3e} = {err / max(scale, 1e-300):.1e}")
    assert not msgs, "; ".join(msgs)


@pytest.mark.gpu
def test_engine_energy_loss_gradients_020_vs_autograd_fp64(model_v020, weights):
    """First-order sweep alone (energy + magmom cotangents) on a batch WITH a zero-angle structure in it, against torch
    autograd through the float64 oracle: 1e-4 relative per tensor."""
    graphs = [load_case_v020(n)[0] for n in ("limno2", "noangle", "s16tri")]
    rng = np.random.default_rng(43)
    ce = rng.normal(size=3)
    t = torch.tensor(ce)
    torch.set_num_threads(8)
    want = _oracle(weights, torch.float64).parameter_gradients(graphs, lambda o: (o["e"] * t).sum())
    try:
        model_v020.forward(graphs, task="e")
        got = model_v020.backward(ce.astype(np.float32))
    finally:
        model_v020.release_forward_state()
    msgs = []
    for k, ref in want.items():
        if k.startswith(("angle_layers.2.", "composition_model", "site_wise")):
            assert not np.any(got[k]), k
            continue
        scale, err = float(np.abs(ref).max()), float(np.abs(got[k] - ref).max())
        if not err <= 1e-4 * scale:
            msgs.append(f"{k}: {err:.3e} / {scale:.3e}")
    assert not msgs, "; ".join(msgs)


@pytest.mark.gpu
def test_train_step_runs_on_the_020_architecture(weights):
    """``TrainStep`` (forward -> CombinedLoss -> backward -> Adam -> weights back on the engine) accepts a model with
    ``mlp_out`` biases now and lowers the loss; the biases move."""
    from chgnet_amd import CHGNet
    from chgnet_amd.trainer import TrainStep

    model = CHGNet(state_dict=weights, **V020_ARGS)
    try:
        graphs = [load_case_v020(n)[0] for n in ("limno2", "s16tri", "s40")]
        ref = model.predict_graph(graphs, task="efsm")
        # labels a small COHERENT shift away from the predictions (like tests/test_gpu_train.py::test_train_step_with_force_and_stress_terms)
        targets = {"e": np.array([r["e"] + 0.05 for r in ref]), "f": [r["f"] * 1.5 for r in ref],
                   "s": [r["s"] + 0.1 * np.eye(3, dtype=np.float32) for r in ref], "m": [r["m"] + 0.1 for r in ref]}
        step = TrainStep(model, targets="efsm", learning_rate=2e-4)
        before = {k: v.copy() for k, v in model.state_dict().items() if k.endswith("mlp_out.layers.1.bias")}
        losses = [step(graphs, targets)["loss"] for _ in range(10)]
        assert np.isfinite(losses).all() and losses[-1] < 0.7 * losses[0], losses
        assert all(np.abs(model.state_dict()[k] - v).max() > 0 for k, v in before.items())
    finally:
        model.release_forward_state()
        if model._engine is not None:
            model._engine.close()
