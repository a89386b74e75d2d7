"""The numpy model of the kernel pipeline (factorised forward + hand-derived backward, the algorithm
the HIP kernels implement) agrees with the autograd oracle -- in float64, to rounding."""

from __future__ import annotations

import numpy as np
import torch

from chgnet_amd.pack import pack_batch
from conftest import load_case
from oracle.chgnet_oracle import OracleCHGNet
from oracle.staged_ref import StagedModel


def _cat(parts):
    return np.concatenate([np.atleast_1d(p) for p in parts])


def test_staged_pipeline_equals_autograd_oracle_fp64(golden_weights, packed_weights):
    graphs = [load_case(n)[0] for n in ("limno2", "noangle", "s16tri")]
    ref = OracleCHGNet(golden_weights, dtype=torch.float64).forward(
        graphs, "efsm", return_site_energies=True, return_atom_feas=True, return_crystal_feas=True)
    out = StagedModel(packed_weights).run(pack_batch(graphs))
    assert np.abs(out["e"] - np.array(ref["e"])).max() < 1e-12
    assert np.abs(out["f"] - _cat(ref["f"])).max() < 1e-12
    assert np.abs(out["s"] - np.stack(ref["s"])).max() < 1e-11
    assert np.abs(out["m"] - _cat(ref["m"])).max() < 1e-12
    assert np.abs(out["site_energies"] - _cat(ref["site_energies"])).max() < 1e-12
    assert np.abs(out["atom_fea"] - _cat(ref["atom_fea"])).max() < 1e-12
    assert np.abs(out["crystal_fea"] - np.stack(ref["crystal_fea"])).max() < 1e-11


def test_staged_pipeline_zero_angle_batch(golden_weights, packed_weights):
    """A batch without any angle skips BondConv / AngleUpdate entirely (model.py:438,460)."""
    g = load_case("noangle")[0]
    ref = OracleCHGNet(golden_weights, dtype=torch.float64).forward([g], "efs")
    out = StagedModel(packed_weights).run(pack_batch([g]))
    assert abs(out["e"][0] - ref["e"][0]) < 1e-12
    assert np.abs(out["f"] - ref["f"][0]).max() < 1e-12
    assert np.abs(out["s"][0] - ref["s"][0]).max() < 1e-11


def _blob_from(wg: dict, pw) -> np.ndarray:
    blob = np.zeros(pw.blob.size, np.float64)
    for name, g in wg.items():
        off, shape = pw.offsets[name]
        assert tuple(np.shape(g)) == tuple(shape), (name, np.shape(g), shape)
        blob[off:off + int(np.prod(shape))] = np.asarray(g, np.float64).reshape(-1)
    return blob


def test_staged_weight_gradients_equal_autograd_fp64(golden_weights, packed_weights):
    """Stage A of the fine-tuning backward: d(sum_b c_b e_b)/d(every parameter) from the
    hand-derived reverse sweep of the factorised pipeline == torch.autograd through the oracle (fp64)."""
    from chgnet_amd.pack import unpack_weight_grads

    graphs = [load_case(n)[0] for n in ("limno2", "noangle", "s16tri")]
    cot = np.array([0.3, -1.2, 0.7])
    oracle = OracleCHGNet(golden_weights, dtype=torch.float64)
    want = oracle.parameter_gradients(graphs, lambda o: (o["e"] * torch.tensor(cot)).sum())
    out = StagedModel(packed_weights).run(pack_batch(graphs), e_cot=cot)
    got = unpack_weight_grads(_blob_from(out["wgrad"], packed_weights), packed_weights)
    assert set(got) == set(want) == set(golden_weights)
    dead = [k for k in want if k.startswith("angle_layers.2.") or k.startswith("site_wise") or k.startswith("composition_model")]
    for k, ref in want.items():
        assert got[k].shape == ref.shape, k
        if k in dead:                        # unused by the energy (model.py:442-496, 484-487, 179-182)
            assert not np.any(ref) and not np.any(got[k]), k
            continue
        scale = np.abs(ref).max()
        assert scale > 0, k
        assert np.abs(got[k] - ref).max() < 1e-10 * scale, (k, np.abs(got[k] - ref).max(), scale)


def test_stage_b_force_and_stress_loss_gradients_equal_double_backward_fp64(golden_weights, packed_weights):
    """Stage B of the fine-tuning backward: d L / d(every parameter) for a loss with energy, force AND
    stress terms -- one tangent sweep + a two-adjoint reverse sweep (oracle/staged_train.py) -- against
    torch double-backward through the oracle (the reference's create_graph=True path, model.py:517-535)."""
    from chgnet_amd.pack import unpack_weight_grads
    from oracle.staged_train import StagedTrainer

    graphs = [load_case(n)[0] for n in ("limno2", "noangle", "s16tri")]
    pb = pack_batch(graphs)
    rng = np.random.default_rng(17)
    gE, gF, gS = rng.normal(size=pb.n_struct), rng.normal(size=(pb.n_atoms, 3)), rng.normal(size=(pb.n_struct, 3, 3))
    oracle = OracleCHGNet(golden_weights, dtype=torch.float64)
    tE, tF, tS = torch.tensor(gE), torch.tensor(gF), torch.tensor(gS)
    for terms in ("efs", "f", "s", "e"):
        cE = gE if "e" in terms else None
        cF = gF if "f" in terms else None
        cS = gS if "s" in terms else None

        def loss(o, terms=terms):
            val = 0.0
            if "e" in terms:
                val = val + (o["e"] * tE).sum()
            if "f" in terms:
                val = val + (o["f"] * tF).sum()
            if "s" in terms:
                val = val + (o["s"] * tS).sum()
            return val

        want = oracle.parameter_gradients(graphs, loss, task="efs")
        out = StagedTrainer(packed_weights).run(pb, gE=cE, gF=cF, gS=cS)
        got = unpack_weight_grads(_blob_from(out["wgrad"], packed_weights), packed_weights)
        for k, ref in want.items():
            if k.startswith(("angle_layers.2.", "site_wise", "composition_model")):
                assert not np.any(got[k]), k
                continue
            scale = np.abs(ref).max()
            if scale == 0:                      # e.g. the last bias does not move forces or stress
                assert np.abs(got[k]).max() < 1e-12, (terms, k)
                continue
            assert np.abs(got[k] - ref).max() < 2e-9 * scale, (terms, k, np.abs(got[k] - ref).max(), scale)
