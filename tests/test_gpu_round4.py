"""Round-4 GPU tests: the atom schedule of the per-atom angle adjoints (k_win_groups / k_win_schedule), the golden vectors through
the per-atom adjoints, the all-reduced gradient blob.  All through the C-ABI (ctypes).

The per-atom kernels normally need a batch with a few atoms per wave; CHGNET_WIN_MIN_ATOMS_PER_WAVE=0 (read at every upload) sends
the small golden cases through them as well, so that every golden vector of the reference pins that path too (before this round
only the 384- and 1024-structure batches did)."""

from __future__ import annotations

import os

import numpy as np
import pytest

from conftest import GOLDEN, load_case

pytestmark = pytest.mark.gpu

TOL = {"e": 5e-6, "f": 1e-5, "s": 1e-4, "m": 1e-5, "site_energies": 1e-5, "atom_fea": 5e-5, "crystal_fea": 3e-4}
CASES = ["limno2", "s40", "s16tri", "li9co7o16"]     # "noangle" has no angles: nothing for the angle kernels to do
N_STRUCT = 384


@pytest.fixture()
def force_per_atom():
    old = os.environ.get("CHGNET_WIN_MIN_ATOMS_PER_WAVE")
    os.environ["CHGNET_WIN_MIN_ATOMS_PER_WAVE"] = "0"
    yield
    if old is None:
        del os.environ["CHGNET_WIN_MIN_ATOMS_PER_WAVE"]
    else:
        os.environ["CHGNET_WIN_MIN_ATOMS_PER_WAVE"] = old


def _split(res, packed):
    off = packed.atom_off
    outs = []
    for i in range(packed.n_struct):
        sl = slice(off[i], off[i + 1])
        d = {"e": res["e"][i]}
        for k in ("f", "m", "site_energies", "atom_fea"):
            d[k] = res[k][sl]
        for k in ("s", "crystal_fea"):
            d[k] = res[k][i]
        outs.append(d)
    return outs


@pytest.fixture(scope="module")
def big_batch():
    import bench
    from chgnet_amd.pack import pack_batch

    return pack_batch(bench.build_workload(N_STRUCT, 7000))


def test_atom_schedule_is_a_balanced_partition(hip_engine, big_batch):
    """k_win_groups / k_win_schedule: every atom with angles is in exactly one wave's list, the lists of a 64-wave group cover
    exactly the group's atom range, and the tile counts of the waves of a group differ by at most one atom's worth."""
    pb = big_batch
    batch = hip_engine.upload(pb)
    try:
        N = pb.n_atoms
        flag = hip_engine.debug_fetch_i32(batch, "win_flag", 4)
        assert flag[0] == 1
        grid = int(flag[3])
        assert grid % 64 == 0 and grid >= 64
        na = hip_engine.debug_fetch_i32(batch, "win_na", N + 1)[:N]
        head = hip_engine.debug_fetch_i32(batch, "win_wave_head", grid * 8)
        nxt = hip_engine.debug_fetch_i32(batch, "win_next_atom", N)
        xatom = hip_engine.debug_fetch_i32(batch, "win_xatom", grid // 8 + 1)
        assert xatom[0] == 0 and xatom[-1] == N and np.all(np.diff(xatom) >= 0)
        seen = np.zeros(N, dtype=np.int32)
        tiles = (na * (na - 1) + 15) // 16 + 1
        spx = grid // 64
        worst = 0
        for gi in range(grid // 8):
            x, sub = divmod(gi, spx)
            loads = []
            for slot in range(64):
                b = (((sub << 3) + slot // 8) << 3) + x
                c = int(head[b * 8 + slot % 8])
                load, prev = 0, -1
                while c >= 0:
                    assert xatom[gi] <= c < xatom[gi + 1] and c > prev and na[c] >= 2
                    seen[c] += 1
                    load += int(tiles[c])
                    prev, c = c, int(nxt[c])
                loads.append(load)
            big = int(tiles[xatom[gi]:xatom[gi + 1]].max()) if xatom[gi + 1] > xatom[gi] else 0
            worst = max(worst, max(loads) - min(loads) - big)
        assert np.array_equal(seen, (na >= 2).astype(np.int32))
        assert worst <= 0, f"loads inside a group differ by more than one atom ({worst} tiles over)"
    finally:
        batch.free()


@pytest.mark.parametrize("name", CASES)
def test_per_atom_kernels_match_reference_golden(hip_engine, force_per_atom, name):
    """The reference's golden outputs (tests/golden/case_*.npz) through the per-atom adjoint kernels (li9co7o16: up to 18 short
    bonds per atom, more than the adjoints' private rows: the direct-atomics fallback of the ranks past them)."""
    g, d = load_case(name)
    batch = hip_engine.upload([g])
    try:
        flag = hip_engine.debug_fetch_i32(batch, "win_flag", 4)
        assert flag[0] == 1, "golden graphs have the canonical angle structure"
        hip_engine.predict(batch, "efsm")
        res = hip_engine.download(batch, "efsm", site_energies=True, atom_feas=True, crystal_feas=True)
        out = _split(res, batch.packed)[0]
    finally:
        batch.free()
    for key, tol in TOL.items():
        ref = d["out_" + key]
        err = float(np.abs(out[key] - ref).max())
        assert np.isfinite(out[key]).all(), key
        assert err < tol, f"{name}:{key} max|d|={err:.3e} tol={tol:.1e}"


def test_per_atom_kernels_on_the_mixed_batch(hip_engine, force_per_atom):
    """Mixed sizes in one batch, a structure without angles in the middle (tests/golden/batch_mixed.npz)."""
    d = np.load(os.path.join(GOLDEN, "batch_mixed.npz"))
    order = [str(x) for x in d["order"]]
    graphs = [load_case(n)[0] for n in order]
    batch = hip_engine.upload(graphs)
    try:
        assert hip_engine.debug_fetch_i32(batch, "win_flag", 4)[0] == 1
        hip_engine.predict(batch, "efsm")
        res = hip_engine.download(batch, "efsm", site_energies=True, atom_feas=True, crystal_feas=True)
        outs = _split(res, batch.packed)
    finally:
        batch.free()
    for n, o in zip(order, outs):
        for key, tol in TOL.items():
            err = float(np.abs(o[key] - d[f"{n}_{key}"]).max()) if d[f"{n}_{key}"].size else 0.0
            assert err < tol, f"{n}:{key} {err:.3e}"


def test_per_atom_adjoint_intermediates_match_the_pipeline_model(hip_engine, packed_weights, force_per_atom):
    """Bond / angle features after every layer and the adjoints the per-atom kernels leave behind vs the float64 pipeline model."""
    from oracle.staged_ref import StagedModel

    graphs = [load_case(n)[0] for n in ("limno2", "s16tri", "li9co7o16")]
    batch = hip_engine.upload(graphs)
    try:
        assert hip_engine.debug_fetch_i32(batch, "win_flag", 4)[0] == 1
        hip_engine.predict(batch, "efsm")
        pb = batch.packed
        buf = StagedModel(packed_weights).run(pb)["buffers"]
        A, Eb = pb.n_angles, pb.n_bnodes
        msgs = []
        for name, shape, tol in [("hbc1", (Eb, 64), 5e-5), ("ang1", (A, 64), 3e-4), ("hbc2", (Eb, 64), 5e-5), ("ang2", (A, 64), 3e-4),
                                 ("hbc3", (Eb, 64), 5e-5), ("Gang", (A, 64), 1e-5), ("Gwbgc", (Eb, 64), 1e-5)]:
            got, want = hip_engine.debug_fetch(batch, name, shape), buf[name]
            err, scale = float(np.abs(got - want).max()), max(1.0, float(np.abs(want).max()))
            if not err < tol * scale:
                msgs.append(f"{name}: max|d|={err:.3e} (scale {scale:.2e}, tol {tol:.1e})")
        assert not msgs, "; ".join(msgs)
    finally:
        batch.free()


def test_atoms_with_more_bonds_than_private_rows(hip_engine, golden_weights, force_per_atom):
    """Dense cells: 14 short bonds per atom (bcc-like, 3 A) and 26 (bond-graph cutoff raised to 4.2 A) -- more than the 13 / 14
    private second-bond rows of the per-atom adjoints: the ranks past them leave as direct row atomics.  Both against the CPU oracle
    on the same graphs."""
    import torch

    from chgnet_amd import CrystalGraphConverter, Structure
    from chgnet_amd.graph.structure import Lattice
    from oracle.chgnet_oracle import OracleCHGNet

    torch.set_num_threads(8)
    oracle = OracleCHGNet(golden_weights)
    rng = np.random.default_rng(5)
    # bcc-like cell, a = 2.9 A: 8 neighbours at 2.51 A + 6 at 2.9 A = 14 short bonds per atom at the 3 A cutoff, 12 more at 4.1 A
    frac = np.array([[i, j, k] for i in range(2) for j in range(2) for k in range(2)], dtype=float) / 2.0
    frac = np.concatenate([frac, frac + 0.25]) + rng.normal(0, 0.004, (16, 3))
    s = Structure(Lattice(np.eye(3) * 5.8), np.full(16, 26), frac)
    for cutoff, want_n in ((3.0, 14), (4.2, 26)):
        g = CrystalGraphConverter(atom_graph_cutoff=6, bond_graph_cutoff=cutoff)(s)
        batch = hip_engine.upload([g])
        try:
            flag = hip_engine.debug_fetch_i32(batch, "win_flag", 4)
            na = hip_engine.debug_fetch_i32(batch, "win_na", 17)[:16]
            assert flag[0] == 1 and (na == want_n).all(), (flag, na)
            hip_engine.predict(batch, "efs")
            res = hip_engine.download(batch, "efs")
        finally:
            batch.free()
        ref = oracle.predict_graph(g, "efs")
        assert abs(res["e"][0] - ref["e"]) < 5e-6 and np.abs(res["f"] - ref["f"]).max() < 2e-5 and np.abs(res["s"][0] - ref["s"]).max() < 2e-4


def test_all_reduced_gradient_carries_the_last_bias_and_survives_a_weight_update(packed_weights, trained_like_weights):
    """chg_backward_allreduce on a one-rank communicator, twice, with a weight update in between: (i) the slot of the readout's last
    bias -- formed on the host, sum_b cot_b n_b -- is in the blob the COLLECTIVE sees (it used to be patched into the host copy after
    the all-reduce, so with several ranks every rank kept its local value: ADVICE r03); (ii) a second call gives the same blob (fp32
    reassociation of the atomics only); (iii) a following chg_engine_update_weights is ordered behind the collective on the engine's
    stream: the gradient after the update is the one a fresh engine with the new weights gives."""
    from chgnet_amd.distributed import RcclComm
    from chgnet_amd.engine import Engine
    from chgnet_amd.pack import pack_weights

    graphs = [load_case(n)[0] for n in ("limno2", "noangle", "s16tri")]
    n_atoms = np.array([len(g.atomic_number) for g in graphs], np.float64)
    cot = np.array([0.5, -1.0, 2.0], np.float32)
    w2 = pack_weights(trained_like_weights)
    b3 = packed_weights.offsets["mlp_b3"][0]
    comm = RcclComm(0, 1, 0)
    eng = Engine(packed_weights, 0)
    try:
        batch = eng.upload(graphs)
        eng.predict(batch, "e")
        plain = eng.backward(batch, cot)
        comm.world = 2                      # route through chg_backward_allreduce (the communicator still has one rank)
        try:
            first = eng.backward(batch, cot, comm=comm)
            second = eng.backward(batch, cot, comm=comm)
            eng.update_weights(w2)
            eng.predict(batch, "e")
            after = eng.backward(batch, cot, comm=comm)
        finally:
            comm.world = 1
        batch.free()
    finally:
        eng.close()
        comm.close()
    want_b3 = float((cot.astype(np.float64) / n_atoms * n_atoms).sum())     # intensive model: cot_b / n_b per site, n_b sites
    for blob in (plain, first, second, after):
        assert abs(blob[b3] - want_b3) < 1e-6, (blob[b3], want_b3)
    scale = float(np.abs(plain).max())
    assert scale > 0 and np.abs(plain - first).max() <= 1e-4 * scale and np.abs(first - second).max() <= 1e-4 * scale
    fresh = Engine(w2, 0)
    try:
        batch = fresh.upload(graphs)
        fresh.predict(batch, "e")
        want = fresh.backward(batch, cot)
        batch.free()
    finally:
        fresh.close()
    assert np.abs(want - plain).max() > 1e-3 * scale                        # the two weight sets do give different gradients
    assert np.abs(after - want).max() <= 1e-4 * float(np.abs(want).max())
