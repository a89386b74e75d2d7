"""Round-3 GPU tests: the per-atom AngleUpdate adjoint (csrc/kernels_angle_w.h) and the multi-GPU entry points on one rank.

All through the C-ABI (ctypes).  The per-atom path needs a batch that gives every wave a few atoms (>= ~12k atoms on
256 CUs); smaller batches -- every other GPU test -- run the plain adjoint."""

from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_STRUCT = 384          # x 40 atoms = 15,360 atoms: enough for the per-atom order on a 256-CU device


def _reordered(pb, perm):
    from chgnet_amd.pack import PackedBatch

    arr = dict(pb.arrays)
    for k in ("a_ctr", "a_b1", "a_d1", "a_b2", "a_d2", "a_b1c", "a_b2c"):
        arr[k] = np.ascontiguousarray(arr[k][perm])
    return PackedBatch(pb.n_struct, pb.n_atoms, pb.n_directed, pb.n_undirected, pb.n_angles, pb.n_bnodes, arr)


@pytest.fixture(scope="module")
def big_batch():
    import bench
    from chgnet_amd.pack import pack_batch

    return pack_batch(bench.build_workload(N_STRUCT, 7000))


def test_per_atom_index_is_a_permutation_with_the_block_structure(hip_engine, big_batch):
    """k_win_*: centre-major row order built on the device -- a bijection of the angles, rows of one atom contiguous and
    complete (n (n - 1) rows for n short bonds), (atom, bond) pair indices consistent with the compact bond indices."""
    pb = big_batch
    batch = hip_engine.upload(pb)
    try:
        A, N = pb.n_angles, pb.n_atoms
        flag = hip_engine.debug_fetch_i32(batch, "win_flag", 4)
        assert flag[0] == 1, "canonical reference-ordered graphs must take the per-atom path at this size"
        q_a = hip_engine.debug_fetch_i32(batch, "win_q_a", A)
        q_ctr = hip_engine.debug_fetch_i32(batch, "win_q_ctr", A)
        na = hip_engine.debug_fetch_i32(batch, "win_na", N + 1)[:N]
        aoff = hip_engine.debug_fetch_i32(batch, "win_aoff", N + 1)
        ab1 = hip_engine.debug_fetch_i32(batch, "win_q_ab1", A)
        ab2 = hip_engine.debug_fetch_i32(batch, "win_q_ab2", A)
        assert np.array_equal(np.sort(q_a), np.arange(A))
        assert np.array_equal(q_ctr, pb.a_ctr[q_a]) and np.all(np.diff(q_ctr) >= 0)
        assert np.array_equal(np.diff(aoff), na * (na - 1)) and aoff[-1] == A
        assert np.array_equal(np.bincount(pb.a_ctr, minlength=N), na * (na - 1))
        boff = np.concatenate([[0], np.cumsum(na)])
        r1, r2 = ab1 - boff[q_ctr], ab2 - boff[q_ctr]
        assert r1.min() >= 0 and (r1 < na[q_ctr]).all() and r2.min() >= 0 and (r2 < na[q_ctr]).all() and (r1 != r2).all()
        # one pair index <-> one compact bond per atom
        for ab, bc in ((ab1, pb.a_b1c[q_a]), (ab2, pb.a_b2c[q_a])):
            first = {}
            sample = np.random.default_rng(0).choice(A, 20000, replace=False)
            for i in sample:
                assert first.setdefault(int(ab[i]), int(bc[i])) == int(bc[i])
    finally:
        batch.free()


@pytest.mark.parametrize("which", ["seed0", "trained_like"])
def test_per_atom_adjoint_equals_the_plain_adjoint(hip_engine, trained_like_weights, big_batch, which):
    """The same batch with its angle rows shuffled inside every structure has no group structure: the device clears the flag
    and the plain adjoints run (BondConv then in the f32 matrix form, per atom in split precision).  Forces / stress of the two
    paths agree to fp32 rounding -- also at trained-checkpoint magnitudes (|F| up to several eV/A, saturated gates)."""
    from chgnet_amd.engine import Engine
    from chgnet_amd.pack import pack_weights

    pb = big_batch
    eng = hip_engine if which == "seed0" else Engine(pack_weights(trained_like_weights), 0)
    tol = {"e": 2e-6, "f": 2e-6, "s": 2e-5} if which == "seed0" else {"e": 1e-5, "f": 1e-4, "s": 1e-3}
    rng = np.random.default_rng(3)
    off = pb.ang_off
    perm = np.concatenate([off[b] + rng.permutation(off[b + 1] - off[b]) for b in range(pb.n_struct)])
    out = []
    try:
        for p, want_flag in ((pb, 1), (_reordered(pb, perm), 0)):
            batch = eng.upload(p)
            try:
                eng.predict(batch, "efs")
                out.append(eng.download(batch, "efs"))
                assert eng.debug_fetch_i32(batch, "win_flag", 4)[0] == want_flag
            finally:
                batch.free()
    finally:
        if which != "seed0":
            eng.close()
    a, b = out
    assert np.isfinite(a["f"]).all() and np.abs(a["f"]).max() > (1e-3 if which == "seed0" else 1.0)
    for k, t in tol.items():
        assert np.abs(a[k] - b[k]).max() < t, (k, float(np.abs(a[k] - b[k]).max()))


def test_per_atom_adjoint_matches_the_oracle(hip_engine, golden_weights, big_batch):
    """A few structures of the big batch against the CPU oracle (the other GPU parity tests run batches too small to take
    the per-atom path)."""
    import bench
    import torch

    from oracle.chgnet_oracle import OracleCHGNet

    pb = big_batch
    batch = hip_engine.upload(pb)
    try:
        hip_engine.predict(batch, "efs")
        res = hip_engine.download(batch, "efs")
    finally:
        batch.free()
    torch.set_num_threads(8)
    oracle = OracleCHGNet(golden_weights)
    graphs = bench.build_workload(N_STRUCT, 7000)
    o = pb.atom_off
    for i in (0, N_STRUCT // 2, N_STRUCT - 1):
        ref = oracle.predict_graph(graphs[i], "efs")
        assert abs(res["e"][i] - ref["e"]) < 5e-6
        assert np.abs(res["f"][o[i]:o[i + 1]] - ref["f"]).max() < 1e-5
        assert np.abs(res["s"][i] - ref["s"]).max() < 1e-4


def test_device_pointer_collectives_on_one_rank(hip_engine, golden_weights):
    """chg_batch_all_gather_energy / chg_backward_allreduce / chg_comm_info through a one-rank RCCL communicator: the table is
    the batch's energies zero-padded to the width, the all-reduced gradient is the plain one, ncclCommCount says 1."""
    from chgnet_amd.distributed import RcclComm
    from conftest import load_case

    graphs = [load_case(n)[0] for n in ("limno2", "noangle", "s16tri")]
    comm = RcclComm(0, 1, 0)
    try:
        assert comm.info() == {"rank": 0, "nranks": 1, "device": 0}
        batch = hip_engine.upload(graphs)
        try:
            hip_engine.predict(batch, "efs")
            res = hip_engine.download(batch, "efs")
            table = hip_engine.all_gather_energy(batch, comm, 8)
            assert table.shape == (1, 8) and np.array_equal(table[0, :3], res["e"]) and not table[0, 3:].any()
            cot = np.array([0.5, -1.0, 2.0], np.float32)
            plain = hip_engine.backward(batch, cot)
            comm.world = 2                      # route through chg_backward_allreduce (the communicator still has one rank)
            try:
                reduced = hip_engine.backward(batch, cot, comm=comm)
            finally:
                comm.world = 1
            scale = float(np.abs(plain).max())      # two sweeps differ by fp32 reassociation (atomics), nothing else
            assert scale > 0 and np.abs(plain - reduced).max() <= 1e-4 * scale
        finally:
            batch.free()
    finally:
        comm.close()


def test_forces_and_stress_are_the_energy_derivatives_at_the_headline_size(golden_weights):
    """Size-independent property at BASELINE's batch size (1024 structures x 40 atoms), where no CPU oracle goes: central
    differences of the ENERGY along a random displacement field and a random symmetric strain reproduce -sum F . dx and
    V sigma : eps / 160.2 per structure (model.py:517-535).  float32 energies bound the accuracy (~2e-5 eV of rounding on
    differences of 1e-3..1e-2 eV): a percent-level check of sign, units and completeness of the whole reverse sweep, structure
    by structure."""
    import bench

    from chgnet_amd import Structure
    from chgnet_amd.graph.structure import Lattice
    from chgnet_amd.model import CHGNet

    model = CHGNet(state_dict=golden_weights)
    structs = bench.workload_structures(1024, 9000)
    rng = np.random.default_rng(17)
    n = np.array([len(s) for s in structs])
    h, hs = 0.01, 0.002                                        # displacement amplitude (Angstrom), strain amplitude
    disp = [rng.normal(size=(k, 3)) / np.sqrt(k) for k in n]   # unit-length displacement field per structure
    eps = rng.normal(size=(len(structs), 3, 3))
    eps = 0.5 * (eps + eps.transpose(0, 2, 1))

    def energies(ss):
        return np.array([o["e"] for o in model.predict_structure(ss, task="e", batch_size=1024)], np.float64) * n

    def moved(sign):
        out = []
        for s, d in zip(structs, disp):
            L = np.asarray(s.lattice.matrix, np.float64)
            out.append(Structure(Lattice(L), s.atomic_numbers, np.asarray(s.frac_coords) + sign * h * d @ np.linalg.inv(L)))
        return out

    def strained(amp):
        return [Structure(Lattice(np.asarray(s.lattice.matrix, np.float64) @ (np.eye(3) + amp * e)), s.atomic_numbers, s.frac_coords)
                for s, e in zip(structs, eps)]

    base = model.predict_structure(structs, task="efs", batch_size=1024)
    want_f = np.array([-(o["f"].astype(np.float64) * d).sum() for o, d in zip(base, disp)])
    got_f = (energies(moved(+1)) - energies(moved(-1))) / (2 * h)
    vol = np.array([s.volume for s in structs])
    want_s = np.array([(o["s"].astype(np.float64) * e).sum() for o, e in zip(base, eps)]) * vol / 160.21766208
    fd1 = (energies(strained(+hs)) - energies(strained(-hs))) / (2 * hs)
    fd2 = (energies(strained(+2 * hs)) - energies(strained(-2 * hs))) / (4 * hs)
    got_s = (4.0 * fd1 - fd2) / 3.0            # Richardson: the h^2 term of the central difference cancels
    assert np.abs(want_f).mean() > 1e-2 and np.abs(want_s).mean() > 1e-1
    # absolute floors: float32 energy rounding (~2e-5 eV per structure) over the step (the strain derivative is extrapolated from two
    # steps: this random-weight model has third derivatives of 1e2..1e4 eV along a strain direction)
    for label, got, want, floor in (("force", got_f, want_f, 5e-3), ("stress", got_s, want_s, 2e-2)):
        err = np.abs(got - want)
        assert (err <= 2e-2 * np.abs(want) + floor).all(), (label, float(err.max()), int(err.argmax()), float(want[err.argmax()]))
    assert np.corrcoef(got_s, want_s)[0, 1] > 0.999 and np.corrcoef(got_f, want_f)[0, 1] > 0.999
    model.release_forward_state()


def test_weight_update_rebuilds_the_prebuilt_kernel_images(golden_weights, trained_like_weights):
    """chg_engine_update_weights: the tile kernels read their weights from images prebuilt per upload (k_*_image, stage_image) --
    an engine updated to a second weight set gives what an engine created with that set gives (also through a
    replayed hipGraph captured before the update), and not what the first set gave."""
    from chgnet_amd.engine import Engine
    from chgnet_amd.pack import pack_weights
    from conftest import load_case

    graphs = [load_case(n)[0] for n in ("limno2", "noangle", "s16tri", "li9co7o16")]
    w1, w2 = pack_weights(golden_weights), pack_weights(trained_like_weights)

    def run(eng, batch):
        eng.predict(batch, "efsm")
        return eng.download(batch, "efsm")

    fresh = Engine(w2, 0)
    try:
        b = fresh.upload(graphs)
        want = run(fresh, b)
        b.free()
    finally:
        fresh.close()
    eng = Engine(w1, 0)
    try:
        b = eng.upload(graphs)
        first = run(eng, b)
        run(eng, b)                      # second call: the captured graph replays
        eng.update_weights(w2)
        got = run(eng, b)                # replay of the graph captured under the first weights
        b2 = eng.upload(graphs)
        got2 = run(eng, b2)
        b.free(); b2.free()
    finally:
        eng.close()
    assert np.abs(first["f"] - want["f"]).max() > 1e-2
    for k in ("e", "f", "s", "m"):           # fp32 atomics: two sweeps differ by reassociation, nothing else
        tol = 2e-5 * float(np.abs(want[k]).max())
        assert np.abs(got[k] - want[k]).max() <= tol and np.abs(got2[k] - want[k]).max() <= tol, k
