"""Round 6 (GPU): the MD-size path -- team-mode angle adjoints, merged prologue / embedding launches, chained row GEMMs, the
one-launch scans of the device graph build -- against the launch sequence of the large batches, the goldens and the oracle; the
ensemble step; two real engines at world size 2 on one GPU (bench.py --shared-device)."""

from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import GOLDEN, REPO, load_case

pytestmark = pytest.mark.gpu

_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from conftest import load_case
from chgnet_amd import Structure
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_weights
W = dict(np.load(sys.argv[1] + "/tests/golden/weights_trained_like.npz"))
_, d = load_case("li9co7o16")
base = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"])
rng = np.random.default_rng(3)
out = {}
eng = Engine(pack_weights(W), 0)
for tag, scale in (("a32", (1, 1, 1)), ("a256", (2, 2, 2)), ("a384", (3, 2, 2))):
    s = base.make_supercell(scale)
    s = Structure(s.lattice, s.atomic_numbers, s.frac_coords + rng.normal(0, 0.01, s.frac_coords.shape))
    b = eng.build_batch([s], 6.0, 3.0)
    for task in ("efsm", "e"):
        eng.predict(b, task)
        r = eng.download(b, task, site_energies=True, crystal_feas=True)
        for k, v in r.items():
            out[f"{tag}/{task}/{k}"] = v
    out[f"{tag}/flag"] = eng.debug_fetch_i32(b, "win_flag", 4)
    try:
        out[f"{tag}/blk_tiles"] = eng.debug_fetch_i32(b, "blk_tiles", 1)
        out[f"{tag}/blk_a"] = eng.debug_fetch_i32(b, "blk_a", 16 * int(out[f"{tag}/blk_tiles"][0]))
    except Exception:
        out[f"{tag}/blk_tiles"] = np.array([0])
    out[f"{tag}/counts"] = np.array([b.packed.n_directed, b.packed.n_angles, b.packed.n_bnodes])
    # the same structure through the HOST builder and an upload (uploaded batches keep the row-order fallback launch behind the team kernel)
    from chgnet_amd import CrystalGraphConverter
    g = CrystalGraphConverter(atom_graph_cutoff=6, bond_graph_cutoff=3)(s)
    b2 = eng.upload([g])
    eng.predict(b2, "efsm")
    for k, v in eng.download(b2, "efsm").items():
        out[f"{tag}/upload/{k}"] = v
    try:
        out[f"{tag}/upload_blk"] = np.array([int(eng.debug_fetch_i32(b2, "blk_tiles", 1)[0]), int(eng.debug_fetch_i32(b2, "win_flag", 4)[0])])
    except Exception:
        out[f"{tag}/upload_blk"] = np.array([0, 0])
    b.free(); b2.free()
# a ragged batch: 40 random cells of 10-100 atoms (the C3 sweep's generator): atoms with 2 ... 30 short bonds, every block shape
import bench
rag = [bench.sweep_structure(i) for i in range(40)]
b = eng.build_batch(rag, 6.0, 3.0)
eng.predict(b, "efsm")
for k, v in eng.download(b, "efsm").items():
    out[f"ragged/efsm/{k}"] = v
out["ragged/counts"] = np.array([b.packed.n_directed, b.packed.n_angles, b.packed.n_bnodes])
try:
    nt = int(eng.debug_fetch_i32(b, "blk_tiles", 1)[0])
    out["ragged/blk_tiles"] = np.array([nt]); out["ragged/blk_a"] = eng.debug_fetch_i32(b, "blk_a", 16 * nt); out["ragged/blk_desc"] = eng.debug_fetch_i32(b, "blk_desc", nt)
except Exception:
    out["ragged/blk_tiles"] = np.array([0])
b.free()
np.savez(sys.argv[2], **out)
'''


def _run_child(env_extra: dict, tmp: str, name: str) -> dict:
    env = dict(os.environ)
    for k in ("CHGNET_TINY_FUSE", "CHGNET_TINY_CHAIN", "CHGNET_TEAM_MIN_ANGLES", "CHGNET_BLK_MAX_ANGLES"):
        env.pop(k, None)
    env.update(env_extra)
    path = os.path.join(tmp, name + ".npz")
    subprocess.run([sys.executable, "-c", _CHILD, REPO, path], check=True, env=env, timeout=600)
    return dict(np.load(path))


def test_md_size_path_equals_the_large_batch_launch_sequence():
    """32 / 256 / 384 atoms of Li9Co7O16 (trained-like weights): the round-6 small-batch path (default: merged launches, chained row
    GEMMs, angle adjoints over 4 x 4 blocked tiles) against the launch sequence of the large batches (CHGNET_TINY_FUSE=0, blocked
    tiles and team mode off) in a child process each -- E / F / S / M, site energies and crystal features to fp32 reassociation; the
    team index is valid (flag 1) for device-built graphs; every angle sits in exactly one slot of the blocked tiles; uploaded == device-built."""
    with tempfile.TemporaryDirectory() as tmp:
        team = _run_child({"CHGNET_TEAM_MIN_ANGLES": "0", "CHGNET_BLK_MAX_ANGLES": "0"}, tmp, "team")   # every batch with angles through the team kernels
        default = _run_child({}, tmp, "default")                                                          # ... through the blocked tiles
        nochain = _run_child({"CHGNET_TINY_CHAIN": "0", "CHGNET_BLK_MAX_ANGLES": "0"}, tmp, "nochain")     # ... the row-order adjoints
        old = _run_child({"CHGNET_TINY_FUSE": "0", "CHGNET_TEAM_MIN_ANGLES": "-1", "CHGNET_BLK_MAX_ANGLES": "0"}, tmp, "old")
    tol = {"e": 2e-6, "f": 4e-5, "s": 4e-4, "m": 2e-5, "site_energies": 3e-5, "crystal_fea": 5e-4}     # the trained-like tolerances of tests/test_v020.py
    # the ragged batch: every angle in exactly one slot, all three block shapes in use, results as without the blocked tiles
    n_ang = int(old["ragged/counts"][1])
    slots = default["ragged/blk_a"]
    assert np.array_equal(np.sort(slots[slots >= 0]), np.arange(n_ang))
    assert {int(d) & 0xFF for d in default["ragged/blk_desc"]} == {0x22, 0x31, 0x13}      # log2 P | log2 Q << 4: 4 x 4, 2 x 8, 8 x 2
    assert slots.size <= 1.35 * n_ang                                                       # at least ~75 % of the slots hold an angle
    for variant in (team, default, nochain):
        for key, ref in old.items():
            if key.startswith("ragged/efsm/"):
                k = key.rsplit("/", 1)[1]
                err = float(np.abs(variant[key] - ref).max())
                assert np.isfinite(variant[key]).all() and err <= tol[k], (key, err)
    for tag in ("a32", "a256", "a384"):
        assert list(team[f"{tag}/flag"][:1]) == [1], tag
        assert np.array_equal(team[f"{tag}/counts"], old[f"{tag}/counts"])
        n_ang = int(old[f"{tag}/counts"][1])
        assert int(default[f"{tag}/blk_tiles"][0]) > 0 and int(old[f"{tag}/blk_tiles"][0]) == 0
        # the uploaded copy of the graph gets the same number of tiles from its centre-major order (k_blk_from_q), flag 1 = canonical
        assert list(default[f"{tag}/upload_blk"]) == [int(default[f"{tag}/blk_tiles"][0]), 1] and int(old[f"{tag}/upload_blk"][0]) == 0
        slots = default[f"{tag}/blk_a"]
        assert np.array_equal(np.sort(slots[slots >= 0]), np.arange(n_ang)), tag      # a permutation of the angles, the rest empty
        for variant in (team, default, nochain):
            for key, ref in old.items():
                if not key.startswith(tag + "/") or key.endswith(("/flag", "/counts", "/blk_tiles", "/blk_a", "/upload_blk")):
                    continue
                k = key.rsplit("/", 1)[1]
                err = float(np.abs(variant[key] - ref).max())
                assert np.isfinite(variant[key]).all() and err <= tol[k], (key, err)


def test_ensemble_prediction_equals_single_predictions(golden_weights):
    """R replicas of one cell with their own displacements through ONE predict_structure call (the MD-ensemble step of bench.py C4)
    == R single calls."""
    from chgnet_amd import Structure
    from chgnet_amd.graph.structure import Lattice
    from chgnet_amd.model import CHGNet

    _, d = load_case("li9co7o16")
    base = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"]).make_supercell([2, 2, 1])
    rng = np.random.default_rng(5)
    reps = [Structure(base.lattice, base.atomic_numbers, base.frac_coords + rng.normal(0, 0.01, base.frac_coords.shape)) for _ in range(6)]
    model = CHGNet(state_dict=golden_weights)
    try:
        together = model.predict_structure(reps, task="ef", batch_size=6, min_atoms_per_batch=0)
        for s, p in zip(reps, together):
            one = model.predict_structure(s, task="ef")
            assert abs(float(p["e"]) - float(one["e"])) < 2e-6
            assert np.abs(p["f"] - one["f"]).max() < 5e-6
    finally:
        if model._engine is not None:
            model._engine.close()


def test_two_engines_at_world_size_two_on_one_gpu():
    """bench.py --gpus 2 --shared-device: two ranks, two engines, GPU 0, gloo collectives -- the REAL C2 step, C3 sweep and C5 steps
    at world size 2: every slot of the gathered energy tables and the all-reduced gradient checked on rank 0 against its own
    single-engine results (no scaling claim: one device)."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--shared-device", "--structures", "64", "--sweep-structures", "60"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["process_group_ranks"] == 2 and line["shared_device"] is True and line["ok"] is True
    assert all(line["parity"].values()) and set(line["parity"]) == {"C2_energy_table", "C3_energy_table", "C5_allreduced_gradient", "C5_weights_identical_on_all_ranks"}


def test_mid_size_scans_of_the_graph_build_are_exact(hip_engine):
    """The one-launch chained scans (3-5 chunks for a 256-atom cell; longer arrays keep the three-launch scan): the device-built
    arrays of a 1,024-atom cell (13-28 chunks) equal the host builder's, bit for bit."""
    from chgnet_amd import CrystalGraphConverter, Structure
    from chgnet_amd.graph.structure import Lattice
    from chgnet_amd.pack import pack_batch

    _, d = load_case("li9co7o16")
    s = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"]).make_supercell([4, 4, 2])
    rng = np.random.default_rng(1)
    s = Structure(s.lattice, s.atomic_numbers, s.frac_coords + rng.normal(0, 0.004, s.frac_coords.shape))
    want = pack_batch([CrystalGraphConverter(atom_graph_cutoff=6, bond_graph_cutoff=3)(s)])
    for _ in range(2):          # the exact pass, then the single-pass (speculative) build
        b = hip_engine.build_batch([s])
        try:
            assert (b.packed.n_directed, b.packed.n_angles, b.packed.n_bnodes) == (want.n_directed, want.n_angles, want.n_bnodes)
            for name in ("e_center", "e_nbr", "e_d2u", "e_rev", "u_u2d", "u_bnode", "bn_und", "a_ctr", "a_b1c", "a_b2c", "a_d1", "a_d2"):
                got = hip_engine.debug_fetch_i32(b, name, len(want.arrays[name]))
                assert np.array_equal(got, want.arrays[name]), name
        finally:
            b.free()


def test_fine_tuning_gradient_of_a_device_built_batch_equals_the_uploaded_one(trained_like_weights):
    """chg_backward (tangent + two-adjoint sweeps) starts from the first-order adjoints the force sweep leaves in the batch: on a batch built on
    the device those come from the blocked-tile angle adjoints, on an uploaded one from the row-order kernels -- same gradient."""
    import bench
    from chgnet_amd import CrystalGraphConverter

    from chgnet_amd.engine import Engine
    from chgnet_amd.pack import pack_weights

    eng = Engine(pack_weights(trained_like_weights), 0)
    structs = [bench.sweep_structure(i) for i in range(12)]
    conv = CrystalGraphConverter(atom_graph_cutoff=6, bond_graph_cutoff=3)
    b_dev = eng.build_batch(structs, 6.0, 3.0)
    b_up = eng.upload([conv(s) for s in structs])
    try:
        n, n_atoms = len(structs), b_dev.packed.n_atoms
        if os.environ.get("CHGNET_BLK_MAX_ANGLES") is None:      # (the suite is also run with the blocked tiles switched off)
            assert int(eng.debug_fetch_i32(b_dev, "blk_tiles", 1)[0]) > 0
        rng = np.random.default_rng(0)
        cot = rng.normal(1, 0.1, n).astype(np.float32)
        gf = rng.normal(0, 0.01, (n_atoms, 3)).astype(np.float32)
        gs = rng.normal(0, 0.01, (n, 3, 3)).astype(np.float32)
        grads = []
        for b in (b_dev, b_up):
            eng.predict(b, "efs")
            grads.append(eng.backward(b, cot, f_grad=gf, s_grad=gs))
        assert np.isfinite(grads[0]).all()
        assert np.abs(grads[0] - grads[1]).max() <= 2e-5 * np.abs(grads[1]).max()
    finally:
        b_dev.free(); b_up.free()
        eng.close()


def test_uploaded_graph_without_the_canonical_angle_order_falls_back_behind_the_blocked_tiles(hip_engine):
    """A small uploaded batch runs its angle adjoints over blocked tiles (index from the centre-major order, k_blk_from_q) when its angle
    rows have the reference's group structure (flag 1); with the rows shuffled inside every structure the device clears the flag, the
    blocked-tile kernels return at once and the row-order adjoints launched behind them do the work -- same forces and stresses."""
    import bench
    from chgnet_amd.pack import pack_batch
    from test_gpu_round3 import _reordered

    pb = pack_batch(bench.build_workload(24, 300))          # 960 atoms
    rng = np.random.default_rng(4)
    off = pb.ang_off
    perm = np.concatenate([off[b] + rng.permutation(off[b + 1] - off[b]) for b in range(pb.n_struct)])
    out = []
    for p_, want_flag in ((pb, 1), (_reordered(pb, perm), 0)):
        batch = hip_engine.upload(p_)
        try:
            hip_engine.predict(batch, "efs")
            out.append(hip_engine.download(batch, "efs"))
            if os.environ.get("CHGNET_BLK_MAX_ANGLES") is None:
                assert int(hip_engine.debug_fetch_i32(batch, "blk_tiles", 1)[0]) > 0
                assert hip_engine.debug_fetch_i32(batch, "win_flag", 4)[0] == want_flag
        finally:
            batch.free()
    a, b = out
    assert np.isfinite(a["f"]).all() and np.abs(a["f"]).max() > 1e-3
    for k, t in {"e": 2e-6, "f": 2e-6, "s": 2e-5}.items():
        assert np.abs(a[k] - b[k]).max() < t, (k, float(np.abs(a[k] - b[k]).max()))
