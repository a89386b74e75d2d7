"""Print max|engine - model| for every pipeline buffer (GPU box; debugging aid, not a test)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_case
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_weights
from oracle.staged_ref import StagedModel

W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
pw = pack_weights(W)
eng = Engine(pw, 0)
rng = np.random.default_rng(0)
for rows, k, nout in [(37, 64, 64), (200, 64, 128), (130, 128, 64)]:
    x = rng.normal(size=(rows, k)).astype(np.float32); wt = rng.normal(size=(nout, k)).astype(np.float32); b = rng.normal(size=nout).astype(np.float32)
    y = eng.test_rows_gemm(x, wt, b)
    print(f"rows_gemm {rows}x{k}->{nout}: err {np.abs(y - (x.astype(np.float64) @ wt.T.astype(np.float64) + b)).max():.3e}", flush=True)
names = sys.argv[1:] or ["limno2", "noangle", "s16tri"]
graphs = [load_case(n)[0] for n in names]
batch = eng.upload(graphs)
pb = batch.packed
eng.predict(batch, "efsm")
res = eng.download(batch, "efsm", site_energies=True, atom_feas=True, crystal_feas=True)
ref = StagedModel(pw).run(pb)
buf = ref["buffers"]
N, Ed, Eu, A, Eb = pb.n_atoms, pb.n_directed, pb.n_undirected, pb.n_angles, pb.n_bnodes
def rep(name, got, want):
    if want.size == 0:
        print(f"{name:12s} empty"); return
    err = np.abs(got - want); i = np.unravel_index(np.nanargmax(err), err.shape)
    print(f"{name:12s} max|d|={np.nanmax(err):.3e}  max|ref|={np.abs(want).max():.3e}  nan={int(np.isnan(got).sum())}  at {i}", flush=True)
ev = eng.debug_fetch(batch, "ev", (Ed, 4)); eu = eng.debug_fetch(batch, "eu", (Ed, 4))
rep("cart", eng.debug_fetch(batch, "cart", (N, 3)), buf["cart"])
rep("bond_vec", ev[:, :3], buf["bond_vec"]); rep("bond_len", ev[:, 3], buf["bond_len"]); rep("bond_unit", eu[:, :3], buf["bond_unit"])
for name, shape in [("hb0", (Eu, 64)), ("wag", (Eu, 64)), ("wbgc", (Eb, 64)), ("ang0", (A, 64)), ("atom0", (N, 64)), ("hbc0", (Eb, 64)),
                    ("atom1", (N, 64)), ("hbc1", (Eb, 64)), ("ang1", (A, 64)), ("atom2", (N, 64)), ("hbc2", (Eb, 64)), ("ang2", (A, 64)),
                    ("atom3", (N, 64)), ("hbc3", (Eb, 64)), ("atom4", (N, 64)),
                    ("Gb", (Eu, 64)), ("Gwag", (Eu, 64)), ("Gwbgc", (Eb, 64)), ("Gang", (A, 64))]:
    rep(name, eng.debug_fetch(batch, name, shape), buf[name])
rep("Grk", eng.debug_fetch(batch, "Grk", (Eu,)), buf["Gr"][pb.u_u2d])
rep("Gu", eng.debug_fetch(batch, "Gu", (Ed, 4))[:, :3], buf["Gu"])
for k in ("e", "f", "s", "m", "site_energies", "atom_fea", "crystal_fea"):
    rep("out." + k, res[k], ref[k])
batch.free(); eng.close()
