"""SURVEY 8d "C3" probe (GPU box): ragged sweep of random 10-100-atom cells, structures in -> E/F/S out,
graphs built on the device, chunked like predict_structure.  Prints end-to-end structures/s."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from chgnet_amd import Structure
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.model import CHGNet

n_struct = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rng = np.random.default_rng(12345)
t0 = time.time()
structs = []
for _ in range(n_struct):
    n = int(rng.integers(10, 101))
    vol = n / 0.103                                   # LiMnO2's number density
    a = vol ** (1 / 3) * rng.uniform(0.85, 1.15); b = vol ** (1 / 3) * rng.uniform(0.85, 1.15)
    lat = np.diag([a, b, vol / (a * b)])
    # jittered lattice of sites: min distance ~1.6 A without an O(n^2) rejection loop
    m = int(np.ceil(n ** (1 / 3)))
    grid = np.array([[i, j, k] for i in range(m) for j in range(m) for k in range(m)], dtype=np.float64)
    pick = rng.choice(len(grid), size=n, replace=False)
    frac = (grid[pick] + 0.5 + rng.uniform(-0.15, 0.15, (n, 3))) / m
    structs.append(Structure(Lattice(lat), rng.choice([3, 25, 27, 8], size=n), frac))
print(f"generated {n_struct} structures, {sum(len(s) for s in structs)} atoms in {time.time()-t0:.1f}s", flush=True)
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
model = CHGNet(state_dict=W)
model.graph_converter.set_isolated_atom_response("ignore")
model.predict_structure(structs[:8], task="efs", batch_size=8)            # warm-up
for rep in range(2):
    t = time.time()
    out = model.predict_structure(structs, task="efs", batch_size=chunk)
    dt = time.time() - t
    print(f"sweep: {n_struct} structures in {dt:.3f}s = {n_struct/dt:.0f} structures/s end to end (chunk {chunk}); "
          f"finite={all(np.isfinite(o['e']) for o in out)}", flush=True)
