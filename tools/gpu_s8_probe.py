"""SURVEY.md §8d asks for S8x1024 beside the S40x1024 headline: 1024 perturbed 8-atom LiMnO2 cells."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from chgnet_amd import Structure
from chgnet_amd.engine import Engine
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.pack import pack_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
lat = Lattice.from_parameters(2.868779, 4.634475, 5.832507, 90, 90, 90)
frac = [[0.5, 0.5, 0.3797505], [0, 0, 0.6202495], [0.5, 0.5, 0.8632525], [0, 0, 0.1367475],
        [0.5, 0, 0.3608245], [0, 0.5, 0.0985135], [0.5, 0, 0.9014865], [0, 0.5, 0.6391755]]
base = Structure(lat, ["Li", "Li", "Mn", "Mn", "O", "O", "O", "O"], frac)
structs = [base.perturb(0.01, np.random.default_rng(i)) for i in range(n)]
eng = Engine(pack_weights(W), 0)
batch = eng.build_batch(structs)
for _ in range(3):
    eng.predict(batch, "efs")
eng.synchronize()
t = time.time()
for _ in range(10):
    eng.predict(batch, "efs")
eng.synchronize()
dt = (time.time() - t) / 10
res = eng.download(batch, "efs")
print(f"S8 x {n}: {dt*1e3:.2f} ms/step = {n/dt:.0f} structures/s (resident, efs); finite={np.isfinite(res['e']).all()}", flush=True)
