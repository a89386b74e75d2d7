"""predict_structure with the reference's default arguments (batch_size=16) on 1024 structures."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from chgnet_amd import Structure
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.model import CHGNet
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
lat = Lattice.from_parameters(2.868779, 4.634475, 5.832507, 90, 90, 90)
frac = [[0.5, 0.5, 0.3797505], [0, 0, 0.6202495], [0.5, 0.5, 0.8632525], [0, 0, 0.1367475],
        [0.5, 0, 0.3608245], [0, 0.5, 0.0985135], [0.5, 0, 0.9014865], [0, 0.5, 0.6391755]]
base = Structure(lat, ["Li", "Li", "Mn", "Mn", "O", "O", "O", "O"], frac).make_supercell([5, 1, 1])
structs = [base.perturb(0.01, np.random.default_rng(i)) for i in range(1024)]
model = CHGNet(state_dict=W)
model.predict_structure(structs[:64], task="efs")
for floor in (40960, 0):
    model.min_atoms_per_batch = floor
    t = time.perf_counter(); out = model.predict_structure(structs, task="efs"); dt = time.perf_counter() - t
    print(f"min_atoms_per_batch={floor}: 1024 structures, default batch_size=16: {dt*1e3:.1f} ms = {1024/dt:.0f} structures/s")
