"""MD-size batch (2x2x2 Li9Co7O16, 256 atoms): one prediction + download on a RESIDENT batch, eager launches against hipGraph replay
(CHGNET_HIP_GRAPHS=0 / 1 in separate processes).  What a rebuilt graph that could replay the captured launch sequence would gain."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_case
from chgnet_amd import Structure
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
scale = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,2,2").split(",")]
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
_, d = load_case("li9co7o16")
s = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"]).make_supercell(scale)
eng = Engine(pack_weights(W), 0)
batch = eng.build_batch([s], 6.0, 3.0)
for _ in range(20):
    eng.predict(batch, "ef"); eng.download(batch, "ef")
t0 = time.perf_counter()
for _ in range(n):
    eng.predict(batch, "ef"); eng.download(batch, "ef")
dt = time.perf_counter() - t0
print(f"{len(s)} atoms, CHGNET_HIP_GRAPHS={os.environ.get('CHGNET_HIP_GRAPHS', '1')}: {1e6 * dt / n:.1f} us per predict + download")
