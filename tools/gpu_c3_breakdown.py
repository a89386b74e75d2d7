"""Where the ragged C3 sweep (bench.py config 3) spends its wall time: cProfile of CHGNet.predict_structure on the GPU box."""
import cProfile, os, pstats, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd.model import CHGNet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
structs = [bench.sweep_structure(i) for i in range(n)]
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
model = CHGNet(state_dict=W)
model.graph_converter.set_isolated_atom_response("ignore")
model.predict_structure(structs[:64], task="efs", batch_size=64)
t = time.time(); out = model.predict_structure(structs, task="efs", batch_size=chunk); dt = time.time() - t
print(f"{n} structures in {dt:.3f} s = {n / dt:.0f} structures/s")
pr = cProfile.Profile(); pr.enable()
out = model.predict_structure(structs, task="efs", batch_size=chunk)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
