"""C5 epoch with and without ``upload_ahead`` (GPU box): ms per 1024-structure step and the host-side split."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd import CrystalGraphConverter
from chgnet_amd.model import CHGNet
from chgnet_amd.trainer import TrainStep
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
model = CHGNet(state_dict=W)
conv = CrystalGraphConverter(atom_graph_cutoff=6, bond_graph_cutoff=3)
distinct = [[conv(s) for s in bench.workload_structures(n, 1000 + i * n)] for i in range(3)]
rng = np.random.default_rng(0)
def labels(b_):
    return {"e": rng.normal(-7, 0.05, len(b_)).astype(np.float32), "f": [rng.normal(0, 0.05, (len(g.atomic_number), 3)).astype(np.float32) for g in b_],
            "s": [rng.normal(0, 0.2, (3, 3)).astype(np.float32) for _ in b_], "m": [np.abs(rng.normal(0.5, 0.2, len(g.atomic_number))).astype(np.float32) for g in b_]}
dl = [labels(b) for b in distinct]
batches = [distinct[i % 3] for i in range(steps)]
lab = [dl[i % 3] for i in range(steps)]
step = TrainStep(model, targets="efsm", learning_rate=1e-5)
step.run_epoch(batches[:1], lab[:1])
eng = model.engine
for rep in range(int(os.environ.get("AB_REPS", "2"))):
    for ahead in (False, True):
        step.seconds.clear()
        t0 = time.perf_counter(); step.run_epoch(batches, lab, upload_ahead=ahead); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"upload_ahead={ahead}: {1e3 * dt / steps:.1f} ms per step = {n * steps / dt:.0f} structures/s; split " +
              ", ".join(f"{k} {1e3 * v / steps:.1f}" for k, v in step.seconds.items() if k != "calls"), flush=True)
