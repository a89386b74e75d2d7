"""Timing probe (GPU box): TrainStep over a few batches (bench config C5), per-step wall time."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd import CrystalGraphConverter
from chgnet_amd.model import CHGNet
from chgnet_amd.trainer import TrainStep
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
model = CHGNet(state_dict=W)
conv = CrystalGraphConverter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batches = [[conv(s) for s in bench.workload_structures(n, 1000 + i * n)] for i in range(3)]
rng = np.random.default_rng(0)
def labels_for(b_):
    return {"e": -7.0 + rng.normal(0, 0.05, len(b_)).astype(np.float32),
            "f": [rng.normal(0, 0.05, (len(g_.atomic_number), 3)).astype(np.float32) for g_ in b_],
            "s": [rng.normal(0, 0.2, (3, 3)).astype(np.float32) for _ in b_],
            "m": [np.abs(rng.normal(0.5, 0.2, len(g_.atomic_number))).astype(np.float32) for g_ in b_]}
labels = [labels_for(b_) for b_ in batches]
step = TrainStep(model, targets="efsm", learning_rate=1e-4)
for rep in range(2):
    for i in range(3):
        t = time.perf_counter(); info = step(batches[i], labels[i]); dt = time.perf_counter() - t
        print(f"rep {rep} step {i}: {dt*1e3:.1f} ms loss {info['loss']:.4g} free/total GB {[round(x/1e9,1) for x in model.engine.memory_info()]}", flush=True)
t = time.perf_counter(); step.run_epoch(batches, labels); print(f"run_epoch 3 steps: {(time.perf_counter()-t)/3*1e3:.1f} ms/step", flush=True)
