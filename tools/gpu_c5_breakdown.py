"""Timing probe (GPU box): where one TrainStep of bench config C5 (1024 structures, targets efsm) spends its wall time."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd import CrystalGraphConverter
from chgnet_amd.model import CHGNet
from chgnet_amd.pack import pack_batch
from chgnet_amd.trainer import TrainStep, allreduce_gradients
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
model = CHGNet(state_dict=W)
conv = CrystalGraphConverter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
graphs = [conv(s) for s in bench.workload_structures(n, 1000)]
rng = np.random.default_rng(0)
labels = {"e": -7.0 + rng.normal(0, 0.05, n).astype(np.float32),
          "f": [rng.normal(0, 0.05, (len(g_.atomic_number), 3)).astype(np.float32) for g_ in graphs],
          "s": [rng.normal(0, 0.2, (3, 3)).astype(np.float32) for _ in graphs],
          "m": [np.abs(rng.normal(0.5, 0.2, len(g_.atomic_number))).astype(np.float32) for g_ in graphs]}
step = TrainStep(model, targets="efsm", learning_rate=1e-4)
t = time.perf_counter(); packed = pack_batch(graphs); print(f"pack_batch {1e3*(time.perf_counter()-t):.1f} ms")
step(packed, labels)
for rep in range(3):
    T = {}
    t0 = time.perf_counter(); pred = model.forward(packed, task=step.task); T["forward (upload + predict + download)"] = time.perf_counter() - t0
    t0 = time.perf_counter(); info, g = step.loss.gradients(labels, pred); T["loss + its gradients (host)"] = time.perf_counter() - t0
    t0 = time.perf_counter(); grads = model.backward(g.get("e"), g.get("m"), g.get("f"), g.get("s")); T["backward (cotangent upload + sweeps + blob download + unpack)"] = time.perf_counter() - t0
    t0 = time.perf_counter(); grads = allreduce_gradients(grads); new = step.optimizer.step(model.state_dict(), grads); T["Adam (host)"] = time.perf_counter() - t0
    t0 = time.perf_counter(); model.load_state_dict(new); T["load_state_dict (pack + upload + transposes)"] = time.perf_counter() - t0
    print(f"rep {rep}: total {1e3*sum(T.values()):.1f} ms")
    for k, v in T.items():
        print(f"    {k:64s} {1e3*v:8.2f} ms")
eng = model.engine
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(packed, labels); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
