import os, sys, time
import numpy as np
REPO = "/root/repo" if os.path.isdir("/root/repo/chgnet_amd") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, REPO)
import bench
from concurrent.futures import ThreadPoolExecutor
from chgnet_amd import CrystalGraphConverter
from chgnet_amd.model import CHGNet
from chgnet_amd.pack import pack_batch
from chgnet_amd.trainer import TrainStep
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
model = CHGNet(state_dict=W)
conv = CrystalGraphConverter()
n = 1024
batches = [[conv(s) for s in bench.workload_structures(n, 1000 + i * n)] for i in range(10)]
rng = np.random.default_rng(0)
def labels_for(b_):
    return {"e": -7.0 + rng.normal(0, 0.05, len(b_)).astype(np.float32),
            "f": [rng.normal(0, 0.05, (len(g_.atomic_number), 3)).astype(np.float32) for g_ in b_],
            "s": [rng.normal(0, 0.2, (3, 3)).astype(np.float32) for _ in b_],
            "m": [np.abs(rng.normal(0.5, 0.2, len(g_.atomic_number))).astype(np.float32) for g_ in b_]}
labels = [labels_for(b_) for b_ in batches]
class _OneRank:          # keeps allreduce_gradients from importing torch (PROBE_NO_TORCH=1)
    world = 1
step = TrainStep(model, targets="efsm", learning_rate=1e-4, comm=_OneRank() if os.environ.get("PROBE_NO_TORCH") else None)
step(batches[0], labels[0]); step(batches[0], labels[0])
packed = [pack_batch(b) for b in batches[:4]]
for i in range(4):
    t = time.perf_counter(); step(packed[i], labels[i]); print(f"pre-packed step {i}: {1e3*(time.perf_counter()-t):.1f} ms")
with ThreadPoolExecutor(max_workers=1) as pool:
    nxt = pool.submit(pack_batch, batches[0])
    for i in range(10):
        t0 = time.perf_counter(); p = nxt.result(); t1 = time.perf_counter()
        nxt = pool.submit(pack_batch, batches[(i + 1) % 10])
        step(p, labels[i]); t2 = time.perf_counter()
        print(f"epoch step {i}: wait for pack {1e3*(t1-t0):.1f} ms, step {1e3*(t2-t1):.1f} ms")
t = time.perf_counter(); pack_batch(batches[0]); print(f"pack alone {1e3*(time.perf_counter()-t):.1f} ms")
step.seconds.clear()
step.run_epoch(batches, labels)
print("torch imported:", "torch" in sys.modules); print("run_epoch split (ms/step):", {k: round(1e3 * v / step.seconds["calls"], 2) for k, v in step.seconds.items() if k != "calls"})
