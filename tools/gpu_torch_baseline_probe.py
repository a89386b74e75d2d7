"""Informative second baseline: the torch restatement of the reference (oracle/)
run through torch-ROCm eager on the same GPU — the "hipified PyTorch" route the engine replaces.
Test infrastructure only; prints structures/s for C2-shaped structures."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from oracle.chgnet_oracle import OracleCHGNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
graphs = bench.build_workload(n, 0)
torch.set_default_device("cuda")
model = OracleCHGNet(W)
for bs in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("16", "64"))]:
    model.predict_graph(graphs[:bs], task="efs", batch_size=bs)   # warm-up (kernel load, allocator)
    torch.cuda.synchronize()
    t = time.time()
    model.predict_graph(graphs, task="efs", batch_size=bs)
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f"torch-ROCm eager, batch_size={bs}: {n} structures in {dt:.2f}s = {n/dt:.1f} structures/s", flush=True)
