import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
import bench
t = time.time(); import torch; print("import torch", time.time() - t, flush=True)
from oracle.chgnet_oracle import OracleCHGNet
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
graphs = bench.build_workload(16, 0)
m = OracleCHGNet(W)
for nt in [int(x) for x in sys.argv[1:]]:
    torch.set_num_threads(nt)
    t = time.time(); m.predict_graph(graphs[0], "efs"); w = time.time() - t
    t = time.time()
    for g in graphs[:4]: m.predict_graph(g, "efs")
    dt1 = (time.time() - t) / 4
    t = time.time(); m.predict_graph(graphs, "efs", batch_size=16); dt16 = (time.time() - t) / 16
    print(f"threads {nt}: warm {w:.2f}s  bs1 {1/dt1:.2f}/s  bs16 {1/dt16:.2f}/s", flush=True)
