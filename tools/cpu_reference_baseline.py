"""The REAL reference (/root/reference, unmodified) timed on this build container's host cores on a sample of the headline workload:
`CHGNet.predict_graph(graphs, task="efs", batch_size=b)` (model/model.py:593-665), random-init 0.3.0 architecture, fp32, CPU.

    python tools/cpu_reference_baseline.py        -> profiles/reference_cpu_baseline.json

The GPU box has no /root/reference (it does not travel), so bench.py times the oracle port there (`cpu_baseline.kind = "port"`) and
quotes this file next to it (`cpu_baseline.reference_in_build_container`): SURVEY 8d / BASELINE.md section 4 ask for the reference itself.
"""
from __future__ import annotations

import json
import os
import statistics
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle._refimport import load_reference  # noqa: E402

load_reference(fast_graph=True)
from chgnet.graph.crystalgraph import CrystalGraph as RefGraph  # noqa: E402
from chgnet.model.model import CHGNet as RefCHGNet  # noqa: E402

import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ncpu = len(os.sched_getaffinity(0))
ours = bench.build_workload(N, 0)
graphs = [RefGraph(atomic_number=torch.as_tensor(g.atomic_number), atom_frac_coord=torch.as_tensor(g.atom_frac_coord),
                   atom_graph=torch.as_tensor(g.atom_graph), atom_graph_cutoff=6.0, neighbor_image=torch.as_tensor(g.neighbor_image),
                   directed2undirected=torch.as_tensor(g.directed2undirected), undirected2directed=torch.as_tensor(g.undirected2directed),
                   bond_graph=torch.as_tensor(g.bond_graph), bond_graph_cutoff=3.0, lattice=torch.as_tensor(g.lattice)) for g in ours]
torch.manual_seed(0)
model = RefCHGNet().eval()
rows = []
for threads in sorted({1, ncpu}):
    torch.set_num_threads(threads)
    for bs in (1, 16):
        n = min(N, 16 if threads == 1 else N)
        model.predict_graph(graphs[:bs], task="efs", batch_size=bs)          # warm-up
        reps = []
        for _ in range(3 if threads == 1 else 5):
            t0 = time.perf_counter()
            model.predict_graph(graphs[:n], task="efs", batch_size=bs)
            reps.append(n / (time.perf_counter() - t0))
        rows.append({"threads": threads, "batch_size": bs, "structures": n, "structures_per_s": round(statistics.median(reps), 3)})
        print(rows[-1], flush=True)
best = max((r for r in rows if r["threads"] == ncpu), key=lambda r: r["structures_per_s"])
out = {"what": "unmodified reference CHGNet.predict_graph(task='efs') on the build container's host cores, headline workload sample "
               "(LiMnO2 5x1x1, 40 atoms), random-init 0.3.0 architecture, torch fp32 CPU; median of the repeats",
       "value": best["structures_per_s"], "unit": "structures/s", "cores": ncpu, "batch_size": best["batch_size"], "kind": "reference",
       "torch": torch.__version__, "rows": rows, "script": "tools/cpu_reference_baseline.py"}
json.dump(out, open(os.path.join(REPO, "profiles", "reference_cpu_baseline.json"), "w"), indent=1)
print(json.dumps(out)[:300])
