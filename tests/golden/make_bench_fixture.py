"""Writes tests/golden/bench_c2_oracle.npz: the CPU oracle's E / |F| / tr(S) for every structure of the headline
bench workload (BASELINE.json configs[1]: 1024 perturbed LiMnO2 5x1x1 cells, seeds 0..1023) with the seeded
random weights, so that the GPU test can check ALL 1024 structures of the full-size batch against the oracle
without spending a minute of CPU time on the GPU box.

    python tests/golden/make_bench_fixture.py        (about 1-2 minutes on 8 cores)

The oracle itself is pinned to the unmodified reference by tests/test_oracle_golden.py.
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from oracle.chgnet_oracle import OracleCHGNet  # noqa: E402

torch.set_num_threads(8)
W = dict(np.load(os.path.join(REPO, "tests", "golden", "weights_seed0.npz")))
graphs = bench.build_workload(1024, 0)
model = OracleCHGNet(W)
e, fn, st, f0 = [], [], [], []
for i in range(0, 1024, 16):
    for p in model.predict_graph(graphs[i:i + 16], "efs", batch_size=16):
        e.append(p["e"]); fn.append(np.sqrt((p["f"].astype(np.float64) ** 2).sum())); st.append(np.trace(p["s"].astype(np.float64)))
        f0.append(p["f"][0])
    print(i, flush=True)
np.savez(os.path.join(REPO, "tests", "golden", "bench_c2_oracle.npz"), e=np.array(e, np.float32), f_norm=np.array(fn, np.float64),
         s_trace=np.array(st, np.float64), f_atom0=np.array(f0, np.float32),
         n_directed=np.array([len(g.atom_graph) for g in graphs], np.int32), n_angles=np.array([len(g.bond_graph) for g in graphs], np.int32))
