import sys, time, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
from chgnet_amd.trainer import CombinedLoss
rng = np.random.default_rng(0)
B, n = 1024, 40
pred = {"e": rng.normal(size=B).astype(np.float32), "f": [rng.normal(size=(n, 3)).astype(np.float32) for _ in range(B)],
        "s": [rng.normal(size=(3, 3)).astype(np.float32) for _ in range(B)], "m": [rng.normal(size=n).astype(np.float32) for _ in range(B)]}
targ = {"e": rng.normal(size=B).astype(np.float32), "f": [rng.normal(size=(n, 3)).astype(np.float32) for _ in range(B)],
        "s": [rng.normal(size=(3, 3)).astype(np.float32) for _ in range(B)], "m": [rng.normal(size=n).astype(np.float32) for _ in range(B)]}
loss = CombinedLoss(target_str="efsm")
for rep in range(3):
    t = time.perf_counter()
    for _ in range(20): loss.gradients(targ, pred)
    print(f"{(time.perf_counter()-t)/20*1e3:.2f} ms per call", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): loss.gradients(targ, pred)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(8)
