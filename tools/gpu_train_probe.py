"""Timing probe (GPU box): chg_backward on the headline batch -- first-order (energy loss) and second-order (E+F+S loss)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
eng = Engine(pack_weights(W), 0)
batch = eng.build_batch(bench.workload_structures(n, 0))
pb = batch.packed
eng.predict(batch, "efs"); eng.synchronize()
cot = np.ones(n, np.float32); gf = np.ones((pb.n_atoms, 3), np.float32) * 0.01; gs = np.ones((n, 3, 3), np.float32) * 0.01
for name, kw in (("first-order (e)", dict()), ("second-order (efs)", dict(f_grad=gf, s_grad=gs))):
    for it in range(3):
        t = time.perf_counter(); eng.predict(batch, "efs"); eng.synchronize(); tf = time.perf_counter() - t
        t = time.perf_counter(); g = eng.backward(batch, cot, **kw); tb = time.perf_counter() - t
        print(f"{name}: forward(efs) {tf*1e3:.2f} ms, backward {tb*1e3:.2f} ms, |grad|max {np.abs(g).max():.3e} finite={np.isfinite(g).all()}", flush=True)
    eng.profile(True); eng.profile_reset()
    g = eng.backward(batch, cot, **kw); eng.synchronize()
    prof = eng.profile_read(); eng.profile(False)
    tot = sum(ms for _, ms in prof.values())
    for k, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"    {k:24s} {cnt:4d} launches {ms:8.3f} ms", flush=True)
    print(f"  {name}: profiled backward {tot:.2f} ms", flush=True)
print("device bytes: batch %.1f GB" % (batch.device_bytes / 1e9))
