"""Timing probe (GPU box): chg_backward (weight gradients of the energy loss) on the headline batch."""
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
print("stream copy GB/s:", [round(eng.stream_copy_gbs(1 << 30, 10)) for _ in range(3)], flush=True)
batch = eng.build_batch(bench.workload_structures(n, 0))
eng.predict(batch, "e"); eng.synchronize()
cot = np.ones(n, np.float32)
for it in range(3):
    t = time.perf_counter(); eng.predict(batch, "e"); eng.synchronize(); tf = time.perf_counter() - t
    t = time.perf_counter(); g = eng.backward(batch, cot); tb = time.perf_counter() - t
    print(f"forward(e) {tf*1e3:.2f} ms, backward {tb*1e3:.2f} ms, |grad|max {np.abs(g).max():.3e} finite={np.isfinite(g).all()}", flush=True)
eng.profile(True); eng.profile_reset()
eng.predict(batch, "e"); g = eng.backward(batch, cot); eng.synchronize()
prof = eng.profile_read()
tot = sum(ms for _, ms in prof.values())
for k, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:24s} {cnt:4d} launches {ms:8.3f} ms", flush=True)
print(f"profiled forward(e)+backward total {tot:.2f} ms -> {n/(tot*1e-3):.0f} structures/s (energy-loss training step, fp32)")
