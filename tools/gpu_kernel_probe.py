"""Per-kernel times of the headline step (GPU box): python tools/gpu_kernel_probe.py [n_struct] -- the library is CHGNET_HIP_LIB."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_batch, pack_weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
task = sys.argv[2] if len(sys.argv) > 2 else "efs"
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
pb = pack_batch(bench.build_workload(n, 0))
eng = Engine(pack_weights(W), 0)
batch = eng.upload(pb)
eng.profile(True)
for it in range(3):
    eng.predict(batch, task); eng.synchronize()
    if it == 0: eng.profile_reset()
prof = eng.profile_read()
tot = sum(ms for _, ms in prof.values()) / 2
for k, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:18s} launches={cnt // 2:4d} per-step={ms / 2:8.3f} ms avg={ms / max(cnt, 1):8.3f} ms", flush=True)
eng.profile(False)
for _ in range(2): eng.predict(batch, task)
eng.synchronize()
t = time.time()
for _ in range(5): eng.predict(batch, task)
eng.synchronize(); dt = (time.time() - t) / 5
res = eng.download(batch, task)
print(f"kernel sum {tot:.2f} ms; steady: {dt * 1e3:.2f} ms/step -> {n / dt:.0f} structures/s; finite={np.isfinite(res['e']).all() and np.isfinite(res['f']).all()}"
      f" e[0]={res['e'][0]:.6f} |f|max={np.abs(res['f']).max():.5f}", flush=True)
