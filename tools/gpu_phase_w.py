"""Per-phase shader-clock breakdown of the windowed angle adjoints (needs a -DCHG_PHASE_TIMING build in CHGNET_HIP_LIB)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_batch, pack_weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
pb = pack_batch(bench.build_workload(n, 0))
eng = Engine(pack_weights(W), 0)
os.environ["CHGNET_HIP_GRAPHS"] = "0"
batch = eng.upload(pb)
eng.predict(batch, "efs"); eng.synchronize()
eng.profile(True)
eng.predict(batch, "efs"); eng.synchronize()
prof = eng.profile_read()
print({k: round(ms / max(c, 1), 3) for k, (c, ms) in prof.items() if "conv_" in k or "angleupd" in k})
ph = eng.debug_fetch(batch, "phase", 64)
names = ["indices+gathers", "GEMMs+gated", "Wang^T+Gang", "LDS scatter", "flush", "wait open", "Gwbgc scatter", "-", "-", "pre-wait"]
tiles = pb.n_angles / 16
for base, label, launches in ((40, "bondconv_bwd_w", 3), (50, "angleupd_bwd_w", 2)):
    v = ph[base:base + 10]
    tot = v.sum()
    print(f"{label}: {tot / (tiles * launches):8.0f} cycles per wave-tile: " + ", ".join(f"{nm} {v[i] / (tiles * launches):.0f}" for i, nm in enumerate(names) if nm != "-"))
