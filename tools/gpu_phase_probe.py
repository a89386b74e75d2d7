"""Per-phase shader-clock breakdown of the angle kernels (needs a -DCHG_PHASE_TIMING build in CHGNET_HIP_LIB)."""
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
PH_WAVES = 4096                                   # kernels_conv.h: phase[kernel][later / first tile][slot][wave]
ph4 = eng.debug_fetch(batch, "phase", 4 * 2 * 10 * PH_WAVES).reshape(4, 2, 10, PH_WAVES).sum(axis=(1, 3))
ph = np.zeros(40)
for k in range(4):
    ph[10 * k:10 * k + 10] = ph4[k]               # slot 9 = prologue (per wave, not per tile)
names = {0: ["idx+ang rows", "table gather", "W_ang GEMM", "gated fwd", "output scatter"],
         10: ["idx+ang rows", "table gather", "W_ang GEMM", "gated fwd", "rows+dE/dy+Gwbgc", "gated bwd", "W_ang^T GEMM", "Gang update", "GR/GS scatter"]}
tiles = pb.n_angles / 16
for base, label, launches in ((0, "bondconv_fwd", 3), (10, "bondconv_bwd", 3), (20, "angleupd_fwd", 2), (30, "angleupd_bwd", 2)):
    v = ph[base:base + 10]
    tot = v.sum()
    print(f"{label}: {tot / (tiles * launches):8.0f} cycles per wave-tile")
    for i, nm in enumerate(names[10 if base % 20 else 0]):
        print(f"   {nm:20s} {v[i] / (tiles * launches):8.0f}  {100 * v[i] / tot:5.1f} %")

# per-atom adjoints (kernels_angle_w.h): kernel slots 4 (BondConv) and 5 (AngleUpdate); tiles padded per atom
try:
    ph6 = eng.debug_fetch(batch, "phase", 6 * 2 * 10 * PH_WAVES).reshape(6, 2, 10, PH_WAVES).sum(axis=(1, 3))
    wn = ["indices + gathers", "dE/dy rows + gated bwd (+W2^T)", "Gang update", "scatter (runs, private rows)", "per-atom flush", "W_ang^T contraction",
          "bond-weight grads", "forward recomputation"]
    for k, label, launches in ((4, "bondconv_bwd per atom", 3), (5, "angleupd_bwd per atom", 2)):
        v = ph6[k]; tot = v.sum()
        if tot <= 0: continue
        print(f"{label}: {tot / (tiles * launches):8.0f} cycles per (unpadded) wave-tile")
        for i, nm in enumerate(wn):
            if v[i] > 0: print(f"   {nm:30s} {v[i] / (tiles * launches):8.0f}  {100 * v[i] / tot:5.1f} %")
except Exception as e:
    print("per-atom phases unavailable:", e)

try:
    ph8 = eng.debug_fetch(batch, "phase", 8 * 2 * 10 * PH_WAVES).reshape(8, 2, 10, PH_WAVES).sum(axis=(1, 3))
    v = ph8[7]; tot = v.sum(); et = pb.n_directed / 16
    if tot > 0:
        print(f"atomconv_bwd: {tot / (et * 4):8.0f} cycles per wave-tile")
        for i, nm in enumerate(["requests + table sums read", "forward recomputation", "bond-weight gradient rows", "gated adjoint", "next gathers issued",
                                "scatter (GQ rows, run sums)", "next gathers landed"]):
            print(f"   {nm:30s} {v[i] / (et * 4):8.0f}  {100 * v[i] / tot:5.1f} %")
except Exception as e:
    print("atomconv phases unavailable:", e)
