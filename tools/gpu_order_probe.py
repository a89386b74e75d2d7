"""Timing experiment (GPU box): the six tile kernels with the angle rows in the reference's order (sorted by owning
bond) and in centre-atom-major order (stable sort by centre), same library, same process.  The per-structure results
must agree to fp32 reassociation; prints per-kernel averages for both orders.
usage: CHGNET_HIP_LIB=... python tools/gpu_order_probe.py [n_structures] [phase]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd.engine import Engine
from chgnet_amd.pack import PackedBatch, pack_batch, pack_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
phase = len(sys.argv) > 2 and sys.argv[2] == "phase"
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
pb = pack_batch(bench.build_workload(n, 0))


def reorder(pb, key):
    perm = np.argsort(key, kind="stable")
    arr = dict(pb.arrays)
    for k in ("a_ctr", "a_b1", "a_d1", "a_b2", "a_d2", "a_b1c", "a_b2c"):
        arr[k] = np.ascontiguousarray(arr[k][perm])
    return PackedBatch(pb.n_struct, pb.n_atoms, pb.n_directed, pb.n_undirected, pb.n_angles, pb.n_bnodes, arr)


def stats(pb, label):
    A = pb.n_angles
    nt = A // 128
    b2 = pb.arrays["a_b2c"][: nt * 128].reshape(nt, 128)
    b1 = pb.arrays["a_b1c"][: nt * 128].reshape(nt, 128)
    ct = pb.arrays["a_ctr"][: nt * 128].reshape(nt, 128)
    s = slice(0, min(nt, 2000))
    d2 = np.mean([len(np.unique(r)) for r in b2[s]])
    d1 = np.mean([len(np.unique(r)) for r in b1[s]])
    du = np.mean([len(np.unique(np.concatenate([x, y]))) for x, y in zip(b1[s], b2[s])])
    dc = np.mean([len(np.unique(r)) for r in ct[s]])
    print(f"[{label}] per 128-angle workgroup tile: distinct b1 {d1:.1f}, b2 {d2:.1f}, b1|b2 {du:.1f}, centres {dc:.1f}", flush=True)


eng = Engine(pack_weights(W), 0)
if phase:
    os.environ["CHGNET_HIP_GRAPHS"] = "0"
results = {}
orders = {"bond-major (reference)": pb, "centre-major": reorder(pb, pb.arrays["a_ctr"])}
for label, p in orders.items():
    stats(p, label)
    batch = eng.upload(p)
    eng.profile(True)
    for it in range(3):
        eng.predict(batch, "efs"); eng.synchronize()
        if it == 0: eng.profile_reset()
    prof = eng.profile_read()
    eng.profile(False)
    line = " | ".join(f"{k} {ms / max(c, 1):.3f}" for k, (c, ms) in sorted(prof.items()) if "conv_" in k or "angleupd_" in k)
    t = time.time()
    for _ in range(3): eng.predict(batch, "efs")
    eng.synchronize(); dt = (time.time() - t) / 3
    print(f"[{label}] {line} | steady {dt * 1e3:.2f} ms -> {n / dt:.0f} structures/s", flush=True)
    results[label] = eng.download(batch, "efs")
    if phase:
        ph = eng.debug_fetch(batch, "phase", 64)
        names = {0: ["idx+ang rows", "table gather", "W_ang GEMM", "gated fwd", "output scatter"],
                 10: ["idx+ang rows", "table gather", "W_ang GEMM", "gated fwd", "rows+dE/dy+Gwbgc", "gated bwd", "W_ang^T GEMM", "Gang update", "GR/GS scatter"]}
        tiles = p.n_angles / 16
        # the phase counters are zeroed at the top of every predict
        for base, lab, launches in ((0, "bondconv_fwd", 3), (10, "bondconv_bwd", 3), (20, "angleupd_fwd", 2), (30, "angleupd_bwd", 2)):
            v = ph[base:base + 10]
            tot = v.sum()
            print(f"  {lab}: {tot / (tiles * launches):8.0f} cycles per wave-tile: " +
                  ", ".join(f"{nm} {v[i] / (tiles * launches):.0f}" for i, nm in enumerate(names[10 if base % 20 else 0])), flush=True)
    batch.free()
a, b = results.values()
print("agreement between orders: e %.2e f %.2e s %.2e" % (np.abs(a["e"] - b["e"]).max(), np.abs(a["f"] - b["f"]).max(), np.abs(a["s"] - b["s"]).max()), flush=True)
