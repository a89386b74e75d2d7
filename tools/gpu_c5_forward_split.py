"""Timing probe (GPU box): the pieces of CHGNet.forward on a packed 1024-structure batch (the 'forward' segment of a fine-tuning step)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd import CrystalGraphConverter
from chgnet_amd.model import CHGNet
from chgnet_amd.pack import pack_batch
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
model = CHGNet(state_dict=W)
conv = CrystalGraphConverter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
packed = pack_batch([conv(s) for s in bench.workload_structures(n, 1000)])
eng = model.engine
print("host bytes of the packed batch: %.1f MB" % (sum(a.nbytes for a in packed.arrays.values()) / 1e6))
model.forward(packed, task="efsm")
gf = np.ones((packed.n_atoms, 3), np.float32) * 0.01
model.backward(np.ones(n, np.float32), None, gf, None)          # training workspace attached to the batch, as in a step
for rep in range(3):
    T = {}
    t = time.perf_counter(); model.release_forward_state(); eng.synchronize(); T["release previous batch (+ workspace)"] = time.perf_counter() - t
    t = time.perf_counter(); b = eng.upload(packed); T["upload: enqueue"] = time.perf_counter() - t
    t = time.perf_counter(); eng.synchronize(); T["upload: drain"] = time.perf_counter() - t
    t = time.perf_counter(); eng.predict(b, "efsm"); T["predict: enqueue"] = time.perf_counter() - t
    t = time.perf_counter(); eng.synchronize(); T["predict: device"] = time.perf_counter() - t
    t = time.perf_counter(); res = eng.download(b, "efsm"); T["download"] = time.perf_counter() - t
    t = time.perf_counter()
    off = packed.atom_off
    f = [res["f"][off[i]:off[i + 1]] for i in range(n)]; m = [res["m"][off[i]:off[i + 1]] for i in range(n)]; s = [res["s"][i] for i in range(n)]
    T["per-structure views"] = time.perf_counter() - t
    model._fwd_batch, model._fwd_task = b, "efsm"
    model.backward(np.ones(n, np.float32), None, gf, None)
    t = time.perf_counter(); out = model.forward(packed, task="efsm"); T["model.forward, whole (for comparison)"] = time.perf_counter() - t
    model.backward(np.ones(n, np.float32), None, gf, None)
    print(f"rep {rep}")
    for k, v in T.items():
        print(f"    {k:44s} {1e3 * v:8.2f} ms")
