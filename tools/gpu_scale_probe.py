"""Timing probe (GPU box): per-kernel profile at a given batch size."""
import os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_batch, pack_weights
n = int(sys.argv[1]); task = sys.argv[2] if len(sys.argv) > 2 else "efs"
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
t = time.time(); graphs = bench.build_workload(n, 0); print(f"build {n}: {time.time()-t:.2f}s", flush=True)
t = time.time(); pb = pack_batch(graphs); print(f"pack: {time.time()-t:.2f}s N={pb.n_atoms} Ed={pb.n_directed} A={pb.n_angles} Eb={pb.n_bnodes}", flush=True)
eng = Engine(pack_weights(W), 0)
t = time.time(); batch = eng.upload(pb); print(f"upload: {time.time()-t:.2f}s bytes={batch.device_bytes/1e9:.2f}GB", flush=True)
eng.profile(True)
for it in range(2):
    t = time.time(); eng.predict(batch, task); print(f"enqueue {time.time()-t:.3f}s", flush=True); eng.synchronize(); print(f"predict[{it}] {time.time()-t:.3f}s", flush=True)
    if it == 0: eng.profile_reset()
prof = eng.profile_read()
for k, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:18s} launches={cnt:4d} total={ms:9.3f} ms avg={ms/max(cnt,1):8.3f} ms", flush=True)
eng.profile(False)
t = time.time()
for _ in range(3): eng.predict(batch, task)
eng.synchronize(); dt = (time.time()-t)/3
print(f"steady: {dt*1e3:.2f} ms/step -> {n/dt:.1f} structures/s", flush=True)
t = time.time(); res = eng.download(batch, task); print(f"download {time.time()-t:.3f}s finite={np.isfinite(res['e']).all()}", flush=True)
# end-to-end variants (host buffers in -> results on host)
import bench as _b
from chgnet_amd import Structure
from chgnet_amd.graph.structure import Lattice
lat = Lattice.from_parameters(2.868779, 4.634475, 5.832507, 90, 90, 90)
frac = [[0.5, 0.5, 0.3797505], [0, 0, 0.6202495], [0.5, 0.5, 0.8632525], [0, 0, 0.1367475],
        [0.5, 0, 0.3608245], [0, 0.5, 0.0985135], [0.5, 0, 0.9014865], [0, 0.5, 0.6391755]]
base = Structure(lat, ["Li", "Li", "Mn", "Mn", "O", "O", "O", "O"], frac).make_supercell([5, 1, 1])
structs = [base.perturb(0.01, np.random.default_rng(i)) for i in range(n)]
batch.free()
for rep in range(2):
    t = time.time(); b2 = eng.build_batch(structs); tb = time.time() - t
    t = time.time(); eng.predict(b2, task); r2 = eng.download(b2, task); tp = time.time() - t
    b2.free()
    print(f"structures -> device graph build {tb*1e3:.1f} ms + predict/download {tp*1e3:.1f} ms = {n/(tb+tp):.0f} structures/s end to end", flush=True)
for rep in range(2):
    t = time.time(); b3 = eng.upload(pb); tu = time.time() - t
    t = time.time(); eng.predict(b3, task); r3 = eng.download(b3, task); tp = time.time() - t
    b3.free()
    print(f"host graphs -> upload {tu*1e3:.1f} ms + predict/download {tp*1e3:.1f} ms = {n/(tu+tp):.0f} structures/s (PCIe-inclusive)", flush=True)
print("e agree:", float(np.abs(r2["e"] - r3["e"]).max()), flush=True)
