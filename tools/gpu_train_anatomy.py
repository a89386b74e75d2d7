"""Anatomy of one fine-tuning step on the host side (GPU box): where the ~55 ms of ``forward`` and the ~20 ms of ``loss`` go."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd import CrystalGraphConverter
from chgnet_amd.model import CHGNet
from chgnet_amd.pack import pack_batch
from chgnet_amd.trainer import TrainStep
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
model = CHGNet(state_dict=W)
conv = CrystalGraphConverter(atom_graph_cutoff=6, bond_graph_cutoff=3)
batches = [[conv(s) for s in bench.workload_structures(n, 1000 + i * n)] for i in range(3)]
rng = np.random.default_rng(0)
def labels(b_):
    return {"e": rng.normal(-7, 0.05, len(b_)).astype(np.float32), "f": [rng.normal(0, 0.05, (len(g.atomic_number), 3)).astype(np.float32) for g in b_],
            "s": [rng.normal(0, 0.2, (3, 3)).astype(np.float32) for _ in b_], "m": [np.abs(rng.normal(0.5, 0.2, len(g.atomic_number))).astype(np.float32) for g in b_]}
lab = [labels(b) for b in batches]
eng = model.engine
T = lambda: time.perf_counter()
for rep in range(3):
    t0 = T(); packed = pack_batch(batches[rep]); t1 = T()
    db = eng.upload(packed); eng.synchronize(); t2 = T()
    eng.predict(db, "efsm"); eng.synchronize(); t3 = T()
    res = eng.download(db, "efsm"); t4 = T()
    db.free(); t5 = T()
    print(f"pack {1e3*(t1-t0):.1f}  upload {1e3*(t2-t1):.1f}  predict(first, eager) {1e3*(t3-t2):.1f}  download {1e3*(t4-t3):.1f}  free {1e3*(t5-t4):.1f} ms", flush=True)
step = TrainStep(model, targets="efsm", learning_rate=1e-4)
step(batches[0], lab[0])
import gc
if os.environ.get("ANATOMY_NOGC"):
    gc.disable()
print("gc enabled:", gc.isenabled(), "objects tracked:", len(gc.get_objects()), flush=True)
_orig = step.loss.gradients
def _timed(targets, pred, flat_targets=None):
    t0 = T(); out = _orig(targets, pred, flat_targets=flat_targets); dt = 1e3 * (T() - t0)
    print(f"    loss.gradients call: {dt:.1f} ms  flat given: {flat_targets is not None}  pred.flat: {getattr(pred, 'flat', None) is not None}", flush=True)
    return out
step.loss.gradients = _timed
for rep in range(2):
    step.seconds.clear()
    t0 = T(); infos = step.run_epoch(batches, lab); eng.synchronize(); dt = T() - t0
    print(f"epoch of 3 steps: {1e3*dt/3:.1f} ms per step; split " + ", ".join(f"{k} {1e3*v/3:.1f}" for k, v in step.seconds.items() if k != "calls"), flush=True)
# ---- the loss call alone / next to a packing thread -----------------------------------------------------------------------
import threading
pred = model.forward(batches[1], task="efsm")
flat = step.loss.flatten_targets(lab[1], pred["atoms_per_graph"])
for what in ("alone", "next to pack_batch", "next to flatten_targets", "alone again"):
    th = None
    if what == "next to pack_batch":
        th = threading.Thread(target=lambda: [pack_batch(batches[2]) for _ in range(3)])
    elif what == "next to flatten_targets":
        th = threading.Thread(target=lambda: [step.loss.flatten_targets(lab[2], pred["atoms_per_graph"]) for _ in range(20)])
    if th: th.start()
    ts = []
    for _ in range(5):
        t0 = T(); step.loss.gradients(lab[1], pred, flat_targets=flat); ts.append(1e3 * (T() - t0))
    if th: th.join()
    print(f"loss.gradients {what}: " + " ".join(f"{x:.1f}" for x in ts) + " ms", flush=True)
model.release_forward_state()
