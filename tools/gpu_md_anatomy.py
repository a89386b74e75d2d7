"""Where one MD step of BASELINE config 4 goes (GPU box): host wall clock around the engine calls of CHGNetCalculator's exact-rebuild path."""
import os, sys, time, collections
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_case
from chgnet_amd import Structure
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.calculator import CHGNetCalculator
from chgnet_amd.md import BerendsenNVT
from chgnet_amd.model import CHGNet
from chgnet_amd import engine as engine_mod

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
scale = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,2,2").split(",")]
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
_, d = load_case("li9co7o16")
s = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"]).make_supercell(scale)
calc = CHGNetCalculator(CHGNet(state_dict=W))
md = BerendsenNVT(s, calc, temperature_K=1000.0, timestep_fs=2.0, task="ef")
md.run(20)
acc = collections.defaultdict(float)


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label or name] += time.perf_counter() - t
    setattr(obj, name, wrapper)


eng = calc.model.engine
for name in ("prepare_structures", "build_prepared", "predict", "download"):
    timed(eng, name)
timed(engine_mod.DeviceBatch, "free", "batch.free")
timed(calc.model, "predict_structure")
timed(calc, "calculate")
t0 = time.perf_counter()
out = md.run(n_steps)
wall = time.perf_counter() - t0
print(f"{len(s)} atoms: {n_steps / wall:.0f} steps/s, {wall / n_steps * 1e3:.3f} ms per step", out)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:22s} {v / n_steps * 1e6:8.1f} us per step")
inner = sum(acc[k] for k in ("prepare_structures", "build_prepared", "predict", "download", "batch.free"))
print(f"  predict_structure outside the engine calls {(acc['predict_structure'] - inner) / n_steps * 1e6:8.1f} us;  calculate outside predict_structure "
      f"{(acc['calculate'] - acc['predict_structure']) / n_steps * 1e6:8.1f} us;  integrator outside calculate {(wall - acc['calculate']) / n_steps * 1e6:8.1f} us")
