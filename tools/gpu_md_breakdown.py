"""Where a single-structure (MD-size) step goes: kernel time vs launch overhead, eager vs hipGraph replay."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_case
from chgnet_amd import Structure
from chgnet_amd.engine import Engine
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.pack import pack_weights

W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
_, d = load_case("li9co7o16")
s = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"]).make_supercell([2, 2, 2])
eng = Engine(pack_weights(W), 0)
task = sys.argv[1] if len(sys.argv) > 1 else "ef"

N_REP = int(os.environ.get("MD_PROBE_REPS", "50"))


def wall(fn, n=None):
    n = n or N_REP
    fn(); eng.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    eng.synchronize()
    return (time.perf_counter() - t) / n * 1e3

b = eng.build_batch([s])
print(f"N={b.packed.n_atoms} Ed={b.packed.n_directed} A={b.packed.n_angles}")
eng.predict(b, task); eng.synchronize()            # first call eager, later calls replay the captured graph
print(f"replay predict            {wall(lambda: eng.predict(b, task)):.3f} ms")
print(f"replay predict + download {wall(lambda: (eng.predict(b, task), eng.download(b, task))):.3f} ms")
def rebuild():
    bb = eng.build_batch([s]); eng.predict(bb, task); r = eng.download(bb, task); bb.free()
print(f"build + eager predict + download {wall(rebuild):.3f} ms")
def build_only():
    bb = eng.build_batch([s]); bb.free()
print(f"build only {wall(build_only):.3f} ms")
eng.profile(True)
bb = eng.build_batch([s]); eng.predict(bb, task); eng.synchronize()
prof = eng.profile_read()
eng.profile(False)
tot = sum(ms for _, ms in prof.values()); n = sum(c for c, _ in prof.values())
print(f"eager: {n} launches, kernel time {tot:.3f} ms")
for k, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {k:18s} launches={cnt:4d} total={ms:7.3f} ms")

# clock-state check: the same replay right after a second of heavy work (a 512-structure batch), and interleaved with it
import bench
heavy = eng.build_batch(bench.workload_structures(512, 0))
eng.predict(heavy, "efsm"); eng.synchronize()
t = time.perf_counter()
while time.perf_counter() - t < 1.0:
    eng.predict(heavy, "efsm")
eng.synchronize()
print(f"replay predict right after 1 s of heavy work {wall(lambda: eng.predict(b, task), 4 * N_REP):.3f} ms")
print(f"replay predict, again                        {wall(lambda: eng.predict(b, task), 4 * N_REP):.3f} ms")
print(f"replay predict, 2000 in a row                {wall(lambda: eng.predict(b, task), 40 * N_REP):.3f} ms")
