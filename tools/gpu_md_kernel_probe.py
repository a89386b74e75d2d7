"""Per-kernel HIP-event times of ONE resident MD-size prediction (2x2x2 Li9Co7O16 = 256 atoms by default; task ef): which of the ~57
dependent launches cost what.  python tools/gpu_md_kernel_probe.py [reps] [scale "2,2,2"]; the library is CHGNET_HIP_LIB."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_case
from chgnet_amd import Structure
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_weights
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
scale = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,2,2").split(",")]
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
_, d = load_case("li9co7o16")
s = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"]).make_supercell(scale)
eng = Engine(pack_weights(W), 0)
R = int(os.environ.get("PROBE_REPLICAS", "1"))
rng = np.random.default_rng(0)
inv = np.linalg.inv(s.lattice.matrix)
structs = [Structure(s.lattice, s.atomic_numbers, s.frac_coords + (rng.normal(0, 0.08, (len(s), 3)) @ inv if R > 1 else 0.0)) for _ in range(R)]
t0 = time.perf_counter(); batch = eng.build_batch(structs, 6.0, 3.0); batch.free()
t0 = time.perf_counter(); batch = eng.build_batch(structs, 6.0, 3.0); print(f"build_batch of {R} structure(s): {1e6 * (time.perf_counter() - t0):.1f} us", flush=True)
pb = batch.packed
print(f"{len(s)} atoms: Ed={pb.n_directed} Eu={pb.n_undirected} A={pb.n_angles} Eb={pb.n_bnodes}", flush=True)
for _ in range(10):
    eng.predict(batch, "ef"); eng.download(batch, "ef")
t0 = time.perf_counter()
for _ in range(10 * reps):
    eng.predict(batch, "ef"); eng.download(batch, "ef")
dt = time.perf_counter() - t0
print(f"predict + download, replayed: {1e5 * dt / reps:.1f} us", flush=True)
eng.profile(True)
for it in range(reps + 1):
    eng.predict(batch, "ef"); eng.synchronize()
    if it == 0: eng.profile_reset()
prof = eng.profile_read()
tot = sum(ms for _, ms in prof.values()) / reps
nl = sum(c for c, _ in prof.values()) // reps
for k, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:18s} launches={cnt // reps:3d} per-step={1e3 * ms / reps:8.1f} us avg={1e3 * ms / max(cnt, 1):7.1f} us", flush=True)
print(f"label sum {1e3 * tot:.1f} us over {nl} labelled scopes")
