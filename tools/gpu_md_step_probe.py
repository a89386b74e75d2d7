"""One MD-like step on FRESH batches (positions perturbed each step, graph rebuilt on the device, first prediction of the batch = eager
launches): wall time per phase and per-label HIP-event times.  python tools/gpu_md_step_probe.py [steps] [scale] [sigma_A]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_case
from chgnet_amd import Structure
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_weights
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
scale = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,2,2").split(",")]
sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 0.08
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
_, d = load_case("li9co7o16")
s0 = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"]).make_supercell(scale)
rng = np.random.default_rng(0)
inv = np.linalg.inv(s0.lattice.matrix)
structs = [Structure(s0.lattice, s0.atomic_numbers, s0.frac_coords + rng.normal(0, sigma, (len(s0), 3)) @ inv) for _ in range(8)]
eng = Engine(pack_weights(W), 0)
if os.environ.get("PROBE_GRAPH_SEARCH"):
    eng.set_graph_search(os.environ["PROBE_GRAPH_SEARCH"])
for profile in (False, True):
    eng.profile(profile)
    tb = tp = td = 0.0
    for it in range(steps + 5):
        if it == 5:
            tb = tp = td = 0.0
            if profile: eng.profile_reset()
        s = structs[it % len(structs)]
        t0 = time.perf_counter(); batch = eng.build_batch([s], 6.0, 3.0); t1 = time.perf_counter()
        eng.predict(batch, "ef"); t2 = time.perf_counter()
        eng.download(batch, "ef"); t3 = time.perf_counter()
        if it == 0:
            pb = batch.packed
            print(f"{len(s)} atoms: Ed={pb.n_directed} A={pb.n_angles} Eb={pb.n_bnodes} win_flag={eng.debug_fetch_i32(batch, 'win_flag', 4)}", flush=True)
        batch.free()
        tb += t1 - t0; tp += t2 - t1; td += t3 - t2
    print(f"profile={profile}: build {1e6 * tb / steps:.1f} us, predict (launch) {1e6 * tp / steps:.1f} us, download (wait) {1e6 * td / steps:.1f} us, "
          f"sum {1e6 * (tb + tp + td) / steps:.1f} us", flush=True)
prof = eng.profile_read()
for k, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k:18s} launches={cnt // steps:3d} per-step={1e3 * ms / steps:8.1f} us avg={1e3 * ms / max(cnt, 1):7.1f} us", flush=True)
