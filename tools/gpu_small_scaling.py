"""Per-kernel time of the eager sweep as the batch grows from one 256-atom cell: where the fixed cost per launch sits."""
import os, sys
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
names = ("bondconv_bwd", "atomconv_bwd", "bondconv_fwd", "angleupd_bwd", "gemm_R", "gemm_GR", "atomconv_fwd", "gemm_P", "readout")
print("copies  angles  " + "  ".join(f"{n:>12s}" for n in names) + "   total_ms")
for copies in (1, 2, 4, 8, 16, 32, 64):
    b = eng.build_batch([s] * copies)
    eng.predict(b, "ef"); eng.synchronize()
    best = None
    for _ in range(5):
        eng.profile(True); eng.profile_reset()
        eng.predict(b, "ef"); eng.synchronize()
        prof = eng.profile_read()
        eng.profile(False)
        row = {k: ms / c for k, (c, ms) in prof.items()}
        row["_total"] = sum(ms for _, ms in prof.values())
        if best is None or row["_total"] < best["_total"]:
            best = row
    print(f"{copies:6d} {b.packed.n_angles:7d}  " + "  ".join(f"{1e3 * best.get(n, float('nan')):10.1f}us" for n in names) + f"   {best['_total']:.3f}")
    b.free()
