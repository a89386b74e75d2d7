"""Per-phase shader clocks of the tile kernels at MD size (one resident 256-/512-atom batch, ONE eager prediction; needs a
-DCHG_PHASE_TIMING build in CHGNET_HIP_LIB): per kernel the mean and the slowest wave's total, split into first / later tiles."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ["CHGNET_HIP_GRAPHS"] = "0"
from conftest import load_case
from chgnet_amd import Structure
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_weights
scale = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,2,2").split(",")]
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
_, d = load_case("li9co7o16")
s = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"]).make_supercell(scale)
eng = Engine(pack_weights(W), 0)
batch = eng.build_batch([s], 6.0, 3.0)
for _ in range(3):
    eng.predict(batch, "ef"); eng.synchronize()
PH_WAVES = 4096
ph = eng.debug_fetch(batch, "phase", 8 * 2 * 10 * PH_WAVES).reshape(8, 2, 10, PH_WAVES) / 3.0
names = {0: "bondconv_fwd x3", 1: "bondconv_bwd row x3", 2: "angleupd_fwd x2", 3: "angleupd_bwd row x2", 4: "bondconv_bwd team x3", 5: "angleupd_bwd team x2",
         7: "atomconv_bwd x4"}
launches = {0: 3, 1: 3, 2: 2, 3: 2, 4: 3, 5: 2, 7: 4}
pb = batch.packed
print(f"{len(s)} atoms: Ed={pb.n_directed} A={pb.n_angles} Eb={pb.n_bnodes}")
for k, nm in names.items():
    v = ph[k] / launches[k]                       # [first/later][slot][wave], per launch
    per_wave = v.sum(axis=(0, 1))
    busy = per_wave > 0
    if not busy.any(): continue
    print(f"{nm}: {busy.sum()} waves with work; per wave and launch: mean {per_wave[busy].mean():8.0f} clocks, max {per_wave.max():8.0f}")
    for which, lab in ((1, "first tile"), (0, "later tiles")):
        tot = v[which][:, busy].mean(axis=1)
        if tot.sum() > 0:
            print("   " + lab + ": " + "  ".join(f"[{i}] {t:6.0f}" for i, t in enumerate(tot) if t > 0))
# entry / exit stamps of the LAST launch of every kernel (kernel slot 6): when its waves really start and end
st = eng.debug_fetch(batch, "phase", 8 * 2 * 10 * PH_WAVES).reshape(8, 20, PH_WAVES)[6].view(np.uint32).reshape(10, 2, PH_WAVES)[:8]
for k, nm in names.items():
    a, e = st[k, 0].astype(np.int64), st[k, 1].astype(np.int64)
    busy = e != 0
    if not busy.any(): continue
    a, e = a[busy], e[busy]
    t0 = a.min()
    d = (e - a)
    print(f"{nm}: {busy.sum()} waves; entry spread {np.percentile(a - t0, [0, 50, 90, 100]).astype(int)}, exit - first entry {np.percentile(e - t0, [0, 50, 90, 100]).astype(int)}, "
          f"wave lifetime {np.percentile(d, [0, 50, 90, 100]).astype(int)} clocks")
    xcd = (np.flatnonzero(busy) // 8) % 8        # workgroup = wave // 8; XCD = workgroup % 8 (observed dispatch)
    print("   first entry per XCD:", [int((a[xcd == x] - t0).min()) for x in range(8) if (xcd == x).any()])
