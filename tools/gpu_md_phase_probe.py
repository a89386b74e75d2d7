"""MD-size batches: where one launch of the angle kernels goes (needs a -DCHG_PHASE_TIMING build in CHGNET_HIP_LIB).

s_memtime ticks per phase, summed over the waves by the kernel: the prologue (slot 9) is paid by every wave of the grid, the
tile phases by the waves that own a wave-tile.  Printed per wave / per wave-tile, next to the launch's duration (HIP events)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ["CHGNET_HIP_GRAPHS"] = "0"
from conftest import load_case
from chgnet_amd import Structure
from chgnet_amd.engine import Engine
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.pack import pack_weights

W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
_, d = load_case("li9co7o16")
eng = Engine(pack_weights(W), 0)
NAMES = {False: ["idx+ang rows", "table gather", "W_ang GEMM", "gated fwd", "output scatter"],
         True: ["idx+ang rows", "table gather", "W_ang GEMM", "gated fwd", "rows+dE/dy+Gwbgc", "gated bwd", "W_ang^T GEMM", "Gang update", "GR/GS scatter"]}
PH_WAVES = 4096
KERNEL = {0: "bondconv_fwd", 10: "bondconv_bwd", 20: "angleupd_fwd", 30: "angleupd_bwd"}


def probe(structs, label):
    b = eng.build_batch(structs)
    pb = b.packed
    for _ in range(3):
        eng.predict(b, "efs")
    eng.synchronize()
    ph = eng.debug_fetch(b, "phase", 4 * 2 * 10 * PH_WAVES).reshape(4, 2, 10, PH_WAVES)   # [kernel][later / first tile][slot][wave]
    eng.profile(True)
    for _ in range(20):
        eng.predict(b, "efs")
    eng.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    tiles = (pb.n_angles + 15) // 16
    cus = int(os.environ.get("CHG_CUS", "256"))
    waves = min(PH_WAVES, 8 * cus * (1 if tiles <= 4 * 8 * cus else 2))          # engine.hip tile_grid
    per_wave = np.array([(w + 1) * tiles // waves - w * tiles // waves for w in range(waves)])
    print(f"== {label}: N={pb.n_atoms} Ed={pb.n_directed} A={pb.n_angles} wave-tiles={tiles} waves={waves} "
          f"(tiles per wave {per_wave.min()}..{per_wave.max()})")
    for k, launches in ((0, 3), (1, 3), (2, 2), (3, 2)):
        name = KERNEL[10 * k]
        ms = prof.get(name, (0, 0.0))
        v = ph[k, :, :, :waves] / launches
        total = v.sum(axis=(0, 1))                      # ticks per wave and launch, prologue to last phase
        print(f"{name}: launch {1e3 * ms[1] / max(1, ms[0]):7.1f} us (timing build)   per wave: total mean {total.mean():8.0f} max {total.max():8.0f} ticks;"
              f" prologue mean {v[1, 9].mean():7.0f} max {v[1, 9].max():7.0f}")
        later = np.maximum(per_wave - 1, 0)
        print(f"   {'phase':20s} {'first tile':>10s} {'(max)':>8s} {'later, per tile':>16s} {'(max)':>8s}")
        for i, nm in enumerate(NAMES[bool(k % 2)]):
            lt = v[0, i][later > 0] / later[later > 0]
            print(f"   {nm:20s} {v[1, i].mean():10.0f} {v[1, i].max():8.0f} {lt.mean() if lt.size else 0:16.0f} {lt.max() if lt.size else 0:8.0f}")
    b.free()


s0 = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"])
probe([s0.make_supercell([2, 2, 2])], "MD cell 2x2x2")
probe([s0.make_supercell([4, 2, 2])], "MD cell 4x2x2")
import bench
probe(bench.workload_structures(256, 0), "256 structures (tick calibration: many tiles per wave)")
