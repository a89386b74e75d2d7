"""BASELINE config 4 probe (GPU box): NVT MD of 2x2x2 Li9Co7O16 (256 atoms) through CHGNetCalculator."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_case
from chgnet_amd import Structure
from chgnet_amd.graph.structure import Lattice
from chgnet_amd.calculator import CHGNetCalculator
from chgnet_amd.md import BerendsenNVT
from chgnet_amd.model import CHGNet
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
_, d = load_case("li9co7o16")
s = Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"]).make_supercell([2, 2, 2])
calc = CHGNetCalculator(CHGNet(state_dict=W))
md = BerendsenNVT(s, calc, temperature_K=1000.0, timestep_fs=2.0, task="ef")
md.run(5)
t = time.perf_counter(); calc.model.predict_structure(md.structure, task="ef"); tps = time.perf_counter() - t
print(f"predict_structure (device graph build + predict + download) {tps*1e3:.2f} ms", flush=True)
t = time.perf_counter(); g = calc.model.graph_converter(md.structure); tg = time.perf_counter() - t
t = time.perf_counter(); calc.model.predict_graph(g, task="ef"); tp = time.perf_counter() - t
print(f"graph build {tg*1e3:.2f} ms, predict_graph (pack+upload+predict+download) {tp*1e3:.2f} ms, N={len(s)} Ed={len(g.atom_graph)} A={len(g.bond_graph)}", flush=True)
out = md.run(n_steps)
print("exact rebuild every step:", out, flush=True)
