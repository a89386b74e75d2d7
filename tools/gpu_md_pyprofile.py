"""cProfile of the MD loop (bench config C4): where the host side of a calculator call goes."""
import cProfile, os, pstats, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from chgnet_amd.calculator import CHGNetCalculator
from chgnet_amd.md import BerendsenNVT
from chgnet_amd.model import CHGNet
W = dict(np.load(os.path.join(REPO, "tests/golden/weights_seed0.npz")))
model = CHGNet(state_dict=W)
md = BerendsenNVT(bench.li9co7o16_supercell(), CHGNetCalculator(model), temperature_K=1000.0, timestep_fs=2.0, task="ef")
md.run(20)
out = md.run(300); print("steps/s", out["steps_per_s"])
pr = cProfile.Profile(); pr.enable(); md.run(300); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
