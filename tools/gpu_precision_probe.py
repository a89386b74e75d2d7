"""Engine vs the float64 oracle (and the fp32 reference fixture) on the golden cases at trained-checkpoint magnitudes."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from conftest import load_case
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_weights
from oracle.chgnet_oracle import OracleCHGNet
for wname, pre in (("weights_seed0.npz", "out_"), ("weights_trained_like.npz", "tl_out_")):
    W = dict(np.load(os.path.join(REPO, "tests/golden", wname)))
    eng = Engine(pack_weights(W), 0)
    torch.set_num_threads(8)
    o64 = OracleCHGNet(W, dtype=torch.float64)
    for name in ("limno2", "s40", "s16tri", "li9co7o16"):
        g, d = load_case(name)
        b = eng.upload([g]); eng.predict(b, "efsm"); r = eng.download(b, "efsm"); b.free()
        t = o64.predict_graph(g, "efsm")
        n = len(g.atomic_number)
        line = f"{wname[8:-4]:13s} {name:10s} |F|max {np.abs(t['f']).max():6.2f}  engine-fp64: e {abs(r['e'][0]-t['e']):.1e} f {np.abs(r['f'][:n]-t['f']).max():.1e} s {np.abs(r['s'][0]-t['s']).max():.1e}"
        line += f"   ref32-fp64: e {abs(d[pre+'e']-t['e']):.1e} f {np.abs(d[pre+'f']-t['f']).max():.1e} s {np.abs(d[pre+'s']-t['s']).max():.1e}"
        line += f"   engine-ref32: f {np.abs(r['f'][:n]-d[pre+'f']).max():.1e}"
        print(line, flush=True)
    eng.close()
