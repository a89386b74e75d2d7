import sys; sys.path.insert(0, '.')
import numpy as np
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_weights
eng = Engine(pack_weights(dict(np.load('tests/golden/weights_seed0.npz'))), 0)
for nb in (1 << 28, 1 << 30, 1 << 32):
    print(nb >> 20, "MiB:", [round(eng.stream_copy_gbs(nb, 5)) for _ in range(2)], "GB/s", flush=True)
