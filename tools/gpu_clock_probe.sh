#!/bin/bash
# shader clock / power while the headline step runs in a loop (is the chip power- or clock-limited under these kernels?)
R=$PWD
python - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from chgnet_amd.engine import Engine
from chgnet_amd.pack import pack_batch, pack_weights
W = dict(np.load("tests/golden/weights_seed0.npz"))
eng = Engine(pack_weights(W), 0)
batch = eng.upload(pack_batch(bench.build_workload(1024, 0)))
for _ in range(3): eng.predict(batch, "efs")
eng.synchronize()
t = time.time(); n = 0
while time.time() - t < 14:
    for _ in range(10): eng.predict(batch, "efs")
    eng.synchronize(); n += 10
print(f"loop: {n} steps, {1e3 * (time.time() - t) / n:.2f} ms/step", flush=True)
PY
sleep 6
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | head -8
  echo "--"; sleep 1.5
done
wait
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
rocm-smi --showmaxpower --showclkfrq 2>/dev/null | grep -E "Max Graphics Package Power|sclk|\*" | head -12
