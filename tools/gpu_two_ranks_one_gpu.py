"""Can two RCCL ranks (two processes) share ONE GPU?  If RCCL allows it, the data-parallel collectives get a real 2-rank test on the
1-GPU boxes.  python tools/gpu_two_ranks_one_gpu.py  (spawns rank 1 itself)"""
import os, subprocess, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rank = int(os.environ.get("RANK", "-1"))
if rank < 0:
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0",
               CHGNET_RCCL_TOKEN="two-on-one")
    procs = [subprocess.Popen([sys.executable, __file__], env=dict(env, RANK=str(r), LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            p.kill(); out = p.communicate()[0] + "\n[timeout]"
        print(f"--- rank {r} rc={p.returncode}\n{out[-1500:]}")
    sys.exit(0)
from chgnet_amd.distributed import RcclComm
t = time.time()
try:
    comm = RcclComm(rank, 2, 0, timeout_s=60)
    print("communicator:", comm.info(), f"{time.time() - t:.1f}s")
    g = comm.all_gather(np.array([rank + 1.0, 10.0 * (rank + 1)], np.float32))
    s = comm.all_reduce_sum(np.array([1.0 + rank, 2.0], np.float32))
    print("all_gather", g, "all_reduce", s)
    comm.close()
except Exception as exc:  # noqa: BLE001
    print("FAILED:", type(exc).__name__, exc)
