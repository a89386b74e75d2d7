"""Multi-GPU driver: structures are independent, so they are sharded across ranks (one process per
GPU, ``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm) with NO collective on the data
path.  The only exchange is an optional all-gather of the per-structure energies (<= 4 B per
structure) so every rank ends up with the full energy table of a dataset sweep.

The reference has nothing to port here: it is single-process, single-device
(SURVEY 2 "Parallelism strategies ... none").
"""

from __future__ import annotations

import heapq
from collections.abc import Callable, Sequence

import numpy as np


def structure_cost(graph) -> float:
    """Cost proxy of one structure = GEMM flops of the reference formulation.  A structure
    without a graph yet (``CHGNet.predict_structure`` builds it on the device) is charged by its atom count:
    bonds and angles per atom are set by the density, which varies little inside one dataset."""
    if not hasattr(graph, "atom_graph"):
        return float(len(graph))
    n_dir = len(graph.atom_graph)
    n_ang = len(graph.bond_graph)
    return 262144.0 * n_dir + 380800.0 * n_ang + 57472.0 * len(graph.atomic_number)


def shard_indices(costs: Sequence[float], world_size: int) -> list[list[int]]:
    """Longest-processing-time-first greedy partition: balanced sum of costs per rank."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    heap = [(0.0, r) for r in range(world_size)]
    heapq.heapify(heap)
    shards: list[list[int]] = [[] for _ in range(world_size)]
    for idx in order:
        load, r = heapq.heappop(heap)
        shards[r].append(int(idx))
        heapq.heappush(heap, (load + float(costs[idx]), r))
    return [sorted(s) for s in shards]


class RcclComm:
    """One RCCL communicator per process through the engine library's own entry points (``chg_comm_*``): the multi-GPU
    path without ``torch.distributed``.  Rank / world size / device come from the launcher's environment
    (``RANK``, ``WORLD_SIZE``, ``LOCAL_RANK``); rank 0's ``ncclUniqueId`` reaches the other ranks of the node through a
    file named after ``MASTER_PORT`` (written atomically, removed by rank 0 on ``close``)."""

    def __init__(self, rank: int | None = None, world: int | None = None, device: int | None = None, *, rendezvous_dir: str | None = None,
                 timeout_s: float = 120.0) -> None:
        import ctypes  # noqa: PLC0415
        import os  # noqa: PLC0415
        import tempfile  # noqa: PLC0415
        import time  # noqa: PLC0415

        from chgnet_amd import _lib  # noqa: PLC0415

        self.lib = _lib.load()
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        self.handle = ctypes.c_void_p()
        ident = (ctypes.c_uint8 * 128)()
        self._id_file = None
        if self.world > 1:
            root = rendezvous_dir or tempfile.gettempdir()
            path = os.path.join(root, f"chgnet_rccl_{os.environ.get('MASTER_PORT', '29500')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}.id")
            if self.rank == 0:
                self._check(self.lib.chg_comm_unique_id(ident))
                with open(path + ".tmp", "wb") as fh:
                    fh.write(bytes(ident))
                os.replace(path + ".tmp", path)
                self._id_file = path
            else:
                started = time.time()
                deadline = started + timeout_s
                # a file left behind by a run that died is older than this process by more than the launcher's spread
                while not (os.path.exists(path) and os.path.getmtime(path) >= started - 30.0):
                    if time.time() > deadline:
                        raise TimeoutError(f"RcclComm: rank 0 did not publish {path}")
                    time.sleep(0.01)
                with open(path, "rb") as fh:
                    ident = (ctypes.c_uint8 * 128).from_buffer_copy(fh.read(128))
        else:
            self._check(self.lib.chg_comm_unique_id(ident))
        self._check(self.lib.chg_comm_create(ident, self.rank, self.world, self.device, ctypes.byref(self.handle)))
        self.barrier()
        if self._id_file:                      # every rank has read it once the first barrier is through
            os.remove(self._id_file)
            self._id_file = None

    def _check(self, status: int) -> None:
        if status != 0:
            raise RuntimeError(f"chg_comm error {status}: {self.lib.chg_comm_last_error(self.handle).decode()}")

    def all_gather(self, values: np.ndarray) -> np.ndarray:
        """Equal-length float32 vectors of every rank, concatenated in rank order (ncclAllGather)."""
        import ctypes  # noqa: PLC0415

        send = np.ascontiguousarray(values, np.float32).reshape(-1)
        recv = np.empty(self.world * send.size, np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        self._check(self.lib.chg_comm_all_gather_f32(self.handle, send.ctypes.data_as(fp), send.size, recv.ctypes.data_as(fp)))
        return recv

    def all_reduce_sum(self, values: np.ndarray) -> np.ndarray:
        """Element-wise sum over the ranks (ncclAllReduce); returns a new float32 array."""
        import ctypes  # noqa: PLC0415

        data = np.array(values, dtype=np.float32, copy=True).reshape(-1)
        self._check(self.lib.chg_comm_all_reduce_sum_f32(self.handle, data.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), data.size))
        return data.reshape(np.shape(values))

    def barrier(self) -> None:
        self._check(self.lib.chg_comm_barrier(self.handle))

    def close(self) -> None:
        if self.handle:
            self.lib.chg_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def all_gather_energies(local_e: np.ndarray, shards: list[list[int]], n_total: int, device=None, comm: RcclComm | None = None) -> np.ndarray:
    """All-gather the per-structure energies of every rank into original order (padded to equal
    counts; one collective of ``4 * max_shard`` bytes per rank).  ``comm``: an ``RcclComm`` instead of the
    ``torch.distributed`` process group."""
    width = max(len(s) for s in shards)
    if comm is not None:
        mine = np.zeros(width, np.float32)
        mine[: len(shards[comm.rank])] = np.asarray(local_e, np.float32)
        table = comm.all_gather(mine).reshape(comm.world, width)
        out = np.empty(n_total, dtype=np.float32)
        for r, idxs in enumerate(shards):
            out[idxs] = table[r, : len(idxs)]
        return out
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.zeros(width, dtype=torch.float32, device=device)
    mine[: len(shards[rank])] = torch.as_tensor(np.asarray(local_e, dtype=np.float32), device=device)
    gathered = torch.empty(world * width, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(gathered, mine)
    table = gathered.cpu().numpy().reshape(world, width)
    out = np.empty(n_total, dtype=np.float32)
    for r, idxs in enumerate(shards):
        out[idxs] = table[r, : len(idxs)]
    return out


def predict_sharded(predict_fn: Callable, graphs: Sequence, *, task: str = "efs", gather_energies: bool = True,
                    comm: RcclComm | None = None, **kwargs):
    """Each rank predicts its own shard with ``predict_fn(list_of_graphs, task=..., **kwargs)``:
    ``CHGNet.predict_graph`` for CrystalGraphs, or ``CHGNet.predict_structure`` for structures -- the
    graph of every structure is then built on the owning rank's GPU (chg_batch_build), nothing is
    converted on the host.

    Returns ``(local, energies)``: ``local`` maps original structure index -> prediction dict for the
    structures this rank owns; ``energies`` is the all-gathered float32 table of every structure (or
    ``None`` when ``gather_energies`` is false / no process group exists)."""
    if comm is not None:                      # RCCL through the engine library: no torch needed
        have_group, world, rank = True, comm.world, comm.rank
    else:
        try:
            import torch.distributed as dist

            have_group = dist.is_available() and dist.is_initialized()
        except ImportError:
            have_group = False
        world = dist.get_world_size() if have_group else 1
        rank = dist.get_rank() if have_group else 0
    shards = shard_indices([structure_cost(g) for g in graphs], world)
    mine = shards[rank]
    preds = predict_fn([graphs[i] for i in mine], task=task, **kwargs) if mine else []
    if isinstance(preds, dict):
        preds = [preds]
    local = dict(zip(mine, preds))
    energies = None
    if gather_energies:
        local_e = np.array([float(p["e"]) for p in preds], dtype=np.float32)
        if have_group and world > 1:
            energies = all_gather_energies(local_e, shards, len(graphs), comm=comm)
        else:
            energies = np.empty(len(graphs), np.float32)
            energies[mine] = local_e
    return local, energies
