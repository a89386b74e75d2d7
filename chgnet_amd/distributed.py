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


def exchange_unique_id(rank: int, world: int, ident: bytes | None, *, addr: str | None = None, port: int | None = None,
                       timeout_s: float = 120.0, nbytes: int = 128) -> bytes:
    """Single-node rendezvous of the RCCL communicator: rank 0 hands its ``ncclUniqueId`` (``ident``, ``nbytes`` bytes) to every
    other rank over a TCP socket on ``addr:port`` (default ``MASTER_ADDR`` : ``CHGNET_RCCL_PORT`` or ``MASTER_PORT + 1``, next to
    the launcher's own store).  Nothing is left on disk and nothing can be stale: a rank that arrives early retries the
    connection until rank 0 listens (any order, any delay up to ``timeout_s``), a rank that arrives late finds rank 0 still
    accepting -- rank 0 serves until all ``world - 1`` peers have been answered.  Every peer introduces itself with a
    launch token (``TORCHELASTIC_RUN_ID`` / ``CHGNET_RCCL_TOKEN``) and its rank; a connection from another job, a repeated rank
    or a rank out of range is refused and does not count.  Returns the id (rank 0: ``ident`` itself)."""
    import os  # noqa: PLC0415
    import socket  # noqa: PLC0415
    import struct  # noqa: PLC0415
    import time  # noqa: PLC0415

    if world <= 1:
        if ident is None:
            raise ValueError("exchange_unique_id: a single rank passes its own id")
        return ident
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    if port is None:
        port = int(os.environ["CHGNET_RCCL_PORT"]) if "CHGNET_RCCL_PORT" in os.environ else int(os.environ.get("MASTER_PORT", "29500")) + 1
    token = (os.environ.get("CHGNET_RCCL_TOKEN") or os.environ.get("TORCHELASTIC_RUN_ID") or "chgnet").encode()[:64].ljust(64, b"\0")
    deadline = time.monotonic() + timeout_s
    if rank == 0:
        if ident is None or len(ident) != nbytes:
            raise ValueError("exchange_unique_id: rank 0 passes the id")
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            served: set[int] = set()
            while len(served) < world - 1:
                left = deadline - time.monotonic()
                if left <= 0:
                    raise TimeoutError(f"RcclComm rendezvous: {world - 1 - len(served)} of {world - 1} ranks did not connect to {addr}:{port}")
                srv.settimeout(left)
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    conn.settimeout(5.0)
                    try:
                        hello = _recv_exact(conn, 68)
                    except (OSError, EOFError):
                        continue
                    peer = struct.unpack("<i", hello[64:])[0]
                    if hello[:64] != token or not (0 < peer < world) or peer in served:
                        continue                              # another job, or a repeated / impossible rank: not ours
                    conn.sendall(ident)
                    served.add(peer)
        return ident
    hello = token + struct.pack("<i", rank)
    while True:
        try:
            with socket.create_connection((addr, port), timeout=max(0.1, min(5.0, deadline - time.monotonic()))) as conn:
                conn.sendall(hello)
                return _recv_exact(conn, nbytes)
        except (OSError, EOFError):
            if time.monotonic() > deadline:
                raise TimeoutError(f"RcclComm rendezvous: rank {rank} could not fetch the id from {addr}:{port}") from None
            time.sleep(0.05)


def _recv_exact(conn, n: int) -> bytes:
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise EOFError("connection closed")
        buf += chunk
    return buf


class RcclComm:
    """One RCCL communicator per process through the engine library's own entry points (``chg_comm_*``): the multi-GPU
    path without ``torch.distributed``.  Rank / world size / device come from the launcher's environment
    (``RANK``, ``WORLD_SIZE``, ``LOCAL_RANK``); rank 0's ``ncclUniqueId`` reaches the other ranks of the node over a TCP
    socket (``exchange_unique_id``)."""

    def __init__(self, rank: int | None = None, world: int | None = None, device: int | None = None, *, timeout_s: float = 120.0,
                 addr: str | None = None, port: int | None = None) -> None:
        import ctypes  # noqa: PLC0415
        import os  # noqa: PLC0415

        from chgnet_amd import _lib  # noqa: PLC0415

        self.lib = _lib.load()
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        self.handle = ctypes.c_void_p()
        ident = (ctypes.c_uint8 * 128)()
        if self.rank == 0:
            self._check(self.lib.chg_comm_unique_id(ident))
        raw = exchange_unique_id(self.rank, self.world, bytes(ident) if self.rank == 0 else None, addr=addr, port=port, timeout_s=timeout_s)
        ident = (ctypes.c_uint8 * 128).from_buffer_copy(raw)
        self._check(self.lib.chg_comm_create(ident, self.rank, self.world, self.device, ctypes.byref(self.handle)))
        self.barrier()

    def info(self) -> dict:
        """What RCCL reports for this communicator: ``{"rank", "nranks" (ncclCommCount), "device"}``."""
        import ctypes  # noqa: PLC0415

        r, n, d = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        self._check(self.lib.chg_comm_info(self.handle, ctypes.byref(r), ctypes.byref(n), ctypes.byref(d)))
        return {"rank": int(r.value), "nranks": int(n.value), "device": int(d.value)}

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
