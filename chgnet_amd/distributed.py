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


def all_gather_energies(local_e: np.ndarray, shards: list[list[int]], n_total: int, device=None) -> np.ndarray:
    """All-gather the per-structure energies of every rank into original order (padded to equal
    counts; one collective of ``4 * max_shard`` bytes per rank)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    width = max(len(s) for s in shards)
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


def predict_sharded(predict_fn: Callable, graphs: Sequence, *, task: str = "efs", gather_energies: bool = True, **kwargs):
    """Each rank predicts its own shard with ``predict_fn(list_of_graphs, task=..., **kwargs)``:
    ``CHGNet.predict_graph`` for CrystalGraphs, or ``CHGNet.predict_structure`` for structures -- the
    graph of every structure is then built on the owning rank's GPU (chg_batch_build), nothing is
    converted on the host.

    Returns ``(local, energies)``: ``local`` maps original structure index -> prediction dict for the
    structures this rank owns; ``energies`` is the all-gathered float32 table of every structure (or
    ``None`` when ``gather_energies`` is false / no process group exists)."""
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
            energies = all_gather_energies(local_e, shards, len(graphs))
        else:
            energies = np.empty(len(graphs), np.float32)
            energies[mine] = local_e
    return local, energies
