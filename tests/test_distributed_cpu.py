"""N > 1 path on CPU: 2 ranks over gloo, the CPU oracle standing in for the per-rank GPU engine."""

from __future__ import annotations

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chgnet_amd.distributed import shard_indices, structure_cost
from conftest import GOLDEN, REPO, load_case

NAMES = ["limno2", "noangle", "s16tri", "s40", "li9co7o16"]


def test_shard_indices_balance_and_cover():
    rng = np.random.default_rng(0)
    costs = rng.uniform(1, 50, size=1000)
    for world in (1, 2, 4, 8):
        shards = shard_indices(costs, world)
        assert sorted(i for s in shards for i in s) == list(range(1000))
        loads = np.array([costs[s].sum() for s in shards])
        assert loads.max() / loads.mean() < 1.01
    assert shard_indices([], 2) == [[], []]
    assert shard_indices([3.0], 4) == [[0], [], [], []]


def _worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chgnet_amd.distributed import predict_sharded
    from oracle.chgnet_oracle import OracleCHGNet

    weights = dict(np.load(os.path.join(GOLDEN, "weights_seed0.npz")))
    oracle = OracleCHGNet(weights)
    graphs = [load_case(n)[0] for n in NAMES]
    local, energies = predict_sharded(lambda gs, task: oracle.predict_graph(list(gs), task, batch_size=16), graphs, task="ef")
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), energies=energies, owned=np.array(sorted(local)),
             **{f"f{i}": p["f"] for i, p in local.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_allgather(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # both ranks hold the full, identical energy table in original order
    assert np.array_equal(r0["energies"], r1["energies"])
    want = np.array([float(load_case(n)[1]["out_e"]) for n in NAMES], dtype=np.float32)
    assert np.abs(r0["energies"] - want).max() < 3e-6
    # shards are disjoint, cover everything, and are balanced by the cost proxy
    owned0, owned1 = set(r0["owned"].tolist()), set(r1["owned"].tolist())
    assert owned0 | owned1 == set(range(len(NAMES))) and not (owned0 & owned1)
    costs = np.array([structure_cost(load_case(n)[0]) for n in NAMES])
    loads = [costs[sorted(o)].sum() for o in (owned0, owned1)]
    assert max(loads) <= sum(loads) * 0.75
    # forces stay rank-local and match the reference goldens
    for r in (r0, r1):
        for i in r["owned"]:
            assert np.abs(r[f"f{i}"] - load_case(NAMES[i])[1]["out_f"]).max() < 3e-6


def test_bench_spawns_the_ranks_it_is_asked_for():
    """`python bench.py --gpus 2` outside a launcher starts 2 ranks itself (torch.distributed.run), builds the
    process group and all-gathers across them; --dry-run swaps RCCL for gloo and skips the GPU work."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout                       # stdout carries ONE line, the result of rank 0: library chatter
    line = json.loads(lines[0])                              # (gloo / RCCL banners, model banners) goes to stderr
    assert line["n_gpus"] == 2 and line["ranks_in_all_gather"] == [0, 1] and line["backend"] == "gloo"
    # the dry run also WALKS the two sharded legs of run_configs with a stand-in model (bench.py DryModel): the C3 sweep (LPT shards,
    # zero-padded all-gather -- every rank checks every slot of the energy table and exits non-zero otherwise) and the C5 epoch
    # (TrainStep, gradient all-reduce, Adam; the profiled steps after the epoch are taken by EVERY rank -- rank 0 alone used to enter
    # that all-reduce, which would have blocked a real multi-GPU run for ever)
    assert line["legs"]["C3_sweep"]["energies_gathered"] == 48 and line["legs"]["C3_sweep"]["shard_atoms_max_over_mean"] < 1.05
    assert line["legs"]["C5_train_epoch"]["ms_per_step"] > 0
    assert line["allreduced_gradient_values"][:3] == [2.0, 3.5, 5.0]        # mean over the ranks of (rank + 1) * call, rank 0 one call ahead
    # ... and every rank pinned itself to its own share of the cores (rank 0: the first half of this box's CPUs)
    aff = line["cpu_affinity"]
    assert aff is not None and aff["cpus"] >= 1 and aff["first_cpu"] == 0
    ncpu = len(os.sched_getaffinity(0))
    assert aff["cpus"] <= max(1, ncpu // 2) or aff["numa_nodes"] > 1
    # under a launcher --gpus must agree with the world size
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run"], env=env2, capture_output=True,
                         text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE is 1" in bad.stderr


def test_sharding_accepts_structures_without_graphs():
    from bench import sweep_structure

    structs = [sweep_structure(i) for i in range(12)]
    costs = [structure_cost(s) for s in structs]
    assert costs == [float(len(s)) for s in structs]
    shards = shard_indices(costs, 2)
    loads = [sum(costs[i] for i in sh) for sh in shards]
    assert abs(loads[0] - loads[1]) <= max(costs)


# ---- RcclComm rendezvous (chgnet_amd/distributed.py:exchange_unique_id): real processes, no GPU --------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rdv_worker(rank: int, world: int, port: int, delay: float, token: str, q):
    import time

    sys.path.insert(0, REPO)
    os.environ["CHGNET_RCCL_TOKEN"] = token
    from chgnet_amd.distributed import exchange_unique_id

    time.sleep(delay)
    ident = bytes(range(128)) if rank == 0 else None
    try:
        got = exchange_unique_id(rank, world, ident, addr="127.0.0.1", port=port, timeout_s=20.0)
        q.put((rank, got == bytes(range(128))))
    except Exception as exc:  # noqa: BLE001
        q.put((rank, repr(exc)))


@pytest.mark.parametrize("delays", [(0.0, 1.5, 0.0), (1.5, 0.0, 0.2)], ids=["peers_late", "rank0_late"])
def test_rccl_rendezvous_any_arrival_order(delays):
    """Three ranks, staggered by more than a second either way: every rank ends up with rank 0's 128 bytes.  (The file
    rendezvous of round 2 rejected ids older than 30 s and accepted ids a crashed run left behind.)"""
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_rdv_worker, args=(r, 3, port, delays[r], "job-a", q)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert res == {0: True, 1: True, 2: True}, res


def test_rccl_rendezvous_ignores_other_jobs_and_times_out():
    """A peer of ANOTHER launch (different token) on the same port is refused and does not count as a rank; without rank 0 a
    peer gives up with TimeoutError instead of hanging."""
    from chgnet_amd.distributed import exchange_unique_id

    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    stranger = ctx.Process(target=_rdv_worker, args=(1, 2, port, 0.0, "job-b", q))      # wrong token: never served
    mine = [ctx.Process(target=_rdv_worker, args=(r, 2, port, 0.3 * r, "job-a", q)) for r in range(2)]
    stranger.start()
    for p in mine:
        p.start()
    first = dict(q.get(timeout=60) for _ in range(2))
    assert first == {0: True, 1: True}, first
    stranger.join(timeout=40)                                                           # runs into its own timeout
    late = q.get(timeout=40)
    assert late[0] == 1 and "TimeoutError" in str(late[1]), late
    for p in mine:
        p.join(timeout=30)
    with pytest.raises(TimeoutError):
        exchange_unique_id(1, 2, None, addr="127.0.0.1", port=_free_port(), timeout_s=0.5)


def test_rank_cpu_sets_are_disjoint_and_cover_one_node_each():
    """``bench.pin_rank_cpus``: the ranks of a node take disjoint, contiguous shares of the allowed CPUs (in a subprocess per rank:
    the call changes the caller's affinity)."""
    import json
    import subprocess

    code = ("import json, os, sys; sys.path.insert(0, %r); import bench; r = bench.pin_rank_cpus(int(sys.argv[1]), 4); "
            "print(json.dumps({'info': r, 'cpus': sorted(os.sched_getaffinity(0)), 'omp': os.environ.get('OMP_NUM_THREADS')}))" % REPO)
    env = {k: v for k, v in os.environ.items() if k != "CHGNET_BENCH_NO_PIN"}
    got = [json.loads(subprocess.run([sys.executable, "-c", code, str(r)], env=env, capture_output=True, text=True, check=True).stdout)
           for r in range(4)]
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        pytest.skip("fewer than four CPUs")
    sets = [set(g["cpus"]) for g in got]
    assert all(s and s <= set(allowed) for s in sets)
    assert all(sets[i].isdisjoint(sets[j]) for i in range(4) for j in range(i + 1, 4))
    assert all(g["omp"] == str(min(len(g["cpus"]), 16)) for g in got)
    off = json.loads(subprocess.run([sys.executable, "-c", code, "0"], env=dict(env, CHGNET_BENCH_NO_PIN="1"), capture_output=True, text=True,
                                    check=True).stdout)
    assert off["info"] is None and off["cpus"] == allowed
