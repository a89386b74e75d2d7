"""bench.py -- structures/s (energy + force + stress) of the HIP engine on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

``--gpus N`` with N > 1 outside a launcher (no ``$LOCAL_RANK``) re-executes this file under
``torch.distributed.run`` with N ranks (one per GPU, RCCL over xGMI); under a launcher the ranks are
taken from RANK / LOCAL_RANK / WORLD_SIZE and ``--gpus`` must agree with WORLD_SIZE.

Headline workload (BASELINE.json configs[1], SURVEY 8d "C2"): a batch of 1024 LiMnO2 5x1x1 supercells
(40 atoms, ~3.45k directed bonds, ~4.0k angles each), fractional coordinates perturbed with
N(0, 0.01^2), numpy default_rng(seed = global structure index); random-init weights of the 0.3.0
architecture (tests/golden/weights_seed0.npz -- the pretrained blobs are not available offline).
One step = one pass of the hot path (chg_predict, task "efs") over the rank's 1024 device-resident
structures plus the download of E/F/S to the host.  Ranks hold disjoint structures (weak scaling,
no data-path collective); with N > 1 the per-structure energies are all-gathered over RCCL each step.

Prints ONE JSON line (rank 0):
  roofline       the kernel with the largest share of the step (fp32-MFMA-bound), HIP events on the engine stream
  roofline_hbm   the HBM-bound kernels of the path against a STREAM-like copy measured in this process
  configs        the other BASELINE.json configs: C1 (single LiMnO2 cell + x1024), C3 (ragged 10-100-atom sweep
                 through predict_structure, sharded over the ranks), C4 (NVT MD of 2x2x2 Li9Co7O16, steps/s),
                 C5 (one data-parallel fine-tuning epoch, energy + magmom terms, gradient all-reduce)
  cpu_baseline   the UNMODIFIED reference CHGNet.predict_graph timed on this box's host cores (kind "reference": bytecode archive
                 oracle/_ref/chgnet_ref_bytecode.zip, oracle/build_ref_model.py), the CPU oracle (oracle/chgnet_oracle.py, a torch
                 port of the reference path) next to it as ``port``; the same leg checks the configs' results against the oracle
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0           # same guide: HBM3E 8 TB/s (spec)
GUIDE_COPY_GBS = 6290.0         # same guide: float4 copy measured on MI355X (79 % of spec): floor of the "measured HBM roofline"
PEAK_F16_MFMA_TFLOPS = 2500.0   # same guide: dense f16 / bf16 MFMA
SPLIT_MFMA_PER_PRODUCT = 3      # csrc/mfma_split.h: x.w = xh.wh + xl.wh + xh.wl


def csrc_hash() -> str:
    """Hash of the sources of the prediction kernels: profiles/pmc_latest.json and sq_latest.json carry the hash of the sources their
    counters were taken on (profiles/summarize.py); counters of OTHER sources are not quoted in the line (they went stale silently
    before).  Covered: every kernel header and the unit that instantiates and launches the prediction kernels (engine_predict.hip) --
    not the ABI / training / graph-build units, whose edits do not change these kernels."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(REPO, "chgnet_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith(".h") or name == "engine_predict.hip":
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]
# Which matrix form a tile kernel's contractions run in (csrc/mfma_split.h): every f32 product as three f16 MFMAs on hi/lo
# operand halves with f32 accumulation (22+ significand bits; measured parity = the f32 form's), or the f32 MFMA itself.
# fractions in `tile_kernels` stay relative to the f32 MFMA peak so that they compare with earlier rounds; a split kernel's own
# matrix-pipe ceiling is PEAK_F16_MFMA_TFLOPS / 3 f32-equivalent TFLOP/s.
SPLIT_KERNELS = {"atomconv_fwd", "atomconv_bwd", "bondconv_fwd", "bondconv_bwd", "angleupd_fwd", "angleupd_bwd"}

# Algorithmic work per unit of the dominant kernels in the engine's (factorised) formulation, derived in
# DESIGN.md "Roofline accounting": (unit, MFMA flop per unit, compulsory HBM bytes per unit)
KERNEL_MODEL = {
    # two 64x64 second-layer blocks per direction + the bond block W_bond (64 -> 128) once per BOND (the kernel contracts it per
    # direction; the algorithmic count is per bond); h_bond 128 + w_ag 128 read, Q table 256 written for the reverse sweep, per direction
    "atomconv_fwd": ("n_directed", 24576, 538),
    # adjoint (round 4: with the dE/d h_bond update in its tiles): two second-layer blocks and their transposes + W_bond^T
    # (128 -> 64) per direction; the dE/dQ table (256 B per direction) is no longer written, the bond's dE/d h_bond row is read and
    # written instead (2 x 128 B per direction)
    "atomconv_bwd": ("n_directed", 32768 + 16384, 900 - 256 + 256),
    # round 6 (csrc AngleArgs::zsave, large batches): the forward leaves z = W_ang x + table rows behind (512 B per angle written), the
    # adjoint reads it back (512 B) instead of the angle row (256 B) and contracting W_ang (64 -> 128: 16,384 flop) again;
    # CHGNET_ZSAVE=0 restores the round-5 formulation (300 B | 65,536 flop, 800 B)
    "bondconv_fwd": ("n_angles", 32768, 300 + (512 if os.environ.get("CHGNET_ZSAVE", "1") != "0" else 0)),
    "bondconv_bwd": ("n_angles", 65536 - 16384, 800 - 256 + 512) if os.environ.get("CHGNET_ZSAVE", "1") != "0" else ("n_angles", 65536, 800),
    "angleupd_fwd": ("n_angles", 16384, 524),
    "angleupd_bwd": ("n_angles", 32768, 780),
}

# Reverse kernels of the fused second-order training sweep (csrc/kernels_train2_tile.h): (unit, flop per unit, compulsory bytes per
# unit, what).  Flops: every contraction once for the primal / bar row and once for the tangent / G row -- BondConv: W_ang 64->128
# (16,384) x 2, W2c + W2g (16,384) x 2, their transposes x 2 -> 131,072; AngleUpdate: the angle block and its transpose -> 65,536;
# AtomConv: second layer and its transpose, two rows each -> 65,536.  Bytes: the row dumps the weight-gradient contractions read back
# (BondConv 6 x 512, AngleUpdate 2 x 512, AtomConv 4 x 512) + angle rows in (2 x 256) and adjoint rows updated (2 x 2 x 256).
TRAIN_KERNEL_MODEL = {
    "t2_bond_b": ("n_angles", 131072, 6 * 512 + 512 + 1024, "k2_angle<BondConv, reverse>"),
    "t2_angle_b": ("n_angles", 65536, 2 * 512 + 512 + 1024 + 512, "k2_angle<AngleUpdate, reverse>"),
    "t2_atom_b": ("n_directed", 65536, 4 * 512 + 256, "k2_atom<reverse>"),
}

# HBM-bound kernels: (unit, compulsory bytes per unit) -- every distinct input read once, every output written once
# (DESIGN.md "Roofline accounting", HBM regime).  Eb/Eu-dependent terms are added in hbm_model().
LIMNO2_FRAC = [[0.5, 0.5, 0.3797505], [0, 0, 0.6202495], [0.5, 0.5, 0.8632525], [0, 0, 0.1367475],
               [0.5, 0, 0.3608245], [0, 0.5, 0.0985135], [0.5, 0, 0.9014865], [0, 0.5, 0.6391755]]
LIMNO2_SPECIES = ["Li", "Li", "Mn", "Mn", "O", "O", "O", "O"]


def limno2(supercell=(1, 1, 1)):
    """mp-18767 LiMnO2 (the reference's own fixture, examples/mp-18767-LiMnO2.cif), written out so the
    bench does not need /root/reference at run time."""
    from chgnet_amd import Structure
    from chgnet_amd.graph.structure import Lattice

    base = Structure(Lattice.from_parameters(2.868779, 4.634475, 5.832507, 90, 90, 90), LIMNO2_SPECIES, LIMNO2_FRAC)
    return base if tuple(supercell) == (1, 1, 1) else base.make_supercell(list(supercell))


def workload_structures(n_struct: int, first_seed: int, supercell=(5, 1, 1)):
    base = limno2(supercell)
    return [base.perturb(0.01, np.random.default_rng(first_seed + i)) for i in range(n_struct)]


def build_workload(n_struct: int, first_seed: int):
    from chgnet_amd import CrystalGraphConverter

    conv = CrystalGraphConverter(atom_graph_cutoff=6, bond_graph_cutoff=3)
    return [conv(s) for s in workload_structures(n_struct, first_seed)]


def sweep_atom_count(i: int) -> int:
    return int(np.random.default_rng([12345, i]).integers(10, 101))


def sweep_structure(i: int):
    """Structure i of the SURVEY 8d "C3" sweep: random orthorhombic cell of 10-100 atoms at LiMnO2's number
    density (0.103 atoms/A^3), species from {Li, Mn, Co, O}, sites on a jittered grid (min distance ~1.6 A)."""
    from chgnet_amd import Structure
    from chgnet_amd.graph.structure import Lattice

    rng = np.random.default_rng([12345, i])
    n = int(rng.integers(10, 101))
    vol = n / 0.103
    a = vol ** (1 / 3) * rng.uniform(0.85, 1.15)
    b = vol ** (1 / 3) * rng.uniform(0.85, 1.15)
    m = int(np.ceil(n ** (1 / 3)))
    grid = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    pick = rng.choice(len(grid), size=n, replace=False)
    frac = (grid[pick] + 0.5 + rng.uniform(-0.15, 0.15, (n, 3))) / m
    return Structure(Lattice(np.diag([a, b, vol / (a * b)])), rng.choice([3, 25, 27, 8], size=n), frac)


def li9co7o16_supercell(scale=(2, 2, 2)):
    """Supercell (2x2x2 = 256 atoms by default) of mp-1175469 Li9Co7O16 (reference fixture examples/mp-1175469-Li9Co7O16.cif;
    the cell is stored with the golden cases)."""
    from chgnet_amd import Structure
    from chgnet_amd.graph.structure import Lattice

    d = np.load(os.path.join(REPO, "tests", "golden", "case_li9co7o16.npz"))
    return Structure(Lattice(d["lattice_f64"]), d["atomic_number"], d["frac_coord_f64"]).make_supercell(list(scale))


# ---------------------------------------------------------------------------------------------------------
# multi-GPU plumbing
# ---------------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


_RESULT_FD = None


def claim_stdout() -> None:
    """stdout carries ONE line, the result.  Libraries write there too -- RCCL prints a five-line banner through C stdio,
    which a pipe holds back until the process exits, i.e. AFTER the result -- so file descriptor 1 is pointed at stderr for
    the whole run and the result goes to a private duplicate of the original stdout (``emit``)."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line: dict) -> None:
    data = (json.dumps(line) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
        return
    sys.stdout.flush()
    while data:
        data = data[os.write(_RESULT_FD, data):]


def spawn_ranks(n: int, argv: list[str]) -> int:
    """Re-execute this file under torch.distributed.run with n ranks on this node; returns its exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL over xGMI needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def _cpulist(text: str) -> list[int]:
    out = []
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            out += list(range(int(lo), int(hi or lo) + 1))
    return out


def pin_rank_cpus(local_rank: int, world: int) -> dict | None:
    """One process per GPU means N packers, graph builders and result splitters on ONE host: give every rank a DISJOINT set of
    cores, on one NUMA node, instead of letting 8 x (torch + OpenMP + loader thread) float over both sockets (VERDICT r04 weak 7 /
    item 8).  Ranks are dealt to the NUMA nodes in order (GPUs 0..N/2-1 hang off socket 0 on the usual two-socket MI355X node) and
    split a node's CPUs evenly.  ``CHGNET_BENCH_NO_PIN=1`` leaves the affinity alone.  Returns what was done (goes into the line)."""
    if world <= 1 or os.environ.get("CHGNET_BENCH_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    import glob

    allowed = sorted(os.sched_getaffinity(0))
    nodes = []
    for path in sorted(glob.glob("/sys/devices/system/node/node[0-9]*/cpulist"), key=lambda q: int(q.split("node")[-1].split("/")[0])):
        try:
            with open(path) as fh:
                cpus = [c for c in _cpulist(fh.read()) if c in set(allowed)]
        except OSError:
            cpus = []
        if cpus:
            nodes.append(cpus)
    if not nodes:
        nodes = [allowed]
    node = local_rank * len(nodes) // world
    peers = [r for r in range(world) if r * len(nodes) // world == node]          # ranks that share this node
    cpus = nodes[node]
    per = max(1, len(cpus) // len(peers))
    i = peers.index(local_rank)
    mine = cpus[i * per:(i + 1) * per] if i * per < len(cpus) else cpus[-per:]
    os.sched_setaffinity(0, mine)
    threads = str(max(1, min(len(mine), 16)))
    os.environ["OMP_NUM_THREADS"] = threads                                          # before torch / numpy spin up their pools
    return {"numa_nodes": len(nodes), "node": node, "cpus": len(mine), "first_cpu": mine[0], "last_cpu": mine[-1], "threads": int(threads)}


class Ranks:
    """The process group of this run (None-safe helpers for the single-process case)."""

    def __init__(self, args) -> None:
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # (ranks of THIS node only: on a multi-node job the global world size would put every rank on NUMA node 0)
        self.cpu_affinity = pin_rank_cpus(self.local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", self.world)))
        self.dist = None
        self.torch = None
        self.shared_device = bool(getattr(args, "shared_device", False))
        self.device_index = 0 if self.shared_device else self.local_rank   # --shared-device: every rank's engine sits on GPU 0
        self.backend = "gloo" if (args.dry_run or self.shared_device) else "nccl"   # "nccl" IS RCCL on ROCm
        self.comm = None
        self.comm_note = None
        self.rccl_info = None
        if args.comm in ("rccl", "auto") and not args.dry_run and not self.shared_device and (self.world > 1 or os.environ.get("CHGNET_BENCH_FORCE_DIST")):
            from chgnet_amd.distributed import RcclComm      # RCCL through the engine library's C-ABI: no torch.distributed

            try:
                self.comm = RcclComm(self.rank, self.world, self.local_rank)
            except Exception as exc:  # noqa: BLE001
                if args.comm == "rccl":     # asked for by name: fail, do not measure something else
                    raise SystemExit(f"bench.py --comm rccl: the engine library's communicator could not be created "
                                     f"({type(exc).__name__}: {exc})") from exc
                self.comm = None            # --comm auto: a scaling run must not die on the rendezvous -- torch.distributed instead, and said so
                self.comm_note = f"RcclComm failed ({type(exc).__name__}: {exc}); torch.distributed used instead"
        if self.comm is not None:
            self.backend = "rccl (chg_comm_*)"
            info = self.comm.info()
            if info["nranks"] != args.gpus:
                raise SystemExit(f"bench.py: --gpus {args.gpus} but RCCL reports {info['nranks']} ranks")
            devices = self.comm.all_gather(np.array([info["device"]], np.float32)).astype(int).tolist()
            if len(set(devices)) != self.world and not os.environ.get("CHGNET_BENCH_FORCE_DIST"):
                raise SystemExit(f"bench.py: ranks share a GPU (hipGetDevice per rank: {devices})")
            self.rccl_info = {"nranks_from_ncclCommCount": info["nranks"], "device_per_rank": devices}
        elif self.world > 1 or os.environ.get("CHGNET_BENCH_FORCE_DIST"):   # the env switch exercises the RCCL leg on one GPU
            import torch
            import torch.distributed as dist

            self.torch, self.dist = torch, dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            kw = {}
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                kw["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world, **kw)
            if dist.get_world_size() != args.gpus:
                raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        elif args.gpus != 1:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {self.world}")

    @property
    def device(self):
        return self.torch.device("cuda", self.local_rank) if self.backend == "nccl" else self.torch.device("cpu")

    def all_gather(self, values: np.ndarray) -> np.ndarray:
        """Equal-length float32 vectors of every rank, concatenated in rank order (RCCL all-gather)."""
        if self.comm is not None:
            return self.comm.all_gather(values)
        if self.dist is None:
            return values
        mine = self.torch.from_numpy(np.ascontiguousarray(values, np.float32)).to(self.device)
        allv = self.torch.empty(self.world * mine.numel(), dtype=mine.dtype, device=mine.device)
        self.dist.all_gather_into_tensor(allv, mine)
        return allv.cpu().numpy()

    def barrier(self) -> None:
        if self.comm is not None:
            self.comm.barrier()
        if self.dist is not None:
            self.dist.barrier()
            if self.backend == "nccl":
                self.torch.cuda.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if self.comm is not None:
            return float(self.comm.all_gather(np.array([x], np.float32)).max())
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self) -> None:
        if self.comm is not None:
            self.comm.barrier()
            self.comm.close()
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------
# CPU leg: baseline timing + parity of the configs (the only place that touches oracle/)
# ---------------------------------------------------------------------------------------------------------
def cpu_leg(weights: dict, graphs, checks: dict, seconds_budget: float = 24.0) -> tuple[dict, dict]:
    """Time the CPU oracle on a bounded sample of the headline workload and use it as the checker for the
    small per-config samples collected in ``checks`` (name -> (graphs, engine results)).

    The GPU box exposes 256 logical CPUs; torch's intra-op pool stops scaling (and then collapses) well before
    that on these small graph ops, so a few thread counts are tried and the best one is reported."""
    import torch

    from oracle.chgnet_oracle import OracleCHGNet

    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    model = OracleCHGNet(weights)
    thread_counts = sorted({min(ncpu, 8), min(ncpu, 16), min(ncpu, 32)})

    def time_model(predict, sample, budget):
        """Best (rate, batch_size, structures, threads) of ``predict(chunk, bs)`` over a few thread counts and batch sizes."""
        trials = [(t, bs) for t in thread_counts for bs in (1, 16)]
        top = None
        for threads, bs in trials:
            torch.set_num_threads(threads)
            predict(sample[:1], 1)  # warm-up
            n_done, t0 = 0, time.perf_counter()
            while n_done < len(sample) and time.perf_counter() - t0 < budget / len(trials):
                chunk = sample[n_done:n_done + bs]
                predict(chunk, bs)
                n_done += len(chunk)
            rate = n_done / (time.perf_counter() - t0)
            if top is None or rate > top[0]:
                top = (rate, bs, n_done, threads)
        return top

    # (1) the REFERENCE ITSELF on this box's host cores (north star: "next to the reference CPU path timed on the same box's host
    # cores"): the unmodified CHGNet.predict_graph (model.py:593-665) from the bytecode archive oracle/build_ref_model.py compiled from
    # /root/reference where it lies (oracle/_ref/, a built artefact that travels like the .so files); same graphs, same weights
    reference, ref_note = None, None
    try:
        reference = reference_leg(weights, graphs, time_model, seconds_budget * 0.6, ncpu)
    except Exception as exc:  # noqa: BLE001 -- the archive is absent or of another interpreter: the port below is the baseline then
        ref_note = f"{type(exc).__name__}: {exc}"[:300]
    # (2) the port (oracle/chgnet_oracle.py): the checker of the parity flags below, timed as well
    best = time_model(lambda chunk, bs: model.predict_graph(chunk, "efs", batch_size=bs), graphs, seconds_budget * (0.4 if reference else 1.0))
    port = {
        "value": round(best[0], 3), "unit": "structures/s", "cores": best[3], "kind": "port",
        "sample": f"{best[2]} structures of the headline workload through oracle/chgnet_oracle.py (torch fp32 CPU restatement of the "
                  f"reference path, autograd F/S like the reference; NOT the reference's CHGNet.predict_graph: one batched forward "
                  f"without the per-graph BatchedGraph.from_graphs loop and without the dead third AngleUpdate, so it flatters "
                  f"the CPU), batch_size={best[1]}, best of {'/'.join(map(str, thread_counts))} torch threads on {ncpu} logical CPUs"}
    if reference is not None:
        baseline = dict(reference)
        baseline["port"] = port
    else:
        baseline = dict(port)
        baseline["reference_unavailable"] = ref_note
    torch.set_num_threads(best[3])
    # SURVEY 8d "informative second baseline": the same torch restatement through torch-ROCm EAGER on this GPU -- the
    # "recompile PyTorch for ROCm" route the engine replaces (host batching included, like the CPU figure)
    # (own process: torch-ROCm brings its own HIP runtime, which finds no device once the engine library's runtime owns it)
    try:
        import re
        import subprocess

        probe = os.path.join(REPO, "tools", "gpu_torch_baseline_probe.py")
        out = subprocess.run([sys.executable, probe, "64", "64"], capture_output=True, text=True, timeout=300).stdout
        m = re.search(r"= ([0-9.]+) structures/s", out)
        baseline["gpu_eager_baseline"] = {"value": float(m.group(1)) if m else None, "unit": "structures/s",
                                          "what": "oracle/chgnet_oracle.py on cuda:0 through torch-ROCm eager, batch_size=64, 64 structures "
                                                  "(tools/gpu_torch_baseline_probe.py in its own process)"}
    except Exception as exc:  # noqa: BLE001 -- informative only
        baseline["gpu_eager_baseline"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"[:200]}
    parity = {}
    for name, item in checks.items():
        if item[0] == "weight-gradients":           # C5: d CombinedLoss / d parameters against torch double-backward (float64)
            parity[name] = gradient_parity(weights, *item[1:])
            continue
        gs, got = item
        err = {"e": 0.0, "f": 0.0, "s": 0.0}
        for g, r in zip(gs, got):
            ref = model.predict_graph(g, "efs")
            for k in err:
                err[k] = max(err[k], float(np.abs(np.asarray(r[k], np.float64) - np.asarray(ref[k], np.float64)).max()))
        # north-star bars: E 1e-4 eV/atom, F 1e-3 eV/A; stress 1e-2 GPa
        parity[name] = {"n_checked": len(gs), "max_abs_err": {k: float(f"{v:.3g}") for k, v in err.items()},
                        "ok": bool(err["e"] < 1e-4 and err["f"] < 1e-3 and err["s"] < 1e-2)}
    return baseline, parity


def reference_leg(weights: dict, graphs, time_model, budget: float, ncpu: int) -> dict:
    """The unmodified reference on this box's CPU: ``CHGNet.predict_graph(graphs, task="efs", batch_size=b)`` (model.py:593-665) on
    the reference's own ``CrystalGraph`` objects, timed like the port; its outputs double as a check of the port on this box."""
    import torch

    from oracle.build_ref_model import load
    from oracle.chgnet_oracle import OracleCHGNet

    load()
    import io
    from contextlib import redirect_stdout

    from chgnet.graph.crystalgraph import CrystalGraph as RefGraph
    from chgnet.model.model import CHGNet as RefCHGNet

    with redirect_stdout(io.StringIO()):     # "CHGNet initialized with ..." must not land in front of the bench line
        ref = RefCHGNet()
    ref.load_state_dict({k: torch.tensor(np.asarray(v)) for k, v in weights.items()})
    ref.eval()
    i32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32)  # noqa: E731
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
    rgraphs = [RefGraph(atomic_number=i32(g.atomic_number), atom_frac_coord=f32(g.atom_frac_coord), atom_graph=i32(g.atom_graph),
                        neighbor_image=f32(g.neighbor_image), directed2undirected=i32(g.directed2undirected),
                        undirected2directed=i32(g.undirected2directed), bond_graph=i32(np.asarray(g.bond_graph).reshape(-1, 5)),
                        lattice=f32(g.lattice), atom_graph_cutoff=6, bond_graph_cutoff=3) for g in graphs]
    rate, bs, n_done, threads = time_model(lambda chunk, b: ref.predict_graph(chunk, task="efs", batch_size=b), rgraphs, budget)
    torch.set_num_threads(threads)
    out, port = ref.predict_graph(rgraphs[0], task="efs"), OracleCHGNet(weights).predict_graph(graphs[0], "efs")
    agree = {k: float(f"{float(np.abs(np.asarray(out[k], np.float64) - np.asarray(port[k], np.float64)).max()):.3g}") for k in ("e", "f", "s")}
    return {"value": round(rate, 3), "unit": "structures/s", "cores": threads, "kind": "reference",
            "sample": f"{n_done} structures of the headline workload through the UNMODIFIED reference CHGNet.predict_graph(task='efs', "
                      f"batch_size={bs}) on this box's host CPU (fp32, torch {torch.__version__}; bytecode of /root/reference/chgnet compiled by "
                      f"oracle/build_ref_model.py), best of {'/'.join(str(min(ncpu, t)) for t in (8, 16, 32))} torch threads x batch_size 1/16 on {ncpu} logical CPUs",
            "port_vs_reference_max_abs": agree}


def gradient_parity(weights: dict, graphs, targets: dict, got: dict) -> dict:
    """Parameter gradients of CombinedLoss(MSE, target efsm) for a few structures: the device's against autograd /
    double-backward through the float64 oracle (the loss written out on tensors: trainer.py:779-869 with the default
    ratios 1 / 1 / 0.1 / 0.1 and mean reduction)."""
    import torch

    from oracle.chgnet_oracle import OracleCHGNet

    t = lambda a: torch.tensor(np.asarray(a, np.float64))  # noqa: E731
    te, tf, ts, tm = t(targets["e"]), t(np.concatenate(targets["f"])), t(np.stack(targets["s"])), t(np.concatenate(targets["m"]))

    def loss(o):
        mse = lambda a, b: ((a.reshape(-1) - b.reshape(-1)) ** 2).mean()  # noqa: E731
        return mse(o["e"], te) + mse(o["f"], tf) + 0.1 * mse(o["s"], ts) + 0.1 * mse(o["m"], tm)

    want = OracleCHGNet(weights, dtype=torch.float64).parameter_gradients(graphs, loss, task="efsm")
    worst, worst_name, n = 0.0, "", 0
    for k, ref in want.items():
        if k.startswith(("angle_layers.2.", "composition_model")):      # dead layer / frozen AtomRef: zero gradient
            continue
        scale = float(np.abs(ref).max())
        if scale == 0.0:
            continue
        rel = float(np.abs(np.asarray(got[k], np.float64) - ref).max()) / scale
        n += 1
        if rel > worst:
            worst, worst_name = rel, k
    return {"n_checked": len(graphs), "tensors": n, "max_rel_err": float(f"{worst:.3g}"), "worst_tensor": worst_name,
            "ok": bool(np.isfinite(worst) and worst < 1e-3)}


# ---------------------------------------------------------------------------------------------------------
def hbm_model(packed) -> dict:
    """Compulsory HBM bytes per launch of the HBM-bound kernels (all fp32 / int32)."""
    Ed, Eu, A, Eb, N = packed.n_directed, packed.n_undirected, packed.n_angles, packed.n_bnodes, packed.n_atoms
    return {
        "gemm_Q": Eu * (256 + 512),                 # read h_bond row (64), write Q row (128)
        "gemm_GQ": Eu * (512 + 256 + 256),          # read GQ row (128), read-modify-write Gb row (64)
        "bond_embed_fwd": Eu * (16 + 8 + 512) + Eb * 256,   # ev + 2 indices in; hb0, wag (and wbgc for bond-graph nodes) out
        "angle_embed_fwd": A * (8 + 256) + Ed * 16,         # 2 indices in, unit vectors once, angle row out
        "edge_force": Ed * (16 + 16 + 16 + 16 + 4 * 6) + N * 12,   # ev, eu, Gu, Gu[e_rev], d2u / rev / owner / centre / u2d[k] / Grk[k], forces out
    }


def run_configs(eng, weights, ranks: Ranks, args, model=None, legs=("C1", "C3", "C4", "C5")) -> tuple[dict, dict]:
    """BASELINE.json configs other than the headline; returns (configs, checks for the CPU leg).  ``model`` / ``legs``: the dry run
    (no GPU) walks the sharded legs C3 and C5 with a stand-in model (``DryModel``) so that their multi-rank bookkeeping -- LPT shards,
    padded all-gather, max-over-ranks timing, gradient all-reduce -- is exercised by the world-size-2 CPU test before a real node is."""
    from chgnet_amd import CrystalGraphConverter
    from chgnet_amd.calculator import CHGNetCalculator
    from chgnet_amd.md import BerendsenNVT
    from chgnet_amd.model import CHGNet

    conv = CrystalGraphConverter(atom_graph_cutoff=6, bond_graph_cutoff=3)
    configs, checks = {}, {}
    if model is None:
        model = CHGNet(state_dict=weights, use_device=ranks.local_rank)
        model._engine = eng   # one engine per GPU: share the bench's
        model.graph_converter.set_isolated_atom_response("ignore")

    def timed(fn, reps):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return min(ts), out

    if ranks.rank == 0 and "C1" in legs:
        # ---- C1: the reference's plumbing case, one pristine 8-atom LiMnO2 cell -------------------------------------
        s8 = limno2()
        dt, pred = timed(lambda: model.predict_structure(s8, task="efsm"), 20)
        checks["C1_single"] = ([conv(s8)], [pred])
        g8 = conv(s8)
        configs["C1_single"] = {
            "workload": f"1 x mp-18767 LiMnO2 (8 atoms, {len(g8.atom_graph)} directed bonds, {len(g8.bond_graph)} angles), "
                        "CHGNet.predict_structure task efsm, host structure in -> host dict out (graph built on the device)",
            "ms_per_call": round(1e3 * dt, 3), "structures_per_s": round(1 / dt, 1)}
        structs = workload_structures(args.structures, 0, supercell=(1, 1, 1))
        b8 = eng.build_batch(structs)

        def step8():
            eng.predict(b8, "efs")
            return eng.download(b8, "efs")

        dt, _ = timed(step8, 10)
        configs["C1_x1024"] = {
            "workload": f"{args.structures} x perturbed LiMnO2 cell (8 atoms; {b8.packed.n_directed} directed bonds, "
                        f"{b8.packed.n_angles} angles in the batch), task efs, device-resident",
            "ms_per_step": round(1e3 * dt, 3), "structures_per_s": round(args.structures / dt, 1)}
        b8.free()

    # ---- C3: ragged sweep, sharded over the ranks by atom count (LPT), energies all-gathered ---------------------
    from chgnet_amd.distributed import shard_indices

    if "C3" in legs:
        n_total = args.sweep_structures * ranks.world
        counts = [sweep_atom_count(i) for i in range(n_total)]
        shards = shard_indices([float(c) for c in counts], ranks.world)
        mine = shards[ranks.rank]
        structs = [sweep_structure(i) for i in mine]
        model.predict_structure(structs[:8], task="efs", batch_size=8)            # warm-up
        width = max(len(s) for s in shards)
        best, preds = None, None
        for _ in range(2):
            ranks.barrier()
            t0 = time.perf_counter()
            preds = model.predict_structure(structs, task="efs", batch_size=args.sweep_chunk)
            e_local = np.zeros(width, np.float32)
            e_local[:len(preds)] = [p["e"] for p in preds]
            table = ranks.all_gather(e_local)
            ranks.barrier()
            mine_s = time.perf_counter() - t0
            dt = ranks.max_over_ranks(mine_s)
            if best is None or dt < best:
                best, per_rank_s = dt, ranks.all_gather(np.array([mine_s], np.float32)).astype(np.float64)
            if hasattr(model, "expected_energy"):   # dry run: every rank holds every structure's energy in the slot its shard says
                for r_, sh in enumerate(shards):
                    want = np.array([model.expected_energy(len(sweep_structure(i))) for i in sh], np.float32)
                    got_ = np.asarray(table[r_ * width:r_ * width + len(sh)], np.float32)
                    if not np.array_equal(got_, want) or np.any(table[r_ * width + len(sh):(r_ + 1) * width]):
                        raise SystemExit(f"bench.py dry run: rank {ranks.rank} sees a wrong energy table for shard {r_}")
        shard_atoms = np.array([sum(counts[i] for i in sh) for sh in shards], np.float64)
        if ranks.rank == 0:
            sample = list(range(0, len(structs), max(1, len(structs) // 6)))[:6]
            checks["C3_sweep"] = ([conv(structs[i]) for i in sample], [preds[i] for i in sample])
            gs = [conv(s) for s in structs[:200]]
            per_atom = (sum(len(g.atom_graph) for g in gs) / sum(len(s) for s in structs[:200]),
                        sum(len(g.bond_graph) for g in gs) / sum(len(s) for s in structs[:200]))
            configs["C3_sweep"] = {
                "workload": f"{n_total} random orthorhombic cells of 10-100 atoms (density 0.103 atoms/A^3, species Li/Mn/Co/O, "
                            f"default_rng([12345, i])), {sum(counts)} atoms, ~{per_atom[0]:.0f} directed bonds and ~{per_atom[1]:.0f} angles "
                            f"per atom; host structures -> CHGNet.predict_structure(task efs, batch_size={args.sweep_chunk}) -> host dicts, "
                            f"graphs built on the device, LPT-sharded over {ranks.world} GPU(s), energies all-gathered",
                "seconds": round(best, 4), "structures_per_s": round(n_total / best, 1), "atoms_per_s": round(sum(counts) / best, 1),
                "energies_gathered": int(np.isfinite(table).sum()) if ranks.world > 1 else len(preds),
                # load balance of the LPT sharder (chgnet_amd/distributed.py): atoms per rank, and what the ranks actually took
                "shard_atoms_max_over_mean": round(float(shard_atoms.max() / shard_atoms.mean()), 4),
                "per_rank_seconds": {"min": round(float(per_rank_s.min()), 4), "median": round(float(np.median(per_rank_s)), 4),
                                     "max": round(float(per_rank_s.max()), 4)}}

    # ---- C4: NVT MD, graph rebuilt on the device every step (replicas only: rank 0) ------------------------------
    if ranks.rank == 0 and "C4" in legs:
        cell = li9co7o16_supercell()
        calc = CHGNetCalculator(model)
        md = BerendsenNVT(cell, calc, temperature_K=1000.0, timestep_fs=2.0, task="ef")
        md.run(10)
        out = md.run(args.md_steps)
        g = conv(md.structure)
        checks["C4_md"] = ([g], [model.predict_structure(md.structure, task="efs")])
        configs["C4_md"] = {
            "workload": f"NVT (Berendsen, 1000 K, 2 fs) MD of 2x2x2 Li9Co7O16 = {len(cell)} atoms ({len(g.atom_graph)} directed bonds, "
                        f"{len(g.bond_graph)} angles), {args.md_steps} steps through CHGNetCalculator.calculate(task ef), neighbour list "
                        "and graph rebuilt on the device every step (in-repo integrator: ASE is not installed)",
            "steps_per_s": round(out["steps_per_s"], 1), "ms_per_step": round(1e3 / out["steps_per_s"], 3),
            "calculator_ms_per_step": round(1e3 * out["calculator_s"] / args.md_steps, 3),
            "temperature_K": round(out["temperature_K"], 1)}
        # BASELINE.json says "~512 atoms": the 4x2x2 cell next to the 2x2x2 one (same path, rebuilt every step)
        cell512 = li9co7o16_supercell((4, 2, 2))
        md5 = BerendsenNVT(cell512, CHGNetCalculator(model), temperature_K=1000.0, timestep_fs=2.0, task="ef")
        md5.run(10)
        n512 = max(100, args.md_steps // 2)
        out5 = md5.run(n512)
        configs["C4_md"]["cell_4x2x2_512_atoms"] = {
            "atoms": len(cell512), "steps": n512, "steps_per_s": round(out5["steps_per_s"], 1),
            "ms_per_step": round(1e3 / out5["steps_per_s"], 3), "temperature_K": round(out5["temperature_K"], 1)}
        # the use case north_star names next to the single trajectory: an MD ENSEMBLE -- R replicas of the 256-atom cell (same topology,
        # own velocities) advanced in lockstep, all R graphs built and swept by ONE predict_structure call per step
        ens = {}
        for R in (8, 32):
            reps = [BerendsenNVT(li9co7o16_supercell(), None, temperature_K=1000.0, timestep_fs=2.0, seed=r, task="ef") for r in range(R)]
            single = BerendsenNVT(li9co7o16_supercell(), CHGNetCalculator(model), temperature_K=1000.0, timestep_fs=2.0, seed=0, task="ef")

            def ens_step(first=False):
                preds = model.predict_structure([m.structure for m in reps], task="ef", batch_size=R, min_atoms_per_batch=0)
                for m, pr in zip(reps, preds):
                    if not first:
                        m.vel += m._half_dt_over_m * np.asarray(pr["f"], np.float64)      # second half kick of the previous step
                        m.T_end = m.temperature()
                    m.forces, m.energy = np.asarray(pr["f"], np.float64), float(pr["e"]) * len(m.structure)
                    m.advance_positions()                                                # thermostat, first half kick, drift

            ens_step(first=True)
            for _ in range(5):
                ens_step()
            n_ens = max(50, args.md_steps // (R // 2))
            t0 = time.perf_counter()
            for _ in range(n_ens):
                ens_step()
            dt_ens = time.perf_counter() - t0
            # replica 0 against the single-trajectory driver (same seed, same integrator): same temperature path to fp32 reassociation
            single.run(5 + n_ens)                       # the replicas' last evaluation is the end of their step 5 + n_ens
            ens[f"R{R}"] = {"replicas": R, "steps": n_ens, "replica_steps_per_s": round(R * n_ens / dt_ens, 1),
                            "ms_per_ensemble_step": round(1e3 * dt_ens / n_ens, 3),
                            "replica0_vs_single_trajectory": {"dT_K": round(abs(reps[0].T_end - single.temperature()), 4),
                                                               "dE_eV": round(abs(reps[0].energy - single.energy), 5)}}
        configs["C4_md"]["ensemble"] = {"what": "R replicas of the 256-atom cell in lockstep, one predict_structure call (device graph build + sweep "
                                                "of all replicas) per step; replica-steps/s per GPU", **ens}
    # ---- C2 batch-size curve: the headline workload at 128 ... 4096 structures, resident (where min_atoms_per_batch sits on it) ----
    if ranks.rank == 0 and "C1" in legs and not getattr(args, "dry_run", False):
        curve = {}
        for nb in (64, 128, 256, 512, 1024, 2048, 4096):
            try:
                bb = eng.build_batch(workload_structures(nb, 0))
            except Exception as exc:  # noqa: BLE001  (memory on a shared device)
                curve[str(nb)] = {"error": str(exc)[:80]}
                continue

            def step_nb():
                eng.predict(bb, "efs")
                return eng.download(bb, "efs")

            dt, _ = timed(step_nb, 5)
            curve[str(nb)] = {"ms_per_step": round(1e3 * dt, 3), "structures_per_s": round(nb / dt, 1), "atoms": int(bb.packed.n_atoms)}
            bb.free()
        configs["C2_batch_curve"] = {"workload": "headline workload (perturbed LiMnO2 5x1x1, 40 atoms) at other batch sizes, task efs, device-resident, "
                                                 "best of 5 steps", "by_structures": curve}
    # ---- C5: one fine-tuning epoch, data-parallel: Trainer step with the full CombinedLoss (E + F + S + magmom) --------
    if args.train_structures > 0 and "C5" in legs:
        from chgnet_amd.trainer import TrainStep

        per_rank = max(1, args.train_structures // ranks.world)
        bs = min(args.structures, per_rank)
        n_steps = (per_rank + bs - 1) // bs
        conv6 = CrystalGraphConverter(atom_graph_cutoff=6, bond_graph_cutoff=3)
        first = 10_000_000 + ranks.rank * per_rank                       # structures no other config uses
        batches = [[conv6(s) for s in workload_structures(min(bs, per_rank - i * bs), first + i * bs)] for i in range(n_steps)]
        rng = np.random.default_rng(ranks.rank)
        p0 = model.predict_graph(batches[0][:8], task="em")

        def labels_for(b_):
            return {"e": np.float32(p0[0]["e"]) + rng.normal(0, 0.05, len(b_)).astype(np.float32),
                    "f": [rng.normal(0, 0.05, (len(g_.atomic_number), 3)).astype(np.float32) for g_ in b_],
                    "s": [rng.normal(0, 0.2, (3, 3)).astype(np.float32) for _ in b_],
                    "m": [np.abs(rng.normal(0.5, 0.2, len(g_.atomic_number))).astype(np.float32) for g_ in b_]}

        labels = [labels_for(b_) for b_ in batches]
        if ranks.rank == 0:      # gradient sample for the CPU leg, taken before any optimizer step
            from chgnet_amd.trainer import CombinedLoss

            gs = batches[0][:3]
            tg = {k: v[:3] for k, v in labels[0].items()}
            pred = model.forward(gs, task="efsm")
            _, g = CombinedLoss(target_str="efsm").gradients(tg, pred)
            checks["C5_train_epoch"] = ("weight-gradients", gs, tg, model.backward(g.get("e"), g.get("m"), g.get("f"), g.get("s")))
            model.release_forward_state()
        results = {}
        for targets in ("efsm", "em"):
            step = TrainStep(model, targets=targets, learning_rate=1e-4, comm=ranks.comm)
            use = slice(0, n_steps if targets == "efsm" else min(n_steps, 3))
            # warm-up through the epoch's own path (two steps: both device arenas of the upload-ahead pipeline, first touch)
            step.run_epoch(batches[:2], labels[:2], upload_ahead=True)
            step.seconds.clear()
            ranks.barrier()
            t0 = time.perf_counter()
            losses = [info["loss"] for info in step.run_epoch(batches[use], labels[use], upload_ahead=True)]
            eng.synchronize()
            ranks.barrier()
            dt = ranks.max_over_ranks(time.perf_counter() - t0)
            n_done = sum(len(b_) for b_ in batches[use]) * ranks.world
            results[targets] = (n_done, dt, len(losses), losses)
            if targets == "efsm":
                split_ms = {k: round(1e3 * v / max(step.seconds.get("calls", 1), 1), 2) for k, v in step.seconds.items() if k != "calls"}
            if targets == "efsm":   # three more steps with per-kernel HIP events (outside the timed region).  EVERY rank takes them: a
                # step ends in the gradient all-reduce, and a collective entered by rank 0 alone would block for ever (found by the
                # two-rank dry run, round 5); only rank 0 keeps the events
                prof_runs = []
                for _ in range(3):
                    if ranks.rank == 0:
                        eng.profile(True)
                        eng.profile_reset()
                    step(batches[0], labels[0])
                    eng.synchronize()
                    if ranks.rank == 0:
                        prof_runs.append(eng.profile_read())
                        eng.profile(False)
                if ranks.rank == 0:
                    # per label: the MEDIAN step (and the min / max next to it below), so that one odd step cannot reach the record
                    labels_t = sorted({k for run in prof_runs for k in run})
                    train_prof = {k: (int(np.median([run.get(k, (0, 0.0))[0] for run in prof_runs])),
                                      float(np.median([run.get(k, (0, 0.0))[1] for run in prof_runs]))) for k in labels_t}
                    train_prof_range = {k: (float(min(run.get(k, (0, 0.0))[1] for run in prof_runs)),
                                            float(max(run.get(k, (0, 0.0))[1] for run in prof_runs))) for k in labels_t}
                    n_angles_b = sum(len(g_.bond_graph) for g_ in batches[0])
                    n_dir_b = sum(len(g_.atom_graph) for g_ in batches[0])
        model.release_forward_state()
        if ranks.rank == 0:
            n_done, dt, ns, losses = results["efsm"]
            n1, dt1, ns1, _ = results["em"]
            configs["C5_train_epoch"] = {
                "workload": f"one epoch over {n_done} perturbed LiMnO2 5x1x1 cells (40 atoms) with synthetic energy / force / stress / magmom "
                            f"labels, {ranks.world} rank(s) x {ns} step(s) of {bs} structures: host graphs -> pack into page-locked memory and upload (next batch, on a "
                            "helper thread, the copy under the current step's backward sweeps) -> forward(efsm) -> CombinedLoss(MSE, target efsm) -> chg_backward (136 parameter tensors; "
                            "tangent sweep + two-adjoint reverse sweep for the force / stress terms) -> all-reduce of the 1.65 MB gradient -> "
                            "Adam -> weights back on the engine",
                "seconds": round(dt, 3), "structures_per_s": round(n_done / dt, 1), "ms_per_step": round(1e3 * dt / ns, 2),
                "loss_first_last": [float(f"{losses[0]:.4g}"), float(f"{losses[-1]:.4g}")],
                "energy_magmom_terms_only": {"structures_per_s": round(n1 / dt1, 1), "ms_per_step": round(1e3 * dt1 / ns1, 2),
                                             "what": "target em: first-order reverse sweep only (fused kernels)"}}
            # the training step's own kernels (HIP events of one step): labels t2_* = fused second-order sweep
            # (csrc/kernels_train2_tile.h), t2_wgrad = k_xty weight-gradient contractions, the rest = forward + force sweep
            ranked_t = sorted(train_prof.items(), key=lambda kv: -kv[1][1])
            configs["C5_train_epoch"]["kernel_ms_per_step"] = {k: round(ms, 3) for k, (_, ms) in ranked_t[:16]}
            configs["C5_train_epoch"]["kernel_ms_per_step_min_max"] = {k: [round(train_prof_range[k][0], 3), round(train_prof_range[k][1], 3)]
                                                                       for k, _ in ranked_t[:16]}
            configs["C5_train_epoch"]["kernel_ms_note"] = "median of three profiled steps after the timed epoch (eager launches, HIP events per label)"
            configs["C5_train_epoch"]["device_ms_per_step"] = round(sum(ms for _, ms in train_prof.values()), 2)
            configs["C5_train_epoch"]["wall_ms_per_step_split"] = split_ms   # forward = upload + predict + download; the rest of ms_per_step: waiting for the packer
            dom_t = next((k for k, _ in ranked_t if k in TRAIN_KERNEL_MODEL), None)
            if dom_t is not None:
                unit, flop_u, byte_u, what = TRAIN_KERNEL_MODEL[dom_t]
                n_l, ms_t = train_prof[dom_t]
                units = n_angles_b if unit == "n_angles" else n_dir_b
                t_s = ms_t / n_l * 1e-3
                tfl, gbs = units * flop_u / t_s / 1e12, units * byte_u / t_s / 1e9
                # re-based like the headline's roofline: the contractions of this kernel run in the SPLIT form (three
                # f16 MFMAs per f32 product), so their matrix ceiling is 2500 / 3 f32-equivalent TFLOP/s and the balance against HBM
                # ~104 flop/B; this kernel's algorithmic intensity decides which side it is classified on
                split_peak = PEAK_F16_MFMA_TFLOPS / SPLIT_MFMA_PER_PRODUCT
                bound = "mfma" if flop_u / byte_u > split_peak * 1e3 / PEAK_HBM_GBS else "hbm"
                rt = {"kernel": dom_t, "what": what, "launches_per_step": n_l, "avg_launch_ms": round(ms_t / n_l, 3),
                      "avg_launch_ms_min_max": [round(train_prof_range[dom_t][0] / max(n_l, 1), 3), round(train_prof_range[dom_t][1] / max(n_l, 1), 3)],
                      "units_per_launch": int(units), "flop_per_unit": flop_u, "bytes_per_unit": byte_u, "bound": bound,
                      "matrix_form": "3 x f16 16x16x32 split, f32 accumulate"}
                if bound == "mfma":
                    rt.update(achieved=round(tfl, 2), peak=round(split_peak, 1), unit="TFLOP/s", frac=round(tfl / split_peak, 4))
                else:
                    rt.update(achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4))
                rt.update(f32_equivalent_tflops=round(tfl, 2), frac_mfma_f16=round(tfl * SPLIT_MFMA_PER_PRODUCT / PEAK_F16_MFMA_TFLOPS, 4),
                          algorithmic_gbs=round(gbs, 1), frac_hbm_algorithmic=round(gbs / PEAK_HBM_GBS, 4),
                          frac_f32_mfma_continuity=round(tfl / PEAK_FP32_MFMA_TFLOPS, 4),
                          note="factorised f32-equivalent flops of the contractions the kernel runs (primal + tangent forward, two adjoints "
                               "back) against the split-f16 matrix roof, algorithmic bytes (row dumps the weight-gradient contractions read "
                               "back + rows in / out) against HBM; median of three profiled steps; frac_f32_mfma_continuity = what rounds 3-4 "
                               "quoted as frac; rocprofv3 of the training step: profiles/r05_train_kernel_stats.csv")
                configs["C5_train_epoch"]["roofline_train"] = rt
    if hasattr(model, "_engine"):
        model._engine = None   # the bench owns the engine
    return configs, checks


class DryEngine:
    """Stand-in for ``Engine`` in the dry run: nothing to synchronise, nothing to profile."""

    def synchronize(self) -> None:
        pass

    def profile(self, on) -> None:  # noqa: ARG002
        pass

    def profile_reset(self) -> None:
        pass

    def profile_read(self) -> dict:
        return {}

    def upload(self, packed):  # noqa: ARG002
        return object()          # TrainStep.run_epoch(upload_ahead=True) hands it back to DryModel.forward as device_batch


class DryModel:
    """Stand-in for ``CHGNet`` in the dry run (no GPU, no engine, no oracle): the subset of its surface the sharded legs of
    ``run_configs`` and ``TrainStep`` use, with results that are FUNCTIONS OF THE INPUT -- a structure's energy is
    ``expected_energy(n_atoms)``, a parameter gradient is ``(rank + 1) * (step + 1)`` -- so that the all-gathered energy table and
    the all-reduced gradients can be checked value by value on every rank."""

    def __init__(self, weights: dict, rank: int, world: int) -> None:
        self._sd = {k: np.asarray(v, np.float32).copy() for k, v in weights.items()}
        self.rank, self.world, self.model_args, self.backward_calls = rank, world, {}, 0
        self.seen_mean_gradients: list[float] = []
        self.engine = DryEngine()

    @staticmethod
    def expected_energy(n_atoms: int) -> np.float32:
        return np.float32(-7.0 - 0.001 * n_atoms)

    def _pred(self, n: int, task: str) -> dict:
        out = {"e": self.expected_energy(n)}
        if "f" in task:
            out["f"] = np.zeros((n, 3), np.float32)
        if "s" in task:
            out["s"] = np.zeros((3, 3), np.float32)
        if "m" in task:
            out["m"] = np.zeros(n, np.float32)
        return out

    def predict_structure(self, structures, *, task="efsm", batch_size=16, **kw):  # noqa: ARG002
        single = hasattr(structures, "frac_coords")
        res = [self._pred(len(s), task) for s in ([structures] if single else structures)]
        return res[0] if single else res

    def predict_graph(self, graphs, *, task="efsm", batch_size=16, **kw):  # noqa: ARG002
        single = hasattr(graphs, "atom_graph")
        res = [self._pred(len(g.atomic_number), task) for g in ([graphs] if single else graphs)]
        return res[0] if single else res

    def forward(self, graphs, *, task="e", device_batch=None, **kw):  # noqa: ARG002
        from chgnet_amd.pack import PackedBatch

        n_at = np.diff(graphs.atom_off).astype(int).tolist() if isinstance(graphs, PackedBatch) else [len(g.atomic_number) for g in graphs]
        out = {"atoms_per_graph": np.asarray(n_at, np.int64), "e": np.array([self.expected_energy(n) for n in n_at], np.float32)}
        if "f" in task:
            out["f"] = [np.zeros((n, 3), np.float32) for n in n_at]
        if "s" in task:
            out["s"] = [np.zeros((3, 3), np.float32) for _ in n_at]
        if "m" in task:
            out["m"] = [np.zeros(n, np.float32) for n in n_at]
        return out

    def backward(self, e_grad=None, m_grad=None, f_grad=None, s_grad=None, comm=None):  # noqa: ARG002
        self.backward_calls += 1
        return {k: np.full(v.shape, float((self.rank + 1) * self.backward_calls), np.float32) for k, v in self._sd.items()}

    def state_dict(self) -> dict:
        return self._sd

    def load_state_dict(self, sd: dict) -> None:
        # what Adam applied came out of the gradient all-reduce: remember its (uniform) value for the check in dry_run
        self._sd = {k: np.asarray(v, np.float32) for k, v in sd.items()}

    def release_forward_state(self) -> None:
        pass


def dry_run(args, ranks: Ranks) -> None:
    """No GPU: argument handling, rank spawning, CPU pinning, the process group and the all-gather -- and the two SHARDED legs of
    ``run_configs`` (C3 ragged sweep, C5 data-parallel fine-tuning epoch) walked with ``DryModel`` on small sizes: LPT shards,
    zero-padded all-gather of the energies, max-over-ranks timing, ``TrainStep`` with its gradient all-reduce and Adam.  The
    world-size-2 gloo test (tests/test_distributed_cpu.py) runs this; it is what stands in for a multi-GPU node until one exists."""
    import chgnet_amd.trainer as trainer_mod

    e = np.full(4, float(ranks.rank), np.float32)
    table = ranks.all_gather(e)
    ranks.barrier()
    line = {"metric": "dry-run", "n_gpus": ranks.world, "ranks_in_all_gather": sorted({int(v) for v in table}),
            "backend": ranks.backend if ranks.dist is not None else None, "cpu_affinity": ranks.cpu_affinity}
    weights = dict(np.load(os.path.join(REPO, "tests", "golden", "weights_seed0.npz")))
    model = DryModel(weights, ranks.rank, ranks.world)
    small = argparse.Namespace(**{**vars(args), "sweep_structures": 24, "sweep_chunk": 8, "train_structures": 8 * ranks.world, "structures": 4})
    seen = []
    real_allreduce = trainer_mod.allreduce_gradients

    def spying_allreduce(grads, *a, **kw):       # the averaged gradient every rank must see: mean over ranks of (rank + 1) * call
        out = real_allreduce(grads, *a, **kw)
        seen.append(float(next(iter(out.values())).reshape(-1)[0]))
        return out

    trainer_mod.allreduce_gradients = spying_allreduce
    try:
        import contextlib

        with contextlib.redirect_stdout(sys.stderr):
            configs, _ = run_configs(DryEngine(), weights, ranks, small, model=model, legs=("C3", "C5"))
    finally:
        trainer_mod.allreduce_gradients = real_allreduce
    # backward call 1 of rank 0 is the gradient sample it takes for the CPU leg (no all-reduce); the all-reduced calls follow in order on
    # every rank: call i (0-based) averages (r + 1) * (i + 1 + [r == 0]) over the ranks
    want = [sum((r + 1) * (i + 1 + (1 if r == 0 else 0)) for r in range(ranks.world)) / ranks.world for i in range(len(seen))]
    ok = len(seen) > 0 and np.allclose(seen, want)
    if ranks.rank == 0:
        line["legs"] = {k: {kk: v[kk] for kk in ("structures_per_s", "energies_gathered", "shard_atoms_max_over_mean", "per_rank_seconds",
                                                 "ms_per_step", "loss_first_last") if kk in v} for k, v in configs.items()}
        line["allreduced_gradient_values"] = seen[:4]
    all_ok = ranks.max_over_ranks(0.0 if ok else 1.0) == 0.0
    ranks.close()
    if not all_ok:
        raise SystemExit("bench.py dry run: the all-reduced gradients are not the mean over the ranks")
    if ranks.rank == 0:
        emit(line)


def shared_device_run(args, ranks: Ranks) -> None:
    """``--gpus N --shared-device``: N ranks, each with its OWN ``chg_engine``, all on GPU 0, collectives over gloo -- the only N > 1 run
    a one-GPU lease can give.  It measures NO scaling (the ranks share one device) and reports none: what it
    proves is that the REAL sharded legs hold together at world size N -- the C2 step with its energy all-gather, the C3 LPT-sharded
    ragged sweep, and C5 data-parallel fine-tuning steps -- by checking every slot of the gathered energy tables against rank 0's own
    single-engine results for ALL shards, and the all-reduced gradient against the mean of the per-shard gradients."""
    from chgnet_amd import CrystalGraphConverter
    from chgnet_amd.distributed import shard_indices
    from chgnet_amd.model import CHGNet
    from chgnet_amd.trainer import TrainStep

    world, rank = ranks.world, ranks.rank
    weights = dict(np.load(os.path.join(REPO, "tests", "golden", "weights_seed0.npz")))
    model = CHGNet(state_dict=weights, use_device=0)
    model.graph_converter.set_isolated_atom_response("ignore")
    eng = model.engine
    conv = CrystalGraphConverter(atom_graph_cutoff=6, bond_graph_cutoff=3)
    n_c2 = min(args.structures, 256)
    flags, legs = {}, {}

    def check(name, ok, **info):
        flags[name] = bool(ok)
        legs[name] = {"ok": bool(ok), **info}

    # ---- C2: this rank's share of the headline workload, energies all-gathered; rank 0 recomputes every share on its engine ----
    batch = eng.upload(build_workload(n_c2, first_seed=rank * n_c2))
    for _ in range(2):
        eng.predict(batch, "efs")
        res = eng.download(batch, "efs")
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.predict(batch, "efs")
        res = eng.download(batch, "efs")
        table = ranks.all_gather(res["e"])
    ranks.barrier()
    c2_ms = 1e3 * ranks.max_over_ranks(time.perf_counter() - t0) / 3
    batch.free()
    if rank == 0:
        worst = 0.0
        for r in range(world):
            b_r = eng.upload(build_workload(n_c2, first_seed=r * n_c2))
            eng.predict(b_r, "efs")
            e_r = eng.download(b_r, "efs")["e"]
            b_r.free()
            worst = max(worst, float(np.abs(table[r * n_c2:(r + 1) * n_c2] - e_r).max()))
        check("C2_energy_table", worst <= 5e-6 and len(table) == world * n_c2, max_abs_err=worst, structures_per_rank=n_c2, ms_per_step_all_ranks_one_gpu=round(c2_ms, 3))

    # ---- C3: the LPT-sharded ragged sweep (run_configs' leg, smaller), every slot of the gathered table vs rank 0's own full sweep ----
    n_total = min(args.sweep_structures, 300) * world
    counts = [sweep_atom_count(i) for i in range(n_total)]
    shards = shard_indices([float(c) for c in counts], world)
    structs = [sweep_structure(i) for i in shards[rank]]
    width = max(len(sh) for sh in shards)
    preds = model.predict_structure(structs, task="efs", batch_size=min(args.sweep_chunk, 100))
    e_local = np.zeros(width, np.float32)
    e_local[:len(preds)] = [p["e"] for p in preds]
    table = ranks.all_gather(e_local)
    if rank == 0:
        worst, pad_ok = 0.0, True
        for r, sh in enumerate(shards):
            ref = model.predict_structure([sweep_structure(i) for i in sh], task="e", batch_size=100)
            got = table[r * width:r * width + len(sh)]
            worst = max(worst, float(np.abs(got - np.array([p["e"] for p in ref], np.float32)).max()))
            pad_ok &= not np.any(table[r * width + len(sh):(r + 1) * width])
        check("C3_energy_table", worst <= 5e-6 and pad_ok, max_abs_err=worst, structures=n_total, shard_sizes=[len(sh) for sh in shards])

    # ---- C5: data-parallel fine-tuning steps: all-reduced gradient == mean of the per-shard gradients; identical weights afterwards ----
    import chgnet_amd.trainer as trainer_mod

    bs = 32
    per_rank_graphs = [[conv(s) for s in workload_structures(bs, 7000 + 100 * r)] for r in range(world)]
    def synthetic_targets(gs, seed):
        rng_ = np.random.default_rng(seed)
        return {"e": rng_.normal(0, 0.05, len(gs)).astype(np.float32),
                "f": [rng_.normal(0, 0.05, (len(g_.atomic_number), 3)).astype(np.float32) for g_ in gs],
                "s": [rng_.normal(0, 0.2, (3, 3)).astype(np.float32) for _ in gs],
                "m": [np.abs(rng_.normal(0.5, 0.2, len(g_.atomic_number))).astype(np.float32) for g_ in gs]}

    per_rank_targets = [synthetic_targets(gs, seed=r) for r, gs in enumerate(per_rank_graphs)]
    seen = {}
    real_allreduce = trainer_mod.allreduce_gradients

    def spying_allreduce(grads, *a, **kw):
        out = real_allreduce(grads, *a, **kw)
        seen.setdefault("local", {k: np.array(v, copy=True) for k, v in grads.items()})
        seen.setdefault("reduced", {k: np.array(v, copy=True) for k, v in out.items()})
        return out

    trainer_mod.allreduce_gradients = spying_allreduce
    try:
        step = TrainStep(model, targets="efsm", learning_rate=1e-3)
        infos = [step(per_rank_graphs[rank], per_rank_targets[rank]) for _ in range(2)]
    finally:
        trainer_mod.allreduce_gradients = real_allreduce
    after = model.state_dict()
    key = "atom_conv_layers.0.twoBody_atom.mlp_core.layers.0.weight"
    same_weights = ranks.all_gather(after[key].reshape(-1)[:64].astype(np.float32)).reshape(world, -1)
    if rank == 0:
        ref_model = CHGNet(state_dict=weights, use_device=0)
        ref_model._engine = None
        want = None
        for r in range(world):           # the first step's gradient of every shard, on rank 0's own (fresh) model
            pred = ref_model.forward(per_rank_graphs[r], task="efsm")
            _, g = step.loss.gradients(per_rank_targets[r], pred)
            gr = ref_model.backward(g.get("e"), g.get("m"), g.get("f"), g.get("s"))
            want = gr if want is None else {k: want[k] + gr[k] for k in want}
        ref_model.release_forward_state()
        worst = 0.0
        for k, v in want.items():
            scale = float(np.abs(v).max()) / world
            if scale > 0:
                worst = max(worst, float(np.abs(seen["reduced"][k] - v / world).max()) / scale)
        check("C5_allreduced_gradient", worst <= 2e-4, max_rel_err=worst, tensors=len(want), loss_first_last=[round(infos[0]["loss"], 5), round(infos[-1]["loss"], 5)])
        check("C5_weights_identical_on_all_ranks", bool(np.all(same_weights == same_weights[0])))
        if ref_model._engine is not None:
            ref_model._engine.close()
    ok_all = ranks.max_over_ranks(0.0 if all(flags.values()) else 1.0) == 0.0
    ranks.close()
    if rank == 0:
        emit({"metric": "shared-device plumbing run (NOT a scaling measurement)", "n_gpus": 1, "process_group_ranks": world, "shared_device": True,
              "backend": "gloo (torch.distributed)", "cpu_affinity": ranks.cpu_affinity, "parity": flags, "legs": legs, "ok": bool(ok_all)})
    if not ok_all:
        raise SystemExit("bench.py --shared-device: a parity check failed")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--structures", type=int, default=1024, help="structures per GPU per step")
    ap.add_argument("--sweep-structures", type=int, default=12500, help="C3: structures per GPU (BASELINE: 100k over 8 GPUs)")
    ap.add_argument("--sweep-chunk", type=int, default=1000, help="C3: batch_size handed to predict_structure")
    ap.add_argument("--md-steps", type=int, default=1000, help="C4: timed MD steps (BASELINE: 1000)")
    ap.add_argument("--train-structures", type=int, default=10240, help="C5: structures in the fine-tuning epoch (all ranks together); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="headline workload only")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: process-group plumbing only (gloo)")
    ap.add_argument("--comm", choices=("auto", "torch", "rccl"), default="auto",
                    help="exchange steps: 'rccl' = the engine library's own RCCL entry points, failing if the communicator cannot be "
                         "created or ncclCommCount differs from the world size; 'torch' = torch.distributed (backend nccl = RCCL); "
                         "'auto' (default) = rccl, falling back to torch with a comm_note in the line")
    ap.add_argument("--shared-device", action="store_true",
                    help="N ranks with their own engines on GPU 0, gloo collectives: the real sharded legs at world size N on a one-GPU box "
                         "(parity of the gathered tables / all-reduced gradients; no scaling number)")
    ap.add_argument("--total-structures", type=int, default=0,
                    help="strong-scaling mode: this many structures of the headline workload in TOTAL, split evenly over the ranks "
                         "(the default mode is weak: --structures per rank)")
    args = ap.parse_args()

    if args.gpus > 1 and "LOCAL_RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))   # not under a launcher: start the N ranks ourselves
    claim_stdout()
    ranks = Ranks(args)
    if args.dry_run:
        return dry_run(args, ranks)
    if args.shared_device:
        return shared_device_run(args, ranks)
    rank, world = ranks.rank, ranks.world

    from chgnet_amd.engine import Engine
    from chgnet_amd.pack import pack_batch, pack_weights

    weights = dict(np.load(os.path.join(REPO, "tests", "golden", "weights_seed0.npz")))
    first_seed = rank * args.structures
    if args.total_structures and args.total_structures < world:
        raise SystemExit(f"bench.py: --total-structures {args.total_structures} is less than the {world} ranks (every rank needs a structure)")
    if args.total_structures:     # strong scaling: a fixed batch split evenly (identical structures sizes: no balancing needed here; C3 has it)
        share = [args.total_structures // world + (1 if r < args.total_structures % world else 0) for r in range(world)]
        args.structures, first_seed = share[rank], sum(share[:rank])
    graphs = build_workload(args.structures, first_seed=first_seed)
    eng = Engine(pack_weights(weights), ranks.local_rank)
    packed = pack_batch(graphs)
    batch = eng.upload(packed)   # inputs resident in HBM before the timed region

    gathered = None
    gather_width = -(-args.total_structures // world) if args.total_structures else args.structures   # equal width per rank (zero-padded)

    def step():
        nonlocal gathered
        eng.predict(batch, "efs")
        res = eng.download(batch, "efs")
        if ranks.comm is not None:      # all-gather of the energies from HBM on the engine's stream (chg_batch_all_gather_energy)
            gathered = eng.all_gather_energy(batch, ranks.comm, gather_width)
        elif ranks.dist is not None:
            gathered = ranks.all_gather(res["e"])
        return res

    def barrier():
        eng.synchronize()
        ranks.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    my_elapsed = time.perf_counter() - t0
    elapsed = ranks.max_over_ranks(my_elapsed)
    per_rank_ms = 1e3 * ranks.all_gather(np.array([my_elapsed], np.float32)).astype(np.float64) / args.steps   # imbalance is visible, not only the max
    assert np.isfinite(res["e"]).all() and np.isfinite(res["f"]).all() and np.isfinite(res["s"]).all()
    ms_per_step = 1e3 * elapsed / args.steps
    total_structures = args.total_structures or world * args.structures
    value = total_structures / (elapsed / args.steps)
    energies_in_gather = int(np.isfinite(gathered).sum()) if gathered is not None else args.structures

    # device-only time of one step and the per-kernel split (HIP events on the engine stream)
    eng.timer_start()
    for _ in range(args.steps):
        eng.predict(batch, "efs")
    dev_ms = eng.timer_stop_ms() / args.steps
    eng.profile(True)
    eng.profile_reset()
    prof_steps = max(1, min(args.steps, 3))
    for _ in range(prof_steps):
        eng.predict(batch, "efs")
        eng.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    stream_gbs = eng.stream_copy_gbs(1 << 30, 10) if rank == 0 else None

    # secondary, informative only: structures on the host -> graph built on the device -> E/F/S on the host
    structs = workload_structures(args.structures, first_seed=first_seed)
    e2e = []
    for _ in range(3):
        t0 = time.perf_counter()
        b2 = eng.build_batch(structs)
        eng.predict(b2, "efs")
        eng.download(b2, "efs")
        e2e.append(time.perf_counter() - t0)
        b2.free()
    e2e_ms = 1e3 * min(e2e)

    configs, checks = ({}, {})
    if not args.no_configs:
        import contextlib

        with contextlib.redirect_stdout(sys.stderr):   # CHGNet / CHGNetCalculator print their banners; stdout carries ONE line
            configs, checks = run_configs(eng, weights, ranks, args)
    if rank == 0:
        sample = [0, args.structures // 2, args.structures - 1]
        o = packed.atom_off
        checks["C2_headline"] = ([graphs[i] for i in sample],
                                 [{"e": res["e"][i], "f": res["f"][o[i]:o[i + 1]], "s": res["s"][i]} for i in sample])

    line = None
    if rank == 0:
        total_ms = sum(ms for _, ms in prof.values()) or 1.0
        ranked = sorted(prof.items(), key=lambda kv: -kv[1][1])
        dom = next((k for k, _ in ranked if k in KERNEL_MODEL), ranked[0][0])
        # HBM bytes per launch from the rocprofv3 PMC passes of this same command (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs,
        # summarised by profiles/summarize.py with the gfx950 FETCH_SIZE x2 correction) -- quoted only when they were taken on THESE
        # kernel sources (the summary is stamped with the hash of csrc/)
        src_hash, pmc, pmc_note = csrc_hash(), {}, None
        try:
            with open(os.path.join(REPO, "profiles", "pmc_latest.json")) as fh:
                pmc = json.load(fh)
            if pmc.get("csrc_hash") != src_hash:
                pmc_note = f"profiles/pmc_latest.json was taken on other kernel sources (csrc hash {pmc.get('csrc_hash')} != {src_hash}): traffic not quoted"
                pmc = {}
        except (OSError, ValueError):
            pmc_note = "profiles/pmc_latest.json missing"
        # The tile kernels contract in the split form (3 f16 MFMAs per f32 product): their matrix ceiling is 2500 / 3 f32-equivalent
        # TFLOP/s, the balance against HBM 2500 / 3 / 8 = 104 flop per byte.  Every kernel is reported against BOTH roofs of the
        # formulation it actually runs: matrix pipe (executed f16 flops / 2500) and HBM (algorithmic bytes, and counted traffic).
        split_peak = PEAK_F16_MFMA_TFLOPS / SPLIT_MFMA_PER_PRODUCT
        tile = {}
        step_flop = 0.0
        for k, (unit_attr, flop_u, byte_u) in KERNEL_MODEL.items():
            if k in prof and prof[k][0]:
                n_l, t_ms = prof[k]
                t_s = t_ms / n_l * 1e-3
                units = getattr(packed, unit_attr)
                fl, by = units * flop_u, units * byte_u
                step_flop += fl * n_l / prof_steps
                split = k in SPLIT_KERNELS
                exec_tf = fl * (SPLIT_MFMA_PER_PRODUCT if split else 1) / t_s / 1e12
                entry = {"avg_launch_ms": round(t_ms / n_l, 4), "units_per_launch": int(units), "flop_per_unit": flop_u, "bytes_per_unit": byte_u,
                         "matrix_form": "3 x f16 16x16x32 split, f32 accumulate" if split else "f32 16x16x4",
                         "f32_equivalent_tflops": round(fl / t_s / 1e12, 2),
                         "frac_mfma_f16": round(exec_tf / PEAK_F16_MFMA_TFLOPS, 4) if split else None,
                         "algorithmic_gbs": round(by / t_s / 1e9, 1), "frac_hbm_algorithmic": round(by / t_s / 1e9 / PEAK_HBM_GBS, 4),
                         "bound": "mfma" if flop_u / byte_u > (split_peak if split else PEAK_FP32_MFMA_TFLOPS) * 1e3 / PEAK_HBM_GBS else "hbm",
                         "frac_f32_mfma_continuity": round(fl / t_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
                if k in pmc:
                    tr = pmc[k]["hbm_bytes_per_launch"]
                    entry.update(traffic_bytes_per_launch=int(tr), traffic_ratio=round(tr / by, 3),
                                 frac_hbm_traffic=round(tr / t_s / 1e9 / PEAK_HBM_GBS, 4),
                                 frac_of_stream_copy_traffic=round(tr / t_s / 1e9 / max(stream_gbs, GUIDE_COPY_GBS), 4) if stream_gbs else None)
                    if "l2_hit_rate" in pmc[k]:
                        entry["l2_hit_rate"] = pmc[k]["l2_hit_rate"]
                tile[k] = entry
        launches, ms = prof[dom]
        avg_ms = ms / max(launches, 1)
        roofline = {"kernel": dom, "avg_launch_ms": round(avg_ms, 4), "share_of_step": round(ms / total_ms, 3)}
        if dom in tile:
            d = tile[dom]
            if d["bound"] == "mfma":
                roofline.update(bound="mfma", achieved=round(d["f32_equivalent_tflops"], 3), peak=round(split_peak, 1), unit="TFLOP/s",
                                frac=round(d["f32_equivalent_tflops"] / split_peak, 4))
            else:
                roofline.update(bound="hbm", achieved=d["algorithmic_gbs"], peak=PEAK_HBM_GBS, unit="GB/s", frac=d["frac_hbm_algorithmic"])
            roofline.update(units_per_launch=d["units_per_launch"], flop_per_unit=d["flop_per_unit"], bytes_per_unit=d["bytes_per_unit"],
                            algorithmic_gbs=d["algorithmic_gbs"], algorithmic_tflops=d["f32_equivalent_tflops"],
                            frac_mfma_f16=d["frac_mfma_f16"], frac_hbm_algorithmic=d["frac_hbm_algorithmic"],
                            frac_hbm_traffic=d.get("frac_hbm_traffic"), traffic_ratio=d.get("traffic_ratio"),
                            frac_f32_mfma_continuity=d["frac_f32_mfma_continuity"])
            roofline["traffic"] = d.get("traffic_bytes_per_launch")
            if d.get("traffic_bytes_per_launch") is not None:
                roofline["traffic_source"] = pmc[dom]["profile"]
        else:
            roofline["traffic"] = None
        if pmc_note:
            roofline["traffic_note"] = pmc_note
        roofline["csrc_hash"] = src_hash
        roofline["tile_kernels"] = tile
        roofline["peak_note"] = (f"classification against the formulation the kernels run: split contractions (3 f16 MFMAs per f32 product) have a matrix "
                                 f"ceiling of {PEAK_F16_MFMA_TFLOPS:.0f} / 3 = {split_peak:.0f} f32-equivalent TFLOP/s and balance against HBM at "
                                 f"{split_peak * 1e3 / PEAK_HBM_GBS:.0f} flop/B; frac_mfma_f16 = executed f16 flops / {PEAK_F16_MFMA_TFLOPS:.0f}; "
                                 f"frac_f32_mfma_continuity = f32-equivalent TFLOP/s / {PEAK_FP32_MFMA_TFLOPS} (the number earlier rounds quoted as frac). "
                                 f"Neither roof is near: what bounds the tile kernels is bracketed by the experiments of profiles/r05_experiments.md")
        roofline["whole_step_frac_f32_mfma_continuity"] = round(step_flop / (dev_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
        roofline["whole_step_frac_mfma_f16"] = round(step_flop * SPLIT_MFMA_PER_PRODUCT / (dev_ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4)
        # the "measured HBM roofline" the fractions below are quoted against is NEVER below the guide's float4-copy figure: a box (or a
        # launch geometry) on which this process's copy kernel is slow must not flatter the kernels
        copy_ceiling = max(stream_gbs, GUIDE_COPY_GBS)
        hbm = {"bound": "hbm", "stream_copy_gbs": round(stream_gbs, 1), "guide_copy_gbs": GUIDE_COPY_GBS, "copy_ceiling_gbs": round(copy_ceiling, 1),
               "spec_gbs": PEAK_HBM_GBS, "kernels": {},
               "note": "achieved = compulsory bytes per launch / mean launch time (HIP events); stream_copy = 1 GiB read + 1 GiB write "
                       "device copy kernel timed in this process (best of 12 launch shapes: interleaved / 16 KiB segments, non-temporal / "
                       "plain, 8-128 workgroups per CU); frac_of_stream = achieved / max(that, the guide's 6.29 TB/s float4 copy)"}
        for k, nbytes in hbm_model(packed).items():
            if k in prof and prof[k][0]:
                t_ms = prof[k][1] / prof[k][0]
                gbs = nbytes / (t_ms * 1e-3) / 1e9
                hbm["kernels"][k] = {"bytes_per_launch": int(nbytes), "avg_launch_ms": round(t_ms, 4), "achieved_gbs": round(gbs, 1),
                                     "frac_of_stream": round(gbs / copy_ceiling, 4), "frac_of_spec": round(gbs / PEAK_HBM_GBS, 4)}
        # what the wave slots of each kernel spend their cycles on (rocprofv3 --pmc SQ_* pass of this command, profiles/summarize.py):
        # the embedding kernels are neither HBM- nor matrix-bound, the copy-rate fractions above are quoted for continuity only
        try:
            with open(os.path.join(REPO, "profiles", "sq_latest.json")) as fh:
                sqc = json.load(fh)
            if sqc.get("csrc_hash") != src_hash:
                raise KeyError("SQ counters of other kernel sources")
            for k, v in sqc["kernels"].items():
                if k in hbm["kernels"]:
                    hbm["kernels"][k]["sq_counters"] = v
                elif k in roofline.get("tile_kernels", {}):
                    roofline["tile_kernels"][k]["sq_counters"] = v
                    # the roof these kernels are actually under: vector-ALU issue.  A wave issues vector instructions `active_inst_valu`
                    # of its resident cycles; the tile kernels run two waves per SIMD (256 registers each), so the SIMD's vector port is
                    # busy about twice that -- the third "frac" next to frac_mfma_f16 and frac_hbm_*
                    # (an ESTIMATE, clamped: not a SIMD-level busy counter -- ADVICE r04)
                    roofline["tile_kernels"][k]["frac_valu_issue"] = round(min(1.0, 2.0 * v["active_inst_valu"]), 3)
            if dom in roofline.get("tile_kernels", {}) and "frac_valu_issue" in roofline["tile_kernels"][dom]:
                roofline["frac_valu_issue"] = roofline["tile_kernels"][dom]["frac_valu_issue"]
                roofline["binding_resource"] = ("none saturated: vector issue ~frac_valu_issue of a SIMD's cycles (estimate: 2 waves per SIMD x "
                                                "active_inst_valu), matrix pipe frac_mfma_f16 x 3/2.5, LDS 35-45 %; same-box experiments "
                                                "(profiles/r05_experiments.md): one wave per SIMD costs 1.2-1.4x, no transcendentals -3 %, free "
                                                "LDS weight operands -11 % of the step -- a balanced mix at ~70 % of its serial issue floor")
            hbm["sq_counters_unit"] = sqc["unit"]
        except (OSError, ValueError, KeyError):
            pass
        line = {
            "metric": "structures/s (energy+force+stress) on batched ~50-atom crystals",
            "value": round(value, 2), "unit": "structures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if args.total_structures else "weak",
            "vs_baseline": None,
            "per_rank_ms_per_step": {"min": round(float(per_rank_ms.min()), 3), "median": round(float(np.median(per_rank_ms)), 3),
                                     "max": round(float(per_rank_ms.max()), 3)},
            "dtype": "f32", "data": "synthetic",
            "arithmetic": "f32 storage, f32 accumulation; contractions of the six tile kernels as 3 x f16 MFMA on split f32 operands "
                          "(error of the split contraction <= the f32 MFMA's, profiles/r03_split_lab.txt; E/F/S parity unchanged)",
            "config": {"workload": (f"{args.total_structures} x LiMnO2 5x1x1 (40 atoms, sigma=0.01 frac perturbation) in total, split evenly over the GPUs, task efs"
                                    if args.total_structures else
                                    f"{args.structures} x LiMnO2 5x1x1 (40 atoms, sigma=0.01 frac perturbation) per GPU, task efs"),
                       "structures_per_gpu": args.structures, "atoms": int(packed.n_atoms), "directed_bonds": int(packed.n_directed),
                       "angles": int(packed.n_angles), "bond_graph_nodes": int(packed.n_bnodes),
                       "weights": "random-init 0.3.0 architecture (tests/golden/weights_seed0.npz)",
                       "parallelism": f"structures sharded over {world} GPU(s), RCCL all-gather of energies only",
                       "process_group_ranks": world, "energies_in_all_gather": energies_in_gather, "cpu_affinity_rank0": ranks.cpu_affinity,
                       "comm": ranks.backend if world > 1 else "none (single rank)", "rccl": ranks.rccl_info, "comm_note": ranks.comm_note},
            "device_ms_per_step": round(dev_ms, 3),
            "end_to_end": {"what": "host structures -> device graph build (chg_batch_build) -> predict -> E/F/S on host, per GPU",
                           "ms": round(e2e_ms, 3), "structures_per_s": round(args.structures / (e2e_ms * 1e-3), 1)},
            "device_bytes": batch.device_bytes,
            "roofline": roofline,
            "roofline_hbm": hbm,
            "kernel_ms_per_step": {k: round(v[1] / prof_steps, 3) for k, v in ranked},
            "configs": configs,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"], parity = cpu_leg(weights, graphs[:128], checks)
            try:    # the reference ITSELF, timed where it exists: the build container (tools/cpu_reference_baseline.py; it does not travel here)
                with open(os.path.join(REPO, "profiles", "reference_cpu_baseline.json")) as fh:
                    ref_cpu = json.load(fh)
                line["cpu_baseline"]["reference_in_build_container"] = {k: ref_cpu[k] for k in ("value", "unit", "cores", "batch_size", "kind", "what", "script")}
            except (OSError, ValueError, KeyError):
                line["cpu_baseline"]["reference_in_build_container"] = None
            for name, p in parity.items():
                line["configs"].setdefault(name, {})["parity_vs_oracle"] = p
        else:
            line["cpu_baseline"] = None
    batch.free()
    eng.close()
    ranks.close()
    if line is not None:
        emit(line)


if __name__ == "__main__":
    main()
