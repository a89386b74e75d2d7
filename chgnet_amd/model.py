"""``CHGNet`` -- host-side mirror of the reference model API for the inference path.

Same constructor keywords, ``predict_structure`` / ``predict_graph`` signatures, task
strings, error types/messages, output keys, dtypes and units as reference
chgnet/model/model.py:544-665, and the same (de)serialisation surface
(``as_dict`` / ``from_dict`` / ``from_file`` / ``load``, model.py:667-745).  All
arithmetic happens in the gfx950 HIP engine (``chgnet_amd.engine``); there is no torch
module graph and no CPU fallback.
"""

from __future__ import annotations

import itertools
import math
import os
from collections.abc import Sequence

import numpy as np

from chgnet_amd import VALID_TASKS
from chgnet_amd.graph import CrystalGraph, CrystalGraphConverter
from chgnet_amd.pack import D, N_ELEM, NUM_ANGULAR, NUM_RADIAL, check_model_args, pack_batch, pack_weights

module_dir = os.path.dirname(os.path.abspath(__file__))

_DEFAULT_ARGS = dict(
    atom_fea_dim=64, bond_fea_dim=64, angle_fea_dim=64, composition_model="MPtrj", num_radial=31, num_angular=31,
    n_conv=4, atom_conv_hidden_dim=64, update_bond=True, bond_conv_hidden_dim=64, update_angle=True,
    angle_layer_hidden_dim=0, conv_dropout=0, read_out="ave", mlp_hidden_dims=(64, 64, 64), mlp_dropout=0,
    mlp_first=True, is_intensive=True, non_linearity="silu", atom_graph_cutoff=6, bond_graph_cutoff=3,
    graph_converter_algorithm="fast", cutoff_coeff=8, learnable_rbf=True, gMLP_norm="layer", readout_norm="layer",
    version=None,
)


def _is_structure(obj) -> bool:
    """pymatgen Structure or ours: anything with frac_coords + lattice that is not a sequence of them."""
    return hasattr(obj, "frac_coords") and hasattr(obj, "lattice")


def _is_graph(obj) -> bool:
    return isinstance(obj, CrystalGraph) or (hasattr(obj, "atom_graph") and hasattr(obj, "bond_graph"))


def random_state_dict(model_args: dict, seed: int = 0) -> dict:
    """Randomly initialised parameters with the shapes of the reference ``state_dict``
    (SURVEY 8.0); uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's Linear default."""
    rng = np.random.default_rng(seed)
    L = int(model_args.get("n_conv", 4))
    sd: dict[str, np.ndarray] = {}

    def lin(name, out, inp, bias=True):
        k = 1.0 / math.sqrt(inp)
        sd[name + ".weight"] = rng.uniform(-k, k, (out, inp)).astype(np.float32)
        if bias:
            sd[name + ".bias"] = rng.uniform(-k, k, out).astype(np.float32)

    def ln(name):
        sd[name + ".weight"] = np.ones(D, np.float32)
        sd[name + ".bias"] = np.zeros(D, np.float32)

    if model_args.get("composition_model", "MPtrj") is not None:
        sd["composition_model.fc.weight"] = np.zeros((1, N_ELEM), np.float32)
    sd["atom_embedding.embedding.weight"] = rng.normal(0, 1, (N_ELEM, D)).astype(np.float32)
    n_rad, n_ang = int(model_args.get("num_radial", NUM_RADIAL)), int(model_args.get("num_angular", NUM_ANGULAR))
    sd["bond_basis_expansion.rbf_expansion_ag.frequencies"] = (np.pi * np.arange(1, n_rad + 1)).astype(np.float32)
    sd["bond_basis_expansion.rbf_expansion_bg.frequencies"] = (np.pi * np.arange(1, n_rad + 1)).astype(np.float32)
    sd["angle_basis_expansion.fourier_expansion.frequencies"] = np.arange(1, (n_ang - 1) // 2 + 1).astype(np.float32)
    for n in ("bond_embedding", "bond_weights_ag", "bond_weights_bg"):
        lin(n, D, n_rad, bias=False)
    lin("angle_embedding", D, n_ang, bias=False)
    mlp_out_bias = bool(model_args.get("mlp_out_bias", False))
    for l in range(L):
        p = f"atom_conv_layers.{l}"
        for br in ("mlp_core", "mlp_gate"):
            lin(f"{p}.twoBody_atom.{br}.layers.0", D, 3 * D)
            lin(f"{p}.twoBody_atom.{br}.layers.3", D, D)
        ln(f"{p}.twoBody_atom.bn1")
        ln(f"{p}.twoBody_atom.bn2")
        lin(f"{p}.mlp_out.layers.1", D, D, bias=mlp_out_bias)
    for l in range(L - 1):
        p = f"bond_conv_layers.{l}"
        for br in ("mlp_core", "mlp_gate"):
            lin(f"{p}.twoBody_bond.{br}.layers.0", D, 4 * D)
            lin(f"{p}.twoBody_bond.{br}.layers.3", D, D)
        ln(f"{p}.twoBody_bond.bn1")
        ln(f"{p}.twoBody_bond.bn2")
        lin(f"{p}.mlp_out.layers.1", D, D, bias=mlp_out_bias)
        p = f"angle_layers.{l}"
        for br in ("mlp_core", "mlp_gate"):
            lin(f"{p}.twoBody_bond.{br}.layers.1", D, 4 * D)
        ln(f"{p}.twoBody_bond.bn1")
        ln(f"{p}.twoBody_bond.bn2")
    lin("site_wise", 1, D)
    ln("readout_norm")
    n_hidden = len(model_args.get("mlp_hidden_dims", (64, 64, 64)))     # functions.py:81-91
    for i in range(n_hidden):
        lin(f"mlp.layers.{2 * i}", D, D)
    lin(f"mlp.layers.{2 * n_hidden + 1}", 1, D)
    return sd


class ForwardResult(dict):
    """``CHGNet.forward``'s batch dictionary; ``flat`` holds the same f / s / m values as whole-batch arrays ([N,3], [B,3,3], [N])."""

    flat: dict | None = None


class CHGNet:
    """Crystal Hamiltonian Graph neural Network, inference path on MI355X."""

    def __init__(self, *, state_dict: dict | None = None, use_device: str | int | None = None, seed: int = 0, **kwargs) -> None:
        args = dict(_DEFAULT_ARGS)
        args.update(kwargs)
        if args.get("version") is None:
            args.pop("version", None)
        self.model_args = args
        check_model_args(args)
        self.atom_fea_dim = args["atom_fea_dim"]
        self.bond_fea_dim = args["bond_fea_dim"]
        self.is_intensive = args["is_intensive"]
        self.n_conv = args["n_conv"]
        self.mlp_first = True
        self.graph_converter = CrystalGraphConverter(
            atom_graph_cutoff=args["atom_graph_cutoff"], bond_graph_cutoff=args["bond_graph_cutoff"],
            algorithm=args.get("graph_converter_algorithm", "fast"), verbose=args.pop("converter_verbose", False))
        self._state_dict = {k: _to_numpy(v) for k, v in (state_dict or random_state_dict(args, seed)).items()}
        if args.get("composition_model") is None:
            self._state_dict.pop("composition_model.fc.weight", None)
        self.composition_model = "AtomRef" if "composition_model.fc.weight" in self._state_dict else None
        self._weights = pack_weights(self._state_dict, args)
        self._device = _parse_device(use_device)
        self._engine = None
        # predict_* pack at least this many atoms into one device batch (see _plan_chunks); 0 = chunk by batch_size only
        self.min_atoms_per_batch = 40960
        version_str = f" v{self.version}" if self.version else ""
        print(f"CHGNet{version_str} initialized with {self.n_params:,} parameters")

    # ---- properties mirrored from the reference (model.py:320-328) --------------------------------
    @property
    def version(self) -> str | None:
        return self.model_args.get("version")

    @property
    def n_params(self) -> int:
        return int(sum(v.size for v in self._state_dict.values()))

    @property
    def device(self) -> str:
        return f"cuda:{self._device}"

    def state_dict(self) -> dict:
        return self._state_dict

    @property
    def engine(self):
        """The HIP engine, created on first use; raises if the extension or a gfx950 GPU is missing."""
        if self._engine is None:
            from chgnet_amd.engine import Engine

            self._engine = Engine(self._weights, self._device)
        return self._engine

    def to(self, device) -> "CHGNet":
        dev = _parse_device(device)
        if dev != self._device:
            self._device = dev
            if self._engine is not None:
                self._engine.close()
                self._engine = None
        return self

    def eval(self) -> "CHGNet":
        return self

    # ---- batched forward (model.py:330-387) -----------------------------------------------------------
    def forward(self, graphs: Sequence, *, task: str = "e", return_site_energies: bool = False,
                return_atom_feas: bool = False, return_crystal_feas: bool = False, device_batch=None) -> dict:
        """One device batch from ``graphs`` -> the reference's batch dictionary (model.py:330-387, 427-542):
        ``atoms_per_graph`` int64 [B], ``e`` float32 [B] (eV/atom when ``is_intensive``), and -- by task --
        ``f`` list of [n,3] (eV/A), ``s`` list of [3,3] (GPa), ``m`` list of [n] (mu_B); optional
        ``site_energies`` list of [n], ``atom_fea`` list of [n,64] (features before the last AtomConv),
        ``crystal_fea`` [B,64].  Values are host numpy arrays (the reference returns torch tensors that
        carry an autograd graph; here parameter gradients come from ``backward``, which consumes the
        device-resident state this call leaves behind)."""
        if task not in VALID_TASKS:
            raise ValueError(f"Invalid {task=}. Must be one of {VALID_TASKS}.")
        from chgnet_amd.pack import PackedBatch  # noqa: PLC0415

        if isinstance(graphs, PackedBatch):           # already packed (a data loader packs the next batch while this one runs)
            packed = graphs
            graphs = range(packed.n_struct)
        else:
            graphs = [graphs] if _is_graph(graphs) else list(graphs)
            packed = pack_batch(graphs)
        eng = self.engine
        self.release_forward_state()
        # device_batch: ``packed`` already uploaded (Engine.upload may run on a loader thread while the previous step computes)
        batch = device_batch if device_batch is not None else eng.upload(packed)
        self._fwd_batch, self._fwd_task = batch, task
        eng.predict(batch, task)
        res = eng.download(batch, task, site_energies=return_site_energies, atom_feas=return_atom_feas,
                           crystal_feas=return_crystal_feas)
        off = packed.atom_off
        split = lambda a: [a[off[i]:off[i + 1]] for i in range(len(graphs))]  # noqa: E731
        out: dict = {"atoms_per_graph": np.diff(off).astype(np.int64), "e": res["e"]}
        for key in ("f", "m", "site_energies", "atom_fea"):
            if key in res:
                out[key] = split(res[key])
        if "s" in res:
            out["s"] = [res["s"][i] for i in range(len(graphs))]
        if "crystal_fea" in res:
            out["crystal_fea"] = res["crystal_fea"]
        # the batch arrays as downloaded ride along as an attribute (the dictionary itself stays the reference's): CombinedLoss
        # works on these instead of re-concatenating the per-structure views
        out = ForwardResult(out)
        out.flat = {k: res[k] for k in ("f", "s", "m") if k in res}
        return out

    __call__ = forward

    def backward(self, e_grad=None, m_grad=None, f_grad=None, s_grad=None, comm=None) -> dict:
        """Parameter gradients of ``sum_b e_grad[b] e[b] + sum_i m_grad[i] m[i] + sum_i f_grad[i].f[i] + sum_b s_grad[b]:s[b]``
        for the batch of the last ``forward`` call -- the cotangents are d loss / d prediction as ``CombinedLoss.gradients``
        returns them (``e_grad`` defaults to ones when it is the only term and to none next to another cotangent, the others to none; ``f_grad`` [N,3] over all atoms of the batch,
        ``s_grad`` [B,3,3]).  Returns ``{state_dict key: float32 array}``: what ``loss.backward()`` leaves in ``param.grad``
        in the reference's train step (trainer.py:399-411).  AtomRef is frozen (model.py:179-182): zeros.  Force / stress
        terms run the second-order sweep (one tangent pass + a two-adjoint reverse pass).  ``comm`` (``RcclComm``): the
        gradients are SUMMED over the ranks on the device before they are returned."""
        from chgnet_amd.pack import unpack_weight_grads  # noqa: PLC0415

        batch = getattr(self, "_fwd_batch", None)
        if batch is None:
            raise RuntimeError("backward() needs the device state of a preceding forward() call")
        return unpack_weight_grads(self.engine.backward(batch, e_grad, m_grad, f_grad, s_grad, comm=comm), self._weights)

    def load_state_dict(self, state_dict: dict) -> None:
        """New parameter values (same keys and shapes), e.g. after an optimizer step; the engine is updated in place."""
        new = {k: _to_numpy(v) for k, v in state_dict.items()}
        if set(new) != set(self._state_dict) or any(new[k].shape != v.shape for k, v in self._state_dict.items()):
            raise ValueError("load_state_dict: keys / shapes differ from the model's")
        self._state_dict = new
        self._weights = pack_weights(new, self.model_args)
        if self._engine is not None:
            self._engine.update_weights(self._weights)

    def release_forward_state(self) -> None:
        """Free the device batch kept by the last ``forward`` call."""
        batch = getattr(self, "_fwd_batch", None)
        if batch is not None:
            batch.free()
        self._fwd_batch = None

    # ---- prediction (model.py:544-665) ---------------------------------------------------------------
    def predict_structure(self, structure, *, task: str = "efsm", return_site_energies: bool = False,
                          return_atom_feas: bool = False, return_crystal_feas: bool = False, batch_size: int = 16,
                          min_atoms_per_batch: int | None = None):
        """Predict from structure(s) (pymatgen ``Structure`` or ``chgnet_amd.Structure``).

        ``batch_size`` is the reference's memory knob (model.py:639-650).  Here it is the MINIMUM number of
        structures per device batch: a chunk keeps growing until it holds ``min_atoms_per_batch`` atoms
        (default ``self.min_atoms_per_batch`` = 40,960), because per-structure results do not depend on the
        chunking and 16 small cells leave most of an MI355X idle.  Pass ``min_atoms_per_batch=0`` to make
        ``batch_size`` the hard cap it is in the reference.  A chunk whose arena does not fit in device
        memory is split in two and retried (``EngineOutOfMemory`` only for a single structure)."""
        if self.graph_converter is None:
            raise ValueError("graph_converter cannot be None!")
        single = _is_structure(structure)
        structures = [structure] if single else list(structure)
        valid_tasks = VALID_TASKS
        if task not in valid_tasks:
            raise ValueError(f"Invalid {task=}. Must be one of {valid_tasks}.")
        # Graphs are built on the GPU (chg_batch_build: bit-for-bit the arrays of the host converter, see
        # tests/test_gpu_parity.py::test_device_graph_build_is_bit_exact); the host converter is only
        # consulted to phrase the reference's isolated-atom error / warning for the offending structure.
        conv, eng = self.graph_converter, self.engine
        flags = dict(site_energies=return_site_energies, atom_feas=return_atom_feas, crystal_feas=return_crystal_feas)

        def launch(chunk, prepared):
            """Build the chunk's graphs on the device and enqueue the sweep (asynchronous)."""
            batch = eng.build_prepared(prepared, conv.atom_graph_cutoff, conv.bond_graph_cutoff)
            try:
                if batch.packed.n_isolated and conv.on_isolated_atoms != "ignore":
                    for struct in chunk:          # raises ValueError / prints the warning like converter.py:161-174
                        conv(struct)
                eng.predict(batch, task)
            except BaseException:
                batch.free()
                raise
            return batch

        def collect(batch):
            try:
                return eng.download(batch, task, **flags), batch.packed.atom_off
            finally:
                batch.free()

        def run(chunk):
            res, atom_off = collect(launch(chunk, eng.prepare_structures(chunk)))
            return _split_results(res, atom_off, len(chunk))

        if len(structures) == 1:
            # one structure (every MD / relaxation step through the calculator): no chunk plan, no pipeline, and the batch arrays ARE the
            # structure's results -- handed out without the per-structure copies (~25 us of interpreter work per call, 3 % of an MD step)
            def run_one(chunk):
                # graph build and sweep enqueued by ONE native call (chg_batch_build_predict); an isolated atom is reported afterwards
                batch = eng.build_prepared(eng.prepare_structures(chunk), conv.atom_graph_cutoff, conv.bond_graph_cutoff, predict_task=task)
                try:
                    if batch.packed.n_isolated and conv.on_isolated_atoms != "ignore":
                        for struct in chunk:          # raises ValueError / prints the warning like converter.py:161-174
                            conv(struct)
                    res = eng.download(batch, task, **flags)
                finally:
                    batch.free()
                pred = {"e": res["e"][0]}
                for key in ("f", "m", "site_energies", "atom_fea"):
                    if key in res:
                        pred[key] = res[key]
                for key in ("s", "crystal_fea"):
                    if key in res:
                        pred[key] = res[key][0]
                return [pred]

            return _run_splitting(run_one, structures)[0]
        floor = self.min_atoms_per_batch if min_atoms_per_batch is None else int(min_atoms_per_batch)
        chunks = [structures[a:b] for a, b in _plan_chunks([len(s) for s in structures], batch_size, floor)]
        predictions = _run_pipelined(chunks, eng.prepare_structures, launch, collect, run)
        # like the reference, which hands the list to predict_graph: a one-element list gives a bare dict (model.py:665)
        return predictions[0] if len(structures) == 1 else predictions

    def predict_graph(self, graph, *, task: str = "efsm", return_site_energies: bool = False,
                      return_atom_feas: bool = False, return_crystal_feas: bool = False, batch_size: int = 16,
                      min_atoms_per_batch: int | None = None):
        """Predict from CrystalGraph(s).

        Returns a dict (single graph) or list of dicts with float32 numpy arrays:
        e () eV/atom, f (n,3) eV/A, s (3,3) GPa, m (n,) mu_B, and optionally
        site_energies (n,), atom_fea (n,64), crystal_fea (64,).
        ``batch_size`` / ``min_atoms_per_batch``: see ``predict_structure``.
        """
        if not (_is_graph(graph) or isinstance(graph, Sequence)):
            raise TypeError(f"{type(graph)=} must be CrystalGraph or list of CrystalGraphs")
        valid_tasks = VALID_TASKS
        if task not in valid_tasks:
            raise ValueError(f"Invalid {task=}. Must be one of {valid_tasks}.")
        graphs = [graph] if _is_graph(graph) else list(graph)
        eng = self.engine
        flags = dict(site_energies=return_site_energies, atom_feas=return_atom_feas, crystal_feas=return_crystal_feas)

        def launch(chunk, packed):  # noqa: ARG001
            batch = eng.upload(packed)
            try:
                eng.predict(batch, task)
            except BaseException:
                batch.free()
                raise
            return batch

        def collect(batch):
            try:
                return eng.download(batch, task, **flags), batch.packed.atom_off
            finally:
                batch.free()

        def run(chunk):
            res, atom_off = collect(launch(chunk, pack_batch(chunk)))
            return _split_results(res, atom_off, len(chunk))

        slots = itertools.cycle((0, 1))     # chunks are packed into two alternating page-locked blocks (one batch alive at a time)

        def pack_pinned(chunk):
            return pack_batch(chunk, alloc=eng.pinned_allocator(next(slots)))

        floor = self.min_atoms_per_batch if min_atoms_per_batch is None else int(min_atoms_per_batch)
        chunks = [graphs[a:b] for a, b in _plan_chunks([len(g.atomic_number) for g in graphs], batch_size, floor)]
        predictions = _run_pipelined(chunks, pack_pinned, launch, collect, run)   # the next chunk is packed during the sweep
        return predictions[0] if len(graphs) == 1 else predictions

    # ---- (de)serialisation (model.py:667-745) ----------------------------------------------------------
    def as_dict(self) -> dict:
        return {"state_dict": self._state_dict, "model_args": self.model_args}

    def todict(self) -> dict:
        return {"model_name": type(self).__name__, "model_args": self.model_args}

    @classmethod
    def from_dict(cls, dct: dict, **kwargs) -> "CHGNet":
        return cls(state_dict=dct["state_dict"], **dct["model_args"], **kwargs)

    @classmethod
    def from_file(cls, path: str, **kwargs) -> "CHGNet":
        """Read a reference checkpoint: ``torch.save({"model": {"state_dict", "model_args"}, ...})``."""
        from .safe_load import load_torch_file  # noqa: PLC0415  (restricted unpickler)

        state = load_torch_file(path)
        return cls.from_dict(state["model"], **kwargs)

    @classmethod
    def load(cls, *, model_name: str = "0.3.0", use_device: str | None = None, check_cuda_mem: bool = False,  # noqa: ARG003
             verbose: bool = True, checkpoint_dir: str | None = None) -> "CHGNet":
        """Load a pretrained checkpoint (same names as the reference, model.py:718-736).  The
        ``.pth.tar`` blobs are not shipped with this repository: point ``checkpoint_dir`` (or
        ``$CHGNET_CHECKPOINT_DIR``) at a reference ``chgnet/pretrained`` directory."""
        rel = {
            "0.3.0": "0.3.0/chgnet_0.3.0_e29f68s314m37.pth.tar",
            "0.2.0": "0.2.0/chgnet_0.2.0_e30f77s348m32.pth.tar",
            "r2scan": "r2scan/chgnet_r2scan_transfer_learning_e15f36s161m23.pth.tar",
        }.get(model_name)
        if rel is None:
            raise ValueError(f"Unknown {model_name=}")
        root = checkpoint_dir or os.environ.get("CHGNET_CHECKPOINT_DIR") or os.path.join(module_dir, "pretrained")
        path = os.path.join(root, rel)
        if not os.path.exists(path):
            raise FileNotFoundError(f"checkpoint {path} not found (set CHGNET_CHECKPOINT_DIR)")
        model = cls.from_file(path, mlp_out_bias=model_name == "0.2.0", version=model_name, use_device=use_device)
        if verbose:
            print(f"CHGNet will run on {model.device}")
        return model


def _run_splitting(run, chunk: list, refused: BaseException | None = None) -> list[dict]:
    """``run(chunk)``; when the device cannot hold the chunk's arena (EngineOutOfMemory) the chunk is halved
    and both halves are run the same way.  A single structure that does not fit raises.  ``refused``: the whole chunk
    has just been refused with this error (it is not tried again)."""
    from chgnet_amd.engine import EngineOutOfMemory  # noqa: PLC0415

    if refused is not None:
        if len(chunk) <= 1:
            raise refused
    else:
        try:
            return run(chunk)
        except EngineOutOfMemory:
            if len(chunk) <= 1:
                raise
    mid = len(chunk) // 2
    return _run_splitting(run, chunk[:mid]) + _run_splitting(run, chunk[mid:])


def _run_pipelined(chunks: list, prepare, launch, collect, run_sync) -> list[dict]:
    """Chunk loop of ``predict_structure`` with the host work hidden behind the device: while the device sweeps chunk i
    the host extracts the arrays of chunk i+1 (``prepare``), and chunk i's results are cut into per-structure
    dictionaries after chunk i+1 has been enqueued.  One batch is alive at a time, as in the plain loop.  A chunk whose
    arena does not fit (``EngineOutOfMemory`` from ``launch``) goes through ``_run_splitting(run_sync, chunk)``."""
    from chgnet_amd.engine import EngineOutOfMemory  # noqa: PLC0415

    out: list[dict] = []
    n = len(chunks)
    if n == 0:
        return out
    i, batch, prepared = 0, None, prepare(chunks[0])
    try:
        while i < n:
            if batch is None:
                try:
                    batch = launch(chunks[i], prepared)
                except EngineOutOfMemory as err:
                    out.extend(_run_splitting(run_sync, chunks[i], refused=err))
                    i += 1
                    prepared = prepare(chunks[i]) if i < n else None
                    continue
            nxt = prepare(chunks[i + 1]) if i + 1 < n else None        # overlaps the sweep of chunk i
            done, batch = batch, None
            res, atom_off = collect(done)                              # waits for chunk i, frees its batch
            if i + 1 < n:
                try:
                    batch = launch(chunks[i + 1], nxt)
                except EngineOutOfMemory:
                    batch = None                                       # retried (and split) at the top of the next round
            out.extend(_split_results(res, atom_off, len(chunks[i])))  # overlaps the sweep of chunk i+1
            i += 1
            prepared = nxt
    finally:
        if batch is not None:                                          # an exception with a sweep in flight
            batch.free()
    return out


def _plan_chunks(n_atoms: list[int], batch_size: int, min_atoms: int) -> list[tuple[int, int]]:
    """Split a list of structures into device batches.  ``batch_size`` (the reference's memory knob,
    model.py:640-650) is the minimum number of structures per chunk; a chunk keeps growing until it holds
    ``min_atoms`` atoms, because a batch below a few thousand atoms leaves most of the GPU idle (16 x 40
    atoms run at a third of the full-batch rate).  Per-structure results do not depend on the chunking
    (disjoint sub-graphs; energies are summed per structure in fixed order)."""
    if batch_size < 1:
        raise ValueError(f"{batch_size=} must be positive")
    chunks, start, n = [], 0, len(n_atoms)
    while start < n:
        stop = min(start + batch_size, n)
        atoms = sum(n_atoms[start:stop])
        while stop < n and atoms < min_atoms:
            atoms += n_atoms[stop]
            stop += 1
        chunks.append((start, stop))
        start = stop
    return chunks


def _split_results(res: dict, atom_off, n: int) -> list[dict]:
    """Batch-wide result arrays -> one dict per structure (reference model.py:651-663)."""
    out = []
    for i in range(n):
        sl = slice(atom_off[i], atom_off[i + 1])
        pred = {"e": res["e"][i]}
        for key in ("f", "m", "site_energies", "atom_fea"):
            if key in res:
                pred[key] = res[key][sl].copy()
        for key in ("s", "crystal_fea"):
            if key in res:
                pred[key] = res[key][i].copy()
        out.append(pred)
    return out


def _to_numpy(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v), dtype=np.float32)


def _parse_device(use_device) -> int:
    """``None``/"cuda"/"cuda:N"/N -> GPU ordinal (env CHGNET_DEVICE like common_utils.py:27; LOCAL_RANK
    for one-process-per-GPU launches).  CPU / MPS are not available in this engine."""
    use_device = use_device if use_device is not None else os.environ.get("CHGNET_DEVICE")
    if use_device is None:
        return int(os.environ.get("LOCAL_RANK", "0"))
    if isinstance(use_device, int):
        return use_device
    s = str(use_device)
    if s in ("cuda", "hip", "gpu"):
        return int(os.environ.get("LOCAL_RANK", "0"))
    if s.startswith("cuda:"):
        return int(s.split(":", 1)[1])
    raise ValueError(f"chgnet_amd runs on MI355X GPUs only; cannot use device {use_device!r}")
