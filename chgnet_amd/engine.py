"""Thin Python owner of a ``chg_engine`` / ``chg_batch`` (include/chgnet_hip.h)."""

from __future__ import annotations

import ctypes
import weakref

import numpy as np

from chgnet_amd import _lib
from chgnet_amd.graph.structure import atomic_numbers_of
from chgnet_amd.pack import D, PackedBatch, PackedWeights, _check_z, pack_batch


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_float_p)


def _ip(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_int_p)


class PreparedStructures:
    """Contiguous host arrays of a list of structures (``Engine.prepare_structures``)."""

    __slots__ = ("n_struct", "z", "atom_off", "frac", "lattice")

    def __init__(self, n_struct: int, z: np.ndarray, atom_off: np.ndarray, frac: np.ndarray, lattice: np.ndarray) -> None:
        self.n_struct, self.z, self.atom_off, self.frac, self.lattice = n_struct, z, atom_off, frac, lattice


class EngineOutOfMemory(RuntimeError):
    """CHG_ENOMEM: the batch arena could not be allocated (or exceeds ``Engine.set_memory_limit``).
    The engine stays usable; ``CHGNet.predict_*`` answer by splitting the chunk."""


class EngineRangeError(RuntimeError, FloatingPointError):
    """CHG_ERANGE: a WEIGHT of magnitude >= 65504 (the tile kernels carry their weights as f16 hi / lo images); raised by ``Engine()`` /
    ``update_weights``.  Activations beyond that range are not an error: ``download`` re-runs such a batch on the wide-range sweep."""


class DeviceBatch:
    """A packed batch resident in HBM together with all its workspace."""

    def __init__(self, engine: "Engine", packed: PackedBatch, handle=None) -> None:
        self.engine = engine
        self.packed = packed
        if handle is not None:          # built on the device (Engine.build_batch)
            self.handle = handle
            return
        self.handle = ctypes.c_void_p()
        self._host = self._host_struct(packed)
        engine._check(engine.lib.chg_batch_upload(engine.handle, ctypes.byref(self._host), ctypes.byref(self.handle)))

    @staticmethod
    def _host_struct(pb: PackedBatch) -> _lib.BatchHost:
        h = _lib.BatchHost()
        h.n_struct, h.n_atoms, h.n_directed = pb.n_struct, pb.n_atoms, pb.n_directed
        h.n_undirected, h.n_angles, h.n_bnodes = pb.n_undirected, pb.n_angles, pb.n_bnodes
        for name in ("frac", "lattice", "e_image"):
            setattr(h, name, _fp(pb.arrays[name]))
        for name in ("z", "atom_owner", "atom_off", "e_center", "e_nbr", "e_d2u", "e_owner", "e_rev", "p_center", "p_nbr", "u_u2d", "u_bnode",
                     "bn_und", "a_ctr", "a_b1c", "a_b2c", "a_d1", "a_d2"):
            setattr(h, name, _ip(pb.arrays[name]))
        return h

    @property
    def device_bytes(self) -> int:
        return int(self.engine.lib.chg_batch_device_bytes(self.handle))

    def update_geometry(self, frac=None, lattice=None) -> None:
        f = np.ascontiguousarray(frac, dtype=np.float32) if frac is not None else None
        l = np.ascontiguousarray(lattice, dtype=np.float32) if lattice is not None else None
        self.engine._check(self.engine.lib.chg_batch_update_geometry(
            self.engine.handle, self.handle, _fp(f) if f is not None else None, _fp(l) if l is not None else None))

    def free(self) -> None:
        if self.handle:
            self.engine.lib.chg_batch_free(self.engine.handle, self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class Engine:
    """One engine per GPU; not thread-safe (serialise calls per engine) -- except ``upload``, which may run on a second thread
    while the engine computes (its copies use their own stream; TrainStep.run_epoch uploads the next batch that way)."""

    def __init__(self, weights: PackedWeights, device: int = 0) -> None:
        self.lib = _lib.load()
        self.weights = weights
        self.device = device
        desc = _lib.ModelDesc(weights.n_conv, weights.cutoff_coeff, int(weights.is_intensive), int(weights.has_composition),
                              weights.atom_graph_cutoff, weights.bond_graph_cutoff, weights.blob.size,
                              int(getattr(weights, "n_mlp_hidden", 3)), int(getattr(weights, "mlp_out_bias", False)))
        self.handle = ctypes.c_void_p()
        blob = np.ascontiguousarray(weights.blob, dtype=np.float32)
        status = self.lib.chg_engine_create(ctypes.byref(desc), _fp(blob), int(device), ctypes.byref(self.handle))
        if status != 0:
            msg = self.lib.chg_last_error(self.handle).decode() if self.handle else ""
            if self.handle:
                self.lib.chg_engine_destroy(self.handle)
                self.handle = ctypes.c_void_p()
            if status == -6:
                raise EngineRangeError(f"chg_engine_create: {msg}")
            raise RuntimeError(f"chg_engine_create failed with status {status}: {msg or 'no usable gfx950 device'}")

    def _check(self, status: int) -> None:
        if status != 0:
            msg = f"chgnet_hip error {status}: {self.lib.chg_last_error(self.handle).decode()}"
            if status == -3:
                raise EngineOutOfMemory(msg)
            if status == -6:
                raise EngineRangeError(msg)
            raise RuntimeError(msg)

    def build_stats(self) -> tuple[int, int]:
        """(single-pass graph builds, capacity overflows that fell back to the exact pass) of ``build_batch``."""
        a, b = ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.chg_engine_build_stats(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def set_graph_search(self, search: str = "auto", cell_min_atoms: int = 0) -> None:
        """Neighbour search of ``build_batch``: "auto" (cell list for structures of at least ``cell_min_atoms`` atoms,
        default 2048), "all_pairs" or "cells".  The graph does not depend on the choice."""
        self._check(self.lib.chg_engine_set_graph_search(self.handle, {"auto": 0, "all_pairs": 1, "cells": 2}[search], int(cell_min_atoms)))

    def cell_stats(self) -> tuple[int, int]:
        """(builds that used the cell list, builds repeated with all pairs because a centre had more than 1024 rows)."""
        a, b = ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.chg_engine_cell_stats(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def set_memory_limit(self, n_bytes: int) -> None:
        """Refuse batch arenas above ``n_bytes`` with ``EngineOutOfMemory`` (0 = no limit)."""
        self._check(self.lib.chg_engine_set_memory_limit(self.handle, int(n_bytes)))

    def memory_info(self) -> tuple[int, int]:
        """(free, total) device bytes; arenas pooled for reuse count as free."""
        free, total = ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.chg_engine_memory_info(self.handle, ctypes.byref(free), ctypes.byref(total)))
        return int(free.value), int(total.value)

    def bytes_required(self, n_struct: int, n_atoms: int, n_directed: int, n_angles: int, n_bnodes: int) -> int:
        """Exact arena size of a batch with these counts (chg_batch_bytes_required)."""
        return int(self.lib.chg_batch_bytes_required(self.weights.n_conv, n_struct, n_atoms, n_directed, n_angles, n_bnodes))

    def close(self) -> None:
        self.__dict__.get("_pinned", {}).clear()      # page-locked packing blocks: freed by their finalizers once no PackedBatch array views them
        if self.handle:
            self.lib.chg_engine_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ------------------------------------------------------------------
    def pinned_allocator(self, slot: int = 0):
        """``alloc`` argument for ``pack_batch``: the packed arrays of a batch carved out of ONE page-locked block per ``slot`` (grown when
        a batch needs more; a block is returned to the system when the pool and every array carved out of it are gone).  ``upload`` from such arrays is asynchronous DMA at the link rate (1024 x 40 atoms:
        ~250 MB in 5 ms instead of 22 ms from pageable memory).  A slot's block is REUSED by the next ``pack_batch`` with the same
        slot: alternate two slots when one batch is packed while the previous one is still in use (``TrainStep.run_epoch`` does)."""
        pools = self.__dict__.setdefault("_pinned", {})
        host_free = self.lib.chg_host_free

        def alloc(spec: dict) -> dict:
            offs, pos = {}, 0
            for name, (shape, dtype) in spec.items():
                offs[name] = pos
                pos += (int(np.prod(shape)) * np.dtype(dtype).itemsize + 255) & ~255
            need = max(pos, 256)
            raw, have = pools.get(slot, (None, 0))
            if have < need:
                # A block that proved too small is dropped from the pool, not freed here: earlier PackedBatch objects may still view it.
                # Every block is ONE ctypes buffer object that its arrays keep alive (numpy holds the exporter); a finalizer returns the
                # page-locked memory when the last of them -- and the pool -- has let go.  New blocks get 50 % headroom, so a slowly
                # growing batch size retires a block every few growths instead of every step (round 5 kept all retired blocks until
                # Engine.close(): up to ~8x the final size in unswappable memory).
                pools.pop(slot, None)
                out = ctypes.c_void_p()
                want = need + need // 2
                if self.lib.chg_host_alloc(want, ctypes.byref(out)) != 0 or not out.value:
                    return {k: np.empty(shape, dtype) for k, (shape, dtype) in spec.items()}    # no page-locked memory: pageable arrays
                raw, have = (ctypes.c_char * want).from_address(out.value), want
                weakref.finalize(raw, host_free, ctypes.c_void_p(out.value))
                pools[slot] = (raw, have)
            arrays = {}
            for name, (shape, dtype) in spec.items():
                n = int(np.prod(shape))
                arrays[name] = np.frombuffer(raw, dtype=dtype, count=n, offset=offs[name]).reshape(shape)
            return arrays

        return alloc

    def upload(self, graphs_or_packed) -> DeviceBatch:
        packed = graphs_or_packed if isinstance(graphs_or_packed, PackedBatch) else pack_batch(graphs_or_packed)
        return DeviceBatch(self, packed)

    def prepare_structures(self, structures) -> "PreparedStructures":
        """Host side of ``build_batch``: structures -> contiguous arrays (atomic numbers checked).  Pure CPU work, so a
        caller can prepare the next chunk while the device is busy with the current one."""
        structures = list(structures)
        n_at = np.array([len(s) for s in structures], dtype=np.int64)
        a_off = np.concatenate([[0], np.cumsum(n_at)]).astype(np.int32)
        z = np.ascontiguousarray(np.concatenate([atomic_numbers_of(s) for s in structures]), dtype=np.int32)
        _check_z(z)
        frac = np.ascontiguousarray(np.concatenate([np.asarray(s.frac_coords, dtype=np.float64).reshape(-1, 3) for s in structures]))
        lattice = np.ascontiguousarray(np.stack([np.asarray(s.lattice.matrix, dtype=np.float64) for s in structures]))
        return PreparedStructures(len(structures), z, a_off, frac, lattice)

    def build_prepared(self, prep: "PreparedStructures", atom_graph_cutoff: float = 6.0, bond_graph_cutoff: float = 3.0,
                       numerical_tol: float = 1e-8, predict_task: str | None = None) -> DeviceBatch:
        """Device side of ``build_batch`` (chg_batch_build).  ``predict_task``: the prediction is enqueued by the same native call
        (chg_batch_build_predict) -- a single-structure caller saves the trip back to Python between the two."""
        host = _lib.StructsHost(prep.n_struct, int(prep.atom_off[-1]), _ip(prep.z), prep.frac.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                prep.lattice.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _ip(prep.atom_off))
        handle = ctypes.c_void_p()
        counts = np.zeros(6, dtype=np.int32)
        if predict_task is not None:
            self._check(self.lib.chg_batch_build_predict(self.handle, ctypes.byref(host), float(atom_graph_cutoff), float(bond_graph_cutoff),
                                                         float(numerical_tol), _lib.task_mask(predict_task), ctypes.byref(handle), _ip(counts)))
        else:
            self._check(self.lib.chg_batch_build(self.handle, ctypes.byref(host), float(atom_graph_cutoff), float(bond_graph_cutoff),
                                                 float(numerical_tol), ctypes.byref(handle), _ip(counts)))
        packed = PackedBatch(prep.n_struct, int(prep.atom_off[-1]), int(counts[0]), int(counts[1]), int(counts[2]), int(counts[3]),
                             {"z": prep.z, "atom_off": prep.atom_off, "frac": prep.frac.astype(np.float32),
                              "lattice": prep.lattice.astype(np.float32)})
        packed.n_isolated = int(counts[4])
        return DeviceBatch(self, packed, handle=handle)

    def build_batch(self, structures, atom_graph_cutoff: float = 6.0, bond_graph_cutoff: float = 3.0,
                    numerical_tol: float = 1e-8) -> DeviceBatch:
        """Structures -> device-resident batch with the graph built ON the GPU (chg_batch_build):
        same arrays, bit for bit, as ``CrystalGraphConverter`` + ``pack_batch`` + ``upload``.
        ``batch.packed.n_isolated`` holds the number of atoms without any neighbour."""
        return self.build_prepared(self.prepare_structures(structures), atom_graph_cutoff, bond_graph_cutoff, numerical_tol)

    def debug_fetch_i32(self, batch: DeviceBatch, name: str, n: int) -> np.ndarray:
        dst = np.empty(max(int(n), 1), np.int32)
        got = ctypes.c_int64()
        self._check(self.lib.chg_debug_fetch_i32(self.handle, batch.handle, name.encode(), _ip(dst), int(n), ctypes.byref(got)))
        if got.value != n:
            raise RuntimeError(f"debug_fetch_i32({name}): expected {n} ints, device buffer has {got.value}")
        return dst[:n]

    def predict(self, batch: DeviceBatch, task: str = "efsm") -> None:
        """Enqueue E(+F,S,M) for the batch on the engine stream (asynchronous)."""
        self._check(self.lib.chg_predict(self.handle, batch.handle, _lib.task_mask(task)))

    def backward(self, batch: DeviceBatch, e_grad=None, m_grad=None, f_grad=None, s_grad=None, comm=None) -> np.ndarray:
        """Gradient blob (weight-blob layout) of ``sum e_grad*e + sum m_grad*m + sum f_grad.f + sum s_grad:s`` after
        ``predict`` on ``batch`` (chg_backward); ``pack.unpack_weight_grads`` turns it into state_dict names.
        ``f_grad`` [N,3] / ``s_grad`` [B,3,3] switch to the second-order sweep.  ``e_grad=None`` is "ones" only when it is the
        ONLY term (``backward(batch)`` = gradient of the summed energies); next to another cotangent it means "no energy
        term".  ``comm`` (an ``RcclComm``): the blob is summed over the ranks in HBM on the engine's stream before it comes to
        the host (chg_backward_allreduce)."""
        pb = batch.packed
        grad = np.zeros(self.weights.blob.size, np.float32)
        if e_grad is None and any(x is not None for x in (m_grad, f_grad, s_grad)):
            e_grad = np.zeros(pb.n_struct, np.float32)      # the C-ABI reads a null energy cotangent as ones

        def arg(x, n, what):
            if x is None:
                return None
            a = np.ascontiguousarray(x, np.float32).reshape(-1)
            if a.size != n:
                raise ValueError(f"{what} has {a.size} entries, the batch needs {n}")
            return a

        cot = arg(e_grad, pb.n_struct, "e_grad")
        mcot = arg(m_grad, pb.n_atoms, "m_grad")
        fcot = arg(f_grad, 3 * pb.n_atoms, "f_grad")
        scot = arg(s_grad, 9 * pb.n_struct, "s_grad")
        ptr = lambda a: _fp(a) if a is not None else None  # noqa: E731
        if comm is not None and getattr(comm, "world", 1) > 1:
            self._check(self.lib.chg_backward_allreduce(self.handle, batch.handle, ptr(cot), ptr(mcot), ptr(fcot), ptr(scot), comm.handle, _fp(grad)))
        else:
            self._check(self.lib.chg_backward(self.handle, batch.handle, ptr(cot), ptr(mcot), ptr(fcot), ptr(scot), _fp(grad)))
        return grad

    def all_gather_energy(self, batch: DeviceBatch, comm, width: int) -> np.ndarray:
        """[nranks, width] table of the per-structure energies of every rank's batch (zero-padded to ``width``), gathered
        from HBM on the engine's stream (chg_batch_all_gather_energy).  After ``predict``."""
        table = np.empty(comm.world * int(width), np.float32)
        self._check(self.lib.chg_batch_all_gather_energy(self.handle, batch.handle, comm.handle, int(width), _fp(table)))
        return table.reshape(comm.world, int(width))

    def update_weights(self, weights: PackedWeights) -> None:
        """Replace the parameter values (optimizer step); same architecture / blob layout."""
        if weights.blob.size != self.weights.blob.size:
            raise ValueError("update_weights: blob length differs from the engine's")
        blob = np.ascontiguousarray(weights.blob, dtype=np.float32)
        self._check(self.lib.chg_engine_update_weights(self.handle, _fp(blob)))
        self.weights = weights

    def synchronize(self) -> None:
        self._check(self.lib.chg_synchronize(self.handle))

    def download(self, batch: DeviceBatch, task: str = "efsm", *, site_energies=False, atom_feas=False,
                 crystal_feas=False) -> dict:
        """Copy results to host; returns batch-wide arrays (split per structure by the caller)."""
        pb = batch.packed
        out = {"e": np.empty(pb.n_struct, np.float32)}
        o = _lib.OutHost()
        o.energy = _fp(out["e"])
        if "f" in task:
            out["f"] = np.empty((pb.n_atoms, 3), np.float32)
            o.force = _fp(out["f"])
        if "s" in task:
            out["s"] = np.empty((pb.n_struct, 3, 3), np.float32)
            o.stress = _fp(out["s"])
        if "m" in task:
            out["m"] = np.empty(pb.n_atoms, np.float32)
            o.magmom = _fp(out["m"])
        if site_energies:
            out["site_energies"] = np.empty(pb.n_atoms, np.float32)
            o.site_energy = _fp(out["site_energies"])
        if atom_feas:
            out["atom_fea"] = np.empty((pb.n_atoms, D), np.float32)
            o.atom_fea = _fp(out["atom_fea"])
        if crystal_feas:
            out["crystal_fea"] = np.empty((pb.n_struct, D), np.float32)
            o.crystal_fea = _fp(out["crystal_fea"])
        self._check(self.lib.chg_batch_download(self.handle, batch.handle, ctypes.byref(o)))
        return out

    # ---- timing / profiling on the engine's own stream ------------------------------------------------
    def timer_start(self) -> None:
        self._check(self.lib.chg_timer_start(self.handle))

    def timer_stop_ms(self) -> float:
        ms = ctypes.c_float()
        self._check(self.lib.chg_timer_stop_ms(self.handle, ctypes.byref(ms)))
        return float(ms.value)

    def stream_copy_gbs(self, n_bytes: int = 1 << 30, iters: int = 10) -> float:
        """Measured HBM copy rate in GB/s (read + write counted), chg_stream_copy."""
        ms = ctypes.c_float()
        self._check(self.lib.chg_stream_copy(self.handle, int(n_bytes), int(iters), ctypes.byref(ms)))
        return 2.0 * n_bytes / (ms.value * 1e-3) / 1e9

    def profile(self, on: bool) -> None:
        self._check(self.lib.chg_profile_enable(self.handle, int(on)))

    def profile_reset(self) -> None:
        self._check(self.lib.chg_profile_reset(self.handle))

    def profile_read(self) -> dict:
        out = {}
        buf = ctypes.create_string_buffer(64)
        for i in range(self.lib.chg_profile_count(self.handle)):
            n, ms = ctypes.c_int64(), ctypes.c_double()
            self._check(self.lib.chg_profile_read(self.handle, i, buf, 64, ctypes.byref(n), ctypes.byref(ms)))
            out[buf.value.decode()] = (int(n.value), float(ms.value))
        return out

    def debug_fetch(self, batch: DeviceBatch, name: str, shape) -> np.ndarray:
        n = int(np.prod(shape))
        dst = np.empty(max(n, 1), np.float32)
        got = ctypes.c_int64()
        self._check(self.lib.chg_debug_fetch(self.handle, batch.handle, name.encode(), _fp(dst), n, ctypes.byref(got)))
        if got.value != n:
            raise RuntimeError(f"debug_fetch({name}): expected {n} floats, device buffer has {got.value}")
        return dst[:n].reshape(shape)

    def test_split_gemm(self, x: np.ndarray, w: np.ndarray, mode: int) -> np.ndarray:
        """The split-precision contraction of the tile kernels on its own (csrc/mfma_split.h): ``w`` [f, 64], f in {64, 128};
        mode 0 / 2: ``x`` [rows, 64] -> ``x @ w.T`` [rows, f] (split images / row-major image); mode 1 / 3: ``x`` [rows, f] ->
        ``x @ w`` [rows, 64] (the adjoint forms, rows scaled by a power of two)."""
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        f = w.shape[0]
        assert w.shape[1] == 64 and x.shape[1] == (f if mode & 1 else 64)
        y = np.empty((x.shape[0], 64 if mode & 1 else f), np.float32)
        self._check(self.lib.chg_test_split_gemm(self.handle, _fp(x), _fp(w), _fp(y), x.shape[0], f, int(mode)))
        return y

    def test_rows_gemm(self, x: np.ndarray, wt: np.ndarray, bias: np.ndarray | None) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        wt = np.ascontiguousarray(wt, np.float32)
        rows, k = x.shape
        nout = wt.shape[0]
        y = np.empty((rows, nout), np.float32)
        b = np.ascontiguousarray(bias, np.float32) if bias is not None else None
        self._check(self.lib.chg_test_rows_gemm(self.handle, _fp(x), _fp(wt), _fp(b) if b is not None else None, _fp(y), rows, k, nout))
        return y
