"""Host-side packing: list[CrystalGraph] -> one SoA batch, state_dict -> one fp32 blob.

The packed batch is what replaces the reference's per-graph Python loop
``BatchedGraph.from_graphs`` (model.py:820-899): all index offsetting happens here, once,
in numpy; geometry and basis functions are evaluated on the device.

Extra index arrays (not in the reference) that the kernels use:
  * ``bn_und`` / ``u_bnode`` -- compact numbering of the undirected bonds that are nodes of
    the bond graph (appear in ``bond_graph[:,1]`` or ``[:,3]``); only those bonds are ever
    updated by BondConv (layers.py:252-258), so per-layer bond features and the per-bond
    partial products live in arrays of ``Eb`` rows (~15 % of ``Eu``);
  * ``p_center`` / ``p_nbr`` -- the directed edges re-listed in bond-pair order (rows 2k, 2k+1 are the
    two directions of undirected bond k) for the AtomConv adjoint, where both directions of a bond
    must sit in one tile so that their gradients can be summed without atomics;
  * global (batch-wide) directed / undirected indices with per-structure offsets applied.
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass, field

import numpy as np

from chgnet_amd.graph.crystalgraph import CrystalGraph

D = 64          # atom = bond = angle feature width of every released CHGNet
NUM_RADIAL = 31     # basis sizes of the KERNELS (0.3.0 / r2scan); smaller expansions (0.2.0: 9 / 9) are zero-padded to them, see pad_bases
NUM_ANGULAR = 31
N_ELEM = 94


@dataclass
class PackedBatch:
    n_struct: int
    n_atoms: int
    n_directed: int
    n_undirected: int
    n_angles: int
    n_bnodes: int
    arrays: dict = field(default_factory=dict)   # name -> contiguous numpy array
    n_isolated: int = 0

    def __getattr__(self, name):
        try:
            return self.__dict__["arrays"][name]
        except KeyError as exc:
            raise AttributeError(name) from exc


class _GraphView(ctypes.Structure):      # include/chgnet_graph.h: chg_graph_view
    _fields_ = [("n_atoms", ctypes.c_int32), ("n_directed", ctypes.c_int32), ("n_undirected", ctypes.c_int32), ("n_angles", ctypes.c_int32),
                ("atomic_number", ctypes.c_void_p), ("frac", ctypes.c_void_p), ("lattice", ctypes.c_void_p), ("atom_graph", ctypes.c_void_p),
                ("image", ctypes.c_void_p), ("directed2undirected", ctypes.c_void_p), ("undirected2directed", ctypes.c_void_p),
                ("bond_graph", ctypes.c_void_p)]


_PACKED_FIELDS = ("z", "atom_owner", "atom_off", "edge_off", "und_off", "ang_off", "frac", "lattice", "e_image", "e_center", "e_nbr", "e_d2u",
                  "e_owner", "e_rev", "p_center", "p_nbr", "u_u2d", "u_bnode", "bn_und", "a_ctr", "a_b1", "a_d1", "a_b2", "a_d2", "a_b1c", "a_b2c")


class _PackedOut(ctypes.Structure):      # include/chgnet_graph.h: chg_packed_out
    _fields_ = [(name, ctypes.c_void_p) for name in _PACKED_FIELDS]


def pack_batch(graphs, alloc=None) -> PackedBatch:
    """Concatenate graphs into one disjoint-union batch with global indices: one pass in native code
    (``chg_pack_batch``, csrc/host_graph.cpp; ~30x the numpy formulation kept below as ``pack_batch_numpy``).
    ``alloc(sizes: {name: (shape, dtype)}) -> {name: array}``: where the output arrays live -- ``Engine.pinned_allocator`` hands out
    page-locked memory, from which ``Engine.upload`` copies at the link rate."""
    from chgnet_amd.graph.converter import graph_lib  # noqa: PLC0415

    graphs = [CrystalGraph.from_reference(g) for g in graphs]
    B = len(graphs)
    lib = graph_lib()
    views = (_GraphView * max(B, 1))()
    N = Ed = Eu = A = 0
    for i, g in enumerate(graphs):
        v = views[i]
        v.n_atoms, v.n_directed, v.n_undirected, v.n_angles = len(g.atomic_number), len(g.atom_graph), len(g.undirected2directed), len(g.bond_graph)
        v.atomic_number, v.frac, v.lattice = g.atomic_number.ctypes.data, g.atom_frac_coord.ctypes.data, g.lattice.ctypes.data
        v.atom_graph, v.image = g.atom_graph.ctypes.data, g.neighbor_image.ctypes.data
        v.directed2undirected, v.undirected2directed = g.directed2undirected.ctypes.data, g.undirected2directed.ctypes.data
        v.bond_graph = g.bond_graph.ctypes.data
        N += v.n_atoms
        Ed += v.n_directed
        Eu += v.n_undirected
        A += v.n_angles
        if len(g.directed2undirected) != v.n_directed or len(g.neighbor_image) != v.n_directed or len(g.atom_frac_coord) != v.n_atoms:
            raise ValueError(f"graph {i}: array lengths are inconsistent")
    if max(N, Ed, Eu, A) >= 2**31 - 1:
        raise ValueError("batch too large for int32 indexing")
    sizes = {"z": N, "atom_owner": N, "atom_off": B + 1, "edge_off": B + 1, "und_off": B + 1, "ang_off": B + 1, "frac": (N, 3),
             "lattice": (B, 3, 3), "e_image": (Ed, 3), "e_center": Ed, "e_nbr": Ed, "e_d2u": Ed, "e_owner": Ed, "e_rev": Ed,
             "p_center": Ed, "p_nbr": Ed, "u_u2d": Eu, "u_bnode": Eu, "bn_und": Eu, "a_ctr": A, "a_b1": A, "a_d1": A, "a_b2": A,
             "a_d2": A, "a_b1c": A, "a_b2c": A}
    spec = {k: (sizes[k] if isinstance(sizes[k], tuple) else (sizes[k],), np.float32 if k in ("frac", "lattice", "e_image") else np.int32)
            for k in _PACKED_FIELDS}
    arr = alloc(spec) if alloc is not None else {k: np.empty(shape, dtype) for k, (shape, dtype) in spec.items()}
    out = _PackedOut(*[arr[k].ctypes.data for k in _PACKED_FIELDS])
    n_bn, bad = ctypes.c_int32(), ctypes.c_int32(-1)
    status = lib.chg_pack_batch(B, views, ctypes.byref(out), ctypes.byref(n_bn), ctypes.byref(bad))
    if status != 0:
        pack_batch_numpy(graphs)          # phrases the IndexError / ValueError for the offending array
        raise ValueError(f"graph {bad.value}: {lib.chg_graph_strerror(status).decode()}")
    arr["bn_und"] = arr["bn_und"][:n_bn.value] if alloc is not None else arr["bn_und"][:n_bn.value].copy()
    if alloc is not None:       # the offset tables outlive the upload (results are split by them): keep them out of the reusable buffer
        for k in ("atom_off", "edge_off", "und_off", "ang_off"):
            arr[k] = arr[k].copy()
    return PackedBatch(B, N, Ed, Eu, A, int(n_bn.value), arr)


def pack_batch_numpy(graphs) -> PackedBatch:
    """The same batch assembled with numpy (reference implementation of ``pack_batch`` for the tests, and the
    source of the detailed error messages)."""
    graphs = [CrystalGraph.from_reference(g) for g in graphs]
    B = len(graphs)
    n_at = np.array([len(g.atomic_number) for g in graphs], dtype=np.int64)
    n_ed = np.array([len(g.atom_graph) for g in graphs], dtype=np.int64)
    n_un = np.array([len(g.undirected2directed) for g in graphs], dtype=np.int64)
    n_an = np.array([len(g.bond_graph) for g in graphs], dtype=np.int64)
    a_off = np.concatenate([[0], np.cumsum(n_at)])
    e_off = np.concatenate([[0], np.cumsum(n_ed)])
    u_off = np.concatenate([[0], np.cumsum(n_un)])
    g_off = np.concatenate([[0], np.cumsum(n_an)])
    N, Ed, Eu, A = int(a_off[-1]), int(e_off[-1]), int(u_off[-1]), int(g_off[-1])
    if max(N, Ed, Eu, A) >= 2**31 - 1:
        raise ValueError("batch too large for int32 indexing")

    def cat(parts, dtype, shape_tail=()):
        parts = [p for p in parts if len(p)]
        if not parts:
            return np.zeros((0, *shape_tail), dtype=dtype)
        return np.ascontiguousarray(np.concatenate(parts), dtype=dtype)

    # one concatenation per field, offsets added afterwards in one vectorised pass (no per-graph temporaries)
    e_owner = np.repeat(np.arange(B, dtype=np.int32), n_ed)
    e_aoff = np.repeat(a_off[:-1], n_ed)
    a_aoff, a_uoff, a_eoff = np.repeat(a_off[:-1], n_an), np.repeat(u_off[:-1], n_an), np.repeat(e_off[:-1], n_an)
    ag = cat([g.atom_graph for g in graphs], np.int64, (2,)).reshape(-1, 2)
    bg = cat([g.bond_graph for g in graphs], np.int64, (5,)).reshape(-1, 5)
    arr = {}
    arr["z"] = cat([g.atomic_number for g in graphs], np.int32)
    arr["frac"] = cat([g.atom_frac_coord for g in graphs], np.float32, (3,))
    arr["lattice"] = np.ascontiguousarray(np.stack([g.lattice for g in graphs]) if B else np.zeros((0, 3, 3)), dtype=np.float32)
    arr["atom_owner"] = np.repeat(np.arange(B, dtype=np.int32), n_at)
    arr["e_center"] = (ag[:, 0] + e_aoff).astype(np.int32)
    arr["e_nbr"] = (ag[:, 1] + e_aoff).astype(np.int32)
    arr["e_image"] = cat([g.neighbor_image for g in graphs], np.float32, (3,))
    arr["e_d2u"] = (cat([g.directed2undirected for g in graphs], np.int64) + np.repeat(u_off[:-1], n_ed)).astype(np.int32)
    arr["e_owner"] = e_owner
    arr["u_u2d"] = (cat([g.undirected2directed for g in graphs], np.int64) + np.repeat(e_off[:-1], n_un)).astype(np.int32)
    arr["a_ctr"] = (bg[:, 0] + a_aoff).astype(np.int32)
    arr["a_b1"] = (bg[:, 1] + a_uoff).astype(np.int32)
    arr["a_d1"] = (bg[:, 2] + a_eoff).astype(np.int32)
    arr["a_b2"] = (bg[:, 3] + a_uoff).astype(np.int32)
    arr["a_d2"] = (bg[:, 4] + a_eoff).astype(np.int32)

    # Range checks before anything reaches the device: graphs may come from user code or from .pt cache
    # files (CrystalGraph.from_file), and an out-of-range index would be an out-of-bounds device read or
    # atomic write.  The reference fails the same inputs with IndexError (nn.Embedding(94), index_select).
    _check_z(arr["z"])
    _check_index(arr["e_center"], e_off, a_off, "atom_graph[:, 0]")
    _check_index(arr["e_nbr"], e_off, a_off, "atom_graph[:, 1]")
    _check_index(arr["e_d2u"], e_off, u_off, "directed2undirected")
    _check_index(arr["u_u2d"], u_off, e_off, "undirected2directed")
    _check_index(arr["a_ctr"], g_off, a_off, "bond_graph[:, 0]")
    _check_index(arr["a_b1"], g_off, u_off, "bond_graph[:, 1]")
    _check_index(arr["a_d1"], g_off, e_off, "bond_graph[:, 2]")
    _check_index(arr["a_b2"], g_off, u_off, "bond_graph[:, 3]")
    _check_index(arr["a_d2"], g_off, e_off, "bond_graph[:, 4]")
    if Eu and not (np.bincount(arr["e_d2u"], minlength=Eu) == 2).all():
        raise ValueError("directed2undirected must map exactly two directed edges onto every undirected edge")
    if Eu and not (arr["e_d2u"][arr["u_u2d"]] == np.arange(Eu)).all():
        raise ValueError("undirected2directed[k] must be a directed edge of undirected edge k")

    # compact numbering of bond-graph nodes (monotone in the undirected index, so the
    # angle rows stay sorted by their owning bond: graph.py:283-327 emits them that way)
    is_node = np.zeros(Eu, dtype=bool)
    is_node[arr["a_b1"]] = True
    is_node[arr["a_b2"]] = True
    bn_und = np.flatnonzero(is_node).astype(np.int32)
    u_bnode = np.full(Eu, -1, dtype=np.int32)
    u_bnode[bn_und] = np.arange(len(bn_und), dtype=np.int32)
    arr["bn_und"] = bn_und
    arr["u_bnode"] = u_bnode
    arr["a_b1c"] = u_bnode[arr["a_b1"]].astype(np.int32) if A else np.zeros(0, np.int32)
    arr["a_b2c"] = u_bnode[arr["a_b2"]].astype(np.int32) if A else np.zeros(0, np.int32)

    # pair-ordered edge list for the AtomConv adjoint: rows 2k, 2k+1 = the two directions of bond k
    # (first = undirected2directed[k], whose centre is nondecreasing in k)
    first = arr["u_u2d"].astype(np.int64)
    # the other directed edge of bond k: (sum of its two edge indices) - first   (exact in float64 below 2^53)
    second = (np.bincount(arr["e_d2u"], weights=np.arange(Ed, dtype=np.float64), minlength=Eu).astype(np.int64) - first) if Eu \
        else np.zeros(0, np.int64)
    rows = np.stack([first, second], axis=1).reshape(-1) if Eu else np.zeros(0, np.int64)
    e_rev = np.empty(Ed, dtype=np.int32)                      # the opposite direction of every directed edge
    e_rev[first] = second
    e_rev[second] = first
    arr["e_rev"] = e_rev
    arr["p_center"] = np.ascontiguousarray(arr["e_center"][rows], dtype=np.int32)
    arr["p_nbr"] = np.ascontiguousarray(arr["e_nbr"][rows], dtype=np.int32)

    arr["atom_off"] = a_off.astype(np.int32)
    arr["edge_off"] = e_off.astype(np.int32)
    arr["und_off"] = u_off.astype(np.int32)
    arr["ang_off"] = g_off.astype(np.int32)
    return PackedBatch(B, N, Ed, Eu, A, int(len(bn_und)), arr)


def _check_z(z) -> None:
    """1 <= Z <= 94 (rows of the atom embedding / AtomRef tables, model.py:432-434)."""
    z = np.asarray(z)
    if z.size and (int(z.min()) < 1 or int(z.max()) > N_ELEM):
        bad = z[(z < 1) | (z > N_ELEM)][0]
        raise IndexError(f"atomic number {int(bad)} is out of range: the atom embedding has {N_ELEM} rows (Z = 1..{N_ELEM})")


def _check_index(idx, seg_off, target_off, what: str) -> None:
    """Rows ``seg_off[b]:seg_off[b+1]`` of ``idx`` belong to structure b and must index into its own
    ``[target_off[b], target_off[b+1])`` (two segmented min / max passes, no per-row temporaries)."""
    if len(idx) == 0:
        return
    has_rows = seg_off[1:] > seg_off[:-1]
    starts = seg_off[:-1][has_rows]
    lo = np.minimum.reduceat(idx, starts)
    hi = np.maximum.reduceat(idx, starts)
    if (lo < target_off[:-1][has_rows]).any() or (hi >= target_off[1:][has_rows]).any():
        raise IndexError(f"{what} holds an index outside its structure")


# ---------------------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------------------
SUPPORTED_MODEL_ARGS = {
    "atom_fea_dim": 64, "bond_fea_dim": 64, "angle_fea_dim": 64,
    "atom_conv_hidden_dim": 64, "bond_conv_hidden_dim": 64, "angle_layer_hidden_dim": 0,
    "update_bond": True, "update_angle": True, "mlp_first": True, "non_linearity": "silu",
    "gMLP_norm": "layer", "readout_norm": "layer", "conv_dropout": 0, "mlp_dropout": 0,
}


def check_model_args(model_args: dict) -> None:
    """The engine implements the architecture family of the released checkpoints -- 0.3.0 / r2scan
    (pretrained/0.3.0/README.md: 31 radial / 31 angular functions, a 64-64-64 energy head, cutoffs 6 / 3,
    envelope exponent 8) and 0.2.0 (pretrained/0.2.0/README.md:12-37: 9 / 9, a 64-64 head, cutoffs 5 / 3,
    exponent 5, ``mlp_out`` biases) -- and what lies between them: 1..31 radial functions, an odd number of
    1..31 angular functions, two or three hidden layers of 64.  Anything else is rejected loudly."""
    for key, want in SUPPORTED_MODEL_ARGS.items():
        got = model_args.get(key, want)
        if isinstance(want, bool) or isinstance(want, str) or want is None:
            ok = got == want
        else:
            ok = (list(got) == [want] if isinstance(got, (list, tuple)) else got == want)
        if not ok:
            raise NotImplementedError(f"chgnet_amd engine does not implement {key}={got!r} (supports {want!r})")
    hid = model_args.get("mlp_hidden_dims", (64, 64, 64))
    if isinstance(hid, int) or list(hid) not in ([64, 64, 64], [64, 64]):
        raise NotImplementedError(f"mlp_hidden_dims={hid!r} not implemented (supports (64, 64, 64) and (64, 64))")
    nr, na = model_args.get("num_radial", NUM_RADIAL), model_args.get("num_angular", NUM_ANGULAR)
    if int(nr) != nr or not 1 <= nr <= NUM_RADIAL:
        raise NotImplementedError(f"num_radial={nr!r} not implemented (supports 1..{NUM_RADIAL})")
    if int(na) != na or not 1 <= na <= NUM_ANGULAR or na % 2 == 0:
        # encoders.py:133-146 / basis.py:11-40: the expansion has 1 + 2*((num_angular - 1) // 2) columns, so an even
        # num_angular cannot feed Linear(num_angular, 64) in the reference either
        raise NotImplementedError(f"num_angular={na!r} not implemented (supports odd values 1..{NUM_ANGULAR})")
    if model_args.get("conv_norm") is not None:
        raise NotImplementedError("conv_norm is not implemented")
    if model_args.get("final_mlp", "MLP") not in {"normal", "MLP"}:
        raise NotImplementedError("gated final_mlp is not implemented")
    if int(model_args.get("n_conv", 4)) < 2:
        raise NotImplementedError("n_conv must be >= 2")
    p = model_args.get("cutoff_coeff", 8)
    if float(p) != int(p) or int(p) < 1:
        # basis.py:170-206: p = 0 means "no envelope" and any positive float is allowed there; the kernels
        # evaluate s^(p-1) by repeated squaring, so only integer p >= 1 (every released checkpoint: 5 or 8)
        raise NotImplementedError(f"cutoff_coeff={p!r} is not implemented (supports integers >= 1)")


def pad_radial(x: np.ndarray, width: int = NUM_RADIAL) -> np.ndarray:
    """Last axis zero-padded to ``width`` (frequencies / the columns of a bias-free embedding Linear)."""
    x = np.asarray(x, np.float32)
    out = np.zeros((*x.shape[:-1], width), np.float32)
    out[..., :x.shape[-1]] = x
    return out


def pad_angular(w: np.ndarray) -> np.ndarray:
    """Columns of ``angle_embedding.weight`` [64, 1 + 2*order] re-laid for the kernels' order-15 expansion
    ``[1/sqrt2 | sin(f_1 t) .. sin(f_15 t) | cos(f_1 t) .. cos(f_15 t)]`` (basis.py:33-40): constant column,
    ``order`` sine columns, zeros, ``order`` cosine columns, zeros."""
    w = np.asarray(w, np.float32)
    order, full = (w.shape[1] - 1) // 2, (NUM_ANGULAR - 1) // 2
    out = np.zeros((w.shape[0], NUM_ANGULAR), np.float32)
    out[:, 0] = w[:, 0]
    out[:, 1:1 + order] = w[:, 1:1 + order]
    out[:, 1 + full:1 + full + order] = w[:, 1 + order:1 + 2 * order]
    return out


def unpad_angular(w31: np.ndarray, num_angular: int) -> np.ndarray:
    """Inverse of ``pad_angular`` (gradients back to the reference's column order)."""
    order, full = (num_angular - 1) // 2, (NUM_ANGULAR - 1) // 2
    return np.concatenate([w31[:, :1 + order], w31[:, 1 + full:1 + full + order]], axis=1)


def _f32(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=np.float32)


@dataclass
class PackedWeights:
    blob: np.ndarray            # float32 [n_values]
    offsets: dict               # name -> (offset, shape)
    n_conv: int
    atom_graph_cutoff: float
    bond_graph_cutoff: float
    cutoff_coeff: int
    is_intensive: bool
    has_composition: bool
    mlp_out_bias: bool = False  # the mlp_out Linears of AtomConv / BondConv carry a bias (0.2.0 checkpoint, model.py:734)
    n_mlp_hidden: int = 3       # hidden layers of the energy head (2 for the 0.2.0 architecture)
    num_radial: int = NUM_RADIAL
    num_angular: int = NUM_ANGULAR

    def get(self, name: str) -> np.ndarray:
        off, shape = self.offsets[name]
        return self.blob[off:off + int(np.prod(shape))].reshape(shape)


def weight_layout(n_conv: int) -> list[tuple[str, tuple]]:
    """Names and shapes of every tensor in the blob, in blob order.  The C side
    (csrc/weights.h) derives the same table from ``n_conv``; both must agree."""
    lay: list[tuple[str, tuple]] = [
        ("atomref", (N_ELEM,)), ("emb", (N_ELEM, D)),
        ("freq_ag", (NUM_RADIAL,)), ("freq_bg", (NUM_RADIAL,)), ("freq_ang", ((NUM_ANGULAR - 1) // 2,)),
        ("w_bond_emb", (D, NUM_RADIAL)), ("w_wag", (D, NUM_RADIAL)), ("w_wbg", (D, NUM_RADIAL)),
        ("w_ang_emb", (D, NUM_ANGULAR)),
    ]
    gated_tail = [("w2c", (D, D)), ("b2c", (D,)), ("w2g", (D, D)), ("b2g", (D,)),
                  ("w2c_t", (D, D)), ("w2g_t", (D, D))]
    ln = [("ln1_g", (D,)), ("ln1_b", (D,)), ("ln2_g", (D,)), ("ln2_b", (D,))]
    for l in range(n_conv):
        p = f"ac{l}."
        lay += [(p + "w_cn", (4 * D, D)),      # rows 0..127: centre block (core|gate), 128..255: neighbour block
                (p + "w_bond", (2 * D, D)),    # bond block (core|gate)
                (p + "b1", (2 * D,)),
                (p + "q_bias", (2 * D,)),      # W_bond . (sum of earlier BondConv mlp_out biases): see pack_weights
                (p + "q_shift", (D,)),         # that sum itself: the constant part of the bond features outside the bond graph (weight gradients)
                *[(p + n, s) for n, s in gated_tail], *[(p + n, s) for n, s in ln],
                (p + "w_out", (D, D)), (p + "b_out", (D,)),
                (p + "w_out_t", (D, D)),       # [in,out] for G(agg) = G(h') . Wout
                (p + "w_cn_t", (2, D, 2 * D)), # [block][in][128]: G(h_atom) += GPc . Wc + GPn . Wn
                (p + "w_bond_t", (D, 2 * D))]
    for l in range(n_conv - 1):
        p = f"bc{l}."
        lay += [(p + "w_bij", (4 * D, D)),     # rows 0..127: bond_i block, 128..255: bond_j block
                (p + "w_ang", (2 * D, D)), (p + "w_ctr", (2 * D, D)), (p + "b1", (2 * D,)),
                *[(p + n, s) for n, s in gated_tail], *[(p + n, s) for n, s in ln],
                (p + "w_out", (D, D)), (p + "b_out", (D,)), (p + "w_out_t", (D, D)),
                (p + "w_bij_t", (2, D, 2 * D)), (p + "w_ang_t", (D, 2 * D)), (p + "w_ctr_t", (D, 2 * D))]
    for l in range(n_conv - 1):
        p = f"au{l}."
        lay += [(p + "w_bij", (4 * D, D)), (p + "w_ang", (2 * D, D)), (p + "w_ctr", (2 * D, D)), (p + "b1", (2 * D,)),
                *[(p + n, s) for n, s in ln],
                (p + "w_bij_t", (2, D, 2 * D)), (p + "w_ang_t", (D, 2 * D)), (p + "w_ctr_t", (D, 2 * D))]
    lay += [("site_w", (D,)), ("site_b", (1,)), ("ro_ln_g", (D,)), ("ro_ln_b", (D,)),
            ("mlp_w0", (D, D)), ("mlp_b0", (D,)), ("mlp_w1", (D, D)), ("mlp_b1", (D,)),
            ("mlp_w2", (D, D)), ("mlp_b2", (D,)), ("mlp_w3", (D,)), ("mlp_b3", (1,)),
            ("mlp_w0_t", (D, D)), ("mlp_w1_t", (D, D)), ("mlp_w2_t", (D, D))]
    return lay


def pack_weights(state_dict: dict, model_args: dict | None = None) -> PackedWeights:
    """Re-lay the reference ``state_dict`` for the kernels."""
    model_args = dict(model_args or {})
    check_model_args(model_args)
    sd = {k: _f32(v) for k, v in state_dict.items()}
    n_conv = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("atom_conv_layers."))
    vals: dict[str, np.ndarray] = {}

    has_comp = "composition_model.fc.weight" in sd
    vals["atomref"] = sd["composition_model.fc.weight"].reshape(-1) if has_comp else np.zeros(N_ELEM, np.float32)
    vals["emb"] = sd["atom_embedding.embedding.weight"]
    # Basis sizes below the kernels' 31 / 31 (0.2.0: 9 / 9) are ZERO-PADDED: a radial function of frequency 0 is
    # identically 0 (sin(0 r) / r, basis.py:108), and a padded column of a bias-free embedding Linear carries weight 0,
    # so the padded model is the same real-number function and the same fp32 sums plus exact zeros.
    n_rad = int(sd["bond_embedding.weight"].shape[1])
    n_ang = int(sd["angle_embedding.weight"].shape[1])
    if not (1 <= n_rad <= NUM_RADIAL and 1 <= n_ang <= NUM_ANGULAR and n_ang % 2 == 1):
        raise NotImplementedError(f"basis sizes {n_rad} / {n_ang} are not implemented")
    order = (n_ang - 1) // 2
    for key in ("bond_basis_expansion.rbf_expansion_ag.frequencies", "bond_basis_expansion.rbf_expansion_bg.frequencies"):
        if sd[key].shape != (n_rad,):
            raise ValueError(f"{key}: expected {n_rad} frequencies")
    if sd["angle_basis_expansion.fourier_expansion.frequencies"].shape != (order,):
        raise ValueError(f"angle frequencies: expected {order}")
    vals["freq_ag"] = pad_radial(sd["bond_basis_expansion.rbf_expansion_ag.frequencies"])
    vals["freq_bg"] = pad_radial(sd["bond_basis_expansion.rbf_expansion_bg.frequencies"])
    vals["freq_ang"] = pad_radial(sd["angle_basis_expansion.fourier_expansion.frequencies"], (NUM_ANGULAR - 1) // 2)
    vals["w_bond_emb"] = pad_radial(sd["bond_embedding.weight"])
    vals["w_wag"] = pad_radial(sd["bond_weights_ag.weight"])
    vals["w_wbg"] = pad_radial(sd["bond_weights_bg.weight"])
    vals["w_ang_emb"] = pad_angular(sd["angle_embedding.weight"])

    def cg(prefix, name):  # stack core | gate along the output axis
        return np.concatenate([sd[f"{prefix}.mlp_core.{name}"], sd[f"{prefix}.mlp_gate.{name}"]], axis=0)

    def gated_tail(p, prefix):
        vals[p + "w2c"] = sd[f"{prefix}.mlp_core.layers.3.weight"]
        vals[p + "b2c"] = sd[f"{prefix}.mlp_core.layers.3.bias"]
        vals[p + "w2g"] = sd[f"{prefix}.mlp_gate.layers.3.weight"]
        vals[p + "b2g"] = sd[f"{prefix}.mlp_gate.layers.3.bias"]
        vals[p + "w2c_t"] = vals[p + "w2c"].T
        vals[p + "w2g_t"] = vals[p + "w2g"].T

    def ln(p, prefix):
        vals[p + "ln1_g"], vals[p + "ln1_b"] = sd[f"{prefix}.bn1.weight"], sd[f"{prefix}.bn1.bias"]
        vals[p + "ln2_g"], vals[p + "ln2_b"] = sd[f"{prefix}.bn2.weight"], sd[f"{prefix}.bn2.bias"]

    def mlp_out(p, prefix):
        vals[p + "w_out"] = sd[f"{prefix}.mlp_out.layers.1.weight"]
        vals[p + "b_out"] = sd.get(f"{prefix}.mlp_out.layers.1.bias", np.zeros(D, np.float32))
        vals[p + "w_out_t"] = vals[p + "w_out"].T

    for l in range(n_conv):
        p, pre = f"ac{l}.", f"atom_conv_layers.{l}.twoBody_atom"
        w1 = cg(pre, "layers.0.weight")                       # [128, 192], columns [centre | bond | nbr] (layers.py:116)
        vals[p + "w_cn"] = np.concatenate([w1[:, 0:D], w1[:, 2 * D:3 * D]], axis=0)
        vals[p + "w_bond"] = w1[:, D:2 * D]
        vals[p + "b1"] = cg(pre, "layers.0.bias")
        gated_tail(p, pre)
        ln(p, pre)
        mlp_out(p, f"atom_conv_layers.{l}")
        vals[p + "w_cn_t"] = np.stack([vals[p + "w_cn"][:2 * D].T, vals[p + "w_cn"][2 * D:].T])
        vals[p + "w_bond_t"] = vals[p + "w_bond"].T
    for l in range(n_conv - 1):
        p, pre = f"bc{l}.", f"bond_conv_layers.{l}.twoBody_bond"
        w1 = cg(pre, "layers.0.weight")                       # [128, 256], columns [bond_i | bond_j | angle | centre] (layers.py:241-243)
        vals[p + "w_bij"] = np.concatenate([w1[:, 0:D], w1[:, D:2 * D]], axis=0)
        vals[p + "w_ang"] = w1[:, 2 * D:3 * D]
        vals[p + "w_ctr"] = w1[:, 3 * D:4 * D]
        vals[p + "b1"] = cg(pre, "layers.0.bias")
        gated_tail(p, pre)
        ln(p, pre)
        mlp_out(p, f"bond_conv_layers.{l}")
        vals[p + "w_bij_t"] = np.stack([vals[p + "w_bij"][:2 * D].T, vals[p + "w_bij"][2 * D:].T])
        for n in ("w_ang", "w_ctr"):
            vals[p + n + "_t"] = vals[p + n].T
    for l in range(n_conv - 1):
        p, pre = f"au{l}.", f"angle_layers.{l}.twoBody_bond"
        w1 = cg(pre, "layers.1.weight")                       # single Linear (functions.py:72-73), same column order (layers.py:351-353)
        vals[p + "w_bij"] = np.concatenate([w1[:, 0:D], w1[:, D:2 * D]], axis=0)
        vals[p + "w_ang"] = w1[:, 2 * D:3 * D]
        vals[p + "w_ctr"] = w1[:, 3 * D:4 * D]
        vals[p + "b1"] = cg(pre, "layers.1.bias")
        ln(p, pre)
        vals[p + "w_bij_t"] = np.stack([vals[p + "w_bij"][:2 * D].T, vals[p + "w_bij"][2 * D:].T])
        for n in ("w_ang", "w_ctr"):
            vals[p + n + "_t"] = vals[p + n].T
    # mlp_out_bias (0.2.0 checkpoint only): the reference aggregates BondConv messages over ALL bonds
    # (layers.py:252-258, num_owner=len(bond_feas)), so every bond -- also those that own no angle --
    # gains the mlp_out bias in every BondConv layer.  Bonds outside the bond graph keep their
    # embedding in the engine, so their layer-l features are hb0 + sum_{m<l} b_out[m]; the constant
    # part enters AtomConv l through Q as this bias row.
    shift = np.zeros(D, np.float64)
    for l in range(n_conv):
        vals[f"ac{l}.q_bias"] = (vals[f"ac{l}.w_bond"].astype(np.float64) @ shift).astype(np.float32)
        vals[f"ac{l}.q_shift"] = shift.astype(np.float32)
        if l < n_conv - 1:
            shift = shift + vals[f"bc{l}.b_out"].astype(np.float64)
    vals["site_w"] = sd["site_wise.weight"].reshape(-1)
    vals["site_b"] = sd["site_wise.bias"].reshape(-1)
    vals["ro_ln_g"], vals["ro_ln_b"] = sd["readout_norm.weight"], sd["readout_norm.bias"]
    # energy head (functions.py:81-91): Linear, act per hidden layer, Dropout, Linear(64, 1) -> the last Linear is
    # layers.{2 n_hidden + 1}: 7 for (64, 64, 64), 5 for (64, 64); an absent third layer leaves zeros in its slots
    n_hidden = mlp_hidden_layers(sd)
    for i in range(3):
        if i < n_hidden:
            vals[f"mlp_w{i}"] = sd[f"mlp.layers.{2 * i}.weight"]
            vals[f"mlp_b{i}"] = sd[f"mlp.layers.{2 * i}.bias"]
        else:
            vals[f"mlp_w{i}"], vals[f"mlp_b{i}"] = np.zeros((D, D), np.float32), np.zeros(D, np.float32)
        vals[f"mlp_w{i}_t"] = vals[f"mlp_w{i}"].T
    vals["mlp_w3"] = sd[f"mlp.layers.{2 * n_hidden + 1}.weight"].reshape(-1)
    vals["mlp_b3"] = sd[f"mlp.layers.{2 * n_hidden + 1}.bias"].reshape(-1)

    offsets, chunks, pos = {}, [], 0
    for name, shape in weight_layout(n_conv):
        v = np.ascontiguousarray(vals[name], dtype=np.float32)
        if tuple(v.shape) != tuple(shape):
            raise ValueError(f"weight {name}: expected shape {shape}, got {v.shape}")
        pad = (-pos) % 4                                     # keep every tensor 16-byte aligned
        if pad:
            chunks.append(np.zeros(pad, np.float32))
            pos += pad
        offsets[name] = (pos, tuple(shape))
        chunks.append(v.reshape(-1))
        pos += v.size
    blob = np.concatenate(chunks)
    return PackedWeights(
        blob=blob, offsets=offsets, n_conv=n_conv,
        atom_graph_cutoff=float(model_args.get("atom_graph_cutoff", 6)),
        bond_graph_cutoff=float(model_args.get("bond_graph_cutoff", 3)),
        cutoff_coeff=int(model_args.get("cutoff_coeff", 8)),
        is_intensive=bool(model_args.get("is_intensive", True)),
        has_composition=has_comp,
        mlp_out_bias=any(k.endswith("mlp_out.layers.1.bias") for k in sd),
        n_mlp_hidden=n_hidden, num_radial=n_rad, num_angular=n_ang,
    )


def mlp_hidden_layers(sd: dict) -> int:
    """Hidden layers of the energy head of a ``state_dict``: its last Linear is ``mlp.layers.{2 n + 1}`` with one output."""
    last = max(int(k.split(".")[2]) for k in sd if k.startswith("mlp.layers.") and k.endswith(".weight"))
    n_hidden = (last - 1) // 2
    if last % 2 != 1 or n_hidden not in (2, 3) or np.asarray(sd[f"mlp.layers.{last}.weight"]).shape[0] != 1:
        raise NotImplementedError(f"energy head with last layer mlp.layers.{last} is not implemented (two or three hidden layers of 64)")
    return n_hidden


def unpack_weight_grads(grad_blob: np.ndarray, pw: PackedWeights) -> dict:
    """Inverse of ``pack_weights`` for gradients: a blob in the weight-blob layout (what
    ``chg_backward`` returns; derived entries such as transposed copies and ``q_bias`` carry no
    gradient of their own and are ignored) -> ``{state_dict key: gradient}`` with the reference's
    tensor shapes, i.e. what ``loss.backward()`` leaves in ``param.grad`` (trainer.py:399-411).
    The frozen AtomRef (model.py:179-182) gets zeros."""
    def G(name):
        off, shape = pw.offsets[name]
        return np.asarray(grad_blob[off:off + int(np.prod(shape))]).reshape(shape)

    out: dict[str, np.ndarray] = {}
    if pw.has_composition:
        out["composition_model.fc.weight"] = np.zeros((1, N_ELEM), grad_blob.dtype)
    out["atom_embedding.embedding.weight"] = G("emb")
    nr, na = pw.num_radial, pw.num_angular                     # the padded tail of a smaller expansion is not a parameter
    out["bond_basis_expansion.rbf_expansion_ag.frequencies"] = G("freq_ag")[:nr]
    out["bond_basis_expansion.rbf_expansion_bg.frequencies"] = G("freq_bg")[:nr]
    out["angle_basis_expansion.fourier_expansion.frequencies"] = G("freq_ang")[:(na - 1) // 2]
    out["bond_embedding.weight"] = G("w_bond_emb")[:, :nr]
    out["bond_weights_ag.weight"] = G("w_wag")[:, :nr]
    out["bond_weights_bg.weight"] = G("w_wbg")[:, :nr]
    out["angle_embedding.weight"] = unpad_angular(G("w_ang_emb"), na)

    def split_cg(pre, name, full):            # rows 0..63 = core, 64..127 = gate
        out[f"{pre}.mlp_core.{name}"], out[f"{pre}.mlp_gate.{name}"] = full[:D], full[D:]

    def gated_tail(p, pre):
        out[f"{pre}.mlp_core.layers.3.weight"], out[f"{pre}.mlp_core.layers.3.bias"] = G(p + "w2c"), G(p + "b2c")
        out[f"{pre}.mlp_gate.layers.3.weight"], out[f"{pre}.mlp_gate.layers.3.bias"] = G(p + "w2g"), G(p + "b2g")

    def ln(p, pre):
        out[f"{pre}.bn1.weight"], out[f"{pre}.bn1.bias"] = G(p + "ln1_g"), G(p + "ln1_b")
        out[f"{pre}.bn2.weight"], out[f"{pre}.bn2.bias"] = G(p + "ln2_g"), G(p + "ln2_b")

    L = pw.n_conv
    for l in range(L):
        p, pre = f"ac{l}.", f"atom_conv_layers.{l}.twoBody_atom"
        w_cn, w_bond = G(p + "w_cn"), G(p + "w_bond")       # [centre(core|gate) ; nbr(core|gate)], [bond(core|gate)]
        split_cg(pre, "layers.0.weight", np.concatenate([w_cn[:2 * D], w_bond, w_cn[2 * D:]], axis=1))   # columns [centre | bond | nbr]
        split_cg(pre, "layers.0.bias", G(p + "b1"))
        gated_tail(p, pre)
        ln(p, pre)
        out[f"atom_conv_layers.{l}.mlp_out.layers.1.weight"] = G(p + "w_out")
        if pw.mlp_out_bias:
            out[f"atom_conv_layers.{l}.mlp_out.layers.1.bias"] = G(p + "b_out")
    for l in range(L - 1):
        p, pre = f"bc{l}.", f"bond_conv_layers.{l}.twoBody_bond"
        w_bij = G(p + "w_bij")
        split_cg(pre, "layers.0.weight", np.concatenate([w_bij[:2 * D], w_bij[2 * D:], G(p + "w_ang"), G(p + "w_ctr")], axis=1))
        split_cg(pre, "layers.0.bias", G(p + "b1"))
        gated_tail(p, pre)
        ln(p, pre)
        out[f"bond_conv_layers.{l}.mlp_out.layers.1.weight"] = G(p + "w_out")
        if pw.mlp_out_bias:
            out[f"bond_conv_layers.{l}.mlp_out.layers.1.bias"] = G(p + "b_out")
    for l in range(L - 1):
        p, pre = f"au{l}.", f"angle_layers.{l}.twoBody_bond"
        w_bij = G(p + "w_bij")
        split_cg(pre, "layers.1.weight", np.concatenate([w_bij[:2 * D], w_bij[2 * D:], G(p + "w_ang"), G(p + "w_ctr")], axis=1))
        split_cg(pre, "layers.1.bias", G(p + "b1"))
        ln(p, pre)
    out["site_wise.weight"], out["site_wise.bias"] = G("site_w").reshape(1, D), G("site_b")
    out["readout_norm.weight"], out["readout_norm.bias"] = G("ro_ln_g"), G("ro_ln_b")
    nh = pw.n_mlp_hidden
    for i in range(nh):
        out[f"mlp.layers.{2 * i}.weight"], out[f"mlp.layers.{2 * i}.bias"] = G(f"mlp_w{i}"), G(f"mlp_b{i}")
    out[f"mlp.layers.{2 * nh + 1}.weight"], out[f"mlp.layers.{2 * nh + 1}.bias"] = G("mlp_w3").reshape(1, D), G("mlp_b3")
    return out

