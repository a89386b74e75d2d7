"""The C-ABI libraries load on a CPU-only box and export every symbol their headers declare;
host-side packing agrees with the C layout.  No compute call is made here (no GPU)."""

from __future__ import annotations

import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, load_case


def _declared(header: str) -> list[str]:
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chg_[a-z0-9_]+)\s*\(", text)))


def test_graph_library_exports_header_symbols():
    from chgnet_amd.graph.converter import graph_lib

    lib = graph_lib()
    names = _declared("chgnet_graph.h")
    assert {"chg_graph_build", "chg_graph_from_neighbors", "chg_graph_free", "chg_graph_strerror"} <= set(names)
    for n in names:
        assert hasattr(lib, n), n


def test_interface_version_of_header_binding_and_library_agree():
    """include/chgnet_hip.h CHG_ABI_VERSION == chgnet_amd._lib.ABI_VERSION == chg_abi_version() of the built library (load() refuses
    any other): chg_engine_create copies *desc, a binding with an older, shorter chg_model_desc would be read past its end."""
    from chgnet_amd import _lib

    with open(os.path.join(REPO, "include", "chgnet_hip.h")) as fh:
        declared = int(re.search(r"#define\s+CHG_ABI_VERSION\s+(\d+)", fh.read()).group(1))
    assert declared == _lib.ABI_VERSION
    assert int(_lib.load().chg_abi_version()) == declared


def test_hip_library_exports_header_symbols():
    from chgnet_amd import _lib

    lib = _lib.load()
    names = _declared("chgnet_hip.h")
    assert len(names) >= 19 and set(names) == set(_lib.EXPORTED_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_weight_blob_layout_agrees_with_c_side(packed_weights):
    from chgnet_amd import _lib
    from chgnet_amd.pack import weight_layout

    lib = _lib.load()
    assert lib.chg_weights_required(packed_weights.n_conv) == packed_weights.blob.size
    assert lib.chg_weights_required(1) == -1 and lib.chg_weights_required(9) == -1
    for off, _ in packed_weights.offsets.values():
        assert off % 4 == 0                         # every tensor 16-byte aligned
    assert [n for n, _ in weight_layout(4)] == list(packed_weights.offsets)


def test_engine_creation_fails_loudly_without_gpu(packed_weights):
    """No silent CPU fallback: without a gfx950 device the product raises."""
    from chgnet_amd import _lib
    from chgnet_amd.engine import Engine

    if _lib.load().chg_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="chg_engine_create failed"):
        Engine(packed_weights, 0)


def test_missing_extension_is_an_error(monkeypatch, tmp_path):
    from chgnet_amd import _lib

    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "hip_lib_path", lambda: str(tmp_path / "libchgnet_hip.so"))
    with pytest.raises(RuntimeError, match="There is no CPU fallback"):
        _lib.load()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(REPO, "chgnet_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f


def test_pack_batch_global_indices_and_bond_nodes():
    from chgnet_amd.pack import pack_batch

    graphs = [load_case(n)[0] for n in ("limno2", "noangle", "s16tri")]
    pb = pack_batch(graphs)
    assert pb.n_struct == 3 and pb.n_atoms == 8 + 2 + 16
    assert pb.n_directed == 2 * pb.n_undirected == sum(len(g.atom_graph) for g in graphs)
    # per-structure blocks keep their local indices after subtracting the offsets
    for i, g in enumerate(graphs):
        e0, e1 = pb.edge_off[i], pb.edge_off[i + 1]
        assert np.array_equal(pb.e_center[e0:e1] - pb.atom_off[i], g.atom_graph[:, 0])
        assert np.array_equal(pb.e_d2u[e0:e1] - pb.und_off[i], g.directed2undirected)
        assert np.all(pb.e_owner[e0:e1] == i)
    # compact bond-graph nodes: monotone, cover exactly the bonds used by angles
    used = np.union1d(pb.a_b1, pb.a_b2)
    assert np.array_equal(pb.bn_und, used) and np.all(np.diff(pb.bn_und) > 0)
    assert np.array_equal(pb.bn_und[pb.a_b1c], pb.a_b1) and np.array_equal(pb.bn_und[pb.a_b2c], pb.a_b2)
    assert np.all(np.diff(pb.a_b1c) >= 0)
    assert (pb.u_bnode >= 0).sum() == pb.n_bnodes
    for arr in pb.arrays.values():
        assert arr.flags["C_CONTIGUOUS"] and arr.dtype in (np.int32, np.float32)


def test_unsupported_architectures_are_rejected(golden_weights):
    from chgnet_amd.pack import pack_weights

    with pytest.raises(NotImplementedError, match="non_linearity"):
        pack_weights(golden_weights, {"non_linearity": "relu"})
    with pytest.raises(NotImplementedError, match="mlp_first"):
        pack_weights(golden_weights, {"mlp_first": False})
    with pytest.raises(NotImplementedError, match="gMLP_norm"):
        pack_weights(golden_weights, {"gMLP_norm": "batch"})


def test_product_build_defines_no_experiment_switch():
    """The wrong-result timing switches of rounds 1-3 (CHG_EXP_*) are gone from the sources; the one diagnostic build switch left,
    CHG_PHASE_TIMING, is refused without CHG_EXPERIMENTS, and the product flags define neither."""
    from chgnet_amd import build

    assert not any("CHG_EXP" in f or "CHG_PHASE_TIMING" in f for f in build.HIP_FLAGS)
    hdr = open(os.path.join(REPO, "chgnet_amd", "csrc", "mfma_tile.h")).read()
    used = set()
    for root, _, files in os.walk(os.path.join(REPO, "chgnet_amd", "csrc")):
        for f in files:
            used |= set(re.findall(r"#\s*if(?:def|ndef)?\s+(?:defined\()?(CHG_EXP_[A-Z0-9_]+)", open(os.path.join(root, f)).read()))
    assert not used, f"experiment switches in the shipping sources: {used}"
    assert "#if defined(CHG_PHASE_TIMING) && !defined(CHG_EXPERIMENTS)" in hdr and "#error" in hdr


def test_ctypes_structs_mirror_the_c_header(tmp_path):
    """``chg_model_desc`` (with the ``n_mlp_hidden`` field the 0.2.0 head needs) and ``chg_out_host`` as gcc lays them
    out from include/chgnet_hip.h == the ctypes mirrors in chgnet_amd/_lib.py, field by field."""
    import subprocess

    from chgnet_amd import _lib

    fields = {"chg_model_desc": _lib.ModelDesc, "chg_out_host": _lib.OutHost, "chg_structs_host": _lib.StructsHost, "chg_batch_host": _lib.BatchHost}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "chgnet_hip.h"', "int main(void) {"]
    for cname, cls in fields.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    subprocess.run(["gcc", f"-I{os.path.join(REPO, 'include')}", str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in fields.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)
