"""Goldens of the RELEASED 0.2.0 ARCHITECTURE from the unmodified reference.

    python tests/golden/make_golden_v020.py      # needs /root/reference (this container only)

The reference model is built with exactly the arguments of ``chgnet/pretrained/0.2.0/README.md:12-37``

    CHGNet(num_radial=9, num_angular=9, mlp_hidden_dims=[64, 64], atom_graph_cutoff=5, bond_graph_cutoff=3,
           cutoff_coeff=5, mlp_out_bias=True, ...defaults...)                     -> 403,126 parameters

(what ``CHGNet.load(model_name="0.2.0")`` instantiates, model.py:718-736; the checkpoint blob itself is absent here:
``.MISSING_LARGE_BLOBS``), ``torch.manual_seed(20)``, every LayerNorm affine and every ``frequencies`` tensor moved off its
initial value by 0.1 randn, the ``mlp_out`` biases by 0.2 randn (they are what makes 0.2.0 different: the reference adds a
BondConv's bias to EVERY bond, layers.py:252-258).  A second weight set is pushed to trained-checkpoint magnitudes like
``weights_trained_like.npz``.

Writes
  weights_v020.npz / weights_v020_trained_like.npz   the two state_dicts
  case_v020_<name>.npz     graph of the reference's OWN converter at cutoffs 5 / 3 (its compiled cygraph; neighbour list
                           from our builder, as in make_golden.py) + ``predict_graph(task="efsm", return_site_energies=True,
                           return_atom_feas=True, return_crystal_feas=True)`` for both weight sets (``out_*`` / ``tl_out_*``)
                           for limno2, s40, s16tri, noangle, li9co7o16
  grad_v020_five.npz       ``p.grad`` of all tensors after ``loss.backward()`` of the seeded E+F+S+M loss of
                           make_grad_golden.py on the five cases in one batch (train mode, create_graph=True: the
                           reference's own double backward, trainer.py:399-411) -- pins the oracle's autograd for this
                           architecture, ``mlp_out`` biases included
"""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from oracle._refimport import load_reference  # noqa: E402

load_reference(fast_graph=True)
from chgnet.graph.converter import CrystalGraphConverter as RefConverter  # noqa: E402
from chgnet.model.model import CHGNet as RefCHGNet  # noqa: E402

from make_golden import make_trained_like, neighbour_list, structures  # noqa: E402
from make_golden_helpers import DuckStructure  # noqa: E402

V020_ARGS = dict(num_radial=9, num_angular=9, mlp_hidden_dims=[64, 64], atom_graph_cutoff=5, bond_graph_cutoff=3,
                 cutoff_coeff=5, mlp_out_bias=True)
CASES = ("limno2", "noangle", "s16tri", "s40", "li9co7o16")


def cotangents(seed: int, n_struct: int, n_atoms: int):
    """Same draw order as make_grad_golden.py (not imported: that module loads the reference without its cygraph)."""
    rng = np.random.default_rng(seed)
    ce, gm = rng.normal(size=n_struct).astype(np.float32), rng.normal(size=n_atoms).astype(np.float32)
    gf, gs = rng.normal(size=(n_atoms, 3)).astype(np.float32), rng.normal(size=(n_struct, 3, 3)).astype(np.float32)
    return ce, gm, gf, gs


def make_v020_model() -> RefCHGNet:
    torch.manual_seed(20)
    model = RefCHGNet(**V020_ARGS)
    gen = torch.Generator().manual_seed(21)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ".bn1." in name or ".bn2." in name or name.startswith("readout_norm") or name.endswith("frequencies"):
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
            elif name.endswith("mlp_out.layers.1.bias"):
                p.add_(0.2 * torch.randn(p.shape, generator=gen))
    model.eval()
    return model


def main() -> None:
    torch.set_num_threads(1)
    model = make_v020_model()
    assert model.n_params == 403126, model.n_params
    model_tl = make_trained_like(make_v020_model())
    for fname, m in (("weights_v020.npz", model), ("weights_v020_trained_like.npz", model_tl)):
        np.savez_compressed(os.path.join(HERE, fname), **{k: v.detach().numpy().copy() for k, v in m.state_dict().items()})
    conv = RefConverter(atom_graph_cutoff=5, bond_graph_cutoff=3, algorithm="fast", on_isolated_atoms="ignore")
    assert conv.algorithm == "fast"
    structs = structures()
    kw = dict(task="efsm", return_site_energies=True, return_atom_feas=True, return_crystal_feas=True)
    graphs = {}
    for name in CASES:
        s = structs[name]
        nl, arrays = neighbour_list(s, 5.0, 3.0)
        rg = conv(DuckStructure(s, nl))
        graphs[name] = rg
        assert np.array_equal(rg.atom_graph.numpy(), arrays["atom_graph"]) and np.array_equal(rg.bond_graph.numpy().reshape(-1, 5), arrays["bond_graph"].reshape(-1, 5))
        save = {
            "atomic_number": s.atomic_numbers, "atom_frac_coord": rg.atom_frac_coord.detach().numpy(),
            "frac_coord_f64": s.frac_coords, "lattice_f64": s.lattice.matrix, "lattice": rg.lattice.detach().numpy(),
            "atom_graph": rg.atom_graph.numpy(), "neighbor_image": rg.neighbor_image.numpy(),
            "directed2undirected": rg.directed2undirected.numpy(), "undirected2directed": rg.undirected2directed.numpy(),
            "bond_graph": rg.bond_graph.numpy().reshape(-1, 5),
        }
        out, out_tl = model.predict_graph(rg, **kw), model_tl.predict_graph(rg, **kw)
        for k, v in out.items():
            save["out_" + k] = np.asarray(v)
        for k, v in out_tl.items():
            save["tl_out_" + k] = np.asarray(v)
        np.savez_compressed(os.path.join(HERE, f"case_v020_{name}.npz"), **save)
        print(name, "N", len(s), "Ed", len(arrays["atom_graph"]), "A", len(arrays["bond_graph"]), "e", float(out["e"]),
              "|f|max", float(np.abs(out["f"]).max()), "| trained-like: e", float(out_tl["e"]), "|f|max", float(np.abs(out_tl["f"]).max()),
              "|s|max", float(np.abs(out_tl["s"]).max()))

    # the reference's own loss.backward() on this architecture (train mode -> create_graph=True)
    train = make_v020_model()
    train.train()
    batch = [graphs[n] for n in CASES]
    n_atoms = sum(len(g.atomic_number) for g in batch)
    ce, gm, gf, gs = cotangents(31, len(batch), n_atoms)
    out = train(batch, task="efsm")
    loss = (out["e"] * torch.tensor(ce)).sum() + (torch.cat(out["m"]) * torch.tensor(gm)).sum() \
        + (torch.cat(out["f"]) * torch.tensor(gf)).sum() + (torch.stack(out["s"]) * torch.tensor(gs)).sum()
    train.zero_grad()
    loss.backward()
    res = {"order": np.array(CASES), "cot_e": ce, "cot_m": gm, "cot_f": gf, "cot_s": gs, "loss": np.float32(loss.item())}
    for k, p in train.named_parameters():
        res["grad/" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "grad_v020_five.npz"), **res)
    g = {k: v for k, v in res.items() if k.startswith("grad/")}
    print("grad_v020_five.npz loss", float(res["loss"]), "tensors", len(g), "max|grad|", max(float(np.abs(v).max()) for v in g.values()))


if __name__ == "__main__":
    main()
