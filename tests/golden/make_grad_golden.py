"""Gradient goldens from the reference's OWN backward (VERDICT r02 "pin the gradient oracle").

    python tests/golden/make_grad_golden.py      # needs /root/reference (this container only)

Runs the unmodified reference ``CHGNet`` in train mode (``model.train()``; ``forward(graphs, task="efsm")`` builds the
double-backward graph, model.py:517-535 ``create_graph=True``), forms

    loss = sum_b ce_b e_b + sum_i gm_i m_i + sum_i gF_i . f_i + sum_b gS_b : s_b          (seeded cotangents)

and calls ``loss.backward()`` exactly like the Trainer (trainer.py:399-411).  Writes ``p.grad`` of all 136 tensors:

  grad_five_seed0.npz          limno2, noangle, s16tri, s40, li9co7o16 in one batch, weights_seed0
  grad_five_trained_like.npz   the same batch, weights_trained_like
  grad_mixed_seed0.npz         limno2, noangle, s16tri (a zero-angle structure in the middle), weights_seed0

Each file also holds the cotangents (``cot_e``, ``cot_m``, ``cot_f``, ``cot_s``) and the case order, so the CPU test
(oracle fp64 vs fixture) and the GPU test (engine vs fixture) rebuild the same loss.  float32, fp32 reference on CPU,
one torch thread.
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

load_reference(fast_graph=False)
from chgnet.graph.crystalgraph import CrystalGraph as RefGraph  # noqa: E402
from chgnet.model.model import CHGNet as RefCHGNet  # noqa: E402

FIVE = ("limno2", "noangle", "s16tri", "s40", "li9co7o16")
MIXED = ("limno2", "noangle", "s16tri")


def ref_graph(name: str) -> RefGraph:
    d = np.load(os.path.join(HERE, f"case_{name}.npz"))
    return RefGraph(
        atomic_number=torch.tensor(d["atomic_number"], dtype=torch.int32),
        atom_frac_coord=torch.tensor(d["atom_frac_coord"], dtype=torch.float32),
        atom_graph=torch.tensor(d["atom_graph"], dtype=torch.int32),
        neighbor_image=torch.tensor(d["neighbor_image"], dtype=torch.float32),
        directed2undirected=torch.tensor(d["directed2undirected"], dtype=torch.int32),
        undirected2directed=torch.tensor(d["undirected2directed"], dtype=torch.int32),
        bond_graph=torch.tensor(d["bond_graph"].reshape(-1, 5), dtype=torch.int32),
        lattice=torch.tensor(d["lattice"], dtype=torch.float32),
        atom_graph_cutoff=6, bond_graph_cutoff=3,
    )


def cotangents(seed: int, n_struct: int, n_atoms: int):
    """Same draw order as tests/test_gpu_train.py::test_force_and_stress_loss_gradients_vs_double_backward."""
    rng = np.random.default_rng(seed)
    ce, gm = rng.normal(size=n_struct).astype(np.float32), rng.normal(size=n_atoms).astype(np.float32)
    gf, gs = rng.normal(size=(n_atoms, 3)).astype(np.float32), rng.normal(size=(n_struct, 3, 3)).astype(np.float32)
    return ce, gm, gf, gs


def reference_gradients(weights: dict, names, seed: int) -> dict:
    model = RefCHGNet()
    model.load_state_dict({k: torch.tensor(v) for k, v in weights.items()})
    model.train()
    graphs = [ref_graph(n) for n in names]
    n_atoms = sum(len(g.atomic_number) for g in graphs)
    ce, gm, gf, gs = cotangents(seed, len(graphs), n_atoms)
    out = model(graphs, task="efsm")
    loss = (out["e"] * torch.tensor(ce)).sum() + (torch.cat(out["m"]) * torch.tensor(gm)).sum() \
        + (torch.cat(out["f"]) * torch.tensor(gf)).sum() + (torch.stack(out["s"]) * torch.tensor(gs)).sum()
    model.zero_grad()
    loss.backward()
    res = {"order": np.array(names), "cot_e": ce, "cot_m": gm, "cot_f": gf, "cot_s": gs, "loss": np.float32(loss.item())}
    for k, p in model.named_parameters():
        res["grad/" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().astype(np.float32)
    return res


def main() -> None:
    torch.set_num_threads(1)
    seed0 = dict(np.load(os.path.join(HERE, "weights_seed0.npz")))
    tl = dict(np.load(os.path.join(HERE, "weights_trained_like.npz")))
    for fname, w, names, seed in (("grad_five_seed0.npz", seed0, FIVE, 23), ("grad_five_trained_like.npz", tl, FIVE, 23),
                                  ("grad_mixed_seed0.npz", seed0, MIXED, 29)):
        res = reference_gradients(w, names, seed)
        np.savez_compressed(os.path.join(HERE, fname), **res)
        g = {k: v for k, v in res.items() if k.startswith("grad/")}
        print(fname, "loss", float(res["loss"]), "tensors", len(g), "max|grad|", max(float(np.abs(v).max()) for v in g.values()))


if __name__ == "__main__":
    main()
