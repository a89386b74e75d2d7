"""Generate the committed golden fixtures by running the UNMODIFIED reference here.

    python tests/golden/make_golden.py          # needs /root/reference (this container only)

Writes (all small, committed):
  weights_seed0.npz      state_dict of reference ``CHGNet()`` (0.3.0 architecture, 412,525
                         parameters), torch.manual_seed(0), then every LayerNorm weight/bias
                         and every basis ``frequencies`` tensor perturbed by 0.1*randn so
                         that non-default affine / learned frequencies are exercised
                         (the pretrained checkpoints are absent: .MISSING_LARGE_BLOBS)
  case_<name>.npz        CrystalGraph arrays (from our native builder) + the reference's
                         ``predict_graph(task="efsm", return_site_energies=True,
                         return_atom_feas=True, return_crystal_feas=True)`` outputs
  graph_<name>.npz       neighbour list fed to the reference converter (legacy Graph.add_edge
                         + line_graph_adjacency_list) and the graph it produced, for the
                         bit-exact indexing tests
The reference model runs on CPU in fp32 (``CHGNet.predict_graph``, model.py:593-665).
"""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle._refimport import REFERENCE_ROOT, load_reference  # noqa: E402

load_reference()
from chgnet.graph.converter import CrystalGraphConverter as RefConverter  # noqa: E402
from chgnet.graph.crystalgraph import CrystalGraph as RefGraph  # noqa: E402
from chgnet.model.model import CHGNet as RefCHGNet  # noqa: E402

from chgnet_amd.graph.converter import build_graph_arrays  # noqa: E402
from chgnet_amd.graph.structure import Lattice, Structure  # noqa: E402


def make_reference_model() -> RefCHGNet:
    torch.manual_seed(0)
    model = RefCHGNet()
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ".bn1." in name or ".bn2." in name or name.startswith("readout_norm") or name.endswith("frequencies"):
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
    model.eval()
    return model


def structures() -> dict[str, Structure]:
    limno2 = Structure.from_file(f"{REFERENCE_ROOT}/examples/mp-18767-LiMnO2.cif")
    licoo = Structure.from_file(f"{REFERENCE_ROOT}/examples/mp-1175469-Li9Co7O16.cif")
    s40 = limno2.make_supercell([5, 1, 1]).perturb(0.01, np.random.default_rng(0))
    s16 = limno2.make_supercell([1, 2, 1]).perturb(0.02, np.random.default_rng(7)).apply_strain(
        [[0.02, 0.01, 0.0], [0.0, -0.015, 0.005], [0.01, 0.0, 0.03]])
    noangle = Structure(Lattice(np.diag([4.2, 4.2, 4.2])), ["Cs", "Cl"], [[0, 0, 0], [0.5, 0.5, 0.5]])
    return {"limno2": limno2, "s40": s40, "s16tri": s16, "noangle": noangle,
            "li9co7o16": licoo.perturb(0.005, np.random.default_rng(3))}


class _DuckStructure:
    """What converter.py:120-134,187 touches on a pymatgen Structure."""

    def __init__(self, s: Structure, nl: dict) -> None:
        self._s, self._nl = s, nl
        self.frac_coords = s.frac_coords
        self.lattice = s.lattice
        self.sites = s.sites
        self.composition = s.composition

    def __len__(self):
        return len(self._s)

    def __iter__(self):
        return iter(self._s)

    def get_neighbor_list(self, r, sites=None, numerical_tol=1e-8):  # noqa: ARG002
        nl = self._nl
        return nl["center"], nl["neighbor"], nl["image"], nl["distance"]


def ref_graph_from(s: Structure, arrays: dict, r_atom=6.0, r_bond=3.0) -> RefGraph:
    return RefGraph(
        atomic_number=torch.tensor(s.atomic_numbers, dtype=torch.int32),
        atom_frac_coord=torch.tensor(s.frac_coords, dtype=torch.float32, requires_grad=True),
        atom_graph=torch.tensor(arrays["atom_graph"], dtype=torch.int32),
        neighbor_image=torch.tensor(arrays["image"], dtype=torch.float32),
        directed2undirected=torch.tensor(arrays["directed2undirected"], dtype=torch.int32),
        undirected2directed=torch.tensor(arrays["undirected2directed"], dtype=torch.int32),
        bond_graph=torch.tensor(arrays["bond_graph"].reshape(-1, 5), dtype=torch.int32),
        lattice=torch.tensor(s.lattice.matrix, dtype=torch.float32, requires_grad=True),
        atom_graph_cutoff=r_atom, bond_graph_cutoff=r_bond,
    )


def main() -> None:
    torch.set_num_threads(1)  # deterministic summation order for the fixtures
    model = make_reference_model()
    sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "weights_seed0.npz"), **sd)
    print("weights:", len(sd), "tensors,", sum(v.size for v in sd.values()), "values")

    structs = structures()
    ref_graphs = {}
    for name, s in structs.items():
        arrays = build_graph_arrays(s.frac_coords, s.lattice.matrix, 6.0, 3.0)
        rg = ref_graph_from(s, arrays)
        ref_graphs[name] = rg
        out = model.predict_graph(rg, task="efsm", return_site_energies=True,
                                  return_atom_feas=True, return_crystal_feas=True)
        save = {
            "atomic_number": s.atomic_numbers, "atom_frac_coord": s.frac_coords.astype(np.float32),
            "frac_coord_f64": s.frac_coords, "lattice_f64": s.lattice.matrix,
            "lattice": s.lattice.matrix.astype(np.float32),
            "atom_graph": arrays["atom_graph"], "neighbor_image": arrays["image"].astype(np.float32),
            "directed2undirected": arrays["directed2undirected"],
            "undirected2directed": arrays["undirected2directed"],
            "bond_graph": arrays["bond_graph"].reshape(-1, 5),
        }
        for k, v in out.items():
            save["out_" + k] = np.asarray(v)
        np.savez_compressed(os.path.join(HERE, f"case_{name}.npz"), **save)
        print(name, "N", len(s), "Ed", len(arrays["atom_graph"]), "A", len(arrays["bond_graph"]),
              "e", float(out["e"]), "|f|max", float(np.abs(out["f"]).max()))

    # one batched call (mixed sizes, a zero-angle structure in the middle): batch == singles
    order = ["limno2", "noangle", "s16tri"]
    outs = model.predict_graph([ref_graphs[n] for n in order], task="efsm", return_site_energies=True,
                               return_atom_feas=True, return_crystal_feas=True, batch_size=16)
    save = {"order": np.array(order)}
    for n, o in zip(order, outs):
        for k, v in o.items():
            save[f"{n}_{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "batch_mixed.npz"), **save)

    # graph-indexing goldens: reference converter (legacy python algorithm) on our neighbour list
    conv = RefConverter(atom_graph_cutoff=6, bond_graph_cutoff=3, algorithm="legacy")
    for name in ("limno2", "s16tri", "noangle"):
        s = structs[name]
        arrays = build_graph_arrays(s.frac_coords, s.lattice.matrix, 6.0, 3.0)
        nl = {"center": arrays["atom_graph"][:, 0].astype(np.int64), "neighbor": arrays["atom_graph"][:, 1].astype(np.int64),
              "image": arrays["image"].astype(np.int64), "distance": arrays["distance"]}
        # shuffle rows inside each centre block so the numbering rules are really exercised
        rng = np.random.default_rng(11)
        perm = np.concatenate([rng.permutation(np.flatnonzero(nl["center"] == c)) for c in range(len(s))])
        nl = {k: v[perm] for k, v in nl.items()}
        g = conv(_DuckStructure(s, nl))
        np.savez_compressed(
            os.path.join(HERE, f"graph_{name}.npz"), n_atoms=len(s), **{"nl_" + k: v for k, v in nl.items()},
            atom_graph=g.atom_graph.numpy(), directed2undirected=g.directed2undirected.numpy(),
            undirected2directed=g.undirected2directed.numpy(), bond_graph=g.bond_graph.numpy().reshape(-1, 5),
            neighbor_image=g.neighbor_image.numpy())
        print("graph golden", name, tuple(g.atom_graph.shape), tuple(g.bond_graph.shape))


if __name__ == "__main__":
    main()
