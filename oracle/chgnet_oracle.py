"""CPU oracle: restatement of the reference CHGNet E/F/S/M path in plain torch (CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The product never imports it.

This is a floating-point path, so the oracle is a torch restatement (fp32 by default,
fp64 on request for "ground truth" error budgets) of exactly the reference's algorithm,
function by function, and -- like the reference -- it obtains forces / stress from
``torch.autograd.grad`` of the summed energy, NOT from the hand-derived backward the HIP
kernels implement; the two derivations are therefore independent.

Pinning: ``tests/test_oracle_golden.py`` checks it against ``tests/golden/case_*.npz``,
which were produced by the unmodified reference (``tests/golden/make_golden.py``), and --
when ``/root/reference`` is present -- against the live reference on fresh random inputs.

Reference lines followed (all relative to /root/reference/chgnet):
  model/model.py:792-913    BatchedGraph.from_graphs (geometry, bases, index offsets)
  model/encoders.py:73-111  BondEncoder.forward      133-146 AngleEncoder.forward
  model/basis.py:33-40      Fourier                   93-116 RadialBessel  188-206 CutoffPolynomial
  model/model.py:389-542    CHGNet._compute
  model/layers.py:81-137    AtomConv.forward  208-265 BondConv.forward  321-363 AngleUpdate.forward
  model/functions.py:10-40  aggregate  98-107 MLP  168-183 GatedMLP
  model/composition_model.py:102-126,175-205  AtomRef
"""

from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

N_ELEM = 94


def _np(x, dtype):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


class OracleCHGNet:
    """Functional CHGNet (0.3.0 architecture family) evaluated from a ``state_dict``."""

    def __init__(self, state_dict: dict, *, atom_graph_cutoff: float = 6.0, bond_graph_cutoff: float = 3.0,
                 cutoff_coeff: int = 8, is_intensive: bool = True, dtype=torch.float32) -> None:
        self.dtype = dtype
        self.w = {k: torch.tensor(_np(v, np.float64), dtype=dtype) for k, v in state_dict.items()}
        self.rc_ag = float(atom_graph_cutoff)
        self.rc_bg = float(bond_graph_cutoff)
        self.p = cutoff_coeff
        self.is_intensive = is_intensive
        self.n_conv = 1 + max(int(k.split(".")[1]) for k in self.w if k.startswith("atom_conv_layers."))
        self.has_comp = "composition_model.fc.weight" in self.w

    # ---- basis functions -------------------------------------------------------------
    def _envelope(self, r, rc):  # basis.py:181-206
        p = self.p
        a = -(p + 1) * (p + 2) / 2
        b = p * (p + 2)
        c = -p * (p + 1) / 2
        s = r / rc
        env = 1 + a * s**p + b * s ** (p + 1) + c * s ** (p + 2)
        return torch.where(s < 1, env, torch.zeros_like(s))

    def _rbf(self, r, rc, freq):  # basis.py:108-116
        dist = r[:, None]
        d_scaled = dist * (1 / rc)
        out = (2 * (1 / rc)) ** 0.5 * torch.sin(freq * d_scaled) / dist
        return self._envelope(dist, rc) * out

    def _fourier(self, theta, freq):  # basis.py:33-40
        order = freq.shape[0]
        result = theta.new_zeros(theta.shape[0], 1 + 2 * order)
        result[:, 0] = 1 / math.sqrt(2.0)
        tmp = torch.outer(theta, freq)
        result[:, 1 : order + 1] = torch.sin(tmp)
        result[:, order + 1 :] = torch.cos(tmp)
        return result / math.sqrt(math.pi)

    # ---- layers ----------------------------------------------------------------------
    def _ln(self, x, prefix):
        return F.layer_norm(x, (x.shape[-1],), self.w[prefix + ".weight"], self.w[prefix + ".bias"], 1e-5)

    def _mlp2(self, x, prefix):  # MLP with one hidden layer: layers.0, SiLU, Dropout, layers.3
        w = self.w
        h = F.silu(F.linear(x, w[prefix + ".layers.0.weight"], w[prefix + ".layers.0.bias"]))
        return F.linear(h, w[prefix + ".layers.3.weight"], w[prefix + ".layers.3.bias"])

    def _gated(self, x, prefix, hidden: bool):  # functions.py:177-183
        w = self.w
        if hidden:
            core = self._mlp2(x, prefix + ".mlp_core")
            gate = self._mlp2(x, prefix + ".mlp_gate")
        else:  # hidden_dim=0 -> [Dropout, Linear] (functions.py:72-73)
            core = F.linear(x, w[prefix + ".mlp_core.layers.1.weight"], w[prefix + ".mlp_core.layers.1.bias"])
            gate = F.linear(x, w[prefix + ".mlp_gate.layers.1.weight"], w[prefix + ".mlp_gate.layers.1.bias"])
        core = F.silu(self._ln(core, prefix + ".bn1"))
        gate = torch.sigmoid(self._ln(gate, prefix + ".bn2"))
        return core * gate

    @staticmethod
    def _aggregate(data, owners, num_owner):  # functions.py:10-40 with average=False
        out = data.new_zeros(num_owner, data.shape[1])
        return out.index_add_(0, owners, data)

    def _mlp_out(self, x, prefix):  # MLP(hidden_dim=0, bias=mlp_out_bias): [Dropout, Linear]
        return F.linear(x, self.w[prefix + ".layers.1.weight"], self.w.get(prefix + ".layers.1.bias"))

    # ---- the path ---------------------------------------------------------------------
    def forward(self, graphs, task: str = "efsm", *, return_site_energies=False, return_atom_feas=False,
                return_crystal_feas=False, return_intermediates=False, as_tensors=False, create_graph=False):
        """graphs: sequence of objects with the CrystalGraph attributes.  Returns per-structure
        numpy arrays exactly like ``CHGNet.predict_graph`` (model.py:651-663), batched once.

        ``as_tensors=True`` returns the batch-wide torch tensors instead (``e`` [B], ``f`` [N,3], ``s`` [B,3,3],
        ``m`` [N]) still attached to the autograd graph; with ``create_graph=True`` forces and stress are
        differentiable like in the reference's training path (model.py:521-530)."""
        w, dt = self.w, self.dtype
        B = len(graphs)
        n_at = [len(_np(g.atomic_number, np.int64)) for g in graphs]
        n_ed = [len(_np(g.atom_graph, np.int64).reshape(-1, 2)) for g in graphs]
        n_un = [len(_np(g.undirected2directed, np.int64)) for g in graphs]
        n_an = [len(_np(g.bond_graph, np.int64).reshape(-1, 5)) for g in graphs]
        a_off = np.concatenate([[0], np.cumsum(n_at)])
        e_off = np.concatenate([[0], np.cumsum(n_ed)])
        u_off = np.concatenate([[0], np.cumsum(n_un)])

        Z = torch.tensor(np.concatenate([_np(g.atomic_number, np.int64) for g in graphs]))
        frac = torch.tensor(np.concatenate([_np(g.atom_frac_coord, np.float64).reshape(-1, 3) for g in graphs]), dtype=dt)
        latt = torch.tensor(np.stack([_np(g.lattice, np.float64).reshape(3, 3) for g in graphs]), dtype=dt)
        ag = torch.tensor(np.concatenate([_np(g.atom_graph, np.int64).reshape(-1, 2) + a_off[i] for i, g in enumerate(graphs)]))
        image = torch.tensor(np.concatenate([_np(g.neighbor_image, np.float64).reshape(-1, 3) for g in graphs]), dtype=dt)
        d2u = torch.tensor(np.concatenate([_np(g.directed2undirected, np.int64) + u_off[i] for i, g in enumerate(graphs)]))
        u2d = torch.tensor(np.concatenate([_np(g.undirected2directed, np.int64) + e_off[i] for i, g in enumerate(graphs)]))
        bgs = []
        for i, g in enumerate(graphs):
            bg = _np(g.bond_graph, np.int64).reshape(-1, 5)
            bgs.append(bg + np.array([a_off[i], u_off[i], e_off[i], u_off[i], e_off[i]]))
        bg = torch.tensor(np.concatenate(bgs)) if sum(n_an) else torch.zeros((0, 5), dtype=torch.long)
        atom_owner = torch.tensor(np.repeat(np.arange(B), n_at))
        edge_owner = torch.tensor(np.repeat(np.arange(B), n_ed))

        # --- BatchedGraph.from_graphs (model.py:820-899), all structures at once ---------
        strain = torch.zeros(B, 3, 3, dtype=dt, requires_grad=True)            # :827
        lattice = latt @ (torch.eye(3, dtype=dt) + strain)                       # :828-830
        volumes = (lattice[:, 0] * torch.linalg.cross(lattice[:, 1], lattice[:, 2])).sum(1)  # :834-836
        cart = (frac[:, None, :] @ lattice[atom_owner]).squeeze(1)              # :840
        center = cart[ag[:, 0]]
        neighbor = cart[ag[:, 1]] + (image[:, None, :] @ lattice[edge_owner]).squeeze(1)  # encoders.py:98
        bond_vec = center - neighbor                                            # :99
        bond_len = torch.norm(bond_vec, dim=1)                                  # :100
        bond_unit = bond_vec / bond_len[:, None]                                # :102
        und_len = bond_len[u2d]                                                 # :106-108
        rbf_ag = self._rbf(und_len, self.rc_ag, w["bond_basis_expansion.rbf_expansion_ag.frequencies"])
        rbf_bg = self._rbf(und_len, self.rc_bg, w["bond_basis_expansion.rbf_expansion_bg.frequencies"])
        n_atoms, n_und, n_ang = len(Z), len(u2d), len(bg)
        if n_ang:
            cosine = (bond_unit[bg[:, 2]] * bond_unit[bg[:, 4]]).sum(1) * (1 - 1e-6)   # encoders.py:144
            theta = torch.acos(cosine)
            ang_basis = self._fourier(theta, w["angle_basis_expansion.fourier_expansion.frequencies"])

        inter = {}
        # --- CHGNet._compute (model.py:427-542) -----------------------------------------
        atom = w["atom_embedding.embedding.weight"][Z - 1]                      # :432-434
        bond = F.linear(rbf_ag, w["bond_embedding.weight"])                     # :435
        w_ag = F.linear(rbf_ag, w["bond_weights_ag.weight"])                    # :436
        w_bg = F.linear(rbf_bg, w["bond_weights_bg.weight"])                    # :437
        if n_ang:
            ang = F.linear(ang_basis, w["angle_embedding.weight"])              # :439
        inter.update(bond_len=bond_len, bond_unit=bond_unit, rbf_ag=rbf_ag, rbf_bg=rbf_bg, atom0=atom, bond0=bond,
                     w_ag=w_ag, w_bg=w_bg)
        if n_ang:
            inter.update(theta=theta, ang0=ang)

        def atom_conv(layer, atom, bond):                                       # layers.py:113-132
            pre = f"atom_conv_layers.{layer}"
            msg = torch.cat([atom[ag[:, 0]], bond[d2u], atom[ag[:, 1]]], dim=1)
            msg = self._gated(msg, pre + ".twoBody_atom", hidden=True) * w_ag[d2u]
            new = self._aggregate(msg, ag[:, 0], n_atoms)
            return self._mlp_out(new, pre + ".mlp_out") + atom

        magmom = atom_fea_out = None
        for layer in range(self.n_conv - 1):                                    # model.py:442-487
            atom = atom_conv(layer, atom, bond)
            inter[f"atom{layer + 1}"] = atom
            if n_ang:
                pre = f"bond_conv_layers.{layer}"                               # layers.py:238-260
                tot = torch.cat([bond[bg[:, 1]], bond[bg[:, 3]], ang, atom[bg[:, 0]]], dim=1)
                upd = self._gated(tot, pre + ".twoBody_bond", hidden=True) * w_bg[bg[:, 1]] * w_bg[bg[:, 3]]
                new = self._aggregate(upd, bg[:, 1], n_und)
                bond = self._mlp_out(new, pre + ".mlp_out") + bond
                inter[f"bond{layer + 1}"] = bond
                pre = f"angle_layers.{layer}"                                   # layers.py:348-360
                tot = torch.cat([bond[bg[:, 1]], bond[bg[:, 3]], ang, atom[bg[:, 0]]], dim=1)
                ang = self._gated(tot, pre + ".twoBody_bond", hidden=False) + ang
                inter[f"ang{layer + 1}"] = ang
            if layer == self.n_conv - 2:                                        # model.py:477-487
                atom_fea_out = atom
                magmom = torch.abs(F.linear(atom, w["site_wise.weight"], w["site_wise.bias"])).view(-1)
        atom = atom_conv(self.n_conv - 1, atom, bond)                           # :490-496
        inter[f"atom{self.n_conv}"] = atom
        atom = self._ln(atom, "readout_norm")                                   # :497-498
        x = atom                                # MLP 64-64-64(-64)-1 (functions.py:81-91): Linear + act per hidden layer, Dropout, Linear
        last = max(int(k.split(".")[2]) for k in w if k.startswith("mlp.layers.") and k.endswith(".weight"))   # 7 (0.3.0) or 5 (0.2.0)
        for k in range(0, last - 1, 2):
            x = F.silu(F.linear(x, w[f"mlp.layers.{k}.weight"], w[f"mlp.layers.{k}.bias"]))
        site_e = F.linear(x, w[f"mlp.layers.{last}.weight"], w[f"mlp.layers.{last}.bias"])  # [N,1]
        energy = self._aggregate(site_e, atom_owner, B).view(-1)                # :503
        crystal = self._aggregate(atom, atom_owner, B)                          # :508-509

        out = {}
        if "f" in task:                                                         # :517-524
            (gpos,) = torch.autograd.grad(energy.sum(), cart, retain_graph=True, create_graph=create_graph)
            out["f"] = -gpos
        if "s" in task:                                                         # :527-535
            (gstrain,) = torch.autograd.grad(energy.sum(), strain, retain_graph=True, create_graph=create_graph)
            scale = 1 / volumes * 160.21766208
            out["s"] = gstrain * scale[:, None, None]
        apg = torch.tensor(n_at, dtype=dt)
        e = energy / apg if self.is_intensive else energy                       # :538-540
        site = site_e.squeeze(1)
        if self.has_comp:                                                       # model.py:356-358,378; composition_model.py:175-205
            wref = w["composition_model.fc.weight"][0]
            comp = torch.stack([torch.bincount(Z[a_off[i]:a_off[i + 1]] - 1, minlength=N_ELEM) for i in range(B)]).to(dt)
            if self.is_intensive:
                comp = comp / apg[:, None]
            e = e + F.linear(comp, w["composition_model.fc.weight"]).view(-1)
            site = site + wref[Z - 1]                                           # model.py:379-386
        out["e"] = e
        if as_tensors:
            out["m"] = magmom
            out["atoms_per_graph"] = apg
            return out

        def split(t, offs):
            return [t[offs[i]:offs[i + 1]].detach().cpu().numpy() for i in range(B)]

        res = {"e": [x for x in out["e"].detach().cpu().numpy()]}
        if "f" in task:
            res["f"] = split(out["f"], a_off)
        if "s" in task:
            res["s"] = [x for x in out["s"].detach().cpu().numpy()]
        if "m" in task:
            res["m"] = split(magmom, a_off)
        if return_site_energies:
            res["site_energies"] = split(site, a_off)
        if return_atom_feas:
            res["atom_fea"] = split(atom_fea_out, a_off)
        if return_crystal_feas:
            res["crystal_fea"] = [x for x in crystal.detach().cpu().numpy()]
        if return_intermediates:
            res["intermediates"] = {k: v.detach().cpu().numpy() for k, v in inter.items()}
        return res

    def predict_graph(self, graph, task="efsm", *, return_site_energies=False, return_atom_feas=False,
                      return_crystal_feas=False, batch_size: int = 16):
        """Same chunking and per-structure dict output as model.py:593-665."""
        single = hasattr(graph, "atomic_number")
        graphs = [graph] if single else list(graph)
        preds = []
        for s in range(0, len(graphs), batch_size):
            chunk = graphs[s:s + batch_size]
            r = self.forward(chunk, task, return_site_energies=return_site_energies,
                             return_atom_feas=return_atom_feas, return_crystal_feas=return_crystal_feas)
            for i in range(len(chunk)):
                preds.append({k: np.asarray(v[i]) for k, v in r.items()})
        return preds[0] if single else preds

    def parameter_gradients(self, graphs, loss_fn, task: str = "e") -> dict:
        """d loss / d parameter for every tensor of the state_dict, by autograd through this restatement
        (what ``loss.backward()`` gives the reference's Trainer, trainer.py:399-411).  ``loss_fn`` maps the
        tensor dictionary of ``forward(as_tensors=True)`` to a scalar.  The frozen AtomRef
        (model.py:179-182) gets a zero gradient."""
        names = [k for k in self.w if k != "composition_model.fc.weight"]
        for k in names:
            self.w[k].requires_grad_(True)
        try:
            out = self.forward(graphs, task, as_tensors=True, create_graph=True)
            loss = loss_fn(out)
            grads = torch.autograd.grad(loss, [self.w[k] for k in names], allow_unused=True)
        finally:
            for k in names:
                self.w[k].requires_grad_(False)
        res = {k: (g.detach().cpu().numpy() if g is not None else np.zeros(tuple(self.w[k].shape))) for k, g in zip(names, grads)}
        if self.has_comp:
            res["composition_model.fc.weight"] = np.zeros(tuple(self.w["composition_model.fc.weight"].shape))
        return res

