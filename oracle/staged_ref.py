"""numpy model of the *kernel pipeline* (factorised forward + hand-derived backward).

TEST INFRASTRUCTURE ONLY.  Where ``chgnet_oracle.py`` restates the reference (concat ->
Linear, autograd), this file restates what the HIP engine actually executes, stage by
stage and buffer by buffer, in float64 numpy:

  * first gated-MLP layer factorised into per-atom / per-bond partial products
    (P = h_atom.[Wc;Wn]^T, Q = h_bond.Wb^T, R = h_bond[nodes].[Wi;Wj]^T, S = h_atom.Wctr^T),
    so the per-edge / per-angle work is gather+add then one 64->64 contraction;
  * bond features kept as ``hb0[Eu]`` + per-layer compact ``hbc[l][Eb]`` (bond-graph nodes);
  * analytic reverse sweep producing dE/dv_e once, from which forces and the virial follow
    (SURVEY Appendix B).

It is checked against the autograd oracle on CPU (tests/test_staged_ref.py); on the GPU
every named buffer here can be compared with ``Engine.debug_fetch(name)`` to localise a
kernel bug to one stage.
"""

from __future__ import annotations

import numpy as np

KAPPA32 = float(np.float32(1 - 1e-6))  # encoders.py:144: python float folded into an fp32 tensor multiply


def silu(x):
    return x / (1 + np.exp(-x))


def sigmoid(x):
    return 1 / (1 + np.exp(-x))


def dsilu(x):
    s = sigmoid(x)
    return s * (1 + x * (1 - s))


def ln_fwd(x, g, b):
    mu = x.mean(1, keepdims=True)
    var = ((x - mu) ** 2).mean(1, keepdims=True)
    rstd = 1 / np.sqrt(var + 1e-5)
    xhat = (x - mu) * rstd
    return xhat * g + b, xhat, rstd


def ln_bwd(gy, g, xhat, rstd):
    gx = gy * g
    return rstd * (gx - gx.mean(1, keepdims=True) - xhat * (gx * xhat).mean(1, keepdims=True))


def envelope(r, rc, p):
    a, b, c = -(p + 1) * (p + 2) / 2, p * (p + 2), -p * (p + 1) / 2
    s = r / rc
    env = 1 + a * s**p + b * s ** (p + 1) + c * s ** (p + 2)
    denv = (a * p * s ** (p - 1) + b * (p + 1) * s**p + c * (p + 2) * s ** (p + 1)) / rc
    m = s < 1
    return np.where(m, env, 0.0), np.where(m, denv, 0.0)


def rbf(r, rc, freq, p):
    """basis [n,31] and d(basis)/dr."""
    r = r[:, None]
    cn = np.sqrt(2 / rc)
    w = freq[None, :] / rc
    sin, cos = np.sin(w * r), np.cos(w * r)
    env, denv = envelope(r, rc, p)
    base = cn * sin / r
    dbase = cn * (w * cos / r - sin / r**2)
    return env * base, denv * base + env * dbase


def fourier(theta, freq):
    n = len(theta)
    t = np.outer(theta, freq)
    out = np.concatenate([np.full((n, 1), 1 / np.sqrt(2)), np.sin(t), np.cos(t)], axis=1) / np.sqrt(np.pi)
    dout = np.concatenate([np.zeros((n, 1)), freq * np.cos(t), -freq * np.sin(t)], axis=1) / np.sqrt(np.pi)
    return out, dout


class StagedModel:
    def __init__(self, pw, dtype=np.float64) -> None:
        self.pw = pw
        self.dt = dtype
        self.kappa = KAPPA32 if dtype == np.float32 else (1 - 1e-6)

    def W(self, name):
        return self.pw.get(name).astype(self.dt)

    # ------------------------------------------------------------------------------
    def gated_fwd(self, z, p, hidden):
        W = self.W
        D = 64
        if hidden:
            H = silu(z)
            c = H[:, :D] @ W(p + "w2c").T + W(p + "b2c")
            g = H[:, D:] @ W(p + "w2g").T + W(p + "b2g")
        else:
            H = None
            c, g = z[:, :D], z[:, D:]
        n1, xh1, rs1 = ln_fwd(c, W(p + "ln1_g"), W(p + "ln1_b"))
        n2, xh2, rs2 = ln_fwd(g, W(p + "ln2_g"), W(p + "ln2_b"))
        a1, a2 = silu(n1), sigmoid(n2)
        cache = (z, H, n1, xh1, rs1, n2, xh2, rs2, a1, a2)
        return a1 * a2, cache

    def gated_bwd(self, gy, cache, p, hidden, wg=None):
        """dE/dy -> dE/dz.  ``wg`` (dict) additionally receives the gradients of the gated MLP's own
        parameters: LayerNorm affine (column sums of the pre-LayerNorm-backward adjoints), and for the
        hidden form the second Linear (outer products of its output adjoint with the hidden activation)."""
        W = self.W
        D = 64
        z, H, n1, xh1, rs1, n2, xh2, rs2, a1, a2 = cache
        gn1 = gy * a2 * dsilu(n1)
        gn2 = gy * a1 * a2 * (1 - a2)
        gc = ln_bwd(gn1, W(p + "ln1_g"), xh1, rs1)
        gg = ln_bwd(gn2, W(p + "ln2_g"), xh2, rs2)
        if wg is not None:
            wg[p + "ln1_g"], wg[p + "ln1_b"] = (gn1 * xh1).sum(0), gn1.sum(0)
            wg[p + "ln2_g"], wg[p + "ln2_b"] = (gn2 * xh2).sum(0), gn2.sum(0)
            if hidden:
                wg[p + "w2c"], wg[p + "b2c"] = gc.T @ H[:, :D], gc.sum(0)
                wg[p + "w2g"], wg[p + "b2g"] = gg.T @ H[:, D:], gg.sum(0)
        if hidden:
            gH = np.concatenate([gc @ W(p + "w2c"), gg @ W(p + "w2g")], axis=1)
            return gH * dsilu(z)
        return np.concatenate([gc, gg], axis=1)

    # ------------------------------------------------------------------------------
    def run(self, pb, want_bwd=True, e_cot=None) -> dict:
        """``e_cot`` [B] (optional): cotangent of the per-structure energies ``e``.  The reverse sweep then
        starts from ``d(sum_b e_cot[b] e[b])/d(site energy)`` instead of 1 and ``out["wgrad"]`` holds the
        gradient of that scalar with respect to every packed weight tensor (stage A: the
        energy part of the fine-tuning loss, trainer.py:779-869)."""
        W, dt, pw = self.W, self.dt, self.pw
        D = 64
        L = pw.n_conv
        B, N, Ed, Eu, A, Eb = pb.n_struct, pb.n_atoms, pb.n_directed, pb.n_undirected, pb.n_angles, pb.n_bnodes
        buf = {}
        c, n, k = pb.e_center, pb.e_nbr, pb.e_d2u
        ctr, b1c, b2c = pb.a_ctr, pb.a_b1c, pb.a_b2c
        bn = pb.bn_und

        # S0 geometry (model.py:826-850, encoders.py:98-102)
        lat = pb.lattice.astype(dt)
        cart = np.einsum("ni,nij->nj", pb.frac.astype(dt), lat[pb.atom_owner])
        v = cart[c] - cart[n] - np.einsum("ei,eij->ej", pb.e_image.astype(dt), lat[pb.e_owner])
        r = np.sqrt((v * v).sum(1))
        u = v / r[:, None]
        vol = np.einsum("bi,bi->b", lat[:, 0], np.cross(lat[:, 1], lat[:, 2]))
        buf.update(cart=cart, bond_vec=v, bond_len=r, bond_unit=u)

        # S1 bond basis + embedding (only the representative directed edge u2d[k] matters)
        rk = r[pb.u_u2d]
        rbf6, drbf6 = rbf(rk, pw.atom_graph_cutoff, W("freq_ag"), pw.cutoff_coeff)
        rbf3, drbf3 = rbf(rk, pw.bond_graph_cutoff, W("freq_bg"), pw.cutoff_coeff)
        hb0 = rbf6 @ W("w_bond_emb").T
        wag = rbf6 @ W("w_wag").T
        wbgc = (rbf3 @ W("w_wbg").T)[bn]
        buf.update(hb0=hb0, wag=wag, wbgc=wbgc)

        # S2 angle basis + embedding
        if A:
            cosv = (u[pb.a_d1] * u[pb.a_d2]).sum(1) * self.kappa
            theta = np.arccos(cosv)
            four, dfour = fourier(theta, W("freq_ang"))
            ang = [four @ W("w_ang_emb").T]
            buf.update(theta=theta)
        # S3 atom embedding
        atom = [W("emb")[pb.z - 1]]
        hbc = [hb0[bn]]

        def hb_full(l):
            # bonds outside the bond graph keep their embedding, plus -- only for models with an mlp_out
            # bias (0.2.0) and only when the batch has angles -- the biases of the earlier BondConv layers
            h = hb0.copy()
            if A:
                for m in range(l):
                    h = h + W(f"bc{m}.b_out")
            h[bn] = hbc[l]
            return h

        ac_cache, bc_cache, au_cache = {}, {}, {}

        def atom_conv(l, bl):
            p = f"ac{l}."
            P = atom[l] @ W(p + "w_cn").T
            P[:, :2 * D] += W(p + "b1")
            Q = hb_full(bl) @ W(p + "w_bond").T
            z = P[c, :2 * D] + P[n, 2 * D:] + Q[k]
            y, cache = self.gated_fwd(z, p, True)
            m = y * wag[k]
            agg = np.zeros((N, D), dt)
            np.add.at(agg, c, m)
            ac_cache[l] = (cache, y, bl)
            buf[f"ac{l}.agg"] = agg
            return agg @ W(p + "w_out").T + W(p + "b_out") + atom[l]

        for l in range(L - 1):
            atom.append(atom_conv(l, l))
            if A:
                p = f"bc{l}."
                S = atom[l + 1] @ W(p + "w_ctr").T + W(p + "b1")
                R = hbc[l] @ W(p + "w_bij").T
                z = R[b1c, :2 * D] + R[b2c, 2 * D:] + S[ctr] + ang[l] @ W(p + "w_ang").T
                y, cache = self.gated_fwd(z, p, True)
                w1, w2 = wbgc[b1c], wbgc[b2c]
                agg = np.zeros((Eb, D), dt)
                np.add.at(agg, b1c, y * w1 * w2)
                bc_cache[l] = (cache, y)
                buf[f"bc{l}.agg"] = agg
                hbc.append(agg @ W(p + "w_out").T + W(p + "b_out") + hbc[l])
                if l < L - 2:  # the last AngleUpdate's output is never consumed (model.py:442-496)
                    p = f"au{l}."
                    S = atom[l + 1] @ W(p + "w_ctr").T + W(p + "b1")
                    R = hbc[l + 1] @ W(p + "w_bij").T
                    z = R[b1c, :2 * D] + R[b2c, 2 * D:] + S[ctr] + ang[l] @ W(p + "w_ang").T
                    y, cache = self.gated_fwd(z, p, False)
                    au_cache[l] = cache
                    ang.append(ang[l] + y)
            else:
                hbc.append(hbc[l])
        atom_fea = atom[L - 1]
        magmom = np.abs(atom_fea @ W("site_w") + W("site_b")[0])
        atom.append(atom_conv(L - 1, L - 1))

        # readout (model.py:497-509)
        x0, xh0, rs0 = ln_fwd(atom[L], W("ro_ln_g"), W("ro_ln_b"))
        l1 = x0 @ W("mlp_w0").T + W("mlp_b0")
        l2 = silu(l1) @ W("mlp_w1").T + W("mlp_b1")
        three = getattr(self.pw, "n_mlp_hidden", 3) == 3          # two hidden layers (0.2.0): the last Linear reads silu(l2)
        l3 = silu(l2) @ W("mlp_w2").T + W("mlp_b2") if three else l2
        site = silu(l3) @ W("mlp_w3") + W("mlp_b3")[0]
        energy = np.zeros(B, dt)
        np.add.at(energy, pb.atom_owner, site)
        crystal = np.zeros((B, D), dt)
        np.add.at(crystal, pb.atom_owner, x0)
        n_at = np.diff(pb.atom_off).astype(dt)
        wref = W("atomref")[pb.z - 1]
        comp = np.zeros(B, dt)
        np.add.at(comp, pb.atom_owner, wref)
        e = energy / n_at if pw.is_intensive else energy
        if pw.has_composition:
            e = e + (comp / n_at if pw.is_intensive else comp)
        out = {"e": e, "m": magmom, "site_energies": site + (wref if pw.has_composition else 0), "atom_fea": atom_fea,
               "crystal_fea": crystal}
        for i, a in enumerate(atom):
            buf[f"atom{i}"] = a
        for i, h in enumerate(hbc):
            buf[f"hbc{i}"] = h
        if A:
            for i, a in enumerate(ang):
                buf[f"ang{i}"] = a
        out["buffers"] = buf
        if not want_bwd:
            return out

        # ------------------------------ reverse sweep -----------------------------------
        wg = {}                                             # weight gradients (packed names), filled when e_cot is given
        train = e_cot is not None
        cot = np.ones(N, dt)
        if train:
            cot = (np.asarray(e_cot, dt) / (n_at if pw.is_intensive else 1.0))[pb.atom_owner]
            wg["mlp_w3"] = (cot[:, None] * silu(l3)).sum(0)
            wg["mlp_b3"] = np.array([cot.sum()])
        g3 = cot[:, None] * W("mlp_w3")[None, :] * dsilu(l3)
        g2 = (g3 @ W("mlp_w2")) * dsilu(l2) if three else g3
        g1 = (g2 @ W("mlp_w1")) * dsilu(l1)
        gx0 = g1 @ W("mlp_w0")
        Ga = ln_bwd(gx0, W("ro_ln_g"), xh0, rs0)          # dE/d atom[L]
        if train:
            if three:
                wg["mlp_w2"], wg["mlp_b2"] = g3.T @ silu(l2), g3.sum(0)
            wg["mlp_w1"], wg["mlp_b1"] = g2.T @ silu(l1), g2.sum(0)
            wg["mlp_w0"], wg["mlp_b0"] = g1.T @ x0, g1.sum(0)
            wg["ro_ln_g"], wg["ro_ln_b"] = (gx0 * xh0).sum(0), gx0.sum(0)
        buf["Ga_readout"] = Ga.copy()
        Gb = np.zeros((Eu, D), dt)                          # dE/d bond features (node rows double as hbc grads)
        Gwag = np.zeros((Eu, D), dt)
        Gwbgc = np.zeros((Eb, D), dt)
        Gang = np.zeros((A, D), dt)

        def atom_conv_bwd(l):
            nonlocal Ga
            p = f"ac{l}."
            cache, y, bl = ac_cache[l]
            if train:
                wg[p + "w_out"], wg[p + "b_out"] = Ga.T @ buf[f"ac{l}.agg"], Ga.sum(0)
            GA = Ga @ W(p + "w_out")
            Gm = GA[c]
            np.add.at(Gwag, k, Gm * y)
            Gz = self.gated_bwd(Gm * wag[k], cache, p, True, wg if train else None)
            GP = np.zeros((N, 4 * D), dt)
            np.add.at(GP[:, :2 * D], c, Gz)
            np.add.at(GP[:, 2 * D:], n, Gz)
            GQ = np.zeros((Eu, 2 * D), dt)
            np.add.at(GQ, k, Gz)
            if train:   # first layer, factorised: the table gradients contract with the rows the tables were made from
                wg[p + "w_cn"] = GP.T @ atom[l]
                wg[p + "b1"] = GP[:, :2 * D].sum(0)
                wg[p + "w_bond"] = GQ.T @ hb_full(bl)
            Ga = Ga + GP @ W(p + "w_cn")
            Gb[:] += GQ @ W(p + "w_bond")
            buf[f"ac{l}.Gz"] = Gz
            buf[f"ac{l}.GP"] = GP
            buf[f"ac{l}.GQ"] = GQ

        def angle_scatter(Gz, p, hb_rows, atom_rows, ang_rows):
            nonlocal Ga
            GR = np.zeros((Eb, 4 * D), dt)
            np.add.at(GR[:, :2 * D], b1c, Gz)
            np.add.at(GR[:, 2 * D:], b2c, Gz)
            GS = np.zeros((N, 2 * D), dt)
            np.add.at(GS, ctr, Gz)
            if train:
                wg[p + "w_bij"] = GR.T @ hb_rows
                wg[p + "w_ctr"], wg[p + "b1"] = GS.T @ atom_rows, GS.sum(0)
                wg[p + "w_ang"] = Gz.T @ ang_rows
            Gb[bn] += GR @ W(p + "w_bij")
            Ga = Ga + GS @ W(p + "w_ctr")
            Gang[:] += Gz @ W(p + "w_ang")

        atom_conv_bwd(L - 1)
        for l in range(L - 2, -1, -1):
            if A:
                if l < L - 2:
                    p = f"au{l}."
                    Gz = self.gated_bwd(Gang.copy(), au_cache[l], p, False, wg if train else None)
                    angle_scatter(Gz, p, hbc[l + 1], atom[l + 1], ang[l])
                p = f"bc{l}."
                cache, y = bc_cache[l]
                if train:
                    # (the bias reaches EVERY bond's layer-(l+1) features, layers.py:252-258: column sum over all Eu rows of the running dE/d bond)
                    wg[p + "w_out"], wg[p + "b_out"] = Gb[bn].T @ buf[f"bc{l}.agg"], Gb.sum(0)
                Gagg = Gb[bn] @ W(p + "w_out")
                Gu = Gagg[b1c]
                w1, w2 = wbgc[b1c], wbgc[b2c]
                np.add.at(Gwbgc, b1c, Gu * y * w2)
                np.add.at(Gwbgc, b2c, Gu * y * w1)
                Gz = self.gated_bwd(Gu * w1 * w2, cache, p, True, wg if train else None)
                angle_scatter(Gz, p, hbc[l], atom[l + 1], ang[l])
            atom_conv_bwd(l)
        buf.update(Gb=Gb.copy(), Gwag=Gwag, Gwbgc=Gwbgc, Gang=Gang.copy())

        # embeddings / bases -> dE/dr_k, dE/du_e
        Grbf6 = Gb @ W("w_bond_emb") + Gwag @ W("w_wag")
        Gwbg_full = np.zeros((Eu, D), dt)
        Gwbg_full[bn] = Gwbgc
        Grbf3 = Gwbg_full @ W("w_wbg")
        if train:
            wg["emb"] = np.zeros((94, D), dt)
            np.add.at(wg["emb"], pb.z - 1, Ga)              # Ga is dE/d atom[0] now
            wg["w_bond_emb"], wg["w_wag"], wg["w_wbg"] = Gb.T @ rbf6, Gwag.T @ rbf6, Gwbg_full.T @ rbf3

            def dfreq(Grbf, rc, freq):                      # d rbf_j / d f_j = env * sqrt(2/rc) * cos(f_j r/rc) / rc
                env, _ = envelope(rk[:, None], rc, pw.cutoff_coeff)
                return (Grbf * env * np.sqrt(2 / rc) * np.cos(freq[None, :] * rk[:, None] / rc) / rc).sum(0)

            wg["freq_ag"] = dfreq(Grbf6, pw.atom_graph_cutoff, W("freq_ag"))
            wg["freq_bg"] = dfreq(Grbf3, pw.bond_graph_cutoff, W("freq_bg"))
            if A:
                Gfour = Gang @ W("w_ang_emb")
                wg["w_ang_emb"] = Gang.T @ four
                t = np.outer(theta, W("freq_ang"))
                nf = len(W("freq_ang"))
                wg["freq_ang"] = ((Gfour[:, 1:1 + nf] * np.cos(t) - Gfour[:, 1 + nf:] * np.sin(t)) * theta[:, None]).sum(0) / np.sqrt(np.pi)
            out["wgrad"] = wg
        Grk = (Grbf6 * drbf6).sum(1) + (Grbf3 * drbf3).sum(1)
        Gr = np.zeros(Ed, dt)
        Gr[pb.u_u2d] = Grk
        Gu_e = np.zeros((Ed, 3), dt)
        if A:
            Gtheta = ((Gang @ W("w_ang_emb")) * dfour).sum(1)
            Gcos = -Gtheta / np.sqrt(1 - cosv * cosv)
            np.add.at(Gu_e, pb.a_d1, (Gcos * self.kappa)[:, None] * u[pb.a_d2])
            np.add.at(Gu_e, pb.a_d2, (Gcos * self.kappa)[:, None] * u[pb.a_d1])
        Gv = Gr[:, None] * u + (Gu_e - (Gu_e * u).sum(1, keepdims=True) * u) / r[:, None]
        buf.update(Gr=Gr, Gu=Gu_e, Gv=Gv)
        force = np.zeros((N, 3), dt)
        np.add.at(force, c, -Gv)
        np.add.at(force, n, Gv)
        virial = np.zeros((B, 3, 3), dt)
        np.add.at(virial, pb.e_owner, v[:, :, None] * Gv[:, None, :])
        out["f"] = force
        out["s"] = virial * (1 / vol * 160.21766208)[:, None, None]
        return out
