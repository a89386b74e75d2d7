"""Stage B of the fine-tuning backward: the exact parameter gradient of a loss that depends on
forces and stress, as a float64 numpy model of the sweep the HIP engine is to execute.

TEST INFRASTRUCTURE ONLY (derivation check; tests/test_staged_ref.py compares it with torch double-backward
through the oracle, i.e. with what ``loss.backward()`` does in the reference: chgnet/model/model.py:517-535
``create_graph=True`` + chgnet/trainer/trainer.py:399-411).

Derivation.  F_i = -dE/dx_i and sigma_b = (kappa / V_b) dE/d eps_b are first derivatives of E_tot = sum_b E_b, so
with cotangents gE, gF, gS of a loss L(e, F, sigma)

    dL/d theta = d/d theta [ sum_b ce_b E_b  +  D E_tot ],      D E_tot = d/d tau E_tot(v + tau vdot) at tau = 0,
    vdot_e = ux[c_e] - ux[n_e] + v_e W_b(e),   ux = -gF,   W_b = (kappa / V_b) gS_b,   ce_b = gE_b / n_b

(all geometry enters through the bond vectors v_e = (frac_c - frac_n - image) L (1 + eps)).  D E_tot is ONE
forward-mode (tangent) sweep through the network -- one direction for the whole loss, not one per atom -- and its
theta-gradient is a reverse sweep over the (primal, tangent) program.  For every op y = f(x; theta) with tangent
ydot = J(x) xdot that reverse sweep carries two adjoints:

    G(y)   = d(D E)/d ydot = dE/dy          the ordinary adjoint (seed 1 at every site energy): what chg_predict
                                            already computes for the forces;
    bar(y) = d Phi/d y,  Phi = sum ce_b E_b + D E      (seed ce at the site energies)

    bar(x)      = J^T bar(y)  +  d/dx [ G(y) . J(x) xdot ]               second-order source term
    bar(theta) += d/d theta [ bar(y) . f ]  +  d/d theta [ G(y) . J xdot ]

i.e. the ordinary backward applied to bar(y), plus source terms at every nonlinearity that need the tangent
activations and G:
    Linear   y = W x:        bar(W) += bar(y) x^T + G(y) xdot^T                      (no source term)
    act      y = phi(x):     bar(x) = phi'(x) bar(y) + phi''(x) xdot G(y)
    product  y = a b:        bar(a) = b bar(y) + bdot G(y)                           (and symmetrically)
    LayerNorm xhat = (x - mu) r:   with h = G(xhat), t = xdot, P(a) = a - mean(a) - xhat mean(a xhat)
             bar(x) = r P(bar(xhat)) - r^2 [ xhat mean(h P(t)) + P(h) mean(xhat t) + P(t) mean(h xhat) ]
    bases    rbf_j(r; f_j), four(theta; g_j):  bar(f) += bar(rbf) d rbf/df + G(rbf) (d^2 rbf / dr df) rdot
Cost: one tangent forward (~1x the forward) and a reverse sweep of ~2x the ordinary one.
"""

from __future__ import annotations

import numpy as np

from oracle.staged_ref import dsilu, envelope, ln_fwd, sigmoid, silu

KAPPA_GPA = 160.21766208


def ddsilu(x):
    s = sigmoid(x)
    return s * (1 - s) * (2 + x * (1 - 2 * s))


def dsigmoid(x):
    s = sigmoid(x)
    return s * (1 - s)


def ddsigmoid(x):
    s = sigmoid(x)
    return s * (1 - s) * (1 - 2 * s)


def _proj(a, xhat):
    return a - a.mean(1, keepdims=True) - xhat * (a * xhat).mean(1, keepdims=True)


def ln_tangent(xd, xhat, rstd):
    return rstd * _proj(xd, xhat)


def ln_bar(bar_xhat, g_xhat, xd, xhat, rstd):
    """(bar(x), G(x)) of x -> xhat given bar(xhat), G(xhat) and the input tangent xd."""
    h, t = g_xhat, xd
    pt, ph = _proj(t, xhat), _proj(h, xhat)
    second = -rstd**2 * (xhat * (h * pt).mean(1, keepdims=True) + ph * (xhat * t).mean(1, keepdims=True)
                         + pt * (h * xhat).mean(1, keepdims=True))
    return rstd * _proj(bar_xhat, xhat) + second, rstd * ph


class StagedTrainer:
    """dL/d(packed weights) for L with cotangents (gE [B], gF [N,3], gS [B,3,3]) of (e, F, sigma)."""

    def __init__(self, pw) -> None:
        self.pw = pw

    def W(self, name):
        return self.pw.get(name).astype(np.float64)

    # ---- gated MLP: forward with tangent, and the two-adjoint backward ----------------------------------
    def gated_fwd(self, z, zd, p, hidden):
        W, D = self.W, 64
        if hidden:
            H, Hd = silu(z), dsilu(z) * zd
            c = H[:, :D] @ W(p + "w2c").T + W(p + "b2c")
            g = H[:, D:] @ W(p + "w2g").T + W(p + "b2g")
            cd, gd = Hd[:, :D] @ W(p + "w2c").T, Hd[:, D:] @ W(p + "w2g").T
        else:
            H = Hd = None
            c, g, cd, gd = z[:, :D], z[:, D:], zd[:, :D], zd[:, D:]
        n1, xh1, r1 = ln_fwd(c, W(p + "ln1_g"), W(p + "ln1_b"))
        n2, xh2, r2 = ln_fwd(g, W(p + "ln2_g"), W(p + "ln2_b"))
        xh1d, xh2d = ln_tangent(cd, xh1, r1), ln_tangent(gd, xh2, r2)
        n1d, n2d = W(p + "ln1_g") * xh1d, W(p + "ln2_g") * xh2d
        a1, a2 = silu(n1), sigmoid(n2)
        a1d, a2d = dsilu(n1) * n1d, dsigmoid(n2) * n2d
        cache = dict(z=z, zd=zd, H=H, Hd=Hd, cd=cd, gd=gd, n1=n1, n2=n2, xh1=xh1, xh2=xh2, r1=r1, r2=r2, xh1d=xh1d, xh2d=xh2d,
                     n1d=n1d, n2d=n2d, a1=a1, a2=a2, a1d=a1d, a2d=a2d)
        return a1 * a2, a1d * a2 + a1 * a2d, cache

    def gated_bwd(self, bar_y, g_y, c, p, hidden, wg):
        W, D = self.W, 64
        bar_a1, bar_a2 = c["a2"] * bar_y + c["a2d"] * g_y, c["a1"] * bar_y + c["a1d"] * g_y
        g_a1, g_a2 = c["a2"] * g_y, c["a1"] * g_y
        bar_n1 = dsilu(c["n1"]) * bar_a1 + ddsilu(c["n1"]) * c["n1d"] * g_a1
        bar_n2 = dsigmoid(c["n2"]) * bar_a2 + ddsigmoid(c["n2"]) * c["n2d"] * g_a2
        g_n1, g_n2 = dsilu(c["n1"]) * g_a1, dsigmoid(c["n2"]) * g_a2
        for q, bar_n, g_n, xh, xhd in (("ln1", bar_n1, g_n1, c["xh1"], c["xh1d"]), ("ln2", bar_n2, g_n2, c["xh2"], c["xh2d"])):
            wg[p + q + "_g"] = (bar_n * xh + g_n * xhd).sum(0)
            wg[p + q + "_b"] = bar_n.sum(0)
        bar_c, g_c = ln_bar(W(p + "ln1_g") * bar_n1, W(p + "ln1_g") * g_n1, c["cd"], c["xh1"], c["r1"])
        bar_g, g_g = ln_bar(W(p + "ln2_g") * bar_n2, W(p + "ln2_g") * g_n2, c["gd"], c["xh2"], c["r2"])
        if not hidden:
            return np.concatenate([bar_c, bar_g], 1), np.concatenate([g_c, g_g], 1)
        H, Hd = c["H"], c["Hd"]
        wg[p + "w2c"], wg[p + "b2c"] = bar_c.T @ H[:, :D] + g_c.T @ Hd[:, :D], bar_c.sum(0)
        wg[p + "w2g"], wg[p + "b2g"] = bar_g.T @ H[:, D:] + g_g.T @ Hd[:, D:], bar_g.sum(0)
        bar_H = np.concatenate([bar_c @ W(p + "w2c"), bar_g @ W(p + "w2g")], 1)
        g_H = np.concatenate([g_c @ W(p + "w2c"), g_g @ W(p + "w2g")], 1)
        return dsilu(c["z"]) * bar_H + ddsilu(c["z"]) * c["zd"] * g_H, dsilu(c["z"]) * g_H

    # ---------------------------------------------------------------------------------------------------
    def run(self, pb, gE=None, gF=None, gS=None) -> dict:
        W, pw, dt, D = self.W, self.pw, np.float64, 64
        L = pw.n_conv
        B, N, Ed, Eu, A, Eb = pb.n_struct, pb.n_atoms, pb.n_directed, pb.n_undirected, pb.n_angles, pb.n_bnodes
        c, n, k = pb.e_center, pb.e_nbr, pb.e_d2u
        ctr, b1c, b2c, bn = pb.a_ctr, pb.a_b1c, pb.a_b2c, pb.bn_und
        n_at = np.diff(pb.atom_off).astype(dt)
        gE = np.zeros(B, dt) if gE is None else np.asarray(gE, dt)
        gF = np.zeros((N, 3), dt) if gF is None else np.asarray(gF, dt)
        gS = np.zeros((B, 3, 3), dt) if gS is None else np.asarray(gS, dt)

        # ---- geometry and the direction ---------------------------------------------------------------
        lat = pb.lattice.astype(dt)
        cart = np.einsum("ni,nij->nj", pb.frac.astype(dt), lat[pb.atom_owner])
        v = cart[c] - cart[n] - np.einsum("ei,eij->ej", pb.e_image.astype(dt), lat[pb.e_owner])
        r = np.sqrt((v * v).sum(1))
        u = v / r[:, None]
        vol = np.einsum("bi,bi->b", lat[:, 0], np.cross(lat[:, 1], lat[:, 2]))
        ux = -gF
        Wst = gS * (KAPPA_GPA / vol)[:, None, None]
        vd = ux[c] - ux[n] + np.einsum("ei,eij->ej", v, Wst[pb.e_owner])
        rd = (u * vd).sum(1)
        ud = (vd - u * rd[:, None]) / r[:, None]
        cot = (gE / (n_at if pw.is_intensive else 1.0))[pb.atom_owner]

        # ---- bases and embeddings, with tangents -------------------------------------------------------
        rk, rkd = r[pb.u_u2d], rd[pb.u_u2d]

        def rbf_all(rc, freq):
            rr = rk[:, None]
            cn, w = np.sqrt(2 / rc), freq[None, :] / rc
            sin, cos = np.sin(w * rr), np.cos(w * rr)
            env, denv = envelope(rr, rc, pw.cutoff_coeff)
            val = env * cn * sin / rr
            dr = denv * cn * sin / rr + env * cn * (w * cos / rr - sin / rr**2)
            df = env * cn * cos / rc
            drdf = cn / rc * (denv * cos - env * w * sin)
            return val, dr, df, drdf

        rbf6, dr6, df6, drdf6 = rbf_all(pw.atom_graph_cutoff, W("freq_ag"))
        rbf3, dr3, df3, drdf3 = rbf_all(pw.bond_graph_cutoff, W("freq_bg"))
        rbf6d, rbf3d = dr6 * rkd[:, None], dr3 * rkd[:, None]
        hb0, hb0d = rbf6 @ W("w_bond_emb").T, rbf6d @ W("w_bond_emb").T
        wag, wagd = rbf6 @ W("w_wag").T, rbf6d @ W("w_wag").T
        wbgc, wbgcd = (rbf3 @ W("w_wbg").T)[bn], (rbf3d @ W("w_wbg").T)[bn]
        kappa = 1 - 1e-6
        if A:
            cosv = (u[pb.a_d1] * u[pb.a_d2]).sum(1) * kappa
            cosd = ((ud[pb.a_d1] * u[pb.a_d2]).sum(1) + (u[pb.a_d1] * ud[pb.a_d2]).sum(1)) * kappa
            theta = np.arccos(cosv)
            thd = -cosd / np.sqrt(1 - cosv * cosv)
            fr = W("freq_ang")
            t = np.outer(theta, fr)
            isp = 1 / np.sqrt(np.pi)
            four = np.concatenate([np.full((A, 1), 1 / np.sqrt(2)), np.sin(t), np.cos(t)], 1) * isp
            dfour = np.concatenate([np.zeros((A, 1)), fr * np.cos(t), -fr * np.sin(t)], 1) * isp
            fourd = dfour * thd[:, None]
            ang, angd = [four @ W("w_ang_emb").T], [fourd @ W("w_ang_emb").T]
        atom, atomd = [W("emb")[pb.z - 1]], [np.zeros((N, D), dt)]
        hbc, hbcd = [hb0[bn]], [hb0d[bn]]

        def full(rows0, rows_l, l=None):    # bond features of all Eu bonds at some layer (nodes carry the layer's rows)
            h = rows0.copy()
            if l is not None and A:         # primal rows only: bonds outside the bond graph carry the mlp_out biases of the
                for m in range(l):          # earlier BondConv layers (0.2.0; layers.py:252-258 aggregates over ALL bonds) -- constants,
                    h = h + W(f"bc{m}.b_out")   # so the tangent rows are unchanged
            h[bn] = rows_l
            return h

        # ---- forward with tangent -----------------------------------------------------------------------
        ac, bc, au = {}, {}, {}

        def atom_conv(l):
            p = f"ac{l}."
            hb, hbd = full(hb0, hbc[l], l), full(hb0d, hbcd[l])
            P, Pd = atom[l] @ W(p + "w_cn").T, atomd[l] @ W(p + "w_cn").T
            P[:, :2 * D] += W(p + "b1")
            Q, Qd = hb @ W(p + "w_bond").T, hbd @ W(p + "w_bond").T
            z = P[c, :2 * D] + P[n, 2 * D:] + Q[k]
            zd = Pd[c, :2 * D] + Pd[n, 2 * D:] + Qd[k]
            y, yd, cache = self.gated_fwd(z, zd, p, True)
            m, md = y * wag[k], yd * wag[k] + y * wagd[k]
            agg, aggd = np.zeros((N, D), dt), np.zeros((N, D), dt)
            np.add.at(agg, c, m)
            np.add.at(aggd, c, md)
            ac[l] = dict(cache=cache, y=y, yd=yd, agg=agg, aggd=aggd, hb=hb, hbd=hbd)
            return agg @ W(p + "w_out").T + W(p + "b_out") + atom[l], aggd @ W(p + "w_out").T + atomd[l]

        def angle_z(p, hrows, hrowsd, atoms, atomsd, angs, angsd):
            S, Sd = atoms @ W(p + "w_ctr").T + W(p + "b1"), atomsd @ W(p + "w_ctr").T
            R, Rd = hrows @ W(p + "w_bij").T, hrowsd @ W(p + "w_bij").T
            z = R[b1c, :2 * D] + R[b2c, 2 * D:] + S[ctr] + angs @ W(p + "w_ang").T
            zd = Rd[b1c, :2 * D] + Rd[b2c, 2 * D:] + Sd[ctr] + angsd @ W(p + "w_ang").T
            return z, zd

        for l in range(L - 1):
            a_new, a_newd = atom_conv(l)
            atom.append(a_new)
            atomd.append(a_newd)
            if A:
                p = f"bc{l}."
                z, zd = angle_z(p, hbc[l], hbcd[l], atom[l + 1], atomd[l + 1], ang[l], angd[l])
                y, yd, cache = self.gated_fwd(z, zd, p, True)
                w1, w2, w1d, w2d = wbgc[b1c], wbgc[b2c], wbgcd[b1c], wbgcd[b2c]
                uu = y * w1 * w2
                uud = yd * w1 * w2 + y * w1d * w2 + y * w1 * w2d
                agg, aggd = np.zeros((Eb, D), dt), np.zeros((Eb, D), dt)
                np.add.at(agg, b1c, uu)
                np.add.at(aggd, b1c, uud)
                bc[l] = dict(cache=cache, y=y, yd=yd, agg=agg, aggd=aggd)
                hbc.append(agg @ W(p + "w_out").T + W(p + "b_out") + hbc[l])
                hbcd.append(aggd @ W(p + "w_out").T + hbcd[l])
                if l < L - 2:
                    p = f"au{l}."
                    z, zd = angle_z(p, hbc[l + 1], hbcd[l + 1], atom[l + 1], atomd[l + 1], ang[l], angd[l])
                    y, yd, cache = self.gated_fwd(z, zd, p, False)
                    au[l] = cache
                    ang.append(ang[l] + y)
                    angd.append(angd[l] + yd)
            else:
                hbc.append(hbc[l])
                hbcd.append(hbcd[l])
        a_new, a_newd = atom_conv(L - 1)
        atom.append(a_new)
        atomd.append(a_newd)

        x0, xh0, rs0 = ln_fwd(atom[L], W("ro_ln_g"), W("ro_ln_b"))
        xh0d = ln_tangent(atomd[L], xh0, rs0)
        x0d = W("ro_ln_g") * xh0d
        ls, lds, ss, sds = [], [], [x0], [x0d]
        nh = getattr(self.pw, "n_mlp_hidden", 3)                  # hidden layers of the energy head (2: the 0.2.0 architecture)
        for i in range(nh):
            li = ss[-1] @ W(f"mlp_w{i}").T + W(f"mlp_b{i}")
            lid = sds[-1] @ W(f"mlp_w{i}").T
            ls.append(li)
            lds.append(lid)
            ss.append(silu(li))
            sds.append(dsilu(li) * lid)
        site = ss[nh] @ W("mlp_w3") + W("mlp_b3")[0]
        sited = sds[nh] @ W("mlp_w3")
        out = {"site": site, "dE": float(sited.sum())}

        # ---- reverse sweep with two adjoints -------------------------------------------------------------
        wg = {}
        wg["mlp_w3"] = (cot[:, None] * ss[nh]).sum(0) + sds[nh].sum(0)
        wg["mlp_b3"] = np.array([cot.sum()])
        bar_s = cot[:, None] * W("mlp_w3")[None, :]
        g_s = np.ones((N, 1)) * W("mlp_w3")[None, :]
        for i in reversed(range(nh)):
            bar_l = dsilu(ls[i]) * bar_s + ddsilu(ls[i]) * lds[i] * g_s
            g_l = dsilu(ls[i]) * g_s
            wg[f"mlp_w{i}"] = bar_l.T @ ss[i] + g_l.T @ sds[i]
            wg[f"mlp_b{i}"] = bar_l.sum(0)
            bar_s, g_s = bar_l @ W(f"mlp_w{i}"), g_l @ W(f"mlp_w{i}")
        wg["ro_ln_g"] = (bar_s * xh0 + g_s * xh0d).sum(0)
        wg["ro_ln_b"] = bar_s.sum(0)
        bar_a, g_a = ln_bar(W("ro_ln_g") * bar_s, W("ro_ln_g") * g_s, atomd[L], xh0, rs0)    # adjoints of atom[L]
        bar_b, g_b = np.zeros((Eu, D), dt), np.zeros((Eu, D), dt)       # bond features (node rows double as hbc adjoints)
        bar_wag, g_wag = np.zeros((Eu, D), dt), np.zeros((Eu, D), dt)
        bar_wbg, g_wbg = np.zeros((Eb, D), dt), np.zeros((Eb, D), dt)
        bar_ang, g_ang = np.zeros((A, D), dt), np.zeros((A, D), dt)

        def atom_conv_bwd(l):
            nonlocal bar_a, g_a
            p = f"ac{l}."
            s = ac[l]
            wg[p + "w_out"] = bar_a.T @ s["agg"] + g_a.T @ s["aggd"]
            wg[p + "b_out"] = bar_a.sum(0)
            bar_m, g_m = (bar_a @ W(p + "w_out"))[c], (g_a @ W(p + "w_out"))[c]
            np.add.at(bar_wag, k, s["y"] * bar_m + s["yd"] * g_m)
            np.add.at(g_wag, k, s["y"] * g_m)
            bar_z, g_z = self.gated_bwd(wag[k] * bar_m + wagd[k] * g_m, wag[k] * g_m, s["cache"], p, True, wg)
            barP, gP = np.zeros((N, 4 * D), dt), np.zeros((N, 4 * D), dt)
            for dst, src in ((barP, bar_z), (gP, g_z)):
                np.add.at(dst[:, :2 * D], c, src)
                np.add.at(dst[:, 2 * D:], n, src)
            barQ, gQ = np.zeros((Eu, 2 * D), dt), np.zeros((Eu, 2 * D), dt)
            np.add.at(barQ, k, bar_z)
            np.add.at(gQ, k, g_z)
            wg[p + "w_cn"] = barP.T @ atom[l] + gP.T @ atomd[l]
            wg[p + "b1"] = barP[:, :2 * D].sum(0)
            wg[p + "w_bond"] = barQ.T @ s["hb"] + gQ.T @ s["hbd"]
            bar_a, g_a = bar_a + barP @ W(p + "w_cn"), g_a + gP @ W(p + "w_cn")
            bar_b[:] += barQ @ W(p + "w_bond")
            g_b[:] += gQ @ W(p + "w_bond")

        def angle_scatter(bar_z, g_z, p, hrows, hrowsd, atoms, atomsd, angs, angsd):
            nonlocal bar_a, g_a
            barR, gR = np.zeros((Eb, 4 * D), dt), np.zeros((Eb, 4 * D), dt)
            barS, gS_ = np.zeros((N, 2 * D), dt), np.zeros((N, 2 * D), dt)
            for dstR, dstS, src in ((barR, barS, bar_z), (gR, gS_, g_z)):
                np.add.at(dstR[:, :2 * D], b1c, src)
                np.add.at(dstR[:, 2 * D:], b2c, src)
                np.add.at(dstS, ctr, src)
            wg[p + "w_bij"] = barR.T @ hrows + gR.T @ hrowsd
            wg[p + "w_ctr"] = barS.T @ atoms + gS_.T @ atomsd
            wg[p + "b1"] = barS.sum(0)
            wg[p + "w_ang"] = bar_z.T @ angs + g_z.T @ angsd
            bar_b[bn] += barR @ W(p + "w_bij")
            g_b[bn] += gR @ W(p + "w_bij")
            bar_a, g_a = bar_a + barS @ W(p + "w_ctr"), g_a + gS_ @ W(p + "w_ctr")
            bar_ang[:] += bar_z @ W(p + "w_ang")
            g_ang[:] += g_z @ W(p + "w_ang")

        atom_conv_bwd(L - 1)
        for l in range(L - 2, -1, -1):
            if A:
                if l < L - 2:
                    p = f"au{l}."
                    bar_z, g_z = self.gated_bwd(bar_ang.copy(), g_ang.copy(), au[l], p, False, wg)
                    angle_scatter(bar_z, g_z, p, hbc[l + 1], hbcd[l + 1], atom[l + 1], atomd[l + 1], ang[l], angd[l])
                p = f"bc{l}."
                s = bc[l]
                wg[p + "w_out"] = bar_b[bn].T @ s["agg"] + g_b[bn].T @ s["aggd"]
                wg[p + "b_out"] = bar_b.sum(0)      # the bias reaches EVERY bond's layer-(l+1) features: column sum over all Eu rows
                bar_u, g_u = (bar_b[bn] @ W(p + "w_out"))[b1c], (g_b[bn] @ W(p + "w_out"))[b1c]
                w1, w2, w1d, w2d = wbgc[b1c], wbgc[b2c], wbgcd[b1c], wbgcd[b2c]
                y, yd = s["y"], s["yd"]
                np.add.at(bar_wbg, b1c, y * w2 * bar_u + (yd * w2 + y * w2d) * g_u)
                np.add.at(bar_wbg, b2c, y * w1 * bar_u + (yd * w1 + y * w1d) * g_u)
                np.add.at(g_wbg, b1c, y * w2 * g_u)
                np.add.at(g_wbg, b2c, y * w1 * g_u)
                bar_z, g_z = self.gated_bwd(w1 * w2 * bar_u + (w1d * w2 + w1 * w2d) * g_u, w1 * w2 * g_u, s["cache"], p, True, wg)
                angle_scatter(bar_z, g_z, p, hbc[l], hbcd[l], atom[l + 1], atomd[l + 1], ang[l], angd[l])
            atom_conv_bwd(l)

        # ---- embeddings, learnable frequencies -------------------------------------------------------------
        wg["emb"] = np.zeros((94, D), dt)
        np.add.at(wg["emb"], pb.z - 1, bar_a)
        bar_wbg_full, g_wbg_full = np.zeros((Eu, D), dt), np.zeros((Eu, D), dt)
        bar_wbg_full[bn], g_wbg_full[bn] = bar_wbg, g_wbg
        wg["w_bond_emb"] = bar_b.T @ rbf6 + g_b.T @ rbf6d
        wg["w_wag"] = bar_wag.T @ rbf6 + g_wag.T @ rbf6d
        wg["w_wbg"] = bar_wbg_full.T @ rbf3 + g_wbg_full.T @ rbf3d
        bar_rbf6 = bar_b @ W("w_bond_emb") + bar_wag @ W("w_wag")
        g_rbf6 = g_b @ W("w_bond_emb") + g_wag @ W("w_wag")
        bar_rbf3, g_rbf3 = bar_wbg_full @ W("w_wbg"), g_wbg_full @ W("w_wbg")
        wg["freq_ag"] = (bar_rbf6 * df6 + g_rbf6 * drdf6 * rkd[:, None]).sum(0)
        wg["freq_bg"] = (bar_rbf3 * df3 + g_rbf3 * drdf3 * rkd[:, None]).sum(0)
        if A:
            wg["w_ang_emb"] = bar_ang.T @ four + g_ang.T @ fourd
            bar_four, g_four = bar_ang @ W("w_ang_emb"), g_ang @ W("w_ang_emb")
            nf = len(fr)
            sn, cs, th = np.sin(t), np.cos(t), theta[:, None]
            wg["freq_ang"] = ((bar_four[:, 1:1 + nf] * th * cs - bar_four[:, 1 + nf:] * th * sn)
                              + (g_four[:, 1:1 + nf] * (cs - t * sn) + g_four[:, 1 + nf:] * (-sn - t * cs)) * thd[:, None]).sum(0) * isp
        out["wgrad"] = wg
        # the ordinary adjoint, continued to the geometry, reproduces the forces (sanity link to StagedModel)
        out["g_bond"] = g_b
        return out
