"""``chgnet_amd.torch_bridge.CHGNetModule``: the engine as a torch module whose outputs carry an autograd graph (reference
chgnet/model/model.py:330-387), driven by the reference's UNMODIFIED Trainer / CombinedLoss / torch optimizers / schedulers
(chgnet/trainer/trainer.py:140-231, 386-411).

CPU: plumbing against a stand-in core whose outputs are a known linear function of the parameters (no engine).
GPU: the reference's loss.backward() fixtures (tests/golden/grad_*.npz) through the bridge + torch optimizers."""

from __future__ import annotations

import os

import numpy as np
import pytest
import torch

from chgnet_amd.model import ForwardResult
from chgnet_amd.torch_bridge import CHGNetModule
from conftest import GOLDEN, load_case

REF = os.path.isdir("/root/reference/chgnet")
NAMES = ("composition_model.fc.weight", "atom_embedding.embedding.weight", "atom_conv_layers.0.twoBody_atom.mlp_core.layers.0.weight",
         "atom_conv_layers.0.twoBody_atom.mlp_core.layers.0.bias", "mlp.layers.7.weight")
SHAPES = ((1, 94), (94, 4), (4, 12), (4,), (1, 4))


class LinearCore:
    """Stand-in for ``chgnet_amd.CHGNet``: every output is J . theta for a fixed matrix per (batch size, output), so parameter
    gradients are J^T . cotangent.  Same surface as the model: state_dict / model_args / forward / backward / load_state_dict."""

    is_intensive = True
    model_args = {"learnable_rbf": True}
    graph_converter = None

    def __init__(self, seed: int = 0) -> None:
        rng = np.random.default_rng(seed)
        self.sd = {n: rng.normal(size=s).astype(np.float32) for n, s in zip(NAMES, SHAPES)}
        self.n_theta = sum(int(np.prod(s)) for n, s in zip(NAMES, SHAPES) if n != NAMES[0])
        self.loads = 0
        self._fwd = None

    def state_dict(self):
        return self.sd

    def load_state_dict(self, sd):
        assert set(sd) == set(self.sd)
        self.sd = {k: np.asarray(v, np.float32).copy() for k, v in sd.items()}
        self.loads += 1

    def theta(self):
        return np.concatenate([self.sd[n].reshape(-1) for n in NAMES[1:]]).astype(np.float64)

    @staticmethod
    def jac(n_at, key, n_theta):
        rows = {"e": len(n_at), "f": 3 * sum(n_at), "s": 9 * len(n_at), "m": sum(n_at)}[key]
        return np.random.default_rng([sum(n_at), "efsm".index(key)]).normal(size=(rows, n_theta)) / 8

    def forward(self, graphs, *, task="e", **flags):  # noqa: ARG002
        n_at = [len(g.atomic_number) for g in graphs]
        th = self.theta()
        z = np.concatenate([np.asarray(getattr(g.atomic_number, "numpy", lambda: g.atomic_number)()) for g in graphs])
        ref = np.array([self.sd[NAMES[0]][0, np.asarray(zz) - 1].mean() for zz in np.split(z, np.cumsum(n_at)[:-1])])
        flat = {k: (self.jac(n_at, k, self.n_theta) @ th).astype(np.float32) for k in task}
        flat["e"] = (flat["e"] + ref).astype(np.float32)
        off = np.concatenate([[0], np.cumsum(n_at)])
        out = ForwardResult({"atoms_per_graph": np.asarray(n_at, np.int64), "e": flat["e"]})
        if "f" in flat:
            flat["f"] = flat["f"].reshape(-1, 3)
            out["f"] = [flat["f"][off[i]:off[i + 1]] for i in range(len(n_at))]
        if "s" in flat:
            flat["s"] = flat["s"].reshape(-1, 3, 3)
            out["s"] = list(flat["s"])
        if "m" in flat:
            out["m"] = [flat["m"][off[i]:off[i + 1]] for i in range(len(n_at))]
        out.flat = {k: flat[k] for k in "fsm" if k in flat}
        out.atomic_numbers = z
        self._fwd = n_at
        return out

    def backward(self, e_grad=None, m_grad=None, f_grad=None, s_grad=None, comm=None):  # noqa: ARG002
        g = np.zeros(self.n_theta)
        for key, cot in (("e", e_grad), ("m", m_grad), ("f", f_grad), ("s", s_grad)):
            if cot is not None:
                g += self.jac(self._fwd, key, self.n_theta).T @ np.asarray(cot, np.float64).reshape(-1)
        out, pos = {NAMES[0]: np.zeros(SHAPES[0], np.float32)}, 0
        for n, s in zip(NAMES[1:], SHAPES[1:]):
            k = int(np.prod(s))
            out[n] = g[pos:pos + k].reshape(s).astype(np.float32)
            pos += k
        return out


class TorchLinear(torch.nn.Module):
    """The same function as ``LinearCore`` written in torch (autograd does the backward): the comparison model."""

    def __init__(self, core: LinearCore) -> None:
        super().__init__()
        self.core = [core]
        self.p = torch.nn.ParameterList([torch.nn.Parameter(torch.tensor(core.sd[n])) for n in NAMES])
        self.p[0].requires_grad = False

    def forward(self, graphs, *, task="e"):
        core = self.core[0]
        n_at = [len(g.atomic_number) for g in graphs]
        th = torch.cat([p.reshape(-1) for p in list(self.p)[1:]]).double()
        out = {"atoms_per_graph": torch.tensor(n_at)}
        ref = torch.stack([self.p[0][0, torch.as_tensor(np.asarray(g.atomic_number)).long() - 1].mean() for g in graphs])
        for k in task:
            v = (torch.tensor(core.jac(n_at, k, core.n_theta)) @ th).float()
            if k == "e":
                out["e"] = v + ref
            elif k == "f":
                out["f"] = list(torch.split(v.reshape(-1, 3), n_at))
            elif k == "s":
                out["s"] = list(v.reshape(-1, 3, 3).unbind(0))
            else:
                out["m"] = list(torch.split(v, n_at))
        return out


class G:
    def __init__(self, z):
        self.atomic_number = np.asarray(z, np.int32)


def _loss(out, rng):
    return (out["e"] * torch.tensor(rng.normal(size=len(out["e"])), dtype=torch.float32)).sum() + sum((f ** 2).sum() for f in out["f"]) \
        + sum((s * 0.3).sum() for s in out["s"]) + sum(m.abs().sum() for m in out["m"])


def test_state_dict_names_parameters_and_frozen_atomref():
    core = LinearCore()
    mod = CHGNetModule(core)
    assert list(mod.state_dict()) == list(NAMES)
    assert {n for n, _ in mod.named_parameters()} == set(NAMES)
    assert not dict(mod.named_parameters())[NAMES[0]].requires_grad
    assert all(p.requires_grad for n, p in mod.named_parameters() if n != NAMES[0])
    assert [p.requires_grad for p in mod.composition_model.parameters()] == [False]      # trainer.py:299-300 walks this sub-module
    assert mod.n_params == sum(int(np.prod(s)) for s in SHAPES)
    assert set(mod.as_dict()) == {"state_dict", "model_args"}


def test_backward_collects_all_cotangents_into_one_engine_call_and_matches_autograd():
    core = LinearCore()
    mod, twin = CHGNetModule(core), TorchLinear(core)
    graphs = [G([3, 8, 8]), G([27, 8]), G([3, 3, 25, 8])]
    calls = []
    orig = core.backward
    core.backward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    _loss(mod(graphs, task="efsm"), np.random.default_rng(5)).backward()
    _loss(twin(graphs, task="efsm"), np.random.default_rng(5)).backward()
    assert len(calls) == 1
    got = dict(mod.named_parameters())
    for n, p in zip(NAMES, twin.p):
        if n == NAMES[0]:
            assert got[n].grad is None
            continue
        assert torch.allclose(got[n].grad, p.grad, atol=1e-5, rtol=1e-5), n
    # outputs the loss never touched reach the core as None (no second-order sweep for an energy-only loss)
    seen = {}
    core.backward = lambda e=None, m=None, f=None, s=None, comm=None: (seen.update(e=e, m=m, f=f, s=s), orig(e, m, f, s))[1]
    mod.zero_grad()
    mod(graphs, task="efsm")["e"].sum().backward()
    assert seen["e"] is not None and seen["f"] is None and seen["s"] is None and seen["m"] is None


def test_stale_graph_and_no_grad_paths():
    core = LinearCore()
    mod = CHGNetModule(core)
    graphs = [G([3, 8]), G([8])]
    first = mod(graphs, task="ef")
    mod(graphs, task="ef")
    with pytest.raises(RuntimeError, match="LAST batch"):
        first["e"].sum().backward()
    with torch.no_grad():
        out = mod(graphs, task="efsm")
    assert out["e"].grad_fn is None and not out["f"][0].requires_grad
    with pytest.raises(ValueError, match="Invalid task"):
        mod(graphs, task="x")


def test_optimizer_steps_reach_the_core_once_per_change():
    core = LinearCore()
    mod = CHGNetModule(core)
    graphs = [G([3, 8]), G([8, 8, 27])]
    opt = torch.optim.AdamW(mod.parameters(), 1e-2)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=1e-4)
    before = {k: v.copy() for k, v in core.sd.items()}
    mod(graphs, task="e")["e"].sum().backward()
    opt.step(); sched.step()
    assert core.loads == 0                      # nothing pushed until the next sweep needs the values
    mod(graphs, task="e")
    assert core.loads == 1
    mod(graphs, task="e")
    assert core.loads == 1                      # unchanged parameters are not packed again
    assert np.array_equal(core.sd[NAMES[0]], before[NAMES[0]])       # frozen AtomRef
    assert not np.array_equal(core.sd[NAMES[1]], before[NAMES[1]])
    assert np.allclose(core.sd[NAMES[1]], dict(mod.named_parameters())[NAMES[1]].detach().numpy())


def test_trainable_atomref_gradient_is_the_composition_histogram():
    core = LinearCore()
    mod, twin = CHGNetModule(core), TorchLinear(core)
    for m in (mod.composition_model, ):
        for p in m.parameters():
            p.requires_grad = True                        # Trainer.train(train_composition_model=True)
    twin.p[0].requires_grad = True
    graphs = [G([3, 8, 8]), G([27, 8])]
    w = torch.tensor([0.7, -1.9])
    (mod(graphs, task="e")["e"] * w).sum().backward()
    (twin(graphs, task="e")["e"] * w).sum().backward()
    assert torch.allclose(dict(mod.named_parameters())[NAMES[0]].grad, twin.p[0].grad, atol=1e-6)


@pytest.mark.skipif(not REF, reason="live reference only in the build container")
@pytest.mark.parametrize("optimizer,scheduler", [("AdamW", "CosLR"), ("SGD", "ExponentialLR")])
def test_the_unmodified_reference_trainer_drives_the_bridge(tmp_path, optimizer, scheduler):
    """chgnet.trainer.Trainer (its optimizers, schedulers, CombinedLoss, checkpoint writer) over the bridge == over the torch twin."""
    from oracle._refimport import load_reference

    load_reference(fast_graph=True)
    import sys

    sys.path.insert(0, os.path.join(GOLDEN))
    from chgnet.graph.crystalgraph import CrystalGraph as RefGraph
    from chgnet.trainer.trainer import Trainer

    def ref_graph(name):
        d = np.load(os.path.join(GOLDEN, f"case_{name}.npz"))
        return RefGraph(atomic_number=torch.tensor(d["atomic_number"], dtype=torch.int32), atom_frac_coord=torch.tensor(d["atom_frac_coord"], dtype=torch.float32),
                        atom_graph=torch.tensor(d["atom_graph"], dtype=torch.int32), neighbor_image=torch.tensor(d["neighbor_image"], dtype=torch.float32),
                        directed2undirected=torch.tensor(d["directed2undirected"], dtype=torch.int32),
                        undirected2directed=torch.tensor(d["undirected2directed"], dtype=torch.int32),
                        bond_graph=torch.tensor(d["bond_graph"].reshape(-1, 5), dtype=torch.int32), lattice=torch.tensor(d["lattice"], dtype=torch.float32),
                        atom_graph_cutoff=6, bond_graph_cutoff=3)

    rng = np.random.default_rng(11)
    batches = []
    for names in (("limno2", "noangle"), ("s16tri",), ("limno2", "s16tri", "noangle")):
        graphs = [ref_graph(n) for n in names]
        n_at = [len(g.atomic_number) for g in graphs]
        targets = {"e": torch.tensor(rng.normal(size=len(n_at)), dtype=torch.float32), "f": [torch.tensor(rng.normal(size=(n, 3)), dtype=torch.float32) for n in n_at],
                   "s": [torch.tensor(rng.normal(size=(3, 3)), dtype=torch.float32) for _ in n_at],
                   "m": [torch.tensor(np.abs(rng.normal(size=n)), dtype=torch.float32) for n in n_at]}
        targets["m"][0] = None
        batches.append((graphs, targets))

    finals = []
    for make in (lambda c: CHGNetModule(c), TorchLinear):
        core = LinearCore(seed=4)
        model = make(core)
        if isinstance(model, TorchLinear):          # the attributes Trainer.train / save touch on a CHGNet
            model.composition_model = torch.nn.Module()
            model.as_dict = lambda m=model: {"state_dict": m.state_dict(), "model_args": {}}
        trainer = Trainer(model=model, targets="efsm", optimizer=optimizer, scheduler=scheduler, criterion="Huber", epochs=2, learning_rate=1e-2,
                          use_device="cpu", print_freq=100)
        trainer.train(batches, batches[:1], save_dir=str(tmp_path / type(model).__name__))
        finals.append([p.detach().clone() for _, p in sorted(model.named_parameters())] if isinstance(model, CHGNetModule)
                      else [p.detach().clone() for p in model.p])
        hist = trainer.training_history
        assert len(hist["e"]["train"]) == 2 and np.isfinite(hist["f"]["val"]).all()
    by_name = dict(zip(sorted(NAMES), finals[0]))
    for n, want in zip(NAMES, finals[1]):
        assert torch.allclose(by_name[n], want, atol=2e-5, rtol=1e-4), n
    saved = [f for f in os.listdir(tmp_path / "CHGNetModule") if f.startswith("epoch")]
    assert saved and set(torch.load(tmp_path / "CHGNetModule" / saved[0], weights_only=False)["model"]["state_dict"]) == set(NAMES)


# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("fname,which", [("grad_five_seed0.npz", "seed0"), ("grad_v020_five.npz", "v020")])
def test_bridge_gradients_equal_the_reference_backward_fixtures(golden_weights, fname, which):
    """loss.backward() on the bridge's outputs fills param.grad with what the UNMODIFIED reference's backward left there
    (tests/golden/make_grad_golden.py, make_golden_v020.py), then torch.optim.AdamW + CosineAnnealingLR take a step and the
    engine predicts with the new weights."""
    from chgnet_amd.model import CHGNet

    d = np.load(os.path.join(GOLDEN, fname))
    want = {k[len("grad/"):]: d[k] for k in d.files if k.startswith("grad/")}
    if which == "v020":
        from test_v020 import V020_ARGS, load_case_v020

        weights = dict(np.load(os.path.join(GOLDEN, "weights_v020.npz")))
        core = CHGNet(state_dict=weights, **V020_ARGS)
        graphs = [load_case_v020(str(n))[0] for n in d["order"]]
    else:
        core = CHGNet(state_dict=golden_weights)
        graphs = [load_case(str(n))[0] for n in d["order"]]
    mod = CHGNetModule(core)
    try:
        out = mod(graphs, task="efsm")
        assert out["e"].grad_fn is not None and out["f"][0].grad_fn is not None
        loss = (out["e"] * torch.tensor(d["cot_e"])).sum() + (torch.cat(out["m"]) * torch.tensor(d["cot_m"])).sum() \
            + (torch.cat(out["f"]) * torch.tensor(d["cot_f"])).sum() + (torch.stack(out["s"]) * torch.tensor(d["cot_s"])).sum()
        assert abs(float(loss.detach()) - float(d["loss"])) <= 2e-4 * max(1.0, abs(float(d["loss"])))
        opt = torch.optim.AdamW(mod.parameters(), 1e-3)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=1e-5)
        opt.zero_grad()
        loss.backward()
        got = {n: p.grad for n, p in mod.named_parameters()}
        msgs, checked = [], 0
        for k, ref in want.items():
            if k.startswith("composition_model"):
                assert got[k] is None
                continue
            if not np.any(ref):
                assert got[k] is None or not torch.any(got[k]), k
                continue
            scale, err = float(np.abs(ref).max()), float(np.abs(got[k].numpy() - ref).max())
            checked += 1
            if not err <= 3.2e-4 * scale:
                msgs.append(f"{k}: {err:.3e} / {scale:.3e}")
        assert not msgs, "; ".join(msgs)
        assert checked >= 120
        e0 = out["e"].detach().clone()
        opt.step(); sched.step()
        with torch.no_grad():
            e1 = mod(graphs, task="e")["e"]
        assert torch.isfinite(e1).all() and not torch.allclose(e0, e1)
        again = core.predict_graph(graphs, task="e")
        assert np.allclose([p["e"] for p in again], e1.numpy(), atol=1e-5)
    finally:
        core.release_forward_state()
        if core._engine is not None:
            core._engine.close()
