"""``CHGNetModule`` -- the engine behind the reference's *training* boundary: a ``torch.nn.Module`` whose ``forward`` returns tensors
that carry an autograd graph (reference chgnet/model/model.py:330-387), so that the reference's UNMODIFIED ``Trainer`` /
``CombinedLoss`` / torch optimizers / learning-rate schedulers (chgnet/trainer/trainer.py:140-231, 386-411) drive the HIP engine:

    model = CHGNetModule(CHGNet(state_dict=...))            # parameters named and shaped like the reference's state_dict()
    prediction = model(graphs, task="efsm")                   # e [B], f list of [n,3], s list of [3,3], m list of [n]: torch tensors
    CombinedLoss(...)(targets, prediction)["loss"].backward() # ONE chg_backward(_allreduce) call -> param.grad
    torch.optim.AdamW(model.parameters()).step()              # the next forward pushes the new values (chg_engine_update_weights)

torch is plumbing here: the parameters' master copy lives in host tensors (what the optimizers update), every number of the forward
and of the backward is computed by the engine.  The graph node of one forward collects the cotangents of e / f / s / m and hands
them to ``CHGNet.backward`` -- the first-order reverse sweep for energy / magmom terms, the tangent + two-adjoint sweep for force /
stress terms (model.py:517-535 ``create_graph=True`` in the reference).  Optional module: nothing else in the package imports it.
"""

from __future__ import annotations

import numpy as np
import torch
from torch import nn

from chgnet_amd import VALID_TASKS

_ATOMREF = "composition_model.fc.weight"


def _attach(root: nn.Module, dotted: str, value: torch.Tensor, *, parameter: bool, requires_grad: bool) -> None:
    """Register ``value`` under the dotted state_dict name, creating plain container modules on the way (the reference's module tree
    seen through its names only: ``atom_conv_layers.0.twoBody_atom.mlp_core.layers.0.weight`` ...)."""
    *path, leaf = dotted.split(".")
    mod = root
    for name in path:
        if name not in mod._modules:
            mod.add_module(name, nn.Module())
        mod = mod._modules[name]
    if parameter:
        mod.register_parameter(leaf, nn.Parameter(value, requires_grad=requires_grad))
    else:
        mod.register_buffer(leaf, value)


class _EngineSweep(torch.autograd.Function):
    """One batch through the engine.  Inputs: the parameters (so that autograd routes d loss / d parameter here); outputs: the flat
    e [B], f [N,3], s [B,3,3], m [N] of the batch (those the task asks for)."""

    @staticmethod
    def forward(ctx, module: "CHGNetModule", graphs, task: str, flags: dict, *params):  # noqa: ARG004
        res = module.core.forward(graphs, task=task, **flags)
        ctx.module, ctx.serial, ctx.keys = module, module._serial, tuple(k for k in "efsm" if k == "e" or k in res.flat)
        ctx.set_materialize_grads(False)          # an output the loss never touched arrives as None, not as a zero tensor
        module._last_result = res
        flat = {"e": res["e"], **res.flat}
        return tuple(torch.from_numpy(np.ascontiguousarray(flat[k])) for k in ctx.keys)

    @staticmethod
    def backward(ctx, *cotangents):
        module = ctx.module
        if ctx.serial != module._serial:
            raise RuntimeError("backward through a CHGNetModule output after a later forward(): the engine keeps the state of the "
                               "LAST batch only (call loss.backward() before the next model(...))")
        cot = {k: (None if c is None else c.detach().cpu().numpy()) for k, c in zip(ctx.keys, cotangents)}
        names = module._param_names
        if all(c is None for c in cot.values()):
            return (None, None, None, None, *[None] * len(names))
        grads = module.core.backward(cot.get("e"), cot.get("m"), cot.get("f"), cot.get("s"), comm=module.comm)
        if module.comm is not None and getattr(module.comm, "world", 1) > 1 and module.average_gradients:
            grads = {k: v / module.comm.world for k, v in grads.items()}
        if _ATOMREF in grads and cot.get("e") is not None and module._atomref_trainable():
            grads[_ATOMREF] = module._atomref_gradient(cot["e"])
        out = []
        for i, name in enumerate(names):
            need = ctx.needs_input_grad[4 + i]
            out.append(torch.from_numpy(np.ascontiguousarray(grads[name], np.float32)).reshape(module._shapes[name]) if need else None)
        return (None, None, None, None, *out)


class CHGNetModule(nn.Module):
    """``chgnet_amd.CHGNet`` with the reference model's torch face.

    ``core``: a ``chgnet_amd.CHGNet`` (anything with its ``state_dict / model_args / forward / backward / load_state_dict``).
    ``comm``: an ``RcclComm`` for data-parallel fine-tuning -- the gradient blob is summed over the ranks on the device inside the
    backward (and divided by the world size when ``average_gradients``, DistributedDataParallel's convention)."""

    def __init__(self, core, comm=None, average_gradients: bool = True) -> None:
        super().__init__()
        object.__setattr__(self, "core", core)          # not a torch module: kept out of _modules
        self.comm, self.average_gradients = comm, average_gradients
        learnable_rbf = bool(getattr(core, "model_args", {}).get("learnable_rbf", True))
        self._param_names, self._shapes = [], {}
        for name, value in core.state_dict().items():
            t = torch.from_numpy(np.array(value, dtype=np.float32, copy=True))
            is_freq = name.endswith(".frequencies")
            if is_freq and not learnable_rbf:          # buffers in the reference (basis.py:31-40, 87-98)
                _attach(self, name, t, parameter=False, requires_grad=False)
                continue
            _attach(self, name, t, parameter=True, requires_grad=name != _ATOMREF)     # AtomRef frozen: model.py:179-182
            self._param_names.append(name)
            self._shapes[name] = tuple(t.shape)
        self._serial = 0
        self._out_device = None
        self._pushed = self._versions()
        self._last_result = None

    # ---- the reference model's attributes the Trainer touches -------------------------------------------
    @property
    def n_params(self) -> int:
        return int(sum(p.numel() for p in self.parameters()))

    @property
    def graph_converter(self):
        return self.core.graph_converter

    @property
    def model_args(self) -> dict:
        return self.core.model_args

    def as_dict(self) -> dict:
        """model.py:667-672: what ``Trainer.save`` writes under "model"."""
        return {"state_dict": self.state_dict(), "model_args": self.core.model_args}

    def todict(self) -> dict:
        return {"model_name": "CHGNet", "model_args": self.core.model_args}

    def predict_structure(self, *args, **kwargs):
        self._push_weights()
        return self.core.predict_structure(*args, **kwargs)

    def predict_graph(self, *args, **kwargs):
        self._push_weights()
        return self.core.predict_graph(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        """``.to(device)`` / ``.cuda()`` (Trainer.train, trainer.py:296): the parameters' master copy stays on the host -- the engine owns
        the device copy -- and only the device the OUTPUTS are handed out on follows, so that a loss against labels the Trainer moved to
        "cuda" finds its predictions there."""
        probe = fn(torch.zeros(1))
        if probe.dtype != torch.float32:
            raise TypeError("CHGNetModule keeps float32 parameters (the engine computes in fp32)")
        self._out_device = None if probe.device.type == "cpu" else probe.device
        return self

    # ---- parameters <-> engine -----------------------------------------------------------------------------
    def _named(self) -> dict:
        return dict(self.state_dict(keep_vars=True))

    def _versions(self) -> tuple:
        sd = self._named()
        return tuple((id(sd[n]), sd[n]._version) for n in sd)

    def _push_weights(self) -> None:
        """New parameter values (an optimizer step, ``load_state_dict``) reach the engine before the next sweep: one repack +
        ``chg_engine_update_weights``, skipped when no tensor was written since the last push (tensor version counters)."""
        now = self._versions()
        if now != self._pushed:
            self.core.load_state_dict({k: v.detach().cpu().numpy() for k, v in self._named().items()})
            self._pushed = now

    def _atomref_trainable(self) -> bool:
        return bool(self._named()[_ATOMREF].requires_grad)

    def _atomref_gradient(self, e_cot: np.ndarray) -> np.ndarray:
        """``Trainer.train(train_composition_model=True)`` (trainer.py:299-300): the energy is linear in the AtomRef row,
        e_b += sum_i w[z_i] (/ n_b when intensive: composition_model.py:98-117), so its gradient is a host-side histogram."""
        res = self._last_result
        n_at = np.asarray(res["atoms_per_graph"], np.int64)
        z = np.asarray(res.atomic_numbers, np.int64)
        per_atom = np.repeat(np.asarray(e_cot, np.float64) / (n_at if self.core.is_intensive else 1.0), n_at)
        g = np.zeros(self._shapes[_ATOMREF], np.float64)
        np.add.at(g[0], z - 1, per_atom)
        return g.astype(np.float32)

    # ---- forward (model.py:330-387) ---------------------------------------------------------------------------
    def forward(self, graphs, *, task: str = "e", return_site_energies: bool = False, return_atom_feas: bool = False,
                return_crystal_feas: bool = False) -> dict:
        """The reference's batch dictionary as torch tensors: ``atoms_per_graph`` int64 [B], ``e`` [B], and by task ``f`` list of
        [n,3], ``s`` list of [3,3], ``m`` list of [n] (+ the optional ``site_energies`` / ``atom_fea`` / ``crystal_fea``, without a
        graph).  With gradients enabled the e / f / s / m tensors hang off one graph node whose backward is the engine's."""
        if task not in VALID_TASKS:
            raise ValueError(f"Invalid {task=}. Must be one of {VALID_TASKS}.")
        self._push_weights()
        self._serial += 1
        flags = dict(return_site_energies=return_site_energies, return_atom_feas=return_atom_feas, return_crystal_feas=return_crystal_feas)
        named = self._named()
        params = [named[n] for n in self._param_names]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            flat = dict(zip("efsm", [None] * 4))
            outs = _EngineSweep.apply(self, graphs, task, flags, *params)
            res = self._last_result
            for k, t in zip([k for k in "efsm" if k == "e" or k in res.flat], outs):
                flat[k] = t
        else:
            res = self.core.forward(graphs, task=task, **flags)
            self._last_result = res
            flat = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in {"e": res["e"], **res.flat}.items()}
        if not hasattr(res, "atomic_numbers"):        # for the AtomRef gradient: the batch's atomic numbers, in batch order
            res.atomic_numbers = getattr(getattr(self.core, "_fwd_batch", None), "packed", None)
            res.atomic_numbers = res.atomic_numbers.z if res.atomic_numbers is not None else None
        if self._out_device is not None:              # (outside the graph node: autograd brings the cotangents back to the host)
            flat = {k: (t.to(self._out_device) if t is not None else None) for k, t in flat.items()}
        sizes = [int(n) for n in res["atoms_per_graph"]]
        out: dict = {"atoms_per_graph": torch.as_tensor(np.asarray(res["atoms_per_graph"], np.int64)), "e": flat["e"]}
        if flat.get("f") is not None:
            out["f"] = list(torch.split(flat["f"].reshape(-1, 3), sizes))
        if flat.get("s") is not None:
            out["s"] = list(flat["s"].reshape(-1, 3, 3).unbind(0))
        if flat.get("m") is not None:
            out["m"] = list(torch.split(flat["m"].reshape(-1), sizes))
        for key in ("site_energies", "atom_fea"):
            if key in res:
                out[key] = [torch.from_numpy(np.ascontiguousarray(a)) for a in res[key]]
        if "crystal_fea" in res:
            out["crystal_fea"] = torch.from_numpy(np.ascontiguousarray(res["crystal_fea"]))
        return out
