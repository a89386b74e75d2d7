"""Fine-tuning on the HIP engine: ``CombinedLoss`` and a data-parallel train step.

Mirrors the part of the reference's training loop that sits on the hot path
(chgnet/trainer/trainer.py): ``CombinedLoss.forward`` (719-869: criterion choice, loss ratios, NaN masking of
missing labels, MAE bookkeeping), one optimisation step (``_train`` 386-411: forward, loss, ``loss.backward()``,
``optimizer.step()``) and -- not in the reference, which is single-device -- the gradient all-reduce of a
one-process-per-GPU data-parallel run (RCCL over xGMI through ``torch.distributed``; 412,525 float32 = 1.65 MB per
step).  Datasets, schedulers, checkpoint bookkeeping and logging are out of scope.

The device differentiates all four terms: energy and magmom by a first-order reverse sweep (fused kernels), forces and
stress -- first derivatives of the energy themselves -- by one tangent sweep plus a reverse sweep with two adjoints per
activation (``chg_backward``; derivation and float64 check in the tests' pipeline model).
"""

from __future__ import annotations

import numpy as np


def _criterion(name: str, delta: float):
    """(value, d value / d prediction) of the mean-reduced criterion over 1-D arrays (torch.nn.MSELoss / L1Loss /
    HuberLoss with the default ``reduction="mean"``)."""
    if name in {"MSE", "mse"}:
        return lambda t, p: (float(np.mean((p - t) ** 2)), 2.0 * (p - t) / p.size)
    if name in {"MAE", "mae", "l1"}:
        return lambda t, p: (float(np.mean(np.abs(p - t))), np.sign(p - t) / p.size)
    if name == "Huber":
        def huber(t, p):
            d = p - t
            small = np.abs(d) <= delta
            val = np.where(small, 0.5 * d * d, delta * (np.abs(d) - 0.5 * delta))
            return float(np.mean(val)), np.where(small, d, delta * np.sign(d)) / p.size
        return huber
    raise NotImplementedError(name)


class CombinedLoss:
    """Energy / force / stress / magmom loss with the reference's semantics (trainer.py:719-869).

    ``forward(targets, prediction)`` takes the reference's dictionaries -- ``e`` [B], ``f`` list of [n,3], ``s`` list
    of [3,3], ``m`` list of [n] or None -- as numpy arrays and returns ``{"loss", "<k>_MAE", "<k>_MAE_size"}``;
    ``gradients`` additionally returns d loss / d prediction for every term (what autograd hands back to the model)."""

    def __init__(self, *, target_str: str = "ef", criterion: str = "MSE", energy_loss_ratio: float = 1, force_loss_ratio: float = 1,
                 stress_loss_ratio: float = 0.1, mag_loss_ratio: float = 0.1, delta: float = 0.1, allow_missing_labels: bool = True) -> None:
        self.criterion = _criterion(criterion, delta)
        self.target_str = target_str
        self.energy_loss_ratio = energy_loss_ratio
        self.force_loss_ratio = force_loss_ratio if "f" in target_str else 0
        self.stress_loss_ratio = stress_loss_ratio if "s" in target_str else 0
        self.mag_loss_ratio = mag_loss_ratio if "m" in target_str else 0
        self.allow_missing_labels = allow_missing_labels

    def gradients(self, targets: dict, prediction: dict, flat_targets: dict | None = None) -> tuple[dict, dict]:
        """``flat_targets``: the result of ``_flat_targets(targets, atoms_per_graph)`` when the caller already holds it (the loader
        thread of ``TrainStep.run_epoch`` flattens the labels of the next batch while the device works): the per-step validity stamp
        of the label cache -- a probe of every label array, 4 of the 6 ms of this call at 1024 structures -- is then skipped."""
        out: dict = {"loss": 0.0}
        grads: dict = {}

        def term(key, ratio, t, p):
            t, p = np.asarray(t, np.float64), np.asarray(p, np.float64)
            if t.size and not np.isnan(t.sum()):       # no label missing (one pass; inf - inf lands in the general path below): the
                tv, pv = t.reshape(-1), p.reshape(-1)  # same 1-D operands as t[valid] / p[valid] without the mask and the copies
                val, gv = self.criterion(tv, pv)
                out["loss"] += ratio * val
                out[f"{key}_MAE"] = float(np.mean(np.abs(tv - pv)))
                return (ratio * gv).reshape(p.shape), int(t.size) if self.allow_missing_labels else int(t.shape[0])
            valid = ~np.isnan(t) if self.allow_missing_labels else np.ones(t.shape, bool)
            g = np.zeros(p.shape, np.float64)
            if valid.any():
                val, gv = self.criterion(t[valid], p[valid])
                out["loss"] += ratio * val
                g[valid] = ratio * gv
                out[f"{key}_MAE"] = float(np.mean(np.abs(t[valid] - p[valid])))
            else:   # torch: mean over an empty tensor is nan; the reference would propagate it
                out["loss"] += float("nan")
                out[f"{key}_MAE"] = float("nan")
            # trainer.py:783-812: with masking the sizes count label ELEMENTS, without it the rows of the [N,3] / [3B,3] targets
            return g, int(valid.sum()) if self.allow_missing_labels else int(t.shape[0])

        # Flat fast path: CHGNet.forward hands over the engine's batch arrays next to the per-structure views (``.flat``), and the
        # label lists of a batch are flattened once and kept with the label dictionary -- 1024-element lists re-concatenated every
        # step were most of this function's time in a training loop.  Same arithmetic, same results (tests/test_trainer_cpu.py).
        pflat = getattr(prediction, "flat", None)
        tflat = (flat_targets if flat_targets is not None else self._flat_targets(targets, prediction["atoms_per_graph"])) if pflat is not None else None
        if "e" in self.target_str:
            grads["e"], _ = term("e", self.energy_loss_ratio, targets["e"], prediction["e"])
            out["e_MAE_size"] = int(np.asarray(prediction["e"]).shape[0])
        if "f" in self.target_str:
            tf = tflat["f"] if tflat is not None else np.concatenate(targets["f"], 0)
            pf = pflat["f"].reshape(-1, 3) if pflat is not None else np.concatenate(prediction["f"], 0)
            g, n = term("f", self.force_loss_ratio, tf, pf)
            grads["f"], out["f_MAE_size"] = g, n
        if "s" in self.target_str:
            ts = tflat["s"] if tflat is not None else np.concatenate(targets["s"], 0)
            ps = pflat["s"].reshape(-1, 3) if pflat is not None else np.concatenate(prediction["s"], 0)
            g, n = term("s", self.stress_loss_ratio, ts, ps)
            grads["s"], out["s_MAE_size"] = g.reshape(-1, 3, 3), n
        if "m" in self.target_str and tflat is not None and "m" in pflat:
            # atoms of the structures whose magmom labels count (trainer.py:846: a missing or partly-NaN label drops the structure)
            keep_atoms, tm = tflat["m_keep_atoms"], tflat["m"]
            pm = np.asarray(pflat["m"], np.float64).reshape(-1)
            gm_flat = np.zeros(pm.shape[0], np.float64)
            if tm.size:
                sel = pm[keep_atoms] if keep_atoms is not None else pm
                val, gv = self.criterion(tm, sel)
                out["loss"] += self.mag_loss_ratio * val
                out["m_MAE"] = float(np.mean(np.abs(tm - sel)))
                if keep_atoms is not None:
                    gm_flat[keep_atoms] = self.mag_loss_ratio * gv
                else:
                    gm_flat = self.mag_loss_ratio * gv
            else:
                out["m_MAE"] = 0.0
            out["m_MAE_size"] = int(tflat["m_size"])
            grads["m"] = gm_flat
        elif "m" in self.target_str:
            preds, targs, keep, size = [], [], [], 0
            for mp, mt in zip(prediction["m"], targets["m"], strict=True):
                ok = (mt is not None and not np.isnan(np.asarray(mt, np.float64)).any()) if self.allow_missing_labels else True
                keep.append(ok)
                if ok:
                    preds.append(np.asarray(mp, np.float64))
                    targs.append(np.asarray(mt, np.float64))
                    size += len(mt)
            gm = [np.zeros(len(mp), np.float64) for mp in prediction["m"]]
            if targs:
                val, gv = self.criterion(np.concatenate(targs), np.concatenate(preds))
                out["loss"] += self.mag_loss_ratio * val
                out["m_MAE"] = float(np.mean(np.abs(np.concatenate(targs) - np.concatenate(preds))))
                pos = 0
                for i, ok in enumerate(keep):
                    if ok:
                        gm[i] = self.mag_loss_ratio * gv[pos:pos + len(gm[i])]
                        pos += len(gm[i])
            else:
                out["m_MAE"] = 0.0
            out["m_MAE_size"] = size
            grads["m"] = np.concatenate(gm) if gm else np.zeros(0)
        return out, grads

    def flatten_targets(self, targets: dict, atoms_per_graph) -> dict:
        """Flattened labels of one batch (no cache): ``f`` [N,3], ``s`` [3B,3], ``m`` of the structures whose magmom labels count
        (trainer.py:846: a missing or partly-NaN label drops the structure) with the atoms they belong to.  One concatenation per key
        and one NaN scan over the flat array -- the per-structure Python loop this replaces held the interpreter lock for 12 ms per
        1024-structure batch on the loader thread, next to the main thread's loss."""
        flat = {}
        if "f" in targets:
            flat["f"] = np.concatenate(targets["f"], 0).astype(np.float64, copy=False) if len(targets["f"]) else np.zeros((0, 3))
        if "s" in targets:
            ts = targets["s"]
            flat["s"] = (np.asarray(ts, np.float64).reshape(-1, 3) if isinstance(ts, np.ndarray) else
                         np.concatenate(ts, 0).astype(np.float64, copy=False) if len(ts) else np.zeros((0, 3)))
        if "m" in targets:
            tm = targets["m"]
            n_at = np.asarray(atoms_per_graph, np.int64).reshape(-1)
            present = np.fromiter((mt is not None for mt in tm), bool, len(tm))
            if not self.allow_missing_labels:
                keep = np.ones(len(tm), bool)
                cat = np.concatenate([np.asarray(mt, np.float64).reshape(-1) for mt in tm]) if len(tm) else np.zeros(0)
                sizes = np.fromiter((np.size(mt) for mt in tm), np.int64, len(tm))
            else:
                have = [np.asarray(mt, np.float64).reshape(-1) for mt in tm if mt is not None]
                sizes = np.zeros(len(tm), np.int64)
                sizes[present] = [h.size for h in have]
                cat = np.concatenate(have) if have else np.zeros(0)
                keep = present.copy()
                if cat.size:
                    bad = np.isnan(cat)
                    if bad.any():                      # structures with a NaN anywhere in their label are dropped as a whole
                        starts = np.concatenate([[0], np.cumsum(sizes[present])[:-1]])
                        nonempty = sizes[present] > 0
                        has_nan = np.zeros(int(present.sum()), bool)
                        has_nan[nonempty] = np.add.reduceat(bad, starts[nonempty]) > 0 if nonempty.any() else False
                        keep[np.flatnonzero(present)[has_nan]] = False
            if keep.all():
                flat["m"], flat["m_keep_atoms"], flat["m_size"] = cat, None, int(sizes.sum())
            else:
                off = np.concatenate([[0], np.cumsum(n_at)])
                moff = np.concatenate([[0], np.cumsum(sizes)])
                kept = np.flatnonzero(keep)
                flat["m"] = np.concatenate([cat_slice for cat_slice in (self._label_slices(cat, moff, present, kept))]) if kept.size else np.zeros(0)
                flat["m_keep_atoms"] = np.concatenate([np.arange(off[i], off[i + 1]) for i in kept]) if kept.size else np.zeros(0, np.int64)
                flat["m_size"] = int(sizes[kept].sum())
        return flat

    @staticmethod
    def _label_slices(cat, moff, present, kept):
        """Slices of the concatenated PRESENT labels that belong to the kept structures (``moff`` counts absent ones as empty)."""
        return [cat[moff[i]:moff[i + 1]] for i in kept]

    def _flat_targets(self, targets: dict, atoms_per_graph) -> dict:
        """Flattened labels of one batch, built once per label dictionary (kept in a small cache on this object, keyed by the
        dictionary's identity: the label sets of an epoch come back every epoch)."""
        cache = self.__dict__.setdefault("_flat_cache", {})
        # a hit must be the same dictionary holding the same label OBJECTS for the same graph sizes: a loader that refills one
        # dictionary (or its lists) in place gets fresh labels, not the previous batch's
        def probe(x):   # identity + the BYTES of the first / last value: an array refilled in place changes the stamp (O(1) per
            # array), and a NaN label -- a missing magmom, allow_missing_labels -- compares equal to itself (nan != nan would
            # make every step a miss)
            if type(x) is np.ndarray:      # the common case, without the detour over a flat iterator (3 x 1024 labels per step)
                return (id(x), x.reshape(-1)[:1].tobytes(), x.reshape(-1)[-1:].tobytes()) if x.size else (id(x), b"", b"")
            a = np.asarray(x, np.float64) if x is not None else np.zeros(0)
            return (id(x), a.reshape(-1)[:1].tobytes(), a.reshape(-1)[-1:].tobytes())

        def items(k):   # the label container of key k: a list of per-structure arrays, or ONE stacked array (stress as [B,3,3])
            v = targets.get(k)
            if v is None:
                return ()
            return (v,) if isinstance(v, np.ndarray) or hasattr(v, "detach") else v     # a stacked array is probed as a whole

        sizes = atoms_per_graph.tobytes() if type(atoms_per_graph) is np.ndarray else tuple(int(n) for n in atoms_per_graph)
        stamp = (tuple(id(targets.get(k)) for k in ("f", "s", "m")), tuple(probe(x) for k in ("f", "s", "m") for x in items(k)),
                 sizes, bool(self.allow_missing_labels))
        hit = cache.get(id(targets))
        if hit is not None and hit[0] is targets and hit[2] == stamp:
            return hit[1]
        flat = self.flatten_targets(targets, atoms_per_graph)
        if len(cache) >= 256:
            cache.clear()
        cache[id(targets)] = (targets, flat, stamp)
        return flat

    def forward(self, targets: dict, prediction: dict) -> dict:
        return self.gradients(targets, prediction)[0]

    __call__ = forward


class Adam:
    """torch.optim.Adam (the reference's default optimizer, trainer.py:97-140) over a ``{name: array}`` state dict;
    ``frozen`` names are left untouched (AtomRef: model.py:179-182)."""

    def __init__(self, params: dict, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, frozen=()) -> None:
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.frozen = set(frozen)
        self.m = {k: np.zeros_like(v, dtype=np.float32) for k, v in params.items() if k not in self.frozen}
        self.v = {k: np.zeros_like(v, dtype=np.float32) for k, v in params.items() if k not in self.frozen}
        self.t = 0

    def step(self, params: dict, grads: dict) -> dict:
        self.t += 1
        b1, b2 = self.betas
        c1, c2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        new = {}
        for k, p in params.items():
            if k in self.frozen:
                new[k] = p
                continue
            g = grads[k].astype(np.float32)
            if self.weight_decay:
                g = g + self.weight_decay * p
            self.m[k] = b1 * self.m[k] + (1 - b1) * g
            self.v[k] = b2 * self.v[k] + (1 - b2) * g * g
            new[k] = (p - self.lr * (self.m[k] / c1) / (np.sqrt(self.v[k] / c2) + self.eps)).astype(np.float32)
        return new


def allreduce_gradients(grads: dict, average: bool = True, comm=None) -> dict:
    """Sum (or average) the gradient dictionaries of all ranks: ONE all-reduce of the flattened 1.65 MB buffer
    (``torch.distributed``: backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests; or ``comm``, a
    ``chgnet_amd.distributed.RcclComm`` -- RCCL through the engine library, no torch).  Without a process group the
    gradients are returned unchanged."""
    if comm is not None:
        if comm.world == 1:
            return grads
        keys = sorted(grads)
        flat = comm.all_reduce_sum(np.concatenate([np.asarray(grads[k], np.float32).reshape(-1) for k in keys]))
        if average:
            flat /= comm.world
        out, pos = {}, 0
        for k in keys:
            n = int(np.prod(grads[k].shape))
            out[k] = flat[pos:pos + n].reshape(grads[k].shape)
            pos += n
        return out
    try:
        import torch
        import torch.distributed as dist
    except ImportError:
        return grads
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return grads
    keys = sorted(grads)
    flat = np.concatenate([np.asarray(grads[k], np.float32).reshape(-1) for k in keys])
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.from_numpy(flat).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if average:
        t /= dist.get_world_size()
    flat = t.cpu().numpy()
    out, pos = {}, 0
    for k in keys:
        n = int(np.prod(grads[k].shape))
        out[k] = flat[pos:pos + n].reshape(grads[k].shape)
        pos += n
    return out


class TrainStep:
    """forward -> CombinedLoss -> backward -> (all-reduce) -> Adam -> new weights on the engine: one iteration of the
    reference's ``Trainer._train`` loop (trainer.py:386-411)."""

    def __init__(self, model, *, targets: str = "ef", criterion: str = "MSE", learning_rate: float = 1e-3, comm=None, **loss_kwargs) -> None:
        if set(targets) - set("efsm") or "e" not in targets:
            raise ValueError(f"training targets {targets!r}: a combination of e, f, s, m that contains e (chgnet/__init__.py:15 TrainTask)")
        self.model = model
        self.targets = targets
        self.comm = comm                      # RcclComm, or None for the torch.distributed process group (if any)
        self.loss = CombinedLoss(target_str=targets, criterion=criterion, **loss_kwargs)
        state = model.state_dict()
        frozen = ["composition_model.fc.weight"]
        if not getattr(model, "model_args", {}).get("learnable_rbf", True):   # buffers in the reference (basis.py:31-40, 87-98): never updated
            frozen += [k for k in state if k.endswith(".frequencies")]
        self.optimizer = Adam(state, lr=learning_rate, frozen=tuple(frozen))
        self.seconds: dict = {}
        self.task = "".join(k for k in "efsm" if k in targets)
        if self.task not in ("e", "ef", "em", "efs", "efsm"):       # the engine's task strings (chgnet/__init__.py:15 PredTask)
            self.task = "efsm" if "m" in targets else "efs"

    def run_epoch(self, batches, targets, *, upload_ahead: bool = False) -> list[dict]:
        """One pass over ``batches`` (lists of CrystalGraphs) with their label dictionaries.  The next batch is packed on a helper
        thread (native code, GIL released) while the device works on the current one -- the role of the reference's DataLoader workers
        (chgnet/data/dataset.py ``get_train_val_test_loader``).

        ``upload_ahead=True`` lets the helper also put the packed batch on the device (``Engine.upload`` on its copy stream).  The
        copy is held back until the CURRENT step's forward results are on the host: it then runs under the backward sweeps (5 ms of
        DMA from the page-locked packing buffer under ~120 ms of kernels) and never next to the forward's launches and download --
        started right after packing it collided with them and doubled the forward's time on some hosts (round 4).  Two batches are
        resident on the device then.  Measured on MI355X boxes (1024-structure batches, tools/gpu_train_ab.py): 3 % more
        structures/s; off by default because a host that packs slower than the device steps gains nothing from it."""
        from concurrent.futures import ThreadPoolExecutor  # noqa: PLC0415

        from chgnet_amd.pack import pack_batch  # noqa: PLC0415

        def prepare(i):          # data-loader work of step i: flatten the label lists (handed to the step as they are), pack the graphs
            # the interpreter-bound part first (the main thread is inside the upload / launch calls of its step then, which release the
            # lock), the native packing after it
            flat = self.loss.flatten_targets(targets[i], np.fromiter((len(g.atomic_number) for g in batches[i]), np.int64, len(batches[i])))
            # packed into page-locked memory (two alternating blocks: batch i + 1 is packed while batch i is in use): the step's upload is
            # then DMA at the link rate
            pinned = getattr(getattr(self.model, "engine", None), "pinned_allocator", None)
            packed = pack_batch(batches[i], alloc=pinned(i & 1) if pinned is not None else None)
            device_batch = None
            if upload_ahead:
                if i > 0:
                    forward_done[i - 1].wait()
                device_batch = self.model.engine.upload(packed)
            return packed, device_batch, flat

        import sys  # noqa: PLC0415

        import threading  # noqa: PLC0415

        forward_done = [threading.Event() for _ in batches]
        infos = []
        # Two interpreter threads: with CPython's default 5 ms switch interval the main thread's loss (a few dozen small numpy calls,
        # 0.3 ms alone) took 20-27 ms next to the loader -- every time it gave the lock up it waited a full interval to get it back
        # (tools/gpu_train_anatomy.py).  A 0.2 ms interval for the epoch keeps both threads moving.
        interval = sys.getswitchinterval()
        sys.setswitchinterval(min(interval, 2e-4))
        try:
            with ThreadPoolExecutor(max_workers=1) as pool:
                nxt = pool.submit(prepare, 0) if len(batches) else None
                for i in range(len(batches)):
                    packed, device_batch, flat = nxt.result()
                    nxt = pool.submit(prepare, i + 1) if i + 1 < len(batches) else None
                    infos.append(self(packed, targets[i], device_batch=device_batch, flat_targets=flat, after_forward=forward_done[i].set))
        finally:
            for ev in forward_done:      # a step that raised must not leave the helper waiting
                ev.set()
            sys.setswitchinterval(interval)
        return infos

    def __call__(self, graphs, targets: dict, device_batch=None, flat_targets: dict | None = None, after_forward=None) -> dict:
        import time  # noqa: PLC0415

        model = self.model
        t0 = time.perf_counter()
        try:
            pred = model.forward(graphs, task=self.task, device_batch=device_batch)
        finally:
            if after_forward is not None:     # run_epoch: the helper thread may start the next batch's upload now
                after_forward()
        t1 = time.perf_counter()
        info, g = self.loss.gradients(targets, pred, flat_targets=flat_targets)
        t2 = time.perf_counter()
        if self.comm is not None and self.comm.world > 1:
            # RCCL straight from the engine library: the 1.65 MB blob is summed in HBM on the engine's stream
            grads = model.backward(g.get("e"), g.get("m"), g.get("f"), g.get("s"), comm=self.comm)
            grads = {k: v / self.comm.world for k, v in grads.items()}
        else:
            grads = model.backward(g.get("e"), g.get("m"), g.get("f"), g.get("s"))
            grads = allreduce_gradients(grads, comm=self.comm)
        t3 = time.perf_counter()
        model.load_state_dict(self.optimizer.step(model.state_dict(), grads))
        t4 = time.perf_counter()
        # wall-clock split of the step (seconds), summed over the calls: what bench.py reports next to the epoch time
        for key, dt in (("forward", t1 - t0), ("loss", t2 - t1), ("backward_allreduce", t3 - t2), ("optimizer_reload", t4 - t3)):
            self.seconds[key] = self.seconds.get(key, 0.0) + dt
        self.seconds["calls"] = self.seconds.get("calls", 0) + 1
        return info
