"""Host side of the fine-tuning path: CombinedLoss against the reference's own class, Adam against torch.optim.Adam,
gradient all-reduce over a 2-rank gloo group."""

from __future__ import annotations

import os
import socket
import sys

import numpy as np
import pytest
import torch

from chgnet_amd.trainer import Adam, CombinedLoss, TrainStep, allreduce_gradients
from conftest import REPO


def _batch(rng, missing: bool):
    n_at = [3, 5, 2]
    pred = {"e": rng.normal(size=3), "f": [rng.normal(size=(n, 3)) for n in n_at], "s": [rng.normal(size=(3, 3)) for _ in n_at],
            "m": [np.abs(rng.normal(size=n)) for n in n_at]}
    targ = {"e": rng.normal(size=3), "f": [rng.normal(size=(n, 3)) for n in n_at], "s": [rng.normal(size=(3, 3)) for _ in n_at],
            "m": [np.abs(rng.normal(size=n)) for n in n_at]}
    if missing:
        targ["e"][1] = np.nan
        targ["f"][0][1] = np.nan
        targ["s"][2][:] = np.nan
        targ["m"][1] = None
        targ["m"][2][0] = np.nan            # a NaN anywhere drops the whole structure's magmoms (trainer.py:846)
    return targ, pred


@pytest.mark.skipif(not os.path.isdir("/root/reference/chgnet"), reason="live reference only in the build container")
@pytest.mark.parametrize("criterion", ["MSE", "MAE", "Huber"])
@pytest.mark.parametrize(("missing", "allow"), [(False, True), (True, True), (False, False)])
def test_combined_loss_equals_the_references(criterion, missing, allow):
    """Loss value, MAEs, MAE sizes and d loss / d prediction against chgnet.trainer.trainer.CombinedLoss + autograd; also without
    label masking, where the reference's sizes count target ROWS ([N,3], [3B,3]) instead of elements (trainer.py:826, 840)."""
    from oracle._refimport import load_reference

    load_reference(fast_graph=True)
    from chgnet.trainer.trainer import CombinedLoss as RefLoss

    rng = np.random.default_rng(3)
    targ, pred = _batch(rng, missing)
    kw = dict(target_str="efsm", criterion=criterion, energy_loss_ratio=1.3, force_loss_ratio=0.7, stress_loss_ratio=0.2,
              mag_loss_ratio=0.4, delta=0.3, allow_missing_labels=allow)
    ours, grads = CombinedLoss(**kw).gradients(targ, pred)
    tp = {"e": torch.tensor(pred["e"], requires_grad=True), "f": [torch.tensor(x, requires_grad=True) for x in pred["f"]],
          "s": [torch.tensor(x, requires_grad=True) for x in pred["s"]], "m": [torch.tensor(x, requires_grad=True) for x in pred["m"]]}
    tt = {"e": torch.tensor(targ["e"]), "f": [torch.tensor(x) for x in targ["f"]], "s": [torch.tensor(x) for x in targ["s"]],
          "m": [None if x is None else torch.tensor(x) for x in targ["m"]]}
    ref = RefLoss(**kw)(tt, tp)
    assert abs(ours["loss"] - float(ref["loss"])) < 1e-12
    for k in "efsm":
        assert abs(ours[f"{k}_MAE"] - float(ref[f"{k}_MAE"])) < 1e-12, k
        assert ours[f"{k}_MAE_size"] == int(ref[f"{k}_MAE_size"]), k
    ref["loss"].backward()
    assert np.allclose(grads["e"], tp["e"].grad.numpy(), atol=1e-14)
    assert np.allclose(grads["f"], np.concatenate([x.grad.numpy() for x in tp["f"]]), atol=1e-14)
    assert np.allclose(grads["s"].reshape(-1, 3), np.concatenate([x.grad.numpy() for x in tp["s"]]), atol=1e-14)
    gm = np.concatenate([np.zeros(len(x)) if x.grad is None else x.grad.numpy() for x in tp["m"]])
    assert np.allclose(grads["m"], gm, atol=1e-14)


def test_adam_equals_torch_adam():
    rng = np.random.default_rng(0)
    params = {"a": rng.normal(size=(4, 3)).astype(np.float32), "b": rng.normal(size=5).astype(np.float32),
              "frozen": np.ones(2, np.float32)}
    tparams = {k: torch.tensor(v.copy(), requires_grad=True) for k, v in params.items() if k != "frozen"}
    topt = torch.optim.Adam(list(tparams.values()), lr=3e-3)
    opt = Adam(params, lr=3e-3, frozen=("frozen",))
    for _ in range(5):
        grads = {k: rng.normal(size=v.shape).astype(np.float32) for k, v in params.items()}
        for k, t in tparams.items():
            t.grad = torch.tensor(grads[k])
        topt.step()
        params = opt.step(params, grads)
    for k, t in tparams.items():
        assert np.allclose(params[k], t.detach().numpy(), atol=2e-6), k
    assert np.array_equal(params["frozen"], np.ones(2, np.float32))


def _allreduce_worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, REPO)
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grads = {"w": np.full((3, 2), float(rank + 1), np.float32), "b": np.arange(4, dtype=np.float32) * (rank + 1)}
    out = allreduce_gradients(grads)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks(tmp_path):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_allreduce_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    for k in ("w", "b"):
        assert np.array_equal(r0[k], r1[k])
    assert np.allclose(r0["w"], 1.5) and np.allclose(r0["b"], np.arange(4) * 1.5)      # mean over the two ranks
    same = {"w": np.ones(2, np.float32)}
    assert allreduce_gradients(same) is same                                             # no process group: unchanged


def test_train_step_argument_checks(golden_weights):
    from chgnet_amd.model import CHGNet

    model = CHGNet(state_dict=golden_weights)
    with pytest.raises(ValueError, match="training targets"):
        TrainStep(model, targets="fs")          # the reference's TrainTask always contains the energy
    assert TrainStep(model, targets="efsm").task == "efsm" and TrainStep(model, targets="ef").task == "ef"


@pytest.mark.parametrize("criterion", ["MSE", "Huber"])
@pytest.mark.parametrize(("missing", "allow"), [(False, True), (True, True), (False, False)])
def test_flat_loss_path_equals_the_list_path(criterion, missing, allow):
    """``CHGNet.forward`` attaches the whole-batch arrays to its dictionary and ``CombinedLoss`` then works on those and on labels
    flattened once per label set: same loss, MAEs, sizes and cotangents as the per-structure lists (also with missing labels)."""
    from chgnet_amd.model import ForwardResult

    rng = np.random.default_rng(11)
    targ, pred = _batch(rng, missing)
    kw = dict(target_str="efsm", criterion=criterion, energy_loss_ratio=1.3, force_loss_ratio=0.7, stress_loss_ratio=0.2,
              mag_loss_ratio=0.4, delta=0.3, allow_missing_labels=allow)
    want, gwant = CombinedLoss(**kw).gradients(targ, pred)
    fast = ForwardResult(dict(pred, atoms_per_graph=np.array([len(x) for x in pred["f"]])))
    fast.flat = {"f": np.concatenate(pred["f"], 0), "s": np.stack(pred["s"]), "m": np.concatenate(pred["m"])}
    loss = CombinedLoss(**kw)
    for _ in range(2):            # second call: labels from the cache
        got, ggot = loss.gradients(targ, fast)
        assert set(got) == set(want)
        for k, v in want.items():
            assert (np.isnan(v) and np.isnan(got[k])) or abs(got[k] - v) < 1e-14, k
        for k in "efsm":
            assert np.allclose(np.asarray(ggot[k]).reshape(-1), np.asarray(gwant[k]).reshape(-1), atol=1e-15), k


def test_flat_label_cache_sees_a_refilled_dictionary():
    """the flattened-label cache is keyed on the label dictionary's identity.  A loader that reuses ONE dictionary --
    replacing its lists, or refilling the arrays in place -- must get the new labels, not the previous batch's."""
    from chgnet_amd.model import ForwardResult

    rng = np.random.default_rng(5)
    targ, pred = _batch(rng, False)
    fast = ForwardResult(dict(pred, atoms_per_graph=np.array([len(x) for x in pred["f"]])))
    fast.flat = {"f": np.concatenate(pred["f"], 0), "s": np.stack(pred["s"]), "m": np.concatenate(pred["m"])}
    loss = CombinedLoss(target_str="efsm", criterion="MSE")
    first = loss.gradients(targ, fast)[0]["loss"]
    assert loss.gradients(targ, fast)[0]["loss"] == first                    # cache hit: same labels
    targ["f"] = [x + 1.0 for x in targ["f"]]                                  # same dictionary, new list of new arrays
    second = loss.gradients(targ, fast)[0]["loss"]
    want = CombinedLoss(target_str="efsm", criterion="MSE").gradients(targ, pred)[0]["loss"]
    assert abs(second - want) < 1e-12 and abs(second - first) > 0.1
    for x in targ["f"]:
        x += 2.0                                                              # same arrays refilled in place
    third = loss.gradients(targ, fast)[0]["loss"]
    want = CombinedLoss(target_str="efsm", criterion="MSE").gradients(targ, pred)[0]["loss"]
    assert abs(third - want) < 1e-12 and abs(third - second) > 0.1


def test_flat_label_cache_with_stacked_stress_and_nan_magmoms():
    """(i) a label container that is ONE stacked array (stress as [B,3,3]) must not be truth-tested
    (``ValueError: truth value of an array is ambiguous``); (ii) NaN labels (missing magmoms under
    ``allow_missing_labels``, the default) must not defeat the cache: the stamp compares bytes, not values (nan != nan)."""
    from chgnet_amd.model import ForwardResult

    rng = np.random.default_rng(6)
    targ, pred = _batch(rng, False)
    targ["s"] = np.stack(targ["s"])                                            # one [B,3,3] array instead of a list
    targ["m"] = [np.asarray(x, np.float64).copy() for x in targ["m"]]
    targ["m"][0][0] = np.nan                                                   # first value of the first label array
    targ["m"][-1][-1] = np.nan
    apg = np.array([len(x) for x in pred["f"]])
    fast = ForwardResult(dict(pred, atoms_per_graph=apg))
    fast.flat = {"f": np.concatenate(pred["f"], 0), "s": np.stack(pred["s"]), "m": np.concatenate(pred["m"])}
    loss = CombinedLoss(target_str="efsm", criterion="MSE", allow_missing_labels=True)
    a = loss._flat_targets(targ, apg)
    assert loss._flat_targets(targ, apg) is a                                  # hit, NaNs and all
    want = CombinedLoss(target_str="efsm", criterion="MSE", allow_missing_labels=True).gradients(
        dict(targ, s=list(targ["s"])), pred)[0]
    got = loss.gradients(targ, fast)[0]
    assert abs(got["loss"] - want["loss"]) < 1e-12 and got["m_MAE_size"] == want["m_MAE_size"]
    targ["s"][0, 0, 0] += 1.0                                                  # the stacked array refilled in place: a miss
    assert loss._flat_targets(targ, apg) is not a


def test_run_epoch_hands_every_batch_its_own_upload_in_order():
    """The loader of ``TrainStep.run_epoch`` with ``upload_ahead=True`` (the helper thread packs AND uploads the next batch while a step
    runs): every step gets the device batch that was uploaded from ITS packed batch, in order, each batch uploaded exactly once, for
    epochs of 1, 2 and 5 batches; with the default (False) the helper only packs and the step uploads.  Stub model and engine."""
    import threading
    import time

    from conftest import load_case

    graph, _ = load_case("limno2")
    n_at = len(graph.atomic_number)

    class StubEngine:
        def __init__(self):
            self.uploads, self.events, self.lock = [], [], threading.Lock()

        def upload(self, packed):
            with self.lock:
                self.uploads.append(packed.n_struct)          # (unique per batch in this test; id() is reused after collection)
                self.events.append(("upload", packed.n_struct))
            return ("device", packed.n_struct)

    class StubModel:
        model_args = {}

        def __init__(self):
            self.engine, self.steps = StubEngine(), []
            self.w = {"w": np.zeros(3, np.float32)}

        def state_dict(self):
            return dict(self.w)

        def load_state_dict(self, sd):
            self.w = dict(sd)

        def forward(self, packed, *, task, device_batch=None):
            if self.expect_upload:
                assert device_batch == ("device", packed.n_struct), "a step must run on the upload of its own packed batch"
            else:
                assert device_batch is None
            self.steps.append(packed.n_struct)
            if self.fail_at == packed.n_struct:
                raise RuntimeError("forward failed")
            time.sleep(0.02)                                  # the helper has packed the next batch long before this returns
            with self.engine.lock:
                self.engine.events.append(("forward done", packed.n_struct))
            b = packed.n_struct
            out = {"atoms_per_graph": np.diff(packed.atom_off).astype(np.int64), "e": np.zeros(b, np.float32),
                   "f": [np.zeros((n_at, 3), np.float32) for _ in range(b)]}
            return out

        def backward(self, e_grad=None, m_grad=None, f_grad=None, s_grad=None, comm=None):
            return {"w": np.ones(3, np.float32)}

    for n_batches, ahead in ((1, True), (2, True), (5, True), (3, False)):
        model = StubModel()
        model.expect_upload, model.fail_at = ahead, -1
        step = TrainStep(model, targets="ef", learning_rate=1e-3)
        batches = [[graph] * (i + 1) for i in range(n_batches)]          # batch i holds i + 1 structures: the order is visible
        labels = [{"e": np.zeros(i + 1, np.float32), "f": [np.zeros((n_at, 3), np.float32) for _ in range(i + 1)]} for i in range(n_batches)]
        infos = step.run_epoch(batches, labels, upload_ahead=ahead)
        assert len(infos) == n_batches and model.steps == [i + 1 for i in range(n_batches)]
        assert model.engine.uploads == ([i + 1 for i in range(n_batches)] if ahead else [])      # each batch once, in order
        # the copy of batch i + 1 is held back until the forward of batch i has returned (it then runs under the backward sweeps,
        # never next to the forward's launches and download)
        ev = model.engine.events
        for i in range(2, n_batches + 1):
            if ahead:
                assert ev.index(("upload", i)) > ev.index(("forward done", i - 1)), ev
    assert step.run_epoch([], []) == []
    # a step that raises does not leave the helper thread waiting for its forward
    model = StubModel()
    model.expect_upload, model.fail_at = True, 2
    step = TrainStep(model, targets="ef", learning_rate=1e-3)
    batches = [[graph] * (i + 1) for i in range(4)]
    labels = [{"e": np.zeros(i + 1, np.float32), "f": [np.zeros((n_at, 3), np.float32) for _ in range(i + 1)]} for i in range(4)]
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="forward failed"):
        step.run_epoch(batches, labels, upload_ahead=True)
    assert time.perf_counter() - t0 < 30 and model.steps == [1, 2]
