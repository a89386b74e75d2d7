"""Files written by the REFERENCE's own save paths, for the readers of SURVEY 8f-4 (chgnet_amd/safe_load.py).

    python tests/golden/make_ref_files.py        # needs /root/reference (this container only)

Writes (committed; the GPU box and the CPU tests only READ them):
  ref_trainer_save.pth.tar   ``Trainer.save`` of the unmodified reference (trainer/trainer.py:614-623): {"model": CHGNet.as_dict()
                             (state_dict + model_args, model.py:667-669), "optimizer": a real torch.optim.Adam state_dict AFTER one
                             step (exp_avg / exp_avg_sq per parameter), "scheduler": CosineAnnealingLR state_dict,
                             "training_history", "trainer_args"}.  The model is make_golden.make_reference_model() (== the
                             committed weights_seed0.npz) moved by that one Adam step -- the expected state_dict is saved next
                             to it (ref_trainer_save_state.npz) from the live module, not from the file.
  ref_graph_<case>.pt        ``CrystalGraph.save`` (graph/crystalgraph.py:138-156: torch.save(self.to_dict())) of two graphs the
                             reference's own converter built (limno2, s16tri) -- the arrays are the ones of case_<name>.npz.
The optimizer step uses a synthetic gradient (0.01 * randn, seeded): what matters is a file with every key torch writes.
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

import make_golden  # noqa: E402  (imports the reference through oracle/_refimport.py)
from chgnet.graph.crystalgraph import CrystalGraph as RefGraph  # noqa: E402
from chgnet.trainer.trainer import Trainer as RefTrainer  # noqa: E402


def main() -> None:
    model = make_golden.make_reference_model()
    model.train()
    trainer = RefTrainer(model=model, targets="efsm", optimizer="Adam", scheduler="CosLR", criterion="MSE", epochs=5,
                         learning_rate=1e-3, use_device="cpu", print_freq=10, torch_seed=7, data_seed=11)
    gen = torch.Generator().manual_seed(5)
    for p in model.parameters():
        if p.requires_grad:
            p.grad = 0.01 * torch.randn(p.shape, generator=gen)
    trainer.optimizer.step()
    trainer.scheduler.step()
    trainer.training_history["e"]["train"].append(0.0123)
    trainer.training_history["f"]["val"].append(0.0456)
    path = os.path.join(HERE, "ref_trainer_save.pth.tar")
    trainer.save(path)
    np.savez_compressed(os.path.join(HERE, "ref_trainer_save_state.npz"),
                        **{k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    print("wrote", path, os.path.getsize(path), "bytes; model_args:", model.model_args)

    for name in ("limno2", "s16tri"):
        d = np.load(os.path.join(HERE, f"case_{name}.npz"))
        t = torch.as_tensor
        g = RefGraph(atomic_number=t(d["atomic_number"]), atom_frac_coord=t(d["atom_frac_coord"]), atom_graph=t(d["atom_graph"]),
                     atom_graph_cutoff=6.0, neighbor_image=t(d["neighbor_image"]), directed2undirected=t(d["directed2undirected"]),
                     undirected2directed=t(d["undirected2directed"]), bond_graph=t(d["bond_graph"]), bond_graph_cutoff=3.0,
                     lattice=t(d["lattice"]), graph_id=name, mp_id=f"mp-{name}", composition="ref")
        out = g.save(fname=f"ref_graph_{name}.pt", save_dir=HERE)
        print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
