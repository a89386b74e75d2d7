"""``CHGNetCalculator`` -- the ASE-facing surface of the reference
(chgnet/model/dynamics.py:58-181) on top of the HIP engine.

ASE is an optional dependency (absent in the build container): when it imports, the class derives
from ``ase.calculators.calculator.Calculator`` so ASE optimisers / MD drivers can use it unchanged;
otherwise a minimal base with the same ``results`` / ``get_*`` contract is used so that the
in-repo MD smoke driver and the tests still work.
"""

from __future__ import annotations

import numpy as np

from chgnet_amd.graph.structure import Lattice, Structure, atomic_numbers_of
from chgnet_amd.model import CHGNet

GPA_TO_EV_A3 = 1.0 / 160.21766208   # ase.units.GPa

try:  # pragma: no cover - exercised only where ASE is installed
    from ase.calculators.calculator import Calculator, all_changes, all_properties

    HAVE_ASE = True
except ImportError:
    HAVE_ASE = False
    all_changes, all_properties = [], []

    class Calculator:  # minimal stand-in with ASE's attribute contract
        def __init__(self, **kwargs) -> None:  # noqa: ARG002
            self.results: dict = {}
            self.atoms = None

        def calculate(self, atoms=None, properties=None, system_changes=None) -> None:  # noqa: ARG002
            self.atoms = atoms

        def get_potential_energy(self, atoms=None):
            self.calculate(atoms)
            return self.results["energy"]

        def get_forces(self, atoms=None):
            self.calculate(atoms)
            return self.results["forces"]

        def get_stress(self, atoms=None):
            self.calculate(atoms)
            return self.results["stress"]


def atoms_to_structure(atoms) -> Structure:
    """ASE ``Atoms`` (or anything with get_cell / get_scaled_positions / get_atomic_numbers) or one
    of our / pymatgen's structures -> ``Structure`` (replaces AseAtomsAdaptor, dynamics.py:156)."""
    if hasattr(atoms, "frac_coords") and hasattr(atoms, "lattice"):
        return atoms
    cell = np.asarray(atoms.get_cell()[:] if hasattr(atoms.get_cell(), "__getitem__") else atoms.get_cell())
    return Structure(Lattice(cell), np.asarray(atoms.get_atomic_numbers()), np.asarray(atoms.get_scaled_positions(wrap=False)))


class CHGNetCalculator(Calculator):
    """CHGNet Calculator for ASE applications."""

    implemented_properties = ("energy", "forces", "stress", "magmoms", "energies")

    def __init__(self, model: CHGNet | None = None, *, use_device: str | None = None, check_cuda_mem: bool = False,  # noqa: ARG002
                 stress_weight: float = GPA_TO_EV_A3, on_isolated_atoms: str = "warn", return_site_energies: bool = False,
                 skin: float = 0.0, **kwargs) -> None:
        """Same arguments as the reference (dynamics.py:63-107) plus ``skin`` (Angstrom, default 0 = the
        reference behaviour: graph rebuilt on the host every call).

        ``skin > 0`` enables the device-resident MD fast path: the graph is built once with both cutoffs
        enlarged by ``skin`` and kept in HBM; while no atom has moved more than ``skin / 2`` (and the cell
        is unchanged) a call only uploads positions (``chg_batch_update_geometry``) and re-runs the
        kernels.  This is exact, not an approximation: the polynomial envelope is identically zero beyond
        the cutoff (basis.py:205), so bonds in the skin shell carry zero features, zero weights and zero
        gradients.  It requires ``mlp_out`` without bias (0.3.0 / r2scan; the 0.2.0 bias would leak into
        skin bonds) and is refused otherwise."""
        super().__init__(**kwargs)
        if model is None:
            self.model = CHGNet.load(verbose=False, use_device=use_device)
        else:
            self.model = model.to(use_device) if use_device is not None else model
        self.device = self.model.device
        self.model.graph_converter.set_isolated_atom_response(on_isolated_atoms)
        self.stress_weight = stress_weight
        self.return_site_energies = return_site_energies
        self.skin = float(skin)
        self._resident = None
        self.n_graph_builds = 0
        if self.skin > 0:
            if any(k.endswith("mlp_out.layers.1.bias") for k in self.model.state_dict()):
                raise ValueError("skin > 0 needs a model without mlp_out bias (0.3.0 / r2scan); got mlp_out_bias=True")
            from chgnet_amd.graph import CrystalGraphConverter

            conv = self.model.graph_converter
            self._skin_converter = CrystalGraphConverter(
                atom_graph_cutoff=conv.atom_graph_cutoff + self.skin, bond_graph_cutoff=conv.bond_graph_cutoff + self.skin,
                on_isolated_atoms=on_isolated_atoms)
        print(f"CHGNet will run on {self.device}")

    @classmethod
    def from_file(cls, path: str, use_device: str | None = None, **kwargs) -> "CHGNetCalculator":
        return cls(model=CHGNet.from_file(path), use_device=use_device, **kwargs)

    @property
    def version(self) -> str | None:
        return self.model.version

    @property
    def n_params(self) -> int:
        return self.model.n_params

    def calculate(self, atoms=None, properties=None, system_changes=None, task: str = "efsm") -> None:
        properties = properties or all_properties
        system_changes = system_changes or all_changes
        super().calculate(atoms=atoms, properties=properties, system_changes=system_changes)
        structure = atoms_to_structure(atoms)
        if self.skin > 0:
            pred = self._predict_resident(structure, task)
        else:
            self.n_graph_builds += 1   # graph rebuilt every call like dynamics.py:156-157, but on the device
            pred = self.model.predict_structure(structure, task=task, return_crystal_feas=True,
                                                return_site_energies=self.return_site_energies)
        extensive_factor = len(structure) if self.model.is_intensive else 1
        key_map = {"e": ("energy", extensive_factor), "f": ("forces", 1), "m": ("magmoms", 1), "s": ("stress", self.stress_weight)}
        self.results.update({long_key: pred[key] * factor for key, (long_key, factor) in key_map.items() if key in pred})
        self.results["free_energy"] = self.results["energy"]
        self.results["crystal_fea"] = pred["crystal_fea"]
        if self.return_site_energies:
            self.results["energies"] = pred["site_energies"]

    # ---- device-resident fast path (skin > 0) ----------------------------------------------------------
    def _predict_resident(self, structure, task: str) -> dict:
        from chgnet_amd import VALID_TASKS

        if task not in VALID_TASKS:
            raise ValueError(f"Invalid {task=}. Must be one of {VALID_TASKS}.")
        eng = self.model.engine
        lattice = np.asarray(structure.lattice.matrix, dtype=np.float64)
        frac = np.asarray(structure.frac_coords, dtype=np.float64)          # unwrapped: images stay valid
        z = atomic_numbers_of(structure)
        cart = frac @ lattice
        res = self._resident
        reuse = (res is not None and len(z) == len(res["z"]) and np.array_equal(z, res["z"])
                 and np.array_equal(lattice, res["lattice"])
                 and float(np.sqrt(((cart - res["cart"]) ** 2).sum(1).max())) < 0.5 * self.skin)
        if reuse:
            res["batch"].update_geometry(frac.astype(np.float32), lattice.astype(np.float32)[None])
        else:
            if res is not None:
                res["batch"].free()
            conv = self._skin_converter
            self.n_graph_builds += 1
            batch = eng.build_batch([structure], conv.atom_graph_cutoff, conv.bond_graph_cutoff)
            if batch.packed.n_isolated and conv.on_isolated_atoms != "ignore":
                conv(structure)   # phrases the reference's isolated-atom error / warning
            res = self._resident = {"z": z, "lattice": lattice.copy(), "cart": cart.copy(), "batch": batch}
        eng.predict(res["batch"], task)
        out = eng.download(res["batch"], task, site_energies=self.return_site_energies, crystal_feas=True)
        pred = {"e": out["e"][0], "crystal_fea": out["crystal_fea"][0]}
        for key in ("f", "m", "site_energies"):
            if key in out:
                pred[key] = out[key]
        if "s" in out:
            pred["s"] = out["s"][0]
        return pred
