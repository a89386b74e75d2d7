"""Minimal NVT molecular-dynamics driver (velocity Verlet + Berendsen thermostat).

The reference drives MD through ASE (``MolecularDynamics``, dynamics.py:433-780: ASE integrators calling
``CHGNetCalculator.calculate`` every step).  ASE is a third-party driver and is absent offline, so this
small integrator exists to exercise and time exactly that calculator path (BASELINE config 4: 2x2x2
Li9Co7O16, graph rebuilt on the host every step like dynamics.py:156-157).  It is not a replacement for
ASE: when ASE is installed use ``CHGNetCalculator`` with any ASE dynamics.

Units: eV, Angstrom, amu, fs (1 ASE time unit = Angstrom*sqrt(amu/eV) = 10.1805 fs).
"""

from __future__ import annotations

import time

import numpy as np

from chgnet_amd.graph.structure import Lattice, Structure

KB_EV = 8.617333262e-5           # eV / K
FS_PER_TIME_UNIT = 10.1805055    # fs per Angstrom*sqrt(amu/eV)

# standard atomic weights (amu) of the elements CHGNet covers (Z = 1..94)
ATOMIC_MASS = np.array([
    0, 1.008, 4.0026, 6.94, 9.0122, 10.81, 12.011, 14.007, 15.999, 18.998, 20.180, 22.990, 24.305, 26.982, 28.085,
    30.974, 32.06, 35.45, 39.948, 39.098, 40.078, 44.956, 47.867, 50.942, 51.996, 54.938, 55.845, 58.933, 58.693,
    63.546, 65.38, 69.723, 72.630, 74.922, 78.971, 79.904, 83.798, 85.468, 87.62, 88.906, 91.224, 92.906, 95.95, 98.0,
    101.07, 102.91, 106.42, 107.87, 112.41, 114.82, 118.71, 121.76, 127.60, 126.90, 131.29, 132.91, 137.33, 138.91,
    140.12, 140.91, 144.24, 145.0, 150.36, 151.96, 157.25, 158.93, 162.50, 164.93, 167.26, 168.93, 173.05, 174.97,
    178.49, 180.95, 183.84, 186.21, 190.23, 192.22, 195.08, 196.97, 200.59, 204.38, 207.2, 208.98, 209.0, 210.0,
    222.0, 223.0, 226.0, 227.0, 232.04, 231.04, 238.03, 237.0, 244.0])


class BerendsenNVT:
    """Velocity Verlet with Berendsen velocity rescaling."""

    def __init__(self, structure: Structure, calculator, *, temperature_K: float = 300.0, timestep_fs: float = 2.0,
                 taut_fs: float = 100.0, seed: int = 0, task: str = "ef") -> None:
        self.structure = structure.copy()
        self.calc = calculator
        self.T0 = temperature_K
        self.dt = timestep_fs / FS_PER_TIME_UNIT
        self.taut = taut_fs / FS_PER_TIME_UNIT
        self.task = task
        self.mass = ATOMIC_MASS[self.structure.atomic_numbers][:, None]
        self._m3 = np.repeat(self.mass, 3, axis=1)                   # per-component masses / half-step factors (step() runs ~10 numpy calls)
        self._half_dt_over_m = 0.5 * (timestep_fs / FS_PER_TIME_UNIT) / self._m3
        rng = np.random.default_rng(seed)
        self.vel = rng.normal(0.0, 1.0, (len(self.structure), 3)) * np.sqrt(KB_EV * temperature_K / self.mass)
        self.vel -= (self.vel * self.mass).sum(0) / self.mass.sum()      # no centre-of-mass drift
        self.forces = None
        self.energy = None
        self.timing = {"graph+predict": 0.0, "steps": 0}

    def temperature(self) -> float:
        ke2 = float(np.vdot(self.vel, self._m3 * self.vel))            # 2 x kinetic energy
        return ke2 / (3.0 * len(self.structure) * KB_EV)

    def _evaluate(self) -> None:
        t0 = time.perf_counter()
        self.calc.calculate(self.structure, task=self.task)
        self.timing["graph+predict"] += time.perf_counter() - t0
        self.forces = np.asarray(self.calc.results["forces"], dtype=np.float64)
        self.energy = float(self.calc.results["energy"])

    def advance_positions(self) -> None:
        """Thermostat, first half kick and drift of one step (needs ``self.forces``); the caller evaluates the new forces and applies
        the second half kick.  ``step`` is exactly this + ``_evaluate`` + the second kick; an ensemble driver (bench.py C4) calls it
        for R replicas around ONE batched prediction."""
        lam = np.sqrt(1.0 + self.dt / self.taut * (self.T0 / max(self.temperature(), 1e-12) - 1.0))
        vel = self.vel
        vel *= min(max(lam, 0.9), 1.1)
        vel += self._half_dt_over_m * self.forces
        lattice = self.structure.lattice                       # fixed cell: the Lattice object and its inverse are reused
        if getattr(self, "_inv_of", None) is not lattice:
            self._inv_of, self._inv = lattice, np.linalg.inv(lattice.matrix)
            self._dt_inv = self.dt * self._inv
        # x(t + dt) = x + dt v in fractional coordinates: frac += (dt v) . L^-1  (one small matrix product; no cartesian round trip)
        new = Structure.__new__(Structure)
        new.lattice, new.atomic_numbers = lattice, self.structure.atomic_numbers
        new.frac_coords = self.structure.frac_coords + vel @ self._dt_inv
        self.structure = new

    def step(self) -> None:
        if self.forces is None:
            self._evaluate()
        self.advance_positions()
        self._evaluate()
        self.vel += self._half_dt_over_m * self.forces
        self.timing["steps"] += 1

    def run(self, n_steps: int) -> dict:
        t0 = time.perf_counter()
        for _ in range(n_steps):
            self.step()
        wall = time.perf_counter() - t0
        return {"steps": n_steps, "wall_s": wall, "steps_per_s": n_steps / wall, "temperature_K": self.temperature(),
                "energy_eV": self.energy, "calculator_s": self.timing["graph+predict"]}
