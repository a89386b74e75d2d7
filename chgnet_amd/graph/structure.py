"""Minimal periodic ``Structure`` / ``Lattice`` (pymatgen is not a dependency).

Only what the hot path touches on a pymatgen ``Structure``
(reference chgnet/graph/converter.py:120-134,187): ``len``, iteration yielding
``site.specie.Z``, ``frac_coords``, ``lattice.matrix``, ``composition.formula``.
A real pymatgen ``Structure`` can be passed to the converter unchanged; these classes
exist so that tests, benches and MD drivers work where pymatgen is absent.
"""

from __future__ import annotations

import math
import re
from collections import Counter
from types import SimpleNamespace

import numpy as np

_SYMBOLS = (
    "H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr "
    "Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm "
    "Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr"
).split()
SYMBOL_TO_Z = {s: i + 1 for i, s in enumerate(_SYMBOLS)}
Z_TO_SYMBOL = {i + 1: s for i, s in enumerate(_SYMBOLS)}


class Lattice:
    """Rows of ``matrix`` are the lattice vectors a, b, c in Angstrom."""

    def __init__(self, matrix) -> None:
        self.matrix = np.array(matrix, dtype=np.float64).reshape(3, 3)

    @classmethod
    def from_parameters(cls, a, b, c, alpha, beta, gamma) -> "Lattice":
        """Same orientation convention as pymatgen: c along z, a in the x-z plane."""
        al, be, ga = (math.radians(x) for x in (alpha, beta, gamma))
        cos_al, cos_be, cos_ga = math.cos(al), math.cos(be), math.cos(ga)
        sin_al, sin_be = math.sin(al), math.sin(be)
        val = (cos_al * cos_be - cos_ga) / (sin_al * sin_be)
        val = max(-1.0, min(1.0, val))
        gamma_star = math.acos(val)
        va = [a * sin_be, 0.0, a * cos_be]
        vb = [-b * sin_al * math.cos(gamma_star), b * sin_al * math.sin(gamma_star), b * cos_al]
        vc = [0.0, 0.0, float(c)]
        return cls([va, vb, vc])

    @property
    def volume(self) -> float:
        m = self.matrix
        return float(abs(np.dot(m[0], np.cross(m[1], m[2]))))


class Structure:
    """Periodic structure: lattice + atomic numbers + fractional coordinates."""

    def __init__(self, lattice, species, frac_coords) -> None:
        self.lattice = lattice if isinstance(lattice, Lattice) else Lattice(lattice)
        if isinstance(species, np.ndarray) and species.dtype.kind in "iu":     # atomic numbers already (an MD driver rebuilds the
            self.atomic_numbers = species.astype(np.int32)                       # structure every step: no per-atom Python work)
        else:
            self.atomic_numbers = np.array([SYMBOL_TO_Z[s] if isinstance(s, str) else int(s) for s in species], dtype=np.int32)
        self.frac_coords = np.array(frac_coords, dtype=np.float64).reshape(len(self.atomic_numbers), 3)

    def __len__(self) -> int:
        return len(self.atomic_numbers)

    def __iter__(self):
        for z, fc in zip(self.atomic_numbers, self.frac_coords):
            yield SimpleNamespace(specie=SimpleNamespace(Z=int(z), symbol=Z_TO_SYMBOL[int(z)]), frac_coords=fc)

    @property
    def sites(self):
        return list(self)

    @property
    def cart_coords(self) -> np.ndarray:
        return self.frac_coords @ self.lattice.matrix

    @property
    def volume(self) -> float:
        return self.lattice.volume

    @property
    def composition(self):
        counts = Counter(Z_TO_SYMBOL[int(z)] for z in self.atomic_numbers)
        # electronegativity order is pymatgen's; alphabetical-by-Z is enough for tracking
        order = sorted(counts, key=lambda s: _FORMULA_ORDER.get(s, SYMBOL_TO_Z[s] + 1000))
        formula = " ".join(f"{s}{counts[s]}" for s in order)
        return SimpleNamespace(formula=formula)

    def copy(self) -> "Structure":
        return Structure(Lattice(self.lattice.matrix.copy()), self.atomic_numbers.copy(), self.frac_coords.copy())

    def make_supercell(self, scaling) -> "Structure":
        """Return an (na, nb, nc) supercell, sites ordered as pymatgen does: every
        original site is replicated over all translations before the next site."""
        na, nb, nc = (int(x) for x in scaling)
        trans = np.array([[i, j, k] for i in range(na) for j in range(nb) for k in range(nc)], dtype=np.float64)
        scale = np.array([na, nb, nc], dtype=np.float64)
        frac, zs = [], []
        for z, fc in zip(self.atomic_numbers, self.frac_coords):
            for t in trans:
                frac.append((fc + t) / scale)
                zs.append(int(z))
        lat = self.lattice.matrix * scale[:, None]
        return Structure(Lattice(lat), zs, np.array(frac))

    def perturb(self, sigma_frac: float, rng: np.random.Generator) -> "Structure":
        """Add N(0, sigma^2) noise to every fractional coordinate (bench config C2)."""
        out = self.copy()
        out.frac_coords = out.frac_coords + rng.normal(0.0, sigma_frac, size=out.frac_coords.shape)
        return out

    def apply_strain(self, strain) -> "Structure":
        strain = np.asarray(strain, dtype=np.float64)
        if strain.ndim == 0:
            strain = np.eye(3) * float(strain)
        elif strain.ndim == 1:
            strain = np.diag(strain)
        out = self.copy()
        out.lattice = Lattice(self.lattice.matrix @ (np.eye(3) + strain))
        return out

    @classmethod
    def from_file(cls, path: str) -> "Structure":
        return cls.from_cif(path)

    @classmethod
    def from_cif(cls, path: str) -> "Structure":
        """Reader for P1 CIFs with explicit sites (both reference fixtures are of this kind)."""
        with open(path) as fh:
            lines = [ln.strip() for ln in fh if ln.strip() and not ln.lstrip().startswith("#")]
        cell = {}
        for ln in lines:
            m = re.match(r"_cell_(length_[abc]|angle_(?:alpha|beta|gamma))\s+([-\d.eE()+]+)", ln)
            if m:
                cell[m.group(1)] = float(re.sub(r"\(.*\)", "", m.group(2)))
        lat = Lattice.from_parameters(cell["length_a"], cell["length_b"], cell["length_c"],
                                      cell["angle_alpha"], cell["angle_beta"], cell["angle_gamma"])
        species, frac = [], []
        i = 0
        while i < len(lines):
            if lines[i] == "loop_":
                j = i + 1
                heads = []
                while j < len(lines) and lines[j].startswith("_"):
                    heads.append(lines[j])
                    j += 1
                if "_atom_site_fract_x" in heads:
                    ix, iy, iz = (heads.index(f"_atom_site_fract_{c}") for c in "xyz")
                    isym = heads.index("_atom_site_type_symbol") if "_atom_site_type_symbol" in heads else heads.index("_atom_site_label")
                    while j < len(lines) and not lines[j].startswith(("_", "loop_", "data_")):
                        tok = lines[j].split()
                        sym = re.match(r"([A-Z][a-z]?)", tok[isym]).group(1)
                        species.append(sym)
                        frac.append([float(re.sub(r"\(.*\)", "", tok[k])) for k in (ix, iy, iz)])
                        j += 1
                i = j
            else:
                i += 1
        if not species:
            raise ValueError(f"no atom sites found in {path}")
        return cls(lat, species, frac)


# rough electronegativity ranks for a pymatgen-like formula string (tracking only)
_FORMULA_ORDER = {s: i for i, s in enumerate(
    "Cs K Rb Ba Na Sr Li Ca La Y Mg Sc Zr Hf Ti Mn Ta Nb V Al Zn Cr Cd In Ga Fe Co Cu Si Ni Ag Sn Hg Ge Bi B Sb Te Mo As P H Ir Ru Os Pd Pt Rh Pb W Au C Se S I Br N Cl O F".split())}


def atomic_numbers_of(structure) -> np.ndarray:
    """Z of every site of a structure-like object.  ``atomic_numbers`` (this module's and pymatgen's
    structures have it) avoids a Python loop over sites, which for a thousand structures costs more
    than building their graphs on the GPU; anything else is read site by site (``site.specie.Z``)."""
    z = getattr(structure, "atomic_numbers", None)
    if z is None:
        z = [site.specie.Z for site in structure]
    return np.asarray(z, dtype=np.int32)
