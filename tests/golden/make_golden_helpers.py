"""Shared by make_golden.py and the live-reference test: what the reference converter touches on a pymatgen
Structure (chgnet/graph/converter.py:120-134, 162, 187), served from one of our structures plus a neighbour list."""


class DuckStructure:
    def __init__(self, s, nl: dict) -> None:
        self._s, self._nl = s, nl
        self.frac_coords = s.frac_coords
        self.lattice = s.lattice
        self.sites = s.sites
        self.composition = s.composition

    def __len__(self):
        return len(self._s)

    def __iter__(self):
        return iter(self._s)

    def get_neighbor_list(self, r, sites=None, numerical_tol=1e-8):  # noqa: ARG002
        nl = self._nl
        return nl["center"], nl["neighbor"], nl["image"], nl["distance"]
