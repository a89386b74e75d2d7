"""Structure -> CrystalGraph (host side, native builder in csrc/host_graph.cpp)."""

from chgnet_amd.graph.converter import CrystalGraphConverter
from chgnet_amd.graph.crystalgraph import CrystalGraph
from chgnet_amd.graph.structure import Lattice, Structure

__all__ = ["CrystalGraph", "CrystalGraphConverter", "Lattice", "Structure"]
