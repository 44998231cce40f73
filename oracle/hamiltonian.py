"""Oracle: J1-J2 XXZ model on the Lx x Ly square lattice traversed as an S-snake.

Restates src/Hamiltonians.cpp:4-147 and include/Hamiltonians.hpp:19-26,89-118,241-265.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from collections import namedtuple

from .qn import OpSm, OpSz, OpSp

# include/Hamiltonians.hpp:19-26
Term = namedtuple("Term", "a Iop Isite Jop Jsite")

OpenBC, PeriodicBC = 0, 1


class J1J2XXZModel_SquareLattice:
    def __init__(self, Lx=4, Ly=4, J1=1.0, J2=1.0, Jz1=0.0, Jz2=0.0, heisenberg=None,
                 BCopen=False, BCperiodic=False):
        # defaults: include/Hamiltonians.hpp:241-265; cylinder = open x, periodic y
        self._Lx, self._Ly = int(Lx), int(Ly)
        self._J1, self._J2, self._Jz1, self._Jz2 = float(J1), float(J2), float(Jz1), float(Jz2)
        if heisenberg is not None:  # include/Hamiltonians.hpp:102-108
            self._Jz1 = float(heisenberg)
            self._J1, self._J2, self._Jz2 = 0.5, 0.0, 0.0
        self._BCx, self._BCy = OpenBC, PeriodicBC
        if BCopen:
            self._BCx = self._BCy = OpenBC
        if BCperiodic:
            self._BCx = self._BCy = PeriodicBC
        self._H_full = None

    def Lx(self):
        return self._Lx

    def Ly(self):
        return self._Ly

    def NumSites(self):
        return self._Lx * self._Ly

    def NumEnvSites(self):
        return self._Ly

    def To1D(self, ix, jy):
        """src/Hamiltonians.cpp:4 (SSNAKE_2D_1D)."""
        Ly = self._Ly
        return (ix * Ly + jy) * (1 - (ix % 2)) + ((ix + 1) * Ly - (jy + 1)) * (ix % 2)

    def To2D(self, idx):
        """src/Hamiltonians.cpp:14-24."""
        ix = idx // self._Ly
        t1 = ix % 2
        jy = (idx % self._Ly) * (1 - 2 * t1) + (self._Ly - 1) * t1
        return ix, jy

    def _nn(self, ix, jy, ns):
        """src/Hamiltonians.cpp:26-46."""
        Lx, Ly = self._Lx, self._Ly
        nn = []
        if (0 <= jy < Ly - 1) or (jy == Ly - 1 and self._BCy == PeriodicBC):
            jy_above = (jy + 1) % Ly
            n1 = self.To1D(ix, jy_above)
            if n1 < ns and jy_above != jy:
                nn.append(n1)
        if (0 <= ix < Lx - 1) or (ix == Lx - 1 and self._BCx == PeriodicBC):
            ix_right = (ix + 1) % Lx
            n1 = self.To1D(ix_right, jy)
            if n1 < ns and ix_right != ix:
                nn.append(n1)
        return nn

    def _nnn(self, ix, jy, ns):
        """src/Hamiltonians.cpp:48-68."""
        Lx, Ly = self._Lx, self._Ly
        out = []
        ycond = (0 <= jy < Ly - 1) or (jy == Ly - 1 and self._BCy == PeriodicBC)
        if ((1 <= ix < Lx) or (ix == 0 and self._BCx == PeriodicBC)) and ycond:
            n1 = self.To1D((ix + Lx - 1) % Lx, (jy + 1) % Ly)
            if n1 < ns:
                out.append(n1)
        if ((0 <= ix < Lx - 1) or (ix == Lx - 1 and self._BCx == PeriodicBC)) and ycond:
            n1 = self.To1D((ix + 1) % Lx, (jy + 1) % Ly)
            if n1 < ns:
                out.append(n1)
        return out

    def H(self, nsites_in=None):
        """Term list for the first ``nsites_in`` snake sites (src/Hamiltonians.cpp:70-122)."""
        ns = self._Lx * self._Ly if nsites_in is None else int(nsites_in)
        full = ns == self._Lx * self._Ly
        if full and self._H_full is not None:
            return list(self._H_full)
        J1, J2, Jz1, Jz2 = self._J1, self._J2, self._Jz1, self._Jz2
        terms = []
        for s in range(ns):
            ix, jy = self.To2D(s)
            if J1 != 0.0 or Jz1 != 0.0:
                for n in self._nn(ix, jy, ns):
                    ia, ib = min(n, s), max(n, s)
                    if J1 != 0.0:
                        terms.append(Term(J1, OpSp, ia, OpSm, ib))
                        terms.append(Term(J1, OpSm, ia, OpSp, ib))
                    if Jz1 != 0.0:
                        terms.append(Term(Jz1, OpSz, ia, OpSz, ib))
            # quirk (:101): NNN terms only if BOTH J2 and Jz2 are non-zero
            if (J2 != 0.0 and Jz2 != 0.0) and self._Lx > 1 and self._Ly > 1:
                for n in self._nnn(ix, jy, ns):
                    il, ir = min(n, s), max(n, s)
                    if J2 != 0.0:
                        terms.append(Term(J2, OpSp, il, OpSm, ir))
                        terms.append(Term(J2, OpSm, il, OpSp, ir))
                    if Jz2 != 0.0:
                        terms.append(Term(Jz2, OpSz, il, OpSz, ir))
        if full:
            self._H_full = list(terms)
        return terms

    def NeighborPairs(self):
        """src/Hamiltonians.cpp:124-147."""
        ns = self._Lx * self._Ly
        out = []
        for s in range(ns):
            ix, jy = self.To2D(s)
            for n in self._nn(ix, jy, ns):
                out.append([min(n, s), max(n, s)])
        return out
