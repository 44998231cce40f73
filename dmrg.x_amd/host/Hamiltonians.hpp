/** @file Hamiltonians.hpp
    Lattice geometry and two-site term list of the spin-1/2 J1-J2 XXZ model on the Lx x Ly square lattice laid out as
    an S-shaped snake.  Public interface, option names, defaults and the reference's documented quirks follow
    reference include/Hamiltonians.hpp:19-26,77-288 and src/Hamiltonians.cpp:4-147; host-only integer code that hands
    the superblock plan its term list. */
#ifndef DMRGX_HAMILTONIANS_HPP
#define DMRGX_HAMILTONIANS_HPP

#include <vector>
#include <string>
#include "petsc_compat.hpp"
#include "DMRGBlock.hpp"

namespace Hamiltonians
{
    /** a * Iop(Isite) (x) Jop(Jsite) */
    struct Term
    {
        PetscScalar a;
        Op_t        Iop;
        PetscInt    Isite;
        Op_t        Jop;
        PetscInt    Jsite;
    };

    typedef enum { OpenBC = 0, PeriodicBC = 1 } BC_t;

    class J1J2XXZModel_SquareLattice
    {
    public:
        J1J2XXZModel_SquareLattice() {}

        /** -J1 -J2 -Jz1 -Jz2 -Lx -Ly -heisenberg -BCopen -BCperiodic  (defaults: 4x4, J1=J2=1, Jz=0, cylinder) */
        PetscErrorCode SetFromOptions()
        {
            PetscErrorCode ierr;
            ierr = PetscOptionsGetReal(NULL, NULL, "-J1", &_J1, NULL); CHKERRQ(ierr);
            ierr = PetscOptionsGetReal(NULL, NULL, "-J2", &_J2, NULL); CHKERRQ(ierr);
            ierr = PetscOptionsGetReal(NULL, NULL, "-Jz1", &_Jz1, NULL); CHKERRQ(ierr);
            ierr = PetscOptionsGetReal(NULL, NULL, "-Jz2", &_Jz2, NULL); CHKERRQ(ierr);
            ierr = PetscOptionsGetInt(NULL, NULL, "-Lx", &_Lx, NULL); CHKERRQ(ierr);
            ierr = PetscOptionsGetInt(NULL, NULL, "-Ly", &_Ly, NULL); CHKERRQ(ierr);
            ierr = PetscOptionsGetBool(NULL, NULL, "-verbose", &verbose, NULL); CHKERRQ(ierr);
            ierr = PetscOptionsGetReal(NULL, NULL, "-heisenberg", &_Jz1, &heisenberg); CHKERRQ(ierr);
            if (heisenberg) { _J1 = 0.50; _J2 = 0.0; _Jz2 = 0.0; }   /* H = sum J (S+S- + S-S+) + Jz SzSz */
            PetscBool BCopen = PETSC_FALSE, BCperiodic = PETSC_FALSE;
            ierr = PetscOptionsGetBool(NULL, NULL, "-BCopen", &BCopen, NULL); CHKERRQ(ierr);
            if (BCopen) { _BCx = OpenBC; _BCy = OpenBC; }
            ierr = PetscOptionsGetBool(NULL, NULL, "-BCperiodic", &BCperiodic, NULL); CHKERRQ(ierr);
            if (BCperiodic) { _BCx = PeriodicBC; _BCy = PeriodicBC; }
            set_from_options = PETSC_TRUE;
            H_full_filled = PETSC_FALSE;
            return 0;
        }
        PetscErrorCode SaveAsOptions(const std::string& filename)
        {
            FILE* fp = fopen(filename.c_str(), "w");
            if (!fp) SETERRQ1(PETSC_COMM_SELF, PETSC_ERR_FILE_OPEN, "cannot open %s", filename.c_str());
            fprintf(fp, "-Lx %lld\n-Ly %lld\n-J1 %.20g\n-Jz1 %.20g\n-J2 %.20g\n-Jz2 %.20g\n", LLD(_Lx), LLD(_Ly), _J1, _Jz1, _J2, _Jz2);
            if (_BCx == OpenBC && _BCy == OpenBC) fprintf(fp, "-BCopen 1\n");
            if (_BCx == PeriodicBC && _BCy == PeriodicBC) fprintf(fp, "-BCperiodic 1\n");
            fclose(fp);
            return 0;
        }
        void PrintOut() const
        {
            printf("HAMILTONIAN: J1J2XXZModel_SquareLattice\n  Lx=%lld Ly=%lld J1=%g Jz1=%g J2=%g Jz2=%g BCx=%s BCy=%s\n",
                   LLD(_Lx), LLD(_Ly), _J1, _Jz1, _J2, _Jz2, _BCx ? "periodic" : "open", _BCy ? "periodic" : "open");
        }
        void SaveOut(FILE* fp) const
        {
            fprintf(fp, "  \"Hamiltonian\": {\n    \"label\":\"J1J2XXZModel_SquareLattice\",\n    \"parameters\": {\n"
                        "      \"Lx\": %lld,\n      \"Ly\": %lld,\n      \"J1\": %g,\n      \"Jz1\": %g,\n      \"J2\": %g,\n      \"Jz2\": %g,\n"
                        "      \"BCx\": \"%s\",\n      \"BCy\": \"%s\"\n    }\n  }",
                    LLD(_Lx), LLD(_Ly), _J1, _Jz1, _J2, _Jz2, _BCx ? "periodic" : "open", _BCy ? "periodic" : "open");
        }

        PetscInt Lx() const { return _Lx; }
        PetscInt Ly() const { return _Ly; }
        PetscInt NumSites() const { return _Lx * _Ly; }
        /** sites of one column: the unit by which the warm-up grows its environment */
        PetscInt NumEnvSites() const { return _Ly; }

        /** (ix,jy) -> position on the snake: even columns run up, odd columns run down */
        PetscInt To1D(const PetscInt ix, const PetscInt jy) const
        {
            return (ix % 2 == 0) ? ix * _Ly + jy : (ix + 1) * _Ly - (jy + 1);
        }
        PetscErrorCode To2D(const PetscInt idx, PetscInt& ix, PetscInt& jy) const
        {
            ix = idx / _Ly;
            jy = (ix % 2 == 0) ? idx % _Ly : _Ly - 1 - idx % _Ly;
            return 0;
        }

        /** Terms among the first nsites_in snake sites (PETSC_DEFAULT: the whole lattice, cached). */
        std::vector<Term> H(const PetscInt& nsites_in = PETSC_DEFAULT)
        {
            const PetscInt ns = (nsites_in == PETSC_DEFAULT) ? _Lx * _Ly : nsites_in;
            const bool full = (ns == _Lx * _Ly);
            if (full && H_full_filled) return H_full;
            std::vector<Term> T;
            T.reserve((size_t)ns * 8);
            for (PetscInt is = 0; is < ns; ++is) {
                PetscInt ix, jy;
                To2D(is, ix, jy);
                if (_J1 != 0.0 || _Jz1 != 0.0)
                    for (PetscInt in : NearestNeighbors(ix, jy, ns)) AddBond(T, _J1, _Jz1, std::min(in, is), std::max(in, is));
                /* reference behaviour: next-nearest terms only when BOTH J2 and Jz2 are non-zero */
                if ((_J2 != 0.0 && _Jz2 != 0.0) && _Lx > 1 && _Ly > 1)
                    for (PetscInt in : NextNearestNeighbors(ix, jy, ns)) AddBond(T, _J2, _Jz2, std::min(in, is), std::max(in, is));
            }
            if (full) { H_full = T; H_full_filled = PETSC_TRUE; }
            return T;
        }

        /** all nearest-neighbour pairs of the full lattice */
        std::vector<std::vector<PetscInt>> NeighborPairs(const PetscInt d = 1) const
        {
            if (d != 1) CPP_CHKERRQ_MSG(1, "Only d=1 supported.");
            std::vector<std::vector<PetscInt>> out;
            const PetscInt ns = _Lx * _Ly;
            for (PetscInt is = 0; is < ns; ++is) {
                PetscInt ix, jy;
                To2D(is, ix, jy);
                for (PetscInt in : NearestNeighbors(ix, jy, ns)) out.push_back({std::min(in, is), std::max(in, is)});
            }
            return out;
        }

    private:
        static void AddBond(std::vector<Term>& T, PetscScalar J, PetscScalar Jz, PetscInt ia, PetscInt ib)
        {
            if (J != 0.0) { T.push_back({J, OpSp, ia, OpSm, ib}); T.push_back({J, OpSm, ia, OpSp, ib}); }
            if (Jz != 0.0) T.push_back({Jz, OpSz, ia, OpSz, ib});
        }
        /** "above" then "right" neighbour, kept only if it lies among the first ns sites */
        std::vector<PetscInt> NearestNeighbors(PetscInt ix, PetscInt jy, PetscInt ns) const
        {
            std::vector<PetscInt> nn;
            if (jy < _Ly - 1 || (jy == _Ly - 1 && _BCy == PeriodicBC)) {
                const PetscInt ja = (jy + 1) % _Ly, n1 = To1D(ix, ja);
                if (n1 < ns && ja != jy) nn.push_back(n1);
            }
            if (ix < _Lx - 1 || (ix == _Lx - 1 && _BCx == PeriodicBC)) {
                const PetscInt ir = (ix + 1) % _Lx, n1 = To1D(ir, jy);
                if (n1 < ns && ir != ix) nn.push_back(n1);
            }
            return nn;
        }
        /** "upper-left" then "upper-right" diagonal neighbour */
        std::vector<PetscInt> NextNearestNeighbors(PetscInt ix, PetscInt jy, PetscInt ns) const
        {
            std::vector<PetscInt> out;
            const bool up = jy < _Ly - 1 || (jy == _Ly - 1 && _BCy == PeriodicBC);
            if (up && (ix >= 1 || _BCx == PeriodicBC)) { const PetscInt n1 = To1D((ix + _Lx - 1) % _Lx, (jy + 1) % _Ly); if (n1 < ns) out.push_back(n1); }
            if (up && (ix < _Lx - 1 || _BCx == PeriodicBC)) { const PetscInt n1 = To1D((ix + 1) % _Lx, (jy + 1) % _Ly); if (n1 < ns) out.push_back(n1); }
            return out;
        }

        PetscBool heisenberg = PETSC_FALSE, set_from_options = PETSC_FALSE, verbose = PETSC_FALSE;
        PetscScalar _Jz1 = 0.0, _Jz2 = 0.0, _J1 = 1.0, _J2 = 1.0;
        PetscInt _Lx = 4, _Ly = 4;
        BC_t _BCx = OpenBC, _BCy = PeriodicBC;
        std::vector<Term> H_full;
        PetscBool H_full_filled = PETSC_FALSE;
    };
}

#endif
