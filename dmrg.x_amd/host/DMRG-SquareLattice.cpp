/** @file DMRG-SquareLattice.cpp
    DMRG executable for the spin-1/2 J1-J2 XXZ model on the square lattice, MI355X engine.  Same call sequence as the
    reference driver (reference src/DMRG-SquareLattice.cpp:15-37): Initialize -> register correlators -> Warmup ->
    Sweeps -> Destroy, against the same class API, so the reference's own driver source also compiles on these
    headers (`make dropin-check`). */
static char help[] = "DMRG executable for the Spin-1/2 J1-J2 XXZ model on a two-dimensional square lattice (MI355X engine).\n";

#include "DMRGBlock.hpp"
#include "Hamiltonians.hpp"
#include "DMRGBlockContainer.hpp"

typedef DMRGBlockContainer<Block::SpinBase, Hamiltonians::J1J2XXZModel_SquareLattice> DMRG_t;

/** The measurements of the reference driver (reference src/DMRG-SquareLattice.cpp:40-185): <Sz_i> on every system
    site, the three bond correlators of every nearest-neighbour pair (their J-weighted sum is the bond energy), the
    Sz string along row 1, the two "Polyakov" columns and the interior "Wilson" loop -- the string operators only on
    lattices that have those rows and columns. */
static PetscErrorCode RegisterCorrelators(DMRG_t& DMRG)
{
    PetscErrorCode ierr;
    const auto& Ham = DMRG.HamiltonianRef();
    const PetscInt Lx = Ham.Lx(), Ly = Ham.Ly();
    auto label = [&](Op_t t, PetscInt idx) {
        PetscInt ix = 0, jy = 0;
        Ham.To2D(idx, ix, jy);
        return OpToStr(t) + "_{" + std::to_string(ix) + "," + std::to_string(jy) + "} ";
    };
    auto string_of_sz = [&](const std::vector<PetscInt>& sites, const std::string& name) -> PetscErrorCode {
        if (sites.empty()) return 0;
        std::vector<Op> ops;
        std::string desc = "< ";
        for (PetscInt idx : sites) { ops.push_back({OpSz, idx}); desc += label(OpSz, idx); }
        return DMRG.SetUpCorrelation(ops, name, desc + ">");
    };
    for (PetscInt idx = 0; idx < Lx * Ly / 2; ++idx) {
        ierr = DMRG.SetUpCorrelation({{OpSz, idx}}, "Magnetization(" + std::to_string(idx) + ")", "< " + label(OpSz, idx) + ">"); CHKERRQ(ierr);
    }
    const Op_t kinds[3][2] = {{OpSz, OpSz}, {OpSp, OpSm}, {OpSm, OpSp}};
    for (const std::vector<PetscInt>& pair : Ham.NeighborPairs()) {
        if (pair.size() != 2) SETERRQ1(PETSC_COMM_WORLD, 1, "Invalid 2-point correlator. Got %lu operators instead.", (unsigned long)pair.size());
        for (const auto& k : kinds) {
            const std::string name = "NearestNeighbor" + OpToStr(k[0]) + OpToStr(k[1]) + "( " + std::to_string(pair[0]) + " " + std::to_string(pair[1]) + " )";
            ierr = DMRG.SetUpCorrelation({{k[0], pair[0]}, {k[1], pair[1]}}, name, "< " + label(k[0], pair[0]) + label(k[1], pair[1]) + ">"); CHKERRQ(ierr);
        }
    }
    if (Ly >= 2) {
        std::vector<PetscInt> row;
        for (PetscInt ix = 0; ix < Lx; ++ix) row.push_back(Ham.To1D(ix, 1));
        ierr = string_of_sz(row, "MagnetizationRowX1"); CHKERRQ(ierr);
    }
    if (Lx >= 3) {
        std::vector<PetscInt> c1, c2;
        for (PetscInt jy = 0; jy < Ly; ++jy) { c1.push_back(Ham.To1D(1, jy)); c2.push_back(Ham.To1D(Lx - 2, jy)); }
        ierr = string_of_sz(c1, "Polyakov"); CHKERRQ(ierr);
        ierr = string_of_sz(c2, "Polyakov2"); CHKERRQ(ierr);
    }
    if (Lx >= 4 && Ly >= 4) {                       /* loop around the interior, clockwise from (1,1) */
        std::vector<PetscInt> loop;
        for (PetscInt jy = 1; jy < Ly - 2; ++jy) loop.push_back(Ham.To1D(1, jy));
        for (PetscInt ix = 1; ix < Lx - 2; ++ix) loop.push_back(Ham.To1D(ix, Ly - 2));
        for (PetscInt jy = Ly - 2; jy > 1; --jy) loop.push_back(Ham.To1D(Lx - 2, jy));
        for (PetscInt ix = Lx - 2; ix > 1; --ix) loop.push_back(Ham.To1D(ix, 1));
        ierr = string_of_sz(loop, "Wilson"); CHKERRQ(ierr);
    }
    return 0;
}

int main(int argc, char** argv)
{
    PetscErrorCode ierr;
    PetscMPIInt nprocs, rank;
    ierr = SlepcInitialize(&argc, &argv, (char*)0, help); CHKERRQ(ierr);
    ierr = MPI_Comm_size(PETSC_COMM_WORLD, &nprocs); CHKERRQ(ierr);
    ierr = MPI_Comm_rank(PETSC_COMM_WORLD, &rank); CHKERRQ(ierr);
    int32_t ndev = 0;
    if (dmrgx_device_count(&ndev)) { fprintf(stderr, "%s\n", dmrgx_last_error()); return DMRGX_ERR_DEVICE; }   /* no CPU fallback */
    {
        DMRG_t DMRG(PETSC_COMM_WORLD);
        ierr = DMRG.Initialize(); CHKERRQ(ierr);
        ierr = RegisterCorrelators(DMRG); CHKERRQ(ierr);
        ierr = DMRG.Warmup(); CHKERRQ(ierr);
        ierr = DMRG.Sweeps(); CHKERRQ(ierr);
        if (!rank) printf("FINAL GSEnergy %.14g  MatMults %lld  ranks %d\n", DMRG.GSEnergy(), LLD(DMRG.TotalMatMults()), nprocs);
        ierr = DMRG.Destroy(); CHKERRQ(ierr);
    }
    ierr = SlepcFinalize(); CHKERRQ(ierr);
    return 0;
}
