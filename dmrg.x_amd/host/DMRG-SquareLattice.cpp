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

/** <Sz_i> on every system site and the three bond correlators of every nearest-neighbour pair */
static PetscErrorCode RegisterCorrelators(DMRG_t& DMRG)
{
    PetscErrorCode ierr;
    const PetscInt nsys = DMRG.HamiltonianRef().Lx() * DMRG.HamiltonianRef().Ly() / 2;
    for (PetscInt idx = 0; idx < nsys; ++idx) {
        PetscInt ix, jy;
        ierr = DMRG.HamiltonianRef().To2D(idx, ix, jy); CHKERRQ(ierr);
        ierr = DMRG.SetUpCorrelation({{OpSz, idx}}, "Magnetization(" + std::to_string(idx) + ")",
                                     "< Sz_{" + std::to_string(ix) + "," + std::to_string(jy) + "} >"); CHKERRQ(ierr);
    }
    for (const std::vector<PetscInt>& pair : DMRG.HamiltonianRef().NeighborPairs()) {
        const std::string tag = "(" + std::to_string(pair[0]) + "," + std::to_string(pair[1]) + ")";
        ierr = DMRG.SetUpCorrelation({{OpSz, pair[0]}, {OpSz, pair[1]}}, "SzSz" + tag, "< Sz Sz >"); CHKERRQ(ierr);
        ierr = DMRG.SetUpCorrelation({{OpSp, pair[0]}, {OpSm, pair[1]}}, "SpSm" + tag, "< S+ S- >"); CHKERRQ(ierr);
        ierr = DMRG.SetUpCorrelation({{OpSm, pair[0]}, {OpSp, pair[1]}}, "SmSp" + tag, "< S- S+ >"); CHKERRQ(ierr);
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
        printf("FINAL GSEnergy %.14g  MatMults %lld\n", DMRG.GSEnergy(), LLD(DMRG.TotalMatMults()));
        ierr = DMRG.Destroy(); CHKERRQ(ierr);
    }
    ierr = SlepcFinalize(); CHKERRQ(ierr);
    return 0;
}
