"""CPU oracle for the DMRG.x hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy/scipy + a small C library, ``kron_ref.c``) of the
reference's algorithm for the superblock MatMult / eigensolve / RDM-truncation / rotation path.
It is the *checker* used by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  Nothing under ``dmrg.x_amd/`` (the product) may import, link or execute it.

Pinning status (see DESIGN.md "Oracle"):
  * index arithmetic (sector-pair ordering, merged-sector ordering, ``idx=(idx_L-bks_L)*NR'+(idx_R-bks_R)+fws``)
    is pinned by the reference's own known-answer tables (tests/UnitTests_DMRGKron.cpp:49-245,
    tests/UnitTests_Misc.cpp:82-136, tests/UnitTests_DMRGBlock.cpp:84-114), transcribed to
    ``tests/golden/*.json``;
  * single-site operators are pinned by src/DMRGBlock.cpp:1131-1136,1193-1195;
  * the eigensolver / LAPACK boundary (SLEPc 3.8.3, PETSc 3.8.4 -- not under /root/reference, pinned only
    by prose in docs/doc_01_installation.dox:10-15) has NO golden vector in the reference: **parity unpinned**
    for solver tolerance, restart and start vector.  Energies are therefore anchored on independent exact
    diagonalisation and on algebraic invariants instead.
The reference itself cannot be built here (needs PETSc/SLEPc/MPI + PETSc private headers,
src/DMRGKron.cpp:9-10), so there is no ``oracle/_ref``.
"""
from .qn import QuantumNumbers, OpSm, OpSz, OpSp, OpEye  # noqa: F401
