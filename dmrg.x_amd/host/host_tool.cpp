/** @file host_tool.cpp
    Line-oriented test harness over the engine's host classes (QuantumNumbers, Block::SpinBase, KronBlocks_t,
    KronEye_Explicit, Hamiltonians) -- everything here is index/metadata work on host-resident cells, so it runs
    without a GPU.  Used by tests/test_host_engine.py to replay the reference's known-answer tables
    (tests/UnitTests_DMRGKron.cpp, tests/UnitTests_DMRGBlock.cpp) against the C++ engine. */
#include <iostream>
#include <sstream>
#include <map>
#include "DMRGBlock.hpp"
#include "Hamiltonians.hpp"
#include "DMRGKron.hpp"
#include "CorrelatorDealing.hpp"

static void dump_mat(const char* tag, PetscInt site, const Mat& m)
{
    if (!m) { printf("op %s %lld null\n", tag, LLD(site)); return; }
    const PetscInt n = m->N();
    printf("op %s %lld %lld\n", tag, LLD(site), LLD(n));
    for (PetscInt r = 0; r < n; ++r) {
        const std::vector<double> row = m->dense_row(r);
        printf("row %lld", LLD(r));
        for (PetscInt c = 0; c < n; ++c) if (row[(size_t)c] != 0.0) printf(" %lld:%.17g", LLD(c), row[(size_t)c]);
        printf("\n");
    }
}

int main()
{
    std::map<std::string, Block::SpinBase> blocks;
    Hamiltonians::J1J2XXZModel_SquareLattice ham;
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream is(line);
        std::string cmd;
        if (!(is >> cmd) || cmd[0] == '#') continue;
        PetscErrorCode ierr = 0;
        if (cmd == "block") {
            std::string name; PetscInt nsites, nsec;
            is >> name >> nsites >> nsec;
            std::vector<PetscReal> qn((size_t)nsec); std::vector<PetscInt> sz((size_t)nsec);
            for (auto& q : qn) is >> q;
            for (auto& s : sz) is >> s;
            ierr = blocks[name].Initialize(PETSC_COMM_WORLD, nsites, qn, sz);
            printf("rc %d\n", ierr);
        } else if (cmd == "single") {
            std::string name; is >> name;
            ierr = blocks[name].Initialize(PETSC_COMM_WORLD, 1, PETSC_DEFAULT);
            printf("rc %d\n", ierr);
        } else if (cmd == "set") {
            std::string name, op; PetscInt site, row, col; double val;
            is >> name >> op >> site >> row >> col >> val;
            Mat m = (op == "Sz") ? blocks[name].Sz(site) : blocks[name].Sp(site);
            ierr = m->set(row, col, val);
            printf("rc %d\n", ierr);
        } else if (cmd == "copy") {              /* shallow copy: the new block shares the operator handles */
            std::string a, b; is >> a >> b;
            blocks[b] = blocks[a];
            printf("rc 0\n");
        } else if (cmd == "destroy") {
            std::string name; is >> name;
            ierr = blocks[name].Destroy();
            printf("rc %d\n", ierr);
        } else if (cmd == "nnz") {               /* number of stored cells of Sz(site) as seen through this block object */
            std::string name; PetscInt site; is >> name >> site;
            long n = -1;
            try { Mat m = blocks[name].Sz(site); n = m ? (long)m->cells.size() : -1; } catch (const std::exception&) { n = -2; }
            printf("nnz %ld\n", n);
        } else if (cmd == "save") {
            std::string name, dir; is >> name >> dir;
            ierr = blocks[name].SaveToDisk(dir);
            printf("rc %d\n", ierr);
        } else if (cmd == "load") {
            std::string name, dir; is >> name >> dir;
            ierr = blocks[name].InitializeFromDisk(PETSC_COMM_WORLD, dir);
            printf("rc %d\n", ierr);
        } else if (cmd == "check") {
            std::string name; is >> name;
            ierr = blocks[name].CheckOperatorBlocks();
            printf("rc %d\n", ierr);
        } else if (cmd == "kroneye") {
            std::string l, r, o; is >> l >> r >> o;
            ierr = KronEye_Explicit(blocks[l], blocks[r], {}, blocks[o]);
            printf("rc %d\n", ierr);
        } else if (cmd == "dump") {
            std::string name; is >> name;
            Block::SpinBase& b = blocks[name];
            printf("sectors %lld", LLD(b.Magnetization.NumSectors()));
            for (PetscReal q : b.Magnetization.List()) printf(" %.17g", q);
            for (PetscInt s : b.Magnetization.Sizes()) printf(" %lld", LLD(s));
            printf("\n");
            for (PetscInt i = 0; i < b.NumSites(); ++i) { dump_mat("Sz", i, b.Sz(i)); dump_mat("Sp", i, b.Sp(i)); }
            printf("end\n");
        } else if (cmd == "kronblocks") {
            std::string l, r; is >> l >> r;
            std::vector<PetscReal> qs; PetscReal q;
            while (is >> q) qs.push_back(q);
            KronBlocks_t kb(blocks[l], blocks[r], qs, NULL, 0);
            printf("kronblocks %lld %lld", LLD(kb.size()), LLD(kb.NumStates()));
            for (PetscInt k = 0; k < kb.size(); ++k) printf(" %lld,%lld,%lld,%lld", LLD(kb.LeftIdx(k)), LLD(kb.RightIdx(k)), LLD(kb.Sizes(k)), LLD(kb.Offsets(k)));
            printf("\n");
        } else if (cmd == "iterate") {           /* walk KronBlocksIterator over [istart, iend) of the (sector-filtered) KronBlocks */
            std::string l, r; PetscInt i0, i1; is >> l >> r >> i0 >> i1;
            std::vector<PetscReal> qs; PetscReal q;
            while (is >> q) qs.push_back(q);
            KronBlocks_t kb(blocks[l], blocks[r], qs, NULL, 0);
            if (i1 < 0) i1 = kb.NumStates();
            KronBlocksIterator it(kb, i0, i1);
            printf("iterate %lld %lld\n", LLD(it.IdxStart()), LLD(it.IdxEnd()));
            for (; it.Loop(); ++it)
                printf("it %lld %lld %lld %lld %lld %lld %lld %lld %lld %d %lld %lld %lld\n", LLD(it.Idx()), LLD(it.BlockIdx()), LLD(it.LocIdx()), LLD(it.BlockIdxLeft()),
                       LLD(it.BlockIdxRight()), LLD(it.LocIdxLeft()), LLD(it.LocIdxRight()), LLD(it.GlobalIdxLeft()), LLD(it.GlobalIdxRight()), (int)it.UpdatedBlock(),
                       LLD(it.Steps()), LLD(it.BlockStartIdx(0)), LLD(it.BlockSize(+1)));
            printf("end\n");
        } else if (cmd == "ham") {
            dmrgx_host::Options::Global().Clear();
            std::string k, v;
            while (is >> k >> v) dmrgx_host::Options::Global().Set(k[0] == '-' ? k.substr(1) : k, v == "_" ? "" : v);
            ham = Hamiltonians::J1J2XXZModel_SquareLattice();
            ierr = ham.SetFromOptions();
            printf("rc %d\n", ierr);
        } else if (cmd == "terms") {
            PetscInt n; is >> n;
            const std::vector<Hamiltonians::Term> T = ham.H(n < 0 ? PETSC_DEFAULT : n);
            printf("terms %zu", T.size());
            for (const auto& t : T) printf(" %.17g,%d,%lld,%d,%lld", t.a, (int)t.Iop, LLD(t.Isite), (int)t.Jop, LLD(t.Jsite));
            printf("\n");
        } else if (cmd == "snake") {
            PetscInt ns = ham.NumSites();
            printf("snake");
            for (PetscInt i = 0; i < ns; ++i) { PetscInt ix, jy; ham.To2D(i, ix, jy); printf(" %lld,%lld,%lld", LLD(ix), LLD(jy), LLD(ham.To1D(ix, jy))); }
            printf("\n");
        } else if (cmd == "qnrange") {
            std::string name; PetscInt blk, shift; is >> name >> blk >> shift;
            PetscInt s = 0, e = 0; PetscBool flg = PETSC_FALSE;
            ierr = blocks[name].Magnetization.OpBlockToGlobalRange(blk, shift, s, e, flg);
            printf("rc %d %lld %lld %d\n", ierr, LLD(s), LLD(e), (int)flg);
        } else if (cmd == "deal") {              /* deal N W ncorr, then per correlator: k site_1 .. site_k  ->  owners and carried weights */
            int64_t N; int W; size_t nc;
            is >> N >> W >> nc;
            std::vector<std::vector<int64_t>> sites(nc);
            for (auto& v : sites) { size_t k; is >> k; v.resize(k); for (auto& x : v) is >> x; }
            std::vector<double> carried;
            const std::vector<int> owner = dmrgx_host::DealCorrelators(sites, N, W, &carried);
            printf("owners");
            for (int o : owner) printf(" %d", o);
            printf("\ncarried");
            for (double c : carried) printf(" %.17g", c);
            printf("\n");
        } else printf("unknown %s\n", cmd.c_str());
        fflush(stdout);
    }
    return 0;
}
