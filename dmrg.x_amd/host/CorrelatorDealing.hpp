#ifndef DMRGX_HOST_CORRELATOR_DEALING_HPP
#define DMRGX_HOST_CORRELATOR_DEALING_HPP
/** @file CorrelatorDealing.hpp
    Which rank measures which correlator (engine extension, multi-rank runs; DMRGBlockContainer::BuildNeedTables).

    A rank keeps -- and rotates at every step of the way back to the centre -- only the site operators ITS correlators read.  Site i
    of the half-lattice block (N / 2 sites) is rotated N / 2 - i times on the way back, so the one- and two-site correlators are cut
    into W consecutive runs by their lowest site such that the largest carried weight -- partner sites included -- is minimal (low
    sites are dear: rank 0's run is the shortest); the longer strings (row, columns, loop) are handed to the last ranks first, whose
    runs then shrink accordingly.  Pure index arithmetic: also replayed on
    the CPU by dmrgx-host-tool (tests/test_host_engine.py).  The reference splits this work the other way round, rotating operators
    on sub-communicators (src/DMRGBlock.cpp:761-773, -rot_nsubcomm). */
#include <algorithm>
#include <cstdint>
#include <vector>

namespace dmrgx_host {

/** sites[c] = block-local site indices correlator c reads (system and environment operators alike); N = sites of the lattice.
    Returns owner[c] in [0, W); W <= 1: all -1 ("measured by every rank").  carried (optional): per rank, sum over its sites of
    the times each is rotated. */
inline std::vector<int> DealCorrelators(const std::vector<std::vector<int64_t>>& sites, int64_t N, int W, std::vector<double>* carried_out = nullptr)
{
    std::vector<int> owner(sites.size(), -1);
    if (carried_out) carried_out->assign((size_t)std::max(W, 1), 0.0);
    if (W <= 1 || sites.empty()) return owner;
    const int64_t H = std::max<int64_t>(N / 2, 1);
    auto weight = [&](int64_t i) { return (double)std::max<int64_t>(H - i, 1); };
    std::vector<std::vector<char>> has((size_t)W, std::vector<char>((size_t)std::max<int64_t>(N, 1), 0));
    std::vector<double> carried((size_t)W, 0.0);
    auto give = [&](size_t c, int r) {
        owner[c] = r;
        for (int64_t i : sites[c]) if (i >= 0 && i < N && !has[(size_t)r][(size_t)i]) { has[(size_t)r][(size_t)i] = 1; carried[(size_t)r] += weight(i); }
    };
    // ---- one- and two-site correlators: consecutive runs of lowest sites; the run boundaries minimise the largest carried weight,
    //      which counts the partner sites a run drags along (a pair reaches up to Ly sites ahead) -- bisection on the bound, greedy fill
    std::vector<std::vector<size_t>> by_low((size_t)std::max<int64_t>(N, 1));
    for (size_t c = 0; c < sites.size(); ++c) {
        const std::vector<int64_t>& v = sites[c];
        if (v.empty()) { owner[c] = 0; continue; }
        if (v.size() > 2) continue;
        int64_t lo = *std::min_element(v.begin(), v.end());
        lo = std::min<int64_t>(std::max<int64_t>(lo, 0), N - 1);
        by_low[(size_t)lo].push_back(c);
    }
    std::vector<int64_t> lows;
    for (int64_t l = 0; l < N; ++l) if (!by_low[(size_t)l].empty()) lows.push_back(l);
    // ---- the strings (row, columns, loop) first, heaviest to the last rank, the next to the one before, ...: a string drags sites
    //      from the whole block along whoever measures it; the last ranks' runs lie at the cheap end of the block and give way
    {
        std::vector<std::pair<double, size_t>> strings;
        for (size_t c = 0; c < sites.size(); ++c) {
            if (sites[c].size() <= 2) continue;
            double w = 0;
            std::vector<char> seen((size_t)N, 0);
            for (int64_t i : sites[c]) if (i >= 0 && i < N && !seen[(size_t)i]) { seen[(size_t)i] = 1; w += weight(i); }
            strings.push_back({w, c});
        }
        std::stable_sort(strings.begin(), strings.end(), [](const std::pair<double, size_t>& a, const std::pair<double, size_t>& b) { return a.first > b.first; });
        for (size_t q = 0; q < strings.size(); ++q) give(strings[q].second, W - 1 - (int)(q % (size_t)W));
    }
    // ranks needed when no rank may carry more than `bound` (every rank takes at least one lowest site); fills `cut` with the first
    // index into `lows` of every rank
    auto fill = [&](double bound, std::vector<size_t>* cut) {
        int used = 0;
        size_t q = 0;
        if (cut) cut->clear();
        while (q < lows.size()) {
            if (cut) cut->push_back(q);
            std::vector<char> in = has[(size_t)std::min(used, W - 1)];          // (what the rank's strings already make it carry)
            double wsum = carried[(size_t)std::min(used, W - 1)];
            size_t q0 = q;
            for (; q < lows.size(); ++q) {
                double add = 0;
                std::vector<int64_t> fresh;
                for (size_t c : by_low[(size_t)lows[q]]) for (int64_t i : sites[c]) if (i >= 0 && i < N && !in[(size_t)i]) { in[(size_t)i] = 1; fresh.push_back(i); add += weight(i); }
                if (q > q0 && wsum + add > bound) { for (int64_t i : fresh) in[(size_t)i] = 0; break; }
                wsum += add;
            }
            ++used;
        }
        return used;
    };
    if (!lows.empty()) {
        double lo_b = 0, hi_b = 0;
        for (int64_t i = 0; i < N; ++i) hi_b += weight(i);
        for (int it = 0; it < 50; ++it) { const double mid = 0.5 * (lo_b + hi_b); if (fill(mid, nullptr) <= W) hi_b = mid; else lo_b = mid; }
        std::vector<size_t> cut;
        fill(hi_b, &cut);
        for (size_t r = 0; r < cut.size(); ++r) {
            const size_t qe = r + 1 < cut.size() ? cut[r + 1] : lows.size();
            for (size_t q = cut[r]; q < qe; ++q) for (size_t c : by_low[(size_t)lows[q]]) give(c, (int)std::min<size_t>(r, (size_t)W - 1));
        }
    }
    if (carried_out) *carried_out = carried;
    return owner;
}

}  // namespace dmrgx_host
#endif
