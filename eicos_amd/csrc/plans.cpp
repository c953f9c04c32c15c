// Host-side construction of the sliced-ELL triangular-solve plans.
#include "plans.hpp"
#include "envknob.hpp"

#include <algorithm>

namespace eicos {

// Append slice `m`, cut into sub-slices of at most ELL_KMAX entries per lane (the depth the kernels prefetch);
// the slot layout (off + k*lanes + lane) is unchanged, sub-slice j simply starts at k = j*ELL_KMAX.
static void push_subslices(std::vector<SliceMeta> &sl, SliceMeta m) {
    const int lanes = m.cnt << m.lg, K = m.K;
    if (K <= ELL_KMAX) { sl.push_back(m); return; }
    for (int k0 = 0; k0 < K; k0 += ELL_KMAX) {
        SliceMeta s = m;
        s.K = std::min(ELL_KMAX, K - k0);
        s.off = m.off + k0 * lanes;
        s.newlev = (k0 == 0) ? m.newlev : 0;
        s.cont = (k0 > 0);
        s.more = (k0 + ELL_KMAX < K);
        sl.push_back(s);
    }
}

TriPlan build_tri_plan(const Symbolic &S, int T, bool forward, bool allow_solo) {
    TriPlan pl;
    pl.pos.assign(S.nnzL, 0);
    const std::vector<int> &ptr = forward ? S.Rp : S.Lp;
    const std::vector<int> &ind = forward ? S.Rj : S.Li;
    // hybrid: the sweeps cover the levels below the cut; a row of the top block keeps only its entries in columns < n0
    // (a prefix: the columns of a row ascend), a column below the cut keeps all its rows
    // (dense apex, Symbolic::apex0: the same cut on the scalar path -- the block above it is swept by apex_solve instead of the tile sweeps)
    const bool apex = S.tile == 0 && S.apex0 >= 0;
    const bool hyb = S.tile == 2 || apex;
    const int cut_n0 = apex ? S.apex0 : S.n0;
    const int nlev = hyb ? (apex ? S.apex_lev : S.lev_cut) : S.nlev;
    auto len = [&](int r) {
        if (hyb && forward && r >= cut_n0) return (int)(std::lower_bound(ind.begin() + ptr[r], ind.begin() + ptr[r + 1], cut_n0) - (ind.begin() + ptr[r]));
        return ptr[r + 1] - ptr[r];
    };
    auto pow2ceil = [](int x) { int p = 1; while (p < x) p <<= 1; return p; };
    // shape of the next slice of a level at workgroup width Tw: rows [r, r+cnt), g lanes per row, K entries per lane
    auto shape = [&](int r, int end, int Tw, int &g, int &cnt, int &K) {
        // lanes per row from the longest row among the next Tw candidates (rows of a level are
        // sorted by decreasing row length, so for the forward plan this is row r itself)
        int mx = 0;
        for (int i = r; i < std::min(end, r + Tw); i++) mx = std::max(mx, len(i));
        g = std::max(1, std::min(64, pow2ceil((mx + ELL_KMAX - 1) / ELL_KMAX)));
        cnt = std::min(Tw / g, end - r);
        mx = 0;
        for (int i = r; i < r + cnt; i++) mx = std::max(mx, len(i));
        K = (mx + g - 1) / g;
    };
    // the rows of the apex against the columns below (their order is the apex's lane order: no sorting): a slice is a RUN of consecutive
    // rows that share the lanes-per-row factor of its first rows -- a much longer row (the root of an MPC tree: 935 entries among rows
    // of 18..26) starts its own slice instead of forcing 64 lanes on every row of the block
    auto shape_run = [&](int r, int end, int Tw, int &g, int &cnt, int &K) {
        auto gof = [&](int m) { return std::max(1, std::min(64, pow2ceil((m + ELL_KMAX - 1) / ELL_KMAX))); };
        int mx = len(r);
        g = gof(mx); cnt = 1;
        while (r + cnt < end && r + cnt != pl.split_row && gof(std::max(mx, len(r + cnt))) == g && (cnt + 1) * g <= Tw) { mx = std::max(mx, len(r + cnt)); cnt++; }
        K = (mx + g - 1) / g;
    };
    auto slices_of = [&](int v, int Tw) {
        int r = S.lev_ptr[v], n = 0;
        while (r < S.lev_ptr[v + 1]) { int g, cnt, K; shape(r, S.lev_ptr[v + 1], Tw, g, cnt, K); r += cnt; n++; }
        return n;
    };
    // dense apex: the one row of the block that would need sub-slices at 64 lanes (> 64 ELL_KMAX entries below the block), cut into parts
    // that fill the spare slots of the sweep vector behind slot N (see TriPlan::split_row)
    if (apex && forward) {
        const int spare = scalar_npad(S.N) - (S.N + 1);
        int longest = -1;
        for (int r = cut_n0; r < S.N; r++) if (len(r) > 64 * ELL_KMAX && (longest < 0 || len(r) > len(longest))) longest = r;
        if (longest >= 0) {
            const int parts = std::min({spare, T / 64, (len(longest) + 64 * ELL_KMAX - 1) / (64 * ELL_KMAX)});
            if (parts >= 2) { pl.split_row = longest; pl.split_slot0 = S.N + 1; pl.split_n = parts; }
        }
    }
    auto emit_split = [&](bool first) { // the parts of split_row: ONE slice, 64 lanes per part
        const int r = pl.split_row, n = len(r), P = pl.split_n, part = (n + P - 1) / P, lanes = 64 * P, K = (part + 63) / 64;
        push_subslices(pl.sl, SliceMeta{pl.split_slot0, P, 6, K, pl.slots, first ? 1 : 0, 0, 0});
        pl.idx.resize((size_t)pl.slots + (size_t)K * lanes, S.N);
        for (int j = 0; j < n; j++) {
            const int e = ptr[r] + j, p = j / part, jj = j % part, q = jj % 64, kk = jj / 64;
            const int slot = pl.slots + kk * lanes + p * 64 + q;
            pl.idx[slot] = ind[e];
            pl.pos[S.Rpos[e]] = slot;
        }
        pl.slots += K * lanes;
    };
    auto emit_level = [&](int v, int Tw) { // v = nlev (hybrid, forward): the rows of the top block
        int r = S.lev_ptr[v];
        const int end = (hyb && v == nlev) ? S.N : S.lev_ptr[v + 1];
        bool first = true;
        while (r < end) {
            if (apex && v == nlev && r == pl.split_row) { emit_split(first); first = false; r++; continue; }
            int g, cnt, K;
            if (apex && v == nlev) shape_run(r, end, Tw, g, cnt, K); else shape(r, end, Tw, g, cnt, K);
            int lg = 0;
            while ((1 << lg) < g) lg++;
            const int lanes = cnt * g;
            push_subslices(pl.sl, SliceMeta{r, cnt, lg, K, pl.slots, first ? 1 : 0, 0, 0});
            pl.idx.resize((size_t)pl.slots + (size_t)K * lanes, S.N); // padding gathers the zero slot N
            for (int i = r; i < r + cnt; i++)
                for (int e = ptr[i]; e < ptr[i] + len(i); e++) {
                    const int j = e - ptr[i], q = j % g, kk = j / g;
                    const int slot = pl.slots + kk * lanes + (i - r) * g + q;
                    pl.idx[slot] = ind[e];
                    pl.pos[forward ? S.Rpos[e] : e] = slot; // indexed by the CSC entry of L
                }
            pl.slots += K * lanes;
            r += cnt;
            first = false;
        }
    };
    // empty slices: no loop tail in the kernel's software pipeline (trips of TRI_TRIP slices, remainder in trips of the queue depth).  Every
    // SECTION of a plan (wide / solo / ext: each is walked by its own tri_sweep call) is padded to a multiple of ITS queue depth, counted
    // from the section's first slice -- the two depths need not divide each other
    size_t sec0 = 0;
    auto pad_to = [&](int depth) { while ((pl.sl.size() - sec0) % depth) pl.sl.push_back(SliceMeta{0, 0, 0, 0, pl.slots, 0, 0, 0}); sec0 = pl.sl.size(); };
    auto pad = [&]() { pad_to(TRI_DEPTH); };
    auto pad_solo = [&]() { pad_to(TRI_DEPTH_SOLO); };
    // levels >= vs form the narrow top of the tree (each fits one wavefront in at most two slices)
    const int v_first = forward ? 1 : 0; // forward (L y = b, unit lower L): level-0 rows have no entries: y = b
    int vs = nlev;
    // (with a dense apex the backward sweep has no single-wavefront part: the few narrow levels left below the apex run as workgroup-wide
    // slices, so that the backward sweep is ONE call whose loads start while wavefront 0 is still in the apex -- measured +0.5 ... 2 %)
    // (a handle that runs one workgroup per CU has no single-wavefront parts at all -- api.cpp passes allow_solo = false; with a second workgroup on
    // the CU the idle wavefronts of a single-wavefront part are issue slots for the neighbour, and the narrow levels stay on it)
    const bool solo_here = allow_solo && !(apex && !forward && env_knob("EICOS_APEX_BSOLO", 0, 0, 1) == 0);
    if (solo_here) while (vs > v_first && slices_of(vs - 1, 64) <= 2) vs--;
    if (forward) {
        for (int v = v_first; v < vs; v++) emit_level(v, T);
        // dense apex with no narrow levels below it: its rows against the columns below are simply the last level of the workgroup-wide
        // part (one tri_sweep call: the loads of these slices are prefetched behind the levels before instead of starting cold)
        const bool ext_merged = apex && std::max(vs, v_first) >= nlev;
        if (ext_merged) emit_level(nlev, T);
        pad(); pl.n_wide = (int)pl.sl.size();
        for (int v = std::max(vs, v_first); v < nlev; v++) emit_level(v, 64);
        pad_solo(); pl.n_solo = (int)pl.sl.size() - pl.n_wide;
        if (hyb && !ext_merged) { emit_level(nlev, T); pad(); pl.n_ext = (int)pl.sl.size() - pl.n_wide - pl.n_solo; }
    } else {
        for (int v = nlev - 1; v >= std::max(vs, v_first); v--) emit_level(v, 64);
        pad_solo(); pl.n_solo = (int)pl.sl.size();
        for (int v = vs - 1; v >= v_first; v--) emit_level(v, T);
        pad(); pl.n_wide = (int)pl.sl.size() - pl.n_solo;
    }
    for (SliceMeta &m : pl.sl) if (m.cnt == 0) m.off = pl.slots; // padding slices read the dummy slot
    pl.idx.push_back(S.N); // slot `slots`: the dummy (index N, value 0) read by inactive lanes
    pl.ulen = pl.slots + 1;
    if (apex) { // the entries inside the apex: dense image behind the dummy slot
        const int na = S.N - S.apex0;
        pl.apex_base = ((pl.slots + 1 + 63) / 64) * 64;
        pl.ulen = pl.apex_base + APEX_IMG;
        (void)na;
        for (int j = S.apex0; j < S.N; j++)
            for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) {
                const int i = S.Li[e] - S.apex0, k = j - S.apex0; // (rows of a column lie above it in the order: i > k, both inside the apex)
                pl.pos[e] = pl.apex_base + apex_img_at(i, k);
            }
    }
    return pl;
}

FactorPlan build_factor_plan(const Symbolic &S, int T, const std::vector<int> &posB, int dummyB,
                             const std::vector<int> &posF, int dummyF) {
    FactorPlan pl;
    auto pow2ceil = [](int x) { int p = 1; while (p < x) p <<= 1; return p; };
    auto gof = [&](int64_t m) { return std::max(1, std::min(64, pow2ceil((int)((m + ELL_KMAX - 1) / ELL_KMAX)))); };
    for (int v = 0; v + 1 < (int)S.ftask_ptr.size(); v++) { // (hybrid: the levels below the cut + one level for the top block)
        int r = S.ftask_ptr[v];
        const int end = S.ftask_ptr[v + 1];
        bool first = true;
        while (r < end) {
            auto len = [&](int q) { return S.tp[S.ftask[q] + 1] - S.tp[S.ftask[q]]; };
            const int g = gof(len(r)); // tasks are sorted by decreasing pair count: task r is the longest
            const int cnt = std::min(T / g, end - r);
            const int K = (int)((len(r) + g - 1) / g), lanes = cnt * g;
            int lg = 0;
            while ((1 << lg) < g) lg++;
            push_subslices(pl.sl, SliceMeta{(int)pl.target.size(), cnt, lg, K, pl.slots, first ? 1 : 0, 0, 0});
            pl.pa.resize((size_t)pl.slots + (size_t)K * lanes, dummyB);
            pl.pb.resize(pl.pa.size(), dummyF);
            pl.pbU.resize(pl.pa.size(), dummyB);
            pl.pk.resize(pl.pa.size(), S.N); // (slot N of the pivot mirror holds 0: a padding pair is 0 * (0 * 0))
            for (int i = r; i < r + cnt; i++) {
                const int tgt = S.ftask[i];
                pl.target.push_back(tgt);
                for (int64_t e = S.tp[tgt]; e < S.tp[tgt + 1]; e++) {
                    const int j = (int)(e - S.tp[tgt]), q = j % g, kk = j / g;
                    const int slot = pl.slots + kk * lanes + (i - r) * g + q;
                    pl.pa[slot] = posB[S.pa[e]]; pl.pb[slot] = posF[S.pb[e]]; // term = U[i,k] * L[j,k]
                    pl.pbU[slot] = posB[S.pb[e]]; pl.pk[slot] = S.pk[e];       // ... = U[i,k] * (U[j,k] * (1 / D[k]))
                }
            }
            pl.slots += K * lanes;
            r += cnt;
            first = false;
        }
    }
    pl.pa.push_back(dummyB); pl.pb.push_back(dummyF); pl.pbU.push_back(dummyB); pl.pk.push_back(S.N); // dummy slot `slots`
    // newlev bit 1: last slice of its level (the kernel runs the level's second phase after it)
    for (size_t i = 0; i < pl.sl.size(); i++)
        if (i + 1 == pl.sl.size() || (pl.sl[i + 1].newlev & 1)) pl.sl[i].newlev |= 2;
    return pl;
}

EllPlan build_ell_plan(const std::vector<int> &ptr, int nrows, int T) {
    EllPlan pl;
    auto len = [&](int r) { return ptr[r + 1] - ptr[r]; };
    auto pow2ceil = [](int x) { int p = 1; while (p < x) p <<= 1; return p; };
    int r = 0;
    while (r < nrows) {
        // grow the slice while the lanes-per-row factor of its longest row still lets it fit in T lanes
        int mx = len(r), cnt = 1;
        auto gof = [&](int m) { return std::max(1, std::min(64, pow2ceil((m + ELL_KMAX - 1) / ELL_KMAX))); };
        while (r + cnt < nrows) {
            const int m2 = std::max(mx, len(r + cnt));
            if ((cnt + 1) * gof(m2) > T) break;
            // do not let one very long row force many lanes on a run of short rows
            if (gof(m2) > gof(mx) && cnt >= 32) break;
            mx = m2; cnt++;
        }
        const int g = gof(mx), K = (mx + g - 1) / g, lanes = cnt * g;
        int lg = 0;
        while ((1 << lg) < g) lg++;
        push_subslices(pl.sl, SliceMeta{r, cnt, lg, K, pl.slots, 0, 0, 0});
        pl.src.resize((size_t)pl.slots + (size_t)K * lanes, -1);
        for (int i = r; i < r + cnt; i++)
            for (int e = ptr[i]; e < ptr[i + 1]; e++) {
                const int j = e - ptr[i], q = j % g, kk = j / g;
                pl.src[pl.slots + kk * lanes + (i - r) * g + q] = e;
            }
        pl.slots += K * lanes;
        r += cnt;
    }
    pl.src.push_back(-1); // dummy slot (index `slots`)
    while (pl.sl.size() % ELL_DEPTH) pl.sl.push_back(SliceMeta{0, 0, 0, 0, pl.slots, 0, 0, 0}); // no loop tail
    return pl;
}

} // namespace eicos
