// Host-side symbolic analysis (see symbolic.hpp).  Plain C++17, no GPU code.
#include "symbolic.hpp"
#include "envknob.hpp"

#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <stdexcept>

namespace eicos {
namespace {

using ivec = std::vector<int>;

void transpose_pattern(int rows, int cols, const ivec &jc, const ivec &ir, ivec &tptr, ivec &tcol, ivec &tpos) {
    const int nnz = (int)ir.size();
    tptr.assign(rows + 1, 0); tcol.resize(nnz); tpos.resize(nnz);
    for (int k = 0; k < nnz; k++) tptr[ir[k] + 1]++;
    for (int r = 0; r < rows; r++) tptr[r + 1] += tptr[r];
    ivec next(tptr.begin(), tptr.end() - 1);
    for (int j = 0; j < cols; j++)
        for (int k = jc[j]; k < jc[j + 1]; k++) { int d = next[ir[k]]++; tcol[d] = j; tpos[d] = k; }
}

// Minimum-degree ordering on the explicit elimination graph.
// mode 1: each round eliminates a maximal independent set of minimum-degree nodes
// (multiple elimination), which keeps the elimination tree bushy -- tree height is the
// number of dependent steps of every GPU triangular solve, so it matters as much as fill.
// `hold` (optional): hold[v] = w >= 0 keeps node w out of the ordering until node v (and every other node that names w)
// has been eliminated -- used for the two expansion columns of a second-order cone (see analyze_mode).
ivec order_min_degree(int N, const ivec &er, const ivec &ec, int mode, const ivec *hold = nullptr) {
    std::vector<ivec> adj(N);
    for (size_t e = 0; e < er.size(); e++)
        if (er[e] != ec[e]) { adj[er[e]].push_back(ec[e]); adj[ec[e]].push_back(er[e]); }
    for (auto &a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
    std::vector<char> dead(N, 0);
    ivec blocked(N, 0); // number of nodes that must be eliminated before this one becomes eligible
    if (hold) for (int v = 0; v < N; v++) if ((*hold)[v] >= 0) blocked[(*hold)[v]]++;
    ivec order; order.reserve(N);
    ivec mark(N, -1), cand, chosen, merged, nv;
    int round = 0, alive = N;
    while (alive > 0) {
        size_t mind = (size_t)-1;
        for (int v = 0; v < N; v++) if (!dead[v] && !blocked[v]) mind = std::min(mind, adj[v].size());
        chosen.clear();
        if (mode == 0) {
            for (int v = 0; v < N; v++) if (!dead[v] && !blocked[v] && adj[v].size() == mind) { chosen.push_back(v); break; }
        } else {
            // mode = 1 + slack: nodes within `slack` of the minimum degree are eligible too,
            // lowest degree first (plain minimum degree peels chain-like graphs from their
            // two ends, which gives a tree as deep as the chain is long).
            const size_t slack = (size_t)(mode - 1);
            cand.clear();
            for (size_t d = mind; d <= mind + slack; d++)
                for (int v = 0; v < N; v++) if (!dead[v] && !blocked[v] && adj[v].size() == d) cand.push_back(v);
            for (int v : cand) {
                if (mark[v] == round) continue;
                chosen.push_back(v);
                mark[v] = round;
                for (int u : adj[v]) mark[u] = round;
            }
        }
        for (int v : chosen) {
            dead[v] = 1; order.push_back(v); alive--;
            if (hold && (*hold)[v] >= 0) blocked[(*hold)[v]]--;
            nv.swap(adj[v]); adj[v].clear();
            for (int u : nv) {
                merged.clear();
                std::set_union(adj[u].begin(), adj[u].end(), nv.begin(), nv.end(), std::back_inserter(merged));
                ivec &au = adj[u]; au.clear();
                for (int w : merged) if (w != u && w != v) au.push_back(w);
            }
        }
        round++;
    }
    return order;
}

// For a symmetric pattern given by upper entries under permutation iperm: per new column,
// the list of new rows < column.
std::vector<ivec> permuted_upper(int N, const ivec &er, const ivec &ec, const ivec &iperm) {
    std::vector<ivec> up(N);
    for (size_t e = 0; e < er.size(); e++) {
        int a = iperm[er[e]], b = iperm[ec[e]];
        if (a == b) continue;
        up[std::max(a, b)].push_back(std::min(a, b));
    }
    return up;
}

// Elimination tree + row patterns of L (row k = set of columns i<k with L[k,i] != 0).
void etree_rows(int N, const std::vector<ivec> &up, ivec &parent, std::vector<ivec> *rows) {
    parent.assign(N, -1);
    ivec flag(N, -1);
    if (rows) rows->assign(N, ivec());
    for (int k = 0; k < N; k++) {
        flag[k] = k;
        for (int i0 : up[k])
            for (int i = i0; flag[i] != k; i = parent[i]) {
                if (parent[i] < 0) parent[i] = k;
                flag[i] = k;
                if (rows) (*rows)[k].push_back(i);
            }
    }
}

} // namespace

static Symbolic analyze_mode(const ProblemPattern &P, int order_mode, bool tile, bool program, bool hybrid = false, bool cone_order = false);

// order_mode < 0: try several slacks and keep the cheapest under a simple cost model:
// every level costs the GPU one workgroup barrier + a dependent memory round trip, which is
// priced here like LEVEL_COST entries of L.
Symbolic analyze(const ProblemPattern &P, int order_mode, int tile) {
    const std::vector<int> modes = order_mode >= 0 ? std::vector<int>{order_mode} : std::vector<int>{1, 2, 3, 4, 6};
    constexpr double LEVEL_COST = 64.0;
    int best_mode = modes[0];
    // Cone-ordered candidates (the expansion columns of every second-order cone after the cone's rows, see analyze_mode):
    // taken when that order is free under the cost model -- at most one level or 2 % dearer than the best unconstrained
    // order -- which is the case for small patterns; on MPC-SOC it costs 9 % fill and 16 % factor pairs, on the dense-front
    // config 35 % tiles, and the unconstrained order stays.  EICOS_CONE_ORDER (experiment knob): 0 never, 1 always.
    const int cone_knob = env_knob("EICOS_CONE_ORDER", -1, 0, 1);
    if (tile != 1) { // structure only (no factor program): pick the ordering, see how dense L is
        double best_cost = -1, best_cost_c = -1;
        long long best_nnzL = 0;
        int best_mode_c = modes[0];
        const int N = P.n + P.p + P.m + 2 * P.nc;
        for (int mode : modes) {
            Symbolic S = analyze_mode(P, mode, false, false);
            const double cost = S.nnzL + LEVEL_COST * S.nlev;
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_mode = mode; best_nnzL = S.nnzL; }
            if (P.nc > 0 && cone_knob != 0) {
                Symbolic C = analyze_mode(P, mode, false, false, false, true);
                const double cc = C.nnzL + LEVEL_COST * C.nlev;
                if (best_cost_c < 0 || cc < best_cost_c) { best_cost_c = cc; best_mode_c = mode; }
            }
        }
        if (tile < 0) tile = (N > 0 && best_nnzL >= 16LL * N) ? 1 : 2; // ~16+ entries per column of L: dense fronts; else hybrid if it pays
        const bool cone_order = best_cost_c >= 0 && (cone_knob == 1 || best_cost_c <= std::max(best_cost + LEVEL_COST, 1.02 * best_cost));
        if (tile == 0 || tile == 2) return analyze_mode(P, cone_order ? best_mode_c : best_mode, false, true, tile == 2, cone_order);
    }
    // tile path: the cost is the number of 16 x 16 tiles (bytes streamed per solve) plus the block levels (barriers)
    Symbolic best;
    double best_cost = -1;
    for (int mode : modes) {
        Symbolic S = analyze_mode(P, mode, true, false, false, cone_knob == 1);
        long long nt = 0;
        { // count the off-diagonal tiles of L
            std::vector<int> blk(S.N);
            for (int b = 0; b < S.nblk; b++) for (int k = S.blk_ptr[b]; k < S.blk_ptr[b + 1]; k++) blk[k] = b;
            std::vector<int> last(S.nblk, -1);
            for (int j = 0; j < S.N; j++)
                for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) {
                    const int I = blk[S.Li[e]], J = blk[j];
                    if (I != J && last[I] != J) { last[I] = J; nt++; } // rows of a column ascend: one hit per (I, J) run
                }
        }
        const double cost = (double)nt + (double)S.nblk + 4.0 * S.nblev;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = std::move(S); }
    }
    return best;
}

static Symbolic analyze_mode(const ProblemPattern &P, int order_mode, bool tile, bool program, bool hybrid, bool cone_order) {
    Symbolic S;
    S.order_mode = order_mode;
    S.tile = tile ? 1 : 0;
    S.n = P.n; S.p = P.p; S.m = P.m; S.nc = P.nc; S.q = P.q;
    int qsum = 0;
    for (int d : P.q) { if (d < 1) throw std::invalid_argument("cone dimension < 1"); qsum += d; }
    S.l = P.m - qsum;
    if (S.l < 0) throw std::invalid_argument("sum(q) > m");
    S.N = P.n + P.p + P.m + 2 * P.nc;
    S.mt = P.m + 2 * P.nc;
    S.nnzA = P.nnzA(); S.nnzG = P.nnzG();
    S.cone_off.resize(P.nc);
    { int o = S.l; for (int c = 0; c < P.nc; c++) { S.cone_off[c] = o; o += P.q[c]; } }
    S.nV = S.l; for (int d : P.q) S.nV += 3 * d + 1;

    transpose_pattern(P.p, P.n, P.Ajc, P.Air, S.At_ptr, S.At_col, S.At_pos);
    transpose_pattern(P.m, P.n, P.Gjc, P.Gir, S.Gt_ptr, S.Gt_col, S.Gt_pos);

    // ---- KKT pattern, reference layout (src/eicos.cpp:1765-1878) ----
    auto addK = [&](int r, int c, int kind, int src) { S.K_row.push_back(r); S.K_col.push_back(c); S.K_kind.push_back(kind); S.K_src.push_back(src); };
    const int n = S.n, p = S.p;
    for (int j = 0; j < n; j++) addK(j, j, SRC_POSDELTA, 0);
    for (int r = 0; r < p; r++) {
        for (int k = S.At_ptr[r]; k < S.At_ptr[r + 1]; k++) addK(S.At_col[k], n + r, SRC_A, S.At_pos[k]);
        addK(n + r, n + r, SRC_NEGDELTA, 0);
    }
    int grow = 0, col = n + p, vs = 0;
    for (int i = 0; i < S.l; i++, grow++, col++) {
        for (int k = S.Gt_ptr[grow]; k < S.Gt_ptr[grow + 1]; k++) addK(S.Gt_col[k], col, SRC_G, S.Gt_pos[k]);
        addK(col, col, SRC_V, vs++);
    }
    for (int c = 0; c < S.nc; c++) {
        const int d = P.q[c], c0 = col, base = vs;
        for (int i = 0; i < d; i++, grow++, col++) {
            for (int k = S.Gt_ptr[grow]; k < S.Gt_ptr[grow + 1]; k++) addK(S.Gt_col[k], col, SRC_G, S.Gt_pos[k]);
            addK(col, col, SRC_V, base + i);                       // D_i
        }
        for (int i = 1; i < d; i++) addK(c0 + i, col, SRC_V, base + d + i); // v_{i-1} at slot d+1+(i-1)
        addK(col, col, SRC_V, base + d);                           // v diagonal
        col++;
        for (int i = 0; i < d; i++) addK(c0 + i, col, SRC_V, base + 2 * d + 1 + i); // u_i
        addK(col, col, SRC_V, base + 2 * d);                       // u diagonal
        col++;
        vs += 3 * d + 1;
    }
    S.nnzK = (int)S.K_row.size();
    const int N = S.N;

    // ---- ordering, then renumber so that etree levels are contiguous ----
    // Second-order cones: the two expansion columns of a cone (v, then u: src/eicos.cpp:1848-1876) are ordered AFTER the
    // cone's own rows, v before u -- the order the reference's column layout has them in and the one the sparse expansion
    // is designed for: eliminating v first turns the cone's diagonal block -eta^2 I into -eta^2 (I - v1^2 q q'), whose
    // pivots cancel (measured on unboundedMaxSqrt, 300 copies perturbed by 1e-16: v and u first -> 300 x NUMERICS; cone
    // rows first -> 140..157 x the DINF certificate the reference's test asserts, the CPU oracle: 182).
    ivec hold;
    if (S.nc > 0 && cone_order) {
        S.cone_order = 1;
        hold.assign(N, -1);
        int k0 = S.n + S.p + S.l;
        for (int c = 0; c < S.nc; c++) {
            const int d = P.q[c];
            for (int i = 0; i < d; i++) hold[k0 + i] = k0 + d; // every cone row holds back v
            hold[k0 + d] = k0 + d + 1;                          // v holds back u
            k0 += d + 2;
        }
    }
    ivec perm0 = order_min_degree(N, S.K_row, S.K_col, order_mode, hold.empty() ? nullptr : &hold);
    ivec iperm0(N);
    for (int k = 0; k < N; k++) iperm0[perm0[k]] = k;
    std::vector<ivec> rows;
    if (!tile) {
        {
            auto up = permuted_upper(N, S.K_row, S.K_col, iperm0);
            ivec par; etree_rows(N, up, par, nullptr);
            ivec level(N, 0);
            for (int k = 0; k < N; k++) if (par[k] >= 0) level[par[k]] = std::max(level[par[k]], level[k] + 1);
            ivec idx(N); std::iota(idx.begin(), idx.end(), 0);
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return level[a] < level[b]; });
            S.perm.resize(N); S.iperm.resize(N);
            for (int k = 0; k < N; k++) { S.perm[k] = perm0[idx[k]]; S.iperm[S.perm[k]] = k; }
            S.nlev = N ? level[idx[N - 1]] + 1 : 0;
            S.lev_ptr.assign(S.nlev + 1, 0);
            for (int k = 0; k < N; k++) S.lev_ptr[level[idx[k]] + 1]++;
            for (int v = 0; v < S.nlev; v++) S.lev_ptr[v + 1] += S.lev_ptr[v];
        }
        // ---- L pattern under the final ordering ----
        // Inside a level the numbering is free (any topological order of the tree gives the same
        // fill): sort each level by decreasing row length of L so that the sliced-ELL structures of
        // the GPU triangular solves (rows of one slice share a lanes-per-row factor) pad little.
        auto up = permuted_upper(N, S.K_row, S.K_col, S.iperm);
        etree_rows(N, up, S.parent, &rows);
        ivec idx(N); std::iota(idx.begin(), idx.end(), 0);
        for (int v = 0; v < S.nlev; v++)
            std::stable_sort(idx.begin() + S.lev_ptr[v], idx.begin() + S.lev_ptr[v + 1],
                             [&](int a, int b) { return rows[a].size() > rows[b].size(); });
        ivec perm2(N);
        for (int k = 0; k < N; k++) perm2[k] = S.perm[idx[k]];
        S.perm = perm2;
        for (int k = 0; k < N; k++) S.iperm[S.perm[k]] = k;
        auto up2 = permuted_upper(N, S.K_row, S.K_col, S.iperm);
        etree_rows(N, up2, S.parent, &rows);
    } else {
        // ---- tile mode: postorder of the elimination tree (subtrees, hence supernodes, become contiguous), cut into
        // blocks of <= 16 nodes at supernode boundaries, blocks renumbered by their level in the block dependency graph.
        // Every step is a topological reordering of the elimination tree, so the fill of the min-degree order is kept.
        ivec par;
        { auto up = permuted_upper(N, S.K_row, S.K_col, iperm0); etree_rows(N, up, par, nullptr); }
        ivec post; post.reserve(N);
        {
            std::vector<ivec> kids(N);
            ivec roots;
            for (int k = 0; k < N; k++) (par[k] >= 0 ? kids[par[k]] : roots).push_back(k);
            ivec stack, it(N, 0);
            for (int r : roots) {
                stack.push_back(r);
                while (!stack.empty()) {
                    const int v = stack.back();
                    if (it[v] < (int)kids[v].size()) stack.push_back(kids[v][it[v]++]);
                    else { post.push_back(v); stack.pop_back(); }
                }
            }
        }
        ivec perm1(N), iperm1(N);
        for (int k = 0; k < N; k++) { perm1[k] = perm0[post[k]]; iperm1[perm1[k]] = k; }
        ivec par1;
        std::vector<ivec> cols(N); // structure of column j of L (rows > j), ascending
        auto structure = [&]() {
            auto up = permuted_upper(N, S.K_row, S.K_col, iperm1);
            etree_rows(N, up, par1, &rows);
            for (auto &c : cols) c.clear();
            for (int i = 0; i < N; i++) for (int j : rows[i]) cols[j].push_back(i);
        };
        structure();
        // supernodes: node k joins node k-1 when struct(k-1) \ {k} == struct(k) (a chain of the tree: fundamental
        // supernode) or struct(k-1) == struct(k) (siblings with one common front, e.g. the rows of one second-order cone)
        auto joins = [&](int a, int b) {
            const ivec &ca = cols[a], &cb = cols[b];
            if (ca.size() == cb.size()) return ca == cb;
            if (ca.size() == cb.size() + 1 && !ca.empty() && ca[0] == b) return std::equal(cb.begin(), cb.end(), ca.begin() + 1);
            return false;
        };
        {   // The TOP of the tree: small supernodes all of whose ancestors are small supernodes too (on the dense-front config the
            // variables shared by two neighbouring cones: 28 supernodes of 1..3 nodes).  In the postorder each of them follows the
            // subtrees of its children, so they are no neighbours and the amalgamation below cannot merge them -- every one stays a
            // block whose tiles hold one to three rows (22 % of that config's tiles hold a single row).  The set is closed under
            // "parent of", so moving it behind all other nodes, in its own postorder, is again a topological order of the tree (same
            // fill); its supernodes then are neighbours.
            ivec sn_of(N), sn_first, sn_size;
            for (int k = 0; k < N; k++) { if (k == 0 || !joins(k - 1, k)) { sn_first.push_back(k); sn_size.push_back(0); } sn_of[k] = (int)sn_first.size() - 1; sn_size.back()++; }
            const int nsn = (int)sn_first.size();
            std::vector<char> top(nsn, 0);
            int ntop = 0;
            for (int s = nsn - 1; s >= 0; s--) { // (a supernode's parent supernode has a larger number: parents follow children)
                const int last = sn_first[s] + sn_size[s] - 1, pr = par1[last];
                top[s] = sn_size[s] <= 8 && (pr < 0 || top[sn_of[pr]]);
                ntop += top[s];
            }
            if (ntop >= 2) {
                ivec ord; ord.reserve(N);
                for (int k = 0; k < N; k++) if (!top[sn_of[k]]) ord.push_back(k);
                for (int k = 0; k < N; k++) if (top[sn_of[k]]) ord.push_back(k);
                ivec perm2(N);
                for (int k = 0; k < N; k++) perm2[k] = perm1[ord[k]];
                perm1.swap(perm2);
                for (int k = 0; k < N; k++) iperm1[perm1[k]] = k;
                structure();
            }
        }
        // Fundamental supernodes first (no size limit), then RELAXED AMALGAMATION of supernodes that are neighbours in the postorder,
        // then every (merged) supernode is cut into blocks of 16.  Any partition into runs of consecutive nodes is correct for the tile
        // code (tiles.cpp derives tiles, pairs and levels from the scalar pattern; structural zeros inside a tile are just zeros), so the
        // partition is purely a cost choice, and the cost is what the kernels stream: off-diagonal tiles + diagonal tiles + level
        // barriers.  Strict supernodes waste tiles on tiny supernodes -- on the dense-front config the v column of every cone (one node:
        // 5 tiles at 1/16 fill, next to a 62-node supernode with the same row structure) and the 1..3-node supernodes at the top of
        // the tree (variables shared by two cones): 59 of its 316 blocks, ~30 % of its 1120 tiles.  A boundary between two
        // neighbouring supernodes is dropped when that lowers the cost (exact recount), greedily in elimination order.
        ivec sn_start; // first node of every fundamental supernode
        for (int k = 0; k < N; k++) if (k == 0 || !joins(k - 1, k)) sn_start.push_back(k);
        auto cut16 = [&](const ivec &sn) { // supernode starts -> block starts (runs of <= 16 nodes inside every supernode)
            ivec bs;
            for (size_t s = 0; s < sn.size(); s++) {
                const int a = sn[s], b = s + 1 < sn.size() ? sn[s + 1] : N;
                for (int k = a; k < b; k += 16) bs.push_back(k);
            }
            return bs;
        };
        ivec blk_of(N), last_hit, lev_of;
        auto cost_of = [&](const ivec &bs) { // tiles + blocks + 4 x block levels (= the tile path's cost model in analyze())
            const int nbk = (int)bs.size();
            for (int b = 0; b < nbk; b++) for (int k = bs[b]; k < (b + 1 < nbk ? bs[b + 1] : N); k++) blk_of[k] = b;
            last_hit.assign(nbk, -1); lev_of.assign(nbk, 0);
            long long nt = 0;
            int maxlev = 0;
            for (int j = 0; j < N; j++)
                for (int i : cols[j]) { const int I = blk_of[i], J = blk_of[j]; if (I != J && last_hit[I] != J) { last_hit[I] = J; nt++; } }
            for (int i = 0; i < N; i++)
                for (int j : rows[i]) { const int I = blk_of[i], J = blk_of[j]; if (I != J && lev_of[I] <= lev_of[J]) { lev_of[I] = lev_of[J] + 1; maxlev = std::max(maxlev, lev_of[I]); } }
            return (double)nt + (double)nbk + 4.0 * (maxlev + 1);
        };
        {
            double cur = cost_of(cut16(sn_start));
            // a merge can only pay when it saves a block: ceil((a + b) / 16) < ceil(a / 16) + ceil(b / 16); the recounts are bounded
            // (each is O(nnz(L))) so that huge patterns do not spend minutes here
            long long nnz_rows = 0;
            for (int i = 0; i < N; i++) nnz_rows += (long long)rows[i].size();
            long long budget = std::min<long long>(4000, 3000000000LL / std::max<long long>(1, 2 * nnz_rows));
            for (size_t s = 0; s + 1 < sn_start.size() && budget > 0;) {
                const int a = sn_start[s + 1] - sn_start[s], b = (s + 2 < sn_start.size() ? sn_start[s + 2] : N) - sn_start[s + 1];
                if ((a + b + 15) / 16 >= (a + 15) / 16 + (b + 15) / 16) { s++; continue; }
                ivec trial(sn_start);
                trial.erase(trial.begin() + s + 1);
                const double c = cost_of(cut16(trial));
                budget--;
                if (c < cur) { cur = c; sn_start.swap(trial); } // merged: the same position now faces the next supernode
                else s++;
            }
        }
        ivec blk_start = cut16(sn_start); // first node of every block
        const int nb = (int)blk_start.size();
        blk_start.push_back(N);
        ivec blk(N);
        for (int b = 0; b < nb; b++) for (int k = blk_start[b]; k < blk_start[b + 1]; k++) blk[k] = b;
        ivec blev(nb, 0);
        for (int i = 0; i < N; i++)
            for (int j : rows[i]) if (blk[j] != blk[i]) blev[blk[i]] = std::max(blev[blk[i]], blev[blk[j]] + 1); // blocks ascend
        ivec bidx(nb); std::iota(bidx.begin(), bidx.end(), 0);
        std::stable_sort(bidx.begin(), bidx.end(), [&](int a, int b) { return blev[a] < blev[b]; });
        S.nblk = nb; S.nblev = nb ? blev[bidx[nb - 1]] + 1 : 0;
        S.blk_ptr.assign(1, 0); S.blev_ptr.assign(S.nblev + 1, 0);
        S.perm.resize(N); S.iperm.resize(N);
        int pos = 0;
        for (int q = 0; q < nb; q++) {
            const int b = bidx[q];
            for (int k = blk_start[b]; k < blk_start[b + 1]; k++) S.perm[pos++] = perm1[k];
            S.blk_ptr.push_back(pos);
            S.blev_ptr[blev[b] + 1] = q + 1;
        }
        for (int v = 0; v < S.nblev; v++) S.blev_ptr[v + 1] = std::max(S.blev_ptr[v + 1], S.blev_ptr[v]);
        for (int k = 0; k < N; k++) S.iperm[S.perm[k]] = k;
        S.nlev = S.nblev; // node ranges of the block levels
        S.lev_ptr.assign(S.nlev + 1, 0);
        for (int v = 0; v < S.nblev; v++) S.lev_ptr[v + 1] = S.blk_ptr[S.blev_ptr[v + 1]];
        auto up2 = permuted_upper(N, S.K_row, S.K_col, S.iperm);
        etree_rows(N, up2, S.parent, &rows);
    }
    S.Rp.assign(N + 1, 0);
    for (int i = 0; i < N; i++) { std::sort(rows[i].begin(), rows[i].end()); S.Rp[i + 1] = S.Rp[i] + (int)rows[i].size(); }
    S.nnzL = S.Rp[N];
    S.Rj.resize(S.nnzL);
    for (int i = 0; i < N; i++) std::copy(rows[i].begin(), rows[i].end(), S.Rj.begin() + S.Rp[i]);
    S.Lp.assign(N + 1, 0);
    for (int e = 0; e < S.nnzL; e++) S.Lp[S.Rj[e] + 1]++;
    for (int j = 0; j < N; j++) S.Lp[j + 1] += S.Lp[j];
    S.Li.resize(S.nnzL); S.Rpos.resize(S.nnzL); S.Cpos.resize(S.nnzL);
    {
        ivec next(S.Lp.begin(), S.Lp.end() - 1);
        for (int i = 0; i < N; i++)
            for (int e = S.Rp[i]; e < S.Rp[i + 1]; e++) { int d = next[S.Rj[e]]++; S.Li[d] = i; S.Rpos[e] = d; S.Cpos[d] = e; }
    }
    for (int i = 0; i < N; i++) S.max_row_len = std::max(S.max_row_len, S.Rp[i + 1] - S.Rp[i]);
    for (int j = 0; j < N; j++) S.max_col_len = std::max(S.max_col_len, S.Lp[j + 1] - S.Lp[j]);

    auto find_csc = [&](int i, int j) { // position of L[i,j], i>j
        auto b = S.Li.begin() + S.Lp[j], e = S.Li.begin() + S.Lp[j + 1];
        auto it = std::lower_bound(b, e, i);
        if (it == e || *it != i) throw std::logic_error("symbolic: entry missing from L pattern");
        return (int)(it - S.Li.begin());
    };

    // ---- numeric sources ----
    S.Lkind.assign(S.nnzL, SRC_ZERO); S.Lsrc.assign(S.nnzL, 0);
    S.Dkind.assign(N, SRC_ZERO); S.Dsrc.assign(N, 0);
    for (int e = 0; e < S.nnzK; e++) {
        int a = S.iperm[S.K_row[e]], b = S.iperm[S.K_col[e]];
        if (a == b) { S.Dkind[a] = S.K_kind[e]; S.Dsrc[a] = S.K_src[e]; }
        else { int pos = find_csc(std::max(a, b), std::min(a, b)); S.Lkind[pos] = S.K_kind[e]; S.Lsrc[pos] = S.K_src[e]; }
    }

    // ---- hybrid: does the scalar schedule end in a chain worth handing to the tile path? ----
    if (!tile && program && hybrid && N > 0) {
        int cut = S.nlev;
        // (EICOS_HYBRID_WMAX, experiment knob: a wider tail, taken whatever its length -- the top of a tree that is not a chain)
        const int wmax = env_knob("EICOS_HYBRID_WMAX", 4, 1, 64);
        const bool forced = wmax != 4;
        while (cut > 0 && S.lev_ptr[cut] - S.lev_ptr[cut - 1] <= wmax) cut--; // maximal tail of levels with <= 4 nodes
        const int nD = N - S.lev_ptr[cut], nbD = (nD + 15) / 16;
        // worth it: the tail is long (each of its levels costs the sweeps a dependent step, one block costs about two)
        if (nD >= 24 && nD <= 1024 && (forced || (S.nlev - cut >= 12 && 3 * nbD <= S.nlev - cut))) {
            S.tile = 2; S.lev_cut = cut; S.n0 = S.lev_ptr[cut];
            S.nblk = nbD; S.nblev = nbD;
            S.blk_ptr.clear(); S.blev_ptr.clear();
            for (int b = 0; b <= nbD; b++) { S.blk_ptr.push_back(std::min(N, S.n0 + 16 * b)); S.blev_ptr.push_back(b); }
        }
    }
    // ---- dense apex: the maximal tail of levels that holds at most APEX_MAX nodes (level 0 always stays on the level schedule) ----
    if (!tile && program && S.tile == 0 && N >= env_knob("EICOS_APEX_MIN_N", APEX_MIN_N, 64, 1 << 20) && env_knob("EICOS_APEX", 1, 0, 1)) {
        int cut = S.nlev;
        while (cut > 1 && N - S.lev_ptr[cut - 1] <= APEX_MAX) cut--;
        if (S.nlev - cut >= APEX_MIN_LEVELS) { S.apex0 = S.lev_ptr[cut]; S.apex_lev = cut; }
    }
    if (tile || !program) { // tile mode has its own (tile-level) program, tiles.cpp; structure-only calls need none
        int64_t np = 0;
        for (int k = 0; k < N; k++) { int64_t c = S.Lp[k + 1] - S.Lp[k]; np += c * (c + 1) / 2; }
        S.npairs = np; S.flops_factor = 3.0 * (double)np + N;
        return S;
    }
    // ---- factor program ----
    const int64_t NT = (int64_t)N + S.nnzL;
    S.tp.assign(NT + 1, 0);
    auto for_each_pair = [&](auto &&fn) {
        const int kend = S.tile == 2 ? S.n0 : N; // hybrid: the columns of the top block update it through the tile program
        for (int k = 0; k < kend; k++) {
            const int b0 = S.Lp[k], b1 = S.Lp[k + 1];
            for (int eb = b0; eb < b1; eb++) {
                const int rb = S.Li[eb];
                fn((int64_t)rb, eb, eb, k);
                for (int ea = eb + 1; ea < b1; ea++) fn((int64_t)N + find_csc(S.Li[ea], rb), ea, eb, k);
            }
        }
    };
    int64_t np = 0;
    for (int k = 0; k < (S.tile == 2 ? S.n0 : N); k++) { int64_t c = S.Lp[k + 1] - S.Lp[k]; np += c * (c + 1) / 2; }
    S.npairs = np;
    if (np > (int64_t)400 * 1000 * 1000)
        throw std::runtime_error("symbolic: scalar factor program too large for a sparse pattern (> 4e8 multiply-subtract pairs)");
    for_each_pair([&](int64_t t, int, int, int) { S.tp[t + 1]++; });
    for (int64_t t = 0; t < NT; t++) S.tp[t + 1] += S.tp[t];
    S.pa.resize(np); S.pb.resize(np); S.pk.resize(np);
    {
        std::vector<int64_t> next(S.tp.begin(), S.tp.end() - 1);
        for_each_pair([&](int64_t t, int ea, int eb, int k) { int64_t d = next[t]++; S.pa[d] = ea; S.pb[d] = eb; S.pk[d] = k; });
    }
    S.flops_factor = 3.0 * (double)np + N;

    // ---- per-level task lists, longest first ----
    // (hybrid: the levels below the cut, then ONE level with every target of the top block)
    const int nlev_f = S.tile == 2 ? S.lev_cut + 1 : S.nlev;
    S.ftask_ptr.assign(nlev_f + 1, 0);
    S.ftask.reserve(NT);
    for (int v = 0; v < nlev_f; v++) {
        const int j0 = S.lev_ptr[v], j1 = (S.tile == 2 && v == S.lev_cut) ? N : S.lev_ptr[v + 1];
        const size_t start = S.ftask.size();
        for (int j = j0; j < j1; j++) S.ftask.push_back(j);
        for (int e = S.Lp[j0]; e < S.Lp[j1]; e++) S.ftask.push_back(N + e);
        std::stable_sort(S.ftask.begin() + start, S.ftask.end(), [&](int a, int b) {
            return (S.tp[a + 1] - S.tp[a]) > (S.tp[b + 1] - S.tp[b]);
        });
        S.ftask_ptr[v + 1] = (int)S.ftask.size();
    }
    return S;
}

} // namespace eicos
