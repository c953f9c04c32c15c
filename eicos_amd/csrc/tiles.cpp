// Host-side construction of the tile plan (see tiles.hpp).  Plain C++17, no GPU code.
#include "tiles.hpp"

#include <cstring>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>

namespace eicos {

TilePlan build_tile_plan(const Symbolic &S) {
    if (!S.tile) throw std::logic_error("tile plan requested for a scalar-mode analysis");
    TilePlan T;
    // n0 = first node on tiles: 0 in pure tile mode; hybrid (Symbolic::tile == 2): the top block starts there, the nodes
    // below keep their scalar slots (slot = elimination position) and only entries with BOTH indices >= n0 live in tiles
    const int N = S.N, nb = S.nblk, n0 = S.tile == 2 ? S.n0 : 0;
    T.nb = nb; T.n0 = n0; T.N16 = n0 + 16 * nb; T.nblev = S.nblev; T.blev_ptr = S.blev_ptr;
    std::vector<int> blk(N, -1), off(N, 0);
    T.slot.resize(N);
    for (int k = 0; k < n0; k++) T.slot[k] = k;
    for (int b = 0; b < nb; b++) {
        if (S.blk_ptr[b + 1] - S.blk_ptr[b] > 16 || S.blk_ptr[b + 1] <= S.blk_ptr[b]) throw std::logic_error("tile plan: bad block size");
        for (int k = S.blk_ptr[b]; k < S.blk_ptr[b + 1]; k++) { blk[k] = b; off[k] = k - S.blk_ptr[b]; T.slot[k] = n0 + 16 * b + off[k]; }
    }
    // ---- tiles from the scalar pattern of L (CSC: column j, rows ascending) ----
    std::vector<std::vector<int>> colrows(nb);
    for (int j = n0; j < N; j++)
        for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) if (blk[S.Li[e]] != blk[j]) colrows[blk[j]].push_back(blk[S.Li[e]]);
    T.tc_ptr.assign(nb + 1, 0);
    for (int J = 0; J < nb; J++) {
        auto &v = colrows[J];
        std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
        for (int I : v) { if (I <= J) throw std::logic_error("tile plan: tile above the diagonal"); T.t_row.push_back(I); T.t_col.push_back(J); }
        T.tc_ptr[J + 1] = (int)T.t_row.size();
    }
    T.nt = (int)T.t_row.size();
    auto tile_of = [&](int I, int J) { // id of tile (I, J), I > J; -1 if absent
        auto b = T.t_row.begin() + T.tc_ptr[J], e = T.t_row.begin() + T.tc_ptr[J + 1];
        auto it = std::lower_bound(b, e, I);
        return (it == e || *it != I) ? -1 : (int)(it - T.t_row.begin());
    };
    T.tr_ptr.assign(nb + 1, 0);
    for (int t = 0; t < T.nt; t++) T.tr_ptr[T.t_row[t] + 1]++;
    for (int I = 0; I < nb; I++) T.tr_ptr[I + 1] += T.tr_ptr[I];
    T.tr_tile.resize(T.nt);
    { std::vector<int> next(T.tr_ptr.begin(), T.tr_ptr.end() - 1); for (int t = 0; t < T.nt; t++) T.tr_tile[next[T.t_row[t]]++] = t; } // CSC order => ascending column

    // ---- block levels must be consistent with the tiles: every tile (I, J) has level(I) > level(J) ----
    std::vector<int> lev(nb, 0);
    for (int v = 0; v < T.nblev; v++) for (int b = T.blev_ptr[v]; b < T.blev_ptr[v + 1]; b++) lev[b] = v;
    for (int t = 0; t < T.nt; t++) if (lev[T.t_row[t]] <= lev[T.t_col[t]]) throw std::logic_error("tile plan: level order violated");

    // ---- factor program: pairs per target ----
    // source block column K with tile rows R = [I1 < I2 < ...]: target (R[b], R[a]) for a <= b gets the pair
    // (tile(R[b], K), tile(R[a], K)); a target tile that is structurally absent receives exact zeros only (every term has a
    // structurally zero factor) and is skipped.
    const int ntgt = nb + T.nt;
    std::vector<int> cnt(ntgt + 1, 0);
    auto for_each_pair = [&](auto &&fn) {
        for (int K = 0; K < nb; K++) {
            const int b0 = T.tc_ptr[K], b1 = T.tc_ptr[K + 1];
            for (int a = b0; a < b1; a++) {
                fn(T.t_row[a], a, a, K); // diagonal target of block row(a)
                for (int b = a + 1; b < b1; b++) { const int t = tile_of(T.t_row[b], T.t_row[a]); if (t >= 0) fn(nb + t, b, a, K); }
            }
        }
    };
    for_each_pair([&](int tg, int, int, int) { cnt[tg + 1]++; });
    for (int q = 0; q < ntgt; q++) cnt[q + 1] += cnt[q];
    std::vector<int> pa0(cnt[ntgt]), pb0(cnt[ntgt]), pk0(cnt[ntgt]);
    { std::vector<int> next(cnt.begin(), cnt.end() - 1); for_each_pair([&](int tg, int a, int b, int K) { const int d = next[tg]++; pa0[d] = a; pb0[d] = b; pk0[d] = K; }); }
    T.npairs = cnt[ntgt];
    // execution order: per level, the diagonal and off-diagonal targets of its block columns, longest first
    T.tgt_lev_ptr.assign(1, 0); T.fin_lev_ptr.assign(1, 0); T.tp_ptr.assign(1, 0);
    for (int v = 0; v < T.nblev; v++) {
        const size_t start = T.tgt.size();
        for (int J = T.blev_ptr[v]; J < T.blev_ptr[v + 1]; J++) {
            T.tgt.push_back(J);
            for (int t = T.tc_ptr[J]; t < T.tc_ptr[J + 1]; t++) { T.tgt.push_back(nb + t); T.fin.push_back(t); }
        }
        std::stable_sort(T.tgt.begin() + start, T.tgt.end(), [&](int a, int b) { return cnt[a + 1] - cnt[a] > cnt[b + 1] - cnt[b]; });
        T.tgt_lev_ptr.push_back((int)T.tgt.size());
        T.fin_lev_ptr.push_back((int)T.fin.size());
    }
    for (int tg : T.tgt) {
        for (int d = cnt[tg]; d < cnt[tg + 1]; d++) { T.pa.push_back(pa0[d]); T.pb.push_back(pb0[d]); T.pk.push_back(pk0[d]); }
        T.tp_ptr.push_back((int)T.pa.size());
    }

    // ---- where every scalar entry lives ----
    T.ident.assign(nb, 1);
    T.Le_img.assign(S.nnzL, -1); T.Le_tile.assign(S.nnzL, 0); T.Le_rc.assign(S.nnzL, 0); T.D_img.assign(N, -1);
    for (int j = n0; j < N; j++) {
        T.D_img[j] = blk[j] * 256 + tile_res(off[j], off[j]);
        for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) {
            const int i = S.Li[e], I = blk[i], J = blk[j];
            if (I == J) { T.Le_tile[e] = -1 - J; T.Le_rc[e] = off[i] * 16 + off[j]; T.Le_img[e] = J * 256 + tile_res(off[i], off[j]); T.ident[J] = 0; }
            else {
                const int t = tile_of(I, J);
                if (t < 0) throw std::logic_error("tile plan: entry without a tile");
                T.Le_tile[e] = t; T.Le_rc[e] = off[i] * 16 + off[j]; T.Le_img[e] = (nb + t) * 256 + tile_res(off[i], off[j]);
            }
        }
    }
    for (int b = 0; b < nb; b++)
        for (int o = S.blk_ptr[b + 1] - S.blk_ptr[b]; o < 16; o++) T.pad_img.push_back(b * 256 + tile_res(o, o));
    return T;
}

TileFactorOps build_tile_factor_ops(const TilePlan &T, int NW, int pf, const std::vector<char> *img_zero) {
    TileFactorOps F;
    F.ptr.assign(1, 0);
    for (int v = 0; v < T.nblev; v++) {
        const int q0 = T.tgt_lev_ptr[v], q1 = T.tgt_lev_ptr[v + 1];
        // weight of a target: its operations, plus the dense 16 x 16 LDL' a diagonal target ends with (about eight operations' time)
        auto weight = [&](int q) { return 1 + (T.tp_ptr[q + 1] - T.tp_ptr[q]) + (T.tgt[q] < T.nb ? 8 : 0); };
        std::vector<int> order(q1 - q0);
        std::iota(order.begin(), order.end(), q0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weight(a) > weight(b); });
        std::vector<std::vector<int>> mine(NW);
        std::vector<long> load(NW, 0);
        for (int q : order) { const int w = (int)(std::min_element(load.begin(), load.end()) - load.begin()); mine[w].push_back(q); load[w] += weight(q); }
        for (int w = 0; w < NW; w++) {
            for (int q : mine[w]) {
                const int tg = T.tgt[q], p0 = T.tp_ptr[q], p1 = T.tp_ptr[q + 1];
                const int zero = (img_zero && (*img_zero)[tg]) ? FOP_ZERO : 0; // (tg indexes the K image: diagonal tiles first, then the off-diagonal ones)
                F.ops.insert(F.ops.end(), {zero ? 0 : tg, 0, 0, FOP_INIT | zero | (p0 == p1 ? FOP_END : 0) | (tg << FOP_SHIFT)});
                for (int e = p0; e < p1; e++) F.ops.insert(F.ops.end(), {T.pa[e], T.pb[e], T.pk[e], (e + 1 == p1 ? FOP_END : 0) | (tg << FOP_SHIFT)});
            }
            while (((int)F.ops.size() / 4 - F.ptr.back()) % pf) F.ops.insert(F.ops.end(), {0, 0, 0, FOP_PAD}); // (unconditional loads: see build_tile_sweeps)
            F.ptr.push_back((int)F.ops.size() / 4);
        }
    }
    return F;
}

TileSweeps build_tile_sweeps(const TilePlan &T, int NW, int pf, int serial_max) {
    TileSweeps W;
    W.NW = NW;
    // A SMALL block system (the dense top of a hybrid pattern: lp_bandm's is a chain of seven blocks, one per level) is swept by ONE wavefront
    // as one flat list in dependency order: a wavefront's LDS accesses execute in order, so no barrier separates its levels and its tile queue
    // runs across them -- level by level the same system costs a barrier and a cold start of the queue per level while seven wavefronts wait.
    // Same operations on the same operands in the same order per block: bit-identical.  (Needs the sweep vector in LDS: api.cpp.)
    W.serial = (T.nt + T.nb) <= serial_max ? 1 : 0;
    const bool allow_split = !(getenv("EICOS_EXPERIMENT") && getenv("EICOS_TILE_SPLIT") && !strcmp(getenv("EICOS_EXPERIMENT"), "1") && !strcmp(getenv("EICOS_TILE_SPLIT"), "0"));
    auto build = [&](bool fwd, std::vector<int> &ops, std::vector<int> &ptr, std::vector<int> &endr, std::vector<int> &split) {
        ptr.assign(1, 0); endr.clear();
        split.assign(T.nblev, 0);
        auto close_range = [&]() { endr.push_back((int)ops.size() / 4); while (((int)ops.size() / 4 - ptr.back()) % pf) ops.insert(ops.end(), {0, T.nb, 0, 0}); ptr.push_back((int)ops.size() / 4); };
        if (W.serial) {
            for (int step = 0; step < T.nblev; step++) {
                const int v = fwd ? step : T.nblev - 1 - step;
                for (int B = T.blev_ptr[v]; B < T.blev_ptr[v + 1]; B++) {
                    if (fwd) for (int e = T.tr_ptr[B]; e < T.tr_ptr[B + 1]; e++) { const int t = T.tr_tile[e]; ops.insert(ops.end(), {t, T.t_col[t], B, 0}); }
                    else for (int t = T.tc_ptr[B]; t < T.tc_ptr[B + 1]; t++) ops.insert(ops.end(), {t, T.t_row[t], B, 0});
                    ops.insert(ops.end(), {T.ident[B] ? 0 : B, B, 0, TOP_DIAG | (T.ident[B] ? TOP_IDENT : 0)});
                }
            }
            close_range(); // (level 0, phase 0, wavefront 0) holds everything; every other range is empty
            for (int r = 1; r < T.nblev * 2 * NW; r++) close_range();
            return;
        }
        for (int step = 0; step < T.nblev; step++) {
            const int v = fwd ? step : T.nblev - 1 - step;
            const int b0 = T.blev_ptr[v], b1 = T.blev_ptr[v + 1];
            auto ntiles = [&](int B) { return fwd ? T.tr_ptr[B + 1] - T.tr_ptr[B] : T.tc_ptr[B + 1] - T.tc_ptr[B]; };
            // work items: whole blocks, or parts of a split block (tile range [e0, e1) of the block's list, partial-sum slot)
            struct Item { int B, e0, e1, slot; };
            std::vector<Item> items;
            struct Closing { int B, slot0, np; };
            std::vector<Closing> closing; // diagonal operations of the split blocks (phase 1)
            long total = 0;
            for (int B = b0; B < b1; B++) total += ntiles(B) + 2;
            const long share = std::max<long>(2L * pf, (total + NW - 1) / NW); // a wavefront's share of the level (never below two trips of the queue)
            int slots = 0;
            for (int B = b0; B < b1; B++) {
                const int n = ntiles(B);
                int parts = 1;
                if (allow_split && NW > 1 && n > share + share / 4) parts = (int)std::min<long>(NW, (n + share - 1) / share);
                if (slots + parts > TILE_PARTS) parts = 1;
                if (parts == 1) { items.push_back({B, 0, n, -1}); continue; }
                closing.push_back({B, slots, parts});
                for (int q = 0; q < parts; q++) items.push_back({B, (int)((long)n * q / parts), (int)((long)n * (q + 1) / parts), slots++});
            }
            auto weight = [&](const Item &it) { return (long)(it.e1 - it.e0) + (it.slot < 0 ? 2 : 1); };
            std::vector<int> order(items.size());
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weight(items[a]) > weight(items[b]); });
            std::vector<std::vector<int>> mine(NW);
            std::vector<long> load(NW, 0);
            for (int i : order) { // longest first onto the least loaded wavefront
                const int w = (int)(std::min_element(load.begin(), load.end()) - load.begin());
                mine[w].push_back(i); load[w] += weight(items[i]);
            }
            // pad to a multiple of `pf` operations with products against the all-zero vector block nb: the kernel's software
            // pipeline then has no conditional around its loads (a conditional load defeats the s_waitcnt counting)
            // (the padding is there for the refills of the kernel's queue only: it stops at `endr`, the end of the real operations)
            auto close_segment = [&]() { endr.push_back((int)ops.size() / 4); while (((int)ops.size() / 4 - ptr.back()) % pf) ops.insert(ops.end(), {0, T.nb, 0, 0}); ptr.push_back((int)ops.size() / 4); };
            for (int w = 0; w < NW; w++) { // phase 0
                for (int i : mine[w]) {
                    const Item &it = items[i];
                    const int B = it.B;
                    if (fwd) for (int e = T.tr_ptr[B] + it.e0; e < T.tr_ptr[B] + it.e1; e++) { const int t = T.tr_tile[e]; ops.insert(ops.end(), {t, T.t_col[t], B, 0}); }
                    else for (int t = T.tc_ptr[B] + it.e0; t < T.tc_ptr[B] + it.e1; t++) ops.insert(ops.end(), {t, T.t_row[t], B, 0});
                    if (it.slot >= 0) ops.insert(ops.end(), {0, T.nb, it.slot, TOP_PART}); // partial sum -> LDS slot (its unconditional load: tile 0)
                    // identity diagonal tile: nothing to multiply; its (unconditional) load is pointed at diagonal tile 0, which stays cached
                    else ops.insert(ops.end(), {T.ident[B] ? 0 : B, B, 0, TOP_DIAG | (T.ident[B] ? TOP_IDENT : 0)});
                }
                close_segment();
            }
            for (int w = 0; w < NW; w++) { // phase 1: the split blocks are closed, dealt round-robin
                for (size_t c = (size_t)w; c < closing.size(); c += (size_t)NW) {
                    const Closing &cl = closing[c];
                    ops.insert(ops.end(), {T.ident[cl.B] ? 0 : cl.B, cl.B, cl.slot0, TOP_DIAG | (T.ident[cl.B] ? TOP_IDENT : 0) | (cl.np << TOP_NPART_SHIFT)});
                }
                close_segment();
            }
            split[step] = closing.empty() ? 0 : 1;
        }
    };
    build(true, W.fops, W.fptr, W.fend, W.fsplit);
    build(false, W.bops, W.bptr, W.bend, W.bsplit);
    // every tile exactly once, every block closed exactly once, partial slots written before they are read (by construction; checked)
    for (int pass = 0; pass < 2; pass++) {
        const std::vector<int> &ops = pass ? W.bops : W.fops, &ptr = pass ? W.bptr : W.fptr;
        std::vector<int> seen_tile(T.nt, 0), closed(T.nb, 0);
        for (int st = 0; st < T.nblev; st++) {
            std::vector<int> written(TILE_PARTS, 0);
            for (int ph = 0; ph < 2; ph++)
                for (int w = 0; w < NW; w++)
                    for (int o = ptr[(st * 2 + ph) * NW + w]; o < ptr[(st * 2 + ph) * NW + w + 1]; o++) {
                        const int x = ops[4 * o], y = ops[4 * o + 1], z = ops[4 * o + 2], fl = ops[4 * o + 3];
                        if (fl & TOP_PART) { if (ph != 0 || z < 0 || z >= TILE_PARTS || written[z]++) throw std::logic_error("tile sweeps: bad partial slot"); }
                        else if (fl & TOP_DIAG) {
                            closed[y]++;
                            const int np = fl >> TOP_NPART_SHIFT;
                            if ((np > 0) != (ph == 1)) throw std::logic_error("tile sweeps: split block closed in the wrong phase");
                            for (int q = 0; q < np; q++) if (!written[z + q]) throw std::logic_error("tile sweeps: partial sum read before it is written");
                        } else if (y != T.nb) seen_tile[x]++;
                    }
        }
        for (int t = 0; t < T.nt; t++) if (seen_tile[t] != 1) throw std::logic_error("tile sweeps: a tile is not swept exactly once");
        for (int b = 0; b < T.nb; b++) if (closed[b] != 1) throw std::logic_error("tile sweeps: a block is not closed exactly once");
    }
    if (getenv("EICOS_PLAN_STATS")) { // developer aid: per level, blocks and the op count of the busiest / the average wavefront (phase 0 + phase 1)
        for (int pass = 0; pass < 2; pass++) {
            const std::vector<int> &ptr = pass ? W.bptr : W.fptr;
            fprintf(stderr, "[tile sweeps NW=%d] %s:", NW, pass ? "backward" : "forward");
            long crit = 0, totall = 0;
            for (int st = 0; st < T.nblev; st++) {
                const int v = pass ? T.nblev - 1 - st : st;
                fprintf(stderr, " L%d[%d blk:", v, T.blev_ptr[v + 1] - T.blev_ptr[v]);
                for (int ph = 0; ph < 2; ph++) {
                    int mx = 0, tot = 0;
                    for (int w = 0; w < NW; w++) { const int c = ptr[(st * 2 + ph) * NW + w + 1] - ptr[(st * 2 + ph) * NW + w]; mx = std::max(mx, c); tot += c; }
                    if (ph == 0 || tot) fprintf(stderr, " max %d / avg %.1f%s", mx, (double)tot / NW, ph ? " (closing)" : "");
                    crit += mx; totall += tot;
                }
                fprintf(stderr, "]");
            }
            fprintf(stderr, "  critical path %ld ops, %.1f per wavefront\n", crit, (double)totall / NW);
        }
    }
    return W;
}

} // namespace eicos
