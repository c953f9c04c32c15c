// Dev: memory-locality metrics of the factor program and of the sweep plans for a fixture -- how many distinct 64-byte
// lines one wavefront's gather instruction touches (the vector L1 looks up about one line per clock, so scattered 8-byte
// gathers are bound by that rate, not by bytes).   g++ -O2 -std=c++17 tools/dev/plan_lines.cpp eicos_amd/csrc/{symbolic,plans}.cpp
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <set>
#include <string>
#include <vector>
#include "../../eicos_amd/csrc/plans.hpp"
#include "../../eicos_amd/csrc/symbolic.hpp"
using namespace eicos;
static bool read_epb(const std::string &path, ProblemPattern &P) {
    std::ifstream f(path, std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), {});
    if (raw.size() < 36 || std::memcmp(raw.data(), "EPB1", 4)) return false;
    const int *hd = reinterpret_cast<const int *>(raw.data() + 4);
    P.n = hd[0]; P.m = hd[1]; P.p = hd[2]; P.l = hd[3]; P.nc = hd[4];
    const int nnzG = hd[5], nnzA = hd[6];
    const int *ip = hd + 8;
    auto take = [&](std::vector<int> &v, int cnt) { v.assign(ip, ip + cnt); ip += cnt; };
    take(P.q, P.nc); take(P.Gjc, P.n + 1); take(P.Gir, nnzG); take(P.Ajc, P.n + 1); take(P.Air, nnzA);
    return true;
}
int main(int argc, char **argv) {
    ProblemPattern P;
    if (argc < 2 || !read_epb(argv[1], P)) return 2;
    const int T = argc > 2 ? atoi(argv[2]) : 256;
    Symbolic S = analyze(P, -1, 0);
    TriPlan pf = build_tri_plan(S, T, true), pb = build_tri_plan(S, T, false);
    FactorPlan px = build_factor_plan(S, T, pb.pos, pb.slots, pf.pos, pf.slots);
    auto lines_of = [&](const std::vector<SliceMeta> &sl, auto &&slot_index, int nsets) {
        long instr = 0, lines = 0;
        for (const SliceMeta &m : sl) {
            const int lanes = m.cnt << m.lg;
            for (int w0 = 0; w0 < lanes; w0 += 64)
                for (int kk = 0; kk < m.K; kk++)
                    for (int set = 0; set < nsets; set++) {
                        std::set<int> ln;
                        for (int t = w0; t < std::min(lanes, w0 + 64); t++) ln.insert(slot_index(set, m.off + kk * lanes + t) >> 3);
                        instr++; lines += (long)ln.size();
                    }
        }
        printf("   %ld wave-gathers, %ld lines (%.1f per gather)\n", instr, lines, instr ? (double)lines / instr : 0.);
        return lines;
    };
    printf("N %d nnzL %d levels %d pairs %lld targets %zu\n", S.N, S.nnzL, S.nlev, (long long)S.npairs, px.target.size());
    printf("factor phase A (pa: U slots, pb: L slots):\n");
    lines_of(px.sl, [&](int set, int slot) { return set ? px.pb[slot] : px.pa[slot]; }, 2);
    { // phase B: per target gather U[dst], invD[col] and scatter UF[dstF]
        std::vector<int> col_of(S.nnzL);
        for (int j = 0; j < S.N; j++) for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) col_of[e] = j;
        long instr = 0, lines = 0;
        for (size_t t0 = 0; t0 < px.target.size(); t0 += 64) {
            std::set<int> a, b, c;
            for (size_t t = t0; t < std::min(px.target.size(), t0 + 64); t++) {
                const int tg = px.target[t];
                if (tg < S.N) continue;
                a.insert(pb.pos[tg - S.N] >> 3); b.insert(col_of[tg - S.N] >> 3); c.insert(pf.pos[tg - S.N] >> 3);
            }
            instr += 3; lines += (long)(a.size() + b.size() + c.size());
        }
        printf("factor phase B: %ld wave-accesses, %ld lines (%.1f per access)\n", instr, lines, (double)lines / instr);
    }
    printf("forward sweep LDS gathers (distinct 8-byte words per wave gather = all, bank conflicts not modelled); index spread:\n");
    lines_of(pf.sl, [&](int, int slot) { return pf.idx[slot]; }, 1);
    printf("backward sweep:\n");
    lines_of(pb.sl, [&](int, int slot) { return pb.idx[slot]; }, 1);
    return 0;
}
