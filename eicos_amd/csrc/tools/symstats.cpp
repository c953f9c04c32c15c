// Dev tool: print symbolic-analysis statistics for an EPB1 fixture.
#include "../symbolic.hpp"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
using namespace eicos;
int main(int argc, char **argv) {
    if (argc < 2) return 1;
    int mode = argc > 2 ? atoi(argv[2]) : 1;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), {});
    const int *h = (const int *)(raw.data() + 4);
    ProblemPattern P; P.n = h[0]; P.m = h[1]; P.p = h[2]; P.l = h[3]; P.nc = h[4];
    int nnzG = h[5], nnzA = h[6];
    const int *ip = h + 8;
    P.q.assign(ip, ip + P.nc); ip += P.nc;
    P.Gjc.assign(ip, ip + P.n + 1); ip += P.n + 1; P.Gir.assign(ip, ip + nnzG); ip += nnzG;
    P.Ajc.assign(ip, ip + P.n + 1); ip += P.n + 1; P.Air.assign(ip, ip + nnzA); ip += nnzA;
    auto t0 = std::chrono::steady_clock::now();
    Symbolic S = analyze(P, mode);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s mode=%d N=%d nnzK=%d nnzL=%d nlev=%d npairs=%lld maxrow=%d maxcol=%d  (%.2fs)\n", argv[1], mode, S.N, S.nnzK, S.nnzL, S.nlev, (long long)S.npairs, S.max_row_len, S.max_col_len, dt);
    if (argc > 3) {
        for (int v = 0; v < S.nlev; v++) {
            long long maxp = 0; 
            for (int t = S.ftask_ptr[v]; t < S.ftask_ptr[v+1]; t++) maxp = std::max<long long>(maxp, S.tp[S.ftask[t]+1]-S.tp[S.ftask[t]]);
            printf("  lev %3d nodes %5d targets %6d maxpairs %lld\n", v, S.lev_ptr[v+1]-S.lev_ptr[v], S.ftask_ptr[v+1]-S.ftask_ptr[v], maxp);
        }
    }
    return 0;
}
