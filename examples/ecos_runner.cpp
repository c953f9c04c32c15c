// Counterpart of the reference's test runner test/ecostester.cpp:54-92 for the problems available as EPB1
// fixtures: every problem goes through the ECOS-style C shim of include/ecos.h -- ECOS_setup -> ECOS_solve
// [-> ECOS_updateData -> ECOS_solve for every further value set, as test/updateData/update_data.h:1662-1683 does]
// -> ECOS_cleanup -- and the exit code of every solve must be one the reference's test header accepts.
//   g++ -std=c++17 -Iinclude examples/ecos_runner.cpp -Leicos_amd -leicos_amd -Wl,-rpath,$PWD/eicos_amd -o ecos_runner
//   ./ecos_runner manifest.txt        (lines: <name> <problem.epb> <accepted exit codes, comma separated>)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

#include "ecos.h"

struct Problem {
    idxint n, m, p, l, nc, nsets;
    std::vector<idxint> q, Gjc, Gir, Ajc, Air;
    struct Set { std::vector<pfloat> Gpr, Apr, c, h, b; };
    std::vector<Set> sets;
};

static bool read_epb(const std::string &path, Problem &P) {
    std::ifstream f(path, std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), {});
    if (raw.size() < 36 || std::memcmp(raw.data(), "EPB1", 4)) return false;
    const int *hd = reinterpret_cast<const int *>(raw.data() + 4);
    P.n = hd[0]; P.m = hd[1]; P.p = hd[2]; P.l = hd[3]; P.nc = hd[4];
    const int nnzG = hd[5], nnzA = hd[6];
    P.nsets = hd[7];
    const int *ip = hd + 8;
    auto take_i = [&](std::vector<idxint> &v, int cnt) { v.assign(ip, ip + cnt); ip += cnt; };
    take_i(P.q, P.nc); take_i(P.Gjc, P.n + 1); take_i(P.Gir, nnzG); take_i(P.Ajc, P.n + 1); take_i(P.Air, nnzA);
    const double *dp = reinterpret_cast<const double *>(ip);
    auto take_d = [&](std::vector<pfloat> &v, int cnt) { v.assign(dp, dp + cnt); dp += cnt; };
    P.sets.resize(P.nsets);
    for (auto &s : P.sets) { take_d(s.Gpr, nnzG); take_d(s.Apr, nnzA); take_d(s.c, P.n); take_d(s.h, P.m); take_d(s.b, P.p); }
    return true;
}

static bool accepted(const std::string &list, idxint code) {
    std::stringstream ss(list);
    std::string tok;
    while (std::getline(ss, tok, ',')) if (std::atoi(tok.c_str()) == code) return true;
    return false;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s manifest.txt\n", argv[0]); return 2; }
    std::ifstream mf(argv[1]);
    std::string name, path, codes;
    int tests_run = 0, failed = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (mf >> name >> path >> codes) {
        Problem P;
        if (!read_epb(path, P)) { std::printf("%s: cannot read %s\n", name.c_str(), path.c_str()); failed++; continue; }
        auto ptr = [](std::vector<pfloat> &v) { return v.empty() ? nullptr : v.data(); };
        Problem::Set &s0 = P.sets[0];
        // NULL groups exactly as the reference's test headers pass them (e.g. feas.h: no A; emptyProblem.h: all NULL)
        const bool haveG = P.m > 0, haveA = P.p > 0;
        pwork *w = ECOS_setup(P.n, P.m, P.p, P.l, P.nc, P.nc ? P.q.data() : nullptr, 0,
                              haveG ? ptr(s0.Gpr) : nullptr, haveG ? P.Gjc.data() : nullptr, haveG ? P.Gir.data() : nullptr,
                              haveA ? ptr(s0.Apr) : nullptr, haveA ? P.Ajc.data() : nullptr, haveA ? P.Air.data() : nullptr,
                              ptr(s0.c), ptr(s0.h), ptr(s0.b));
        bool ok = true;
        std::string seen;
        for (idxint k = 0; k < P.nsets; k++) {
            if (k > 0 && w) ECOS_updateData(w, ptr(P.sets[k].Gpr), ptr(P.sets[k].Apr), ptr(P.sets[k].c), ptr(P.sets[k].h), ptr(P.sets[k].b));
            const idxint exitflag = w ? ECOS_solve(w) : ECOS_FATAL;
            seen += (seen.empty() ? "" : ",") + std::to_string(exitflag);
            if (!accepted(codes, exitflag)) ok = false;
        }
        ECOS_cleanup(w, 0);
        tests_run++;
        std::printf("%-18s exit %-6s accepted {%s}  %s\n", name.c_str(), seen.c_str(), codes.c_str(), ok ? "ok" : "FAILED");
        if (!ok) failed++;
    }
    if (!failed) std::printf("\nALL TESTS PASSED\n");
    std::printf("Tests run: %d\n", tests_run);
    std::printf("Test time: %f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return failed != 0;
}
