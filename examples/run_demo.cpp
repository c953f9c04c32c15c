// Counterpart of the reference's demo driver src/run.cpp:7-52: setup -> solve -> updateData(same data)
// -> solve, with wall-clock timings, through the EiCOS::Solver surface of include/eicos.hpp.
// Reads a problem in the EPB1 container (eicos_amd/problem_io.py) instead of the missing data_MPC01.hpp.
//   g++ -std=c++17 -Iinclude examples/run_demo.cpp -Leicos_amd -leicos_amd -Wl,-rpath,$PWD/eicos_amd -o run_demo
//   ./run_demo tests/golden/MPC02.epb
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "eicos.hpp"

static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s problem.epb\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), {});
    if (raw.size() < 36 || std::memcmp(raw.data(), "EPB1", 4)) { std::fprintf(stderr, "not an EPB1 file\n"); return 2; }
    const int *hd = reinterpret_cast<const int *>(raw.data() + 4);
    int n = hd[0], m = hd[1], p = hd[2], l = hd[3], nc = hd[4], nnzG = hd[5], nnzA = hd[6];
    const int *ip = hd + 8;
    std::vector<int> q(ip, ip + nc); ip += nc;
    std::vector<int> Gjc(ip, ip + n + 1); ip += n + 1;
    std::vector<int> Gir(ip, ip + nnzG); ip += nnzG;
    std::vector<int> Ajc(ip, ip + n + 1); ip += n + 1;
    std::vector<int> Air(ip, ip + nnzA); ip += nnzA;
    const double *dp = reinterpret_cast<const double *>(ip);
    std::vector<double> Gpr(dp, dp + nnzG); dp += nnzG;
    std::vector<double> Apr(dp, dp + nnzA); dp += nnzA;
    std::vector<double> c(dp, dp + n); dp += n;
    std::vector<double> h(dp, dp + m); dp += m;
    std::vector<double> b(dp, dp + p);

    auto t0 = std::chrono::steady_clock::now();
    EiCOS::Solver solver(n, m, p, l, nc, q.data(), m ? Gpr.data() : nullptr, Gjc.data(), Gir.data(),
                         p ? Apr.data() : nullptr, Ajc.data(), Air.data(), c.data(), h.data(), b.data());
    std::printf("Time for setup:    %.3f ms\n", ms_since(t0));
    t0 = std::chrono::steady_clock::now();
    EiCOS::exitcode code = solver.solve();
    std::printf("Time for solve:    %.3f ms  (exit %d, %zu iterations, pcost %.9g)\n", ms_since(t0), int(code),
                solver.getInfo().iter, solver.getInfo().pcost);
    t0 = std::chrono::steady_clock::now();
    solver.updateData(m ? Gpr.data() : nullptr, p ? Apr.data() : nullptr, c.data(), h.data(), b.data());
    std::printf("Time for update:   %.3f ms\n", ms_since(t0));
    t0 = std::chrono::steady_clock::now();
    code = solver.solve();
    std::printf("Time for solve:    %.3f ms  (exit %d)\n", ms_since(t0), int(code));
    return (code == EiCOS::exitcode::optimal || code == EiCOS::exitcode::close_to_optimal) ? 0 : 1;
}
