// Multi-GPU from host C++ (no torch, no RCCL): ONE EiCOS::BatchSolver over a list of devices -- the batch is cut into contiguous
// shards, one per list entry, each with its own handle and HIP stream (include/eicos_amd.h: eicos_multi_*).  The problem file's
// data is replicated `batch` times with a small per-instance change of h, solved on the device list and again on the first device
// alone; the two runs must agree bit for bit (instances are independent, every shard runs the same kernels).
//   g++ -std=c++17 -Iinclude examples/multi_gpu_demo.cpp -Leicos_amd -leicos_amd -Wl,-rpath,$PWD/eicos_amd -o multi_gpu_demo
//   ./multi_gpu_demo tests/golden/MPC02.epb 64 0,1,2,3     (a device may be listed twice: 0,0)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <vector>

#include "eicos.hpp"

int main(int argc, char **argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s problem.epb batch dev[,dev...]\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), {});
    if (raw.size() < 36 || std::memcmp(raw.data(), "EPB1", 4)) { std::fprintf(stderr, "not an EPB1 file\n"); return 2; }
    const int B = std::atoi(argv[2]);
    std::vector<int> devs;
    { std::stringstream ss(argv[3]); std::string tok; while (std::getline(ss, tok, ',')) devs.push_back(std::atoi(tok.c_str())); }
    const int *hd = reinterpret_cast<const int *>(raw.data() + 4);
    const int n = hd[0], m = hd[1], p = hd[2], nc = hd[4], nnzG = hd[5], nnzA = hd[6];
    const int *ip = hd + 8;
    std::vector<int> q(ip, ip + nc); ip += nc;
    std::vector<int> Gjc(ip, ip + n + 1); ip += n + 1;
    std::vector<int> Gir(ip, ip + nnzG); ip += nnzG;
    std::vector<int> Ajc(ip, ip + n + 1); ip += n + 1;
    std::vector<int> Air(ip, ip + nnzA); ip += nnzA;
    const double *dp = reinterpret_cast<const double *>(ip);
    const double *Gpr = dp, *Apr = Gpr + nnzG, *c = Apr + nnzA, *h = c + n, *b = h + m;
    // [batch][...] arrays in global instance order; instance i relaxes every inequality by 1e-3 i (1 + |h|)
    std::vector<double> G((size_t)B * nnzG), A((size_t)B * nnzA), C((size_t)B * n), H((size_t)B * m), Bv((size_t)B * p);
    for (int i = 0; i < B; i++) {
        std::copy(Gpr, Gpr + nnzG, G.begin() + (size_t)i * nnzG); std::copy(Apr, Apr + nnzA, A.begin() + (size_t)i * nnzA);
        std::copy(c, c + n, C.begin() + (size_t)i * n); std::copy(b, b + p, Bv.begin() + (size_t)i * p);
        for (int k = 0; k < m; k++) H[(size_t)i * m + k] = h[k] + 1e-3 * i * (1 + (h[k] < 0 ? -h[k] : h[k]));
    }
    auto run = [&](const std::vector<int> &ids, std::vector<double> &x, std::vector<size_t> &iters) {
        auto t0 = std::chrono::steady_clock::now();
        EiCOS::BatchSolver s(n, m, p, nc, q.data(), m ? Gjc.data() : nullptr, m ? Gir.data() : nullptr, p ? Ajc.data() : nullptr, p ? Air.data() : nullptr, B, ids);
        s.updateData(m ? G.data() : nullptr, p ? A.data() : nullptr, C.data(), m ? H.data() : nullptr, p ? Bv.data() : nullptr);
        auto t1 = std::chrono::steady_clock::now();
        const std::vector<EiCOS::exitcode> codes = s.solve();
        auto t2 = std::chrono::steady_clock::now();
        x = s.solution();
        int ok = 0;
        for (auto cd : codes) ok += cd == EiCOS::exitcode::optimal;
        for (const auto &inf : s.getInfo()) iters.push_back(inf.iter);
        std::printf("%d shard(s): setup+update %.1f ms, solve %.1f ms, %d / %d optimal\n", s.num_shards(),
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(), ok, B);
        return ok;
    };
    std::vector<double> x_multi, x_one;
    std::vector<size_t> it_multi, it_one;
    const int ok_multi = run(devs, x_multi, it_multi), ok_one = run({devs[0]}, x_one, it_one);
    const bool same = x_multi.size() == x_one.size() && std::memcmp(x_multi.data(), x_one.data(), x_one.size() * sizeof(double)) == 0 && it_multi == it_one;
    std::printf("sharded vs single-device results: %s\n", same ? "bit-identical" : "DIFFERENT");
    // the one-call form on PINNED arrays: updateData(...) + solve() as one call -- every shard's solve kernel pulls its instances' inputs over
    // PCIe itself and writes x into the pinned result array
    bool same_fused = false;
    {
        EiCOS::BatchSolver s(n, m, p, nc, q.data(), m ? Gjc.data() : nullptr, m ? Gir.data() : nullptr, p ? Ajc.data() : nullptr, p ? Air.data() : nullptr, B, devs);
        auto pin = [&](const std::vector<double> &v) { double *d = EiCOS::BatchSolver::hostAlloc(std::max<size_t>(v.size(), 1)); std::copy(v.begin(), v.end(), d); return d; };
        double *pG = pin(G), *pA = pin(A), *pC = pin(C), *pH = pin(H), *pB = pin(Bv), *px = EiCOS::BatchSolver::hostAlloc((size_t)B * n);
        const std::vector<EiCOS::exitcode> codes = s.solve(m ? pG : nullptr, p ? pA : nullptr, pC, m ? pH : nullptr, p ? pB : nullptr, px);
        int ok = 0;
        for (auto cd : codes) ok += cd == EiCOS::exitcode::optimal;
        same_fused = ok == ok_multi && std::memcmp(px, x_multi.data(), x_multi.size() * sizeof(double)) == 0;
        std::printf("one-call solve on pinned arrays vs updateData + solve: %s\n", same_fused ? "bit-identical" : "DIFFERENT");
        for (double *d : {pG, pA, pC, pH, pB, px}) EiCOS::BatchSolver::hostFree(d);
    }
    return (same && same_fused && ok_multi == ok_one && ok_multi > 0) ? 0 : 1;
}
