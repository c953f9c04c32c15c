// A caller in the shape of the reference's demo driver (reference src/run.cpp:11-50): raw CSC arrays mapped into
// Eigen types, EiCOS::Solver(G, A, c, h, b, q) -> solve -> updateData(G, A, c, h, b) -> solve, reading the result
// through `const Eigen::VectorXd &solution()`.  Data comes from an EPB1 fixture (value set 0, then set 1 if present)
// instead of the missing data_MPC01.hpp.  Built against tests/eigen_standin (Eigen is absent from this image).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "eicos.hpp"

#ifndef EICOS_HAVE_EIGEN
#error "the Eigen-typed surface was not enabled: Eigen/Sparse must be on the include path"
#endif

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), {});
    if (raw.size() < 36 || std::memcmp(raw.data(), "EPB1", 4)) return 2;
    int *hd = reinterpret_cast<int *>(raw.data() + 4);
    const int n = hd[0], m = hd[1], p = hd[2], ncones = hd[4], nnzG = hd[5], nnzA = hd[6], nsets = hd[7];
    int *q = hd + 8, *Gjc = q + ncones, *Gir = Gjc + n + 1, *Ajc = Gir + nnzG, *Air = Ajc + n + 1;
    double *vals = reinterpret_cast<double *>(Air + nnzA);
    const size_t per = (size_t)nnzG + nnzA + n + m + p;

    Eigen::SparseMatrix<double> G_, A_;
    Eigen::VectorXd c_, h_, b_;
    Eigen::VectorXi q_;
    auto map_set = [&](int k) {
        double *Gpr = vals + k * per, *Apr = Gpr + nnzG, *c = Apr + nnzA, *h = c + n, *b = h + m;
        if (m > 0) {
            G_ = Eigen::Map<Eigen::SparseMatrix<double>>(m, n, Gjc[n], Gjc, Gir, Gpr);
            q_ = Eigen::Map<Eigen::VectorXi>(q, ncones);
            h_ = Eigen::Map<Eigen::VectorXd>(h, m);
        }
        if (p > 0) {
            A_ = Eigen::Map<Eigen::SparseMatrix<double>>(p, n, Ajc[n], Ajc, Air, Apr);
            b_ = Eigen::Map<Eigen::VectorXd>(b, p);
        }
        c_ = Eigen::Map<Eigen::VectorXd>(c, n);
    };
    map_set(0);
    EiCOS::Solver solver(G_, A_, c_, h_, b_, q_);
    EiCOS::exitcode exitcode = solver.solve();
    const Eigen::VectorXd &x = solver.solution(); // reference include/eicos.hpp:160
    double cx = 0;
    for (Eigen::Index j = 0; j < x.size(); j++) cx += c_(j) * x(j);
    std::printf("solve 1: exit %d pcost %.7f c'x %.7f\n", int(exitcode), solver.getInfo().pcost, cx);
    if (exitcode != EiCOS::exitcode::optimal) return 1;

    map_set(nsets > 1 ? 1 : 0);
    solver.updateData(G_, A_, c_, h_, b_);
    exitcode = solver.solve();
    cx = 0;
    for (Eigen::Index j = 0; j < x.size(); j++) cx += c_(j) * x(j); // same reference: solver-owned storage
    std::printf("solve 2: exit %d pcost %.7f c'x %.7f\n", int(exitcode), solver.getInfo().pcost, cx);
    return exitcode == EiCOS::exitcode::optimal ? 0 : 1;
}
