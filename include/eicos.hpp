// include/eicos.hpp -- drop-in re-creation of the reference's public C++ surface
// (reference include/eicos.hpp:8-73,137-163) on top of the MI355X C ABI (eicos_amd.h).
//
//   EiCOS::Solver(n,m,p,l,ncones,q,Gpr,Gjc,Gir,Apr,Ajc,Air,c,h,b)   raw ctor     (ref :151-154)
//   updateData(Gpr,Apr,c,h,b)                                        raw update   (ref :155-156)
//   solve(verbose) / solution() / getInfo() / getSettings()                       (ref :158-163)
//   + the Eigen-typed ctor/updateData (ref :138-148) when <Eigen/Sparse> is available.
//
// One Solver = one pattern + ONE instance on the GPU (batch = 1 of the batched engine);
// EiCOS::BatchSolver below exposes the batched updateData path the hardware is built for.
// Header-only; link with libeicos_amd.so.  Errors of the C ABI surface as exitcode::fatal
// from solve() (the reference has no exceptions by design) or std::runtime_error from ctors.
#pragma once

#include <cstddef>
#include <cstdio>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "eicos_amd.h"

#if __has_include(<Eigen/Sparse>)
#include <Eigen/Sparse>
#define EICOS_HAVE_EIGEN 1
#endif

namespace EiCOS
{

    enum class exitcode // values of reference include/eicos.hpp:8-21
    {
        optimal = 0,
        primal_infeasible = 1,
        dual_infeasible = 2,
        maxit = -1,
        numerics = -2,
        outcone = -3,
        fatal = -7,
        close_to_optimal = 10,
        close_to_primal_infeasible = 11,
        close_to_dual_infeasible = 12,
        not_converged_yet = -87
    };

    struct Settings // reference include/eicos.hpp:23-47; all but `verbose` are compile-time
    {               // constants of the GPU kernels (eicos_amd/csrc/kernels.hip)
        const double gamma = 0.99;
        const double delta = 2e-7;
        const double deltastat = 7e-8;
        const double eps = 1e13;
        const double feastol = 1e-8;
        const double abstol = 1e-8;
        const double reltol = 1e-8;
        const double feastol_inacc = 1e-4;
        const double abstol_inacc = 5e-5;
        const double reltol_inacc = 5e-5;
        const size_t nitref = 9;
        const size_t maxit = 100;
        bool verbose = false;
        const double linsysacc = 1e-14;
        const double irerrfact = 6;
        const double stepmin = 1e-6;
        const double stepmax = 0.999;
        const double sigmamin = 1e-4;
        const double sigmamax = 1.;
        const size_t equil_iters = 3;
        const size_t iter_max = 100;
        const size_t safeguard = 500;
    };

    struct Information // reference include/eicos.hpp:49-73
    {
        double pcost = 0, dcost = 0, pres = 0, dres = 0;
        bool pinf = false, dinf = false;
        std::optional<double> pinfres, dinfres;
        double gap = 0;
        std::optional<double> relgap;
        double sigma = 0, mu = 0, step = 0, step_aff = 0, kapovert = 0;
        size_t iter = 0, iter_max = 100, nitref1 = 0, nitref2 = 0, nitref3 = 0;

        static Information from(const eicos_info &i)
        {
            Information o;
            o.pcost = i.pcost; o.dcost = i.dcost; o.pres = i.pres; o.dres = i.dres;
            o.pinf = i.pinf != 0; o.dinf = i.dinf != 0;
            if (i.has_pinfres) o.pinfres = i.pinfres;
            if (i.has_dinfres) o.dinfres = i.dinfres;
            o.gap = i.gap;
            if (i.has_relgap) o.relgap = i.relgap;
            o.sigma = i.sigma; o.mu = i.mu; o.step = i.step; o.step_aff = i.step_aff; o.kapovert = i.kapovert;
            o.iter = (size_t)i.iter; o.nitref1 = (size_t)i.nitref1; o.nitref2 = (size_t)i.nitref2; o.nitref3 = (size_t)i.nitref3;
            return o;
        }
    };

    namespace detail
    {
        inline void check(int rc, const char *what)
        {
            if (rc != EICOS_OK) throw std::runtime_error(std::string(what) + ": " + eicos_last_error());
        }
    }

    // Batched engine: one pattern, `batch` instances.  Arrays are [batch][...] row-major in global instance order.
    // One GPU (device, -1 = current) or several: with a list of device ids the batch is cut into contiguous shards, one per
    // list entry, solved concurrently (eicos_multi_* of eicos_amd.h; a device may be listed more than once).
    class BatchSolver
    {
    public:
        BatchSolver(int n, int m, int p, int ncones, const int *q,
                    const int *Gjc, const int *Gir, const int *Ajc, const int *Air, int batch, int device = -1)
            : BatchSolver(n, m, p, ncones, q, Gjc, Gir, Ajc, Air, batch, std::vector<int>{device}) {}
        BatchSolver(int n, int m, int p, int ncones, const int *q,
                    const int *Gjc, const int *Gir, const int *Ajc, const int *Air, int batch, const std::vector<int> &device_ids)
            : n_(n), m_(m), p_(p), batch_(batch)
        {
            mcheck(eicos_multi_create(n, m, p, -1, ncones, q, Gjc, Gir, Ajc, Air, batch, device_ids.data(), (int)device_ids.size(), &h_), "eicos_multi_create");
            eicos_dims d; eicos_batch_dims(handle(), &d);
            n_ = d.n; m_ = d.m; p_ = d.p;
        }
        BatchSolver(const BatchSolver &) = delete;
        BatchSolver &operator=(const BatchSolver &) = delete;
        ~BatchSolver() { eicos_multi_destroy(h_); }

        void updateData(const double *Gpr, const double *Apr, const double *c, const double *h, const double *b,
                        int first = 0, int count = -1)
        {
            mcheck(eicos_multi_update(h_, first, count < 0 ? batch_ : count, Gpr, Apr, c, h, b), "eicos_multi_update");
        }
        // inputs already resident in the HBM of GPU `src_device`: no PCIe traffic; shards on other GPUs pull their rows over xGMI
        void updateDataDevice(int src_device, const double *dGpr, const double *dApr, const double *dc, const double *dh, const double *db,
                              int first = 0, int count = -1)
        {
            mcheck(eicos_multi_update_device(h_, src_device, first, count < 0 ? batch_ : count, dGpr, dApr, dc, dh, db), "eicos_multi_update_device");
        }
        // Extension (not in the reference): re-solves start from the previous solution, see eicos_amd.h
        void setWarmStart(double shift) { mcheck(eicos_multi_set_warm_start(h_, shift), "eicos_multi_set_warm_start"); }
        // Extension: ECOS-style dynamic regularisation (the reference's Settings::delta / ::eps are never read)
        void setDynamicRegularization(double delta, double eps)
        {
            mcheck(eicos_multi_set_dynamic_regularization(h_, delta, eps), "eicos_multi_set_dynamic_regularization");
        }
        std::vector<exitcode> solve()
        {
            std::vector<int> codes(batch_);
            mcheck(eicos_multi_solve(h_, codes.data()), "eicos_multi_solve");
            std::vector<exitcode> out(batch_);
            for (int i = 0; i < batch_; i++) out[i] = static_cast<exitcode>(codes[i]);
            return out;
        }
        // updateData(...) + solve() in ONE call (reference include/eicos.hpp:155-158 back to back): with arrays from hostAlloc / hostRegister the
        // solve kernel's workgroups pull every instance's inputs over PCIe themselves, behind each other's compute, and write x into a pinned
        // `x_out` ([batch][n], optional) as instances finish; plain arrays take updateData + solve.  Same results on every path.
        std::vector<exitcode> solve(const double *Gpr, const double *Apr, const double *c, const double *h, const double *b, double *x_out = nullptr)
        {
            std::vector<int> codes(batch_);
            mcheck(eicos_multi_update_solve(h_, Gpr, Apr, c, h, b, x_out, codes.data()), "eicos_multi_update_solve");
            std::vector<exitcode> out(batch_);
            for (int i = 0; i < batch_; i++) out[i] = static_cast<exitcode>(codes[i]);
            return out;
        }
        void solveAsync() { mcheck(eicos_multi_solve_async(h_), "eicos_multi_solve_async"); } // enqueue on every shard's stream
        void sync() { mcheck(eicos_multi_sync(h_), "eicos_multi_sync"); }
        std::vector<double> solution() const
        {
            std::vector<double> x((size_t)batch_ * n_);
            if (n_ > 0) mcheck(eicos_multi_solution(h_, x.data()), "eicos_multi_solution");
            return x;
        }
        // x into caller-owned storage [batch][n] (pinned memory from hostAlloc: one strided copy per shard, no bounce)
        void solution(double *x) const { if (n_ > 0) mcheck(eicos_multi_solution(h_, x), "eicos_multi_solution"); }
        // Pinned host arrays (eicos_host_alloc): updateData reads them in place over PCIe instead of through the bounce buffers -- for the
        // arrays a closed loop rewrites every sample.  Not in the reference (which computes on the host).
        static double *hostAlloc(size_t doubles)
        {
            void *ptr = eicos_host_alloc(doubles * sizeof(double));
            if (!ptr) throw std::runtime_error(std::string("eicos_host_alloc: ") + eicos_last_error());
            return static_cast<double *>(ptr);
        }
        static void hostFree(double *ptr) { eicos_host_free(ptr); }
        // ... or pin storage the caller already owns (a std::vector's data()) in place; unregister before it is freed or reallocated
        static void hostRegister(double *ptr, size_t doubles)
        {
            if (eicos_host_register(ptr, doubles * sizeof(double)) != EICOS_OK) throw std::runtime_error(std::string("eicos_host_register: ") + eicos_last_error());
        }
        static void hostUnregister(double *ptr) { eicos_host_unregister(ptr); }
        std::vector<Information> getInfo() const
        {
            std::vector<eicos_info> raw(batch_);
            mcheck(eicos_multi_info(h_, raw.data()), "eicos_multi_info");
            std::vector<Information> out;
            for (auto &r : raw) out.push_back(Information::from(r));
            return out;
        }
        int batch() const { return batch_; }
        int n_var() const { return n_; }
        int num_shards() const { return eicos_multi_num_shards(h_); }
        eicos_batch *handle(int shard = 0) const
        {
            eicos_batch *b = nullptr;
            mcheck(eicos_multi_shard(h_, shard, &b, nullptr, nullptr, nullptr), "eicos_multi_shard");
            return b;
        }
        eicos_multi *multi_handle() const { return h_; }

    private:
        static void mcheck(int rc, const char *what)
        {
            if (rc != EICOS_OK) throw std::runtime_error(std::string(what) + ": " + eicos_multi_last_error());
        }
        eicos_multi *h_ = nullptr;
        int n_, m_, p_, batch_;
    };

    class Solver
    {
    public:
        // traditional interface (reference include/eicos.hpp:151-154); `l` is ignored as in the
        // reference (src/eicos.cpp:91); NULL groups are allowed (src/eicos.cpp:103-117)
        Solver(int n, int m, int p, int /*l*/, int ncones, int *q,
               double *Gpr, int *Gjc, int *Gir,
               double *Apr, int *Ajc, int *Air,
               double *c, double *h, double *b)
        {
            const bool haveG = Gpr && Gjc && Gir, haveA = Apr && Ajc && Air;
            if (!c) n = 0;
            detail::check(eicos_batch_create(n, haveG ? m : 0, haveA ? p : 0, -1 /* derived, ref src/eicos.cpp:91 */, haveG ? ncones : 0, q,
                                             haveG ? Gjc : nullptr, haveG ? Gir : nullptr,
                                             haveA ? Ajc : nullptr, haveA ? Air : nullptr, 1, -1, &h_),
                          "eicos_batch_create");
            eicos_dims d; eicos_batch_dims(h_, &d);
            resize_solution(d.n);
            // first data set: every group that exists must be supplied
            static double dummy = 0.0;
            detail::check(eicos_batch_update(h_, 0, 1, haveG ? Gpr : nullptr, haveA ? Apr : nullptr,
                                             d.n ? c : &dummy, haveG ? h : nullptr, haveA ? b : nullptr),
                          "eicos_batch_update");
        }
        Solver(const Solver &) = delete;
        Solver &operator=(const Solver &) = delete;
        ~Solver() { eicos_batch_destroy(h_); }

        // reference include/eicos.hpp:155-156 : NULL = keep; h is read only with Gpr, b only with Apr
        void updateData(double *Gpr, double *Apr, double *c, double *h, double *b)
        {
            detail::check(eicos_batch_update(h_, 0, 1, Gpr, Apr, c, h, b), "eicos_batch_update");
        }

#ifdef EICOS_HAVE_EIGEN
        // Eigen-typed surface (reference include/eicos.hpp:138-148).  Inputs are copied.
        Solver(const Eigen::SparseMatrix<double> &G, const Eigen::SparseMatrix<double> &A,
               const Eigen::VectorXd &c, const Eigen::VectorXd &h, const Eigen::VectorXd &b,
               const Eigen::VectorXi &soc_dims)
            : Solver(int(c.size()), int(G.rows()), int(A.rows()), 0, int(soc_dims.size()),
                     const_cast<int *>(soc_dims.data()),
                     const_cast<double *>(G.valuePtr()), const_cast<int *>(G.outerIndexPtr()), const_cast<int *>(G.innerIndexPtr()),
                     A.rows() ? const_cast<double *>(A.valuePtr()) : nullptr,
                     A.rows() ? const_cast<int *>(A.outerIndexPtr()) : nullptr,
                     A.rows() ? const_cast<int *>(A.innerIndexPtr()) : nullptr,
                     const_cast<double *>(c.data()), const_cast<double *>(h.data()), const_cast<double *>(b.data())) {}
        void updateData(const Eigen::SparseMatrix<double> &G, const Eigen::SparseMatrix<double> &A,
                        const Eigen::VectorXd &c, const Eigen::VectorXd &h, const Eigen::VectorXd &b)
        {
            updateData(const_cast<double *>(G.valuePtr()), A.rows() ? const_cast<double *>(A.valuePtr()) : nullptr,
                       const_cast<double *>(c.data()), const_cast<double *>(h.data()), const_cast<double *>(b.data()));
        }
#endif

        exitcode solve(bool verbose = false) // reference include/eicos.hpp:158
        {
            settings_.verbose = verbose;
            int code = EICOS_FATAL;
            if (eicos_batch_solve(h_, &code) != EICOS_OK) return exitcode::fatal;
            eicos_info raw;
            if (eicos_batch_info(h_, &raw) != EICOS_OK) return exitcode::fatal;
            info_ = Information::from(raw);
            if (x_.size() > 0) eicos_batch_solution(h_, x_.data());
            if (verbose)
                std::printf("EiCOS(MI355X): exit %d after %d iterations, pcost %.9g dcost %.9g pres %.1e dres %.1e gap %.1e\n",
                            code, raw.iter, raw.pcost, raw.dcost, raw.pres, raw.dres, raw.gap);
            return static_cast<exitcode>(code);
        }

        // reference include/eicos.hpp:160: `const Eigen::VectorXd &solution() const` -- a reference to solver-owned
        // storage, valid until the next solve() / destruction.  Same type when Eigen is available; without Eigen
        // (raw-pointer callers only) the storage is a std::vector<double>.
#ifdef EICOS_HAVE_EIGEN
        using SolutionVector = Eigen::VectorXd;
#else
        using SolutionVector = std::vector<double>;
#endif
        const SolutionVector &solution() const { return x_; }
        Settings &getSettings() { return settings_; }
        const Information &getInfo() const { return info_; }
        // Extension (not in the reference, which cold-starts every solve): warm-start the next solves after updateData
        void setWarmStart(double shift) { eicos_batch_set_warm_start(h_, shift); }
        void setDynamicRegularization(double delta, double eps) { eicos_batch_set_dynamic_regularization(h_, delta, eps); }

    private:
        eicos_batch *h_ = nullptr;
        Settings settings_;
        Information info_;
        SolutionVector x_;
        void resize_solution(int n)
        {
#ifdef EICOS_HAVE_EIGEN
            x_.setZero(n);
#else
            x_.assign((size_t)n, 0.0);
#endif
        }
    };

} // namespace EiCOS
