/*
 * oracle/eicos_oracle.h -- C ABI of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The product path
 * (eicos_amd/, include/) never includes, links or calls anything in oracle/.
 *
 * The oracle is a CPU restatement (own code, no Eigen) of the algorithm in the reference
 * /root/reference/src/eicos.cpp; every function in eicos_oracle.cpp cites the reference
 * lines it follows.  Parity pins: see the header comment of eicos_oracle.cpp.
 */
#ifndef EICOS_ORACLE_H
#define EICOS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Mirror of EiCOS::Information (reference include/eicos.hpp:49-73) with the std::optional
 * members flattened into value + has_* flag, plus tau/kap and the exit code. */
typedef struct oracle_info {
    double pcost, dcost, pres, dres, gap, relgap, sigma, mu, step, step_aff, kapovert;
    double pinfres, dinfres, tau, kap;
    int has_relgap, has_pinfres, has_dinfres, pinf, dinf;
    int iter, nitref1, nitref2, nitref3, exitcode;
    /* bookkeeping the reference does not expose: totals over the whole solve() */
    int n_factor; /* numeric factorisations (1 + completed passes)       */
    int n_ldlsolve; /* LDL solves (first solve + refinement solves)        */
} oracle_info;

/* Same argument convention as the reference's raw constructor
 * (include/eicos.hpp:151-154): NULL groups allowed, `l` ignored. */
void *oracle_create(int n, int m, int p, int l, int ncones, const int *q,
                    const double *Gpr, const int *Gjc, const int *Gir,
                    const double *Apr, const int *Ajc, const int *Air,
                    const double *c, const double *h, const double *b);
/* reference updateData(double*...) include/eicos.hpp:155-156 : NULL = keep */
void oracle_update(void *s, const double *Gpr, const double *Apr,
                   const double *c, const double *h, const double *b);
int oracle_solve(void *s);
void oracle_get_info(void *s, oracle_info *out);
/* x is the only vector the reference exposes (solution()); y,z,s are extras for tests */
void oracle_get_x(void *s, double *x);
void oracle_get_yzs(void *s, double *y, double *z, double *sl);
void oracle_get_dims(void *s, int *dimK, int *nnzK, int *nnzL);
/* per-iteration history of the last solve: rows of {pcost,dcost,gap,pres,dres,kap/tau,mu,step,sigma,tau,kap,nitref3};
 * returns the number of rows available */
int oracle_get_trace(void *s, double *out, int max_rows);
/* N3 (not in the reference): shift > 0 enables the warm start described at Solver::warm_init; 0 = cold start */
void oracle_set_warm_start(void *s, double shift);
/* N4 (not in the reference): ECOS-style dynamic regularisation of the LDL' pivots; delta = 0 switches it off */
void oracle_set_dynamic_regularization(void *s, double delta, double eps);
void oracle_destroy(void *s);
/* test hooks: updateScalings + updateKKTScalings for a given (s, z) -> scaling block of K in cacheIndices order
 * (ref :411-479, :1691-1732, :1944-1987); the KKT matrix as it stands (upper CSC; returns nnz) */
int oracle_debug_scalings(void *s, const double *s_in, const double *z_in, double *V_out);
int oracle_debug_kkt(void *s, int *Kp, int *Ki, double *Kx);

/* CPU-baseline driver: solve `batch` instances that share one pattern, one instance at a
 * time per thread.  Each thread constructs its solver once (pattern setup, untimed), then all
 * threads start together and every instance goes through oracle_update + oracle_solve (the
 * reference's updateData path, src/run.cpp:34-50).  Arrays are [batch][...].  Returns the wall
 * seconds attributed to solve; the updateData share of the same wall time goes to *update_s. */
double oracle_batch_solve(int n, int m, int p, int ncones, const int *q,
                          const int *Gjc, const int *Gir, const int *Ajc, const int *Air,
                          int batch, const double *Gpr, const double *Apr,
                          const double *c, const double *h, const double *b,
                          int nthreads, int *exitcodes, int *iters, double *pcost,
                          double *x_out /* [batch][n] or NULL */, double *update_s,
                          long long *total_ldlsolves);

#ifdef __cplusplus
}
#endif
#endif
