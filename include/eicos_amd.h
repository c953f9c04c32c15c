/*
 * include/eicos_amd.h -- C ABI of the MI355X batched SOCP interior-point solver.
 *
 * This is the drop-in boundary for the EiCOS hot path.  The reference has no FFI of its own;
 * its public surface is class EiCOS::Solver (reference include/eicos.hpp:137-163) and the
 * closest thing to a C ABI is the ECOS shim of reference test/ecos.h:11-34
 * (ECOS_setup / ECOS_solve / ECOS_updateData / ECOS_cleanup).  Each entry point below names
 * the reference interface it replaces.  include/eicos.hpp (this repo) re-creates
 * class EiCOS::Solver on top of this ABI with batch = 1; INTEGRATION.md shows the bindings.
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer it passes; all
 * functions return 0 on success or a negative EICOS_E_* code (never throw across the ABI);
 * eicos_last_error() returns a thread-local message for the last failure.
 * All arithmetic is fp64, indices are 32-bit int (Eigen's default StorageIndex).
 *
 * One handle = one sparsity pattern (G: m x n CSC, A: p x n CSC, cone sizes q) analysed once
 * on the host + `batch` numeric instances resident in HBM on one GPU.  Instances are
 * independent: several GPUs of one node are driven from ONE process through eicos_multi_* (below: contiguous shards, one handle
 * and stream per device), or from one process per GPU with one handle each (bench.py --gpus N under torch.distributed).
 */
#ifndef EICOS_AMD_H
#define EICOS_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct eicos_batch eicos_batch; /* opaque */

enum {
    EICOS_OK = 0,
    EICOS_E_INVALID = -1,   /* bad argument / inconsistent dimensions          */
    EICOS_E_NOGPU = -2,     /* no usable HIP device (there is NO CPU fallback) */
    EICOS_E_HIP = -3,       /* a HIP runtime call failed                       */
    EICOS_E_UNSUPPORTED = -4 /* pattern too dense for the current factor path  */
};

/* exit codes per instance: values of enum class EiCOS::exitcode (reference include/eicos.hpp:8-21) */
enum {
    EICOS_OPTIMAL = 0, EICOS_PINF = 1, EICOS_DINF = 2, EICOS_MAXIT = -1, EICOS_NUMERICS = -2,
    EICOS_OUTCONE = -3, EICOS_FATAL = -7, EICOS_INACC_OFFSET = 10
};

/* Mirror of struct EiCOS::Information (reference include/eicos.hpp:49-73); the three
 * std::optional<double> members are flattened into value + has_* flag.  tau/kap/exitcode and
 * the n_* counters are additions (the counters feed the roofline byte model of bench.py). */
typedef struct eicos_info {
    double pcost, dcost, pres, dres, gap, relgap, sigma, mu, step, step_aff, kapovert;
    double pinfres, dinfres, tau, kap;
    int has_relgap, has_pinfres, has_dinfres, pinf, dinf;
    int iter, nitref1, nitref2, nitref3, exitcode;
    int n_factor;   /* numeric LDL' factorisations in the last solve (1 + completed passes) */
    int n_ldlsolve; /* LDL' solves in the last solve (incl. refinement solves)             */
    int n_sweep;    /* passes over the factor L in the last solve: = n_ldlsolve, except that a solve of two right-hand
                     * sides at once (eicos_dims.dual_rhs) counts one pass for both                  */
    int reserved_;
    double solve_us; /* device wall time of this instance's last solve (its workgroup, microseconds): max / p95 over the batch
                      * against eicos_batch_last_solve_ms shows how much of a launch is its slowest instances */
} eicos_info;

/* Pattern / size report of a handle (for byte accounting and tests). */
typedef struct eicos_dims {
    int n, m, p, l, ncones, dim_K, nnzA, nnzG, nnzK, nnzL, nlevels, order_mode, batch, device;
    long long factor_pairs;      /* multiply-subtract pairs of one numeric factorisation */
    size_t inst_bytes, work_bytes, pattern_bytes;
    int threads_per_block;
    int resident_blocks; /* instances resident on the GPU at a time = resident workgroups */
    int lds_bytes; /* dynamic LDS per workgroup (solve vector staged in LDS), 0 if in HBM */
    int instances_per_block; /* always 1 (field kept for ABI stability: the lock-step pairs of round 2 were measured slower and removed) */
    int lds_resident; /* 1: small pattern, the solve works on LDS copies of the instance's slabs (lds_bytes includes them) */
    int factor_path;  /* 0 scalar level-scheduled program, 1 dense 16x16 tiles (MFMA), 2 hybrid: tiles for the top of the tree */
    int cone_order;   /* 1: the two expansion columns of every second-order cone are eliminated after the cone's own rows (the
                       * numerically preferable order, csrc/symbolic.cpp); 0: unconstrained minimum degree (or no cones) */
    int dual_rhs;     /* 1: the two independent KKT systems of the initialisation and of every pass are solved in one sweep */
    /* the arithmetic path of this handle (what decides the rounding of a result besides the data): the profile it was created under
     * (eicos_set_arithmetic_profile), the nodes of the dense apex (0 = none: level schedule to the root) and the slices of the
     * single-wavefront part of the two sweep plans (0 = no split); two handles on one pattern with equal threads_per_block, factor_path,
     * apex_nodes and solo_slices give bit-identical results for equal data */
    int arithmetic_profile, apex_nodes, solo_slices;
} eicos_dims;

/* Process-wide choice of how a handle's PLANS are shaped, read by eicos_batch_create / eicos_multi_create (no reference counterpart: the
 * reference has one code path).  0 (default): by the launch -- workgroup size, dense apex and the single-wavefront tree top follow the
 * batch size, so the last bits of an instance's result can depend on the batch (or shard) it is solved in.  1: by the pattern alone, as
 * for a batch beyond one workgroup per CU -- an instance gives the same bits in a batch of 1, of 4096 and in any shard of an eicos_multi
 * (at the price of the small-batch plan choices: a few percent below profile 0 there).  Returns EICOS_OK or EICOS_E_INVALID. */
int eicos_set_arithmetic_profile(int profile);
int eicos_get_arithmetic_profile(void);

/* ---- construction: replaces Solver::Solver(n,m,p,l,ncones,q,Gpr,Gjc,Gir,Apr,Ajc,Air,c,h,b)
 * (reference include/eicos.hpp:151-154, src/eicos.cpp:91-120) + build() (:132-187), for a
 * whole batch.  Pattern only; values arrive through eicos_batch_update*.  The reference ignores `l`
 * and derives it as m - sum(q) (src/eicos.cpp:91,155): pass l < 0 for exactly that; l >= 0 is checked
 * (l + sum(q) must equal m, else EICOS_E_INVALID).  The arrays are trusted to hold n+1 column pointers,
 * jc[n] row indices and ncones cone sizes.  NULL Gjc/Gir (or Ajc/Air) mean "no G" ("no A").
 * device < 0 selects the current HIP device. */
int eicos_batch_create(int n, int m, int p, int l, int ncones, const int *q,
                       const int *Gjc, const int *Gir, const int *Ajc, const int *Air,
                       int batch, int device, eicos_batch **out);

/* ---- updateData: replaces Solver::updateData(double*Gpr,double*Apr,double*c,double*h,double*b)
 * (reference include/eicos.hpp:155-156, src/eicos.cpp:2053-2082) for instances
 * [first, first+count).  Arrays are [count][nnzG], [count][nnzA], [count][n], [count][m],
 * [count][p], row-major, HOST pointers; NULL = keep that group (h is read only with Gpr, b only
 * with Apr, as in the reference).  Runs un-equilibrate -> copy-in -> setEquilibration (:302-374)
 * -> transposes -> KKT value refresh on the GPU. */
int eicos_batch_update(eicos_batch *hd, int first, int count,
                       const double *Gpr, const double *Apr,
                       const double *c, const double *h, const double *b);
/* How the host arrays travel (csrc/api.cpp: eicos_internal_update_staged).  PAGEABLE arrays (plain malloc / std::vector / numpy) go through
 * two pinned bounce buffers in chunks of ~16 MB: the host copies chunk k + 1 in while the updateData kernel of chunk k reads its inputs
 * straight out of the other buffer over PCIe; the call returns once the last chunk has been copied (the arrays are the caller's again),
 * with the kernels still in flight on the handle's stream.  PINNED arrays (eicos_host_alloc below, hipHostMalloc, hipHostRegister) are
 * read in place by ONE kernel launch; the call waits for it, so the arrays may be overwritten on return -- as with the reference's
 * synchronous updateData.  eicos_batch_last_update_path tells which path the most recent call took (1 bounce, 2 pinned in place,
 * 3 peer GPU in place, 4 staged peer copies). */
void *eicos_host_alloc(size_t bytes); /* pinned host memory the GPU addresses directly; NULL on failure */
int eicos_host_free(void *p);
/* ... or pin arrays the caller already owns IN PLACE (hipHostRegister): they then take the pinned path as well.  Registering costs about
 * as much as a few bounce copies of the same bytes -- worth it for arrays reused across calls; unregister before freeing them.  An array is
 * read or written in place only when its WHOLE extent is pinned (first byte, last byte and a probe every 2 MB are checked per call; anything
 * else takes the bounce path -- never a GPU page fault).  Registering a base pointer that is already registered returns EICOS_OK when the
 * existing registration covers [p, p + bytes) and EICOS_E_INVALID when it is smaller (the runtime would map nothing new); a registration
 * belongs to whoever made it: eicos_host_unregister(p) ends it for every user of that memory.  paths 5 / 6: eicos_batch_update_solve below. */
int eicos_host_register(void *p, size_t bytes);
int eicos_host_unregister(void *p);
int eicos_batch_last_update_path(eicos_batch *hd);
/* updateData + solve in ONE synchronous call for the whole batch: replaces Solver::updateData(double *...) followed by Solver::solve()
 * (reference include/eicos.hpp:155-158, src/eicos.cpp:2053-2082 + :848).  Same array conventions as eicos_batch_update (NULL = keep the group).
 * When every given array is memory the GPU addresses directly -- eicos_host_alloc / eicos_host_register memory, or device memory -- each
 * workgroup of the solve kernel runs updateData for the instance it is about to solve: the PCIe transfer is spread over the launch behind
 * the other workgroups' compute instead of preceding it (path 5 of eicos_batch_last_update_path).  PAGEABLE arrays take the bounce
 * pipeline of eicos_batch_update, then the solve (an experiment switch stages them instead -- the kernel is launched at once and the host
 * copies the arrays chunk by chunk into a pinned staging buffer WHILE it runs, one flag per chunk, path 6: faster on some hosts, slower on
 * others, off by default).  x_out: optional [batch][n] result array
 * (pinned host / device memory is written by the kernel as instances finish).  Handles without an LDS vector (patterns too large for LDS)
 * run eicos_batch_update + eicos_batch_solve (+ eicos_batch_solution).  Results are bit-identical on every path.  exitcodes: optional [batch]. */
int eicos_batch_update_solve(eicos_batch *hd, const double *Gpr, const double *Apr, const double *c, const double *h, const double *b,
                             double *x_out, int *exitcodes);
/* Same, DEVICE pointers (inputs already resident in HBM; no PCIe traffic). */
int eicos_batch_update_device(eicos_batch *hd, int first, int count,
                              const double *dGpr, const double *dApr,
                              const double *dc, const double *dh, const double *db);

/* ---- solve: replaces exitcode Solver::solve(bool) (reference include/eicos.hpp:158,
 * src/eicos.cpp:848-1262) for every instance of the batch.  exitcodes (host, [batch]) may be
 * NULL.  Synchronous: returns after the GPU work has completed. */
int eicos_batch_solve(eicos_batch *hd, int *exitcodes);
/* Asynchronous half: enqueue on the handle's stream and return; eicos_batch_sync waits. */
int eicos_batch_solve_async(eicos_batch *hd);
int eicos_batch_sync(eicos_batch *hd);

/* ---- results: replaces solution() (reference include/eicos.hpp:160) / getInfo() (:163).
 * x: [batch][n] host.  y,z,s are extras the reference keeps private; any may be NULL.  A pinned destination receives one strided
 * device-to-host copy; a pageable one is filled through the pinned bounce buffers (copy of chunk k + 1 in flight while chunk k is copied out). */
int eicos_batch_solution(eicos_batch *hd, double *x);
int eicos_batch_duals(eicos_batch *hd, double *y, double *z, double *s);
int eicos_batch_info(eicos_batch *hd, eicos_info *info /* [batch] */);
/* Device-resident results (no copy): pointer to instance 0's x and the stride in doubles. */
int eicos_batch_solution_device(eicos_batch *hd, const double **dx, size_t *stride_doubles);

/* ---- warm start (N3 of SURVEY.md 8f; NOT in the reference, whose solve() always cold-starts, src/eicos.cpp:855-984).
 * shift > 0: a solve of an instance whose previous solve ended OPTIMAL skips the two initialisation solves and starts
 * from that solution -- re-equilibrated, with s and z pushed into the cone (LP rows floored at shift * mean|.|, cone
 * heads at ||tail|| + the same margin), tau = kap = 1.  shift = 0 (default) restores the reference behaviour.
 * 0.1 is a good value for MPC re-solves (1 % data perturbation: 13-15 -> 8-10 iterations). */
int eicos_batch_set_warm_start(eicos_batch *hd, double shift);

/* ---- dynamic regularisation (N4 of SURVEY.md 8f; NOT in the reference, whose Settings::delta / ::eps are dead,
 * include/eicos.hpp:26,28).  delta > 0: during the numeric LDL' a pivot whose sign disagrees with the quasi-definite
 * sign pattern of the KKT matrix, or whose magnitude is below eps, is replaced by sign * delta (ECOS: delta = 2e-7,
 * eps = 1e-13).  delta = 0 (default): static regularisation only, an exactly zero pivot ends the solve with
 * EICOS_FATAL as in the reference (src/eicos.cpp:1166-1170). */
int eicos_batch_set_dynamic_regularization(eicos_batch *hd, double delta, double eps);

/* ---- plumbing */
int eicos_batch_dims(eicos_batch *hd, eicos_dims *out);
/* Which compilation of the solve kernel this handle launches (chosen at creation from pattern size and batch; no reference
 * counterpart): 0 = default build, 1 = LDS-resident build for small patterns (eicos_dims.lds_resident), 2 = the 256-thread kernel
 * compiled for two waves per SIMD (launches of at most two workgroups per CU), 3 = the 256- / 512-thread kernel with the factor operand
 * array resident in LDS (launches of one workgroup per CU whose factor fits the idle LDS).  Negative: error code. */
int eicos_batch_kernel_build(eicos_batch *hd);
/* Use a caller-owned HIP stream (hipStream_t passed as void*); NULL restores the own stream. */
int eicos_batch_set_stream(eicos_batch *hd, void *hip_stream);
/* HIP-event timing of the most recent solve / update kernels on the handle's stream (ms). */
int eicos_batch_last_solve_ms(eicos_batch *hd, float *ms);
int eicos_batch_last_update_ms(eicos_batch *hd, float *ms);
/* The durations (ms) of the most recent launches, oldest first: which = 0 the solve launches, 1 the updateData calls, 2 the span of a
 * whole step (start of the updateData call that preceded a solve launch -> end of that solve; meaningful when the two alternate).  The handle keeps a
 * ring of 64 event pairs, so a caller that enqueues K steps back to back (update + solve_async, no host synchronisation in between) can
 * read every launch's duration afterwards.  Returns the number written (<= cap, <= 64) or a negative error; waits for the most recent
 * launch to finish.  No reference counterpart (measurement only). */
int eicos_batch_ms_history(eicos_batch *hd, int which, float *ms, int cap);
/* replaces the Solver destructor / ECOS_cleanup (reference test/ecos.h:31-34) */
int eicos_batch_destroy(eicos_batch *hd);

/* ---- single-instance surface: the same entry points for one problem, i.e. exactly what
 * class EiCOS::Solver needs.  A handle made by eicos_create is a batch of one; every eicos_batch_* call works on it.
 *   eicos_create   <-> Solver::Solver(n,m,p,l,ncones,q,Gpr,Gjc,Gir,Apr,Ajc,Air,c,h,b)  include/eicos.hpp:151-154
 *                      / ECOS_setup (reference test/ecos.h:11-24); values are copied, NULL groups allowed
 *   eicos_update   <-> Solver::updateData(Gpr,Apr,c,h,b)  include/eicos.hpp:155-156 / ECOS_updateData test/ecos.h:28-29
 *   eicos_solve    <-> exitcode Solver::solve()  include/eicos.hpp:158 / ECOS_solve test/ecos.h:26; returns the
 *                      exit code through *exitcode (the function's own return value is the EICOS_E_* status)
 *   eicos_solution <-> Solver::solution()  include/eicos.hpp:160 (x[n], host)
 *   eicos_info_get <-> Solver::getInfo()   include/eicos.hpp:163
 *   eicos_destroy  <-> ~Solver / ECOS_cleanup test/ecos.h:31-34 */
int eicos_create(int n, int m, int p, int l, int ncones, const int *q,
                 const double *Gpr, const int *Gjc, const int *Gir,
                 const double *Apr, const int *Ajc, const int *Air,
                 const double *c, const double *h, const double *b, int device, eicos_batch **out);
int eicos_update(eicos_batch *hd, const double *Gpr, const double *Apr, const double *c, const double *h, const double *b);
int eicos_solve(eicos_batch *hd, int *exitcode);
int eicos_solution(eicos_batch *hd, double *x);
int eicos_info_get(eicos_batch *hd, eicos_info *info);
int eicos_destroy(eicos_batch *hd);

/* ---- multi-GPU (SURVEY.md 8b "eicos_batch_create(pattern, B, device_ids...)", 8e): ONE pattern, `batch` instances in contiguous
 * shards over the listed devices -- shard s = instances [s*base + min(s, rem), ...) with base = batch / ndev, rem = batch % ndev, the
 * first `rem` shards one instance longer.  Host C++ above the single-GPU entry points: one eicos_batch handle and one HIP stream per
 * list entry, no collective on the data path (instances are independent), no torch / RCCL dependency.  The reference has no
 * counterpart (EiCOS::Solver solves one problem on one core, include/eicos.hpp:137-163); every call mirrors its eicos_batch_*
 * namesake over the whole batch, arrays [batch][...] row-major in GLOBAL instance order.  A device may be listed several times: its
 * shards then run concurrently on separate streams of that GPU.  Errors: eicos_multi_last_error() (names the failing shard). */
typedef struct eicos_multi eicos_multi; /* opaque */
int eicos_multi_create(int n, int m, int p, int l, int ncones, const int *q,
                       const int *Gjc, const int *Gir, const int *Ajc, const int *Air,
                       int batch, const int *device_ids, int ndev, eicos_multi **out);
/* updateData from HOST arrays: every shard moves its rows over its own GPU's PCIe link (eicos_batch_update's pinned bounce pipeline), all
 * shards in parallel -- one persistent host thread per shard, started by eicos_multi_create */
int eicos_multi_update(eicos_multi *mh, int first, int count, const double *Gpr, const double *Apr,
                       const double *c, const double *h, const double *b);
/* updateData + solve in one synchronous call (eicos_batch_update_solve on every shard, concurrently; whole batch; x_out / exitcodes optional) */
int eicos_multi_update_solve(eicos_multi *mh, const double *Gpr, const double *Apr, const double *c, const double *h, const double *b,
                             double *x_out, int *exitcodes);
/* updateData from arrays resident in the HBM of ONE GPU (src_device): shards on that GPU read them in place; the others read their rows
 * in place as well, over xGMI (peer access is enabled between the listed devices at creation), or -- without peer access -- pull them with
 * staged hipMemcpyPeerAsync copies on their own streams: the "batch scatter" of north_star without a collective.
 * ASYNCHRONOUS, like eicos_batch_update_device: the call returns with the updateData kernels enqueued on the shards' streams, and those
 * kernels read the source buffers IN PLACE (on src_device itself and, with peer access, from the other GPUs).  The source buffers must stay
 * valid and unmodified until eicos_multi_sync (or a solve / result call, which synchronise) has returned; work that PRODUCES them on
 * src_device must have completed before the call (the shards' streams are not ordered against the producer's stream). */
int eicos_multi_update_device(eicos_multi *mh, int src_device, int first, int count, const double *dGpr, const double *dApr,
                              const double *dc, const double *dh, const double *db);
/* solve: async = enqueue every shard's kernels on its stream and return; sync waits for all; eicos_multi_solve = both (+ exit codes, may be NULL) */
int eicos_multi_solve_async(eicos_multi *mh);
int eicos_multi_sync(eicos_multi *mh);
int eicos_multi_solve(eicos_multi *mh, int *exitcodes);
/* results gathered into the caller's host arrays in global instance order (the "gather" of north_star: per-device copies) */
int eicos_multi_solution(eicos_multi *mh, double *x);
int eicos_multi_duals(eicos_multi *mh, double *y, double *z, double *s);
int eicos_multi_info(eicos_multi *mh, eicos_info *info /* [batch] */);
int eicos_multi_set_warm_start(eicos_multi *mh, double shift);
int eicos_multi_set_dynamic_regularization(eicos_multi *mh, double delta, double eps);
/* the shards: their number, and shard s's single-GPU handle (every eicos_batch_* call works on it), instance range and device */
int eicos_multi_num_shards(eicos_multi *mh);
int eicos_multi_shard(eicos_multi *mh, int s, eicos_batch **handle, int *first, int *count, int *device);
/* HIP-event duration of the most recent solve (ms): per_shard ([num_shards], optional) = every shard's own launch; ms_max = the slowest
 * DEVICE -- for a device that holds several shards, from the start of its first launch to the end of its last one */
int eicos_multi_last_solve_ms(eicos_multi *mh, float *ms_max, float *per_shard);
int eicos_multi_destroy(eicos_multi *mh);
const char *eicos_multi_last_error(void);

const char *eicos_last_error(void);
/* number of visible HIP devices (0 when there is no GPU); never initialises a context */
int eicos_device_count(void);

/* Debug/parity hooks (tests only): numeric LDL' of instance `inst` with the KKT values as they
 * stand, and one LDL' solve. Host buffers. */
int eicos_debug_factor(eicos_batch *hd, int inst, double *D /*[dim_K], permuted*/, double *U /*[nnzL] CSC, permuted*/);
/* per-iteration history of instance `inst` in the last solve: out[102][12] = {pcost,dcost,gap,pres,dres,
 * kap/tau,mu,step,sigma,tau,kap,nitref3} per IPM pass (valid while batch <= resident workgroups). */
int eicos_debug_trace(eicos_batch *hd, int inst, double *out);
int eicos_debug_pattern(eicos_batch *hd, int *perm /*[dim_K]*/, int *Lp /*[dim_K+1]*/, int *Li /*[nnzL]*/);
/* upper triangle of instance `inst`'s KKT matrix as the numeric factorisation reads it (equilibrated A/G values,
 * scaling block, +-delta), coordinate form in the reference's column layout (src/eicos.cpp:1734-1890); each [nnzK] */
int eicos_debug_kkt(eicos_batch *hd, int inst, int *rows, int *cols, double *vals);
/* the solver's own updateScalings + updateKKTScalings stage (reference src/eicos.cpp:1160-1162) for a given (s, z):
 * V[l + sum(3 q_i + 1)] = scaling block of K in the slot order of reference cacheIndices (:1944-1987); *ran = 1 if
 * the stage reached the scalings.  The cone state persists between calls, as it does between iterations. */
int eicos_debug_scalings(eicos_batch *hd, int inst, const double *s, const double *z, double *V, int *ran);

/* Host-only self check of the symbolic analysis + factor/solve programs (no GPU needed):
 * returns ||K x - b||_inf / ||b||_inf for random quasi-definite values, < 0 on error. */
double eicos_debug_host_check(int n, int m, int p, int ncones, const int *q, const int *Gjc, const int *Gir,
                              const int *Ajc, const int *Air, unsigned seed, int order_mode, int *stats);

/* the same for the tile (dense-front) path: block factorisation over 16 x 16 tiles + tile sweeps emulated on the host
 * with the device's index structures; stats = {dim_K, nnzK, nnzL, block levels, tile pairs, order_mode, blocks, tiles} */
double eicos_debug_host_check_tiles(int n, int m, int p, int ncones, const int *q, const int *Gjc, const int *Gir,
                                    const int *Ajc, const int *Air, unsigned seed, int order_mode, int *stats);

/* host-only: elimination order (perm[dim_K]: new -> KKT index), block partition (blk_ptr[blocks + 1]) and L pattern of the tile path; returns the
 * number of blocks; stats = {dim_K, nnzL, blocks, off-diagonal tiles, block levels, tile pairs, order_mode, cone_order} */
int eicos_debug_host_tile_order(int n, int m, int p, int ncones, const int *q, const int *Gjc, const int *Gir,
                                const int *Ajc, const int *Air, int order_mode, int *perm, int *blk_ptr, int *stats,
                                int *Lp /* [dim_K + 1] or NULL */, int *Li /* [nnzL] or NULL: the pattern of L in that order, CSC */);

/* the same for the hybrid path (scalar programs below the cut, tiles on the top block of the tree); returns -10 when the
 * pattern's schedule has no tail worth handing to the tile path */
double eicos_debug_host_check_hybrid(int n, int m, int p, int ncones, const int *q, const int *Gjc, const int *Gir,
                                     const int *Ajc, const int *Air, unsigned seed, int order_mode, int *stats);

#ifdef __cplusplus
}
#endif
#endif
