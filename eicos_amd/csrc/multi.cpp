// Multi-GPU layer of the C ABI (include/eicos_amd.h: eicos_multi_*): one sparsity pattern, `batch` instances in contiguous
// shards over a list of devices -- one eicos_batch handle (own stream) per list entry, no data-path collective (instances are
// independent: SURVEY.md 8e).  Host C++ only: no torch, no RCCL; inputs reach a shard over its own GPU's PCIe link (host
// pointers) or by peer copies over xGMI (inputs resident on one GPU).  A device may be listed more than once: its shards then
// run concurrently on separate streams of that GPU (what a single-GPU box can exercise, tests/test_gpu_parity.py).
#include "../../include/eicos_amd.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" int eicos_internal_update_staged(eicos_batch *h, int first, int count, const double *G, const double *A,
                                            const double *c, const double *hh, const double *b, int src_dev);
extern "C" int eicos_internal_device(const eicos_batch *h);
extern "C" int eicos_internal_solve_span_ms(eicos_batch *from, eicos_batch *to, float *ms);

namespace {
// One persistent host thread per shard: blocking calls of the shards (symbolic analysis at creation, the chunked host-pointer
// updateData, result copies) overlap across GPUs without a thread being created per call.  post() hands the worker one job, wait()
// blocks until it has run.
class Worker {
  public:
    Worker() : th_([this] { run(); }) {}
    ~Worker() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu_); job_ = std::move(f); has_ = true; done_ = false; } cv_.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return done_; }); }
  private:
    void run() {
        for (;;) {
            std::function<void()> f;
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return stop_ || has_; }); if (!has_) return; f = std::move(job_); has_ = false; }
            f();
            { std::lock_guard<std::mutex> lk(mu_); done_ = true; }
            cv_.notify_all();
        }
    }
    std::mutex mu_; std::condition_variable cv_;
    std::function<void()> job_;
    bool has_ = false, done_ = true, stop_ = false;
    std::thread th_; // (last member: the thread starts after the state above is initialised)
};
} // namespace

struct eicos_multi {
    int batch = 0, n = 0, m = 0, p = 0, nnzG = 0, nnzA = 0;
    std::vector<eicos_batch *> shard;
    std::vector<int> first, count, device;
    std::vector<std::unique_ptr<Worker>> worker; // one per shard (none for a single shard: the caller's thread does the work)
    std::mutex call_mu;                          // one fan-out at a time (the workers hold one job each)
};

namespace {
thread_local std::string g_merr;
int mfail(int code, const std::string &msg) { g_merr = msg; return code; }
// run fn(s) for every shard on the shard's host thread; returns the first failing shard's code and message
template <class F> int for_shards(eicos_multi *mh, F &&fn) {
    const int ns = (int)mh->shard.size();
    std::vector<int> rc(ns, EICOS_OK);
    std::vector<std::string> msg(ns);
    auto body = [&](int s) { rc[s] = fn(s); if (rc[s] != EICOS_OK) msg[s] = eicos_last_error(); };
    if (ns == 1 || mh->worker.empty()) { for (int s = 0; s < ns; s++) body(s); }
    else {
        std::lock_guard<std::mutex> lk(mh->call_mu);
        for (int s = 0; s < ns; s++) mh->worker[s]->post([&body, s] { body(s); });
        for (int s = 0; s < ns; s++) mh->worker[s]->wait();
    }
    for (int s = 0; s < ns; s++) if (rc[s] != EICOS_OK) return mfail(rc[s], "shard " + std::to_string(s) + " (device " + std::to_string(mh->device[s]) + "): " + msg[s]);
    return EICOS_OK;
}
// shards that intersect the instance range [first, first + count): fn(s, first index inside the shard, count, offset in the caller's arrays)
template <class F> int for_range(eicos_multi *mh, int first, int count, F &&fn) {
    if (!mh) return mfail(EICOS_E_INVALID, "NULL handle");
    if (first < 0 || count < 0 || first + count > mh->batch) return mfail(EICOS_E_INVALID, "instance range out of bounds");
    return for_shards(mh, [&](int s) {
        const int a = std::max(first, mh->first[s]), b = std::min(first + count, mh->first[s] + mh->count[s]);
        if (a >= b) return (int)EICOS_OK;
        return fn(s, a - mh->first[s], b - a, (size_t)(a - first));
    });
}
const double *at(const double *p, size_t row, int width) { return p ? p + row * (size_t)width : nullptr; }
} // namespace

extern "C" {

const char *eicos_multi_last_error(void) { return g_merr.c_str(); }

int eicos_multi_create(int n, int m, int p, int l, int ncones, const int *q, const int *Gjc, const int *Gir, const int *Ajc, const int *Air,
                       int batch, const int *device_ids, int ndev, eicos_multi **out) {
    if (!out) return mfail(EICOS_E_INVALID, "out is NULL");
    *out = nullptr;
    if (ndev < 1 || !device_ids) return mfail(EICOS_E_INVALID, "need at least one device id");
    if (batch < ndev) return mfail(EICOS_E_INVALID, "batch smaller than the number of shards");
    eicos_multi *mh = new eicos_multi();
    mh->batch = batch;
    mh->shard.assign(ndev, nullptr); mh->first.resize(ndev); mh->count.resize(ndev); mh->device.assign(device_ids, device_ids + ndev);
    // a negative id means "the caller's current device": resolved HERE, on the calling thread (a worker thread's current device is 0)
    {
        int cur = 0, nvis = 0;
        if (hipGetDeviceCount(&nvis) != hipSuccess || nvis == 0) {
            const int d0 = mh->device[0];
            delete mh;
            return mfail(EICOS_E_NOGPU, "shard 0 (device " + std::to_string(d0) + "): no HIP device visible: the solver has no CPU fallback");
        }
        if (hipGetDevice(&cur) != hipSuccess) cur = 0;
        for (int &d : mh->device) { if (d < 0) d = cur; if (d >= nvis) { delete mh; return mfail(EICOS_E_INVALID, "device index out of range (" + std::to_string(nvis) + " visible)"); } }
    }
    if (ndev > 1) for (int s = 0; s < ndev; s++) mh->worker.emplace_back(new Worker());
    const int base = batch / ndev, rem = batch % ndev; // contiguous shards, the first `rem` one instance longer (= eicos_amd.generate.shard_range)
    for (int s = 0; s < ndev; s++) { mh->first[s] = s * base + std::min(s, rem); mh->count[s] = base + (s < rem ? 1 : 0); }
    // every shard analyses the pattern and sets up its device on its own host thread (the analysis is deterministic: identical plans)
    const int rc = for_shards(mh, [&](int s) {
        return eicos_batch_create(n, m, p, l, ncones, q, Gjc, Gir, Ajc, Air, mh->count[s], mh->device[s], &mh->shard[s]);
    });
    if (rc != EICOS_OK) { const std::string keep = g_merr; eicos_multi_destroy(mh); g_merr = keep; return rc; }
    eicos_dims d;
    eicos_batch_dims(mh->shard[0], &d);
    mh->n = d.n; mh->m = d.m; mh->p = d.p; mh->nnzG = d.nnzG; mh->nnzA = d.nnzA;
    // Peer access between every pair of distinct devices of the list, both directions: eicos_multi_update_device then reads inputs that
    // live on one GPU in place over xGMI.  A pair without it (or a failure to enable it) falls back to staged peer copies -- noted once.
    int caller_dev = 0;
    if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = 0;
    for (int a = 0; a < ndev; a++)
        for (int b = 0; b < ndev; b++) {
            const int da = mh->device[a], db = mh->device[b];
            if (da == db) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess) { (void)hipGetLastError(); can = 0; }
            hipError_t e = hipErrorUnknown;
            if (can && hipSetDevice(da) == hipSuccess) e = hipDeviceEnablePeerAccess(db, 0);
            (void)hipGetLastError();
            if (!can || (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)) {
                static bool noted = false;
                if (!noted) { noted = true; std::fprintf(stderr, "[eicos_amd] no peer access from device %d to device %d: eicos_multi_update_device will stage peer copies\n", da, db); }
            }
        }
    (void)hipSetDevice(caller_dev); // (enabling peer access switched devices: the caller's current device is restored)
    *out = mh;
    return EICOS_OK;
}

int eicos_multi_destroy(eicos_multi *mh) {
    if (!mh) return EICOS_OK;
    for (eicos_batch *h : mh->shard) eicos_batch_destroy(h);
    delete mh;
    return EICOS_OK;
}

int eicos_multi_num_shards(eicos_multi *mh) { return mh ? (int)mh->shard.size() : mfail(EICOS_E_INVALID, "NULL handle"); }

int eicos_multi_shard(eicos_multi *mh, int s, eicos_batch **handle, int *first, int *count, int *device) {
    if (!mh || s < 0 || s >= (int)mh->shard.size()) return mfail(EICOS_E_INVALID, "bad shard index");
    if (handle) *handle = mh->shard[s];
    if (first) *first = mh->first[s];
    if (count) *count = mh->count[s];
    if (device) *device = mh->device[s];
    return EICOS_OK;
}

int eicos_multi_update(eicos_multi *mh, int first, int count, const double *Gpr, const double *Apr, const double *c, const double *h, const double *b) {
    return for_range(mh, first, count, [&](int s, int f, int cnt, size_t off) {
        return eicos_batch_update(mh->shard[s], f, cnt, at(Gpr, off, mh->nnzG), at(Apr, off, mh->nnzA), at(c, off, mh->n), at(h, off, mh->m), at(b, off, mh->p));
    });
}

int eicos_multi_update_device(eicos_multi *mh, int src_device, int first, int count, const double *dGpr, const double *dApr, const double *dc,
                              const double *dh, const double *db) {
    if (src_device < 0) return mfail(EICOS_E_INVALID, "src_device must name the GPU that holds the inputs");
    return for_range(mh, first, count, [&](int s, int f, int cnt, size_t off) {
        const double *G = at(dGpr, off, mh->nnzG), *A = at(dApr, off, mh->nnzA), *cc = at(dc, off, mh->n), *hh = at(dh, off, mh->m), *bb = at(db, off, mh->p);
        // (experiment knob, honoured like the others only under EICOS_EXPERIMENT=1 and read on EVERY call: take the other-GPU path even on
        // the source GPU, so that a single-GPU box exercises it; eicos_batch_last_update_path tells which path a shard took)
        const char *e_ = std::getenv("EICOS_EXPERIMENT"), *k_ = std::getenv("EICOS_MULTI_FORCE_PEER");
        const bool force_peer = e_ && !std::strcmp(e_, "1") && k_ && !std::strcmp(k_, "1");
        if (mh->device[s] == src_device && !force_peer) return eicos_batch_update_device(mh->shard[s], f, cnt, G, A, cc, hh, bb); // already in this GPU's HBM: no copy
        return eicos_internal_update_staged(mh->shard[s], f, cnt, G, A, cc, hh, bb, src_device);                  // peer copies (xGMI), then the same kernel
    });
}

int eicos_multi_solve_async(eicos_multi *mh) {
    if (!mh) return mfail(EICOS_E_INVALID, "NULL handle");
    for (size_t s = 0; s < mh->shard.size(); s++) { // enqueue only: every shard's kernels start on its own stream, the call returns at once
        const int rc = eicos_batch_solve_async(mh->shard[s]);
        if (rc != EICOS_OK) return mfail(rc, "shard " + std::to_string(s) + ": " + eicos_last_error());
    }
    return EICOS_OK;
}

int eicos_multi_sync(eicos_multi *mh) {
    if (!mh) return mfail(EICOS_E_INVALID, "NULL handle");
    for (size_t s = 0; s < mh->shard.size(); s++) {
        const int rc = eicos_batch_sync(mh->shard[s]);
        if (rc != EICOS_OK) return mfail(rc, "shard " + std::to_string(s) + ": " + eicos_last_error());
    }
    return EICOS_OK;
}

int eicos_multi_info(eicos_multi *mh, eicos_info *info) {
    if (!mh || !info) return mfail(EICOS_E_INVALID, "NULL argument");
    return for_shards(mh, [&](int s) { return eicos_batch_info(mh->shard[s], info + mh->first[s]); });
}

int eicos_multi_solve(eicos_multi *mh, int *exitcodes) {
    int rc = eicos_multi_solve_async(mh);
    if (rc == EICOS_OK) rc = eicos_multi_sync(mh);
    if (rc != EICOS_OK || !exitcodes) return rc;
    std::vector<eicos_info> info(mh->batch);
    rc = eicos_multi_info(mh, info.data());
    if (rc != EICOS_OK) return rc;
    for (int i = 0; i < mh->batch; i++) exitcodes[i] = info[i].exitcode;
    return EICOS_OK;
}

// updateData + solve in one call over every shard (eicos_batch_update_solve per shard on its rows, the shards concurrently)
int eicos_multi_update_solve(eicos_multi *mh, const double *G, const double *A, const double *c, const double *hh, const double *b,
                             double *x_out, int *exitcodes) {
    if (!mh) return mfail(EICOS_E_INVALID, "NULL handle");
    return for_shards(mh, [&](int s) {
        const size_t r = (size_t)mh->first[s];
        return eicos_batch_update_solve(mh->shard[s], at(G, r, mh->nnzG), at(A, r, mh->nnzA), at(c, r, mh->n), at(hh, r, mh->m), at(b, r, mh->p),
                                        x_out ? x_out + r * (size_t)mh->n : nullptr, exitcodes ? exitcodes + r : nullptr);
    });
}

int eicos_multi_solution(eicos_multi *mh, double *x) {
    if (!mh || !x) return mfail(EICOS_E_INVALID, "NULL argument");
    if (mh->n == 0) return EICOS_OK;
    return for_shards(mh, [&](int s) { return eicos_batch_solution(mh->shard[s], x + (size_t)mh->first[s] * mh->n); });
}

int eicos_multi_duals(eicos_multi *mh, double *y, double *z, double *sl) {
    if (!mh) return mfail(EICOS_E_INVALID, "NULL handle");
    return for_shards(mh, [&](int s) {
        const size_t f = (size_t)mh->first[s];
        return eicos_batch_duals(mh->shard[s], y ? y + f * mh->p : nullptr, z ? z + f * mh->m : nullptr, sl ? sl + f * mh->m : nullptr);
    });
}

int eicos_multi_set_warm_start(eicos_multi *mh, double shift) {
    if (!mh) return mfail(EICOS_E_INVALID, "NULL handle");
    for (eicos_batch *h : mh->shard) { const int rc = eicos_batch_set_warm_start(h, shift); if (rc != EICOS_OK) return mfail(rc, eicos_last_error()); }
    return EICOS_OK;
}

int eicos_multi_set_dynamic_regularization(eicos_multi *mh, double delta, double eps) {
    if (!mh) return mfail(EICOS_E_INVALID, "NULL handle");
    for (eicos_batch *h : mh->shard) { const int rc = eicos_batch_set_dynamic_regularization(h, delta, eps); if (rc != EICOS_OK) return mfail(rc, eicos_last_error()); }
    return EICOS_OK;
}

int eicos_multi_last_solve_ms(eicos_multi *mh, float *ms_max, float *per_shard) {
    if (!mh || !ms_max) return mfail(EICOS_E_INVALID, "NULL argument");
    *ms_max = 0.f;
    const size_t ns = mh->shard.size();
    for (size_t s = 0; s < ns; s++) {
        float ms = 0.f;
        const int rc = eicos_batch_last_solve_ms(mh->shard[s], &ms);
        if (rc != EICOS_OK) return mfail(rc, eicos_last_error());
        if (per_shard) per_shard[s] = ms;
        *ms_max = std::max(*ms_max, ms);
    }
    // A device that holds several shards runs their launches partly one after the other: its solve lasted from the first start to the last
    // end over ITS shards (HIP events of one device are comparable across streams), not as long as its slowest shard alone.
    for (size_t a = 0; a < ns; a++)
        for (size_t b = 0; b < ns; b++) {
            if (a == b || mh->device[a] != mh->device[b]) continue;
            float span = 0.f;
            if (eicos_internal_solve_span_ms(mh->shard[a], mh->shard[b], &span) == EICOS_OK) *ms_max = std::max(*ms_max, span);
        }
    return EICOS_OK;
}

} // extern "C"
