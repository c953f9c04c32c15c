// Multi-GPU layer of the C ABI (include/eicos_amd.h: eicos_multi_*): one sparsity pattern, `batch` instances in contiguous
// shards over a list of devices -- one eicos_batch handle (own stream) per list entry, no data-path collective (instances are
// independent: SURVEY.md 8e).  Host C++ only: no torch, no RCCL; inputs reach a shard over its own GPU's PCIe link (host
// pointers) or by peer copies over xGMI (inputs resident on one GPU).  A device may be listed more than once: its shards then
// run concurrently on separate streams of that GPU (what a single-GPU box can exercise, tests/test_gpu_parity.py).
#include "../../include/eicos_amd.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" int eicos_internal_update_staged(eicos_batch *h, int first, int count, const double *G, const double *A,
                                            const double *c, const double *hh, const double *b, int src_dev);
extern "C" int eicos_internal_device(const eicos_batch *h);

struct eicos_multi {
    int batch = 0, n = 0, m = 0, p = 0, nnzG = 0, nnzA = 0;
    std::vector<eicos_batch *> shard;
    std::vector<int> first, count, device;
};

namespace {
thread_local std::string g_merr;
int mfail(int code, const std::string &msg) { g_merr = msg; return code; }
// run fn(s) for every shard on its own host thread (blocking calls such as the chunked host-pointer updateData overlap across
// GPUs); returns the first failing shard's code and message
template <class F> int for_shards(eicos_multi *mh, F &&fn) {
    const int ns = (int)mh->shard.size();
    std::vector<int> rc(ns, EICOS_OK);
    std::vector<std::string> msg(ns);
    auto body = [&](int s) { rc[s] = fn(s); if (rc[s] != EICOS_OK) msg[s] = eicos_last_error(); };
    if (ns == 1) body(0);
    else {
        std::vector<std::thread> th;
        for (int s = 0; s < ns; s++) th.emplace_back(body, s);
        for (auto &t : th) t.join();
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
    const int base = batch / ndev, rem = batch % ndev; // contiguous shards, the first `rem` one instance longer (= eicos_amd.generate.shard_range)
    for (int s = 0; s < ndev; s++) { mh->first[s] = s * base + std::min(s, rem); mh->count[s] = base + (s < rem ? 1 : 0); }
    // every shard analyses the pattern and sets up its device on its own host thread (the analysis is deterministic: identical plans)
    const int rc = for_shards(mh, [&](int s) {
        return eicos_batch_create(n, m, p, l, ncones, q, Gjc, Gir, Ajc, Air, mh->count[s], device_ids[s], &mh->shard[s]);
    });
    if (rc != EICOS_OK) { const std::string keep = g_merr; eicos_multi_destroy(mh); g_merr = keep; return rc; }
    eicos_dims d;
    eicos_batch_dims(mh->shard[0], &d);
    mh->n = d.n; mh->m = d.m; mh->p = d.p; mh->nnzG = d.nnzG; mh->nnzA = d.nnzA;
    for (int s = 0; s < ndev; s++) mh->device[s] = eicos_internal_device(mh->shard[s]); // (a negative id was resolved to the current device)
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
        // (experiment knob, honoured like the others only under EICOS_EXPERIMENT=1: take the peer-copy path even on the source GPU, so
        // that a single-GPU box exercises it -- hipMemcpyPeerAsync with equal devices is a device-to-device copy)
        static const bool force_peer = [] { const char *e = std::getenv("EICOS_EXPERIMENT"), *k = std::getenv("EICOS_MULTI_FORCE_PEER");
                                            return e && !std::strcmp(e, "1") && k && !std::strcmp(k, "1"); }();
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
    for (size_t s = 0; s < mh->shard.size(); s++) {
        float ms = 0.f;
        const int rc = eicos_batch_last_solve_ms(mh->shard[s], &ms);
        if (rc != EICOS_OK) return mfail(rc, eicos_last_error());
        if (per_shard) per_shard[s] = ms;
        *ms_max = std::max(*ms_max, ms);
    }
    return EICOS_OK;
}

} // extern "C"
