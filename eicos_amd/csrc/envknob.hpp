// Experiment knobs (EICOS_* environment variables).  They steer which numeric path a handle takes (factor path, workgroup
// size, LDS placement, ...) and exist for tests that must reach every kernel variant and for A/B measurements.  A host
// application's environment must never switch paths silently, so the knobs are honoured ONLY when EICOS_EXPERIMENT=1 is
// set, values are range-checked (out of range -> the default), and the chosen paths are reported through eicos_batch_dims.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace eicos {
inline bool experiments_enabled() {
    const char *e = std::getenv("EICOS_EXPERIMENT");
    return e && std::strcmp(e, "1") == 0;
}
// A knob that is set but not honoured (no opt-in, or a value outside its range) is reported on stderr -- once per knob and process --
// so that an A/B script cannot silently measure the default path.
inline void knob_warn(const char *name, const char *why) {
    static std::mutex mu; // (handles are created on parallel host threads: eicos_multi_create)
    std::lock_guard<std::mutex> lk(mu);
    static char seen[32][40];
    static int nseen = 0;
    for (int i = 0; i < nseen; i++) if (std::strncmp(seen[i], name, 39) == 0) return;
    if (nseen < 32) { std::strncpy(seen[nseen], name, 39); seen[nseen][39] = '\0'; nseen++; }
    std::fprintf(stderr, "[eicos_amd] %s is set but ignored: %s\n", name, why);
}
inline int env_knob(const char *name, int dflt, int lo, int hi) {
    const char *v = std::getenv(name);
    if (!v || !*v) return dflt;
    if (!experiments_enabled()) { knob_warn(name, "experiment knobs need EICOS_EXPERIMENT=1"); return dflt; }
    char *end = nullptr;
    const long x = std::strtol(v, &end, 10);
    if (end == v || *end != '\0' || x < lo || x > hi) { knob_warn(name, "value outside the knob's range, default used"); return dflt; }
    return (int)x;
}
} // namespace eicos
