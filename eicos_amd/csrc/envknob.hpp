// Experiment knobs (EICOS_* environment variables).  They steer which numeric path a handle takes (factor path, workgroup
// size, LDS placement, ...) and exist for tests that must reach every kernel variant and for A/B measurements.  A host
// application's environment must never switch paths silently, so the knobs are honoured ONLY when EICOS_EXPERIMENT=1 is
// set, values are range-checked (out of range -> the default), and the chosen paths are reported through eicos_batch_dims.
#pragma once
#include <cstdlib>
#include <cstring>

namespace eicos {
inline bool experiments_enabled() {
    const char *e = std::getenv("EICOS_EXPERIMENT");
    return e && std::strcmp(e, "1") == 0;
}
inline int env_knob(const char *name, int dflt, int lo, int hi) {
    if (!experiments_enabled()) return dflt;
    const char *v = std::getenv(name);
    if (!v || !*v) return dflt;
    char *end = nullptr;
    const long x = std::strtol(v, &end, 10);
    if (end == v || *end != '\0' || x < lo || x > hi) return dflt;
    return (int)x;
}
} // namespace eicos
