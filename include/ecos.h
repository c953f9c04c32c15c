/*
 * include/ecos.h -- ECOS-style C shim over the MI355X library, drop-in for the reference's
 * test/ecos.h:11-34 (ECOS_setup / ECOS_solve / ECOS_updateData / ECOS_cleanup + exit-code macros
 * :36-44).  With this header (and minunit.h) on the include path the reference's own test data
 * headers (test/MPC/MPC02.h, test/LPnetlib/lp_*.h, ...) compile and run against libeicos_amd.so.
 * Only the C ABI of eicos_amd.h is used.
 */
#pragma once
#include "eicos_amd.h"

typedef int idxint;
typedef double pfloat;
typedef struct pwork { eicos_batch *h; } pwork;

static inline pwork *ECOS_setup(idxint n, idxint m, idxint p, idxint l, idxint ncones, idxint *q, idxint nexc,
                                pfloat *Gpr, idxint *Gjc, idxint *Gir, pfloat *Apr, idxint *Ajc, idxint *Air,
                                pfloat *c, pfloat *h, pfloat *b) {
    (void)nexc; /* no exponential cones (the reference ignores the argument too, test/ecos.h:11) */
    static pfloat zero = 0.0;
    pwork *w = new pwork{nullptr};
    if (eicos_create(c ? n : 0, m, p, l, ncones, q, Gpr, Gjc, Gir, Apr, Ajc, Air, c ? c : &zero, h, b, -1, &w->h) != EICOS_OK) {
        delete w;
        return nullptr;
    }
    return w;
}
static inline idxint ECOS_solve(pwork *w) {
    int code = EICOS_FATAL;
    if (eicos_solve(w->h, &code) != EICOS_OK) return EICOS_FATAL;
    return code;
}
static inline void ECOS_updateData(pwork *w, pfloat *Gpr, pfloat *Apr, pfloat *c, pfloat *h, pfloat *b) {
    eicos_update(w->h, Gpr, Apr, c, h, b);
}
static inline void ECOS_cleanup(pwork *w, idxint keepvars) {
    (void)keepvars;
    if (w) { eicos_destroy(w->h); delete w; }
}

#define ECOS_OPTIMAL (0)
#define ECOS_PINF (1)
#define ECOS_DINF (2)
#define ECOS_INACC_OFFSET (10)
#define ECOS_MAXIT (-1)
#define ECOS_NUMERICS (-2)
#define ECOS_OUTCONE (-3)
#define ECOS_SIGINT (-4)
#define ECOS_FATAL (-7)
