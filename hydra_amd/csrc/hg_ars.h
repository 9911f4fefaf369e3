// hg_ars.h -- adaptive rejection sampling for BayesW's scalar conditionals, and the
// uniform stream it draws from.
//
// What it replaces: arms() of src/BayesW_arms.cpp as BayesW calls it (src/BayesW.cpp:1336-1355,
// :1389, :1437, :1584): four starting abscissae, bounds [xl, xr], at most 100 envelope points,
// no Metropolis step, one sample; uniforms ((double)rand() + 0.5) / (RAND_MAX + 1.0) from libc
// rand() (src/BayesW_arms.cpp:914-919), seeded with srand(seed) (src/BayesW.cpp:1012).
//
// Shape here: the piecewise-exponential upper hull lives in one array kept sorted by x.  Entries
// alternate  bound/crossing, curve point, crossing, curve point, ..., bound  so "left neighbour"
// is index - 1 and a new curve point is spliced in by shifting the tail by two.  No allocation,
// no recursion, a functor for the log density: the same code compiles for the host driver and for
// a device lane.  The generator is a private restatement of glibc's TYPE_3 additive-feedback
// random() so that each chain owns its stream (libc's is process-global).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HG_ARS_HD __host__ __device__
#else
#define HG_ARS_HD
#endif

namespace hg {

// glibc random_r.c, TYPE_3 (x^31 + x^3 + 1), as srand()/rand() run it: 31 words of state seeded by
// the Lehmer generator 16807 mod 2^31-1 (Schrage form), 310 outputs discarded, output = word >> 1.
struct GlibcRand {
    int32_t r[31];
    int32_t f, b; // front / rear index

    HG_ARS_HD void seed(uint32_t s)
    {
        if (s == 0) s = 1;
        r[0] = (int32_t)s;
        int32_t word = (int32_t)s;
        for (int i = 1; i < 31; ++i) {
            const long hi = word / 127773, lo = word % 127773;
            word = (int32_t)(16807 * lo - 2836 * hi);
            if (word < 0) word += 2147483647;
            r[i] = word;
        }
        f = 3;
        b = 0;
        for (int i = 0; i < 310; ++i) (void)next();
    }

    HG_ARS_HD int32_t next()
    {
        const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
        r[f] = (int32_t)v;
        if (++f >= 31) f = 0;
        if (++b >= 31) b = 0;
        return (int32_t)(v >> 1);
    }

    HG_ARS_HD double uniform() { return ((double)next() + 0.5) / (2147483647.0 + 1.0); }
};

namespace ars {

constexpr double XEPS = 0.00001; // src/BayesW_arms.cpp:56-59
constexpr double YEPS = 0.1;
constexpr double EYEPS = 0.001;
constexpr double YCEIL = 50.;
constexpr int MAX_POINTS = 100; // npoint at every call site
constexpr int NINIT = 4;

enum Error {
    OK = 0,
    BAD_BOUNDS = 1003,   // starting abscissae not strictly inside (xl, xr)
    NOT_ORDERED = 1004,  // starting abscissae not increasing
    NOT_CONCAVE = 2000,  // the hull found the log density non-concave
    OUT_OF_PIECE = 2001, // inversion left its piece (the reference exits the process here)
};

struct Hull {
    double x[MAX_POINTS], y[MAX_POINTS], ey[MAX_POINTS], cum[MAX_POINTS];
    int n;       // entries in use; even index = bound/crossing, odd index = curve point
    double ymax; // largest y on the hull
};

HG_ARS_HD inline double expshift(double y, double y0) { return (y - y0 > -2.0 * YCEIL) ? exp(y - y0 + YCEIL) : 0.0; }
HG_ARS_HD inline double logshift(double y, double y0) { return log(y) + y0 - YCEIL; }

// crossing of the chords through the curve points either side of even entry q
// (src/BayesW_arms.cpp:683-799); false = concavity violated
HG_ARS_HD inline bool place_crossing(Hull& h, int q)
{
    const bool hasL = q - 1 >= 0, hasR = q + 1 < h.n;
    const bool il = hasL && q - 3 >= 0, ir = hasR && q + 3 < h.n, irl = hasL && hasR;
    double gl = 0, gr = 0, grl = 0, dl = 0, dr = 0;
    if (il) gl = (h.y[q - 1] - h.y[q - 3]) / (h.x[q - 1] - h.x[q - 3]);
    if (ir) gr = (h.y[q + 1] - h.y[q + 3]) / (h.x[q + 1] - h.x[q + 3]);
    if (irl) grl = (h.y[q + 1] - h.y[q - 1]) / (h.x[q + 1] - h.x[q - 1]);
    if (irl && il && gl < grl) return false;
    if (irl && ir && gr > grl) return false;
    if (il && irl) {
        dr = (gl - grl) * (h.x[q + 1] - h.x[q - 1]);
        if (dr < YEPS) dr = YEPS;
    }
    if (ir && irl) {
        dl = (grl - gr) * (h.x[q + 1] - h.x[q - 1]);
        if (dl < YEPS) dl = YEPS;
    }
    if (il && ir && irl) {
        h.x[q] = (dl * h.x[q + 1] + dr * h.x[q - 1]) / (dl + dr);
        h.y[q] = (dl * h.y[q + 1] + dr * h.y[q - 1] + dl * dr) / (dl + dr);
    } else if (il && irl) {
        h.x[q] = h.x[q + 1];
        h.y[q] = h.y[q + 1] + dr;
    } else if (ir && irl) {
        h.x[q] = h.x[q - 1];
        h.y[q] = h.y[q - 1] + dl;
    } else if (il) { // right bound
        h.y[q] = h.y[q - 1] + gl * (h.x[q] - h.x[q - 1]);
    } else if (ir) { // left bound
        h.y[q] = h.y[q + 1] - gr * (h.x[q + 1] - h.x[q]);
    } else {
        return false;
    }
    return !((hasL && h.x[q] < h.x[q - 1]) || (hasR && h.x[q] > h.x[q + 1]));
}

// exponentiate relative to the maximum and integrate piece by piece (src/BayesW_arms.cpp:649-679, :803-825)
HG_ARS_HD inline void integrate(Hull& h)
{
    h.ymax = h.y[0];
    for (int i = 1; i < h.n; ++i)
        if (h.y[i] > h.ymax) h.ymax = h.y[i];
    for (int i = 0; i < h.n; ++i) h.ey[i] = expshift(h.y[i], h.ymax);
    h.cum[0] = 0.;
    for (int i = 1; i < h.n; ++i) {
        double a;
        if (h.x[i - 1] == h.x[i]) a = 0.;
        else if (fabs(h.y[i] - h.y[i - 1]) < YEPS) a = 0.5 * (h.ey[i] + h.ey[i - 1]) * (h.x[i] - h.x[i - 1]);
        else a = ((h.ey[i] - h.ey[i - 1]) / (h.y[i] - h.y[i - 1])) * (h.x[i] - h.x[i - 1]);
        h.cum[i] = h.cum[i - 1] + a;
    }
}

struct Draw {
    double x, y, ey;
    int right; // index of the hull entry to the right of the draw
};

// the point with hull mass `prob` to its left (src/BayesW_arms.cpp:378-451)
HG_ARS_HD inline int invert(const Hull& h, double prob, Draw& d)
{
    int q = h.n - 1;
    const double u = prob * h.cum[q];
    while (h.cum[q - 1] > u) --q;
    d.right = q;
    const double xl = h.x[q - 1], xr = h.x[q], yl = h.y[q - 1], yr = h.y[q], eyl = h.ey[q - 1], eyr = h.ey[q];
    const double prop = (u - h.cum[q - 1]) / (h.cum[q] - h.cum[q - 1]);
    if (xl == xr) {
        d.x = xr;
        d.y = yr;
        d.ey = eyr;
        return OK;
    }
    if (fabs(yr - yl) < YEPS) {
        if (fabs(eyr - eyl) > EYEPS * fabs(eyr + eyl)) d.x = xl + ((xr - xl) / (eyr - eyl)) * (-eyl + sqrt((1. - prop) * eyl * eyl + prop * eyr * eyr));
        else d.x = xl + (xr - xl) * prop;
        d.ey = ((d.x - xl) / (xr - xl)) * (eyr - eyl) + eyl;
        d.y = logshift(d.ey, h.ymax);
    } else {
        d.x = xl + ((xr - xl) / (yr - yl)) * (-yl + logshift(((1. - prop) * eyl + prop * eyr), h.ymax));
        d.y = ((d.x - xl) / (xr - xl)) * (yr - yl) + yl;
        d.ey = expshift(d.y, h.ymax);
    }
    return (d.x < xl || d.x > xr) ? OUT_OF_PIECE : OK;
}

// splice the evaluated point (x, y) in before entry `right` (src/BayesW_arms.cpp:556-645)
template <class LogDensity>
HG_ARS_HD inline int splice(Hull& h, int right, double x, double y, LogDensity& f, int& neval)
{
    if (h.n > MAX_POINTS - 2) return OK; // full: the point is dropped
    // neighbours are (curve, crossing) or (crossing, curve); either way two entries go in at `right`
    for (int i = h.n - 1; i >= right; --i) {
        h.x[i + 2] = h.x[i];
        h.y[i + 2] = h.y[i];
    }
    h.n += 2;
    const int q = (right & 1) ? right : right + 1; // the new curve point takes the odd slot
    h.x[q] = x;
    h.y[q] = y;
    // too close to the neighbouring curve point (or bound): nudge it inwards and re-evaluate
    const int ql = (q - 2 >= 0) ? q - 2 : q - 1, qr = (q + 2 < h.n) ? q + 2 : q + 1;
    const double lo = (1. - XEPS) * h.x[ql] + XEPS * h.x[qr], hi = XEPS * h.x[ql] + (1. - XEPS) * h.x[qr];
    if (h.x[q] < lo) {
        h.x[q] = lo;
        h.y[q] = f(h.x[q]);
        ++neval;
    } else if (h.x[q] > hi) {
        h.x[q] = hi;
        h.y[q] = f(h.x[q]);
        ++neval;
    }
    if (!place_crossing(h, q - 1)) return NOT_CONCAVE;
    if (!place_crossing(h, q + 1)) return NOT_CONCAVE;
    if (q - 2 >= 0 && !place_crossing(h, q - 3)) return NOT_CONCAVE;
    if (q + 2 < h.n && !place_crossing(h, q + 3)) return NOT_CONCAVE;
    integrate(h);
    return OK;
}

// One draw from the density exp(f) restricted to [xl, xr].  neval counts density evaluations.
template <class LogDensity, class Uniform>
HG_ARS_HD inline int sample(const double (&xinit)[NINIT], double xl, double xr, LogDensity& f, Uniform& unif, Hull& h, double& out, int& neval)
{
    neval = 0;
    if (xinit[0] <= xl || xinit[NINIT - 1] >= xr) return BAD_BOUNDS;
    for (int i = 1; i < NINIT; ++i)
        if (xinit[i] <= xinit[i - 1]) return NOT_ORDERED;
    h.n = 2 * NINIT + 1;
    for (int j = 0; j < h.n; ++j) {
        h.x[j] = 0.;
        h.y[j] = 0.;
    }
    h.x[0] = xl;
    h.x[h.n - 1] = xr;
    for (int k = 0; k < NINIT; ++k) {
        h.x[2 * k + 1] = xinit[k];
        h.y[2 * k + 1] = f(xinit[k]);
        ++neval;
    }
    for (int j = 0; j < h.n; j += 2)
        if (!place_crossing(h, j)) return NOT_CONCAVE;
    integrate(h);

    for (;;) {
        Draw d;
        const int err = invert(h, unif(), d);
        if (err) return err;
        const double u = unif() * d.ey;
        const double y = logshift(u, h.ymax);
        const int left = d.right - 1;
        if (left - 1 >= 0 && d.right + 1 < h.n) { // squeeze under the chord of the enclosing curve points
            const int a = (left & 1) ? left : left - 1, b = (d.right & 1) ? d.right : d.right + 1;
            const double ysq = (h.y[b] * (d.x - h.x[a]) + h.y[a] * (h.x[b] - d.x)) / (h.x[b] - h.x[a]);
            if (y <= ysq) {
                out = d.x;
                return OK;
            }
        }
        const double ynew = f(d.x);
        ++neval;
        const int serr = splice(h, d.right, d.x, ynew, f, neval);
        if (serr) return NOT_CONCAVE;
        if (y < ynew) {
            out = d.x;
            return OK;
        }
    }
}

} // namespace ars
} // namespace hg
