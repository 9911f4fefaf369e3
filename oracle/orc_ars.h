/* oracle/orc_ars.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Derivative-free adaptive rejection sampling (Gilks 1992; Gilks, Best & Tan
 * 1995 without the Metropolis step) as hydra's BayesW calls it:
 *   arms(xinit, 4, &xl, &xr, dens, data, &convex=1.0, npoint=100, dometrop=0,
 *        &xprev, xsamp, nsamp=1, qcent, xcent, ncent=4, &neval)
 * (src/BayesW.cpp:1336-1355, :1389, :1437, :1584; sampler in src/BayesW_arms.cpp).
 * The four requested centiles are computed by the reference and never read; they
 * consume no random numbers and are not computed here.
 *
 * PINNED: tests/test_bayesw_oracle.py runs this restatement and the reference's
 * own arms() (oracle/_ref/libarms.so, built from src/BayesW_arms.cpp where it
 * lies) on the same densities and the same libc rand() stream and requires
 * identical samples, evaluation counts and error codes.
 *
 * The envelope is a pool of nodes linked by index; nodes alternate between
 * bounds/chord intersections (on_curve = 0) and evaluated points of the log
 * density (on_curve = 1).  Uniforms come from libc rand(), as in the reference
 * (src/BayesW_arms.cpp:914-919).
 */
#pragma once
#include <math.h>
#include <stdlib.h>

#define ORC_ARS_XEPS 0.00001 /* critical relative x-value difference (src/BayesW_arms.cpp:56-59) */
#define ORC_ARS_YEPS 0.1     /* critical y-value difference */
#define ORC_ARS_EYEPS 0.001  /* critical relative exp(y) difference */
#define ORC_ARS_YCEIL 50.    /* maximum y avoiding overflow in exp(y) */
#define ORC_ARS_MAXNODES 128

typedef double (*orc_logdens_fn)(double x, void* data);

typedef struct {
    double x, y, ey, cum;
    int on_curve;
    int lt, rt; /* neighbours, -1 = none */
} orc_ars_node;

typedef struct {
    orc_ars_node n[ORC_ARS_MAXNODES];
    int used, cap;
    double ymax;
    orc_logdens_fn f;
    void* data;
    int* neval;
} orc_ars_env;

static inline double orc_ars_uniform(void) { return ((double)rand() + 0.5) / ((double)RAND_MAX + 1.0); }

static inline double orc_ars_expshift(double y, double y0)
{
    return (y - y0 > -2.0 * ORC_ARS_YCEIL) ? exp(y - y0 + ORC_ARS_YCEIL) : 0.0;
}

static inline double orc_ars_logshift(double y, double y0) { return log(y) + y0 - ORC_ARS_YCEIL; }

static inline double orc_ars_eval(orc_ars_env* e, double x)
{
    double y = e->f(x, e->data);
    (*e->neval)++;
    return y;
}

/* where the chords through the neighbouring curve points cross: src/BayesW_arms.cpp:683-799.
 * returns 0 ok, 1 envelope violation (log-concavity broken), >1 internal inconsistency */
static inline int orc_ars_meet(orc_ars_env* e, int q)
{
    orc_ars_node* N = e->n;
    const int L = N[q].lt, R = N[q].rt;
    int il = 0, ir = 0, irl = 0;
    double gl = 0, gr = 0, grl = 0, dl = 0, dr = 0;
    if (N[q].on_curve) return 2030;
    if (L >= 0 && N[N[L].lt].lt >= 0) {
        const int LL = N[N[L].lt].lt;
        gl = (N[L].y - N[LL].y) / (N[L].x - N[LL].x);
        il = 1;
    }
    if (R >= 0 && N[N[R].rt].rt >= 0) {
        const int RR = N[N[R].rt].rt;
        gr = (N[R].y - N[RR].y) / (N[R].x - N[RR].x);
        ir = 1;
    }
    if (L >= 0 && R >= 0) {
        grl = (N[R].y - N[L].y) / (N[R].x - N[L].x);
        irl = 1;
    }
    if (irl && il && gl < grl) return 1;
    if (irl && ir && gr > grl) return 1;
    if (il && irl) {
        dr = (gl - grl) * (N[R].x - N[L].x);
        if (dr < ORC_ARS_YEPS) dr = ORC_ARS_YEPS;
    }
    if (ir && irl) {
        dl = (grl - gr) * (N[R].x - N[L].x);
        if (dl < ORC_ARS_YEPS) dl = ORC_ARS_YEPS;
    }
    if (il && ir && irl) {
        N[q].x = (dl * N[R].x + dr * N[L].x) / (dl + dr);
        N[q].y = (dl * N[R].y + dr * N[L].y + dl * dr) / (dl + dr);
    } else if (il && irl) {
        N[q].x = N[R].x;
        N[q].y = N[R].y + dr;
    } else if (ir && irl) {
        N[q].x = N[L].x;
        N[q].y = N[L].y + dl;
    } else if (il) {
        N[q].y = N[L].y + gl * (N[q].x - N[L].x);
    } else if (ir) {
        N[q].y = N[R].y - gr * (N[R].x - N[q].x);
    } else {
        return 2031;
    }
    if ((L >= 0 && N[q].x < N[L].x) || (R >= 0 && N[q].x > N[R].x)) return 2032;
    return 0;
}

/* exponentiate and integrate the envelope: src/BayesW_arms.cpp:649-679, :803-825 */
static inline void orc_ars_cumulate(orc_ars_env* e)
{
    orc_ars_node* N = e->n;
    int first = 0, q;
    while (N[first].lt >= 0) first = N[first].lt;
    e->ymax = N[first].y;
    for (q = N[first].rt; q >= 0; q = N[q].rt)
        if (N[q].y > e->ymax) e->ymax = N[q].y;
    for (q = first; q >= 0; q = N[q].rt) N[q].ey = orc_ars_expshift(N[q].y, e->ymax);
    N[first].cum = 0.;
    for (q = N[first].rt; q >= 0; q = N[q].rt) {
        const orc_ars_node* p = &N[N[q].lt];
        double a;
        if (p->x == N[q].x) a = 0.;
        else if (fabs(N[q].y - p->y) < ORC_ARS_YEPS) a = 0.5 * (N[q].ey + p->ey) * (N[q].x - p->x);
        else a = ((N[q].ey - p->ey) / (N[q].y - p->y)) * (N[q].x - p->x);
        N[q].cum = p->cum + a;
    }
}

/* x with envelope mass `prob` to its left: src/BayesW_arms.cpp:378-451.  w is a scratch node. */
static inline int orc_ars_invert(orc_ars_env* e, double prob, orc_ars_node* w)
{
    orc_ars_node* N = e->n;
    int q = 0;
    double u, prop;
    while (N[q].rt >= 0) q = N[q].rt;
    u = prob * N[q].cum;
    while (N[N[q].lt].cum > u) q = N[q].lt;
    w->lt = N[q].lt;
    w->rt = q;
    w->on_curve = 0;
    w->cum = u;
    {
        const orc_ars_node* a = &N[N[q].lt];
        const orc_ars_node* b = &N[q];
        prop = (u - a->cum) / (b->cum - a->cum);
        if (a->x == b->x) {
            w->x = b->x;
            w->y = b->y;
            w->ey = b->ey;
            return 0;
        }
        if (fabs(b->y - a->y) < ORC_ARS_YEPS) {
            if (fabs(b->ey - a->ey) > ORC_ARS_EYEPS * fabs(b->ey + a->ey))
                w->x = a->x + ((b->x - a->x) / (b->ey - a->ey)) * (-a->ey + sqrt((1. - prop) * a->ey * a->ey + prop * b->ey * b->ey));
            else
                w->x = a->x + (b->x - a->x) * prop;
            w->ey = ((w->x - a->x) / (b->x - a->x)) * (b->ey - a->ey) + a->ey;
            w->y = orc_ars_logshift(w->ey, e->ymax);
        } else {
            w->x = a->x + ((b->x - a->x) / (b->y - a->y)) * (-a->y + orc_ars_logshift(((1. - prop) * a->ey + prop * b->ey), e->ymax));
            w->y = ((w->x - a->x) / (b->x - a->x)) * (b->y - a->y) + a->y;
            w->ey = orc_ars_expshift(w->y, e->ymax);
        }
        if (w->x < a->x || w->x > b->x) return 2001; /* the reference exit(1)s here */
    }
    return 0;
}

/* splice the evaluated point w into the envelope: src/BayesW_arms.cpp:556-645 */
static inline int orc_ars_insert(orc_ars_env* e, const orc_ars_node* w)
{
    orc_ars_node* N = e->n;
    int q, m, ql, qr, err;
    if (!w->on_curve || e->used > e->cap - 2) return 0; /* no room: the point is dropped */
    q = e->used++;
    N[q].x = w->x;
    N[q].y = w->y;
    N[q].on_curve = 1;
    m = e->used++;
    N[m].on_curve = 0;
    if (N[w->lt].on_curve && !N[w->rt].on_curve) { /* lt(curve) m q rt */
        N[m].lt = w->lt;
        N[m].rt = q;
        N[q].lt = m;
        N[q].rt = w->rt;
        N[N[m].lt].rt = m;
        N[N[q].rt].lt = q;
    } else if (!N[w->lt].on_curve && N[w->rt].on_curve) { /* lt q m rt(curve) */
        N[m].rt = w->rt;
        N[m].lt = q;
        N[q].rt = m;
        N[q].lt = w->lt;
        N[N[m].rt].lt = m;
        N[N[q].lt].rt = q;
    } else {
        return 2010;
    }
    ql = (N[N[q].lt].lt >= 0) ? N[N[q].lt].lt : N[q].lt;
    qr = (N[N[q].rt].rt >= 0) ? N[N[q].rt].rt : N[q].rt;
    if (N[q].x < (1. - ORC_ARS_XEPS) * N[ql].x + ORC_ARS_XEPS * N[qr].x) {
        N[q].x = (1. - ORC_ARS_XEPS) * N[ql].x + ORC_ARS_XEPS * N[qr].x;
        N[q].y = orc_ars_eval(e, N[q].x);
    } else if (N[q].x > ORC_ARS_XEPS * N[ql].x + (1. - ORC_ARS_XEPS) * N[qr].x) {
        N[q].x = ORC_ARS_XEPS * N[ql].x + (1. - ORC_ARS_XEPS) * N[qr].x;
        N[q].y = orc_ars_eval(e, N[q].x);
    }
    if ((err = orc_ars_meet(e, N[q].lt))) return err;
    if ((err = orc_ars_meet(e, N[q].rt))) return err;
    if (N[N[q].lt].lt >= 0 && (err = orc_ars_meet(e, N[N[N[q].lt].lt].lt))) return err;
    if (N[N[q].rt].rt >= 0 && (err = orc_ars_meet(e, N[N[N[q].rt].rt].rt))) return err;
    orc_ars_cumulate(e);
    return 0;
}

/* Same argument list and error codes as the reference's arms() so the two are
 * interchangeable behind one function pointer.  dometrop must be 0. */
static inline int orc_ars_arms(double* xinit, int ninit, double* xl, double* xr, orc_logdens_fn f, void* data, double* convex, int npoint,
                               int dometrop, double* xprev, double* xsamp, int nsamp, double* qcent, double* xcent, int ncent, int* neval)
{
    orc_ars_env env;
    orc_ars_env* e = &env;
    orc_ars_node* N = env.n;
    int i, j, k, total, got = 0;
    (void)xprev;
    (void)xcent;
    for (i = 0; i < ncent; i++)
        if (qcent[i] < 0.0 || qcent[i] > 100.0) return 1005;
    if (dometrop) return 1099; /* not restated: hydra never asks for the Metropolis step */
    if (ninit < 3) return 1001;
    total = 2 * ninit + 1;
    if (npoint < total) return 1002;
    if (npoint > ORC_ARS_MAXNODES) return 1006;
    if (xinit[0] <= *xl || xinit[ninit - 1] >= *xr) return 1003;
    for (i = 1; i < ninit; i++)
        if (xinit[i] <= xinit[i - 1]) return 1004;
    if (*convex < 0.0) return 1008;
    e->f = f;
    e->data = data;
    e->neval = neval;
    *neval = 0;
    e->cap = npoint;
    /* bound, point, crossing, point, ..., point, bound (src/BayesW_arms.cpp:300-343) */
    for (j = 0, k = 0; j < total; j++) {
        N[j].lt = j - 1;
        N[j].rt = (j + 1 < total) ? j + 1 : -1;
        N[j].on_curve = j % 2;
        N[j].y = 0.;
        if (j % 2) {
            N[j].x = xinit[k++];
            N[j].y = orc_ars_eval(e, N[j].x);
        }
    }
    N[0].x = *xl;
    N[total - 1].x = *xr;
    for (j = 0; j < total; j += 2)
        if (orc_ars_meet(e, j)) return 2000;
    orc_ars_cumulate(e);
    e->used = total;

    do {
        orc_ars_node w;
        double u, y, ynew;
        int err = orc_ars_invert(e, orc_ars_uniform(), &w);
        if (err) return err;
        /* rejection, squeezing: src/BayesW_arms.cpp:455-514 */
        u = orc_ars_uniform() * w.ey;
        y = orc_ars_logshift(u, e->ymax);
        if (N[w.lt].lt >= 0 && N[w.rt].rt >= 0) {
            const orc_ars_node* a = N[w.lt].on_curve ? &N[w.lt] : &N[N[w.lt].lt];
            const orc_ars_node* b = N[w.rt].on_curve ? &N[w.rt] : &N[N[w.rt].rt];
            const double ysq = (b->y * (w.x - a->x) + a->y * (b->x - w.x)) / (b->x - a->x);
            if (y <= ysq) {
                xsamp[got++] = w.x;
                continue;
            }
        }
        ynew = orc_ars_eval(e, w.x);
        w.y = ynew;
        w.ey = orc_ars_expshift(w.y, e->ymax);
        w.on_curve = 1;
        if (orc_ars_insert(e, &w)) return 2000;
        if (y < ynew) xsamp[got++] = w.x;
    } while (got < nsamp);
    return 0;
}
