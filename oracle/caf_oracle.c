/*
 * caf_oracle.c -- CPU restatement of the caf_rust filterbank CAF hot path.
 * TEST INFRASTRUCTURE ONLY (see caf_oracle.h).  Parity pin: the ten KATs of
 * caf_rust/tests/test.rs on tests/golden/data (tests/test_oracle_kats.py).
 *
 * Each function cites the reference lines it follows.  Arithmetic order is
 * kept where the reference states one (mixer recurrence, divide-by-n before the
 * inverse transform, strict '>' scans); the FFT is this file's own (rustfft's
 * source is not in the reference tree).
 */
#include "caf_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846264338327950288
#endif

/* ------------------------------------------------------------------ FFT -- */
/* Stockham autosort, radix-4 with one radix-2 stage when log2(n) is odd.
 * Unnormalised in both directions, like rustfft (xcor_rustfft.rs:32-35). */

typedef struct {
    size_t n;
    double *tw; /* tw[2k],tw[2k+1] = cos,sin(2*pi*k/n), k in [0,n) */
} fft_plan;

static int is_pow2(size_t n) { return n && !(n & (n - 1)); }

static fft_plan *plan_new(size_t n)
{
    fft_plan *p = (fft_plan *)malloc(sizeof *p);
    if (!p) return NULL;
    p->n = n;
    p->tw = (double *)malloc(2 * n * sizeof(double));
    if (!p->tw) { free(p); return NULL; }
    for (size_t k = 0; k < n; ++k) {
        double ang = 2.0 * M_PI * (double)k / (double)n;
        p->tw[2 * k] = cos(ang);
        p->tw[2 * k + 1] = sin(ang);
    }
    return p;
}

static void plan_free(fft_plan *p)
{
    if (p) { free(p->tw); free(p); }
}

/* x -> result ends up in x or y; returns pointer to the buffer holding it.
 * sgn = -1 forward (e^{-i}), +1 inverse (e^{+i}). */
static double *fft_run(const fft_plan *pl, double *x, double *y, int sgn)
{
    const size_t N = pl->n;
    const double *tw = pl->tw;
    size_t n = N, s = 1;
    const double sg = (double)sgn;
    while (n >= 4) {
        const size_t n1 = n / 4;
        const size_t tstep = N / n;
        for (size_t p = 0; p < n1; ++p) {
            const double w1r = tw[2 * (p * tstep)], w1i = sg * tw[2 * (p * tstep) + 1];
            const double w2r = tw[2 * (2 * p * tstep)], w2i = sg * tw[2 * (2 * p * tstep) + 1];
            const double w3r = tw[2 * (3 * p * tstep)], w3i = sg * tw[2 * (3 * p * tstep) + 1];
            for (size_t q = 0; q < s; ++q) {
                const double *a = x + 2 * (q + s * (p + 0 * n1));
                const double *b = x + 2 * (q + s * (p + 1 * n1));
                const double *c = x + 2 * (q + s * (p + 2 * n1));
                const double *d = x + 2 * (q + s * (p + 3 * n1));
                const double apcr = a[0] + c[0], apci = a[1] + c[1];
                const double amcr = a[0] - c[0], amci = a[1] - c[1];
                const double bpdr = b[0] + d[0], bpdi = b[1] + d[1];
                /* (sgn*i)*(b-d) */
                const double bmdr = b[0] - d[0], bmdi = b[1] - d[1];
                const double jr = -sg * bmdi, ji = sg * bmdr;
                double *y0 = y + 2 * (q + s * (4 * p + 0));
                double *y1 = y + 2 * (q + s * (4 * p + 1));
                double *y2 = y + 2 * (q + s * (4 * p + 2));
                double *y3 = y + 2 * (q + s * (4 * p + 3));
                y0[0] = apcr + bpdr;
                y0[1] = apci + bpdi;
                const double t1r = amcr + jr, t1i = amci + ji;
                y1[0] = w1r * t1r - w1i * t1i;
                y1[1] = w1r * t1i + w1i * t1r;
                const double t2r = apcr - bpdr, t2i = apci - bpdi;
                y2[0] = w2r * t2r - w2i * t2i;
                y2[1] = w2r * t2i + w2i * t2r;
                const double t3r = amcr - jr, t3i = amci - ji;
                y3[0] = w3r * t3r - w3i * t3i;
                y3[1] = w3r * t3i + w3i * t3r;
            }
        }
        { double *t = x; x = y; y = t; }
        n = n1;
        s *= 4;
    }
    if (n == 2) {
        for (size_t q = 0; q < s; ++q) {
            const double *a = x + 2 * q;
            const double *b = x + 2 * (q + s);
            double *y0 = y + 2 * q;
            double *y1 = y + 2 * (q + s);
            y0[0] = a[0] + b[0];
            y0[1] = a[1] + b[1];
            y1[0] = a[0] - b[0];
            y1[1] = a[1] - b[1];
        }
        { double *t = x; x = y; y = t; }
    }
    return x;
}

int oracle_fft(const double *in, double *out, size_t n, int inverse)
{
    if (!is_pow2(n)) return -1;
    fft_plan *pl = plan_new(n);
    double *x = (double *)malloc(2 * n * sizeof(double));
    double *y = (double *)malloc(2 * n * sizeof(double));
    if (!pl || !x || !y) { plan_free(pl); free(x); free(y); return -2; }
    memcpy(x, in, 2 * n * sizeof(double));
    double *r = fft_run(pl, x, y, inverse ? +1 : -1);
    memcpy(out, r, 2 * n * sizeof(double));
    plan_free(pl); free(x); free(y);
    return 0;
}

/* ---------------------------------------------------------------- mixer -- */
/* mod.rs:46-65.  dt = 1.0/(fs as f64); shift = from_polar(1, 2*PI*f*dt)
 * (evaluated left to right: ((2.0*PI)*f)*dt); accum starts at 1+0j and each
 * sample is multiplied BEFORE the accumulator advances. */
void oracle_apply_freq_shift(const double *in, size_t n, double freq_shift,
                             uint32_t fs, double *out)
{
    const double dt = 1.0 / (double)fs;
    const double ph = 2.0 * M_PI * freq_shift * dt;
    const double sr = cos(ph), si = sin(ph); /* from_polar(1.0, ph) */
    double ar = 1.0, ai = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double xr = in[2 * i], xi = in[2 * i + 1];
        out[2 * i] = xr * ar - xi * ai;     /* *samp *= accum_shift  (mod.rs:59) */
        out[2 * i + 1] = xr * ai + xi * ar;
        const double nr = ar * sr - ai * si; /* accum_shift *= shift  (mod.rs:60) */
        const double ni = ar * si + ai * sr;
        ar = nr; ai = ni;
    }
}

/* ----------------------------------------------------------------- xcor -- */
struct oracle_xcor {
    size_t n;
    fft_plan *plan;      /* shared between clones (xcor_rustfft.rs:82-93) */
    int owns_plan;
    double *a, *b, *c, *t; /* scratch (xcor_rustfft.rs:17-19) + ping-pong */
};

static oracle_xcor *xcor_alloc(size_t n, fft_plan *shared)
{
    oracle_xcor *x = (oracle_xcor *)calloc(1, sizeof *x);
    if (!x) return NULL;
    x->n = n;
    x->plan = shared ? shared : plan_new(n);
    x->owns_plan = shared == NULL;
    x->a = (double *)calloc(2 * n, sizeof(double));
    x->b = (double *)calloc(2 * n, sizeof(double));
    x->c = (double *)calloc(2 * n, sizeof(double));
    x->t = (double *)calloc(2 * n, sizeof(double));
    if (!x->plan || !x->a || !x->b || !x->c || !x->t) { oracle_xcor_free(x); return NULL; }
    return x;
}

oracle_xcor *oracle_xcor_new(size_t n)
{
    if (!is_pow2(n)) return NULL;
    return xcor_alloc(n, NULL);
}

void oracle_xcor_free(oracle_xcor *x)
{
    if (!x) return;
    if (x->owns_plan) plan_free(x->plan);
    free(x->a); free(x->b); free(x->c); free(x->t);
    free(x);
}

static void fft_into(const fft_plan *pl, const double *in, double *dst,
                     double *s0, double *s1, int sgn)
{
    memcpy(s0, in, 2 * pl->n * sizeof(double));
    double *r = fft_run(pl, s0, s1, sgn);
    memcpy(dst, r, 2 * pl->n * sizeof(double));
}

/* B = FFT(b) conj'd and multiplied against a given A = FFT(a); out = IFFT(A*conj(B)/n) */
static void xcor_tail(oracle_xcor *x, const double *A, const double *b, double *out)
{
    const size_t n = x->n;
    fft_into(x->plan, b, x->c, x->a, x->t, -1);          /* xcor_rustfft.rs:60-61 */
    for (size_t i = 0; i < n; ++i) x->c[2 * i + 1] = -x->c[2 * i + 1]; /* :64-66 */
    const double nn = (double)n;
    for (size_t i = 0; i < n; ++i) {                     /* :69-73  (a*b)/n */
        const double pr = A[2 * i] * x->c[2 * i] - A[2 * i + 1] * x->c[2 * i + 1];
        const double pi = A[2 * i] * x->c[2 * i + 1] + A[2 * i + 1] * x->c[2 * i];
        x->a[2 * i] = pr / nn;
        x->a[2 * i + 1] = pi / nn;
    }
    double *r = fft_run(x->plan, x->a, x->t, +1);        /* :76 unnormalised */
    memcpy(out, r, 2 * n * sizeof(double));              /* :77 */
}

int oracle_xcor_run(oracle_xcor *x, const double *a, const double *b, double *out)
{
    if (!x) return -1;
    fft_into(x->plan, a, x->b, x->a, x->t, -1);          /* xcor_rustfft.rs:58-59 */
    xcor_tail(x, x->b, b, out);
    return 0;
}

/* -------------------------------------------------------------- surface -- */
typedef struct {
    const double *needle_p, *haystack_p, *H; /* padded (2n) inputs, hoisted FFT or NULL */
    size_t L;
    const double *freqs;
    size_t nfreq;
    uint32_t fs;
    double *surface;
    uint64_t *row_idx;
    double *row_val;
    fft_plan *plan;
    size_t next;           /* work counter (threads variant) */
    pthread_mutex_t mu;
} surf_job;

/* One row: mod.rs:138-161 */
static void surf_row(const surf_job *j, oracle_xcor *xc, double *shifted,
                     double *res, size_t r)
{
    const size_t L = j->L;
    oracle_apply_freq_shift(j->needle_p, L, j->freqs[r], j->fs, shifted); /* :138 */
    if (j->H) {
        xcor_tail(xc, j->H, shifted, res);
    } else {
        oracle_xcor_run(xc, j->haystack_p, shifted, res);                  /* :139 */
    }
    double max = 0.0;                                                      /* :143 */
    uint64_t argmax = 0;
    double *mag = j->surface ? j->surface + r * L : NULL;
    for (size_t i = 0; i < L; ++i) {
        const double m = res[2 * i] * res[2 * i] + res[2 * i + 1] * res[2 * i + 1]; /* norm_sqr :147 */
        if (m > max) { max = m; argmax = i; }                               /* :148-151 */
        if (mag) mag[i] = m;
    }
    j->row_idx[r] = argmax;
    j->row_val[r] = max;
}

static int surf_setup(surf_job *j, const double *needle, const double *haystack,
                      size_t n, int hoist, double **np_, double **hp_, double **H_)
{
    const size_t L = 2 * n;                                   /* mod.rs:130-131 */
    if (!is_pow2(L)) return -1;
    double *np = (double *)calloc(2 * L, sizeof(double));
    double *hp = (double *)calloc(2 * L, sizeof(double));
    if (!np || !hp) { free(np); free(hp); return -2; }
    memcpy(np, needle, 2 * n * sizeof(double));               /* zeros at the END */
    memcpy(hp, haystack, 2 * n * sizeof(double));
    j->plan = plan_new(L);
    if (!j->plan) { free(np); free(hp); return -2; }
    double *H = NULL;
    if (hoist) {
        H = (double *)malloc(2 * L * sizeof(double));
        double *s0 = (double *)malloc(2 * L * sizeof(double));
        double *s1 = (double *)malloc(2 * L * sizeof(double));
        if (!H || !s0 || !s1) { free(np); free(hp); free(H); free(s0); free(s1); return -2; }
        fft_into(j->plan, hp, H, s0, s1, -1);
        free(s0); free(s1);
    }
    j->needle_p = np; j->haystack_p = hp; j->H = H; j->L = L;
    *np_ = np; *hp_ = hp; *H_ = H;
    return 0;
}

int oracle_caf_surface(const double *needle, const double *haystack, size_t n,
                       const double *freqs_hz, size_t nfreq, uint32_t fs,
                       double *surface, uint64_t *row_idx, double *row_val,
                       int hoist)
{
    return oracle_caf_surface_threads(needle, haystack, n, freqs_hz, nfreq, fs,
                                      surface, row_idx, row_val, hoist, 1);
}

static void *surf_worker(void *arg)
{
    surf_job *j = (surf_job *)arg;
    oracle_xcor *xc = xcor_alloc(j->L, j->plan);   /* Xcor::clone, xcor_rustfft.rs:82-93 */
    double *shifted = (double *)malloc(2 * j->L * sizeof(double));
    double *res = (double *)malloc(2 * j->L * sizeof(double));
    if (xc && shifted && res) {
        for (;;) {
            pthread_mutex_lock(&j->mu);
            size_t r = j->next++;
            pthread_mutex_unlock(&j->mu);
            if (r >= j->nfreq) break;
            surf_row(j, xc, shifted, res, r);
        }
    }
    oracle_xcor_free(xc); free(shifted); free(res);
    return NULL;
}

int oracle_caf_surface_threads(const double *needle, const double *haystack,
                               size_t n, const double *freqs_hz, size_t nfreq,
                               uint32_t fs, double *surface, uint64_t *row_idx,
                               double *row_val, int hoist, int nthreads)
{
    surf_job j;
    memset(&j, 0, sizeof j);
    double *np = NULL, *hp = NULL, *H = NULL;
    int rc = surf_setup(&j, needle, haystack, n, hoist, &np, &hp, &H);
    if (rc) return rc;
    j.freqs = freqs_hz; j.nfreq = nfreq; j.fs = fs;
    j.surface = surface; j.row_idx = row_idx; j.row_val = row_val;
    pthread_mutex_init(&j.mu, NULL);
    if (nthreads <= 1) {
        surf_worker(&j);
    } else {
        pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof *th);
        int started = 0;
        for (int t = 0; t < nthreads; ++t)
            if (pthread_create(&th[started], NULL, surf_worker, &j) == 0) ++started;
        if (started == 0) surf_worker(&j);
        for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
        free(th);
    }
    pthread_mutex_destroy(&j.mu);
    plan_free(j.plan); free(np); free(hp); free(H);
    return 0;
}

/* mod.rs:31-42 */
void oracle_find_peak(const double *freqs_hz, const uint64_t *row_idx,
                      const double *row_val, size_t nfreq, double *best_freq,
                      uint64_t *best_idx)
{
    double bf = 0.0, bv = 0.0;
    uint64_t bi = 0;
    for (size_t r = 0; r < nfreq; ++r) {
        if (row_val[r] > bv) { bv = row_val[r]; bf = freqs_hz[r]; bi = row_idx[r]; }
    }
    *best_freq = bf;
    *best_idx = bi;
}

/* tests/test.rs:335-352: `as i32` / `as usize` truncate toward zero. */
size_t oracle_gen_float_shifts(double start, double end, double step,
                               double *out, size_t cap)
{
    const int32_t s = (int32_t)(start * 1000.0);
    const int32_t e = (int32_t)(end * 1000.0);
    const size_t st = (size_t)(step * 1000.0);
    size_t cnt = 0;
    if (st == 0) return 0; /* Rust's step_by(0) panics; callers never pass it */
    for (int64_t m = s; m < e; m += (int64_t)st) {
        if (out && cnt < cap) out[cnt] = (double)m / 1e3;
        ++cnt;
    }
    return cnt;
}
