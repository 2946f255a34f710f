/* TEST INFRASTRUCTURE -- not part of the product.  PARITY UNPINNED (see oracle/ref_numpy.py and DESIGN.md section 5).
 *
 * C / OpenMP restatement of the calamity fit step on the ragged problem layout, used as
 *   (1) a second, independently written checker of the HIP path at sizes the NumPy restatement cannot reach, and
 *   (2) the "strong CPU baseline" of bench.py (all host cores, fused, un-padded), next to the reference-faithful
 *       padded NumPy restatement.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Follows /root/reference/calamity/calibration.py:
 *   fg_model   :1587-1590   v = A c            (A real, c complex in split form)
 *   data_model :1593-1605   m = g_i conj(g_j) v
 *   mse        :1608-1609   sum w |d - m|^2
 *   mse_chunked_sum_regularized :1623-1656   + (sum w m_r - P_r)^2 + (sum w m_i - P_i)^2
 *   train_step :663-668     gradient of the above w.r.t. re and im parts as independent variables, then
 *   OPTIMIZERS :17-27       Keras Adam / Adamax (epsilon outside the bias correction)
 * The adjoints are the hand-derived ones of SURVEY.md section 8 (a-6), checked against autograd in tests/test_oracle.py.
 *
 * Built twice (REAL = float, double) by oracle/Makefile into one shared library.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL double
#define SUF f64
#endif
#define CAT2(a, b) a##_##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(name, SUF)

typedef struct {
  int nants, nfreqs, ngrps, nbls;
  const long long* basis_off; /* [nbasis] element offset of each unique block in basis_data */
  const int* basis_nvec;      /* [nbasis] */
  const REAL* basis_data;     /* blocks, each (nrowblk * nfreqs, nvec) row-major */
  const int* grp_basis;       /* [ngrps] */
  const int* grp_bl_start;    /* [ngrps + 1] */
  const long long* grp_coff;  /* [ngrps + 1] offsets into the flat coefficient vectors */
  const int* bl_ant0;
  const int* bl_ant1;
  const int* bl_rowblk;
  const REAL* data_r; /* [nbls][nfreqs] */
  const REAL* data_i;
  const REAL* wgts;
} FN(ref_problem);

static int nthreads_or_default(int nthreads) {
#ifdef _OPENMP
  return nthreads > 0 ? nthreads : omp_get_max_threads();
#else
  (void)nthreads;
  return 1;
#endif
}

/* One pass over all baselines.  want_grads == 0: loss, S_r, S_i only.  alpha_*: the regulariser's 2 (S - P), applied to
 * e = -2 w r + alpha w (calibration.py:1650-1655 differentiated); pass zeros for the un-regularised loss.
 * gg_* [nants * nfreqs] and gc_* [ncoef] are overwritten. */
static void FN(pass)(const FN(ref_problem) * P, const REAL* g_r, const REAL* g_i, const REAL* c_r, const REAL* c_i,
                     int want_grads, double alpha_r, double alpha_i, double* loss, double* s_r, double* s_i, REAL* gg_r,
                     REAL* gg_i, REAL* gc_r, REAL* gc_i, int nthreads) {
  const int F = P->nfreqs;
  const int nt = nthreads_or_default(nthreads);
  const size_t ng = (size_t)P->nants * F;
  double* part = (double*)calloc((size_t)nt * 3, sizeof(double));
  REAL* ggp = NULL; /* per-thread gain-gradient buffers: the reference's scatter-add (gather backward) without atomics */
  if (want_grads) ggp = (REAL*)calloc((size_t)nt * 2 * ng, sizeof(REAL));
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int me = omp_get_thread_num();
#else
    const int me = 0;
#endif
    REAL* my_r = want_grads ? ggp + (size_t)me * 2 * ng : NULL;
    REAL* my_i = want_grads ? my_r + ng : NULL;
    REAL* gv_r = (REAL*)malloc(sizeof(REAL) * (size_t)F);
    REAL* gv_i = (REAL*)malloc(sizeof(REAL) * (size_t)F);
    double l = 0.0, sr = 0.0, si = 0.0;
#pragma omp for schedule(dynamic, 16)
    for (int g = 0; g < P->ngrps; ++g) {
      const int u = P->grp_basis[g];
      const int nvec = P->basis_nvec[u];
      const REAL* cr = c_r + P->grp_coff[g];
      const REAL* ci = c_i + P->grp_coff[g];
      REAL* gcr = want_grads ? gc_r + P->grp_coff[g] : NULL;
      REAL* gci = want_grads ? gc_i + P->grp_coff[g] : NULL;
      if (want_grads)
        for (int k = 0; k < nvec; ++k) gcr[k] = gci[k] = 0;
      for (int b = P->grp_bl_start[g]; b < P->grp_bl_start[g + 1]; ++b) {
        const REAL* A = P->basis_data + P->basis_off[u] + (size_t)P->bl_rowblk[b] * F * nvec;
        const int a0 = P->bl_ant0[b], a1 = P->bl_ant1[b];
        const REAL* d_r = P->data_r + (size_t)b * F;
        const REAL* d_i = P->data_i + (size_t)b * F;
        const REAL* w = P->wgts + (size_t)b * F;
        for (int f = 0; f < F; ++f) {
          const REAL* row = A + (size_t)f * nvec;
          REAL vr = 0, vi = 0;
#pragma omp simd reduction(+ : vr, vi)
          for (int k = 0; k < nvec; ++k) { /* fg_model :1587-1590 */
            vr += row[k] * cr[k];
            vi += row[k] * ci[k];
          }
          const REAL g0r = g_r[(size_t)a0 * F + f], g0i = g_i[(size_t)a0 * F + f];
          const REAL g1r = g_r[(size_t)a1 * F + f], g1i = g_i[(size_t)a1 * F + f];
          const REAL Gr = g0r * g1r + g0i * g1i; /* data_model :1598-1601 */
          const REAL Gi = g0i * g1r - g0r * g1i;
          const REAL mr = Gr * vr - Gi * vi;
          const REAL mi = Gi * vr + Gr * vi;
          const REAL rr = d_r[f] - mr, ri = d_i[f] - mi;
          l += (double)(w[f] * (rr * rr + ri * ri)); /* mse :1608-1609 */
          sr += (double)(w[f] * mr);
          si += (double)(w[f] * mi);
          if (want_grads) {
            const REAL er = (REAL)-2 * w[f] * rr + (REAL)alpha_r * w[f];
            const REAL ei = (REAL)-2 * w[f] * ri + (REAL)alpha_i * w[f];
            gv_r[f] = Gr * er + Gi * ei; /* gbar_v = conj(G) e */
            gv_i[f] = Gr * ei - Gi * er;
            const REAL qr = vr * er + vi * ei; /* gbar_G = conj(v) e */
            const REAL qi = vr * ei - vi * er;
            my_r[(size_t)a0 * F + f] += qr * g1r - qi * g1i; /* grad g_i += gbar_G g_j */
            my_i[(size_t)a0 * F + f] += qr * g1i + qi * g1r;
            my_r[(size_t)a1 * F + f] += qr * g0r + qi * g0i; /* grad g_j += conj(gbar_G) g_i */
            my_i[(size_t)a1 * F + f] += qr * g0i - qi * g0r;
          }
        }
        if (want_grads) {
          for (int f = 0; f < F; ++f) { /* grad c = A^T gbar_v */
            const REAL* row = A + (size_t)f * nvec;
            const REAL a = gv_r[f], bb = gv_i[f];
#pragma omp simd
            for (int k = 0; k < nvec; ++k) {
              gcr[k] += row[k] * a;
              gci[k] += row[k] * bb;
            }
          }
        }
      }
    }
    part[me * 3 + 0] = l;
    part[me * 3 + 1] = sr;
    part[me * 3 + 2] = si;
    free(gv_r);
    free(gv_i);
  }
  double l = 0, sr = 0, si = 0;
  for (int t = 0; t < nt; ++t) {
    l += part[t * 3];
    sr += part[t * 3 + 1];
    si += part[t * 3 + 2];
  }
  *loss = l;
  *s_r = sr;
  *s_i = si;
  if (want_grads) {
#pragma omp parallel for num_threads(nt)
    for (long long x = 0; x < (long long)ng; ++x) {
      REAL a = 0, b = 0;
      for (int t = 0; t < nt; ++t) {
        a += ggp[(size_t)t * 2 * ng + x];
        b += ggp[(size_t)t * 2 * ng + ng + x];
      }
      gg_r[x] = a;
      gg_i[x] = b;
    }
    free(ggp);
  }
  free(part);
}

/* loss (and gradients when gg_r != NULL).  reg != 0: the "sum" regulariser with priors (prior_r, prior_i). */
int FN(ref_loss_grads)(const FN(ref_problem) * P, const REAL* g_r, const REAL* g_i, const REAL* c_r, const REAL* c_i, int reg,
                       double prior_r, double prior_i, double* loss, REAL* gg_r, REAL* gg_i, REAL* gc_r, REAL* gc_i,
                       int nthreads) {
  double l, sr, si;
  const int want = gg_r != NULL;
  if (!reg) {
    FN(pass)(P, g_r, g_i, c_r, c_i, want, 0.0, 0.0, &l, &sr, &si, gg_r, gg_i, gc_r, gc_i, nthreads);
    *loss = l;
    return 0;
  }
  FN(pass)(P, g_r, g_i, c_r, c_i, 0, 0.0, 0.0, &l, &sr, &si, NULL, NULL, NULL, NULL, nthreads);
  *loss = l + (sr - prior_r) * (sr - prior_r) + (si - prior_i) * (si - prior_i);
  if (want) {
    double l2, a, b;
    FN(pass)(P, g_r, g_i, c_r, c_i, 1, 2.0 * (sr - prior_r), 2.0 * (si - prior_i), &l2, &a, &b, gg_r, gg_i, gc_r, gc_i, nthreads);
  }
  return 0;
}

/* Keras Adam (optimizer 0) / Adamax (1) on one real array; t is the 1-based iteration of this update */
static void FN(update)(int optimizer, REAL* p, const REAL* g, REAL* m, REAL* v, long long n, long long t, double lr, double b1,
                       double b2, double eps, int nthreads) {
  const double bc1 = 1.0 - pow(b1, (double)t);
  const double lr_t = lr * sqrt(1.0 - pow(b2, (double)t)) / bc1;
  const double lr_u = lr / bc1;
#pragma omp parallel for num_threads(nthreads_or_default(nthreads))
  for (long long x = 0; x < n; ++x) {
    const REAL gx = g[x];
    m[x] = (REAL)b1 * m[x] + (REAL)(1.0 - b1) * gx;
    if (optimizer == 0) {
      v[x] = (REAL)b2 * v[x] + (REAL)(1.0 - b2) * gx * gx;
      p[x] -= (REAL)lr_t * m[x] / ((REAL)sqrt((double)v[x]) + (REAL)eps);
    } else {
      const REAL u = (REAL)b2 * v[x];
      v[x] = u > (REAL)fabs((double)gx) ? u : (REAL)fabs((double)gx);
      p[x] -= (REAL)lr_u * m[x] / (v[x] + (REAL)eps);
    }
  }
}

/* nsteps updates; losses[k] is the loss BEFORE update k (calibration.py:699-701).  Parameters and moments
 * (mom: 8 arrays in the order m,v of g_r, g_i, c_r, c_i; zero them for a fresh optimizer) are updated in place;
 * t0 = number of updates already applied. */
int FN(ref_fit)(const FN(ref_problem) * P, REAL* g_r, REAL* g_i, REAL* c_r, REAL* c_i, long long ncoef, int reg, double prior_r,
                double prior_i, int optimizer, double lr, double b1, double b2, double eps, long long t0, int nsteps,
                int freeze_model, REAL** mom, double* losses, int nthreads) {
  const long long ng = (long long)P->nants * P->nfreqs;
  REAL* gg_r = (REAL*)malloc(sizeof(REAL) * (size_t)ng);
  REAL* gg_i = (REAL*)malloc(sizeof(REAL) * (size_t)ng);
  REAL* gc_r = (REAL*)malloc(sizeof(REAL) * (size_t)ncoef);
  REAL* gc_i = (REAL*)malloc(sizeof(REAL) * (size_t)ncoef);
  if (!gg_r || !gg_i || !gc_r || !gc_i) return -1;
  for (int s = 0; s < nsteps; ++s) {
    FN(ref_loss_grads)(P, g_r, g_i, c_r, c_i, reg, prior_r, prior_i, &losses[s], gg_r, gg_i, gc_r, gc_i, nthreads);
    const long long t = t0 + s + 1;
    FN(update)(optimizer, g_r, gg_r, mom[0], mom[1], ng, t, lr, b1, b2, eps, nthreads);
    FN(update)(optimizer, g_i, gg_i, mom[2], mom[3], ng, t, lr, b1, b2, eps, nthreads);
    if (!freeze_model) {
      FN(update)(optimizer, c_r, gc_r, mom[4], mom[5], ncoef, t, lr, b1, b2, eps, nthreads);
      FN(update)(optimizer, c_i, gc_i, mom[6], mom[7], ncoef, t, lr, b1, b2, eps, nthreads);
    }
  }
  free(gg_r);
  free(gg_i);
  free(gc_r);
  free(gc_i);
  return 0;
}

int FN(ref_max_threads)(void) { return nthreads_or_default(0); }
