/*
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's dense-tableau simplex arithmetic, used
 * only as (a) the parity checker for the HIP engine in tests/ and smoke(),
 * and (b) the "port" CPU baseline timed by bench.py.  Nothing under
 * linear-programming_amd/ may link, load or call this file.
 *
 * The arithmetic itself lives upstream in an un-vendored git submodule
 * (github.com/QimingZheng/tableau, pin floating: reference .gitmodules:1-3),
 * so this file restates the library's observable behaviour as pinned by the
 * reference's own golden files; oracle/pin_oracle.py replays every one of
 * those goldens (12 tests/ + 15 netlib + the lp_test.cc tableau strings)
 * through this code and compares the text byte-for-byte.
 *
 * Layout: column-major fp64, T[i + j*lda], columns 0..N-1 = variables in
 * Variable::operator< order (reference base.cc:11-29), column N = right-hand
 * side ("constant" column, reference lp.cc:167), column N+1 = phase-1
 * artificial (reference lp_simplex_tableau.cc:70); ncols = N+2.
 *
 * Build: gcc -O3 -ffp-contract=off -fopenmp -shared -fPIC (see Makefile).
 * Every * + / is rounded separately, exactly as the reference's x86-64
 * baseline build does (no FMA contraction).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* reference base.h:61-66 -- predicates take float: the double is narrowed
 * first, then compared against 1e-6f. */
static const float kEps32 = 1e-6f;
static inline int is_pos(double x) { return (float)x >= kEps32; }     /* _IsPositive    base.h:64 */
static inline int is_neg(double x) { return (float)x <= -kEps32; }    /* _IsNegative    base.h:63 */
static inline int is_zero(double x) { return fabsf((float)x) < kEps32; } /* _IsZero     base.h:65 */
static inline int is_nonneg(double x) { return (float)x > -kEps32; }  /* _IsNonNegative base.h:66 */
/* reference lp_simplex_tableau.cc:5-8: _IsZeroT<double> compares in double
 * against the float constant widened to double; the tableau library drops
 * such entries after every Add. */
static inline double flushz(double v) { return fabs(v) < (double)kEps32 ? 0.0 : v; }

enum { LPO_ERROR = 0, LPO_UNBOUNDED = 1, LPO_NOSOLUTION = 2, LPO_SOLVED = 3, LPO_LIMIT = 4 };
enum { LPO_MAX_COST = 2, LPO_MAX_REDUCTION = 3 }; /* reference lp.h:74-79 */

typedef struct {
  int64_t m, ncols, lda, N;
  double *T;          /* lda * ncols */
  double *obj;        /* ncols */
  uint8_t *is_basic;  /* ncols */
  /* options */
  int32_t skip_zero_cols; /* 1: leave columns whose pivot-row entry is exactly 0 untouched */
  int32_t fast_basis;     /* 1: leaving variable from basis_of_row[] instead of the row scan */
  int64_t *basis_of_row;  /* m, only used when fast_basis */
  /* trace (may be NULL) */
  int64_t *trace_e, *trace_r, *trace_d;
  int64_t trace_cap, n_pivots;
  int64_t max_pivots;     /* <0: unlimited */
  /* timing of the rank-1 update alone, seconds, accumulated */
  double update_seconds;
} lpo_t;

static double now_s(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}

int lpo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* reference lp_simplex_tableau.cc:10-52 (TableauPivot): objective row first,
 * then T += col' (x) row' with the zero-drop applied to every sum. */
int lpo_pivot(lpo_t *p, int64_t d, int64_t e, int64_t r) {
  const int64_t m = p->m, nc = p->ncols, lda = p->lda;
  double *T = p->T;
  if (e < 0 || e >= nc || e == p->N || r < 0 || r >= m) return -1;
  const double a = T[r + e * lda];
  const double ce = p->obj[e];
  const double g = ce * (-1.0 / a);   /* lp_simplex_tableau.cc:34-35 */
  const double f = -1.0 / a;          /* lp_simplex_tableau.cc:40 */
  double *rowp = (double *)malloc(sizeof(double) * (size_t)nc);
  double *colp = (double *)malloc(sizeof(double) * (size_t)m);
  if (!rowp || !colp) return -2;
  for (int64_t j = 0; j < nc; j++) {
    const double t = T[r + j * lda];
    p->obj[j] = flushz(p->obj[j] + t * g);  /* :36 */
    rowp[j] = t * f;
  }
  p->obj[e] = 0.0;                          /* :37 */
  for (int64_t i = 0; i < m; i++) colp[i] = T[i + e * lda];
  colp[r] = 1.0 + a;                        /* :42-44 */
  const int skip = p->skip_zero_cols;
  const double t0 = now_s();
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t j = 0; j < nc; j++) {
    const double rj = rowp[j];
    if (skip && rj == 0.0) continue;
    double *col = T + j * lda;
    for (int64_t i = 0; i < m; i++) col[i] = flushz(col[i] + colp[i] * rj);  /* :45-46 */
  }
  p->update_seconds += now_s() - t0;
  if (d >= 0 && d < nc) p->is_basic[d] = 0; /* :27 */
  p->is_basic[e] = 1;                       /* :28 */
  if (p->basis_of_row) p->basis_of_row[r] = e;
  if (p->trace_e && p->n_pivots < p->trace_cap) {
    p->trace_e[p->n_pivots] = e;
    p->trace_r[p->n_pivots] = r;
    p->trace_d[p->n_pivots] = d;
  }
  p->n_pivots++;
  free(rowp);
  free(colp);
  return 0;
}

/* reference lp_simplex_tableau.cc:54-62 */
int lpo_need_init(const lpo_t *p) {
  for (int64_t i = 0; i < p->m; i++)
    if (is_neg(p->T[i + p->N * p->lda])) return 1;
  return 0;
}

/* reference lp_simplex_tableau.cc:188-198: strict '>' so the lowest column
 * index wins ties. */
int64_t lpo_price_max_cost(const lpo_t *p) {
  int64_t e = -1;
  double best = -DBL_MAX;
  for (int64_t j = 0; j < p->ncols; j++) {
    if (j == p->N) continue;
    if (p->is_basic[j]) continue;
    const double c = p->obj[j];
    if (is_pos(c) && c > best) { best = c; e = j; }
  }
  return e;
}

/* first basic column (ascending, RHS excluded) with a non-zero entry in row i:
 * the inner loop of reference lp_simplex_tableau.cc:215-226. */
static int64_t basic_var_of_row(const lpo_t *p, int64_t i) {
  if (p->fast_basis && p->basis_of_row) return p->basis_of_row[i];
  for (int64_t j = 0; j < p->ncols; j++) {
    if (j == p->N) continue;
    if (!p->is_basic[j]) continue;
    if (!is_zero(p->T[i + j * p->lda])) return j;
  }
  return -1;
}

/* reference lp_simplex_tableau.cc:207-227: strict '<' so the lowest row wins. */
void lpo_ratio_test(const lpo_t *p, int64_t e, int64_t *r_out, int64_t *d_out) {
  int64_t r = -1, d = -1;
  double best = DBL_MAX;
  const double *col = p->T + e * p->lda, *rhs = p->T + p->N * p->lda;
  for (int64_t i = 0; i < p->m; i++) {
    const double a = col[i];
    if (is_zero(a)) continue;
    if (is_nonneg(a)) continue;
    const double ratio = rhs[i] / (-a);
    if (best > ratio) {
      const int64_t dj = basic_var_of_row(p, i);
      if (dj >= 0) { best = ratio; d = dj; r = i; }
    }
  }
  *r_out = r;
  *d_out = d;
}

/* reference lp_simplex_tableau.cc:235-374 (both the Map branch at 253-331 and
 * the sequential branch at 332-364 reduce to this fold). */
void lpo_price_max_reduction(const lpo_t *p, int64_t *e_out, int64_t *r_out, int64_t *d_out) {
  const int64_t m = p->m, nc = p->ncols, lda = p->lda, N = p->N;
  const double *rhs = p->T + N * lda;
  /* basis[row] discovery, :240-251 */
  int64_t *basis = (int64_t *)malloc(sizeof(int64_t) * (size_t)m);
  if (p->fast_basis && p->basis_of_row) {
    memcpy(basis, p->basis_of_row, sizeof(int64_t) * (size_t)m);
  } else {
    for (int64_t i = 0; i < m; i++) basis[i] = -1;
    for (int64_t j = 0; j < nc; j++) {
      if (j == N || !p->is_basic[j]) continue;
      for (int64_t i = 0; i < m; i++)
        if (!is_zero(p->T[i + j * lda])) { basis[i] = j; break; }
    }
  }
  double *cmin = (double *)malloc(sizeof(double) * (size_t)nc);
  int64_t *crow = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t j = 0; j < nc; j++) {
    crow[j] = -2; /* not a candidate */
    if (j == N || p->is_basic[j] || !is_pos(p->obj[j])) continue;
    double mn = DBL_MAX;
    int64_t ri = -1;
    const double *col = p->T + j * lda;
    for (int64_t i = 0; i < m; i++) {
      const double a = col[i];
      if (is_nonneg(a)) continue;               /* :293 / :343 */
      const double ratio = rhs[i] / (-a);
      if (mn > ratio) { mn = ratio; ri = i; }   /* :297 / :347 */
    }
    cmin[j] = mn;
    crow[j] = ri;
  }
  int64_t e = -1, r = -1, d = -1;
  double best = -DBL_MAX;
  for (int64_t j = 0; j < nc; j++) {
    if (crow[j] == -2) continue;
    if (e < 0) e = j;                            /* :353 / :315-317 */
    if (crow[j] < 0) continue;
    /* the leaving variable must be defined for the candidate to count (:354) */
    if (basis[crow[j]] < 0) continue;
    const double score = p->obj[j] * cmin[j];
    if (score > best) { best = score; e = j; r = crow[j]; d = basis[r]; }
  }
  *e_out = e;
  *r_out = r;
  *d_out = d;
  free(basis);
  free(cmin);
  free(crow);
}

static int lpo_main_loop(lpo_t *p, int strategy);

/* reference lp_simplex_tableau.cc:64-171 */
int lpo_phase1(lpo_t *p, int strategy) {
  const int64_t m = p->m, nc = p->ncols, lda = p->lda, N = p->N, A = p->N + 1;
  if (!lpo_need_init(p)) return LPO_SOLVED;
  double *saved = (double *)malloc(sizeof(double) * (size_t)nc);
  memcpy(saved, p->obj, sizeof(double) * (size_t)nc);
  for (int64_t j = 0; j < nc; j++) p->obj[j] = 0.0;
  p->obj[A] = -1.0;                                  /* :79 */
  p->is_basic[A] = 0;                                /* :71 */
  for (int64_t i = 0; i < m; i++) p->T[i + A * lda] = 1.0; /* :82-86 */
  /* MinReduce fold, ties go to the later row (:88-91) */
  int64_t r = -1;
  double mv = 0.0;
  for (int64_t i = 0; i < m; i++) {
    const double b = p->T[i + N * lda];
    if (!(mv < b)) { mv = b; r = i; }
  }
  int rc = LPO_ERROR;
  if (r < 0) goto fail;
  {
    int64_t d = -1;
    for (int64_t j = 0; j < nc; j++) {               /* :94-104 */
      if (j == N) continue;
      if (is_zero(p->T[r + j * lda])) continue;
      if (p->is_basic[j]) { d = j; break; }
    }
    if (d < 0) goto fail;
    if (lpo_pivot(p, d, A, r)) goto fail;
  }
  for (int64_t i = 0; i < m; i++)                    /* :106-108 */
    if (!is_nonneg(p->T[i + N * lda])) goto fail;
  rc = lpo_main_loop(p, strategy);                   /* :110 */
  if (rc == LPO_LIMIT) { free(saved); return LPO_LIMIT; }
  if (rc != LPO_SOLVED) { rc = LPO_ERROR; goto fail; }
  if (is_neg(p->obj[N])) { free(saved); return LPO_NOSOLUTION; } /* :113-115 */
  if (p->is_basic[A]) {                              /* :117-135 */
    int pivoted = 0;
    for (int64_t i = 0; i < m && !pivoted; i++) {
      if (is_zero(p->T[i + A * lda])) continue;
      for (int64_t j = 0; j < nc; j++) {
        if (p->is_basic[j]) continue;
        if (is_zero(p->T[i + j * lda])) continue;
        if (j == N) { rc = LPO_ERROR; goto fail; }   /* reference would assert-abort */
        if (lpo_pivot(p, A, j, i)) { rc = LPO_ERROR; goto fail; }
        pivoted = 1;
        break;
      }
    }
    if (!pivoted) { rc = LPO_ERROR; goto fail; }
    p->is_basic[A] = 0;
  }
  for (int64_t i = 0; i < m; i++) p->T[i + A * lda] = 0.0;   /* :143 */
  memcpy(p->obj, saved, sizeof(double) * (size_t)nc);        /* :144 */
  p->obj[A] = 0.0;
  {
    double *acc = (double *)calloc((size_t)nc, sizeof(double));
    for (int64_t j = 0; j < nc; j++) {               /* :149-165 */
      if (!p->is_basic[j]) continue;
      if (is_zero(p->obj[j])) continue;
      int64_t rho = -1;
      double bv = (double)kEps32;                    /* MaxAbsReduce init {-1, kEpsilonF} */
      for (int64_t i = 0; i < m; i++) {
        const double v = p->T[i + j * lda];
        if (!(fabs(bv) > fabs(v))) { bv = v; rho = i; }
      }
      if (rho < 0) { free(acc); rc = LPO_ERROR; goto fail; }
      const double s = -p->obj[j] / p->T[rho + j * lda];   /* :162 */
      for (int64_t k = 0; k < nc; k++) acc[k] = flushz(acc[k] + p->T[rho + k * lda] * s);
    }
    for (int64_t k = 0; k < nc; k++) p->obj[k] = flushz(p->obj[k] + acc[k]);  /* :166 */
    free(acc);
  }
  free(saved);
  return LPO_SOLVED;
fail:
  free(saved);
  return rc;
}

/* reference lp_simplex_tableau.cc:181-377 */
static int lpo_main_loop(lpo_t *p, int strategy) {
  for (;;) {
    if (p->max_pivots >= 0 && p->n_pivots >= p->max_pivots) return LPO_LIMIT;
    int64_t e = -1, r = -1, d = -1;
    if (strategy == LPO_MAX_COST) {
      e = lpo_price_max_cost(p);
      if (e < 0) return LPO_SOLVED;
      lpo_ratio_test(p, e, &r, &d);
      if (d < 0) return LPO_UNBOUNDED;
    } else if (strategy == LPO_MAX_REDUCTION) {
      lpo_price_max_reduction(p, &e, &r, &d);
      if (e < 0) return LPO_SOLVED;
      if (d < 0) return LPO_UNBOUNDED;
    } else {
      return LPO_ERROR; /* reference: live assert at lp_simplex_tableau.cc:14 aborts */
    }
    if (lpo_pivot(p, d, e, r)) return LPO_ERROR;
  }
}

/* reference lp_simplex_tableau.cc:173-379 */
int lpo_solve(lpo_t *p, int strategy) {
  if (lpo_need_init(p)) {
    const int rc = lpo_phase1(p, strategy);
    if (rc == LPO_NOSOLUTION || rc == LPO_LIMIT) return rc;
    if (rc != LPO_SOLVED) return LPO_ERROR;
  }
  if (lpo_need_init(p)) return LPO_ERROR;            /* :179 */
  return lpo_main_loop(p, strategy);
}

/* reference lp_simplex_tableau.cc:381-399 (values of all columns; the caller
 * maps them to names and rebuilds substituted free variables). */
void lpo_column_values(const lpo_t *p, double *x) {
  for (int64_t j = 0; j < p->ncols; j++) {
    x[j] = 0.0;
    if (j == p->N || !p->is_basic[j]) continue;
    for (int64_t i = 0; i < p->m; i++) {
      const double v = p->T[i + j * p->lda];
      if (!is_zero(v)) x[j] = -p->T[i + p->N * p->lda] / v;   /* last such row wins */
    }
  }
}

/* ---- synthetic dense LP, counter-based so any shard can be regenerated ---- */
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static inline double u01(uint64_t seed, uint64_t i, uint64_t j) {
  const uint64_t h = splitmix64(splitmix64(seed ^ (i * 0xD1B54A32D192ED03ull)) ^ (j * 0x8CB92BA72F3D8DD7ull));
  return (double)(h >> 11) * 0x1.0p-53;
}
/* value of entry (i, j) of the m x (n+m+2) extended tableau; row i == m is the
 * objective row.  Structural j<n: -A_ij with A ~ U(0.1,1.1); slack n+i: -1;
 * RHS column N=n+m: b_i ~ U(1,2)*n; objective c_j ~ U(0.5,1.5). */
double lpo_synth_entry(uint64_t seed, int64_t m, int64_t n, int64_t i, int64_t j) {
  const int64_t N = n + m;
  if (i == m) return j < n ? 0.5 + u01(seed, (uint64_t)m, (uint64_t)j) : 0.0;
  if (j < n) return -(0.1 + u01(seed, (uint64_t)i, (uint64_t)j));
  if (j < N) return (j - n == i) ? -1.0 : 0.0;
  if (j == N) return (1.0 + u01(seed, (uint64_t)i, (uint64_t)N)) * (double)n;
  return 0.0;
}
void lpo_synth_fill(lpo_t *p, uint64_t seed, int64_t n) {
  const int64_t m = p->m;
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < p->ncols; j++) {
    for (int64_t i = 0; i < m; i++) p->T[i + j * p->lda] = lpo_synth_entry(seed, m, n, i, j);
    p->obj[j] = lpo_synth_entry(seed, m, n, m, j);
    p->is_basic[j] = (j >= n && j < n + m) ? 1 : 0;
  }
  if (p->basis_of_row)
    for (int64_t i = 0; i < m; i++) p->basis_of_row[i] = n + i;
}

/* ======================================================================
 * Revised simplex on the tableau (reference lp_revised_simplex_tableau.cc).
 * Dense restatement: the basis inverse is a dense m x m matrix (row i = basis
 * position i, column = constraint row), sums over the contracted index run in
 * ascending order with every product and sum rounded separately and no zero
 * drop inside the accumulation; the zero drop (flushz) applies to the result of
 * List/Tableau::Add only, i.e. to the reduced costs (:238) and to the rank-1
 * update of the inverse (:61), not to basic_feasible_solution->AddScaled (:52).
 * This is the variant that reproduces all seven revised-simplex goldens of the
 * reference byte for byte (tests/sol{15,16,17}_revised_simplex_tableau.txt,
 * datasets/netlib_revised_simplex_sol/{afiro,adlittle,kb2,israel}.txt).
 * ====================================================================== */
typedef struct {
  int64_t m, N;        /* rows; variables (RHS column index = N) */
  int64_t ncols;       /* allocated columns: N + 1 + m (artificials at N+1 .. N+m) */
  int64_t lda;
  double *T;           /* lda * ncols, column-major */
  double *obj;         /* ncols */
  uint8_t *is_basic;   /* ncols */
  int64_t *basis;      /* m: in = initial basis columns in Variable order, out = final basis */
  double *Binv;        /* m * m, row-major */
  double *xB, *cB;     /* m */
  int32_t opt_reverted;/* in/out */
  double optimum;      /* out */
  int64_t *trace_e, *trace_l;   /* entering column, leaving basis position */
  int64_t trace_cap, n_pivots, max_pivots;
} lpo_rev_t;

static void rev_init_basis(lpo_rev_t *p, int64_t ncols_active) {
  const int64_t m = p->m, lda = p->lda, N = p->N;
  (void)ncols_active;
  /* reference :212-233.  The basis matrix is a signed permutation here (slack or
   * artificial columns), so its inverse is the transpose scaled by 1/entry. */
  memset(p->Binv, 0, sizeof(double) * (size_t)m * (size_t)m);
  for (int64_t i = 0; i < m; i++) {
    const int64_t b = p->basis[i];
    p->cB[i] = p->obj[b];
    p->xB[i] = 0.0;
    for (int64_t r = 0; r < m; r++) {
      const double v = p->T[r + b * lda];
      if (!is_zero(v)) {
        p->xB[i] = p->T[r + N * lda] / (-v);             /* :223-225 */
        p->Binv[i * m + r] = 1.0 / (-v);                  /* inverse of the permutation column */
        break;
      }
    }
  }
}

/* reference :42-62 */
static void rev_pivot(lpo_rev_t *p, int64_t leaving, int64_t e, double *mu, double ratio) {
  const int64_t m = p->m;
  p->is_basic[e] = 1;
  p->is_basic[p->basis[leaving]] = 0;
  for (int64_t i = 0; i < m; i++) p->xB[i] = p->xB[i] + mu[i] * ratio;   /* AddScaled, no zero drop */
  p->xB[leaving] = ratio;
  p->basis[leaving] = e;
  p->cB[leaving] = p->obj[e];
  double *rowl = p->Binv + leaving * m;
  const double s = -1.0 / mu[leaving];
  for (int64_t k = 0; k < m; k++) rowl[k] = rowl[k] * s;
  mu[leaving] = 0.0;
  /* dense view of Tableau::Add (:61): Binv = flush(Binv + mu (x) scaled row).  The zero
   * drop reaches every entry, also rows with mu[i] == 0 and the scaled leaving row
   * itself (needed for the israel golden), so the scaled row is copied first. */
  double *scaled = (double *)malloc(sizeof(double) * (size_t)m);
  memcpy(scaled, rowl, sizeof(double) * (size_t)m);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    const double mi = mu[i];
    double *row = p->Binv + i * m;
    for (int64_t k = 0; k < m; k++) row[k] = flushz(row[k] + mi * scaled[k]);
  }
  free(scaled);
  if (p->trace_e && p->n_pivots < p->trace_cap) {
    p->trace_e[p->n_pivots] = e;
    p->trace_l[p->n_pivots] = leaving;
  }
  p->n_pivots++;
}

/* mu = Binv * T[:, col]  (Tableau::Times, :155 and :280): ascending constraint row */
static void rev_times(const lpo_rev_t *p, int64_t col, double *mu) {
  const int64_t m = p->m;
  const double *c = p->T + col * p->lda;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    double acc = 0.0;
    const double *row = p->Binv + i * m;
    for (int64_t r = 0; r < m; r++) acc = acc + row[r] * c[r];
    mu[i] = acc;
  }
}

static void rev_optimum(lpo_rev_t *p) {   /* :253-257 */
  double v = p->obj[p->N];
  for (int64_t i = 0; i < p->m; i++) v += p->obj[p->basis[i]] * p->xB[i];
  if (p->opt_reverted) v *= -1;
  p->optimum = v;
}

/* reference :235-300 */
static int rev_main_loop(lpo_rev_t *p, int64_t ncols_active) {
  const int64_t m = p->m, lda = p->lda, N = p->N;
  double *pv = (double *)malloc(sizeof(double) * (size_t)m);
  double *cost = (double *)malloc(sizeof(double) * (size_t)ncols_active);
  double *mu = (double *)malloc(sizeof(double) * (size_t)m);
  int rc = LPO_ERROR;
  for (;;) {
    if (p->max_pivots >= 0 && p->n_pivots >= p->max_pivots) { rc = LPO_LIMIT; break; }
    /* p = sum_i cB[i] * Binv[i, :]   (:236) */
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < m; r++) {
      double acc = 0.0;
      for (int64_t i = 0; i < m; i++) acc = acc + p->cB[i] * p->Binv[i * m + r];
      pv[r] = acc;
    }
    /* cost = obj + sum_r p[r] * T[r, :]   (:237-238), zero drop on the sum */
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ncols_active; j++) {
      const double *col = p->T + j * lda;
      double acc = 0.0;
      for (int64_t r = 0; r < m; r++) acc = acc + pv[r] * col[r];
      cost[j] = flushz(p->obj[j] + acc);
    }
    int64_t e = -1;
    double best = DBL_MAX;
    for (int64_t j = 0; j < ncols_active; j++) {       /* :242-251 */
      if (j == N || p->is_basic[j]) continue;
      if (is_nonneg(cost[j])) continue;
      if (best > cost[j]) { best = cost[j]; e = j; }
    }
    if (e < 0) { rev_optimum(p); rc = LPO_SOLVED; break; }
    rev_times(p, e, mu);                                 /* :280 */
    int64_t leaving = -1;
    double mr = DBL_MAX;
    for (int64_t i = 0; i < m; i++) {                    /* :284-291 */
      if (is_neg(mu[i]) && -(p->xB[i] / mu[i]) < mr) { mr = -p->xB[i] / mu[i]; leaving = i; }
    }
    if (leaving < 0) { rc = LPO_UNBOUNDED; break; }
    rev_pivot(p, leaving, e, mu, mr);
  }
  free(pv);
  free(cost);
  free(mu);
  return rc;
}

/* reference :199-302 with :92-197 inlined for the phase-1 branch */
int lpo_revised_solve(lpo_rev_t *p) {
  const int64_t m = p->m, lda = p->lda, N = p->N, W0 = N + 1;
  for (int64_t j = 0; j < W0; j++) p->obj[j] = p->obj[j] * -1.0;   /* :201 */
  p->opt_reverted = !p->opt_reverted;                              /* :202 */
  int need = 0;
  for (int64_t i = 0; i < m; i++) need |= is_neg(p->T[i + N * lda]);
  if (!need) {
    rev_init_basis(p, W0);
    return rev_main_loop(p, W0);
  }
  /* phase 1: rows with a negative right-hand side are negated (:95-114) */
  for (int64_t i = 0; i < m; i++) {
    if (!is_neg(p->T[i + N * lda])) continue;
    for (int64_t j = 0; j < W0; j++) p->T[i + j * lda] = -1 * p->T[i + j * lda];
  }
  double *raw_obj = (double *)malloc(sizeof(double) * (size_t)p->ncols);
  memcpy(raw_obj, p->obj, sizeof(double) * (size_t)p->ncols);
  const int raw_rev = p->opt_reverted;
  int64_t *saved_basis = NULL;
  (void)saved_basis;
  for (int64_t j = 0; j < p->ncols; j++) { p->obj[j] = 0.0; p->is_basic[j] = 0; }
  for (int64_t i = 0; i < m; i++) {                                /* :127-137 */
    const int64_t c = W0 + i;
    for (int64_t r = 0; r < m; r++) p->T[r + c * lda] = 0.0;
    p->T[i + c * lda] = -1.0;
    p->obj[c] = -1.0;
    p->is_basic[c] = 1;
    p->basis[i] = c;
  }
  /* nested TableauRevisedSimplexSolve (:139): negate the auxiliary objective */
  for (int64_t j = 0; j < p->ncols; j++) p->obj[j] = p->obj[j] * -1.0;
  p->opt_reverted = !p->opt_reverted;
  rev_init_basis(p, p->ncols);
  int rc = rev_main_loop(p, p->ncols);
  if (rc == LPO_LIMIT) { free(raw_obj); return rc; }
  if (rc != LPO_SOLVED) { free(raw_obj); return LPO_ERROR; }      /* :140 live assert */
  if (!is_zero(p->optimum)) { free(raw_obj); return LPO_NOSOLUTION; }   /* :142 */
  double *probing = (double *)malloc(sizeof(double) * (size_t)m);
  for (int64_t i = 0; i < m; i++) {                                /* :145-170 */
    if (p->basis[i] < N) continue;      /* only artificial columns (index >= N + 1) qualify */
    int pivoted = 0;
    for (int64_t col = 0; col < W0 && !pivoted; col++) {
      if (col == N || p->is_basic[col]) continue;
      rev_times(p, col, probing);
      if (!is_zero(probing[i])) {
        rev_pivot(p, i, col, probing, -(p->xB[i] / probing[i]));
        pivoted = 1;
      }
    }
    if (!pivoted) { free(probing); free(raw_obj); return LPO_ERROR; }   /* :165 live assert */
  }
  free(probing);
  for (int64_t c = W0; c < p->ncols; c++) {                        /* :171 */
    for (int64_t r = 0; r < m; r++) p->T[r + c * lda] = 0.0;
    p->is_basic[c] = 0;
  }
  memcpy(p->obj, raw_obj, sizeof(double) * (size_t)p->ncols);      /* :192-193 */
  p->opt_reverted = raw_rev;
  free(raw_obj);
  for (int64_t i = 0; i < m; i++) p->cB[i] = p->obj[p->basis[i]];  /* :194-195 */
  return rev_main_loop(p, W0);
}

/* ======================================================================
 * Storage-free restatement for the SYNTHETIC LP (test infrastructure).
 * The two largest benchmark configurations do not fit the host (m=65536
 * n=262144 stores 172 GB), so the dense restatement above cannot check them.
 * Every entry of the synthetic tableau is a pure function of (seed, i, j), and
 * a pivot changes entry (i, j) by the rank-1 formula of reference
 * lp_simplex_tableau.cc:45-46 only, so the entry after s pivots is
 *   flushz(... flushz(flushz(T0_ij + col'_0[i]*row'_0[j]) + col'_1[i]*row'_1[j]) ...)
 * with the col'/row' vectors of the pivots so far (m and ncols doubles each).
 * The functions below run the main loop of reference
 * lp_simplex_tableau.cc:181-377 on entries regenerated this way: same
 * operations in the same order as lpo_pivot / lpo_price_* / lpo_ratio_test
 * (a step whose row' entry for the column is exactly 0 is skipped, which is
 * what skip_zero_cols does and changes no bit), no tableau in memory.
 * tests/test_oracle_goldens.py checks it against the dense restatement.
 * ====================================================================== */
typedef struct {
  uint64_t seed;
  int64_t m, n, N, ncols, kmax, k;   /* k = pivots applied so far */
  double *colp;                       /* kmax * m */
  double *rowp;                       /* kmax * ncols */
  double *obj, *rhs;                  /* ncols, m: current objective row and right-hand side */
  uint8_t *is_basic;                  /* ncols */
  int64_t *basis;                     /* m */
  int64_t *trace_e, *trace_r;         /* kmax */
} lpo_stream_t;

void lpo_stream_free(lpo_stream_t *h) {
  if (!h) return;
  free(h->colp); free(h->rowp); free(h->obj); free(h->rhs); free(h->is_basic); free(h->basis);
  free(h->trace_e); free(h->trace_r); free(h);
}

lpo_stream_t *lpo_stream_create(uint64_t seed, int64_t m, int64_t n, int64_t kmax) {
  lpo_stream_t *h = (lpo_stream_t *)calloc(1, sizeof(lpo_stream_t));
  if (!h) return NULL;
  h->seed = seed; h->m = m; h->n = n; h->N = n + m; h->ncols = n + m + 2; h->kmax = kmax; h->k = 0;
  h->colp = (double *)malloc(sizeof(double) * (size_t)(kmax * m));
  h->rowp = (double *)malloc(sizeof(double) * (size_t)(kmax * h->ncols));
  h->obj = (double *)malloc(sizeof(double) * (size_t)h->ncols);
  h->rhs = (double *)malloc(sizeof(double) * (size_t)m);
  h->is_basic = (uint8_t *)malloc((size_t)h->ncols);
  h->basis = (int64_t *)malloc(sizeof(int64_t) * (size_t)m);
  h->trace_e = (int64_t *)malloc(sizeof(int64_t) * (size_t)kmax);
  h->trace_r = (int64_t *)malloc(sizeof(int64_t) * (size_t)kmax);
  if (!h->colp || !h->rowp || !h->obj || !h->rhs || !h->is_basic || !h->basis || !h->trace_e || !h->trace_r) {
    lpo_stream_free(h);
    return NULL;
  }
  for (int64_t j = 0; j < h->ncols; j++) {
    h->obj[j] = lpo_synth_entry(seed, m, n, m, j);
    h->is_basic[j] = (j >= n && j < n + m) ? 1 : 0;
  }
  for (int64_t i = 0; i < m; i++) {
    h->rhs[i] = lpo_synth_entry(seed, m, n, i, h->N);
    h->basis[i] = n + i;
  }
  return h;
}

/* column j as the pivots so far leave it */
void lpo_stream_column(const lpo_stream_t *h, int64_t j, double *out) {
  const int64_t m = h->m;
  for (int64_t i = 0; i < m; i++) out[i] = lpo_synth_entry(h->seed, m, h->n, i, j);
  for (int64_t s = 0; s < h->k; s++) {
    const double rj = h->rowp[s * h->ncols + j];
    if (rj == 0.0) continue;
    const double *c = h->colp + s * m;
    for (int64_t i = 0; i < m; i++) out[i] = flushz(out[i] + c[i] * rj);   /* lp_simplex_tableau.cc:45-46 */
  }
}

static double stream_entry(const lpo_stream_t *h, int64_t i, int64_t j) {
  double v = lpo_synth_entry(h->seed, h->m, h->n, i, j);
  for (int64_t s = 0; s < h->k; s++) {
    const double rj = h->rowp[s * h->ncols + j];
    if (rj != 0.0) v = flushz(v + h->colp[s * h->m + i] * rj);
  }
  return v;
}

/* At most `budget` further pivots (never beyond kmax in total). Returns LPO_LIMIT when the budget
 * or kmax ends the run, otherwise the terminal Result. */
int lpo_stream_run(lpo_stream_t *h, int strategy, int64_t budget) {
  const int64_t m = h->m, nc = h->ncols, N = h->N;
  double *col = (double *)malloc(sizeof(double) * (size_t)m);
  double *cmin = (double *)malloc(sizeof(double) * (size_t)nc);
  int64_t *crow = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
  int rc = LPO_LIMIT;
  for (int64_t done = 0; done < budget && h->k < h->kmax; done++) {
    int64_t e = -1, r = -1;
    if (strategy == LPO_MAX_COST) {
      double best = -DBL_MAX;                               /* :188-198 */
      for (int64_t j = 0; j < nc; j++) {
        if (j == N || h->is_basic[j]) continue;
        const double c = h->obj[j];
        if (is_pos(c) && c > best) { best = c; e = j; }
      }
      if (e < 0) { rc = LPO_SOLVED; break; }
      lpo_stream_column(h, e, col);
      double bestr = DBL_MAX;                               /* :207-227 */
      for (int64_t i = 0; i < m; i++) {
        const double a = col[i];
        if (is_zero(a) || is_nonneg(a)) continue;
        const double ratio = h->rhs[i] / (-a);
        if (bestr > ratio && h->basis[i] >= 0) { bestr = ratio; r = i; }
      }
      if (r < 0) { rc = LPO_UNBOUNDED; break; }
    } else if (strategy == LPO_MAX_REDUCTION) {
#pragma omp parallel
      {
        double *w = (double *)malloc(sizeof(double) * (size_t)m);
#pragma omp for schedule(dynamic, 4)
        for (int64_t j = 0; j < nc; j++) {                  /* :275-309 / :333-352 */
          crow[j] = -2;
          if (j == N || h->is_basic[j] || !is_pos(h->obj[j])) continue;
          lpo_stream_column(h, j, w);
          double mn = DBL_MAX;
          int64_t ri = -1;
          for (int64_t i = 0; i < m; i++) {
            const double a = w[i];
            if (is_nonneg(a)) continue;
            const double ratio = h->rhs[i] / (-a);
            if (mn > ratio) { mn = ratio; ri = i; }
          }
          cmin[j] = mn;
          crow[j] = ri;
        }
        free(w);
      }
      double best = -DBL_MAX;                               /* :310-325 / :353-362 */
      for (int64_t j = 0; j < nc; j++) {
        if (crow[j] == -2) continue;
        if (e < 0) e = j;
        if (crow[j] < 0 || h->basis[crow[j]] < 0) continue;
        const double score = h->obj[j] * cmin[j];
        if (score > best) { best = score; e = j; r = crow[j]; }
      }
      if (e < 0) { rc = LPO_SOLVED; break; }
      if (r < 0) { rc = LPO_UNBOUNDED; break; }
      lpo_stream_column(h, e, col);
    } else {
      rc = LPO_ERROR;
      break;
    }
    /* TableauPivot, reference lp_simplex_tableau.cc:10-52, on regenerated entries */
    const int64_t s = h->k, d = h->basis[r];
    const double a = col[r];
    const double ce = h->obj[e];
    const double g = ce * (-1.0 / a);
    const double f = -1.0 / a;
    double *rowp = h->rowp + s * nc, *colp = h->colp + s * m;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < nc; j++) {
      const double t = j == N ? h->rhs[r] : stream_entry(h, r, j);
      h->obj[j] = flushz(h->obj[j] + t * g);
      rowp[j] = t * f;
    }
    h->obj[e] = 0.0;
    for (int64_t i = 0; i < m; i++) colp[i] = col[i];
    colp[r] = 1.0 + a;
    const double rn = rowp[N];
    if (rn != 0.0)
      for (int64_t i = 0; i < m; i++) h->rhs[i] = flushz(h->rhs[i] + colp[i] * rn);
    if (d >= 0 && d < nc) h->is_basic[d] = 0;
    h->is_basic[e] = 1;
    h->basis[r] = e;
    h->trace_e[s] = e;
    h->trace_r[s] = r;
    h->k = s + 1;
  }
  free(col); free(cmin); free(crow);
  return rc;
}

int64_t lpo_stream_state(const lpo_stream_t *h, int64_t *trace_e, int64_t *trace_r, double *obj, double *rhs, uint8_t *is_basic) {
  if (trace_e) memcpy(trace_e, h->trace_e, sizeof(int64_t) * (size_t)h->k);
  if (trace_r) memcpy(trace_r, h->trace_r, sizeof(int64_t) * (size_t)h->k);
  if (obj) memcpy(obj, h->obj, sizeof(double) * (size_t)h->ncols);
  if (rhs) memcpy(rhs, h->rhs, sizeof(double) * (size_t)h->m);
  if (is_basic) memcpy(is_basic, h->is_basic, (size_t)h->ncols);
  return h->k;
}
