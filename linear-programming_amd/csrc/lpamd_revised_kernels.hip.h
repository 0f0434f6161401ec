// Device kernels of the revised simplex on the tableau (gfx950 / CDNA4, wave64).
//
// The method keeps the constraint matrix untouched and carries a dense inverse of
// the basis instead (reference lp_revised_simplex_tableau.cc).  Every contraction of
// the reference runs over its index in ascending order with each product and sum
// rounded on its own, so a sum is never split across lanes here: one lane owns one
// (or two adjacent) output entries and walks the contracted index sequentially,
// while the lanes of a wavefront cover adjacent outputs, which makes every load
// a coalesced row segment.
//
// Data layout (one GPU):
//   Trm    row-major fp64 copy of the tableau, m rows of ldw doubles: columns
//          0..N-1 variables, N the right-hand side, N+1+k the k-th artificial
//          column of the feasibility phase (allocated only when that phase runs).
//   Binv   row-major m x ldb: row i = basis position i, column r = constraint row.
//   BinvT  its transpose, kept in step by the same update kernel, so that both
//          B^-1 * column and c_B^T * B^-1 read rows of a row-major matrix.
//   obj, cost  ldw doubles; xB, cB, pvec, mu, colbuf, scaled  ldb doubles.
#pragma once
#include "lpamd_kernels.hip.h"

namespace lpamd {

struct RevState {      // device resident, one per revised engine
  int32_t status;      // ST_* as in the tableau engine
  int32_t pad0;
  int64_t iter;        // pivots applied so far
  int64_t max_iter;    // pricing reports ST_LIMIT once iter reaches this
  int64_t e;           // entering column of the pivot in flight
  int64_t leaving;     // leaving basis position
  double ratio;
};

struct RevShape {
  int64_t m, N;        // rows; variables (right-hand side at column N)
  int64_t W;           // allocated columns: N + 1, or N + 1 + m with the artificial block
  int64_t ldw, ldb;    // row pitch of Trm / of Binv and BinvT (multiples of 64 doubles)
};

constexpr int kRevThreads = 256;
constexpr int kRevSelectThreads = 1024;

// out[j] = sum_r coef[r] * M[r * ld + j]  (r ascending, no zero drop inside the sum);
// with kCost the result is flushz(base[j] + sum), the reduced cost of column j
// (reference lp_revised_simplex_tableau.cc:236-238, :155, :280).
// One lane owns columns 2t and 2t+1 (kVec == 2) or column t (kVec == 1); ld is even
// and covers the rounded-up column count, padding entries are finite.
template <int kVec, int kUnroll, bool kCost>
__global__ __launch_bounds__(kRevThreads) void k_rev_colsum(
    const RevState* __restrict__ st, const double* __restrict__ M, int64_t ld, int64_t nrows, int64_t ncols,
    const double* __restrict__ coef, const double* __restrict__ base, double* __restrict__ out) {
  if (st->status != ST_RUNNING) return;
  const int64_t j = ((int64_t)blockIdx.x * kRevThreads + threadIdx.x) * kVec;
  if (j >= ncols) return;
  const double* __restrict__ p = M + j;
  double a0 = 0.0, a1 = 0.0;
  int64_t r = 0;
  if constexpr (kVec == 2) {
    for (; r + kUnroll <= nrows; r += kUnroll) {
      double2 v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; u++) v[u] = *reinterpret_cast<const double2*>(p + (r + u) * ld);
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
        const double c = coef[r + u];
        a0 = __dadd_rn(a0, __dmul_rn(c, v[u].x));
        a1 = __dadd_rn(a1, __dmul_rn(c, v[u].y));
      }
    }
    for (; r < nrows; r++) {
      const double2 v = *reinterpret_cast<const double2*>(p + r * ld);
      const double c = coef[r];
      a0 = __dadd_rn(a0, __dmul_rn(c, v.x));
      a1 = __dadd_rn(a1, __dmul_rn(c, v.y));
    }
  } else {
    for (; r + kUnroll <= nrows; r += kUnroll) {
      double v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; u++) v[u] = p[(r + u) * ld];
#pragma unroll
      for (int u = 0; u < kUnroll; u++) a0 = __dadd_rn(a0, __dmul_rn(coef[r + u], v[u]));
    }
    for (; r < nrows; r++) a0 = __dadd_rn(a0, __dmul_rn(coef[r], p[r * ld]));
  }
  if constexpr (kCost) {
    a0 = flushz(__dadd_rn(base[j], a0));
    if (kVec == 2 && j + 1 < ncols) a1 = flushz(__dadd_rn(base[j + 1], a1));
  }
  out[j] = a0;
  if (kVec == 2 && j + 1 < ncols) out[j + 1] = a1;
}

// kFirstNonZero == false: entering column = non-basic column (not the right-hand
//   side) with the most negative reduced cost, lowest index on ties; none left means
//   the basis is optimal (reference lp_revised_simplex_tableau.cc:240-252).
// kFirstNonZero == true: lowest non-basic column whose probe entry is non-zero, the
//   column that replaces an artificial variable still basic after the feasibility
//   phase (reference :148-163); none is ST_ERROR (the reference asserts).
// The winner's column of Trm is then copied to colbuf for the B^-1 * column product.
template <bool kFirstNonZero>
__global__ __launch_bounds__(kRevSelectThreads) void k_rev_select_entering(
    RevState* __restrict__ st, RevShape sh, int64_t ncols_active, const double* __restrict__ cost,
    const uint8_t* __restrict__ is_basic, const double* __restrict__ Trm, double* __restrict__ colbuf) {
  __shared__ double s_v[kRevSelectThreads / kWave];
  __shared__ int64_t s_i[kRevSelectThreads / kWave];
  __shared__ int64_t s_e;
  if (st->status != ST_RUNNING) return;
  const bool limit = !kFirstNonZero && st->iter >= st->max_iter;
  double bv = 0.0;
  int64_t bi = -1;
  for (int64_t j = threadIdx.x; j < ncols_active; j += kRevSelectThreads) {
    if (j == sh.N || is_basic[j]) continue;
    const double c = cost[j];
    if (kFirstNonZero) {
      if (!is_zero(c)) take_min(bv, bi, (double)j, j);
    } else {
      if ((float)c > -1e-6f) continue;   // _IsNonNegative, reference base.h:65
      take_min(bv, bi, c, j);
    }
  }
  block_arg<false, kRevSelectThreads>(bv, bi, s_v, s_i);
  if (threadIdx.x == 0) {
    s_e = -1;
    if (limit) st->status = ST_LIMIT;
    else if (bi < 0) st->status = kFirstNonZero ? ST_ERROR : ST_SOLVED;
    else { st->e = bi; s_e = bi; }
  }
  __syncthreads();
  const int64_t e = s_e;
  if (e < 0) return;
  for (int64_t r = threadIdx.x; r < sh.m; r += kRevSelectThreads) colbuf[r] = Trm[r * sh.ldw + e];
}

// Ratio test over mu = B^-1 * column (reference :282-297) and everything of the
// pivot that is a vector operation (reference :42-60): the basic solution moves
// along mu without a zero drop, the leaving row of the inverse is scaled by
// -1 / mu[leaving] into `scaled`, mu[leaving] becomes zero, the basis bookkeeping
// and the trace are written.  forced_leaving >= 0 skips the test (replacement of a
// basic artificial variable, reference :156-160).
__global__ __launch_bounds__(kRevSelectThreads) void k_rev_select_leaving(
    RevState* __restrict__ st, RevShape sh, int64_t forced_leaving, double* __restrict__ mu, double* __restrict__ xB,
    double* __restrict__ cB, const double* __restrict__ obj, int64_t* __restrict__ basis, uint8_t* __restrict__ is_basic,
    const double* __restrict__ Binv, double* __restrict__ scaled, int64_t* __restrict__ trace, int64_t trace_cap) {
  __shared__ double s_v[kRevSelectThreads / kWave];
  __shared__ int64_t s_i[kRevSelectThreads / kWave];
  __shared__ double s_q;
  __shared__ int64_t s_l;
  if (st->status != ST_RUNNING) return;
  const int64_t e = st->e, it = st->iter;
  double bv = 0.0;
  int64_t bi = -1;
  if (forced_leaving < 0) {
    for (int64_t i = threadIdx.x; i < sh.m; i += kRevSelectThreads) {
      const double mi = mu[i];
      if (!is_neg(mi)) continue;
      const double q = -(xB[i] / mi);
      if (q < DBL_MAX) take_min(bv, bi, q, i);
    }
    block_arg<false, kRevSelectThreads>(bv, bi, s_v, s_i);
  } else {
    bi = forced_leaving;
    bv = -(xB[bi] / mu[bi]);
  }
  if (threadIdx.x == 0) {
    s_l = bi;
    s_q = bv;
    if (bi < 0) st->status = ST_UNBOUNDED;
  }
  __syncthreads();
  const int64_t l = s_l;
  if (l < 0) return;
  const double ratio = s_q;
  const double s = -1.0 / mu[l];
  const double* __restrict__ rowl = Binv + l * sh.ldb;
  for (int64_t i = threadIdx.x; i < sh.m; i += kRevSelectThreads) {
    xB[i] = i == l ? ratio : __dadd_rn(xB[i], __dmul_rn(mu[i], ratio));
    scaled[i] = __dmul_rn(rowl[i], s);
  }
  __syncthreads();   // every lane has read mu[l]
  if (threadIdx.x == 0) {
    mu[l] = 0.0;
    is_basic[e] = 1;
    is_basic[basis[l]] = 0;
    basis[l] = e;
    cB[l] = obj[e];
    if (it < trace_cap) { trace[2 * it] = e; trace[2 * it + 1] = l; }
    st->leaving = l;
    st->ratio = ratio;
    st->iter = it + 1;
  }
}

// Binv = flushz(Binv' + mu (x) scaled) over every entry, where Binv' is Binv with
// its leaving row replaced by `scaled` (reference :56-61: the zero drop of
// Tableau::Add reaches all entries, the scaled row included).  blockIdx.z == 1
// applies the same update to the transposed copy: there the roles of the two
// vectors swap and the replaced line is column `leaving`.
// Lane t of block x owns columns 2 * (256 x + t) .. +1 of a chunk of rows.
template <int kUnroll>
__global__ __launch_bounds__(kRevThreads) void k_rev_update(
    const RevState* __restrict__ st, RevShape sh, double* __restrict__ Binv, double* __restrict__ BinvT,
    const double* __restrict__ mu, const double* __restrict__ scaled, int64_t rows_per_block) {
  if (st->status != ST_RUNNING) return;
  const bool tr = blockIdx.z == 1;
  const int64_t l = st->leaving;
  double* __restrict__ M = tr ? BinvT : Binv;
  const double* __restrict__ u = tr ? scaled : mu;   // indexed by matrix row
  const double* __restrict__ v = tr ? mu : scaled;   // indexed by matrix column
  const int64_t b = ((int64_t)blockIdx.x * kRevThreads + threadIdx.x) * 2;
  if (b >= sh.m) return;
  const double v0 = v[b], v1 = b + 1 < sh.m ? v[b + 1] : 0.0;
  const int64_t a_begin = (int64_t)blockIdx.y * rows_per_block;
  const int64_t a_end = a_begin + rows_per_block < sh.m ? a_begin + rows_per_block : sh.m;
  double* __restrict__ p = M + b;
  auto apply = [&](double2 x, int64_t a) {
    const double ua = u[a];
    if (!tr && a == l) { x.x = v0; x.y = v1; }
    if (tr) {
      if (b == l) x.x = ua;
      if (b + 1 == l) x.y = ua;
    }
    x.x = flushz(__dadd_rn(x.x, __dmul_rn(ua, v0)));
    x.y = flushz(__dadd_rn(x.y, __dmul_rn(ua, v1)));
    return x;
  };
  int64_t a = a_begin;
  for (; a + kUnroll <= a_end; a += kUnroll) {
    double2 x[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; k++) x[k] = *reinterpret_cast<const double2*>(p + (a + k) * sh.ldb);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < kUnroll; k++) *reinterpret_cast<double2*>(p + (a + k) * sh.ldb) = apply(x[k], a + k);
  }
  for (; a < a_end; a++) *reinterpret_cast<double2*>(p + a * sh.ldb) = apply(*reinterpret_cast<const double2*>(p + a * sh.ldb), a);
}

// ---- set-up -----------------------------------------------------------------
// Basis of slack or artificial columns: each has one non-zero entry v at row
// rows[i], so the inverse holds 1 / -v at (i, rows[i]) and the basic solution is
// b / -v (reference :212-233).  Binv and BinvT are zero on entry.
__global__ void k_rev_init_basis(RevShape sh, const int64_t* __restrict__ basis, const int64_t* __restrict__ rows,
                                 const double* __restrict__ Trm, const double* __restrict__ obj, double* __restrict__ Binv,
                                 double* __restrict__ BinvT, double* __restrict__ xB, double* __restrict__ cB) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sh.m) return;
  const int64_t bcol = basis[i], r = rows[i];
  cB[i] = obj[bcol];
  double x = 0.0;
  if (r >= 0) {
    const double v = Trm[r * sh.ldw + bcol];
    x = Trm[r * sh.ldw + sh.N] / (-v);
    const double inv = 1.0 / (-v);
    Binv[i * sh.ldb + r] = inv;
    BinvT[r * sh.ldb + i] = inv;
  }
  xB[i] = x;
}

__global__ void k_rev_gather_basis_cost(int64_t m, const int64_t* __restrict__ basis, const double* __restrict__ obj,
                                        double* __restrict__ cB) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) cB[i] = obj[basis[i]];
}

// Rows flagged in `negate` are multiplied by -1 over columns 0..N (reference :95-114).
__global__ void k_rev_negate_rows(RevShape sh, const uint8_t* __restrict__ negate, double* __restrict__ Trm) {
  const int64_t j = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.x;
  if (j > sh.N || !negate[r]) return;
  Trm[r * sh.ldw + j] = __dmul_rn(-1.0, Trm[r * sh.ldw + j]);
}

// Artificial block: column N + 1 + k carries -1 in row k (reference :127-137).
__global__ void k_rev_fill_artificials(RevShape sh, double* __restrict__ Trm) {
  const int64_t k = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.x;
  if (k < sh.m) Trm[r * sh.ldw + sh.N + 1 + k] = k == r ? -1.0 : 0.0;
}

// Row-major image of the synthetic dense LP of k_fill_synthetic, with its slack basis.
__global__ __launch_bounds__(256) void k_rev_fill_synthetic(RevShape sh, double* __restrict__ Trm, uint8_t* __restrict__ is_basic,
                                                          int64_t* __restrict__ basis, int64_t* __restrict__ rows, int64_t n,
                                                          uint64_t seed) {
  const int64_t j = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.x;
  if (j < sh.ldw) Trm[r * sh.ldw + j] = j <= sh.N ? synth_entry(seed, sh.m, n, r, j) : 0.0;
  if (r == 0 && j < sh.W) is_basic[j] = (j >= n && j < n + sh.m) ? 1 : 0;
  if (j == 0) { basis[r] = n + r; rows[r] = r; }
}

}  // namespace lpamd
