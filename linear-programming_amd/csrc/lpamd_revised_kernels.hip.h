// Device kernels of the revised simplex on the tableau (gfx950 / CDNA4, wave64).
//
// The method keeps the constraint matrix untouched and carries a dense inverse of
// the basis instead (reference lp_revised_simplex_tableau.cc).  Every contraction of
// the reference runs over its index in ascending order with each product and sum
// rounded on its own, so a sum is never split across lanes here: one lane owns an
// output entry and walks the contracted index sequentially.  Terms whose
// coefficient is an exact zero are left out (acc + x * 0 == acc for finite x, and an
// accumulator that starts at +0.0 never becomes -0.0), which is what makes sparse
// columns, sparse basis costs and sparse leaving rows cheap.
//
// Data layout (one GPU):
//   Trm    row-major fp64 copy of the tableau, m rows of ldw doubles: columns
//          0..N-1 variables, N the right-hand side, N+1+k the k-th artificial
//          column of the feasibility phase (allocated only when that phase runs).
//   BinvT  the basis inverse stored transposed, row-major m x ldb: row k =
//          constraint row, column i = basis position, BinvT[k][i] = B^-1[i][k].
//          B^-1 * column then streams rows of BinvT (one lane per basis position),
//          the rank-1 update touches only the rows k whose leaving-row entry is
//          non-zero, and c_B^T * B^-1 is a dot product along every row, done through
//          LDS tiles so that the global loads stay coalesced.
//   obj, cost  ldw doubles; xB, cB, pvec, mu, colbuf, scaled  ldb doubles;
//   idx_upd  ldb int32: compacted list of the rows of BinvT the pending update has to touch.
//   Unit rows.  The start basis is a signed permutation, so a row k of BinvT starts with one
//   non-zero (at position pos[k]) and keeps that shape until an update lists it; `dense`
//   (a byte per row) and idx_dense (ascending list) record the rows that have been listed.
//   The two products over the inverse then run over the dense rows only and take the single
//   term of a unit row directly, which is the same sum with its exact-zero terms left out.
#pragma once
#include "lpamd_kernels.hip.h"

namespace lpamd {

struct RevState {      // device resident, one per revised engine
  int32_t status;      // ST_* as in the tableau engine
  int32_t full_update; // 1: the next update visits every row (entries set at start-up may still need the zero drop)
  int64_t iter;        // pivots applied so far
  int64_t max_iter;    // pricing reports ST_LIMIT once iter reaches this
  int64_t e;           // entering column of the pivot in flight
  int64_t leaving;     // leaving basis position
  double ratio;
  int32_t n_upd;       // entries of idx_upd
  int32_t n_rows;      // entries of idx_rows (rows with a non-zero multiplier in the pricing product)
  int32_t n_dense;     // entries of idx_dense (rows of BinvT an update has touched)
  int32_t pad0;
};

struct RevShape {
  int64_t m, N;        // rows; variables (right-hand side at column N)
  int64_t W;           // allocated columns: N + 1, or N + 1 + m with the artificial block
  int64_t ldw, ldb;    // row pitch of Trm / of BinvT (multiples of 64 doubles)
};

constexpr int kRevThreads = 256;
constexpr int kRevSelectThreads = 1024;

// Exclusive prefix sum of one int per thread over a kRevSelectThreads block.
__device__ __forceinline__ int block_exclusive_scan(int v, int* s_w /* [kRevSelectThreads / kWave + 1] */, int& total) {
  constexpr int kWaves = kRevSelectThreads / kWave;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  int inc = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int o = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += o;
  }
  if (lane == kWave - 1) s_w[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    const int w = lane < kWaves ? s_w[lane] : 0;
    int winc = w;
#pragma unroll
    for (int off = 1; off < kWaves; off <<= 1) {
      const int o = __shfl_up(winc, off, kWave);
      if (lane >= off) winc += o;
    }
    if (lane < kWaves) s_w[lane] = winc - w;
    if (lane == kWaves - 1) s_w[kWaves] = winc;
  }
  __syncthreads();
  total = s_w[kWaves];
  const int res = s_w[wave] + inc - v;
  __syncthreads();   // s_w may be reused by the caller
  return res;
}

// out[j] = sum over rows r (ascending) of coef[r] * M[r * ld + j]; with kCost the
// result is flushz(base[j] + sum), the reduced cost of column j (reference
// lp_revised_simplex_tableau.cc:237-238; Tableau::Times, :155 and :280).
// One lane owns columns 2t and 2t+1 (kVec == 2) or column t (kVec == 1) and keeps kUnroll
// row loads in flight; ld is even and covers the rounded-up column count, padding
// entries are finite.  This is the plain form of the product; the engine runs the
// LDS-tiled k_rev_coltile by default (LPAMD_REV_TILE=0 selects this one).
template <int kVec, int kUnroll, bool kCost, bool kListed = false>
__global__ __launch_bounds__(kRevThreads) void k_rev_colsum(
    const RevState* __restrict__ st, const double* __restrict__ M, int64_t ld, int64_t nrows_all, int64_t ncols,
    const double* __restrict__ coef, const double* __restrict__ base, double* __restrict__ out,
    const int32_t* __restrict__ rows = nullptr, const int32_t* __restrict__ n_listed = nullptr,
    double* __restrict__ sel_v = nullptr, int64_t* __restrict__ sel_i = nullptr,
    const uint8_t* __restrict__ is_basic = nullptr, int64_t rhs_col = -1) {
  __shared__ double s_v[kRevThreads / kWave];
  __shared__ int64_t s_i[kRevThreads / kWave];
  if (st->status != ST_RUNNING) return;
  const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * kVec;   // 64..256 lanes per workgroup
  const bool pricing = kCost && sel_v != nullptr;   // then every lane stays for the workgroup's arg-min
  if (j >= ncols && !pricing) return;
  const double* __restrict__ p = M + (j < ncols ? j : 0);
  const int64_t nrows = j < ncols ? (kListed ? (int64_t)*n_listed : nrows_all) : 0;   // positions in the ascending row list, or rows
  auto row_at = [&](int64_t pos) -> int64_t { return kListed ? (int64_t)rows[pos] : pos; };
  double a0 = 0.0, a1 = 0.0;
  int64_t r = 0;
  for (; r + kUnroll <= nrows; r += kUnroll) {
    int64_t rr[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) rr[u] = row_at(r + u);
    if constexpr (kVec == 2) {
      double2 v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
        const double* q = p + rr[u] * ld;
        v[u].x = __builtin_nontemporal_load(q);
        v[u].y = __builtin_nontemporal_load(q + 1);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
        const double c = coef[rr[u]];
        a0 = __dadd_rn(a0, __dmul_rn(c, v[u].x));
        a1 = __dadd_rn(a1, __dmul_rn(c, v[u].y));
      }
    } else {
      double v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; u++) v[u] = __builtin_nontemporal_load(p + rr[u] * ld);
#pragma unroll
      for (int u = 0; u < kUnroll; u++) a0 = __dadd_rn(a0, __dmul_rn(coef[rr[u]], v[u]));
    }
  }
  for (; r < nrows; r++) {
    const int64_t row = row_at(r);
    const double c = coef[row];
    a0 = __dadd_rn(a0, __dmul_rn(c, p[row * ld]));
    if constexpr (kVec == 2) a1 = __dadd_rn(a1, __dmul_rn(c, p[row * ld + 1]));
  }
  if (j < ncols) {
    if constexpr (kCost) {
      a0 = flushz(__dadd_rn(base[j], a0));
      if (kVec == 2 && j + 1 < ncols) a1 = flushz(__dadd_rn(base[j + 1], a1));
    }
    out[j] = a0;
    if (kVec == 2 && j + 1 < ncols) out[j + 1] = a1;
  }
  if constexpr (kCost) {
    if (pricing) {
      // pricing partial of this workgroup's outputs (reference :240-252): most negative reduced cost
      // among the non-basic columns, lowest column on ties; k_rev_enter_and_gather folds them
      double bv = 0.0;
      int64_t bi = -1;
      if (j < ncols && j != rhs_col && !is_basic[j] && !((float)a0 > -1e-6f)) { bv = a0; bi = j; }
      if (kVec == 2 && j + 1 < ncols && j + 1 != rhs_col && !is_basic[j + 1] && !((float)a1 > -1e-6f)) take_min(bv, bi, a1, j + 1);
      wave_arg<false>(bv, bi);
      const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, n_waves = blockDim.x / kWave;
      if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int w = 1; w < n_waves; w++) take_min(bv, bi, s_v[w], s_i[w]);
        sel_v[blockIdx.x] = bv;
        sel_i[blockIdx.x] = bi;
      }
    }
  }
}

// The same sum as k_rev_colsum, out[j] = sum_r coef[r] * M[r * ld + j] (+ base and zero drop with
// kCost), staged through LDS: a workgroup owns kW adjacent outputs and sweeps the rows in chunks
// of 4096 / kW; all 256 lanes load the chunk (16 bytes each, eight loads in flight) while kW lanes
// of the first wavefront add it up in row order.  Against one-lane-per-output this puts 4 to 16
// times more loads in flight per output, which is what narrow or short matrices need; chunks
// whose coefficients are all zero are skipped.  kListed: the rows are rows[0..*n_listed), the
// ascending list of the rows with a non-zero coefficient (k_rev_list_nonzero), so that a
// coefficient vector with few, scattered non-zeros costs only its own rows.
// kUnit (with kListed): the list holds the dense rows of the transposed inverse; output j also takes the
// single term of its unit row rowof[j] (if that row is still a unit row), merged at its place in the
// ascending row order.
template <int kW, bool kCost, bool kListed, bool kUnit = false>
__global__ __launch_bounds__(kRevThreads) void k_rev_coltile(
    const RevState* __restrict__ st, const double* __restrict__ M, int64_t ld, int64_t nrows_all, int64_t ncols,
    const double* __restrict__ coef, const double* __restrict__ base, double* __restrict__ out,
    const int32_t* __restrict__ rows, const int32_t* __restrict__ n_listed,
    const int64_t* __restrict__ rowof = nullptr, const uint8_t* __restrict__ dense = nullptr,
    double* __restrict__ sel_v = nullptr, int64_t* __restrict__ sel_i = nullptr,
    const uint8_t* __restrict__ is_basic = nullptr, int64_t rhs_col = -1) {
  constexpr int kH = 4096 / kW;                    // rows per chunk
  constexpr int kRowsPerPass = 2 * kRevThreads / kW;
  constexpr int kPasses = kH / kRowsPerPass;       // = 8
  __shared__ double tile[kH * kW];
  __shared__ double s_coef[kH];
  __shared__ int32_t s_row[kUnit ? kH : 1];
  if (st->status != ST_RUNNING) return;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int64_t j0 = (int64_t)blockIdx.x * kW;
  const int lrow = tid / (kW / 2), lcol = (tid % (kW / 2)) * 2;
  const int64_t nrows = kListed ? (int64_t)*n_listed : nrows_all;   // positions in the list, or rows
  const int64_t nchunks = (nrows + kH - 1) / kH;
  auto row_at = [&](int64_t pos) -> int64_t { return kListed ? (int64_t)rows[pos] : pos; };
  auto next_active = [&](int64_t c) {
    if (kListed) return c;   // every listed row counts
    for (; c < nchunks; c++) {
      bool nz = false;
      for (int q = lane; q < kH; q += kWave) {
        const int64_t r = c * kH + q;
        nz |= r < nrows && coef[r] != 0.0;
      }
      if (__ballot(nz) != 0) break;
    }
    return c;
  };
  double2 reg[kPasses];
  double creg[(kH + kRevThreads - 1) / kRevThreads];
  int32_t rreg[(kH + kRevThreads - 1) / kRevThreads];
  // the single term of this output's unit row, and the row it belongs to (INT32_MAX: none)
  double uterm = 0.0;
  int32_t urow = INT32_MAX;
  if (kUnit && tid < kW && j0 + tid < ncols) {
    const int64_t k = rowof[j0 + tid];
    if (k >= 0 && !dense[k]) {
      urow = (int32_t)k;
      uterm = __dmul_rn(coef[k], M[k * ld + j0 + tid]);
    }
  }
  auto issue_loads = [&](int64_t c) {
#pragma unroll
    for (int q = 0; q < kPasses; q++) {
      const int64_t pos = c * kH + q * kRowsPerPass + lrow;
      if (pos < nrows && j0 + lcol < ld) {
        const double* src = M + row_at(pos) * ld + j0 + lcol;
        reg[q].x = __builtin_nontemporal_load(src);
        reg[q].y = __builtin_nontemporal_load(src + 1);
      } else {
        reg[q] = make_double2(0.0, 0.0);
      }
    }
#pragma unroll
    for (int q = 0; q * kRevThreads < kH; q++) {
      const int64_t pos = c * kH + q * kRevThreads + tid;
      const bool in = q * kRevThreads + tid < kH && pos < nrows;
      creg[q] = in ? coef[row_at(pos)] : 0.0;
      rreg[q] = in ? (int32_t)row_at(pos) : INT32_MAX;
    }
  };
  double acc = 0.0;
  int64_t c = next_active(0);
  if (c < nchunks) issue_loads(c);
  while (c < nchunks) {
    __syncthreads();   // the previous tile has been consumed
#pragma unroll
    for (int q = 0; q < kPasses; q++)
      *reinterpret_cast<double2*>(tile + (q * kRowsPerPass + lrow) * kW + lcol) = reg[q];
#pragma unroll
    for (int q = 0; q * kRevThreads < kH; q++)
      if (q * kRevThreads + tid < kH) {
        s_coef[q * kRevThreads + tid] = creg[q];
        if (kUnit) s_row[q * kRevThreads + tid] = rreg[q];
      }
    __syncthreads();
    const int64_t cn = next_active(c + 1);
    if (cn < nchunks) issue_loads(cn);
    if (tid < kW) {
      if (kUnit) {
#pragma unroll 8
        for (int rr = 0; rr < kH; rr++) {
          if (s_row[rr] > urow) {            // the unit row sorts before this dense row: its term goes in first
            acc = __dadd_rn(acc, uterm);
            urow = INT32_MAX;
          }
          acc = __dadd_rn(acc, __dmul_rn(s_coef[rr], tile[rr * kW + tid]));
        }
      } else {
#pragma unroll 16
        for (int rr = 0; rr < kH; rr++) acc = __dadd_rn(acc, __dmul_rn(s_coef[rr], tile[rr * kW + tid]));
      }
    }
    c = cn;
  }
  if (kUnit && urow != INT32_MAX) acc = __dadd_rn(acc, uterm);   // the unit row sorts after every dense row
  const int64_t j = j0 + tid;
  double res = 0.0;
  if (tid < kW && j < ncols) {
    res = kCost ? flushz(__dadd_rn(base[j], acc)) : acc;
    out[j] = res;
  }
  if (kCost && sel_v != nullptr && tid < kWave) {
    // pricing partial of this workgroup's outputs (reference :240-252): most negative reduced cost
    // among the non-basic columns, lowest column on ties; k_rev_enter_and_gather folds them
    double bv = 0.0;
    int64_t bi = -1;
    if (tid < kW && j < ncols && j != rhs_col && !is_basic[j] && !((float)res > -1e-6f)) { bv = res; bi = j; }
    wave_arg<false>(bv, bi);
    if (tid == 0) { sel_v[blockIdx.x] = bv; sel_i[blockIdx.x] = bi; }
  }
  if (!kCost && sel_v != nullptr && tid < kWave) {
    // The product is mu = B^-1 * column and `base` carries the basic solution: ratio-test partial of
    // this workgroup's positions (reference :282-297), smallest -x_B / mu over mu <= -1e-6f, lowest
    // position on ties; k_rev_select_leaving folds the partials instead of scanning mu again.
    double bv = 0.0;
    int64_t bi = -1;
    if (tid < kW && j < ncols && is_neg(res)) {
      const double ratio = -(base[j] / res);
      if (ratio < DBL_MAX) { bv = ratio; bi = j; }
    }
    wave_arg<false>(bv, bi);
    if (tid == 0) { sel_v[blockIdx.x] = bv; sel_i[blockIdx.x] = bi; }
  }
}

// out[r] = sum_i coef[i] * M[r * ld + i], i ascending: a dot product along every row
// (p = c_B^T * B^-1 on the transposed inverse, reference :236).  A workgroup owns kR
// rows and sweeps the columns in chunks of kC = 4096 / kR: all 256 lanes load the
// kR x kC tile with coalesced 16-byte accesses into registers, park it in LDS
// (odd pitch, conflict-free for the transposed read), and kR lanes of the first
// wavefront add their row of the tile in order while the loads of the next chunk
// are already in flight (two chunks are kept in flight).  Chunks whose coefficients are
// all zero are skipped (their list is tabulated up front, so no test sits between two
// chunks), and inside a chunk the adding lanes walk only the positions with a non-zero coefficient
// (compacted in LDS, ascending): the basis costs are non-zero only at the positions that
// hold a structural variable, so the sequential chain of a row has as many steps as
// such positions, not m.
// kListed: the workgroup's rows are rows[blockIdx.x * kR ..) of the list of *n_listed dense rows; the
// other (unit) rows get their single term from k_rev_unit_rows_p.
template <int kR, bool kListed>
__global__ __launch_bounds__(kRevThreads) void k_rev_rowdot(
    const RevState* __restrict__ st, const double* __restrict__ M, int64_t ld, int64_t nrows_all, int64_t ncols,
    const double* __restrict__ coef, double* __restrict__ out, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ n_listed) {
  constexpr int kC = 4096 / kR;
  constexpr int kCWaves = kC / kWave;
  constexpr int kWaves = kRevThreads / kWave;
  constexpr int kPitch = kC + 1;
  constexpr int kRowsPerPass = 2 * kRevThreads / kC;
  constexpr int kPasses = kR / kRowsPerPass;
  constexpr int kSeg = 2048;            // chunks whose activity is tabulated at a time
  static_assert(kC >= kWave && kC <= kRevThreads, "one coefficient per lane");
  __shared__ double tile[kR * kPitch];
  __shared__ double s_cval[kC];
  __shared__ int32_t s_cidx[kC];
  __shared__ int32_t s_wcnt[kCWaves];
  __shared__ uint8_t s_flag[kSeg];
  __shared__ int32_t s_list[kSeg];      // ascending list of the segment's chunks with a non-zero coefficient
  __shared__ int32_t s_nlist;
  if (st->status != ST_RUNNING) return;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int64_t r0 = (int64_t)blockIdx.x * kR;
  const int64_t nrows = kListed ? (int64_t)*n_listed : nrows_all;   // positions in the list, or rows
  if (r0 >= nrows) return;
  auto row_at = [&](int64_t pos) -> int64_t { return kListed ? (int64_t)rows[pos] : pos; };
  const int lrow = tid / (kC / 2), lcol = (tid % (kC / 2)) * 2;
  const int64_t nchunks = (ncols + kC - 1) / kC;
  double2 ra[kPasses], rb[kPasses];
  double ca = 0.0, cb = 0.0;
  int64_t grow[kPasses];   // matrix rows of this lane's tile rows (-1: past the end)
#pragma unroll
  for (int q = 0; q < kPasses; q++) {
    const int64_t pos = r0 + q * kRowsPerPass + lrow;
    grow[q] = pos < nrows ? row_at(pos) : -1;
  }
  auto issue = [&](double2 (&reg)[kPasses], double& cr, int64_t c) {
    const int64_t col = c * kC + lcol;
#pragma unroll
    for (int q = 0; q < kPasses; q++)
      reg[q] = (grow[q] >= 0 && col < ld) ? *reinterpret_cast<const double2*>(M + grow[q] * ld + col) : make_double2(0.0, 0.0);
    if (tid < kC) cr = c * kC + tid < ncols ? coef[c * kC + tid] : 0.0;
  };
  double acc = 0.0;
  // parks a chunk in LDS, compacts its non-zero coefficients in order and adds the listed terms
  auto consume = [&](const double2 (&reg)[kPasses], double cr, auto&& refill) {
    __syncthreads();   // the previous tile and its coefficient list have been consumed
#pragma unroll
    for (int q = 0; q < kPasses; q++) {
      double* dst = tile + (q * kRowsPerPass + lrow) * kPitch + lcol;
      dst[0] = reg[q].x;
      dst[1] = reg[q].y;
    }
    const bool nz = tid < kC && cr != 0.0;
    const unsigned long long mask = __ballot(nz);
    if (tid < kC && lane == 0) s_wcnt[wave] = __popcll(mask);
    __syncthreads();
    int before = 0, n_terms = 0;
#pragma unroll
    for (int w = 0; w < kCWaves; w++) {
      const int n = s_wcnt[w];
      if (w < wave) before += n;
      n_terms += n;
    }
    if (nz) {
      const int at = before + __popcll(mask & ((1ull << lane) - 1ull));
      s_cidx[at] = tid;
      s_cval[at] = cr;
    }
    __syncthreads();
    refill();          // the registers are free again: the loads of the chunk after next go out before the adds
    if (tid < kR) {
      const double* row = tile + tid * kPitch;
#pragma unroll 4
      for (int b = 0; b < n_terms; b++) acc = __dadd_rn(acc, __dmul_rn(s_cval[b], row[s_cidx[b]]));
    }
  };
  for (int64_t seg0 = 0; seg0 < nchunks; seg0 += kSeg) {
    const int nseg = (int)(nchunks - seg0 < kSeg ? nchunks - seg0 : kSeg);
    // which chunks of the segment hold a non-zero coefficient: a wavefront tests four chunks per round trip
    __syncthreads();   // the list of the previous segment has been consumed
    for (int ch0 = wave * 4; ch0 < nseg; ch0 += kWaves * 4) {
      bool nzc[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        nzc[u] = false;
#pragma unroll
        for (int q = 0; q < kCWaves; q++) {
          const int64_t i = (seg0 + ch0 + u) * kC + q * kWave + lane;
          nzc[u] |= ch0 + u < nseg && i < ncols && coef[i] != 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool any = __ballot(nzc[u]) != 0;
        if (lane == 0 && ch0 + u < nseg) s_flag[ch0 + u] = any ? 1 : 0;
      }
    }
    __syncthreads();
    if (wave == 0) {   // ordered compaction of the flags
      int n = 0;
      for (int b = 0; b < nseg; b += kWave) {
        const bool f = b + lane < nseg && s_flag[b + lane] != 0;
        const unsigned long long mask = __ballot(f);
        if (f) s_list[n + __popcll(mask & ((1ull << lane) - 1ull))] = b + lane;
        n += __popcll(mask);
      }
      if (lane == 0) s_nlist = n;
    }
    __syncthreads();
    const int nl = s_nlist;
    // two chunks in flight: while one is parked in LDS and added up, the loads of the next are under way
    if (nl > 0) issue(ra, ca, seg0 + s_list[0]);
    if (nl > 1) issue(rb, cb, seg0 + s_list[1]);
    for (int k = 0; k < nl; k += 2) {
      consume(ra, ca, [&]() { if (k + 2 < nl) issue(ra, ca, seg0 + s_list[k + 2]); });
      if (k + 1 >= nl) break;
      consume(rb, cb, [&]() { if (k + 3 < nl) issue(rb, cb, seg0 + s_list[k + 3]); });
    }
  }
  if (tid < kR && r0 + tid < nrows) out[row_at(r0 + tid)] = acc;
}

__device__ __forceinline__ double read_lane_f64(double x, int src /* wave-uniform */) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}

// The same dot product with one workgroup per row, for a short list of dense rows (a few
// thousand at most): the LDS-tiled kernel above puts 16 rows on a workgroup and leaves most of
// the chip idle then.  The four wavefronts stream the row together, 2 * 64 * kU columns each per
// round, with coalesced 16-byte loads; each leaves the products of its non-zero coefficients in
// LDS, compacted in column order, and the first wavefront adds the four lists one after the
// other, so the sum still runs in index order (64 products per LDS read, handed from lane to
// lane with read-lane; the accumulator lives redundantly in all its lanes).
template <bool kListed, int kU>
__global__ __launch_bounds__(kRevThreads) void k_rev_rowdot_row(
    const RevState* __restrict__ st, const double* __restrict__ M, int64_t ld, int64_t nrows_all, int64_t ncols,
    const double* __restrict__ coef, double* __restrict__ out, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ n_listed) {
  constexpr int kWaves = kRevThreads / kWave;
  constexpr int kCols = 2 * kWave * kU;            // columns of one wavefront per round
  __shared__ double s_prod[kWaves][kCols];
  __shared__ int s_n[kWaves];
  if (st->status != ST_RUNNING) return;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const unsigned long long below = (1ull << lane) - 1ull;
  const int64_t nrows = kListed ? (int64_t)*n_listed : nrows_all;
  for (int64_t pos = blockIdx.x; pos < nrows; pos += gridDim.x) {
    const int64_t r = kListed ? (int64_t)rows[pos] : pos;
    const double* __restrict__ row = M + r * ld;
    double acc = 0.0;
    for (int64_t base = 0; base < ncols; base += (int64_t)kWaves * kCols) {
      double2 v[kU], c[kU];
#pragma unroll
      for (int u = 0; u < kU; u++) {
        const int64_t col = base + (int64_t)wave * kCols + (int64_t)u * 2 * kWave + 2 * lane;   // ld is even and covers ncols rounded up
        v[u] = col < ld ? *reinterpret_cast<const double2*>(row + col) : make_double2(0.0, 0.0);
        c[u].x = col < ncols ? coef[col] : 0.0;
        c[u].y = col + 1 < ncols ? coef[col + 1] : 0.0;
      }
      int n = 0;
#pragma unroll
      for (int u = 0; u < kU; u++) {
        const bool nx = c[u].x != 0.0, ny = c[u].y != 0.0;
        const unsigned long long mx = __ballot(nx), my = __ballot(ny);
        const int at = n + __popcll(mx & below) + __popcll(my & below);
        if (nx) s_prod[wave][at] = __dmul_rn(c[u].x, v[u].x);
        if (ny) s_prod[wave][at + (nx ? 1 : 0)] = __dmul_rn(c[u].y, v[u].y);
        n += __popcll(mx) + __popcll(my);
      }
      if (lane == 0) s_n[wave] = n;
      __syncthreads();
      if (wave == 0) {
        for (int w = 0; w < kWaves; w++) {
          const int cnt = s_n[w];
          for (int k0 = 0; k0 < cnt; k0 += kWave) {
            const double p = k0 + lane < cnt ? s_prod[w][k0 + lane] : 0.0;
            const int lim = cnt - k0 < kWave ? cnt - k0 : kWave;
            for (int t = 0; t < lim; t++) acc = __dadd_rn(acc, read_lane_f64(p, t));
          }
        }
      }
      __syncthreads();   // the lists are free again
    }
    if (threadIdx.x == 0) out[r] = acc;
  }
}

// Flagged elements of this 256-thread workgroup -> counts[blockIdx.x] (first half of the chip-wide
// ordered compaction, see compact_flag_wide).
__device__ __forceinline__ void count_flag_wide(bool flag, int32_t* __restrict__ counts, int* s_w /* [kRevThreads / kWave] */) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int n = __popcll(__ballot(flag));
  if (lane == 0) s_w[wave] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < kRevThreads / kWave; w++) t += s_w[w];
    counts[blockIdx.x] = t;
  }
  __syncthreads();
}

// Second half: element i of this workgroup (one per thread, blocks of 256 in index order) is
// appended to idx at its rank among all flagged elements; counts[] holds the per-workgroup
// totals of an earlier kernel over the same flags.  The last workgroup writes the total.
__device__ __forceinline__ void compact_flag_wide(bool flag, int64_t i, const int32_t* __restrict__ counts,
                                                  int32_t* __restrict__ idx, int32_t* __restrict__ total_out,
                                                  int* s_w /* [2 * kRevThreads / kWave] */) {
  constexpr int kWaves = kRevThreads / kWave;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  int before = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kRevThreads) before += counts[b];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) before += __shfl_xor(before, off, kWave);
  const unsigned long long mask = __ballot(flag);
  if (lane == 0) { s_w[wave] = before; s_w[kWaves + wave] = __popcll(mask); }
  __syncthreads();
  int base = 0, here = 0;
#pragma unroll
  for (int w = 0; w < kWaves; w++) {
    base += s_w[w];
    if (w < wave) base += s_w[kWaves + w];
    here += s_w[kWaves + w];
  }
  if (flag) idx[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)i;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    int blocks_before = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) blocks_before += s_w[w];
    *total_out = blocks_before + here;
  }
  __syncthreads();   // s_w may be reused by the caller
}

// p[r] for the rows of the transposed inverse that are still unit rows: the sum over the row has one
// term that can be non-zero, c_B[pos[r]] * BinvT[r][pos[r]], and starts from +0.0.  With `counts` the
// kernel also leaves, per 256 rows, how many entries of p are non-zero (dense rows were written by
// k_rev_rowdot before), the first half of listing the rows the pricing product has to read.
__global__ __launch_bounds__(kRevThreads) void k_rev_unit_rows_p(
    const RevState* __restrict__ st, const double* __restrict__ M, int64_t ld, int64_t nrows, const double* __restrict__ coef,
    const int64_t* __restrict__ pos, const uint8_t* __restrict__ dense, double* __restrict__ out, int32_t* __restrict__ counts) {
  __shared__ int s_w[kRevThreads / kWave];
  if (st->status != ST_RUNNING) return;
  const int64_t r = (int64_t)blockIdx.x * kRevThreads + threadIdx.x;
  double v = 0.0;
  if (r < nrows) {
    if (dense[r]) {
      v = out[r];
    } else {
      const int64_t i = pos[r];
      v = i >= 0 ? __dadd_rn(0.0, __dmul_rn(coef[i], M[r * ld + i])) : 0.0;
      out[r] = v;
    }
  }
  if (counts) count_flag_wide(v != 0.0, counts, s_w);
}

// Ascending list of the indices i < n with v[i] != 0 over the whole chip, given the per-256 counts
// (same job as k_rev_list_nonzero, which does it in one workgroup).
__global__ __launch_bounds__(kRevThreads) void k_rev_list_nonzero_wide(
    const RevState* __restrict__ st, const double* __restrict__ v, int64_t n, const int32_t* __restrict__ counts,
    int32_t* __restrict__ idx, int32_t* __restrict__ count) {
  __shared__ int s_w[2 * kRevThreads / kWave];
  if (st->status != ST_RUNNING) return;
  const int64_t i = (int64_t)blockIdx.x * kRevThreads + threadIdx.x;
  compact_flag_wide(i < n && v[i] != 0.0, i, counts, idx, count, s_w);
}

// colbuf[r] = M[r * ld + col] for rows r = threadIdx.x, threadIdx.x + blockDim.x, ...; the strided
// loads go out eight at a time: a single workgroup hides their latency only through loads in flight.
__device__ __forceinline__ void gather_column(const double* __restrict__ M, int64_t ld, int64_t col, int64_t nrows,
                                              double* __restrict__ colbuf) {
  for (int64_t rb = threadIdx.x; rb < nrows; rb += 8 * (int64_t)blockDim.x) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int64_t r = rb + q * (int64_t)blockDim.x;
      v[q] = r < nrows ? M[r * ld + col] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int64_t r = rb + q * (int64_t)blockDim.x;
      if (r < nrows) colbuf[r] = v[q];
    }
  }
}

// kFirstNonZero == false: entering column = non-basic column (not the right-hand
//   side) with the most negative reduced cost, lowest index on ties; none left means
//   the basis is optimal (reference lp_revised_simplex_tableau.cc:240-252).
// kFirstNonZero == true: lowest non-basic column whose probe entry is non-zero, the
//   column that replaces an artificial variable still basic after the feasibility
//   phase (reference :148-163); none is ST_ERROR (the reference asserts).
// With kGather the winner's column of Trm is then copied to colbuf for the B^-1 * column
// product; large problems leave that to k_rev_gather_wide, whose strided loads (one page
// per row) are spread over the whole chip.
template <bool kFirstNonZero, bool kGather>
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
  // a single workgroup hides the load latency only through loads in flight: eight per lane
  for (int64_t jb = threadIdx.x; jb < ncols_active; jb += 8 * kRevSelectThreads) {
    double c[8];
    uint8_t basic[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int64_t j = jb + q * kRevSelectThreads;
      c[q] = j < ncols_active ? cost[j] : 0.0;
      basic[q] = j < ncols_active ? is_basic[j] : 1;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int64_t j = jb + q * kRevSelectThreads;
      if (j >= ncols_active || j == sh.N || basic[q]) continue;
      if (kFirstNonZero) {
        if (!is_zero(c[q])) take_min(bv, bi, (double)j, j);
      } else {
        if ((float)c[q] > -1e-6f) continue;   // _IsNonNegative, reference base.h:65
        take_min(bv, bi, c[q], j);
      }
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
  if (!kGather || e < 0) return;
  gather_column(Trm, sh.ldw, e, sh.m, colbuf);
}

// Pricing from the partials the cost product left (one per workgroup), same outcome as
// k_rev_select_entering<false, false> without a second pass over the costs, and the chip-wide fetch
// of the entering column, in one launch: every workgroup folds the partials itself (a few hundred
// values from L2), so all of them know the entering column without a kernel boundary; the first
// one records the outcome.
__global__ __launch_bounds__(kRevThreads) void k_rev_enter_and_gather(
    RevState* __restrict__ st, const double* __restrict__ sel_v, const int64_t* __restrict__ sel_i, int n_parts,
    const double* __restrict__ M, int64_t ld, int64_t nrows, double* __restrict__ out) {
  __shared__ double s_v[kRevThreads / kWave];
  __shared__ int64_t s_i[kRevThreads / kWave];
  if (st->status != ST_RUNNING) return;
  const bool limit = st->iter >= st->max_iter;
  double bv;
  int64_t bi;
  fold_partials<false, kRevThreads>(sel_v, sel_i, n_parts, bv, bi, s_v, s_i);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (limit) st->status = ST_LIMIT;
    else if (bi < 0) st->status = ST_SOLVED;
    else st->e = bi;
  }
  if (limit || bi < 0) return;
  const int64_t r = (int64_t)blockIdx.x * kRevThreads + threadIdx.x;
  if (r < nrows) out[r] = M[r * ld + bi];
}

// out[r] = M[r * ld + col] with one row per lane over the whole grid; col < 0 reads the
// entering column chosen by k_rev_select_entering.
__global__ __launch_bounds__(kRevThreads) void k_rev_gather_wide(
    const RevState* __restrict__ st, const double* __restrict__ M, int64_t ld, int64_t nrows, int64_t col, double* __restrict__ out) {
  if (st->status != ST_RUNNING) return;
  const int64_t r = (int64_t)blockIdx.x * kRevThreads + threadIdx.x;
  if (col < 0) col = st->e;
  if (r < nrows) out[r] = M[r * ld + col];
}

// Ascending list of the indices i < n with bytes[i] != 0; one workgroup, every thread a
// contiguous run, loads eight at a time.
__device__ __forceinline__ void list_flagged(const uint8_t* __restrict__ bytes, int64_t n, int32_t* __restrict__ idx, int* s_w,
                                             int32_t* __restrict__ count) {
  const int64_t seg = (n + kRevSelectThreads - 1) / kRevSelectThreads;
  const int64_t begin = threadIdx.x * seg, end = begin + seg < n ? begin + seg : n;
  int hits = 0;
  for (int64_t b = begin; b < end; b += 8) {
    bool f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) f[q] = b + q < end && bytes[b + q] != 0;
#pragma unroll
    for (int q = 0; q < 8; q++) hits += f[q];
  }
  int total;
  int at = block_exclusive_scan(hits, s_w, total);
  for (int64_t i = begin; i < end; i++)
    if (bytes[i] != 0) idx[at++] = (int32_t)i;
  if (threadIdx.x == 0) *count = total;
}

// Ascending list of the indices i < n with v[i] != 0 (the rows a sparse coefficient vector makes a
// column product read); one workgroup, every thread a contiguous run, loads eight at a time.
__global__ __launch_bounds__(kRevSelectThreads) void k_rev_list_nonzero(
    const RevState* __restrict__ st, const double* __restrict__ v, int64_t n, int32_t* __restrict__ idx, int32_t* __restrict__ count) {
  __shared__ int s_w[kRevSelectThreads / kWave + 1];
  if (st->status != ST_RUNNING) return;
  const int64_t seg = (n + kRevSelectThreads - 1) / kRevSelectThreads;
  const int64_t begin = threadIdx.x * seg, end = begin + seg < n ? begin + seg : n;
  int hits = 0;
  for (int64_t b = begin; b < end; b += 8) {
    double x[8];
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = b + q < end ? v[b + q] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) hits += x[q] != 0.0;
  }
  int total;
  int at = block_exclusive_scan(hits, s_w, total);
  for (int64_t i = begin; i < end; i++)
    if (v[i] != 0.0) idx[at++] = (int32_t)i;
  if (threadIdx.x == 0) *count = total;
}

// Ratio test over mu = B^-1 * column (reference :282-297) and everything of the
// pivot that is a vector operation (reference :42-60): the basic solution moves
// along mu without a zero drop, the leaving row of the inverse (column `leaving`
// of BinvT) is scaled by -1 / mu[leaving] into `scaled`, mu[leaving] becomes zero,
// the basis bookkeeping and the trace are written, and the rows of BinvT the rank-1
// update has to visit are listed in idx_upd: those whose leaving-row entry is
// non-zero before or after the scaling (all rows while full_update is set).
// forced_leaving >= 0 skips the test (replacement of a basic artificial variable,
// reference :156-160).
// Without kVectors the kernel stops after the test (leaving position and ratio in the
// state); k_rev_pivot_vectors and k_rev_finish_pivot then do the rest over the whole chip.
template <bool kVectors>
__global__ __launch_bounds__(kRevSelectThreads) void k_rev_select_leaving(
    RevState* __restrict__ st, RevShape sh, int64_t forced_leaving, double* __restrict__ mu, double* __restrict__ xB,
    double* __restrict__ cB, const double* __restrict__ obj, int64_t* __restrict__ basis, uint8_t* __restrict__ is_basic,
    const double* __restrict__ BinvT, double* __restrict__ scaled, int32_t* __restrict__ idx_upd, int64_t* __restrict__ trace,
    int64_t trace_cap, uint8_t* __restrict__ dense, int32_t* __restrict__ idx_dense,
    const double* __restrict__ part_v = nullptr, const int64_t* __restrict__ part_i = nullptr, int n_parts = 0) {
  __shared__ double s_v[kRevSelectThreads / kWave];
  __shared__ int64_t s_i[kRevSelectThreads / kWave];
  __shared__ int s_w[kRevSelectThreads / kWave + 1];
  __shared__ double s_q;
  __shared__ int64_t s_l;
  if (st->status != ST_RUNNING) return;
  const int64_t e = st->e, it = st->iter;
  const bool full = st->full_update != 0;
  double bv = 0.0;
  int64_t bi = -1;
  if (forced_leaving < 0 && n_parts > 0) {   // the B^-1 * column product left ratio-test partials
    for (int k = threadIdx.x; k < n_parts; k += kRevSelectThreads) take_min(bv, bi, part_v[k], part_i[k]);
    block_arg<false, kRevSelectThreads>(bv, bi, s_v, s_i);
  } else if (forced_leaving < 0) {
    for (int64_t ib = threadIdx.x; ib < sh.m; ib += 8 * kRevSelectThreads) {
      double mi[8], xi[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int64_t i = ib + q * kRevSelectThreads;
        mi[q] = i < sh.m ? mu[i] : 0.0;
        xi[q] = i < sh.m ? xB[i] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (!is_neg(mi[q])) continue;
        const double ratio_q = -(xi[q] / mi[q]);
        if (ratio_q < DBL_MAX) take_min(bv, bi, ratio_q, ib + q * kRevSelectThreads);
      }
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
    else if (!kVectors) { st->leaving = bi; st->ratio = bv; }
  }
  if (!kVectors) return;
  __syncthreads();
  const int64_t l = s_l;
  if (l < 0) return;
  const double ratio = s_q;
  const double s = -1.0 / mu[l];
  const int64_t seg = (sh.m + kRevSelectThreads - 1) / kRevSelectThreads;
  const int64_t k_begin = threadIdx.x * seg, k_end = k_begin + seg < sh.m ? k_begin + seg : sh.m;
  int touched = 0;
  uint64_t mask = 0;   // bit (k - k_begin) for the first 64 rows of the run; longer runs re-read
  for (int64_t kb = k_begin; kb < k_end; kb += 8) {
    double old[8], xk[8], mk[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {   // strided: eight in flight
      const bool in = kb + q < k_end;
      old[q] = in ? BinvT[(kb + q) * sh.ldb + l] : 0.0;
      xk[q] = in ? xB[kb + q] : 0.0;
      mk[q] = in ? mu[kb + q] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int64_t k = kb + q;
      if (k < k_end) {
        xB[k] = k == l ? ratio : __dadd_rn(xk[q], __dmul_rn(mk[q], ratio));
        const double sc = __dmul_rn(old[q], s);
        scaled[k] = sc;
        const bool changes = old[q] != 0.0 || sc != 0.0;   // the update makes this row dense
        if (changes) dense[k] = 1;
        const bool hit = full || changes;
        touched += hit;
        if (hit && k - k_begin < 64) mask |= 1ull << (k - k_begin);
      }
    }
  }
  int total;
  int at = block_exclusive_scan(touched, s_w, total);   // also orders the reads of mu[l] above before the write below
  for (int64_t k = k_begin; k < k_end; k++) {
    const bool hit = k - k_begin < 64 ? ((mask >> (k - k_begin)) & 1) != 0
                                      : (full || BinvT[k * sh.ldb + l] != 0.0 || scaled[k] != 0.0);
    if (hit) idx_upd[at++] = (int32_t)k;
  }
  __syncthreads();
  list_flagged(dense, sh.m, idx_dense, s_w, &st->n_dense);
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
    st->n_upd = total;
    st->full_update = 0;
  }
}

// Vector part of the pivot for the position and ratio left by k_rev_select_leaving<false>,
// one row per lane over the whole grid: basic solution, scaled leaving row, and a byte per
// row of BinvT telling whether the update has to visit it.
__global__ __launch_bounds__(kRevThreads) void k_rev_pivot_vectors(
    const RevState* __restrict__ st, RevShape sh, const double* __restrict__ mu, double* __restrict__ xB,
    const double* __restrict__ BinvT, double* __restrict__ scaled, uint8_t* __restrict__ touched, uint8_t* __restrict__ dense,
    int32_t* __restrict__ cnt_touched, int32_t* __restrict__ cnt_dense) {
  __shared__ int s_w[kRevThreads / kWave];
  if (st->status != ST_RUNNING) return;
  const int64_t k = (int64_t)blockIdx.x * kRevThreads + threadIdx.x;
  bool visit = false, is_dense = false;
  if (k < sh.m) {
    const int64_t l = st->leaving;
    const double ratio = st->ratio;
    const double s = -1.0 / mu[l];
    const double old = BinvT[k * sh.ldb + l];
    xB[k] = k == l ? ratio : __dadd_rn(xB[k], __dmul_rn(mu[k], ratio));
    const double sc = __dmul_rn(old, s);
    scaled[k] = sc;
    const bool changes = old != 0.0 || sc != 0.0;   // the update makes this row dense
    is_dense = changes || dense[k] != 0;
    if (changes) dense[k] = 1;
    visit = st->full_update != 0 || changes;
    touched[k] = visit ? 1 : 0;
  }
  // per-256-row totals for the ordered lists k_rev_finish_pivot_wide writes
  count_flag_wide(visit, cnt_touched, s_w);
  count_flag_wide(is_dense, cnt_dense, s_w);
}

// Closes the pivot after k_rev_pivot_vectors: list of the rows to update, mu[leaving] = 0,
// basis bookkeeping, trace, pivot count.
__global__ __launch_bounds__(kRevSelectThreads) void k_rev_finish_pivot(
    RevState* __restrict__ st, RevShape sh, double* __restrict__ mu, double* __restrict__ cB, const double* __restrict__ obj,
    int64_t* __restrict__ basis, uint8_t* __restrict__ is_basic, const uint8_t* __restrict__ touched, int32_t* __restrict__ idx_upd,
    int64_t* __restrict__ trace, int64_t trace_cap, const uint8_t* __restrict__ dense, int32_t* __restrict__ idx_dense) {
  __shared__ int s_w[kRevSelectThreads / kWave + 1];
  if (st->status != ST_RUNNING) return;
  const int64_t e = st->e, it = st->iter, l = st->leaving;
  list_flagged(touched, sh.m, idx_upd, s_w, &st->n_upd);
  list_flagged(dense, sh.m, idx_dense, s_w, &st->n_dense);
  if (threadIdx.x == 0) {
    mu[l] = 0.0;
    is_basic[e] = 1;
    is_basic[basis[l]] = 0;
    basis[l] = e;
    cB[l] = obj[e];
    if (it < trace_cap) { trace[2 * it] = e; trace[2 * it + 1] = l; }
    st->iter = it + 1;
    st->full_update = 0;
  }
}

// The same over the whole chip, one row per lane, with the per-256-row totals k_rev_pivot_vectors left.
__global__ __launch_bounds__(kRevThreads) void k_rev_finish_pivot_wide(
    RevState* __restrict__ st, RevShape sh, double* __restrict__ mu, double* __restrict__ cB, const double* __restrict__ obj,
    int64_t* __restrict__ basis, uint8_t* __restrict__ is_basic, const uint8_t* __restrict__ touched, int32_t* __restrict__ idx_upd,
    int64_t* __restrict__ trace, int64_t trace_cap, const uint8_t* __restrict__ dense, int32_t* __restrict__ idx_dense,
    const int32_t* __restrict__ cnt_touched, const int32_t* __restrict__ cnt_dense) {
  __shared__ int s_w[2 * kRevThreads / kWave];
  if (st->status != ST_RUNNING) return;
  const int64_t k = (int64_t)blockIdx.x * kRevThreads + threadIdx.x;
  const bool in = k < sh.m;
  compact_flag_wide(in && touched[k] != 0, k, cnt_touched, idx_upd, &st->n_upd, s_w);
  compact_flag_wide(in && dense[k] != 0, k, cnt_dense, idx_dense, &st->n_dense, s_w);
  if (k == 0) {
    const int64_t e = st->e, it = st->iter, l = st->leaving;
    mu[l] = 0.0;
    is_basic[e] = 1;
    is_basic[basis[l]] = 0;
    basis[l] = e;
    cB[l] = obj[e];
    if (it < trace_cap) { trace[2 * it] = e; trace[2 * it + 1] = l; }
    st->iter = it + 1;
    st->full_update = 0;
  }
}

// B^-1 = flushz(B^-1' + mu (x) scaled), where B^-1' is B^-1 with its leaving row
// replaced by `scaled` (reference :56-61: the zero drop of Tableau::Add reaches every
// entry, the scaled row included).  On the transposed storage that is, for every
// listed row k: BinvT[k][i] = flushz(x + scaled[k] * mu[i]) with x = scaled[k] for
// i == leaving and the stored value otherwise.  Rows that are not listed have
// scaled[k] == 0 and a zero leaving-row entry, so none of their entries would change
// (every stored entry is already a fixed point of the zero drop after the first,
// full, update).  A work item is one run of kUnroll * 256 16-byte vectors of one
// row; all loads of a run are issued before its first store.
template <int kUnroll, bool kNT>
__global__ __launch_bounds__(kRevThreads, 2) void k_rev_update(
    const RevState* __restrict__ st, RevShape sh, double* __restrict__ BinvT, const double* __restrict__ mu,
    const double* __restrict__ scaled, const int32_t* __restrict__ idx_upd) {
  if (st->status != ST_RUNNING) return;
  const int64_t l = st->leaving;
  const int n_vec = (int)(sh.ldb / 2);
  constexpr int kRun = kUnroll * kRevThreads;
  const int n_runs = (n_vec + kRun - 1) / kRun;
  const int64_t n_items = (int64_t)st->n_upd * n_runs;
  const double2* __restrict__ mu2 = reinterpret_cast<const double2*>(mu);
  for (int64_t w = blockIdx.x; w < n_items; w += gridDim.x) {
    const int64_t k = idx_upd[w / n_runs];
    const int v0 = (int)(w % n_runs) * kRun + threadIdx.x;
    const double u = scaled[k];
    double2* __restrict__ p = reinterpret_cast<double2*>(BinvT + k * sh.ldb);
    double2 x[kUnroll], c[kUnroll];
#pragma unroll
    for (int q = 0; q < kUnroll; q++) {
      const int v = v0 + q * kRevThreads;
      if (v < n_vec) { x[q] = ld16<kNT>(p + v); c[q] = mu2[v]; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < kUnroll; q++) {
      const int v = v0 + q * kRevThreads;
      if (v < n_vec) {
        if (2 * (int64_t)v == l) x[q].x = u;
        if (2 * (int64_t)v + 1 == l) x[q].y = u;
        x[q].x = flushz(__dadd_rn(x[q].x, __dmul_rn(c[q].x, u)));
        x[q].y = flushz(__dadd_rn(x[q].y, __dmul_rn(c[q].y, u)));
        st16<kNT>(p + v, x[q]);
      }
    }
  }
}

// ---- set-up -----------------------------------------------------------------
// Basis of slack or artificial columns: each has one non-zero entry v at row
// rows[i], so the inverse holds 1 / -v at (i, rows[i]) and the basic solution is
// b / -v (reference :212-233).  BinvT is zero on entry.
__global__ void k_rev_init_basis(RevShape sh, const int64_t* __restrict__ basis, const int64_t* __restrict__ rows,
                                 const double* __restrict__ Trm, const double* __restrict__ obj, double* __restrict__ BinvT,
                                 double* __restrict__ xB, double* __restrict__ cB, int64_t* __restrict__ pos) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sh.m) return;
  const int64_t bcol = basis[i], r = rows[i];
  cB[i] = obj[bcol];
  double x = 0.0;
  if (r >= 0) {
    const double v = Trm[r * sh.ldw + bcol];
    x = Trm[r * sh.ldw + sh.N] / (-v);
    const double inv = 1.0 / (-v);
    BinvT[r * sh.ldb + i] = inv;
    pos[r] = i;   // the single non-zero of row r of the transposed inverse
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
