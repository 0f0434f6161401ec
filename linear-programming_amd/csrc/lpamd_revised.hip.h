// Host side of the lpamd_rev_* entry points of include/lpamd.h: the revised simplex
// on the tableau (reference lp_revised_simplex_tableau.cc) with the tableau, the
// basis inverse and all vectors resident on one GPU.  Included at the end of
// lpamd_engine.hip (one translation unit, shared error handling).
#include "lpamd_revised_kernels.hip.h"

struct lpamd_rev_engine {
  RevShape sh{};
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // device memory
  double* Trm = nullptr;
  double* BinvT = nullptr;      // basis inverse, stored transposed
  double* obj = nullptr;
  double* cost = nullptr;
  uint8_t* is_basic = nullptr;
  uint8_t* negate = nullptr;
  int64_t* basis = nullptr;
  int64_t* rows = nullptr;      // row of the single non-zero of every initial basis column
  double* xB = nullptr;
  double* cB = nullptr;
  double* pvec = nullptr;
  double* mu = nullptr;
  double* colbuf = nullptr;
  double* scaled = nullptr;
  int32_t* idx_upd = nullptr;   // rows of BinvT the pending update visits
  int32_t* idx_rows = nullptr;  // rows of the tableau with a non-zero entry of p (pricing product)
  int32_t* idx_dense = nullptr; // rows of BinvT that an update has touched (ascending)
  uint8_t* dense = nullptr;     // the same as a byte per row
  int64_t* pos = nullptr;       // row of BinvT -> position of its single start entry (-1: none)
  uint8_t* touched = nullptr;   // one byte per row of BinvT: the pending update visits it
  int32_t* cnt_touched = nullptr;   // per 256 rows: flagged entries of touched / dense / non-zeros of p, the
  int32_t* cnt_dense = nullptr;     // first half of the chip-wide ordered lists
  int32_t* cnt_p = nullptr;
  double* sel_v = nullptr;      // pricing partials of the cost product, one per workgroup
  int64_t* sel_i = nullptr;
  int32_t sel_parts = 0;        // partials the last cost product left (0: none)
  int32_t rt_parts = 0;         // ratio-test partials the last B^-1 * column product left (0: none)
  bool p_counted = false;       // cnt_p describes pvec
  RevState* st = nullptr;
  int64_t* trace = nullptr;
  int64_t trace_cap = 0;
  // host mirrors
  RevState h_st{};
  std::vector<double> h_obj;    // W entries, the objective row as the reference holds it at this point
  std::vector<double> h_rhs;    // m entries, the right-hand side at upload time
  bool opt_reverted = false;
  bool loaded = false, started = false, finished = false;
  int32_t final_result = LPAMD_ERROR;
  double optimum = 0.0;
  // kernel shape overrides (0 = by size), for tests and tuning: LPAMD_REV_COLSUM_VEC=1|2,
  // LPAMD_REV_ROWDOT_ROWS=16|64, LPAMD_REV_UPDATE_UNROLL=4|16
  // LPAMD_REV_SPLIT=0|1: strided gathers inside the single-workgroup selection kernels / as
  // chip-wide kernels of their own (default: by size)
  // LPAMD_REV_TILE=0|16|64: column products one lane per output / through LDS tiles 16 or 64 outputs wide
  // LPAMD_REV_LIST=0: no row list in the pricing product (64-row chunk skipping only)
  int force_vec = 0, force_rowdot_rows = 0, force_update_unroll = 0, force_split = -1, force_tile = -1, force_colsum_threads = 0;
  int price_tile = -1;   // LPAMD_REV_PRICE_TILE: tile width of the pricing product alone (0 = one lane per output)
  bool list_rows = true;
  // LPAMD_REV_UNIT=0: the products over the inverse read every row instead of the dense rows only
  bool unit_rows = true;
  // LPAMD_REV_WIDE=0: ordered lists and the pricing arg-min in single workgroups instead of chip-wide
  bool wide = true;
  bool split() const { return force_split >= 0 ? force_split != 0 : sh.m > 8192; }
};

namespace {

#define CHECK_REV(h)                                                     \
  do {                                                                   \
    if (!(h)) return set_err(LPAMD_E_ARG, "null revised-simplex engine"); \
    HIP_TRY(hipSetDevice((h)->device));                                  \
  } while (0)

void rev_free(lpamd_rev_engine* h) {
  void* ptrs[] = {h->Trm, h->BinvT, h->obj, h->cost, h->is_basic, h->negate, h->basis, h->rows, h->xB,
                  h->cB, h->pvec, h->mu, h->colbuf, h->scaled, h->idx_upd, h->idx_rows, h->idx_dense, h->dense, h->pos, h->touched, h->st, h->trace,
                  h->cnt_touched, h->cnt_dense, h->cnt_p, h->sel_v, h->sel_i};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  h->Trm = h->BinvT = h->obj = h->cost = h->xB = h->cB = h->pvec = h->mu = h->colbuf = h->scaled = nullptr;
  h->is_basic = h->negate = h->touched = nullptr;
  h->basis = h->rows = h->trace = nullptr;
  h->idx_upd = h->idx_rows = h->idx_dense = nullptr;
  h->dense = nullptr;
  h->pos = nullptr;
  h->cnt_touched = h->cnt_dense = h->cnt_p = nullptr;
  h->sel_v = nullptr;
  h->sel_i = nullptr;
  h->sel_parts = 0;
  h->p_counted = false;
  h->st = nullptr;
  h->trace_cap = 0;
}

// Device buffers for W columns; everything is zero-filled.
int rev_allocate(lpamd_rev_engine* h, int64_t W) {
  rev_free(h);
  RevShape& sh = h->sh;
  sh.W = W;
  sh.ldw = round_up(W, 64);
  sh.ldb = round_up(sh.m, 64);
  const size_t tbytes = sizeof(double) * (size_t)sh.m * sh.ldw, bbytes = sizeof(double) * (size_t)sh.m * sh.ldb;
  hipError_t err = hipMalloc(&h->Trm, tbytes);
  if (err == hipSuccess) err = hipMalloc(&h->BinvT, bbytes);
  if (err != hipSuccess)
    return set_err(LPAMD_E_HIP, "hipMalloc of %.2f GB (tableau) + %.2f GB (basis inverse) failed: %s", tbytes / 1e9,
                   bbytes / 1e9, hipGetErrorString(err));
  HIP_TRY(hipMalloc(&h->obj, sizeof(double) * sh.ldw));
  HIP_TRY(hipMalloc(&h->cost, sizeof(double) * sh.ldw));
  HIP_TRY(hipMalloc(&h->is_basic, sh.ldw));
  HIP_TRY(hipMalloc(&h->negate, sh.ldb));
  HIP_TRY(hipMalloc(&h->basis, sizeof(int64_t) * sh.ldb));
  HIP_TRY(hipMalloc(&h->rows, sizeof(int64_t) * sh.ldb));
  HIP_TRY(hipMalloc(&h->idx_upd, sizeof(int32_t) * sh.ldb));
  HIP_TRY(hipMalloc(&h->idx_rows, sizeof(int32_t) * sh.ldb));
  HIP_TRY(hipMalloc(&h->idx_dense, sizeof(int32_t) * sh.ldb));
  HIP_TRY(hipMalloc(&h->dense, sh.ldb));
  HIP_TRY(hipMalloc(&h->pos, sizeof(int64_t) * sh.ldb));
  HIP_TRY(hipMalloc(&h->touched, sh.ldb));
  const size_t nblk = (size_t)(sh.ldb + kRevThreads - 1) / kRevThreads, nparts = (size_t)(sh.ldw + 15) / 16;
  HIP_TRY(hipMalloc(&h->cnt_touched, sizeof(int32_t) * nblk));
  HIP_TRY(hipMalloc(&h->cnt_dense, sizeof(int32_t) * nblk));
  HIP_TRY(hipMalloc(&h->cnt_p, sizeof(int32_t) * nblk));
  HIP_TRY(hipMalloc(&h->sel_v, sizeof(double) * nparts));
  HIP_TRY(hipMalloc(&h->sel_i, sizeof(int64_t) * nparts));
  double** vecs[] = {&h->xB, &h->cB, &h->pvec, &h->mu, &h->colbuf, &h->scaled};
  for (double** v : vecs) {
    HIP_TRY(hipMalloc(v, sizeof(double) * sh.ldb));
    HIP_TRY(hipMemsetAsync(*v, 0, sizeof(double) * sh.ldb, h->stream));
  }
  HIP_TRY(hipMalloc(&h->st, sizeof(RevState)));
  HIP_TRY(hipMemsetAsync(h->Trm, 0, tbytes, h->stream));
  HIP_TRY(hipMemsetAsync(h->obj, 0, sizeof(double) * sh.ldw, h->stream));
  HIP_TRY(hipMemsetAsync(h->cost, 0, sizeof(double) * sh.ldw, h->stream));
  HIP_TRY(hipMemsetAsync(h->is_basic, 0, sh.ldw, h->stream));
  HIP_TRY(hipMemsetAsync(h->negate, 0, sh.ldb, h->stream));
  HIP_TRY(hipMemsetAsync(h->st, 0, sizeof(RevState), h->stream));
  h->trace_cap = 4096;
  HIP_TRY(hipMalloc(&h->trace, sizeof(int64_t) * 2 * h->trace_cap));
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::memset(&h->h_st, 0, sizeof(RevState));
  h->h_st.max_iter = std::numeric_limits<int64_t>::max();
  h->started = h->finished = false;
  return LPAMD_OK;
}

int rev_ensure_trace(lpamd_rev_engine* h, int64_t need) {
  if (need <= h->trace_cap) return LPAMD_OK;
  int64_t cap = h->trace_cap;
  while (cap < need) cap *= 2;
  int64_t* nt = nullptr;
  HIP_TRY(hipMalloc(&nt, sizeof(int64_t) * 2 * cap));
  HIP_TRY(hipMemcpyAsync(nt, h->trace, sizeof(int64_t) * 2 * h->trace_cap, hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipFree(h->trace));
  h->trace = nt;
  h->trace_cap = cap;
  return LPAMD_OK;
}

int rev_pull(lpamd_rev_engine* h) {
  HIP_TRY(hipMemcpyAsync(&h->h_st, h->st, sizeof(RevState), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return LPAMD_OK;
}
int rev_push(lpamd_rev_engine* h) {
  HIP_TRY(hipMemcpyAsync(h->st, &h->h_st, sizeof(RevState), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return LPAMD_OK;
}
int rev_upload_obj(lpamd_rev_engine* h) {
  HIP_TRY(hipMemcpyAsync(h->obj, h->h_obj.data(), sizeof(double) * h->sh.W, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return LPAMD_OK;
}

// out[0..ncols) = sum over rows of coef[r] * M[r][.] (+ base with the zero drop when it is given)
// list_rows: first list the rows with a non-zero coefficient and read only those (pays when the
// non-zeros are few and scattered, as those of p are for thousands of pivots).
// over_inverse: M is the transposed inverse; only its dense rows are read and every output takes
// the single term of its unit row (k_rev_coltile<.., kUnit>).
// price: the cost product also leaves one pricing partial per workgroup (h->sel_parts of them).
// nz_counts: per-256 non-zero counts of coef, if a kernel already left them (chip-wide listing).
int rev_colsum(lpamd_rev_engine* h, const double* M, int64_t ld, int64_t nrows, int64_t ncols, const double* coef,
               const double* base, double* out, bool list_rows = false, bool over_inverse = false, bool price = false,
               const int32_t* nz_counts = nullptr) {
  h->sel_parts = 0;
  h->rt_parts = 0;
  // LDS-tiled product: 64 outputs per workgroup once that fills the chip, 16 below.  Measured at
  // m=32768 n=65536 with a dense coefficient vector: 2.87 / 1.18 ms against 3.58 / 1.49 ms for one
  // lane per output (pricing / B^-1 column); at m=4096: 0.30 against 0.47 ms per pivot.
  int tile = h->force_tile >= 0 ? h->force_tile : (ncols >= 64 * 256 ? 64 : 16);
  // The pricing product over a short row list (the non-zeros of p stay few for thousands of pivots)
  // streams best with one lane per output: whole rows go by in 2 KB runs per workgroup, and the few
  // terms per output need no LDS staging.  Measured at m=32768 n=65536 with 665 listed rows: 80 us
  // (6.5 TB/s) against 146 us tiled.  The choice uses the list length the host last saw.
  bool listed_lanes = false;
  if (price && list_rows && h->list_rows && h->force_tile < 0 && ncols >= 64 * 256 && h->h_st.n_rows <= 8192) { tile = 0; listed_lanes = true; }
  if (price && h->price_tile >= 0) { tile = h->price_tile; listed_lanes = false; }
  if (tile == 64 || tile == 16) {
    const dim3 tgrid((unsigned)((ncols + tile - 1) / tile)), tblock(kRevThreads);
    if (over_inverse && h->unit_rows && !base) {
      const int32_t* n_dense = &h->st->n_dense;
      // mu = B^-1 * column: with the basic solution at hand the workgroups also leave ratio-test partials
      double* rt_v = h->wide ? h->sel_v : nullptr;
      const double* xb = rt_v ? h->xB : nullptr;
      if (rt_v) h->rt_parts = (int32_t)tgrid.x;
      if (tile == 64)
        hipLaunchKernelGGL((k_rev_coltile<64, false, true, true>), tgrid, tblock, 0, h->stream, h->st, M, ld, nrows, ncols, coef, xb, out,
                           h->idx_dense, n_dense, h->rows, h->dense, rt_v, h->sel_i);
      else
        hipLaunchKernelGGL((k_rev_coltile<16, false, true, true>), tgrid, tblock, 0, h->stream, h->st, M, ld, nrows, ncols, coef, xb, out,
                           h->idx_dense, n_dense, h->rows, h->dense, rt_v, h->sel_i);
      return check_launch("k_rev_coltile");
    }
    const int32_t* rows = nullptr;
    const int32_t* n_listed = nullptr;
    if (list_rows && h->list_rows) {
      if (nz_counts)
        hipLaunchKernelGGL(k_rev_list_nonzero_wide, dim3((unsigned)((nrows + kRevThreads - 1) / kRevThreads)), dim3(kRevThreads), 0, h->stream,
                           h->st, coef, nrows, nz_counts, h->idx_rows, &h->st->n_rows);
      else
        hipLaunchKernelGGL(k_rev_list_nonzero, dim3(1), dim3(kRevSelectThreads), 0, h->stream, h->st, coef, nrows, h->idx_rows, &h->st->n_rows);
      LP_TRY(check_launch("k_rev_list_nonzero"));
      rows = h->idx_rows;
      n_listed = &h->st->n_rows;
    }
    const bool parts = price && base && h->wide;
    double* sel_v = parts ? h->sel_v : nullptr;
    if (parts) h->sel_parts = (int32_t)tgrid.x;
#define LPAMD_REV_COLTILE(W, C, L)                                                                                                       \
  hipLaunchKernelGGL((k_rev_coltile<W, C, L>), tgrid, tblock, 0, h->stream, h->st, M, ld, nrows, ncols, coef, base, out, rows, n_listed, \
                     (const int64_t*)nullptr, (const uint8_t*)nullptr, sel_v, h->sel_i, (const uint8_t*)h->is_basic, h->sh.N)
#define LPAMD_REV_COLTILE_L(W, C)                         \
  do {                                                    \
    if (rows) LPAMD_REV_COLTILE(W, C, true);              \
    else LPAMD_REV_COLTILE(W, C, false);                  \
  } while (0)
    if (tile == 64) { if (base) LPAMD_REV_COLTILE_L(64, true); else LPAMD_REV_COLTILE_L(64, false); }
    else { if (base) LPAMD_REV_COLTILE_L(16, true); else LPAMD_REV_COLTILE_L(16, false); }
#undef LPAMD_REV_COLTILE_L
#undef LPAMD_REV_COLTILE
    return check_launch("k_rev_coltile");
  }
  // one lane per output: two columns (16 bytes) per lane once that still leaves a wavefront per CU,
  // one wavefront per workgroup on narrow outputs so that the lanes spread over all CUs
  const int32_t* rows = nullptr;
  const int32_t* n_listed = nullptr;
  if (list_rows && h->list_rows && !over_inverse) {
    if (nz_counts)
      hipLaunchKernelGGL(k_rev_list_nonzero_wide, dim3((unsigned)((nrows + kRevThreads - 1) / kRevThreads)), dim3(kRevThreads), 0, h->stream,
                         h->st, coef, nrows, nz_counts, h->idx_rows, &h->st->n_rows);
    else
      hipLaunchKernelGGL(k_rev_list_nonzero, dim3(1), dim3(kRevSelectThreads), 0, h->stream, h->st, coef, nrows, h->idx_rows, &h->st->n_rows);
    LP_TRY(check_launch("k_rev_list_nonzero"));
    rows = h->idx_rows;
    n_listed = &h->st->n_rows;
  }
  const bool vec2 = h->force_vec ? h->force_vec == 2 : (ncols >= 2 * 64 * 128 && !listed_lanes);
  int threads = ncols / (vec2 ? 2 : 1) >= 512 * kRevThreads ? kRevThreads : kWave;
  if (listed_lanes) threads = kRevThreads;
  if (h->force_colsum_threads) threads = h->force_colsum_threads;
  const int64_t per_block = (int64_t)threads * (vec2 ? 2 : 1);
  const dim3 grid((unsigned)((ncols + per_block - 1) / per_block)), block(threads);
  const bool parts = price && base && h->wide;
  double* sel_v = parts ? h->sel_v : nullptr;
  if (parts) h->sel_parts = (int32_t)grid.x;
#define LPAMD_REV_COLSUM(V, C, L)                                                                                                             \
  hipLaunchKernelGGL((k_rev_colsum<V, 16, C, L>), grid, block, 0, h->stream, h->st, M, ld, nrows, ncols, coef, base, out, rows, n_listed, \
                     sel_v, h->sel_i, (const uint8_t*)h->is_basic, h->sh.N)
#define LPAMD_REV_COLSUM_L(V, C)                 \
  do {                                           \
    if (rows) LPAMD_REV_COLSUM(V, C, true);      \
    else LPAMD_REV_COLSUM(V, C, false);          \
  } while (0)
  if (base) { if (vec2) LPAMD_REV_COLSUM_L(2, true); else LPAMD_REV_COLSUM_L(1, true); }
  else { if (vec2) LPAMD_REV_COLSUM_L(2, false); else LPAMD_REV_COLSUM_L(1, false); }
#undef LPAMD_REV_COLSUM_L
#undef LPAMD_REV_COLSUM
  return check_launch("k_rev_colsum");
}

// p = c_B^T * B^-1: a dot product along every row of the transposed inverse
int rev_rowdot(lpamd_rev_engine* h) {
  const RevShape& sh = h->sh;
  // Short list of dense rows (thousands of pivots pass before it grows beyond a few thousand): one
  // workgroup per row, so that the chip is busy although the rows are few.  Long list: 16 rows per
  // workgroup through LDS tiles, where 16 lanes add 16 rows side by side (64-row tiles lost against
  // 16-row tiles at every list length measured).  Both kernels take any list length; the choice uses
  // the length the host last saw, which only costs speed when it is stale.
  // LPAMD_REV_ROWDOT_ROWS=1|16|64 pins the variant.
  const int32_t* rows = h->unit_rows ? h->idx_dense : nullptr;
  const int32_t* n_listed = h->unit_rows ? &h->st->n_dense : nullptr;
  const int64_t n_hint = rows ? (int64_t)h->h_st.n_dense : sh.m;
  const int kr = h->force_rowdot_rows ? h->force_rowdot_rows : (n_hint <= 4096 ? 1 : 16);
  if (kr == 1) {
    const dim3 grid((unsigned)std::min<int64_t>(sh.m, 2048)), block(kRevThreads);   // rows are dealt round-robin
    if (rows)
      hipLaunchKernelGGL((k_rev_rowdot_row<true, 8>), grid, block, 0, h->stream, h->st, h->BinvT, sh.ldb, sh.m, sh.m, h->cB, h->pvec, rows, n_listed);
    else
      hipLaunchKernelGGL((k_rev_rowdot_row<false, 8>), grid, block, 0, h->stream, h->st, h->BinvT, sh.ldb, sh.m, sh.m, h->cB, h->pvec, rows, n_listed);
  } else {
    const bool rows64 = kr == 64;
    const dim3 grid((unsigned)((sh.m + (rows64 ? 64 : 16) - 1) / (rows64 ? 64 : 16))), block(kRevThreads);
#define LPAMD_REV_ROWDOT(R, L) \
  hipLaunchKernelGGL((k_rev_rowdot<R, L>), grid, block, 0, h->stream, h->st, h->BinvT, sh.ldb, sh.m, sh.m, h->cB, h->pvec, rows, n_listed)
    if (rows64) { if (rows) LPAMD_REV_ROWDOT(64, true); else LPAMD_REV_ROWDOT(64, false); }
    else { if (rows) LPAMD_REV_ROWDOT(16, true); else LPAMD_REV_ROWDOT(16, false); }
#undef LPAMD_REV_ROWDOT
  }
  LP_TRY(check_launch("k_rev_rowdot"));
  if (rows) {
    hipLaunchKernelGGL(k_rev_unit_rows_p, dim3((unsigned)((sh.m + kRevThreads - 1) / kRevThreads)), dim3(kRevThreads), 0, h->stream, h->st,
                       h->BinvT, sh.ldb, sh.m, h->cB, h->pos, h->dense, h->pvec, h->wide ? h->cnt_p : nullptr);
    LP_TRY(check_launch("k_rev_unit_rows_p"));
  }
  h->p_counted = rows && h->wide;
  return LPAMD_OK;
}

int rev_enqueue_update(lpamd_rev_engine* h) {
  const RevShape& sh = h->sh;
  const int grid = 4096;   // work items (listed row x run of vectors) are dealt round-robin
  const bool unroll16 = h->force_update_unroll ? h->force_update_unroll == 16 : sh.ldb >= 8192;
  // nt loads/stores once the inverse exceeds the Infinity Cache, as in the tableau update
  const bool nt = (double)sh.m * (double)sh.ldb * sizeof(double) > 512e6;
#define LPAMD_REV_UPDATE(U, NT) \
  hipLaunchKernelGGL((k_rev_update<U, NT>), dim3(grid), dim3(kRevThreads), 0, h->stream, h->st, sh, h->BinvT, h->mu, h->scaled, h->idx_upd)
  if (unroll16) { if (nt) LPAMD_REV_UPDATE(16, true); else LPAMD_REV_UPDATE(16, false); }
  else { if (nt) LPAMD_REV_UPDATE(4, true); else LPAMD_REV_UPDATE(4, false); }
#undef LPAMD_REV_UPDATE
  return check_launch("k_rev_update");
}

// Entering column (pricing, or the first column with a non-zero probe entry) and its copy in colbuf.
int rev_enqueue_entering(lpamd_rev_engine* h, bool first_nonzero, int64_t ncols_active) {
  const RevShape& sh = h->sh;
  const dim3 one(1), sel(kRevSelectThreads);
#define LPAMD_REV_ENTER(F, G) \
  hipLaunchKernelGGL((k_rev_select_entering<F, G>), one, sel, 0, h->stream, h->st, sh, ncols_active, h->cost, h->is_basic, h->Trm, h->colbuf)
  if (!h->split()) {
    if (first_nonzero) LPAMD_REV_ENTER(true, true); else LPAMD_REV_ENTER(false, true);
    return check_launch("k_rev_select_entering");
  }
  if (first_nonzero) LPAMD_REV_ENTER(true, false); else LPAMD_REV_ENTER(false, false);
#undef LPAMD_REV_ENTER
  LP_TRY(check_launch("k_rev_select_entering"));
  hipLaunchKernelGGL(k_rev_gather_wide, dim3((unsigned)((sh.m + kRevThreads - 1) / kRevThreads)), dim3(kRevThreads), 0, h->stream, h->st,
                     h->Trm, sh.ldw, sh.m, (int64_t)-1, h->colbuf);
  return check_launch("k_rev_gather_wide");
}

// B^-1 * colbuf, ratio test (or the forced leaving
// position), vector part of the pivot, rank-1 update.
int rev_enqueue_pivot_tail(lpamd_rev_engine* h, int64_t forced_leaving) {
  const RevShape& sh = h->sh;
  const dim3 one(1), sel(kRevSelectThreads);
  LP_TRY(rev_colsum(h, h->BinvT, sh.ldb, sh.m, sh.m, h->colbuf, nullptr, h->mu, false, true));
  if (!h->split()) {
    hipLaunchKernelGGL((k_rev_select_leaving<true>), one, sel, 0, h->stream, h->st, sh, forced_leaving, h->mu, h->xB, h->cB, h->obj,
                       h->basis, h->is_basic, h->BinvT, h->scaled, h->idx_upd, h->trace, h->trace_cap, h->dense, h->idx_dense,
                       h->sel_v, h->sel_i, (int)h->rt_parts);
    LP_TRY(check_launch("k_rev_select_leaving"));
  } else {
    hipLaunchKernelGGL((k_rev_select_leaving<false>), one, sel, 0, h->stream, h->st, sh, forced_leaving, h->mu, h->xB, h->cB, h->obj,
                       h->basis, h->is_basic, h->BinvT, h->scaled, h->idx_upd, h->trace, h->trace_cap, h->dense, h->idx_dense,
                       h->sel_v, h->sel_i, (int)h->rt_parts);
    LP_TRY(check_launch("k_rev_select_leaving"));
    hipLaunchKernelGGL(k_rev_pivot_vectors, dim3((unsigned)((sh.m + kRevThreads - 1) / kRevThreads)), dim3(kRevThreads), 0, h->stream,
                       h->st, sh, h->mu, h->xB, h->BinvT, h->scaled, h->touched, h->dense, h->cnt_touched, h->cnt_dense);
    LP_TRY(check_launch("k_rev_pivot_vectors"));
    if (h->wide)
      hipLaunchKernelGGL(k_rev_finish_pivot_wide, dim3((unsigned)((sh.m + kRevThreads - 1) / kRevThreads)), dim3(kRevThreads), 0, h->stream,
                         h->st, sh, h->mu, h->cB, h->obj, h->basis, h->is_basic, h->touched, h->idx_upd, h->trace, h->trace_cap, h->dense,
                         h->idx_dense, h->cnt_touched, h->cnt_dense);
    else
      hipLaunchKernelGGL(k_rev_finish_pivot, one, sel, 0, h->stream, h->st, sh, h->mu, h->cB, h->obj, h->basis, h->is_basic, h->touched,
                         h->idx_upd, h->trace, h->trace_cap, h->dense, h->idx_dense);
    LP_TRY(check_launch("k_rev_finish_pivot"));
  }
  return rev_enqueue_update(h);
}

// One iteration of the main loop (reference lp_revised_simplex_tableau.cc:235-299).
int rev_enqueue_iteration(lpamd_rev_engine* h, int64_t ncols_active) {
  const RevShape& sh = h->sh;
  LP_TRY(rev_rowdot(h));
  LP_TRY(rev_colsum(h, h->Trm, sh.ldw, sh.m, ncols_active, h->pvec, h->obj, h->cost, true, false, true,
                    h->p_counted ? h->cnt_p : nullptr));
  if (h->sel_parts > 0) {
    // the cost product left pricing partials: fold them and fetch the entering column chip-wide
    hipLaunchKernelGGL(k_rev_enter_and_gather, dim3((unsigned)((sh.m + kRevThreads - 1) / kRevThreads)), dim3(kRevThreads), 0, h->stream,
                       h->st, h->sel_v, h->sel_i, (int)h->sel_parts, h->Trm, sh.ldw, sh.m, h->colbuf);
    LP_TRY(check_launch("k_rev_enter_and_gather"));
  } else {
    LP_TRY(rev_enqueue_entering(h, false, ncols_active));
  }
  return rev_enqueue_pivot_tail(h, -1);
}

// Runs the device-resident loop until the status leaves RUNNING or `max_pivots`
// (total count, < 0: none) is reached.
int rev_main_loop(lpamd_rev_engine* h, int64_t ncols_active, int64_t max_pivots, int32_t* result) {
  LP_TRY(rev_pull(h));
  h->h_st.status = ST_RUNNING;
  h->h_st.max_iter = max_pivots < 0 ? std::numeric_limits<int64_t>::max() : max_pivots;
  LP_TRY(rev_push(h));
  int64_t batch = 4;
  for (;;) {
    int64_t todo = batch;
    if (max_pivots >= 0) todo = std::max<int64_t>(1, std::min<int64_t>(batch, max_pivots - h->h_st.iter + 1));
    LP_TRY(rev_ensure_trace(h, h->h_st.iter + todo));
    const int64_t before = h->h_st.iter;
    for (int64_t k = 0; k < todo; k++) LP_TRY(rev_enqueue_iteration(h, ncols_active));
    LP_TRY(rev_pull(h));
    if (h->h_st.status != ST_RUNNING) break;
    if (h->h_st.iter == before) {
      h->h_st.status = ST_ERROR;
      set_err(LPAMD_E_STATE, "revised simplex loop made no progress");
      break;
    }
    batch = std::min<int64_t>(batch * 2, 256);
  }
  *result = status_to_result(h->h_st.status);
  return LPAMD_OK;
}

int rev_read_basis_solution(lpamd_rev_engine* h, std::vector<int64_t>& basis, std::vector<double>& xB) {
  basis.resize(h->sh.m);
  xB.resize(h->sh.m);
  HIP_TRY(hipMemcpyAsync(basis.data(), h->basis, sizeof(int64_t) * h->sh.m, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(xB.data(), h->xB, sizeof(double) * h->sh.m, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return LPAMD_OK;
}

// reference lp_revised_simplex_tableau.cc:253-257
int rev_compute_optimum(lpamd_rev_engine* h) {
  std::vector<int64_t> basis;
  std::vector<double> xB;
  LP_TRY(rev_read_basis_solution(h, basis, xB));
  double v = h->h_obj[h->sh.N];
  for (int64_t i = 0; i < h->sh.m; i++) v += h->h_obj[basis[i]] * xB[i];
  if (h->opt_reverted) v *= -1;
  h->optimum = v;
  return LPAMD_OK;
}

int rev_init_basis(lpamd_rev_engine* h) {
  const RevShape& sh = h->sh;
  HIP_TRY(hipMemsetAsync(h->BinvT, 0, sizeof(double) * (size_t)sh.m * sh.ldb, h->stream));
  HIP_TRY(hipMemsetAsync(h->dense, 0, sh.ldb, h->stream));
  HIP_TRY(hipMemsetAsync(h->pos, 0xFF, sizeof(int64_t) * sh.ldb, h->stream));
  hipLaunchKernelGGL(k_rev_init_basis, dim3((unsigned)((sh.m + 255) / 256)), dim3(256), 0, h->stream, sh, h->basis, h->rows, h->Trm,
                     h->obj, h->BinvT, h->xB, h->cB, h->pos);
  LP_TRY(check_launch("k_rev_init_basis"));
  // an initial entry below the zero-drop threshold survives until the first update, which
  // therefore has to visit every row once
  LP_TRY(rev_pull(h));
  h->h_st.full_update = 1;
  h->h_st.n_dense = 0;
  return rev_push(h);
}

// Feasibility phase (reference lp_revised_simplex_tableau.cc:92-197).
int rev_phase1(lpamd_rev_engine* h, int64_t max_pivots, int32_t* result) {
  const RevShape& sh = h->sh;
  const int64_t m = sh.m, N = sh.N, W0 = N + 1;
  if (sh.W < W0 + m) return set_err(LPAMD_E_STATE, "the feasibility phase needs the artificial block (upload decides it from the right-hand side)");
  // rows with a negative right-hand side are negated (:95-114)
  std::vector<uint8_t> neg(m, 0);
  for (int64_t i = 0; i < m; i++) neg[i] = h_is_neg(h->h_rhs[i]) ? 1 : 0;
  HIP_TRY(hipMemcpyAsync(h->negate, neg.data(), m, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_rev_negate_rows, dim3((unsigned)m, (unsigned)((W0 + 255) / 256)), dim3(256), 0, h->stream, sh, h->negate, h->Trm);
  LP_TRY(check_launch("k_rev_negate_rows"));
  const std::vector<double> raw_obj = h->h_obj;
  const bool raw_reverted = h->opt_reverted;
  // auxiliary problem: one artificial column per row, objective -sum(artificials) (:116-137),
  // negated once more by the nested solve (:201-202)
  std::vector<uint8_t> basic(sh.W, 0);
  std::vector<int64_t> basis(m), rows(m);
  for (int64_t j = 0; j < sh.W; j++) h->h_obj[j] = 0.0;
  for (int64_t i = 0; i < m; i++) {
    h->h_obj[W0 + i] = -1.0;
    basic[W0 + i] = 1;
    basis[i] = W0 + i;
    rows[i] = i;
  }
  for (int64_t j = 0; j < sh.W; j++) h->h_obj[j] = h->h_obj[j] * -1.0;
  h->opt_reverted = !h->opt_reverted;
  hipLaunchKernelGGL(k_rev_fill_artificials, dim3((unsigned)m, (unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, sh, h->Trm);
  LP_TRY(check_launch("k_rev_fill_artificials"));
  HIP_TRY(hipMemcpyAsync(h->is_basic, basic.data(), sh.W, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->basis, basis.data(), sizeof(int64_t) * m, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->rows, rows.data(), sizeof(int64_t) * m, hipMemcpyHostToDevice, h->stream));
  LP_TRY(rev_upload_obj(h));   // synchronises: the host vectors above may go out of scope
  LP_TRY(rev_init_basis(h));
  int32_t r1 = LPAMD_ERROR;
  LP_TRY(rev_main_loop(h, sh.W, max_pivots, &r1));
  if (r1 == LPAMD_LIMIT) { *result = LPAMD_LIMIT; return LPAMD_OK; }
  if (r1 != LPAMD_SOLVED) { *result = LPAMD_ERROR; return LPAMD_OK; }        // :140, a live assert upstream
  LP_TRY(rev_compute_optimum(h));
  if (!h_is_zero(h->optimum)) { *result = LPAMD_NOSOLUTION; return LPAMD_OK; }   // :142
  // artificial variables still basic are pivoted out on the first structural column
  // with a non-zero entry in their row of B^-1 * A (:145-170)
  std::vector<double> xB;
  LP_TRY(rev_read_basis_solution(h, basis, xB));
  h->h_st.status = ST_RUNNING;
  h->h_st.max_iter = std::numeric_limits<int64_t>::max();
  LP_TRY(rev_push(h));
  for (int64_t i = 0; i < m; i++) {
    if (basis[i] < N) continue;
    LP_TRY(rev_ensure_trace(h, h->h_st.iter + 1));
    // row i of B^-1 (column i of the transposed storage) times the tableau = row i of B^-1 A
    hipLaunchKernelGGL(k_rev_gather_wide, dim3((unsigned)((m + kRevThreads - 1) / kRevThreads)), dim3(kRevThreads), 0, h->stream, h->st,
                       h->BinvT, sh.ldb, m, i, h->colbuf);
    LP_TRY(check_launch("k_rev_gather_wide"));
    LP_TRY(rev_colsum(h, h->Trm, sh.ldw, m, W0, h->colbuf, nullptr, h->cost));
    LP_TRY(rev_enqueue_entering(h, true, W0));
    LP_TRY(rev_enqueue_pivot_tail(h, i));
    LP_TRY(rev_pull(h));
    if (h->h_st.status != ST_RUNNING) { *result = LPAMD_ERROR; return LPAMD_OK; }   // :165, a live assert upstream
    basis[i] = h->h_st.e;
  }
  // back to the real objective (:171-195); the artificial block is simply left out of the pricing from here on
  h->h_obj = raw_obj;
  h->opt_reverted = raw_reverted;
  LP_TRY(rev_upload_obj(h));
  hipLaunchKernelGGL(k_rev_gather_basis_cost, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, m, h->basis, h->obj, h->cB);
  LP_TRY(check_launch("k_rev_gather_basis_cost"));
  *result = LPAMD_SOLVED;
  return LPAMD_OK;
}

}  // namespace

extern "C" {

int lpamd_rev_create(lpamd_rev_engine** out, int64_t m, int64_t N, void* hip_stream) {
  if (!out) return set_err(LPAMD_E_ARG, "out is null");
  *out = nullptr;
  if (m <= 0 || N <= 0) return set_err(LPAMD_E_ARG, "bad shape m=%lld N=%lld", (long long)m, (long long)N);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(LPAMD_E_NODEVICE, "no HIP device visible");
  lpamd_rev_engine* h = new lpamd_rev_engine();
  h->sh.m = m;
  h->sh.N = N;
  if (hipGetDevice(&h->device) != hipSuccess) { delete h; return set_err(LPAMD_E_HIP, "hipGetDevice failed"); }
  if (hip_stream) {
    h->stream = (hipStream_t)hip_stream;
  } else if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return set_err(LPAMD_E_HIP, "hipStreamCreate failed");
  } else {
    h->own_stream = true;
  }
  if (const char* env = getenv("LPAMD_REV_COLSUM_VEC")) h->force_vec = atoi(env);
  if (const char* env = getenv("LPAMD_REV_ROWDOT_ROWS")) h->force_rowdot_rows = atoi(env);
  if (const char* env = getenv("LPAMD_REV_UPDATE_UNROLL")) h->force_update_unroll = atoi(env);
  if (const char* env = getenv("LPAMD_REV_SPLIT")) h->force_split = atoi(env);
  if (const char* env = getenv("LPAMD_REV_TILE")) h->force_tile = atoi(env);
  if (const char* env = getenv("LPAMD_REV_WIDE")) h->wide = atoi(env) != 0;
  if (const char* env = getenv("LPAMD_REV_PRICE_TILE")) h->price_tile = atoi(env);
  if (const char* env = getenv("LPAMD_REV_COLSUM_THREADS")) h->force_colsum_threads = atoi(env);
  if (const char* env = getenv("LPAMD_REV_LIST")) h->list_rows = atoi(env) != 0;
  if (const char* env = getenv("LPAMD_REV_UNIT")) h->unit_rows = atoi(env) != 0;
  *out = h;
  return LPAMD_OK;
}

int lpamd_rev_destroy(lpamd_rev_engine* h) {
  if (!h) return LPAMD_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  rev_free(h);
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return LPAMD_OK;
}

int lpamd_rev_upload(lpamd_rev_engine* h, const double* T, int64_t ld, const double* obj, const uint8_t* is_basic, int32_t opt_reverted) {
  CHECK_REV(h);
  const int64_t m = h->sh.m, N = h->sh.N, W0 = N + 1;
  if (!T || !obj || !is_basic || ld < m) return set_err(LPAMD_E_ARG, "rev_upload: null buffer or ld < m");
  bool need_init = false;
  for (int64_t i = 0; i < m; i++) need_init |= h_is_neg(T[N * ld + i]);   // reference :8-16
  // initial basis: the basic columns in index order, each with its first non-zero row (:216-232)
  std::vector<int64_t> basis, rows;
  for (int64_t j = 0; j < N; j++) {
    if (!is_basic[j]) continue;
    int64_t r = -1;
    for (int64_t i = 0; i < m; i++)
      if (!h_is_zero(T[j * ld + i])) { r = i; break; }
    basis.push_back(j);
    rows.push_back(r);
  }
  if ((int64_t)basis.size() != m)
    return set_err(LPAMD_E_ARG, "rev_upload: %lld basic columns for %lld rows", (long long)basis.size(), (long long)m);
  LP_TRY(rev_allocate(h, need_init ? W0 + m : W0));
  const RevShape& sh = h->sh;
  std::vector<double> rm((size_t)m * sh.ldw, 0.0);
  for (int64_t j = 0; j < W0; j++)
    for (int64_t i = 0; i < m; i++) rm[(size_t)i * sh.ldw + j] = T[j * ld + i];
  std::vector<uint8_t> basic(sh.W, 0);
  for (int64_t j = 0; j < N; j++) basic[j] = is_basic[j] ? 1 : 0;
  h->h_obj.assign(sh.W, 0.0);
  for (int64_t j = 0; j < W0; j++) h->h_obj[j] = obj[j];
  h->h_rhs.assign(T + N * ld, T + N * ld + m);
  h->opt_reverted = opt_reverted != 0;
  HIP_TRY(hipMemcpyAsync(h->Trm, rm.data(), sizeof(double) * rm.size(), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->is_basic, basic.data(), sh.W, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->basis, basis.data(), sizeof(int64_t) * m, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->rows, rows.data(), sizeof(int64_t) * m, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->loaded = true;
  return LPAMD_OK;
}

int lpamd_rev_fill_synthetic(lpamd_rev_engine* h, int64_t n_struct, uint64_t seed) {
  CHECK_REV(h);
  const int64_t m = h->sh.m, N = h->sh.N;
  if (n_struct <= 0 || n_struct + m != N) return set_err(LPAMD_E_ARG, "rev_fill_synthetic: N must equal n_struct + m");
  LP_TRY(rev_allocate(h, N + 1));
  const RevShape& sh = h->sh;
  hipLaunchKernelGGL(k_rev_fill_synthetic, dim3((unsigned)m, (unsigned)((sh.ldw + 255) / 256)), dim3(256), 0, h->stream, sh, h->Trm,
                     h->is_basic, h->basis, h->rows, n_struct, seed);
  LP_TRY(check_launch("k_rev_fill_synthetic"));
  h->h_obj.assign(sh.W, 0.0);
  for (int64_t j = 0; j <= N; j++) h->h_obj[j] = synth_entry(seed, m, n_struct, m, j);
  h->h_rhs.resize(m);
  for (int64_t i = 0; i < m; i++) h->h_rhs[i] = synth_entry(seed, m, n_struct, i, N);
  h->opt_reverted = false;
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->loaded = true;
  return LPAMD_OK;
}

int lpamd_rev_solve(lpamd_rev_engine* h, int64_t max_pivots, int32_t* result, int64_t* n_pivots) {
  CHECK_REV(h);
  if (!h->loaded) return set_err(LPAMD_E_STATE, "no tableau uploaded yet");
  if (!result) return set_err(LPAMD_E_ARG, "result is null");
  const RevShape& sh = h->sh;
  const int64_t W0 = sh.N + 1;
  int32_t res = LPAMD_ERROR;
  if (h->finished) return set_err(LPAMD_E_STATE, "rev_solve: this tableau has been solved already (upload again)");
  if (!h->started) {
    h->started = true;
    for (int64_t j = 0; j < W0; j++) h->h_obj[j] = h->h_obj[j] * -1.0;   // reference :201
    h->opt_reverted = !h->opt_reverted;                                   // :202
    bool need_init = false;
    for (int64_t i = 0; i < sh.m; i++) need_init |= h_is_neg(h->h_rhs[i]);
    if (need_init) {
      LP_TRY(rev_phase1(h, max_pivots, &res));
      if (res != LPAMD_SOLVED) {
        h->finished = true;
        h->final_result = res;
        *result = res;
        if (n_pivots) *n_pivots = h->h_st.iter;
        return LPAMD_OK;
      }
    } else {
      LP_TRY(rev_upload_obj(h));
      LP_TRY(rev_init_basis(h));
    }
  }
  LP_TRY(rev_main_loop(h, W0, max_pivots, &res));
  if (res == LPAMD_SOLVED) LP_TRY(rev_compute_optimum(h));
  if (res != LPAMD_LIMIT) h->finished = true;   // a budgeted run may be continued with a larger budget
  h->final_result = res;
  *result = res;
  if (n_pivots) *n_pivots = h->h_st.iter;
  return LPAMD_OK;
}

int lpamd_rev_optimum(lpamd_rev_engine* h, double* out) {
  CHECK_REV(h);
  if (!h->finished || h->final_result != LPAMD_SOLVED) return set_err(LPAMD_E_STATE, "rev_optimum: not solved");
  *out = h->optimum;
  return LPAMD_OK;
}

int lpamd_rev_column_values(lpamd_rev_engine* h, double* x) {
  CHECK_REV(h);
  if (!h->started) return set_err(LPAMD_E_STATE, "rev_column_values: rev_solve has not run");
  std::vector<int64_t> basis;
  std::vector<double> xB;
  LP_TRY(rev_read_basis_solution(h, basis, xB));
  for (int64_t j = 0; j < h->sh.N + 2; j++) x[j] = 0.0;
  for (int64_t i = 0; i < h->sh.m; i++)
    if (basis[i] >= 0 && basis[i] < h->sh.N) x[basis[i]] = xB[i];
  return LPAMD_OK;
}

int lpamd_rev_pivot_count(lpamd_rev_engine* h, int64_t* n) {
  CHECK_REV(h);
  *n = h->h_st.iter;
  return LPAMD_OK;
}

int lpamd_rev_list_sizes(lpamd_rev_engine* h, int64_t* n_update_rows, int64_t* n_pricing_rows, int64_t* n_dense_rows) {
  CHECK_REV(h);
  if (n_update_rows) *n_update_rows = h->h_st.n_upd;
  if (n_pricing_rows) *n_pricing_rows = h->h_st.n_rows;
  if (n_dense_rows) *n_dense_rows = h->h_st.n_dense;
  return LPAMD_OK;
}

int lpamd_rev_trace(lpamd_rev_engine* h, int64_t* te, int64_t* tl, int64_t cap, int64_t* n) {
  CHECK_REV(h);
  if (!h->started) { *n = 0; return LPAMD_OK; }
  const int64_t k = std::min<int64_t>({cap, h->h_st.iter, h->trace_cap});
  std::vector<int64_t> t(2 * std::max<int64_t>(k, 1));
  if (k > 0) {
    HIP_TRY(hipMemcpyAsync(t.data(), h->trace, sizeof(int64_t) * 2 * k, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  for (int64_t i = 0; i < k; i++) { te[i] = t[2 * i]; tl[i] = t[2 * i + 1]; }
  *n = k;
  return LPAMD_OK;
}

int lpamd_rev_read_state(lpamd_rev_engine* h, double* Binv, double* BinvT, double* xB, double* cB, int64_t* basis) {
  CHECK_REV(h);
  if (!h->started) return set_err(LPAMD_E_STATE, "rev_read_state: rev_solve has not run");
  const RevShape& sh = h->sh;
  const size_t row = sizeof(double) * sh.m;
  std::vector<double> tmp;
  double* bt = BinvT;
  if (Binv && !BinvT) {
    tmp.resize((size_t)sh.m * sh.m);
    bt = tmp.data();
  }
  if (bt) HIP_TRY(hipMemcpy2DAsync(bt, row, h->BinvT, sizeof(double) * sh.ldb, row, sh.m, hipMemcpyDeviceToHost, h->stream));
  if (xB) HIP_TRY(hipMemcpyAsync(xB, h->xB, row, hipMemcpyDeviceToHost, h->stream));
  if (cB) HIP_TRY(hipMemcpyAsync(cB, h->cB, row, hipMemcpyDeviceToHost, h->stream));
  if (basis) HIP_TRY(hipMemcpyAsync(basis, h->basis, sizeof(int64_t) * sh.m, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (Binv)
    for (int64_t i = 0; i < sh.m; i++)
      for (int64_t k = 0; k < sh.m; k++) Binv[i * sh.m + k] = bt[k * sh.m + i];
  return LPAMD_OK;
}

}  // extern "C"
