// Step j >= 1 of a pass of MAX_REDUCTION pivots on one GPU: everything behind the bounds in ONE launch.
//
// k_prb_first -> k_prb_list -> k_prb_eval -> k_prb_pick -> k_chain_finalize are five dependent launches of kernels that
// do microseconds of work each (45 of the 95 us a step costs at m=32768 n=65536, two thirds of a step at m=4096
// n=8192).  Here the workgroups stay resident and hand each other what the launch boundaries handed over, as tagged
// records (see lpamd_fused_kernels.hip.h), and the evaluation is organised so that ONE hand-over is the usual case:
//
//   1. every workgroup picks, from what k_prb_bounds left per workgroup (its candidate with the largest bound, and its
//      second largest bound), the kTailCols candidates with the largest bounds, and T = the largest bound among all
//      OTHER candidates (the next record in line, or the second bound of a chosen record's workgroup);
//   2. those columns are evaluated exactly, in row slices across all workgroups; every workgroup publishes the result
//      of its slice and folds all of them;
//   3. s = the largest exact score.  If s > T no other column can win or tie (its score is at most its bound), and the
//      reference's fold over the evaluated columns (lp_simplex_tableau.cc:310-325 / 353-362) names the pivot;
//      otherwise every workgroup evaluates, whole columns at a time, the candidates of its share of the columns whose
//      bound reaches s, publishes its best, and a second fold over those records joins in.  Which columns are
//      evaluated never changes the result, only how soon it is known: every candidate whose bound reaches the winning
//      score has been evaluated exactly when the fold is taken, as with k_prb_list.
//   4. the set-up of the step (reference lp_simplex_tableau.cc:10-44), by the same row / column workgroups as
//      k_chain_finalize<row given, column rebuilt>.
//
// Every workgroup publishes a record in step 2 after it has read the state that decides whether the step exists, and
// nobody passes step 2 without everybody's record: the state is written (pivot count, step record) only after all
// workgroups have read it, however late one of them starts.  "No candidate" and "no eligible row in any candidate" end
// the chain here; the first pricing of the next pass finds them again and reports them.
#pragma once
#include "lpamd_prbound_kernels.hip.h"
#include "lpamd_fused_kernels.hip.h"

namespace lpamd {

constexpr int kTailCols = 4;              // columns evaluated up front: one wavefront of a workgroup folds the slices of each
constexpr int kTailEmit = kTailCols + 1;  // records a wavefront passes on: the next in line bounds everything left out
constexpr int kTailEvalWords = 3;         // ratio (2), row (1)
constexpr int kTailBackWords = 6;         // score (2), column (1), row (1), objective entry of the column (2)
constexpr int kTailMaxGroups = 512;       // workgroups of a launch: their records are polled by 256 threads, 6 / 12 words each
constexpr int kTailMaxSlices = 128;

struct TailPick {          // the pivot a fold names
  double score;
  int32_t col;             // -1: no evaluated column has an eligible row
  int64_t row;
};

// kRec: records of k_prb_bounds per thread (n_top <= kRec * 256)
// kShard: column shards.  The pivot the fold names is this rank's CANDIDATE, not yet the step's pivot: instead of the set-up
// (4.) the launch packs it for the exchange, the record into slot_cand and the column, rebuilt through the steps so far,
// into slot_col (what k_prb_pick and k_prb_export do after k_prb_first / list / eval: five launches in one); "no candidate"
// travels as gcol = -1, and the set-up follows the exchange (k_chain_finalize from the slots).
template <int kRec, bool kShard>
__global__ __launch_bounds__(kPartialThreads, 2) void k_prb_tail(
    Tab T, Shape sh, PivotState* __restrict__ st, ChainStep* __restrict__ steps, int j, int n_rb,
    double* __restrict__ colp, double* __restrict__ rowp, int64_t rstride, double* __restrict__ rhsv,
    double* __restrict__ obj, const uint8_t* __restrict__ is_basic, const int64_t* __restrict__ basis,
    int64_t* __restrict__ trace, int64_t trace_cap,
    const double* __restrict__ top_u1, const int32_t* __restrict__ top_l1, const double* __restrict__ top_u2, int n_top,
    const double* __restrict__ ub, const uint8_t* __restrict__ ucand, PrbInfo* __restrict__ info,
    uint64_t* __restrict__ rec_eval, uint64_t* __restrict__ rec_back, uint32_t epoch,
    Candidate* __restrict__ slot_cand, double* __restrict__ slot_col) {
  constexpr int kThreads = kPartialThreads;
  constexpr int kWaves = kThreads / kWave;
  static_assert(kWaves == kTailCols, "one wavefront per column evaluated up front");
  __shared__ double s_v[kWaves];
  __shared__ int64_t s_i[kWaves];
  __shared__ uint64_t s_tkey[kWaves * kTailEmit];
  __shared__ uint32_t s_tidx[kWaves * kTailEmit];
  __shared__ int32_t s_tl[kWaves * kTailEmit];
  __shared__ double s_tu2[kWaves * kTailEmit];
  __shared__ int32_t s_col[kTailCols];
  __shared__ double s_cobj[kTailCols], s_cu2[kTailCols], s_next;
  __shared__ int s_ncols, s_nvalid;
  __shared__ double s_mv[kTailCols];
  __shared__ int64_t s_mi[kTailCols];
  __shared__ unsigned long long s_mask[kWaves];
  __shared__ int64_t s_row;
  __shared__ uint32_t s_words[12 * kThreads];
  __shared__ double s_ce;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int G = (int)gridDim.x, w = (int)blockIdx.x;
  const int my_set = w % kFusedReplicas;
  const int64_t eval_region = fused_region_words(G, kTailEvalWords), back_region = fused_region_words(G, kTailBackWords);
  const uint32_t tag_eval = (epoch << 8) | 1u, tag_back = (epoch << 8) | 2u;
  const int ncl = sh.nloc + 1;
  LPAMD_CLK(j, 2, 0, w == 0);
  LPAMD_CLK(j, 3, 0, w == n_rb);
  // ---- requested before the first wait: the state that decides whether the step exists, the records of the steps so
  // far (one per lane), this thread's share of the bounds' records
  int32_t q_in = -1, q_out = -1;
  int64_t q_e = -1, q_r = -1;
  if (lane == 0) { q_in = st->l_e; q_out = st->l_d; q_e = st->e; q_r = st->r; }
  else if (lane < j) { q_in = steps[lane].l_in; q_out = steps[lane].l_out; q_e = steps[lane].e; q_r = steps[lane].r; }
  const int64_t it = st->iter;
  uint64_t keys[kRec];
  int32_t rl[kRec];
  double ru2[kRec];
#pragma unroll
  for (int q = 0; q < kRec; q++) {
    const int k = tid + q * kThreads;
    const bool in = k < n_top;
    rl[q] = in ? top_l1[k] : -1;
    const double u1 = in ? top_u1[k] : 0.0;
    ru2[q] = in ? top_u2[k] : -DBL_MAX;
    keys[q] = rl[q] >= 0 ? arg_key<true>(u1, 0) : 0;
  }
  auto no_candidate = [&]() {              // kShard: what the other ranks read when this one has nothing to offer
    if (kShard && blockIdx.x == 0 && threadIdx.x == 0) { Candidate c; c.key = -DBL_MAX; c.gcol = -1; c.row = -1; c.ce = 0.0; *slot_cand = c; }
  };
  if (!prb_step_alive(st, steps, j)) { no_candidate(); return; }
  const double* __restrict__ rhs = prb_rhs(T, sh, rhsv, j);

  LPAMD_CLK(j, 2, 1, w == 0);
  LPAMD_CLK(j, 3, 1, w == n_rb);
  // ---- 1. the kTailCols largest bounds.  Every wavefront passes on its kTailEmit best records (larger bound, lower
  // record on ties); the best kTailEmit of those are the best of all.
#pragma unroll 1
  for (int r = 0; r < kTailEmit; r++) {
    uint64_t bk = 0;
    uint32_t bx = kNoIndex;
#pragma unroll
    for (int q = 0; q < kRec; q++)
      if (keys[q] > bk) { bk = keys[q]; bx = (uint32_t)(tid + q * kThreads); }     // a thread's records ascend with q
    uint64_t wk = bk;
    uint32_t wx = bx;
    wave_best(wk, wx);
#pragma unroll
    for (int q = 0; q < kRec; q++) {
      if (wk != 0 && (uint32_t)(tid + q * kThreads) == wx) {       // the owner hands the record over and retires it
        s_tkey[wave * kTailEmit + r] = wk; s_tidx[wave * kTailEmit + r] = wx;
        s_tl[wave * kTailEmit + r] = rl[q]; s_tu2[wave * kTailEmit + r] = ru2[q];
        keys[q] = 0;
      }
    }
    if (wk == 0 && lane == 0) { s_tkey[wave * kTailEmit + r] = 0; s_tidx[wave * kTailEmit + r] = kNoIndex; s_tl[wave * kTailEmit + r] = -1; s_tu2[wave * kTailEmit + r] = -DBL_MAX; }
  }
  __syncthreads();
  if (wave == 0) {
    constexpr int kAll = kWaves * kTailEmit;
    const bool in = lane < kAll;
    const uint64_t mk = in ? s_tkey[lane] : 0;
    const uint32_t mx = in ? s_tidx[lane] : kNoIndex;
    int rank = 0;
    for (int q = 0; q < kAll; q++) {
      const uint64_t ok = s_tkey[q];
      const uint32_t ox = s_tidx[q];
      rank += (ok > mk || (ok == mk && ox < mx)) ? 1 : 0;
    }
    const bool valid = mk != 0;
    const int n_valid = __popcll(__ballot(valid));
    if (valid && rank < kTailCols) {
      const int32_t l = s_tl[lane];
      s_col[rank] = l;
      s_cu2[rank] = s_tu2[lane];
      s_cobj[rank] = obj[l];             // before this workgroup's record: the set-up overwrites the objective row
    }
    if (valid && rank == kTailCols) s_next = arg_value<true>(mk);
    if (lane == 0) { s_nvalid = n_valid; s_ncols = min(n_valid, kTailCols); }
  }
  __syncthreads();
  const int n_cols = s_ncols;
  if (n_cols == 0) { no_candidate(); return; }   // no candidate column: optimal after step j-1
  double t_rest = s_nvalid > kTailCols ? s_next : -DBL_MAX;    // T: the largest bound among the candidates left out
  for (int c = 0; c < n_cols; c++) t_rest = fmax(t_rest, s_cu2[c]);
  const bool none_left = t_rest == -DBL_MAX;

  LPAMD_CLK(j, 2, 2, w == 0);
  LPAMD_CLK(j, 3, 2, w == n_rb);
  // ---- 2. exact evaluation in row slices, one (column, slice) per workgroup
  const int n_slices = max(1, min(G / n_cols, kTailMaxSlices));
  const int n_items = n_cols * n_slices;
  {
    double bv = DBL_MAX;
    int64_t bi = -1;
    if (w < n_items)
      prb_column_min(T, sh, s_col[w / n_slices], j, colp, rowp, rstride, rhs, (int64_t)(w % n_slices) * kThreads + tid,
                     (int64_t)n_slices * kThreads, bv, bi);
    block_arg<false, kThreads>(bv, bi, s_v, s_i);
    wait_stores();                       // every load above is back (the objective entries of the columns among them)
  LPAMD_CLK(j, 2, 3, w == 0);
  LPAMD_CLK(j, 3, 3, w == n_rb);
    if (tid < kFusedReplicas * kTailEvalWords) {
      const int x = tid % kTailEvalWords;
      const uint64_t vb = (uint64_t)__double_as_longlong(bv);
      rec_put(rec_eval + (int64_t)(tid / kTailEvalWords) * eval_region + (int64_t)w * kTailEvalWords + x,
              x == 0 ? (uint32_t)vb : x == 1 ? (uint32_t)(vb >> 32) : (bi < 0 ? kFusedNoRow : (uint32_t)bi), tag_eval);
    }
  }
  {
    const bool ok = poll_words<6>(rec_eval + (int64_t)my_set * eval_region, G * kTailEvalWords, tag_eval, tid, kThreads, s_words);
    if (__syncthreads_or(!ok)) { if (tid == 0) st->status = ST_ERROR; return; }
  }
  LPAMD_CLK(j, 2, 4, w == 0);
  LPAMD_CLK(j, 3, 4, w == n_rb);
  if (wave < n_cols) {                   // wavefront c: smallest ratio of column c over its slices, lowest row on ties
    double mv = DBL_MAX;
    int64_t mi = -1;
    for (int q = lane; q < n_slices; q += kWave) {
      const uint32_t* d = s_words + (wave * n_slices + q) * kTailEvalWords;
      if (d[2] != kFusedNoRow) take_min(mv, mi, rec_f64(d[0], d[1]), (int64_t)d[2]);
    }
    wave_arg<false>(mv, mi);
    if (lane == 0) { s_mv[wave] = mv; s_mi[wave] = mi; }
  }
  __syncthreads();
  // ---- 3. the fold over the evaluated columns: largest score among columns that have a row, lowest column on ties
  TailPick pick;
  pick.score = -DBL_MAX; pick.col = -1; pick.row = -1;
  for (int c = 0; c < n_cols; c++) {
    if (s_mi[c] < 0) continue;
    const double score = __dmul_rn(s_cobj[c], s_mv[c]);
    if (pick.col < 0 || score > pick.score || (score == pick.score && s_col[c] < pick.col)) { pick.score = score; pick.col = s_col[c]; pick.row = s_mi[c]; }
  }
  int64_t n_exact = n_cols;
  double back_ce = 0.0;
  bool from_back = false;
  const bool settled = pick.col >= 0 ? (none_left || pick.score > t_rest) : none_left;
  if (!settled) {
    // the candidates of this workgroup's share of the columns whose bound reaches the score so far, whole columns at a time
    const int per = (sh.nloc + G - 1) / G;
    const int l_begin = w * per, l_end = min(sh.nloc, l_begin + per);
    double wv = -DBL_MAX, wce = 0.0;
    int64_t wl = -1, wr = -1;
    for (int l0 = l_begin; l0 < l_end; l0 += kThreads) {
      const int l = l0 + tid;
      bool take = l < l_end && ucand[l] != 0 && (pick.col < 0 || ub[l] >= pick.score);
      if (pick.col >= 0)                 // evaluated already
        for (int c = 0; c < n_cols; c++) take = take && s_col[c] != l;
      const unsigned long long mask = __ballot(take);
      __syncthreads();
      if (lane == 0) s_mask[wave] = mask;
      __syncthreads();
      for (int wq = 0; wq < kWaves; wq++) {
        unsigned long long todo = s_mask[wq];
        while (todo) {
          const int b = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          const int lc = l0 + wq * kWave + b;
          double bv = DBL_MAX;
          int64_t bi = -1;
          prb_column_min(T, sh, lc, j, colp, rowp, rstride, rhs, tid, kThreads, bv, bi);
          block_arg<false, kThreads>(bv, bi, s_v, s_i);
          if (bi >= 0) {
            const double c_lc = obj[lc];
            const double score = __dmul_rn(c_lc, bv);
            if (wl < 0 || score > wv || (score == wv && lc < wl)) { wv = score; wl = lc; wr = bi; wce = c_lc; }
          }
        }
      }
    }
    wait_stores();
    __syncthreads();
    if (tid < kFusedReplicas * kTailBackWords) {
      const int x = tid % kTailBackWords;
      const uint64_t vb = (uint64_t)__double_as_longlong(wv), cb2 = (uint64_t)__double_as_longlong(wce);
      rec_put(rec_back + (int64_t)(tid / kTailBackWords) * back_region + (int64_t)w * kTailBackWords + x,
              x == 0 ? (uint32_t)vb : x == 1 ? (uint32_t)(vb >> 32) : x == 2 ? (wl < 0 ? kFusedNoRow : (uint32_t)wl)
              : x == 3 ? (uint32_t)(wr < 0 ? 0 : wr) : x == 4 ? (uint32_t)cb2 : (uint32_t)(cb2 >> 32), tag_back);
    }
    if (w == n_rb && tid == 0) info->tail_back += 1;
    const bool ok = poll_words<12>(rec_back + (int64_t)my_set * back_region, G * kTailBackWords, tag_back, tid, kThreads, s_words);
    if (__syncthreads_or(!ok)) { if (tid == 0) st->status = ST_ERROR; return; }
    double fv = -DBL_MAX;
    int64_t fl = -1;
    for (int k = tid; k < G; k += kThreads) {
      const uint32_t* d = s_words + k * kTailBackWords;
      if (d[2] != kFusedNoRow) take_max(fv, fl, rec_f64(d[0], d[1]), (int64_t)d[2]);
    }
    block_arg<true, kThreads>(fv, fl, s_v, s_i);
    if (fl >= 0) {
      for (int k = tid; k < G; k += kThreads) {
        const uint32_t* d = s_words + k * kTailBackWords;
        if (d[2] != kFusedNoRow && (int64_t)d[2] == fl) { s_row = (int64_t)d[3]; s_ce = rec_f64(d[4], d[5]); }     // one record names a column
      }
    }
    __syncthreads();
    if (fl >= 0 && (pick.col < 0 || fv > pick.score || (fv == pick.score && fl < pick.col))) { pick.score = fv; pick.col = (int32_t)fl; pick.row = s_row; back_ce = s_ce; from_back = true; }
    n_exact += 1;     // counted as one more evaluation round
  }
  if (pick.col < 0) { no_candidate(); return; }   // no candidate has an eligible row: unbounded after step j-1 (the next pass's first pricing reports it)

  LPAMD_CLK(j, 2, 5, w == 0);
  LPAMD_CLK(j, 3, 5, w == n_rb);
  // ---- 4. the set-up of step j for entering column pick.col and pivot row pick.row
  const int l_e2 = pick.col;
  const int64_t r2 = pick.row;
  const int64_t e2 = gcol_of((int64_t)l_e2 * sh.world + sh.rank, sh.N);
  const bool row_block = w < n_rb;
  const int cb = w - n_rb, n_cb = G - n_rb;
  const double* __restrict__ ecol = T.col(l_e2);
  const double* __restrict__ rhs_prev = rhs;
  // the objective entry of the entering column, read before anybody's set-up rewrites the objective row
  double ce2 = back_ce;
  if (!from_back)
    for (int c = 0; c < n_cols; c++) if (s_col[c] == l_e2) ce2 = s_cobj[c];
  if (kShard) {
    // ---- 4'. this rank's candidate, packed for the exchange
    if (w == 0 && tid == 0) {
      Candidate c;
      c.key = pick.score; c.gcol = e2; c.row = r2; c.ce = ce2;
      *slot_cand = c;
      info->priced += 1;
      info->listed += n_exact - 1;
    }
    if (row_block) {
      const int64_t stride = (int64_t)n_rb * kThreads;
      for (int64_t i = (int64_t)w * kThreads + tid; i < sh.lda; i += stride)
        slot_col[i] = i < sh.m ? chain_apply(ecol[i], j, colp + i, sh.lda, rowp + l_e2, rstride) : 0.0;
    }
    (void)cb; (void)n_cb; (void)rhs_prev; (void)it;
    return;
  }
  const double a2 = chain_apply(ecol[r2], j, colp + r2, sh.lda, rowp + l_e2, rstride);
  const double f2 = -1.0 / a2;
  const double g2 = __dmul_rn(ce2, f2);
  if (row_block) {
    double* __restrict__ cj = colp + (int64_t)j * sh.lda;
    double* __restrict__ rhs_now = rhsv + (int64_t)j * sh.lda;
    const double rp_rhs = __dmul_rn(rhs_prev[r2], f2);   // row'_j of the right-hand-side column
    const int64_t stride = (int64_t)n_rb * kThreads;
    bool big = false;
    for (int64_t i = (int64_t)w * kThreads + tid; i < sh.m; i += stride) {
      const double ci = chain_apply(ecol[i], j, colp + i, sh.lda, rowp + l_e2, rstride);
      const double rhs_i = rhs_prev[i];
      const double cpi = (i == r2) ? __dadd_rn(1.0, a2) : ci;   // lp_simplex_tableau.cc:41-44
      big = big || !(fabs(cpi) < kDirtyBound);
      cj[i] = cpi;
      rhs_now[i] = rp_rhs != 0.0 ? flushz(__dadd_rn(rhs_i, __dmul_rn(cpi, rp_rhs))) : rhs_i;
    }
    if (big) st->dirty_step = j + 1;   // PivotState::dirty
    LPAMD_CLK(j, 2, 7, w == 0);
    return;
  }
  // the basic variable of row r2 after steps 0..j-1: the last step so far that pivoted on this row, else what the pass before left
  int64_t d2 = basis[r2];
  {
    const unsigned long long hit = __ballot(lane < j && q_r == r2);
    if (hit) d2 = readlane_i64(q_e, 63 - __clzll((long long)hit));
  }
  int l_d2 = -1;
  if (d2 >= 0 && d2 != sh.N) {
    const int64_t o = ordinal_of(d2, sh.N);
    if (ord_rank(o, sh.world) == sh.rank) l_d2 = ord_local(o, sh.world);
  }
  const double* __restrict__ c_r = colp + r2;       // col'_s[r2] at c_r[s * lda]
  LPAMD_CLK(j, 3, 6, w == n_rb && l_d2 != -7);
  for (int l = cb * kThreads + tid; l < ncl; l += n_cb * kThreads) {
    const double t = chain_apply(T.col(l)[r2], j, c_r, sh.lda, rowp + l, rstride);
    const double rp2 = __dmul_rn(t, f2);
    double o = flushz(__dadd_rn(obj[l], __dmul_rn(t, g2)));   // :36
    if (l == l_e2) o = 0.0;                                     // :37
    obj[l] = o;
    rowp[(int64_t)j * rstride + l] = rp2;
  }
  if (cb == 0 && tid == 0) {
    ChainStep z;
    z.done = 1; z.slot = 0; z.l_in = l_e2; z.l_out = l_d2;
    z.e = e2; z.r = r2; z.d = d2; z.a = a2; z.ce = ce2; z.f = f2; z.g = g2;
    steps[j] = z;
    if (it < trace_cap) { trace[2 * it] = e2; trace[2 * it + 1] = r2; }
    st->iter = it + 1;
    info->priced += 1;
    info->listed += n_exact - 1;
  }
  LPAMD_CLK(j, 3, 7, w == n_rb);
  (void)q_in; (void)q_out; (void)is_basic;
}

}  // namespace lpamd
