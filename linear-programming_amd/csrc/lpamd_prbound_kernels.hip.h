// MAX_REDUCTION pricing without a pass over the tableau (reference lp_simplex_tableau.cc:235-374).
//
// The reference scores every candidate column j with c_j * min_i b_i / (-T_ij) over its eligible
// rows and takes the largest score (lowest column on ties).  Done literally that is a read of the
// whole tableau per pivot, after the update of the pivot before: one pass per pivot, no chains.
// The same decision can be reached from far fewer entries:
//   - for ANY set R of rows, u_j = c_j * min_{i in R, eligible} b_i / (-T_ij) is an upper bound of
//     column j's score (a minimum over fewer rows; the rounded division and the rounded product with
//     c_j > 0 are monotone, so the bound holds for the computed values, not just the exact ones);
//   - the exact score s1 of ONE column is a lower bound of the winning score;
//   - so every column with u_j < s1 is out, whatever its exact score is.
// R = the rows with the smallest right-hand side: that is where the minimum ratio of almost every
// column sits, so the bounds are tight and hardly any column survives: the column with the largest
// bound is evaluated exactly (s1), then the few columns with u_j >= s1, and the fold over those
// exact results applies the reference's rule.  Cost per pivot: |R| entries per candidate column
// instead of m.  Every entry is taken as the pivots set up so far will leave it (chain_apply), so
// these pivots chain like the MAX_COST ones: one pass over the tableau per chain.
// When no bound helps (first iteration of a degenerate start, columns without an eligible row in R:
// u_j = +inf) the survivors are simply all candidates and the scheme degrades to the full evaluation.
#pragma once

namespace lpamd {

constexpr int kPrbRows = 32;
constexpr int kPrbRecent = 8;    // rows of the latest pivots that are always part of R

struct PrbInfo {
  int32_t n_rows;          // rows of R in use
  int32_t n_list;          // columns to evaluate exactly besides l1
  int32_t l1;              // local column with the largest bound (-1: no candidate column)
  int32_t first;           // lowest candidate column (the reference reports it when no candidate has a row)
  double s1;               // exact score of l1 (-DBL_MAX: it has no eligible row)
  int32_t rows[kPrbRows];
  double b[kPrbRows];      // right-hand side at those rows, as the pivots set up so far leave it
  int64_t priced, listed;  // running totals: pricings done, columns they evaluated exactly besides l1
  int64_t tail_back;       // pricings of k_prb_tail that needed the second round
  // Row cache: R changes by a few rows from one pricing of a pass to the next, and gathering a row of a column-major
  // tableau costs a 64-byte sector per column.  Every row of R is therefore gathered once, into a slot of a
  // row-major side buffer, as the steps set up until then leave it (jf of them); the pricings after that read the
  // slot (coalesced) and apply the steps since.  The slots are emptied at step 0: the pass has changed the tableau.
  int32_t slot[kPrbRows];      // slot of row k of R
  int32_t jf[kPrbRows];        // steps its slot already holds
  int32_t slot_row[kPrbRows];  // row a slot holds (-1: none)
  int32_t slot_jf[kPrbRows];
  // The slots outlive the pass when nothing but that pass has changed the tableau since: the first pricing of the next
  // pass then brings them through the steps of the pass they still lack (its col' and row' vectors are intact until
  // the set-up of the new pass's first pivot) instead of gathering every row again (a 64-byte sector per column and
  // row: 140 us per pass at m=32768 n=65536).  cache_iter: the pivot count at which slots + steps describe the tableau;
  // steps_seen: steps of the pass the slots count in; both written by every k_prb_rows launch that goes ahead.
  int64_t cache_iter;
  int32_t steps_seen;
  int32_t j_apply;             // steps k_prb_bounds applies in this pricing: j, or the steps of the pass before (carried)
  int32_t carried;             // this pricing (the first of a pass) works from the slots of the pass before
  int32_t pad_;
};

// the slots of the pass before are good for the first pricing of this pass (see PrbInfo)
__device__ __forceinline__ bool prb_carried(const PivotState* st, const PrbInfo* info, const PassInfo* pass, int j, int carry) {
  if (j != 0 || !carry) return false;
  const int64_t iter = st->iter, cache_iter = info->cache_iter;
  const int32_t seen = info->steps_seen, applied = pass->n_steps;
  return cache_iter == iter && seen >= 1 && seen == applied;
}

// true while step j of the pass may still be set up (the same tests k_chain_ratio makes)
__device__ __forceinline__ bool prb_step_alive(const PivotState* st, const ChainStep* steps, int j) {
  // requested together: one round trip instead of three
  const int32_t status = st->status;
  const int32_t prev_done = j > 1 ? steps[j - 1].done : 1;
  const int64_t iter = st->iter, max_iter = st->max_iter;
  const int32_t dirty_step = st->dirty_step;
  if (status != ST_RUNNING) return false;
  if (!prev_done) return false;                     // the chain ended before this step
  if (dirty_step != 0 && dirty_step - 1 < j) return false;   // a step before this one ends the chain (PivotState::dirty)
  if (j >= 1 && iter >= max_iter) return false;     // the budget ends with step j-1 (step 0 reports LIMIT itself)
  return true;
}

// right-hand side as steps 0..j-1 leave it
__device__ __forceinline__ const double* prb_rhs(const Tab& T, const Shape& sh, const double* rhsv, int j) {
  return j == 0 ? T.col(sh.nloc) : rhsv + (int64_t)(j - 1) * sh.lda;
}

// 1. The rows of R, in one launch of a single workgroup.  Exactly the rows with the smallest right-hand side
// matter (a row of rank 3 that is missed costs a full evaluation of thousands of columns when it holds the minimum
// of the column with the largest bound), so the selection is exact: (a) the minimum of every one of 128 segments of
// the rows; (b) the kPrbRows-th smallest of those minima is a threshold that at least kPrbRows rows meet, and
// hardly more (about 40): all rows that meet it are collected in LDS; (c) the rows of the latest pivots come first
// (a pivot scales its row by 1 / |pivot element|, and a row of large entries holds the minimum ratio of many
// columns without having a small right-hand side), then the collected rows in the order of their right-hand side
// (lowest row first on equal values) until R is full.  The right-hand side is m doubles that sit in the L2: one
// workgroup reads it twice in less time than the three dependent launches cost that used to share this work.
// The kernel also settles the budget at step 0, keeps the cache slots of the rows (see PrbInfo) and leaves the col'
// entries of the steps so far at the rows of R in prb_c.
constexpr int kPrbCollect = 512;   // rows (b) can hold; more than that met the threshold only for short columns
constexpr int kPrbSegments = 128;
constexpr int kPrbRowThreads = 1024;
template <int kKeep>   // values of the right-hand side a thread keeps in registers (segments of at most 8 * kKeep rows); 0: none, see (a)
__global__ __launch_bounds__(kPrbRowThreads) void k_prb_rows(
    Tab T, Shape sh, PivotState* __restrict__ st, ChainStep* __restrict__ steps, int j, int depth,
    const double* __restrict__ rhsv, const double* __restrict__ colp, int max_rows, int renew,
    const int64_t* __restrict__ trace, int64_t trace_cap, PrbInfo* __restrict__ info, double* __restrict__ prb_c,
    const PassInfo* __restrict__ pass, int carry) {
  __shared__ double s_seg[kPrbSegments];
  __shared__ double s_tau;
  __shared__ int s_go, s_ncoll, s_nrecent, s_nfresh, s_japply;
  __shared__ int64_t s_iter;
  __shared__ int32_t s_crow[kPrbCollect];
  __shared__ double s_cb[kPrbCollect];
  __shared__ uint8_t s_fresh[kPrbCollect];
  __shared__ int32_t s_R[kPrbRows];
  __shared__ int32_t s_slot_row[kPrbRows], s_slot_jf[kPrbRows];
  __shared__ int64_t s_recent[kPrbRecent];
  const int tid = threadIdx.x;
  LPAMD_CLK(j, 0, 0, true);
  // everything this kernel reads before it can start, requested by different threads at once (one round trip)
  if (tid < kPrbRecent) {
    const int64_t tq = st->iter - 1 - tid;
    s_recent[tid] = (tq >= 0 && tq < trace_cap) ? trace[2 * tq + 1] : -1;
  } else if (tid >= kWave && tid < kWave + kPrbRows) {
    const int q = tid - kWave;
    const int32_t row_q = info->slot_row[q];
    s_slot_jf[q] = info->slot_jf[q];
    s_slot_row[q] = (j == 0 && !prb_carried(st, info, pass, j, carry)) ? -1 : row_q;   // otherwise step 0 starts from empty slots
  } else if (tid == 2 * kWave) {
    int go = 1;
    if (st->status != ST_RUNNING) {
      go = 0;
    } else {
      // the step records of the pass before are cleared even when the budget ends with step 0 (k_chain_list counts them)
      if (j == 1) for (int s = 1; s < depth; s++) { steps[s].done = 0; steps[s].e = -1; }
      if (j == 0 && st->dirty == 1) { st->status = ST_DIRTY; go = 0; }
      else if (j == 0 && st->iter >= st->max_iter) { st->status = ST_LIMIT; go = 0; }   // as k_pr_fold does
      else if (!prb_step_alive(st, steps, j)) go = 0;
    }
    s_go = go;
    s_iter = st->iter;
    s_japply = prb_carried(st, info, pass, j, carry) ? info->steps_seen : j;
    s_ncoll = 0;
    s_nfresh = 0;
    s_tau = DBL_MAX;            // fewer minima than rows wanted: every row qualifies
  }
  __syncthreads();
  if (!s_go) return;
  LPAMD_CLK(j, 0, 1, true);
  if (tid == 0) {               // the rows of the latest pivots, the latest first, each once
    int n_recent = 0;
    for (int q = 0; q < kPrbRecent && n_recent < max_rows; q++) {
      const int64_t row = s_recent[q];
      if (row < 0) break;
      bool dup = false;
      for (int k = 0; k < n_recent; k++) dup = dup || s_R[k] == (int32_t)row;
      if (!dup) s_R[n_recent++] = (int32_t)row;
    }
    s_nrecent = n_recent;
  }
  const double* __restrict__ rhs = prb_rhs(T, sh, rhsv, j);
  // (a) eight threads per segment, all loads of a thread in flight together.  Up to 256 rows per segment (m <= 32768)
  // a thread's values (kKeep of them: 4 for m <= 4096 .. 32) stay in registers and the rows that meet the threshold are
  // picked from them in (b); longer columns are read a second time.
  const int m32 = (int)sh.m;     // rows fit 31 bits (checked at create): 32-bit index arithmetic, a third of the instructions
  const int n_seg = min(kPrbSegments, m32);
  const int seg = (m32 + n_seg - 1) / n_seg;
  constexpr int kPerSeg = kPrbRowThreads / kPrbSegments;
  constexpr int kLoads = 16;   // loads in flight per thread otherwise (unconditional: the index is clamped, the value masked)
  constexpr bool keep = kKeep > 0;
  const int sidx = tid / kPerSeg, sub = tid % kPerSeg;
  const int i0 = sidx * seg + sub, i1 = sidx < n_seg ? min((sidx + 1) * seg, m32) : 0;
  double kept[keep ? kKeep : 1];
  double mymin = DBL_MAX;
  if constexpr (keep) {
    if (i0 < i1) {
#pragma unroll
      for (int u = 0; u < kKeep; u++) kept[u] = rhs[min(i0 + u * kPerSeg, i1 - 1)];   // a row read twice does not change a minimum
#pragma unroll
      for (int u = 0; u < kKeep; u++) mymin = fmin(mymin, kept[u]);
    }
  } else if (sidx < n_seg) {
    for (int i = i0; i < i1; i += kPerSeg * kLoads) {
      double v[kLoads];
#pragma unroll
      for (int u = 0; u < kLoads; u++) v[u] = rhs[min(i + u * kPerSeg, i1 - 1)];
#pragma unroll
      for (int u = 0; u < kLoads; u++) mymin = fmin(mymin, v[u]);
    }
  }
  {
    double bv = mymin;
#pragma unroll
    for (int off = kPerSeg / 2; off > 0; off >>= 1) bv = fmin(bv, __shfl_xor(bv, off, kWave));
    if (sub == 0 && sidx < n_seg) s_seg[sidx] = bv;
  }
  __syncthreads();
  LPAMD_CLK(j, 0, 2, true);
  // (b) the threshold, then the rows that meet it
  {   // rank of every segment minimum among the minima: the eight threads of a segment count a share each
    const double mine = sidx < n_seg ? s_seg[sidx] : 0.0;
    int rank = 0;
    if (sidx < n_seg) {
#pragma unroll 4
      for (int q = sub; q < n_seg; q += kPerSeg) {
        const double other = s_seg[q];
        rank += (other < mine || (other == mine && q < sidx)) ? 1 : 0;
      }
    }
    rank += dpp_i32<kDppXor1>(rank);
    rank += dpp_i32<kDppXor2>(rank);
    rank += dpp_i32<kDppHalfMirror>(rank);     // the eight lanes of a segment sit in one half row
    if (sub == 0 && sidx < n_seg && rank == max_rows - 1) s_tau = mine;
  }
  __syncthreads();
  LPAMD_CLK(j, 0, 3, true);
  const double tau = s_tau;
  if constexpr (keep) {
    if (mymin <= tau) {        // a few threads only
#pragma unroll
      for (int u = 0; u < kKeep; u++) {
        const int i = i0 + u * kPerSeg;
        if (i < i1 && kept[u] <= tau) {
          const int k = atomicAdd(&s_ncoll, 1);
          if (k < kPrbCollect) { s_crow[k] = (int32_t)i; s_cb[k] = kept[u]; }
        }
      }
    }
  } else {
    for (int i = tid; i < m32; i += kPrbRowThreads * kLoads) {
      double v[kLoads];
#pragma unroll
      for (int u = 0; u < kLoads; u++) v[u] = rhs[min(i + u * kPrbRowThreads, m32 - 1)];
#pragma unroll
      for (int u = 0; u < kLoads; u++) {
        if (v[u] <= tau && i + u * kPrbRowThreads < m32) {
          const int k = atomicAdd(&s_ncoll, 1);
          if (k < kPrbCollect) { s_crow[k] = (int32_t)(i + u * kPrbRowThreads); s_cb[k] = v[u]; }
        }
      }
    }
  }
  __syncthreads();
  LPAMD_CLK(j, 0, 4, true);
  // (c) the collected rows that are not pivot rows already, ranked by (right-hand side, row)
  const int n_coll = min(s_ncoll, kPrbCollect), n_recent = s_nrecent;
  if (tid < n_coll) {
    bool fresh = true;
    for (int k = 0; k < n_recent; k++) fresh = fresh && s_R[k] != s_crow[tid];
    s_fresh[tid] = fresh ? 1 : 0;
    if (fresh) atomicAdd(&s_nfresh, 1);
  }
  __syncthreads();
  if (n_coll <= kPrbRowThreads / kPerSeg) {      // the usual case: eight threads share the count of a row
    const int c = tid / kPerSeg;
    const bool counted = c < n_coll && s_fresh[c];
    const double mv = counted ? s_cb[c] : 0.0;
    const int32_t mr = counted ? s_crow[c] : 0;
    int rank = 0;
    if (counted) {
#pragma unroll 4
      for (int q = tid % kPerSeg; q < n_coll; q += kPerSeg) {
        const double ob = s_cb[q];
        const int32_t orow = s_crow[q];
        const bool ofresh = s_fresh[q] != 0;
        rank += (ofresh && (ob < mv || (ob == mv && orow < mr))) ? 1 : 0;
      }
    }
    rank += dpp_i32<kDppXor1>(rank);
    rank += dpp_i32<kDppXor2>(rank);
    rank += dpp_i32<kDppHalfMirror>(rank);
    if (counted && tid % kPerSeg == 0 && n_recent + rank < max_rows) s_R[n_recent + rank] = mr;
  } else if (tid < n_coll && s_fresh[tid]) {
    const double mv = s_cb[tid];
    const int32_t mr = s_crow[tid];
    int rank = 0;
#pragma unroll 8
    for (int q = 0; q < n_coll; q++) {
      const double ob = s_cb[q];
      const int32_t orow = s_crow[q];
      const bool ofresh = s_fresh[q] != 0;
      rank += (ofresh && (ob < mv || (ob == mv && orow < mr))) ? 1 : 0;
    }
    if (n_recent + rank < max_rows) s_R[n_recent + rank] = mr;
  }
  __syncthreads();
  LPAMD_CLK(j, 0, 5, true);
  const int n_rows = min(max_rows, n_recent + s_nfresh);
  const int ja = s_japply;
  const bool carried = ja != j;      // only at step 0: the steps are those of the pass before
  for (int t = tid; t < n_rows * ja; t += kPrbRowThreads) {     // independent loads, one or two per thread
    const int k = t % n_rows, sidx = t / n_rows;
    prb_c[sidx * kPrbRows + k] = colp[(int64_t)sidx * sh.lda + s_R[k]];
  }
  LPAMD_CLK(j, 0, 6, true);
  if (tid >= kWave) return;
  // the first wavefront: lane k keeps row k of R
  const int lane = tid;
  const int64_t mine = lane < n_rows ? s_R[lane] : -1;
  if (lane < n_rows) { info->rows[lane] = (int32_t)mine; info->b[lane] = rhs[mine]; }
  // cache slots: a row that a slot holds keeps it; the others take the slots no row of R occupies, lowest first
  int my_slot = -1;
  if (lane < n_rows)
    for (int q = 0; q < kPrbRows; q++)
      if (s_slot_row[q] == (int32_t)mine) my_slot = q;
  const bool hit = lane < n_rows && my_slot >= 0, miss = lane < n_rows && my_slot < 0;
  unsigned used = hit ? 1u << my_slot : 0u;      // slots the rows of R occupy already: OR over the wavefront
  used |= (unsigned)dpp_i32<kDppXor1>((int)used);
  used |= (unsigned)dpp_i32<kDppXor2>((int)used);
  used |= (unsigned)dpp_i32<kDppHalfMirror>((int)used);
  used |= (unsigned)dpp_i32<kDppMirror>((int)used);
  used = (unsigned)__builtin_amdgcn_readlane((int)used, 0) | (unsigned)__builtin_amdgcn_readlane((int)used, 16) |
         (unsigned)__builtin_amdgcn_readlane((int)used, 32) | (unsigned)__builtin_amdgcn_readlane((int)used, 48);
  const unsigned long long missing = __ballot(miss);
  const int nth = __popcll(missing & ((1ull << lane) - 1ull));
  if (miss) {
    unsigned free_slots = ~used;
    for (int t = 0; t < nth; t++) free_slots &= free_slots - 1;
    my_slot = __ffs(free_slots) - 1;       // n_rows <= kPrbRows slots: there is one
    info->slot_row[my_slot] = (int32_t)mine;
    info->slot_jf[my_slot] = j;
  }
  if (lane < n_rows) { info->slot[lane] = my_slot; info->jf[lane] = hit ? s_slot_jf[my_slot] : ja; }
  // k_prb_bounds brings a slot that is `renew` steps behind up to date (same test there), and every slot when carried
  if (hit && (carried || j - s_slot_jf[my_slot] >= renew)) info->slot_jf[my_slot] = j;
  if (j == 0) {                              // slots no row of R holds are empty from here on
    unsigned taken = lane < n_rows ? 1u << my_slot : 0u;
    taken |= (unsigned)dpp_i32<kDppXor1>((int)taken);
    taken |= (unsigned)dpp_i32<kDppXor2>((int)taken);
    taken |= (unsigned)dpp_i32<kDppHalfMirror>((int)taken);
    taken |= (unsigned)dpp_i32<kDppMirror>((int)taken);
    taken = (unsigned)__builtin_amdgcn_readlane((int)taken, 0) | (unsigned)__builtin_amdgcn_readlane((int)taken, 16) |
            (unsigned)__builtin_amdgcn_readlane((int)taken, 32) | (unsigned)__builtin_amdgcn_readlane((int)taken, 48);
    if (lane < kPrbRows && !((taken >> lane) & 1u)) info->slot_row[lane] = -1;
  }
  if (lane == 0) {
    info->n_rows = n_rows; info->n_list = 0; info->l1 = -1; info->first = -1; info->s1 = -DBL_MAX;
    info->j_apply = ja; info->carried = carried ? 1 : 0;
    info->steps_seen = j + 1; info->cache_iter = s_iter + 1;     // as they will be once this step is set up
  }
  LPAMD_CLK(j, 0, 7, true);
}

// basis flag of local column l after the pivots set up so far (the pass kernel flips is_basic[] later).  q_in / q_out:
// lane s of the wavefront holds the local columns step s turned basic / non-basic (lane 0: pivot 0, from PivotState), so
// a step costs two lane reads instead of two scalar loads and a wait.
__device__ __forceinline__ bool prb_basic(int l, int j, const uint8_t* is_basic, int q_in, int q_out) {
  bool basic = is_basic[l] != 0;
#pragma unroll 1
  for (int s = 0; s < j; s++) {
    const int in_s = __builtin_amdgcn_readlane(q_in, s), out_s = __builtin_amdgcn_readlane(q_out, s);
    if (l == in_s) basic = true; else if (l == out_s) basic = false;
  }
  return basic;
}

// 2. bound u_l per candidate column from the rows of R; per-workgroup arg-max (lowest column on ties)
// and lowest candidate column.  The rows new to R in this pricing are gathered from the tableau here, for every
// local column (a 64-byte sector per column and row), brought through steps 0..j-1 and left in their cache slots.  A workgroup takes tiles of 64 columns; its four wavefronts share the rows of R
// (a quarter each, combined through LDS): one thread per column and all rows is a chain of a few thousand dependent
// instructions on a chip that this kernel fills with one wavefront per SIMD.
constexpr int kPrbGroups = kPartialThreads / kWave;
constexpr int kPrbPerGroup = kPrbRows / kPrbGroups;
__global__ __launch_bounds__(kPartialThreads) void k_prb_bounds(
    Tab T, Shape sh, const PivotState* __restrict__ st, const ChainStep* __restrict__ steps, int j,
    const double* __restrict__ obj, const uint8_t* __restrict__ is_basic,
    const double* __restrict__ rowp, int64_t rstride, const PrbInfo* __restrict__ info, const double* __restrict__ prb_c, int renew,
    double* __restrict__ cache,         // nullptr: the rows are gathered from the tableau itself
    double* __restrict__ ub, uint8_t* __restrict__ ucand,
    double* __restrict__ pv, int64_t* __restrict__ pi, int64_t* __restrict__ pfirst,
    double* __restrict__ top_u1 = nullptr, int32_t* __restrict__ top_l1 = nullptr, double* __restrict__ top_u2 = nullptr) {
  __shared__ double s_mn[kPrbGroups][kWave];
  const int lane = threadIdx.x & (kWave - 1);
  LPAMD_CLK(j, 1, 0, blockIdx.x == 0);
  int q_in = -1, q_out = -1;     // the records of the steps so far, one per lane
  if (lane == 0) { q_in = st->l_e; q_out = st->l_d; }
  else if (lane < j) { q_in = steps[lane].l_in; q_out = steps[lane].l_out; }
  // requested together with the state (one round trip): the rows of this wavefront's share of R
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int k0 = grp * kPrbPerGroup;
  const int n_rows = info->n_rows;
  const int ja = cache ? info->j_apply : j;      // steps applied to cached entries (the steps of the pass before when carried)
  const bool carried = cache && info->carried != 0;
  int32_t i_jf[kPrbPerGroup], i_row[kPrbPerGroup], i_slot[kPrbPerGroup];
  double i_b[kPrbPerGroup];
#pragma unroll
  for (int k = 0; k < kPrbPerGroup; k++) { i_jf[k] = info->jf[k0 + k]; i_row[k] = info->rows[k0 + k]; i_slot[k] = info->slot[k0 + k]; i_b[k] = info->b[k0 + k]; }
  if (!prb_step_alive(st, steps, j)) return;
  int jf_min = ja;
  int jfs[kPrbPerGroup], slots[kPrbPerGroup], rows[kPrbPerGroup];
  bool fresh[kPrbPerGroup];      // new to R in this pricing: gathered from the tableau here and left in its cache slot
  bool store[kPrbPerGroup];      // ... or its slot is `renew` steps and more behind: what this pricing computes goes back to the slot
  bool any_fresh = false;
  double bs[kPrbPerGroup];
#pragma unroll
  for (int k = 0; k < kPrbPerGroup; k++) {
    const bool live = k0 + k < n_rows;
    const int jf = live ? (cache ? i_jf[k] : 0) : ja;
    rows[k] = live ? i_row[k] : 0;
    fresh[k] = live && cache && jf == ja;         // slots filled by earlier pricings of the pass hold fewer steps
    store[k] = fresh[k] || (live && cache && (carried || ja - jf >= renew));
    jfs[k] = fresh[k] ? (carried ? ja : 0) : jf;  // a fresh row starts from the stored entry and takes every step of this pass
    slots[k] = live ? (cache ? i_slot[k] : rows[k]) : 0;   // slot of the cache, or the row itself
    bs[k] = live ? i_b[k] : 0.0;
    jf_min = min(jf_min, jfs[k]);
    any_fresh = any_fresh || store[k];
  }
  LPAMD_CLK(j, 1, 1, blockIdx.x == 0 && jf_min >= 0);
  double bv = -DBL_MAX;
  int64_t bi = -1, fi = -1;
  // for k_prb_tail: this workgroup's candidate with the largest bound (a candidate without a bound counts as DBL_MAX)
  // and the second largest bound among its candidates
  double t1 = -DBL_MAX, t2 = -DBL_MAX;
  int32_t tl = -1;
  const int n_tiles = (sh.nloc + kWave - 1) / kWave;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int l = tile * kWave + lane;
    const double c = l < sh.nloc ? obj[l] : 0.0;
    const bool cand = l < sh.nloc && !prb_basic(l, j, is_basic, q_in, q_out) && is_pos(c);
    double mn = DBL_MAX;   // no eligible row among this wavefront's rows
    if (cand || (any_fresh && l < sh.nloc)) {      // a row is cached for EVERY local column: any of them may price later
      double a[kPrbPerGroup];
#pragma unroll
      for (int k = 0; k < kPrbPerGroup; k++)
        a[k] = k0 + k >= n_rows ? 0.0 : ((cache && !fresh[k]) ? cache[(int64_t)slots[k] * sh.nloc + l] : T.col(l)[rows[k]]);
      constexpr int kBatch = 8;   // row' values fetched together: the steps depend on each other, their loads do not
      for (int s0 = jf_min; s0 < ja; s0 += kBatch) {
        double r[kBatch];
#pragma unroll
        for (int q = 0; q < kBatch; q++) r[q] = s0 + q < ja ? rowp[(int64_t)(s0 + q) * rstride + l] : 0.0;
#pragma unroll
        for (int q = 0; q < kBatch; q++) {
          const int s = s0 + q;
          if (s >= ja || r[q] == 0.0) continue;   // the update leaves the column alone in that step
#pragma unroll
          for (int k = 0; k < kPrbPerGroup; k++)
            if (s >= jfs[k]) a[k] = flushz(__dadd_rn(a[k], __dmul_rn(prb_c[s * kPrbRows + k0 + k], r[q])));   // a slot holds the steps before jf
        }
      }
#pragma unroll
      for (int k = 0; k < kPrbPerGroup; k++) {
        if (store[k]) cache[(int64_t)slots[k] * sh.nloc + l] = a[k];   // as steps 0..j-1 leave it (jf = j)
        if (cand && k0 + k < n_rows && (float)a[k] <= -1e-6f) {
          const double ratio = bs[k] / (-a[k]);
          if (ratio < mn) mn = ratio;
        }
      }
    }
    s_mn[grp][lane] = mn;
    __syncthreads();
    if (grp == 0 && l < sh.nloc) {
#pragma unroll
      for (int g = 1; g < kPrbGroups; g++) mn = fmin(mn, s_mn[g][lane]);
      const bool any = mn < DBL_MAX;
      const double u = any ? __dmul_rn(c, mn) : DBL_MAX;   // no eligible row in R: no bound
      ucand[l] = cand ? 1 : 0;
      if (cand) {
        ub[l] = u;
        if (fi < 0) fi = l;
        // The column evaluated first should have a TIGHT bound, so that its exact score rules the others out; a
        // column without a bound is listed for evaluation anyway and would make a poor yardstick.
        if (any && u > bv) { bv = u; bi = l; }
        if (tl < 0) { t1 = u; tl = l; }
        else if (u > t1) { t2 = t1; t1 = u; tl = l; }
        else if (u > t2) t2 = u;
      }
    }
    __syncthreads();
  }
  LPAMD_CLK(j, 1, 2, blockIdx.x == 0);
  if (grp == 0) {
    wave_arg<true>(bv, bi);
    double fv = fi < 0 ? 0.0 : -(double)fi;
    wave_arg<true>(fv, fi);
    if (lane == 0) { pv[blockIdx.x] = bv; pi[blockIdx.x] = bi; pfirst[blockIdx.x] = fi; }
    if (top_u1) {
      uint64_t key = tl >= 0 ? arg_key<true>(t1, 0) : 0;
      uint32_t idx = tl >= 0 ? (uint32_t)tl : kNoIndex;
      wave_best(key, idx);                                    // lowest column on equal bounds
      const double mine = tl < 0 ? -DBL_MAX : ((uint32_t)tl == idx ? t2 : t1);
      uint64_t key2 = mine > -DBL_MAX ? arg_key<true>(mine, 0) : 0;
      uint32_t any_lane = (uint32_t)lane;
      wave_best(key2, any_lane);
      if (lane == 0) {
        top_u1[blockIdx.x] = key ? arg_value<true>(key) : -DBL_MAX;
        top_l1[blockIdx.x] = key ? (int32_t)idx : -1;
        top_u2[blockIdx.x] = key2 ? arg_value<true>(key2) : -DBL_MAX;
      }
    }
    LPAMD_CLK(j, 1, 3, blockIdx.x == 0);
  }
}

// exact minimum ratio of local column l over the rows [i0, m) strided by `stride`, as the steps so far leave it
__device__ __forceinline__ void prb_column_min(const Tab& T, const Shape& sh, int l, int j, const double* __restrict__ colp,
                                               const double* __restrict__ rowp, int64_t rstride, const double* __restrict__ rhs,
                                               int64_t i0, int64_t stride, double& bv, int64_t& bi) {
  const double* __restrict__ col = T.col(l);
  for (int64_t i = i0; i < sh.m; i += stride) {
    const double a = chain_apply(col[i], j, colp + i, sh.lda, rowp + l, rstride);
    if ((float)a > -1e-6f) continue;                       // _IsNonNegative (reference :293 / :343)
    const double ratio = rhs[i] / (-a);
    if (bv > ratio) { bv = ratio; bi = i; }                // strict: lowest row on ties (:297 / :347)
  }
}

// 3. the column with the largest bound, evaluated exactly by all workgroups together (ratio partials)
__global__ __launch_bounds__(kPartialThreads) void k_prb_first(
    Tab T, Shape sh, const PivotState* __restrict__ st, const ChainStep* __restrict__ steps, int j,
    const double* __restrict__ colp, const double* __restrict__ rowp, int64_t rstride, const double* __restrict__ rhsv,
    const double* __restrict__ pv, const int64_t* __restrict__ pi, const int64_t* __restrict__ pfirst, int n_part,
    PrbInfo* __restrict__ info, double* __restrict__ rv, int64_t* __restrict__ ri) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  if (!prb_step_alive(st, steps, j)) return;
  double u1;
  int64_t l1;
  fold_partials<true, kPartialThreads>(pv, pi, n_part, u1, l1, s_v, s_i);
  double fv = 0.0;
  int64_t fi = -1;
  for (int k = threadIdx.x; k < n_part; k += kPartialThreads) {
    const int64_t f = pfirst[k];
    if (f >= 0) take_max(fv, fi, -(double)f, f);
  }
  block_arg<true, kPartialThreads>(fv, fi, s_v, s_i);
  if (threadIdx.x == 0) s_i[0] = fi;
  __syncthreads();
  fi = s_i[0];
  __syncthreads();
  if (l1 < 0) l1 = fi;           // no column has a bound: start from the lowest candidate
  if (blockIdx.x == 0 && threadIdx.x == 0) { info->l1 = (int32_t)l1; info->first = (int32_t)fi; }
  double bv = DBL_MAX;
  int64_t bi = -1;
  if (l1 >= 0)
    prb_column_min(T, sh, (int)l1, j, colp, rowp, rstride, prb_rhs(T, sh, rhsv, j), (int64_t)blockIdx.x * kPartialThreads + threadIdx.x,
                   (int64_t)gridDim.x * kPartialThreads, bv, bi);
  block_arg<false, kPartialThreads>(bv, bi, s_v, s_i);
  if (threadIdx.x == 0) { rv[blockIdx.x] = bv; ri[blockIdx.x] = bi; }
}

// 4. exact score s1 of that column; the columns whose bound reaches it are listed for exact evaluation
__global__ __launch_bounds__(kPartialThreads) void k_prb_list(
    Shape sh, const PivotState* __restrict__ st, const ChainStep* __restrict__ steps, int j,
    const double* __restrict__ obj, const double* __restrict__ rv, const int64_t* __restrict__ ri, int n_ratio_part,
    const double* __restrict__ ub, const uint8_t* __restrict__ ucand, PrbInfo* __restrict__ info,
    int32_t* __restrict__ elist, double* __restrict__ escore, int64_t* __restrict__ erow) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  if (!prb_step_alive(st, steps, j)) return;
  const int l1 = info->l1;
  if (l1 < 0) return;
  double min1;
  int64_t row1;
  fold_partials<false, kPartialThreads>(rv, ri, n_ratio_part, min1, row1, s_v, s_i);
  const double s1 = row1 >= 0 ? __dmul_rn(obj[l1], min1) : -DBL_MAX;
  if (blockIdx.x == 0 && threadIdx.x == 0) { info->s1 = s1; escore[l1] = s1; erow[l1] = row1; }
  for (int l0 = blockIdx.x * kPartialThreads; l0 < sh.nloc; l0 += gridDim.x * kPartialThreads) {
    const int l = l0 + threadIdx.x;
    const bool take = l < sh.nloc && l != l1 && ucand[l] != 0 && ub[l] >= s1;
    const unsigned long long mask = __ballot(take);
    if (mask) {
      const int lane = threadIdx.x & (kWave - 1);
      int base = 0;
      if (lane == 0) base = atomicAdd(&info->n_list, __popcll(mask));
      base = __shfl(base, 0, kWave);
      if (take) elist[base + __popcll(mask & ((1ull << lane) - 1ull))] = l;
    }
  }
}

// 5. exact evaluation of the listed columns.  A short list (the usual case: a handful of columns) is spread
// over the chip, one workgroup per column and slice of rows, and leaves ratio partials per slice; a long
// one (nothing could be ruled out) takes one workgroup per column at a time and leaves the results.
constexpr int kPrbSliced = 64;     // longest list evaluated in slices
constexpr int kPrbSlices = 128;    // most slices per column

__global__ __launch_bounds__(kPartialThreads) void k_prb_eval(
    Tab T, Shape sh, const PivotState* __restrict__ st, const ChainStep* __restrict__ steps, int j,
    const double* __restrict__ obj, const double* __restrict__ colp, const double* __restrict__ rowp, int64_t rstride,
    const double* __restrict__ rhsv, const PrbInfo* __restrict__ info, const int32_t* __restrict__ elist,
    double* __restrict__ escore, int64_t* __restrict__ erow, int n_slices, double* __restrict__ ep_v, int64_t* __restrict__ ep_i) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  if (!prb_step_alive(st, steps, j)) return;
  const int n_list = info->n_list;
  const double* __restrict__ rhs = prb_rhs(T, sh, rhsv, j);
  if (n_list <= kPrbSliced) {
    for (int w = blockIdx.x; w < n_list * n_slices; w += gridDim.x) {
      const int l = elist[w / n_slices], slice = w % n_slices;
      double bv = DBL_MAX;
      int64_t bi = -1;
      prb_column_min(T, sh, l, j, colp, rowp, rstride, rhs, (int64_t)slice * kPartialThreads + threadIdx.x,
                     (int64_t)n_slices * kPartialThreads, bv, bi);
      block_arg<false, kPartialThreads>(bv, bi, s_v, s_i);
      if (threadIdx.x == 0) { ep_v[w] = bv; ep_i[w] = bi; }
    }
    return;
  }
  for (int k = blockIdx.x; k < n_list; k += gridDim.x) {
    const int l = elist[k];
    double bv = DBL_MAX;
    int64_t bi = -1;
    prb_column_min(T, sh, l, j, colp, rowp, rstride, rhs, threadIdx.x, kPartialThreads, bv, bi);
    block_arg<false, kPartialThreads>(bv, bi, s_v, s_i);
    if (threadIdx.x == 0) {
      escore[l] = bi >= 0 ? __dmul_rn(obj[l], bv) : -DBL_MAX;
      erow[l] = bi;
    }
  }
}

// 6. the reference's fold (lp_simplex_tableau.cc:310-325 / 353-362) over the exact results: largest score
// among columns that have a row, lowest column on ties; no such column: the lowest candidate with row -1
// (=> unbounded); no candidate: gcol -1 (=> optimal).  A later step of the chain records its entering
// column for k_chain_finalize (step 0 hands the record to k_pivot_finalize; k_prb_rows has settled the budget).
__global__ __launch_bounds__(kPartialThreads) void k_prb_pick(
    Shape sh, PivotState* __restrict__ st, ChainStep* __restrict__ steps, int j,
    const double* __restrict__ obj, const PrbInfo* __restrict__ info, const int32_t* __restrict__ elist,
    double* __restrict__ escore, int64_t* __restrict__ erow, int n_slices, const double* __restrict__ ep_v,
    const int64_t* __restrict__ ep_i, Candidate* __restrict__ cand, int record_step) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  if (!prb_step_alive(st, steps, j)) return;
  const int l1 = info->l1, n_list = info->n_list;
  if (n_list <= kPrbSliced) {     // a short list was evaluated in slices: one wavefront folds the slices of a column
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int k = wave; k < n_list; k += kPartialThreads / kWave) {
      double mv = DBL_MAX;
      int64_t mi = -1;
      for (int q = lane; q < n_slices; q += kWave) take_min(mv, mi, ep_v[k * n_slices + q], ep_i[k * n_slices + q]);
      wave_arg<false>(mv, mi);
      if (lane == 0) {
        const int l = elist[k];
        escore[l] = mi >= 0 ? __dmul_rn(obj[l], mv) : -DBL_MAX;
        erow[l] = mi;
      }
    }
    __syncthreads();
  }
  double bv = -DBL_MAX;
  int64_t bi = -1;
  for (int k = threadIdx.x; k <= n_list; k += kPartialThreads) {
    const int l = k == n_list ? l1 : elist[k];
    if (l >= 0 && erow[l] >= 0) take_max(bv, bi, escore[l], (int64_t)l);
  }
  block_arg<true, kPartialThreads>(bv, bi, s_v, s_i);
  if (threadIdx.x == 0) {
    Candidate c;
    if (bi >= 0) {
      c.key = bv; c.gcol = gcol_of(bi * sh.world + sh.rank, sh.N); c.row = erow[bi]; c.ce = obj[bi];
    } else if (info->first >= 0) {
      c.key = -DBL_MAX; c.gcol = gcol_of((int64_t)info->first * sh.world + sh.rank, sh.N); c.row = -1; c.ce = obj[info->first];
    } else {
      c.key = -DBL_MAX; c.gcol = -1; c.row = -1; c.ce = 0.0;
    }
    *cand = c;
    PrbInfo* winfo = const_cast<PrbInfo*>(info);        // nothing else runs in this launch
    winfo->priced += 1;
    winfo->listed += n_list;
    if (j == 0) st->n_active = 0;                       // the set-up appends the active list from zero
    else if (record_step && c.gcol >= 0 && c.row >= 0) { steps[j].e = c.gcol; steps[j].ce = c.ce; }   // column shards: the winner among all ranks is recorded after the exchange
  }
}

// Column shards: this rank's candidate record (k_prb_pick) and that column as the pivots set up so far leave
// it, packed for the exchange.  A step that may not be set up any more (budget, chain ended) sends "no candidate".
__global__ __launch_bounds__(kPartialThreads) void k_prb_export(
    Tab T, Shape sh, const PivotState* __restrict__ st, const ChainStep* __restrict__ steps, int j,
    const double* __restrict__ colp, const double* __restrict__ rowp, int64_t rstride,
    Candidate* __restrict__ slot_cand, double* __restrict__ slot_col) {
  if (st->status != ST_RUNNING) return;
  if (!prb_step_alive(st, steps, j)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { slot_cand->gcol = -1; slot_cand->row = -1; }
    return;
  }
  const int64_t e = slot_cand->gcol;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sh.lda) return;
  double v = 0.0;
  if (e >= 0 && i < sh.m) {
    const int l = ord_local(ordinal_of(e, sh.N), sh.world);      // a local column by construction
    v = chain_apply(T.col(l)[i], j, colp + i, sh.lda, rowp + l, rstride);
  }
  slot_col[i] = v;
}

}  // namespace lpamd
