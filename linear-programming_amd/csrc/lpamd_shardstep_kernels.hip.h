// Column shards, MAX_COST, a step inside a chain: the set-up of step j from the gathered slots AND this rank's slot for
// step j+1, in ONE launch.
//
// Between two collectives a chained step used to be two launches: k_chain_finalize<slots> (set-up of step j from what
// the all-gather delivered) and k_export_lookahead<ratio> (this rank's candidate for step j+1: record, column as steps
// 0..j will leave it, ratio-test partials on it).  The second needs what the first produces in two different sets of
// workgroups: the pricing result of the column workgroups (which local column is the candidate, row'_j of it) and
// col'_j / the right-hand side after the step, which the row workgroups write.  Here the row workgroups stay resident
// behind their own part of the set-up, wait for the column workgroups' pricing results, which arrive as tagged records
// (see lpamd_fused_kernels.hip.h: data + tag per word, agent-scope accesses, no barrier), and write the slot from the
// col'_j and right-hand-side entries they still hold in registers: one launch and the collective per step.
//
// The records are published by workgroups that never wait for anybody, so the launch drains whatever the order in which
// its workgroups start; a poll that does not complete within kFusedSpinLimit rounds marks the state ST_ERROR.  The
// state that decides whether the step exists is not written by this launch except by its last writer's own fields (the
// pivot count travels to the row workgroups inside the first column workgroup's record).
//
// The exchange itself can move into the launch as well (kIn / kOut, ShardLink): the row workgroups store the slot straight
// into every rank's receive buffer over xGMI (peer-mapped device memory) instead of into a send buffer that a collective
// then carries, and the next step's launch reads what it needs of the ranks' slots from its own buffer.  No flag, counter or
// fence says "the slot has arrived": every 64-bit payload word travels as TWO tagged words (32 bits of data, 32-bit tag = the
// number of the exporting launch; each an aligned 8-byte store, which memory keeps whole), and a reader polls exactly the
// words it consumes until both carry the tag it expects, so arrival and content are one access whatever order the stores land
// in.  Three receive buffers take turns (tag mod 3): a rank can export tag t+3 only after it has consumed every rank's
// record of tag t+2, which each rank wrote from a launch that its stream ran after the launch that consumed tag t.  A poll
// gives up after kLinkTimeoutTicks of the constant 100 MHz clock, marks the state ST_ERROR and leaves: the grid always drains.
#pragma once
#include "lpamd_fused_kernels.hip.h"

namespace lpamd {

constexpr int kShardRecWords = 5;        // reduced cost (2), local column (1), pivot count after the step (2; first column workgroup)
constexpr int kShardMaxColGroups = 384;  // their records are polled by 256 threads, 8 words each at most
constexpr int kLinkBuffers = 3;
constexpr long long kLinkTimeoutTicks = 10LL * 100000000LL;   // ten seconds of the 100 MHz clock: a peer may be late by a pass, not by this
constexpr uint32_t kLinkMaxRounds = 1u << 27;                  // ... and a bound that does not depend on the clock

struct ShardLink {
  uint64_t* const* peers;   // device array [ranks of the exchange]: every rank's receive buffers as this process maps them
  const uint64_t* in;       // this rank's own receive buffers
  int64_t ll_words;         // tagged words per slot (twice the payload words)
  int32_t my_rank;          // where this rank's slot lives among the slots of a buffer
  uint32_t tag_in;          // tag of the slots this launch reads (kIn) ...
  uint32_t tag_out;         // ... and of the slot it writes (kOut)
  long long timeout;        // ticks of the 100 MHz clock a reader waits for a word (0: kLinkTimeoutTicks)
};

// payload word -> its pair of tagged words at p[0], p[1] (low half, high half)
__device__ __forceinline__ void ll_put(uint64_t* p, uint64_t v, uint32_t tag) {
  __hip_atomic_store(p, (v << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(p + 1, (v & 0xFFFFFFFF00000000ull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// false: the pair did not carry `tag` in time
__device__ __forceinline__ bool ll_get(const uint64_t* p, uint32_t tag, uint64_t& v, long long timeout = kLinkTimeoutTicks) {
  long long t0 = 0;
  for (uint32_t round = 0; round < kLinkMaxRounds; round++) {
    const uint64_t a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint64_t b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((uint32_t)a == tag && (uint32_t)b == tag) { v = (a >> 32) | (b & 0xFFFFFFFF00000000ull); return true; }
    if (round == 64) t0 = wall_clock64();
    else if (round > 64 && (round & 63) == 0 && wall_clock64() - t0 > timeout) return false;
    if (round < 16) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8);
  }
  return false;
}
// The same for up to kN payload words at once (those whose bit is set in `want`): all their loads are out before the first
// tag is looked at, so a thread that needs several words pays one round trip to memory, not one per word.
template <int kN>
__device__ __forceinline__ bool ll_get_many(const uint64_t* const (&p)[kN], uint32_t tag, uint64_t (&v)[kN], uint32_t want, long long timeout = kLinkTimeoutTicks) {
  uint32_t pending = want;
  long long t0 = 0;
  for (uint32_t round = 0; round < kLinkMaxRounds && pending; round++) {
    uint64_t a[kN], b[kN];
#pragma unroll
    for (int q = 0; q < kN; q++)
      if ((pending >> q) & 1u) {
        a[q] = __hip_atomic_load(p[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        b[q] = __hip_atomic_load(p[q] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
#pragma unroll
    for (int q = 0; q < kN; q++)
      if (((pending >> q) & 1u) && (uint32_t)a[q] == tag && (uint32_t)b[q] == tag) {
        v[q] = (a[q] >> 32) | (b[q] & 0xFFFFFFFF00000000ull);
        pending &= ~(1u << q);
      }
    if (!pending) break;
    if (round == 64) t0 = wall_clock64();
    else if (round > 64 && (round & 63) == 0 && wall_clock64() - t0 > timeout) return false;
    if (round < 16) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8);
  }
  return pending == 0;
}
__device__ __forceinline__ bool ll_get_f64(const uint64_t* p, uint32_t tag, double& v) {
  uint64_t b = 0;
  const bool ok = ll_get(p, tag, b);
  v = __longlong_as_double((long long)b);
  return ok;
}

// k_export_lookahead<ratio> with the slot stored straight into every rank's receive buffer: this rank's candidate for
// step 1 of a pass (record, column as the pivots set up so far will leave it, ratio-test partials; one workgroup per 256
// rows, so every pair of partials has its writer).
__global__ __launch_bounds__(kPartialThreads) void k_export_lookahead_link(
    Tab T, Shape sh, const PivotState* __restrict__ st, const double* __restrict__ pv, const int64_t* __restrict__ pi, int n_price_part,
    const double* __restrict__ colp, const double* __restrict__ rowp, int64_t rstride, const ChainStep* __restrict__ steps, int pending,
    const double* __restrict__ rhsv, int slot_world, int slot_header, ShardLink link) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  const int32_t status = st->status;
  const int64_t iter = st->iter, max_iter = st->max_iter;
  const int n = pending > 0 ? chain_steps_done(steps, pending) : 0;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const double* __restrict__ rhs = n == 0 ? T.col(sh.nloc) : rhsv + (int64_t)(n - 1) * sh.lda;
  const double rhs_i = i < sh.m ? rhs[i] : 0.0;
  double ce = -DBL_MAX;
  int64_t l = -1;
  fold_share<true, kPartialThreads>(pv, pi, n_price_part, ce, l);
  if (status != ST_RUNNING) return;
  block_arg<true, kPartialThreads>(ce, l, s_v, s_i);
  if (iter >= max_iter) l = -1;
  const int64_t out_at = ((int64_t)(link.tag_out % kLinkBuffers) * slot_world + link.my_rank) * link.ll_words;
  const int n_parts = (int)gridDim.x;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t gcol = l < 0 ? -1 : gcol_of(l * sh.world + sh.rank, sh.N);
    for (int p = 0; p < slot_world; p++) {
      uint64_t* dst = link.peers[p] + out_at;
      ll_put(dst, (uint64_t)__double_as_longlong(ce), link.tag_out);
      ll_put(dst + 2, (uint64_t)gcol, link.tag_out);
      ll_put(dst + 4, ~0ull, link.tag_out);
      ll_put(dst + 6, (uint64_t)__double_as_longlong(ce), link.tag_out);
    }
  }
  double v = 0.0;
  if (l >= 0 && i < sh.m) v = chain_apply(T.col(l)[i], n, colp + i, sh.lda, rowp + l, rstride);
  if (i < sh.lda)
    for (int p = 0; p < slot_world; p++) ll_put(link.peers[p] + out_at + 2 * ((int64_t)slot_header + i), (uint64_t)__double_as_longlong(v), link.tag_out);
  double bv = DBL_MAX;
  int64_t bi = -1;
  if (l >= 0 && i < sh.m && (float)v <= -1e-6f) { bv = rhs_i / (-v); bi = i; }   // reference lp_simplex_tableau.cc:207-227
  block_arg<false, kPartialThreads>(bv, bi, s_v, s_i);
  if (bi >= 0 ? i == bi : threadIdx.x == 0)     // the thread that owns the winning row holds the column's entry there
    for (int p = 0; p < slot_world; p++) {
      uint64_t* dst = link.peers[p] + out_at + 2 * ((int64_t)slot_header + sh.lda);
      ll_put(dst + 2 * (int64_t)blockIdx.x, (uint64_t)__double_as_longlong(bv), link.tag_out);
      ll_put(dst + 2 * ((int64_t)n_parts + blockIdx.x), (uint64_t)bi, link.tag_out);
      ll_put(dst + 2 * (2 * (int64_t)n_parts + blockIdx.x), (uint64_t)__double_as_longlong(bi >= 0 ? v : 0.0), link.tag_out);
    }
}

template <bool kIn, bool kOut>
__global__ __launch_bounds__(kPartialThreads, 2) void k_shard_step(
    Tab T, Shape sh, PivotState* __restrict__ st, ChainStep* __restrict__ steps, int j, int n_ratio_part, int n_row_blocks,
    double* __restrict__ colp, double* __restrict__ rowp, int64_t rstride, double* __restrict__ rhsv,
    double* __restrict__ obj, const uint8_t* __restrict__ is_basic, const int64_t* __restrict__ basis,
    int64_t* __restrict__ trace, int64_t trace_cap, double* __restrict__ next_pv, int64_t* __restrict__ next_pi,
    const uint64_t* __restrict__ slots, int slot_world, int64_t slot_words, int slot_header, int depth,
    uint64_t* __restrict__ my_slot,      // nullptr: no slot for a next step (last step of the pass); kOut: any other value
    uint64_t* __restrict__ rec, uint32_t epoch, ShardLink link) {
  constexpr int kThreads = kPartialThreads;
  __shared__ double s_v[kThreads / kWave];
  __shared__ int64_t s_i[kThreads / kWave];
  __shared__ int s_w;
  __shared__ int64_t s_e2;
  __shared__ double s_ce2;
  __shared__ double s_a2;
  __shared__ uint32_t s_words[8 * kThreads];
  __shared__ __attribute__((aligned(16))) double s_u[kMaxChain];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const bool row_block = (int)blockIdx.x < n_row_blocks;
  const int cb = (int)blockIdx.x - n_row_blocks, n_cb = (int)gridDim.x - n_row_blocks;
  const int ncl = sh.nloc + 1;
  LPAMD_CLKR_DECL;
  const int clk_who = blockIdx.x == 0 ? 0 : cb == 0 ? 2 : cb == n_cb - 1 ? 3 : -1;   // timeline (experiments only): who this workgroup is
  (void)clk_who;
  LPAMD_CLKR(0, clk_who >= 0);
  const uint32_t tag = (epoch << 8) | 1u;
  const int64_t region = fused_region_words(n_cb, kShardRecWords);
  const double* __restrict__ rhs_prev = rhsv + (int64_t)(j - 1) * sh.lda;
  // ---- requested before the first wait (as in k_chain_finalize): the gating state, the exchange records (one per lane of
  // the first wavefront), the step records (one per lane), the entries of the row or column this thread owns
  const int32_t status = st->status;
  const int32_t prev_done = j > 1 ? steps[j - 1].done : 1;
  const int32_t dirty_step = st->dirty_step;
  const int64_t max_iter = st->max_iter;
  Candidate c_rec;
  c_rec.key = -DBL_MAX; c_rec.gcol = -1; c_rec.row = -1; c_rec.ce = 0.0;
  if (!kIn && slot_world <= kWave && tid < slot_world) c_rec = *reinterpret_cast<const Candidate*>(slots + (size_t)tid * slot_words);
  // kIn: a first look at the ranks' records goes out with the loads above (whether the step exists at all is only known
  // once those are back, and a step that does not exist must not wait for records nobody sends)
  const uint64_t* rec_at = kIn ? link.in + ((int64_t)(link.tag_in % kLinkBuffers) * slot_world + (tid < slot_world ? tid : 0)) * link.ll_words : nullptr;
  uint64_t first_a[4] = {0, 0, 0, 0}, first_b[4] = {0, 0, 0, 0};
  if (kIn && tid < slot_world) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      first_a[q] = __hip_atomic_load(rec_at + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      first_b[q] = __hip_atomic_load(rec_at + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // kIn, column workgroups (they are the ones everybody waits for): the ratio-test partials of EVERY rank go out with that
  // first look, up to four triples per thread, so that once the winner is known its partials are already here and the pivot
  // needs no second round trip to uncached memory (2 us each).  The row workgroups, which wait for the column workgroups'
  // pricing anyway, read only the winner's.
  const int n_spec = (int)blockIdx.x >= n_row_blocks ? slot_world * n_ratio_part : 0;
  const bool spec = kIn && n_spec > 0 && n_spec <= 4 * kThreads;
  uint64_t sp_a[4][3], sp_b[4][3];
  if (spec) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int t = tid + q * kThreads;
      if (t < n_spec) {
        const int k = t / n_ratio_part, x = t - k * n_ratio_part;
        const uint64_t* part = link.in + ((int64_t)(link.tag_in % kLinkBuffers) * slot_world + k) * link.ll_words + 2 * ((int64_t)slot_header + sh.lda);
#pragma unroll
        for (int c = 0; c < 3; c++) {
          sp_a[q][c] = __hip_atomic_load(part + 2 * ((int64_t)c * n_ratio_part + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          sp_b[q][c] = __hip_atomic_load(part + 2 * ((int64_t)c * n_ratio_part + x) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
  int32_t q_in = -1, q_out = -1;
  int64_t q_e = -1, q_r = -1;
  if (lane == 0) { q_in = st->l_e; q_out = st->l_d; q_e = st->e; q_r = st->r; }
  else if (lane < j) { q_in = steps[lane].l_in; q_out = steps[lane].l_out; q_e = steps[lane].e; q_r = steps[lane].r; }
  const int64_t i0 = (int64_t)blockIdx.x * kThreads + tid;         // row workgroups: one row per thread (the host checks m)
  const bool have_row = row_block && i0 < sh.m;
  const double rhs0 = have_row ? rhs_prev[i0] : 0.0;
  if (status != ST_RUNNING) return;
  const bool keeper = cb == 0 && tid == 0;
  if (j == 1 && keeper) for (int s = 1; s < depth; s++) { steps[s].done = 0; steps[s].e = -1; }
  if (!prev_done) return;
  if (dirty_step != 0 && dirty_step - 1 < j) return;
  bool late = false;                      // kIn: a word this thread needs did not arrive
  if (kIn && tid < slot_world) {          // (the host admits at most kWave ranks)
    uint64_t w4[4] = {0, ~0ull, ~0ull, 0};
    uint32_t missing = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if ((uint32_t)first_a[q] == link.tag_in && (uint32_t)first_b[q] == link.tag_in) w4[q] = (first_a[q] >> 32) | (first_b[q] & 0xFFFFFFFF00000000ull);
      else missing |= 1u << q;
    }
    if (missing) {
      const uint64_t* const p4[4] = {rec_at, rec_at + 2, rec_at + 4, rec_at + 6};
      late = !ll_get_many<4>(p4, link.tag_in, w4, missing, link.timeout ? link.timeout : kLinkTimeoutTicks);
    }
    c_rec.key = __longlong_as_double((long long)w4[0]); c_rec.gcol = (int64_t)w4[1]; c_rec.row = (int64_t)w4[2];
    c_rec.ce = __longlong_as_double((long long)w4[3]);
    if (late) c_rec.gcol = -1;
  }
  // ---- the winner among the ranks' records (larger key, lower global column; reference lp_simplex_tableau.cc:188-198)
  if (slot_world <= kWave) {
    if (tid < kWave) {
      const bool valid = c_rec.gcol >= 0;
      double bk = valid ? c_rec.key : -DBL_MAX;
      const int64_t gc = valid ? c_rec.gcol : -1;
      int64_t bg = gc;
      wave_arg<true>(bk, bg);
      const unsigned long long owner = __ballot(gc >= 0 && gc == bg);
      const int w = bg < 0 ? -1 : __ffsll((long long)owner) - 1;
      if (lane == 0) s_w = w;
      if (lane == w) { s_e2 = c_rec.gcol; s_ce2 = c_rec.ce; }
    }
  } else if (tid == 0) {
    int w = -1;
    Candidate best = c_rec;
    for (int k = 0; k < slot_world; k++) {
      const Candidate c = *reinterpret_cast<const Candidate*>(slots + (size_t)k * slot_words);
      if (c.gcol < 0) continue;
      if (w < 0 || c.key > best.key || (c.key == best.key && c.gcol < best.gcol)) { w = k; best = c; }
    }
    s_w = w; s_e2 = best.gcol; s_ce2 = best.ce;
  }
  if (kIn) {
    if (__syncthreads_or(late)) { if (tid == 0) st->status = ST_ERROR; return; }
  } else {
    __syncthreads();
  }
  const int w = s_w;
  LPAMD_CLKR(1, clk_who >= 0);
  if (w < 0) return;                      // optimal after step j-1, or the budget ends with it
  const uint64_t* __restrict__ mine = slots + (size_t)(kIn ? 0 : w) * slot_words;
  const int64_t e2 = s_e2;
  const double ce2 = s_ce2;
  const double* __restrict__ colv = reinterpret_cast<const double*>(mine + slot_header);
  const double* __restrict__ rv = colv + sh.lda;      // the owner's ratio-test partials: values, then rows
  const int64_t* __restrict__ ri = reinterpret_cast<const int64_t*>(rv + n_ratio_part);
  // kIn: the winner's slot in this rank's receive buffer, payload word x at ll_mine[2 * x]
  const uint64_t* __restrict__ ll_mine = kIn ? link.in + ((int64_t)(link.tag_in % kLinkBuffers) * slot_world + w) * link.ll_words : nullptr;
  double colv0 = 0.0;
  double best = DBL_MAX;
  int64_t r2 = -1;
  double my_a = 0.0;      // kIn: the winner's column at the row of this thread's best partial
  int64_t my_r = -1;
  if (spec) {
    // the winner's triples among those this thread asked for on entry; one that had not arrived then is polled now
    const uint64_t* part = ll_mine + 2 * ((int64_t)slot_header + sh.lda);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int t = tid + q * kThreads;
      const int k = t < n_spec ? t / n_ratio_part : -1;
      if (k == w) {
        const int x = t - k * n_ratio_part;
        uint64_t w3[3] = {0, ~0ull, 0};
        uint32_t missing = 0;
#pragma unroll
        for (int c = 0; c < 3; c++) {
          if ((uint32_t)sp_a[q][c] == link.tag_in && (uint32_t)sp_b[q][c] == link.tag_in) w3[c] = (sp_a[q][c] >> 32) | (sp_b[q][c] & 0xFFFFFFFF00000000ull);
          else missing |= 1u << c;
        }
        if (missing) {
          const uint64_t* const p3[3] = {part + 2 * (int64_t)x, part + 2 * ((int64_t)n_ratio_part + x), part + 2 * (2 * (int64_t)n_ratio_part + x)};
          late |= !ll_get_many<3>(p3, link.tag_in, w3, missing, link.timeout ? link.timeout : kLinkTimeoutTicks);
        }
        if (!late) {
          const int64_t before = r2;
          take_min(best, r2, __longlong_as_double((long long)w3[0]), (int64_t)w3[1]);
          if (r2 != before) my_a = __longlong_as_double((long long)w3[2]);
        }
      }
    }
    my_r = r2;
  } else if (kIn) {
    // this thread's row of the column and its first triple of ratio-test partials (ratio, row, the column's entry at that
    // row) in one round trip; the entry travels with the partial so that the pivot element needs no read of its own
    const uint64_t* part = ll_mine + 2 * ((int64_t)slot_header + sh.lda);
    {
      uint64_t w4[4] = {0, 0, ~0ull, 0};
      const uint64_t* const p4[4] = {ll_mine + 2 * ((int64_t)slot_header + (have_row ? i0 : 0)), part + 2 * (int64_t)tid,
                                     part + 2 * ((int64_t)n_ratio_part + tid), part + 2 * (2 * (int64_t)n_ratio_part + tid)};
      const uint32_t want = (have_row ? 1u : 0u) | (tid < n_ratio_part ? 14u : 0u);
      late |= !ll_get_many<4>(p4, link.tag_in, w4, want, link.timeout ? link.timeout : kLinkTimeoutTicks);
      colv0 = __longlong_as_double((long long)w4[0]);
      if (!late && tid < n_ratio_part) {
        const double pv = __longlong_as_double((long long)w4[1]);
        const int64_t before = r2;
        take_min(best, r2, pv, (int64_t)w4[2]);
        if (r2 != before) my_a = __longlong_as_double((long long)w4[3]);
      }
    }
    for (int k = tid + kThreads; k < n_ratio_part; k += kThreads) {
      uint64_t w3[3] = {0, ~0ull, 0};
      const uint64_t* const p3[3] = {part + 2 * (int64_t)k, part + 2 * ((int64_t)n_ratio_part + k), part + 2 * (2 * (int64_t)n_ratio_part + k)};
      late |= !ll_get_many<3>(p3, link.tag_in, w3, 7u, link.timeout ? link.timeout : kLinkTimeoutTicks);
      if (!late) {
        const int64_t before = r2;
        take_min(best, r2, __longlong_as_double((long long)w3[0]), (int64_t)w3[1]);
        if (r2 != before) my_a = __longlong_as_double((long long)w3[2]);
      }
    }
    my_r = r2;
  } else {
    colv0 = have_row ? colv[i0] : 0.0;
    fold_share<false, kThreads>(rv, ri, n_ratio_part, best, r2);
  }
  block_arg<false, kThreads>(best, r2, s_v, s_i);
  if (kIn && __syncthreads_or(late)) { if (tid == 0) st->status = ST_ERROR; return; }
  if (r2 < 0 || r2 >= sh.m) return;       // unbounded after step j-1: found again, and reported, by the next iteration
  r2 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(r2 >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r2));
  auto local_id = [&](int64_t g) -> int {
    if (g < 0 || g == sh.N) return -1;
    const int64_t o = ordinal_of(g, sh.N);
    return (ord_rank(o, sh.world) == sh.rank) ? ord_local(o, sh.world) : -1;
  };
  const int l_e2 = local_id(e2);
  double a2 = 0.0;
  if (kIn) {
    if (my_r == r2) s_a2 = my_a;           // exactly one partial stands for row r2
    __syncthreads();
    a2 = s_a2;
  } else {
    a2 = colv[r2];
  }
  const double f2 = -1.0 / a2;
  const double g2 = __dmul_rn(ce2, f2);
  LPAMD_CLKR(2, clk_who >= 0);

  if (row_block) {
    // ---- col'_j and the right-hand side after the step for this thread's row
    double cpi = 0.0, rhs_i = 0.0;
    if (have_row) {
      const double rp_rhs = __dmul_rn(rhs_prev[r2], f2);   // row'_j of the right-hand-side column
      cpi = (i0 == r2) ? __dadd_rn(1.0, a2) : colv0;      // lp_simplex_tableau.cc:41-44
      rhs_i = rp_rhs != 0.0 ? flushz(__dadd_rn(rhs0, __dmul_rn(cpi, rp_rhs))) : rhs0;
      colp[(int64_t)j * sh.lda + i0] = cpi;
      rhsv[(int64_t)j * sh.lda + i0] = rhs_i;
      if (!(fabs(cpi) < kDirtyBound)) st->dirty_step = j + 1;   // PivotState::dirty
    }
    LPAMD_CLKR(3, clk_who >= 0);
    if (!my_slot) { LPAMD_CLKR_FLUSH(j, clk_who, clk_who >= 0); return; }
    // ---- this rank's slot for step j+1.  While the column workgroups are at work: col'_s of this thread's row for the
    // steps so far, into registers (they do not depend on which column the candidate will be)
    double h[kMaxChain];
#pragma unroll
    for (int s0 = 0; s0 < kMaxChain; s0 += 4) {
      if (s0 < j) {
#pragma unroll
        for (int q = 0; q < 4; q++) h[s0 + q] = (s0 + q < j && have_row) ? colp[(int64_t)(s0 + q) * sh.lda + i0] : 0.0;
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) h[s0 + q] = 0.0;
      }
    }
#pragma unroll
    for (int s = 1; s < kMaxChain; s++) if (s == j) h[s] = cpi;
    // The pricing after step j: the column workgroups' records
    const bool ok = poll_words<8>(rec + (int64_t)((int)blockIdx.x % kFusedReplicas) * region, n_cb * kShardRecWords, tag, tid, kThreads, s_words);
    if (__syncthreads_or(!ok)) { if (tid == 0) st->status = ST_ERROR; return; }
    LPAMD_CLKR(4, clk_who >= 0);
    double ce = -DBL_MAX;
    int64_t lw = -1;
    for (int k = tid; k < n_cb; k += kThreads) {
      const uint32_t* d = s_words + k * kShardRecWords;
      take_max(ce, lw, rec_f64(d[0], d[1]), (int64_t)(int32_t)d[2]);
    }
    const int64_t iter_after = (int64_t)(((uint64_t)s_words[4] << 32) | s_words[3]);
    block_arg<true, kThreads>(ce, lw, s_v, s_i);
    if (iter_after >= max_iter) lw = -1;           // a used-up budget travels as "no candidate"
    // kOut: where this rank's slot starts in a receive buffer
    const int64_t out_at = ((int64_t)(link.tag_out % kLinkBuffers) * slot_world + link.my_rank) * link.ll_words;
    if (blockIdx.x == 0 && tid == 0) {
      Candidate c;
      c.key = ce; c.ce = ce; c.row = -1;
      c.gcol = lw < 0 ? -1 : gcol_of(lw * sh.world + sh.rank, sh.N);
      if (kOut) {
        for (int p = 0; p < slot_world; p++) {
          uint64_t* dst = link.peers[p] + out_at;
          ll_put(dst, (uint64_t)__double_as_longlong(c.key), link.tag_out);
          ll_put(dst + 2, (uint64_t)c.gcol, link.tag_out);
          ll_put(dst + 4, (uint64_t)c.row, link.tag_out);
          ll_put(dst + 6, (uint64_t)__double_as_longlong(c.ce), link.tag_out);
        }
      } else {
        *reinterpret_cast<Candidate*>(my_slot) = c;
      }
    }
    double* __restrict__ xlocal = reinterpret_cast<double*>(my_slot + slot_header);
    double* __restrict__ rpart_v = xlocal + sh.lda;
    int64_t* __restrict__ rpart_i = reinterpret_cast<int64_t*>(rpart_v + n_ratio_part);
    // the candidate column as steps 0..j will leave it: steps 0..j-1 from what the launches before left, step j from the
    // col' entry this thread has just computed and the row' entry its column workgroup has just written
    double v = 0.0;
    const int l = lw >= 0 ? __builtin_amdgcn_readfirstlane((int)lw) : 0;
    const double t0 = (lw >= 0 && have_row) ? T.col(l)[i0] : 0.0;
    if (lw >= 0 && tid <= j) s_u[tid] = ld_shared_f64(rowp + (int64_t)tid * rstride + l);   // row'_s of the candidate, one round trip
    __syncthreads();
    if (lw >= 0 && have_row) v = fused_chain_reg<true, kMaxChain>(t0, j + 1, s_u, h);
    if (kOut) {
      if (i0 < sh.lda)
        for (int p = 0; p < slot_world; p++) ll_put(link.peers[p] + out_at + 2 * ((int64_t)slot_header + i0), (uint64_t)__double_as_longlong(v), link.tag_out);
    } else if (i0 < sh.lda) {
      xlocal[i0] = v;
    }
    LPAMD_CLKR(5, clk_who >= 0);
    double bv = DBL_MAX;
    int64_t bi = -1;
    if (lw >= 0 && have_row && (float)v <= -1e-6f) { bv = rhs_i / (-v); bi = i0; }   // reference lp_simplex_tableau.cc:207-227
    block_arg<false, kThreads>(bv, bi, s_v, s_i);
    if (kOut) {
      // one pair of partials per row workgroup; the first one also fills the pairs no workgroup stands for (every word a
      // reader may poll must be written)
      auto put_triple = [&](int k, double pv, int64_t pr, double pa) {
        for (int p = 0; p < slot_world; p++) {
          uint64_t* dst = link.peers[p] + out_at + 2 * ((int64_t)slot_header + sh.lda);
          ll_put(dst + 2 * (int64_t)k, (uint64_t)__double_as_longlong(pv), link.tag_out);
          ll_put(dst + 2 * ((int64_t)n_ratio_part + k), (uint64_t)pr, link.tag_out);
          ll_put(dst + 2 * (2 * (int64_t)n_ratio_part + k), (uint64_t)__double_as_longlong(pa), link.tag_out);
        }
      };
      if (bi >= 0 && i0 == bi) put_triple((int)blockIdx.x, bv, bi, v);      // the thread that owns the winning row holds the entry
      if (bi < 0 && tid == 0) put_triple((int)blockIdx.x, DBL_MAX, -1, 0.0);
      if (blockIdx.x == 0)
        for (int k = n_row_blocks + tid; k < n_ratio_part; k += kThreads) put_triple(k, DBL_MAX, -1, 0.0);
    } else if (tid == 0) {
      rpart_v[blockIdx.x] = bv; rpart_i[blockIdx.x] = bi;
    }
    LPAMD_CLKR(6, clk_who >= 0);
    LPAMD_CLKR_FLUSH(j, clk_who, clk_who >= 0);
    return;
  }

  // ---- column workgroups: the pivot row through the steps so far, row'_j, the objective row, the pricing after the step
  int64_t d2 = basis[r2];
  {
    const unsigned long long hit = __ballot(lane < j && q_r == r2);
    if (hit) d2 = readlane_i64(q_e, 63 - __clzll((long long)hit));
  }
  const int l_d2 = local_id(d2);
  const double* __restrict__ c_r = colp + r2;       // col'_s[r2] at c_r[s * lda]
  double nbv = -DBL_MAX;
  int64_t nbi = -1;
  for (int l = cb * kThreads + tid; l < ncl; l += n_cb * kThreads) {
    const double t = chain_apply(T.col(l)[r2], j, c_r, sh.lda, rowp + l, rstride);
    const double rp2 = __dmul_rn(t, f2);
    double o = flushz(__dadd_rn(obj[l], __dmul_rn(t, g2)));   // :36
    if (l == l_e2) o = 0.0;                                     // :37
    obj[l] = o;
    st_shared_f64(rowp + (int64_t)j * rstride + l, rp2);
    if (l < sh.nloc) {
      bool basic = is_basic[l] != 0;   // flags as before the pass, then the steps set up so far
#pragma unroll 1
      for (int s = 0; s < j; s++) {
        const int in_s = __builtin_amdgcn_readlane(q_in, s), out_s = __builtin_amdgcn_readlane(q_out, s);
        if (l == in_s) basic = true; else if (l == out_s) basic = false;
      }
      if (l == l_e2) basic = true; else if (l == l_d2) basic = false;
      if (!basic && is_pos(o) && o > nbv) { nbv = o; nbi = l; }
    }
  }
  LPAMD_CLKR(3, clk_who >= 0);
  wait_stores();                          // row'_j is out before this workgroup's record
  block_arg<true, kThreads>(nbv, nbi, s_v, s_i);
  __shared__ int64_t s_iter;
  if (tid == 0) { next_pv[cb] = nbv; next_pi[cb] = nbi; s_iter = cb == 0 ? st->iter + 1 : 0; }
  __syncthreads();
  const int64_t iter_after = s_iter;
  if (my_slot && tid < kFusedReplicas * kShardRecWords) {
    // every thread holds the workgroup's result: thread c * kShardRecWords + x writes word x of copy c
    const int x = tid % kShardRecWords;
    const uint64_t vb = (uint64_t)__double_as_longlong(nbv);
    const uint32_t word = x == 0 ? (uint32_t)vb : x == 1 ? (uint32_t)(vb >> 32) : x == 2 ? (uint32_t)(int32_t)nbi
                        : x == 3 ? (uint32_t)(uint64_t)iter_after : (uint32_t)((uint64_t)iter_after >> 32);
    rec_put(rec + (int64_t)(tid / kShardRecWords) * region + (int64_t)cb * kShardRecWords + x, word, tag);
  }
  LPAMD_CLKR(4, clk_who >= 0);
  LPAMD_CLKR_FLUSH(j, clk_who, clk_who >= 0);
  if (keeper) {
    ChainStep z;
    z.done = 1; z.slot = 0; z.l_in = l_e2; z.l_out = l_d2;
    z.e = e2; z.r = r2; z.d = d2; z.a = a2; z.ce = ce2; z.f = f2; z.g = g2;
    steps[j] = z;
    const int64_t it = iter_after - 1;
    if (it < trace_cap) { trace[2 * it] = e2; trace[2 * it + 1] = r2; }
    st->iter = iter_after;
  }
}

}  // namespace lpamd
