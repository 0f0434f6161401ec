// Steps 1 .. K-1 of a chain of MAX_COST pivots set up by ONE launch (one GPU).
//
// k_chain_ratio / k_chain_finalize set a step up with two launches, and each launch pays the end of one kernel, the
// start of the next and a fresh look at the state that says whether the step exists: two thirds of a step at
// m=4096 n=8192, a tenth of a pass at m=32768 n=65536.  Here the workgroups stay resident for the whole chain and hand
// each other what a kernel boundary used to hand over:
//
//   row workgroups  [0, n_rb)     own rows.  Per step: fold the pricing results, rebuild the entering column through
//                                 the steps so far, ratio test (reference lp_simplex_tableau.cc:188-227); publish the
//                                 workgroup's best row, fold all rows' records, write col'_j and the right-hand
//                                 side after the step.
//   column workgroups [n_rb, G)   own columns.  Per step: fold the row records, gather the pivot row through the steps
//                                 so far, row'_j, objective row, basis flags, the pricing result for the next step
//                                 (reference lp_simplex_tableau.cc:10-44); publish it.
//
// There is no barrier and no counter.  A result travels as a RECORD of 8-byte words, each word 32 bits of data and a
// 32-bit tag (launch number and step); the reader polls the record until every word carries the tag it waits for, so
// "has it arrived" and "what is it" are one round trip.  Records and the vectors other workgroups read inside this
// launch (row'_s of the entering column, col'_s at the pivot row) are written and read with agent-scope relaxed
// atomics: each such access goes past the XCD's own L2 by itself, no cache-wide write-back or invalidate is needed,
// and a plain wait for the stores (vmcnt) before the workgroup's record is published orders them (measured beside a
// counter barrier with release / acquire fences in scripts/grid_barrier_latency.hip: 1.4 against 2.6 .. 7.5 us).
// The records of a kind are folded by the FIRST WAVEFRONT of a workgroup, one record per lane (every poll goes out to
// memory, and a few hundred workgroups polling with all their wavefronts slow each other and the working ones down:
// 45 us per step measured that way at m=32768 n=65536); the rest of the workgroup waits at the barrier behind it.
// Record sets alternate with the parity of the step: a writer reaches step j+2 only after every reader has
// consumed step j (it needs their step j+1 records to get there).
//
// Every workgroup takes the same decisions from the same records, so all of them leave the loop at the same step
// (optimum, no eligible row, budget, depth); a poll that does not complete within kFusedSpinLimit rounds gives up,
// marks the state ST_ERROR and leaves, and so then does everybody who waits for that workgroup: the grid always
// drains.  The state a workgroup reads on entry (status, pivot count, budget, pivot 0) is not written by this launch
// (a workgroup may start late): the pivot count is advanced by k_chain_list afterwards.
#pragma once
#include "lpamd_kernels.hip.h"

namespace lpamd {

constexpr int kFusedRatioWords = 6;     // ratio (2), row (1), entering column (1), its reduced cost (2); 48-byte records
constexpr int kFusedPriceWords = 3;     // reduced cost (2), local column (1); 24-byte records
constexpr uint32_t kFusedNoRow = 0xFFFFFFFFu;   // the workgroup's rows hold no eligible entry
constexpr uint32_t kFusedEnd = 0xFFFFFFFEu;     // no entering column: the chain ends before this step
constexpr uint32_t kFusedSpinLimit = 1u << 22;
// Every record is published kFusedReplicas times, each set of copies in a region of its own (a multiple of 4 KiB apart),
// and a workgroup polls the set blockIdx % kFusedReplicas: several hundred workgroups polling the same few KiB went out to
// the same memory channels and a hand-over took 7 us instead of 2.
constexpr int kFusedReplicas = 8;
__host__ __device__ __forceinline__ int64_t fused_region_words(int n_records, int words) {   // words between two sets of copies
  return ((int64_t)n_records * words + 511) / 512 * 512;
}

__device__ __forceinline__ void rec_put(uint64_t* w, uint32_t data, uint32_t tag) {
  __hip_atomic_store(w, ((uint64_t)data << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rec_put_f64(uint64_t* w, double v, uint32_t tag) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  rec_put(w, (uint32_t)b, tag);
  rec_put(w + 1, (uint32_t)(b >> 32), tag);
}
__device__ __forceinline__ double rec_f64(uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo)); }

// `pollers` threads (whole wavefronts) poll a set of n_words <= pollers * kMaxW tagged words laid out back to back
// (thread t takes words t, t + pollers, ...: a wavefront's request is one contiguous run) and leave the data halves in
// s_words[] once a word carries `tag`.  A wavefront returns when all its lanes have theirs.  False: some word of this
// wavefront did not arrive within kFusedSpinLimit rounds.
template <int kMaxW>
__device__ __forceinline__ bool poll_words(const uint64_t* words, int n_words, uint32_t tag, int t, int pollers, uint32_t* s_words) {
  uint32_t pending = 0;
#pragma unroll
  for (int q = 0; q < kMaxW; q++) if (t + q * pollers < n_words) pending |= 1u << q;
  for (uint32_t round = 0; round < kFusedSpinLimit; round++) {
    uint64_t w[kMaxW];
#pragma unroll
    for (int q = 0; q < kMaxW; q++)
      if ((pending >> q) & 1u) w[q] = __hip_atomic_load(words + t + q * pollers, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int q = 0; q < kMaxW; q++) {
      if (((pending >> q) & 1u) && (uint32_t)w[q] == tag) {
        s_words[t + q * pollers] = (uint32_t)(w[q] >> 32);
        pending &= ~(1u << q);
      }
    }
    if (__ballot(pending != 0) == 0) return true;
    if (round < 16) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(6);   // back off: every poll goes out to memory
  }
  return false;
}

__device__ __forceinline__ double ld_shared_f64(const double* p) {   // written by another workgroup of this launch
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_shared_f64(double* p, double v) {   // read by other workgroups of this launch
  __hip_atomic_store(reinterpret_cast<uint64_t*>(p), (uint64_t)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// v through steps 0..j-1 with the operands of one side in LDS (su: the values every entry of this workgroup shares)
// and the other side in global memory at `own` (stride `pitch`): the update's own arithmetic, see chain_apply.
template <bool kSharedIsRow, int kBatch>
__device__ __forceinline__ double fused_chain_batch(double v, int s, const double* __restrict__ su, const double* __restrict__ own, int64_t pitch) {
  double o[kBatch];
#pragma unroll
  for (int q = 0; q < kBatch; q++) o[q] = own[(int64_t)(s + q) * pitch];
#pragma unroll
  for (int q = 0; q < kBatch; q++) {
    const double r = kSharedIsRow ? su[s + q] : o[q], c = kSharedIsRow ? o[q] : su[s + q];
    if (r != 0.0) v = flushz(__dadd_rn(v, __dmul_rn(c, r)));
  }
  return v;
}
template <bool kSharedIsRow>
__device__ __forceinline__ double fused_chain(double v, int j, const double* __restrict__ su, const double* __restrict__ own, int64_t pitch) {
  int s = 0;
  for (; s + 8 <= j; s += 8) v = fused_chain_batch<kSharedIsRow, 8>(v, s, su, own, pitch);
  if (s + 4 <= j) { v = fused_chain_batch<kSharedIsRow, 4>(v, s, su, own, pitch); s += 4; }
  if (s + 2 <= j) { v = fused_chain_batch<kSharedIsRow, 2>(v, s, su, own, pitch); s += 2; }
  if (s < j) v = fused_chain_batch<kSharedIsRow, 1>(v, s, su, own, pitch);
  return v;
}
// The same with this thread's side held in registers (h[s], s < j <= kHist; unrolled so that h stays in registers, four
// steps per uniform branch so that a short chain does not walk the whole history).
template <bool kSharedIsRow, int kHist>
__device__ __forceinline__ double fused_chain_reg(double v, int j, const double* __restrict__ su, const double (&h)[kHist]) {
#pragma unroll
  for (int s0 = 0; s0 < kHist; s0 += 4) {
    if (s0 < j) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int s = s0 + q;
        if (s < kHist) {
          const double r = kSharedIsRow ? su[s] : h[s], c = kSharedIsRow ? h[s] : su[s];
          if (s < j && r != 0.0) v = flushz(__dadd_rn(v, __dmul_rn(c, r)));
        }
      }
    }
  }
  return v;
}

// What the first wavefront of a workgroup leaves in LDS for the others after folding the records of a kind.
struct FusedFold {
  double v;        // winning value (reduced cost / ratio)
  double a;        // ratio records: the rebuilt entering column at the winning row; b: the right-hand side there
  double b;
  double ce;       // ratio records: reduced cost of the entering column
  int64_t idx;     // winning column / row, -1 none
  int64_t d_old;   // ratio records: basis[] entry of the winning row
  int32_t l;       // ratio records: entering column
  int32_t flags;   // 1: a poll gave up; 2: the chain ends before this step; 4: the entering column of this step holds an
                   // entry of kDirtyBound or more (PivotState::dirty): the step is set up, the chain ends behind it
};

// kReg: every thread owns at most ONE row (row workgroups) or column (column workgroups) and keeps what the steps so far
// left for it in registers (col'_s of its row / row'_s of its column, the right-hand side entry, the objective entry);
// otherwise a thread walks several rows / columns and reads those values back from memory.  kHist >= depth: entries
// of the register history (24 leave room for two workgroups of 512 threads per CU).
template <int kThreads, bool kReg, int kHist>
__global__ __launch_bounds__(kThreads, kThreads == 256 ? 3 : kThreads == 512 ? 2 : 1) void k_chain_fused(
    Tab T, Shape sh, PivotState* __restrict__ st, ChainStep* __restrict__ steps, int depth, int n_rb,
    double* __restrict__ pv, int64_t* __restrict__ pi, int n_price_part,
    double* __restrict__ colp, double* __restrict__ rowp, int64_t rstride, double* __restrict__ rhsv, double* __restrict__ colv,
    double* __restrict__ obj, const uint8_t* __restrict__ is_basic, const int64_t* __restrict__ basis,
    int64_t* __restrict__ trace, int64_t trace_cap, uint64_t* __restrict__ rec_ratio, uint64_t* __restrict__ rec_price, uint32_t epoch) {
  __shared__ double s_v[kThreads / kWave];
  __shared__ int64_t s_i[kThreads / kWave];
  __shared__ __attribute__((aligned(16))) double s_u[kMaxChain];   // the operands all entries of the workgroup share: row'_s of the entering column / col'_s at the pivot row
  __shared__ FusedFold s_f;
  __shared__ uint32_t s_words[6 * kThreads];   // the data halves of a polled set of records
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int my_set = (int)(blockIdx.x % kFusedReplicas);
  const int64_t ratio_region = fused_region_words(n_rb, kFusedRatioWords), price_region = fused_region_words((int)gridDim.x - n_rb, kFusedPriceWords);
  const bool wave0 = tid < kWave;
  const bool row_wg = (int)blockIdx.x < n_rb;
  const int n_cb = (int)gridDim.x - n_rb, cb = (int)blockIdx.x - n_rb;
  const int ncl = sh.nloc + 1;
  // ---- entry state: not written by this launch (a workgroup that gives up marks the status, nothing else)
  const int32_t status = st->status;
  const int64_t it0 = st->iter, max_iter = st->max_iter;   // it0 counts pivot 0 of the pass
  const int32_t l_e0 = st->l_e, l_d0 = st->l_d;
  int64_t q_e = -1, q_r = -1;      // step `lane` (lane 0: pivot 0): entering variable and pivot row
  if (lane == 0) { q_e = st->e; q_r = st->r; }
  if (status != ST_RUNNING) return;
  if (st->dirty_step == 1) return;   // pivot 0 ends the chain (PivotState::dirty); this launch itself only ever writes 2 and more
  const bool keeper = cb == 0 && tid == 0;
  if (keeper) for (int s = 1; s < depth; s++) { steps[s].done = 0; steps[s].e = -1; }
  // step j exists budget-wise while it0 + (j - 1) < max_iter
  const int64_t room = max_iter - it0;
  const int limit = room >= (int64_t)depth ? depth : (int)(room > 0 ? room + 1 : 1);

  // The first wavefront folds the ratio records of step j (smallest ratio, lowest row on ties; reference
  // lp_simplex_tableau.cc:207-227) and fetches what hangs on the winning row; the result lands in s_f.
  auto fold_ratio_records = [&](int j, uint32_t tag, bool want_rhs, bool want_basis, bool want_col_r) {
    double best = DBL_MAX, ce2 = 0.0;
    int64_t r2 = -1;
    int32_t l2 = -1, flags = 0;
    const bool ok = poll_words<12>(rec_ratio + ((int64_t)(j & 1) * kFusedReplicas + my_set) * ratio_region, n_rb * kFusedRatioWords, tag, lane, kWave, s_words);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wavefront's own LDS writes, read back by other lanes
    if (ok) {
      for (int k = lane; k < n_rb; k += kWave) {
        const uint32_t* d = s_words + k * kFusedRatioWords;
        l2 = (int32_t)(d[3] & 0x7FFFFFFFu);
        ce2 = rec_f64(d[4], d[5]);
        if (d[2] == kFusedEnd) { flags |= 2; continue; }
        if (d[3] >> 31) flags |= 4;
        if (d[2] == kFusedNoRow) continue;
        const double ratio = rec_f64(d[0], d[1]);
        const int64_t row = (int64_t)d[2];
        if (r2 < 0 || ratio < best || (ratio == best && row < r2)) { best = ratio; r2 = row; }
      }
    }
    flags = (ok ? 0 : 1) | (__ballot(flags & 2) ? 2 : 0) | (__ballot(flags & 4) ? 4 : 0);
    wave_arg<false>(best, r2);
    l2 = __builtin_amdgcn_readfirstlane(l2);       // lane 0 always holds a record, and every record names the same column
    ce2 = rec_f64((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__double_as_longlong(ce2)),
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)__double_as_longlong(ce2) >> 32)));
    if ((flags & 3) == 0 && r2 >= 0) {
      // what hangs on the pivot row: requested together
      if (want_col_r && lane < j) s_u[lane] = ld_shared_f64(colp + (int64_t)lane * sh.lda + r2);
      if (lane == 0) {
        s_f.a = ld_shared_f64(colv + (int64_t)(j & 1) * sh.lda + r2);
        if (want_rhs) s_f.b = ld_shared_f64(rhsv + (int64_t)(j - 1) * sh.lda + r2);
        if (want_basis) s_f.d_old = basis[r2];
      }
    }
    if (lane == 0) { s_f.v = best; s_f.idx = r2; s_f.l = l2; s_f.ce = ce2; s_f.flags = flags; }
  };

  if (row_wg) {
    const int64_t stride = (int64_t)n_rb * kThreads;
    const int64_t i0 = (int64_t)blockIdx.x * kThreads + tid;
    const bool have = i0 < sh.m;
    double h[kHist];                 // kReg: col'_s of this thread's row
    double rhs_reg = 0.0;            // kReg: its right-hand side entry after the steps so far
    if (kReg) {
#pragma unroll
      for (int s = 0; s < kHist; s++) h[s] = 0.0;
      if (have) { h[0] = colp[i0]; rhs_reg = rhsv[i0]; }
    }
    for (int j = 1; j < limit; j++) {
      const uint32_t tag = (epoch << 8) | (uint32_t)j;
      // this workgroup's record in every set of copies: lane c * kFusedRatioWords + x of the first wavefront writes word x of copy c
      uint64_t* mine = rec_ratio + ((int64_t)(j & 1) * kFusedReplicas + tid / kFusedRatioWords) * ratio_region + (int64_t)blockIdx.x * kFusedRatioWords + tid % kFusedRatioWords;
      const bool publisher = tid < kFusedReplicas * kFusedRatioWords;
      const double* __restrict__ rhs = rhsv + (int64_t)(j - 1) * sh.lda;
      LPAMD_CLK(j, 0, 0, blockIdx.x == 0);
      // -- pricing after step j-1 (reference lp_simplex_tableau.cc:188-198): the records of the column workgroups (there
      // may be several hundred), two per thread
      double ce = -DBL_MAX;
      int64_t lw = -1;
      bool ok = true;
      if (j == 1) {                         // left by the set-up of pivot 0
        fold_share<true, kThreads>(pv, pi, n_price_part, ce, lw);
      } else {
        ok = poll_words<6>(rec_price + ((int64_t)(j & 1) * kFusedReplicas + my_set) * price_region, n_cb * kFusedPriceWords, tag, tid, kThreads, s_words);
      }
      if (__syncthreads_or(!ok)) { if (tid == 0) st->status = ST_ERROR; return; }
      if (j > 1) {
        for (int k = tid; k < n_cb; k += kThreads) {
          const uint32_t* d = s_words + k * kFusedPriceWords;
          take_max(ce, lw, rec_f64(d[0], d[1]), (int64_t)(int32_t)d[2]);
        }
      }
      block_arg<true, kThreads>(ce, lw, s_v, s_i);
      if (lw < 0) {                       // optimal after step j-1: tell the column workgroups
        if (publisher) rec_put(mine, kFusedEnd, tag);
        return;
      }
      const int l = __builtin_amdgcn_readfirstlane((int)lw);
      if (tid < j) s_u[tid] = ld_shared_f64(rowp + (int64_t)tid * rstride + l);   // row'_s of the entering column
      __syncthreads();
      LPAMD_CLK(j, 0, 1, blockIdx.x == 0);
      // -- the entering column as steps 0..j-1 leave it; ratio test (reference lp_simplex_tableau.cc:207-227)
      // (two copies of the rebuilt column take turns: a workgroup that is still fetching the pivot element of step j
      // must not find the column of step j+1 there, and its owner may be that far ahead)
      const double* __restrict__ col = T.col(l);
      double* __restrict__ cv = colv + (int64_t)(j & 1) * sh.lda;
      double bv = DBL_MAX, a_reg = 0.0;
      int64_t bi = -1;
      bool big = false;
      if (kReg) {
        if (have) {
          a_reg = fused_chain_reg<true, kHist>(col[i0], j, s_u, h);
          big = !(fabs(a_reg) < kDirtyBound);
          st_shared_f64(cv + i0, a_reg);
          if (!((float)a_reg > -1e-6f)) { bv = rhs_reg / (-a_reg); bi = i0; }
        }
      } else {
        for (int64_t i = i0; i < sh.m; i += stride) {
          const double a = fused_chain<true>(col[i], j, s_u, colp + i, sh.lda);
          big = big || !(fabs(a) < kDirtyBound);
          st_shared_f64(cv + i, a);
          if ((float)a > -1e-6f) continue;
          const double ratio = ld_shared_f64(rhs + i) / (-a);
          if (bv > ratio) { bv = ratio; bi = i; }
        }
      }
      LPAMD_CLK(j, 0, 2, blockIdx.x == 0);
      wait_stores();                      // the rebuilt column is out before the record that may name one of its rows
      const bool any_big = __syncthreads_or(big) != 0;
      block_arg<false, kThreads>(bv, bi, s_v, s_i);
      if (publisher) {                    // every thread holds the workgroup's result
        const uint64_t vb = (uint64_t)__double_as_longlong(bv), cb2 = (uint64_t)__double_as_longlong(ce);
        const int x = tid % kFusedRatioWords;
        const uint32_t word = x == 0 ? (uint32_t)vb : x == 1 ? (uint32_t)(vb >> 32) : x == 2 ? (bi < 0 ? kFusedNoRow : (uint32_t)bi)
                            : x == 3 ? ((uint32_t)l | (any_big ? 0x80000000u : 0u)) : x == 4 ? (uint32_t)cb2 : (uint32_t)(cb2 >> 32);
        rec_put(mine, word, tag);
      }
      LPAMD_CLK(j, 0, 3, blockIdx.x == 0);
      // -- all rows' records: the pivot row, its pivot element and right-hand side
      if (wave0) fold_ratio_records(j, tag, true, false, false);
      __syncthreads();
      if (s_f.flags & 1) { if (tid == 0) st->status = ST_ERROR; return; }
      const int64_t r2 = s_f.idx;
      if (r2 < 0) return;                 // no eligible row after step j-1: found again, and reported, by the next iteration
      LPAMD_CLK(j, 0, 4, blockIdx.x == 0);
      const double a2 = s_f.a, rhs_r2 = s_f.b;
      const double f2 = -1.0 / a2;
      const double rp_rhs = __dmul_rn(rhs_r2, f2);   // row'_j of the right-hand-side column
      double* __restrict__ cj = colp + (int64_t)j * sh.lda;
      double* __restrict__ rhs_now = rhsv + (int64_t)j * sh.lda;
      if (kReg) {
        if (have) {
          const double cpi = (i0 == r2) ? __dadd_rn(1.0, a2) : a_reg;   // lp_simplex_tableau.cc:41-44
          st_shared_f64(cj + i0, cpi);
          if (rp_rhs != 0.0) rhs_reg = flushz(__dadd_rn(rhs_reg, __dmul_rn(cpi, rp_rhs)));
          st_shared_f64(rhs_now + i0, rhs_reg);
#pragma unroll
          for (int s = 1; s < kHist; s++) if (s == j) h[s] = cpi;
        }
      } else {
        for (int64_t i = i0; i < sh.m; i += stride) {
          const double ci = ld_shared_f64(cv + i);
          const double rhs_i = ld_shared_f64(rhs + i);
          const double cpi = (i == r2) ? __dadd_rn(1.0, a2) : ci;   // lp_simplex_tableau.cc:41-44
          st_shared_f64(cj + i, cpi);
          st_shared_f64(rhs_now + i, rp_rhs != 0.0 ? flushz(__dadd_rn(rhs_i, __dmul_rn(cpi, rp_rhs))) : rhs_i);
        }
      }
      const bool ends_here = (s_f.flags & 4) != 0;
      wait_stores();       // col'_j and the right-hand side are out before this workgroup's next record
      __syncthreads();     // (and s_f / s_u are free again)
      LPAMD_CLK(j, 0, 5, blockIdx.x == 0);
      if (ends_here) return;   // PivotState::dirty: no step is set up behind this one
    }
    return;
  }

  // ---- column workgroups.  Basis flags of this thread's columns (column base + q * pitch is bit q) as after pivot 0;
  // every step then sets / clears one bit at most.  The host keeps the columns per thread within 32.
  const int l_base = cb * kThreads + tid, l_pitch = n_cb * kThreads;
  uint32_t basic_bits = 0;
  {
    int q = 0;
    for (int l = l_base; l < sh.nloc; l += l_pitch, q++) {
      const bool basic = l == l_e0 ? true : (l == l_d0 ? false : is_basic[l] != 0);
      basic_bits |= (basic ? 1u : 0u) << q;
    }
  }
  double h[kHist];                   // kReg: row'_s of this thread's column
  double obj_reg = 0.0;              // kReg: its objective entry
  if (kReg) {
#pragma unroll
    for (int s = 0; s < kHist; s++) h[s] = 0.0;
    if (l_base < ncl) { h[0] = rowp[l_base]; obj_reg = obj[l_base]; }
  }
  for (int j = 1; j < limit; j++) {
    const uint32_t tag = (epoch << 8) | (uint32_t)j;
    LPAMD_CLK(j, 2, 0, cb == 0);
    LPAMD_CLK(j, 3, 0, cb == n_cb - 1);
    if (wave0) fold_ratio_records(j, tag, false, true, true);
    __syncthreads();
    if (s_f.flags & 1) { if (tid == 0) st->status = ST_ERROR; return; }
    if (s_f.flags & 2) return;            // optimal after step j-1
    const int64_t r2 = s_f.idx;
    if (r2 < 0) return;                   // no eligible row
    LPAMD_CLK(j, 2, 1, cb == 0);
    LPAMD_CLK(j, 3, 1, cb == n_cb - 1);
    const double a2 = s_f.a, ce2 = s_f.ce;
    const int32_t l_e2 = s_f.l;
    const bool ends_here = (s_f.flags & 4) != 0;
    const double f2 = -1.0 / a2;                     // lp_simplex_tableau.cc:40
    const double g2 = __dmul_rn(ce2, f2);            // lp_simplex_tableau.cc:34-35
    LPAMD_CLK(j, 2, 2, cb == 0);
    // the basic variable of row r2 after steps 0..j-1: the last step so far that pivoted on this row, else what the pass before left
    int64_t d2 = s_f.d_old;
    {
      const unsigned long long hit = __ballot(lane < j && q_r == r2);
      if (hit) d2 = readlane_i64(q_e, 63 - __clzll((long long)hit));
    }
    const int64_t e2 = gcol_of((int64_t)l_e2 * sh.world + sh.rank, sh.N);
    int l_d2 = -1;
    if (d2 >= 0 && d2 != sh.N) {
      const int64_t o = ordinal_of(d2, sh.N);
      if (ord_rank(o, sh.world) == sh.rank) l_d2 = ord_local(o, sh.world);
    }
    double nbv = -DBL_MAX;
    int64_t nbi = -1;
    if (kReg) {
      const int l = l_base;
      if (l < ncl) {
        const double tv = T.col(l)[r2];
        LPAMD_CLK(j, 2, 5, cb == 0 && tv == tv);
        const double t = fused_chain_reg<false, kHist>(tv, j, s_u, h);
        LPAMD_CLK(j, 2, 6, cb == 0 && t == t);
        const double rp2 = __dmul_rn(t, f2);
        double o = flushz(__dadd_rn(obj_reg, __dmul_rn(t, g2)));   // :36
        if (l == l_e2) o = 0.0;                                      // :37
        obj_reg = o;
        obj[l] = o;
        st_shared_f64(rowp + (int64_t)j * rstride + l, rp2);
#pragma unroll
        for (int s = 1; s < kHist; s++) if (s == j) h[s] = rp2;
        if (l < sh.nloc) {
          if (l == l_e2) basic_bits |= 1u; else if (l == l_d2) basic_bits &= ~1u;
          if (!(basic_bits & 1u) && is_pos(o)) { nbv = o; nbi = l; }
        }
      }
    } else {
      int q = 0;
      for (int l = l_base; l < ncl; l += l_pitch, q++) {
        const double t = fused_chain<false>(T.col(l)[r2], j, s_u, rowp + l, rstride);
        const double rp2 = __dmul_rn(t, f2);
        double o = flushz(__dadd_rn(obj[l], __dmul_rn(t, g2)));   // :36
        if (l == l_e2) o = 0.0;                                     // :37
        obj[l] = o;
        st_shared_f64(rowp + (int64_t)j * rstride + l, rp2);
        if (l < sh.nloc) {
          if (l == l_e2) basic_bits |= 1u << q; else if (l == l_d2) basic_bits &= ~(1u << q);
          if (!((basic_bits >> q) & 1u) && is_pos(o) && o > nbv) { nbv = o; nbi = l; }
        }
      }
    }
    LPAMD_CLK(j, 2, 3, cb == 0);
    LPAMD_CLK(j, 3, 3, cb == n_cb - 1);
    wait_stores();                        // row'_j is out before this workgroup's record
    block_arg<true, kThreads>(nbv, nbi, s_v, s_i);
    if (tid < kFusedReplicas * kFusedPriceWords) {   // every thread holds the workgroup's result: lane c * kFusedPriceWords + x writes word x of copy c
      const int x = tid % kFusedPriceWords;
      const uint64_t vb = (uint64_t)__double_as_longlong(nbv);
      rec_put(rec_price + ((int64_t)((j + 1) & 1) * kFusedReplicas + tid / kFusedPriceWords) * price_region + (int64_t)cb * kFusedPriceWords + x,
              x == 0 ? (uint32_t)vb : x == 1 ? (uint32_t)(vb >> 32) : (uint32_t)(int32_t)nbi, (epoch << 8) | (uint32_t)(j + 1));
    }
    if (tid == 0) {
      // the same result where the first kernel of the next pass looks for it (n_price_part entries)
      pv[cb] = nbv;
      pi[cb] = nbi;
      for (int k = cb + n_cb; k < n_price_part; k += n_cb) { pv[k] = -DBL_MAX; pi[k] = -1; }
    }
    LPAMD_CLK(j, 2, 4, cb == 0);
    if (lane == j) { q_e = e2; q_r = r2; }
    if (keeper) {
      ChainStep z;
      z.done = 1; z.slot = 0; z.l_in = l_e2; z.l_out = l_d2;
      z.e = e2; z.r = r2; z.d = d2; z.a = a2; z.ce = ce2; z.f = f2; z.g = g2;
      steps[j] = z;
      const int64_t it = it0 + (j - 1);
      if (it < trace_cap) { trace[2 * it] = e2; trace[2 * it + 1] = r2; }
      if (ends_here) st->dirty_step = j + 1;
    }
    if (ends_here) return;   // PivotState::dirty: no step is set up behind this one
  }
}

}  // namespace lpamd
