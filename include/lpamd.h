/*
 * lpamd -- C ABI of the MI355X dense simplex-tableau engine.
 *
 * This is the drop-in seam for the reference's tableau path
 * (`solver <file> simplex_tableau [max_cost|max_reduction]`).  The reference
 * has no FFI of its own: LPModel holds raw `Tableau<real_t>*` / `List<real_t>*`
 * pointers into a header-only template library (reference lp.h:273-279) and
 * calls its methods from lp_simplex_tableau.cc.  Each entry point below names
 * the reference interface whose work it takes over.
 *
 * Conventions
 *  - plain C types only; every function returns 0 on success and a negative
 *    LPAMD_E_* code on failure, and never throws.  lpamd_last_error() gives the
 *    message of the calling thread's last failure.
 *  - solver outcomes travel through an `int* result` out-parameter using the
 *    reference's `enum Result` values (reference lp.h:32-37), plus LPAMD_LIMIT.
 *  - columns are addressed by GLOBAL column id: 0..N-1 = variables in
 *    Variable::operator< order (reference base.cc:11-29, lp.cc:158-166),
 *    N = right-hand-side column (`constant_index_`, reference lp.cc:167),
 *    N+1 = phase-1 artificial column (reference lp_simplex_tableau.cc:70).
 *  - host matrices are column-major fp64 with leading dimension `ld` >= m and
 *    N+2 columns.  Host buffers are borrowed for the duration of the call;
 *    outputs go into caller-allocated buffers.
 *  - an engine is driven by one host thread at a time.  It owns its device
 *    memory (HBM) until lpamd_destroy.
 *  - multi-GPU: one engine per process/GPU holds the columns whose ordinal
 *    (global id, with the RHS column skipped) is congruent to `rank` modulo
 *    `world`, plus a replica of the RHS column.  Either the library runs the shards of an
 *    LP itself, RCCL collectives included (lpamd_group_*), or the caller drives the engines
 *    step by step and moves the exchange buffers between ranks with collectives of its own.
 */
#ifndef LPAMD_H_
#define LPAMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lpamd_engine lpamd_engine;

/* reference lp.h:32-37 (enum Result) + one extra value for a pivot budget */
enum { LPAMD_ERROR = 0, LPAMD_UNBOUNDED = 1, LPAMD_NOSOLUTION = 2, LPAMD_SOLVED = 3, LPAMD_LIMIT = 4 };
/* reference lp.h:74-79 (enum LPModel::PivotingStrategy) */
enum { LPAMD_PIVOTING_UNKNOWN = 0, LPAMD_SMALLEST_SUBSCRIPT = 1, LPAMD_MAX_COST = 2, LPAMD_MAX_REDUCTION = 3 };
/* status codes */
enum {
  LPAMD_OK = 0,
  LPAMD_E_ARG = -1,      /* bad argument / shape mismatch */
  LPAMD_E_HIP = -2,      /* a HIP runtime call failed */
  LPAMD_E_STATE = -3,    /* call not valid in the engine's current state */
  LPAMD_E_NODEVICE = -4, /* no usable GPU */
  LPAMD_E_UNSUPPORTED = -5
};

/* size in bytes of one candidate record / of the pivot-column exchange buffer */
#define LPAMD_CANDIDATE_BYTES 32
int64_t lpamd_exchange_bytes(int64_t m);

const char* lpamd_last_error(void);
/* version string of the library; also proves the symbol table loads without a GPU */
const char* lpamd_version(void);

/* ---- lifetime ----------------------------------------------------------
 * Replaces `new Tableau<real_t>(rows, cols, format)`, `new List<real_t>()` and
 * `new bool[...]` in LPModel::ToTableau (reference lp.cc:147-197): allocates
 * the device-resident shard for an m-row tableau over N variables.
 * `hip_stream` (a hipStream_t) may be NULL: the engine then creates its own. */
int lpamd_create(lpamd_engine** out, int64_t m, int64_t N, int32_t rank, int32_t world, void* hip_stream);
int lpamd_destroy(lpamd_engine* eng);

/* ---- data in / out ------------------------------------------------------
 * Replaces the row/objective appends of LPModel::ToTableau (reference
 * lp.cc:171-196) and the `tableau_is_base_variable_` fill (lp.cc:156-166).
 * T: m x (N+2) column-major, obj: N+2, is_basic: N+2 (entries N, N+1 ignored). */
int lpamd_upload(lpamd_engine* eng, const double* T, int64_t ld, const double* obj, const uint8_t* is_basic);
/* Fills the shard on the device with the synthetic dense LP used by the
 * benchmark (m rows, n structural + m slack columns, N must equal n + m);
 * entry (i,j) depends only on (seed, i, j).  No reference counterpart. */
int lpamd_fill_synthetic(lpamd_engine* eng, int64_t n_struct, uint64_t seed);
/* Reads the shard back (what LPModel::PrintTableau iterates, reference
 * lp.cc:199-219).  Columns owned by other ranks are left untouched. Any of
 * T / obj / is_basic may be NULL. */
int lpamd_download(const lpamd_engine* eng, double* T, int64_t ld, double* obj, uint8_t* is_basic);
int lpamd_read_column(const lpamd_engine* eng, int64_t col, double* out_m);   /* Tableau::Col(j)  */
int lpamd_read_row(const lpamd_engine* eng, int64_t row, double* out_Np2);    /* Tableau::Row(i), local columns only */
int lpamd_write_objective(lpamd_engine* eng, const double* obj_Np2);          /* swaps opt_obj_tableau_, reference lp_simplex_tableau.cc:76-79,144 */
int lpamd_basis_of_row(const lpamd_engine* eng, int64_t* out_m);              /* basic column of every row */

/* ---- the hot-path steps, one call each (every call synchronises) ---------
 * needTableauInitialization, reference lp_simplex_tableau.cc:54-62 */
int lpamd_need_init(lpamd_engine* eng, int32_t* out);
/* MAX_COST pricing loop, reference lp_simplex_tableau.cc:188-198; *e = -1 if none */
int lpamd_price_max_cost(lpamd_engine* eng, int64_t* e);
/* MAX_COST ratio test, reference lp_simplex_tableau.cc:207-227; *r = *d = -1 if unbounded */
int lpamd_ratio_test(lpamd_engine* eng, int64_t e, int64_t* r, int64_t* d);
/* MAX_REDUCTION pricing, reference lp_simplex_tableau.cc:235-364 */
int lpamd_price_max_reduction(lpamd_engine* eng, int64_t* e, int64_t* r, int64_t* d);
/* LPModel::TableauPivot(base, non_base, constraint_id), reference
 * lp_simplex_tableau.cc:10-52 (d = column of `base`, e = column of `non_base`) */
int lpamd_pivot(lpamd_engine* eng, int64_t d, int64_t e, int64_t r);
/* LPModel::TableauSimplexInitialize, reference lp_simplex_tableau.cc:64-171 */
int lpamd_phase1(lpamd_engine* eng, int32_t strategy, int32_t* result);
/* LPModel::TableauSimplexSolve, reference lp_simplex_tableau.cc:173-379
 * (phase 1 included).  max_pivots < 0: no budget. */
int lpamd_solve(lpamd_engine* eng, int32_t strategy, int64_t max_pivots, int32_t* result, int64_t* n_pivots);
/* The `while (true)` body of TableauSimplexSolve (reference
 * lp_simplex_tableau.cc:181-377) run device-resident for at most k pivots with
 * no host round trip inside; *result = LPAMD_LIMIT if the budget ran out. */
int lpamd_run_pivots(lpamd_engine* eng, int32_t strategy, int64_t k, int32_t* result, int64_t* done);
/* GetTableauSimplexSolution's per-column values (reference
 * lp_simplex_tableau.cc:381-399): x[j] for every global column (0 for
 * non-basic, RHS and other ranks' columns); and the objective constant read by
 * GetTableauSimplexOptimum (reference lp_simplex_tableau.cc:412-414). */
int lpamd_column_values(lpamd_engine* eng, double* x_Np2);
int lpamd_objective_constant(lpamd_engine* eng, double* out);

/* (e, r) of every pivot applied since creation, in order. */
int lpamd_pivot_count(const lpamd_engine* eng, int64_t* n);
int lpamd_trace(const lpamd_engine* eng, int64_t* e, int64_t* r, int64_t cap, int64_t* n);

/* HIP-event timing of the rank-1 update kernel (off by default).  active_columns: columns the
 * pivots of one launch touch, summed over its pivots and averaged over the launches since
 * profiling was switched on (a launch may apply a chain of pivots, see lpamd_pass_stats). */
int lpamd_set_profiling(lpamd_engine* eng, int32_t on);
int lpamd_update_stats(lpamd_engine* eng, double* total_ms, int64_t* launches, int64_t* active_columns);
/* The device loop of lpamd_run_pivots / lpamd_solve applies a chain of pivots per pass over the
 * tableau wherever a pass is bound by memory (m > 2048 on one GPU, every column shard), for both
 * strategies: up to 32 per pass; 16 for shards that fit the last-level cache, 26 for column
 * shards, and on one GPU with an HBM-sized tableau the depth is measured when the engine is
 * created (17..24 for MAX_COST, 28 for MAX_REDUCTION).  LPAMD_CHAIN=1..32 sets it by hand (1: one
 * pivot per pass), lpamd_loop_depth reports it.  The column and the row of every further pivot
 * of a pass are rebuilt from the col'/row' vectors of the pivots before it (the arithmetic of
 * reference lp_simplex_tableau.cc:33-46, entry by entry in the same order), so the results are
 * those of one pass per pivot.  For the last pass: the pivots it applied, the columns it
 * streamed, and the sum of the columns the pivots touch one by one (what lpamd_update_stats
 * reports as active_columns). */
int lpamd_pass_stats(lpamd_engine* eng, int32_t* pivots, int64_t* streamed_columns, int64_t* touched_columns);
/* Totals over all passes since lpamd_set_profiling(eng, 1): milliseconds inside the update kernel (HIP events on
 * the engine's stream), passes (= launches of that kernel), pivots they applied, columns they streamed (a column
 * is read and written once per pass: 16*m bytes of HBM traffic each) and columns the pivots touch one by one
 * (the algorithmic 16*m bytes per column and pivot).  Any pointer may be NULL. */
int lpamd_profile_totals(lpamd_engine* eng, double* total_ms, int64_t* passes, int64_t* pivots, int64_t* streamed_columns,
                         int64_t* touched_columns);
/* MAX_REDUCTION pricing from bounds (long columns, one GPU): pricings done since the engine was created and the
 * candidate columns they had to evaluate exactly, over all rows (the reference evaluates every candidate column
 * every time, lp_simplex_tableau.cc:275-309). */
int lpamd_pricing_stats(lpamd_engine* eng, int64_t* pricings, int64_t* columns_evaluated);
/* Pivots one pass of the device loop (lpamd_run_pivots / lpamd_solve) applies at most for `strategy`. */
int lpamd_loop_depth(const lpamd_engine* eng, int32_t strategy, int32_t* depth);

/* ---- multi-GPU steps (column shards; the caller runs the collectives) ----
 * These split one iteration of reference lp_simplex_tableau.cc:181-377 at the
 * two points where column shards must talk.
 * 1. every rank: best local entering candidate -> 32-byte record in device
 *    memory `cand_dev` (asynchronous on the engine's stream). */
int lpamd_local_candidate(lpamd_engine* eng, int32_t strategy, void* cand_dev);
/* 2. caller all-gathers the records; every rank: pick the global winner with
 *    the reference's lowest-index tie-break.  Synchronises (reads world*32 B).
 *    *result = LPAMD_LIMIT while running, else the terminal Result. */
int lpamd_select_candidate(lpamd_engine* eng, int32_t strategy, const void* cands_dev, int32_t world,
                           int64_t* e, int64_t* r, int32_t* owner_rank, int32_t* result);
/*    The fold itself, on `world` records in HOST memory; touches no device, so it
 *    also runs on a machine without a GPU.  winner_out (32 B) may be NULL. */
int lpamd_fold_candidates(int32_t strategy, const void* records_host, int32_t world, int64_t N, int64_t* e, int64_t* r,
                          int32_t* owner_rank, int32_t* result, void* winner_out);
/* 3. owner of column e: copy it into the exchange buffer (asynchronous). */
int lpamd_export_column(lpamd_engine* eng, int64_t e, void* xbuf_dev);
/* 4. caller broadcasts xbuf from the owner; every rank: ratio test on the
 *    broadcast column (or the given row for MAX_REDUCTION), then the rank-1
 *    update of its shard and RHS replica (asynchronous). */
int lpamd_apply_pivot(lpamd_engine* eng, int32_t strategy, int64_t e, int64_t r, const void* xbuf_dev);
/* Host-free variant of steps 2 and 3, for a loop that never waits for the host:
 * 2'. every rank folds the gathered records on the device (same rule as
 *     lpamd_fold_candidates) and keeps the winner; a terminal outcome is left in
 *     the engine's status (asynchronous).
 * 3'. every rank fills the exchange buffer with the winning column if it owns it,
 *     with zeros otherwise (asynchronous); the caller then sum-all-reduces the
 *     buffers as 64-bit integers, which hands every rank the exact column, and
 *     calls lpamd_apply_pivot with e = r = -1 (they are read on the device). */
int lpamd_select_candidate_device(lpamd_engine* eng, int32_t strategy, const void* cands_dev, int32_t world);
int lpamd_export_column_or_zero(lpamd_engine* eng, void* xbuf_dev);
/* Look-ahead variant for MAX_COST (reference lp_simplex_tableau.cc:188-227 + TableauPivot :10-52,
 * same results): the exchange of pivot k+1 travels on a second stream while the rank-1 update
 * of pivot k streams the shard, which takes both collectives off the critical path.  This
 * works because the set-up of pivot k already leaves the objective row of iteration k+1, so
 * every rank knows its next candidate before its update has run.  Per pivot, on the engine's
 * stream unless stated otherwise:
 * a. lpamd_lookahead_setup: ratio test on the delivered column `xbuf_dev` for the winner kept
 *    by step d, then the set-up of the pivot (scalars, col', row', objective row, bookkeeping)
 *    WITHOUT the rank-1 update;
 * b. lpamd_lookahead_export: best local candidate of the NEXT iteration -> `cand_dev` (32-byte
 *    record), and that column as the pending update will leave it -> `xlocal_dev` (the tableau
 *    is not written; with no update pending the column is copied as stored);
 * c. lpamd_lookahead_update: the rank-1 update of the pivot set up in step a;
 * d. meanwhile, on `hip_stream` (ordered after step b by the caller): the caller all-gathers the
 *    records, lpamd_lookahead_select folds them, keeps the winner inside the engine and fills
 *    `xbuf_dev` with `xlocal_dev` on the winner's rank and zeros elsewhere; the caller
 *    sum-all-reduces `xbuf_dev` as 64-bit integers and orders step a of the next pivot after it.
 * A run starts with steps b and d.  All four calls are asynchronous; the exchange buffers
 * hold lpamd_exchange_bytes(m) bytes. */
int lpamd_lookahead_setup(lpamd_engine* eng, const void* xbuf_dev);
int lpamd_lookahead_export(lpamd_engine* eng, void* cand_dev, void* xlocal_dev);
int lpamd_lookahead_update(lpamd_engine* eng);
int lpamd_lookahead_select(lpamd_engine* eng, const void* cands_dev, int32_t world, const void* xlocal_dev, void* xbuf_dev,
                           void* hip_stream);
/* Several pivots per pass over column shards (MAX_COST; same arithmetic as reference
 * lp_simplex_tableau.cc:10-52 and 188-227 pivot by pivot, same results).  After step a has set a
 * pivot up, the next one can be set up as well before any rank-1 update has run: step b exports
 * the next local candidate as ALL pivots set up so far will leave it, step d delivers the winner
 * (this time the engine's stream waits for it), and lpamd_chain_step runs the ratio test on the
 * delivered column and sets that pivot up too.  Up to lpamd_chain_depth pivots (step a included)
 * may be set up this way; step c then applies all of them in ONE pass over the shard.  A chain
 * that cannot go on (budget, optimum, no eligible row) turns the remaining lpamd_chain_step calls
 * into no-ops and the next step a reports the outcome. */
int lpamd_chain_depth(const lpamd_engine* eng, int32_t* depth);
int lpamd_chain_step(lpamd_engine* eng, void* xbuf_dev);
/* Pivot budget of the step-wise loop: after max_pivots more pivots (< 0: no limit)
 * the device steps turn into no-ops.  Also re-arms a loop that hit its budget.
 * lpamd_poll reports LPAMD_LIMIT both while running and when the budget is
 * spent; lpamd_pivot_count tells how far the loop got.  Synchronises. */
int lpamd_set_pivot_budget(lpamd_engine* eng, int64_t max_pivots);
/* Pieces of LPModel::TableauSimplexInitialize (reference lp_simplex_tableau.cc:64-171) for a
 * caller that runs the feasibility phase over column shards (on one GPU lpamd_phase1 does it
 * all).  The scans of that phase work on single rows, columns and the objective row, which
 * the caller assembles from the shards with lpamd_read_row / lpamd_read_column /
 * lpamd_download (each rank fills the entries of its own columns) and hands back with
 * lpamd_write_objective.
 * - lpamd_fill_column: sets every entry of a column (the artificial column: all ones at :82-86,
 *   removed at :143); a no-op on ranks that do not own it.
 * - lpamd_set_basic_flag: tableau_is_base_variable_[col] = flag (:71); no-op on non-owners.
 * - lpamd_pivot_shard: LPModel::TableauPivot(base, non_base, r) (:10-52) on a shard: d = column
 *   of `base` (< 0: none), e = column of `non_base`, ce = objective coefficient of e, xbuf_dev =
 *   column e as exported by its owner (lpamd_export_column) and delivered to every rank (may be
 *   NULL on the owner).  Synchronises. */
int lpamd_fill_column(lpamd_engine* eng, int64_t col, double value);
int lpamd_set_basic_flag(lpamd_engine* eng, int64_t col, int32_t flag);
int lpamd_pivot_shard(lpamd_engine* eng, int64_t d, int64_t e, int64_t r, double ce, const void* xbuf_dev);
/* terminal status after step 4 (synchronises): LPAMD_LIMIT = still running */
int lpamd_poll(lpamd_engine* eng, int32_t* result);

/* ---- column shards over several GPUs, RCCL inside the library --------------
 * One LP, its columns dealt round-robin over `world` shards (one lpamd_engine each, see the
 * multi-GPU steps above), driven through TableauSimplexSolve (reference
 * lp_simplex_tableau.cc:173-379) by the library itself.  The reference has no counterpart: its only
 * parallelism is OpenMP inside List::Map (reference lp_simplex_tableau.cc:253-309).  Per pivot the
 * shards trade ONE exchange: every rank sends its best local candidate, record and column
 * and (MAX_COST) the ratio-test partials it took on that column (world * (64 + 8.06 * lda) bytes
 * arrive per rank), every rank folds the records with the reference's tie-break (largest key,
 * lowest column) and takes the winner's column; both strategies set up several pivots per pass
 * over the shard (lpamd_group_chain), the exchange of the pass after travels on a second stream
 * beside the rank-1 update.  The exchange is an RCCL all-gather, or, where the ranks can map each
 * other's memory, stores into the peers' receive buffers from inside the launches
 * (lpamd_group_link_active).  Results equal the single-engine loop bit for bit.
 *
 * lpamd_group_create: ONE process drives n_shards GPUs (devices[k] for shard k, NULL = 0..n-1),
 * one host thread and one communicator per shard (ncclCommInitAll).  With
 * LPAMD_GROUP_LOCAL_EXCHANGE the shards trade through device copies instead of RCCL and may
 * share a GPU (RCCL refuses that): for tests on a one-GPU machine.
 * lpamd_group_create_rank: one process PER GPU; this process holds shard `rank` on the current
 * device and joins the communicator named by the 128-byte id that rank 0 made with
 * lpamd_comm_unique_id and the caller delivered (any channel).  Collective: every rank calls it.
 * Every lpamd_group_* call below is collective in that mode. */
typedef struct lpamd_group lpamd_group;
#define LPAMD_COMM_ID_BYTES 128
enum {
  LPAMD_GROUP_LOCAL_EXCHANGE = 1,
  /* measurement aid: only shard 0 of the n_shards-way deal is created and trades with itself (RCCL world of
   * one): the per-rank cost of an n_shards-GPU run on one GPU; the pivots are not those of the LP */
  LPAMD_GROUP_REHEARSE_RANK0 = 2
};
int lpamd_comm_unique_id(void* id_out);
int lpamd_group_create(lpamd_group** out, int64_t m, int64_t N, int32_t n_shards, const int32_t* devices, int32_t flags);
int lpamd_group_create_rank(lpamd_group** out, int64_t m, int64_t N, int32_t rank, int32_t world, const void* comm_id);
/* The same with the collective supplied by the CALLER instead of RCCL: a host function that hands every rank the
 * `bytes` bytes of every rank, in rank order (MPI_Allgather, a gloo all_gather, ...; returns 0 on success).  The library
 * stages through host memory around it, so this form is for machines or set-ups RCCL does not serve (two ranks on one
 * GPU, which RCCL refuses: that is how the exchange inside the launches is tested between two processes on a one-GPU
 * machine), not for speed; the steps of a MAX_COST pass still trade inside their launches where the ranks can map each
 * other's memory.  No counterpart in the reference. */
typedef int (*lpamd_host_allgather_fn)(void* ctx, const void* send, void* recv, int64_t bytes);
int lpamd_group_create_rank_hosted(lpamd_group** out, int64_t m, int64_t N, int32_t rank, int32_t world,
                                   lpamd_host_allgather_fn gather, void* ctx);
int lpamd_group_destroy(lpamd_group* g);
/* shards held by this process / shards of the LP; the engine of a local shard (owned by the group:
 * for traces, statistics and read-back through the lpamd_* calls, not for pivoting) */
int lpamd_group_shards(const lpamd_group* g, int32_t* n_local, int32_t* world);
/* ranks the RCCL communicator of the first local shard really holds (ncclCommCount; 0 with the in-process exchange):
 * what a multi-GPU run reports so that a mis-formed communicator is seen, not inferred */
int lpamd_group_comm_ranks(lpamd_group* g, int32_t* ranks);
/* 1 when the steps of a MAX_COST chain trade their slots INSIDE the launches (stores into the peers' receive buffers
 * over xGMI, tagged words, no collective; only the pass boundary keeps the all-gather), 0 when every step runs the
 * all-gather.  Decided when the group is formed: the ranks map each other's buffers (peer access inside one
 * process, IPC handles traded through the communicator otherwise), run a self-test and agree on the outcome;
 * LPAMD_SHARD_LINK=0 keeps the collective.  No counterpart in the reference (its tableau lives in one address space). */
int lpamd_group_link_active(lpamd_group* g, int32_t* active);
int lpamd_group_engine(lpamd_group* g, int32_t local_index, lpamd_engine** out);
/* pivots one pass over a shard sets up at most in the MAX_COST loop (LPAMD_SHARD_CHAIN overrides) */
int lpamd_group_chain(lpamd_group* g, int32_t* pivots_per_pass);
/* as lpamd_upload / lpamd_fill_synthetic / lpamd_download, every local shard taking or filling its own columns */
int lpamd_group_upload(lpamd_group* g, const double* T, int64_t ld, const double* obj, const uint8_t* is_basic);
int lpamd_group_fill_synthetic(lpamd_group* g, int64_t n_struct, uint64_t seed);
int lpamd_group_download(lpamd_group* g, double* T, int64_t ld, double* obj, uint8_t* is_basic);
/* needTableauInitialization / TableauSimplexInitialize / TableauSimplexSolve / the loop body / TableauPivot
 * (reference lp_simplex_tableau.cc:54-62, 64-171, 173-379, 181-377, 10-52) over the shards */
int lpamd_group_need_init(lpamd_group* g, int32_t* out);
int lpamd_group_phase1(lpamd_group* g, int32_t strategy, int32_t* result);
int lpamd_group_solve(lpamd_group* g, int32_t strategy, int64_t max_pivots, int32_t* result, int64_t* n_pivots);
int lpamd_group_run_pivots(lpamd_group* g, int32_t strategy, int64_t k, int32_t* result, int64_t* done);
int lpamd_group_pivot(lpamd_group* g, int64_t d, int64_t e, int64_t r);
/* as lpamd_column_values (assembled over all ranks) / lpamd_objective_constant / lpamd_pivot_count / lpamd_trace */
int lpamd_group_column_values(lpamd_group* g, double* x_Np2);
int lpamd_group_objective_constant(lpamd_group* g, double* out);
int lpamd_group_pivot_count(lpamd_group* g, int64_t* n);
int lpamd_group_trace(lpamd_group* g, int64_t* e, int64_t* r, int64_t cap, int64_t* n);

/* ---- allocation-free ratio test -------------------------------------------
 * No counterpart in the reference.  For callers that cannot allocate device memory: code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It is the
 * MAX_COST ratio test of lpamd_ratio_test (reference lp_simplex_tableau.cc:207-227: rows with
 * (float)a <= -1e-6f, ratio b / (-a), strict minimum, lowest row on ties) on two device vectors
 * the caller owns: col_dev = the entering column, rhs_dev = the right-hand side, m doubles each.
 * *row_out_dev (device memory) receives the pivot row, or -1 if no row is eligible (unbounded);
 * *ratio_out_dev (device memory, may be NULL) its ratio.  The call is asynchronous on
 * `hip_stream` (NULL = the default stream of the current device) and calls no hipMalloc /
 * hipFree: the per-workgroup partial results live in a 16 KiB array inside the library, one copy
 * per GPU.  Limits: 1 <= m <= 16777216 (LPAMD_E_ARG beyond).  All calls on one GPU share that
 * array: calls that overlap on one GPU (two streams, two host threads) race; calls issued on one
 * stream are ordered. */
int lpamd_ratio_test_noalloc(const double* col_dev, const double* rhs_dev, int64_t m, int64_t* row_out_dev,
                             double* ratio_out_dev, void* hip_stream);

/* ---- revised simplex on the tableau (one GPU) ------------------------------
 * The reference's other tableau algorithm, `solver <file> revised_simplex_tableau`
 * (reference solver.cc:139-144): the tableau stays as uploaded and a dense basis
 * inverse is carried along.  The handle owns a row-major copy of the tableau,
 * the inverse (stored transposed) and all vectors; the loop runs on the device. */
typedef struct lpamd_rev_engine lpamd_rev_engine;
/* m constraint rows, N variable columns (right-hand side at column N).
 * hip_stream as in lpamd_create.  Device memory is taken at upload time. */
int lpamd_rev_create(lpamd_rev_engine** out, int64_t m, int64_t N, void* hip_stream);
int lpamd_rev_destroy(lpamd_rev_engine* eng);
/* T, obj, is_basic laid out as for lpamd_upload (columns 0..N are read; exactly m
 * basic columns, each with a single non-zero entry, as LPModel::ToTableau leaves
 * them, reference lp.cc:147-197).  opt_reverted: the model's objective was negated
 * on the way to standard form (reference lp.h opt_reverted_).  A negative
 * right-hand side reserves the m artificial columns of the feasibility phase. */
int lpamd_rev_upload(lpamd_rev_engine* eng, const double* T, int64_t ld, const double* obj, const uint8_t* is_basic,
                     int32_t opt_reverted);
/* The synthetic dense LP of lpamd_fill_synthetic, generated on the device in this
 * engine's layout (N == n_struct + m, slack basis, positive right-hand side). */
int lpamd_rev_fill_synthetic(lpamd_rev_engine* eng, int64_t n_struct, uint64_t seed);
/* LPModel::TableauRevisedSimplexSolve (reference lp_revised_simplex_tableau.cc:199-302)
 * including TableauRevisedSimplexInitialize (:92-197) and the pivot (:42-62).
 * max_pivots < 0: no limit; otherwise *result == LPAMD_LIMIT once that many pivots
 * (feasibility phase included) have been applied, and a further call with a larger
 * limit continues a run that stopped in the main phase. */
int lpamd_rev_solve(lpamd_rev_engine* eng, int64_t max_pivots, int32_t* result, int64_t* n_pivots);
/* GetTableauRevisedSimplexOptimum (reference :253-257, :304-306): sign already applied. */
int lpamd_rev_optimum(lpamd_rev_engine* eng, double* out);
/* Values of all N + 2 columns as in lpamd_column_values: basic_feasible_solution
 * scattered by basis_indices, zero elsewhere (reference :259-263). */
int lpamd_rev_column_values(lpamd_rev_engine* eng, double* x);
int lpamd_rev_pivot_count(lpamd_rev_engine* eng, int64_t* n);
/* Sizes of the exact-sparsity lists as of the last pivot of the last lpamd_rev_solve call (for
 * traffic accounting; no reference counterpart): rows of the inverse its rank-1 update visited,
 * rows of the tableau the pricing product read (non-zeros of c_B^T B^-1), rows of the inverse
 * that are no longer unit rows.  Any pointer may be NULL. */
int lpamd_rev_list_sizes(lpamd_rev_engine* eng, int64_t* n_update_rows, int64_t* n_pricing_rows, int64_t* n_dense_rows);
/* (entering column, leaving basis position) of every pivot so far */
int lpamd_rev_trace(lpamd_rev_engine* eng, int64_t* entering, int64_t* leaving_pos, int64_t cap, int64_t* n);
/* Debug/test access; any pointer may be NULL.  Binv, BinvT: m x m row-major
 * (basis_inverse of the reference and its transpose, the form the engine stores); xB = basic_feasible_solution,
 * cB = basis_coeff, basis = basis_indices (reference lp.h:281-284). */
int lpamd_rev_read_state(lpamd_rev_engine* eng, double* Binv, double* BinvT, double* xB, double* cB, int64_t* basis);

#ifdef __cplusplus
}
#endif
#endif /* LPAMD_H_ */
