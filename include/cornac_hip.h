/*
 * cornac_hip.h — C ABI of libcornac_hip.so, the MI355X (gfx950) backend for the
 * embedding-SGD + scoring hot path of PreferredAI/cornac.
 *
 * The reference has no FFI: its plug-in boundary is four Cython call sites.
 * Every entry point below names the reference interface it replaces (paths are
 * relative to the reference repository root).  The reference-side bindings a
 * maintainer would add are shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no C++ exceptions cross the boundary; every function returns an
 *     int status (0 = CORNAC_HIP_OK) and cornac_hip_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - host buffers are caller-owned; device buffers are library-owned unless a
 *     *_bind_device call hands in caller-owned device pointers (used by the
 *     multi-GPU driver so torch.distributed/RCCL can all-reduce them in place);
 *   - calls on one handle are not re-entrant (one HIP stream per handle);
 *     different handles may be driven from different threads / devices;
 *   - all calls block until the device work is complete unless stated otherwise.
 */
#ifndef CORNAC_HIP_H_
#define CORNAC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CORNAC_HIP_OK 0
#define CORNAC_HIP_ERR_INVALID 1   /* bad argument (message says which)            */
#define CORNAC_HIP_ERR_HIP 2       /* a HIP runtime call failed                    */
#define CORNAC_HIP_ERR_NO_DEVICE 3 /* no gfx950 device visible                     */
#define CORNAC_HIP_ERR_UNSUPPORTED 4

/* Execution mode.
 * DETERMINISTIC reproduces the reference's seeded run (seed != None forces
 * num_threads = 1, cornac/models/bpr/recom_bpr.pyx:132-133, recom_mf.py:124-125):
 * every update observes all earlier ones, in the reference's sample order.
 * HOGWILD is the throughput mode, the counterpart of the reference's racy
 * multi-thread path (num_threads > 1, recom_bpr.pyx:228-267, backend_cpu.pyx:62). */
#define CORNAC_HIP_MODE_DETERMINISTIC 0
#define CORNAC_HIP_MODE_HOGWILD 1

/* Negative-item population of the triplet sampler. */
#define CORNAC_HIP_NEG_UNIFORM 0    /* BPR : neg_item_ids = arange(num_items)  (recom_bpr.pyx:186)  */
#define CORNAC_HIP_NEG_POPULARITY 1 /* WBPR: neg_item_ids = X.indices          (recom_wbpr.pyx:135) */

const char *cornac_hip_last_error(void);
const char *cornac_hip_version(void);
int cornac_hip_device_count(int *count);
/* name (<=255 chars), CU count and HBM bytes of a device */
int cornac_hip_device_info(int device, char *name, int name_len, int *compute_units, int64_t *hbm_bytes);
/* Memory-system calibration of the box a benchmark runs on (bench.py reports it beside the scale leg, whose
 * throughput differs between boxes): over two scratch buffers of `bytes` each, out3 = {device-to-device copy GB/s
 * (read + written), streaming read GB/s, random 512-byte row gather GB/s} */
int cornac_hip_device_probe(int device, int64_t bytes, double *out3);

/* ------------------------------------------------------------------------- *
 * BPR / WBPR trainer.
 * Replaces: BPR._fit_sgd(rng_pos, rng_neg, num_threads, user_ids, item_ids,
 *           neg_item_ids, indptr, U, V, B) -> (correct, skipped)
 *           cornac/models/bpr/recom_bpr.pyx:208-269, its caller loop
 *           BPR.fit :188-201 and WBPR.fit cornac/models/bpr/recom_wbpr.pyx:127-142,
 *           the sampler RNGVector recom_bpr.pyx:54-62 and has_non_zero :46-51.
 * ------------------------------------------------------------------------- */
typedef struct cornac_hip_bpr *cornac_hip_bpr_t;

/* CSR interaction matrix = train_set.matrix (int32 indptr[n_users+1], int32
 * indices[nnz], sorted per row — cornac/data/dataset.py:226-235).  U is
 * [total_users, k], V is [total_items, k], B is [total_items] like BPR._init
 * (recom_bpr.pyx:145-152); n_users/n_items are the TRAIN counts used by the
 * sampler. */
int cornac_hip_bpr_create(cornac_hip_bpr_t *out, int device, int64_t n_users, int64_t n_items,
                          int64_t total_users, int64_t total_items, int k, const int32_t *indptr,
                          const int32_t *indices, int64_t nnz);
int cornac_hip_bpr_destroy(cornac_hip_bpr_t h);

/* host -> device / device -> host copies of the factor tables (fp32, C order).
 * Any pointer may be NULL to skip that table. */
int cornac_hip_bpr_set_factors(cornac_hip_bpr_t h, const float *U, const float *V, const float *B);
int cornac_hip_bpr_get_factors(cornac_hip_bpr_t h, float *U, float *V, float *B);

/* float64 tables.  `_fit_sgd` is a fused-type (`floating`) function (recom_bpr.pyx:211-214): a model given float64
 * U / V / Bi through init_params trains in double, every local of the step included (:219-224).  set_factors_f64 puts
 * the handle into that state (all three tables together; set_factors puts it back), fit_epochs_f64 runs the SEQUENTIAL
 * semantics (the deterministic engine with the mt19937 streams of seed_mt19937; lr / reg as doubles) — there is no
 * float64 throughput kernel: an unseeded float64 model is trained by this engine too (cornac_amd/bpr.py). */
int cornac_hip_bpr_set_factors_f64(cornac_hip_bpr_t h, const double *U, const double *V, const double *B);
int cornac_hip_bpr_get_factors_f64(cornac_hip_bpr_t h, double *U, double *V, double *B);
int cornac_hip_bpr_fit_epochs_f64(cornac_hip_bpr_t h, int n_epochs, double lr, double reg, int use_bias, int neg_population,
                                  int64_t *correct, int64_t *skipped);

/* Use caller-owned device buffers (same shapes) instead of the library's. */
int cornac_hip_bpr_bind_device(cornac_hip_bpr_t h, float *dU, float *dV, float *dB);
int cornac_hip_bpr_device_ptrs(cornac_hip_bpr_t h, float **dU, float **dV, float **dB);
/* Swap caller-owned item tables for other caller-owned ones WITHOUT synchronising the handle's stream (bind_device waits
 * for it): work already enqueued keeps its pointers, later work sees the new ones.  For drivers that move the item
 * table between launches — the ring conveyor of cornac_amd/dist.py binds the block that has just arrived. */
int cornac_hip_bpr_rebind_items(cornac_hip_bpr_t h, float *dV, float *dB);
/* run the handle's work on a caller-provided hipStream_t (NULL = the handle's own); waits for the previous stream */
int cornac_hip_bpr_set_stream(cornac_hip_bpr_t h, void *hip_stream);
/* the same without waiting for the work already queued on the previous stream: for callers that alternate between two
 * streams and order them with events themselves (cornac_amd/dist.py:RowShardedBprTrainer, stages A / B) */
int cornac_hip_bpr_switch_stream(cornac_hip_bpr_t h, void *hip_stream);

/* Deterministic-mode sampler state = the two boost::random::mt19937 engines of
 * RNGVector(1, ...) (recom_bpr.pyx:188-191): pass the ALREADY-DERIVED 32-bit
 * mt19937 seeds (RandomState(seed_a).randint(2**31)).  shared_stream != 0 is
 * WBPR's single generator used for both draws (recom_wbpr.pyx:131).  The
 * streams persist across fit_epochs calls exactly like the reference's
 * RNGVector objects persist across epochs. */
int cornac_hip_bpr_seed_mt19937(cornac_hip_bpr_t h, uint32_t mt_seed_pos, uint32_t mt_seed_neg, int shared_stream);
/* Hogwild-mode sampler state: counter-based Philox4x32-10 keyed by `seed`;
 * the epoch counter persists across calls. */
int cornac_hip_bpr_seed_hogwild(cornac_hip_bpr_t h, uint64_t seed);

/* Popularity-weighted negatives (CORNAC_HIP_NEG_POPULARITY) draw the item of a uniformly chosen entry of `neg_item_ids`
 * (recom_wbpr.pyx:135-139: neg_item_ids = X.indices, the handle's own interactions by default).  A handle that holds only
 * a SLICE of the users (multi-GPU) can be given the population of the whole matrix instead: items[n], any multiset of
 * train item ids whose multiplicities are (proportional to) the global item degrees; n = 0 restores the default.  With a
 * caller's population the popularity draw runs in the fused kernel (the LDS-bin form weights by the handle's own CSC). */
int cornac_hip_bpr_set_negative_population(cornac_hip_bpr_t h, const int32_t *items, int64_t n);

/* Run n_epochs epochs of nnz samples each.  correct/skipped accumulate the
 * reference's per-epoch counters over the epochs run (either may be NULL).
 * hogwild_flags (0 = default).  Bits 16..19 select the form of a hogwild call:
 * 0 = automatic, 1 = the fused kernel (every item-row update a device-scope fp32 atomic; also bit7), 2 = XCD strata
 * (csrc/bpr_strata.inc: 8 launches per epoch, an item row is touched by one XCD per launch and updated by plain
 * read-modify-write like the reference's threads), 3 = LDS-resident item bins (csrc/bpr_ldsbin.inc: the item rows
 * of a bin live in one CU's LDS for the epoch, exact updates, user rows by atomics).  Automatic = LDS bins when the
 * item table fits the LDS in at most max_rounds rounds with at least min_candidates items per bin, else XCD strata
 * for item tables of >= 2^20 rows, else the fused kernel.  A chunk of an epoch (hogwild_enqueue) runs in the same form:
 * an LDS-bin launch takes its share of every bin's draws, an XCD-strata chunk runs the partition phases that begin
 * inside it.  Popularity negatives (WBPR) have the LDS-bin form too — the negative is the item of a second interaction
 * drawn from the bin's own draw space, hot items dealt to every bin — but not the XCD-strata one; every experiment
 * switch below runs the fused kernel.
 * Experiment switches of the fused kernel: bit0 = plain (racy,
 * non-atomic, XCD-incoherent) row stores instead of fp32 atomics; bit1 = the
 * float4-per-lane row layout; bit2 = no user-row ownership (all rows atomic);
 * bit4 = (k in 33..64) the four sampling lanes 4g..4g+3 share one negative item and its row gets ONE combined atomic
 * update — a different joint distribution of the draws than the reference's, kept as a measured experiment;
 * bit6 = the segmented ("binned") item-update path. */
int cornac_hip_bpr_fit_epochs(cornac_hip_bpr_t h, int n_epochs, float lr, float reg, int use_bias, int neg_population,
                              int mode, int hogwild_flags, int64_t *correct, int64_t *skipped);
/* Same, but only enqueues `n_samples` hogwild samples (sample counter and
 * epoch continue) on the handle's stream and returns without synchronising;
 * counters are added into device memory and fetched by ..._sync.  Used by the
 * multi-GPU driver to interleave training chunks with RCCL reductions. */
int cornac_hip_bpr_hogwild_enqueue(cornac_hip_bpr_t h, int64_t n_samples, float lr, float reg, int use_bias,
                                   int neg_population, int hogwild_flags);
int cornac_hip_bpr_sync(cornac_hip_bpr_t h, int64_t *correct, int64_t *skipped);

/* Test hooks of the deterministic sampler: draw `n` values from stream 0/1
 * (boost uniform_int_distribution<long>(0, hi), uniform_int_distribution.hpp:188-227). */
int cornac_hip_bpr_debug_draw(cornac_hip_bpr_t h, int stream, uint64_t hi, int64_t n, int64_t *out);
/* Test hook of the hogwild sampler's user-row ownership: *n_waves receives the
 * width W of the persistent grid (0 when ownership is not used for this
 * handle); wave_ptr[W+1], own_u[nnz] (negative = shared heavy user, stored as
 * ~u), own_i[nnz] receive the tables when non-NULL. */
int cornac_hip_bpr_debug_ownership(cornac_hip_bpr_t h, int64_t *n_waves, int64_t *wave_ptr, int32_t *own_u,
                                   int32_t *own_i);
/* Tuning and inspection of the XCD-strata form.  strata_config: the hot item rows (atomic updates) are the most
 * popular ranks that are touched at least hot_min_mult_x100 / 100 times as often as the average row and together
 * receive at most hot_permille / 1000 of the item-row touches (defaults 200, 120); the item partitions are re-dealt
 * every rehash_period epochs (default 1).  strata_stats: out4 = {hot rows (-1: tables not built yet), workgroup
 * launches that were surplus on their XCD since create (an XCD handed more than an eighth of a launch: those fell back
 * to atomics; parts are claimed at run time from HW_REG_XCC_ID, so blockIdx placement does not matter), bucket builds, grid
 * width in waves (0: the strata form has not run)}.  debug_strata (test hook): deals the partitions of `epoch` and
 * returns the bucket offsets sptr[8 W + 1], the bucketed records rec_u / rec_i [nnz] (rec_i: bit 31 = hot row),
 * rank_item [n_items] (popularity rank -> item) and the epoch key; any pointer may be NULL. */
int cornac_hip_bpr_strata_config(cornac_hip_bpr_t h, int hot_permille, int hot_min_mult_x100, int rehash_period);
/* Inside cornac_hip_bpr_fit_epochs the XCD-strata form trains on packed item records (row + its bias line = one random
 * location per item, csrc/bpr_strata.inc); every other entry point gets the dense V / B back first.  enable = 1 lets the
 * chunk API (cornac_hip_bpr_hogwild_enqueue) keep the records from call to call as well — for a caller that touches the
 * (bound) dense table between chunks only through cornac_hip_bpr_table_delta_begin / _step / _finish, which then work on
 * the records, and reads it after cornac_hip_bpr_sync, which writes the records back (the multi-GPU driver of
 * cornac_amd/dist.py with the dense exchange).  Default 0. */
int cornac_hip_bpr_chunk_records(cornac_hip_bpr_t h, int enable);
int cornac_hip_bpr_strata_stats(cornac_hip_bpr_t h, int64_t *out4);
int cornac_hip_bpr_debug_strata(cornac_hip_bpr_t h, uint32_t epoch, int64_t *sptr, int32_t *rec_u, int32_t *rec_i,
                                int32_t *rank_item, uint32_t *key);
/* Tuning and inspection of the LDS-bin form.  ldsbin_config: an item is hot (rows in global memory under atomics, its
 * interactions dealt to all bins) when its degree exceeds hot_x1000 / 1000 of a bin's share nnz / bins (default 75;
 * passing bins: of nnz / (4 x CUs) — a quarter of what one of the 2 x CUs concurrent slots draws per epoch);
 * the form is used when every bin holds at least min_candidates items (default 48: the negative of a draw comes from
 * the positive's bin) and the table fits in max_rounds rounds of one bin per CU (default 4: "resident bins", a bin owns
 * its CU for the epoch).  ldsbin_pass_config: "passing bins" for item tables beyond that — the table passes through the
 * LDS once per epoch in as many rounds as it takes, `waves` waves (4 / 8 / 16; default 8) and at most lds_kb KiB
 * (default 64: two workgroups per CU) per bin, used when an epoch draws at least min_draws_x100 / 100 interactions per
 * item row (default 200) and for launches of at least a quarter of an epoch; enable = 0 switches the regime off
 * (automatic then falls through to XCD strata / the fused kernel).  ldsbin_stats: out8 =
 * {bins (0: the shape does not use the form), LDS rows per bin, hot items, their interactions, bitmap words per user
 * (0: CSR binary search), dynamic LDS bytes per workgroup, row-lock spins that hit their bound since create (0
 * unless there is a bug: fetched with the epoch counters; the resident exchange's row duties count here too), threads
 * per workgroup (1024 resident bins, 64 x waves passing bins)}. */
int cornac_hip_bpr_ldsbin_config(cornac_hip_bpr_t h, int hot_x1000, int min_candidates, int max_rounds);
int cornac_hip_bpr_ldsbin_pass_config(cornac_hip_bpr_t h, int enable, int waves, int lds_kb, int min_draws_x100);
int cornac_hip_bpr_ldsbin_stats(cornac_hip_bpr_t h, int64_t *out8);
/* The per-epoch deal of the LDS-bin form (csrc/bpr_ldsbin.inc, ldsbin_deal_rank).  deal_config: the popularity ranks
 * are permuted (epoch-keyed) inside strata of ~strata_groups * bins consecutive ranks before they are cut into the
 * groups of `bins` items that are dealt one item to each bin (default 16; 1 = the groups are cut from the static rank
 * order, so two items of one group never share a bin); the hot interactions are dealt to the bins in runs that level
 * the bins' work with a hot draw priced at hot_cost_x16 / 16 cold draws (default 32; 0 = an even split).
 * debug_ldsbin_deal (test hook): deals epoch `epoch` of hogwild seed `seed` and returns bin_of_item [n_items],
 * cold_mass [bins] (interactions of the bin's own items), hot_off [bins + 1] (the bin's run of the hot list) and the
 * shuffled hot list hot_u / hot_i [hot interactions]; any pointer may be NULL. */
int cornac_hip_bpr_ldsbin_deal_config(cornac_hip_bpr_t h, int strata_groups, int hot_cost_x16);
int cornac_hip_bpr_debug_ldsbin_deal(cornac_hip_bpr_t h, uint64_t seed, uint32_t epoch, int32_t *bin_of_item,
                                     uint32_t *cold_mass, uint32_t *hot_off, int32_t *hot_u, int32_t *hot_i);
/* HIP-event timing of the hogwild SGD kernel launches, recorded on the handle's
 * stream: returns the summed duration and count of the launches recorded since
 * the previous call, then enables/disables recording for the following ones. */
int cornac_hip_bpr_kernel_timing(cornac_hip_bpr_t h, int enable, double *total_ms, int64_t *launches);
/* time spent (ms) in the last fit_epochs call: [0] sampler, [1] host level
 * scheduling, [2] SGD kernels, [3] total */
int cornac_hip_bpr_last_timing(cornac_hip_bpr_t h, double *ms4);

/* ------------------------------------------------------------------------- *
 * Building blocks of the row-sharded item table (multi-GPU regime 2): the same
 * _fit_sgd step (cornac/models/bpr/recom_bpr.pyx:208-269) cut into sample / fetch / apply / push
 * because the item rows of a triplet live on other GPUs.  All pointers are DEVICE pointers; all
 * launches go to the handle's stream.  Driven by cornac_amd/dist.py:RowShardedBprTrainer.
 * ------------------------------------------------------------------------- */
/* draws n_draws (u, i, j) with the hogwild sampler (continues the handle's sample counter); a draw whose
 * negative is a positive of u ("skipped", recom_bpr.pyx:236-238) is written as u = i = j = -1 */
int cornac_hip_bpr_sample_triplets(cornac_hip_bpr_t h, int64_t n_draws, int neg_population, int32_t *d_u,
                                   int32_t *d_i, int32_t *d_j);
/* BPR update of the handle's U rows and of STAGED item rows: d_slot_i/j index rows of d_rows [n_slots, k]
 * and d_bias [n_slots * bias_stride]; entries with d_u < 0 are ignored */
int cornac_hip_bpr_apply_triplets(cornac_hip_bpr_t h, const int32_t *d_u, const int32_t *d_slot_i,
                                  const int32_t *d_slot_j, int64_t n, float *d_rows, float *d_bias, int bias_stride,
                                  float lr, float reg, int use_bias);
/* The same two steps through the hogwild kernel's user-row ownership (k in 33..256): emit_triplets runs the owned
 * sampler of the fused kernel and writes the triplets of tile t of wave w at ((t * waves + w) * 64 + lane) — u with
 * bit 30 set for a shared (heavy) user, u = i = j = -1 for a skipped or padding slot; *n_slots (a multiple of
 * waves * 64, at most what staged_slots(n_draws) returns, <= slots_cap) is the array length written.  apply_staged
 * takes the same arrays with i / j replaced by staging-table slots: every user row is again touched by one wave only
 * (plain loads/stores, atomics only on the staged item rows), exactly the fused kernel's update.
 * staged_slots returns 0 when the shape has no owned kernel (use sample_triplets / apply_triplets). */
int cornac_hip_bpr_staged_slots(cornac_hip_bpr_t h, int64_t n_draws, int64_t *n_slots);
int cornac_hip_bpr_emit_triplets(cornac_hip_bpr_t h, int64_t n_draws, int neg_population, int32_t *d_u, int32_t *d_i,
                                 int32_t *d_j, int64_t slots_cap, int64_t *n_slots);
int cornac_hip_bpr_apply_staged(cornac_hip_bpr_t h, const int32_t *d_u, const int32_t *d_slot_i, const int32_t *d_slot_j,
                                int64_t n_slots, float *d_rows, float *d_bias, int bias_stride, float lr, float reg,
                                int use_bias);
/* De-duplication of a micro-batch's item ids without a sort.  Owner-major index of item i:
 * g(i) = (i % world) * rows_per_rank + i / world.  shard_mark sets d_mark[g] = 1 for the items of every valid triplet
 * (d_mark: int32 [world * rows_per_rank], zeroed by the caller); the caller scans it (inclusive); shard_slots writes
 * slot = d_scan[g] - 1 per triplet side (-1 for skipped draws); shard_uniq writes the request lists
 * d_uniq_local[d_scan[g] - 1] = g % rows_per_rank, one contiguous run per owner. */
int cornac_hip_bpr_shard_mark(cornac_hip_bpr_t h, const int32_t *d_i, const int32_t *d_j, int64_t n, int world,
                              int64_t rows_per_rank, int32_t *d_mark);
int cornac_hip_bpr_shard_slots(cornac_hip_bpr_t h, const int32_t *d_i, const int32_t *d_j, int64_t n, int world,
                               int64_t rows_per_rank, const int32_t *d_scan, int32_t *d_slot_i, int32_t *d_slot_j);
int cornac_hip_bpr_shard_uniq(cornac_hip_bpr_t h, const int32_t *d_mark, const int32_t *d_scan, int64_t n_rows,
                              int64_t rows_per_rank, int32_t *d_uniq_local);
/* d_table[d_ids[r], :] += (d_now[r, :] - d_before[r, :]) * (d_scale ? d_scale[r] : 1)  (atomic): the owner's side of
 * a push — d_before is the row as the owner sent it, d_now the row as the requester returned it */
int cornac_hip_bpr_scatter_diff_rows(cornac_hip_bpr_t h, float *d_table, const int32_t *d_ids, int64_t n, int width,
                                     const float *d_now, const float *d_before, const float *d_scale);
/* d_out[r, :] = d_table[d_ids[r], :] and d_table[d_ids[r], :] += d_delta[r, :] (atomic), rows of `width` floats */
int cornac_hip_bpr_gather_rows(cornac_hip_bpr_t h, const float *d_table, const int32_t *d_ids, int64_t n, int width,
                               float *d_out);
int cornac_hip_bpr_scatter_add_rows(cornac_hip_bpr_t h, float *d_table, const int32_t *d_ids, int64_t n, int width,
                                    const float *d_delta);
/* Replicated item table (multi-GPU regime 1): the elementwise passes around the all-reduce of the table deltas.
 * flat = [V (n_items*k) | B (n_items)] is the replica the kernels train on, base its value at the last exchange.
 * begin : bucket = [flat - base | V rows touched (n_items) | biases touched (n_items)], local = copy of flat - base
 * finish: (after bucket was sum-all-reduced) R = delta / sqrt(max(touching ranks, 1)) per row; flat += R - local;
 *         base += R */
int cornac_hip_bpr_table_delta_begin(cornac_hip_bpr_t h, const float *d_flat, const float *d_base, int64_t n_items,
                                     int k, float *d_bucket, float *d_local);
int cornac_hip_bpr_table_delta_finish(cornac_hip_bpr_t h, float *d_flat, float *d_base, const float *d_bucket,
                                      const float *d_local, int64_t n_items, int k);
/* finish of the previous exchange followed by begin of the next one, in one pass (adjacent in the overlapped schedule) */
int cornac_hip_bpr_table_delta_step(cornac_hip_bpr_t h, float *d_flat, float *d_base, const float *d_bucket_prev,
                                    const float *d_local_prev, int64_t n_items, int k, float *d_bucket, float *d_local);
/* The three passes with the reconciliation rule as an argument (op: 0 = begin, 1 = finish, 2 = step; rule: 0 = "sqrt",
 * 1 = "align", described below), on the handle's stream and on its packed records where it keeps them: the regime-1
 * driver takes "align" when it exchanges less than a few times per epoch (sparse item sides: DESIGN.md 5). */
int cornac_hip_bpr_table_delta(cornac_hip_bpr_t h, int op, int rule, float *d_flat, float *d_base, const float *d_bucket_prev,
                               const float *d_local_prev, int64_t n_items, int k, float *d_bucket, float *d_local);
/* The same passes for a caller without a BPR handle (the MF driver): op & 15: 0 = begin, 1 = finish, 2 = step, on
 * `hip_stream` of `device`; begin ignores the *_prev pointers, finish the next-exchange pointers.  op >> 4 selects the
 * reconciliation rule: 0 = "sqrt" (the one above: per-row slots of the bucket = touched flags, R = S / sqrt(count)),
 * 1 = "align" (per-row slots = |delta row|^2 of this rank, R = S * min(1, sum of the slots / |S|^2): the sum of
 * orthogonal deltas, the mean of identical ones — MF's default, cornac_amd/dist.py ItemTableReplica). */
int cornac_hip_table_delta(int op, int device, void *hip_stream, float *d_flat, float *d_base, const float *d_bucket_prev,
                           const float *d_local_prev, int64_t n_items, int k, float *d_bucket, float *d_local);
/* Resident exchange: multi-GPU regime 1 for the LDS-bin form WITHOUT chunk launches (the reference has no counterpart:
 * BPR._fit_sgd, recom_bpr.pyx:208-269, is one process; the update it distributes is :252-265).  ONE launch per epoch
 * publishes the item-table deltas of this rank at n_exchanges points and applies the all-reduced sums of the earlier
 * points as soon as their `landed` flag is set — the launch never waits for the collective (csrc/bpr_ldsbin.inc).
 *   d_base     [nt k + nt], nt = total_items: as in the table_delta passes; table - base = steps not yet published
 *   d_buckets  exchange e at + e * bucket_stride floats: [dV (nt k) | dB (nt) | wV (nt) | wB (nt)], all-reduced in place by
 *              the caller; d_keeps exchange e at + e * keep_stride: this rank's own [dV | dB]
 *   d_arrive   [n_exchanges] zero on entry; reaches *n_arrivals (the launch's workgroups) when bucket e is complete
 *   d_landed   [n_exchanges] zero on entry; the caller sets [e] != 0 after the all-reduce of bucket e (in order)
 *   d_applied  [n_items] out: row i has applied the exchanges [0, d_applied[i])
 * The caller's communication stream runs, for e = 0 .. n_exchanges-1: cornac_hip_stream_wait_counter(d_arrive + e,
 * *n_arrivals) -> all-reduce of bucket e -> cornac_hip_stream_set_flag(d_landed + e); when the last one has landed,
 * cornac_hip_bpr_resident_flush (on the handle's stream) applies what the launch did not (d_landed: the landed exchanges only).  The bound tables
 * (cornac_hip_bpr_bind_device) are the replica.  cornac_hip_bpr_resident_exchange_bins: *n_bins = the launch's
 * workgroup count, 0 when this shape / these flags do not take the LDS-bin form (use hogwild_enqueue chunks then). */
int cornac_hip_bpr_resident_exchange_bins(cornac_hip_bpr_t h, int neg_population, int hogwild_flags, int *n_bins);
int cornac_hip_bpr_epoch_resident_enqueue(cornac_hip_bpr_t h, float lr, float reg, int use_bias, int neg_population,
                                          int hogwild_flags, int n_exchanges, int rule, float *d_base, float *d_buckets,
                                          int64_t bucket_stride, float *d_keeps, int64_t keep_stride, uint32_t *d_arrive,
                                          const uint32_t *d_landed, uint32_t *d_applied, int *n_arrivals);
int cornac_hip_bpr_resident_flush(cornac_hip_bpr_t h, int n_exchanges, int rule, float *d_base, const float *d_buckets,
                                  int64_t bucket_stride, const float *d_keeps, int64_t keep_stride,
                                  const uint32_t *d_applied, const uint32_t *d_landed);
/* Conveyor layout of the LDS-bin form — multi-GPU regime 2 ("V rows owned by ..." — here the item table is
 * sharded by row into n_blocks blocks that rotate over the ranks, cornac_amd/dist.py BinConveyorBprTrainer).  No reference
 * counterpart (cornac is one process); a launch computes recom_bpr.pyx:231-267 for the draws of its bins.
 *   conveyor_setup   plans the bins as n_blocks equal ranges (block B = the bins [B bpb, (B + 1) bpb) of the epoch's deal: a
 *                    block's ITEMS change with the deal key like any bin's, so every (positive, negative) pair of items can
 *                    meet), no hot items; rank_item [n_items]: the popularity order every rank of the fit agrees on (NULL: the
 *                    handle's own interactions'); deal_seed: shared by the ranks (the draws keep the handle's hogwild seed);
 *                    release_item_tables != 0 frees the handle's own V / B (the rows live in the caller's block buffers).
 *                    Out: *n_bins, *bins_per_block, *cap (slots per bin).  A block buffer is [bpb cap k rows | bpb cap biases]
 *                    floats, row (bin - first bin of the block) cap + slot.
 *   conveyor_layout  on the handle's stream: d_slot_item [n_bins cap] = the item at (bin, slot) under the deal of layout_epoch
 *                    (-1: none), d_item_slot [n_items] = its inverse (either may be NULL)
 *   conveyor_enqueue one launch on the handle's stream: all of this handle's draws of `epoch` whose positive lies in the blocks
 *                    first_block[0 .. n_ranges) (distinct; n_ranges <= 8), rows read from / written to d_rows[r]; the deal is
 *                    that of layout_epoch (the caller re-deals the buffers when it changes).  Counters: cornac_hip_bpr_sync. */
int cornac_hip_bpr_conveyor_setup(cornac_hip_bpr_t h, int n_blocks, const int32_t *rank_item, uint64_t deal_seed,
                                  int release_item_tables, int *n_bins, int *bins_per_block, int *cap);
int cornac_hip_bpr_conveyor_layout(cornac_hip_bpr_t h, uint32_t layout_epoch, int32_t *d_slot_item, int32_t *d_item_slot);
int cornac_hip_bpr_conveyor_enqueue(cornac_hip_bpr_t h, uint32_t epoch, uint32_t layout_epoch, int n_ranges,
                                    const int32_t *first_block, float *const *d_rows, float lr, float reg, int use_bias,
                                    int neg_population, int hogwild_flags);
/* on `hip_stream` of `device`: wait until *d_counter >= target (after timeout_ms: *d_error = 1 and the stream moves on;
 * every later wait on the same d_error returns at once); set *d_flag = value unless d_unless != NULL and *d_unless != 0.
 * The resident exchange passes its d_error as d_unless: a bucket that was all-reduced incomplete never gets its landed
 * flag, so neither the launch nor cornac_hip_bpr_resident_flush (d_landed) applies it; the caller reports d_error. */
int cornac_hip_stream_wait_counter(int device, void *hip_stream, const uint32_t *d_counter, uint32_t target,
                                   uint32_t *d_error, int timeout_ms);
int cornac_hip_stream_set_flag(int device, void *hip_stream, uint32_t *d_flag, uint32_t value, const uint32_t *d_unless);
/* Measurement aid (no reference counterpart): what a ring all-reduce costs THIS rank's memory system and CUs when the
 * process group has a single member and RCCL therefore launches nothing — n_workgroups workgroups of 512 threads stream
 * n_floats floats from d_src (read cyclically over src_floats) to d_dst (written cyclically over dst_floats) on
 * hip_stream.  cornac_amd/dist.py puts it where the collective sits when asked to emulate a world of N ranks
 * (n_floats = 2 (N - 1) / N x the bucket). */
int cornac_hip_stream_ring_standin(int device, void *hip_stream, const float *d_src, int64_t src_floats, float *d_dst,
                                   int64_t dst_floats, int64_t n_floats, int n_workgroups);

/* ------------------------------------------------------------------------- *
 * VEBPR (view-enhanced BPR) on the same handle.
 * Replaces: VEBPR._fit_sgd_viewloss(rng_pos, rng_view, rng_neg, ..., U, V)
 *           cornac/models/bpr/recom_vebpr.pyx:211-337 and its caller loop :189-207.
 * The handle's CSR is the purchase matrix; set_views adds the view matrix
 * (train_set.view_matrix: views minus purchases, sorted rows,
 * cornac/data/dataset.py:1400-1440).  No item biases.
 * ------------------------------------------------------------------------- */
int cornac_hip_bpr_set_views(cornac_hip_bpr_t h, const int32_t *view_indptr, const int32_t *view_indices,
                             int64_t nnz_view);
/* third mt19937 engine (rng_view, recom_vebpr.pyx:192); call after cornac_hip_bpr_seed_mt19937 */
int cornac_hip_bpr_seed_view_stream(cornac_hip_bpr_t h, uint32_t mt_seed_view);
int cornac_hip_vebpr_fit_epochs(cornac_hip_bpr_t h, int n_epochs, float lr, float reg, float alpha, int mode,
                                int64_t *correct, int64_t *skipped);
/* The float64 instantiation of the same fused-type function (recom_vebpr.pyx:219 `floating[:, :] U, floating[:, :] V`,
 * reached by float64 init_params): tables set with cornac_hip_bpr_set_factors_f64 (its bias table is not used), all
 * locals of the step in double, sequential semantics (the three mt19937 streams) only, like cornac_hip_bpr_fit_epochs_f64. */
int cornac_hip_vebpr_fit_epochs_f64(cornac_hip_bpr_t h, int n_epochs, double lr, double reg, double alpha, int64_t *correct,
                                    int64_t *skipped);
/* Hogwild mode (the reference's prange over samples, recom_vebpr.pyx:211-337) gives every wave of the grid a fixed set of
 * users (k > 32 and enough interactions for one 64-sample tile per wave): positives come from the wave's own users, so
 * an exclusive user's row has one writer and is updated by plain load / store; the three item rows keep fp32 atomics.
 * OR this bit into `mode` for the all-atomic form (diagnostics, A/B); vebpr_hogwild_form reports what the last hogwild
 * epoch ran (1 = user-row ownership). */
#define CORNAC_HIP_VEBPR_NO_OWNERSHIP 0x100
int cornac_hip_vebpr_hogwild_form(cornac_hip_bpr_t h, int *owned);

/* ------------------------------------------------------------------------- *
 * Matrix factorisation trainer.
 * Replaces: backend_cpu.fit_sgd(rid, cid, val, U, V, Bu, Bi, lr, reg, mu,
 *           max_iter, num_threads, use_bias, early_stop, verbose)
 *           cornac/models/mf/backend_cpu.pyx:35-97 (called from
 *           MF._fit_cpu, cornac/models/mf/recom_mf.py:189-209).
 * ------------------------------------------------------------------------- */
typedef struct cornac_hip_mf *cornac_hip_mf_t;

/* rid/cid are the int64 COO arrays of train_set.uir_tuple in stored order, val fp32. */
int cornac_hip_mf_create(cornac_hip_mf_t *out, int device, int64_t n_users, int64_t n_items, int k,
                         const int64_t *rid, const int64_t *cid, const float *val, int64_t nnz);
int cornac_hip_mf_destroy(cornac_hip_mf_t h);
int cornac_hip_mf_set_factors(cornac_hip_mf_t h, const float *U, const float *V, const float *Bu, const float *Bi);
int cornac_hip_mf_get_factors(cornac_hip_mf_t h, float *U, float *V, float *Bu, float *Bi);
/* Runs up to max_iter epochs; loss_per_epoch[e] = 0.5 * sum(err^2) (may be NULL);
 * early_stop: stop when |loss - last_loss| < 1e-5 (backend_cpu.pyx:89-93).
 * epochs_run receives the number of epochs executed. */
int cornac_hip_mf_fit(cornac_hip_mf_t h, int max_iter, float lr, float reg, float mu, int use_bias, int early_stop,
                      int mode, float *loss_per_epoch, int *epochs_run);
/* Form of the hogwild MF epoch: 0 = automatic, 1 = the fused kernel (user rows owned by waves, item rows by fp32
 * atomics), 2 = the block rotation (csrc/mf_blocks.inc: item bins in the CUs' LDS, user blocks rotating among the 32
 * workgroups of an XCD, 8 launches x 32 barrier-separated sub-rounds per epoch; no atomics, every update applied exactly
 * once).  Automatic = the block rotation for k in 33..256, >= 256 items and >= 2^22 ratings on a 256-CU / 8-XCD device.
 * 3 = the STEP form, for handles that hold one step of a larger schedule — few item rows, most ratings of the launch in
 * flight at once — such as the per-block handles of dist.MfBlockRotationTrainer: the fused kernel launched with only ~4
 * ratings in flight per item row, and item rows that still take more than 32 concurrent updates trained through copies
 * merged after the launch, AT ANY SIZE (forms 0..2 split rows only from 2^20 ratings on: an item holding > 0.1 % of them).
 * Choose it before the handle's first epoch.
 * hogwild_stats: out4 = {form of the last epoch (1 / 2, 0: none yet), tiles of the block schedule, LDS rows per bin,
 * 1 if the rotation gave up once (workgroup placement / barrier bound) and the handle went back to the fused kernel}. */
int cornac_hip_mf_hogwild_form(cornac_hip_mf_t h, int form);
int cornac_hip_mf_hogwild_stats(cornac_hip_mf_t h, int64_t *out4);
/* One-shot form with the reference's exact argument list (host buffers, in place). */
int cornac_hip_mf_fit_sgd(int device, const int64_t *rid, const int64_t *cid, const float *val, int64_t nnz, float *U,
                          float *V, float *Bu, float *Bi, int64_t n_users, int64_t n_items, int k, float lr, float reg,
                          float mu, int max_iter, int use_bias, int early_stop, int mode, float *loss_per_epoch,
                          int *epochs_run);
/* Multi-GPU driver surface (no counterpart in the reference; cornac_amd/dist.py ShardedMfTrainer): every rank holds the
 * ratings of its own users (U, Bu local), the item side [V | Bi] is replicated in caller-owned device memory and
 * reconciled by the caller between slices.  bind_items: train into caller-owned V (n_items x k) and Bi (n_items);
 * set_stream: run on a caller stream (NULL = the handle's own); epoch_enqueue: ratings [nnz*part/n_parts,
 * nnz*(part+1)/n_parts) of the stored order, hogwild semantics, NO host synchronisation (n_parts = 1: the whole epoch in
 * the form cornac_hip_mf_fit would pick); sync: wait, and return the sum of squared errors enqueued since the last sync. */
int cornac_hip_mf_bind_items(cornac_hip_mf_t h, float *dV, float *dBi);
/* bind_users: train into caller-owned U (n_users x k) and Bu (n_users) — cornac_amd/dist.py MfBlockRotationTrainer: the handles
 * of a rank's item blocks (each created over the rank's ratings of one block, item ids local to it) share the rank's user side
 * and are bound, step by step, to whichever buffer holds their block. */
int cornac_hip_mf_bind_users(cornac_hip_mf_t h, float *dU, float *dBu);
int cornac_hip_mf_set_stream(cornac_hip_mf_t h, void *hip_stream);
int cornac_hip_mf_epoch_enqueue(cornac_hip_mf_t h, int part, int n_parts, float lr, float reg, float mu, int use_bias);
int cornac_hip_mf_sync(cornac_hip_mf_t h, double *sq_err_sum);
int cornac_hip_mf_kernel_timing(cornac_hip_mf_t h, int enable, double *total_ms, int64_t *launches);
int cornac_hip_mf_last_timing(cornac_hip_mf_t h, double *ms4);

/* Minibatch path with dense optimisers on the same handle.
 * Replaces: backend_pt.learn(model, train_set, n_epochs, batch_size, learning_rate, reg, optimizer)
 *           cornac/models/mf/backend_pt.py:67-106 and the forward of backend_pt.MF (:56-65), selected by
 *           MF(backend="pytorch", optimizer=...) (cornac/models/mf/recom_mf.py:211-252).
 * order: indices into the handle's rating arrays in visiting order (the concatenated batches of
 * Dataset.uir_iter(batch_size, shuffle=True), cornac/data/dataset.py:445-488); consecutive slices of batch_size
 * form the optimiser steps (the last one may be shorter).  Optimiser state persists across calls. */
#define CORNAC_HIP_OPT_SGD 0
#define CORNAC_HIP_OPT_ADAM 1
#define CORNAC_HIP_OPT_RMSPROP 2
#define CORNAC_HIP_OPT_ADAGRAD 3
int cornac_hip_mf_fit_minibatch(cornac_hip_mf_t h, const int64_t *order, int64_t n_total, int batch_size,
                                int optimizer, float lr, float reg, float mu, int use_bias, double *loss_sum);
/* The same with the reference's dropout on the gathered rows (backend_pt.py:42,59: nn.Dropout(p) on the user rows and on
 * the item rows of a batch): keep_u / keep_i are [n_total][k] bytes, row b belongs to order[b], non-zero = the factor is
 * kept and scaled by keep_scale = 1 / (1 - p).  The masks are an input: the host draws them as the reference's run
 * draws them from torch's CPU generator (cornac_amd/mf.py). */
int cornac_hip_mf_fit_minibatch_dropout(cornac_hip_mf_t h, const int64_t *order, int64_t n_total, int batch_size,
                                        int optimizer, float lr, float reg, float mu, int use_bias, const uint8_t *keep_u,
                                        const uint8_t *keep_i, float keep_scale, double *loss_sum);
int cornac_hip_mf_reset_optimizer(cornac_hip_mf_t h);

/* ------------------------------------------------------------------------- *
 * VBPR (visual BPR) minibatch trainer.
 * Replaces: the per-batch body of VBPR._fit_torch (forward, autograd backward,
 *           torch.optim.Adam step over all tables) cornac/models/vbpr/recom_vbpr.py:228-262.
 * The (u, i, j) batches are produced by the host sampler Dataset.uij_iter
 * (cornac/data/dataset.py:490-526) exactly as in the reference and handed over
 * per call; Adam moments and the step counter persist in the handle.
 * ------------------------------------------------------------------------- */
typedef struct cornac_hip_vbpr *cornac_hip_vbpr_t;

/* features: item visual features [n_items, n_feat] fp32 (train_set.item_image.features) */
int cornac_hip_vbpr_create(cornac_hip_vbpr_t *out, int device, int64_t n_users, int64_t n_items, int k, int k2,
                           int n_feat, const float *features);
int cornac_hip_vbpr_destroy(cornac_hip_vbpr_t h);
/* Bi [n_items], Gu [n_users,k], Gi [n_items,k], Tu [n_users,k2], E [n_feat,k2], Bp [n_feat]; NULL skips */
int cornac_hip_vbpr_set_params(cornac_hip_vbpr_t h, const float *Bi, const float *Gu, const float *Gi, const float *Tu,
                               const float *E, const float *Bp);
int cornac_hip_vbpr_get_params(cornac_hip_vbpr_t h, float *Bi, float *Gu, float *Gi, float *Tu, float *E, float *Bp);
/* runs ceil(n_total / batch_size) Adam steps over consecutive batches of the triplet arrays;
 * sum_nll (may be NULL) receives sum over all triplets of -logsigmoid(x_uij) */
int cornac_hip_vbpr_fit_batches(cornac_hip_vbpr_t h, const int32_t *u, const int32_t *i, const int32_t *j,
                                int64_t n_total, int batch_size, float lr, float lambda_w, float lambda_b,
                                float lambda_e, double *sum_nll);
/* theta_item = F E [n_items, k2], visual_bias = F Bp [n_items] (recom_vbpr.py:132-133) */
int cornac_hip_vbpr_item_tables(cornac_hip_vbpr_t h, float *theta_item, float *visual_bias);

/* ------------------------------------------------------------------------- *
 * WMF (weighted matrix factorisation) minibatch trainer.
 * Replaces: the TensorFlow graph of cornac/models/wmf/wmf.py:34-55 (P = U V_b^T, weighted
 *           squared error, gradient clipping to [-5, 5], tf.train.AdamOptimizer) and the
 *           per-batch sess.run of cornac/models/wmf/recom_wmf.py:181-199.
 * Batches of item ids come from the host iterator Dataset.item_iter (cornac/data/dataset.py:546-562)
 * exactly as in the reference; Adam moments and the step counter persist in the handle and are
 * reset by set_factors (the reference builds a fresh graph per fit).
 * ------------------------------------------------------------------------- */
typedef struct cornac_hip_wmf *cornac_hip_wmf_t;

/* CSC of the rating matrix (train_set.csc_matrix): indptr int64[n_items+1], rows = user ids, vals fp32 */
int cornac_hip_wmf_create(cornac_hip_wmf_t *out, int device, int64_t n_users, int64_t n_items, int k,
                          const int64_t *csc_indptr, const int32_t *csc_rows, const float *csc_vals, int64_t nnz);
void cornac_hip_wmf_destroy(cornac_hip_wmf_t h);
/* U [n_users,k], V [n_items,k] fp32 */
int cornac_hip_wmf_set_factors(cornac_hip_wmf_t h, const float *U, const float *V);
int cornac_hip_wmf_get_factors(cornac_hip_wmf_t h, float *U, float *V);
/* one Adam step per batch b = item_ids[batch_ptr[b] .. batch_ptr[b+1]) (1..128 distinct items), in order;
 * loss_out[b] (may be NULL) = that step's loss value (`_loss`, recom_wmf.py:195) */
int cornac_hip_wmf_fit_batches(cornac_hip_wmf_t h, const int32_t *item_ids, const int64_t *batch_ptr,
                               int64_t n_batches, float lambda_u, float lambda_v, float a, float b,
                               float learning_rate, double *loss_out);
/* HIP-event timing of fit_batches (device ms of the last call) */
int cornac_hip_wmf_kernel_timing(cornac_hip_wmf_t h, int enabled);
int cornac_hip_wmf_last_timing(cornac_hip_wmf_t h, double *device_ms);

/* ------------------------------------------------------------------------- *
 * Scoring / ranking.
 * Replaces: fast_dot(vec, mat, output)  cornac/utils/fast_dot.pyx:40-43 as used by
 *           BPR.score (recom_bpr.pyx:288-291) and MF.score (recom_mf.py:273-278),
 *           and the per-user argsort/argpartition of Recommender.rank
 *           (cornac/models/recommender.py:503-530), batched over users.
 * score(u, i) = item_base[i] + user_base[u] + sum_f U[u,f]*V[i,f], accumulated
 * as an index-ordered fp32 fma chain (what the gfx950 fp32 MFMA computes).
 * ------------------------------------------------------------------------- */
typedef struct cornac_hip_scorer *cornac_hip_scorer_t;

/* Exclusion lists per USER id, kept on the device: CSR (indptr int64[n_users + 1], indices int32) of the items
 * ranking must skip for each user — in the evaluation loop the user's training (+ validation) positives
 * (cornac/eval_methods/base_method.py:176-205 builds the same list per user with numpy set operations).  NULL indptr
 * drops them.  cornac_hip_rank_topk_resident ranks `n` users (users[0..n), or u0 .. u0+n-1 when users is NULL) with
 * those lists applied, like cornac_hip_rank_topk with per-call lists but without moving the lists again; items_out /
 * scores_out may be NULL (results stay on the device), device_ms (optional) receives the HIP-event time of the
 * bitmap build + fused top-k + merge kernels. */
int cornac_hip_scorer_set_exclusions(cornac_hip_scorer_t h, const int64_t *indptr, const int32_t *indices);
/* Page-locked host memory of at least `bytes` bytes owned by the scorer (re-used across calls, freed with it): a caller
 * that passes it as items_out / scores_out gets its results at the full PCIe rate instead of through the pageable-copy
 * staging path (5.5 MB of top-10 ids for 138 493 users: 0.2 instead of 0.5 ms).  A later, larger request returns a new
 * buffer; outgrown ones stay valid (and allocated) until the scorer is destroyed. */
int cornac_hip_scorer_host_buffer(cornac_hip_scorer_t h, size_t bytes, void **out);
int cornac_hip_rank_topk_resident(cornac_hip_scorer_t h, const int32_t *users, int64_t u0, int64_t n, int topk,
                                  int32_t *items_out, float *scores_out, double *device_ms);

int cornac_hip_scorer_create(cornac_hip_scorer_t *out, int device, int64_t n_users, int64_t n_items, int k);
int cornac_hip_scorer_destroy(cornac_hip_scorer_t h);
/* item_base: BPR -> i_biases; MF -> global_mean + i_biases.  user_base: MF -> u_biases; NULL = 0. */
int cornac_hip_scorer_set(cornac_hip_scorer_t h, const float *U, const float *V, const float *item_base,
                          const float *user_base);
/* out[n_items] for one user  (= model.score(user_idx)) */
int cornac_hip_score_user(cornac_hip_scorer_t h, int64_t user, float *out);
/* fast_dot's float64 variant (cornac/utils/fast_dot.pyx:25-43, ddot): the scorer can additionally hold the float64 tables
 * of a model trained in double; score_user_f64 = item_base + user_base + <U[user], V[i]> summed in index order in double.
 * The batched top-k kernels stay float32 (MFMA): a float64 model ranks through score_user_f64 + the reference's own
 * host-side ordering (cornac/models/recommender.py:503-530 is NumPy). */
int cornac_hip_scorer_set_f64(cornac_hip_scorer_t h, const double *U, const double *V, const double *item_base,
                              const double *user_base);
int cornac_hip_score_user_f64(cornac_hip_scorer_t h, int64_t user, double *out);
/* out[n * n_items] for a block of users */
int cornac_hip_score_block(cornac_hip_scorer_t h, const int32_t *users, int64_t n, float *out);
/* out[p] = score(users[p], items[p]) for n pairs, optionally clipped to [lo, hi]:
 * the batched form of Recommender.rate() (cornac/models/recommender.py:447-474) as
 * called once per test rating by rating_eval (cornac/eval_methods/base_method.py:35-105). */
int cornac_hip_score_pairs(cornac_hip_scorer_t h, const int32_t *users, const int32_t *items, int64_t n, int clip,
                           float lo, float hi, float *out);
/* Batched rank(): for each of the n users, the topk best items in descending
 * score order (ties: higher item index first — the oracle's pinned tie rule),
 * excluding the items listed in the optional CSR (excl_indptr[n+1],
 * excl_indices) row of that user.  topk == n_items gives the full ranking.
 * items_out / scores_out are [n, topk]; slots beyond the number of candidates
 * are filled with -1 / -inf. */
int cornac_hip_rank_topk(cornac_hip_scorer_t h, const int32_t *users, int64_t n, int topk, const int64_t *excl_indptr,
                         const int32_t *excl_indices, int32_t *items_out, float *scores_out);
/* Where listed items stand in each user's ranking, without producing the ranking: for user users[b] the
 * "targets" tgt_indices[tgt_indptr[b] .. tgt_indptr[b+1]) (the evaluation loop's test positives,
 * cornac/eval_methods/base_method.py:185-206) among that user's candidates (all items minus the optional
 * exclusion CSR row, as in cornac_hip_rank_topk).  Per target, in tgt_indices order:
 *   greater_out = candidates with a strictly higher score,
 *   pos_out     = its 0-based position in the order cornac_hip_rank_topk produces (ties: higher index first),
 *   ge_out      = candidates with score >= its own (itself included),
 *   tgt_scores_out = its score.
 * These are what AUC / MAP / MRR (cornac/metrics/ranking.py:185-527) read off `rank(k=-1)` output.  A target that is
 * out of range or excluded gets -1 / -1 / -1 / -inf. */
int cornac_hip_rank_positions(cornac_hip_scorer_t h, const int32_t *users, int64_t n, const int64_t *excl_indptr,
                              const int32_t *excl_indices, const int64_t *tgt_indptr, const int32_t *tgt_indices,
                              int32_t *greater_out, int32_t *pos_out, int32_t *ge_out, float *tgt_scores_out);
/* Device-resident throughput probe used by bench.py: ranks users
 * [u0, u0 + n) with fused top-k, keeps results on device; returns elapsed ms of
 * the kernels (hipEvent) in *ms. */
int cornac_hip_rank_topk_device(cornac_hip_scorer_t h, int64_t u0, int64_t n, int topk, int repeats, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* CORNAC_HIP_H_ */
