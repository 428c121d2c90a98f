/* rescan_hip.h — C ABI of librescan_hip.so, the MI355X (gfx950) implementation of Rescan's
 * per-scan geometric hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference tree, mhalber/Rescan).  Matrices are column-major float[16] exactly like
 * msh_mat4_t (lib/msh/msh_vec_math.h:187-191); point arrays are AoS float[3*n] exactly like
 * msh_vec3_t* (lib/msh/msh_vec_math.h:159-164).
 *
 * Return convention: functions returning int give 0 on success and a negative RS_HIP_E_*
 * code otherwise; rs_hip_last_error() describes the failure.  There is NO CPU fallback:
 * without a usable HIP device every compute entry point fails with RS_HIP_E_NODEVICE.
 */
#ifndef RESCAN_HIP_H
#define RESCAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_HIP_OK            0
#define RS_HIP_E_NODEVICE   -1   /* no HIP device / runtime error at init */
#define RS_HIP_E_ARG        -2   /* bad argument */
#define RS_HIP_E_RUNTIME    -3   /* HIP runtime error during the call */
#define RS_HIP_E_CAPACITY   -4   /* problem exceeds a documented limit */

/* ---- runtime ----------------------------------------------------------------------- */

/* Bind the calling process to HIP device `device` (one process per GPU). */
int         rs_hip_init( int device );
const char* rs_hip_last_error( void );
/* Use an existing hipStream_t (e.g. torch's current stream) for all launches; NULL = the
 * library's own stream. */
int         rs_hip_set_stream( void* hip_stream );
int         rs_hip_synchronize( void );
/* Restrict the calling thread's own stream to the compute units whose bits are set in mask (n_words x 32 bits, CU 0 = bit 0 of
 * word 0): independent operators issued from different threads can be kept off each other's CUs (a latency-bound chain beside
 * a throughput-bound batch).  Replaces the thread's stream; pending work on the old one is waited for.  (NULL, 0) goes back to an
 * unrestricted stream — do that before the process exits when a profiler is attached: rocprofv3 crashes in its finalisation when
 * masked streams are still alive. */
int         rs_hip_stream_cu_mask( const uint32_t* mask, int32_t n_words );
/* The HIP stream (hipStream_t) the calling thread's launches go to: its own, or the one given to rs_hip_set_stream.  NULL if the
 * library cannot be initialised. */
void*       rs_hip_get_stream( void );
/* ABI/version string, e.g. "rescan_hip 0.1 gfx950". */
const char* rs_hip_version( void );

/* Per-kernel timing with HIP events recorded on the launch stream.  While enabled, each
 * launch of a hot kernel is bracketed by an event pair; rs_hip_profile_read() synchronises
 * and returns launch count and summed milliseconds for kernel `name`
 * ("nn_icp", "icp_moments", "nn_score", "nn_label", "nn_rows", "edges", "coverage").  The name "candidates" is
 * not a kernel: its launch count is the number of candidates the search kernels staged and evaluated
 * since the last reset (each of them by the 64 query lanes of its wave) — SURVEY.md §8d's second figure. */
int         rs_hip_profile_enable( int on );
int         rs_hip_profile_reset( void );
int         rs_hip_profile_read( const char* name, int64_t* launches, double* total_ms );
/* A named no-op kernel (rs::k_step_marker) on the calling thread's stream: marks where a caller's unit of work begins in a kernel trace. */
int         rs_hip_profile_marker( void );

/* ---- device-resident clouds -------------------------------------------------------- */

/* A cloud = positions (+ optional normals) laid out in HBM as 16-byte records
 * {x,y,z,orig_index} sorted by uniform-grid cell, plus the cell offset table: the device
 * counterpart of a cloud level and its msh_hash_grid_t
 * (lib/rs/rs_pointcloud.h:77-97,849-863; lib/msh/msh_hash_grid.h:248-269,388-541).
 * cell_size > 0 : grid cell edge in metres (msh_hash_grid uses 2*radius; any value gives the
 *                 same search results, it only changes speed).
 * cell_size < 0 : chosen from the cloud's sampling density (about two sample spacings) —
 *                 the fastest choice for the staged search.
 * cell_size = 0 : one cell holding everything ("brute-tile" layout). */
typedef struct rs_hip_cloud rs_hip_cloud_t;

rs_hip_cloud_t* rs_hip_cloud_create( const float* pos, const float* nor /* may be NULL */,
                                     int32_t n, float cell_size );
void            rs_hip_cloud_destroy( rs_hip_cloud_t* c );
int32_t         rs_hip_cloud_size( const rs_hip_cloud_t* c );
/* bytes of HBM held by the cloud */
int64_t         rs_hip_cloud_bytes( const rs_hip_cloud_t* c );
/* (diagnostics) Where cloud construction went, in seconds of the calling threads' wall clock, summed over every cloud the process
 * built since the last reset: out[0] host copy of the arrays, [1] upload + bounds, [2] cell index (sort by cell, offset table),
 * [3] Hilbert order + tiles.  Returns the number of clouds counted; reset != 0 clears the counters. */
int64_t         rs_hip_cloud_build_seconds( double out[4], int32_t reset );

/* ---- bounded-K radius search ------------------------------------------------------- */

/* msh_hash_grid_radius_search (lib/msh/msh_hash_grid.h:1090-1259): for every query the
 * (at most) k nearest target points with dist² < radius², row-major rows of stride k.
 * Rows are always written in ascending (dist², index) order (the reference sorts when
 * search_desc.sort != 0 and leaves heap order otherwise; ascending is a valid answer to
 * both).  n_neighbors may be NULL.  Host pointers in, host pointers out.
 * Returns the total neighbour count through *total (may be NULL). */
int rs_hip_radius_search( const rs_hip_cloud_t* target, const float* query, int64_t n_query,
                          float radius, int32_t k,
                          float* distances_sq, int32_t* indices, size_t* n_neighbors,
                          uint64_t* total );

/* ---- point-to-plane ICP ------------------------------------------------------------- */

/* icp_align (lib/rs/icp.h:416-500): aligns T1*source to T2*target; *T1 is updated in place,
 * the last RMS error is returned through *err (1e6 if no iteration produced one).
 * `source` and `target` must have normals.  max_iter: the reference uses 100.
 * fixed_iters != 0 disables the convergence test at icp.h:489 (benchmark mode).
 * n_iters (may be NULL) receives the number of correspondence searches made. */
int rs_hip_icp_align( const rs_hip_cloud_t* source, const rs_hip_cloud_t* target,
                      float* T1, const float* T2, float max_dist, float max_angle,
                      int32_t max_iter, int32_t fixed_iters, float* err, int32_t* n_iters );

/* The same call, also leaving the error after every iteration in errs_per_iter[0 .. *n_iters) (room for max_iter floats): what
 * icp_align( ..., verbose = true ) prints per iteration (icp.h:482-486).  Same result; the loop state is read after every iteration
 * instead of every few, so it is slower by a synchronisation per iteration. */
int rs_hip_icp_align_traced( const rs_hip_cloud_t* source, const rs_hip_cloud_t* target,
                             float* T1, const float* T2, float max_dist, float max_angle,
                             int32_t max_iter, int32_t fixed_iters, float* err, int32_t* n_iters, float* errs_per_iter );

/* The estimator step (icp.h:136-148,210-298,393-402), by source size (round 6; priced against the bar on every reference fixture:
 * profiles/r06/estimator_policy.txt).  Sources of at most `n_points` points use the reference's own accumulation order and
 * precisions (one sequential fp32 chain per accumulator): poses, errors and iteration counts are bit-identical to the
 * reference's.  Default 16384 — every level-2 object of the reference's call sites — (environment RS_HIP_REF_ORDER_BELOW); 65536 covers all of them, at
 * 520 instead of 110 us per iteration on a 50 k-point source.  Larger sources: see rs_hip_icp_replay_below (opt-in),
 * rs_hip_icp_lane_chains_below (default up to 65536 points), beyond that the grid chains — all three centre the step on the
 * reference's own fp32 centroid sums, bit for bit, and differ from it only where it rounds the 33 accumulators of the normal
 * equations after every addition (<= 3e-6 in the pose on its fixtures, equal iteration counts).  n_points < 0 only reads.  Returns
 * the previous threshold.  Applies to rs_hip_icp_align, _batch, _multi and rs_hip_icp_estimate_pt2pl. */
int32_t rs_hip_icp_reference_order_below( int32_t n_points );
/* Sources larger than that, up to `n_points` points (default 0 = off since round 6; environment RS_HIP_REPLAY_BELOW), get the SAME sums —
 * the reference's sequential fp32 / fp64 chains, bit for bit — computed in parallel: segments of 128 points are added
 * speculatively from a guessed start, for both parities of its last mantissa bit, and a walk over the segments with the exact
 * value accepts a segment when the accumulator provably stayed inside its binade.  0 = never; n_points < 0 only
 * reads.  Returns the previous threshold.  rs_hip_icp_replay_redone(): segments the last call's walks had to re-add one addend
 * after the other (a diagnostic: binade crossings, chain starts). */
int32_t rs_hip_icp_replay_below( int32_t n_points );
int32_t rs_hip_icp_replay_redone( void );
/* Round 6.  Sources above both thresholds and of at most `n_points` points take the LANE chains: the reference's 2.5 sigma cut, its
 * seven weighted-centroid sums (lib/rs/icp.h:136-148) as the sequential fp32 chains they are, bit for bit — one wave per chain, 256
 * addends per step on the integer grid of the running sum's binade, in fp32 one after the other wherever that does not hold — and the
 * normal equations (:226-252) as a parallel fp64 reduction centred on those centroids.  Not the reference's bits: measured on every
 * reference fixture of that size at most 3e-6 from its pose, equal iteration counts (profiles/r06/estimator_policy.txt).  Any
 * number of differently sized problems run side by side (rs_hip_icp_align_multi).  Larger sources take the grid chains (the same
 * sums spread over the chip) — and so does a call with ONE problem from 28672 points on: same sums, same poses bit for bit, faster with
 * the chip to itself (tools/lane_vs_grid.py); a threshold set beyond 65536 keeps such calls on the lane chains too.  n_points < 0 only
 * reads; returns the previous threshold.  Environment: RS_HIP_LANE_CHAINS_BELOW.
 * rs_hip_icp_lane_chains_sequential(): addends those walks added one by one since rs_hip_init (a diagnostic: chain starts, binade
 * changes, ties, sums that hover around zero). */
int32_t rs_hip_icp_lane_chains_below( int32_t n_points );
int64_t rs_hip_icp_lane_chains_sequential( void );
/* The stop test's guard (round 6).  The estimators above that are not the reference's own order follow its per-iteration errors to
 * 1e-8 ... 6e-7; icp_align's stop test |err - prev_err| < 1e-5 (icp.h:489) decided by less than that can fall the other way (one
 * iteration more or less: 1e-4 in the pose).  A problem whose decisive difference comes within 1.5e-6 of the threshold (environment
 * RS_HIP_STOP_GUARD, 0 = off) is run AGAIN in the reference's own order — its bits — where one exists (sources up to 262144 points).
 * Returns how many problems were, since rs_hip_init. */
int64_t rs_hip_icp_stop_guard_redone( void );
/* Plain early iterations (round 6).  The chains above fix where the iteration CONVERGES; far from the end an iteration only has to bring the
 * pose near.  With this on (default; environment RS_HIP_EARLY_PLAIN=0 turns it off) a call on a scan-sized source (more than 65536 points) runs the
 * plain step — fp64 moments centred on their own fp64 centroids, two launches instead of six — in its early iterations, keeping chain
 * iterations before any iteration whose result can be returned: a fixed-length call (fixed_iters) runs plain until TWO before its end; a call
 * with the stop test in its first four iterations (three chain iterations before the first decision), and only for sources above 262144
 * points (below, the stop test's guard wants the reference's errors from the start).  Measured on six 1 M-point rooms: <= 3.7e-6 from the
 * reference's pose (<= 1.2e-6 without; with ONE chain iteration 1.1e-5: not shipped; profiles/r06/early_plain.txt).
 * Object-sized sources (up to 65536 points, whichever kernels run their chains) keep them in every iteration.  on < 0 only reads; returns the previous setting. */
int32_t rs_hip_icp_early_plain( int32_t on );
float   rs_hip_icp_stop_guard( float guard );      /* sets the guard's width (0: off); < 0 only reads; returns the previous width */
/* The sequential estimator (sources up to rs_hip_icp_reference_order_below) runs the reference's dist² statistics and its weighted
 * centroids in ONE pass, the 2.5 sigma cut of the weights (lib/rs/icp.h:396-401) taken at a guess of sigma; the pass stands when no
 * dist² lies between the guessed and the real cut, else the centroids are summed again.  Iterations that had to, over every calling
 * thread of the process, since the first ICP call (diagnostics). */
int32_t rs_hip_icp_faith_redone( void );
/* Test switch of that guess, in thousandths: 1000 (default) the guess as made, 0 no guess (statistics, centroids and normal equations
 * in three passes, as up to round 3), any other value scales the guessed cut — a guess that fails, for the tests of the check and of
 * the fallback.  Returns the previous value; < 0 only queries. */
int32_t rs_hip_icp_faith_guess( int32_t permille );
/* Sources above both thresholds (whole million-point scans): the parallel fp64 reduction, centred on the REFERENCE'S centroids —
 * the seven sums behind icp__compute_weighted_centroid (icp.h:136-148: Σw, Σw·p, Σw·q) are computed as the reference's own
 * sequential fp32 chains, bit for bit.  Those chains carry a systematic rounding drift (parts in a thousand of Σw once the running
 * sum's grid is coarser than the spread of the addends) that moves the reference's converged pose by up to 3e-4 from the exact
 * least-squares one; with its centroids the fp64 step lands within ~1e-6 of the reference's pose.  on = 0: plain fp64
 * moments (faster by the chains' cost, within 1e-4 of the reference in most runs only); on < 0 only reads; environment
 * RS_HIP_EXACT_CENTROIDS.  Returns the previous setting. */
int32_t rs_hip_icp_exact_centroids( int32_t on );
/* How many rs_hip_icp_align[_batch] calls the grid chains gave up so far (sums that keep changing binade — coordinates that straddle
 * the origin in a cancelling order) and ran again with the centroid sums by pass 2 of the replay: same result, ~10x the estimator time. */
int32_t rs_hip_icp_chains_gave_up( void );
/* A source cloud whose chains gave a problem up goes straight to the replay on its next `calls` calls as an ICP source (default 15;
 * environment RS_HIP_CHAINS_RETRY_AFTER) instead of paying for the failed attempt each time; 0 = every call tries the chains first
 * (tests that compare the two paths on centred scans use this); calls < 0 only reads.  The skip is per source cloud, whatever the
 * poses and the target.  Returns the previous value. */
int32_t rs_hip_icp_chains_retry_after( int32_t calls );

/* Many independent icp_align problems of one (source, target) pair, one per start pose
 * (apps/pose_proposal/main.cpp:190-202 runs exactly this loop): T1s is float[16*n], errs
 * float[n], iters int32[n] (may be NULL).  All problems advance in lock-step launches. */
int rs_hip_icp_align_batch( const rs_hip_cloud_t* source, const rs_hip_cloud_t* target,
                            float* T1s, int32_t n, const float* T2, float max_dist, float max_angle,
                            int32_t max_iter, int32_t fixed_iters, float* errs, int32_t* iters );

/* Many independent icp_align problems with a DIFFERENT source each against one target — the per-placement refine loop of
 * lib/rs/rs_database.h:220-230 (rsdb_refine_alignment_of_objects_to_scene) and apps/pose_proposal/main.cpp:190-202 — as one
 * call: problem p aligns sources[p] from T1s[16 p].  Sources of at most rs_hip_icp_reference_order_below() points (every call
 * site of the reference) advance in lock-step launches, grid.y = problem, each problem on its own source view; a batch that
 * holds a larger source runs problem by problem.  Every problem's pose, error and iteration count are what rs_hip_icp_align
 * returns for it alone, bit for bit. */
int rs_hip_icp_align_multi( const rs_hip_cloud_t* const* sources, const rs_hip_cloud_t* target,
                            float* T1s, int32_t n, const float* T2, float max_dist, float max_angle,
                            int32_t max_iter, int32_t fixed_iters, float* errs, int32_t* iters );

/* icp_find_corrs (lib/rs/icp.h:306-412), one call: compacted correspondences in source
 * order.  Output arrays are caller-allocated with capacity 3*n_source floats (weights:
 * n_source); *n_corrs receives the count. */
int rs_hip_icp_find_corrs( const rs_hip_cloud_t* source, const rs_hip_cloud_t* target,
                           const float* T1, const float* T2, float max_dist, float max_angle,
                           float* corr_pts1, float* corr_nor1, float* corr_pts2, float* corr_nor2,
                           float* weights, int32_t* n_corrs );

/* ---- alignment score ---------------------------------------------------------------- */

/* mgs_compute_object_alignment_score (apps/pose_proposal/pose_proposal.cpp:93-158) for
 * n_poses poses at once: scores[p] = score of `object` placed by poses[16*p..] against
 * `scene`.  radius = search_radii[search_lvl] (0.1 at the reference's search_lvl = 1, also
 * used as the distance sigma), max_n_neigh = 64 (proposal/verification) or 32 (refinement). */
int rs_hip_alignment_scores( const rs_hip_cloud_t* object, const rs_hip_cloud_t* scene,
                             const float* poses, int32_t n_poses, float radius, int32_t max_n_neigh,
                             float* scores );
/* Batches of at least `n_queries` (poses x object points) on a scene with a cell grid take the scene-space route: every transformed
 * query is keyed by the scene-aligned block it falls in (and the way its normal faces), the keys are radix-sorted, and a wave
 * searches 64 queries of one block whatever poses they come from — a third of the candidate evaluations of the object-space
 * launch for the same bits.  Default 65536 (environment RS_HIP_SCORE_SCENE_MIN; RS_HIP_SCORE_SCENE=0: never), and only
 * batches whose queries are dense in the scene take it (>= 128 per 0.1 m block of the lattice they span, RS_HIP_SCORE_SCENE_DENSITY: a
 * proposal's verification, not a grid search over the whole room); a threshold of 0 takes it whatever the density.
 * n_queries < 0 only reads.  Returns the previous threshold. */
int64_t rs_hip_score_scene_space_from( int64_t n_queries );

/* ---- label transfer ----------------------------------------------------------------- */

typedef struct rs_hip_placement
{
  float                 pose[16];   /* rs_obj_plcmnt_t.pose */
  const rs_hip_cloud_t* object;     /* the placed object's level-1 cloud */
  float                 radius;     /* search radius for this placement */
} rs_hip_placement_t;

/* rspf__assign_temporary_labels (lib/rs/rs_pointcloud_filters.cpp:738-778) over
 * placements[0..n) in the given order, continuing from the caller's labels / min_dists
 * (int8 / float, length = scene size, scene input order).  labels[j] receives
 * label_base + i + 1 for the winning placement i. */
int rs_hip_assign_labels( const rs_hip_cloud_t* scene, const rs_hip_placement_t* placements,
                          int32_t n, int32_t label_base, int8_t* labels, float* min_dists );

/* The per-placement "unary cost rows" of the same loop, for sharding placements across
 * GPUs: rows[i*scene_n + j] = dist² of scene point j to placement i's nearest object point
 * if one lies within the radius AND passes the 70° normal gate, +inf otherwise.
 * rows_device = 0: `rows` is a host pointer.  rows_device = 1: a device pointer (e.g. a torch tensor that will be
 * all-gathered), rows indexed by scene point in input order like the host form.  rows_device = 2: a device pointer, rows
 * indexed by the scene cloud's QUERY slot — the order the kernel produces them in (coalesced, no re-ordering pass);
 * rs_hip_fold_label_rows_device takes such rows when it is given the same scene cloud (every rank of the multi-GPU route
 * builds the same cloud from the same scene, so the rows of all ranks share that order). */
int rs_hip_label_rows( const rs_hip_cloud_t* scene, const rs_hip_placement_t* placements,
                       int32_t n, float* rows, int rows_device );

/* Ordered arg-min over gathered rows, host side: applies rows 0..n-1 in order with the
 * strict `<` of rs_pointcloud_filters.cpp:763 (earlier placement wins ties). */
void rs_hip_combine_label_rows( const float* rows, int32_t n_rows, int64_t scene_n, int32_t label_base,
                                int8_t* labels, float* min_dists );

/* The same ordered arg-min over rows that already sit in DEVICE memory — the gathered send buffers of the multi-GPU
 * route (placements sharded across GPUs, rows all-gathered over RCCL, then folded in the sorted order of
 * rs_pointcloud_filters.cpp:823-848).  Row k starts at rows_device + row_offsets[k] (in floats; row_offsets is a host
 * array, so rows of different ranks may sit anywhere in one gathered buffer) and holds scene_n floats.  labels / min_dists
 * are host arrays: continued from the caller's values like rs_hip_assign_labels, or — fresh != 0 — started on the device
 * from the loop's initial state (label 0, min_dist 1e9: rs_pointcloud_filters.cpp:799-802,820) without an upload. */
int rs_hip_fold_label_rows_device( const float* rows_device, const int64_t* row_offsets, int32_t n_rows, int64_t scene_n,
                                   int32_t label_base, int8_t* labels, float* min_dists, int32_t fresh,
                                   const rs_hip_cloud_t* rows_in_query_order_of /* NULL: rows in input order */ );

/* The lighter exchange of the same split: instead of one row per placement, a rank sends the PARTIAL of its own contiguous run of
 * the sorted arrangement — the loop's running (min_dist, label) after that run, started from (1e9, 0): 5 bytes per scene point
 * whatever the number of placements (4.9 MB instead of 31 MB per rank at 8 placements per rank and 1 M scene points).  Folding the
 * ranks' partials in rank order with the loop's strict `<` gives the bits of the sequential loop, because the runs are contiguous
 * and in order.  rs_hip_label_partial_device writes the partial of placements[0..n) into device memory (e.g. the RCCL send
 * buffer), indexed by the scene cloud's query slot; labels carry label_base + i + 1.  rs_hip_fold_label_partials_device folds
 * n_parts gathered partials (rank r's min_dists at base_device + min_offsets[r] floats, its labels at (int8*)base_device +
 * label_offsets[r] bytes; host arrays of offsets) and returns the result in input order. */
int rs_hip_label_partial_device( const rs_hip_cloud_t* scene, const rs_hip_placement_t* placements, int32_t n, int32_t label_base,
                                 float* min_dists_device, int8_t* labels_device );
int rs_hip_fold_label_partials_device( const float* base_device, const int64_t* min_offsets, const int64_t* label_offsets, int32_t n_parts,
                                       int64_t scene_n, int8_t* labels, float* min_dists, const rs_hip_cloud_t* in_query_order_of );

/* rspf_arrangement_to_labels ordering + two passes (lib/rs/rs_pointcloud_filters.cpp:780-848):
 * sorts placement indices (dynamic first, then by class index; stable), runs the dynamic
 * pass with `radius` and the static pass with 1.5*radius (or resets min_dists when
 * prioritize_static).  is_static / class_idx are per placement.  sorted_order (may be NULL)
 * receives the permutation; labels index into the sorted order, 1-based, 0 = unlabelled.  min_dists may be NULL (the reference
 * frees its own before it returns, :871-872: the shim's rsd_arrangement_to_labels does not download them). */
int rs_hip_arrangement_to_labels( const rs_hip_cloud_t* scene,
                                  const float* poses /* 16*n */, const rs_hip_cloud_t* const* objects /* n */,
                                  const int32_t* is_static, const int32_t* class_idx, int32_t n,
                                  float radius, int prioritize_static,
                                  int8_t* labels, float* min_dists, int32_t* sorted_order );

/* The same plus the function's tail (lib/rs/rs_pointcloud_filters.cpp:851-869): temporary labels -> in_pc->class_ids[lvl] /
 * instance_ids[lvl] — class of the labelled placement's object and the placement's uidx; label 0 gives
 * (unlabelled_class_idx = rsdb_get_class_idx( rsdb, "unlabelled" ), RSPF_MAX_INSTANCES = 1024).  uidx is per placement
 * (rs_obj_plcmnt_t.uidx).  labels / min_dists / sorted_order may be NULL. */
int rs_hip_arrangement_to_ids( const rs_hip_cloud_t* scene,
                               const float* poses /* 16*n */, const rs_hip_cloud_t* const* objects /* n */,
                               const int32_t* is_static, const int32_t* class_idx, const int32_t* uidx, int32_t n,
                               float radius, int prioritize_static, int32_t unlabelled_class_idx,
                               int32_t* class_ids, int32_t* instance_ids, int8_t* labels, float* min_dists, int32_t* sorted_order );

/* ---- level builder ---------------------------------------------------- */

/* rs_pointcloud__compute_level_poisson (lib/rs/rs_pointcloud.h:984-1106): Poisson-disk subsample of
 * `cloud` in INPUT order — the first point no earlier sample has marked becomes a sample and marks
 * every point within `radius` of it (the reference's searches return at most max_n_neigh points;
 * for level L it uses radius = voxel_size[L] = 0.005·2^L and max_n_neigh = 1024·L/4, 256 for L = 0,
 * :995-996,1011).  sample_idx (capacity = cloud size) receives the sample indices in increasing
 * order — the order in which the reference fills the level's arrays (:1090-1099); n_rounds (may be
 * NULL) the number of decision rounds the device needed.  If some point has more than max_n_neigh
 * points within the radius a reference search would be truncated to its max_n_neigh nearest; that
 * case is not reproduced: the call fails with RS_HIP_E_CAPACITY and writes nothing. */
int rs_hip_level_samples( const rs_hip_cloud_t* cloud, float radius, int32_t max_n_neigh,
                          int32_t* sample_idx, int32_t* n_samples, int32_t* n_rounds );

/* The level as a cloud of its own, built without leaving the device: the samples of `base` (as above), the gather of
 * their positions / normals (rs_pointcloud.h:1090-1099) and the level's search index
 * (rs_pointcloud_compute_search_grid, :849-863; cell_size as for rs_hip_cloud_create).  sample_idx (may be NULL,
 * capacity = size of base) and n_samples (may be NULL) receive the samples.  NULL on failure. */
rs_hip_cloud_t* rs_hip_cloud_create_level( const rs_hip_cloud_t* base, float radius, int32_t max_n_neigh, float cell_size,
                                           int32_t* sample_idx, int32_t* n_samples );

/* The gathers that end the level builder (lib/rs/rs_pointcloud.h:1090-1099): for each of n_arrays per-point arrays of the
 * base level (host pointers, `words[a]` 32-bit words per point: 3 for positions / normals / colours, 1 for radii, qualities,
 * class and instance ids), dst[a][i] = src[a][sample_idx[i]].  Entries with a NULL src or dst are skipped. */
int rs_hip_gather_attributes( const int32_t* sample_idx, int32_t count, int32_t n_src,
                              const void* const* src, const int32_t* words, void* const* dst, int32_t n_arrays );

/* ---- neighbourhood graph ------------------------------------------ */

/* rspf_compute_neighborhood (lib/rs/rs_pointcloud_filters.cpp:674-722): K = max_nn self-search
 * within sqrt(radius_sq) over `cloud` (needs normals), one weighted edge per (point, neighbour),
 * de-duplicated so that every undirected pair appears once, oriented as the reference's insertion
 * order leaves it ({i,j}, i<j, is (i,j) when j is among i's neighbours, (j,i) otherwise); self pairs
 * (i,i) are included, as in the reference.  weight = (1 - pow(d²/(4 r²), dist_exp)) *
 * powf(clamp(n·m,0,1), angle_exp) (the reference calls it with 8, 0.05², 15, 16).
 * Edges are written ordered by idx1, then by (dist², idx2); capacity must be >= n*max_nn. */
int rs_hip_compute_neighborhood( const rs_hip_cloud_t* cloud, int32_t max_nn, float radius_sq,
                                 float dist_exp, float angle_exp,
                                 int32_t* idx1, int32_t* idx2, float* weight, int64_t capacity, int64_t* n_edges );

/* ---- scene-coverage term of the arrangement optimiser ---------------- */

/* The voxel grid of lib/rs/intersect.h:59-109 over a scene's bounding box (isect_grid3d_init: fattened by
 * 0.3, ceilf(extent/voxel)+1 cells per axis) with the scene's level-2 points rasterised into it
 * (rsao_rasterize_scene_to_grid, apps/segment_transfer/arrangement_optimization.cpp:1064-1079: points
 * with quality < threshold are skipped; quality may be NULL).  One bit per voxel on the device. */
typedef struct rs_hip_coverage rs_hip_coverage_t;
rs_hip_coverage_t* rs_hip_coverage_create( const float bbox_min[3], const float bbox_max[3], float voxel_size,
                                           const float* scene_pos, const float* scene_quality, int64_t n_scene,
                                           float quality_threshold );
void rs_hip_coverage_destroy( rs_hip_coverage_t* c );
int  rs_hip_coverage_info( const rs_hip_coverage_t* c, int32_t res[3], float origin[3], int64_t* n_cells, int64_t* valid_cells );
/* Copies the scene grid out as the reference's byte array (1 = RSAO_CELL_ACTIVE), n_cells bytes. */
int  rs_hip_coverage_scene_grid( const rs_hip_coverage_t* c, uint8_t* data );

/* rsao__compute_scene_coverage_score (:344-373) for a batch of arrangements in one launch: arrangement a
 * consists of the placements [first_placement[a], first_placement[a+1]); placement k puts objects[k] (its
 * level-2 cloud) at poses[16k..] and is skipped when is_static[k] (rsdb_is_object_static, :1095-1096).
 * scores[a] = agreeing / valid scene voxels (0 when the scene grid is empty); agree (optional) receives
 * the integer numerators.  The simulated-annealing loop evaluates one arrangement per iteration
 * (:388); candidate moves can be scored together. */
int  rs_hip_coverage_scores( rs_hip_coverage_t* c, const rs_hip_cloud_t* const* objects, const float* poses,
                             const int32_t* is_static, const int32_t* first_placement, int32_t n_arrangements,
                             float* scores, int32_t* agree );

/* ---- host-side helpers shared by the drop-in shim (exact reference arithmetic) ------- */

/* msh_mat4_inverse / msh_mat4_mul (lib/msh/msh_vec_math.h:1818-1905, 1441-1476) */
void rs_hip_mat4_inverse( const float* m, float* out );
void rs_hip_mat4_mul( const float* a, const float* b, float* out );
/* The device's sinf/cosf — a restatement of the host libm's algorithm, since the reference's poses carry libm's bits
 * (msh_rotate, msh_vec_math.h:2091-2092) — evaluated on the host for n arguments, so that a CPU test can hold it
 * against the libm of the machine (no GPU needed). */
void rs_hip_sincosf_model( const float* x, int64_t n, float* sin_out, float* cos_out );
/* icp_estimate_rigid_xform_pt2pl (lib/rs/icp.h:210-298) on host arrays (device reduction) */
int  rs_hip_icp_estimate_pt2pl( const float* pts1, const float* pts2, const float* nor2,
                                const float* weights, int32_t n, float* T1, float* err );

#ifdef __cplusplus
}
#endif
#endif /* RESCAN_HIP_H */
