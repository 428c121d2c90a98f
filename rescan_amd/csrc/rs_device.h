// Shared host/device declarations for librescan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace rs {

// Device view of a cloud: 16-byte records sorted by grid cell, cell offset table.
// pos[s] = {x, y, z, bitcast(original index)};  nor[s] = {nx, ny, nz, 0}.
// cell id = (z*h + y)*w + x, so the cells of one (y,z) row are contiguous in x and a row's
// x-interval [x0,x1] is the single span cell_start[row+x0] .. cell_start[row+x1+1].
constexpr int EVAL_SHARDS = 1024;
constexpr int STAT_SHARDS = 256;    // integer accumulators the tiles of an ICP search add their dist² statistics to

struct GridView
{
  const float4*   pos;
  const float4*   nor;          // may be null
  const uint32_t* cell_start;   // w*h*d + 1 entries
  float minx, miny, minz;       // grid origin
  float inv_cell;               // 1 / cell edge (0 for the one-cell brute layout)
  float cell;                   // cell edge
  int   w, h, d;
  int   n;
  unsigned long long* evals;    // profiling only (null otherwise): EVAL_SHARDS counters, 64 B apart, of candidates staged and evaluated
};

struct Xform { float m[16]; };  // column-major, passed by value (lands in SGPRs)

// Device view of a cloud used as a QUERY set: the same points in Hilbert-curve order, cut
// into tiles of at most 64 consecutive points with a bounded spatial extent.  One wave
// handles one tile, so the 64 lanes of a wave are always spatial neighbours and share one
// small candidate box.  pos[s].w = bitcast(original index).
struct QueryView
{
  const float4*   pos;
  const float4*   nor;     // may be null
  const uint32_t* tiles;   // n_tiles + 1 offsets into pos/nor
  int n, n_tiles;
};

// One placement of the label kernel.
struct PlacementDev
{
  GridView g;         // the placed object's cloud
  Xform    inv;       // msh_mat4_inverse(pose)
  Xform    nmat;      // msh_mat4_transpose(pose)
  float    radius;    // search radius
  float    radius_sq; // (float)((double)radius*(double)radius)
};

enum { ICP_NMOM = 35,     // raw moments reduced per ICP iteration (see k_icp_moments)
       ICP_NRES = 39 };   // per problem result record: 35 moments + {n_corr, mean, stddev, -}

// ---- launchers (the rs_*.hip kernel files) ----------------------------------------------------------
constexpr int HEAVY_SLOTS = 2048;   // wave slots at the front of phase A's grid reserved for the previous iteration's slow tiles
constexpr int HEAVY_CLASSES = 8;     // one list per XCD class of the natural order: a listed tile stays on the XCD whose L2 holds its part of the target
constexpr int HEAVY_PER_CLASS = HEAVY_SLOTS / HEAVY_CLASSES;
constexpr int HEAVY_MEAN = HEAVY_CLASSES;                             // word: mean candidates streamed per tile of the launch (a sample) / HEAVY_MEAN_REF, in 1/256ths, written when the iteration ends
constexpr int HEAVY_HDR = HEAVY_CLASSES + 8;                         // header words before the lists
constexpr int HEAVY_MEAN_REF = 320;  // mean up to which the thresholds below apply as they are (the 1 M-point headline: 300 after the cold launch, 200 later); beyond, they grow with it
__host__ __device__ inline size_t heavy_stride( int n_tiles ) { return (size_t)n_tiles + HEAVY_SLOTS + HEAVY_HDR; }

// One problem of a multi-source batch: its own source view and where its rows of the per-point / per-tile arrays begin.
struct IcpProblem
{
  QueryView  src;
  const int* by_orig;          // original source index -> query slot
  long long  pt_off, tile_off, heavy_off;
};

struct IcpLaunch
{
  GridView     tgt;
  QueryView    src;            // source cloud (query order); multi-source batches: bound per problem on the device (icp_bind)
  const IcpProblem* multi;     // device, n_prob entries (null: one source for all problems)
  long long    pt_off, tile_off, heavy_off;   // set by icp_bind: where this problem's rows begin (points / tiles / slow-tile lists)
  int          max_n, max_tiles;              // largest problem's (grid sizes); single source: src.n, src.n_tiles
  int          n_prob;         // batch size
  float*       T1;             // device, n_prob x 16: current poses (updated on the device by every iteration)
  int*         active;         // device, n_prob flags (0 = skip; cleared on the device when a problem stops)
  Xform        T2i;
  float        radius, radius_sq, gate_tmin;
  int          K;
  // outputs / workspace (device)
  int*    m_slot;   // n_prob x nq : matched target slot or -1
  float*  m_d2;     // n_prob x nq
  float*  m_dot;    // n_prob x nq
  // dist² statistics of the correspondences: n_prob x STAT_SHARDS x {Σ1, Σd²·stat_s1, Σd⁴·stat_s2, -} as integers
  // (null: not wanted — reference-order estimator, find_corrs / estimate-only entry points)
  unsigned long long* stat_acc;
  float   faith_guess_scale;       // 1: the guess of the cut as made; 0: no guess (three passes); anything else: the guess scaled (tests: a guess that fails)
  int*    faith_redone;            // (diagnostics, cumulative; may be null) iterations of k_icp_faithful whose one-pass statistics + centroids did not stand
  double  stat_s1, stat_s2, stat_i1, stat_i2;   // fixed-point scales (powers of two chosen from the radius) and their inverses
  double* mom_part;   // n_prob x n_mom_blocks x ICP_NMOM
  double* res;        // n_prob x ICP_NRES : moments [0,35), then n_corr, mean, stddev, queued tiles (written by k_icp_moments)
  int     n_mom_blocks;
  int*    queue;        // n_prob x n_tiles : tiles handed to the cooperative kernel
  int*    queue_count;  // n_prob
  int     solo_stages;  // candidates a lone wave streams before handing an unsettled tile off
  int     coop_waves;   // waves per queued tile in the cooperative kernel (2, 4 or 8)
  int     coop_all;     // 1: no phase A, the cooperative kernel searches every tile (small launches)
  // loop state kept on the device (lib/rs/icp.h:441-493): k_icp_update solves and updates it
  int     solve;        // 0: reductions only (estimate-only / find_corrs entry points), 1: also solve + update the state
  int     iter_index;   // i of icp.h:444
  int     fixed_iters;  // benchmark mode: no convergence test
  int*    iters;        // n_prob: find_corrs calls made
  float*  err;          // n_prob: last RMS error (icp.h:253)
  float*  prev_err;     // n_prob
  int*    queued;       // n_prob: tiles phase A handed off in the last iteration (host heuristics, diagnostics)
  int*    ticket;       // n_prob (of 2 n_prob words, zero at the start of a call): [prob] = 1 when a stop test of the problem came within stop_guard of its threshold
  float   stop_guard;   // (0: off) estimators that are not the reference's own order flag a problem whose |delta err| passes within this of 1e-5 (icp.h:489): the host runs it again in reference order
  int     warm;         // m_slot holds last iteration's matches: use them as starting candidates
  int     seed;         // (when !warm) start from the best usable point of the query's own cell
  int     bounded_only; // (when warm) phase A only takes tiles whose lanes all start from a candidate; the rest goes straight to the cooperative kernel
  int     by_rows;      // (when warm) per-row sweep of the tiles whose lanes all start from a candidate (rs_search.h: sweep_by_rows)
  // certificates issued by every search and consulted when a point has no usable previous match (rs_icp_search.hip: icp_certificate); null = off
  float*  cert_r;       // n_prob x nq
  float*  cert_dot;     // n_prob x nq
  float*  cert_slack;   // n_prob x nq: rank certificates (null: off) — see icp_certificate
  float*  T1_prev;      // device, n_prob x 16: the poses the previous iteration searched with
  float   tgt_nor_max;  // max |normal| over the target cloud
  unsigned long long* dbg;   // diagnostic builds only: per-tile {cycles, candidates} of phase A (null otherwise)
  const float* w_explicit;   // if non-null: weights given per query (estimate-only entry point)
  // slowest-first start of phase A's tiles: every iteration lists its slow tiles for the next one (null: off)
  //   per problem (heavy_stride words): HEAVY_CLASSES counts | mean candidates streamed per tile | pad |
  //   HEAVY_CLASSES x HEAVY_PER_CLASS tile ids | n_tiles x (bits 0-1: 1 a front slot has it, 2 handed to the cooperative kernel at once; bits 2..: candidates it streamed)
  const int* heavy_in;
  int*    heavy_out;
  int     heavy_streamed;   // a tile that streamed at least this many candidates is listed
  int     heavy_total;      // warm launch: a bounded tile swept tile-wide (boxes too tall for the per-row sweep) is handed off too when the first 64 cell rows of its box hold this many candidates (0: never)
  int     heavy_handoff;    // ... and one that streamed this many in a warm launch goes to the cooperative kernel from the next iteration on (flag 2)
  // reference-order estimator (rs_icp_estimate.hip: k_icp_faithful); faith == null: fp64 moments
  const int* by_orig;   // original source index -> query slot (null: identity)
  float*  faith;        // n_prob x FAITH_REC x nq: the correspondences in the source's own order
  // "exact centroids" estimator (large sources): the fp64 moments, but the seven sums behind the two weighted centroids
  // (icp.h:136-148: Σw, Σw·p, Σw·q) as the reference's own sequential fp32 chains — see launch_icp_exact_centroids
  int     exact_centroids;
  const double* centroid_totals;   // n_prob x 3 x ICP_NMOM (ReplayBufs::totals): [ICP_NMOM + 0..6] = the seven chain totals
  // ... and their fast form (rs_icp_estimate.hip: "grid chains"): every search writes one 48-byte record per source point at the point's
  // ORIGINAL index — {p.xyz, dist² (< 0: no match)} {q.xyz, dot} {n.xyz, -} — which the estimator's kernels then read in the
  // reference's own order, coalesced (null: not wanted)
  float4* rec;                     // n_prob x n x REC_F4
};

// The seven centroid chains (Σw, Σw·p, Σw·q: icp.h:136-148) as sequential fp32 sums, computed on the integer grid of the
// running sum's binade (rs_icp_estimate.hip: "grid chains").  Segments of 64 source points (original order), blocks of 64 segments.
constexpr int CH_ROWS = 7, CH_SEG = 64, CH_BLK = 64;
constexpr int REC_F4 = 3;        // float4 per correspondence record
struct ChainRec { int e_sign; int lo[3], hi[3], D[3]; };   // exponent guess | sign << 8; per exponent e-1, e, e+1: the start mantissas it holds for and the advance
struct ChainBufs
{
  int       n_seg, n_blk;
  double*   segsum;     // n_prob x CH_ROWS x n_seg : fp64 sums of the segments' addends
  double*   blksum;     // n_prob x CH_ROWS x 4 n_blk : ... and of the quarter blocks' (the guess of the running sum at a segment's start = their prefix)
  ChainRec* seg;        // n_prob x CH_ROWS x n_seg
  ChainRec* blk;        // n_prob x CH_ROWS x n_blk : 64 segments composed
  double*   totals;     // n_prob x 3 x ICP_NMOM (the layout of ReplayBufs::totals; the chains fill [ICP_NMOM + 0..6])
  int*      guess;      // n_prob x CH_ROWS x n_seg : exponent | sign << 8 of the guess at every segment's start, kept from the last refresh
  int       refresh;    // 1: the guesses are made anew from this iteration's fp64 sums (moments before records); 0: the kept ones serve again
  int*      failed;     // n_prob: a walk gave up (too many binade changes for this method: chain_walk_row) — the host runs the problem again another way
  int*      done;       // n_prob, zero between launches: k_icp_update_wide's count of finished workgroups
  int*      resolved;   // n_prob: segments the walks had to add up one addend after the other (diagnostics)
  int*      chk;        // (RS_HIP_CHAIN_DEBUG) per chain 4 + 3 x 4096 words: the walk's steps {end segment, value bits, kind}, checked against the plain sum by the walk itself
  int       dbg_reps;   // (RS_HIP_CHAIN_DEBUG=n: k_chain_walk walks n times, the stamps are the last walk's — warm caches)
  int*      dbg;        // RS_HIP_CHAIN_DEBUG: per chain 1 + 64 x 8 words — count, then {segment, value bits, guess, lo / hi / D of the class tried} of the first 64 such segments
  float*    addends;    // (lane chains only) the seven addend rows of every problem: 8 x ( total source points + 4 n_prob ) floats
};
void   launch_icp_chain_centroids( const IcpLaunch& L, const ChainBufs& B, hipStream_t st );
void   launch_icp_plain_from_records( const IcpLaunch& L, const ChainBufs& B, hipStream_t st );      // fp64 moments + update, no chains (L.exact_centroids == 0)
// the same seven sums for object-sized sources, one wave per chain, any number of (differently sized) problems: B.totals, B.done, B.resolved, B.addends only
void   launch_icp_lane_chains( const IcpLaunch& L, const ChainBufs& B, hipStream_t st );
inline int chain_segments( int n ) { return ( n + CH_SEG - 1 ) / CH_SEG; }
inline int chain_blocks( int n ) { return ( chain_segments( n ) + CH_BLK - 1 ) / CH_BLK; }
void launch_icp_corr( const IcpLaunch& L, hipStream_t st );     // phase A, phase B (the tiles add their dist² statistics to L.stat_acc)
void launch_icp_moments( const IcpLaunch& L, hipStream_t st );  // weights + moments (+ solve and loop-state update if L.solve)
constexpr int FAITH_REC = 11;   // per correspondence: dist² (< 0: none), dot | weight, p, q, n
void launch_icp_faithful( const IcpLaunch& L, hipStream_t st ); // the same step with the reference's own accumulation order and precisions
// ... and the same bits computed in parallel (rs_icp_estimate.hip: "replay"): per problem and accumulator row (ICP_NMOM rows per pass)
struct ReplaySeg;
struct ReplayBufs
{
  int     n_seg;        // segments of 128 source points
  int     n_super;      // superblocks of 64 segments
  double* segsum;       // n_prob x ICP_NMOM x n_seg
  double* guess;        // n_prob x 3 passes x ICP_NMOM x n_seg
  ReplaySeg* seg;       // n_prob x ICP_NMOM x n_seg
  ReplaySeg* super;     // n_prob x ICP_NMOM x n_super: 64 segments composed
  double* totals;       // n_prob x 3 passes x ICP_NMOM: the accumulators' final values
  int*    redone;       // n_prob: segments re-added sequentially (diagnostics; may be null)
};
void   launch_icp_replay( const IcpLaunch& L, const ReplayBufs& B, hipStream_t st );
// fp64 moments + the reference's own fp32 chains for Σw, Σw·p, Σw·q (pass 2 of the replay) + solve with those centroids
void   launch_icp_exact_centroids( const IcpLaunch& L, const ReplayBufs& B, hipStream_t st );
void   launch_icp_exact_centroids_from_records( const IcpLaunch& L, const ReplayBufs& B, const ChainBufs& C, hipStream_t st );
int    replay_segments( int n_source );
int    replay_superblocks( int n_source );
size_t replay_seg_bytes();

struct ScoreLaunch
{
  GridView     scene;
  QueryView    obj;
  int          n_poses;
  const float* poses;      // device n_poses x 16
  float        radius_sq, gate_tmin;
  int          K;
  double       sigma;      // (double)radius
  double*      part;       // n_poses x n_tiles
  float*       scores;     // n_poses
  int*         queue;      // n_poses x n_tiles items (pose*n_tiles + tile) for the cooperative kernel
  int*         queue_count;
  int          solo_stages;
  unsigned long long* hist;   // diagnostic builds only (RS_HIP_SCORE_HIST): 6 x 65 counters, see k_score
  // Scene-space batches (rs_score.hip: k_score_keys / k_score_scene / k_score_gather): the n_poses x n transformed queries sorted by
  // the scene-aligned block ("parent", edge sq_parent ~ the radius) they fall in, so that a wave's queries share one neighbourhood
  // whatever pose they come from.  sq_key_a == null: the object-space launch (k_score).
  uint32_t*    sq_key_a;   // items: keys as computed (parent index << sq_fine_bits | direction of the normal, octant of the parent); sq_n_parents << sq_fine_bits = no candidate can exist
  uint32_t*    sq_key_b;   // ... sorted
  uint32_t*    sq_val_a;   // items: pose * n + query slot
  uint32_t*    sq_val_b;
  double*      sq_pq;      // n_poses x n: score of every (pose, query slot)
  void*        sq_tmp;     // radix sort workspace
  size_t       sq_tmp_bytes;
  int          sq_bits;    // key bits to sort
  uint32_t*    sq_hist;    // round 6: 2^sq_bits + 1 counters — the keys are ordered by COUNTING (k_score_keys counts, a scan, k_score_scatter places) instead of
                           // by three radix passes; null: the radix sort (more key bits than a table is worth)
  int          sq_fine_bits;   // low key bits below the parent index (6)
  int          sq_n_parents, sq_dpx, sq_dpy, sq_dpz;
  float        sq_ox, sq_oy, sq_oz;     // origin of the parent lattice (the scene grid's, moved out by whole parents)
  float        sq_inv_fine;             // 4 / parent edge
  float        sq_lox, sq_hix, sq_loy, sq_hiy, sq_loz, sq_hiz;   // the scene grid's box grown by the radius: a query outside has nothing to match
  int          sq_cull;    // sweeps skip the cells farther from the wave's queries than they look (rs_search.h: Cull)
  int          sq_nbin;    // the key's low bits carry the query normal's dominant direction (0: the quarter-parent sub-cell instead)
};
void launch_score( const ScoreLaunch& L, hipStream_t st );

struct LabelLaunch
{
  QueryView    scene;          // scene cloud (query order)
  const PlacementDev* pl;      // device array
  int          n_pl, label_base;
  float        gate_tmin;
  int8_t*      labels;         // scene QUERY order (chain mode) or null
  float*       min_d;          // scene QUERY order (chain mode) or null
  float*       rows;           // n_pl x ns rows, QUERY order (row mode) or null
  int          fresh;          // chain mode: start from (label 0, min_dist 1e9) instead of reading labels / min_d
};
void launch_label( const LabelLaunch& L, hipStream_t st );
// between a cloud's query order and its input order, by gathering (n_f float arrays of n entries back to back, and/or one int8 array)
void launch_label_to_input_order( const int* by_orig, long long n, const float* in_f, float* out_f, int n_f, const int8_t* in_b, int8_t* out_b, hipStream_t st );
void launch_label_to_query_order( const float4* qpos, long long n, const float* in_f, float* out_f, const int8_t* in_b, int8_t* out_b, hipStream_t st );
void launch_label_ids_to_input_order( const int* by_orig, long long n, const int8_t* labels_q, const float* mind_q, const int* plc_class, const int* plc_uidx,
                                      int unlabelled_class, int* class_ids, int* instance_ids, int8_t* labels, float* min_d, hipStream_t st );
void launch_gather_words( const uint32_t* src, const int* idx, long long count, int words, uint32_t* dst, hipStream_t st );
// ordered fold of device-resident rows (row k at rows + offsets[k], n floats each; offsets is a device array)
void launch_label_fold( const float* rows, const long long* offsets, int n_rows, long long n, int label_base, int8_t* labels, float* min_d, bool fresh, hipStream_t st );
// ordered fold of per-rank partials (min_dists at base + min_off[r] floats, int8 labels at (int8*)base + lab_off[r] bytes; device arrays of offsets)
void launch_label_fold_partials( const float* base, const long long* min_off, const long long* lab_off, int n_parts, long long n, int8_t* labels, float* min_d, hipStream_t st );

struct RowsLaunch
{
  GridView     tgt;
  QueryView    q;          // queries (w = original query index)
  int          K;
  float        radius, radius_sq;
  float*       d2;         // nq x K (original query order)
  int*         idx;        // nq x K
  int*         nn;         // nq
};
void launch_rows( const RowsLaunch& L, hipStream_t st );
// one wave per query, queries in the caller's order (AoS xyz on the device); *overflow is set (and nn = -1 written) for a
// query with more than 1024 points within the radius
void launch_rows_wave( const GridView& g, const float* q3, int nq, int K, float radius, float radius_sq,
                       float* out_d2, int* out_idx, int* out_nn, int* overflow, hipStream_t st );

// Scene-coverage term of the arrangement optimiser (apps/segment_transfer/arrangement_optimization.cpp:344-373,
// 1064-1106 on the voxel grid of lib/rs/intersect.h:59-109): bitmaps instead of byte grids.
struct VoxGrid { int x_res, y_res, z_res, n_cells; float ox, oy, oz, inv_voxel; };
struct CoveragePlacement
{
  const float4* pos;     // object level cloud (any order)
  int           n;
  int           arrangement;
  Xform         pose;
};
struct CoverageLaunch
{
  VoxGrid      grid;
  const uint32_t* scene_bits;     // n_words
  uint32_t*    arr_bits;          // n_arr x n_words, zero on entry
  int          n_words;
  const CoveragePlacement* plc;   // device array, non-static placements only
  int          n_plc, max_pts;
  int*         agree;             // n_arr, zero on entry
};
void launch_voxel_mark( const VoxGrid& g, const float* pos /* AoS xyz */, const float* quality /* or null */, float threshold,
                        long long n, uint32_t* bits, hipStream_t st );
void launch_popcount( const uint32_t* bits, int n_words, int* out /* zero on entry */, hipStream_t st );
void launch_coverage( const CoverageLaunch& L, hipStream_t st );

// Device-side cloud construction (rs_build.hip)
void   launch_build_bounds( const float* pos3, const float* nor3, int n, unsigned* out8, hipStream_t st );
void   launch_build_mark( const float* pos3, int n, const float mn[3], float inv, unsigned long long db, unsigned long long dc, uint32_t* bits, hipStream_t st );
void   launch_build_cellids( const float* pos3, int n, const float mn[3], float inv_cell, const int dims[3], uint32_t* cid, uint32_t* iota, uint32_t* counts, hipStream_t st );
void   launch_build_gather( const float* pos3, const float* nor3, const uint32_t* order, int n, float4* spos, float4* snor, hipStream_t st );
void   launch_build_count_runs( const uint32_t* sorted, int n, int* out, hipStream_t st );
void   launch_build_hilbert( const float* pos3, int n, const float mn[3], float scale, uint32_t* key, uint32_t* iota, hipStream_t st );
void   launch_build_tile_flags( const float4* qpos, int n, float max_extent, uint32_t* flags, uint32_t* jump_a, uint32_t* jump_b, hipStream_t st );
void   launch_build_tile_scatter( const uint32_t* flags, const uint32_t* scanned, int n, uint32_t* tiles, hipStream_t st );
void   launch_build_inverse( const float4* qpos, int n, int* by_orig, hipStream_t st );
size_t build_sort_temp_bytes( int n, int bits );
int    build_sort_pairs( void* tmp, size_t bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, int n, int bits, hipStream_t st );
size_t build_scan_temp_bytes( size_t n );
int    build_exclusive_scan( void* tmp, size_t bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t st );

// Neighbourhood graph (rspf_compute_neighborhood): from self-search rows to unique weighted edges.
// Level builder (lib/rs/rs_pointcloud.h:984-1106): a self-search of one cloud listing, per point, the LATER points
// (larger original index) within the radius, then a propagation of decisions along those rows (rs_rows.hip).
struct LevelLaunch
{
  GridView     tgt;
  QueryView    q;            // the same cloud's query layout
  const int*   by_orig;      // original index -> query slot
  int          n;
  float        radius, radius_sq;
  int*         n_earlier;    // n (query slot): earlier points within the radius
  int*         n_later;      // n + 1 (query slot): later points within the radius (row lengths)
  const unsigned* offset;    // n + 1: exclusive scan of n_later
  int*         adj;          // original indices of the later neighbours, rows in query-slot order
  int*         word;         // n (original index): earlier neighbours not yet known to be covered | LEVEL_COVERED
  int*         state;        // n (original index): 1 = sample
  const int*   front_in;     // frontier of this step: (original index << 1) | is_sample
  const int*   front_count_in;
  int*         front_out;
  int*         front_count_out;
  int*         over_cap;     // device flag: some point has more than max_n_neigh points within the radius
  int          max_n_neigh;
  unsigned*    flags;        // n + 1 (original index): 1 = sample (for the final compaction)
  const unsigned* flag_scan; // n + 1
  int*         samples;      // output: sample indices, increasing
};
void launch_level_neighbours( const LevelLaunch& L, bool write, hipStream_t st );
void launch_level_init( const LevelLaunch& L, hipStream_t st );
void launch_level_frontier( const LevelLaunch& L, int lanes_per_item /* 1, 8 or 64 */, int blocks, hipStream_t st );
void launch_level_flags( const LevelLaunch& L, hipStream_t st );
void launch_level_scatter( const LevelLaunch& L, hipStream_t st );
// out3[i] = xyz of the base cloud's point samples[i] (query layout + original -> slot map)
void launch_level_gather( const int* samples, int count, const int* by_orig, const float4* qpos, const float4* qnor,
                          float* pos3, float* nor3, hipStream_t st );

struct EdgeLaunch
{
  int          n, K;
  const float* row_d2;    // n x K  (original order, ascending)
  const int*   row_idx;   // n x K
  const int*   row_nn;    // n
  const float* nor;       // original-order AoS normals, 3*n floats (device)
  float        radius_sq, dist_exp, angle_exp;
  int          dist_int, angle_int;   // exponents as small non-negative integers, or -1 (use pow/powf)
  int*         count;     // n       : kept edges of row i
  unsigned*    offset;    // n + 1   : exclusive scan of count
  int*         e1; int* e2; float* ew;   // outputs
};
void launch_edge_count( const EdgeLaunch& L, hipStream_t st );   // fills count
void launch_edge_scan( const EdgeLaunch& L, hipStream_t st );    // count -> offset (single workgroup, fixed order)
void launch_edge_write( const EdgeLaunch& L, hipStream_t st );   // fills e1/e2/ew

} // namespace rs
