/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * rs_oracle: a plain-C, single-threaded CPU restatement of the Rescan hot path
 * (bounded-K radius search, point-to-plane ICP, alignment score, label
 * transfer), written from the reference's behaviour, each function citing the
 * reference file:line it follows (paths relative to /root/reference).
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every function
 * here bit-for-bit against oracle/_ref (the real reference compiled in place)
 * on seeded inputs when /root/reference is present, and tests/golden/ holds
 * vectors generated from oracle/_ref by oracle/gen_golden.py.  That includes
 * the label loops and the neighbourhood graph (lib/rs/rs_pointcloud_filters.cpp:
 * 674-879): that TU includes the un-vendored gco-v3.0 header, but only its last
 * function (rspf_smooth_labels, :881-) uses gco, so oracle/Makefile compiles the
 * file's own lines 1-14 + 16-879 — every line but the gco include, none edited,
 * <cassert>/<cstring> force-included — into oracle/_ref/libref_filters.so
 * (driver: oracle/ref_filters_driver.cpp), and orc_arrangement_to_labels,
 * orc_assign_labels and orc_compute_neighborhood are held against
 * rspf_arrangement_to_labels, rspf__assign_temporary_labels and
 * rspf_compute_neighborhood themselves (tests/test_oracle_vs_ref.py::
 * test_labels_vs_reference_text, ::test_neighborhood_vs_reference_text; the
 * labels_*.npz, neighborhood_*.npz and bench_seed*.npz label fields come from
 * that build).  Post-smoothing labels (gco) are out of scope and unpinned.
 * The scene-coverage term (orc_voxgrid_*, orc_rasterize_*, orc_coverage_score)
 * is pinned against the reference TU itself: apps/segment_transfer/
 * arrangement_optimization.cpp compiles from its own sources (oracle/_ref/libref_ao.so).
 * The level builder (orc_level_poisson) is pinned against the reference's own
 * rs_pointcloud__compute_level_poisson, which that same library carries (lib/rs/rs_pointcloud.h is
 * header-only; oracle/ref_ao_driver.cpp: ref_level_poisson), and against tests/golden/level.npz.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library.  The product (rescan_amd/, include/) never links or loads it.
 */
#ifndef RS_ORACLE_H
#define RS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_grid orc_grid_t;

/* lib/msh/msh_hash_grid.h:388-541 (dim = 3) */
orc_grid_t* orc_grid_create( const float* pts, int32_t n_pts, float radius );
void        orc_grid_destroy( orc_grid_t* g );
void        orc_grid_info( const orc_grid_t* g, int64_t dims[3], double* cell, float minp[3], uint32_t* max_in_bin );

/* lib/msh/msh_hash_grid.h:1090-1259; nn[] is int64 (reference: size_t) */
uint64_t orc_radius_search( const orc_grid_t* g, const float* query, int64_t nq, float radius,
                            int64_t k, int sort, float* dists, int32_t* inds, int64_t* nn );

/* lib/msh/msh_vec_math.h:1441,1554,1818,1938,2064,2089,868 — column-major float[16] */
void orc_mat4_mul( const float* a, const float* b, float* out );
void orc_mat4_inverse( const float* m, float* out );
void orc_mat4_transpose( const float* m, float* out );
void orc_translate( const float* m, const float* t, float* out );
void orc_rotate( const float* m, float angle, const float* axis, float* out );
void orc_xform_points( const float* m, const float* in, int64_t n, int is_point, float* out );
void orc_normalize( const float* in, int64_t n, float* out );
float orc_mean( const float* v, int n );                 /* lib/msh/msh_std.h:1811-1816 */
float orc_stddev( float mean, const float* v, int n );   /* lib/msh/msh_std.h:1820-1825 */

/* lib/rs/icp.h:306-412; output arrays have capacity n1; returns n_corrs */
int32_t orc_icp_find_corrs( const float* pts1, const float* nor1, int32_t n1,
                            const float* pts2, const float* nor2, int32_t n2, const orc_grid_t* index2,
                            const float* T1, const float* T2, float max_dist, float max_angle,
                            float* c_pts1, float* c_nor1, float* c_pts2, float* c_nor2, float* w );
/* lib/rs/icp.h:210-298 */
float orc_icp_estimate_pt2pl( const float* p1, const float* p2, const float* n2, const float* w,
                              int32_t n, float* T1 );
void orc_hover_bound( double out[8] );   /* mode 4's last estimator call: per centroid chain the sum of half-ulps over its hovering addends; [7] = the weight total */
/* WHAT-IF estimators (rs_oracle.c): icp_align's loop with the estimator's accumulators swapped (mode 0 = the reference's own) */
float orc_icp_iterate_variant( const float* pts1, const float* nor1, int32_t n1, const float* pts2, const float* nor2, int32_t n2,
                               float* T1, const float* T2, float max_dist, float max_angle, int32_t n_iters, int32_t stop_test,
                               int32_t mode, int32_t* iters_done, float* errs_out );
/* lib/rs/icp.h:416-500; n_iters (optional) receives the number of find_corrs calls made */
float orc_icp_align( const float* pts1, const float* nor1, int32_t n1,
                     const float* pts2, const float* nor2, int32_t n2,
                     float* T1, const float* T2, float max_dist, float max_angle, int32_t* n_iters );

/* apps/pose_proposal/pose_proposal.cpp:93-158 with search_lvl = 1 (radius 0.1, sigma 0.1);
 * scene_grid must have been built with radius 0.05 (lib/rs/rs_pointcloud.h:862). */
void orc_alignment_scores( const orc_grid_t* scene_grid, const float* scene_nor,
                           const float* obj_pos, const float* obj_nor, int32_t n_obj,
                           const float* poses, int32_t n_poses, int32_t max_n_neigh, float* scores );

/* lib/rs/rs_pointcloud_filters.cpp:724-879 on raw arrays.
 * placements: pose (16 floats each), object index, uidx; objects: level-1 cloud + grid
 * (radius 0.05), class idx, static flag.  Outputs: labels (int8, 1-based index into the
 * *sorted* arrangement, 0 = none), min_dists, sorted_order (permutation applied by the
 * qsort at :826), class_ids / instance_ids per scene point (:851-869). */
typedef struct orc_object
{
  const float* pos; const float* nor; int32_t n; const orc_grid_t* grid;
  int32_t class_idx; int32_t is_static;
} orc_object_t;

typedef struct orc_placement
{
  float pose[16]; int32_t object_idx; int32_t uidx;
} orc_placement_t;

void orc_arrangement_to_labels( const float* scene_pos, const float* scene_nor, int32_t n_scene,
                                const orc_object_t* objects, const orc_placement_t* placements, int32_t n_plc,
                                float radius, int prioritize_static, int32_t unlabelled_class_idx,
                                int8_t* labels, float* min_dists, int32_t* sorted_order,
                                int32_t* class_ids, int32_t* instance_ids );

/* One rspf__assign_temporary_labels pass (:738-778) over placements [start,end) of an
 * already-ordered arrangement. */
void orc_assign_labels( const float* scene_pos, const float* scene_nor, int32_t n_scene,
                        const orc_object_t* objects, const orc_placement_t* placements,
                        int32_t start, int32_t end, float radius, int8_t* labels, float* min_dists );

/* rspf_compute_neighborhood (lib/rs/rs_pointcloud_filters.cpp:674-722): K = max_nn unsorted self-search
 * over the cloud, one candidate edge per (point, neighbour) with weight
 * (1 - pow(d²/(4 r²), dist_exp)) * powf(clamp(n·m, 0, 1), angle_exp); edges are de-duplicated by the
 * int32 key max(i,j)*n + min(i,j) — first insertion wins, the key wraps for n > 46340 exactly as the
 * reference's int arithmetic does.  Output: edges sorted by (key, then insertion order); returns the
 * count.  idx1/idx2/weight have capacity n*max_nn. */
int64_t orc_compute_neighborhood( const orc_grid_t* grid, const float* pos, const float* nor, int32_t n,
                                  int32_t max_nn, float radius_sq, float dist_exp, float angle_exp,
                                  int32_t* idx1, int32_t* idx2, float* weight );
float orc_edge_cost( float nn_dist, float dot_nm, float radius_sq, float dist_exp, float angle_exp );   /* :706-708 */

/* Scene-coverage term of the arrangement optimiser.
 * isect_grid3d_init (lib/rs/intersect.h:59-75): bbox fattened by 0.3 on every side, resolution
 * ceilf(extent / voxel) + 1 per axis; cell (x,y,z) lives at y*x_res*z_res + z*x_res + x (:91-109). */
typedef struct orc_voxgrid { int32_t x_res, y_res, z_res, n_cells; float voxel_size; float origin[3]; } orc_voxgrid_t;
void    orc_voxgrid_init( orc_voxgrid_t* g, const float bbox_min[3], const float bbox_max[3], float voxel_size );
int64_t orc_voxgrid_cell( const orc_voxgrid_t* g, const float p[3] );          /* -1 = outside (:97-109) */
/* rsao_rasterize_scene_to_grid (apps/segment_transfer/arrangement_optimization.cpp:1064-1079); quality may be NULL (= all pass) */
void    orc_rasterize_scene( const orc_voxgrid_t* g, const float* pos, const float* quality, int64_t n, float quality_threshold, uint8_t* data );
/* rsao__rasterize_arrangement_to_grid (:1082-1106): data zeroed, then every point of every non-static placement, transformed by its pose */
void    orc_rasterize_arrangement( const orc_voxgrid_t* g, const float* const* obj_pos, const int64_t* obj_n, const float* poses,
                                   const int32_t* is_static, int32_t n_plc, uint8_t* data );
/* rsao__compute_scene_coverage_score (:344-373): agreeing / valid scene cells (0 if no valid cell) */
float   orc_coverage_score( const uint8_t* scene_data, const uint8_t* arr_data, int64_t n_cells, int32_t* agree, int32_t* valid );

/* Level builder: rs_pointcloud__compute_level_poisson (lib/rs/rs_pointcloud.h:984-1106) — indices of the
 * samples, increasing; the caller gathers the level's attribute arrays with them (:1090-1099).  For level L the
 * reference passes radius = voxel_size[L] and max_n_neigh = 1024*L/4 (256 for L = 0), :995-996. */
int32_t orc_level_poisson( const float* pts, int32_t n, float radius, int32_t max_n_neigh, int32_t* sample_idx );

/* The three normal gates, on a raw dot value (for threshold pinning). */
int orc_icp_gate( float dot, float max_angle );    /* lib/rs/icp.h:372-374 */
int orc_score_gate( float dot );                   /* pose_proposal.cpp:138-141 */
int orc_label_gate( float dot );                   /* rs_pointcloud_filters.cpp:769-770 */

#ifdef __cplusplus
}
#endif
#endif
