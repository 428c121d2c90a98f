/* rescan_dropin.h — the reference's own entry points for the hot path, re-implemented on top of
 * librescan_hip.so and exported under the reference's names by librescan_dropin.so, so that
 * apps/pose_proposal and apps/segment_transfer link against it unchanged (INTEGRATION.md shows
 * the two shadow headers a maintainer adds).
 *
 * Types are layout-identical to the reference's (sizes/offsets are checked by static asserts in
 * rs_dropin.cpp and by tests/test_dropin_cpu.py):
 *   msh_vec3_t        12 bytes {x,y,z}                       lib/msh/msh_vec_math.h:159-164
 *   msh_mat4_t        64 bytes, column-major float[16]       lib/msh/msh_vec_math.h:187-191
 *   msh_hash_grid_t   120 bytes, caller-allocated, zeroed    lib/msh/msh_hash_grid.h:248-269
 *   msh_hash_grid_search_desc_t                              lib/msh/msh_hash_grid.h:196-216
 */
#ifndef RESCAN_DROPIN_H
#define RESCAN_DROPIN_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rsd_vec3 { float x, y, z; } rsd_vec3_t;                 /* == msh_vec3_t */
typedef struct rsd_mat4 { float data[16]; } rsd_mat4_t;               /* == msh_mat4_t */

typedef struct rsd_hash_grid                                           /* == msh_hash_grid_t */
{
  size_t width, height, depth;
  double cell_size;
  float  min_pt[3];
  float  max_pt[3];
  void*  bin_table;        /* unused by the shim (NULL) */
  void*  data_buffer;      /* holds the shim's handle of this grid (host copy of the points + lazily built device cloud) */
  void*  offsets;          /* unused by the shim (NULL) */
  int32_t  _slab_size;
  double   _inv_cell_size;
  uint8_t  _pts_dim;
  uint16_t _num_threads;
  int32_t  _dont_use_omp;
  uint32_t max_n_pts_in_bin;
  size_t   _n_pts;
} rsd_hash_grid_t;

typedef struct rsd_search_desc                                         /* == msh_hash_grid_search_desc_t */
{
  float*   query_pts;
  size_t   n_query_pts;
  float*   distances_sq;
  int32_t* indices;
  size_t*  n_neighbors;
  float    radius;
  union { size_t k; size_t max_n_neigh; };
  int      sort;
} rsd_search_desc_t;

/* lib/msh/msh_hash_grid.h:218-230 */
void   msh_hash_grid_init_3d( rsd_hash_grid_t* hg, const float* pts, const int32_t n_pts, const float radius );
void   msh_hash_grid_init_2d( rsd_hash_grid_t* hg, const float* pts, const int32_t n_pts, const float radius );   /* :544-548: (x, y) pairs, kept as (x, y, 0) */
void   msh_hash_grid_term( rsd_hash_grid_t* hg );
size_t msh_hash_grid_radius_search( const rsd_hash_grid_t* hg, rsd_search_desc_t* search_desc );
/* :1294-1447 — the reference's shell-by-shell traversal (not an exact k-nearest search: it stops one shell of bins after k points are
 * stored), restated on the host on a grid of the reference's own geometry; no caller on the hot path.  Rows ascending in
 * (dist², index); ends where the reference would overrun its 128-bin stack array or never return (rs_dropin.cpp: KnnGrid). */
size_t msh_hash_grid_knn_search( const rsd_hash_grid_t* hg, rsd_search_desc_t* search_desc );

/* lib/rs/icp.h:83-115 */
float icp_align( rsd_vec3_t* pts1, rsd_vec3_t* nor1, int32_t n_pts1,
                 rsd_vec3_t* pts2, rsd_vec3_t* nor2, int32_t n_pts2,
                 rsd_mat4_t* T1, rsd_mat4_t T2, float max_dist, float max_angle, bool verbose );
float icp_estimate_rigid_xform_pt2pt( rsd_vec3_t* pts1, rsd_vec3_t* pts2, float* weights, int32_t n_pts, rsd_mat4_t* T1 );   /* icp.h:153-206: a stub in the reference (returns 1.0f) */
float icp_estimate_rigid_xform_pt2pl( rsd_vec3_t* pts1, rsd_vec3_t* pts2, rsd_vec3_t* nor2, float* weights,
                                      int32_t n_pts, rsd_mat4_t* T1 );
void  icp_find_corrs( rsd_vec3_t* pts1, rsd_vec3_t* nor1, int32_t n_pts1, rsd_hash_grid_t* idx1,
                      rsd_vec3_t* pts2, rsd_vec3_t* nor2, int32_t n_pts2, rsd_hash_grid_t* idx2,
                      rsd_mat4_t T1, rsd_mat4_t T2,
                      rsd_vec3_t** corr_pts1, rsd_vec3_t** corr_nor1, rsd_vec3_t** corr_pts2, rsd_vec3_t** corr_nor2,
                      float** weights, int32_t* n_corrs, float max_dist, float max_angle );

/* The two C++-linkage consumers take rs_pointcloud_t / rsdb_t, which the shim does not know; the
 * shadow copies of those functions call these flat entry points instead.
 *
 * mgs_compute_object_alignment_score (apps/pose_proposal/pose_proposal.cpp:93-158) for one pose:
 * obj_* = object->positions/normals[query_lvl], scn_* = scene->positions/normals[search_lvl]. */
float rsd_alignment_score( const rsd_vec3_t* obj_pos, const rsd_vec3_t* obj_nor, int32_t n_obj,
                           const rsd_vec3_t* scn_pos, const rsd_vec3_t* scn_nor, int32_t n_scn,
                           rsd_mat4_t xform, float search_radius, int32_t max_n_neigh );
/* the same for a batch of poses (the grid-search loops at pose_proposal.cpp:213-244, 283-298) */
int   rsd_alignment_scores( const rsd_vec3_t* obj_pos, const rsd_vec3_t* obj_nor, int32_t n_obj,
                            const rsd_vec3_t* scn_pos, const rsd_vec3_t* scn_nor, int32_t n_scn,
                            const rsd_mat4_t* xforms, int32_t n_poses, float search_radius, int32_t max_n_neigh,
                            float* scores );
/* rspf_arrangement_to_labels' search part (lib/rs/rs_pointcloud_filters.cpp:796-848): placements in
 * arrangement order with per-placement static flag and class index; writes int8 labels (1-based
 * index into the sorted arrangement, whose permutation is returned in sorted_order). */
int   rsd_arrangement_to_labels( const rsd_vec3_t* scn_pos, const rsd_vec3_t* scn_nor, int32_t n_scn,
                                 const rsd_vec3_t* const* obj_pos, const rsd_vec3_t* const* obj_nor, const int32_t* obj_n,
                                 const rsd_mat4_t* poses, const int32_t* is_static, const int32_t* class_idx, int32_t n_plc,
                                 float radius, bool prioritize_static, int8_t* labels, int32_t* sorted_order );

/* The whole of rspf_arrangement_to_labels (lib/rs/rs_pointcloud_filters.cpp:780-879) including its tail (:851-869): the
 * outputs are in_pc->class_ids[lvl] / instance_ids[lvl] themselves — class of the labelled placement's object, the
 * placement's uidx, (rsdb_get_class_idx( rsdb, "unlabelled" ), RSPF_MAX_INSTANCES) where no placement claimed the point. */
int   rsd_arrangement_to_ids( const rsd_vec3_t* scn_pos, const rsd_vec3_t* scn_nor, int32_t n_scn,
                              const rsd_vec3_t* const* obj_pos, const rsd_vec3_t* const* obj_nor, const int32_t* obj_n,
                              const rsd_mat4_t* poses, const int32_t* is_static, const int32_t* class_idx, const int32_t* uidx, int32_t n_plc,
                              float radius, bool prioritize_static, int32_t unlabelled_class_idx,
                              int32_t* class_ids, int32_t* instance_ids );

/* rspf_compute_neighborhood (lib/rs/rs_pointcloud_filters.cpp:674-722) on pc->positions/normals[lvl]:
 * fills the caller's edge arrays (capacity >= n*max_nn; the reference's hashtable would hold as many),
 * idx1/idx2/weight being the fields of its edge_t (:664-669).  Returns the edge count (< 0: error). */
int64_t rsd_compute_neighborhood( const rsd_vec3_t* pos, const rsd_vec3_t* nor, int32_t n,
                                  int32_t max_nn, float radius_sq, float dist_exp, float angle_exp,
                                  int32_t* idx1, int32_t* idx2, float* weight );

/* rs_pointcloud__compute_level_poisson (lib/rs/rs_pointcloud.h:984-1106) on pc->positions[0]: the indices of the
 * level's samples, increasing — the caller fills positions/normals/colors/... [level][i] = [0][sample_idx[i]] as the
 * reference does (:1090-1099).  voxel_size = pc->voxel_size[level]; the reference's cap on a search,
 * max_n_neigh = 1024 * level / (RSPC_N_LEVELS - 1) or 256 (:995-996), is derived from `level`.  sample_idx has
 * capacity n.  Returns the number of samples (< 0: error; RS_HIP_E_CAPACITY if a point has more than max_n_neigh
 * points within voxel_size, where a reference search would be truncated). */
int32_t rsd_level_poisson( const rsd_vec3_t* pos, int32_t n, float voxel_size, int32_t level, int32_t* sample_idx );

/* Scene-coverage term of the arrangement optimiser (apps/segment_transfer/arrangement_optimization.cpp:344-373).
 * rsd_coverage_create replaces isect_grid3d_init + rsao_rasterize_scene_to_grid for opts->scn_grd
 * (apps/segment_transfer/main.cpp:323-339); rsd_coverage_score replaces the body of
 * rsao__compute_scene_coverage_score: obj_* = rsdb->objects[arrangement[i].object_idx].shape level 2,
 * is_static[i] = rsdb_is_object_static(...).  Returns the score; < 0 on a device error. */
void* rsd_coverage_create( const rsd_vec3_t* bbox_min, const rsd_vec3_t* bbox_max, float voxel_size,
                           const rsd_vec3_t* scene_pos, const float* scene_quality, int32_t n_scene, float quality_threshold );
float rsd_coverage_score( void* coverage, const rsd_vec3_t* const* obj_pos, const int32_t* obj_n,
                          const rsd_mat4_t* poses, const int32_t* is_static, int32_t n_plc );
void  rsd_coverage_destroy( void* coverage );

/* ---- the on-disk formats either side of the path ---------------------------------
 * Pose-proposal blob (writer apps/pose_proposal/main.cpp:61-89, reader apps/segment_transfer/main.cpp:143-193):
 * int32 n_arrays; int32 count[n_arrays]; then per proposal 16 floats (column-major pose) + 1 float score,
 * arrays back to back, host byte order.  `records` holds sum(count) x 17 floats.  The reader allocates
 * *counts and *records with malloc (caller frees); like the reference's, it accepts a truncated file only
 * up to the last complete array and reports the failure (RSD_FORMAT_ERR). */
#define RSD_FORMAT_ERR (-20)
int rsd_pose_bin_write( const char* path, int32_t n_arrays, const int32_t* counts, const float* records );
int rsd_pose_bin_read( const char* path, int32_t* n_arrays, int32_t** counts, float** records );
/* One `pose` line of an .rsdb database (lib/rs/rs_database.h:379-401 parser, :597-606 writer): the 4x4 pose
 * is printed row by row with "%f", i.e. rounded to 6 decimals — poses lose precision between pose_proposal and
 * segment_transfer, and that is reference behaviour.  format returns the length written (no newline). */
int rsd_rsdb_format_pose_line( char* out, size_t capacity, int32_t uidx, int32_t arrangement_idx, int32_t object_idx,
                               float score, const rsd_mat4_t* pose );
int rsd_rsdb_parse_pose_line( const char* line, int32_t* uidx, int32_t* arrangement_idx, int32_t* object_idx,
                              float* score, rsd_mat4_t* pose );

/* The shim caches device uploads of host arrays by (pointers, count, content hash) — SURVEY.md §8b's "(pointer, n,
 * generation)" key.  Arrays of up to 4 MB each (RS_DROPIN_FULL_HASH_BELOW, bytes) are hashed whole on every call: an in-place
 * edit of any byte is seen.  Larger arrays are keyed by a ~1 KB sample and their full hash is re-checked on the first and then
 * every 16th hit: a caller that edits a LARGE array in place (same pointer, same count) MUST call rsd_cache_invalidate( ptr )
 * before the next call that passes it — or set RS_DROPIN_FULL_HASH=1 (environment: whole arrays hashed on every call, 40 us
 * per MB).  msh_hash_grid_term() invalidates what was built from its grid's array (the reference terminates a level's grid
 * right before it frees or rebuilds the level, lib/rs/rs_pointcloud.h:879-901); rsd_cache_clear drops everything.  A cached
 * cloud that a call is using stays alive until that call returns, whatever is evicted or invalidated meanwhile. */
void  rsd_cache_clear( void );
void  rsd_cache_invalidate( const void* host_array );

/* msh_hash_grid_radius_search calls that hit a HIP error on an initialised grid and were therefore answered from the grid's
 * host copy (the reference's search always fills its rows and counts, lib/msh/msh_hash_grid.h:1090-1259, and its callers
 * walk them; the first such call complains on stderr).  0 in a healthy run: tests assert it. */
unsigned long long rsd_device_failures( void );

#ifdef __cplusplus
}
#endif
#endif
