// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// rsd_alignment_scores (include/rescan_dropin.h) answered by the REFERENCE's own mgs_compute_object_alignment_score
// (apps/pose_proposal/pose_proposal.cpp:93-158), one call per pose, on the CPU.  oracle/Makefile links this file with the
// reference's unchanged apps/pose_proposal sources and shadow/apps/pose_proposal_batched.cpp into
// oracle/_ref/pose_proposal_batched_cpu, so that the HOST logic of the batched grid-search driver (pose enumeration, per-cell
// best rotation, thresholds, survivor bookkeeping) can be held against the reference app's proposal .bin without a GPU
// (tests/test_app_batched_cpu.py).  The product never links this file.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <cassert>

#include "msh/msh_std.h"
#include "msh/msh_vec_math.h"
#include "msh/msh_geometry.h"
#include "msh/msh_hash_grid.h"
#include "mg/hashtable.h"
#include "msh/msh_ply.h"
#include "rs_pointcloud.h"
#include "rs_distance_function.h"
#include "rs_database.h"
#include "pose_proposal.h"

extern "C" int rsd_alignment_scores( const msh_vec3_t* obj_pos, const msh_vec3_t* obj_nor, int32_t n_obj,
                                     const msh_vec3_t* scn_pos, const msh_vec3_t* scn_nor, int32_t n_scn,
                                     const msh_mat4_t* xforms, int32_t n_poses, float search_radius, int32_t max_n_neigh, float* scores )
{
  static rs_pointcloud_t scene; static const msh_vec3_t* scene_of = NULL;
  if( search_radius != 0.1f ) return -1;                       // search level 1 (pose_proposal.cpp:98)
  if( scene_of != scn_pos )                                    // the level-1 grid of rs_pointcloud_compute_search_grid (rs_pointcloud.h:849-863)
  {
    memset( &scene, 0, sizeof(scene) );
    scene.positions[1] = (msh_vec3_t*)scn_pos; scene.normals[1] = (msh_vec3_t*)scn_nor; scene.n_pts[1] = (size_t)n_scn;
    rs_pointcloud_compute_search_grid( &scene, 1 );
    scene_of = scn_pos;
  }
  rs_pointcloud_t object; memset( &object, 0, sizeof(object) );
  object.positions[0] = (msh_vec3_t*)obj_pos; object.normals[0] = (msh_vec3_t*)obj_nor; object.n_pts[0] = (size_t)n_obj;
  tmp_score_calc_storage_t storage = allocate_tmp_calc_storage( n_obj, n_scn, max_n_neigh );
  for( int32_t p = 0; p < n_poses; ++p ) scores[p] = mgs_compute_object_alignment_score( &object, &scene, 1, 0, xforms[p], &storage );
  free_tmp_calc_storage( &storage );
  return 0;
}
