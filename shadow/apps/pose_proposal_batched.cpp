// shadow/apps/pose_proposal_batched.cpp — the app-level caller of the BATCHED score entry point.
//
// apps/pose_proposal spends its time in mgs_propose_poses (apps/pose_proposal/pose_proposal.cpp:325-369): a grid search
// over (cell x, cell z, yaw) at level 4 and a re-scoring of the survivors at levels 3 and 2, one
// mgs_compute_object_alignment_score call — one small radius search — per pose (:213-244, :283-298; ~11 500 calls per object
// and room).  This TU is a replacement for that ONE function which gathers a level's poses of an object and scores them
// with ONE rsd_alignment_scores call (include/rescan_dropin.h; k_score: all poses of a level in one launch).
//
// How it is linked (oracle/Makefile: _ref/pose_proposal_hip3): the reference's app sources stay
// byte-unchanged; its pose_proposal.cpp is compiled with -Dmgs_propose_poses=mgs_propose_poses_reference, so that
// main.cpp's call binds to the definition below, and everything else of that TU — mgs_init_opts, the single-pose
// mgs_compute_object_alignment_score main.cpp calls after each icp_align, NMS, sorting — stays the reference's.
//
// Semantics kept (each cited where it is reproduced): the grid's float accumulation, the yaw loop's bound, per-cell best
// rotation by strict '>' from 0, thresholds per level, the -1 marking of rejected poses, the final |score| > 1e-6 copy.
// Written against the reference's headers (types, msh_rotate, msh_array, rsdb_is_object_static): nothing of
// pose_proposal.cpp is included or copied.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <cassert>
#include <vector>

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

extern "C" {
// include/rescan_dropin.h: rsd_vec3_t / rsd_mat4_t are layout-identical to msh_vec3_t / msh_mat4_t (tests/test_dropin.py)
int rsd_alignment_scores( const msh_vec3_t* obj_pos, const msh_vec3_t* obj_nor, int32_t n_obj,
                          const msh_vec3_t* scn_pos, const msh_vec3_t* scn_nor, int32_t n_scn,
                          const msh_mat4_t* xforms, int32_t n_poses, float search_radius, int32_t max_n_neigh, float* scores );
}

namespace {

const int32_t kSearchLevel = 1;        // pose_proposal.cpp:178,264: the scene is always searched at level 1 ...
const float   kSearchRadius = 0.1f;    // ... whose radius is search_radii[1] (:98,121)
const int32_t kMaxNeigh = 64;          // :179,265

// :160-168
float level_threshold( int32_t lvl )
{
  if( lvl == RSPC_N_LEVELS - 1 ) return 0.25f;
  if( lvl == RSPC_N_LEVELS - 2 ) return 0.35f;
  if( lvl == RSPC_N_LEVELS - 3 ) return 0.40f;
  return 0.50f;
}

// one launch for all poses of one object at one level
bool score_batch( rs_pointcloud_t* object, rs_pointcloud_t* scan, int32_t lvl, const std::vector<msh_mat4_t>& poses, std::vector<float>& scores )
{
  scores.assign( poses.size(), 0.0f );
  if( poses.empty() ) return true;
  const int rc = rsd_alignment_scores( object->positions[lvl], object->normals[lvl], (int32_t)object->n_pts[lvl],
                                       scan->positions[kSearchLevel], scan->normals[kSearchLevel], (int32_t)scan->n_pts[kSearchLevel],
                                       poses.data(), (int32_t)poses.size(), kSearchRadius, kMaxNeigh, scores.data() );
  if( rc != 0 ) { fprintf( stderr, "[rescan_hip] pose_proposal_batched: rsd_alignment_scores failed (%d)\n", rc ); return false; }
  return true;
}

// The level-4 search of one object (:197-248): every (cell, yaw) pose in the reference's own enumeration order, scored together.
void initial_proposals( rsdb_t* rsdb, rs_pointcloud_t* scan, int32_t lvl, int32_t obj_idx, const mgs_opts_t* opts,
                        msh_array(pose_proposal_t)* out )
{
  rs_object_t* object = &rsdb->objects[obj_idx];
  const msh_vec3_t origin = scan->bbox.min_p;
  const float spacing = opts->search_grid_spacing, y_angle_inc = opts->search_grid_angle_delta;
  const float length_x = scan->bbox.max_p.x - scan->bbox.min_p.x, length_z = scan->bbox.max_p.z - scan->bbox.min_p.z;
  const float height = 0.0f;
  std::vector<msh_mat4_t> poses; std::vector<int32_t> cell_first;
  // the loops of :213-219, as written there: float counters accumulated by +=, bounds compared in float (ox, oz) and against the
  // double literal MSH_TWO_PI (y_angle)
  for( float ox = -spacing; ox < length_x + spacing; ox += spacing )
    for( float oz = -spacing; oz < length_z + spacing; oz += spacing )
    {
      cell_first.push_back( (int32_t)poses.size() );
      for( float y_angle = 0.0f; y_angle < MSH_TWO_PI; y_angle += y_angle_inc )
      {
        msh_mat4_t xform = msh_rotate( msh_mat4_identity(), y_angle, msh_vec3( 0.0f, 1.0f, 0.0f ) );     // :221
        xform.col[3] = msh_vec4( origin.x + ox, height, origin.z + oz, 1.0f );                            // :222
        poses.push_back( xform );
      }
    }
  cell_first.push_back( (int32_t)poses.size() );
  std::vector<float> scores;
  if( !score_batch( object->shape, scan, lvl, poses, scores ) ) return;
  const float threshold = level_threshold( lvl );
  float max_score = -1e9;
  for( size_t c = 0; c + 1 < cell_first.size(); ++c )
  {
    float best = 0; int32_t best_k = -1;                                                                  // :217-218
    for( int32_t k = cell_first[c]; k < cell_first[c + 1]; ++k )
      if( scores[k] > best ) { best = scores[k]; best_k = k; if( best > max_score ) max_score = best; }   // :230-235
    if( best > threshold )                                                                                // :238 (best_k >= 0: threshold > 0)
    {
      pose_proposal_t proposal; proposal.xform = poses[best_k]; proposal.score = best;
      msh_array_push( *out, proposal );
    }
  }
  printf( "POSE_PROPOSAL:         --> Found %zu potential poses among %zu scored in one batch. (Max score: %f)\n",
          (size_t)msh_array_len( *out ), poses.size(), max_score );
}

// Levels 3 and 2 (:256-303): the surviving poses of one object, scored together; a pose at or below the threshold is marked -1
void verify_proposals( rsdb_t* rsdb, rs_pointcloud_t* scan, int32_t lvl, int32_t obj_idx, msh_array(pose_proposal_t) proposals )
{
  const int32_t n_poses = (int32_t)msh_array_len( proposals );
  std::vector<msh_mat4_t> poses; std::vector<int32_t> which;
  for( int32_t j = 0; j < n_poses; ++j )
    if( proposals[j].score > 0.0f ) { poses.push_back( proposals[j].xform ); which.push_back( j ); }      // :282
  std::vector<float> scores;
  if( !score_batch( rsdb->objects[obj_idx].shape, scan, lvl, poses, scores ) ) return;
  const float threshold = level_threshold( lvl );
  for( size_t k = 0; k < which.size(); ++k )
    proposals[which[k]].score = scores[k] > threshold ? scores[k] : -1.0f;                                // :289-290
}

} // namespace

// apps/pose_proposal/pose_proposal.cpp:325-369
void mgs_propose_poses( rsdb_t* rsdb, rs_pointcloud_t* input_scan, msh_array(msh_array(pose_proposal_t)) *proposed_poses,
                        const mgs_opts_t* opts, int verbose )
{
  const uint64_t gst = msh_time_now();
  const int32_t n_objects = (int32_t)msh_array_len( rsdb->objects );
  msh_array(msh_array(pose_proposal_t)) storage = NULL;
  for( int32_t i = 0; i < n_objects; ++i ) msh_array_push( storage, NULL );                               // :181-184
  for( int32_t lvl = RSPC_N_LEVELS - 1; lvl > RSPC_N_LEVELS - 4; lvl-- )                                   // :337
  {
    const uint64_t st = msh_time_now();
    msh_cprintf( verbose, "POSE PROPOSAL: Working on level: %d | Threshold: %6.4f (batched)\n", lvl, level_threshold( lvl ) );
    for( int32_t i = 0; i < n_objects; ++i )
    {
      if( rsdb_is_object_static( rsdb, i ) ) continue;                                                    // :200-203, :271
      if( lvl == RSPC_N_LEVELS - 1 ) initial_proposals( rsdb, input_scan, lvl, i, opts, &storage[i] );
      else if( msh_array_len( storage[i] ) ) verify_proposals( rsdb, input_scan, lvl, i, storage[i] );
    }
    msh_cprintf( verbose, "POSE PROPOSAL: Level %d processing time: %fs\n", lvl, msh_time_diff_sec( msh_time_now(), st ) );
  }
  for( int32_t i = 0; i < n_objects; ++i )                                                                // :347-359: copy valid poses
  {
    msh_array(pose_proposal_t) cur = NULL;
    for( size_t j = 0; j < msh_array_len( storage[i] ); ++j )
      if( fabsf( storage[i][j].score ) > 0.000001f ) msh_array_push( cur, storage[i][j] );
    msh_array_push( *proposed_poses, cur );
  }
  for( int32_t i = 0; i < n_objects; ++i ) if( storage[i] ) msh_array_free( storage[i] );
  if( storage ) msh_array_free( storage );
  msh_cprintf( verbose, "POSE PROPOSAL: Done in %fs\n", msh_time_diff_sec( msh_time_now(), gst ) );
}
