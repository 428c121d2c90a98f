// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// ref_ao_driver: C-ABI entry points around the REAL reference's scene-coverage term of the
// arrangement optimiser, for parity pinning.  oracle/Makefile compiles this
// file together with /root/reference/apps/segment_transfer/arrangement_optimization.cpp (in place,
// unmodified) into oracle/_ref/libref_ao.so; nothing here restates reference logic: the grids come
// from isect_grid3d_init, the scene grid from rsao_rasterize_scene_to_grid, the score from
// rsao__compute_scene_coverage_score (arrangement_optimization.cpp:344-373,1064-1106), on a
// database assembled with the reference's own rsdb_init / rsdb_add_class / rsdb_add_object.
//
// The implementation macros below are the ones apps/segment_transfer/main.cpp:6-17 defines, minus
// the two that arrangement_optimization.cpp:1-2 defines itself.
#define MSH_STD_IMPLEMENTATION
#define MSH_ARGPARSE_IMPLEMENTATION
#define MSH_VEC_MATH_IMPLEMENTATION
#define MSH_PLY_IMPLEMENTATION
#define MSH_HASH_GRID_IMPLEMENTATION
#define RS_DATABASE_IMPLEMENTATION
#define RS_POINTCLOUD_IMPLEMENTATION
#define RS_DISTANCE_FUNCTION_IMPLEMENTATION
#define FILEPATH_HELPERS_IMPLEMENTATION
#define HASHTABLE_IMPLEMENTATION

#include <cassert>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <cstdarg>
#include <cstddef>
#include <cstdbool>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <cctype>

#include "msh/msh_std.h"
#include "msh/msh_argparse.h"
#include "msh/msh_vec_math.h"
#include "msh/msh_geometry.h"
#include "msh/msh_ply.h"
#include "msh/msh_hash_grid.h"
#include "mg/hashtable.h"
#include "filepath_helpers.h"
#include "rs_pointcloud.h"
#include "rs_distance_function.h"
#include "rs_database.h"
#include "intersect.h"
#include "arrangement_optimization.h"

// arrangement_optimization.cpp uses these msh_array instantiations (apps/segment_transfer/main.cpp instantiates
// them for the app the same way)
template int* msh_array__grow<int>(int* arr, unsigned long long new_len, unsigned long long elem_size );
template rs_object_placement* msh_array__grow<rs_object_placement>(rs_object_placement* arr, unsigned long long new_len, unsigned long long elem_size );

// defined in arrangement_optimization.cpp, declared there at :37 and used by the SA loop
void  rsao__rasterize_arrangement_to_grid( rsdb_t* rsdb, msh_array(rs_obj_plcmnt_t) arrangement, isect_grid3d_t* grd );
float rsao__compute_scene_coverage_score( rsdb_t* rsdb, msh_array(rs_obj_plcmnt_t) arrangement, rsao_opts_t* opts, int32_t verbose );
void  rsao_rasterize_scene_to_grid( rs_scene_t* scn, isect_grid3d_t* grd, float quality_threshold );

typedef struct ref_ao
{
  rsdb_t* rsdb;
  rs_pointcloud_t scene_pc;
  rs_scene_t scene;
  isect_grid3d_t scn_grd, arr_grd;
  rsao_opts_t opts;
  msh_array(rs_pointcloud_t*) shapes;
} ref_ao_t;

extern "C" {

// class_names/class_ids: the class table (must contain "wall", "floor", "unlabelled" for the static rule,
// rs_database.h:257-288).  Scene level-2 points + qualities; bbox = the scene cloud's (rs_pointcloud.h:840-847).
void* ref_ao_create( const char** class_names, const int32_t* class_ids, int32_t n_classes,
                     float* scene_pos, float* scene_quality, int64_t n_scene,
                     const float* bbox_min, const float* bbox_max, float voxel_size, float quality_threshold )
{
  ref_ao_t* h = (ref_ao_t*)calloc( 1, sizeof(ref_ao_t) );
  h->rsdb = rsdb_init();
  for( int32_t i = 0; i < n_classes; ++i ) rsdb_add_class( h->rsdb, strdup( class_names[i] ), class_ids[i] );
  h->scene_pc.positions[2] = (msh_vec3_t*)scene_pos; h->scene_pc.qualities[2] = scene_quality; h->scene_pc.n_pts[2] = (size_t)n_scene;
  h->scene_pc.bbox.min_p = msh_vec3( bbox_min[0], bbox_min[1], bbox_min[2] );
  h->scene_pc.bbox.max_p = msh_vec3( bbox_max[0], bbox_max[1], bbox_max[2] );
  h->scene.shape = &h->scene_pc;
  isect_grid3d_init( &h->scn_grd, &h->scene_pc.bbox, voxel_size );        // apps/segment_transfer/main.cpp:323-325
  isect_grid3d_init( &h->arr_grd, &h->scene_pc.bbox, voxel_size );
  rsao_rasterize_scene_to_grid( &h->scene, &h->scn_grd, quality_threshold );   // main.cpp:339
  rsao_init_opts( &h->opts );
  h->opts.scn_grd = &h->scn_grd; h->opts.arrangement_grd = &h->arr_grd;
  return h;
}

void ref_ao_info( void* hp, int32_t res[3], float origin[3], int64_t* n_cells )
{
  ref_ao_t* h = (ref_ao_t*)hp;
  res[0] = h->scn_grd.x_res; res[1] = h->scn_grd.y_res; res[2] = h->scn_grd.z_res;
  origin[0] = h->scn_grd.bbox.min_p.x; origin[1] = h->scn_grd.bbox.min_p.y; origin[2] = h->scn_grd.bbox.min_p.z;
  *n_cells = h->scn_grd.n_cells;
}

// level-2 cloud of one database object; returns its object index
int32_t ref_ao_add_object( void* hp, float* pos, int64_t n, int32_t class_idx, int32_t uidx )
{
  ref_ao_t* h = (ref_ao_t*)hp;
  rs_pointcloud_t* pc = (rs_pointcloud_t*)calloc( 1, sizeof(rs_pointcloud_t) );
  pc->positions[2] = (msh_vec3_t*)pos; pc->n_pts[2] = (size_t)n;
  msh_array_push( h->shapes, pc );
  rs_object_t o = rsdb_object_init();
  o.uidx = uidx; o.class_idx = class_idx; o.shape = pc;
  return rsdb_add_object( h->rsdb, &o );
}

// rsao__compute_scene_coverage_score for one arrangement (object indices + poses, column-major 4x4)
float ref_ao_coverage( void* hp, const int32_t* object_idx, const float* poses, int32_t n_plc )
{
  ref_ao_t* h = (ref_ao_t*)hp;
  msh_array(rs_obj_plcmnt_t) arr = 0;
  for( int32_t i = 0; i < n_plc; ++i )
  {
    rs_obj_plcmnt_t p; memset( &p, 0, sizeof(p) );
    p.object_idx = object_idx[i]; p.uidx = h->rsdb->objects[object_idx[i]].uidx;
    memcpy( p.pose.data, poses + 16 * i, 64 );
    msh_array_push( arr, p );
  }
  float s = rsao__compute_scene_coverage_score( h->rsdb, arr, &h->opts, 0 );
  msh_array_free( arr );
  return s;
}

// rs_pointcloud__compute_level_poisson (lib/rs/rs_pointcloud.h:984-1106) of the REAL reference on a bare
// cloud: level 0 holds the given points, its class ids carry the point indices, so the level's class ids come back as
// the sample indices.  voxel_size > 0 overrides the level's resolution (rs_pointcloud.h:148).
int32_t ref_level_poisson( const float* pts, int32_t n, int32_t level, float voxel_size, int32_t* out_idx )
{
  rs_pointcloud_t* pc = rs_pointcloud_init( 1 );
  rs_pointcloud__allocate_level( pc, 0, n );
  memcpy( pc->positions[0], pts, (size_t)n * sizeof(msh_vec3_t) );
  memset( pc->normals[0], 0, (size_t)n * sizeof(msh_vec3_t) ); memset( pc->colors[0], 0, (size_t)n * sizeof(msh_vec3_t) );
  memset( pc->radii[0], 0, (size_t)n * 4 ); memset( pc->qualities[0], 0, (size_t)n * 4 ); memset( pc->instance_ids[0], 0, (size_t)n * 4 );
  for( int32_t i = 0; i < n; ++i ) pc->class_ids[0][i] = i;
  if( voxel_size > 0.0f ) pc->voxel_size[level] = voxel_size;
  rs_pointcloud__compute_level_poisson( pc, level );
  const int32_t n_out = (int32_t)pc->n_pts[level];
  memcpy( out_idx, pc->class_ids[level], (size_t)n_out * 4 );
  rs_pointcloud__free_level( pc, level );
  rs_pointcloud__free_level( pc, 0 );
  free( pc );
  return n_out;
}

const uint8_t* ref_ao_scene_grid( void* hp ) { return ((ref_ao_t*)hp)->scn_grd.data; }
const uint8_t* ref_ao_arrangement_grid( void* hp ) { return ((ref_ao_t*)hp)->arr_grd.data; }   // as left by the last ref_ao_coverage

void ref_ao_destroy( void* hp )
{
  ref_ao_t* h = (ref_ao_t*)hp;
  isect_grid3d_term( &h->scn_grd ); isect_grid3d_term( &h->arr_grd );
  for( size_t i = 0; i < msh_array_len( h->shapes ); ++i ) free( h->shapes[i] );
  msh_array_free( h->shapes );
  free( h );       // the rsdb's tables are left to process exit (rsdb_free expects file-loaded clouds)
}

}
