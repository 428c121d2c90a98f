// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// ref_filters_driver: C-ABI entry points around the REAL reference's label transfer and neighbourhood graph
// (SURVEY.md §8 rows a9, a10, f1), for parity pinning.  oracle/Makefile links this file with
// a scratch file which it GENERATES for the duration of the compile (never kept, never committed) from
// /root/reference/lib/rs/rs_pointcloud_filters.cpp: lines 1-14 and 16-879, i.e. the file's own text minus the one
// `#include "GCoptimization.h"` (line 15, un-vendored gco-v3.0) and minus rspf_smooth_labels (:881-, the only user
// of gco, :955-971).  No reference line is edited and no stand-in header exists: <cassert>/<cstring>, which that TU
// used to receive through the gco header, are force-included on the command line.
//
// Nothing here restates the loops under test: rspf_arrangement_to_labels, rspf__assign_temporary_labels and
// rspf_compute_neighborhood below are the reference's compiled text, run on a database assembled with the
// reference's own rsdb_init / rsdb_add_class / rsdb_add_object and level grids from
// rs_pointcloud_compute_search_grid (rs_pointcloud.h:849-863).
//
// The implementation macros are the ones apps/segment_transfer/main.cpp:6-17 defines.
#define MSH_STD_IMPLEMENTATION
#define MSH_ARGPARSE_IMPLEMENTATION
#define MSH_VEC_MATH_IMPLEMENTATION
#define MSH_GEOMETRY_IMPLEMENTATION
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
#include <algorithm>

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
#include "rs_pointcloud_filters.h"

// the msh_array instantiations the filters TU uses (apps/segment_transfer/main.cpp:51-55 instantiates them for the app
// the same way; any TU that links that file needs them)
template int32_t* msh_array__grow<int32_t>(int32_t* arr, unsigned long long new_len, unsigned long long elem_size );
template rs_object_placement* msh_array__grow<rs_object_placement>(rs_object_placement* arr, unsigned long long new_len, unsigned long long elem_size );
template rspf_plane_model* msh_array__grow<rspf_plane_model>(rspf_plane_model* arr, unsigned long long new_len, unsigned long long elem_size );
template rspf_edge* msh_array__grow<rspf_edge>(rspf_edge* arr, unsigned long long new_len, unsigned long long elem_size );
template vec3f* msh_array__grow<vec3f>(vec3f* arr, unsigned long long new_len, unsigned long long elem_size );
template rs_pointcloud_t** msh_array__grow<rs_pointcloud_t*>(rs_pointcloud_t** arr, unsigned long long new_len, unsigned long long elem_size );

// defined (external linkage) in the generated TU at rs_pointcloud_filters.cpp:738-778
void rspf__assign_temporary_labels( rsdb_t* rsdb, rs_pointcloud_t* pc, msh_array(rs_obj_plcmnt_t) arrangement,
                                    msh_hash_grid_search_desc_t* search_opts, int8_t* labels, float* min_dists,
                                    size_t start, size_t end, int32_t lvl );

typedef struct ref_filters
{
  rsdb_t* rsdb;
  msh_array(rs_pointcloud_t*) shapes;
} ref_filters_t;

static rs_pointcloud_t* level1_cloud( float* pos, float* nor, int64_t n, int with_grid )
{
  rs_pointcloud_t* pc = (rs_pointcloud_t*)calloc( 1, sizeof(rs_pointcloud_t) );
  pc->positions[1] = (msh_vec3_t*)pos; pc->normals[1] = (msh_vec3_t*)nor; pc->n_pts[1] = (size_t)n;
  if( with_grid ) rs_pointcloud_compute_search_grid( pc, 1 );
  return pc;
}

static void level1_cloud_free( rs_pointcloud_t* pc )
{
  if( pc->search_grids[1] ) { msh_hash_grid_term( pc->search_grids[1] ); free( pc->search_grids[1] ); }
  free( pc->class_ids[1] ); free( pc->instance_ids[1] );
  free( pc );
}

static msh_array(rs_obj_plcmnt_t) make_arrangement( ref_filters_t* h, const int32_t* object_idx, const int32_t* uidx,
                                                    const float* poses, int32_t n_plc )
{
  msh_array(rs_obj_plcmnt_t) arr = 0;
  for( int32_t i = 0; i < n_plc; ++i )
  {
    rs_obj_plcmnt_t p; memset( &p, 0, sizeof(p) );
    p.object_idx = object_idx[i]; p.uidx = uidx[i]; p.arrangement_idx = 0; p.pose_idx = i;
    memcpy( p.pose.data, poses + 16 * i, 64 );
    msh_array_push( arr, p );
  }
  return arr;
}

extern "C" {

// One class table per process: rsdb_is_class_static caches the class indices in function statics (rs_database.h:260-271).
void* ref_filters_create( const char** class_names, const int32_t* class_ids, int32_t n_classes )
{
  ref_filters_t* h = (ref_filters_t*)calloc( 1, sizeof(ref_filters_t) );
  h->rsdb = rsdb_init();
  for( int32_t i = 0; i < n_classes; ++i ) rsdb_add_class( h->rsdb, strdup( class_names[i] ), class_ids[i] );
  return h;
}

// level-1 cloud of one database object (arrays stay the caller's); returns its object index
int32_t ref_filters_add_object( void* hp, float* pos, float* nor, int64_t n, int32_t class_idx, int32_t uidx )
{
  ref_filters_t* h = (ref_filters_t*)hp;
  rs_pointcloud_t* pc = level1_cloud( pos, nor, n, 1 );
  msh_array_push( h->shapes, pc );
  rs_object_t o = rsdb_object_init();
  o.uidx = uidx; o.class_idx = class_idx; o.shape = pc;
  return rsdb_add_object( h->rsdb, &o );
}

int32_t ref_filters_is_object_static( void* hp, int32_t object_idx )
{
  return rsdb_is_object_static( ((ref_filters_t*)hp)->rsdb, object_idx );
}

int32_t ref_filters_class_idx( void* hp, const char* name ) { return rsdb_get_class_idx( ((ref_filters_t*)hp)->rsdb, name ); }

// rspf_arrangement_to_labels (rs_pointcloud_filters.cpp:780-879), whole: class / instance ids of the scene's level 1
void ref_filters_arrangement_to_labels( void* hp, float* scene_pos, float* scene_nor, int64_t n_scene,
                                        const int32_t* object_idx, const int32_t* uidx, const float* poses, int32_t n_plc,
                                        float radius, int32_t prioritize_static, int32_t* class_ids, int32_t* instance_ids )
{
  ref_filters_t* h = (ref_filters_t*)hp;
  rs_pointcloud_t* scn = level1_cloud( scene_pos, scene_nor, n_scene, 0 );
  scn->class_ids[1]    = (int32_t*)malloc( (size_t)( n_scene > 0 ? n_scene : 1 ) * 4 );
  scn->instance_ids[1] = (int32_t*)malloc( (size_t)( n_scene > 0 ? n_scene : 1 ) * 4 );
  msh_array(rs_obj_plcmnt_t) arr = make_arrangement( h, object_idx, uidx, poses, n_plc );
  rspf_arrangement_to_labels( h->rsdb, scn, arr, radius, prioritize_static != 0 );
  memcpy( class_ids, scn->class_ids[1], (size_t)n_scene * 4 );
  memcpy( instance_ids, scn->instance_ids[1], (size_t)n_scene * 4 );
  msh_array_free( arr );
  level1_cloud_free( scn );
}

// rspf__assign_temporary_labels (:738-778) over placements [start, end) of an arrangement given IN THE ORDER TO VISIT;
// labels / min_dists are the caller's running state, exactly as rspf_arrangement_to_labels hands them over (:837-848).
void ref_filters_assign_labels( void* hp, float* scene_pos, float* scene_nor, int64_t n_scene,
                                const int32_t* object_idx, const int32_t* uidx, const float* poses, int32_t n_plc,
                                int32_t start, int32_t end, float radius, int8_t* labels, float* min_dists )
{
  ref_filters_t* h = (ref_filters_t*)hp;
  rs_pointcloud_t* scn = level1_cloud( scene_pos, scene_nor, n_scene, 0 );
  msh_array(rs_obj_plcmnt_t) arr = make_arrangement( h, object_idx, uidx, poses, n_plc );
  size_t cap = (size_t)( n_scene > 0 ? n_scene : 1 );
  msh_vec3_t* pts = (msh_vec3_t*)malloc( cap * sizeof(msh_vec3_t) );           // the scratch of :805-818
  int32_t* indices = (int32_t*)malloc( cap * 4 ); float* dists_sq = (float*)malloc( cap * 4 );
  size_t* n_neighbors = (size_t*)malloc( cap * sizeof(size_t) );
  msh_hash_grid_search_desc_t so; memset( &so, 0, sizeof(so) );
  so.query_pts = &pts[0].x; so.n_query_pts = (size_t)n_scene; so.distances_sq = dists_sq; so.indices = indices;
  so.n_neighbors = n_neighbors; so.max_n_neigh = 1; so.radius = radius;
  rspf__assign_temporary_labels( h->rsdb, scn, arr, &so, labels, min_dists, (size_t)start, (size_t)end, 1 );
  free( pts ); free( indices ); free( dists_sq ); free( n_neighbors );
  msh_array_free( arr );
  level1_cloud_free( scn );
}

// rspf_compute_neighborhood (:674-722) on a bare level-1 cloud; edges in the order the reference returns them
// (hashtable_items).  Returns the edge count; out arrays must hold n * max_nn entries.
int64_t ref_filters_compute_neighborhood( float* pos, float* nor, int64_t n, int32_t max_nn, float radius_sq,
                                          float dist_exp, float angle_exp, int32_t* idx1, int32_t* idx2, float* weight )
{
  rs_pointcloud_t* pc = level1_cloud( pos, nor, n, 1 );
  msh_array(rspf_edge_t) edges = rspf_compute_neighborhood( pc, 1, max_nn, radius_sq, dist_exp, angle_exp );
  int64_t m = (int64_t)msh_array_len( edges );
  for( int64_t k = 0; k < m; ++k ) { idx1[k] = edges[k].idx1; idx2[k] = edges[k].idx2; weight[k] = edges[k].weight; }
  msh_array_free( edges );
  level1_cloud_free( pc );
  return m;
}

void ref_filters_destroy( void* hp )
{
  ref_filters_t* h = (ref_filters_t*)hp;
  for( size_t i = 0; i < msh_array_len( h->shapes ); ++i ) level1_cloud_free( h->shapes[i] );
  msh_array_free( h->shapes );
  free( h );       // the rsdb's tables are left to process exit (rsdb_free expects file-loaded clouds)
}

}
