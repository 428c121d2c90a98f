// librescan_dropin.so — the reference's own symbol names for the hot path, forwarding to the
// C ABI of librescan_hip.so (include/rescan_hip.h).  See include/rescan_dropin.h.
//
// Error behaviour follows the reference: no error codes on this path.  A failing
// HIP call prints to stderr and leaves the caller's data untouched (icp_align returns the 1e6 it
// would have returned had it never iterated, lib/rs/icp.h:441-442).  There is no CPU fallback.

#include "../../include/rescan_dropin.h"
#include "../../include/rescan_hip.h"

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <mutex>
#include <vector>

static_assert( sizeof(rsd_vec3_t) == 12, "msh_vec3_t is 12 bytes" );
static_assert( sizeof(rsd_mat4_t) == 64, "msh_mat4_t is 64 bytes" );
static_assert( sizeof(rsd_hash_grid_t) == 120, "msh_hash_grid_t is 120 bytes (lib/msh/msh_hash_grid.h:248-269)" );
static_assert( offsetof(rsd_hash_grid_t, data_buffer) == 64, "data_buffer offset" );
static_assert( offsetof(rsd_hash_grid_t, _n_pts) == 112, "_n_pts offset" );
static_assert( sizeof(rsd_search_desc_t) == 64, "msh_hash_grid_search_desc_t is 64 bytes" );

namespace {

void complain( const char* where )
{
  fprintf( stderr, "[rescan_hip] %s: %s\n", where, rs_hip_last_error() );
}

// ---- device-cloud cache ------------------------------------------------------------------
// The reference hands the same host arrays to the hot path over and over (the scene level for
// every proposal, apps/pose_proposal/main.cpp:190-202).  Uploads are cached by
// (pointers, count, cell size, content hash) and evicted least-recently-used.
struct Entry
{
  const void* pos; const void* nor; int32_t n; float cell; uint64_t hash;
  rs_hip_cloud_t* cloud;
};
std::list<Entry> g_cache;
std::mutex g_cache_mutex;
const size_t kMaxEntries = 64;

uint64_t content_hash( const void* a, const void* b, size_t bytes )
{
  // four independent multiply-xor lanes over 32-byte blocks (the dependent chain of a single lane caps at
  // ~8 GB/s; the arrays are hashed on every call, so this sits next to a 4 ms index build)
  uint64_t h = 0x9e3779b97f4a7c15ull ^ bytes;
  for( const void* src : { a, b } )
  {
    if( !src ) { h = ( h ^ 0x51ull ) * 0xff51afd7ed558ccdull; continue; }
    const unsigned char* p = (const unsigned char*)src;
    uint64_t l0 = h, l1 = h ^ 0x165667b19e3779f9ull, l2 = h ^ 0x27d4eb2f165667c5ull, l3 = h ^ 0x85ebca77c2b2ae63ull;
    size_t i = 0;
    for( ; i + 32 <= bytes; i += 32 )
    {
      uint64_t w[4]; std::memcpy( w, p + i, 32 );
      l0 = ( l0 ^ w[0] ) * 0xff51afd7ed558ccdull; l0 ^= l0 >> 29;
      l1 = ( l1 ^ w[1] ) * 0xc4ceb9fe1a85ec53ull; l1 ^= l1 >> 31;
      l2 = ( l2 ^ w[2] ) * 0x9fb21c651e98df25ull; l2 ^= l2 >> 30;
      l3 = ( l3 ^ w[3] ) * 0xd6e8feb86659fd93ull; l3 ^= l3 >> 32;
    }
    h = ( ( l0 * 31 + l1 ) * 31 + l2 ) * 31 + l3;
    for( ; i + 8 <= bytes; i += 8 ) { uint64_t w; std::memcpy( &w, p + i, 8 ); h = ( h ^ w ) * 0xff51afd7ed558ccdull; h ^= h >> 32; }
    for( ; i < bytes; ++i ) { h = ( h ^ p[i] ) * 0x100000001b3ull; }
  }
  return h;
}

rs_hip_cloud_t* cached_cloud( const rsd_vec3_t* pos, const rsd_vec3_t* nor, int32_t n, float cell )
{
  const uint64_t h = content_hash( pos, nor, (size_t)( n > 0 ? n : 0 ) * 12 );
  std::lock_guard<std::mutex> lock( g_cache_mutex );
  for( auto it = g_cache.begin(); it != g_cache.end(); ++it )
    if( it->pos == pos && it->nor == nor && it->n == n && it->cell == cell && it->hash == h )
    {
      g_cache.splice( g_cache.begin(), g_cache, it );
      return g_cache.front().cloud;
    }
  rs_hip_cloud_t* c = rs_hip_cloud_create( (const float*)pos, (const float*)nor, n, cell );
  if( !c ) { complain( "cloud upload" ); return nullptr; }
  g_cache.push_front( Entry{ pos, nor, n, cell, h, c } );
  while( g_cache.size() > kMaxEntries ) { rs_hip_cloud_destroy( g_cache.back().cloud ); g_cache.pop_back(); }
  return c;
}

} // namespace

extern "C" {

void rsd_cache_clear( void )
{
  std::lock_guard<std::mutex> lock( g_cache_mutex );
  for( auto& e : g_cache ) rs_hip_cloud_destroy( e.cloud );
  g_cache.clear();
}

// ---- msh_hash_grid ------------------------------------------------------------------------

void msh_hash_grid_init_3d( rsd_hash_grid_t* hg, const float* pts, const int32_t n_pts, const float radius )
{
  // cell = 2*radius like the reference (msh_hash_grid.h:443); a non-positive radius asks the
  // reference for an extent-derived cell (:444) — any positive cell gives the same results here.
  float cell = radius > 0.0f ? 2.0f * radius : 0.1f;
  rs_hip_cloud_t* c = rs_hip_cloud_create( pts, nullptr, n_pts, cell );
  if( !c ) complain( "msh_hash_grid_init_3d" );
  hg->data_buffer = c;
  hg->bin_table = nullptr; hg->offsets = nullptr;
  hg->cell_size = cell; hg->_inv_cell_size = 1.0 / cell;
  hg->_pts_dim = 3; hg->_num_threads = 1; hg->_n_pts = (size_t)( n_pts > 0 ? n_pts : 0 );
}

void msh_hash_grid_term( rsd_hash_grid_t* hg )
{
  if( hg->data_buffer ) rs_hip_cloud_destroy( (rs_hip_cloud_t*)hg->data_buffer );
  std::memset( hg, 0, sizeof(*hg) );                     // the reference zeroes what it owns (:558-573)
}

size_t msh_hash_grid_radius_search( const rsd_hash_grid_t* hg, rsd_search_desc_t* d )
{
  if( !hg || !hg->data_buffer || !d ) return 0;
  uint64_t total = 0;
  int rc = rs_hip_radius_search( (const rs_hip_cloud_t*)hg->data_buffer, d->query_pts, (int64_t)d->n_query_pts, d->radius,
                                 (int32_t)d->max_n_neigh, d->distances_sq, d->indices, d->n_neighbors, &total );
  if( rc ) { complain( "msh_hash_grid_radius_search" ); return 0; }
  return (size_t)total;
}

// ---- icp ----------------------------------------------------------------------------------

float icp_align( rsd_vec3_t* pts1, rsd_vec3_t* nor1, int32_t n_pts1, rsd_vec3_t* pts2, rsd_vec3_t* nor2, int32_t n_pts2,
                 rsd_mat4_t* T1, rsd_mat4_t T2, float max_dist, float max_angle, bool verbose )
{
  // the reference builds its grids with radius = max_dist (icp.h:436-437); the cell size does not
  // change results here, so the clouds use the density-derived cell (fastest, and shared with the
  // score / label entry points through the cache)
  rs_hip_cloud_t* src = cached_cloud( pts1, nor1, n_pts1, -1.0f );
  rs_hip_cloud_t* tgt = cached_cloud( pts2, nor2, n_pts2, -1.0f );
  float err = 1e6f; int32_t iters = 0;
  if( !src || !tgt ) return err;
  rsd_mat4_t t = *T1;
  int rc = rs_hip_icp_align( src, tgt, t.data, T2.data, max_dist, max_angle, 100, 0, &err, &iters );
  if( rc ) { complain( "icp_align" ); return 1e6f; }
  *T1 = t;
  if( verbose ) printf( " ICP: %d iterations on the GPU, final error %7.5f\n", iters, err );
  return err;
}

float icp_estimate_rigid_xform_pt2pl( rsd_vec3_t* pts1, rsd_vec3_t* pts2, rsd_vec3_t* nor2, float* weights,
                                      int32_t n_pts, rsd_mat4_t* T1 )
{
  float err = 0.0f;
  if( rs_hip_icp_estimate_pt2pl( (const float*)pts1, (const float*)pts2, (const float*)nor2, weights, n_pts, T1->data, &err ) )
    complain( "icp_estimate_rigid_xform_pt2pl" );
  return err;
}

void icp_find_corrs( rsd_vec3_t* pts1, rsd_vec3_t* nor1, int32_t n_pts1, rsd_hash_grid_t* idx1,
                     rsd_vec3_t* pts2, rsd_vec3_t* nor2, int32_t n_pts2, rsd_hash_grid_t* idx2,
                     rsd_mat4_t T1, rsd_mat4_t T2,
                     rsd_vec3_t** corr_pts1, rsd_vec3_t** corr_nor1, rsd_vec3_t** corr_pts2, rsd_vec3_t** corr_nor2,
                     float** weights, int32_t* n_corrs, float max_dist, float max_angle )
{
  (void)idx1;
  // Output arrays come from libc malloc because reference code frees them (icp.h:317-327,
  // including its habit of never freeing corr_nor1).
  if( *corr_pts1 ) { free( *corr_pts1 ); *corr_pts1 = NULL; }
  if( *corr_pts2 ) { free( *corr_pts2 ); *corr_pts2 = NULL; }
  if( *corr_nor2 ) { free( *corr_nor2 ); *corr_nor2 = NULL; }
  if( *weights )   { free( *weights );   *weights = NULL; }
  const size_t cap = (size_t)( n_pts1 > 0 ? n_pts1 : 1 );
  *corr_pts1 = (rsd_vec3_t*)malloc( cap * sizeof(rsd_vec3_t) );
  *corr_pts2 = (rsd_vec3_t*)malloc( cap * sizeof(rsd_vec3_t) );
  *corr_nor1 = (rsd_vec3_t*)malloc( cap * sizeof(rsd_vec3_t) );
  *corr_nor2 = (rsd_vec3_t*)malloc( cap * sizeof(rsd_vec3_t) );
  *weights   = (float*)malloc( cap * sizeof(float) );
  *n_corrs = 0;
  (void)idx2;
  rs_hip_cloud_t* src = cached_cloud( pts1, nor1, n_pts1, -1.0f );
  rs_hip_cloud_t* tgt = cached_cloud( pts2, nor2, n_pts2, -1.0f );
  if( !src || !tgt ) return;
  if( rs_hip_icp_find_corrs( src, tgt, T1.data, T2.data, max_dist, max_angle, (float*)*corr_pts1, (float*)*corr_nor1,
                             (float*)*corr_pts2, (float*)*corr_nor2, *weights, n_corrs ) )
  { complain( "icp_find_corrs" ); *n_corrs = 0; }
}

// ---- the two C++-linkage consumers, flattened ----------------------------------------------

int rsd_alignment_scores( const rsd_vec3_t* obj_pos, const rsd_vec3_t* obj_nor, int32_t n_obj,
                          const rsd_vec3_t* scn_pos, const rsd_vec3_t* scn_nor, int32_t n_scn,
                          const rsd_mat4_t* xforms, int32_t n_poses, float search_radius, int32_t max_n_neigh, float* scores )
{
  // (the reference's level grids use radius 0.05 -> cell 0.10, lib/rs/rs_pointcloud.h:862)
  rs_hip_cloud_t* obj = cached_cloud( obj_pos, obj_nor, n_obj, -1.0f );
  rs_hip_cloud_t* scn = cached_cloud( scn_pos, scn_nor, n_scn, -1.0f );
  if( !obj || !scn ) return RS_HIP_E_RUNTIME;
  int rc = rs_hip_alignment_scores( obj, scn, (const float*)xforms, n_poses, search_radius, max_n_neigh, scores );
  if( rc ) complain( "alignment_scores" );
  return rc;
}

float rsd_alignment_score( const rsd_vec3_t* obj_pos, const rsd_vec3_t* obj_nor, int32_t n_obj,
                           const rsd_vec3_t* scn_pos, const rsd_vec3_t* scn_nor, int32_t n_scn,
                           rsd_mat4_t xform, float search_radius, int32_t max_n_neigh )
{
  float s = 0.0f;
  rsd_alignment_scores( obj_pos, obj_nor, n_obj, scn_pos, scn_nor, n_scn, &xform, 1, search_radius, max_n_neigh, &s );
  return s;
}

int rsd_arrangement_to_labels( const rsd_vec3_t* scn_pos, const rsd_vec3_t* scn_nor, int32_t n_scn,
                               const rsd_vec3_t* const* obj_pos, const rsd_vec3_t* const* obj_nor, const int32_t* obj_n,
                               const rsd_mat4_t* poses, const int32_t* is_static, const int32_t* class_idx, int32_t n_plc,
                               float radius, bool prioritize_static, int8_t* labels, int32_t* sorted_order )
{
  rs_hip_cloud_t* scn = cached_cloud( scn_pos, scn_nor, n_scn, -1.0f );
  if( !scn ) return RS_HIP_E_RUNTIME;
  std::vector<const rs_hip_cloud_t*> objs( (size_t)( n_plc > 0 ? n_plc : 0 ) );
  for( int i = 0; i < n_plc; ++i )
  {
    objs[i] = cached_cloud( obj_pos[i], obj_nor[i], obj_n[i], -1.0f );
    if( !objs[i] ) return RS_HIP_E_RUNTIME;
  }
  std::vector<float> min_dists( (size_t)( n_scn > 0 ? n_scn : 0 ) );
  int rc = rs_hip_arrangement_to_labels( scn, (const float*)poses, objs.data(), is_static, class_idx, n_plc, radius,
                                         prioritize_static ? 1 : 0, labels, min_dists.data(), sorted_order );
  if( rc ) complain( "arrangement_to_labels" );
  return rc;
}

int64_t rsd_compute_neighborhood( const rsd_vec3_t* pos, const rsd_vec3_t* nor, int32_t n,
                                  int32_t max_nn, float radius_sq, float dist_exp, float angle_exp,
                                  int32_t* idx1, int32_t* idx2, float* weight )
{
  rs_hip_cloud_t* c = cached_cloud( pos, nor, n, -1.0f );
  if( !c ) return RS_HIP_E_RUNTIME;
  int64_t n_edges = 0;
  int rc = rs_hip_compute_neighborhood( c, max_nn, radius_sq, dist_exp, angle_exp, idx1, idx2, weight,
                                        (int64_t)n * max_nn, &n_edges );
  if( rc ) { complain( "compute_neighborhood" ); return rc; }
  return n_edges;
}

int32_t rsd_level_poisson( const rsd_vec3_t* pos, int32_t n, float voxel_size, int32_t level, int32_t* sample_idx )
{
  rs_hip_cloud_t* c = cached_cloud( pos, nullptr, n, -1.0f );
  if( !c ) return RS_HIP_E_RUNTIME;
  size_t max_n_neigh = (size_t)( 1024 * ( ( level ) / (float)( 5 - 1 ) ) );      // rs_pointcloud.h:995 (RSPC_N_LEVELS = 5)
  if( !max_n_neigh ) max_n_neigh = 256;                                           // :996
  int32_t n_samples = 0;
  int rc = rs_hip_level_samples( c, voxel_size, (int32_t)max_n_neigh, sample_idx, &n_samples, nullptr );
  if( rc ) { complain( "level_poisson" ); return rc; }
  return n_samples;
}

void* rsd_coverage_create( const rsd_vec3_t* bbox_min, const rsd_vec3_t* bbox_max, float voxel_size,
                           const rsd_vec3_t* scene_pos, const float* scene_quality, int32_t n_scene, float quality_threshold )
{
  rs_hip_coverage_t* c = rs_hip_coverage_create( (const float*)bbox_min, (const float*)bbox_max, voxel_size, (const float*)scene_pos,
                                                 scene_quality, n_scene, quality_threshold );
  if( !c ) complain( "coverage_create" );
  return c;
}

float rsd_coverage_score( void* coverage, const rsd_vec3_t* const* obj_pos, const int32_t* obj_n,
                          const rsd_mat4_t* poses, const int32_t* is_static, int32_t n_plc )
{
  std::vector<const rs_hip_cloud_t*> objs( (size_t)( n_plc > 0 ? n_plc : 0 ) );
  for( int i = 0; i < n_plc; ++i )
  {
    objs[i] = is_static[i] ? nullptr : cached_cloud( obj_pos[i], nullptr, obj_n[i], -1.0f );
    if( !is_static[i] && !objs[i] ) return -1.0f;
  }
  const int32_t first[2] = { 0, n_plc };
  float score = 0.0f;
  int rc = rs_hip_coverage_scores( (rs_hip_coverage_t*)coverage, objs.data(), (const float*)poses, is_static, first, 1, &score, nullptr );
  if( rc ) { complain( "coverage_score" ); return -1.0f; }
  return score;
}

void rsd_coverage_destroy( void* coverage ) { rs_hip_coverage_destroy( (rs_hip_coverage_t*)coverage ); }

// ---- on-disk formats -------------------------------------------------------------------------

int rsd_pose_bin_write( const char* path, int32_t n_arrays, const int32_t* counts, const float* records )
{
  if( !path || n_arrays < 0 || ( n_arrays > 0 && !counts ) ) return RS_HIP_E_ARG;
  FILE* fp = fopen( path, "wb" );
  if( !fp ) { fprintf( stderr, "[rescan_hip] could not open %s\n", path ); return RSD_FORMAT_ERR; }
  size_t total = 0;
  bool ok = fwrite( &n_arrays, sizeof(int32_t), 1, fp ) == 1;                         // main.cpp:69-70
  for( int32_t i = 0; ok && i < n_arrays; ++i ) { ok = counts[i] >= 0 && fwrite( &counts[i], sizeof(int32_t), 1, fp ) == 1; total += (size_t)counts[i]; }   // :71-75
  if( ok && total ) ok = records && fwrite( records, 17 * sizeof(float), total, fp ) == total;   // :77-87: 16 floats + score, arrays in order
  fclose( fp );
  return ok ? 0 : RSD_FORMAT_ERR;
}

int rsd_pose_bin_read( const char* path, int32_t* n_arrays, int32_t** counts, float** records )
{
  if( !path || !n_arrays || !counts || !records ) return RS_HIP_E_ARG;
  *n_arrays = 0; *counts = nullptr; *records = nullptr;
  FILE* fp = fopen( path, "rb" );
  if( !fp ) return RSD_FORMAT_ERR;
  int32_t n = -1;
  if( fread( &n, sizeof(int32_t), 1, fp ) != 1 || n < 0 ) { fclose( fp ); return RSD_FORMAT_ERR; }     // segment_transfer/main.cpp:154-155
  int32_t* cnt = (int32_t*)malloc( (size_t)( n > 0 ? n : 1 ) * sizeof(int32_t) );
  size_t total = 0;
  for( int32_t i = 0; i < n; ++i )
  {
    if( fread( &cnt[i], sizeof(int32_t), 1, fp ) != 1 || cnt[i] < 0 ) { free( cnt ); fclose( fp ); return RSD_FORMAT_ERR; }   // :157-163
    total += (size_t)cnt[i];
  }
  float* rec = (float*)malloc( ( total ? total : 1 ) * 17 * sizeof(float) );
  const size_t got = total ? fread( rec, 17 * sizeof(float), total, fp ) : 0;                             // :175-176
  fclose( fp );
  if( got != total ) { free( cnt ); free( rec ); return RSD_FORMAT_ERR; }
  *n_arrays = n; *counts = cnt; *records = rec;
  return 0;
}

int rsd_rsdb_format_pose_line( char* out, size_t capacity, int32_t uidx, int32_t arrangement_idx, int32_t object_idx,
                               float score, const rsd_mat4_t* pose )
{
  if( !out || !pose ) return RS_HIP_E_ARG;
  const float* m = pose->data;   // column-major: col[c].{x,y,z,w} = m[4c + {0,1,2,3}]; printed row by row (rs_database.h:599-605)
  int len = snprintf( out, capacity, "pose %d %d %d %f   %f %f %f %f  %f %f %f %f  %f %f %f %f  %f %f %f %f",
                      uidx, arrangement_idx, object_idx, score,
                      m[0], m[4], m[8], m[12],  m[1], m[5], m[9], m[13],  m[2], m[6], m[10], m[14],  m[3], m[7], m[11], m[15] );
  return ( len < 0 || (size_t)len >= capacity ) ? RSD_FORMAT_ERR : len;
}

int rsd_rsdb_parse_pose_line( const char* line, int32_t* uidx, int32_t* arrangement_idx, int32_t* object_idx,
                              float* score, rsd_mat4_t* pose )
{
  if( !line || !uidx || !arrangement_idx || !object_idx || !score || !pose ) return RS_HIP_E_ARG;
  char cmd[128] = { 0 };
  float* m = pose->data;
  for( int i = 0; i < 16; ++i ) m[i] = ( i % 5 == 0 ) ? 1.0f : 0.0f;       // msh_mat4_identity(), rs_database.h:386
  *uidx = *arrangement_idx = *object_idx = -1; *score = 0.0f;
  const int got = sscanf( line, "%127s  %d %d %d %f  %f %f %f %f  %f %f %f %f  %f %f %f %f  %f %f %f %f",   // :387-393
                          cmd, uidx, arrangement_idx, object_idx, score,
                          &m[0], &m[4], &m[8], &m[12],  &m[1], &m[5], &m[9], &m[13],  &m[2], &m[6], &m[10], &m[14],  &m[3], &m[7], &m[11], &m[15] );
  return got == 21 ? 0 : RSD_FORMAT_ERR;
}

} // extern "C"
