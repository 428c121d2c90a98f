// librescan_dropin.so — the reference's own symbol names for the hot path, forwarding to the
// C ABI of librescan_hip.so (include/rescan_hip.h).  See include/rescan_dropin.h.
//
// Error behaviour follows the reference: no error codes on this path.  A failing
// HIP call prints to stderr and leaves the caller's data untouched (icp_align returns the 1e6 it
// would have returned had it never iterated, lib/rs/icp.h:441-442).  There is no CPU fallback for
// compute: without a usable HIP device a grid is not even initialised.  The one exception is a HIP
// error DURING a batched msh_hash_grid_radius_search on an initialised grid: the reference's search
// always fills n_neighbors and the rows (lib/msh/msh_hash_grid.h:1090-1259) and its callers walk
// them unconditionally, so such a call is answered from the grid's own host copy (the one that
// serves one-query calls), loudly: a complaint on stderr and a count in rsd_device_failures().

#include "../../include/rescan_dropin.h"
#include "../../include/rescan_hip.h"

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <list>
#include <memory>
#include <mutex>
#include <vector>

static_assert( sizeof(rsd_vec3_t) == 12, "msh_vec3_t is 12 bytes" );
static_assert( sizeof(rsd_mat4_t) == 64, "msh_mat4_t is 64 bytes" );
static_assert( sizeof(rsd_hash_grid_t) == 120, "msh_hash_grid_t is 120 bytes (lib/msh/msh_hash_grid.h:248-269)" );
static_assert( offsetof(rsd_hash_grid_t, data_buffer) == 64, "data_buffer offset" );
static_assert( offsetof(rsd_hash_grid_t, _n_pts) == 112, "_n_pts offset" );
static_assert( sizeof(rsd_search_desc_t) == 64, "msh_hash_grid_search_desc_t is 64 bytes" );

namespace {

// The unchanged apps wait for ~37 k small searches per run: ask librescan_hip.so for busy-waiting completion waits (27 instead of
// ~45 us per call), unless the environment says otherwise.  (Set before the library's first use: it reads the variable in rs_hip_init.)
struct AskForSpinWaits { AskForSpinWaits() { setenv( "RS_HIP_SCHEDULE", "spin", 0 ); } } g_ask_for_spin_waits;

void complain( const char* where )
{
  fprintf( stderr, "[rescan_hip] %s: %s\n", where, rs_hip_last_error() );
}

// RS_DROPIN_STATS=1: where the shim's time went, printed when the process ends (calls / seconds per entry point and route)
struct Stats
{
  bool on = getenv( "RS_DROPIN_STATS" ) != nullptr;
  struct Row { const char* name; unsigned long long calls = 0, items = 0; double seconds = 0.0; };
  Row rows[8] = { { "grid init (host index)" }, { "grid device index (lazy)" }, { "radius_search, host path" }, { "radius_search, device path" },
                  { "icp_align" }, { "scores (flat entry)" }, { "labels (flat entry)" }, { "other" } };
  ~Stats()
  {
    if( !on ) return;
    for( const Row& r : rows ) if( r.calls ) fprintf( stderr, "[rescan_hip stats] %-28s %9llu calls %12llu items %9.3f s\n", r.name, r.calls, r.items, r.seconds );
  }
} g_stats;
struct Timed
{
  Stats::Row* r; std::chrono::steady_clock::time_point t0;
  Timed( int row, unsigned long long items ) : r( g_stats.on ? &g_stats.rows[row] : nullptr ) { if( r ) { r->calls++; r->items += items; t0 = std::chrono::steady_clock::now(); } }
  ~Timed() { if( r ) r->seconds += std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count(); }
};

// ---- device-cloud cache ------------------------------------------------------------------
// The reference hands the same host arrays to the hot path over and over (the scene level for every proposal,
// apps/pose_proposal/main.cpp:190-202; every placement's level-2 cloud, lib/rs/rs_database.h:220-230).  Uploads are
// cached by (pointers, count, cell size, fingerprint) — SURVEY.md §8b's "(pointer, n, generation)" — and evicted
// least-recently-used.  The fingerprint is a SAMPLE of the arrays (first and last 64 bytes plus 62 blocks spread over
// the rest: ~1 KB read per array, whatever its size), enough to notice an array that was freed and re-used, rebuilt or
// transformed; the full content hash the shim started with cost 2.3 ms per 57 MB on every call, i.e. tens of seconds
// over the ~35 k score calls of one pose_proposal run.  What a sample cannot see — a caller editing a few points in
// place — is covered by explicit invalidation: msh_hash_grid_term() drops every entry built from the array its grid was
// built on (the reference terminates a level's grid right before it frees or rebuilds the level's arrays,
// lib/rs/rs_pointcloud.h:879-901), rsd_cache_invalidate( ptr ) does the same for a caller's own arrays, rsd_cache_clear()
// for everything; RS_DROPIN_FULL_HASH=1 brings the full hash back.
struct CloudDeleter { void operator()( rs_hip_cloud_t* c ) const { if( c ) rs_hip_cloud_destroy( c ); } };
// A cached cloud is handed out as a shared reference: the caller keeps it for the duration of its call, so an eviction (a
// call with more placements than the cache has entries) or an invalidation from another thread can drop the cache's entry
// at any time — the device memory goes when the last holder lets go.
typedef std::shared_ptr<rs_hip_cloud_t> CloudRef;
struct Entry
{
  const void* pos; const void* nor; int32_t n; float cell; uint64_t hash; uint64_t generation;
  uint64_t full; bool sampled; unsigned hits;      // sampled keys: the arrays' full hash when the entry was made, re-checked every kVerifyEvery hits
  CloudRef cloud;
};
std::list<Entry> g_cache;
std::mutex g_cache_mutex;
uint64_t g_generation = 1;          // bumped by every invalidation; an entry remembers the generation it was made in
const size_t kMaxEntries = 64;
const unsigned kVerifyEvery = 16;

inline uint64_t mix( uint64_t h, uint64_t w ) { h = ( h ^ w ) * 0xff51afd7ed558ccdull; return h ^ ( h >> 29 ); }

uint64_t full_hash( const void* a, const void* b, size_t bytes )
{
  // four independent multiply-xor lanes over 32-byte blocks (the dependent chain of a single lane caps at ~8 GB/s)
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

// Arrays of up to RS_DROPIN_FULL_HASH_BELOW bytes each (default 4 MB: ~0.15 ms) are keyed by their FULL content hash — every
// object level and every level-2 scan the reference's call sites pass; an in-place edit of any byte is seen.  Larger arrays are
// keyed by a sample (first and last 64 bytes plus 62 blocks spread over the rest) and the full hash is re-checked on the first
// and then every kVerifyEvery-th hit; RS_DROPIN_FULL_HASH=1 hashes everything on every call.
size_t full_hash_below()
{
  static const size_t v = getenv( "RS_DROPIN_FULL_HASH" ) ? ~(size_t)0 : getenv( "RS_DROPIN_FULL_HASH_BELOW" ) ? (size_t)atoll( getenv( "RS_DROPIN_FULL_HASH_BELOW" ) ) : ( (size_t)4 << 20 );
  return v;
}

uint64_t sampled_hash( const void* a, const void* b, size_t bytes )
{
  uint64_t h = 0x9e3779b97f4a7c15ull ^ bytes;
  for( const void* src : { a, b } )
  {
    if( !src ) { h = mix( h, 0x51ull ); continue; }
    const unsigned char* p = (const unsigned char*)src;
    uint64_t w[8];
    std::memcpy( w, p, 64 );               for( uint64_t v : w ) h = mix( h, v );
    std::memcpy( w, p + bytes - 64, 64 );  for( uint64_t v : w ) h = mix( h, v );
    const size_t step = ( ( bytes - 128 ) / 62 ) & ~(size_t)3;
    for( int k = 0; k < 62; ++k ) { std::memcpy( w, p + 64 + (size_t)k * step, 16 ); h = mix( mix( h, w[0] ), w[1] ); }
  }
  return h;
}

// Round 6 — the middle tier.  An arrangement's placements are 8-40 arrays of 0.1-0.6 MB each: full hashes of all of them were the
// 0.5 ms the label entry spent on the host before its 0.38 ms kernel.  Arrays between RS_DROPIN_FULL_HASH_SMALL
// (default 64 KB: every level-2 object of the reference's call sites stays on the full hash) and RS_DROPIN_FULL_HASH_BELOW are keyed by
// every eighth 64-byte block plus both ends — an edit of 512 consecutive bytes anywhere is seen at once, any edit at the next
// periodic full-hash check (first hit, then every kVerifyEvery-th), rsd_cache_invalidate / msh_hash_grid_term at once.
size_t full_hash_small()
{
  static const size_t v = getenv( "RS_DROPIN_FULL_HASH" ) ? ~(size_t)0 : getenv( "RS_DROPIN_FULL_HASH_SMALL" ) ? (size_t)atoll( getenv( "RS_DROPIN_FULL_HASH_SMALL" ) ) : ( (size_t)64 << 10 );
  return v;
}
uint64_t strided_hash( const void* a, const void* b, size_t bytes )
{
  uint64_t h = 0x2545f4914f6cdd1dull ^ bytes;
  for( const void* src : { a, b } )
  {
    if( !src ) { h = mix( h, 0x51ull ); continue; }
    const unsigned char* p = (const unsigned char*)src;
    uint64_t w[8];
    uint64_t l0 = h, l1 = h ^ 0x165667b19e3779f9ull, l2 = h ^ 0x27d4eb2f165667c5ull, l3 = h ^ 0x85ebca77c2b2ae63ull;
    auto block = [&]( const unsigned char* q )
    {
      std::memcpy( w, q, 64 );
      l0 = ( l0 ^ w[0] ^ ( w[4] << 1 ) ) * 0xff51afd7ed558ccdull; l0 ^= l0 >> 29;
      l1 = ( l1 ^ w[1] ^ ( w[5] << 1 ) ) * 0xc4ceb9fe1a85ec53ull; l1 ^= l1 >> 31;
      l2 = ( l2 ^ w[2] ^ ( w[6] << 1 ) ) * 0x9fb21c651e98df25ull; l2 ^= l2 >> 30;
      l3 = ( l3 ^ w[3] ^ ( w[7] << 1 ) ) * 0xd6e8feb86659fd93ull; l3 ^= l3 >> 32;
    };
    for( size_t i = 0; i + 64 <= bytes; i += 512 ) block( p + i );
    block( p + bytes - 64 );
    h = ( ( l0 * 31 + l1 ) * 31 + l2 ) * 31 + l3;
  }
  return h;
}

CloudRef cached_cloud( const rsd_vec3_t* pos, const rsd_vec3_t* nor, int32_t n, float cell )
{
  const size_t bytes = (size_t)( n > 0 ? n : 0 ) * 12;
  const bool sparse = bytes > full_hash_below() && bytes > 2048;
  const bool sampled = sparse || ( bytes > full_hash_small() && bytes > 2048 );
  const uint64_t h = sparse ? sampled_hash( pos, nor, bytes ) : sampled ? strided_hash( pos, nor, bytes ) : full_hash( pos, nor, bytes );
  {
    std::unique_lock<std::mutex> lock( g_cache_mutex );
    for( auto it = g_cache.begin(); it != g_cache.end(); ++it )
      if( it->pos == pos && it->nor == nor && it->n == n && it->cell == cell && it->sampled == sampled && it->hash == h )
      {
        if( sampled && ( it->hits++ % kVerifyEvery ) == 0 )
        {
          const uint64_t expect = it->full;
          lock.unlock();                                    // (hashing tens of MB: not under the lock)
          const bool same = full_hash( pos, nor, bytes ) == expect;
          lock.lock();
          it = g_cache.begin();
          while( it != g_cache.end() && !( it->pos == pos && it->nor == nor && it->n == n && it->cell == cell && it->sampled && it->hash == h ) ) ++it;
          if( it == g_cache.end() ) break;                  // gone meanwhile: rebuild
          if( !same ) { g_cache.erase( it ); break; }       // edited in place where the sample does not look: rebuild
        }
        g_cache.splice( g_cache.begin(), g_cache, it );
        return g_cache.front().cloud;
      }
  }
  const uint64_t full = sampled ? full_hash( pos, nor, bytes ) : h;
  rs_hip_cloud_t* c = rs_hip_cloud_create( (const float*)pos, (const float*)nor, n, cell );
  if( !c ) { complain( "cloud upload" ); return CloudRef(); }
  CloudRef ref( c, CloudDeleter() );
  std::lock_guard<std::mutex> lock( g_cache_mutex );
  g_cache.push_front( Entry{ pos, nor, n, cell, h, g_generation, full, sampled, 0u, ref } );      // hits = 0: the FIRST hit re-checks the full hash, then every kVerifyEvery-th
  while( g_cache.size() > kMaxEntries ) g_cache.pop_back();        // (holders of the evicted cloud keep it alive)
  return ref;
}

void invalidate_pointer( const void* p )
{
  if( !p ) return;
  std::lock_guard<std::mutex> lock( g_cache_mutex );
  ++g_generation;
  for( auto it = g_cache.begin(); it != g_cache.end(); )
    if( it->pos == p || it->nor == p ) it = g_cache.erase( it ); else ++it;
}

// ---- host side of a grid -------------------------------------------------------------------
// msh_hash_grid_init_3d keeps a copy of the points, like the reference's (msh_hash_grid.h:511-532): sorted by cell,
// {x, y, z, original index}, with a dense cell offset table.  It serves two things.  (1) Searches of a handful of queries:
// rs_pointcloud__compute_level_poisson (lib/rs/rs_pointcloud.h:1007-1037) issues ONE query per call, the next one
// depending on the previous one's result, up to ~10^6 times per cloud; a kernel launch, two copies and a synchronisation
// per point would be ~30 us each against ~1 us of arithmetic, so such calls (n_query_pts <= RS_DROPIN_HOST_QUERIES,
// default 4) are answered on the host from this copy — a dispatch by call size, not a fallback: a grid cannot be
// initialised without a HIP device, and everything batched goes to the GPU.  (2) The device cloud of a grid is built
// lazily, at the first batched search, from this copy (the caller's array may be gone by then): most level grids
// (rs_pointcloud_compute_search_grid builds five per cloud, :849-863) are never searched at all.
struct HostGrid
{
  int32_t n = 0;
  float cell = 0.0f, inv_cell = 0.0f, mn[3] = { 0, 0, 0 };
  int dims[3] = { 1, 1, 1 };
  std::vector<uint32_t> cell_start;                 // dims product + 1
  std::vector<float> rec;                           // 4 floats per point, cell order: x, y, z, bitcast(original index)

  inline int bin( float v, int a ) const
  {
    float f = std::floor( ( v - mn[a] ) * inv_cell );
    if( !( f >= 0.0f ) ) f = 0.0f;
    if( f > (float)( dims[a] - 1 ) ) f = (float)( dims[a] - 1 );
    return (int)f;
  }
  void build( const float* pts, int32_t n_pts, float cell_size )
  {
    n = n_pts > 0 ? n_pts : 0;
    float mx[3] = { 0, 0, 0 };
    bool any = false;
    for( int32_t i = 0; i < n; ++i )
    {
      const float* p = pts + 3 * (size_t)i;
      if( !( std::isfinite( p[0] ) && std::isfinite( p[1] ) && std::isfinite( p[2] ) ) ) continue;
      for( int a = 0; a < 3; ++a ) { if( !any || p[a] < mn[a] ) mn[a] = p[a]; if( !any || p[a] > mx[a] ) mx[a] = p[a]; }
      any = true;
    }
    cell = cell_size;
    for( ;; )
    {
      double cells = 1.0;
      for( int a = 0; a < 3; ++a ) { dims[a] = (int)std::floor( ( (double)mx[a] - (double)mn[a] ) / (double)cell ) + 1; cells *= dims[a]; }
      if( cells <= 16.0e6 ) break;
      cell *= 2.0f;                                   // keep the dense table small; any cell size gives the same results
    }
    inv_cell = 1.0f / cell;
    const size_t n_cells = (size_t)dims[0] * dims[1] * dims[2];
    cell_start.assign( n_cells + 1, 0u );
    std::vector<uint32_t> id( (size_t)n );
    for( int32_t i = 0; i < n; ++i )
    {
      const float* p = pts + 3 * (size_t)i;
      id[i] = (uint32_t)( ( (size_t)bin( p[2], 2 ) * dims[1] + bin( p[1], 1 ) ) * dims[0] + bin( p[0], 0 ) );
      cell_start[id[i] + 1]++;
    }
    for( size_t c = 0; c < n_cells; ++c ) cell_start[c + 1] += cell_start[c];
    std::vector<uint32_t> cur( cell_start.begin(), cell_start.end() - 1 );
    rec.resize( (size_t)n * 4 );
    for( int32_t i = 0; i < n; ++i )                 // input order inside a cell, like the reference
    {
      const size_t s = cur[id[i]]++;
      std::memcpy( &rec[4 * s], pts + 3 * (size_t)i, 12 );
      std::memcpy( &rec[4 * s + 3], &i, 4 );
    }
  }
  // the (at most) k nearest with dist² < radius² (msh_hash_grid.h:852-857,1111), ascending (dist², index)
  size_t search( const float* q, float radius, size_t k, float* out_d2, int32_t* out_idx ) const
  {
    if( n == 0 || !( std::isfinite( q[0] ) && std::isfinite( q[1] ) && std::isfinite( q[2] ) ) ) return 0;
    const float r2 = (float)( (double)radius * (double)radius );
    int lo[3], hi[3];
    for( int a = 0; a < 3; ++a )
    {
      float fa = std::floor( ( q[a] - radius - mn[a] ) * inv_cell - 0.01f ), fb = std::floor( ( q[a] + radius - mn[a] ) * inv_cell + 0.01f );
      if( fa < 0.0f ) fa = 0.0f;
      if( fb > (float)( dims[a] - 1 ) ) fb = (float)( dims[a] - 1 );
      if( !( fb >= fa ) ) return 0;
      lo[a] = (int)fa; hi[a] = (int)fb;
    }
    thread_local std::vector<std::pair<float, int32_t>> hits;
    hits.clear();
    for( int z = lo[2]; z <= hi[2]; ++z )
      for( int y = lo[1]; y <= hi[1]; ++y )
      {
        const size_t row = ( (size_t)z * dims[1] + y ) * dims[0];
        const uint32_t s0 = cell_start[row + lo[0]], s1 = cell_start[row + hi[0] + 1];
        for( uint32_t s = s0; s < s1; ++s )
        {
          const float* p = &rec[4 * (size_t)s];
          const float vx = p[0] - q[0], vy = p[1] - q[1], vz = p[2] - q[2];
          const float d2 = vx * vx + vy * vy + vz * vz;
          if( d2 < r2 ) { int32_t i; std::memcpy( &i, p + 3, 4 ); hits.emplace_back( d2, i ); }
        }
      }
    if( hits.size() > k ) { std::nth_element( hits.begin(), hits.begin() + (ptrdiff_t)k, hits.end() ); hits.resize( k ); }
    std::sort( hits.begin(), hits.end() );
    for( size_t t = 0; t < hits.size(); ++t ) { out_d2[t] = hits[t].first; out_idx[t] = hits[t].second; }
    return hits.size();
  }
};

// msh_hash_grid_knn_search (lib/msh/msh_hash_grid.h:1291-1447) is NOT an exact k-nearest search: it visits the bins around the query's
// own bin shell by shell (Chebyshev distance 0, 1, 2, ... in bins), skips a bin farther than the k-th distance so far, and stops ONE
// shell after k points are stored — what it returns depends on the grid's own geometry (cell = 2 radius, the bounding box grown by
// 1e-4, bins by truncation: :413-448,471-475).  Nothing on the hot path calls it (SURVEY §8b lists it with the boundary's exports,
// VERDICT r05 missing 3): it is restated here on the HOST, on a grid of exactly that geometry built from the handle's copy of the
// points at the first call, so that a maintainer's translation unit that uses it links against the shim and gets the reference's
// rows.  Where the reference's behaviour is undefined or it does not return, the shim answers: a third shell's 218 bins overrun the
// reference's 128-entry stack array (:1330,1412-1414) — a vector here; a cloud with fewer than k points, or a query outside the
// grid's box, makes the reference spin forever — here the search ends when every bin has been visited.  Rows are returned ascending
// in (dist², index), like every row of this library.
struct KnnGrid
{
  float min_pt[3] = { 0, 0, 0 };
  double cell = 0.0, inv_cell = 0.0;
  int64_t w = 1, h = 1, d = 1;
  std::vector<uint32_t> start;            // w h d + 1
  std::vector<float> rec;                 // x, y, z, bitcast(index), bin order, input order inside a bin
  void build( const HostGrid& hg, float radius )
  {
    const int32_t n = hg.n;
    std::vector<float> pts( (size_t)n * 3 );
    for( int32_t s = 0; s < n; ++s ) { int32_t i; std::memcpy( &i, &hg.rec[4 * (size_t)s + 3], 4 ); std::memcpy( &pts[3 * (size_t)i], &hg.rec[4 * (size_t)s], 12 ); }
    float mn[3] = { 1e9f, 1e9f, 1e9f }, mx[3] = { -1e9f, -1e9f, -1e9f };                        // :413-414
    for( int32_t i = 0; i < n; ++i ) for( int a = 0; a < 3; ++a ) { const float v = pts[3 * (size_t)i + a]; mn[a] = ( mn[a] > v ) ? v : mn[a]; mx[a] = ( mx[a] < v ) ? v : mx[a]; }
    for( int a = 0; a < 3; ++a ) { mx[a] += 0.0001f; mn[a] -= 0.0001f; min_pt[a] = mn[a]; }   // :433-434
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    const float max_dim = std::max( dx, std::max( dy, dz ) );
    if( radius > 0.0 ) cell = 2.0 * radius; else cell = max_dim / ( 32 * sqrtf( 3.0f ) );          // :443-444
    w = (int)( dx / cell + 1.0 ); h = (int)( dy / cell + 1.0 ); d = (int)( dz / cell + 1.0 );     // :446-448
    w = std::max<int64_t>( w, 1 ); h = std::max<int64_t>( h, 1 ); d = std::max<int64_t>( d, 1 );
    inv_cell = 1.0f / cell;                                                                       // :449
    const size_t n_bins = (size_t)w * h * d;
    start.assign( n_bins + 1, 0u );
    std::vector<uint64_t> id( (size_t)n );
    for( int32_t i = 0; i < n; ++i )
    {
      const float* p = &pts[3 * (size_t)i];
      uint64_t ix = (uint64_t)( ( p[0] - min_pt[0] ) * inv_cell ), iy = (uint64_t)( ( p[1] - min_pt[1] ) * inv_cell ), iz = (uint64_t)( ( p[2] - min_pt[2] ) * inv_cell );   // :471-473
      ix = std::min<uint64_t>( ix, (uint64_t)w - 1 ); iy = std::min<uint64_t>( iy, (uint64_t)h - 1 ); iz = std::min<uint64_t>( iz, (uint64_t)d - 1 );       // (never for finite points: the box is the points' own)
      id[i] = ( iz * (uint64_t)h + iy ) * (uint64_t)w + ix;
      start[id[i] + 1]++;
    }
    for( size_t b = 0; b < n_bins; ++b ) start[b + 1] += start[b];
    std::vector<uint32_t> cur( start.begin(), start.end() - 1 );
    rec.resize( (size_t)n * 4 );
    for( int32_t i = 0; i < n; ++i ) { const size_t s = cur[id[i]]++; std::memcpy( &rec[4 * s], &pts[3 * (size_t)i], 12 ); std::memcpy( &rec[4 * s + 3], &i, 4 ); }
  }
  // the reference's traversal for one query (q: 3 floats; a 2-D grid's queries carry z = 0, like its points)
  size_t search( const float* q, size_t k, float* out_d2, int32_t* out_idx ) const
  {
    const float px = q[0] - min_pt[0], py = q[1] - min_pt[1], pz = q[2] - min_pt[2];           // pt_prime (:1350-1361)
    auto base = []( float v, double inv, int64_t dim ) { double t = (double)v * inv; if( !( t >= 0.0 ) ) t = 0.0; if( t > (double)( dim - 1 ) ) t = (double)( dim - 1 ); return (int64_t)t; };
    const int64_t ix = base( px, inv_cell, w ), iy = base( py, inv_cell, h ), iz = base( pz, inv_cell, d );      // :1363-1365 (clamped: see above)
    const float cs = (float)cell;                                                                                   // `float cs = hg->cell_size` (:1312)
    thread_local std::vector<std::pair<float, int32_t>> kept;
    thread_local std::vector<int64_t> bins;
    kept.clear();
    float max_dist = 0.0f;      // the k-th distance so far (valid once kept.size() >= k)
    bool should_break = false;
    const int64_t last_layer = std::max( w, std::max( h, d ) );
    for( int64_t layer = 0; layer <= last_layer; ++layer )
    {
      bins.clear();
      for( int64_t oz = -layer; oz <= layer; ++oz )
      {
        const int64_t cz = iz + oz;
        if( cz < 0 || cz >= d ) continue;
        float ddz; if( oz < 0 ) ddz = pz - ( cz + 1 ) * cs; else if( oz > 0 ) ddz = cz * cs - pz; else ddz = 0.0f;                 // :1378-1380
        for( int64_t oy = -layer; oy <= layer; ++oy )
        {
          const int64_t cy = iy + oy;
          if( cy < 0 || cy >= h ) continue;
          float ddy; if( oy < 0 ) ddy = py - ( cy + 1 ) * cs; else if( oy > 0 ) ddy = cy * cs - py; else ddy = 0.0f;               // :1388-1390
          const int64_t inc_x = ( std::llabs( oy ) != layer && std::llabs( oz ) != layer ) ? 2 * layer : 1;                          // :1392-1393 (the shell's faces only)
          for( int64_t ox = -layer; ox <= layer; ox += inc_x )
          {
            const int64_t cx = ix + ox;
            if( cx < 0 || cx >= w ) continue;
            float ddx; if( ox < 0 ) ddx = px - ( cx + 1 ) * cs; else if( ox > 0 ) ddx = cx * cs - px; else ddx = 0.0f;             // :1400-1402
            const float dist_sq = ddz * ddz + ddy * ddy + ddx * ddx;                                                                // :1404
            if( kept.size() >= k && dist_sq > max_dist ) continue;                                                                  // :1406-1407
            bins.push_back( ( cz * h + cy ) * w + cx );
          }
        }
      }
      for( int64_t b : bins )
        for( uint32_t s = start[(size_t)b]; s < start[(size_t)b + 1]; ++s )
        {
          const float* p = &rec[4 * (size_t)s];
          const float vx = p[0] - q[0], vy = p[1] - q[1], vz = p[2] - q[2];                                                         // :1278-1287
          int32_t i; std::memcpy( &i, p + 3, 4 );
          kept.emplace_back( vx * vx + vy * vy + vz * vz, i );
        }
      if( kept.size() > k ) { std::nth_element( kept.begin(), kept.begin() + (ptrdiff_t)k, kept.end() ); kept.resize( k ); }   // the heap of the k nearest (:797-824)
      if( kept.size() >= k ) { max_dist = 0.0f; for( const auto& e : kept ) max_dist = std::max( max_dist, e.first ); }
      if( should_break ) break;                                                                                                      // :1428-1429: one more shell after k were stored
      if( kept.size() >= k ) should_break = true;
    }
    std::sort( kept.begin(), kept.end() );
    for( size_t t = 0; t < kept.size(); ++t ) { out_d2[t] = kept[t].first; out_idx[t] = kept[t].second; }
    return kept.size();
  }
};

struct GridHandle
{
  const float* src = nullptr;         // the array the grid was built on (identity only: never dereferenced after init)
  HostGrid host;
  rs_hip_cloud_t* dev = nullptr;      // built at the first batched search
  bool dev_failed = false;
  int dim = 3;                        // 2: msh_hash_grid_init_2d — points and queries are (x, y), kept as (x, y, 0)
  float radius = 0.0f;                // as given to init (the k-NN grid's geometry)
  std::unique_ptr<KnnGrid> knn;       // built at the first msh_hash_grid_knn_search
  std::mutex mutex;
};

rs_hip_cloud_t* device_cloud_of( GridHandle* h )
{
  std::lock_guard<std::mutex> lock( h->mutex );
  if( h->dev || h->dev_failed ) return h->dev;
  Timed t( 1, (unsigned long long)h->host.n );
  std::vector<float> pts( (size_t)h->host.n * 3 );
  for( int32_t s = 0; s < h->host.n; ++s )
  {
    int32_t i; std::memcpy( &i, &h->host.rec[4 * (size_t)s + 3], 4 );
    std::memcpy( &pts[3 * (size_t)i], &h->host.rec[4 * (size_t)s], 12 );
  }
  h->dev = rs_hip_cloud_create( pts.data(), nullptr, h->host.n, h->host.cell );
  if( !h->dev ) { h->dev_failed = true; complain( "msh_hash_grid: device index" ); }
  return h->dev;
}

} // namespace

extern "C" {

void rsd_cache_clear( void )
{
  std::lock_guard<std::mutex> lock( g_cache_mutex );
  ++g_generation;
  g_cache.clear();
}

void rsd_cache_invalidate( const void* host_array ) { invalidate_pointer( host_array ); }

// ---- msh_hash_grid ------------------------------------------------------------------------

static void hash_grid_init( rsd_hash_grid_t* hg, const float* pts, const int32_t n_pts, const float radius, int dim );
void msh_hash_grid_init_3d( rsd_hash_grid_t* hg, const float* pts, const int32_t n_pts, const float radius ) { hash_grid_init( hg, pts, n_pts, radius, 3 ); }
// msh_hash_grid.h:544-548: points are (x, y) pairs; the reference keeps them as (x, y, 0) in the same structure, and so does the shim
void msh_hash_grid_init_2d( rsd_hash_grid_t* hg, const float* pts, const int32_t n_pts, const float radius ) { hash_grid_init( hg, pts, n_pts, radius, 2 ); }

static void hash_grid_init( rsd_hash_grid_t* hg, const float* pts, const int32_t n_pts, const float radius, int dim )
{
  // cell = 2*radius like the reference (msh_hash_grid.h:443); a non-positive radius asks the
  // reference for an extent-derived cell (:444) — any positive cell gives the same results here.
  const float cell = radius > 0.0f ? 2.0f * radius : 0.1f;
  hg->bin_table = nullptr; hg->offsets = nullptr; hg->data_buffer = nullptr;
  hg->cell_size = cell; hg->_inv_cell_size = 1.0 / cell;
  hg->_pts_dim = (uint8_t)dim; hg->_num_threads = 1; hg->_n_pts = (size_t)( n_pts > 0 ? n_pts : 0 );
  // No HIP device: no grid (there is no CPU fallback; every search on it returns 0 neighbours).  RS_DROPIN_INIT_WITHOUT_DEVICE=1
  // is for tests of the failure path on a machine without a GPU: the grid is initialised, and every batched search then
  // fails on the device side exactly as after a runtime error.
  if( rs_hip_synchronize() != RS_HIP_OK && !getenv( "RS_DROPIN_INIT_WITHOUT_DEVICE" ) ) { complain( "msh_hash_grid_init_3d" ); return; }
  GridHandle* h = new GridHandle();
  h->src = pts; h->dim = dim; h->radius = radius;
  {
    Timed t( 0, (unsigned long long)( n_pts > 0 ? n_pts : 0 ) );
    if( dim == 2 )
    {
      std::vector<float> p3( (size_t)( n_pts > 0 ? n_pts : 0 ) * 3 );
      for( int32_t i = 0; i < n_pts; ++i ) { p3[3 * (size_t)i] = pts[2 * (size_t)i]; p3[3 * (size_t)i + 1] = pts[2 * (size_t)i + 1]; p3[3 * (size_t)i + 2] = 0.0f; }
      h->host.build( p3.data(), n_pts, cell );
    }
    else h->host.build( pts, n_pts, cell );
  }
  hg->data_buffer = h;
  hg->width = (size_t)h->host.dims[0]; hg->height = (size_t)h->host.dims[1]; hg->depth = (size_t)h->host.dims[2];
}

void msh_hash_grid_term( rsd_hash_grid_t* hg )
{
  if( hg->data_buffer )
  {
    GridHandle* h = (GridHandle*)hg->data_buffer;
    invalidate_pointer( h->src );                        // whatever else was built from that array goes with its grid
    if( h->dev ) rs_hip_cloud_destroy( h->dev );
    delete h;
  }
  std::memset( hg, 0, sizeof(*hg) );                     // the reference zeroes what it owns (:558-573)
}

static std::atomic<unsigned long long> g_device_failures{ 0 };
unsigned long long rsd_device_failures( void ) { return g_device_failures.load(); }

// all queries of a call from the grid's host copy (the reference's own contract: every row and every count is written)
static size_t host_search_all( const GridHandle* h, rsd_search_desc_t* d )
{
  size_t total = 0;
  for( size_t i = 0; i < d->n_query_pts; ++i )
  {
    const size_t c = h->host.search( d->query_pts + 3 * i, d->radius, d->max_n_neigh, d->distances_sq + i * d->max_n_neigh, d->indices + i * d->max_n_neigh );
    if( d->n_neighbors ) d->n_neighbors[i] = c;
    total += c;
  }
  return total;
}

static size_t radius_search_3d( GridHandle* h, rsd_search_desc_t* d );
size_t msh_hash_grid_radius_search( const rsd_hash_grid_t* hg, rsd_search_desc_t* d )
{
  if( !hg || !d || !d->query_pts || !d->distances_sq || !d->indices || d->max_n_neigh == 0 ) return 0;
  // the reference writes n_neighbors[i] for every query (msh_hash_grid.h:1242) and its callers read them unconditionally:
  // a call that finds nothing — no grid, a non-positive radius — still says so per query
  if( !hg->data_buffer || !( d->radius > 0.0f ) )
  {
    if( d->n_neighbors ) for( size_t i = 0; i < d->n_query_pts; ++i ) d->n_neighbors[i] = 0;
    return 0;
  }
  GridHandle* h = (GridHandle*)hg->data_buffer;
  if( h->dim == 2 )
  {
    // (x, y) queries of a 2-D grid (msh_hash_grid_init_2d): the same search on (x, y, 0), like the reference's own (:1150-1162)
    std::vector<float> q3( d->n_query_pts * 3 );
    for( size_t i = 0; i < d->n_query_pts; ++i ) { q3[3 * i] = d->query_pts[2 * i]; q3[3 * i + 1] = d->query_pts[2 * i + 1]; q3[3 * i + 2] = 0.0f; }
    rsd_search_desc_t d3 = *d; d3.query_pts = q3.data();
    return radius_search_3d( h, &d3 );
  }
  return radius_search_3d( h, d );
}

static size_t radius_search_3d( GridHandle* h, rsd_search_desc_t* d )
{
  const size_t host_queries = getenv( "RS_DROPIN_HOST_QUERIES" ) ? (size_t)atoll( getenv( "RS_DROPIN_HOST_QUERIES" ) ) : 4;      // (read per call: tests switch it)
  if( d->n_query_pts <= host_queries )
  {
    Timed t( 2, d->n_query_pts );
    return host_search_all( h, d );
  }
  rs_hip_cloud_t* c = device_cloud_of( h );
  int rc = RS_HIP_E_RUNTIME;
  uint64_t total = 0;
  if( c )
  {
    Timed t( 3, d->n_query_pts );
    rc = rs_hip_radius_search( c, d->query_pts, (int64_t)d->n_query_pts, d->radius,
                               (int32_t)d->max_n_neigh, d->distances_sq, d->indices, d->n_neighbors, &total );
    if( rc ) complain( "msh_hash_grid_radius_search" );
  }
  if( rc == RS_HIP_OK ) return (size_t)total;
  // A HIP error in the middle of an app run ("print to stderr ... never abort the app"): the caller's rows may be
  // half written and its loops will walk them whatever we return, so the call is answered from the grid's host copy — the same
  // arithmetic and order as the one-query route (tests/test_dropin.py pins it against the reference's rows) — and counted.
  if( g_device_failures.fetch_add( 1 ) == 0 )
    fprintf( stderr, "[rescan_hip] msh_hash_grid_radius_search: the device path failed; this call and any later failing one are answered "
                     "from the grid's host copy (slow).  rsd_device_failures() counts them.\n" );
  Timed t( 2, d->n_query_pts );
  return host_search_all( h, d );
}

// msh_hash_grid.h:1294-1447 — see KnnGrid.  Host code: no caller of the hot path uses it; rows ascending in (dist², index).
size_t msh_hash_grid_knn_search( const rsd_hash_grid_t* hg, rsd_search_desc_t* d )
{
  if( !hg || !d || !d->query_pts || !d->distances_sq || !d->indices || d->k == 0 ) return 0;
  if( !hg->data_buffer )
  {
    if( d->n_neighbors ) for( size_t i = 0; i < d->n_query_pts; ++i ) d->n_neighbors[i] = 0;
    return 0;
  }
  GridHandle* h = (GridHandle*)hg->data_buffer;
  {
    std::lock_guard<std::mutex> lock( h->mutex );
    if( !h->knn ) { h->knn.reset( new KnnGrid() ); h->knn->build( h->host, h->radius ); }
  }
  Timed t( 2, d->n_query_pts );
  size_t total = 0;
  for( size_t i = 0; i < d->n_query_pts; ++i )
  {
    float q[3] = { d->query_pts[(size_t)h->dim * i], d->query_pts[(size_t)h->dim * i + 1], h->dim == 3 ? d->query_pts[3 * i + 2] : 0.0f };
    const size_t c = h->knn->search( q, d->k, d->distances_sq + i * d->k, d->indices + i * d->k );
    if( d->n_neighbors ) d->n_neighbors[i] = c;
    total += c;
  }
  return total;
}

// ---- icp ----------------------------------------------------------------------------------

float icp_align( rsd_vec3_t* pts1, rsd_vec3_t* nor1, int32_t n_pts1, rsd_vec3_t* pts2, rsd_vec3_t* nor2, int32_t n_pts2,
                 rsd_mat4_t* T1, rsd_mat4_t T2, float max_dist, float max_angle, bool verbose )
{
  // the reference builds its grids with radius = max_dist (icp.h:436-437); the cell size does not
  // change results here, so the clouds use the density-derived cell (fastest, and shared with the
  // score / label entry points through the cache)
  Timed timed( 4, (unsigned long long)( n_pts1 > 0 ? n_pts1 : 0 ) );
  const CloudRef src = cached_cloud( pts1, nor1, n_pts1, -1.0f );      // (held until the call returns: see CloudRef)
  const CloudRef tgt = cached_cloud( pts2, nor2, n_pts2, -1.0f );
  float err = 1e6f; int32_t iters = 0;
  if( !src || !tgt ) return err;
  rsd_mat4_t t = *T1;
  if( !verbose )
  {
    int rc = rs_hip_icp_align( src.get(), tgt.get(), t.data, T2.data, max_dist, max_angle, 100, 0, &err, &iters );
    if( rc ) { complain( "icp_align" ); return 1e6f; }
    *T1 = t;
    return err;
  }
  // verbose: the reference's lines (icp.h:440,482-486,498).  Its two times per iteration are the search's and the estimator's host
  // times; here an iteration is one piece of device work: its share of the call's time is printed as the first, 0 as the second.
  const auto t0 = std::chrono::steady_clock::now();
  float errs[100];
  int rc = rs_hip_icp_align_traced( src.get(), tgt.get(), t.data, T2.data, max_dist, max_angle, 100, 0, &err, &iters, errs );
  if( rc ) { complain( "icp_align" ); return 1e6f; }
  *T1 = t;
  const double ms = std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count();
  printf( " ICP: Search indexes build time: %fms\n", 0.0 );          // (the clouds' indices are cached device objects)
  float prev = 1e6f, md = max_dist;
  for( int i = 0; i < iters; ++i )
  {
    printf( " ICP: Iter %3d {Error: %7.5f; Err.Delta: %7.5f; Params: (%5.4f, %5.4f); Times: (%5.3fms, %5.3fms)}\n",
            i, errs[i], fabsf( prev - errs[i] ), md, max_angle, ms / ( iters > 0 ? iters : 1 ), 0.0 );
    prev = errs[i];
    const double nd = md * 0.95; md = (float)( nd > 0.05 ? nd : 0.05 );          // icp.h:493
  }
  printf( " ICP: Full time to estimate transform: %fms\n", ms );
  return err;
}

// lib/rs/icp.h:153-206: the point-to-point estimator's body is commented out in the reference — it returns 1.0f and leaves *T1
// alone (and is unreachable from icp_align, :473).  Exported so that a TU which names it still links against the shim.
float icp_estimate_rigid_xform_pt2pt( rsd_vec3_t*, rsd_vec3_t*, float*, int32_t, rsd_mat4_t* ) { return 1.0f; }

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
  const CloudRef src = cached_cloud( pts1, nor1, n_pts1, -1.0f );
  const CloudRef tgt = cached_cloud( pts2, nor2, n_pts2, -1.0f );
  if( !src || !tgt ) return;
  if( rs_hip_icp_find_corrs( src.get(), tgt.get(), T1.data, T2.data, max_dist, max_angle, (float*)*corr_pts1, (float*)*corr_nor1,
                             (float*)*corr_pts2, (float*)*corr_nor2, *weights, n_corrs ) )
  { complain( "icp_find_corrs" ); *n_corrs = 0; }
}

// ---- the two C++-linkage consumers, flattened ----------------------------------------------

int rsd_alignment_scores( const rsd_vec3_t* obj_pos, const rsd_vec3_t* obj_nor, int32_t n_obj,
                          const rsd_vec3_t* scn_pos, const rsd_vec3_t* scn_nor, int32_t n_scn,
                          const rsd_mat4_t* xforms, int32_t n_poses, float search_radius, int32_t max_n_neigh, float* scores )
{
  // (the reference's level grids use radius 0.05 -> cell 0.10, lib/rs/rs_pointcloud.h:862)
  const CloudRef obj = cached_cloud( obj_pos, obj_nor, n_obj, -1.0f );
  const CloudRef scn = cached_cloud( scn_pos, scn_nor, n_scn, -1.0f );
  if( !obj || !scn ) return RS_HIP_E_RUNTIME;
  int rc = rs_hip_alignment_scores( obj.get(), scn.get(), (const float*)xforms, n_poses, search_radius, max_n_neigh, scores );
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
  const CloudRef scn = cached_cloud( scn_pos, scn_nor, n_scn, -1.0f );
  if( !scn ) return RS_HIP_E_RUNTIME;
  // every placement's cloud is HELD for the call: an arrangement may have more distinct objects than the cache has entries
  // (the library takes up to 127 placements), and evicting one of them — or the scene — must not free what the call is using
  std::vector<CloudRef> held( (size_t)( n_plc > 0 ? n_plc : 0 ) );
  std::vector<const rs_hip_cloud_t*> objs( held.size() );
  for( int i = 0; i < n_plc; ++i )
  {
    held[i] = cached_cloud( obj_pos[i], obj_nor[i], obj_n[i], -1.0f );
    if( !held[i] ) return RS_HIP_E_RUNTIME;
    objs[i] = held[i].get();
  }
  // (the reference frees its min_dists before it returns, rs_pointcloud_filters.cpp:871-872: nobody wants them back — round 6: no 4 MB
  //  vector, no second download)
  int rc = rs_hip_arrangement_to_labels( scn.get(), (const float*)poses, objs.data(), is_static, class_idx, n_plc, radius,
                                         prioritize_static ? 1 : 0, labels, nullptr, sorted_order );
  if( rc ) complain( "arrangement_to_labels" );
  return rc;
}

int rsd_arrangement_to_ids( const rsd_vec3_t* scn_pos, const rsd_vec3_t* scn_nor, int32_t n_scn,
                            const rsd_vec3_t* const* obj_pos, const rsd_vec3_t* const* obj_nor, const int32_t* obj_n,
                            const rsd_mat4_t* poses, const int32_t* is_static, const int32_t* class_idx, const int32_t* uidx, int32_t n_plc,
                            float radius, bool prioritize_static, int32_t unlabelled_class_idx,
                            int32_t* class_ids, int32_t* instance_ids )
{
  const CloudRef scn = cached_cloud( scn_pos, scn_nor, n_scn, -1.0f );
  if( !scn ) return RS_HIP_E_RUNTIME;
  // every placement's cloud is HELD for the call: an arrangement may have more distinct objects than the cache has entries
  // (the library takes up to 127 placements), and evicting one of them — or the scene — must not free what the call is using
  std::vector<CloudRef> held( (size_t)( n_plc > 0 ? n_plc : 0 ) );
  std::vector<const rs_hip_cloud_t*> objs( held.size() );
  for( int i = 0; i < n_plc; ++i )
  {
    held[i] = cached_cloud( obj_pos[i], obj_nor[i], obj_n[i], -1.0f );
    if( !held[i] ) return RS_HIP_E_RUNTIME;
    objs[i] = held[i].get();
  }
  int rc = rs_hip_arrangement_to_ids( scn.get(), (const float*)poses, objs.data(), is_static, class_idx, uidx, n_plc, radius,
                                      prioritize_static ? 1 : 0, unlabelled_class_idx, class_ids, instance_ids, nullptr, nullptr, nullptr );
  if( rc ) complain( "arrangement_to_ids" );
  return rc;
}

int64_t rsd_compute_neighborhood( const rsd_vec3_t* pos, const rsd_vec3_t* nor, int32_t n,
                                  int32_t max_nn, float radius_sq, float dist_exp, float angle_exp,
                                  int32_t* idx1, int32_t* idx2, float* weight )
{
  const CloudRef c = cached_cloud( pos, nor, n, -1.0f );
  if( !c ) return RS_HIP_E_RUNTIME;
  int64_t n_edges = 0;
  int rc = rs_hip_compute_neighborhood( c.get(), max_nn, radius_sq, dist_exp, angle_exp, idx1, idx2, weight,
                                        (int64_t)n * max_nn, &n_edges );
  if( rc ) { complain( "compute_neighborhood" ); return rc; }
  return n_edges;
}

int32_t rsd_level_poisson( const rsd_vec3_t* pos, int32_t n, float voxel_size, int32_t level, int32_t* sample_idx )
{
  const CloudRef c = cached_cloud( pos, nullptr, n, -1.0f );
  if( !c ) return RS_HIP_E_RUNTIME;
  size_t max_n_neigh = (size_t)( 1024 * ( ( level ) / (float)( 5 - 1 ) ) );      // rs_pointcloud.h:995 (RSPC_N_LEVELS = 5)
  if( !max_n_neigh ) max_n_neigh = 256;                                           // :996
  int32_t n_samples = 0;
  int rc = rs_hip_level_samples( c.get(), voxel_size, (int32_t)max_n_neigh, sample_idx, &n_samples, nullptr );
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
  std::vector<CloudRef> held( (size_t)( n_plc > 0 ? n_plc : 0 ) );
  std::vector<const rs_hip_cloud_t*> objs( held.size() );
  for( int i = 0; i < n_plc; ++i )
  {
    if( !is_static[i] ) held[i] = cached_cloud( obj_pos[i], nullptr, obj_n[i], -1.0f );
    objs[i] = held[i].get();
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
