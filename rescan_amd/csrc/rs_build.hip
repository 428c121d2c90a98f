// Device-side construction of a cloud's two layouts (the MI355X counterpart of msh_hash_grid_init_3d,
// lib/msh/msh_hash_grid.h:388-541): bounds, the density-derived cell size, the cell-sorted target
// layout with its dense offset table, and the Hilbert-ordered, tiled query layout — from the caller's
// raw AoS arrays, uploaded once.  Sorting and scanning use hipCUB (rocPRIM) device primitives; the
// kernels around them are below.  Orchestration (buffers, the few host decisions) is in rs_api.hip.
//
// Nothing here decides a search result: results depend on neither the cell size, nor the order of
// points inside a cell, nor the tiling.  The one expression that must agree with the search kernels is
// the cell coordinate of a stored point (cell_of_dev; rs_search.h: axis_range covers it with a 0.01-cell margin).
#include "rs_device.h"
#include <hipcub/hipcub.hpp>
#include <cfloat>
#include <algorithm>

namespace rs {

#define B_BLOCK 256

// order-preserving float <-> unsigned encoding for atomicMin/atomicMax
__device__ __forceinline__ unsigned enc( float f ) { unsigned u = __float_as_uint( f ); return ( u & 0x80000000u ) ? ~u : ( u | 0x80000000u ); }

// out[0..2] = min xyz, out[3..5] = max xyz (encoded), out[6] = max |normal|² (plain bits; non-negative floats order as uints)
__global__ __launch_bounds__( B_BLOCK ) void k_build_bounds( const float* pos, const float* nor, int n, unsigned* out )
{
  float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX }, n2 = 0.0f;
  for( int i = blockIdx.x * B_BLOCK + threadIdx.x; i < n; i += gridDim.x * B_BLOCK )
  {
#pragma unroll
    for( int a = 0; a < 3; ++a ) { const float v = pos[3 * (size_t)i + a]; if( v < lo[a] ) lo[a] = v; if( v > hi[a] ) hi[a] = v; }   // NaN never wins, as on the host
    if( nor ) { const float x = nor[3 * (size_t)i], y = nor[3 * (size_t)i + 1], z = nor[3 * (size_t)i + 2]; const float l = x * x + y * y + z * z; if( !( l <= n2 ) ) n2 = l; }   // NaN/inf propagate: certificates then stay off
  }
  for( int o = 32; o > 0; o >>= 1 )
  {
#pragma unroll
    for( int a = 0; a < 3; ++a ) { lo[a] = fminf( lo[a], __shfl_down( lo[a], o ) ); hi[a] = fmaxf( hi[a], __shfl_down( hi[a], o ) ); }
    const float other = __shfl_down( n2, o ); if( !( other <= n2 ) ) n2 = other;
  }
  if( ( threadIdx.x & 63 ) == 0 )
  {
#pragma unroll
    for( int a = 0; a < 3; ++a ) { atomicMin( out + a, enc( lo[a] ) ); atomicMax( out + 3 + a, enc( hi[a] ) ); }
    if( n2 == n2 ) atomicMax( out + 6, __float_as_uint( n2 ) ); else out[7] = 1u;      // [7]: a NaN normal was seen
  }
}

// occupancy of a trial grid: one bit per cell
__global__ __launch_bounds__( B_BLOCK ) void k_build_mark( const float* pos, int n, float mx, float my, float mz, float inv, unsigned long long db, unsigned long long dc, uint32_t* bits )
{
  const int i = blockIdx.x * B_BLOCK + threadIdx.x;
  if( i >= n ) return;
  const unsigned long long a = (unsigned long long)fmaxf( 0.0f, floorf( ( pos[3 * (size_t)i] - mx ) * inv ) ),
                           b = (unsigned long long)fmaxf( 0.0f, floorf( ( pos[3 * (size_t)i + 1] - my ) * inv ) ),
                           c = (unsigned long long)fmaxf( 0.0f, floorf( ( pos[3 * (size_t)i + 2] - mz ) * inv ) );
  const unsigned long long id = ( a * db + b ) * dc + c;
  atomicOr( bits + ( id >> 5 ), 1u << ( id & 31 ) );
}

// cell coordinate of a stored point along one axis (the search kernels' axis_range covers this expression with a 0.01-cell margin)
__device__ __forceinline__ int cell_of_dev( float v, float gmin, float inv_cell, int dim )
{
  float c = floorf( ( v - gmin ) * inv_cell );
  if( !( c >= 0.0f ) ) c = 0.0f;
  if( c > (float)( dim - 1 ) ) c = (float)( dim - 1 );
  return (int)c;
}

__global__ __launch_bounds__( B_BLOCK ) void k_build_cellids( const float* pos, int n, float mx, float my, float mz, float inv_cell, int w, int h, int d,
                                                              uint32_t* cid, uint32_t* iota, uint32_t* counts )
{
  const int i = blockIdx.x * B_BLOCK + threadIdx.x;
  if( i >= n ) return;
  const int cx = cell_of_dev( pos[3 * (size_t)i], mx, inv_cell, w ), cy = cell_of_dev( pos[3 * (size_t)i + 1], my, inv_cell, h ), cz = cell_of_dev( pos[3 * (size_t)i + 2], mz, inv_cell, d );
  const uint32_t id = (uint32_t)( ( (size_t)cz * h + cy ) * w + cx );
  cid[i] = id; iota[i] = (uint32_t)i;
  atomicAdd( counts + id, 1u );
}

__global__ __launch_bounds__( B_BLOCK ) void k_build_gather( const float* pos, const float* nor, const uint32_t* order, int n, float4* spos, float4* snor )
{
  const int s = blockIdx.x * B_BLOCK + threadIdx.x;
  if( s >= n ) return;
  const uint32_t i = order[s];
  spos[s] = make_float4( pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2], __uint_as_float( i ) );
  if( nor ) snor[s] = make_float4( nor[3 * (size_t)i], nor[3 * (size_t)i + 1], nor[3 * (size_t)i + 2], 0.0f );
}

__global__ __launch_bounds__( B_BLOCK ) void k_build_count_runs( const uint32_t* sorted, int n, int* out )
{
  const int s = blockIdx.x * B_BLOCK + threadIdx.x;
  int c = ( s < n && ( s == 0 || sorted[s] != sorted[s - 1] ) ) ? 1 : 0;
  for( int o = 32; o > 0; o >>= 1 ) c += __shfl_down( c, o );
  if( ( threadIdx.x & 63 ) == 0 && c ) atomicAdd( out, c );
}

// Index of a cell on a 3-D Hilbert curve with `bits` bits per axis (Skilling's transpose form)
__device__ __forceinline__ uint32_t hilbert3_dev( uint32_t x, uint32_t y, uint32_t z, int bits )
{
  uint32_t X[3] = { x, y, z };
  const uint32_t M = 1u << ( bits - 1 );
  for( uint32_t Q = M; Q > 1; Q >>= 1 )
  {
    const uint32_t P = Q - 1;
#pragma unroll
    for( int i = 0; i < 3; ++i )
    {
      if( X[i] & Q ) X[0] ^= P;
      else { const uint32_t t = ( X[0] ^ X[i] ) & P; X[0] ^= t; X[i] ^= t; }
    }
  }
  X[1] ^= X[0]; X[2] ^= X[1];
  uint32_t t = 0;
  for( uint32_t Q = M; Q > 1; Q >>= 1 ) if( X[2] & Q ) t ^= Q - 1;
  X[0] ^= t; X[1] ^= t; X[2] ^= t;
  uint32_t h = 0;
  for( int b = bits - 1; b >= 0; --b )
#pragma unroll
    for( int i = 0; i < 3; ++i ) h = ( h << 1 ) | ( ( X[i] >> b ) & 1u );
  return h;
}

__global__ __launch_bounds__( B_BLOCK ) void k_build_hilbert( const float* pos, int n, float mx, float my, float mz, float scale, uint32_t* key, uint32_t* iota )
{
  const int i = blockIdx.x * B_BLOCK + threadIdx.x;
  if( i >= n ) return;
  const float mn[3] = { mx, my, mz };
  uint32_t c[3];
#pragma unroll
  for( int a = 0; a < 3; ++a )
  {
    float f = ( pos[3 * (size_t)i + a] - mn[a] ) * scale;
    if( !( f >= 0.0f ) ) f = 0.0f;
    if( f > 1023.0f ) f = 1023.0f;
    c[a] = (uint32_t)f;
  }
  key[i] = hilbert3_dev( c[0], c[1], c[2], 10 );
  iota[i] = (uint32_t)i;
}

// Tiling of the Hilbert sequence, greedy from the front: a tile ends after 64 points or when adding the next
// point would stretch its bounding box beyond max_extent on any axis (the curve crosses empty space there).
// Sequential as stated; in parallel: next[s] = where the tile that STARTS at s would end (independent for every
// s), then the tile starts are the nodes reachable from 0 along next[], marked by pointer doubling.
__global__ __launch_bounds__( 64 ) void k_build_tile_next( const float4* qpos, int n, float max_extent, uint32_t* next )
{
  __shared__ float px[128], py[128], pz[128];
  const int lane = threadIdx.x, base = blockIdx.x * 64;
  for( int k = lane; k < 128; k += 64 )
  {
    const int s = base + k;
    const float4 p = s < n ? qpos[s] : make_float4( 0, 0, 0, 0 );
    px[k] = p.x; py[k] = p.y; pz[k] = p.z;
  }
  __syncthreads();
  const int s = base + lane;
  if( s >= n ) return;
  float lx = px[lane], hx = lx, ly = py[lane], hy = ly, lz = pz[lane], hz = lz;
  int len = 1;
  const int most = min( 64, n - s );
  for( ; len < most; ++len )
  {
    const float x = px[lane + len], y = py[lane + len], z = pz[lane + len];
    const float nlx = fminf( lx, x ), nhx = fmaxf( hx, x ), nly = fminf( ly, y ), nhy = fmaxf( hy, y ), nlz = fminf( lz, z ), nhz = fmaxf( hz, z );
    if( ( nhx - nlx > max_extent ) | ( nhy - nly > max_extent ) | ( nhz - nlz > max_extent ) ) break;
    lx = nlx; hx = nhx; ly = nly; hy = nhy; lz = nlz; hz = nhz;
  }
  next[s] = (uint32_t)( s + len );        // == n for the last tile
}

// one doubling round: everything one jump from a marked node gets marked; the jumps double (out-of-place)
__global__ __launch_bounds__( B_BLOCK ) void k_build_tile_round( const uint32_t* jump, uint32_t* jump_out, uint32_t* mark, int n )
{
  const int s = blockIdx.x * B_BLOCK + threadIdx.x;
  if( s >= n ) return;
  const uint32_t j = jump[s];
  if( mark[s] && j < (uint32_t)n ) mark[j] = 1u;
  jump_out[s] = j < (uint32_t)n ? jump[j] : (uint32_t)n;
}

__global__ __launch_bounds__( B_BLOCK ) void k_build_tile_scatter( const uint32_t* flags, const uint32_t* scanned, int n, uint32_t* tiles )
{
  const int s = blockIdx.x * B_BLOCK + threadIdx.x;
  if( s < n && flags[s] ) tiles[scanned[s]] = (uint32_t)s;
}

static inline unsigned blocks_for( long long n ) { return (unsigned)std::max<long long>( 1, ( n + B_BLOCK - 1 ) / B_BLOCK ); }

void launch_build_bounds( const float* pos3, const float* nor3, int n, unsigned* out8, hipStream_t st )
{
  hipLaunchKernelGGL( k_build_bounds, dim3( std::min( 1024u, blocks_for( n ) ) ), dim3( B_BLOCK ), 0, st, pos3, nor3, n, out8 );
}
void launch_build_mark( const float* pos3, int n, const float mn[3], float inv, unsigned long long db, unsigned long long dc, uint32_t* bits, hipStream_t st )
{
  hipLaunchKernelGGL( k_build_mark, dim3( blocks_for( n ) ), dim3( B_BLOCK ), 0, st, pos3, n, mn[0], mn[1], mn[2], inv, db, dc, bits );
}
void launch_build_cellids( const float* pos3, int n, const float mn[3], float inv_cell, const int dims[3], uint32_t* cid, uint32_t* iota, uint32_t* counts, hipStream_t st )
{
  hipLaunchKernelGGL( k_build_cellids, dim3( blocks_for( n ) ), dim3( B_BLOCK ), 0, st, pos3, n, mn[0], mn[1], mn[2], inv_cell, dims[0], dims[1], dims[2], cid, iota, counts );
}
void launch_build_gather( const float* pos3, const float* nor3, const uint32_t* order, int n, float4* spos, float4* snor, hipStream_t st )
{
  hipLaunchKernelGGL( k_build_gather, dim3( blocks_for( n ) ), dim3( B_BLOCK ), 0, st, pos3, nor3, order, n, spos, snor );
}
void launch_build_count_runs( const uint32_t* sorted, int n, int* out, hipStream_t st )
{
  hipLaunchKernelGGL( k_build_count_runs, dim3( blocks_for( n ) ), dim3( B_BLOCK ), 0, st, sorted, n, out );
}
void launch_build_hilbert( const float* pos3, int n, const float mn[3], float scale, uint32_t* key, uint32_t* iota, hipStream_t st )
{
  hipLaunchKernelGGL( k_build_hilbert, dim3( blocks_for( n ) ), dim3( B_BLOCK ), 0, st, pos3, n, mn[0], mn[1], mn[2], scale, key, iota );
}
// flags[0..n) = 1 where a tile starts; jump_a / jump_b: two scratch arrays of n words
void launch_build_tile_flags( const float4* qpos, int n, float max_extent, uint32_t* flags, uint32_t* jump_a, uint32_t* jump_b, hipStream_t st )
{
  hipLaunchKernelGGL( k_build_tile_next, dim3( ( n + 63 ) / 64 ), dim3( 64 ), 0, st, qpos, n, max_extent, jump_a );
  (void)hipMemsetAsync( flags, 0, (size_t)n * 4, st );
  const uint32_t one = 1u;
  (void)hipMemcpyAsync( flags, &one, 4, hipMemcpyHostToDevice, st );
  // a marked node at distance d from 0 is reached after ceil(log2(d+1)) rounds; the chain has at most n nodes
  // (concurrent marks inside a round only ever ADD nodes of the chain: mark[] is closed under next[] by then or later)
  for( long long reach = 1; reach < (long long)n; reach <<= 1 )
  {
    hipLaunchKernelGGL( k_build_tile_round, dim3( blocks_for( n ) ), dim3( B_BLOCK ), 0, st, jump_a, jump_b, flags, n );
    std::swap( jump_a, jump_b );
  }
}
void launch_build_tile_scatter( const uint32_t* flags, const uint32_t* scanned, int n, uint32_t* tiles, hipStream_t st )
{
  hipLaunchKernelGGL( k_build_tile_scatter, dim3( blocks_for( n ) ), dim3( B_BLOCK ), 0, st, flags, scanned, n, tiles );
}

// by_orig[original index] = query slot (the inverse of the Hilbert permutation kept in qpos[s].w)
__global__ __launch_bounds__( B_BLOCK ) void k_build_inverse( const float4* qpos, int n, int* by_orig )
{
  const int s = blockIdx.x * B_BLOCK + threadIdx.x;
  if( s < n ) by_orig[__float_as_int( qpos[s].w )] = s;
}
void launch_build_inverse( const float4* qpos, int n, int* by_orig, hipStream_t st )
{
  hipLaunchKernelGGL( k_build_inverse, dim3( blocks_for( n ) ), dim3( B_BLOCK ), 0, st, qpos, n, by_orig );
}

size_t build_sort_temp_bytes( int n, int bits )
{
  size_t b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs( nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, bits );
  return b;
}
int build_sort_pairs( void* tmp, size_t bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, int n, int bits, hipStream_t st )
{
  return (int)hipcub::DeviceRadixSort::SortPairs( tmp, bytes, kin, kout, vin, vout, n, 0, bits, st );     // stable
}
size_t build_scan_temp_bytes( size_t n )
{
  size_t b = 0;
  (void)hipcub::DeviceScan::ExclusiveSum( nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n );
  return b;
}
int build_exclusive_scan( void* tmp, size_t bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t st )
{
  return (int)hipcub::DeviceScan::ExclusiveSum( tmp, bytes, in, out, (int)n, st );
}

} // namespace rs
