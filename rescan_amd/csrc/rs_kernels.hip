// librescan_hip device code — written for gfx950 (MI355X, wave64) only.
//
// One primitive underlies all three consumers of the reference's
// msh_hash_grid_radius_search (lib/msh/msh_hash_grid.h:1090-1259): a WAVE of 64 spatially
// adjacent query points sweeps the grid cells overlapping its bounding box (+radius).
// Candidate points of those cells stream HBM/L2 -> LDS in 64-record chunks (one coalesced
// 1 KiB global load per wave), and every lane tests the same candidate at the same time
// through an LDS broadcast read (ds_read_b128, all lanes one address: conflict-free).
// The four waves of a workgroup share ONE tile of 64 queries and split its candidate chunks
// round-robin, so the time of the heaviest tile (cluttered corners hold ~6x the average
// candidate count) is cut by four; they meet once, at the end, to merge their per-lane
// results through LDS.  Inside the sweep waves never synchronise with each other.
//
// Arithmetic that decides *which* neighbour wins is kept in the reference's own order and
// precision (the file is compiled with -ffp-contract=off):
//   dist² = vx*vx + vy*vy + vz*vz with v = candidate - query   (msh_hash_grid.h:852-855)
//   in-range test dist² < (float)((double)r*(double)r)          (msh_hash_grid.h:857,1111)
//   transforms m0*x + m4*y + m8*z + w*m12                       (msh_vec_math.h:1554-1561)
// Neighbour order is (dist², original index) — the reference's order among exactly equal
// distances is an accident of its quicksort/heap and is not reproduced.
//
// The reference keeps the K nearest in a heap and lets each consumer walk them in
// ascending order until a normal gate passes.  That is restated as: c = the nearest
// candidate that passes the gate; accept c iff fewer than K candidates are closer than c.
// It needs no per-lane heap, does not diverge, and costs the same for K = 16, 32 or 64.

#include "rs_device.h"
#include <cfloat>
#include <climits>

namespace rs {

#define WAVE 64
#define BLOCK 256
#define WAVES_PER_BLOCK (BLOCK / WAVE)
#define TW WAVES_PER_BLOCK   // waves cooperating on one tile

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ int uni( int v ) { return __builtin_amdgcn_readfirstlane( v ); }

// Order LDS traffic of one wave: the LDS executes a wave's DS instructions in issue order,
// so a store by one lane is visible to a later load by another lane of the SAME wave; the
// only thing needed is that the compiler keeps the program order.
__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float wave_min( float v ) {
#pragma unroll
  for( int o = 32; o > 0; o >>= 1 ) v = fminf( v, __shfl_xor( v, o ) );
  return v;
}
__device__ __forceinline__ float wave_max( float v ) {
#pragma unroll
  for( int o = 32; o > 0; o >>= 1 ) v = fmaxf( v, __shfl_xor( v, o ) );
  return v;
}
__device__ __forceinline__ double wave_sum( double v ) {
#pragma unroll
  for( int o = 32; o > 0; o >>= 1 ) v += __shfl_xor( v, o );
  return v;
}

// msh_mat4_vec3_mul (msh_vec_math.h:1554-1561); w = 1 for points, 0 for directions.
__device__ __forceinline__ void xform3( const Xform& M, float x, float y, float z, float w,
                                        float& ox, float& oy, float& oz )
{
  ox = M.m[0] * x + M.m[4] * y + M.m[ 8] * z + w * M.m[12];
  oy = M.m[1] * x + M.m[5] * y + M.m[ 9] * z + w * M.m[13];
  oz = M.m[2] * x + M.m[6] * y + M.m[10] * z + w * M.m[14];
}

// Cells of one axis that can hold a point within `r` of the interval [lo,hi].  Binning of
// the stored points (host, rs_api.hip) and this range use the same float expression; the
// 0.01-cell margin is far above the rounding error of either, so the range is a superset.
__device__ __forceinline__ void axis_range( float lo, float hi, float r, float gmin, float inv_cell, int dim,
                                            int& c0, int& c1 )
{
  float a = floorf( ( lo - r - gmin ) * inv_cell - 0.01f );
  float b = floorf( ( hi + r - gmin ) * inv_cell + 0.01f );
  a = fmaxf( a, 0.0f );
  b = fminf( b, (float)( dim - 1 ) );
  c0 = (int)a;
  c1 = ( b >= a ) ? (int)b : -1;      // empty -> c1 < c0
}

struct CellBox { int x0, x1, y0, y1, z0, z1; bool empty; };

// Bounding box of the wave's (active) queries -> cell box of grid g, wave-uniform.
__device__ __forceinline__ CellBox wave_cell_box( const GridView& g, bool active, float qx, float qy, float qz, float r )
{
  const float big = FLT_MAX;
  float lx = wave_min( active ? qx : big ),  hx = wave_max( active ? qx : -big );
  float ly = wave_min( active ? qy : big ),  hy = wave_max( active ? qy : -big );
  float lz = wave_min( active ? qz : big ),  hz = wave_max( active ? qz : -big );
  CellBox b;
  axis_range( lx, hx, r, g.minx, g.inv_cell, g.w, b.x0, b.x1 );
  axis_range( ly, hy, r, g.miny, g.inv_cell, g.h, b.y0, b.y1 );
  axis_range( lz, hz, r, g.minz, g.inv_cell, g.d, b.z0, b.z1 );
  b.x0 = uni( b.x0 ); b.x1 = uni( b.x1 ); b.y0 = uni( b.y0 ); b.y1 = uni( b.y1 ); b.z0 = uni( b.z0 ); b.z1 = uni( b.z1 );
  b.empty = ( b.x1 < b.x0 ) || ( b.y1 < b.y0 ) || ( b.z1 < b.z0 ) || !( hx >= lx );
  return b;
}

// Stream every candidate of the cell box through this wave's LDS slice and call
// f( P, j, slot ) for each: P = {x,y,z,bitcast(index)}, j = position inside the chunk (so
// the matching normal is sn[j]), slot = position in the sorted cloud.  j and slot are
// wave-uniform.  A chunk is always processed as 64 slots rounded up to a multiple of 4:
// lanes beyond the span store a sentinel at +FLT_MAX whose dist² is +inf, so it can never be
// "within the radius" and the inner loop needs no remainder handling (f may be called with
// such sentinels; slot is then >= the span end and must only be used when P qualified).
// Of the chunks met in sweep order, a wave takes those with (chunk number % n_share) == share.
template <bool WITH_NOR, class F>
__device__ __forceinline__ void sweep_box( const GridView& g, const CellBox& b, float4* sp, float4* sn, int lane,
                                           int share, int n_share, F&& f )
{
  int chunk = 0;
  for( int z = b.z0; z <= b.z1; ++z )
  {
    for( int y = b.y0; y <= b.y1; ++y )
    {
      const int row = ( z * g.h + y ) * g.w;
      const uint32_t s = (uint32_t)uni( (int)g.cell_start[row + b.x0] );
      const uint32_t e = (uint32_t)uni( (int)g.cell_start[row + b.x1 + 1] );
      for( uint32_t c0 = s; c0 < e; c0 += WAVE )
      {
        if( ( chunk++ % n_share ) != share ) continue;
        const uint32_t cnt = ( e - c0 < WAVE ) ? ( e - c0 ) : WAVE;
        float4 P = make_float4( FLT_MAX, FLT_MAX, FLT_MAX, 0.0f ), N = make_float4( 0.0f, 0.0f, 0.0f, 0.0f );
        if( (uint32_t)lane < cnt )
        {
          P = g.pos[c0 + lane];
          if( WITH_NOR ) N = g.nor[c0 + lane];
        }
        sp[lane] = P;
        if( WITH_NOR ) sn[lane] = N;
        wave_lds_fence();
        const uint32_t cnt4 = ( cnt + 3u ) & ~3u;
        for( uint32_t j = 0; j < cnt4; j += 4 )
        {
          const float4 P0 = sp[j], P1 = sp[j + 1], P2 = sp[j + 2], P3 = sp[j + 3];
          f( P0, (int)j, (int)( c0 + j ) );
          f( P1, (int)j + 1, (int)( c0 + j + 1 ) );
          f( P2, (int)j + 2, (int)( c0 + j + 2 ) );
          f( P3, (int)j + 3, (int)( c0 + j + 3 ) );
        }
        wave_lds_fence();
      }
    }
  }
}

// (dist², index) lexicographic "a before b"
__device__ __forceinline__ bool lex_less( float d2a, int ia, float d2b, int ib )
{
  return ( d2a < d2b ) || ( d2a == d2b && ia < ib );
}

// Result of a gated search for one query.
struct Match { float d2; int idx; float dot; int slot; bool found; };

// Per-workgroup LDS: candidate staging of each wave + the per-lane merge slots.
struct TileLds
{
  float4 pos[TW][WAVE];
  float4 nor[TW][WAVE];
  float  m_d2[TW][WAVE];
  int    m_idx[TW][WAVE];
  float  m_dot[TW][WAVE];
  int    m_slot[TW][WAVE];
  int    m_cnt[TW][WAVE];
};

// Nearest candidate within the radius whose normal passes  tmin <= max(dot,0) <= 1, accepted
// only if fewer than K candidates (of any normal) precede it in (dist², index) order.
// That is the reference's "first normal-compatible entry of the K-nearest list"
// (lib/rs/icp.h:361-380, apps/pose_proposal/pose_proposal.cpp:127-147).
// All TW waves of the workgroup call this with the SAME queries; each sweeps its share of the
// candidate chunks and all return the same merged result.
__device__ __forceinline__ Match gated_search( const GridView& g, bool active,
                                               float qx, float qy, float qz, float nx, float ny, float nz,
                                               float radius, float radius_sq, float tmin, int K,
                                               TileLds& lds, int wib, int lane )
{
  Match m; m.d2 = INFINITY; m.idx = INT_MAX; m.dot = 0.0f; m.slot = -1; m.found = false;
  CellBox box = wave_cell_box( g, active, qx, qy, qz, radius );
  if( box.empty ) return m;                       // identical in every wave of the workgroup

  int seen_closer = 0;   // candidates that were no farther than this wave's best-so-far when met
  sweep_box<true>( g, box, lds.pos[wib], lds.nor[wib], lane, wib, TW, [&]( float4 P, int j, int slot )
  {
    float vx = P.x - qx, vy = P.y - qy, vz = P.z - qz;
    float d2 = vx * vx + vy * vy + vz * vz;
    // cheap superset of "precedes the best so far": ties are settled inside the rare branch
    const bool maybe = active & ( d2 < radius_sq ) & ( d2 <= m.d2 );
    seen_closer += maybe ? 1 : 0;
    if( __any( maybe ) )
    {
      const int idx = __float_as_int( P.w );
      float4 N = lds.nor[wib][j];
      float dot = N.x * nx + N.y * ny + N.z * nz;         // msh_vec3_dot( m, n )
      float dc = dot > 0.0f ? dot : 0.0f;                 // msh_max( dot, 0.0f )
      if( maybe && lex_less( d2, idx, m.d2, m.idx ) && dc >= tmin && dc <= 1.0f )
      { m.d2 = d2; m.idx = idx; m.dot = dc; m.slot = slot; m.found = true; }
    }
  } );

  // merge the TW partial results of each lane
  lds.m_d2[wib][lane] = m.d2; lds.m_idx[wib][lane] = m.idx; lds.m_dot[wib][lane] = m.dot;
  lds.m_slot[wib][lane] = m.found ? m.slot : -1; lds.m_cnt[wib][lane] = seen_closer;
  __syncthreads();
  int seen_total = 0;
  m.d2 = INFINITY; m.idx = INT_MAX; m.dot = 0.0f; m.slot = -1; m.found = false;
#pragma unroll
  for( int w = 0; w < TW; ++w )
  {
    seen_total += lds.m_cnt[w][lane];
    const float d = lds.m_d2[w][lane]; const int ix = lds.m_idx[w][lane]; const int sl = lds.m_slot[w][lane];
    if( sl >= 0 && lex_less( d, ix, m.d2, m.idx ) ) { m.d2 = d; m.idx = ix; m.dot = lds.m_dot[w][lane]; m.slot = sl; m.found = true; }
  }

  // Every candidate that precedes the final match was counted by the wave that met it (it was
  // no farther than that wave's then-best, which the final match precedes or equals), and so
  // was the match itself: seen_total - 1 >= rank.  Only when that bound does not settle
  // rank < K, count exactly.
  bool need_rank = m.found && ( seen_total - 1 >= K );
  if( __any( need_rank ) )                         // same decision in every wave (same data)
  {
    int rank = 0;
    sweep_box<false>( g, box, lds.pos[wib], lds.nor[wib], lane, wib, TW, [&]( float4 P, int, int )
    {
      float vx = P.x - qx, vy = P.y - qy, vz = P.z - qz;
      float d2 = vx * vx + vy * vy + vz * vz;
      int idx = __float_as_int( P.w );
      rank += ( need_rank & ( d2 < radius_sq ) & ( ( d2 < m.d2 ) | ( ( d2 == m.d2 ) & ( idx < m.idx ) ) ) ) ? 1 : 0;
    } );
    __syncthreads();                               // everyone is done reading the first merge
    lds.m_cnt[wib][lane] = rank;
    __syncthreads();
    rank = 0;
#pragma unroll
    for( int w = 0; w < TW; ++w ) rank += lds.m_cnt[w][lane];
    if( need_rank && rank >= K ) { m.found = false; m.slot = -1; }
  }
  __syncthreads();                                 // merge slots may be reused by the caller's next search
  return m;
}

// ------------------------------------------------------------------------------------------
// ICP: correspondence search  (lib/rs/icp.h:339-391)
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__( BLOCK ) void k_icp_corr( IcpLaunch L )
{
  __shared__ TileLds lds;
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int wib = threadIdx.x / WAVE;
  const int tile = blockIdx.x;                           // one workgroup per tile
  const int i = (int)L.src.tiles[tile] + lane;
  const bool active = i < (int)L.src.tiles[tile + 1];
  const int nq = L.src.n;

  Xform T1;
#pragma unroll
  for( int k = 0; k < 16; ++k ) T1.m[k] = L.T1[prob * 16 + k];

  float qx = 0, qy = 0, qz = 0, nx = 0, ny = 0, nz = 0;
  if( active )
  {
    float4 p = L.src.pos[i], n = L.src.nor[i];
    float tx, ty, tz;
    xform3( T1, p.x, p.y, p.z, 1.0f, tx, ty, tz );   xform3( L.T2i, tx, ty, tz, 1.0f, qx, qy, qz );
    xform3( T1, n.x, n.y, n.z, 0.0f, tx, ty, tz );   xform3( L.T2i, tx, ty, tz, 0.0f, nx, ny, nz );
  }
  Match m = gated_search( L.tgt, active, qx, qy, qz, nx, ny, nz, L.radius, L.radius_sq, L.gate_tmin, L.K,
                          lds, wib, lane );
  if( wib != 0 ) return;                                 // every wave holds the merged result; wave 0 writes it
  const size_t o = (size_t)prob * nq + i;
  if( active ) { L.m_slot[o] = m.found ? m.slot : -1; L.m_d2[o] = m.d2; L.m_dot[o] = m.dot; }

  // statistics of dist² over correspondences (msh_compute_mean/stddev, msh_std.h:1800-1825)
  double c = ( active && m.found ) ? 1.0 : 0.0;
  double s1 = ( active && m.found ) ? (double)m.d2 : 0.0;
  double s2 = ( active && m.found ) ? (double)( m.d2 * m.d2 ) : 0.0;
  c = wave_sum( c ); s1 = wave_sum( s1 ); s2 = wave_sum( s2 );
  if( lane == 0 )
  {
    double* out = L.corr_part + ( (size_t)prob * L.src.n_tiles + tile ) * 3;
    out[0] = c; out[1] = s1; out[2] = s2;
  }
}

// One block per problem: fixed-order sum of the per-wave partials -> n_corr, mean, stddev.
__global__ __launch_bounds__( BLOCK ) void k_icp_stats( IcpLaunch L )
{
  __shared__ double red[3][BLOCK];
  const int prob = blockIdx.x;
  if( L.active[prob] == 0 ) return;
  const int n_waves = L.src.n_tiles;
  const double* in = L.corr_part + (size_t)prob * n_waves * 3;
  double a = 0, b = 0, c = 0;
  for( int w = threadIdx.x; w < n_waves; w += BLOCK ) { a += in[3*w]; b += in[3*w+1]; c += in[3*w+2]; }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c;
  __syncthreads();
  for( int s = BLOCK / 2; s > 0; s >>= 1 )
  {
    if( threadIdx.x < s ) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; red[2][threadIdx.x] += red[2][threadIdx.x + s]; }
    __syncthreads();
  }
  if( threadIdx.x == 0 )
  {
    double n = red[0][0];
    float mean = (float)( red[1][0] / n );                 // sum / (float)n
    float sqm = (float)( red[2][0] / n );                  // sq_sum / (float)n
    float var = sqm - mean * mean;
    float sd = (float)sqrt( (double)var );                 // (float)sqrt( ... ), msh_std.h:1824
    double* st = L.stats + (size_t)prob * 4;
    st[0] = n; st[1] = mean; st[2] = sd; st[3] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// ICP: weights + normal-equation moments  (lib/rs/icp.h:387,393-402,210-252)
//
// The reference centres the correspondences on their weighted centroids c1, c2 and
// accumulates  Σw·c cᵀ, Σw·c nᵀ, Σw·n nᵀ, Σw·(c,n)·s, Σw·s²  with c = (p-c1)×n,
// s = ((p-c1)-(q-c2))·n.  All of those are polynomials in the UNcentred moments below, so
// one pass in fp64 suffices and the host finishes the algebra (rs_api.hip: icp_solve):
//   [0] Σw   [1..3] Σw·p   [4..6] Σw·q   [7..12] Σw·a aᵀ (xx,xy,xz,yy,yz,zz), a = p×n
//   [13..21] Σw·a nᵀ (row-major a_i n_j)   [22..27] Σw·n nᵀ   [28..30] Σw·a·e   [31..33] Σw·n·e
//   [34] Σw·e²,  e = (p-q)·n
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__( BLOCK ) void k_icp_moments( IcpLaunch L )
{
  __shared__ double red[WAVES_PER_BLOCK][ICP_NMOM];
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  Xform T1;
#pragma unroll
  for( int k = 0; k < 16; ++k ) T1.m[k] = L.T1[prob * 16 + k];
  const float sd = (float)L.stats[(size_t)prob * 4 + 2];
  const bool use_sd = sd > 0.000001;
  const float cut = 2.5f * sd;

  double acc[ICP_NMOM];
#pragma unroll
  for( int k = 0; k < ICP_NMOM; ++k ) acc[k] = 0.0;

  for( int i = blockIdx.x * BLOCK + threadIdx.x; i < L.src.n; i += gridDim.x * BLOCK )
  {
    const size_t o = (size_t)prob * L.src.n + i;
    const int slot = L.m_slot[o];
    if( slot < 0 ) continue;
    const float d2 = L.m_d2[o];
    float w;
    if( L.w_explicit ) { w = L.w_explicit[o]; }
    else
    {
      w = ( 1.0f - __fdiv_rn( d2, L.radius ) ) * L.m_dot[o];         // icp.h:387
      if( use_sd && d2 > cut ) w = 0.0f;                              // icp.h:396-401
    }
    float4 p4 = L.src.pos[i];
    float tx, ty, tz, px, py, pz;
    xform3( T1, p4.x, p4.y, p4.z, 1.0f, tx, ty, tz );
    xform3( L.T2i, tx, ty, tz, 1.0f, px, py, pz );
    const float4 q4 = L.tgt.pos[slot], n4 = L.tgt.nor[slot];
    const double W = w, p[3] = { px, py, pz }, q[3] = { q4.x, q4.y, q4.z }, n[3] = { n4.x, n4.y, n4.z };
    const double a[3] = { p[1] * n[2] - p[2] * n[1], p[2] * n[0] - p[0] * n[2], p[0] * n[1] - p[1] * n[0] };
    const double e = ( p[0] - q[0] ) * n[0] + ( p[1] - q[1] ) * n[1] + ( p[2] - q[2] ) * n[2];
    acc[0] += W;
    acc[1] += W * p[0]; acc[2] += W * p[1]; acc[3] += W * p[2];
    acc[4] += W * q[0]; acc[5] += W * q[1]; acc[6] += W * q[2];
    acc[7]  += W * a[0] * a[0]; acc[8]  += W * a[0] * a[1]; acc[9]  += W * a[0] * a[2];
    acc[10] += W * a[1] * a[1]; acc[11] += W * a[1] * a[2]; acc[12] += W * a[2] * a[2];
#pragma unroll
    for( int r = 0; r < 3; ++r )
#pragma unroll
      for( int c = 0; c < 3; ++c ) acc[13 + 3 * r + c] += W * a[r] * n[c];
    acc[22] += W * n[0] * n[0]; acc[23] += W * n[0] * n[1]; acc[24] += W * n[0] * n[2];
    acc[25] += W * n[1] * n[1]; acc[26] += W * n[1] * n[2]; acc[27] += W * n[2] * n[2];
    acc[28] += W * a[0] * e; acc[29] += W * a[1] * e; acc[30] += W * a[2] * e;
    acc[31] += W * n[0] * e; acc[32] += W * n[1] * e; acc[33] += W * n[2] * e;
    acc[34] += W * e * e;
  }
  const int lane = threadIdx.x & ( WAVE - 1 ), wib = threadIdx.x / WAVE;
#pragma unroll
  for( int k = 0; k < ICP_NMOM; ++k ) { double v = wave_sum( acc[k] ); if( lane == 0 ) red[wib][k] = v; }
  __syncthreads();
  if( threadIdx.x < ICP_NMOM )
  {
    double v = 0.0;
    for( int w = 0; w < WAVES_PER_BLOCK; ++w ) v += red[w][threadIdx.x];
    L.mom_part[( (size_t)prob * L.n_mom_blocks + blockIdx.x ) * ICP_NMOM + threadIdx.x] = v;
  }
}

// fixed-order tree over the per-block partials: one workgroup per (moment, problem)
__global__ __launch_bounds__( BLOCK ) void k_icp_moments_final( IcpLaunch L )
{
  __shared__ double red[BLOCK];
  const int prob = blockIdx.y, k = blockIdx.x;
  if( L.active[prob] == 0 ) return;
  const double* in = L.mom_part + (size_t)prob * L.n_mom_blocks * ICP_NMOM;
  double v = 0.0;
  for( int b = threadIdx.x; b < L.n_mom_blocks; b += BLOCK ) v += in[(size_t)b * ICP_NMOM + k];
  red[threadIdx.x] = v;
  __syncthreads();
  for( int s = BLOCK / 2; s > 0; s >>= 1 ) { if( threadIdx.x < s ) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if( threadIdx.x == 0 ) L.moments[(size_t)prob * ICP_NMOM + k] = red[0];
}

void launch_icp_corr( const IcpLaunch& L, hipStream_t st )
{
  dim3 grid( L.src.n_tiles, L.n_prob );
  hipLaunchKernelGGL( k_icp_corr, grid, dim3( BLOCK ), 0, st, L );
}
void launch_icp_stats( const IcpLaunch& L, hipStream_t st )
{
  hipLaunchKernelGGL( k_icp_stats, dim3( L.n_prob ), dim3( BLOCK ), 0, st, L );
}
void launch_icp_moments( const IcpLaunch& L, hipStream_t st )
{
  hipLaunchKernelGGL( k_icp_moments, dim3( L.n_mom_blocks, L.n_prob ), dim3( BLOCK ), 0, st, L );
  hipLaunchKernelGGL( k_icp_moments_final, dim3( ICP_NMOM, L.n_prob ), dim3( BLOCK ), 0, st, L );
}

// ------------------------------------------------------------------------------------------
// Alignment score  (apps/pose_proposal/pose_proposal.cpp:93-158), all poses in one launch
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__( BLOCK ) void k_score( ScoreLaunch L )
{
  __shared__ TileLds lds;
  const int pose = blockIdx.y;
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int wib = threadIdx.x / WAVE;
  const int tile = blockIdx.x;
  const int n_tiles = L.obj.n_tiles;
  const int i = (int)L.obj.tiles[tile] + lane;
  const bool active = i < (int)L.obj.tiles[tile + 1];

  Xform X;
#pragma unroll
  for( int k = 0; k < 16; ++k ) X.m[k] = L.poses[pose * 16 + k];
  float qx = 0, qy = 0, qz = 0, nx = 0, ny = 0, nz = 0;
  if( active )
  {
    float4 p = L.obj.pos[i], n = L.obj.nor[i];
    xform3( X, p.x, p.y, p.z, 1.0f, qx, qy, qz );      // :110
    xform3( X, n.x, n.y, n.z, 0.0f, nx, ny, nz );      // :111
  }
  const float radius = (float)L.sigma;
  Match m = gated_search( L.scene, active, qx, qy, qz, nx, ny, nz, radius, L.radius_sq, L.gate_tmin, L.K,
                          lds, wib, lane );
  if( wib != 0 ) return;
  double s = 0.0;
  if( active && m.found )
  {
    const double angle = acos( (double)m.dot );                                   // :140
    const double normals_compat = exp( -( angle * angle ) / ( 2.0 * 0.5 * 0.5 ) ); // :149
    const double dist_compat = exp( -(double)m.d2 / ( 2.0 * L.sigma * L.sigma ) ); // :150, :36-40
    s = 0.05 * normals_compat + ( 1.0 - 0.05 ) * dist_compat;                      // :102-103,151
  }
  s = wave_sum( s );
  if( lane == 0 ) L.part[(size_t)pose * n_tiles + tile] = s;
}

// fixed-order sum over tiles, / n, narrowed to float (:156-157)
__global__ __launch_bounds__( BLOCK ) void k_score_final( ScoreLaunch L )
{
  __shared__ double red[BLOCK];
  const int pose = blockIdx.x;
  const int n_tiles = L.obj.n_tiles;
  const double* in = L.part + (size_t)pose * n_tiles;
  double a = 0.0;
  for( int t = threadIdx.x; t < n_tiles; t += BLOCK ) a += in[t];
  red[threadIdx.x] = a;
  __syncthreads();
  for( int s = BLOCK / 2; s > 0; s >>= 1 ) { if( threadIdx.x < s ) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if( threadIdx.x == 0 ) L.scores[pose] = (float)( red[0] / (double)L.obj.n );
}

void launch_score( const ScoreLaunch& L, hipStream_t st )
{
  dim3 grid( L.obj.n_tiles, L.n_poses );
  hipLaunchKernelGGL( k_score, grid, dim3( BLOCK ), 0, st, L );
  hipLaunchKernelGGL( k_score_final, dim3( L.n_poses ), dim3( BLOCK ), 0, st, L );
}

// ------------------------------------------------------------------------------------------
// Label transfer  (lib/rs/rs_pointcloud_filters.cpp:738-778)
// Every scene point carries its own (min_dist, label) chain through the placements, in
// order, so one launch covers the whole loop; a wave skips a placement outright when its
// 64 points, moved into the object's frame, miss the object's grid.
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ void unit3( float& x, float& y, float& z )
{
  // msh_vec3_normalize (msh_vec_math.h:868): 1.0f / sqrtf(x*x + y*y + z*z), three multiplies
  float inv = __fdiv_rn( 1.0f, __fsqrt_rn( x * x + y * y + z * z ) );
  x = x * inv; y = y * inv; z = z * inv;
}

__global__ __launch_bounds__( BLOCK ) void k_label( LabelLaunch L )
{
  __shared__ TileLds lds;
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int wib = threadIdx.x / WAVE;
  const int tile = blockIdx.x;
  const int i = (int)L.scene.tiles[tile] + lane;
  const bool active = i < (int)L.scene.tiles[tile + 1];
  float4 p = make_float4( 0, 0, 0, 0 ), n = make_float4( 0, 0, 0, 0 );
  if( active ) { p = L.scene.pos[i]; n = L.scene.nor[i]; }
  const int orig = __float_as_int( p.w );

  float best_min = 1e9f;
  int label = 0;
  if( active && L.min_d ) { best_min = L.min_d[orig]; label = L.labels[orig]; }

  for( int k = 0; k < L.n_pl; ++k )
  {
    const PlacementDev& pl = L.pl[k];
    float qx, qy, qz;
    xform3( pl.inv, p.x, p.y, p.z, 1.0f, qx, qy, qz );                         // :755
    CellBox box = wave_cell_box( pl.g, active, qx, qy, qz, pl.radius );        // same in all TW waves
    float bd2 = INFINITY; int bidx = INT_MAX, bslot = -1;
    if( !box.empty )
    {
      const float r2 = pl.radius_sq;
      sweep_box<false>( pl.g, box, lds.pos[wib], nullptr, lane, wib, TW, [&]( float4 P, int, int slot )
      {
        float vx = P.x - qx, vy = P.y - qy, vz = P.z - qz;
        float d2 = vx * vx + vy * vy + vz * vz;
        int idx = __float_as_int( P.w );
        if( active & ( d2 < r2 ) & ( ( d2 < bd2 ) | ( ( d2 == bd2 ) & ( idx < bidx ) ) ) ) { bd2 = d2; bidx = idx; bslot = slot; }
      } );
      lds.m_d2[wib][lane] = bd2; lds.m_idx[wib][lane] = bidx; lds.m_slot[wib][lane] = bslot;
      __syncthreads();
      bd2 = INFINITY; bidx = INT_MAX; bslot = -1;
#pragma unroll
      for( int w = 0; w < TW; ++w )
      {
        const float d = lds.m_d2[w][lane]; const int ix = lds.m_idx[w][lane]; const int sl = lds.m_slot[w][lane];
        if( sl >= 0 && lex_less( d, ix, bd2, bidx ) ) { bd2 = d; bidx = ix; bslot = sl; }
      }
      __syncthreads();
    }
    // :762-775 — found, strictly closer than the running minimum, and within 70° (either sign)
    bool ok = false;
    if( active && bslot >= 0 && ( L.rows != nullptr || bd2 < best_min ) )
    {
      float n1x, n1y, n1z;
      xform3( pl.nmat, n.x, n.y, n.z, 0.0f, n1x, n1y, n1z );                   // :766
      float4 m4 = pl.g.nor[bslot];
      float n2x = m4.x, n2y = m4.y, n2z = m4.z;
      unit3( n1x, n1y, n1z ); unit3( n2x, n2y, n2z );
      float dot = fabsf( n1x * n2x + n1y * n2y + n1z * n2z );                  // :769
      ok = ( dot >= L.gate_tmin ) && ( dot <= 1.0f );
    }
    if( L.rows ) { if( active && wib == 0 ) L.rows[(size_t)k * L.scene.n + orig] = ok ? bd2 : INFINITY; }
    else if( ok ) { best_min = bd2; label = L.label_base + k + 1; }
  }
  if( active && wib == 0 && L.min_d ) { L.min_d[orig] = best_min; L.labels[orig] = (int8_t)label; }
}

void launch_label( const LabelLaunch& L, hipStream_t st )
{
  hipLaunchKernelGGL( k_label, dim3( L.scene.n_tiles ), dim3( BLOCK ), 0, st, L );
}

// ------------------------------------------------------------------------------------------
// Generic rows: the k nearest within the radius, ascending  (msh_hash_grid.h:1090-1259)
// Compatibility path for callers that want the whole neighbour list.  Selection by
// successive minima: pass t finds, per query, the smallest (dist², index) greater than the
// one found in pass t-1.  No per-lane storage, rows come out sorted.
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__( BLOCK ) void k_rows( RowsLaunch L )
{
  __shared__ float4 s_pos[WAVES_PER_BLOCK][WAVE];
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int wib = threadIdx.x / WAVE;
  const int wave = blockIdx.x * WAVES_PER_BLOCK + wib;
  const int n_waves = L.q.n_tiles;
  if( wave >= n_waves ) return;
  const int i = (int)L.q.tiles[wave] + lane;
  const bool active = i < (int)L.q.tiles[wave + 1];
  float4 q = make_float4( 0, 0, 0, 0 );
  if( active ) q = L.q.pos[i];
  const int orig = __float_as_int( q.w );
  CellBox box = wave_cell_box( L.tgt, active, q.x, q.y, q.z, L.radius );

  float pd2 = -1.0f; int pidx = -1;      // previous pick; dist² >= 0 so (-1,-1) precedes everything
  int count = 0;
  bool more = active && !box.empty;
  for( int t = 0; t < L.K; ++t )
  {
    if( !__any( more ) ) break;
    float bd2 = INFINITY; int bidx = INT_MAX;
    sweep_box<false>( L.tgt, box, s_pos[wib], nullptr, lane, 0, 1, [&]( float4 P, int, int )
    {
      float vx = P.x - q.x, vy = P.y - q.y, vz = P.z - q.z;
      float d2 = vx * vx + vy * vy + vz * vz;
      int idx = __float_as_int( P.w );
      const bool after_prev = ( pd2 < d2 ) | ( ( pd2 == d2 ) & ( pidx < idx ) );
      const bool before_best = ( d2 < bd2 ) | ( ( d2 == bd2 ) & ( idx < bidx ) );
      if( more & ( d2 < L.radius_sq ) & after_prev & before_best ) { bd2 = d2; bidx = idx; }
    } );
    if( more )
    {
      if( bidx != INT_MAX ) { L.d2[(size_t)orig * L.K + t] = bd2; L.idx[(size_t)orig * L.K + t] = bidx; pd2 = bd2; pidx = bidx; count++; }
      else more = false;
    }
  }
  if( active ) L.nn[orig] = count;
}

void launch_rows( const RowsLaunch& L, hipStream_t st )
{
  const int n_waves = L.q.n_tiles;
  dim3 grid( ( n_waves + WAVES_PER_BLOCK - 1 ) / WAVES_PER_BLOCK );
  hipLaunchKernelGGL( k_rows, grid, dim3( BLOCK ), 0, st, L );
}

} // namespace rs
