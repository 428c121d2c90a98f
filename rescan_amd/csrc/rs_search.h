// Search primitives shared by every kernel translation unit of librescan_hip (gfx950, wave64): wave helpers, cell boxes,
// the shell sweep, the candidate step and the three tile searches.  Everything here is __device__ __forceinline__.
#pragma once
// librescan_hip device code — written for gfx950 (MI355X, wave64) only.
//
// One primitive underlies all three consumers of the reference's
// msh_hash_grid_radius_search (lib/msh/msh_hash_grid.h:1090-1259): a WAVE owns a tile of up to
// 64 spatially adjacent query points (Hilbert order, rs_api.hip) and searches the grid cells
// around the tile's bounding box in EXPANDING SHELLS:
//
//   the tile's own cells first, then the box grown by one cell, two, four, ... up to the box grown
//   by the search radius; of each shell only the part within reach of a lane that is still unsettled.
//   After a shell a lane is settled when its best match lies closer than the nearest face of the
//   swept box that can still grow (nothing unseen can precede it), and the wave stops when all its
//   lanes are.  A tile whose lanes all start from a genuine candidate (ICP: last iteration's match,
//   or a point of the query's own cell) needs no shells at all: one sweep of the cells within those
//   candidates' distances settles it — a hundred or two candidates instead of everything within the
//   radius — and that sweep is done per row of 16 lanes, each row streaming only the cells its own lanes
//   reach (sweep_by_rows).  Tiles that stay unsettled are handed to a second kernel that gives each of them a whole
//   workgroup (coop_search); what cannot be bounded at all — a point with nothing to match — is
//   remembered from one ICP iteration to the next (icp_certificate).
//
// The row pieces of a shell (one or two x-intervals per (y,z) row of cells, each a contiguous
// span of the cell-sorted cloud) are gathered by the lanes in parallel, prefix-summed, and
// consumed as ONE flattened stream: every lane fetches "candidate number j" of the stream
// (binary search over the piece offsets), so each 64-record chunk staged in LDS is full.
// All lanes then test the same candidate at the same time through an LDS broadcast read
// (ds_read_b128, one address for the whole wave: conflict-free).  In phase A waves never synchronise
// with each other; the cooperative kernel merges its waves' results through LDS after every shell.
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
#include "rs_math.h"
#include <cfloat>
#include <climits>
#include <algorithm>

// Per-tile timers and counters (RS_HIP_DEBUG_CYCLES) exist only in the diagnostic build
// (tools/variant.sh dbg -DRS_DBG=1): in the production build DBG() is a constant null pointer and
// every diagnostic statement, array and argument folds away.
#ifndef RS_DBG
#define RS_DBG 0
#endif
#define DBG( L ) ( RS_DBG ? ( L ).dbg : (unsigned long long*)nullptr )

namespace rs {

#define WAVE 64
#define BLOCK 256
#define WAVES_PER_BLOCK (BLOCK / WAVE)
#define COOP_WAVES 4                 // waves that share one queued (cluttered) tile
#define COOP_BLOCK (COOP_WAVES * WAVE)

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ int uni( int v ) { return __builtin_amdgcn_readfirstlane( v ); }
// "does any lane ...": a compare of the wave's lane mask with zero on the scalar unit (HIP's __any goes through a VGPR: two VALU
// instructions per question, and the candidate step asks up to six per group of four candidates)
__device__ __forceinline__ bool wave_any( bool p ) { return __builtin_amdgcn_ballot_w64( p ) != 0ull; }

// Order LDS traffic of one wave: the LDS executes a wave's DS instructions in issue order,
// so a store by one lane is visible to a later load by another lane of the SAME wave; the
// only thing needed is that the compiler keeps the program order.
__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  __builtin_amdgcn_wave_barrier();
}

// Wave-wide reductions and the lane prefix sum through DPP lane moves (data-parallel primitives: one VALU instruction
// per step, no LDS round trip), instead of ds_bpermute shuffles, whose six dependent LDS round trips per reduction
// were ~15 % of phase A's instructions and a few microseconds of every tile's latency chain.
//   quad_perm [1,0,3,2] / [2,3,0,1]: lane ^ 1, lane ^ 2;  row_half_mirror / row_mirror: reversed within 8 / 16 lanes
//   (after the quad steps every lane of a row of 16 holds the row's result);  row_bcast:15 into rows 1 and 3, then
//   row_bcast:31 into rows 2 and 3: lane 63 ends with the whole wave's result and is read back as a scalar.
// (experiment: -DRS_CHAIN_PRIO=3 raises the issue priority of the ICP chain's waves over a batch kernel's that share their CUs)
#ifdef RS_CHAIN_PRIO
#define RS_CHAIN_SETPRIO() __builtin_amdgcn_s_setprio( RS_CHAIN_PRIO )
#else
#define RS_CHAIN_SETPRIO()
#endif
#define RS_DPP_QUAD_XOR1   0xB1
#define RS_DPP_QUAD_XOR2   0x4E
#define RS_DPP_ROW_SHR( n ) ( 0x110 + ( n ) )
#define RS_DPP_ROW_MIRROR  0x140
#define RS_DPP_HALF_MIRROR 0x141
#define RS_DPP_BCAST15     0x142
#define RS_DPP_BCAST31     0x143
// lanes whose source is outside the row / disabled by ROW_MASK keep `old`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f( float old, float v )
{
  return __int_as_float( __builtin_amdgcn_update_dpp( __float_as_int( old ), __float_as_int( v ), CTRL, ROW_MASK, 0xf, false ) );
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u( uint32_t old, uint32_t v )
{
  return (uint32_t)__builtin_amdgcn_update_dpp( (int)old, (int)v, CTRL, ROW_MASK, 0xf, false );
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d( double old, double v )
{
  const long long o = __double_as_longlong( old ), x = __double_as_longlong( v );
  const uint32_t lo = dpp_u<CTRL, ROW_MASK>( (uint32_t)o, (uint32_t)x );
  const uint32_t hi = dpp_u<CTRL, ROW_MASK>( (uint32_t)( (unsigned long long)o >> 32 ), (uint32_t)( (unsigned long long)x >> 32 ) );
  return __longlong_as_double( (long long)( ( (unsigned long long)hi << 32 ) | lo ) );
}
__device__ __forceinline__ float lane63( float v ) { return __int_as_float( __builtin_amdgcn_readlane( __float_as_int( v ), 63 ) ); }

// One instruction per step, in place (v = op(v moved, v); lanes of rows outside row_mask keep v).  The hazard
// recogniser does not see inside inline assembly, so the wait states are spelled out: 5 after a possible VALU write
// of EXEC before the first DPP read, 2 between a VALU write of a VGPR and a DPP read of it (CDNA3 ISA §4.5).
#define RS_DPP_REDUCE( OP, v )                                                              \
  asm volatile( "s_nop 4\n\t"                                                               \
                OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" \
                OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" \
                OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"     \
                OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"          \
                OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"        \
                OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"            \
                : "+v"( v ) )
// the same within every row of 16 lanes: all 16 end with their row's result
#define RS_DPP_ROW_REDUCE( OP, v )                                                          \
  asm volatile( "s_nop 4\n\t"                                                               \
                OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" \
                OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" \
                OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"     \
                OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"               \
                : "+v"( v ) )
__device__ __forceinline__ float row_min( float v ) { RS_DPP_ROW_REDUCE( "v_min_f32_dpp", v ); return v; }
__device__ __forceinline__ float row_max( float v ) { RS_DPP_ROW_REDUCE( "v_max_f32_dpp", v ); return v; }
__device__ __forceinline__ uint32_t row_max_u( uint32_t v ) { RS_DPP_ROW_REDUCE( "v_max_u32_dpp", v ); return v; }
// inclusive prefix sum within every row of 16 lanes
__device__ __forceinline__ uint32_t row_scan( uint32_t v ) {
  asm volatile( "s_nop 4\n\t"
                "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                : "+v"( v ) );
  return v;
}
__device__ __forceinline__ float wave_min( float v ) { RS_DPP_REDUCE( "v_min_f32_dpp", v ); return lane63( v ); }
__device__ __forceinline__ float wave_max( float v ) { RS_DPP_REDUCE( "v_max_f32_dpp", v ); return lane63( v ); }
// fixed association: ((quad) + mirrored quad) + mirrored half-row, then rows 0..3 in order
__device__ __forceinline__ double wave_sum( double v ) {
  v += dpp_d<RS_DPP_QUAD_XOR1, 0xf>( 0.0, v );
  v += dpp_d<RS_DPP_QUAD_XOR2, 0xf>( 0.0, v );
  v += dpp_d<RS_DPP_HALF_MIRROR, 0xf>( 0.0, v );
  v += dpp_d<RS_DPP_ROW_MIRROR, 0xf>( 0.0, v );
  v += dpp_d<RS_DPP_BCAST15, 0xa>( 0.0, v );
  v += dpp_d<RS_DPP_BCAST31, 0xc>( 0.0, v );
  const long long r = __double_as_longlong( v );
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane( (int)(uint32_t)r, 63 );
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane( (int)(uint32_t)( (unsigned long long)r >> 32 ), 63 );
  return __longlong_as_double( (long long)( ( (unsigned long long)hi << 32 ) | lo ) );
}
// N sums over the 64 lanes at once, as a butterfly: at every level a lane keeps HALF of its values — lanes with the level's bit clear
// the even ones, the others the odd ones — and adds its partner's copies of the same; after six levels lane l holds the total of
// value l (lanes >= N: nothing).  N / 2 + N / 4 + ... exchanges instead of 6 N: a third of the instructions of N wave_sum()s, which
// was half of all the moments' kernel executed.  Fixed association (partner order 1, 2, 4, 8, 16, 32).
template <int XOR> __device__ __forceinline__ double lane_xor_d( double v )
{
  if( XOR == 1 ) return dpp_d<RS_DPP_QUAD_XOR1, 0xf>( 0.0, v );
  if( XOR == 2 ) return dpp_d<RS_DPP_QUAD_XOR2, 0xf>( 0.0, v );
  if( XOR == 32 ) return __shfl_xor( v, 32 );
  const long long x = __double_as_longlong( v );
  const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_swizzle( (int)(uint32_t)x, ( XOR << 10 ) | 0x1f );
  const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_swizzle( (int)(uint32_t)( (unsigned long long)x >> 32 ), ( XOR << 10 ) | 0x1f );
  return __longlong_as_double( (long long)( ( (unsigned long long)hi << 32 ) | lo ) );
}
template <int XOR, int N>
__device__ __forceinline__ void wave_sums_level( const double ( &a )[N], double ( &o )[( N + 1 ) / 2], int lane )
{
  const bool upper = ( lane & XOR ) != 0;
#pragma unroll
  for( int j = 0; j < ( N + 1 ) / 2; ++j )
  {
    const double lo = a[2 * j], hi = 2 * j + 1 < N ? a[2 * j + 1] : 0.0;
    const double keep = upper ? hi : lo, send = upper ? lo : hi;
    o[j] = keep + lane_xor_d<XOR>( send );
  }
}
template <int N>
__device__ __forceinline__ double wave_sums( const double ( &a )[N], int lane )
{
  static_assert( N <= WAVE && N > 32, "six levels" );
  constexpr int N1 = ( N + 1 ) / 2, N2 = ( N1 + 1 ) / 2, N3 = ( N2 + 1 ) / 2, N4 = ( N3 + 1 ) / 2, N5 = ( N4 + 1 ) / 2;
  double b[N1], c[N2], d[N3], e[N4], f[N5], g[1];
  wave_sums_level<1>( a, b, lane ); wave_sums_level<2>( b, c, lane ); wave_sums_level<4>( c, d, lane );
  wave_sums_level<8>( d, e, lane ); wave_sums_level<16>( e, f, lane ); wave_sums_level<32>( f, g, lane );
  return g[0];
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_u64( unsigned long long v )
{
  const uint32_t lo = dpp_u<CTRL, ROW_MASK>( 0u, (uint32_t)v ), hi = dpp_u<CTRL, ROW_MASK>( 0u, (uint32_t)( v >> 32 ) );
  return ( (unsigned long long)hi << 32 ) | lo;
}
__device__ __forceinline__ unsigned long long wave_sum_u64( unsigned long long v ) {
  v += dpp_u64<RS_DPP_QUAD_XOR1, 0xf>( v );
  v += dpp_u64<RS_DPP_QUAD_XOR2, 0xf>( v );
  v += dpp_u64<RS_DPP_HALF_MIRROR, 0xf>( v );
  v += dpp_u64<RS_DPP_ROW_MIRROR, 0xf>( v );
  v += dpp_u64<RS_DPP_BCAST15, 0xa>( v );
  v += dpp_u64<RS_DPP_BCAST31, 0xc>( v );
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane( (int)(uint32_t)v, 63 );
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane( (int)(uint32_t)( v >> 32 ), 63 );
  return ( (unsigned long long)hi << 32 ) | lo;
}
// inclusive prefix sum over the 64 lanes: Kogge-Stone within each row of 16 (row_shr 1, 2, 4, 8: lanes without a
// source add 0), then the totals of the rows before (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ uint32_t wave_scan( uint32_t v, int lane ) {
  (void)lane;
  asm volatile( "s_nop 4\n\t"
                "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                : "+v"( v ) );
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

// (dist², index) lexicographic "a before b"
__device__ __forceinline__ bool lex_less( float d2a, int ia, float d2b, int ib )
{
  return ( d2a < d2b ) | ( ( d2a == d2b ) & ( ia < ib ) );
}

// ------------------------------------------------------------------------------------------
// cell boxes
// ------------------------------------------------------------------------------------------

struct CellBox { int x0, x1, y0, y1, z0, z1; };

__device__ __forceinline__ bool box_empty( const CellBox& b ) { return ( b.x1 < b.x0 ) | ( b.y1 < b.y0 ) | ( b.z1 < b.z0 ); }
__device__ __forceinline__ bool box_same( const CellBox& a, const CellBox& b )
{ return a.x0 == b.x0 && a.x1 == b.x1 && a.y0 == b.y0 && a.y1 == b.y1 && a.z0 == b.z0 && a.z1 == b.z1; }

// Cells of one axis that can hold a point within `r` of the interval [lo,hi].  Binning of
// the stored points (host, rs_api.hip: cell_of) and this range use the same float
// expression; the 0.01-cell margin is far above the rounding error of either, so the range
// is a superset.
__device__ __forceinline__ void axis_range( float lo, float hi, float r, float gmin, float inv_cell, int dim,
                                            int& c0, int& c1 )
{
  float a = floorf( ( lo - r - gmin ) * inv_cell - 0.01f );
  float b = floorf( ( hi + r - gmin ) * inv_cell + 0.01f );
  // (clamped to the grid as INTEGERS: as floats the three dim - 1 are wave-uniform values that live in vector registers across every
  //  search loop — three of the eleven registers k_score spilled; the literals below are encoded in their instructions)
  a = fminf( fmaxf( a, 0.0f ), 16777216.0f );
  b = fminf( fmaxf( b, -1.0f ), 16777216.0f );
  c0 = (int)a;
  const int bi = min( (int)b, dim - 1 );
  c1 = ( bi >= c0 ) ? bi : -1;        // empty -> c1 < c0
}

struct TileBounds { float lx, hx, ly, hy, lz, hz; bool any; };

__device__ __forceinline__ TileBounds wave_bounds( bool active, float qx, float qy, float qz )
{
  const float big = FLT_MAX;
  TileBounds t;
  t.lx = wave_min( active ? qx : big );  t.hx = wave_max( active ? qx : -big );
  t.ly = wave_min( active ? qy : big );  t.hy = wave_max( active ? qy : -big );
  t.lz = wave_min( active ? qz : big );  t.hz = wave_max( active ? qz : -big );
  t.any = t.hx >= t.lx;
  return t;
}

__device__ __forceinline__ CellBox cell_box( const GridView& g, const TileBounds& t, float r )
{
  CellBox b;
  axis_range( t.lx, t.hx, r, g.minx, g.inv_cell, g.w, b.x0, b.x1 );
  axis_range( t.ly, t.hy, r, g.miny, g.inv_cell, g.h, b.y0, b.y1 );
  axis_range( t.lz, t.hz, r, g.minz, g.inv_cell, g.d, b.z0, b.z1 );
  b.x0 = uni( b.x0 ); b.x1 = uni( b.x1 ); b.y0 = uni( b.y0 ); b.y1 = uni( b.y1 ); b.z0 = uni( b.z0 ); b.z1 = uni( b.z1 );
  return b;
}

// core grown by k cells on every side, clipped to full
__device__ __forceinline__ CellBox box_grow( const CellBox& core, int k, const CellBox& full )
{
  CellBox b;
  b.x0 = max( core.x0 - k, full.x0 ); b.x1 = min( core.x1 + k, full.x1 );
  b.y0 = max( core.y0 - k, full.y0 ); b.y1 = min( core.y1 + k, full.y1 );
  b.z0 = max( core.z0 - k, full.z0 ); b.z1 = min( core.z1 + k, full.z1 );
  return b;
}

// Distance from q to the nearest face of `cur` that can still move outward (a face already at
// `full` never hides a point within the radius).  Made conservative by a margin far above the
// rounding of the face coordinates and of the binning.
__device__ __forceinline__ float box_cover( const GridView& g, const CellBox& cur, const CellBox& full,
                                            float qx, float qy, float qz )
{
  float c = FLT_MAX;
  if( cur.x0 > full.x0 ) c = fminf( c, qx - ( g.minx + (float)cur.x0 * g.cell ) );
  if( cur.x1 < full.x1 ) c = fminf( c, ( g.minx + (float)( cur.x1 + 1 ) * g.cell ) - qx );
  if( cur.y0 > full.y0 ) c = fminf( c, qy - ( g.miny + (float)cur.y0 * g.cell ) );
  if( cur.y1 < full.y1 ) c = fminf( c, ( g.miny + (float)( cur.y1 + 1 ) * g.cell ) - qy );
  if( cur.z0 > full.z0 ) c = fminf( c, qz - ( g.minz + (float)cur.z0 * g.cell ) );
  if( cur.z1 < full.z1 ) c = fminf( c, ( g.minz + (float)( cur.z1 + 1 ) * g.cell ) - qz );
  return c - ( 1e-4f * g.cell + 2e-5f );
}

// ------------------------------------------------------------------------------------------
// shell sweep
// ------------------------------------------------------------------------------------------

// Per-wave LDS.  CAP = staged candidates per round (the per-row cold search of the score batch stages more than a wave's worth).
template <int CAP>
struct __attribute__(( aligned( 16 ) )) WaveLdsT      // (aligned: the compiler splits the 128-bit reads into pairs of 64-bit ones otherwise)
{
  float    px[CAP], py[CAP], pz[CAP];      // staged candidates, one array per coordinate so that four
  int      pidx[CAP];                      // consecutive candidates load as one ds_read_b128 per coordinate
  float    nx[CAP], ny[CAP], nz[CAP];      // their normals, same layout
  uint32_t slot[CAP];      // their positions in the cell-sorted cloud
  uint32_t seg_a[WAVE], len_a[WAVE], seg_b[WAVE], pre[WAVE];   // row pieces of the current batch
  uint32_t evals;          // profiling only (lane 0): candidates this wave staged and evaluated, flushed once by EvalScope
};
typedef WaveLdsT<WAVE> WaveLds;

// Profiling only (GridView::evals non-null): the wave's candidate count goes to the sharded device counters ONCE, when the
// wave leaves the kernel — an atomic per sweep was most of what WRITE_SIZE saw of k_label (75 MB per launch for 5 MB of
// results: atomics execute at the memory side, 64 B each) and a good part of k_icp_corr's.
struct EvalScope
{
  unsigned long long* evals; uint32_t& count; int lane;
  template <class LDS>
  __device__ __forceinline__ EvalScope( unsigned long long* e, LDS& l, int ln ) : evals( e ), count( l.evals ), lane( ln ) { if( evals && lane == 0 ) count = 0u; }
  __device__ __forceinline__ ~EvalScope()
  {
    if( evals && lane == 0 && count ) atomicAdd( evals + 8 * ( ( blockIdx.x + 37 * blockIdx.y ) & ( EVAL_SHARDS - 1 ) ), (unsigned long long)count );   // sharded, one cache line each
  }
};

// Stream every point of (out \ in) through the wave's LDS and call f( X, Y, Z, k ) for every
// group of four staged candidates k..k+3 (X = their four x coordinates, ...; L.pidx[k+i],
// L.nx/ny/nz[k+i], L.slot[k+i] belong to them); k is wave-uniform.  `in` (if in_valid) is any box.  Chunks are padded to a multiple of 4 with sentinels at +FLT_MAX whose
// dist² is +inf: they can never be "within the radius", so f needs no validity test.
// When several waves sweep the same shell together, wave `share` of `n_share` takes the chunks
// whose running number is congruent to it.
// `cull` (optional): the lanes the sweep is for lie in the box [lx,hx] x [ly,hy] x [lz,hz] and none of them looks farther than
// sqrt( R2 ) — a cell farther than that from the box holds nothing for them, so a (y,z) row of cells beyond it is skipped and the
// others are clipped in x to what the remaining distance allows: the swept volume is the box grown by a BALL, not by a cube
// (two thirds of it for a 5 cm tile and a 10 cm reach; half and less of a surface that passes the tile at a distance).
struct Cull { float lx, hx, ly, hy, lz, hz, R2; };
#ifndef RS_SWEEP_CENTER_OUT
#define RS_SWEEP_CENTER_OUT 1
#endif
template <bool WITH_NOR, class F>
__device__ __forceinline__ uint32_t sweep_shell( const GridView& g, const CellBox& out, const CellBox& in, bool in_valid,
                                                 WaveLds& L, int lane, int share, int n_share, F&& f, uint32_t give_up_from = 0xffffffffu,
                                                 const bool culled = false, const Cull cull = Cull{} )
{
  const int ny = out.y1 - out.y0 + 1, nz = out.z1 - out.z0 + 1;
  const int n_rows = ny * nz;
  const float inv_ny = 1.0f / (float)ny;
  uint32_t streamed = 0, evaluated = 0;
  // (Cooperating waves all enumerate the same rows and split the chunks; giving each wave whole
  //  row batches instead balanced worse and measured slower.)
  const int c_share = share, c_nshare = n_share;
  for( int r0 = 0; r0 < n_rows; r0 += WAVE )
  {
    // each lane describes one (y,z) row of cells: up to two x-pieces
    const int r = r0 + lane;
    uint32_t sa = 0, la = 0, sb = 0, lb = 0;
    if( r < n_rows )
    {
      // r / ny, r % ny without the integer-division sequence: r < 2^20, so the float quotient is off by at most one
      int rz = (int)( (float)r * inv_ny );
      rz -= ( rz * ny > r ) ? 1 : 0;
      rz += ( ( rz + 1 ) * ny <= r ) ? 1 : 0;
      int y = out.y0 + ( r - rz * ny ), z = out.z0 + rz;
      if( culled && RS_SWEEP_CENTER_OUT )
      {
        // rows from the middle of the box outwards (c, c + 1, c - 1, c + 2, ...; z slowest): the nearest candidates arrive first, the
        // lanes' bounds are tight before the bulk comes — fewer candidates pass the bound test (no gate for them) and fewer are
        // counted as "closer than the match so far", the count that decides whether a rank pass is needed
        const int iy = r - rz * ny, cy = ( out.y0 + out.y1 ) >> 1, cz = ( out.z0 + out.z1 ) >> 1;
        y = cy + ( ( iy & 1 ) ? ( ( iy + 1 ) >> 1 ) : -( iy >> 1 ) );
        z = cz + ( ( rz & 1 ) ? ( ( rz + 1 ) >> 1 ) : -( rz >> 1 ) );
      }
      const uint32_t* cs = g.cell_start + (size_t)( z * g.h + y ) * g.w;
      const bool inside = in_valid && y >= in.y0 && y <= in.y1 && z >= in.z0 && z <= in.z1;
      int cx0 = out.x0, cx1 = out.x1;
      if( culled )
      {
        // distance of the row's cells from the lanes' box in y and z (cell faces as box_cover computes them; R2 carries the margin)
        const float y_lo = g.miny + (float)y * g.cell, y_hi = g.miny + (float)( y + 1 ) * g.cell;
        const float z_lo = g.minz + (float)z * g.cell, z_hi = g.minz + (float)( z + 1 ) * g.cell;
        const float dy = fmaxf( fmaxf( cull.ly - y_hi, y_lo - cull.hy ), 0.0f ), dz = fmaxf( fmaxf( cull.lz - z_hi, z_lo - cull.hz ), 0.0f );
        const float rem = cull.R2 - ( dy * dy + dz * dz );
        if( rem > 0.0f )
        {
          int a, b;
          axis_range( cull.lx, cull.hx, sqrtf( rem ), g.minx, g.inv_cell, g.w, a, b );
          cx0 = max( cx0, a ); cx1 = min( cx1, b );
        }
        else cx1 = cx0 - 1;
      }
      if( cx1 < cx0 ) { }
      else if( !inside ) { sa = cs[cx0]; la = cs[cx1 + 1] - sa; }
      else
      {
        // the row minus in's x-range (which may stick out of, or miss, out's)
        const int a1 = min( in.x0 - 1, cx1 ), b0 = max( in.x1 + 1, cx0 );
        if( a1 >= cx0 ) { sa = cs[cx0]; la = cs[a1 + 1] - sa; }
        if( b0 <= cx1 ) { sb = cs[b0]; lb = cs[cx1 + 1] - sb; }
      }
    }
    const uint32_t incl = wave_scan( la + lb, lane );
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane( (int)incl, WAVE - 1 );
    // (give_up_from: the caller would rather not stream this much with one wave; decided on the first batch of cell rows, before
    //  anything was evaluated — returns ~0)
    if( r0 == 0 && total >= give_up_from ) return 0xffffffffu;
    L.seg_a[lane] = sa; L.len_a[lane] = la; L.seg_b[lane] = sb; L.pre[lane] = incl - ( la + lb );
    streamed += total;
    wave_lds_fence();

    // This wave's chunks of the batch: share, share + n_share, ...  The global loads of chunk
    // c+1 are issued before chunk c is evaluated, so their latency hides under the evaluation.
    auto fetch = [&]( uint32_t c0, float4& P, float4& N, uint32_t& src )
    {
      const uint32_t j = c0 + lane;
      P = make_float4( FLT_MAX, FLT_MAX, FLT_MAX, 0.0f ); N = make_float4( 0.0f, 0.0f, 0.0f, 0.0f ); src = 0;
      if( j < total )
      {
        int row = 0;                                   // last row whose first candidate number is <= j
#pragma unroll
        for( int step = WAVE / 2; step > 0; step >>= 1 ) { if( L.pre[row + step] <= j ) row += step; }
        const uint32_t off = j - L.pre[row];
        const uint32_t la_r = L.len_a[row];
        src = ( off < la_r ) ? ( L.seg_a[row] + off ) : ( L.seg_b[row] + ( off - la_r ) );
        P = g.pos[src];
        if( WITH_NOR ) N = g.nor[src];
      }
    };
    const uint32_t stride = (uint32_t)c_nshare * WAVE;
    uint32_t c0 = (uint32_t)c_share * WAVE;
    float4 P, N; uint32_t src;
    if( c0 < total ) fetch( c0, P, N, src );
    while( c0 < total )
    {
      L.px[lane] = P.x; L.py[lane] = P.y; L.pz[lane] = P.z; L.pidx[lane] = __float_as_int( P.w );
      if( WITH_NOR ) { L.nx[lane] = N.x; L.ny[lane] = N.y; L.nz[lane] = N.z; }
      L.slot[lane] = src;
      wave_lds_fence();
      const uint32_t cn = c0 + stride;
      if( cn < total ) fetch( cn, P, N, src );         // in flight during the loop below
      const uint32_t cnt = ( total - c0 < WAVE ) ? ( total - c0 ) : WAVE;
      const uint32_t cnt4 = ( cnt + 3u ) & ~3u;
      evaluated += cnt;
#pragma unroll 1
      for( uint32_t k = 0; k < cnt4; k += 4 )
      {
        const float4 X = *reinterpret_cast<const float4*>( &L.px[k] );
        const float4 Y = *reinterpret_cast<const float4*>( &L.py[k] );
        const float4 Z = *reinterpret_cast<const float4*>( &L.pz[k] );
        f( X, Y, Z, (int)k );
      }
      wave_lds_fence();
      c0 = cn;
    }
    wave_lds_fence();
  }
  if( g.evals && lane == 0 ) L.evals += evaluated;       // (flushed by the kernel's EvalScope)
  return streamed;
}

// The sweep of a tile whose lanes all start from a candidate, done per ROW of 16 lanes (a quarter of the tile: 16
// Hilbert-consecutive queries, a patch a few centimetres across).  The wave-wide sweep streams the cells of the whole
// tile's box past all 64 lanes — about 2.3 candidates per query, every one of them tested by every lane; here each row
// streams only the cells its own lanes reach, 16 candidates per row and round, and a lane tests its row's candidates
// only: a third of the distance evaluations, which are what phase A's VALU time goes to.  The staged candidates of row r
// occupy entries [16 r, 16 r + 16) of the wave's LDS arrays; the four rows' ds_read_b128 addresses differ, which the LDS
// serves at the same rate (it processes 16 lanes of a b128 read at a time anyway).  Candidates that two rows both
// reach are staged twice — a lane still meets each candidate once.  Returns 0 (nothing done) when a row's box has
// more than 16 rows of cells: the caller then sweeps the tile's common box as before; 1 when the sweep is done.
template <bool WITH_NOR, class F>
__device__ __forceinline__ int sweep_by_rows( const GridView& g, const CellBox& clip, bool mask, float reach,
                                              float qx, float qy, float qz, WaveLds& L, int lane, F&& f, uint32_t& streamed )
{
  const float big = FLT_MAX;
  const float lx = row_min( mask ? qx - reach : big ), hx = row_max( mask ? qx + reach : -big );
  const float ly = row_min( mask ? qy - reach : big ), hy = row_max( mask ? qy + reach : -big );
  const float lz = row_min( mask ? qz - reach : big ), hz = row_max( mask ? qz + reach : -big );
  int x0, x1, y0, y1, z0, z1;
  axis_range( lx, hx, 0.0f, g.minx, g.inv_cell, g.w, x0, x1 );
  axis_range( ly, hy, 0.0f, g.miny, g.inv_cell, g.h, y0, y1 );
  axis_range( lz, hz, 0.0f, g.minz, g.inv_cell, g.d, z0, z1 );
  x0 = max( x0, clip.x0 ); x1 = min( x1, clip.x1 ); y0 = max( y0, clip.y0 ); y1 = min( y1, clip.y1 ); z0 = max( z0, clip.z0 ); z1 = min( z1, clip.z1 );
  const bool empty = ( hx < lx ) | ( x1 < x0 ) | ( y1 < y0 ) | ( z1 < z0 );
  const int ny = y1 - y0 + 1;
  const int n_rows = empty ? 0 : ny * ( z1 - z0 + 1 );
  if( __any( n_rows > 16 ) ) return 0;
  const int l16 = lane & 15, base = lane & 48;
  uint32_t sa = 0, la = 0;
  if( l16 < n_rows )
  {
    int rz = (int)( (float)l16 / (float)ny );               // l16 < 16, ny <= 16: exact enough to be off by at most one
    rz -= ( rz * ny > l16 ) ? 1 : 0;
    rz += ( ( rz + 1 ) * ny <= l16 ) ? 1 : 0;
    const int y = y0 + ( l16 - rz * ny ), z = z0 + rz;
    const uint32_t* cs = g.cell_start + (size_t)( z * g.h + y ) * g.w;
    sa = cs[x0]; la = cs[x1 + 1] - sa;
  }
  const uint32_t incl = row_scan( la );
  const uint32_t total = row_max_u( incl );                 // of this lane's row
  // entries past a row's last cell row carry pre = total: the binary search below never selects them
  L.seg_a[lane] = sa; L.pre[lane] = incl - la;
  wave_lds_fence();
  const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane( (int)total, 15 ), t1 = (uint32_t)__builtin_amdgcn_readlane( (int)total, 31 );
  const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane( (int)total, 47 ), t3 = (uint32_t)__builtin_amdgcn_readlane( (int)total, 63 );
  const uint32_t longest = max( max( t0, t1 ), max( t2, t3 ) );
  streamed += t0 + t1 + t2 + t3;

  auto fetch = [&]( uint32_t c0, float4& P, float4& N, uint32_t& src )
  {
    const uint32_t j = c0 + (uint32_t)l16;
    P = make_float4( FLT_MAX, FLT_MAX, FLT_MAX, 0.0f ); N = make_float4( 0.0f, 0.0f, 0.0f, 0.0f ); src = 0;
    if( j < total )
    {
      int row = 0;                                   // last cell row of this lane's row whose first candidate number is <= j
#pragma unroll
      for( int step = 8; step > 0; step >>= 1 ) { if( L.pre[base + row + step] <= j ) row += step; }
      src = L.seg_a[base + row] + ( j - L.pre[base + row] );
      P = g.pos[src];
      if( WITH_NOR ) N = g.nor[src];
    }
  };
  float4 P, N; uint32_t src;
  uint32_t c0 = 0;
  if( c0 < longest ) fetch( c0, P, N, src );
  while( c0 < longest )
  {
    L.px[lane] = P.x; L.py[lane] = P.y; L.pz[lane] = P.z; L.pidx[lane] = __float_as_int( P.w );
    if( WITH_NOR ) { L.nx[lane] = N.x; L.ny[lane] = N.y; L.nz[lane] = N.z; }
    L.slot[lane] = src;
    wave_lds_fence();
    const uint32_t cn = c0 + 16u;
    if( cn < longest ) fetch( cn, P, N, src );       // in flight during the evaluation
    const uint32_t left = longest - c0;
    const int n4 = left >= 16u ? 4 : (int)( ( left + 3u ) >> 2 );
#pragma unroll 1
    for( int k4 = 0; k4 < n4; ++k4 )
    {
      const int k = base + 4 * k4;
      const float4 X = *reinterpret_cast<const float4*>( &L.px[k] );
      const float4 Y = *reinterpret_cast<const float4*>( &L.py[k] );
      const float4 Z = *reinterpret_cast<const float4*>( &L.pz[k] );
      f( X, Y, Z, k );
    }
    wave_lds_fence();
    c0 = cn;
  }
  if( g.evals && lane == 0 ) L.evals += ( t0 + t1 + t2 + t3 ) / 4;   // (each candidate is tested by 16 lanes, not 64)
  return 1;
}

// Result of a search for one query.
// `fail_max`: the largest gate value max(dot,0) among the candidates that were inside the lane's bound when
// met and failed the gate — every candidate closer than the final match (or, without one, within the
// radius) is among them.  Only the ICP certificates below read it.
// `rank_slack` (ICP only, > 0 when set): the match was rejected for its rank, and at least K candidates lie closer to
// the query than (distance of the match - rank_slack) — see icp_certificate.
struct Match { float d2; int idx; float dot; int slot; bool found; float fail_max; float rank_slack; };
__device__ __forceinline__ Match no_match() { Match m; m.d2 = INFINITY; m.idx = INT_MAX; m.dot = 0.0f; m.slot = -1; m.found = false; m.fail_max = 0.0f; m.rank_slack = 0.0f; return m; }

typedef float f32x2 __attribute__(( ext_vector_type( 2 ) ));

// dist² of four candidates to one query, two at a time in packed fp32: each v_pk_add/v_pk_mul
// rounds its two halves exactly like the scalar instruction, and the order is the reference's
// (vx*vx + vy*vy) + vz*vz (msh_hash_grid.h:852-855).  No fused multiply-add is formed
// (-ffp-contract=off).
__device__ __forceinline__ void dist2x4( const float4& X, const float4& Y, const float4& Z, float qx, float qy, float qz,
                                         float& d0, float& d1, float& d2, float& d3 )
{
  const f32x2 q_x = { qx, qx }, q_y = { qy, qy }, q_z = { qz, qz };
  f32x2 ax = f32x2{ X.x, X.y } - q_x, ay = f32x2{ Y.x, Y.y } - q_y, az = f32x2{ Z.x, Z.y } - q_z;
  f32x2 bx = f32x2{ X.z, X.w } - q_x, by = f32x2{ Y.z, Y.w } - q_y, bz = f32x2{ Z.z, Z.w } - q_z;
  f32x2 a = ax * ax + ay * ay + az * az;
  f32x2 b = bx * bx + by * by + bz * bz;
  d0 = a.x; d1 = a.y; d2 = b.x; d3 = b.y;
}

// The candidate step shared by all searches, four staged candidates at a time: update the best
// match `m` of this lane.
// `bound` folds three tests into one compare: a candidate can only matter if
// dist² < bound, where bound = radius² until a match exists and then the float just above the
// match's dist² (so "<= best" including ties, which the last branch settles by index);
// inactive lanes carry bound = -1.  seen_closer counts the candidates that passed it (against the
// bound at the start of the group: looser than one by one, still an upper bound of the rank); with SELF
// the lane's current match is not counted when it meets itself (whatever precedes the final match
// preceded every earlier best too, so the count still bounds the rank from above).
// Three levels: (1) distances only — most groups end here; (2) some lane has a candidate inside its
// bound (lanes without a match see that for everything within the radius): the gate of all four,
// packed like the distances; (3) a candidate passed both: settle it one by one.
// (Questions about the whole wave — "is any lane ...?" — are asked of LANE MASKS: ballot( compare ) is the compare's own SGPR
//  result, and and / or / "!= 0" of such masks are scalar instructions.  HIP's __any( a | b ) goes through a VGPR instead,
//  two VALU instructions per question, six questions per group of four candidates.)
typedef unsigned long long lanemask;
#define RS_BALLOT( c ) __builtin_amdgcn_ballot_w64( c )
// cnt += 1 in the lanes of `mask`: one v_addc (the mask is the carry-in), not a select and an add
__device__ __forceinline__ void count_lanes( int& cnt, lanemask mask )
{
  lanemask carry_out;
  asm( "v_addc_co_u32 %0, %1, 0, %0, %2" : "+v"( cnt ), "=s"( carry_out ) : "s"( mask ) );
}

template <bool GATED, bool SELF, class LDS>
__device__ __forceinline__ void consider4( const float4& X, const float4& Y, const float4& Z, int k, const LDS& L,
                                           float qx, float qy, float qz, float nx, float ny, float nz,
                                           float tmin, float& bound, Match& m, int& seen_closer )
{
  float d[4];
  dist2x4( X, Y, Z, qx, qy, qz, d[0], d[1], d[2], d[3] );
  lanemask in[4];
#pragma unroll
  for( int i = 0; i < 4; ++i ) in[i] = RS_BALLOT( d[i] < bound );
  if( ( in[0] | in[1] | in[2] | in[3] ) == 0ull ) return;
  int4 I = make_int4( 0, 0, 0, 0 );
  if( SELF )
  {
    // searches seeded with a starting candidate (ICP iterations >= 2): a lane's match meets itself in the
    // stream exactly once (d == its dist² < bound) — that is most of what gets here once the bounds are
    // tight, and it is no news
    I = *reinterpret_cast<const int4*>( &L.pidx[k] );
    in[0] &= RS_BALLOT( I.x != m.idx ); in[1] &= RS_BALLOT( I.y != m.idx ); in[2] &= RS_BALLOT( I.z != m.idx ); in[3] &= RS_BALLOT( I.w != m.idx );
    if( ( in[0] | in[1] | in[2] | in[3] ) == 0ull ) return;
  }
#pragma unroll
  for( int i = 0; i < 4; ++i ) count_lanes( seen_closer, in[i] );
  float dot[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
  lanemask pass[4] = { in[0], in[1], in[2], in[3] };
  if( GATED )
  {
    const float4 NX = *reinterpret_cast<const float4*>( &L.nx[k] );
    const float4 NY = *reinterpret_cast<const float4*>( &L.ny[k] );
    const float4 NZ = *reinterpret_cast<const float4*>( &L.nz[k] );
    const f32x2 n_x = { nx, nx }, n_y = { ny, ny }, n_z = { nz, nz };
    const f32x2 a = f32x2{ NX.x, NX.y } * n_x + f32x2{ NY.x, NY.y } * n_y + f32x2{ NZ.x, NZ.y } * n_z;   // msh_vec3_dot( m, n )
    const f32x2 b = f32x2{ NX.z, NX.w } * n_x + f32x2{ NY.z, NY.w } * n_y + f32x2{ NZ.z, NZ.w } * n_z;
    dot[0] = a.x; dot[1] = a.y; dot[2] = b.x; dot[3] = b.y;
    if( SELF )     // only the ICP instantiation issues certificates (icp_emit); it needs the exact gate of every in-bound candidate
    {
      const lanemask lane_bit = 1ull << ( threadIdx.x & ( WAVE - 1 ) );
#pragma unroll
      for( int i = 0; i < 4; ++i )
      {
        const float dc = dot[i] > 0.0f ? dot[i] : 0.0f;                                                   // msh_max( dot, 0.0f )
        pass[i] &= RS_BALLOT( dc >= tmin ) & RS_BALLOT( dc <= 1.0f );
        m.fail_max = fmaxf( m.fail_max, ( ( in[i] & ~pass[i] ) & lane_bit ) ? dc : 0.0f );
      }
    }
    else
    {
      // One compare per candidate here, the exact gate  tmin <= max(dot,0) <= 1  only for what survives it (below): with
      // tmin > 0, max(dot,0) >= tmin implies !(dot < tmin) (NaN included); a gate that lets max(dot,0) = 0 pass (tmin <= 0)
      // filters nothing at this stage.
      const float pre = tmin > 0.0f ? tmin : -INFINITY;
#pragma unroll
      for( int i = 0; i < 4; ++i ) pass[i] &= RS_BALLOT( !( dot[i] < pre ) );
    }
  }
  if( ( pass[0] | pass[1] | pass[2] | pass[3] ) == 0ull ) return;
#pragma unroll
  for( int i = 0; i < 4; ++i )
  {
    if( pass[i] != 0ull )
    {
      const int idx = SELF ? ( i == 0 ? I.x : i == 1 ? I.y : i == 2 ? I.z : I.w ) : L.pidx[k + i];
      const float dc = dot[i] > 0.0f ? dot[i] : 0.0f;                                                   // msh_max( dot, 0.0f )
      // (the lane's bit of pass[i] is implied by the tests below but for the gate's prefilter and SELF's own-match skip: the exact
      //  gate re-checks the former, d < bound with the match at its own bound the latter — a match never beats itself in lex_less)
      const bool gate = !GATED | ( ( dc >= tmin ) & ( dc <= 1.0f ) );
      const bool take = gate & ( d[i] < bound ) & lex_less( d[i], idx, m.d2, m.idx );   // the bound may have tightened within the group
      if( take )
      {
        m.d2 = d[i]; m.idx = idx; m.dot = dc; m.slot = (int)L.slot[k + i]; m.found = true;
        bound = __int_as_float( __float_as_int( d[i] ) + 1 );   // next float above d2 (d2 >= 0, finite, < radius²)
      }
    }
  }
}

// count, among four candidates, those that precede (bd2, bidx) within the radius.  bd2 < radius² (it is a match's dist²), so
// "within the radius" is implied by "closer than the match"; a candidate AT the match's distance precedes it by its index — a
// tie, looked at only when some lane has one.
template <class LDS>
__device__ __forceinline__ int precede4( const float4& X, const float4& Y, const float4& Z, int k, const LDS& L,
                                         float qx, float qy, float qz, float radius_sq, float bd2, int bidx )
{
  (void)radius_sq;
  float d[4];
  dist2x4( X, Y, Z, qx, qy, qz, d[0], d[1], d[2], d[3] );
  int c = 0;
#pragma unroll
  for( int i = 0; i < 4; ++i ) count_lanes( c, RS_BALLOT( d[i] < bd2 ) );
  const lanemask eq = RS_BALLOT( d[0] == bd2 ) | RS_BALLOT( d[1] == bd2 ) | RS_BALLOT( d[2] == bd2 ) | RS_BALLOT( d[3] == bd2 );
  if( eq != 0ull )
  {
#pragma unroll
    for( int i = 0; i < 4; ++i ) c += ( ( d[i] == bd2 ) & ( L.pidx[k + i] < bidx ) ) ? 1 : 0;
  }
  return c;
}

// The rank pass of a search that issues certificates (ICP): besides the exact rank, how many candidates lie
// within 0.5, 0.75, 0.9 and 0.97 of the match's distance.
struct RankBands { float dm; int c1, c2, c3, c4; };
#define RANK_BAND_1 0.5f
#define RANK_BAND_2 0.75f
#define RANK_BAND_3 0.9f
#define RANK_BAND_4 0.97f
__device__ __forceinline__ RankBands rank_bands( const Match& m )
{
  RankBands b; b.dm = sqrtf( m.d2 ); b.c1 = b.c2 = b.c3 = b.c4 = 0;
  return b;
}
__device__ __forceinline__ int precede4_bands( const float4& X, const float4& Y, const float4& Z, int k, const WaveLds& L,
                                               float qx, float qy, float qz, float radius_sq, float bd2, int bidx, RankBands& b )
{
  float d[4];
  dist2x4( X, Y, Z, qx, qy, qz, d[0], d[1], d[2], d[3] );
  const float t1 = RANK_BAND_1 * b.dm, t2 = RANK_BAND_2 * b.dm, t3 = RANK_BAND_3 * b.dm, t4 = RANK_BAND_4 * b.dm;
  const float s1 = t1 * t1, s2 = t2 * t2, s3 = t3 * t3, s4 = t4 * t4;
  int c = 0;
#pragma unroll
  for( int i = 0; i < 4; ++i )
  {
    c += ( ( d[i] < radius_sq ) & lex_less( d[i], L.pidx[k + i], bd2, bidx ) ) ? 1 : 0;
    b.c1 += d[i] < s1 ? 1 : 0; b.c2 += d[i] < s2 ? 1 : 0; b.c3 += d[i] < s3 ? 1 : 0; b.c4 += d[i] < s4 ? 1 : 0;
  }
  return c;
}
// Widest band that holds K candidates, as a distance margin (0: none).  3e-4 m is far above the fp32 rounding of the
// distances involved (<= 1e-7 m at these radii) and above the 1e-4 m the gate certificate's radius gives away.
__device__ __forceinline__ float rank_slack_of( const RankBands& b, int K )
{
  // (the same products as in precede4_bands; a chain of selects on values, so that nothing here needs an address)
  float t = b.dm;
  t = b.c4 >= K ? RANK_BAND_4 * b.dm : t;
  t = b.c3 >= K ? RANK_BAND_3 * b.dm : t;
  t = b.c2 >= K ? RANK_BAND_2 * b.dm : t;
  t = b.c1 >= K ? RANK_BAND_1 * b.dm : t;
  const float s = b.dm - t - 3e-4f;
  return s > 0.0f ? s : 0.0f;
}

// bound for a lane before / after a merge
__device__ __forceinline__ float bound_of( bool active, float radius_sq, const Match& m )
{
  if( !active ) return -1.0f;
  return m.found ? __int_as_float( __float_as_int( m.d2 ) + 1 ) : radius_sq;
}

// How far from its query a lane still has to look: to its match (a candidate that precedes the match,
// or ties with it, is no farther; the factor and the offset are far above the fp32 rounding of dist²
// and of the square root), or the whole radius while it has none.
__device__ __forceinline__ float reach_of( const Match& m, float radius )
{
  return m.found ? sqrtf( m.d2 ) * 1.0001f + 1e-5f : radius;
}

// Cells overlapping the boxes [q - reach, q + reach] of the lanes in `mask`, clipped to `clip`.  Wave-uniform.
__device__ __forceinline__ CellBox reach_box( const GridView& g, const CellBox& clip, bool mask, float reach,
                                              float qx, float qy, float qz )
{
  const float big = FLT_MAX;
  TileBounds t;
  t.lx = wave_min( mask ? qx - reach : big );  t.hx = wave_max( mask ? qx + reach : -big );
  t.ly = wave_min( mask ? qy - reach : big );  t.hy = wave_max( mask ? qy + reach : -big );
  t.lz = wave_min( mask ? qz - reach : big );  t.hz = wave_max( mask ? qz + reach : -big );
  t.any = true;
  CellBox b = cell_box( g, t, 0.0f );
  b.x0 = max( b.x0, clip.x0 ); b.x1 = min( b.x1, clip.x1 );
  b.y0 = max( b.y0, clip.y0 ); b.y1 = min( b.y1, clip.y1 );
  b.z0 = max( b.z0, clip.z0 ); b.z1 = min( b.z1, clip.z1 );
  return b;
}

__device__ __forceinline__ CellBox box_clip( const CellBox& a, const CellBox& c )
{
  CellBox b;
  b.x0 = max( a.x0, c.x0 ); b.x1 = min( a.x1, c.x1 );
  b.y0 = max( a.y0, c.y0 ); b.y1 = min( a.y1, c.y1 );
  b.z0 = max( a.z0, c.z0 ); b.z1 = min( a.z1, c.z1 );
  return b;
}

// Nearest candidate within the radius [whose normal passes tmin <= max(dot,0) <= 1, if GATED],
// accepted only if fewer than K candidates (of any normal) precede it in (dist², index) order.
// GATED: the reference's "first normal-compatible entry of the K-nearest list"
// (lib/rs/icp.h:361-380, apps/pose_proposal/pose_proposal.cpp:127-147).
// !GATED (K = 1): the plain nearest neighbour of rs_pointcloud_filters.cpp:758.
//
// `max_stages` (historical name) is the hand-off threshold: a tile that is still unsettled
// after its lone wave has streamed that many candidates sits in a populated neighbourhood, so
// the rest of its box is heavy; *handoff is set, the result is meaningless, and the caller
// queues the tile for the cooperative kernel, which sweeps the whole box with several waves
// (a lone wave needs ~0.7 ms for the ~10^4 candidates of a cluttered corner; the bulk of the
// tiles settle in the first shell with a few hundred).
// BOUNDED_ONLY (the warm ICP iterations' phase A): only the one-sweep path of a tile whose lanes all start from a candidate;
// any other tile is handed off at once (*handoff) — the cooperative kernel gives it a workgroup straight away instead of
// after a lone wave's first shells, and this instantiation carries no shell loop (registers: phase A then fits 6 waves per
// SIMD without scratch).
// The lanes of `mask`, their common box and farthest reach, for sweep_shell's culling.
__device__ __forceinline__ Cull cull_of( const GridView& g, bool mask, float reach, float qx, float qy, float qz )
{
  const TileBounds t = wave_bounds( mask, qx, qy, qz );
  const float R = wave_max( mask ? reach : 0.0f ) + ( 1e-4f * g.cell + 2e-5f );      // (the margin of box_cover: far above the rounding of the cell faces and of the binning)
  Cull c; c.lx = t.lx; c.hx = t.hx; c.ly = t.ly; c.hy = t.hy; c.lz = t.lz; c.hz = t.hz; c.R2 = R * R;
  return c;
}

// ... and the cells that ball-grown box touches, clipped to `clip`
__device__ __forceinline__ CellBox cull_cells( const GridView& g, const Cull& c, const CellBox& clip )
{
  const float R = sqrtf( c.R2 ) * 1.0001f;
  TileBounds t; t.lx = c.lx - R; t.hx = c.hx + R; t.ly = c.ly - R; t.hy = c.hy + R; t.lz = c.lz - R; t.hz = c.hz + R; t.any = true;
  return box_clip( cell_box( g, t, 0.0f ), clip );
}

template <bool GATED, bool WARM = false, bool BOUNDED_ONLY = false, bool CULL = false>
__device__ __forceinline__ Match tile_search( const GridView& g, bool active,
                                              float qx, float qy, float qz, float nx, float ny, float nz,
                                              float radius, float radius_sq, float tmin, int K,
                                              WaveLds& L, int lane, int max_stages, bool* handoff, int* dbg_unsettled,
                                              Match m /* starting candidate: empty, or a genuine one (within radius, gate passed) that only tightens the bounds */,
                                              int* n_sweeps = nullptr /* out: shells swept + rank pass: the tile's cost class */,
                                              bool by_rows = false /* WARM: sweep_by_rows for tiles whose lanes all start from a candidate */,
                                              uint32_t* n_streamed = nullptr /* out: candidates streamed, rank pass included */,
                                              int bounded_give_up_total = 0 /* BOUNDED_ONLY: hand a bounded tile off too when, swept tile-wide, the first 64 cell rows of its box hold this many candidates (0: never) */ )
{
  if( handoff ) *handoff = false;
  int sweeps = 0;
  if( !__any( active ) ) return m;
  int seen_closer = 0;   // candidates that were no farther than the best-so-far when they were met
  float bound = bound_of( active, radius_sq, m );
  // how far a lane still has to look: to its match, or — without one — the radius
  auto reach = [&]() -> float { return reach_of( m, radius ); };
  uint32_t streamed = 0;
  const bool grid = g.inv_cell > 0.0f;
  const bool all_bounded = WARM && grid && !__any( active & !m.found );
  CellBox full, core, cur, prev;
  if( all_bounded ) { full.x0 = full.y0 = full.z0 = 0; full.x1 = g.w - 1; full.y1 = g.h - 1; full.z1 = g.d - 1; core = full; }   // only clips reach_box below
  else
  {
    const TileBounds tb = wave_bounds( active, qx, qy, qz );
    full = cell_box( g, tb, radius );
    if( box_empty( full ) ) return m;
    core = cell_box( g, tb, 0.0f );
    core = box_grow( core, 0, full );
    if( box_empty( core ) ) core = full;           // the tile lies outside the grid but within reach of it
  }
  cur = core; prev = core;
  bool have_prev = false;
  if( BOUNDED_ONLY && !all_bounded ) { *handoff = true; return m; }
  if( all_bounded )
  {
    // Every lane starts from a genuine candidate (ICP iterations >= 2): whatever can beat or precede it
    // lies within its distance, so ONE sweep of the cells those small boxes touch settles the tile —
    // no shells, no cover test.
    auto step = [&]( const float4& X, const float4& Y, const float4& Z, int k4 )
    { consider4<GATED, WARM>( X, Y, Z, k4, L, qx, qy, qz, nx, ny, nz, tmin, bound, m, seen_closer ); };
    const float reach = reach_of( m, radius );
    // (Handing off per-row sweeps with a long row as well — 160 to 320 candidates for one row of 16 lanes — changed nothing.)
    const int swept = by_rows ? sweep_by_rows<GATED>( g, full, active, reach, qx, qy, qz, L, lane, step, streamed ) : 0;
    if( swept == 1 ) cur = full;   // (cur only clips the rank pass's own box)
    else
    {
      cur = reach_box( g, full, active, reach, qx, qy, qz );
      if( !box_empty( cur ) )
      {
        const uint32_t st = sweep_shell<GATED>( g, cur, cur, false, L, lane, 0, 1, step, ( BOUNDED_ONLY && bounded_give_up_total > 0 ) ? (uint32_t)bounded_give_up_total : 0xffffffffu );
        // hundreds of candidates past all 64 lanes of a lone wave (35-60 us: the launch's tail): the cooperative kernel does it with a workgroup
        if( BOUNDED_ONLY && st == 0xffffffffu ) { *handoff = true; return m; }
        streamed += st;
      }
    }
    if( dbg_unsettled ) { dbg_unsettled[1] = (int)streamed; dbg_unsettled[3] = 1; }
    sweeps = 1;
  }
  else if( !BOUNDED_ONLY )
  {
  // shells: the tile's own cells first (they hold the nearest candidates, so the per-lane bounds are
  // tight before the bulk arrives), then grown by 1, 2, 4, ... cells.  Of each shell only the part within
  // reach of a lane that is still unsettled is swept: for such a lane every cell of grow(core,k) that its
  // own box [q - reach, q + reach] touches has then been examined (its reach only shrinks), which is all the
  // cover test below relies on.
  bool unsettled = active;
  // (CULL — the scene-space score batch, whose tiles are a few centimetres across: the tile's own cells and the first ring in ONE sweep;
  //  a sweep of a dozen candidates costs its set-up, not its candidates)
  for( int k = ( CULL && grid ) ? 1 : 0; ; k = k ? 2 * k : 1 )
  {
    cur = grid ? box_grow( core, k, full ) : full;
    Cull cl{};
    if( CULL && grid ) cl = cull_of( g, unsettled, reach(), qx, qy, qz );
    // (CULL: the box from the lanes' common box and farthest reach — seven wave reductions instead of thirteen; the rows are clipped to the ball anyway)
    const CellBox out = !grid ? full : CULL ? cull_cells( g, cl, cur ) : reach_box( g, cur, unsettled, reach(), qx, qy, qz );
    if( dbg_unsettled && sweeps < 5 ) { dbg_unsettled[4 + 2 * sweeps] = __popcll( __ballot( unsettled ) ); dbg_unsettled[5 + 2 * sweeps] = -(int)streamed; }     // [4 + 2 s]: lanes shell s is swept for, [5 + 2 s]: candidates it streamed
    if( !box_empty( out ) )
    {
      streamed += sweep_shell<GATED>( g, out, prev, have_prev, L, lane, 0, 1, [&]( const float4& X, const float4& Y, const float4& Z, int k4 )
      { consider4<GATED, WARM>( X, Y, Z, k4, L, qx, qy, qz, nx, ny, nz, tmin, bound, m, seen_closer ); }, 0xffffffffu, CULL && grid, cl );
    }
    if( dbg_unsettled ) { dbg_unsettled[1] = (int)streamed; if( sweeps < 5 ) dbg_unsettled[5 + 2 * sweeps] += (int)streamed; }
    ++sweeps;
    if( n_sweeps ) *n_sweeps = sweeps;
    if( box_same( cur, full ) ) break;
    // a lane is settled when nothing outside `cur` can precede its match (or reach it at all: beyond the radius)
    const float cov = box_cover( g, cur, full, qx, qy, qz );
    const bool settled = !active | ( cov >= radius ) | ( m.found & ( cov > 0.0f ) & ( m.d2 < cov * cov ) );
    unsettled = !settled;
    if( dbg_unsettled ) { if( k == 1 ) dbg_unsettled[0] = __popcll( __ballot( !settled ) ); dbg_unsettled[1] = (int)streamed; dbg_unsettled[3] += 1; }
    if( !__any( !settled ) ) break;
    // Unsettled in a populated neighbourhood, or facing a shell of many cell rows: the rest of the
    // box is heavy (or latency-bound for one wave), let a whole workgroup do it.
    if( handoff && max_stages != 0x7fffffff )
    {
      const int kn = k ? 2 * k : 1;
      const CellBox nxt = box_grow( core, kn, full );
      const int next_rows = ( nxt.y1 - nxt.y0 + 1 ) * ( nxt.z1 - nxt.z0 + 1 );
      const int thr = max_stages & 0xffff, k_always = max_stages >> 16;     // k_always: hand off whenever still unsettled after shell k >= that (0: never)
      if( streamed >= (uint32_t)thr || next_rows > 2 * WAVE || ( k_always && k >= k_always ) ) { *handoff = true; return m; }
    }
    prev = cur; have_prev = true;
  }
  }

  if( K > 1 || GATED )
  {
    // Every candidate that precedes the final match was counted in seen_closer (it was no
    // farther than the then-best, which the final match precedes or equals), and so was the
    // match itself: seen_closer - 1 >= rank (WARM: a starting candidate does not count itself, so
    // only seen_closer >= rank holds).  Only when that bound does not settle rank < K,
    // count exactly (every such candidate lies inside `cur`: it is closer than the match).
    bool need_rank = m.found && ( seen_closer - ( WARM ? 0 : 1 ) >= K );
    if( __any( need_rank ) )
    {
      int rank = 0;
      Cull cl{};
      if( CULL && grid ) cl = cull_of( g, need_rank, reach_of( m, radius ), qx, qy, qz );
      const CellBox rb = ( CULL && grid ) ? cull_cells( g, cl, cur ) : reach_box( g, cur, need_rank, reach_of( m, radius ), qx, qy, qz );
      uint32_t rs = 0;
      RankBands rbands = rank_bands( m );
      if( !box_empty( rb ) )
      rs = sweep_shell<false>( g, rb, rb, false, L, lane, 0, 1, [&]( const float4& X, const float4& Y, const float4& Z, int k4 )
      {
        if( WARM ) { const int c = precede4_bands( X, Y, Z, k4, L, qx, qy, qz, radius_sq, m.d2, m.idx, rbands ); rank += need_rank ? c : 0; }
        else rank += need_rank ? precede4( X, Y, Z, k4, L, qx, qy, qz, radius_sq, m.d2, m.idx ) : 0;
      }, 0xffffffffu, CULL && grid, cl );
      if( dbg_unsettled ) { dbg_unsettled[2] = (int)rs; dbg_unsettled[15] = __popcll( __ballot( need_rank ) ); }
      if( need_rank && rank >= K ) { m.found = false; m.slot = -1; if( WARM ) m.rank_slack = rank_slack_of( rbands, K ); }
      ++sweeps;
      streamed += rs;
    }
  }
  if( n_sweeps ) *n_sweeps = sweeps;
  if( n_streamed ) *n_streamed = streamed;
  return m;
}

template <int NW>
struct CoopLds
{
  float m_d2[NW][WAVE];
  int   m_idx[NW][WAVE];
  float m_dot[NW][WAVE];
  int   m_slot[NW][WAVE];
  int   m_cnt[NW][WAVE];
  float m_fail[NW][WAVE];
  int   m_bands[NW][WAVE];   // rank pass: the four band counts of a wave's share, saturated at 31, 8 bits each
};

// The same staged search, done by all NW waves of a workgroup for ONE tile: every
// wave holds the same queries and sweeps its share of each shell's chunks; after every shell the
// per-lane bests are merged through LDS, so all waves take the same continue/stop decision and
// carry the tightest bound into the next shell.
template <bool GATED, int NW, bool WARM = false>
__device__ __forceinline__ Match coop_search( const GridView& g, bool active,
                                              float qx, float qy, float qz, float nx, float ny, float nz,
                                              float radius, float radius_sq, float tmin, int K,
                                              WaveLds& L, CoopLds<NW>& C, int wib, int lane, Match m /* starting candidate, see tile_search */,
                                              uint32_t* dbg_streamed = nullptr, unsigned long long* dbg_t = nullptr )
{
  uint32_t streamed = 0;
  int dbg_k = 0;
  if( dbg_t ) dbg_t[dbg_k++] = wall_clock64();
  const TileBounds tb = wave_bounds( active, qx, qy, qz );
  if( !tb.any ) return m;                          // identical in every wave of the workgroup
  const CellBox full = cell_box( g, tb, radius );
  if( box_empty( full ) ) return m;
  CellBox core = cell_box( g, tb, 0.0f );
  core = box_grow( core, 0, full );
  if( box_empty( core ) ) core = full;

  int seen_closer = 0;
  float bound = bound_of( active, radius_sq, m );
  CellBox cur = core, prev = core;
  bool have_prev = false;
  // shells: the tile's own cells first (they hold the nearest candidates, so the per-lane bounds are
  // tight before the bulk arrives), then grown by 1, 2, 4, ... cells
  // Queued tiles were unsettled after the first shells in a populated neighbourhood; most of them
  // have no match at all, so the ladder of small shells only adds row enumerations and barriers:
  // one shell of two cells, then the whole box.
  bool unsettled = active;
  // (WARM, round 5: a tile with a lane that starts from nothing — it had no match last iteration either, or lost it to the shrinking
  //  radius — almost always ends with that lane still unmatched, i.e. after the whole box: the two-cell shell first only adds a row
  //  enumeration, a chunk's round trip and two barriers to a tile that is ~20 us of dependent round trips and nothing else)
  const int k_first = ( WARM && __any( active & !m.found ) ) ? ( 1 << 20 ) : 2;      // (every wave holds the same 64 queries: the same decision)
  for( int k = k_first; ; k = 1 << 20 )
  {
    cur = ( g.inv_cell > 0.0f ) ? box_grow( core, k, full ) : full;
    // only the part of the shell within reach of a still-unsettled lane (see tile_search); identical in every wave
    const CellBox out = ( g.inv_cell > 0.0f ) ? reach_box( g, cur, unsettled, reach_of( m, radius ), qx, qy, qz ) : full;
    if( !box_empty( out ) )
    streamed += sweep_shell<GATED>( g, out, prev, have_prev, L, lane, wib, NW, [&]( const float4& X, const float4& Y, const float4& Z, int k4 )
    { consider4<GATED, WARM>( X, Y, Z, k4, L, qx, qy, qz, nx, ny, nz, tmin, bound, m, seen_closer ); } );
    if( dbg_t && dbg_k < 7 ) dbg_t[dbg_k++] = wall_clock64();
    // merge the per-lane bests of the waves; every wave continues with the merged best
    C.m_d2[wib][lane] = m.d2; C.m_idx[wib][lane] = m.idx; C.m_dot[wib][lane] = m.dot; C.m_slot[wib][lane] = m.found ? m.slot : -1;
    __syncthreads();
#pragma unroll
    for( int w = 0; w < NW; ++w )
    {
      const float d = C.m_d2[w][lane]; const int ix = C.m_idx[w][lane]; const int sl = C.m_slot[w][lane];
      if( sl >= 0 && lex_less( d, ix, m.d2, m.idx ) ) { m.d2 = d; m.idx = ix; m.dot = C.m_dot[w][lane]; m.slot = sl; m.found = true; }
    }
    bound = bound_of( active, radius_sq, m );
    __syncthreads();
    if( box_same( cur, full ) ) break;
    const float cov = box_cover( g, cur, full, qx, qy, qz );
    const bool settled = !active | ( cov >= radius ) | ( m.found & ( cov > 0.0f ) & ( m.d2 < cov * cov ) );
    unsettled = !settled;
    if( !__any( !settled ) ) break;                // same decision in every wave (same merged data)
    prev = cur; have_prev = true;
  }

  if( K > 1 || GATED )
  {
    // each wave's count bounds the rank contribution of its own share (see tile_search)
    C.m_cnt[wib][lane] = seen_closer; C.m_fail[wib][lane] = m.fail_max;
    __syncthreads();
    int seen_total = 0;
#pragma unroll
    for( int w = 0; w < NW; ++w ) { seen_total += C.m_cnt[w][lane]; m.fail_max = fmaxf( m.fail_max, C.m_fail[w][lane] ); }
    bool need_rank = m.found && ( seen_total - ( WARM ? 0 : 1 ) >= K );
    if( __any( need_rank ) )
    {
      int rank = 0;
      const CellBox rb = reach_box( g, cur, need_rank, reach_of( m, radius ), qx, qy, qz );
      RankBands rbands = rank_bands( m );
      if( !box_empty( rb ) )
      sweep_shell<false>( g, rb, rb, false, L, lane, wib, NW, [&]( const float4& X, const float4& Y, const float4& Z, int k4 )
      {
        if( WARM ) { const int c = precede4_bands( X, Y, Z, k4, L, qx, qy, qz, radius_sq, m.d2, m.idx, rbands ); rank += need_rank ? c : 0; }
        else rank += need_rank ? precede4( X, Y, Z, k4, L, qx, qy, qz, radius_sq, m.d2, m.idx ) : 0;
      } );
      __syncthreads();                             // everyone is done reading the counts
      C.m_cnt[wib][lane] = rank;
      if( WARM ) C.m_bands[wib][lane] = min( rbands.c1, 31 ) | ( min( rbands.c2, 31 ) << 8 ) | ( min( rbands.c3, 31 ) << 16 ) | ( min( rbands.c4, 31 ) << 24 );
      __syncthreads();
      rank = 0;
      int bands = 0;
#pragma unroll
      for( int w = 0; w < NW; ++w ) { rank += C.m_cnt[w][lane]; if( WARM ) bands += C.m_bands[w][lane]; }
      if( need_rank && rank >= K )
      {
        m.found = false; m.slot = -1;
        // (saturated sums decide "at least K" correctly while K <= 31, and NW * 31 fits the 8 bits)
        if( WARM && K <= 31 && NW <= 8 ) { rbands.c1 = bands & 255; rbands.c2 = ( bands >> 8 ) & 255; rbands.c3 = ( bands >> 16 ) & 255; rbands.c4 = ( bands >> 24 ) & 255; m.rank_slack = rank_slack_of( rbands, K ); }
      }
    }
  }
  if( dbg_streamed ) *dbg_streamed = streamed;
  if( dbg_t ) { while( dbg_k < 7 ) dbg_t[dbg_k++] = wall_clock64(); }
  return m;
}

} // namespace rs
