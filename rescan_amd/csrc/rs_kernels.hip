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
template <bool WITH_NOR, class F>
__device__ __forceinline__ uint32_t sweep_shell( const GridView& g, const CellBox& out, const CellBox& in, bool in_valid,
                                                 WaveLds& L, int lane, int share, int n_share, F&& f, uint32_t give_up_from = 0xffffffffu )
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
      const int y = out.y0 + ( r - rz * ny ), z = out.z0 + rz;
      const uint32_t* cs = g.cell_start + (size_t)( z * g.h + y ) * g.w;
      const bool inside = in_valid && y >= in.y0 && y <= in.y1 && z >= in.z0 && z <= in.z1;
      if( !inside ) { sa = cs[out.x0]; la = cs[out.x1 + 1] - sa; }
      else
      {
        // the row minus in's x-range (which may stick out of, or miss, out's)
        const int a1 = min( in.x0 - 1, out.x1 ), b0 = max( in.x1 + 1, out.x0 );
        if( a1 >= out.x0 ) { sa = cs[out.x0]; la = cs[a1 + 1] - sa; }
        if( b0 <= out.x1 ) { sb = cs[b0]; lb = cs[out.x1 + 1] - sb; }
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

// K-cap of a cold gated search (KCAP; the score batch): the consumers only walk the K NEAREST candidates, so once a lane has met K
// candidates closer than some distance, nothing at or beyond that distance can ever be its match (its rank would be >= K).  One
// such distance is tested, tau2 = a fixed fraction of radius²: a lane counts what it meets below it (cap_count) and, at K, lowers
// its bound to it — a match it may hold beyond is dropped (rank >= K, proven).  Exact, and it turns the lanes that have nothing
// compatible nearby — whose bound otherwise stays at the radius, which keeps the whole wave gating every candidate and sweeping
// the full box — into lanes with the reach of their K-th neighbour.
struct KCap { float tau2; int count; };
template <bool GATED, bool SELF, bool KCAP = false, class LDS>
__device__ __forceinline__ void consider4( const float4& X, const float4& Y, const float4& Z, int k, const LDS& L,
                                           float qx, float qy, float qz, float nx, float ny, float nz,
                                           float tmin, float& bound, Match& m, int& seen_closer, KCap* cap = nullptr, int K = 0 )
{
  float d[4];
  dist2x4( X, Y, Z, qx, qy, qz, d[0], d[1], d[2], d[3] );
  lanemask in[4];
#pragma unroll
  for( int i = 0; i < 4; ++i ) in[i] = RS_BALLOT( d[i] < bound );
  if( ( in[0] | in[1] | in[2] | in[3] ) == 0ull ) return;
  if( KCAP )
  {
    // (a candidate below tau2 that is not below the lane's bound any more is closer than nothing the lane still cares about:
    //  not counting it only delays the cap)
#pragma unroll
    for( int i = 0; i < 4; ++i ) count_lanes( cap->count, in[i] & RS_BALLOT( d[i] < cap->tau2 ) );
    const bool capit = cap->count >= K && bound > cap->tau2;
    if( RS_BALLOT( capit ) != 0ull )                     // (uniform: the lane masks below are taken with every lane present)
    {
      const bool drop = capit && m.found && !( m.d2 < cap->tau2 );
      bound = capit ? cap->tau2 : bound;
      m.found = drop ? false : m.found; m.slot = drop ? -1 : m.slot; m.d2 = drop ? INFINITY : m.d2; m.idx = drop ? INT_MAX : m.idx;
#pragma unroll
      for( int i = 0; i < 4; ++i ) in[i] = RS_BALLOT( d[i] < bound );
      if( ( in[0] | in[1] | in[2] | in[3] ) == 0ull ) return;
    }
  }
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
template <bool GATED, bool WARM = false, bool BOUNDED_ONLY = false, bool KCAP = false>
__device__ __forceinline__ Match tile_search( const GridView& g, bool active,
                                              float qx, float qy, float qz, float nx, float ny, float nz,
                                              float radius, float radius_sq, float tmin, int K,
                                              WaveLds& L, int lane, int max_stages, bool* handoff, int* dbg_unsettled,
                                              Match m /* starting candidate: empty, or a genuine one (within radius, gate passed) that only tightens the bounds */,
                                              int* n_sweeps = nullptr /* out: shells swept + rank pass: the tile's cost class */,
                                              bool by_rows = false /* WARM: sweep_by_rows for tiles whose lanes all start from a candidate */,
                                              uint32_t* n_streamed = nullptr /* out: candidates streamed, rank pass included */,
                                              int bounded_give_up_total = 0 /* BOUNDED_ONLY: hand a bounded tile off too when, swept tile-wide, the first 64 cell rows of its box hold this many candidates (0: never) */,
                                              float kcap_frac = 0.5f /* KCAP: tau² / radius² */ )
{
  if( handoff ) *handoff = false;
  int sweeps = 0;
  if( !__any( active ) ) return m;
  int seen_closer = 0;   // candidates that were no farther than the best-so-far when they were met
  float bound = bound_of( active, radius_sq, m );
  KCap cap; cap.tau2 = radius_sq * kcap_frac; cap.count = 0;
  // how far a lane still has to look: to its match, or — without one — as far as its bound lets anything in (the radius, or the K-cap)
  auto reach = [&]() -> float { return ( KCAP && !m.found && active ) ? sqrtf( bound ) * 1.0001f + 1e-5f : reach_of( m, radius ); };
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
  for( int k = 0; ; k = k ? 2 * k : 1 )
  {
    cur = grid ? box_grow( core, k, full ) : full;
    const CellBox out = grid ? reach_box( g, cur, unsettled, reach(), qx, qy, qz ) : full;
    if( dbg_unsettled && sweeps < 5 ) { dbg_unsettled[4 + 2 * sweeps] = __popcll( __ballot( unsettled ) ); dbg_unsettled[5 + 2 * sweeps] = -(int)streamed; }     // [4 + 2 s]: lanes shell s is swept for, [5 + 2 s]: candidates it streamed
    if( !box_empty( out ) )
      streamed += sweep_shell<GATED>( g, out, prev, have_prev, L, lane, 0, 1, [&]( const float4& X, const float4& Y, const float4& Z, int k4 )
      { consider4<GATED, WARM, KCAP>( X, Y, Z, k4, L, qx, qy, qz, nx, ny, nz, tmin, bound, m, seen_closer, &cap, K ); } );
    if( dbg_unsettled ) { dbg_unsettled[1] = (int)streamed; if( sweeps < 5 ) dbg_unsettled[5 + 2 * sweeps] += (int)streamed; }
    ++sweeps;
    if( n_sweeps ) *n_sweeps = sweeps;
    if( box_same( cur, full ) ) break;
    // a lane is settled when nothing outside `cur` can precede its match (or reach it at all: beyond the radius, or beyond its K-cap)
    const float cov = box_cover( g, cur, full, qx, qy, qz );
    const bool settled = !active | ( cov >= radius ) | ( m.found & ( cov > 0.0f ) & ( m.d2 < cov * cov ) ) |
                         ( KCAP & !m.found & ( cov > 0.0f ) & ( bound <= cov * cov ) );
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
      const CellBox rb = reach_box( g, cur, need_rank, reach_of( m, radius ), qx, qy, qz );
      uint32_t rs = 0;
      RankBands rbands = rank_bands( m );
      if( !box_empty( rb ) )
      rs = sweep_shell<false>( g, rb, rb, false, L, lane, 0, 1, [&]( const float4& X, const float4& Y, const float4& Z, int k4 )
      {
        if( WARM ) { const int c = precede4_bands( X, Y, Z, k4, L, qx, qy, qz, radius_sq, m.d2, m.idx, rbands ); rank += need_rank ? c : 0; }
        else rank += need_rank ? precede4( X, Y, Z, k4, L, qx, qy, qz, radius_sq, m.d2, m.idx ) : 0;
      } );
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

// ------------------------------------------------------------------------------------------
// Row-wise cold search (the score batch)
//
// tile_search sweeps, for all 64 queries of a tile, the union of their search regions: a 13 cm patch with a 10 cm radius is a
// 33 cm box, every candidate of which is tested by every lane (~840 evaluations per query in the score batch, most of them far
// outside the lane's own ball).  Here the wave's four ROWS of 16 lanes (16 Hilbert-consecutive queries: a 6.5 cm patch) each
// run their own shells around their own boxes, streaming their own candidates into their quarter of the wave's LDS arrays
// (like sweep_by_rows), and every lane keeps count of the candidates within its cover distance: once K of them are known,
// nothing unseen can be among the K nearest (the consumers only walk those), so a query without a gate-passing neighbour — most
// queries of a bad pose — stops at the distance of its K-th neighbour (~5.6 cm at 6 400 pts/m², K = 64) instead of the radius.
// At tile granularity that rule was useless (the cover distance of a lane at the rim of a 13 cm tile lags far behind); with
// 6.5 cm rows it fires.  Same results as tile_search (the K-nearest rule is the reference's own).
// ------------------------------------------------------------------------------------------
struct RowBox { int x0, x1, y0, y1, z0, z1; };            // per lane; identical within a row of 16 lanes
__device__ __forceinline__ bool rbox_empty( const RowBox& b ) { return ( b.x1 < b.x0 ) | ( b.y1 < b.y0 ) | ( b.z1 < b.z0 ); }
__device__ __forceinline__ bool rbox_same( const RowBox& a, const RowBox& b )
{ return a.x0 == b.x0 && a.x1 == b.x1 && a.y0 == b.y0 && a.y1 == b.y1 && a.z0 == b.z0 && a.z1 == b.z1; }

// cells that can hold a point within r of the row's masked lanes' boxes [q - reach, q + reach] (empty if no lane is masked)
__device__ __forceinline__ RowBox row_cell_box( const GridView& g, bool mask, float reach, float r, float qx, float qy, float qz )
{
  const float big = FLT_MAX;
  const float lx = row_min( mask ? qx - reach : big ), hx = row_max( mask ? qx + reach : -big );
  const float ly = row_min( mask ? qy - reach : big ), hy = row_max( mask ? qy + reach : -big );
  const float lz = row_min( mask ? qz - reach : big ), hz = row_max( mask ? qz + reach : -big );
  RowBox b;
  axis_range( lx, hx, r, g.minx, g.inv_cell, g.w, b.x0, b.x1 );
  axis_range( ly, hy, r, g.miny, g.inv_cell, g.h, b.y0, b.y1 );
  axis_range( lz, hz, r, g.minz, g.inv_cell, g.d, b.z0, b.z1 );
  if( hx < lx ) { b.x0 = 0; b.x1 = -1; }
  return b;
}
__device__ __forceinline__ RowBox rbox_grow( const RowBox& core, int k, const RowBox& full )
{
  RowBox b;
  b.x0 = max( core.x0 - k, full.x0 ); b.x1 = min( core.x1 + k, full.x1 );
  b.y0 = max( core.y0 - k, full.y0 ); b.y1 = min( core.y1 + k, full.y1 );
  b.z0 = max( core.z0 - k, full.z0 ); b.z1 = min( core.z1 + k, full.z1 );
  return b;
}
__device__ __forceinline__ RowBox rbox_clip( const RowBox& a, const RowBox& c )
{
  RowBox b;
  b.x0 = max( a.x0, c.x0 ); b.x1 = min( a.x1, c.x1 ); b.y0 = max( a.y0, c.y0 ); b.y1 = min( a.y1, c.y1 ); b.z0 = max( a.z0, c.z0 ); b.z1 = min( a.z1, c.z1 );
  return b;
}
__device__ __forceinline__ float rbox_cover( const GridView& g, const RowBox& cur, const RowBox& full, float qx, float qy, float qz )
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

// Every row streams (its out \ its in) through its quarter of the wave's LDS arrays, RB candidates per row and round (RB / 16
// loads in flight per lane: with 16 a round is four candidate groups per lane, too little work to cover the loads of the next):
// f( X, Y, Z, k ) for every group of four staged candidates k..k+3 of the calling lane's row (k differs between rows).
// Sentinels as in sweep_shell.  LDS = WaveLdsT<4 * RB>.
template <bool WITH_NOR, int RB, class LDS, class F>
__device__ __forceinline__ uint32_t sweep_rows_shell( const GridView& g, const RowBox& out, const RowBox& in, bool in_valid,
                                                      LDS& L, int lane, F&& f )
{
  constexpr int NF = RB / 16;
  const int l16 = lane & 15, tbase = lane & 48, sbase = ( lane >> 4 ) * RB;
  const int ny = out.y1 - out.y0 + 1, nz = out.z1 - out.z0 + 1;
  const int n_rows = rbox_empty( out ) ? 0 : ny * nz;
  const float inv_ny = 1.0f / (float)max( ny, 1 );
  uint32_t streamed = 0;
  for( int r0 = 0; __any( r0 < n_rows ); r0 += 16 )
  {
    const int r = r0 + l16;
    uint32_t sa = 0, la = 0, sb = 0, lb = 0;
    if( r < n_rows )
    {
      int rz = (int)( (float)r * inv_ny );
      rz -= ( rz * ny > r ) ? 1 : 0;
      rz += ( ( rz + 1 ) * ny <= r ) ? 1 : 0;
      const int y = out.y0 + ( r - rz * ny ), z = out.z0 + rz;
      const uint32_t* cs = g.cell_start + (size_t)( z * g.h + y ) * g.w;
      const bool inside = in_valid && y >= in.y0 && y <= in.y1 && z >= in.z0 && z <= in.z1;
      if( !inside ) { sa = cs[out.x0]; la = cs[out.x1 + 1] - sa; }
      else
      {
        const int a1 = min( in.x0 - 1, out.x1 ), b0 = max( in.x1 + 1, out.x0 );
        if( a1 >= out.x0 ) { sa = cs[out.x0]; la = cs[a1 + 1] - sa; }
        if( b0 <= out.x1 ) { sb = cs[b0]; lb = cs[out.x1 + 1] - sb; }
      }
    }
    const uint32_t incl = row_scan( la + lb );
    const uint32_t total = row_max_u( incl );                 // of this lane's row
    L.seg_a[lane] = sa; L.len_a[lane] = la; L.seg_b[lane] = sb; L.pre[lane] = incl - ( la + lb );
    wave_lds_fence();
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane( (int)total, 15 ), t1 = (uint32_t)__builtin_amdgcn_readlane( (int)total, 31 );
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane( (int)total, 47 ), t3 = (uint32_t)__builtin_amdgcn_readlane( (int)total, 63 );
    const uint32_t longest = max( max( t0, t1 ), max( t2, t3 ) );
    streamed += t0 + t1 + t2 + t3;
    float4 P[NF], N[NF]; uint32_t src[NF];
    auto fetch = [&]( uint32_t c0 )
    {
#pragma unroll
      for( int q = 0; q < NF; ++q )
      {
        const uint32_t j = c0 + (uint32_t)( 16 * q + l16 );
        P[q] = make_float4( FLT_MAX, FLT_MAX, FLT_MAX, 0.0f ); N[q] = make_float4( 0.0f, 0.0f, 0.0f, 0.0f ); src[q] = 0;
        if( j < total )
        {
          int row = 0;                                 // last cell row of this lane's row whose first candidate number is <= j
#pragma unroll
          for( int step = 8; step > 0; step >>= 1 ) { if( L.pre[tbase + row + step] <= j ) row += step; }
          const uint32_t off = j - L.pre[tbase + row];
          const uint32_t la_r = L.len_a[tbase + row];
          src[q] = ( off < la_r ) ? ( L.seg_a[tbase + row] + off ) : ( L.seg_b[tbase + row] + ( off - la_r ) );
          P[q] = g.pos[src[q]];
          if( WITH_NOR ) N[q] = g.nor[src[q]];
        }
      }
    };
    uint32_t c0 = 0;
    if( c0 < longest ) fetch( c0 );
    while( c0 < longest )
    {
#pragma unroll
      for( int q = 0; q < NF; ++q )
      {
        const int e = sbase + 16 * q + l16;
        L.px[e] = P[q].x; L.py[e] = P[q].y; L.pz[e] = P[q].z; L.pidx[e] = __float_as_int( P[q].w );
        if( WITH_NOR ) { L.nx[e] = N[q].x; L.ny[e] = N[q].y; L.nz[e] = N[q].z; }
        L.slot[e] = src[q];
      }
      wave_lds_fence();
      const uint32_t cn = c0 + (uint32_t)RB;
      if( cn < longest ) fetch( cn );                  // in flight during the evaluation
      const uint32_t left = longest - c0;
      const int n4 = left >= (uint32_t)RB ? RB / 4 : (int)( ( left + 3u ) >> 2 );
#pragma unroll 1
      for( int k4 = 0; k4 < n4; ++k4 )
      {
        const int k = sbase + 4 * k4;
        const float4 X = *reinterpret_cast<const float4*>( &L.px[k] );
        const float4 Y = *reinterpret_cast<const float4*>( &L.py[k] );
        const float4 Z = *reinterpret_cast<const float4*>( &L.pz[k] );
        f( X, Y, Z, k );
      }
      wave_lds_fence();
      c0 = cn;
    }
    wave_lds_fence();
  }
  if( g.evals && lane == 0 ) L.evals += streamed / 4;      // (each candidate is tested by 16 lanes, not 64)
  return streamed;
}

// The cold search of one tile, row by row.  Same result as tile_search<GATED>( ..., no hand-off, no starting candidate ).
template <bool GATED, int RB, class LDS>
__device__ __forceinline__ Match tile_search_rows( const GridView& g, bool active,
                                                   float qx, float qy, float qz, float nx, float ny, float nz,
                                                   float radius, float radius_sq, float tmin, int K, LDS& L, int lane )
{
  Match m = no_match();
  if( !__any( active ) ) return m;
  int seen_closer = 0, within_cover = 0;
  float bound = bound_of( active, radius_sq, m );
  const RowBox full = row_cell_box( g, active, 0.0f, radius, qx, qy, qz );
  RowBox core = rbox_grow( row_cell_box( g, active, 0.0f, 0.0f, qx, qy, qz ), 0, full );
  if( rbox_empty( core ) ) core = full;                // the row lies outside the grid but within reach of it
  RowBox cur = core, prev = core;
  bool have_prev = false, row_done = rbox_empty( full );
  bool unsettled = active & !row_done;
  for( int k = 0; ; ++k )
  {
    cur = row_done ? cur : rbox_grow( core, k, full );
    RowBox out = rbox_clip( row_cell_box( g, unsettled, reach_of( m, radius ), 0.0f, qx, qy, qz ), cur );
    if( row_done ) { out.x0 = 0; out.x1 = -1; }
    const float cov = rbox_cover( g, cur, full, qx, qy, qz );
    const float cov_sq = cov > 0.0f ? cov * cov : 0.0f;
    sweep_rows_shell<GATED, RB>( g, out, prev, have_prev, L, lane, [&]( const float4& X, const float4& Y, const float4& Z, int k4 )
    {
      // how many candidates of this shell lie within the lane's cover distance (a lower bound of those within it overall)
      float d0, d1, d2, d3;
      dist2x4( X, Y, Z, qx, qy, qz, d0, d1, d2, d3 );
      within_cover += ( d0 < cov_sq ? 1 : 0 ) + ( d1 < cov_sq ? 1 : 0 ) + ( d2 < cov_sq ? 1 : 0 ) + ( d3 < cov_sq ? 1 : 0 );
      consider4<GATED, false>( X, Y, Z, k4, L, qx, qy, qz, nx, ny, nz, tmin, bound, m, seen_closer );
    } );
    if( !row_done )
    {
      const bool at_full = rbox_same( cur, full );
      bool settled = !active | at_full | ( cov >= radius ) | ( m.found & ( cov > 0.0f ) & ( m.d2 < cov * cov ) );
      if( !settled && within_cover >= K )
      {
        // K candidates within the cover distance: nothing unseen can be among the K nearest, and neither can a match beyond it
        settled = true; m.found = false; m.slot = -1; bound = -1.0f;
      }
      unsettled = !settled;
      const bool none_left = row_max_u( unsettled ? 1u : 0u ) == 0u;
      row_done = at_full | none_left;
      prev = cur; have_prev = true;
    }
    if( !__any( !row_done ) ) break;
  }
  if( K > 1 || GATED )
  {
    // the exact rank where the running count does not settle it (see tile_search); everything closer than a match lies inside
    // the row's swept box: a match is only kept when it is closer than the cover distance or the row swept its whole region
    bool need_rank = m.found && ( seen_closer - 1 >= K );
    if( __any( need_rank ) )
    {
      int rank = 0;
      const RowBox rb = rbox_clip( row_cell_box( g, need_rank, reach_of( m, radius ), 0.0f, qx, qy, qz ), cur );
      sweep_rows_shell<false, RB>( g, rb, rb, false, L, lane, [&]( const float4& X, const float4& Y, const float4& Z, int k4 )
      { rank += need_rank ? precede4( X, Y, Z, k4, L, qx, qy, qz, radius_sq, m.d2, m.idx ) : 0; } );
      if( need_rank && rank >= K ) { m.found = false; m.slot = -1; }
    }
  }
  return m;
}

// Merge slots of the cooperative search.
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
  for( int k = 2; ; k = 1 << 20 )
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

// ------------------------------------------------------------------------------------------
// ICP: correspondence search  (lib/rs/icp.h:339-391)
// ------------------------------------------------------------------------------------------

// A kernel's view of ITS problem.  One source for the whole batch (rs_hip_icp_align_batch: n start poses of one cloud): the
// problem's rows of the per-point / per-tile arrays begin at prob * n; a multi-source batch (rs_hip_icp_align_multi: the
// per-placement refine loop of lib/rs/rs_database.h:220-230, every problem its own cloud) carries one IcpProblem per problem.
__device__ __forceinline__ void icp_bind( IcpLaunch& L, int prob )
{
  if( L.multi )
  {
    const IcpProblem& P = L.multi[prob];
    L.src = P.src; L.by_orig = P.by_orig; L.pt_off = P.pt_off; L.tile_off = P.tile_off; L.heavy_off = P.heavy_off;
  }
  else
  {
    L.pt_off = (long long)prob * L.src.n; L.tile_off = (long long)prob * L.src.n_tiles;
    L.heavy_off = (long long)( (size_t)prob * heavy_stride( L.src.n_tiles ) );
  }
}

// Source point i of problem `prob` in the target's frame (icp.h:339-347).
__device__ __forceinline__ void icp_query( const IcpLaunch& L, const Xform& T1, int i, bool active,
                                           float& qx, float& qy, float& qz, float& nx, float& ny, float& nz )
{
  qx = qy = qz = nx = ny = nz = 0.0f;
  if( active )
  {
    float4 p = L.src.pos[i], n = L.src.nor[i];
    float tx, ty, tz;
    xform3( T1, p.x, p.y, p.z, 1.0f, tx, ty, tz );   xform3( L.T2i, tx, ty, tz, 1.0f, qx, qy, qz );
    xform3( T1, n.x, n.y, n.z, 0.0f, tx, ty, tz );   xform3( L.T2i, tx, ty, tz, 0.0f, nx, ny, nz );
  }
}

// Warm start (iterations >= 2): last iteration's match of this source point, re-evaluated under
// the current pose.  If it is still within the radius and passes the gate it is a legitimate
// candidate, so starting the search from it changes nothing in the result and lets most
// candidates fail the very first compare.
// First iteration: no previous match to start from.  The first few points of the query's OWN cell serve the
// same purpose: whichever of them lies within the radius and passes the gate is a legitimate candidate, and
// starting from it turns the cold search (every candidate within the radius passes the bound test of a lane
// without a match) into the bounded one the later iterations run.
template <bool GATED>
__device__ __forceinline__ Match cell_seed( const GridView& g, bool active, float qx, float qy, float qz, float nx, float ny, float nz,
                                            float radius_sq, float tmin )
{
  Match m = no_match();
  if( !active || !( g.inv_cell > 0.0f ) ) return m;
  const float fx = floorf( ( qx - g.minx ) * g.inv_cell ), fy = floorf( ( qy - g.miny ) * g.inv_cell ), fz = floorf( ( qz - g.minz ) * g.inv_cell );
  if( !( fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < (float)g.w && fy < (float)g.h && fz < (float)g.d ) ) return m;     // outside the grid (or NaN)
  const size_t id = ( (size_t)(int)fz * g.h + (int)fy ) * g.w + (int)fx;
  const uint32_t s0 = g.cell_start[id], s1 = g.cell_start[id + 1];
  const uint32_t n = min( s1 - s0, 4u );
  for( uint32_t t = 0; t < n; ++t )
  {
    const uint32_t s = s0 + t;
    const float4 P = g.pos[s];
    const float vx = P.x - qx, vy = P.y - qy, vz = P.z - qz;
    const float d2 = vx * vx + vy * vy + vz * vz;
    float dc = 0.0f;
    bool ok = d2 < radius_sq;
    if( GATED )
    {
      const float4 N = g.nor[s];
      const float dot = N.x * nx + N.y * ny + N.z * nz;
      dc = dot > 0.0f ? dot : 0.0f;
      ok = ok && dc >= tmin && dc <= 1.0f;
    }
    const int idx = __float_as_int( P.w );
    if( ok && lex_less( d2, idx, m.d2, m.idx ) ) { m.d2 = d2; m.idx = idx; m.dot = dc; m.slot = (int)s; m.found = true; }
  }
  return m;
}
__device__ __forceinline__ Match icp_cell_seed( const IcpLaunch& L, bool active, float qx, float qy, float qz, float nx, float ny, float nz )
{ return cell_seed<true>( L.tgt, active, qx, qy, qz, nx, ny, nz, L.radius_sq, L.gate_tmin ); }

__device__ __forceinline__ Match icp_warm_start( const IcpLaunch& L, int prob, int i, bool active,
                                                 float qx, float qy, float qz, float nx, float ny, float nz )
{
  Match m = no_match();
  if( !active ) return m;
  if( !L.warm ) return L.seed ? icp_cell_seed( L, active, qx, qy, qz, nx, ny, nz ) : m;
  const int s = L.m_slot[(size_t)L.pt_off + i];
  if( s < 0 ) return m;
  const float4 P = L.tgt.pos[s], N = L.tgt.nor[s];
  float vx = P.x - qx, vy = P.y - qy, vz = P.z - qz;
  float d2 = vx * vx + vy * vy + vz * vz;
  float dot = N.x * nx + N.y * ny + N.z * nz;
  float dc = dot > 0.0f ? dot : 0.0f;
  if( d2 < L.radius_sq && dc >= L.gate_tmin && dc <= 1.0f ) { m.d2 = d2; m.idx = __float_as_int( P.w ); m.dot = dc; m.slot = s; m.found = true; }
  return m;
}

// Certificates.  Every search ends knowing, for its query, a distance within which EVERY candidate failed the
// gate, and by how much:
//   cert_r   = (distance to the nearest gated candidate — the match, or one rejected for its rank — or the radius
//              when there is none) - margin.  Each later iteration subtracts how far the query has moved since;
//   cert_dot = tmin - fail_max - margin: how much the gate value of any of those candidates may still rise.  Each
//              later iteration subtracts |delta n| * max|m|  (dot(m, n') - dot(m, n) <= |m| |n' - n|).
// A query whose previous match is no longer usable (there was none, or it left the shrinking radius —
// icp.h:493 — or its gate) consults the certificate: while radius <= cert_r and cert_dot >= 0 the triangle
// inequality proves that every candidate within the current radius was examined then and cannot pass the gate
// now, so the result (unmatched) is exactly what the full sweep would return, and the sweep is skipped.  The
// margins (1e-4 m, 1e-5) are orders of magnitude above the fp32 rounding of dist², dot and the displacement.
// This removes the one case a search cannot bound — the full-radius sweep of a source point with nothing to
// match, repeated every iteration — including the points that become unmatched because the radius shrinks.
//
// Returns whether the search of this query may be skipped, and writes the aged certificate back at once
// (keeping it in registers across the search costs a wave of occupancy).  A tile that phase A hands off
// is aged a second time by the cooperative kernel: that only makes the certificate more conservative.
__device__ __forceinline__ bool icp_certificate( const IcpLaunch& L, int prob, int i, bool active,
                                                 float qx, float qy, float qz, float nx, float ny, float nz )
{
  if( !L.cert_r || !L.warm || !active ) return false;
  const size_t o = (size_t)L.pt_off + i;
  const float r = L.cert_r[o];
  if( !( r > 0.0f ) ) return false;
  // the same query under the previous iteration's pose (identical float operations as then)
  Xform Tp;
#pragma unroll
  for( int k = 0; k < 16; ++k ) Tp.m[k] = L.T1_prev[prob * 16 + k];
  float4 p = L.src.pos[i], n = L.src.nor[i];
  float tx, ty, tz, px, py, pz, mx, my, mz;
  xform3( Tp, p.x, p.y, p.z, 1.0f, tx, ty, tz );   xform3( L.T2i, tx, ty, tz, 1.0f, px, py, pz );
  xform3( Tp, n.x, n.y, n.z, 0.0f, tx, ty, tz );   xform3( L.T2i, tx, ty, tz, 0.0f, mx, my, mz );
  const float dq = sqrtf( ( qx - px ) * ( qx - px ) + ( qy - py ) * ( qy - py ) + ( qz - pz ) * ( qz - pz ) );
  const float dn = sqrtf( ( nx - mx ) * ( nx - mx ) + ( ny - my ) * ( ny - my ) + ( nz - mz ) * ( nz - mz ) );
  const float moved = dq * 1.0001f + 1e-5f;
  const float r_now = r - moved;
  const float dot_now = L.cert_dot[o] - ( dn * L.tgt_nor_max * 1.0001f + 1e-6f );
  bool skip = ( L.radius <= r_now ) & ( dot_now >= 0.0f );
  if( L.cert_slack )
  {
    // rank certificate: the nearest gated candidate (at cert_r) was rejected because K candidates precede it, and K of
    // them lie closer than cert_r - slack.  They have come closer to the query by at most `moved`, whatever passes
    // the gate now lies no closer than r_now = cert_r - moved (everything inside failed it, by the margin above): while
    // slack - 2 moved > 0 those K still precede every candidate that could be chosen, so the point stays unmatched.
    const float s = L.cert_slack[o];
    const float s_now = s > 0.0f ? s - 2.0f * moved : 0.0f;
    skip |= ( s_now > 0.0f ) & ( dot_now >= 0.0f ) & ( r_now > 0.0f );
    L.cert_slack[o] = s_now;
  }
  L.cert_r[o] = skip ? r_now : -1.0f;
  L.cert_dot[o] = dot_now;
  return skip;
}

__device__ __forceinline__ void icp_emit( const IcpLaunch& L, int prob, int tile, int i, bool active, int lane, const Match& m,
                                          bool skipped )
{
  const size_t o = (size_t)L.pt_off + i;
  if( active ) { L.m_slot[o] = m.found ? m.slot : -1; if( !L.rec ) { L.m_d2[o] = m.d2; L.m_dot[o] = m.dot; } }
  if( active && L.rec )
  {
    // the correspondence as the estimator wants it, at the source point's ORIGINAL index: three 16-byte stores into one 48-byte
    // record (the estimator's kernels then read the reference's order coalesced, instead of gathering slot / dist² / dot /
    // source / target point / target normal at random: 5 transactions per point)
    Xform T1;
#pragma unroll
    for( int k = 0; k < 16; ++k ) T1.m[k] = L.T1[prob * 16 + k];
    float qx, qy, qz, nx, ny, nz;
    icp_query( L, T1, i, true, qx, qy, qz, nx, ny, nz );           // (the same float operations the search used)
    float4 P = make_float4( 0.0f, 0.0f, 0.0f, 0.0f ), N = P;
    if( m.found ) { P = L.tgt.pos[m.slot]; N = L.tgt.nor[m.slot]; }
    const int orig = __float_as_int( L.src.pos[i].w );
    float4* R = L.rec + ( (size_t)L.pt_off + orig ) * REC_F4;
    R[0] = make_float4( qx, qy, qz, m.found ? m.d2 : -1.0f );
    R[1] = make_float4( P.x, P.y, P.z, m.dot );
    R[2] = make_float4( N.x, N.y, N.z, 0.0f );
  }
  if( active && L.cert_r && !skipped )
  {
    // fresh certificate (m.idx != INT_MAX: a gated candidate exists at dist² m.d2, even if its rank rejected it)
    const float r = ( m.idx != INT_MAX ? sqrtf( m.d2 ) : L.radius ) - 1e-4f;
    const float d = L.gate_tmin - m.fail_max - 1e-5f;
    L.cert_r[o] = ( d >= 0.0f ) ? r : -1.0f; L.cert_dot[o] = d;
    if( L.cert_slack ) L.cert_slack[o] = ( !m.found && m.idx != INT_MAX ) ? m.rank_slack : 0.0f;
  }
  if( RS_DBG >= 2 && DBG( L ) )
  {
    unsigned long long* cat = DBG( L ) + 6 * (size_t)L.src.n_tiles;
    const bool margin = L.gate_tmin - m.fail_max - 1e-5f >= 0.0f;
    const int c_skip = __popcll( __ballot( skipped ) ), c_fresh = __popcll( __ballot( active && !skipped && !m.found && margin ) );
    const int c_rank = __popcll( __ballot( active && !skipped && !m.found && m.idx != INT_MAX ) );
    const int c_loose = __popcll( __ballot( active && !skipped && !m.found && !margin && m.idx == INT_MAX ) );
    if( lane == 0 ) { atomicAdd( cat + 0, (unsigned long long)c_skip ); atomicAdd( cat + 1, (unsigned long long)c_fresh ); atomicAdd( cat + 2, (unsigned long long)c_rank ); atomicAdd( cat + 3, (unsigned long long)c_loose ); }
  }
  // statistics of dist² over the correspondences (msh_compute_mean/stddev, msh_std.h:1800-1825): Σ1, Σd², Σd⁴ of the
  // tile, added as INTEGERS (fixed point, scaled to the radius) to one of STAT_SHARDS accumulators — integer addition
  // does not care in which order the tiles arrive, so the totals are bit-reproducible without a fixed-order pass
  // (and without the kernel launch that pass used to be).
  if( L.stat_acc )
  {
    const unsigned long long c = (unsigned long long)__popcll( __ballot( active && m.found ) );
    double s1 = ( active && m.found ) ? (double)m.d2 : 0.0;
    double s2 = ( active && m.found ) ? (double)( m.d2 * m.d2 ) : 0.0;
    s1 = wave_sum( s1 ); s2 = wave_sum( s2 );
    if( lane == 0 && c != 0 )
    {
      unsigned long long* a = L.stat_acc + ( (size_t)prob * STAT_SHARDS + ( tile & ( STAT_SHARDS - 1 ) ) ) * 4;
      atomicAdd( a + 0, c );
      atomicAdd( a + 1, (unsigned long long)( s1 * L.stat_s1 ) );
      atomicAdd( a + 2, (unsigned long long)( s2 * L.stat_s2 ) );
    }
  }
}

// What has to be reset between two searches of a problem (done by the workgroup that ends the iteration).
// Called by EVERY thread of the workgroup that ends an iteration (at least one full wave).
__device__ __forceinline__ void icp_iteration_reset( const IcpLaunch& L, int prob )
{
  if( threadIdx.x == 0 )
  {
    if( L.queued ) L.queued[prob] = L.queue_count[prob];       // tiles phase A handed off (diagnostics)
    L.queue_count[prob] = 0;                                   // ready for the next iteration's phase A
  }
  const int stat_wave = blockDim.x >= 2 * WAVE ? WAVE : 0;      // (not the wave whose first thread goes on to solve: its time is the iteration's)
  if( L.heavy_out && (int)threadIdx.x >= stat_wave && (int)threadIdx.x < stat_wave + WAVE )
  {
    const int sl = (int)threadIdx.x - stat_wave;
    // What this iteration's phase A streamed per tile, on average (a sample of 256 tiles, by the first wave): the next launch's
    // "slow tile" thresholds are absolute numbers of candidates (a lone wave's time) up to a mean of HEAVY_MEAN_REF and scale with
    // the mean beyond — on a target four times as dense EVERY tile streams four times as many, and a third of them went to the
    // cooperative kernel (2x the search time at 4 M points per scan).
    int* ho = L.heavy_out + (size_t)L.heavy_off;
    const int n_t = L.src.n_tiles, step = max( 1, n_t / ( 4 * WAVE ) );
    unsigned long long acc = 0;                          // sum << 32 | count
#pragma unroll
    for( int q = 0; q < 4; ++q )
    {
      const int t = ( q * WAVE + sl ) * step;
      const unsigned w = t < n_t ? (unsigned)ho[HEAVY_HDR + HEAVY_SLOTS + t] >> 2 : 0u;
      if( w ) acc += ( (unsigned long long)w << 32 ) | 1ull;
    }
    acc = wave_sum_u64( acc );
    if( sl == 0 ) { const unsigned cnt = (unsigned)acc; const unsigned mean = cnt ? (unsigned)( ( acc >> 32 ) / cnt ) : 0u; ho[HEAVY_MEAN] = (int)( mean * 256u / HEAVY_MEAN_REF ); }      // (x 256)
  }
  if( L.stat_acc )
    for( int k = threadIdx.x; k < STAT_SHARDS * 4; k += blockDim.x ) L.stat_acc[(size_t)prob * STAT_SHARDS * 4 + k] = 0ull;
  if( L.heavy_in && threadIdx.x == 0 )      // consumed: it is the next iteration's output buffer
    for( int k = 0; k < HEAVY_CLASSES; ++k ) const_cast<int*>( L.heavy_in )[(size_t)L.heavy_off + k] = 0;
}

#ifndef RS_XCD_MAP
#define RS_XCD_MAP 1
#endif
// workgroups of phase A's natural part per XCD class (see k_icp_corr)
// Waves per workgroup of phase A (k_icp_corr).  A workgroup's wave slots and LDS are released when its LAST wave ends, and a
// warm tile takes its wave 12 us at the median, 17 at the 90th percentile: with four tiles per workgroup a quarter of the slot time
// was spent waiting for the slowest of four (4 400 of 6 144 slots occupied in the launch's steady state, 5 000-5 200 with one; the
// concurrent chain's searches 2.22 -> 2.08 ms per step, serial 1.68 -> 1.65: profiles/r02/ab_*experiments.txt).
#ifndef RS_PA_WAVES
#define RS_PA_WAVES 1
#endif
constexpr int PA_WAVES = RS_PA_WAVES;
__host__ __device__ inline int icp_blocks_per_xcd( int n_tiles ) { return ( ( n_tiles + PA_WAVES - 1 ) / PA_WAVES + 7 ) / 8; }

// Phase A: one wave per tile, first shell(s) only; unsettled tiles are queued.
// Waves per SIMD the register allocation aims at.  5 = 83 / 95 VGPRs (warm / cold instantiation), no scratch; 6 = 80 VGPRs with
// 16 / 40 B of scratch per lane.  Measured on the bench with the consumers on disjoint CUs (interleaved repeats,
// profiles/r02/ab_*experiments.txt), step time / HBM-side traffic per search: both 6: 2.53-2.55 ms, 148 MB; warm 6, cold 5:
// 2.55-2.59 ms, 134 MB; both 5: 2.61-2.62 ms, 122 MB (70.5 MB are algorithmic).  The ICP chain is the step's critical path, so
// the warm launches (nine of ten) keep their sixth wave; the cold one, whose scratch is the larger, does without.
#ifndef RS_ICP_WARM_OCC
#define RS_ICP_WARM_OCC 6
#endif
#ifndef RS_ICP_OCC
#define RS_ICP_OCC 5
#endif
template <bool BOUNDED_ONLY>
__global__ __launch_bounds__( PA_WAVES * WAVE, BOUNDED_ONLY ? RS_ICP_WARM_OCC : RS_ICP_OCC ) void k_icp_corr( IcpLaunch L )
{
  RS_CHAIN_SETPRIO();
  __shared__ WaveLds lds[PA_WAVES];
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  icp_bind( L, prob );
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int wib = PA_WAVES == 1 ? 0 : uni( (int)threadIdx.x / WAVE );
  EvalScope eval_scope( L.tgt.evals, lds[wib], lane );
  // Slowest first: the kernel ends when its slowest tile does, and the slow tiles (several shells, a rank
  // pass) are the same from one iteration to the next.  The previous iteration listed them; the first
  // HEAVY_SLOTS wave slots of the grid take that list, the rest walk the tiles in their natural (Hilbert)
  // order — which the caches depend on — and skip the listed ones.
  int slot = blockIdx.x * PA_WAVES + wib;
  int tile;
  // XCD-aware order of the natural (Hilbert) part: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8
  // share one), each XCD with its own 4 MB L2.  Walking the tiles in plain order would have every XCD touch every part of
  // the 32 MB target cloud; instead XCD class c = b mod 8 walks the c-th eighth of the Hilbert order, so an XCD's L2 only
  // ever sees its own part of the scene (and the seams).  Which XCD a class lands on does not matter.
  auto natural_tile = [&]( int block ) -> int
  {
#if RS_XCD_MAP
    const int per = icp_blocks_per_xcd( L.src.n_tiles );
    if( ( block >> 3 ) >= per ) return INT_MAX;      // (a multi-source batch's grid is the largest problem's: beyond this problem's own eighths)
    return ( ( block & 7 ) * per + ( block >> 3 ) ) * PA_WAVES + wib;
#else
    return block * PA_WAVES + wib;
#endif
  };
  if( L.heavy_in )
  {
    const int* hv = L.heavy_in + (size_t)L.heavy_off;
    if( slot < HEAVY_SLOTS )
    {
      // front block b serves XCD class b mod 8 (it runs on the XCD the class's natural blocks run on), entry (b / 8) * 4 + wave
      const int c = (int)blockIdx.x & ( HEAVY_CLASSES - 1 ), p = ( (int)blockIdx.x >> 3 ) * PA_WAVES + wib;
      if( p >= min( uni( hv[c] ), HEAVY_PER_CLASS ) ) return;
      tile = uni( hv[HEAVY_HDR + c * HEAVY_PER_CLASS + p] );
    }
    else
    {
      tile = natural_tile( (int)blockIdx.x - HEAVY_SLOTS / PA_WAVES );
      if( tile >= L.src.n_tiles ) return;
      const int flag = uni( hv[HEAVY_HDR + HEAVY_SLOTS + tile] ) & 3;
      if( flag == 1 ) return;                                            // a front slot has it
      if( BOUNDED_ONLY && flag == 2 )
      {
        // a tile whose one sweep was so long that a lone wave IS the launch's tail (600+ candidates: 45-60 us, against a launch
        // that could end after ~45): a workgroup of the cooperative kernel takes it from now on, like an unbounded tile
        if( lane == 0 )
        {
          int q = atomicAdd( L.queue_count + prob, 1 ); L.queue[(size_t)L.tile_off + q] = tile;
          if( L.heavy_out ) L.heavy_out[(size_t)L.heavy_off + HEAVY_HDR + HEAVY_SLOTS + tile] = 2;
          if( DBG( L ) ) { DBG( L )[2 * tile] = wall_clock64(); DBG( L )[2 * tile + 1] = 1ull << 20; }      // (handed off, no time spent)
        }
        return;
      }
    }
  }
  else { tile = natural_tile( (int)blockIdx.x ); if( tile >= L.src.n_tiles ) return; }
  const int i = (int)L.src.tiles[tile] + lane;
  const bool active = i < (int)L.src.tiles[tile + 1];

  Xform T1;
#pragma unroll
  for( int k = 0; k < 16; ++k ) T1.m[k] = __int_as_float( uni( __float_as_int( L.T1[prob * 16 + k] ) ) );
  float qx, qy, qz, nx, ny, nz;
  const unsigned long long t_begin = DBG( L ) ? wall_clock64() : 0ull;
  icp_query( L, T1, i, active, qx, qy, qz, nx, ny, nz );
  bool handoff;
  int sweeps = 0;
  uint32_t streamed = 0;
  int unsettled[16] = { 0 };
  const Match init = icp_warm_start( L, prob, i, active, qx, qy, qz, nx, ny, nz );
  const bool search = active & !icp_certificate( L, prob, i, active & !init.found, qx, qy, qz, nx, ny, nz );
  // thresholds in candidates: as given up to a mean of HEAVY_MEAN_REF candidates per tile in the previous launch, growing with it
  // beyond (the factor, in 1/256ths, was worked out when that iteration ended: icp_iteration_reset)
  const unsigned scale_q8 = L.heavy_in ? (unsigned)uni( L.heavy_in[(size_t)L.heavy_off + HEAVY_MEAN] ) : 256u;
  const unsigned sq8 = scale_q8 < 256u ? 256u : ( scale_q8 > 65536u ? 65536u : scale_q8 );
  const int thr_total = (int)( ( (unsigned)min( L.heavy_total, 0xffff ) * sq8 ) >> 8 );
  const uint32_t thr_streamed = ( (unsigned)min( L.heavy_streamed, 0xffff ) * sq8 ) >> 8;
  const uint32_t thr_handoff = ( (unsigned)min( L.heavy_handoff, 0xffff ) * sq8 ) >> 8;
  Match m = tile_search<true, true, BOUNDED_ONLY>( L.tgt, search, qx, qy, qz, nx, ny, nz, L.radius, L.radius_sq, L.gate_tmin, L.K,
                               lds[wib], lane, L.solo_stages, &handoff, DBG( L ) ? unsettled : nullptr, init, &sweeps, L.by_rows != 0, &streamed, thr_total );
  if( L.heavy_out && lane == 0 )
  {
    int* hv = L.heavy_out + (size_t)L.heavy_off;
    int listed = 0;
    // What will be slow next time.  A warm launch hands its unbounded tiles off at once — they cost it nothing — and its slow
    // tiles are the ones that stream many candidates in their one sweep (p50 150 candidates / 12 us, p99.9 750 / 40 us: left in
    // natural order those start half way through the launch and ARE its tail); the cold launch's are its multi-shell tiles.
    const bool slow = BOUNDED_ONLY ? ( !handoff && streamed >= thr_streamed ) : ( handoff || sweeps >= 2 || streamed >= thr_streamed );
    if( BOUNDED_ONLY && !handoff && streamed >= thr_handoff ) listed = 2;
    else if( slow )
    {
#if RS_XCD_MAP
      const int c = min( ( tile / PA_WAVES ) / icp_blocks_per_xcd( L.src.n_tiles ), HEAVY_CLASSES - 1 );      // the class whose natural range holds the tile
#else
      const int c = ( tile / PA_WAVES ) & ( HEAVY_CLASSES - 1 );
#endif
      const int pos = atomicAdd( hv + c, 1 );
      if( pos < HEAVY_PER_CLASS ) { hv[HEAVY_HDR + c * HEAVY_PER_CLASS + pos] = tile; listed = 1; }
    }
    // per tile: flag (bits 0-1) | candidates streamed by this launch (handed off: 0, not counted) — one store; the mean over a sample
    // of these words is the next launch's yardstick (icp_iteration_reset).  (Summing them with atomics cost 33 us per launch.)
    hv[HEAVY_HDR + HEAVY_SLOTS + tile] = listed | ( handoff ? 0 : (int)( min( streamed, 0x0fffffffu ) << 2 ) );
  }
  if( DBG( L ) && lane == 0 )
  {
    // [0] start (absolute, 10 ns ticks) | [1] duration 20 bits | handoff 1 | shells 4 | streamed 16 | rank-pass streamed 16 | unsettled lanes after shell 1: 7
    auto clipv = []( unsigned long long v, unsigned long long mx ) { return v > mx ? mx : v; };
    DBG( L )[2 * tile] = t_begin;
    DBG( L )[2 * tile + 1] = clipv( wall_clock64() - t_begin, 0xfffff ) | ( (unsigned long long)( handoff ? 1 : 0 ) << 20 ) | ( clipv( unsettled[3], 15 ) << 21 ) |
                             ( clipv( unsettled[1], 0xffff ) << 25 ) | ( clipv( unsettled[2], 0xffff ) << 41 ) | ( clipv( unsettled[0], 127 ) << 57 );
  }
  if( handoff )
  {
    if( lane == 0 ) { int q = atomicAdd( L.queue_count + prob, 1 ); L.queue[(size_t)L.tile_off + q] = tile; }
    return;
  }
  icp_emit( L, prob, tile, i, active, lane, m, active & !search );
}

// One tile searched by all NW waves of its workgroup: queries, warm start, certificates (checked by ONE wave — the check ages
// the certificate in place, and the waves must agree on who searches), cooperative search, results written by wave 0.
// Returns whether some lane had to search without a starting candidate (the tile is "not bounded": worth a workgroup again
// next iteration).  Ends with a barrier: the merge slots may be reused at once.
template <int NW>
__device__ __forceinline__ bool icp_coop_tile( const IcpLaunch& L, const Xform& T1, int prob, int tile, WaveLds& lds, CoopLds<NW>& coop,
                                               unsigned long long& s_skip, int wib, int lane, int dbg_slot )
{
  const int i = (int)L.src.tiles[tile] + lane;
  const bool active = i < (int)L.src.tiles[tile + 1];
  float qx, qy, qz, nx, ny, nz;
  icp_query( L, T1, i, active, qx, qy, qz, nx, ny, nz );
  const unsigned long long t_begin = DBG( L ) ? wall_clock64() : 0ull;
  unsigned long long stamps[8];
  const Match init = icp_warm_start( L, prob, i, active, qx, qy, qz, nx, ny, nz );
  if( wib == 0 )
  {
    const unsigned long long skip_mask = __ballot( icp_certificate( L, prob, i, active & !init.found, qx, qy, qz, nx, ny, nz ) );
    if( lane == 0 ) s_skip = skip_mask;
  }
  __syncthreads();
  const bool search = active & !( ( s_skip >> lane ) & 1ull );
  uint32_t streamed = 0;
  Match m = coop_search<true, NW, true>( L.tgt, search, qx, qy, qz, nx, ny, nz, L.radius, L.radius_sq, L.gate_tmin, L.K,
                               lds, coop, wib, lane, init, DBG( L ) ? &streamed : nullptr, DBG( L ) ? stamps : nullptr );
  if( DBG( L ) && wib == 0 && dbg_slot >= 0 )
  {
    const int n_search = __popcll( __ballot( search ) ), n_unm = __popcll( __ballot( search & !m.found ) );
    if( lane == 0 )
    {
      unsigned long long* d = DBG( L ) + 2 * (size_t)L.src.n_tiles + 4 * (size_t)dbg_slot;
      const unsigned long long t_end = wall_clock64();
      // [0] total | [1] streamed | lanes | phases packed: setup, shell 1, shell 2, rest (each 16 bits, ticks of 10 ns)
      auto clip = []( unsigned long long v ) { return v > 0xffffull ? 0xffffull : v; };
      d[0] = t_end - t_begin; d[1] = streamed; d[2] = (unsigned long long)n_search | ( (unsigned long long)n_unm << 8 );
      d[3] = clip( stamps[0] - t_begin ) | ( clip( stamps[1] - stamps[0] ) << 16 ) | ( clip( stamps[2] - stamps[1] ) << 32 ) | ( clip( t_end - stamps[2] ) << 48 );
    }
  }
  if( wib == 0 ) icp_emit( L, prob, tile, i, active, lane, m, active & !search );
  const bool not_bounded = __any( search & !init.found );
  __syncthreads();                               // merge slots are reused by the next tile
  return not_bounded;
}

// Phase B: one workgroup per queued tile, whole box, chunks shared by its waves.
#ifndef RS_COOP_OCC
#define RS_COOP_OCC 5      // waves per SIMD the cooperative kernel's register allocation aims at (96 VGPRs: no spills; 6 = 80 VGPRs spilt 56 B per lane for no gain in time)
#endif
template <int NW>
__global__ __launch_bounds__( NW * WAVE, RS_COOP_OCC ) void k_icp_corr_coop( IcpLaunch L )
{
  RS_CHAIN_SETPRIO();
  __shared__ WaveLds lds[NW];
  __shared__ CoopLds<NW> coop;
  __shared__ unsigned long long s_skip;
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  icp_bind( L, prob );
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int wib = threadIdx.x / WAVE;
  EvalScope eval_scope( L.tgt.evals, lds[wib], lane );
  const int n_queued = L.coop_all ? L.src.n_tiles : L.queue_count[prob];     // coop_all: phase A was not launched, every tile is searched here
  Xform T1;
#pragma unroll
  for( int k = 0; k < 16; ++k ) T1.m[k] = L.T1[prob * 16 + k];
  for( int b = blockIdx.x; b < n_queued; b += gridDim.x )
  {
    const int tile = L.coop_all ? b : L.queue[(size_t)L.tile_off + b];
    icp_coop_tile<NW>( L, T1, prob, tile, lds[wib], coop, s_skip, wib, lane, b );
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
  icp_bind( L, prob );
  Xform T1;
#pragma unroll
  for( int k = 0; k < 16; ++k ) T1.m[k] = L.T1[prob * 16 + k];
  // n_corr, mean, stddev of dist² (icp.h:393-402) from the integer accumulators: every workgroup adds up the same
  // STAT_SHARDS integers, so all of them hold the same bits
  __shared__ unsigned long long s_stat[WAVES_PER_BLOCK][3];
  float sd = 0.0f;
  if( L.stat_acc )
  {
    static_assert( STAT_SHARDS == BLOCK, "one shard per thread" );
    const unsigned long long* a = L.stat_acc + ( (size_t)prob * STAT_SHARDS + threadIdx.x ) * 4;
    const unsigned long long c0 = wave_sum_u64( a[0] ), c1 = wave_sum_u64( a[1] ), c2 = wave_sum_u64( a[2] );
    if( ( threadIdx.x & ( WAVE - 1 ) ) == 0 ) { unsigned long long* o = s_stat[threadIdx.x / WAVE]; o[0] = c0; o[1] = c1; o[2] = c2; }
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
    for( int w = 0; w < WAVES_PER_BLOCK; ++w ) { t0 += s_stat[w][0]; t1 += s_stat[w][1]; t2 += s_stat[w][2]; }
    const double n = (double)t0;
    const float mean = (float)( (double)t1 * L.stat_i1 / n );           // sum / (float)n
    const float sqm = (float)( (double)t2 * L.stat_i2 / n );            // sq_sum / (float)n
    const float var = sqm - mean * mean;
    sd = (float)sqrt( (double)var );                                    // (float)sqrt( ... ), msh_std.h:1824
    if( blockIdx.x == 0 && threadIdx.x == 0 )
    {
      double* st = L.res + (size_t)prob * ICP_NRES + ICP_NMOM;
      st[0] = n; st[1] = mean; st[2] = sd; st[3] = (double)L.queue_count[prob];
    }
  }
  const bool use_sd = sd > 0.000001;
  const float cut = 2.5f * sd;

  double acc[ICP_NMOM];
#pragma unroll
  for( int k = 0; k < ICP_NMOM; ++k ) acc[k] = 0.0;

  for( int i = blockIdx.x * BLOCK + threadIdx.x; i < L.src.n; i += gridDim.x * BLOCK )
  {
    const size_t o = (size_t)L.pt_off + i;
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

// The rest of an iteration once the moments are summed in L.res (the whole workgroup comes in; thread 0 solves).
__device__ __forceinline__ void icp_update_tail( const IcpLaunch& L, int prob )
{
  double* res = L.res + (size_t)prob * ICP_NRES;
  if( !L.solve ) return;
  icp_iteration_reset( L, prob );
  if( threadIdx.x != 0 ) return;
  // ---- icp.h:455-493 for this problem ----
  L.prev_err[prob] = L.err[prob];
  L.iters[prob] += 1;
  if( res[ICP_NMOM] == 0.0 ) { L.active[prob] = 0; return; }            // icp.h:455-459: no correspondences
  Mat4 T;
  for( int k = 0; k < 16; ++k ) { T.m[k] = L.T1[prob * 16 + k]; L.T1_prev[prob * 16 + k] = T.m[k]; }
  float e;
  float cen[6];
  if( L.exact_centroids )
  {
    // the reference's own centroids: c = Σw·p * ( 1.0f / Σw ), every sum its sequential fp32 chain (icp.h:136-148)
    const double* t2 = L.centroid_totals + ( (size_t)prob * 3 + 1 ) * ICP_NMOM;
    const float total = (float)t2[0];
    if( total <= 1e-7 ) { L.active[prob] = 0; return; }                 // icp.h:466-470
    const float inv = __fdiv_rn( 1.0f, total );
    for( int a = 0; a < 6; ++a ) cen[a] = (float)t2[1 + a] * inv;
  }
  if( !icp_solve( res, T, e, L.exact_centroids ? cen : nullptr ) ) { L.active[prob] = 0; return; }         // icp.h:466-470: weights vanished
  for( int k = 0; k < 16; ++k ) L.T1[prob * 16 + k] = T.m[k];           // icp.h:295
  L.err[prob] = e;
  const float delta = fabsf( L.prev_err[prob] - e );
  if( !L.fixed_iters && L.iter_index > 5 && delta < 1e-5 ) L.active[prob] = 0;   // icp.h:489
}

// One workgroup per problem: fixed-order sum of the per-workgroup partials (moment k by wave k mod 16;
// lane l adds partials l, l+64, l+128, ..., then the wave tree) and — inside the ICP loop — the
// rest of the iteration (icp.h:455-493), which the reference runs on the CPU: 6x6 solve, pose
// update, stop tests.  Nothing goes back to the host between two searches.
#define UPDATE_WAVES 16
__global__ __launch_bounds__( UPDATE_WAVES * WAVE ) void k_icp_update( IcpLaunch L )
{
  const int prob = blockIdx.x;
  if( L.active[prob] == 0 ) return;
  icp_bind( L, prob );
  const int lane = threadIdx.x & ( WAVE - 1 ), wib = threadIdx.x / WAVE;
  const double* in = L.mom_part + (size_t)prob * L.n_mom_blocks * ICP_NMOM;
  double* res = L.res + (size_t)prob * ICP_NRES;
  for( int k = wib; k < ICP_NMOM; k += UPDATE_WAVES )
  {
    double v = 0.0;
    if( L.rec )
    {
      // k_chain_moments' layout (moment-major: coalesced), eight loads in flight per lane; the order of the additions is fixed
      for( int b0 = lane; b0 < L.n_mom_blocks; b0 += 8 * WAVE )
      {
        double t[8];
#pragma unroll
        for( int j = 0; j < 8; ++j ) { const int b = b0 + j * WAVE; t[j] = b < L.n_mom_blocks ? in[(size_t)k * L.n_mom_blocks + b] : 0.0; }
#pragma unroll
        for( int j = 0; j < 8; ++j ) v += t[j];
      }
    }
    else for( int b = lane; b < L.n_mom_blocks; b += WAVE ) v += in[(size_t)b * ICP_NMOM + k];
    v = wave_sum( v );
    if( lane == 0 ) res[k] = v;
  }
  __syncthreads();
  icp_update_tail( L, prob );
}

// The same for the chains' estimator, whose partials come by the thousand (one per 1 024 source points, moment-major): one
// workgroup per MOMENT sums its row (a single workgroup took 14 us over the 400 KB of a 1 M-point scan), the last one to finish
// does the rest of the iteration.  Fixed order throughout: thread t adds partials t, t + 256, ..., then the wave tree, then the
// four waves in turn.
__global__ __launch_bounds__( BLOCK ) void k_icp_update_wide( IcpLaunch L, int* done )
{
  RS_CHAIN_SETPRIO();
  __shared__ double s_part[WAVES_PER_BLOCK];
  __shared__ int s_last;
  const int prob = blockIdx.y, k = blockIdx.x;
  if( L.active[prob] == 0 ) return;
  icp_bind( L, prob );
  const int lane = threadIdx.x & ( WAVE - 1 ), wib = threadIdx.x / WAVE;
  const double* in = L.mom_part + ( (size_t)prob * ICP_NMOM + k ) * L.n_mom_blocks;
  double v = 0.0;
  for( int b = threadIdx.x; b < L.n_mom_blocks; b += BLOCK ) v += in[b];
  v = wave_sum( v );
  if( lane == 0 ) s_part[wib] = v;
  __syncthreads();
  if( threadIdx.x == 0 )
  {
    double t = 0.0;
    for( int w = 0; w < WAVES_PER_BLOCK; ++w ) t += s_part[w];
    L.res[(size_t)prob * ICP_NRES + k] = t;
    __threadfence();
    s_last = atomicAdd( done + prob, 1 ) == ICP_NMOM - 1 ? 1 : 0;
  }
  __syncthreads();
  if( !s_last ) return;
  __threadfence();                                       // (the other workgroups' sums)
  if( threadIdx.x == 0 ) done[prob] = 0;
  icp_update_tail( L, prob );
}

// ------------------------------------------------------------------------------------------
// ICP estimator in the reference's own order and precisions  (lib/rs/icp.h:136-148,210-298,387-402)
//
// The reference sums everything one correspondence after the other in source order, in fp32 (dist²
// statistics, Σw, the two centroids, the 3x3 blocks, the right-hand side) and in fp64 only Σw·s² and Σw.
// On clouds of a few thousand points that rounding is part of its result: poses drift from the exact
// least-squares step in the 5th digit and the stop test (|Δerr| < 1e-5) can fire an iteration earlier
// or later.  k_icp_moments above is the fast, more accurate step; this one reproduces the reference bit
// for bit.  fp32 addition does not associate, so each accumulator is ONE sequential chain over the
// correspondences — but the 35 accumulators are independent chains, and producing the addends is parallel:
//   k_icp_faith_gather   the correspondences in the source's original order, SoA, all threads
//   k_icp_faithful       one workgroup per problem, six waves:
//                          waves 2-5  turn 128 correspondences at a time into the addends of every accumulator (LDS; two threads per
//                                     correspondence)
//                          wave 0     lane a adds row a, entry after entry, to fp32 accumulator a
//                          wave 1     the same for the two fp64 accumulators (their rows arrive as doubles)
//                        double-buffered, so the chains never wait for the producers.
//   Two passes: dist² statistics AND weights, centroids — the statistics reach the centroids only through the 2.5 sigma cut of the
//   weights, taken at a guess and checked afterwards (a third pass, the centroids alone, when the guess cut differently: not once
//   in 349 iterations of 24 object-to-scan runs) — then the normal equations, which need the centroids; thread 0 runs the rest of
//   the iteration exactly as k_icp_update does.  What a pass costs is its longest chain: ~8.6 cycles per addition and row of
//   fp32 (6.5 with nothing else on the CU's LDS), 11.5 of fp64.
// Unmatched source points add +0 (no effect on an accumulator that started at +0).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__( BLOCK ) void k_icp_faith_gather( IcpLaunch L )
{
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  icp_bind( L, prob );
  const int i = blockIdx.x * BLOCK + threadIdx.x, n = L.src.n;
  if( i >= n ) return;
  const int s = L.by_orig ? L.by_orig[i] : i;
  const size_t o = (size_t)L.pt_off + s;
  float* F = L.faith + (size_t)FAITH_REC * (size_t)L.pt_off + i;
  const int slot = L.m_slot[o];
  if( slot < 0 ) { F[0] = -1.0f; return; }
  Xform T1;
#pragma unroll
  for( int k = 0; k < 16; ++k ) T1.m[k] = L.T1[prob * 16 + k];
  const float4 p4 = L.src.pos[s];
  float tx, ty, tz, px, py, pz;
  xform3( T1, p4.x, p4.y, p4.z, 1.0f, tx, ty, tz );
  xform3( L.T2i, tx, ty, tz, 1.0f, px, py, pz );
  const float4 q4 = L.tgt.pos[slot], n4 = L.tgt.nor[slot];
  const size_t N = (size_t)n;
  F[0] = L.w_explicit ? 0.0f : L.m_d2[o];
  F[N] = L.w_explicit ? L.w_explicit[o] : L.m_dot[o];
  F[2 * N] = px;   F[3 * N] = py;   F[4 * N] = pz;
  F[5 * N] = q4.x; F[6 * N] = q4.y; F[7 * N] = q4.z;
  F[8 * N] = n4.x; F[9 * N] = n4.y; F[10 * N] = n4.z;
}

#define FAITH_CHUNK 128
#define FAITH_PITCH ( FAITH_CHUNK + 4 )          // rows stay 16-byte aligned (128-bit LDS reads) and a quarter-wave of them covers all banks once
// Two threads per correspondence of a chunk — four producer waves, two of them beside the chain waves on their SIMDs: the normal
// equations' 35 addends (~200 vector instructions, 35 + 2 LDS writes per correspondence) in two halves.  With one thread each the two
// producer waves set the pace of the last pass (880 k cycles at work over 420 chunks against 465 k of the fp32 chain wave, 620 k
// of the fp64 one: tools/faith_timing.py); with two they take 575 k and the fp64 chain does.  RS_FAITH_SPLIT=0: one thread each.
#ifndef RS_FAITH_SPLIT
#define RS_FAITH_SPLIT 1
#endif
#define FAITH_PRODUCERS ( ( RS_FAITH_SPLIT + 1 ) * FAITH_CHUNK )
#define FAITH_THREADS ( 2 * WAVE + FAITH_PRODUCERS )   // two chain waves + the producer waves

struct FaithRec { float v[FAITH_REC]; };
struct FaithPar
{
  bool  w_explicit, use_sd;
  float max_dist, cut;
  float c1[3], c2[3];
};

__device__ __forceinline__ void faith_load( const float* F, int n, int i, FaithRec& r )
{
  r.v[0] = -1.0f;
  if( i < n )
  {
#pragma unroll
    for( int k = 0; k < FAITH_REC; ++k ) r.v[k] = F[(size_t)k * n + i];
  }
}

__device__ __forceinline__ float faith_weight( const FaithRec& r, const FaithPar& P )
{
  if( P.w_explicit ) return r.v[1];
  float w = ( 1.0f - __fdiv_rn( r.v[0], P.max_dist ) ) * r.v[1];      // icp.h:387
  if( P.use_sd && r.v[0] > P.cut ) w = 0.0f;                           // icp.h:396-401
  return w;
}

// the addends of one correspondence for pass PASS, written to column t of `term` (RS_FAITH_SPLIT: of PASS 3 rows [0,18) for half 0,
// [18,35) for half 1; half 2: all)
template <int PASS>
__device__ __forceinline__ void faith_terms( const FaithRec& r, const FaithPar& P, float ( *term )[FAITH_PITCH], int t, int half = 2,
                                             double ( *termd )[FAITH_PITCH] = nullptr /* PASS 3: rows 33, 34 also as doubles */ )
{
  if( PASS != 3 && half == 1 ) return;
  const bool m = r.v[0] >= 0.0f;
  if( PASS == 1 || PASS == 12 )
  {
    constexpr int R0 = PASS == 12 ? 7 : 0;
    term[R0 + 0][t] = m ? r.v[0] : 0.0f;            // msh_compute_mean
    term[R0 + 1][t] = m ? r.v[0] * r.v[0] : 0.0f;   // msh_compute_stddev
    term[R0 + 2][t] = m ? 1.0f : 0.0f;              // n_corrs (exact in fp32 below 2^24)
  }
  if( PASS == 1 ) return;
  if( PASS == 2 || PASS == 12 )
  {
    const float w = m ? faith_weight( r, P ) : 0.0f;
    term[0][t] = w;                                 // icp.h:141  total += w
#pragma unroll
    for( int a = 0; a < 3; ++a )
    {
      term[1 + a][t] = m ? r.v[2 + a] * w : 0.0f;   // icp.h:142  c = c + p*w
      term[4 + a][t] = m ? r.v[5 + a] * w : 0.0f;
    }
  }
  else
  {
    if( !m )
    {
      if( half != 1 ) {
#pragma unroll
        for( int a = 0; a < 18; ++a ) term[a][t] = 0.0f;
      }
      if( half != 0 ) {
#pragma unroll
        for( int a = 18; a < ICP_NMOM; ++a ) term[a][t] = 0.0f;
        if( termd ) { termd[0][t] = 0.0; termd[1][t] = 0.0; }
      }
      return;
    }
    const float wi = faith_weight( r, P );
    const float p[3] = { r.v[2] - P.c1[0], r.v[3] - P.c1[1], r.v[4] - P.c1[2] };
    const float q[3] = { r.v[5] - P.c2[0], r.v[6] - P.c2[1], r.v[7] - P.c2[2] };
    const float nv[3] = { r.v[8], r.v[9], r.v[10] };
    const float d[3] = { p[0] - q[0], p[1] - q[1], p[2] - q[2] };
    const float cv[3] = { p[1] * nv[2] - p[2] * nv[1], p[2] * nv[0] - p[0] * nv[2], p[0] * nv[1] - p[1] * nv[0] };
    const float sd = d[0] * nv[0] + d[1] * nv[1] + d[2] * nv[2];
    if( half != 1 )
    {
#pragma unroll
      for( int col = 0; col < 3; ++col )
#pragma unroll
        for( int row = 0; row < 3; ++row )
        {
          term[3 * col + row][t]      = ( cv[row] * cv[col] ) * wi;   // icp.h:239-241, column-major blocks
          term[9 + 3 * col + row][t]  = ( cv[row] * nv[col] ) * wi;
        }
    }
    if( half == 0 ) return;
#pragma unroll
    for( int col = 0; col < 3; ++col )
#pragma unroll
      for( int row = 0; row < 3; ++row ) term[18 + 3 * col + row][t] = ( nv[row] * nv[col] ) * wi;
#pragma unroll
    for( int a = 0; a < 3; ++a )
    {
      term[27 + a][t] = wi * cv[a] * sd;            // icp.h:242-247
      term[30 + a][t] = wi * nv[a] * sd;
    }
    term[33][t] = wi * sd * sd;                     // icp.h:249 (a float product, summed in fp64)
    term[34][t] = wi;                               // icp.h:250
    if( termd ) { termd[0][t] = (double)( wi * sd * sd ); termd[1][t] = (double)wi; }
  }
}

// A chain wave's chunk: its row's 128 entries added one after the other, straight from LDS, sixteen at a time — the reads of the next
// sixteen go out before this group's additions.  Columns past the end of the cloud hold +0, so every chunk is a full one.
// (A whole row staged in registers first — two of them, 256 registers — pushed the kernel's allocation into AGPRs, one
// v_accvgpr_read per addend; reads placed between the additions one by one took 50 % longer.  None of it shows: a chunk takes what
// its 128 dependent additions take, ~8 cycles each.)
// (the two fp64 accumulators' rows arrive as doubles — the producers' conversion: with a v_cvt_f64_f32 in front of every addition the
//  fp64 wave took 1.5 times as long as the fp32 wave, and set the pace of the last pass together with the producers)
__device__ __forceinline__ void faith_chain_lds( const double* row, double& acc )
{
  const double2* row2 = reinterpret_cast<const double2*>( row );
  double2 a[4], b[4];
#pragma unroll
  for( int j = 0; j < 4; ++j ) a[j] = row2[j];
#pragma unroll
  for( int g = 0; g < FAITH_CHUNK / 8; ++g )
  {
    double2* cur = ( g & 1 ) ? b : a; double2* nxt = ( g & 1 ) ? a : b;
    if( g + 1 < FAITH_CHUNK / 8 )
    {
#pragma unroll
      for( int j = 0; j < 4; ++j ) nxt[j] = row2[( g + 1 ) * 4 + j];
    }
#pragma unroll
    for( int j = 0; j < 4; ++j ) { acc += cur[j].x; acc += cur[j].y; }
    __builtin_amdgcn_sched_barrier( 0 );
  }
}
template <class ACC>
__device__ __forceinline__ void faith_chain_lds( const float* row, ACC& acc )
{
  const float4* row4 = reinterpret_cast<const float4*>( row );
  float4 a[4], b[4];
#pragma unroll
  for( int j = 0; j < 4; ++j ) a[j] = row4[j];
#pragma unroll
  for( int g = 0; g < FAITH_CHUNK / 16; ++g )
  {
    float4* cur = ( g & 1 ) ? b : a; float4* nxt = ( g & 1 ) ? a : b;
    if( g + 1 < FAITH_CHUNK / 16 )
    {
#pragma unroll
      for( int j = 0; j < 4; ++j ) nxt[j] = row4[( g + 1 ) * 4 + j];
    }
#pragma unroll
    for( int j = 0; j < 4; ++j ) { acc += (ACC)cur[j].x; acc += (ACC)cur[j].y; acc += (ACC)cur[j].z; acc += (ACC)cur[j].w; }
    __builtin_amdgcn_sched_barrier( 0 );          // (the scheduler would gather all the row's reads in front of the chain)
  }
}

// One pass over the correspondences: rows [0,NF) end in accf of wave 0's lanes, rows [NF,NF+ND) in accd of wave 1's.
// Iteration k: the producers write chunk k (and keep FAITH_AHEAD chunks of loads in flight: a chunk is consumed faster
// than a load returns); the chain waves add up chunk k-1 from LDS.
#define FAITH_AHEAD 4
// PASS 12: statistics and centroids in ONE pass, the weights cut at a guess of 2.5 sigma (P.cut); band[0] / band[1] end as the largest
// dist² not above the guess and the smallest above it (bits: dist² >= 0) — the guess cut the weights exactly as the real value
// does iff the real value lies in [band[0], band[1]).
template <int PASS, int NF, int ND>
__device__ __forceinline__ void faith_pass( const float* F, int n, const FaithPar& P, float ( *term )[ICP_NMOM][FAITH_PITCH], double ( *termd )[2][FAITH_PITCH],
                                            float& accf, double& accd, unsigned* band = nullptr, int* timing = nullptr /* (experiment: -DRS_FAITH_TIMING=<pass>) */ )
{
  const int wib = threadIdx.x / WAVE, lane = threadIdx.x & ( WAVE - 1 );
  const int t = ( threadIdx.x - 2 * WAVE ) & ( FAITH_CHUNK - 1 );              // producer column ...
  const int half = RS_FAITH_SPLIT ? uni( (int)( threadIdx.x - 2 * WAVE ) / FAITH_CHUNK ) : 2;      // ... and which of its addends (RS_FAITH_SPLIT: half of them; wave-uniform)
  const int n_chunks = ( n + FAITH_CHUNK - 1 ) / FAITH_CHUNK;
  const int my_row = wib == 0 ? ( lane < NF ? lane : -1 ) : ( wib == 1 && lane < ND ? NF + lane : -1 );
  accf = 0.0f; accd = 0.0;
  FaithRec ring[FAITH_AHEAD];
  float d_below = 0.0f, d_above = INFINITY;
  const bool producer = wib >= 2;
#ifdef RS_FAITH_TIMING
  long long t_work = 0, t_wait = 0;
#endif
  if( producer )
  {
#pragma unroll
    for( int u = 0; u < FAITH_AHEAD; ++u ) faith_load( F, n, u * FAITH_CHUNK + t, ring[u] );
  }
  for( int k0 = 0; k0 <= n_chunks; k0 += FAITH_AHEAD )
  {
#pragma unroll
    for( int u = 0; u < FAITH_AHEAD; ++u )
    {
      const int k = k0 + u;
#ifdef RS_FAITH_TIMING
      const long long c0 = clock64();
#endif
      if( wib >= 2 )
      {
        if( producer && k < n_chunks )
        {
          if( PASS == 12 && half != 1 && ring[u].v[0] >= 0.0f )
          {
            if( ring[u].v[0] > P.cut ) d_above = fminf( d_above, ring[u].v[0] ); else d_below = fmaxf( d_below, ring[u].v[0] );
          }
          faith_terms<PASS>( ring[u], P, term[k & 1], t, half, PASS == 3 ? termd[k & 1] : nullptr );
          faith_load( F, n, ( k + FAITH_AHEAD ) * FAITH_CHUNK + t, ring[u] );
        }
      }
      else if( my_row >= 0 )
      {
        // chunk k - 1 (written during step k - 1, the barrier since), while the producers write chunk k into the other buffer
        if( k >= 1 && k <= n_chunks )
        {
          if( wib == 0 ) faith_chain_lds( term[( k - 1 ) & 1][my_row], accf ); else faith_chain_lds( termd[( k - 1 ) & 1][my_row - NF], accd );
        }
      }
#ifdef RS_FAITH_TIMING
      const long long c1 = clock64();
#endif
      __syncthreads();
#ifdef RS_FAITH_TIMING
      t_work += c1 - c0; t_wait += clock64() - c1;
#endif
    }
  }
#ifdef RS_FAITH_TIMING
  if( lane == 0 && PASS == RS_FAITH_TIMING && timing && wib < 4 ) { int* o = timing + 2 + 3 * wib; o[0] = (int)t_work; o[1] = (int)t_wait; o[2] = n_chunks; }
#endif
  if( PASS == 12 && producer ) { atomicMax( band, __float_as_uint( d_below ) ); atomicMin( band + 1, __float_as_uint( d_above ) ); }
}

__global__ __launch_bounds__( FAITH_THREADS ) void k_icp_faithful( IcpLaunch L )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) float term[2][ICP_NMOM][FAITH_PITCH];
  __shared__ __attribute__( ( aligned( 16 ) ) ) double termd[2][2][FAITH_PITCH];
  __shared__ float s_f[ICP_NMOM];
  __shared__ double s_d[2], s_g[FAITH_THREADS / WAVE][3];
  __shared__ unsigned s_band[2];
  const int prob = blockIdx.x;
  if( L.active[prob] == 0 ) return;
  icp_bind( L, prob );
  if( L.solve ) icp_iteration_reset( L, prob );            // (the search of this iteration is over: its queue has been consumed)
  const int wib = threadIdx.x / WAVE, lane = threadIdx.x & ( WAVE - 1 );
  const int n = L.src.n;
  const float* F = L.faith + (size_t)FAITH_REC * (size_t)L.pt_off;
  FaithPar P;
  P.w_explicit = L.w_explicit != nullptr; P.use_sd = false; P.max_dist = L.radius; P.cut = 0.0f;
  P.c1[0] = P.c1[1] = P.c1[2] = P.c2[0] = P.c2[1] = P.c2[2] = 0.0f;
  float accf; double accd;

  // ---- icp.h:393-402: mean and standard deviation of dist² over the correspondences; icp.h:136-148: Σw and the two weighted centroids ----
  // The statistics only reach the centroids through the 2.5 sigma cut of the weights (icp.h:396-401), a COMPARISON: with a guess of
  // sigma (fp64 sums, all threads, a few microseconds) both sets of chains run in one pass — ten rows instead of three, then
  // seven: a chain wave's time does not depend on how many of its lanes are rows — and the pass stands if no dist² lies between
  // the guessed and the real cut (and they agree on whether to cut at all).  Otherwise the centroids again, as before.
  bool have_centroids = false;
  if( !P.w_explicit )
  {
    {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      for( int i = threadIdx.x; i < n; i += FAITH_THREADS ) { const float d = F[i]; if( d >= 0.0f ) { a0 += (double)d; a1 += (double)d * (double)d; a2 += 1.0; } }
      a0 = wave_sum( a0 ); a1 = wave_sum( a1 ); a2 = wave_sum( a2 );
      if( lane == 0 ) { s_g[wib][0] = a0; s_g[wib][1] = a1; s_g[wib][2] = a2; }
      if( threadIdx.x == 0 ) { s_band[0] = 0u; s_band[1] = __float_as_uint( INFINITY ); }
      __syncthreads();
      double t0 = 0.0, t1 = 0.0, t2 = 0.0;
      for( int w = 0; w < FAITH_THREADS / WAVE; ++w ) { t0 += s_g[w][0]; t1 += s_g[w][1]; t2 += s_g[w][2]; }
      const double mean_g = t2 > 0.0 ? t0 / t2 : 0.0, var_g = t2 > 0.0 ? t1 / t2 - mean_g * mean_g : 0.0;
      const float sd_g = (float)sqrt( var_g > 0.0 ? var_g : 0.0 );
      P.use_sd = sd_g > 0.000001; P.cut = 2.5f * sd_g * L.faith_guess_scale;
    }
    const bool guessed_use = P.use_sd;
    if( L.faith_guess_scale != 0.0f ) faith_pass<12, 10, 0>( F, n, P, term, termd, accf, accd, s_band, L.faith_redone );
    else                              faith_pass<1, 3, 0>( F, n, P, term, termd, accf, accd );
    const int r0 = L.faith_guess_scale != 0.0f ? 7 : 0;
    if( wib == 0 && lane < 10 ) s_f[lane] = accf;
    __syncthreads();
    const float cnt = s_f[r0 + 2];
    if( cnt == 0.0f )                                                   // icp.h:455-459: no correspondences
    {
      if( threadIdx.x == 0 && L.solve ) { L.prev_err[prob] = L.err[prob]; L.iters[prob] += 1; L.active[prob] = 0; }
      return;
    }
    const float mean = __fdiv_rn( s_f[r0], cnt );                       // msh_std.h:1800-1825
    const float var = __fdiv_rn( s_f[r0 + 1], cnt ) - mean * mean;
    const float sd = (float)sqrt( (double)var );
    P.use_sd = sd > 0.000001;
    P.cut = 2.5f * sd;
    have_centroids = L.faith_guess_scale != 0.0f && P.use_sd == guessed_use && ( !P.use_sd || ( __uint_as_float( s_band[0] ) <= P.cut && P.cut < __uint_as_float( s_band[1] ) ) );
    if( !have_centroids && L.faith_guess_scale != 0.0f && threadIdx.x == 0 && L.faith_redone ) atomicAdd( L.faith_redone, 1 );
    __syncthreads();                                                    // (s_f is rewritten below if the pass does not stand)
  }

  if( !have_centroids )
  {
    faith_pass<2, 7, 0>( F, n, P, term, termd, accf, accd );
    if( wib == 0 && lane < 7 ) s_f[lane] = accf;
    __syncthreads();
  }
  const float total = s_f[0];
  if( total <= 1e-7 )                                                   // icp.h:466-470: the weights vanished
  {
    if( threadIdx.x == 0 && L.solve ) { L.prev_err[prob] = L.err[prob]; L.iters[prob] += 1; L.active[prob] = 0; }
    return;
  }
  const float inv = __fdiv_rn( 1.0f, total );
#pragma unroll
  for( int a = 0; a < 3; ++a ) { P.c1[a] = s_f[1 + a] * inv; P.c2[a] = s_f[4 + a] * inv; }
  __syncthreads();

  // ---- icp.h:221-252: the normal equations ----
  faith_pass<3, 33, 2>( F, n, P, term, termd, accf, accd, nullptr, L.faith_redone );
  if( wib == 0 && lane < 33 ) s_f[lane] = accf;
  if( wib == 1 && lane < 2 ) s_d[lane] = accd;
  __syncthreads();
  if( threadIdx.x != 0 ) return;

  // ---- icp.h:253-295 and the loop's bookkeeping (icp.h:455-493), as in k_icp_update ----
  float A[33];
  for( int k = 0; k < 33; ++k ) A[k] = s_f[k];
  Mat4 T;
  for( int k = 0; k < 16; ++k ) { T.m[k] = L.T1[prob * 16 + k]; if( L.solve ) L.T1_prev[prob * 16 + k] = T.m[k]; }
  float e;
  icp_solve_ref_order( A, s_d[0], s_d[1], P.c1, T, e );
  for( int k = 0; k < 16; ++k ) L.T1[prob * 16 + k] = T.m[k];
  if( !L.solve ) { L.err[prob] = e; return; }
  L.prev_err[prob] = L.err[prob];
  L.iters[prob] += 1;
  L.err[prob] = e;
  const float delta = fabsf( L.prev_err[prob] - e );
  if( !L.fixed_iters && L.iter_index > 5 && delta < 1e-5 ) L.active[prob] = 0;   // icp.h:489
}

// ------------------------------------------------------------------------------------------
// The reference-order estimator, in parallel ("replay")
//
// k_icp_faithful above runs each of the reference's accumulators as ONE sequential chain: 10-12 ns per source point and
// iteration, 1.5 ms per iteration on a 134 k-point scan.  The same bits can be had in parallel, because of what an IEEE
// addition S + x does while S stays inside one binade [2^e, 2^(e+1)): it adds x ROUNDED TO THE BINADE'S GRID (ulp u), and that
// rounding does not depend on S — except for an exact tie (x mod u = u/2), which goes to the even neighbour, i.e. depends on
// the parity of S's mantissa.  So over a stretch of addends during which the accumulator stays inside its binade, the
// sequential sum is   S_out = S_in + D(parity of S_in),   with D a constant of the stretch.
//
//   k_replay_sums   cuts the source (original order) into segments of 128 points and sums every accumulator's addends per
//                   segment in fp64;
//   k_replay_scan   prefix-sums those per accumulator: a GUESS of the accumulator's value at every segment start (good to a
//                   few thousand ulps: the real chain's own rounding is what it misses);
//   k_replay_run    runs, for every segment and accumulator in parallel, the real fp32 (fp64) chain over the segment from the
//                   guess — once per CLASS of the start's mantissa modulo 4 — and records for which exact starts of that class
//                   the chain is the exact chain shifted: the shift delta = (exact start - class start), in units of the
//                   start's ulp, must keep every intermediate value strictly inside the binade the class's chain visits at
//                   that step (an interval for delta, intersected over the 128 steps — the chain may cross binades), and must
//                   be an EVEN number of grid steps in every binade visited (so that ties round the same way: delta a multiple
//                   of 2^(k+1) where the grid is 2^k coarser than the start's);
//   k_replay_walk   one wave per accumulator walks the segments in order with the EXACT value: if its sign / exponent are the
//                   guess's and delta passes the class's tests, the segment's result is the class's end value shifted by delta
//                   (in the end binade's grid) — exact, by the argument above; otherwise (a sign change inside the segment,
//                   the start of a chain, a guess in the wrong binade) the wave re-adds the segment's 128 addends one after
//                   the other.
//
// Three passes like k_icp_faithful (statistics -> weights and centroids -> normal equations), each needing the previous one's
// totals; every block recomputes the few scalars between passes itself.  Addends come from the same faith_terms<PASS> the
// sequential kernel uses.  Result: the reference's bits (tests: against k_icp_faithful and the reference-generated fixtures).
// ------------------------------------------------------------------------------------------
#define REPLAY_SEG 128
#define REPLAY_PITCH ( REPLAY_SEG + 4 )

template <int PASS> struct ReplayRows;
template <> struct ReplayRows<1> { enum { NF = 3, ND = 0 }; };     // Σd², Σd⁴, count
template <> struct ReplayRows<2> { enum { NF = 7, ND = 0 }; };     // Σw, Σw·p (3), Σw·q (3)
template <> struct ReplayRows<3> { enum { NF = 33, ND = 2 }; };    // 3x3 blocks, rhs | Σw·s², Σw (fp64 in the reference)

#define REPLAY_CLS 4
// What k_replay_run records per (accumulator, segment) and class c = (start mantissa mod 4) — start = the guess's bits with
// the two low mantissa bits cleared, class start = start | c.  With delta = m - (class start's mantissa): the record is
// usable iff sign + exponent match, dmin <= delta <= dmax (dmax < dmin: never) and delta is a multiple of 2 << need_k; the
// result is then `end` with its mantissa advanced by delta >> k_end (k_end >= 0) or delta << -k_end.
struct ReplayCls { long long dmin, dmax; unsigned long long end; int need_k, k_end; };
struct ReplaySeg { unsigned long long start; unsigned long long pad; ReplayCls cls[REPLAY_CLS]; };

// the scalars between the passes, from the totals of the finished passes (identical code to k_icp_faithful's)
__device__ __forceinline__ bool replay_params( const IcpLaunch& L, int prob, int pass, const double* totals /* ICP_NMOM per pass */, FaithPar& P )
{
  P.w_explicit = L.w_explicit != nullptr; P.use_sd = false; P.max_dist = L.radius; P.cut = 0.0f;
  P.c1[0] = P.c1[1] = P.c1[2] = P.c2[0] = P.c2[1] = P.c2[2] = 0.0f;
  if( pass >= 2 && !P.w_explicit && L.exact_centroids )
  {
    // the cut of k_icp_moments (same expressions, same bits: both take n, mean, stddev from the integer statistics of the search)
    const double* st = L.res + (size_t)prob * ICP_NRES + ICP_NMOM;
    if( st[0] == 0.0 ) return false;
    const float sd = (float)st[2];
    P.use_sd = sd > 0.000001;
    P.cut = 2.5f * sd;
  }
  else if( pass >= 2 && !P.w_explicit )
  {
    const double* t1 = totals;                                         // pass 1: Σd², Σd⁴, count (floats kept in doubles)
    const float cnt = (float)t1[2];
    if( cnt == 0.0f ) return false;
    const float mean = __fdiv_rn( (float)t1[0], cnt );                 // msh_std.h:1800-1825
    const float var = __fdiv_rn( (float)t1[1], cnt ) - mean * mean;
    const float sd = (float)sqrt( (double)var );
    P.use_sd = sd > 0.000001;
    P.cut = 2.5f * sd;
  }
  if( pass >= 3 )
  {
    const double* t2 = totals + ICP_NMOM;                              // pass 2: Σw, Σw·p, Σw·q
    const float total = (float)t2[0];
    if( total <= 1e-7 ) return false;
    const float inv = __fdiv_rn( 1.0f, total );
#pragma unroll
    for( int a = 0; a < 3; ++a ) { P.c1[a] = (float)t2[1 + a] * inv; P.c2[a] = (float)t2[4 + a] * inv; }
  }
  return true;
}

// One correspondence as the estimators' kernels take it: from k_icp_faith_gather's arrays — or, where the searches left their 48-byte
// records at the points' ORIGINAL indices (whole scans: the centroid sums by pass 2 here, when the grid chains give a scan up), straight
// from those: the same eleven numbers, without the gather launch (117 us at a million points).
template <int PASS>
__device__ __forceinline__ void replay_load( const IcpLaunch& L, int prob, int i, FaithRec& r )
{
  const int n = L.src.n;
#pragma unroll
  for( int k = 0; k < FAITH_REC; ++k ) r.v[k] = 0.0f;
  if( !L.rec ) { faith_load( L.faith + (size_t)prob * FAITH_REC * n, n, i, r ); return; }
  r.v[0] = -1.0f;
  if( i < n )
  {
    const float4* R = L.rec + ( (size_t)prob * n + i ) * REC_F4;
    const float4 a = R[0], b = R[1];
    r.v[0] = a.w; r.v[1] = b.w; r.v[2] = a.x; r.v[3] = a.y; r.v[4] = a.z; r.v[5] = b.x; r.v[6] = b.y; r.v[7] = b.z;
    if( PASS == 3 ) { const float4 c = R[2]; r.v[8] = c.x; r.v[9] = c.y; r.v[10] = c.z; }      // (the target's normal: the normal equations' alone)
  }
}
// the addends of segment g for pass PASS, into term[row][t] (all rows of the pass, 128 columns; columns past the cloud hold +0)
template <int PASS>
__device__ __forceinline__ void replay_terms( const IcpLaunch& L, int prob, int g, const FaithPar& P, float ( *term )[REPLAY_PITCH] )
{
  for( int t = threadIdx.x; t < REPLAY_SEG; t += blockDim.x )
  {
    FaithRec r; replay_load<PASS>( L, prob, g * REPLAY_SEG + t, r );
    faith_terms<PASS>( r, P, reinterpret_cast<float ( * )[FAITH_PITCH]>( term ), t );
  }
}
static_assert( REPLAY_PITCH == FAITH_PITCH, "replay_terms reuses faith_terms' LDS layout" );

// ONE accumulator's addend of one correspondence — the expressions of faith_terms<PASS>, row by row (the walk re-adds a segment of ONE
// row: producing all 35 rows' addends for it, as the kernels above do, was most of what a re-added segment cost)
template <int PASS>
__device__ __forceinline__ float faith_term_one( const FaithRec& r, const FaithPar& P, int row /* uniform */ )
{
  const bool m = r.v[0] >= 0.0f;
  if( PASS == 1 ) return row == 0 ? ( m ? r.v[0] : 0.0f ) : ( row == 1 ? ( m ? r.v[0] * r.v[0] : 0.0f ) : ( m ? 1.0f : 0.0f ) );
  if( PASS == 2 )
  {
    const float w = m ? faith_weight( r, P ) : 0.0f;
    if( row == 0 ) return w;
    const float v = row == 1 ? r.v[2] : row == 2 ? r.v[3] : row == 3 ? r.v[4] : row == 4 ? r.v[5] : row == 5 ? r.v[6] : r.v[7];
    return m ? v * w : 0.0f;
  }
  if( !m ) return 0.0f;
  const float wi = faith_weight( r, P );
  if( row == 34 ) return wi;
  const float p[3] = { r.v[2] - P.c1[0], r.v[3] - P.c1[1], r.v[4] - P.c1[2] };
  const float q[3] = { r.v[5] - P.c2[0], r.v[6] - P.c2[1], r.v[7] - P.c2[2] };
  const float nv[3] = { r.v[8], r.v[9], r.v[10] };
  const float d[3] = { p[0] - q[0], p[1] - q[1], p[2] - q[2] };
  const float cv[3] = { p[1] * nv[2] - p[2] * nv[1], p[2] * nv[0] - p[0] * nv[2], p[0] * nv[1] - p[1] * nv[0] };
  const float sd = d[0] * nv[0] + d[1] * nv[1] + d[2] * nv[2];
  auto pick = []( const float ( &x )[3], int k ) -> float { return k == 0 ? x[0] : ( k == 1 ? x[1] : x[2] ); };
  if( row < 27 )
  {
    const int blk = row / 9, in = row % 9, col = in / 3, rw = in % 3;
    const float a = blk == 2 ? pick( nv, rw ) : pick( cv, rw );              // cv cv | cv nv | nv nv
    const float b = blk == 0 ? pick( cv, col ) : pick( nv, col );
    return ( a * b ) * wi;                                                 // icp.h:239-241
  }
  if( row < 30 ) return wi * pick( cv, row - 27 ) * sd;                     // icp.h:242-247
  if( row < 33 ) return wi * pick( nv, row - 30 ) * sd;
  return wi * sd * sd;                                                     // icp.h:249
}
template <int PASS>
__device__ __forceinline__ void replay_term_row( const IcpLaunch& L, int prob, int g, const FaithPar& P, float ( *term )[REPLAY_PITCH], int row )
{
  for( int t = threadIdx.x; t < REPLAY_SEG; t += blockDim.x )
  {
    FaithRec r; replay_load<PASS>( L, prob, g * REPLAY_SEG + t, r );
    term[row][t] = faith_term_one<PASS>( r, P, row );
  }
}

template <int PASS>
__global__ __launch_bounds__( REPLAY_SEG ) void k_replay_sums( IcpLaunch L, ReplayBufs B )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) float term[ICP_NMOM][REPLAY_PITCH];
  const int prob = blockIdx.y, g = blockIdx.x;
  if( L.active[prob] == 0 ) return;
  FaithPar P;
  const double* totals = B.totals + (size_t)prob * 3 * ICP_NMOM;
  if( !replay_params( L, prob, PASS, totals, P ) ) return;
  replay_terms<PASS>( L, prob, g, P, term );
  __syncthreads();
  constexpr int NR = ReplayRows<PASS>::NF + ReplayRows<PASS>::ND;
  if( threadIdx.x < NR )
  {
    double a = 0.0;
    for( int t = 0; t < REPLAY_SEG; ++t ) a += (double)term[threadIdx.x][t];
    B.segsum[( (size_t)prob * ICP_NMOM + threadIdx.x ) * B.n_seg + g] = a;
  }
}

// exclusive prefix over the segments, per accumulator: the guesses (as the accumulator's own type: float rows, double rows).  One
// workgroup per accumulator: every thread a contiguous run of segments, the runs' totals scanned through LDS (any association will
// do: this is a guess).  (One wave looping over the segments 64 at a time took 76 us on a 1.15 M-point scan's 9 007 segments.)
#define REPLAY_SCAN_THREADS 256
template <int PASS>
__global__ __launch_bounds__( REPLAY_SCAN_THREADS ) void k_replay_scan( IcpLaunch L, ReplayBufs B )
{
  __shared__ double part[REPLAY_SCAN_THREADS];
  const int prob = blockIdx.y, row = blockIdx.x;
  if( L.active[prob] == 0 ) return;
  const int t = threadIdx.x;
  const double* in = B.segsum + ( (size_t)prob * ICP_NMOM + row ) * B.n_seg;
  double* out = B.guess + ( ( (size_t)prob * 3 + ( PASS - 1 ) ) * ICP_NMOM + row ) * B.n_seg;
  const int per = ( B.n_seg + REPLAY_SCAN_THREADS - 1 ) / REPLAY_SCAN_THREADS;
  const int g0 = min( t * per, B.n_seg ), g1 = min( g0 + per, B.n_seg );
  double sum = 0.0;
  for( int g = g0; g < g1; ++g ) sum += in[g];
  part[t] = sum;
  __syncthreads();
  for( int d = 1; d < REPLAY_SCAN_THREADS; d <<= 1 )       // inclusive scan of the runs' totals
  {
    const double up = t >= d ? part[t - d] : 0.0;
    __syncthreads();
    part[t] += up;
    __syncthreads();
  }
  double carry = part[t] - sum;                            // what the runs before this one add
  for( int g = g0; g < g1; ++g ) { out[g] = carry; carry += in[g]; }
}

// one accumulator type: the bit-level view of fp32 / fp64 the replay needs
template <class T> struct Bits;
template <> struct Bits<float>
{
  typedef uint32_t U; enum { MBITS = 23 };
  static __device__ __forceinline__ U of( float v ) { return __float_as_uint( v ); }
  static __device__ __forceinline__ float from( U b ) { return __uint_as_float( b ); }
};
template <> struct Bits<double>
{
  typedef unsigned long long U; enum { MBITS = 52 };
  static __device__ __forceinline__ U of( double v ) { return (U)__double_as_longlong( v ); }
  static __device__ __forceinline__ double from( U b ) { return __longlong_as_double( (long long)b ); }
};

// the chain of one (accumulator, segment, class)
// (fp32 rows keep their shift bounds in 32-bit integers: a value that RISES more than 7 binades inside one segment makes the record
//  unusable — the walk re-adds that segment — where the 64-bit form allowed 20; the loop is bound by instruction issue, and 64-bit
//  compares, selects and shifts were most of its ~40 instructions per addend)
template <class T> struct ReplayRunInt;
template <> struct ReplayRunInt<float>  { typedef int W; enum { KUP = 7 }; };
template <> struct ReplayRunInt<double> { typedef long long W; enum { KUP = 8 }; };
template <class T>
__device__ __forceinline__ void replay_run_chain( const float* row, double guess, int c, ReplaySeg& out )
{
  typedef typename Bits<T>::U U;
  typedef typename ReplayRunInt<T>::W W;
  constexpr int MB = Bits<T>::MBITS, EB = 8 * sizeof(T) - 1 - MB;
  constexpr int KUP = ReplayRunInt<T>::KUP, KDN = MB == 23 ? 20 : 8;      // (shifted mantissas must fit W)
  const U mmask = ( (U)1 << MB ) - 1, emax = ( (U)1 << EB ) - 1;
  const T gT = (T)guess;
  const U gb = Bits<T>::of( gT ) & ~(U)( REPLAY_CLS - 1 );
  const int e0 = (int)( ( gb >> MB ) & emax );
  const bool usable = e0 != 0 && e0 != (int)emax;               // normal, finite, non-zero
  if( c == 0 ) { out.start = (unsigned long long)gb; out.pad = 0ull; }
  const U sb = gb | (U)c;
  T acc = Bits<T>::from( sb );
  const W m0 = (W)( sb & mmask );
  W dmin = 1 - m0, dmax = (W)mmask - 1 - m0;
  int need_k = 0, k = 0;
  bool valid = usable;
  for( int t = 0; t < REPLAY_SEG; ++t )
  {
    acc += (T)row[t];
    const U b = Bits<T>::of( acc );
    const int e = (int)( ( b >> MB ) & emax );
    k = e - e0;
    valid = valid && !( ( b ^ sb ) >> ( 8 * sizeof(T) - 1 ) ) && e != 0 && e != (int)emax && k <= KUP && k >= -KDN;
    const int kk = valid ? k : 0;                              // (keeps the shifts below defined once the chain is lost)
    const W M = (W)( b & mmask );
    W lo = 1 - M, hi = (W)mmask - 1 - M;                       // allowed shift of this value, in its own binade's grid steps
    if( kk >= 0 ) { lo *= ( (W)1 << kk ); hi *= ( (W)1 << kk ); need_k = kk > need_k ? kk : need_k; }
    else
    {
      const int sh = -kk; const W rnd = ( (W)1 << sh ) - 1;
      lo = lo >= 0 ? ( ( lo + rnd ) >> sh ) : -( ( -lo ) >> sh );          // ceil( lo / 2^sh )
      hi = hi >= 0 ? ( hi >> sh ) : -( ( -hi + rnd ) >> sh );              // floor( hi / 2^sh )
    }
    dmin = lo > dmin ? lo : dmin; dmax = hi < dmax ? hi : dmax;
  }
  ReplayCls r;
  r.dmin = valid ? (long long)dmin : 1; r.dmax = valid ? (long long)dmax : 0;
  r.end = (unsigned long long)Bits<T>::of( acc );
  r.need_k = need_k; r.k_end = valid ? k : 0;
  out.cls[c] = r;
}

#define REPLAY_RUN_THREADS 192        // >= 35 rows x 4 classes and >= REPLAY_SEG term producers
template <int PASS>
__global__ __launch_bounds__( REPLAY_RUN_THREADS ) void k_replay_run( IcpLaunch L, ReplayBufs B )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) float term[ICP_NMOM][REPLAY_PITCH];
  const int prob = blockIdx.y, g = blockIdx.x;
  if( L.active[prob] == 0 ) return;
  FaithPar P;
  const double* totals = B.totals + (size_t)prob * 3 * ICP_NMOM;
  if( !replay_params( L, prob, PASS, totals, P ) ) return;
  replay_terms<PASS>( L, prob, g, P, term );
  __syncthreads();
  constexpr int NF = ReplayRows<PASS>::NF, NR = NF + ReplayRows<PASS>::ND;
  static_assert( NR * REPLAY_CLS <= REPLAY_RUN_THREADS, "one thread per (row, class)" );
  const int row = threadIdx.x / REPLAY_CLS, c = threadIdx.x % REPLAY_CLS;
  if( row < NR )
  {
    const size_t o = ( (size_t)prob * ICP_NMOM + row ) * B.n_seg + g;
    const double guess = B.guess[( ( (size_t)prob * 3 + ( PASS - 1 ) ) * ICP_NMOM + row ) * B.n_seg + g];
    if( row < NF ) replay_run_chain<float>( term[row], guess, c, B.seg[o] );
    else           replay_run_chain<double>( term[row], guess, c, B.seg[o] );
  }
}

// 64 consecutive segments composed into ONE record of the same form, per class of the first segment's start: the walk can then
// take 8 192 addends in a step.  With delta the (class-aligned) offset of the exact value at the superblock's start, the exact
// value entering segment s is  base_s + delta * 2^-K_s  (base_s: where the guess chains lead when delta = 0; K_s: how much
// coarser the grid has become); segment s's own tests on its delta_s = (base_s - its class start) + delta * 2^-K_s turn into an
// interval and a divisibility condition on delta, and its result into the next base.  Anything that does not fit (a base in
// another binade than the segment's guess, a constant part that fails the segment's divisibility) makes the class unusable, and
// the walk then steps through the superblock's segments one by one.
#define REPLAY_SUPER 64
template <class T>
__device__ __forceinline__ void replay_compose_chain( const ReplaySeg* segs, int n, int c0, ReplaySeg& out )
{
  typedef typename Bits<T>::U U;
  constexpr int MB = Bits<T>::MBITS;
  constexpr int KMAX = MB == 23 ? 20 : 8;
  const U mmask = ( (U)1 << MB ) - 1;
  const long long big = 1ll << ( MB + 3 );                        // |delta| < 2^MB
  U base = (U)segs[0].start | (U)c0;
  if( c0 == 0 ) { out.start = segs[0].start; out.pad = 0ull; }
  long long dmin = -big, dmax = big;
  int modlog = 2, K = 0;                                           // delta is a multiple of 4 (class); grid shift so far
  bool valid = true;
  for( int s = 0; s < n && valid; ++s )
  {
    const U st = (U)segs[s].start;
    if( ( ( base ^ st ) & ~mmask ) != 0 ) { valid = false; break; }
    const int c = (int)( base & ( REPLAY_CLS - 1 ) );
    const ReplayCls r = segs[s].cls[c];
    if( r.dmax < r.dmin ) { valid = false; break; }
    const long long cst = (long long)( base & mmask ) - (long long)( ( st & mmask ) | (U)c );        // a multiple of 4
    if( ( cst & ( ( 2ll << r.need_k ) - 1 ) ) != 0 ) { valid = false; break; }
    // r.dmin <= cst + delta * 2^-K <= r.dmax
    long long lo = r.dmin - cst, hi = r.dmax - cst;
    lo = lo < -big ? -big : lo; hi = hi > big ? big : hi;
    if( K >= 0 ) { lo *= ( 1ll << K ); hi *= ( 1ll << K ); }
    else { const int sh = -K; const long long rnd = ( 1ll << sh ) - 1; lo = lo >= 0 ? ( ( lo + rnd ) >> sh ) : -( ( -lo ) >> sh ); hi = hi >= 0 ? ( hi >> sh ) : -( ( -hi + rnd ) >> sh ); }
    dmin = lo > dmin ? lo : dmin; dmax = hi < dmax ? hi : dmax;
    // delta * 2^-K must keep the class (multiple of 4) and the segment's divisibility
    const int need = K + ( r.need_k + 1 > 2 ? r.need_k + 1 : 2 );
    modlog = need > modlog ? need : modlog;
    // next base, next grid
    const U eb = (U)r.end;
    const long long adv = r.k_end >= 0 ? ( cst >> r.k_end ) : cst * ( 1ll << -r.k_end );
    base = ( eb & ~mmask ) | (U)( (long long)( eb & mmask ) + adv );
    K += r.k_end;
    if( K > KMAX || K < -KMAX || modlog > MB ) valid = false;
  }
  ReplayCls q;
  q.dmin = valid ? dmin : 1; q.dmax = valid ? dmax : 0;
  q.end = (unsigned long long)base; q.need_k = modlog - 1; q.k_end = valid ? K : 0;
  out.cls[c0] = q;
}

// four accumulator rows per workgroup: their superblock's 64 records are staged in LDS (coalesced), then one thread per
// (row, class) composes from there (a thread chasing 64 dependent records in global memory took 50-100 us)
#define REPLAY_COMPOSE_ROWS 4
template <int PASS>
__global__ __launch_bounds__( WAVE ) void k_replay_compose( IcpLaunch L, ReplayBufs B )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) ReplaySeg stage[REPLAY_COMPOSE_ROWS][REPLAY_SUPER];
  static_assert( sizeof( ReplaySeg ) % 16 == 0, "staged with 16-byte copies" );
  const int prob = blockIdx.z, sb = blockIdx.x, row0 = blockIdx.y * REPLAY_COMPOSE_ROWS;
  if( L.active[prob] == 0 ) return;
  constexpr int NF = ReplayRows<PASS>::NF, NR = NF + ReplayRows<PASS>::ND;
  const int g0 = sb * REPLAY_SUPER, n = min( REPLAY_SUPER, B.n_seg - g0 );
  constexpr int Q = sizeof( ReplaySeg ) / 16;
  for( int r = 0; r < REPLAY_COMPOSE_ROWS && row0 + r < NR; ++r )
  {
    const uint4* src = reinterpret_cast<const uint4*>( B.seg + ( (size_t)prob * ICP_NMOM + row0 + r ) * B.n_seg + g0 );
    uint4* dst = reinterpret_cast<uint4*>( stage[r] );
    for( int k = threadIdx.x; k < n * Q; k += WAVE ) dst[k] = src[k];
  }
  __syncthreads();
  const int r = threadIdx.x / REPLAY_CLS, c = threadIdx.x % REPLAY_CLS, row = row0 + r;
  if( r >= REPLAY_COMPOSE_ROWS || row >= NR ) return;
  ReplaySeg& out = B.super[( (size_t)prob * ICP_NMOM + row ) * B.n_super + sb];
  if( row < NF ) replay_compose_chain<float>( stage[r], n, c, out );
  else           replay_compose_chain<double>( stage[r], n, c, out );
}

// The walk of one accumulator over its segments, with the exact value.  Lane l of the wave holds the record of segment g0 + l
// in registers; step j fetches lane j's fields with v_readlane (j is uniform), so the running value, the record and all the
// arithmetic of a step live in SCALAR registers: no memory access and no vector-ALU latency on the chain of 10^3-10^4 dependent
// steps (a version that read the records from LDS spent 740 cycles per step).
template <class T> struct ReplayFields;
template <> struct ReplayFields<float>  { typedef int I; };
template <> struct ReplayFields<double> { typedef long long I; };

__device__ __forceinline__ int rl( int v, int lane ) { return __builtin_amdgcn_readlane( v, lane ); }
__device__ __forceinline__ uint32_t rl( uint32_t v, int lane ) { return (uint32_t)__builtin_amdgcn_readlane( (int)v, lane ); }
__device__ __forceinline__ long long rl( long long v, int lane )
{
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane( (int)(uint32_t)v, lane ), hi = (uint32_t)__builtin_amdgcn_readlane( (int)(uint32_t)( (unsigned long long)v >> 32 ), lane );
  return (long long)( ( (unsigned long long)hi << 32 ) | lo );
}
__device__ __forceinline__ unsigned long long rl( unsigned long long v, int lane ) { return (unsigned long long)rl( (long long)v, lane ); }
__device__ __forceinline__ uint32_t first_lane( uint32_t v ) { return (uint32_t)__builtin_amdgcn_readfirstlane( (int)v ); }
__device__ __forceinline__ unsigned long long first_lane( unsigned long long v )
{
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane( (int)(uint32_t)v ), hi = (uint32_t)__builtin_amdgcn_readfirstlane( (int)(uint32_t)( v >> 32 ) );
  return ( (unsigned long long)hi << 32 ) | lo;
}

// one lane's record, compact and typed, for the scalar walk.  Per class the tests are prepared as whole-word bounds — the value's sign,
// exponent and mantissa together: lo <= value <= hi holds the exponent / sign match and the interval of the shift at once — so that a
// step of the walk is five register reads and a dozen scalar instructions.
template <class T> struct ReplayLaneRec
{
  typedef typename Bits<T>::U U; typedef typename ReplayFields<T>::I I;
  U lo[REPLAY_CLS], hi[REPLAY_CLS], base[REPLAY_CLS], end[REPLAY_CLS]; int meta[REPLAY_CLS];
  __device__ __forceinline__ void clear() { for( int c = 0; c < REPLAY_CLS; ++c ) { lo[c] = 1; hi[c] = 0; base[c] = 0; end[c] = 0; meta[c] = 64 << 8; } }
  __device__ __forceinline__ void load( const ReplaySeg& q )
  {
    const U mmask = ( (U)1 << Bits<T>::MBITS ) - 1;
    const long long clampv = 1ll << ( Bits<T>::MBITS + 2 );       // |delta| < 2^MBITS: bounds beyond that say nothing
    const U st = (U)q.start, top = st & ~mmask;
#pragma unroll
    for( int c = 0; c < REPLAY_CLS; ++c )
    {
      const long long dlo = q.cls[c].dmin, dhi = q.cls[c].dmax;
      const long long bm = (long long)( ( st & mmask ) | (U)c );                                   // the class start's mantissa
      long long l = bm + ( dlo < -clampv ? -clampv : ( dlo > clampv ? clampv : dlo ) ), h = bm + ( dhi < -clampv ? -clampv : ( dhi > clampv ? clampv : dhi ) );
      l = l < 0 ? 0 : l; h = h > (long long)mmask ? (long long)mmask : h;
      const bool never = dhi < dlo || l > h;
      lo[c] = never ? (U)1 : ( top | (U)l ); hi[c] = never ? (U)0 : ( top | (U)h );
      base[c] = top | (U)bm;
      end[c] = (U)q.cls[c].end;
      meta[c] = q.cls[c].need_k | ( ( q.cls[c].k_end + 64 ) << 8 );
    }
  }
  // lane j's record applied to the (uniform) value sb: true and sb advanced, or false
  __device__ __forceinline__ bool apply( int j, U& sb ) const
  {
    const U mmask = ( (U)1 << Bits<T>::MBITS ) - 1;
    const int c = (int)( sb & ( REPLAY_CLS - 1 ) );
    U l, h, bs, eb; int m;
    switch( c )                                          // (uniform)
    {
      case 0:  l = rl( lo[0], j ); h = rl( hi[0], j ); bs = rl( base[0], j ); eb = rl( end[0], j ); m = rl( meta[0], j ); break;
      case 1:  l = rl( lo[1], j ); h = rl( hi[1], j ); bs = rl( base[1], j ); eb = rl( end[1], j ); m = rl( meta[1], j ); break;
      case 2:  l = rl( lo[2], j ); h = rl( hi[2], j ); bs = rl( base[2], j ); eb = rl( end[2], j ); m = rl( meta[2], j ); break;
      default: l = rl( lo[3], j ); h = rl( hi[3], j ); bs = rl( base[3], j ); eb = rl( end[3], j ); m = rl( meta[3], j ); break;
    }
    const I d = (I)( sb - bs );                                                               // (same sign and exponent where it counts: the mantissas' difference, a multiple of 4)
    const int need_k = m & 255, k_end = ( m >> 8 ) - 64;
    const bool ok = sb >= l && sb <= h && ( d & ( ( (I)2 << need_k ) - 1 ) ) == 0;
    if( ok )
    {
      const I adv = k_end >= 0 ? ( d >> k_end ) : d * ( (I)1 << -k_end );
      sb = ( eb & ~mmask ) | (U)( (I)( eb & mmask ) + adv );
    }
    return ok;
  }
};

template <class T, int PASS>
__device__ __forceinline__ T replay_walk_row( const IcpLaunch& L, const ReplayBufs& B, int prob, int row, const FaithPar& P,
                                               float ( *term )[REPLAY_PITCH], int lane, int* n_redone )
{
  typedef typename Bits<T>::U U;
  const ReplaySeg* segs = B.seg + ( (size_t)prob * ICP_NMOM + row ) * B.n_seg;
  const ReplaySeg* sups = B.super + ( (size_t)prob * ICP_NMOM + row ) * B.n_super;
  U sb = 0;                                             // the running value's bits (uniform): +0
  int redone = 0;
  // (Fetching superblock S + 1's segment records while S is walked — they do not depend on the value — was tried for the chains that fit
  //  none of their superblocks' composed records: 1 028 instead of 970 us per iteration on the centred 1.15 M-point scans, 640 instead of
  //  628 at 84 k points: the fetch and its 60 instructions of unpacking are then paid at EVERY superblock.)
  for( int s0 = 0; s0 < B.n_super; s0 += WAVE )
  {
    ReplayLaneRec<T> sup; sup.clear();                  // lane l: superblock s0 + l
    if( s0 + lane < B.n_super ) sup.load( sups[s0 + lane] );
    const int n_sup = min( WAVE, B.n_super - s0 );
    for( int js = 0; js < n_sup; ++js )
    {
      if( sup.apply( js, sb ) ) continue;               // 64 segments in one step
      // step through the superblock's segments
      const int g0 = ( s0 + js ) * REPLAY_SUPER, n_here = min( REPLAY_SUPER, B.n_seg - g0 );
      ReplayLaneRec<T> seg; seg.clear();
      if( lane < n_here ) seg.load( segs[g0 + lane] );
      for( int j = 0; j < n_here; ++j )
      {
        if( seg.apply( j, sb ) ) continue;
        // re-add the segment's addends one after the other (uniform over the wave: the value and the record are)
        ++redone;
        __syncthreads();                                // (one wave per block: orders the reuse of `term`)
        replay_term_row<PASS>( L, prob, g0 + j, P, term, row );
        __syncthreads();
        T acc = Bits<T>::from( sb );
        const float4* row4 = reinterpret_cast<const float4*>( term[row] );
#pragma unroll 8
        for( int t4 = 0; t4 < REPLAY_SEG / 4; ++t4 ) { const float4 v = row4[t4]; acc += (T)v.x; acc += (T)v.y; acc += (T)v.z; acc += (T)v.w; }
        sb = first_lane( Bits<T>::of( acc ) );
      }
    }
  }
  if( n_redone ) *n_redone = redone;
  return Bits<T>::from( sb );
}

template <int PASS>
__global__ __launch_bounds__( WAVE ) void k_replay_walk( IcpLaunch L, ReplayBufs B )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) float term[ICP_NMOM][REPLAY_PITCH];
  const int prob = blockIdx.y, row = blockIdx.x;
  if( L.active[prob] == 0 ) return;
  FaithPar P;
  double* totals = B.totals + (size_t)prob * 3 * ICP_NMOM;
  if( !replay_params( L, prob, PASS, totals, P ) ) return;
  constexpr int NF = ReplayRows<PASS>::NF;
  const int lane = threadIdx.x;
  int redone = 0;
  double v;
  if( row < NF ) v = (double)replay_walk_row<float, PASS>( L, B, prob, row, P, term, lane, &redone );
  else           v = replay_walk_row<double, PASS>( L, B, prob, row, P, term, lane, &redone );
  if( lane == 0 )
  {
    totals[( PASS - 1 ) * ICP_NMOM + row] = v;        // (floats are exact in a double)
    if( B.redone ) atomicAdd( B.redone + prob, redone );
  }
}

// the rest of the iteration (icp.h:253-295, 455-493), as k_icp_faithful's last thread does it
__global__ __launch_bounds__( WAVE ) void k_replay_finish( IcpLaunch L, ReplayBufs B )
{
  const int prob = blockIdx.x;
  if( L.active[prob] == 0 ) return;
  icp_bind( L, prob );
  if( L.solve ) icp_iteration_reset( L, prob );            // (by the whole wave: it averages a sample of per-tile counts)
  if( threadIdx.x != 0 ) return;
  const double* totals = B.totals + (size_t)prob * 3 * ICP_NMOM;
  FaithPar P;
  const bool ok = replay_params( L, prob, 3, totals, P );
  if( !ok )                                             // icp.h:455-459 / 466-470: no correspondences, or the weights vanished
  {
    if( L.solve ) { L.prev_err[prob] = L.err[prob]; L.iters[prob] += 1; L.active[prob] = 0; }
    return;
  }
  const double* t3 = totals + 2 * ICP_NMOM;
  float A[33];
  for( int k = 0; k < 33; ++k ) A[k] = (float)t3[k];
  Mat4 T;
  for( int k = 0; k < 16; ++k ) { T.m[k] = L.T1[prob * 16 + k]; if( L.solve ) L.T1_prev[prob * 16 + k] = T.m[k]; }
  float e;
  icp_solve_ref_order( A, t3[33], t3[34], P.c1, T, e );
  for( int k = 0; k < 16; ++k ) L.T1[prob * 16 + k] = T.m[k];
  if( !L.solve ) { L.err[prob] = e; return; }
  L.prev_err[prob] = L.err[prob];
  L.iters[prob] += 1;
  L.err[prob] = e;
  const float delta = fabsf( L.prev_err[prob] - e );
  if( !L.fixed_iters && L.iter_index > 5 && delta < 1e-5 ) L.active[prob] = 0;   // icp.h:489
}

template <int PASS>
static void launch_replay_pass( const IcpLaunch& L, const ReplayBufs& B, hipStream_t st )
{
  constexpr int NR = ReplayRows<PASS>::NF + ReplayRows<PASS>::ND;
  hipLaunchKernelGGL( k_replay_sums<PASS>, dim3( B.n_seg, L.n_prob ), dim3( REPLAY_SEG ), 0, st, L, B );
  hipLaunchKernelGGL( k_replay_scan<PASS>, dim3( NR, L.n_prob ), dim3( REPLAY_SCAN_THREADS ), 0, st, L, B );
  hipLaunchKernelGGL( k_replay_run<PASS>, dim3( B.n_seg, L.n_prob ), dim3( REPLAY_RUN_THREADS ), 0, st, L, B );
  hipLaunchKernelGGL( k_replay_compose<PASS>, dim3( B.n_super, ( NR + REPLAY_COMPOSE_ROWS - 1 ) / REPLAY_COMPOSE_ROWS, L.n_prob ), dim3( WAVE ), 0, st, L, B );
  hipLaunchKernelGGL( k_replay_walk<PASS>, dim3( NR, L.n_prob ), dim3( WAVE ), 0, st, L, B );
}
void launch_icp_replay( const IcpLaunch& L, const ReplayBufs& B, hipStream_t st )
{
  hipLaunchKernelGGL( k_icp_faith_gather, dim3( ( L.src.n + BLOCK - 1 ) / BLOCK, L.n_prob ), dim3( BLOCK ), 0, st, L );
  if( !L.w_explicit ) launch_replay_pass<1>( L, B, st );
  launch_replay_pass<2>( L, B, st );
  launch_replay_pass<3>( L, B, st );
  hipLaunchKernelGGL( k_replay_finish, dim3( L.n_prob ), dim3( WAVE ), 0, st, L, B );
}
// Large sources: k_icp_moments (parallel fp64) for everything but the two weighted centroids, whose seven sums run as the
// reference's sequential fp32 chains (pass 2 of the replay, with the moments' own 2.5-sigma cut); k_icp_update centres on them.
void launch_icp_exact_centroids( const IcpLaunch& L, const ReplayBufs& B, hipStream_t st )
{
  hipLaunchKernelGGL( k_icp_moments, dim3( L.n_mom_blocks, L.n_prob ), dim3( BLOCK ), 0, st, L );      // (also leaves n, mean, stddev in L.res)
  hipLaunchKernelGGL( k_icp_faith_gather, dim3( ( L.src.n + BLOCK - 1 ) / BLOCK, L.n_prob ), dim3( BLOCK ), 0, st, L );
  launch_replay_pass<2>( L, B, st );
  hipLaunchKernelGGL( k_icp_update, dim3( L.n_prob ), dim3( UPDATE_WAVES * WAVE ), 0, st, L );
}
// ------------------------------------------------------------------------------------------
// Grid chains: the reference's seven centroid sums (icp.h:136-148), bit for bit, at the cost of a reduction
//
// A sequential fp32 sum  s <- RN( s + x )  is an INTEGER sum while s stays inside one binade: with u = ulp( s ), s = M u,
// RN( s + x ) = ( M + rndne( x / u ) ) u  unless x / u sits exactly half way between two integers (then the parity of M decides) —
// and integer addition is associative.  So for a stretch of addends and an exponent E the whole effect on the chain is three
// integers: D = Σ rndne( x_i / u ) and the smallest / largest partial sum, which say for which start mantissas M the chain stays
// inside the binade all the way (a margin of one grid step at the ends keeps clear of the neighbouring binades' grids).  Such
// records compose (intervals intersect, advances add).  A tie is part of the record too (ChainFn below: what it adds depends on the
// parity of the start alone).  The only sequential part left is the handful of places where the chain really changes binade (~15
// times on the way from 0 to 2^21) or sign: there the addends of one segment are added one after the other in fp32.
//
//   the searches    leave one 48-byte record per source point at the point's ORIGINAL index (icp_emit);
//   k_chain_segrecs one record per (segment of 64 points, chain): the functions for the binades e-1, e, e+1 around a guess e of the
//                   running sum's exponent there, and what the segment adds (the guessed binade's own advance, i.e. the chain's sum
//                   with its rounding drift — what the walks forecast with);
//   k_chain_compose a block's 64 segment records composed (per chain and binade), the quarter blocks' sums;
//   k_chain_walk_and_moments   ONE launch for three things that do not need each other:
//       the walks   one workgroup per chain (chain_walk_row: forecasts, fetches, the walk proper — ~35 us for a 1 M-point scan);
//       the moments the fp64 moments of k_icp_moments, read from the records in the reference's order (a quarter block per workgroup);
//       the guesses for the NEXT iteration's records, from this iteration's sums (k_chain_guess's work, a block per workgroup);
//   k_icp_update_wide   finishes the iteration (icp.h:253-295,455-493), centred on the chains' centroids.
// The first iteration has no guesses yet: k_chain_moments (with the segments' fp64 sums) and k_chain_guess run before the records,
// k_chain_walk alone after them.
// ------------------------------------------------------------------------------------------
struct ChainPar { bool use_sd; float cut, max_dist; };

// the seven addends of one source point (faith_terms<2>: icp.h:141-142, weights icp.h:387,396-401)
__device__ __forceinline__ void chain_addends( const float4& A, const float4& Bq, const ChainPar& P, float x[CH_ROWS], float& w )
{
  const bool m = A.w >= 0.0f;
  w = 0.0f;
  if( m )
  {
    w = ( 1.0f - __fdiv_rn( A.w, P.max_dist ) ) * Bq.w;
    if( P.use_sd && A.w > P.cut ) w = 0.0f;
  }
  x[0] = w;
  x[1] = m ? A.x * w : 0.0f;  x[2] = m ? A.y * w : 0.0f;  x[3] = m ? A.z * w : 0.0f;
  x[4] = m ? Bq.x * w : 0.0f; x[5] = m ? Bq.y * w : 0.0f; x[6] = m ? Bq.z * w : 0.0f;
}

// n_corr, mean, stddev of dist² from the searches' integer statistics (as k_icp_moments); every thread of the block gets the same bits
__device__ __forceinline__ float chain_stats( const IcpLaunch& L, int prob, unsigned long long ( *s_stat )[3], double* st_out )
{
  static_assert( STAT_SHARDS == BLOCK, "one shard per thread" );
  const unsigned long long* a = L.stat_acc + ( (size_t)prob * STAT_SHARDS + ( threadIdx.x & ( BLOCK - 1 ) ) ) * 4;
  const bool mine = threadIdx.x < BLOCK;
  const unsigned long long c0 = wave_sum_u64( mine ? a[0] : 0ull ), c1 = wave_sum_u64( mine ? a[1] : 0ull ), c2 = wave_sum_u64( mine ? a[2] : 0ull );
  if( mine && ( threadIdx.x & ( WAVE - 1 ) ) == 0 ) { unsigned long long* o = s_stat[threadIdx.x / WAVE]; o[0] = c0; o[1] = c1; o[2] = c2; }
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
  for( int w = 0; w < WAVES_PER_BLOCK; ++w ) { t0 += s_stat[w][0]; t1 += s_stat[w][1]; t2 += s_stat[w][2]; }
  const double n = (double)t0;
  const float mean = (float)( (double)t1 * L.stat_i1 / n );           // sum / (float)n
  const float sqm = (float)( (double)t2 * L.stat_i2 / n );            // sq_sum / (float)n
  const float var = sqm - mean * mean;
  const float sd = (float)sqrt( (double)var );                        // (float)sqrt( ... ), msh_std.h:1824
  if( st_out ) { st_out[0] = n; st_out[1] = mean; st_out[2] = sd; st_out[3] = (double)L.queue_count[prob]; }
  return sd;
}

// One workgroup per QUARTER of a block of 64 segments (1 024 source points in the reference's order): the fp64 moments' partials, the
// seven chains' fp64 sums per segment, and per quarter block.  Block 0 of the launch leaves n, mean, stddev and the cut's stddev
// in L.res for the kernels that follow.
#define CH_QUARTERS 4
struct ChainMomLds { double red[WAVES_PER_BLOCK][ICP_NMOM]; double bsum[WAVES_PER_BLOCK][CH_ROWS]; unsigned long long stat[WAVES_PER_BLOCK][3]; };
__device__ __forceinline__ void chain_moments_block( const IcpLaunch& L, const ChainBufs& B, int prob, int qb, ChainMomLds& S )
{
  // quarter block qb: segments [16 qb, 16 qb + 16)
  const float sd = chain_stats( L, prob, S.stat, ( qb == 0 && threadIdx.x == 0 ) ? L.res + (size_t)prob * ICP_NRES + ICP_NMOM : nullptr );
  ChainPar P; P.use_sd = sd > 0.000001; P.cut = 2.5f * sd; P.max_dist = L.radius;
  const int lane = threadIdx.x & ( WAVE - 1 ), wib = threadIdx.x / WAVE;
  const float4* R = L.rec + (size_t)prob * L.src.n * REC_F4;

  double acc[ICP_NMOM], bs[CH_ROWS];
#pragma unroll
  for( int k = 0; k < ICP_NMOM; ++k ) acc[k] = 0.0;
#pragma unroll
  for( int r = 0; r < CH_ROWS; ++r ) bs[r] = 0.0;
  constexpr int SEGS = CH_BLK / CH_QUARTERS;
  for( int sb = wib; sb < SEGS; sb += WAVES_PER_BLOCK )              // a wave's 64 lanes = one segment
  {
    const int seg = qb * SEGS + sb;
    if( seg >= B.n_seg ) break;
    const int i = seg * CH_SEG + lane;
    float4 A = make_float4( 0.0f, 0.0f, 0.0f, -1.0f ), Q = make_float4( 0.0f, 0.0f, 0.0f, 0.0f ), N4 = Q;
    if( i < L.src.n ) { A = R[(size_t)i * REC_F4]; Q = R[(size_t)i * REC_F4 + 1]; N4 = R[(size_t)i * REC_F4 + 2]; }
    float x[CH_ROWS], w;
    chain_addends( A, Q, P, x, w );
    if( B.refresh )
    {
#pragma unroll
      for( int r = 0; r < CH_ROWS; ++r )
      {
        const double v = wave_sum( (double)x[r] );
        bs[r] += v;
        if( lane == 0 ) B.segsum[( (size_t)prob * CH_ROWS + r ) * B.n_seg + seg] = v;
      }
    }
    if( A.w < 0.0f ) continue;
    const double W = w, p[3] = { A.x, A.y, A.z }, q[3] = { Q.x, Q.y, Q.z }, n[3] = { N4.x, N4.y, N4.z };
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
  { const double v = wave_sums( acc, lane ); if( lane < ICP_NMOM ) S.red[wib][lane] = v; }
  if( lane == 0 ) { for( int r = 0; r < CH_ROWS; ++r ) S.bsum[wib][r] = bs[r]; }
  __syncthreads();
  if( threadIdx.x < ICP_NMOM )
  {
    double v = 0.0;
    for( int w = 0; w < WAVES_PER_BLOCK; ++w ) v += S.red[w][threadIdx.x];
    L.mom_part[( (size_t)prob * ICP_NMOM + threadIdx.x ) * L.n_mom_blocks + qb] = v;      // (moment-major: k_icp_update reads a moment's partials coalesced)
  }
  if( B.refresh && threadIdx.x >= WAVE && threadIdx.x < WAVE + CH_ROWS )
  {
    double v = 0.0;
    for( int w = 0; w < WAVES_PER_BLOCK; ++w ) v += S.bsum[w][threadIdx.x - WAVE];
    B.blksum[( (size_t)prob * CH_ROWS + ( threadIdx.x - WAVE ) ) * ( B.n_blk * CH_QUARTERS ) + qb] = v;
  }
}
__global__ __launch_bounds__( BLOCK ) void k_chain_moments( IcpLaunch L, ChainBufs B )
{
  RS_CHAIN_SETPRIO();
  __shared__ ChainMomLds S;
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  chain_moments_block( L, B, prob, blockIdx.x, S );
}

#define CH_M_LO ( 1 << 23 )
#define CH_M_HI ( ( 1 << 24 ) - 1 )
// A record's function for one exponent is  M -> M + D, valid for lo <= M <= hi  (lo > hi: never).  A run of records f_0 .. f_l
// applies to a start mantissa M iff  lo_j <= M + D_0 + .. + D_(j-1) <= hi_j  for every j, i.e. iff
//     max_j ( lo_j - Dex_j )  <=  M  <=  min_j ( hi_j - Dex_j ),      Dex_j = the advance of the records before j,
// and then advances it by D_0 + .. + D_l: three integer prefix scans over the lanes (sum, max, min), each six DPP instructions.
// (A never-record has lo - Dex > hi - Dex, so the max passes the min from its lane on: nothing fits any more.)
//
// Ties.  An addend that lands exactly half way between two grid points is rounded to the EVEN one: M + k + ( ( M + k ) & 1 ) — what
// it adds depends on the parity of the value it meets, i.e. on the parity of the record's start mantissa.  After it the value is
// even, whatever it was: later ties of the same record are decided.  So a record with ties is  M -> M + D + tau[ M & 1 ]  with two
// small numbers tau[0], tau[1] (a segment: { c, 1 - c }; a block: composed, chain_compose) — kept in the low four bits of the D
// word — and its interval is narrowed by max tau.  The scans below take D alone; whoever applies a run of records adds, in order,
// the tau each start's parity picks (advance in chain_walk_row, chain_compose_block), having left room for the most they can add.
struct ChainFn { int lo, hi, D, tau; };
__device__ __forceinline__ ChainFn chain_never() { ChainFn f; f.lo = CH_M_HI; f.hi = CH_M_LO; f.D = 0; f.tau = 0; return f; }
__device__ __forceinline__ ChainFn chain_identity() { ChainFn f; f.lo = CH_M_LO; f.hi = CH_M_HI; f.D = 0; f.tau = 0; return f; }
__device__ __forceinline__ int chain_tau( int tau, int parity ) { return ( tau >> ( 2 * ( parity & 1 ) ) ) & 3; }
// the function of a record for the (biased) exponent E and sign bit sg of the running value
__device__ __forceinline__ ChainFn chain_select( const ChainRec& r, int E, int sg )
{
  const int c = E - ( r.e_sign & 255 ) + 1;
  ChainFn f = chain_never();
  if( ( ( r.e_sign >> 8 ) & 1 ) == sg )
  {
    int d = 0;
    if( c == 0 ) { f.lo = r.lo[0]; f.hi = r.hi[0]; d = r.D[0]; }
    if( c == 1 ) { f.lo = r.lo[1]; f.hi = r.hi[1]; d = r.D[1]; }
    if( c == 2 ) { f.lo = r.lo[2]; f.hi = r.hi[2]; d = r.D[2]; }
    f.D = d >> 4; f.tau = d & 15;
  }
  return f;
}
// ... for the binade of the value with the bits vb.  A value of exactly zero has no binade: it stays zero through a stretch whose
// addends are ALL zero (bit 16 of e_sign — the unmatched points a scan may well begin with; without it every such segment would be
// added up addend by addend), anything else from there is added one by one (as are denormals, inf and NaN).
#define CH_ALL_ZERO ( 1 << 16 )
__device__ __forceinline__ ChainFn chain_fn_for( const ChainRec& r, uint32_t vb )
{
  const int E = (int)( ( vb >> 23 ) & 255u ), sg = (int)( vb >> 31 );
  if( E == 0 ) return ( ( vb << 1 ) == 0u && ( r.e_sign & CH_ALL_ZERO ) ) ? chain_identity() : chain_never();
  if( E == 255 ) return chain_never();
  return chain_select( r, E, sg );
}
// inclusive prefix max / min over the 64 lanes (signed), like wave_scan: lanes without a source lane keep their own value
#define RS_DPP_PREFIX( OP, v )                                                              \
  asm volatile( "s_nop 4\n\t"                                                               \
                OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"        \
                OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"        \
                OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"        \
                OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"        \
                OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"     \
                OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"         \
                : "+v"( v ) )
// Lane l ends with the run f_0 .. f_l as one function (normalised like a record; the caller's M test is  lo <= M <= hi).
__device__ __forceinline__ ChainFn chain_prefix( const ChainFn& f, int lane )
{
  const int incl = (int)wave_scan( (uint32_t)f.D, lane );
  const int ex = incl - f.D;
  int a = f.lo - ex, b = f.hi - ex;                 // (|D| < 2^24 per valid record, 0 for a never-record: no overflow over 64 lanes)
  RS_DPP_PREFIX( "v_max_i32_dpp", a );
  RS_DPP_PREFIX( "v_min_i32_dpp", b );
  ChainFn g;
  g.lo = max( a, CH_M_LO ); g.hi = min( b, CH_M_HI ); g.D = incl; g.tau = f.tau;
  const bool never = g.lo > g.hi;
  g.lo = never ? CH_M_HI : g.lo; g.hi = never ? CH_M_LO : g.hi; g.D = never ? 0 : g.D;
  return g;
}

// The guesses (refresh iterations): every chain's fp64 prefix at each segment's start — the quarter blocks before from their sums,
// then a scan of the block's own 64 segment sums — kept as exponent | sign << 8 per (chain, segment).  One workgroup per block.
__device__ __forceinline__ double wave_scan_f64( double v, int lane )      // inclusive
{
#pragma unroll
  for( int d = 1; d < WAVE; d <<= 1 ) { const double up = __shfl_up( v, d ); if( lane >= d ) v += up; }
  return v;
}
__device__ __forceinline__ void chain_guess_block( const IcpLaunch& L, const ChainBufs& B, int prob, int blk, int n_waves )
{
  const int lane = threadIdx.x & ( WAVE - 1 );
  for( int r = threadIdx.x / WAVE; r < CH_ROWS; r += n_waves )
  {
    const double* bsum = B.blksum + ( (size_t)prob * CH_ROWS + r ) * ( B.n_blk * CH_QUARTERS );
    double before = 0.0;
    for( int b = lane; b < blk * CH_QUARTERS; b += WAVE ) before += bsum[b];
    before = wave_sum( before );
    const int seg = blk * CH_BLK + lane;
    const double v = seg < B.n_seg ? B.segsum[( (size_t)prob * CH_ROWS + r ) * B.n_seg + seg] : 0.0;
    const double incl = wave_scan_f64( v, lane );
    const uint32_t gb = __float_as_uint( (float)( before + ( incl - v ) ) );
    if( seg < B.n_seg ) B.guess[( (size_t)prob * CH_ROWS + r ) * B.n_seg + seg] = (int)( ( ( gb >> 23 ) & 255u ) | ( ( gb >> 31 ) << 8 ) );
  }
}
__global__ __launch_bounds__( CH_ROWS * WAVE ) void k_chain_guess( IcpLaunch L, ChainBufs B )
{
  RS_CHAIN_SETPRIO();
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  chain_guess_block( L, B, prob, blockIdx.x, CH_ROWS );
}

// The block records: a block's 64 segment records composed per chain and exponent (around the block's first guess), and the quarter
// blocks' sums of the segments' (the walks' forecasts, the next guesses).
__device__ __forceinline__ void chain_rec_copy_fwd( ChainRec& d, const ChainRec& r )
{
  d.e_sign = r.e_sign;
#pragma unroll
  for( int c = 0; c < 3; ++c ) { d.lo[c] = r.lo[c]; d.hi[c] = r.hi[c]; d.D[c] = r.D[c]; }
}
// One workgroup per (block, chain): three waves, one per binade; the first also sums the quarters.  (One workgroup per block with all
// seven chains — 21 scans over 8 waves, 18 KB staged — took 11 us per launch, nearly all of it the latency of that one workgroup's
// load -> scans -> store; 2 000 small ones take 7.)
__device__ __forceinline__ void chain_compose_block( const ChainBufs& B, int prob, int blk, int r )
{
  const int lane = threadIdx.x & ( WAVE - 1 ), c = uni( (int)threadIdx.x / WAVE );
  const int seg = blk * CH_BLK + lane;
  ChainRec mine; mine.e_sign = -1;
  const ChainRec* src = B.seg + ( (size_t)prob * CH_ROWS + r ) * B.n_seg;
  chain_rec_copy_fwd( mine, src[min( seg, B.n_seg - 1 )] );
  if( seg >= B.n_seg ) mine.e_sign = -1;                                  // (past the end of the cloud)
  if( c == 0 )
  {
    const double v = seg < B.n_seg ? B.segsum[( (size_t)prob * CH_ROWS + r ) * B.n_seg + seg] : 0.0;
    // the quarter blocks' sums: a row of 16 lanes each (fixed order: lane 0's quad tree)
    double q = v;
    q += dpp_d<RS_DPP_QUAD_XOR1, 0xf>( 0.0, q ); q += dpp_d<RS_DPP_QUAD_XOR2, 0xf>( 0.0, q );
    q += dpp_d<RS_DPP_HALF_MIRROR, 0xf>( 0.0, q ); q += dpp_d<RS_DPP_ROW_MIRROR, 0xf>( 0.0, q );
    if( ( lane & 15 ) == 0 ) B.blksum[( (size_t)prob * CH_ROWS + r ) * ( B.n_blk * CH_QUARTERS ) + blk * CH_QUARTERS + ( lane >> 4 )] = q;
  }
  const int first = __builtin_amdgcn_readlane( mine.e_sign, 0 );
  const int E = ( first & 255 ) - 1 + c, sg = ( first >> 8 ) & 1;
  const bool all_zero = RS_BALLOT( mine.e_sign != -1 && !( mine.e_sign & CH_ALL_ZERO ) ) == 0ull;
  const ChainFn f0 = mine.e_sign == -1 ? chain_identity() : chain_select( mine, E, sg );
  ChainFn f = chain_prefix( f0, lane );
  // the ties inside, in order: what the block adds for an even / an odd start (each record's tau picked by the parity of ITS start)
  unsigned long long tm = RS_BALLOT( f0.tau != 0 );
  const int ex = f.D - f0.D;
  int t0 = 0, t1 = 0, tmax = 0;
  while( tm != 0ull )
  {
    const int k = __builtin_ctzll( tm ); tm &= tm - 1ull;
    const int exk = __builtin_amdgcn_readlane( ex, k ), tk = __builtin_amdgcn_readlane( f0.tau, k );
    t0 += chain_tau( tk, exk + t0 ); t1 += chain_tau( tk, 1 + exk + t1 ); tmax += max( tk & 3, tk >> 2 );
  }
  if( lane == WAVE - 1 )
  {
    ChainRec* out = B.blk + ( (size_t)prob * CH_ROWS + r ) * B.n_blk + blk;
    if( c == 0 ) out->e_sign = ( first & 0x1ff ) | ( all_zero ? CH_ALL_ZERO : 0 );
    const int hi = f.hi - tmax;
    const bool ok = f.lo <= hi && t0 <= 3 && t1 <= 3;                   // (a never-record has lo > hi already)
    out->lo[c] = ok ? f.lo : CH_M_HI; out->hi[c] = ok ? hi : CH_M_LO; out->D[c] = ok ? f.D * 16 + t0 + 4 * t1 : 0;
  }
}

// The segment records: a wave stages THREE consecutive segments' addends in LDS (lane = point), then lane = (segment, exponent
// class, chain) runs down its segment's 64 addends in integers — no cross-lane traffic, 63 records at once.  The guesses are the
// kept ones (k_chain_guess): between two ICP iterations the sums move by a few per cent at most (the radius shrinks by 5 %, a per
// cent of the correspondences change), far less than the factor of two a record's three exponents cover.
#define CHAIN_REC_TASK 3          // segments per wave and round
#define CHAIN_REC_ROUNDS 1
__global__ __launch_bounds__( BLOCK ) void k_chain_segrecs( IcpLaunch L, ChainBufs B )
{
  RS_CHAIN_SETPRIO();
  // (rows of 65: lane = (segment, class, chain) reads row (segment, chain) at column j — with rows of 64 all 21 rows' column j sit in
  //  ONE bank, a 21-way conflict on every read of the loop below)
  __shared__ float s_x[WAVES_PER_BLOCK][CHAIN_REC_TASK][CH_ROWS][CH_SEG + 1];
  __shared__ int s_zero[WAVES_PER_BLOCK][CHAIN_REC_TASK][CH_ROWS];           // all 64 addends of (segment, chain) are zero
  __shared__ unsigned long long s_stat[WAVES_PER_BLOCK][3];
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  const int lane = threadIdx.x & ( WAVE - 1 ), wib = threadIdx.x / WAVE;
  // (CHAIN_REC_ROUNDS tasks per wave, the loads of all first.  One: with two — the second's loads in flight while the first is worked on,
  //  half as many waves — the launch took 29 us instead of 25: the kernel is bound by its ~1 100 vector instructions per task, not by
  //  the loads.)
  const float4* R = L.rec + (size_t)prob * L.src.n * REC_F4;
  float4 A2[CHAIN_REC_ROUNDS][CHAIN_REC_TASK], Q2[CHAIN_REC_ROUNDS][CHAIN_REC_TASK];
#pragma unroll
  for( int t = 0; t < CHAIN_REC_ROUNDS; ++t )
#pragma unroll
    for( int q = 0; q < CHAIN_REC_TASK; ++q )               // (the loads first: the cut's reduction below runs while they are in flight)
    {
      const int sg = ( ( blockIdx.x * CHAIN_REC_ROUNDS + t ) * WAVES_PER_BLOCK + wib ) * CHAIN_REC_TASK + q, i = sg * CH_SEG + lane;
      const size_t ic = (size_t)min( i, L.src.n - 1 );
      A2[t][q] = R[ic * REC_F4]; Q2[t][q] = R[ic * REC_F4 + 1];
      if( sg >= B.n_seg || i >= L.src.n ) { A2[t][q] = make_float4( 0.0f, 0.0f, 0.0f, -1.0f ); Q2[t][q] = make_float4( 0.0f, 0.0f, 0.0f, 0.0f ); }
    }
  const float sd = chain_stats( L, prob, s_stat, nullptr );                  // (its own: in the iterations that keep their guesses this kernel runs BEFORE the moments)
  ChainPar P; P.use_sd = sd > 0.000001; P.cut = 2.5f * sd; P.max_dist = L.radius;
#pragma unroll
  for( int t = 0; t < CHAIN_REC_ROUNDS; ++t )
  {
  const int task = ( blockIdx.x * CHAIN_REC_ROUNDS + t ) * WAVES_PER_BLOCK + wib, seg0 = task * CHAIN_REC_TASK;
  if( t > 0 ) wave_lds_fence();                                              // (the round before has read its rows)
#pragma unroll
  for( int q = 0; q < CHAIN_REC_TASK; ++q )
  {
    float x[CH_ROWS], w;
    chain_addends( A2[t][q], Q2[t][q], P, x, w );
#pragma unroll
    for( int r = 0; r < CH_ROWS; ++r )
    {
      s_x[wib][q][r][lane] = x[r];
      const unsigned long long nz = RS_BALLOT( x[r] != 0.0f );
      if( lane == 0 ) s_zero[wib][q][r] = nz == 0ull ? CH_ALL_ZERO : 0;
    }
  }
  wave_lds_fence();
  const int q = lane / ( CH_ROWS * 3 ), combo = lane % ( CH_ROWS * 3 );
  const int r = combo % CH_ROWS, c = combo / CH_ROWS;
  const int seg = seg0 + q;
  if( q < CHAIN_REC_TASK && seg < B.n_seg )
  {
  const int es = B.guess[( (size_t)prob * CH_ROWS + r ) * B.n_seg + seg];
  const int eg = es & 255, sg = es >> 8;
  const int E = eg - 1 + c;                                            // s = M * 2^(E - 150), M in [2^23, 2^24)
  int Pj = 0, pmin = 0, pmax = 0;                                      // partial sums, the start included
  const bool bad_e = E < 1 || E > 254;
  bool odd = bad_e, big = bad_e, seen = false;
  int cpar = 0;
  const float* xs = &s_x[wib][q][r][0];
#pragma unroll 8
  for( int j = 0; j < CH_SEG; ++j )
  {
    const float xv = xs[j];
    const float y = ldexpf( sg ? -xv : xv, 150 - E );                  // x / ulp( s ): exact (a power of two), or 0 / inf at the ends; the chain of |s| for negative s
    const float rn = rintf( y );                                       // to nearest, ties to even
    odd |= (int)!( fabsf( y ) < 8388608.0f ) | (int)( fabsf( y - rn ) == 0.5f );  // too big for this binade (or NaN), or a tie: M's parity decides
    Pj += (int)rn;
    pmin = min( pmin, Pj ); pmax = max( pmax, Pj );
  }
  float fsum = 0.0f;
  if( odd )        // one segment in a few hundred: again, telling the two apart and taking the ties as they fall (see ChainFn)
  {
    Pj = 0; pmin = 0; pmax = 0;
    int Pfa = 0;
    for( int j = 0; j < CH_SEG; ++j )
    {
      const float xv = xs[j];
      const float y = ldexpf( sg ? -xv : xv, 150 - E );
      const float rn = rintf( y );
      big |= !( fabsf( y ) < 8388608.0f );
      if( fabsf( y - rn ) == 0.5f )                                    // to the even neighbour
      {
        const int kl = (int)floorf( y );
        if( !seen ) { seen = true; cpar = ( Pj + kl ) & 1; Pj += kl; Pfa = Pj; }      // M + Pj + ( ( M + cpar ) & 1 ): even from here on, = "M' + ( Pj - Pfa )"
        else Pj += kl + ( ( Pj - Pfa + kl ) & 1 );
      }
      else Pj += (int)rn;
      pmin = min( pmin, Pj ); pmax = max( pmax, Pj );
      fsum += xv;
    }
    pmax += seen ? 1 : 0;
  }
  const bool bad = big;
  // What the segment adds to the chain, for the walks' forecasts of where the chain changes binade (chain_walk_row) and for the
  // next iteration's guesses: the guessed binade's own advance — D grid steps, i.e. the CHAIN's sum, its rounding drift included —
  // or, where the addends do not fit that grid, their plain sum.
  if( c == 1 )
    B.segsum[( (size_t)prob * CH_ROWS + r ) * B.n_seg + seg] = big ? (double)fsum : ldexp( (double)( sg ? -Pj : Pj ), E - 150 );
  // every value on the way, the start included, at least one grid step inside the binade: the neighbouring binades' grids
  // (half / twice as fine) then play no part in any of the roundings
  long long lo = (long long)CH_M_LO + 1 - pmin, hi = (long long)CH_M_HI - 1 - pmax;
  lo = lo < CH_M_LO ? CH_M_LO : lo; hi = hi > CH_M_HI ? CH_M_HI : hi;
  const bool ok = !bad && lo <= hi;
  ChainRec* out = B.seg + ( (size_t)prob * CH_ROWS + r ) * B.n_seg + seg;
  if( c == 0 ) out->e_sign = es | s_zero[wib][q][r];
  out->lo[c] = ok ? (int)lo : CH_M_HI; out->hi[c] = ok ? (int)hi : CH_M_LO; out->D[c] = ok ? Pj * 16 + ( seen ? ( cpar ? 1 : 4 ) : 0 ) : 0;      // tau = { cpar, 1 - cpar }
  }
  }
}

// (One launch per block.  Having the k_chain_segrecs workgroup that completes a block compose it — a counter per block, the last of
//  its six or seven to arrive — was tried: correct, and 300 us per launch instead of 25 + 11, because the 1 500 workgroups' release
//  fences each write back their XCD's L2, which the search has just filled with dirty records.)
__global__ __launch_bounds__( 3 * WAVE ) void k_chain_compose( IcpLaunch L, ChainBufs B )
{
  RS_CHAIN_SETPRIO();
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  chain_compose_block( B, prob, blockIdx.x, blockIdx.z );
}

// One chain walked by a workgroup of four waves.  `s` (wave 0's, uniform) is the exact running value.
//
// What a walk costs is set by two things.  Memory latency: a block that does not fit needs its segments' records, the segment that
// does not fit its 64 addends — two dependent round trips of ~2 us, ~20 times per chain.  And instruction issue: ONE wave issues
// ~250 instructions per microsecond, a wave-wide scan of 64 records is ~130.  So the walk FORECASTS where the chain will change
// binade, fetches ahead — every load of a round in flight together, the work of a round shared by the four waves — and has
// everything around a forecast crossing composed into single records beforehand.  Per 512 blocks (2 M source points):
//   1  the blocks' sums (quarter sums: k_chain_compose) and records.  The forecast of the chain at every block's start = the exact
//      value so far + the prefix of the blocks' sums (the records' own advances, their rounding drift included: k_chain_segrecs);
//      of each block record only the function for the forecast's binade is kept (S.ones); the first CH_PRE_BLKS blocks inside which
//      the forecast comes within CH_EPS of a power of two (or of zero) are the "fetched" blocks;
//   2  their segment records and sums: the same forecast by segments; the runs of segments between two forecast crossings composed
//      into one record each, for the binade the forecast has there (S.piece: one segmented scan per block);
//   3  the addends and the records of the first CH_PRE_SEGS segments the forecasts point at (S.xs, S.fseg).
// The walk then: a wave-wide scan over 64 block functions at a time; a fetched block piece by piece and crossing by crossing, one
// record each; a segment whose record does not hold the value is added up addend by addend (the reference's own operations).
// Whatever was not forecast (off by more than CH_EPS, more crossings than fit) is fetched when the walk gets there and scanned.
#define CH_PRE_BLKS 12
#define CH_PRE_SEGS 32
#define CH_SUPER 8                 // chunks of 64 blocks per round of forecasts
#define CH_EPS ( 1.0f / 2048.0f )
#define CH_CHK_MAX 4096            // (RS_HIP_CHAIN_DEBUG) steps of a walk logged for the self-check
#ifndef CH_BUDGET
#define CH_BUDGET 384              // segments a walk may add up addend by addend before it gives the problem up (ChainBufs::failed)
#endif
#define CH_PIECES 16               // pieces of a fetched block
#define CH_PIECE_BIG ( 1 << 27 )
struct ChainPiece { int es, lo, hi, D; };      // exponent | sign << 8 it is made for; M -> M + D [+ tau: ptau / bptau] for lo <= M <= hi
struct ChainOne { int es, lo, hi, Dt; };       // a block record's function for one binade: Dt = D * 16 + tau
// One step of the walk, made ahead: the record of a run of blocks / of a run of segments / of a segment the forecast has a crossing
// in, for the binade the forecast has there (tp: ChainWalkLds::ptau's form) — and what to fall back on when it does not hold the value:
// kind = type | chunk << 2 | block in chunk << 5 | from (or the segment) << 11 | to << 17 | slot in S.xs << 24 (63: not fetched)
enum { CH_IT_BLOCKS = 0, CH_IT_SEGS = 1, CH_IT_SEG = 2, CH_IT_BLOCK = 3 };       // a run of blocks, a run of segments, one segment, one whole block by its segments
struct ChainItem { int es, lo, hi, D, tp, kind; };
#ifndef CH_ITEMS
#define CH_ITEMS 448                // (12 fetched blocks of at most 15 crossing segments and the 16 runs around them, the runs of blocks between)
#endif
__device__ __forceinline__ int chain_item_kind( int type, int c, int at, int from, int to, int slot ) { return type | ( c << 2 ) | ( at << 5 ) | ( from << 11 ) | ( to << 17 ) | ( slot << 24 ); }
struct ChainWalkLds
{
  ChainOne ones[CH_SUPER * WAVE];               // the block records' functions for the forecast binade
  ChainPiece bpiece[CH_SUPER * WAVE];           // [c * 64 + l]: blocks (the forecast's last crossing block before l, l] of chunk c as one record
  int bptau[CH_SUPER][CH_PIECES];               // per piece: tau[0] | tau[1] << 4 | max tau << 8
  float bst[CH_SUPER * WAVE];                   // the forecast at the blocks' starts
  ChainPiece piece[CH_PRE_BLKS][CH_SEG];        // the same by segments inside fetched block k
  int ptau[CH_PRE_BLKS][CH_PIECES];
  ChainRec fseg[CH_PRE_SEGS];                   // the records of the segments the forecasts point at ...
  float xs[CH_PRE_SEGS + 1][CH_SEG];            // ... and their addends (the last row: those of a segment fetched on the way)
  unsigned long long stat[WAVES_PER_BLOCK][3], flag[CH_PRE_BLKS], fmask[CH_SUPER];
  int pblk[CH_PRE_BLKS], at_seg[CH_PRE_SEGS], mode[CH_PRE_BLKS], bmode[CH_SUPER];
  float pst[CH_PRE_BLKS], tot[CH_SUPER], s0;
  int round_end;                                // the first block this round does not cover
  ChainItem items[CH_ITEMS];                    // the walk's steps, in order
  int ctot[CH_SUPER], kbase[CH_PRE_BLKS];       // items per chunk; a fetched block's first item
};
// does a chain that goes from a to b (forecasts) change binade on the way, give or take a relative eps?
__device__ __forceinline__ bool chain_crosses( float a, float b, float eps )
{
  if( a == 0.0f && b == 0.0f ) return false;             // (a chain that has not left zero yet)
  if( !( a * b > 0.0f ) ) return true;                   // zero, a sign change, NaN
  const float lo = fminf( fabsf( a ), fabsf( b ) ) * ( 1.0f - eps ), hi = fmaxf( fabsf( a ), fabsf( b ) ) * ( 1.0f + eps );
  return ( __float_as_uint( lo ) >> 23 ) != ( __float_as_uint( hi ) >> 23 );
}
// does the record's function hold a forecast value, give or take CH_EPS of it?  (f: the record for the forecast's binade)

__device__ __forceinline__ bool chain_fits_forecast( const ChainFn& f, uint32_t vb )
{
  const int M = (int)( vb & 0x7fffffu ) | CH_M_LO, marg = (int)( CH_EPS * 8388608.0f );
  return f.lo <= f.hi && M - marg >= f.lo && M + marg + max( f.tau & 3, f.tau >> 2 ) <= f.hi;
}
// (a forecast needs four digits, not sixteen: fp32 prefix sums, six DPP adds each)
__device__ __forceinline__ float wave_scan_f32( float v ) { RS_DPP_PREFIX( "v_add_f32_dpp", v ); return v; }
__device__ __forceinline__ float rl( float v, int lane ) { return __int_as_float( __builtin_amdgcn_readlane( __float_as_int( v ), lane ) ); }
__device__ __forceinline__ unsigned long long below( int bit ) { return ( 1ull << bit ) - 1ull; }
// the lowest n set bits of m
__device__ __forceinline__ unsigned long long lowest_bits( unsigned long long m, int n )
{
  for( int c = __builtin_popcountll( m ); c > n && m != 0ull; --c ) m &= ~( 1ull << ( 63 - __builtin_clzll( m ) ) );
  return n > 0 ? m : 0ull;
}
// (field by field: a conditional copy of the whole struct is a memcpy through private memory, which then stays in scratch)
__device__ __forceinline__ void chain_rec_copy( ChainRec& d, const ChainRec& r )
{
  d.e_sign = r.e_sign;
#pragma unroll
  for( int c = 0; c < 3; ++c ) { d.lo[c] = r.lo[c]; d.hi[c] = r.hi[c]; d.D[c] = r.D[c]; }
}
__device__ __forceinline__ ChainFn chain_one_fn( const ChainOne& p, uint32_t vb )
{
  ChainFn f = chain_never();
  if( p.es == (int)( vb >> 23 ) ) { f.lo = p.lo; f.hi = p.hi; f.D = p.Dt >> 4; f.tau = p.Dt & 15; }
  return f;
}
// The runs of records between the forecast's crossings (the set bits of m; f0 there: the identity), each composed into ONE record
// for the binade `es` the forecast has there: one segmented scan makes them all — the exclusive prefix of D restarts after every
// crossing, the prefix max / min carry the piece number in the high bits — and lane l ends up with the record of (the last
// crossing before l, l].  The ties inside, piece by piece, as chain_compose_block.  (m has fewer than CH_PIECES bits.)
__device__ __forceinline__ void chain_pieces( const ChainFn& f0, int es, unsigned long long m, int lane, ChainPiece* out, int* ptau )
{
  const unsigned long long before = m & below( lane );
  const int pid = __builtin_popcountll( before );
  const int incD = (int)wave_scan( (uint32_t)f0.D, lane ), ex = incD - f0.D;
  const int first = before != 0ull ? 64 - __builtin_clzll( before ) : 0;       // my piece starts after the last crossing before me
  const int exs = ex - __shfl( ex, first );
  const bool wild = abs( exs ) > ( 1 << 25 );                                   // (no valid run adds that much inside one binade; keeps the sums below in range)
  int a = ( wild ? CH_M_HI : f0.lo - exs ) + pid * CH_PIECE_BIG, b = ( wild ? CH_M_LO : f0.hi - exs ) - pid * CH_PIECE_BIG;
  RS_DPP_PREFIX( "v_max_i32_dpp", a );
  RS_DPP_PREFIX( "v_min_i32_dpp", b );
  ChainPiece pc;
  pc.es = es; pc.lo = max( a - pid * CH_PIECE_BIG, CH_M_LO ); pc.hi = min( b + pid * CH_PIECE_BIG, CH_M_HI ); pc.D = exs + f0.D;
  out[lane] = pc;
  if( lane < CH_PIECES ) ptau[lane] = 0;
  wave_lds_fence();
  // (taus that outgrow their four bits: a 'max tau' no interval has room for — the piece is then never taken whole.  Clamping them
  //  instead was wrong by as many grid steps as were cut off: a run of 37 blocks, nine of them with ties, 4 ulps.)
  auto pack = []( int t0, int t1, int tmax ) -> int { return ( t0 > 15 || t1 > 15 ) ? ( 0x7fffff << 8 ) : ( t0 | ( t1 << 4 ) | ( tmax << 8 ) ); };
  int cur = -1, t0 = 0, t1 = 0, tmax = 0;
  for( unsigned long long tm = RS_BALLOT( f0.tau != 0 ); tm != 0ull; tm &= tm - 1ull )
  {
    const int kk = __builtin_ctzll( tm );
    const int pk = __builtin_amdgcn_readlane( pid, kk ), exk = __builtin_amdgcn_readlane( exs, kk ), tk = __builtin_amdgcn_readlane( f0.tau, kk );
    if( pk != cur ) { if( cur >= 0 && lane == 0 ) ptau[cur] = pack( t0, t1, tmax ); cur = pk; t0 = 0; t1 = 0; tmax = 0; }
    t0 += chain_tau( tk, exk + t0 ); t1 += chain_tau( tk, 1 + exk + t1 ); tmax += max( tk & 3, tk >> 2 );
  }
  if( cur >= 0 && lane == 0 ) ptau[cur] = pack( t0, t1, tmax );
}
// (every round's loads are unconditional, from clamped indices, masked afterwards: a load inside a branch is waited for there,
//  one round trip after the other)
__device__ __forceinline__ void chain_walk_row( const IcpLaunch& L, const ChainBufs& B, int prob, int row, ChainWalkLds& S )
{
  // (the wave's number through readfirstlane: the compiler then KNOWS that "wave 0 only" is uniform control flow — otherwise every
  //  loop of the walk is compiled as divergent, its counters in vector registers and an exec-mask dance around every branch)
  const int lane = threadIdx.x & ( WAVE - 1 ), wib = uni( (int)threadIdx.x / WAVE );
  const bool walker = wib == 0;
  const unsigned long long t_start = B.dbg ? wall_clock64() : 0ull;
  const ChainRec* blks = B.blk + ( (size_t)prob * CH_ROWS + row ) * B.n_blk;
  const ChainRec* segs = B.seg + ( (size_t)prob * CH_ROWS + row ) * B.n_seg;
  const double* ssum = B.segsum + ( (size_t)prob * CH_ROWS + row ) * B.n_seg;
  const double* qsum = B.blksum + ( (size_t)prob * CH_ROWS + row ) * ( B.n_blk * CH_QUARTERS );
  const float* Rf = reinterpret_cast<const float*>( L.rec + (size_t)prob * L.src.n * REC_F4 );
  const int comp = row == 0 ? 3 : ( row <= 3 ? row - 1 : row );      // a record's words: p.xyz at 0..2, dist² at 3, q.xyz at 4..6, dot at 7
  int* dbg = ( B.dbg && threadIdx.x == 0 ) ? B.dbg + ( (size_t)prob * CH_ROWS + row ) * ( 4 + 64 * 8 ) : nullptr;
  auto stamp = [&]( int k ) { if( dbg ) dbg[4 + 63 * 8 + k] = (int)( wall_clock64() - t_start ); };
  constexpr int CHUNKS_PER_WAVE = CH_SUPER / WAVES_PER_BLOCK, BLKS_PER_WAVE = CH_PRE_BLKS / WAVES_PER_BLOCK, SEGS_PER_WAVE = CH_PRE_SEGS / WAVES_PER_BLOCK;
  static_assert( CH_SUPER % WAVES_PER_BLOCK == 0 && CH_PRE_BLKS % WAVES_PER_BLOCK == 0 && CH_PRE_SEGS % WAVES_PER_BLOCK == 0, "the rounds' work is dealt to the waves" );
  static_assert( CH_PRE_BLKS <= WAVE && CH_SUPER <= WAVE, "one lane per fetched block / per chunk" );

  float bsl[CHUNKS_PER_WAVE]; ChainRec rc[CHUNKS_PER_WAVE];
  auto round1_loads = [&]( int B0 )
  {
#pragma unroll
    for( int i = 0; i < CHUNKS_PER_WAVE; ++i )
    {
      const int b = B0 + ( wib + i * WAVES_PER_BLOCK ) * WAVE + lane, bc = min( b, B.n_blk - 1 );
      const double2* q = reinterpret_cast<const double2*>( qsum + (size_t)bc * CH_QUARTERS );
      const double2 u = q[0], v = q[1];
      chain_rec_copy( rc[i], blks[bc] );
      bsl[i] = b < B.n_blk ? (float)( ( u.x + u.y ) + ( v.x + v.y ) ) : 0.0f;
      if( b >= B.n_blk ) rc[i].e_sign = -1;
    }
  };
  round1_loads( 0 );
  const float sd = chain_stats( L, prob, S.stat, nullptr );                  // (its loads go out with round 1's)
  ChainPar P; P.use_sd = sd > 0.000001; P.cut = 2.5f * sd; P.max_dist = L.radius;
  stamp( 0 );

  auto addend_of = [&]( float d2, float dt, float cv, bool inside ) -> float        // chain_addends, this chain's
  {
    const bool mt = d2 >= 0.0f && inside;
    float w = 0.0f;
    if( mt ) { w = ( 1.0f - __fdiv_rn( d2, P.max_dist ) ) * dt; if( P.use_sd && d2 > P.cut ) w = 0.0f; }
    return row == 0 ? w : ( mt ? cv * w : 0.0f );
  };

  float s = 0.0f;
  int resolved = 0, stuck = 0, steps = 0, hits = 0, piece_steps = 0, scans = 0, n_chk = 0;
  int* chk = B.chk ? B.chk + ( (size_t)prob * CH_ROWS + row ) * ( 4 + 3 * CH_CHK_MAX ) : nullptr;
  // advance over the records held by the lanes [from, count): as far as the value fits; returns the first lane that does not (count: all done)
  // (select( value bits ): the lane's record as a function for that value's binade)
  auto advance = [&]( auto&& select, int from, int count ) -> int
  {
    const uint32_t sb = (uint32_t)uni( __float_as_int( s ) );             // (uniform, and the compiler is told so: everything derived from it is scalar work)
    const int M = (int)( sb & 0x7fffffu ) | CH_M_LO;
    const bool mine_in = lane >= from && lane < count;
    const ChainFn f0 = !mine_in ? chain_identity() : select( sb );
    const ChainFn f = chain_prefix( f0, lane );
    ++scans;
    // records with ties inside (ChainFn): each adds the tau its own start's parity picks — until those are known the most they can add
    const unsigned long long tl = RS_BALLOT( f0.tau != 0 );
    int taumax = 0;
    for( unsigned long long t = tl; t != 0ull; t &= t - 1ull ) { const int tk = __builtin_amdgcn_readlane( f0.tau, __builtin_ctzll( t ) ); taumax += max( tk & 3, tk >> 2 ); }
    const bool fits = f.lo <= f.hi && M >= f.lo && M + taumax <= f.hi;
    const unsigned long long good = RS_BALLOT( fits );
    const int stop = good == ~0ull ? WAVE : __builtin_ctzll( ~good );       // the fitting lanes are a prefix: the intervals only shrink
    const int last = min( stop, count ) - 1;
    if( last >= from )
    {
      int extra = 0;
      for( unsigned long long t = tl & ( last >= WAVE - 1 ? ~0ull : below( last + 1 ) ); t != 0ull; t &= t - 1ull )
      {
        const int k = __builtin_ctzll( t );
        const int exk = __builtin_amdgcn_readlane( f.D, k ) - __builtin_amdgcn_readlane( f0.D, k );
        extra += chain_tau( __builtin_amdgcn_readlane( f0.tau, k ), M + exk + extra );
      }
      const int D = __builtin_amdgcn_readlane( f.D, last ) + extra;
      s = __uint_as_float( ( sb & 0xff800000u ) | ( (uint32_t)( M + D ) & 0x7fffffu ) );
    }
    return min( stop, count );
  };
  long long walk_cycles = 0;
  // A round covers up to 512 blocks — and ends early at the first crossing block its fetch slots do not hold (S.round_end):
  // the next round starts there, from the exact value, with fresh slots.
  for( int B0 = 0; B0 < B.n_blk; )
  {
    const int n_chunks = min( CH_SUPER, ( B.n_blk - B0 + WAVE - 1 ) / WAVE );
    if( B0 > 0 ) round1_loads( B0 );
    if( threadIdx.x == 0 ) S.round_end = min( B0 + CH_SUPER * WAVE, B.n_blk );
    // ---- the forecasts at the blocks' starts: the chunks' totals first ...
    float incl[CHUNKS_PER_WAVE];
#pragma unroll
    for( int i = 0; i < CHUNKS_PER_WAVE; ++i ) { incl[i] = wave_scan_f32( bsl[i] ); if( lane == WAVE - 1 ) S.tot[wib + i * WAVES_PER_BLOCK] = incl[i]; }
    if( threadIdx.x == 0 ) S.s0 = s;
    __syncthreads();
    unsigned long long fm_mine[CHUNKS_PER_WAVE]; float st_mine[CHUNKS_PER_WAVE];
    float tot_before;                      // lane c: what the chunks before c add (one read, one scan: a loop of dependent LDS reads costs ~100 cycles a turn)
    { const float t = lane < CH_SUPER ? S.tot[lane] : 0.0f; tot_before = wave_scan_f32( t ) - t; }
#pragma unroll
    for( int i = 0; i < CHUNKS_PER_WAVE; ++i )
    {
      const int c = wib + i * WAVES_PER_BLOCK;
      const float base = S.s0 + rl( tot_before, c );
      const float st = base + ( incl[i] - bsl[i] ), en = base + incl[i];
      const int b = B0 + c * WAVE + lane;
      // ... of a block's record the function for the binade the forecast has at its start (if the value gets there in another: by its segments)
      const uint32_t vb = __float_as_uint( st );
      const ChainFn f = rc[i].e_sign == -1 ? chain_never() : chain_fn_for( rc[i], vb );
      ChainOne one; one.es = (int)( vb >> 23 ); one.lo = f.lo; one.hi = f.hi; one.Dt = f.D * 16 + f.tau;
      S.ones[c * WAVE + lane] = one; S.bst[c * WAVE + lane] = st;
      // a crossing block: the forecast changes binade between its ends — or its record does not hold the forecast (the chain leaves the
      // binade INSIDE the block and is back at its end: a sum that hovers at a power of two does that block after block, and each such
      // block, not forecast, cost a scan of its segments and a round trip per segment that did not fit)
      const unsigned long long m = RS_BALLOT( b < B.n_blk && ( b == 0 || chain_crosses( st, en, CH_EPS ) || !chain_fits_forecast( f, vb ) ) );
      fm_mine[i] = m; st_mine[i] = st;
      // ... and the runs of blocks between the forecast's crossing blocks as one record each
      // (a round ends at its (CH_PRE_BLKS + 1)-th crossing block at the latest: the pieces beyond a chunk's first CH_PIECES - 1 crossing
      //  blocks are never walked — a chunk with more of them used to be taken "whole, by scans": a round trip per segment that did not fit)
      static_assert( CH_PRE_BLKS + 1 < CH_PIECES, "the pieces cover every block a round can reach" );
      const unsigned long long mp = lowest_bits( m, CH_PIECES - 1 );
      if( lane == 0 ) { S.fmask[c] = m; S.bmode[c] = 1; }
      chain_pieces( ( b >= B.n_blk || ( ( mp >> lane ) & 1ull ) ) ? chain_identity() : f, (int)( vb >> 23 ), mp, lane, &S.bpiece[c * WAVE], S.bptau[c] );
    }
    __syncthreads();
    // ---- the blocks to fetch: the first CH_PRE_BLKS of those, in order — every lane knows its block's rank
    int flagged_before, flagged_total;     // lane c: the crossing blocks in the chunks before c
    {
      const uint32_t cnt = lane < CH_SUPER ? (uint32_t)__builtin_popcountll( S.fmask[lane] ) : 0u;
      const uint32_t inc = wave_scan( cnt, lane );
      flagged_before = (int)( inc - cnt ); flagged_total = __builtin_amdgcn_readlane( (int)inc, WAVE - 1 );
#pragma unroll
      for( int i = 0; i < CHUNKS_PER_WAVE; ++i )
      {
        const int c = wib + i * WAVES_PER_BLOCK;
        const int rank = __builtin_amdgcn_readlane( flagged_before, c ) + __builtin_popcountll( fm_mine[i] & below( lane ) );
        if( ( ( fm_mine[i] >> lane ) & 1ull ) && rank < CH_PRE_BLKS ) { S.pblk[rank] = B0 + c * WAVE + lane; S.pst[rank] = st_mine[i]; }
        if( ( ( fm_mine[i] >> lane ) & 1ull ) && rank == CH_PRE_BLKS ) S.round_end = B0 + c * WAVE + lane;      // (the first one without a slot)
      }
      if( walker && lane >= flagged_total && lane < CH_PRE_BLKS ) { S.pblk[lane] = -1; S.pst[lane] = 0.0f; }
    }
    __syncthreads();
    const int round_end = S.round_end;
    stamp( 1 );
    // ---- round 2: those blocks' segment records and sums; the forecast by segments; the pieces
    {
      ChainRec got[BLKS_PER_WAVE]; float sv[BLKS_PER_WAVE]; int pb[BLKS_PER_WAVE];
#pragma unroll
      for( int i = 0; i < BLKS_PER_WAVE; ++i )
      {
        pb[i] = uni( S.pblk[wib * BLKS_PER_WAVE + i] );
        const int sg = max( pb[i], 0 ) * CH_BLK + lane, sgc = min( sg, B.n_seg - 1 );
        chain_rec_copy( got[i], segs[sgc] ); sv[i] = (float)ssum[sgc];
        if( pb[i] < 0 || sg >= B.n_seg ) { got[i].e_sign = -1; sv[i] = 0.0f; }
      }
#pragma unroll
      for( int i = 0; i < BLKS_PER_WAVE; ++i )
      {
        const int k = wib * BLKS_PER_WAVE + i;
        const float incs = wave_scan_f32( sv[i] ), st0 = S.pst[k];
        const float vst = st0 + ( incs - sv[i] );
        const bool in = pb[i] >= 0 && pb[i] * CH_BLK + lane < B.n_seg;
        const uint32_t vb = __float_as_uint( vst );
        const ChainFn fseg = in ? chain_fn_for( got[i], vb ) : chain_identity();
        const unsigned long long m = RS_BALLOT( in && ( chain_crosses( vst, st0 + incs, CH_EPS ) || !chain_fits_forecast( fseg, vb ) ) );      // (as for the blocks)
        const bool pieces = pb[i] >= 0 && __builtin_popcountll( m ) < CH_PIECES;
        // (a block of 16 crossing segments or more — a chain's first block when the coordinates straddle the origin — is walked by scans
        //  of its records, its segments' addends fetched on the way: it takes none of the round's fetch slots — it used to take them all, and
        //  every later crossing segment of the round then went without.  Loading such a block's 4 096 addends in one go was built too:
        //  the centred bench step 3.84 instead of 3.89 ms — and 15 us per iteration MORE on scans in one octant, which never run that
        //  code: the walk is one wave's instruction stream, and it got longer)
        if( lane == 0 ) { S.flag[k] = pieces ? m : 0ull; S.mode[k] = pieces ? 1 : 0; }
        if( pieces )
        {
          const ChainFn f0 = ( !in || ( ( m >> lane ) & 1ull ) ) ? chain_identity() : fseg;
          chain_pieces( f0, (int)( vb >> 23 ), m, lane, S.piece[k], S.ptau[k] );
        }
      }
    }
    __syncthreads();                                                      // (S.flag, S.piece)
    stamp( 2 );
    // ---- the first CH_PRE_SEGS of the segments the forecasts point at, in order: lane k (of every wave) works out block k's share
    unsigned long long l_segs = 0ull;     // lane k: the segments of fetched block k whose addends and records are in S.xs / S.fseg, from slot l_sbase on
    int l_sbase = 0;
    {
      const unsigned long long mine = lane < CH_PRE_BLKS ? S.flag[lane] : 0ull;
      const uint32_t cnt = (uint32_t)__builtin_popcountll( mine );
      l_sbase = (int)( wave_scan( cnt, lane ) - cnt );
      l_segs = lowest_bits( mine, CH_PRE_SEGS - l_sbase );
      if( walker )
      {
        const int pbk = lane < CH_PRE_BLKS ? S.pblk[lane] : 0;
        int slot = l_sbase;
        for( unsigned long long m = l_segs; m != 0ull; m &= m - 1ull ) S.at_seg[slot++] = pbk * CH_BLK + __builtin_ctzll( m );
        const int total = __builtin_amdgcn_readlane( l_sbase + (int)__builtin_popcountll( l_segs ), CH_PRE_BLKS - 1 );
        if( lane >= total && lane < CH_PRE_SEGS ) S.at_seg[lane] = 0;       // (none: segment 0's, not used)
      }
    }
    __syncthreads();                                                      // (S.at_seg)
    // ---- round 3: their addends and records (the loads now; what they bring is put away after the items below, which do not need it)
    float xd[SEGS_PER_WAVE], xw[SEGS_PER_WAVE], xc[SEGS_PER_WAVE]; int at3[SEGS_PER_WAVE], rw[SEGS_PER_WAVE];
    constexpr int REC_WORDS = sizeof( ChainRec ) / 4;
#pragma unroll
    for( int i = 0; i < SEGS_PER_WAVE; ++i )
    {
      const int sgm = uni( S.at_seg[wib * SEGS_PER_WAVE + i] );
      at3[i] = sgm * CH_SEG + lane;
      const float* rp = Rf + (size_t)min( at3[i], L.src.n - 1 ) * ( REC_F4 * 4 );
      xd[i] = rp[3]; xw[i] = rp[7]; xc[i] = rp[comp];
      rw[i] = reinterpret_cast<const int*>( segs + min( sgm, B.n_seg - 1 ) )[min( lane, REC_WORDS - 1 )];
    }
    // ---- the walk's steps, in order, one item each (ChainItem): how many per chunk ...
    auto lanes_below = [&]( int n ) -> unsigned long long { return n >= WAVE ? ~0ull : below( n ); };
    int it_cnt[CHUNKS_PER_WAVE], it_off[CHUNKS_PER_WAVE], it_k[CHUNKS_PER_WAVE];
#pragma unroll
    for( int i = 0; i < CHUNKS_PER_WAVE; ++i )
    {
      const int c = wib + i * WAVES_PER_BLOCK, nb = max( 0, min( WAVE, round_end - ( B0 + c * WAVE ) ) );
      const unsigned long long fm = fm_mine[i] & lanes_below( nb );
      const int rank = __builtin_amdgcn_readlane( flagged_before, c ) + __builtin_popcountll( fm & below( lane ) );
      const bool flagged = ( fm >> lane ) & 1ull, mine_in = lane < nb;
      int cnt = 0; it_k[i] = -1;
      if( c < n_chunks && mine_in )
      {
        if( !S.bmode[c] ) cnt = lane == 0 ? 1 : 0;                           // (too many crossings for pieces: the whole chunk by scans)
        else if( flagged )
        {
          cnt = 1;
          if( rank < CH_PRE_BLKS && S.mode[rank] )
          {
            // a fetched block: its crossing segments and the runs between them
            const int ns = min( CH_BLK, B.n_seg - ( B0 + c * WAVE + lane ) * CH_BLK );
            const unsigned long long m = S.flag[rank], in = lanes_below( ns );
            cnt = __builtin_popcountll( m & in ) + __builtin_popcountll( ~m & ( ( m << 1 ) | 1ull ) & in );
            it_k[i] = rank;
          }
        }
        else cnt = ( lane == nb - 1 || ( ( fm >> ( lane + 1 ) ) & 1ull ) ) ? 1 : 0;     // the last block of a run
      }
      it_cnt[i] = cnt;
      const int inc = (int)wave_scan( (uint32_t)cnt, lane );
      it_off[i] = inc - cnt;
      if( lane == WAVE - 1 ) S.ctot[c] = inc;
    }
    __syncthreads();
    // ... the runs of blocks and the blocks taken whole; where a fetched block's items start
    int n_items, items_before;             // lane c: the items of the chunks before c
    {
      const uint32_t cnt = lane < CH_SUPER ? (uint32_t)S.ctot[lane] : 0u;
      const uint32_t inc = wave_scan( cnt, lane );
      items_before = (int)( inc - cnt ); n_items = __builtin_amdgcn_readlane( (int)inc, WAVE - 1 );
    }
#pragma unroll
    for( int i = 0; i < CHUNKS_PER_WAVE; ++i )
    {
      const int c = wib + i * WAVES_PER_BLOCK, nb = max( 0, min( WAVE, round_end - ( B0 + c * WAVE ) ) );
      const int at = __builtin_amdgcn_readlane( items_before, c ) + it_off[i];
      if( it_cnt[i] > 0 && at < CH_ITEMS )
      {
        const unsigned long long fm = fm_mine[i] & lanes_below( nb );
        ChainItem it; it.es = -1; it.lo = CH_M_HI; it.hi = CH_M_LO; it.D = 0; it.tp = 0;
        if( !S.bmode[c] ) { it.kind = chain_item_kind( CH_IT_BLOCKS, c, 0, 0, nb, 63 ); S.items[at] = it; }
        else if( ( fm >> lane ) & 1ull )
        {
          if( it_k[i] >= 0 ) S.kbase[it_k[i]] = at;
          else { it.kind = chain_item_kind( CH_IT_BLOCK, c, lane, 0, 0, 63 ); S.items[at] = it; }
        }
        else
        {
          const unsigned long long before = fm & below( lane );
          const int from = before != 0ull ? 64 - __builtin_clzll( before ) : 0;
          const ChainPiece pc = S.bpiece[c * WAVE + lane];
          it.es = pc.es; it.lo = pc.lo; it.hi = pc.hi; it.D = pc.D; it.tp = S.bptau[c][__builtin_popcountll( before )];
          it.kind = chain_item_kind( CH_IT_BLOCKS, c, 0, from, lane + 1, 63 );
          S.items[at] = it;
        }
      }
    }
#pragma unroll
    for( int i = 0; i < SEGS_PER_WAVE; ++i )
    {
      S.xs[wib * SEGS_PER_WAVE + i][lane] = addend_of( xd[i], xw[i], xc[i], at3[i] < L.src.n );
      if( lane < REC_WORDS ) reinterpret_cast<int*>( &S.fseg[wib * SEGS_PER_WAVE + i] )[lane] = rw[i];
    }
    __syncthreads();                                                      // (S.xs, S.fseg, S.kbase)
    // ... and the fetched blocks' own
#pragma unroll
    for( int i = 0; i < BLKS_PER_WAVE; ++i )
    {
      const int k = wib * BLKS_PER_WAVE + i, pbk = S.pblk[k];
      if( pbk < 0 || !S.mode[k] ) continue;
      const int ns = min( CH_BLK, B.n_seg - pbk * CH_BLK ), c = ( pbk - B0 ) / WAVE, at_blk = ( pbk - B0 ) % WAVE;
      const unsigned long long m = S.flag[k];
      const bool flagged = ( m >> lane ) & 1ull;
      const bool run_end = !flagged && ( lane == ns - 1 || ( ( m >> ( lane + 1 ) ) & 1ull ) );
      const unsigned long long ends = RS_BALLOT( lane < ns && ( flagged || run_end ) );
      const int at = S.kbase[k] + __builtin_popcountll( ends & below( lane ) );
      if( ( ( ends >> lane ) & 1ull ) && at < CH_ITEMS )
      {
        const unsigned long long have = rl( l_segs, k );
        const int sbase = __builtin_amdgcn_readlane( l_sbase, k );
        const ChainPiece pc = S.piece[k][lane];
        ChainItem it; it.es = pc.es; it.lo = CH_M_HI; it.hi = CH_M_LO; it.D = 0; it.tp = 0;
        if( flagged )
        {
          int slot = 63;
          if( ( have >> lane ) & 1ull )
          {
            slot = sbase + __builtin_popcountll( have & below( lane ) );
            ChainRec r; chain_rec_copy( r, S.fseg[slot] );
            const ChainFn f = chain_fn_for( r, (uint32_t)pc.es << 23 );
            it.lo = f.lo; it.hi = f.hi; it.D = f.D; it.tp = chain_tau( f.tau, 0 ) | ( chain_tau( f.tau, 1 ) << 4 ) | ( max( f.tau & 3, f.tau >> 2 ) << 8 );
          }
          it.kind = chain_item_kind( CH_IT_SEG, c, at_blk, lane, lane + 1, slot );
        }
        else
        {
          const unsigned long long before = m & below( lane );
          const int from = before != 0ull ? 64 - __builtin_clzll( before ) : 0;
          it.lo = pc.lo; it.hi = pc.hi; it.D = pc.D; it.tp = S.ptau[k][__builtin_popcountll( before )];
          it.kind = chain_item_kind( CH_IT_SEGS, c, at_blk, from, lane + 1, 63 );
        }
        S.items[at] = it;
      }
    }
    __syncthreads();
    if( B0 == 0 && dbg ) dbg[1] = (int)( wall_clock64() - t_start );

    // ---- the walk: item after item, 64 of them in wave 0's registers at a time; an item that does not hold the value (the forecast
    // was off, or it is the segment where the chain changes binade) falls back on its own range — addend by addend for a segment,
    // wave-wide scans of the records for a run
    if( walker && !( stuck & 2 ) )
    {
      const long long c_walk = dbg ? clock64() : 0;
      // segment g: its 64 addends one after the other, in fp32 — the reference's own operations.  From LDS, four at a time, every
      // lane the same address: 16 reads + 64 adds (by readlane from a register: 64 + 64).
      auto one_by_one = [&]( int g, int slot )
      {
        if( dbg && resolved < 62 )
        {
          int* d = dbg + 4 + resolved * 8;
          d[0] = g; d[1] = __float_as_int( s ); d[2] = slot < CH_PRE_SEGS ? 3 : 0; d[3] = 0; d[7] = (int)( wall_clock64() - t_start );
        }
        hits += slot < CH_PRE_SEGS ? 1 : 0;
        if( slot >= CH_PRE_SEGS )
        {
          const int i = g * CH_SEG + lane;
          const float* rp = Rf + (size_t)min( i, L.src.n - 1 ) * ( REC_F4 * 4 );
          slot = CH_PRE_SEGS;
          S.xs[slot][lane] = addend_of( rp[3], rp[7], rp[comp], i < L.src.n );
          wave_lds_fence();
        }
        const float4* xp = reinterpret_cast<const float4*>( &S.xs[slot][0] );
        float4 xa[4], xb[4];                                          // (two sets of 16 addends in turn: 32 registers)
        auto add4 = [&]( const float4* x ) {
#pragma unroll
          for( int j = 0; j < 4; ++j ) { s = s + x[j].x; s = s + x[j].y; s = s + x[j].z; s = s + x[j].w; } };
#pragma unroll
        for( int j = 0; j < 4; ++j ) { xa[j] = xp[j]; xb[j] = xp[4 + j]; }
        add4( xa );
#pragma unroll
        for( int j = 0; j < 4; ++j ) xa[j] = xp[8 + j];
        add4( xb );
#pragma unroll
        for( int j = 0; j < 4; ++j ) xb[j] = xp[12 + j];
        add4( xa );
        add4( xb );
        if( ++resolved > CH_BUDGET ) stuck |= 2;
      };
      // the segments [sat, to) of the block that starts at segment g0 by wave-wide scans of their records (fetched now)
      auto segs_by_scans = [&]( int g0, int sat, int to )
      {
        const int ns = min( CH_BLK, B.n_seg - g0 );
        ChainRec smine; smine.e_sign = -1;
        if( lane < ns ) chain_rec_copy( smine, segs[g0 + lane] );
        while( sat < to && !( stuck & 2 ) )
        {
          const int sat_was = sat;
          if( ++steps > 8 * B.n_seg + 4096 ) { stuck |= 2; break; }       // (a walk takes at most one step per block + two per segment: guards against a loop that does not end)
          sat = advance( [&]( uint32_t vb ) -> ChainFn { return smine.e_sign == -1 ? chain_never() : chain_fn_for( smine, vb ); }, sat, to );
          if( sat < sat_was ) { stuck |= 1; sat = sat_was; }              // (cannot happen: the lanes before `sat` hold the identity — guards the loop against a wrong scan)
          if( sat >= to ) break;
          one_by_one( g0 + sat, 63 );
          ++sat;
        }
      };
      // the blocks [at, to) of chunk c by wave-wide scans of their functions
      auto blocks_by_scans = [&]( int c, int at, int to )
      {
        const ChainOne mine = S.ones[c * WAVE + lane];
        while( at < to && !( stuck & 2 ) )
        {
          const int at_was = at;
          if( ++steps > 8 * B.n_seg + 4096 ) { stuck |= 2; break; }
          at = advance( [&]( uint32_t vb ) -> ChainFn { return chain_one_fn( mine, vb ); }, at, to );
          if( at < at_was ) { stuck |= 1; at = at_was; }
          if( at >= to ) break;
          const int g0 = ( B0 + c * WAVE + at ) * CH_BLK;
          segs_by_scans( g0, 0, min( CH_BLK, B.n_seg - g0 ) );
          ++at;
        }
      };
      for( int i0 = 0; i0 < min( n_items, CH_ITEMS ) && !( stuck & 2 ); i0 += WAVE )
      {
        const ChainItem mine = S.items[min( i0 + lane, CH_ITEMS - 1 )];
        const int n_here = min( WAVE, min( n_items, CH_ITEMS ) - i0 );
        for( int i = 0; i < n_here && !( stuck & 2 ); ++i )
        {
          const uint32_t sb = (uint32_t)uni( __float_as_int( s ) );
          const int M = (int)( sb & 0x7fffffu ) | CH_M_LO;
          const int es = __builtin_amdgcn_readlane( mine.es, i ), lo = __builtin_amdgcn_readlane( mine.lo, i ), hi = __builtin_amdgcn_readlane( mine.hi, i );
          const int D = __builtin_amdgcn_readlane( mine.D, i ), tp = __builtin_amdgcn_readlane( mine.tp, i ), kind = __builtin_amdgcn_readlane( mine.kind, i );
          const int type = kind & 3, c = ( kind >> 2 ) & 7, at = ( kind >> 5 ) & 63, from = ( kind >> 11 ) & 63, to = ( kind >> 17 ) & 127, slot = ( kind >> 24 ) & 63;
          const int g0 = ( B0 + c * WAVE + at ) * CH_BLK;
          const bool whole = es == (int)( sb >> 23 ) && lo <= hi && M >= lo && M + ( tp >> 8 ) <= hi;
          if( whole )
          {
            const float s_rec = __uint_as_float( ( sb & 0xff800000u ) | ( (uint32_t)( M + D + ( ( M & 1 ) ? ( tp >> 4 ) & 15 : tp & 15 ) ) & 0x7fffffu ) );
            if( chk )      // (RS_HIP_CHAIN_DEBUG: the same step by scans / addend by addend, from the same value)
            {
              if( type == CH_IT_SEG ) one_by_one( g0 + from, slot );
              else if( type == CH_IT_SEGS ) segs_by_scans( g0, from, to );
              else if( type == CH_IT_BLOCK ) segs_by_scans( g0, 0, min( CH_BLK, B.n_seg - g0 ) );
              else blocks_by_scans( c, from, to );
              s = __int_as_float( uni( __float_as_int( s ) ) );
              if( __float_as_int( s ) != __float_as_int( s_rec ) && dbg && dbg[4 + 63 * 8 + 4] == 0 )
              {
                dbg[4 + 63 * 8 + 4] = 1 + n_chk; dbg[4 + 63 * 8 + 5] = es; dbg[4 + 63 * 8 + 6] = lo; dbg[4 + 63 * 8 + 7] = hi;
                dbg[4 + 62 * 8 + 0] = D; dbg[4 + 62 * 8 + 1] = tp; dbg[4 + 62 * 8 + 2] = (int)sb; dbg[4 + 62 * 8 + 3] = __float_as_int( s ); dbg[4 + 62 * 8 + 4] = __float_as_int( s_rec ); dbg[4 + 62 * 8 + 5] = kind;
              }
            }
            s = s_rec;
            ++piece_steps;
          }
          else if( type == CH_IT_SEG ) one_by_one( g0 + from, slot );
          else if( type == CH_IT_SEGS ) segs_by_scans( g0, from, to );
          else if( type == CH_IT_BLOCK ) segs_by_scans( g0, 0, min( CH_BLK, B.n_seg - g0 ) );
          else blocks_by_scans( c, from, to );
          if( chk && n_chk < CH_CHK_MAX )       // (RS_HIP_CHAIN_DEBUG: where this step ends and with what — held against the plain sum below)
          {
            const int end_seg = type == CH_IT_BLOCKS ? min( ( B0 + c * WAVE + to ) * CH_BLK, B.n_seg ) : ( type == CH_IT_SEGS ? g0 + to : ( type == CH_IT_SEG ? g0 + from + 1 : min( g0 + CH_BLK, B.n_seg ) ) );
            s = __int_as_float( uni( __float_as_int( s ) ) );
            if( lane == 0 ) { chk[4 + 3 * n_chk] = end_seg; chk[4 + 3 * n_chk + 1] = __float_as_int( s ); chk[4 + 3 * n_chk + 2] = kind | ( whole ? 1 << 30 : 0 ); }
            ++n_chk;
          }
        }
      }
      if( n_items > CH_ITEMS ) stuck |= 2;                                   // (more steps than the list holds: hundreds of binade changes in one round)
      if( dbg ) { s = __int_as_float( uni( __float_as_int( s ) ) ); walk_cycles += clock64() - c_walk; }
    }
    __syncthreads();                                                      // (the next round overwrites what this walk read)
    B0 = round_end;
  }
  if( chk && walker )
  {
    // the plain sum, 64 addends at a time, compared with what the walk had where each of its steps ended
    float sp = 0.0f; int next = 0, bad = -1, bad_bits = 0, crossing = 0;
    for( int g = 0; g < B.n_seg; ++g )
    {
      const uint32_t sp_was = __float_as_uint( sp );
      const int i = g * CH_SEG + lane;
      const float* rp = Rf + (size_t)min( i, L.src.n - 1 ) * ( REC_F4 * 4 );
      const float xr = addend_of( rp[3], rp[7], rp[comp], i < L.src.n );
#pragma unroll
      for( int j = 0; j < CH_SEG; ++j ) sp = sp + __int_as_float( __builtin_amdgcn_readlane( __float_as_int( xr ), j ) );
      sp = __int_as_float( uni( __float_as_int( sp ) ) );
      crossing += ( sp_was >> 23 ) != ( __float_as_uint( sp ) >> 23 ) ? 1 : 0;      // (segments that END in another binade than they start in: a lower bound of those that change binade)
      while( next < n_chk && chk[4 + 3 * next] <= g + 1 )
      {
        if( chk[4 + 3 * next] == g + 1 && bad < 0 && chk[4 + 3 * next + 1] != __float_as_int( sp ) ) { bad = next; bad_bits = __float_as_int( sp ); }
        ++next;
      }
    }
    if( lane == 0 ) { chk[0] = n_chk; chk[1] = bad; chk[2] = bad < 0 ? crossing : bad_bits; chk[3] = __float_as_int( sp ); }
  }
  // A chain that wanders around zero — coordinates that straddle the origin, summed in an order that keeps cancelling — changes binade
  // not fifteen times but thousands of times, and every such segment is 64 dependent additions on this one wave: milliseconds.  The
  // walk gives up after CH_BUDGET of them (or when its list of steps overflows, or a loop guard fires): the problem is marked failed
  // and inactive — every later launch of the call is a no-op for it — and the host runs it again with the seven sums by pass 2 of
  // the replay (rs_hip_icp_align_batch), whose speculative segments do not mind.
  if( threadIdx.x == 0 && ( stuck & 2 ) && B.failed ) { B.failed[prob] = 1; L.active[prob] = 0; }
  if( threadIdx.x == 0 )
  {
    B.totals[( (size_t)prob * 3 + 1 ) * ICP_NMOM + row] = (double)s;
    if( B.resolved ) atomicAdd( B.resolved + prob, resolved );
    if( dbg ) { dbg[0] = resolved | ( stuck << 30 ) | ( scans << 16 ); dbg[2] = hits | ( piece_steps << 16 ); dbg[3] = (int)( wall_clock64() - t_start ); dbg[4 + 63 * 8 + 3] = (int)walk_cycles; }
  }
}

__global__ __launch_bounds__( BLOCK ) void k_chain_walk( IcpLaunch L, ChainBufs B )
{
  RS_CHAIN_SETPRIO();
  __shared__ ChainWalkLds S;
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  for( int rep = 0; rep < ( B.dbg ? B.dbg_reps : 1 ); ++rep ) { chain_walk_row( L, B, prob, blockIdx.x, S ); __syncthreads(); }
}
// The walks, the moments and the next iteration's guesses in ONE launch (none needs another's results): workgroups 0-6 walk a chain
// each with their first wave, the next 4 n_blk take a quarter block of the moments each, the last n_blk a block of the guesses (from
// the sums k_chain_segrecs and k_chain_compose have just left: read by the NEXT iteration's k_chain_segrecs).
__global__ __launch_bounds__( BLOCK ) void k_chain_walk_and_moments( IcpLaunch L, ChainBufs B )
{
  RS_CHAIN_SETPRIO();
  __shared__ union U { ChainWalkLds w; ChainMomLds m; __device__ U() {} } S;
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  if( blockIdx.x < CH_ROWS )
  {
    chain_walk_row( L, B, prob, blockIdx.x, S.w );
    return;
  }
  const int qb = (int)blockIdx.x - CH_ROWS;
  if( qb < B.n_blk * CH_QUARTERS ) chain_moments_block( L, B, prob, qb, S.m );
  else chain_guess_block( L, B, prob, qb - B.n_blk * CH_QUARTERS, WAVES_PER_BLOCK );
}

void launch_icp_chain_centroids( const IcpLaunch& L, const ChainBufs& B, hipStream_t st )
{
  const int n_tasks = ( B.n_seg + CHAIN_REC_TASK - 1 ) / CHAIN_REC_TASK;
  const dim3 rec_grid( ( n_tasks + WAVES_PER_BLOCK * CHAIN_REC_ROUNDS - 1 ) / ( WAVES_PER_BLOCK * CHAIN_REC_ROUNDS ), L.n_prob );
  if( B.refresh )
  {
    // the guesses anew: this iteration's fp64 sums first
    hipLaunchKernelGGL( k_chain_moments, dim3( B.n_blk * CH_QUARTERS, L.n_prob ), dim3( BLOCK ), 0, st, L, B );         // (L.n_mom_blocks == 4 B.n_blk)
    hipLaunchKernelGGL( k_chain_guess, dim3( B.n_blk, L.n_prob ), dim3( CH_ROWS * WAVE ), 0, st, L, B );
    hipLaunchKernelGGL( k_chain_segrecs, rec_grid, dim3( BLOCK ), 0, st, L, B );
    hipLaunchKernelGGL( k_chain_compose, dim3( B.n_blk, L.n_prob, CH_ROWS ), dim3( 3 * WAVE ), 0, st, L, B );
    hipLaunchKernelGGL( k_chain_walk, dim3( CH_ROWS, L.n_prob ), dim3( BLOCK ), 0, st, L, B );
  }
  else
  {
    hipLaunchKernelGGL( k_chain_segrecs, rec_grid, dim3( BLOCK ), 0, st, L, B );
    hipLaunchKernelGGL( k_chain_compose, dim3( B.n_blk, L.n_prob, CH_ROWS ), dim3( 3 * WAVE ), 0, st, L, B );
    hipLaunchKernelGGL( k_chain_walk_and_moments, dim3( CH_ROWS + B.n_blk * ( CH_QUARTERS + 1 ), L.n_prob ), dim3( BLOCK ), 0, st, L, B );
  }
  hipLaunchKernelGGL( k_icp_update_wide, dim3( ICP_NMOM, L.n_prob ), dim3( BLOCK ), 0, st, L, B.done );                // (centred on the chains' totals: L.exact_centroids)
}

// The same from the searches' records (L.rec: the grid chains' own inputs — what a scan the chains give up is run with): the fp64
// moments by the chains' moment kernel (n, mean, stddev into L.res), pass 2 of the replay reading the records, the chains' update.
void launch_icp_exact_centroids_from_records( const IcpLaunch& L, const ReplayBufs& B, const ChainBufs& C, hipStream_t st )
{
  hipLaunchKernelGGL( k_chain_moments, dim3( C.n_blk * CH_QUARTERS, L.n_prob ), dim3( BLOCK ), 0, st, L, C );          // (L.n_mom_blocks == 4 C.n_blk)
  launch_replay_pass<2>( L, B, st );
  hipLaunchKernelGGL( k_icp_update_wide, dim3( ICP_NMOM, L.n_prob ), dim3( BLOCK ), 0, st, L, C.done );
}
int replay_segments( int n_source ) { return ( n_source + REPLAY_SEG - 1 ) / REPLAY_SEG; }
int replay_superblocks( int n_source ) { return ( replay_segments( n_source ) + REPLAY_SUPER - 1 ) / REPLAY_SUPER; }
size_t replay_seg_bytes() { return sizeof( ReplaySeg ); }

void launch_icp_corr( const IcpLaunch& L, hipStream_t st )
{
  // queue_count is zero on entry: cleared once by the host, then by the workgroup that ends every iteration (icp_iteration_reset)
  // A launch of a few hundred tiles leaves every wave alone on its SIMD, i.e. latency-bound, and phase A's slowest tile
  // sets its time: such launches skip phase A and give every tile a workgroup straight away (coop_all).
  static_assert( HEAVY_SLOTS % ( 8 * PA_WAVES ) == 0, "the front slots must not shift the XCD class of the natural part" );
  dim3 grid( ( L.heavy_in ? HEAVY_SLOTS / PA_WAVES : 0 ) + 8 * icp_blocks_per_xcd( L.max_tiles ), L.n_prob );      // (max_tiles: the largest problem's)
  if( !L.coop_all )
  {
    if( L.warm && L.bounded_only ) hipLaunchKernelGGL( k_icp_corr<true>, grid, dim3( PA_WAVES * WAVE ), 0, st, L );
    else                           hipLaunchKernelGGL( k_icp_corr<false>, grid, dim3( PA_WAVES * WAVE ), 0, st, L );
  }
  // the queue length is only known on the device: a fixed grid strides over it
  int coop_blocks = L.max_tiles < 2048 ? L.max_tiles : 2048;
  const dim3 cgrid( coop_blocks > 0 ? coop_blocks : 1, L.n_prob );
  // a short queue is latency-bound by its heaviest tile: give every tile more waves
  if( L.coop_waves >= 8 ) hipLaunchKernelGGL( k_icp_corr_coop<8>, cgrid, dim3( 8 * WAVE ), 0, st, L );
  else                    hipLaunchKernelGGL( k_icp_corr_coop<COOP_WAVES>, cgrid, dim3( COOP_BLOCK ), 0, st, L );
}
void launch_icp_faithful( const IcpLaunch& L, hipStream_t st )
{
  hipLaunchKernelGGL( k_icp_faith_gather, dim3( ( L.max_n + BLOCK - 1 ) / BLOCK, L.n_prob ), dim3( BLOCK ), 0, st, L );
  hipLaunchKernelGGL( k_icp_faithful, dim3( L.n_prob ), dim3( FAITH_THREADS ), 0, st, L );
}
void launch_icp_moments( const IcpLaunch& L, hipStream_t st )
{
  hipLaunchKernelGGL( k_icp_moments, dim3( L.n_mom_blocks, L.n_prob ), dim3( BLOCK ), 0, st, L );
  hipLaunchKernelGGL( k_icp_update, dim3( L.n_prob ), dim3( UPDATE_WAVES * WAVE ), 0, st, L );
}

// ------------------------------------------------------------------------------------------
// Alignment score  (apps/pose_proposal/pose_proposal.cpp:93-158), all poses in one launch
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ void score_query( const ScoreLaunch& L, const Xform& X, int i, bool active,
                                             float& qx, float& qy, float& qz, float& nx, float& ny, float& nz )
{
  qx = qy = qz = nx = ny = nz = 0.0f;
  if( active )
  {
    float4 p = L.obj.pos[i], n = L.obj.nor[i];
    xform3( X, p.x, p.y, p.z, 1.0f, qx, qy, qz );      // :110
    xform3( X, n.x, n.y, n.z, 0.0f, nx, ny, nz );      // :111
  }
}

__device__ __forceinline__ void score_emit( const ScoreLaunch& L, int pose, int tile, bool active, int lane, const Match& m )
{
  double s = 0.0;
  if( active && m.found )
  {
    const double angle = acos( (double)m.dot );                                   // :140
    const double normals_compat = exp( -( angle * angle ) / ( 2.0 * 0.5 * 0.5 ) ); // :149
    const double dist_compat = exp( -(double)m.d2 / ( 2.0 * L.sigma * L.sigma ) ); // :150, :36-40
    s = 0.05 * normals_compat + ( 1.0 - 0.05 ) * dist_compat;                      // :102-103,151
  }
  s = wave_sum( s );
  if( lane == 0 ) L.part[(size_t)pose * L.obj.n_tiles + tile] = s;
}

// (7 waves per SIMD — 72 VGPRs, 44 B of scratch per lane — since round 3: on its quarter of the CUs the batch is bound by vector issue,
//  and a seventh wave fills more of it than the spills cost: 2.87 -> 2.77 ms there, four interleaved repeats; 5 waves, no spills: 3.00)
#ifndef RS_SCORE_OCC
#define RS_SCORE_OCC 7
#endif
#ifndef RS_SCORE_ROWS_OCC
#define RS_SCORE_ROWS_OCC 5
#endif
// RB = 0: the tile-wide search (with hand-off to k_score_coop); RB = 16: the cold search row by row, RB candidates per row and round
// Waves per workgroup of the score batch's search.  As for phase A of the ICP search: a workgroup's slots are released when its
// last wave ends, and the four tiles of a workgroup do not take equally long (~180 us each, +-30 %): one tile per workgroup is
// 1.21 -> 1.09 ms for the batch alone and 2.60 -> 2.29 ms on its 3/8 of the CUs beside the ICP chain.
#ifndef RS_SC_WAVES
#define RS_SC_WAVES 1
#endif
constexpr int SC_WAVES = RS_SC_WAVES;
template <int RB, bool KCAP = false>
__global__ __launch_bounds__( SC_WAVES * WAVE, RB ? RS_SCORE_ROWS_OCC : RS_SCORE_OCC ) void k_score( ScoreLaunch L )
{
  typedef WaveLdsT<( RB ? 4 * RB : WAVE )> Lds;
  __shared__ Lds lds[SC_WAVES];
  const int pose = blockIdx.y;
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int wib = SC_WAVES == 1 ? 0 : uni( (int)threadIdx.x / WAVE );      // (told to be uniform: as threadIdx.x / 64 the tile's number lived in a vector register pair for the whole kernel)
  EvalScope eval_scope( L.scene.evals, lds[wib], lane );
  const int tile = blockIdx.x * SC_WAVES + wib;
  if( tile >= L.obj.n_tiles ) return;
  const int i = (int)L.obj.tiles[tile] + lane;
  const bool active = i < (int)L.obj.tiles[tile + 1];
  Xform X;
#pragma unroll
  for( int k = 0; k < 16; ++k ) X.m[k] = __int_as_float( uni( __float_as_int( L.poses[pose * 16 + k] ) ) );
  float qx, qy, qz, nx, ny, nz;
  score_query( L, X, i, active, qx, qy, qz, nx, ny, nz );
  bool handoff;
  // (starting from the query's own cell, as the cold ICP search does, measured 10 % slower here: bad poses leave
  //  most lanes without a usable point in their cell, and the mixed tiles pay for the seed without skipping the shells)
  Match m;
  if constexpr( RB > 0 )
  { handoff = false; m = tile_search_rows<true, RB>( L.scene, active, qx, qy, qz, nx, ny, nz, (float)L.sigma, L.radius_sq, L.gate_tmin, L.K, lds[wib], lane ); }
  else
  {
    if constexpr( KCAP )
      m = tile_search<true, false, false, true>( L.scene, active, qx, qy, qz, nx, ny, nz, (float)L.sigma, L.radius_sq, L.gate_tmin, L.K,
                             lds[wib], lane, L.solo_stages, &handoff, nullptr, no_match(), nullptr, false, nullptr, 0, L.kcap_frac );
    else
    {
      int slog[16] = { 0 };
      m = tile_search<true>( L.scene, active, qx, qy, qz, nx, ny, nz, (float)L.sigma, L.radius_sq, L.gate_tmin, L.K,
                             lds[wib], lane, L.solo_stages, &handoff, ( RS_DBG && L.hist ) ? slog : nullptr, no_match() );
      if( RS_DBG && L.hist && lane == 0 )
      {
        // diagnostic builds (RS_HIP_SCORE_HIST): candidates streamed by shell s for u unsettled lanes -> hist[s][u]; by the rank pass for u
        // lanes that need their rank -> hist[5][u]
        for( int sh = 0; sh < 5; ++sh ) if( slog[5 + 2 * sh] > 0 ) atomicAdd( L.hist + sh * 65 + min( slog[4 + 2 * sh], 64 ), (unsigned long long)slog[5 + 2 * sh] );
        if( slog[2] > 0 ) atomicAdd( L.hist + 5 * 65 + min( slog[15], 64 ), (unsigned long long)slog[2] );
      }
    }
  }
  if( handoff )
  {
    if( lane == 0 ) { int q = atomicAdd( L.queue_count, 1 ); L.queue[q] = pose * L.obj.n_tiles + tile; }
    return;
  }
  score_emit( L, pose, tile, active, lane, m );
}

__global__ __launch_bounds__( COOP_BLOCK ) void k_score_coop( ScoreLaunch L )
{
  __shared__ WaveLds lds[COOP_WAVES];
  __shared__ CoopLds<COOP_WAVES> coop;
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int wib = threadIdx.x / WAVE;
  EvalScope eval_scope( L.scene.evals, lds[wib], lane );
  const int n_queued = *L.queue_count;
  for( int b = blockIdx.x; b < n_queued; b += gridDim.x )
  {
    const int item = L.queue[b];
    const int pose = item / L.obj.n_tiles, tile = item % L.obj.n_tiles;
    const int i = (int)L.obj.tiles[tile] + lane;
    const bool active = i < (int)L.obj.tiles[tile + 1];
    Xform X;
#pragma unroll
    for( int k = 0; k < 16; ++k ) X.m[k] = L.poses[pose * 16 + k];
    float qx, qy, qz, nx, ny, nz;
    score_query( L, X, i, active, qx, qy, qz, nx, ny, nz );
    Match m = coop_search<true, COOP_WAVES>( L.scene, active, qx, qy, qz, nx, ny, nz, (float)L.sigma, L.radius_sq, L.gate_tmin, L.K,
                                 lds[wib], coop, wib, lane, no_match() );
    if( wib == 0 ) score_emit( L, pose, tile, active, lane, m );
    __syncthreads();
  }
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
  (void)hipMemsetAsync( L.queue_count, 0, sizeof(int), st );
  dim3 grid( ( L.obj.n_tiles + SC_WAVES - 1 ) / SC_WAVES, L.n_poses );
  // by_rows: big batches on a cell grid only — nothing is handed off there.  (16 candidates per row and round; 32 and 64 were
  // measured too: 1.70 and 2.15 ms against 1.53 — rows of unequal length evaluate sentinels up to the longest one's count.)
  if( L.by_rows && L.solo_stages == 0x7fffffff && L.scene.inv_cell > 0.0f ) hipLaunchKernelGGL( k_score<16>, grid, dim3( SC_WAVES * WAVE ), 0, st, L );
  else if( L.kcap_frac > 0.0f ) hipLaunchKernelGGL( ( k_score<0, true> ), grid, dim3( SC_WAVES * WAVE ), 0, st, L );      // (opt-in experiment: RS_HIP_SCORE_KCAP)
  else
  {
    // RS_HIP_SCORE_LDS_PAD=<bytes>: dynamic LDS nobody uses — caps how many of this kernel's single-wave workgroups a CU holds, so
    // that a latency-bound chain of kernels issued beside the batch finds free slots on every CU (bench.py: the alternative to
    // confining the two to disjoint CUs)
    static const int pad = getenv( "RS_HIP_SCORE_LDS_PAD" ) ? atoi( getenv( "RS_HIP_SCORE_LDS_PAD" ) ) : 0;
    hipLaunchKernelGGL( ( k_score<0, false> ), grid, dim3( SC_WAVES * WAVE ), (size_t)( pad > 0 ? pad : 0 ), st, L );
  }
  long long items = (long long)L.obj.n_tiles * L.n_poses;
  hipLaunchKernelGGL( k_score_coop, dim3( items < 4096 ? (int)( items > 0 ? items : 1 ) : 4096 ), dim3( COOP_BLOCK ), 0, st, L );
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

#ifndef RS_LABEL_OCC
#define RS_LABEL_OCC 6
#endif
__global__ __launch_bounds__( BLOCK, RS_LABEL_OCC ) void k_label( LabelLaunch L )
{
  __shared__ WaveLds lds[WAVES_PER_BLOCK];
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int wib = threadIdx.x / WAVE;
  EvalScope eval_scope( ( L.n_pl > 0 ? L.pl[0].g.evals : nullptr ), lds[wib], lane );
  const int tile = blockIdx.x * WAVES_PER_BLOCK + wib;
  if( tile >= L.scene.n_tiles ) return;
  const int i = (int)L.scene.tiles[tile] + lane;
  const bool active = i < (int)L.scene.tiles[tile + 1];
  float4 p = make_float4( 0, 0, 0, 0 ), n = make_float4( 0, 0, 0, 0 );
  if( active ) { p = L.scene.pos[i]; n = L.scene.nor[i]; }

  // Everything this kernel reads and writes per point is indexed by the point's QUERY slot i: coalesced.  (Indexed by the
  // original index — a random permutation of the slots — every 4-byte access was its own memory transaction: 75 MB written
  // and 215 MB fetched per launch for 5 MB of results.)  k_label_to_input_order / k_label_to_query_order move whole arrays
  // between the two orders by GATHERING, whose random side is a read that the L2 absorbs.
  float best_min = 1e9f;                                                       // :799-802,820
  int label = 0;
  if( active && L.min_d && !L.fresh ) { best_min = L.min_d[i]; label = L.labels[i]; }

  for( int k = 0; k < L.n_pl; ++k )
  {
    const PlacementDev& pl = L.pl[k];
    float qx, qy, qz;
    xform3( pl.inv, p.x, p.y, p.z, 1.0f, qx, qy, qz );                         // :755
    Match m = tile_search<false>( pl.g, active, qx, qy, qz, 0.0f, 0.0f, 0.0f, pl.radius, pl.radius_sq, 0.0f, 1,
                                  lds[wib], lane, 0, nullptr, nullptr, no_match() );   // :758 (K = 1)
    // :762-775 — found, strictly closer than the running minimum, and within 70° (either sign)
    bool ok = false;
    if( active && m.found && ( L.rows != nullptr || m.d2 < best_min ) )
    {
      float n1x, n1y, n1z;
      xform3( pl.nmat, n.x, n.y, n.z, 0.0f, n1x, n1y, n1z );                   // :766
      float4 m4 = pl.g.nor[m.slot];
      float n2x = m4.x, n2y = m4.y, n2z = m4.z;
      unit3( n1x, n1y, n1z ); unit3( n2x, n2y, n2z );
      float dot = fabsf( n1x * n2x + n1y * n2y + n1z * n2z );                  // :769
      ok = ( dot >= L.gate_tmin ) && ( dot <= 1.0f );
    }
    if( L.rows ) { if( active ) L.rows[(size_t)k * L.scene.n + i] = ok ? m.d2 : INFINITY; }
    else if( ok ) { best_min = m.d2; label = L.label_base + k + 1; }
  }
  if( active && L.min_d ) { L.min_d[i] = best_min; L.labels[i] = (int8_t)label; }
}

void launch_label( const LabelLaunch& L, hipStream_t st )
{
  hipLaunchKernelGGL( k_label, dim3( ( L.scene.n_tiles + WAVES_PER_BLOCK - 1 ) / WAVES_PER_BLOCK ), dim3( BLOCK ), 0, st, L );
}

// query order -> input order: thread j (an original index) reads its slot's values.  n_f float arrays of n entries, one after
// the other, and (optionally) one int8 array.
__global__ __launch_bounds__( BLOCK ) void k_label_to_input_order( const int* by_orig, long long n, const float* in_f, float* out_f, int n_f,
                                                                   const int8_t* in_b, int8_t* out_b )
{
  const long long j = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if( j >= n ) return;
  const int s = by_orig[j];
  for( int a = 0; a < n_f; ++a ) out_f[(size_t)a * n + j] = in_f[(size_t)a * n + s];
  if( in_b ) out_b[j] = in_b[s];
}
// input order -> query order: thread s (a slot) reads the values of its original index (pos[s].w)
__global__ __launch_bounds__( BLOCK ) void k_label_to_query_order( const float4* qpos, long long n, const float* in_f, float* out_f,
                                                                   const int8_t* in_b, int8_t* out_b )
{
  const long long s = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if( s >= n ) return;
  const int j = __float_as_int( qpos[s].w );
  if( in_f ) out_f[s] = in_f[j];
  if( in_b ) out_b[s] = in_b[j];
}
// The tail of rspf_arrangement_to_labels (lib/rs/rs_pointcloud_filters.cpp:851-869): temporary labels -> class / instance
// ids, written in input order together with the state's move out of query order.  label 0: (unlabelled class, 1024).
__global__ __launch_bounds__( BLOCK ) void k_label_ids_to_input_order( const int* by_orig, long long n, const int8_t* labels_q, const float* mind_q,
                                                                       const int* plc_class, const int* plc_uidx, int unlabelled_class,
                                                                       int* class_ids, int* instance_ids, int8_t* labels, float* min_d )
{
  const long long j = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if( j >= n ) return;
  const int s = by_orig[j];
  const int l = labels_q[s];
  class_ids[j] = l == 0 ? unlabelled_class : plc_class[l - 1];            // :856-866
  instance_ids[j] = l == 0 ? 1024 : plc_uidx[l - 1];                      // RSPF_MAX_INSTANCES (:20)
  labels[j] = (int8_t)l; min_d[j] = mind_q[s];
}
void launch_label_ids_to_input_order( const int* by_orig, long long n, const int8_t* labels_q, const float* mind_q, const int* plc_class, const int* plc_uidx,
                                      int unlabelled_class, int* class_ids, int* instance_ids, int8_t* labels, float* min_d, hipStream_t st )
{
  hipLaunchKernelGGL( k_label_ids_to_input_order, dim3( (unsigned)( ( n + BLOCK - 1 ) / BLOCK ) ), dim3( BLOCK ), 0, st, by_orig, n, labels_q, mind_q,
                      plc_class, plc_uidx, unlabelled_class, class_ids, instance_ids, labels, min_d );
}

// dst[i] = src[idx[i]] for records of `words` 32-bit words (the attribute gathers of a level, lib/rs/rs_pointcloud.h:1090-1099)
__global__ __launch_bounds__( BLOCK ) void k_gather_words( const uint32_t* src, const int* idx, long long count, int words, uint32_t* dst )
{
  const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if( t >= count * words ) return;
  const long long i = t / words; const int w = (int)( t - i * words );
  dst[t] = src[(size_t)idx[i] * words + w];
}
void launch_gather_words( const uint32_t* src, const int* idx, long long count, int words, uint32_t* dst, hipStream_t st )
{
  if( count <= 0 ) return;
  hipLaunchKernelGGL( k_gather_words, dim3( (unsigned)( ( count * words + BLOCK - 1 ) / BLOCK ) ), dim3( BLOCK ), 0, st, src, idx, count, words, dst );
}

void launch_label_to_input_order( const int* by_orig, long long n, const float* in_f, float* out_f, int n_f, const int8_t* in_b, int8_t* out_b, hipStream_t st )
{
  hipLaunchKernelGGL( k_label_to_input_order, dim3( (unsigned)( ( n + BLOCK - 1 ) / BLOCK ) ), dim3( BLOCK ), 0, st, by_orig, n, in_f, out_f, n_f, in_b, out_b );
}
void launch_label_to_query_order( const float4* qpos, long long n, const float* in_f, float* out_f, const int8_t* in_b, int8_t* out_b, hipStream_t st )
{
  hipLaunchKernelGGL( k_label_to_query_order, dim3( (unsigned)( ( n + BLOCK - 1 ) / BLOCK ) ), dim3( BLOCK ), 0, st, qpos, n, in_f, out_f, in_b, out_b );
}

// Ordered arg-min over per-placement rows that already sit in device memory (the gathered send buffers of the sharded
// route): rows 0..n-1 applied in order with the strict `<` of rs_pointcloud_filters.cpp:763, so an earlier
// placement wins a tie exactly as in the sequential loop.  Row k starts at rows + offsets[k] (floats).
__global__ __launch_bounds__( BLOCK ) void k_label_fold( const float* rows, const long long* offsets, int n_rows, long long n, int label_base,
                                                         int8_t* labels, float* min_d, bool fresh )
{
  const long long j = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if( j >= n ) return;
  float best = 1e9f; int label = 0;                                            // :799-802,820
  if( !fresh ) { best = min_d[j]; label = labels[j]; }
  for( int k = 0; k < n_rows; ++k )
  {
    const float v = rows[offsets[k] + j];
    if( v < best ) { best = v; label = label_base + k + 1; }                  // :763,772-773
  }
  min_d[j] = best; labels[j] = (int8_t)label;
}
// The same over per-RANK partials of the loop — rank r's (min_dist, label) after its own contiguous run of the sorted arrangement,
// labels already carrying the run's base — folded in rank order with the same strict `<`: a later run only takes a point it is
// strictly closer to, exactly as the sequential loop would have.
__global__ __launch_bounds__( BLOCK ) void k_label_fold_partials( const float* base, const long long* min_off, const long long* lab_off, int n_parts, long long n,
                                                                  int8_t* labels, float* min_d )
{
  const long long j = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if( j >= n ) return;
  float best = 1e9f; int label = 0;                                            // :799-802,820
  for( int r = 0; r < n_parts; ++r )
  {
    const float v = base[min_off[r] + j];
    if( v < best ) { best = v; label = reinterpret_cast<const int8_t*>( base )[lab_off[r] + j]; }
  }
  min_d[j] = best; labels[j] = (int8_t)label;
}
void launch_label_fold_partials( const float* base, const long long* min_off, const long long* lab_off, int n_parts, long long n, int8_t* labels, float* min_d, hipStream_t st )
{
  hipLaunchKernelGGL( k_label_fold_partials, dim3( (unsigned)( ( n + BLOCK - 1 ) / BLOCK ) ), dim3( BLOCK ), 0, st, base, min_off, lab_off, n_parts, n, labels, min_d );
}
void launch_label_fold( const float* rows, const long long* offsets, int n_rows, long long n, int label_base, int8_t* labels, float* min_d, bool fresh, hipStream_t st )
{
  hipLaunchKernelGGL( k_label_fold, dim3( (unsigned)( ( n + BLOCK - 1 ) / BLOCK ) ), dim3( BLOCK ), 0, st, rows, offsets, n_rows, n, label_base, labels, min_d, fresh );
}

// ------------------------------------------------------------------------------------------
// Generic rows: the k nearest within the radius, ascending  (msh_hash_grid.h:1090-1259)
// Compatibility path for callers that want the whole neighbour list.  Selection by
// successive minima: pass t finds, per query, the smallest (dist², index) greater than the
// one found in pass t-1.  No per-lane storage, rows come out sorted.
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__( BLOCK ) void k_rows( RowsLaunch L )
{
  __shared__ WaveLds lds[WAVES_PER_BLOCK];
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int wib = threadIdx.x / WAVE;
  EvalScope eval_scope( L.tgt.evals, lds[wib], lane );
  const int tile = blockIdx.x * WAVES_PER_BLOCK + wib;
  if( tile >= L.q.n_tiles ) return;
  const int i = (int)L.q.tiles[tile] + lane;
  const bool active = i < (int)L.q.tiles[tile + 1];
  float4 q = make_float4( 0, 0, 0, 0 );
  if( active ) q = L.q.pos[i];
  const int orig = __float_as_int( q.w );
  const TileBounds tb = wave_bounds( active, q.x, q.y, q.z );
  CellBox box = cell_box( L.tgt, tb, L.radius );
  const bool searchable = tb.any && !box_empty( box );

  float pd2 = -1.0f; int pidx = -1;      // previous pick; dist² >= 0 so (-1,-1) precedes everything
  int count = 0;
  bool more = active && searchable;
  for( int t = 0; t < L.K; ++t )
  {
    if( !__any( more ) ) break;
    float bd2 = INFINITY; int bidx = INT_MAX;
    sweep_shell<false>( L.tgt, box, box, false, lds[wib], lane, 0, 1, [&]( const float4& X, const float4& Y, const float4& Z, int k4 )
    {
      float d[4];
      dist2x4( X, Y, Z, q.x, q.y, q.z, d[0], d[1], d[2], d[3] );
#pragma unroll
      for( int c = 0; c < 4; ++c )
      {
        const int idx = lds[wib].pidx[k4 + c];
        if( (int)more & (int)( d[c] < L.radius_sq ) & (int)lex_less( pd2, pidx, d[c], idx ) & (int)lex_less( d[c], idx, bd2, bidx ) ) { bd2 = d[c]; bidx = idx; }
      }
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
  hipLaunchKernelGGL( k_rows, dim3( ( L.q.n_tiles + WAVES_PER_BLOCK - 1 ) / WAVES_PER_BLOCK ), dim3( BLOCK ), 0, st, L );
}

// ------------------------------------------------------------------------------------------
// Generic rows, one WAVE per query: the form msh_hash_grid_radius_search takes when the reference's own
// consumers call it unchanged (mgs_compute_object_alignment_score: a few hundred object points per call, K = 64 / 32,
// apps/pose_proposal/pose_proposal.cpp:115-124, tens of thousands of calls).  Queries arrive in the caller's order —
// no Hilbert sort, no tiling, nothing but this launch between the upload and the download.  The 64 lanes stream the
// cells within the radius of their ONE query (same flattened row-piece stream as sweep_shell), every lane tests its own
// candidate, hits are appended to the wave's LDS list by ballot / prefix count, the list is sorted by (dist², index)
// with a bitonic network, and the first K entries are the row.  A query with more than ROWS_CAP points within the radius
// raises `overflow` and is left to k_rows (successive minima need no storage).
// ------------------------------------------------------------------------------------------
#define ROWS_CAP 1024
#define ROWS_WAVES 4
struct RowsWaveLds { float d2[ROWS_CAP]; int idx[ROWS_CAP]; uint32_t seg[WAVE], pre[WAVE]; };

// Bitonic sort of 64 U entries by (dist², index), entry e = 64 u + lane held in registers: exchanges at distance >= 64 are
// between a lane's own registers, the others one lane permute per register — against one LDS round trip (read, compare, write,
// fence) per stage when the row sits in LDS: 28 stages cost 1.2 us instead of 6 for 128 hits, and the sort is half of what a
// wave of a small call does.
template <int U>
__device__ __forceinline__ void rows_sort_regs( float ( &d )[U], int ( &ix )[U], int lane )
{
#pragma unroll
  for( int k = 2; k <= WAVE * U; k <<= 1 )
  {
#pragma unroll
    for( int jj = k >> 1; jj > 0; jj >>= 1 )
    {
      if( jj >= WAVE )
      {
        const int du = jj / WAVE;
#pragma unroll
        for( int u = 0; u < U; ++u )
        {
          if( u & du ) continue;
          const int v = u | du;
          const bool up = ( ( u * WAVE + lane ) & k ) == 0;
          const bool sw = up ? lex_less( d[v], ix[v], d[u], ix[u] ) : lex_less( d[u], ix[u], d[v], ix[v] );
          const float td = d[u]; const int ti = ix[u];
          d[u] = sw ? d[v] : td; ix[u] = sw ? ix[v] : ti;
          d[v] = sw ? td : d[v]; ix[v] = sw ? ti : ix[v];
        }
      }
      else
      {
#pragma unroll
        for( int u = 0; u < U; ++u )
        {
          const float pd = __shfl_xor( d[u], jj, WAVE ); const int pi = __shfl_xor( ix[u], jj, WAVE );
          const bool keep_min = ( ( lane & jj ) == 0 ) == ( ( ( u * WAVE + lane ) & k ) == 0 );
          const bool take = keep_min ? lex_less( pd, pi, d[u], ix[u] ) : lex_less( d[u], ix[u], pd, pi );
          d[u] = take ? pd : d[u]; ix[u] = take ? pi : ix[u];
        }
      }
    }
  }
}
template <int U>
__device__ __forceinline__ void rows_sort_emit( const RowsWaveLds& L, uint32_t count, uint32_t n_out, int lane, float* out_d2, int* out_idx )
{
  float d[U]; int ix[U];
#pragma unroll
  for( int u = 0; u < U; ++u )
  {
    const uint32_t e = (uint32_t)( u * WAVE + lane );
    d[u] = e < count ? L.d2[e] : INFINITY; ix[u] = e < count ? L.idx[e] : INT_MAX;
  }
  rows_sort_regs<U>( d, ix, lane );
#pragma unroll
  for( int u = 0; u < U; ++u )
  {
    const uint32_t e = (uint32_t)( u * WAVE + lane );
    if( e < n_out ) { out_d2[e] = d[u]; out_idx[e] = ix[u]; }
  }
}

__global__ __launch_bounds__( ROWS_WAVES * WAVE ) void k_rows_wave( GridView g, const float* q3, int nq, int K, float radius, float radius_sq,
                                                                    float* out_d2, int* out_idx, int* out_nn, int* overflow )
{
  __shared__ RowsWaveLds lds[ROWS_WAVES];
  const int lane = threadIdx.x & ( WAVE - 1 ), wib = threadIdx.x / WAVE;
  const int qi = blockIdx.x * ROWS_WAVES + wib;
  if( qi >= nq ) return;
  RowsWaveLds& L = lds[wib];
  const float qx = q3[3 * qi], qy = q3[3 * qi + 1], qz = q3[3 * qi + 2];
  int x0 = 0, x1 = 0, y0 = 0, y1 = 0, z0 = 0, z1 = 0;
  const bool grid = g.inv_cell > 0.0f;       // (the one-cell brute layout has a one-entry table: cell (0,0,0) is the whole cloud)
  if( grid )
  {
    axis_range( qx, qx, radius, g.minx, g.inv_cell, g.w, x0, x1 );
    axis_range( qy, qy, radius, g.miny, g.inv_cell, g.h, y0, y1 );
    axis_range( qz, qz, radius, g.minz, g.inv_cell, g.d, z0, z1 );
  }
  const bool finite = fabsf( qx ) <= FLT_MAX && fabsf( qy ) <= FLT_MAX && fabsf( qz ) <= FLT_MAX;       // (false for NaN)
  const bool empty = !finite | ( x1 < x0 ) | ( y1 < y0 ) | ( z1 < z0 ) | ( g.n == 0 );
  const int ny = y1 - y0 + 1;
  const int n_rows = empty ? 0 : ny * ( z1 - z0 + 1 );
  uint32_t count = 0;                                   // hits so far (wave-uniform)
  for( int r0 = 0; r0 < n_rows; r0 += WAVE )
  {
    const int r = r0 + lane;
    uint32_t sa = 0, la = 0;
    if( r < n_rows )
    {
      const int rz = r / ny, y = y0 + ( r - rz * ny ), z = z0 + rz;
      const uint32_t* cs = g.cell_start + (size_t)( z * g.h + y ) * g.w;
      sa = cs[x0]; la = cs[x1 + 1] - sa;
    }
    const uint32_t incl = wave_scan( la, lane );
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane( (int)incl, WAVE - 1 );
    L.seg[lane] = sa; L.pre[lane] = incl - la;
    wave_lds_fence();
    // four chunks of 64 candidates per round, their loads issued together: a small call is one wave per query with nothing else
    // on its SIMD, so every dependent load is exposed latency (a 3 x 3 x 3-cell neighbourhood is 4-6 chunks)
    for( uint32_t c0 = 0; c0 < total; c0 += 4 * WAVE )
    {
      float4 P[4]; bool ok[4];
#pragma unroll
      for( int u = 0; u < 4; ++u )
      {
        const uint32_t j = c0 + (uint32_t)( u * WAVE + lane );
        ok[u] = j < total;
        P[u] = make_float4( 0.0f, 0.0f, 0.0f, 0.0f );
        if( ok[u] )
        {
          int row = 0;
#pragma unroll
          for( int step = WAVE / 2; step > 0; step >>= 1 ) { if( L.pre[row + step] <= j ) row += step; }
          P[u] = g.pos[L.seg[row] + ( j - L.pre[row] )];
        }
      }
#pragma unroll
      for( int u = 0; u < 4; ++u )
      {
        if( c0 + (uint32_t)( u * WAVE ) >= total ) break;
        const float vx = P[u].x - qx, vy = P[u].y - qy, vz = P[u].z - qz;
        const float d2 = vx * vx + vy * vy + vz * vz;     // msh_hash_grid.h:852-855
        const int idx = __float_as_int( P[u].w );
        const bool hit = ok[u] && d2 < radius_sq;         // :857
        const unsigned long long mask = __ballot( hit );
        if( hit )
        {
          const uint32_t at = count + (uint32_t)__builtin_amdgcn_mbcnt_hi( (uint32_t)( mask >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)mask, 0u ) );
          if( at < ROWS_CAP ) { L.d2[at] = d2; L.idx[at] = idx; }
        }
        count += (uint32_t)__popcll( mask );
      }
    }
    wave_lds_fence();
  }
  if( count > ROWS_CAP ) { if( lane == 0 ) { *(volatile int*)overflow = 1; out_nn[qi] = -1; } return; }      // (plain stores: the words may live in host memory)
  // (Ordering a row by counting — every hit counts the hits that precede it and stores itself at that position — instead of
  //  sorting was tried: 128 hits cost about the same as the 28 LDS round trips of the bitonic network, more hits cost more.)
  const uint32_t n_out = count < (uint32_t)K ? count : (uint32_t)K;
  if( count <= 4 * WAVE )
  {
    float* od = out_d2 + (size_t)qi * K; int* oi = out_idx + (size_t)qi * K;
    if( count <= WAVE )          rows_sort_emit<1>( L, count, n_out, lane, od, oi );
    else if( count <= 2 * WAVE ) rows_sort_emit<2>( L, count, n_out, lane, od, oi );
    else                         rows_sort_emit<4>( L, count, n_out, lane, od, oi );
    // the count is the row's "ready" flag for a host that polls it (rows and counts in host memory): rows first, system-wide
    __threadfence_system();
    if( lane == 0 ) out_nn[qi] = (int)n_out;
    return;
  }
  // more than 256 hits: bitonic sort in LDS of the first `count` entries (padded with +inf up to a power of two) by (dist², index)
  uint32_t m = WAVE; while( m < count ) m <<= 1;
  for( uint32_t t = count + lane; t < m; t += WAVE ) { L.d2[t] = INFINITY; L.idx[t] = INT_MAX; }
  wave_lds_fence();
  if( count > 1 )
  for( uint32_t k = 2; k <= m; k <<= 1 )
    for( uint32_t jj = k >> 1; jj > 0; jj >>= 1 )
    {
      for( uint32_t t = lane; t < ( m >> 1 ); t += WAVE )
      {
        const uint32_t lo = ( ( t & ~( jj - 1 ) ) << 1 ) | ( t & ( jj - 1 ) ), hi = lo | jj;
        const bool up = ( lo & k ) == 0;
        const float da = L.d2[lo], db = L.d2[hi]; const int ia = L.idx[lo], ib = L.idx[hi];
        const bool swap = up ? lex_less( db, ib, da, ia ) : lex_less( da, ia, db, ib );
        if( swap ) { L.d2[lo] = db; L.idx[lo] = ib; L.d2[hi] = da; L.idx[hi] = ia; }
      }
      wave_lds_fence();
    }
  for( uint32_t t = lane; t < n_out; t += WAVE ) { out_d2[(size_t)qi * K + t] = L.d2[t]; out_idx[(size_t)qi * K + t] = L.idx[t]; }
  __threadfence_system();
  if( lane == 0 ) out_nn[qi] = (int)n_out;
}

void launch_rows_wave( const GridView& g, const float* q3, int nq, int K, float radius, float radius_sq,
                       float* out_d2, int* out_idx, int* out_nn, int* overflow, hipStream_t st )
{
  hipLaunchKernelGGL( k_rows_wave, dim3( ( nq + ROWS_WAVES - 1 ) / ROWS_WAVES ), dim3( ROWS_WAVES * WAVE ), 0, st, g, q3, nq, K, radius, radius_sq,
                      out_d2, out_idx, out_nn, overflow );
}

// ------------------------------------------------------------------------------------------
// Level builder: Poisson-disk subsample in input order  (lib/rs/rs_pointcloud.h:984-1106)
//
// The reference walks the points in input order: the first unmarked point becomes a sample and marks every point its
// radius search returns (the max_n_neigh nearest within the radius, itself included).  Equivalent statement, as long
// as no search is truncated by max_n_neigh (checked: n_within <= max_n_neigh for every point):
//     point i is a sample  <=>  no EARLIER point within the radius is a sample.
// That is the lexicographically first maximal independent set of the "within radius" graph, decided here in
// dependence order instead of index order, every edge touched once:
//   word[k] = number of earlier neighbours of k not yet known to be covered  (| COVERED once a sample marks k)
//   a SAMPLE j   ORs COVERED into the word of each later neighbour k; the first one to do so puts k on the frontier
//   a COVERED j  decrements the word of each later neighbour k; the decrement that makes it 0 (all earlier neighbours
//                covered, hence none of them a sample: nobody can still set COVERED) makes k a sample, onto the frontier
// One launch per frontier; the number of launches is the longest dependence chain — a handful for shuffled input, of
// the order of the cloud's extent in sample spacings for raster-like vertex orders.  The result is the reference's
// sample set, bit for bit.
//   k_level_neighbours<false>  counts per point its earlier / later neighbours and all points within the radius
//   (scan)                     row offsets
//   k_level_neighbours<true>   writes the later neighbours' original indices
//   k_level_init               word = number of earlier neighbours; points without any are the first frontier (samples)
//   k_level_frontier           one step
//   k_level_flags / scatter    the samples in increasing index order
// ------------------------------------------------------------------------------------------
#define LEVEL_COVERED 0x40000000
template <bool WRITE>
__global__ __launch_bounds__( BLOCK ) void k_level_neighbours( LevelLaunch L )
{
  __shared__ WaveLds lds[WAVES_PER_BLOCK];
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int wib = threadIdx.x / WAVE;
  EvalScope eval_scope( L.tgt.evals, lds[wib], lane );
  const int tile = blockIdx.x * WAVES_PER_BLOCK + wib;
  if( tile >= L.q.n_tiles ) return;
  const int i = (int)L.q.tiles[tile] + lane;
  const bool active = i < (int)L.q.tiles[tile + 1];
  float4 q = make_float4( 0, 0, 0, 0 );
  if( active ) q = L.q.pos[i];
  const int orig = __float_as_int( q.w );
  const TileBounds tb = wave_bounds( active, q.x, q.y, q.z );
  CellBox box = cell_box( L.tgt, tb, L.radius );
  int earlier = 0, later = 0, within = 0;
  int* row = WRITE && active ? L.adj + L.offset[i] : nullptr;
  if( tb.any && !box_empty( box ) )
  {
    sweep_shell<false>( L.tgt, box, box, false, lds[wib], lane, 0, 1, [&]( const float4& X, const float4& Y, const float4& Z, int k4 )
    {
      float d[4];
      dist2x4( X, Y, Z, q.x, q.y, q.z, d[0], d[1], d[2], d[3] );       // candidate - query, msh_hash_grid.h:852-855 (the square is the same either way round)
#pragma unroll
      for( int c = 0; c < 4; ++c )
      {
        if( active & ( d[c] < L.radius_sq ) )                          // strict, :857
        {
          const int idx = lds[wib].pidx[k4 + c];
          within++;
          if( idx < orig ) earlier++;
          if( idx > orig ) { if( WRITE ) row[later] = idx; later++; }
        }
      }
    } );
  }
  if( !WRITE && active )
  {
    L.n_earlier[i] = earlier; L.n_later[i] = later;
    if( within > L.max_n_neigh ) *L.over_cap = 1;
  }
}

__global__ __launch_bounds__( BLOCK ) void k_level_init( LevelLaunch L )
{
  const int s = blockIdx.x * BLOCK + threadIdx.x;
  if( s >= L.n ) return;
  const int i = __float_as_int( L.q.pos[s].w );
  const int ne = L.n_earlier[s];
  L.word[i] = ne;
  L.state[i] = ne == 0 ? 1 : 0;
  if( ne == 0 ) L.front_out[atomicAdd( L.front_count_out, 1 )] = ( i << 1 ) | 1;
}

// frontier items: (original index << 1) | (1 = sample, 0 = covered).  G lanes per item, one lane per later neighbour:
// the atomics of an item are in flight together (one thread per item would wait for each of its ~100 in turn at
// level 4; at level 1, with two neighbours per point, a whole wave per item would idle), and a wave's new frontier
// entries take consecutive slots with one counter update.
template <int G>
__global__ __launch_bounds__( BLOCK ) void k_level_frontier( LevelLaunch L )
{
  const int m = *L.front_count_in;
  const int lane = threadIdx.x & ( WAVE - 1 );
  const int sub = threadIdx.x & ( G - 1 );
  const int n_groups = gridDim.x * ( BLOCK / G );
  // (every lane of a wave runs the same number of outer and inner iterations: the ballot below needs them all)
  const int m_pad = ( m + ( WAVE / G ) - 1 ) / ( WAVE / G ) * ( WAVE / G );
  for( int t = blockIdx.x * ( BLOCK / G ) + threadIdx.x / G; t < m_pad; t += n_groups )
  {
    int j = 0; bool sample = false; unsigned e0 = 0, e1 = 0;
    if( t < m )
    {
      const int item = L.front_in[t];
      j = item >> 1; sample = item & 1;
      const int s = L.by_orig[j];
      e0 = L.offset[s]; e1 = L.offset[s + 1];
    }
    // longest row among the items this wave is working on
    unsigned len = e1 - e0;
#pragma unroll
    for( int o = WAVE / 2; o >= G; o >>= 1 ) len = max( len, (unsigned)__shfl_xor( (int)len, o ) );
    for( unsigned eb = 0; eb < len; eb += G )
    {
      const unsigned e = e0 + eb + (unsigned)sub;
      int push = -1;
      if( e < e1 )
      {
        const int k = L.adj[e];
        if( sample )
        {
          const int old = atomicOr( L.word + k, LEVEL_COVERED );
          if( !( old & LEVEL_COVERED ) ) push = k << 1;
        }
        else
        {
          const int now = atomicSub( L.word + k, 1 ) - 1;
          if( now == 0 ) { L.state[k] = 1; push = ( k << 1 ) | 1; }
        }
      }
      const unsigned long long mask = __ballot( push >= 0 );
      if( mask )
      {
        int base = 0;
        if( lane == 0 ) base = atomicAdd( L.front_count_out, (int)__popcll( mask ) );
        base = __builtin_amdgcn_readfirstlane( base );
        if( push >= 0 ) L.front_out[base + (int)__popcll( mask & ( ( 1ull << lane ) - 1ull ) )] = push;
      }
    }
  }
}

__global__ __launch_bounds__( BLOCK ) void k_level_flags( LevelLaunch L )
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if( i <= L.n ) L.flags[i] = ( i < L.n && L.state[i] == 1 ) ? 1u : 0u;
}
__global__ __launch_bounds__( BLOCK ) void k_level_scatter( LevelLaunch L )
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if( i < L.n && L.flags[i] ) L.samples[L.flag_scan[i]] = i;
}
__global__ __launch_bounds__( BLOCK ) void k_level_gather( const int* samples, int count, const int* by_orig, const float4* qpos, const float4* qnor,
                                                           float* pos3, float* nor3 )
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if( i >= count ) return;
  const int s = by_orig[samples[i]];
  const float4 p = qpos[s];
  pos3[3 * i] = p.x; pos3[3 * i + 1] = p.y; pos3[3 * i + 2] = p.z;
  if( nor3 ) { const float4 m = qnor[s]; nor3[3 * i] = m.x; nor3[3 * i + 1] = m.y; nor3[3 * i + 2] = m.z; }
}
void launch_level_gather( const int* samples, int count, const int* by_orig, const float4* qpos, const float4* qnor,
                          float* pos3, float* nor3, hipStream_t st )
{ hipLaunchKernelGGL( k_level_gather, dim3( ( count + BLOCK - 1 ) / BLOCK ), dim3( BLOCK ), 0, st, samples, count, by_orig, qpos, qnor, pos3, nor3 ); }
void launch_level_neighbours( const LevelLaunch& L, bool write, hipStream_t st )
{
  const dim3 grid( ( L.q.n_tiles + WAVES_PER_BLOCK - 1 ) / WAVES_PER_BLOCK );
  if( write ) hipLaunchKernelGGL( k_level_neighbours<true>, grid, dim3( BLOCK ), 0, st, L );
  else        hipLaunchKernelGGL( k_level_neighbours<false>, grid, dim3( BLOCK ), 0, st, L );
}
void launch_level_init( const LevelLaunch& L, hipStream_t st )
{ hipLaunchKernelGGL( k_level_init, dim3( ( L.n + BLOCK - 1 ) / BLOCK ), dim3( BLOCK ), 0, st, L ); }
void launch_level_frontier( const LevelLaunch& L, int lanes_per_item, int blocks, hipStream_t st )
{
  const dim3 grid( std::max( 1, blocks ) );
  if( lanes_per_item >= 64 )     hipLaunchKernelGGL( k_level_frontier<64>, grid, dim3( BLOCK ), 0, st, L );
  else if( lanes_per_item >= 8 ) hipLaunchKernelGGL( k_level_frontier<8>, grid, dim3( BLOCK ), 0, st, L );
  else                           hipLaunchKernelGGL( k_level_frontier<1>, grid, dim3( BLOCK ), 0, st, L );
}
void launch_level_flags( const LevelLaunch& L, hipStream_t st )
{ hipLaunchKernelGGL( k_level_flags, dim3( ( L.n + BLOCK ) / BLOCK ), dim3( BLOCK ), 0, st, L ); }
void launch_level_scatter( const LevelLaunch& L, hipStream_t st )
{ hipLaunchKernelGGL( k_level_scatter, dim3( ( L.n + BLOCK - 1 ) / BLOCK ), dim3( BLOCK ), 0, st, L ); }

// ------------------------------------------------------------------------------------------
// Neighbourhood graph  (lib/rs/rs_pointcloud_filters.cpp:674-722)
// The K = 8 self-search is k_rows; these kernels turn its rows into the de-duplicated edge list.
// Reference order: rows i ascending, first insertion of an undirected pair wins, so {i,j} (i<j)
// is stored as (i,j) when j is in row i and as (j,i) otherwise.  Equivalent rule per directed
// entry i -> j:  keep it iff  i <= j  or  i is not in row j.
// (The reference's int32 key max*n+min wraps for n > 46340 and then drops whichever edges
// collide; that accident is not reproduced — see DESIGN.md §4.)
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ bool edge_kept( const EdgeLaunch& L, int i, int j )
{
  if( i <= j ) return true;
  const int nj = L.row_nn[j];
  for( int t = 0; t < nj; ++t ) if( L.row_idx[(size_t)j * L.K + t] == i ) return false;
  return true;
}

// x^e for a small integer e >= 0 in double-double arithmetic (error-free products via fma), rounded
// once to double: agrees with a correctly rounded pow() for these arguments.
__device__ __forceinline__ double powi_dd( double x, int e )
{
  double rh = 1.0, rl = 0.0, bh = x, bl = 0.0;
  while( e > 0 )
  {
    if( e & 1 )
    {
      double ph = rh * bh, pl = fma( rh, bh, -ph ) + ( rh * bl + rl * bh );
      double sh = ph + pl; rl = pl - ( sh - ph ); rh = sh;
    }
    e >>= 1;
    if( e )
    {
      double ph = bh * bh, pl = fma( bh, bh, -ph ) + 2.0 * ( bh * bl );
      double sh = ph + pl; bl = pl - ( sh - ph ); bh = sh;
    }
  }
  return rh + rl;
}

// rs_pointcloud_filters.cpp:706-708: (float)(1.0f - pow( d2/(4.0*r2), dist_exp )) * powf( clamp(dot,0,1), angle_exp )
__device__ __forceinline__ float edge_weight( const EdgeLaunch& L, float d2, float dot )
{
  const double y = (double)d2 / ( 4.0 * (double)L.radius_sq );
  const double p = L.dist_int >= 0 ? powi_dd( y, L.dist_int ) : pow( y, (double)L.dist_exp );
  const float dist_cost = (float)( 1.0 - p );
  float c = dot > 0.0f ? dot : 0.0f;
  c = c < 1.0f ? c : 1.0f;
  float norm_cost;
  if( L.angle_int >= 0 )
  {
    double b = c, r = 1.0; int e = L.angle_int;          // powf computes in double and rounds once
    while( e > 0 ) { if( e & 1 ) r *= b; e >>= 1; if( e ) b *= b; }
    norm_cost = (float)r;
  }
  else norm_cost = powf( c, L.angle_exp );
  return dist_cost * norm_cost;
}

__global__ __launch_bounds__( BLOCK ) void k_edge_count( EdgeLaunch L )
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if( i >= L.n ) return;
  int c = 0;
  const int ni = L.row_nn[i];
  for( int t = 0; t < ni; ++t ) c += edge_kept( L, i, L.row_idx[(size_t)i * L.K + t] ) ? 1 : 0;
  L.count[i] = c;
}

// exclusive scan of count[0..n) by one workgroup: each thread sums a contiguous slice, the slice
// totals are scanned through LDS, then each thread writes its slice.  Fixed order, no atomics.
__global__ __launch_bounds__( 1024 ) void k_edge_scan( EdgeLaunch L )
{
  __shared__ unsigned part[1024];
  const int T = 1024, t = threadIdx.x;
  const int per = ( L.n + T - 1 ) / T;
  const int lo = min( t * per, L.n ), hi = min( lo + per, L.n );
  unsigned s = 0;
  for( int i = lo; i < hi; ++i ) s += (unsigned)L.count[i];
  part[t] = s;
  __syncthreads();
  for( int o = 1; o < T; o <<= 1 )
  {
    unsigned v = ( t >= o ) ? part[t - o] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  unsigned run = part[t] - s;
  for( int i = lo; i < hi; ++i ) { L.offset[i] = run; run += (unsigned)L.count[i]; }
  if( t == T - 1 ) L.offset[L.n] = part[T - 1];
}

__global__ __launch_bounds__( BLOCK ) void k_edge_write( EdgeLaunch L )
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if( i >= L.n ) return;
  unsigned at = L.offset[i];
  const int ni = L.row_nn[i];
  const float nx = L.nor[3*i], ny = L.nor[3*i+1], nz = L.nor[3*i+2];
  for( int t = 0; t < ni; ++t )
  {
    const int j = L.row_idx[(size_t)i * L.K + t];
    if( !edge_kept( L, i, j ) ) continue;
    const float dot = nx * L.nor[3*j] + ny * L.nor[3*j+1] + nz * L.nor[3*j+2];     // msh_vec3_dot( n, m )
    L.e1[at] = i; L.e2[at] = j; L.ew[at] = edge_weight( L, L.row_d2[(size_t)i * L.K + t], dot );
    ++at;
  }
}

void launch_edge_count( const EdgeLaunch& L, hipStream_t st )
{ hipLaunchKernelGGL( k_edge_count, dim3( ( L.n + BLOCK - 1 ) / BLOCK ), dim3( BLOCK ), 0, st, L ); }
void launch_edge_scan( const EdgeLaunch& L, hipStream_t st )
{ hipLaunchKernelGGL( k_edge_scan, dim3( 1 ), dim3( 1024 ), 0, st, L ); }
void launch_edge_write( const EdgeLaunch& L, hipStream_t st )
{ hipLaunchKernelGGL( k_edge_write, dim3( ( L.n + BLOCK - 1 ) / BLOCK ), dim3( BLOCK ), 0, st, L ); }

// ------------------------------------------------------------------------------------------
// Scene-coverage term  (arrangement_optimization.cpp:344-373, 1064-1106; intersect.h:97-109)
// One bit per voxel.  The scene bitmap is built once; an arrangement's score only needs the
// scene-active voxels its points hit, so a point whose voxel is not scene-active is dropped at
// once and the others race on atomicOr — the first to set a bit counts it.
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ int voxel_of( const VoxGrid& g, float x, float y, float z )
{
  const int cx = (int)floorf( ( x - g.ox ) * g.inv_voxel );     // intersect.h:101-103
  const int cy = (int)floorf( ( y - g.oy ) * g.inv_voxel );
  const int cz = (int)floorf( ( z - g.oz ) * g.inv_voxel );
  if( cx < 0 || cx >= g.x_res || cy < 0 || cy >= g.y_res || cz < 0 || cz >= g.z_res ) return -1;
  return cy * g.x_res * g.z_res + cz * g.x_res + cx;            // :108
}

__global__ __launch_bounds__( BLOCK ) void k_voxel_mark( VoxGrid g, const float* pos, const float* quality, float threshold, long long n, uint32_t* bits )
{
  const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if( i >= n ) return;
  if( quality && quality[i] < threshold ) return;               // arrangement_optimization.cpp:1073-1074
  const int c = voxel_of( g, pos[3*i], pos[3*i+1], pos[3*i+2] );
  if( c >= 0 ) atomicOr( bits + ( c >> 5 ), 1u << ( c & 31 ) );
}

__global__ __launch_bounds__( BLOCK ) void k_popcount( const uint32_t* bits, int n_words, int* out )
{
  int c = 0;
  for( int w = blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += gridDim.x * BLOCK ) c += __popc( bits[w] );
  for( int o = WAVE / 2; o > 0; o >>= 1 ) c += __shfl_down( c, o );
  if( ( threadIdx.x & ( WAVE - 1 ) ) == 0 && c ) atomicAdd( out, c );
}

__global__ __launch_bounds__( BLOCK ) void k_coverage( CoverageLaunch L )
{
  const CoveragePlacement& P = L.plc[blockIdx.y];
  int hit = 0;
  const uint32_t* scene = L.scene_bits;
  uint32_t* mine = L.arr_bits + (size_t)P.arrangement * L.n_words;
  for( int i = blockIdx.x * BLOCK + threadIdx.x; i < P.n; i += gridDim.x * BLOCK )
  {
    const float4 p = P.pos[i];
    float x, y, z;
    xform3( P.pose, p.x, p.y, p.z, 1.0f, x, y, z );            // msh_mat4_vec3_mul( pose, p, 1 ), :1101
    const int c = voxel_of( L.grid, x, y, z );
    if( c < 0 ) continue;
    const uint32_t m = 1u << ( c & 31 );
    if( !( scene[c >> 5] & m ) ) continue;                      // only cells with scn_cell > 0 can agree (:363)
    if( !( atomicOr( mine + ( c >> 5 ), m ) & m ) ) ++hit;
  }
  for( int o = WAVE / 2; o > 0; o >>= 1 ) hit += __shfl_down( hit, o );
  if( ( threadIdx.x & ( WAVE - 1 ) ) == 0 && hit ) atomicAdd( L.agree + P.arrangement, hit );
}

void launch_voxel_mark( const VoxGrid& g, const float* pos, const float* quality, float threshold, long long n, uint32_t* bits, hipStream_t st )
{
  if( n > 0 ) hipLaunchKernelGGL( k_voxel_mark, dim3( (unsigned)( ( n + BLOCK - 1 ) / BLOCK ) ), dim3( BLOCK ), 0, st, g, pos, quality, threshold, n, bits );
}
void launch_popcount( const uint32_t* bits, int n_words, int* out, hipStream_t st )
{
  hipLaunchKernelGGL( k_popcount, dim3( std::max( 1, std::min( 256, ( n_words + BLOCK - 1 ) / BLOCK ) ) ), dim3( BLOCK ), 0, st, bits, n_words, out );
}
void launch_coverage( const CoverageLaunch& L, hipStream_t st )
{
  if( L.n_plc <= 0 || L.max_pts <= 0 ) return;
  const int bx = std::max( 1, std::min( 64, ( L.max_pts + BLOCK - 1 ) / BLOCK ) );
  hipLaunchKernelGGL( k_coverage, dim3( bx, L.n_plc ), dim3( BLOCK ), 0, st, L );
}

} // namespace rs
