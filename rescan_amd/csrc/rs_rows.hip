// librescan_hip device code (gfx950, wave64) — label transfer, full rows, level builder, neighbourhood edges, coverage voxels
#include "rs_search.h"

namespace rs {

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
