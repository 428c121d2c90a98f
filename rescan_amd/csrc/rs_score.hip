// librescan_hip device code (gfx950, wave64) — the alignment-score batch (apps/pose_proposal/pose_proposal.cpp:93-158)
#include "rs_search.h"

namespace rs {

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

} // namespace rs
