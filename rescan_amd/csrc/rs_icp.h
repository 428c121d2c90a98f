// Shared by the two ICP translation units: a kernel's view of its problem, the source point in the target's frame, and what the
// workgroup that ends an iteration resets.
#pragma once
#include "rs_search.h"

namespace rs {

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

} // namespace rs
