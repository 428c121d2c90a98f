// librescan_hip device code (gfx950, wave64) — the ICP estimators: fp64 moments, reference-order chains, replay, grid chains (lib/rs/icp.h:136-148,210-298)
#include "rs_search.h"
#include "rs_icp.h"

namespace rs {

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
  // (an estimator that is not the reference's own order follows the reference's errors to ~1e-7; a stop test decided by less than
  //  that margin may fall the other way — one iteration more or less, 1e-4 in the pose.  Such a problem is flagged: rs_api.hip)
  if( !L.fixed_iters && L.iter_index > 5 && L.stop_guard > 0.0f && fabsf( delta - 1e-5f ) < L.stop_guard ) L.ticket[prob] = 1;
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
// (R: the problem's records — one source for the whole batch: L.rec + prob * n; a multi-source batch: L.rec + pt_off, see k_lane_chains_and_moments)
// (xrows: where to leave the seven addends of every point of the quarter block, row r at xrows + r * ( ( n + 3 ) & ~3 ) — the lane chains' input; null: not wanted)
__device__ __forceinline__ void chain_moments_block( const IcpLaunch& L, const ChainBufs& B, int prob, int qb, ChainMomLds& S, const float4* R, float* xrows = nullptr )
{
  // quarter block qb: segments [16 qb, 16 qb + 16)
  const float sd = chain_stats( L, prob, S.stat, ( qb == 0 && threadIdx.x == 0 ) ? L.res + (size_t)prob * ICP_NRES + ICP_NMOM : nullptr );
  ChainPar P; P.use_sd = sd > 0.000001; P.cut = 2.5f * sd; P.max_dist = L.radius;
  const int lane = threadIdx.x & ( WAVE - 1 ), wib = threadIdx.x / WAVE;

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
    if( xrows )
    {
      const int ns = ( L.src.n + 3 ) & ~3;               // (the row's padding beyond n: zeros)
#pragma unroll
      for( int r = 0; r < CH_ROWS; ++r ) if( i < ns ) xrows[(size_t)r * ns + i] = x[r];
    }
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
  chain_moments_block( L, B, prob, blockIdx.x, S, L.rec + (size_t)prob * L.src.n * REC_F4 );
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
  if( qb < B.n_blk * CH_QUARTERS ) chain_moments_block( L, B, prob, qb, S.m, L.rec + (size_t)prob * L.src.n * REC_F4 );
  else chain_guess_block( L, B, prob, qb - B.n_blk * CH_QUARTERS, WAVES_PER_BLOCK );
}

// ------------------------------------------------------------------------------------------
// Lane chains (round 6): the same seven sums for OBJECT-sized sources — every icp_align call site of the reference
// (apps/pose_proposal/main.cpp:195-197, lib/rs/rs_database.h:227-229, apps/segment_transfer/database_update.cpp:65-67) —
// one wave per chain, any number of problems side by side (grid.y = problem, each bound to its own source: icp_bind).
//
// The grid chains above pay five launches and a forecast machinery to spread ONE chain over the chip; a source of a few
// ten thousand points does not need that.  A wave takes 256 addends at a time, four consecutive ones per lane, and applies the
// same fact — inside a binade the sum is an integer sum — directly: every addend scaled by 1 / ulp( s ) and rounded to
// nearest-even is what it adds to the mantissa; the four of a lane are prefixed in registers, the lanes by one DPP scan; if every
// partial mantissa stays inside the binade (one grid step clear of its ends, as the records above) and no addend sits exactly
// half way between two grid points, the 256 additions ARE that integer addition.  Otherwise the stretch is cut at the first
// lane that does not fit: the lanes before it are taken by the integers, that lane's four addends are added in fp32 one
// after the other — the reference's own operation, so a change of binade, a tie, a sum that is zero, denormal or not finite
// need no case of their own — and the rest is tried again from the new value.  A sum that hovers (coordinates of both
// signs) makes little headway per attempt; then the fp32 stretch doubles, up to 16 lanes: the worst case is the plain
// sequential sum plus an attempt per 64 addends.  Nothing is given up and nothing is forecast: the result is the
// sequential sum for any input (tests/test_gpu_parity.py: test_lane_chains_are_the_sequential_sums).
//   k_lane_chains_and_moments   workgroups 0-1: the seven walks (a wave each); the rest: the fp64 moments of a quarter block of
//                               1 024 source points each (chain_moments_block, shared with the grid chains)
//   k_icp_update_wide           the rest of the iteration, centred on the chains' centroids (L.exact_centroids)
// What it is NOT: the reference's bits.  The 27 + 6 accumulators of the normal equations are summed in fp64 where the reference
// rounds after every addition — measured on every reference fixture of this size (profiles/r06/estimator_policy.txt): at most
// 3e-6 from the reference's pose, iteration counts equal; rs_hip_icp_reference_order_below() brings the bits back.
// ------------------------------------------------------------------------------------------
#define LC_PER_LANE 16
#define LC_SUPER ( LC_PER_LANE * WAVE )
#define LC_DEPTH 4

// Where problem `prob`'s seven addend rows live (floats): eight rows' worth per problem so that every row starts on a 16-byte
// boundary whatever the sizes of the problems before it; a row holds ( n + 3 ) & ~3 floats.
__device__ __forceinline__ size_t lane_rows_base( const IcpLaunch& L, int prob ) { return 8 * ( (size_t)L.pt_off + 4 * (size_t)prob ); }
__device__ __forceinline__ int lane_row_stride( int n ) { return ( n + 3 ) & ~3; }

// (unconditional loads — a quad beyond the row reads the row's last quad instead and is zeroed afterwards — so that the loads of the
//  stretches ahead stay countable: a load under a branch made the compiler wait for ALL loads in flight at every step)
__device__ __forceinline__ void lane_chain_load( const float* row, int ns, int base, int lane, float4 ( &X )[LC_PER_LANE / 4] )
{
#pragma unroll
  for( int q = 0; q < LC_PER_LANE / 4; ++q )
  {
    const int i = base + lane * LC_PER_LANE + 4 * q;
    X[q] = *(const float4*)( row + min( i, ns - 4 ) );     // (ns is a multiple of 4, >= 4: lane_row_stride)
  }
}
__device__ __forceinline__ float wave_scan_add_f32( float v ) { RS_DPP_PREFIX( "v_add_f32_dpp", v ); return v; }      // inclusive; lanes without a source keep their value

// The sequential fp32 sum of the n addends of `row` (the reference's order), by one wave.  dbg (diagnostics, may be null): addends
// added one by one, attempts made.
//
// The integers are kept as FLOATS: with S = |s| in [2^e, 2^(e+1)), u = ulp( S ) and C = 1.5 * 2^e (same binade, even mantissa),
// RN( C + x ) - C  is x rounded to the grid of u, ties to the even grid point, for every |x| < 2^(e-1) — the addition rounds for us, no
// scaling, no conversion — and sums of such multiples of u below 2^(e+1) are exact in fp32.  So a lane's sixteen addends are
// rounded (two additions each), prefixed in registers, the lanes' totals by one DPP scan of float additions, and the partial sums
// S + prefix compared with the binade's ends as floats.  Every quantity that decides something for a lane is a sum over a
// contiguous range of addends BEFORE the first lane that does not fit, hence a multiple of u below 2^(e+1), hence exact; what lies
// beyond that lane may have rounded and is not looked at.
__device__ __forceinline__ float lane_chain_walk( const float* row, int n, int lane, int* dbg )
{
  float s = 0.0f;                                      // wave-uniform throughout
  int n_seq = 0, n_try = 0;
  const int ns = lane_row_stride( n );
  // Two groups of LC_DEPTH stretches in registers: the one being walked and the next, all of whose loads are issued before the walk of
  // the current group begins — the rows were written a launch ago by other CUs (another XCD's L2: they come from memory, 1-2 us
  // away), a group is walked in about that time.
  float4 X[LC_DEPTH][LC_PER_LANE / 4], Y[LC_DEPTH][LC_PER_LANE / 4];
#pragma unroll
  for( int k = 0; k < LC_DEPTH; ++k ) lane_chain_load( row, ns, k * LC_SUPER, lane, X[k] );
  for( int base0 = 0; base0 < n; base0 += LC_DEPTH * LC_SUPER )
  {
  const bool more = base0 + LC_DEPTH * LC_SUPER < n;
  if( more )
  {
#pragma unroll
    for( int k = 0; k < LC_DEPTH; ++k ) lane_chain_load( row, ns, base0 + ( LC_DEPTH + k ) * LC_SUPER, lane, Y[k] );
  }
#pragma unroll
  for( int k = 0; k < LC_DEPTH; ++k )
  {
    const int base = base0 + k * LC_SUPER;
    if( base >= n ) break;
    float x[LC_PER_LANE];
    {
      const int i0 = base + lane * LC_PER_LANE;
#pragma unroll
      for( int q = 0; q < LC_PER_LANE / 4; ++q )
      {
        const bool in = i0 + 4 * q < ns;
        x[4 * q] = in ? X[k][q].x : 0.0f; x[4 * q + 1] = in ? X[k][q].y : 0.0f; x[4 * q + 2] = in ? X[k][q].z : 0.0f; x[4 * q + 3] = in ? X[k][q].w : 0.0f;
      }
    }
    // the lane's addends' magnitudes, summed: decides whether the lane can go by the grid at all (below), and is inf or NaN when an
    // addend is not finite (or the sum overflows: beyond any binade's reach anyway)
    float mag = 0.0f;
#pragma unroll
    for( int j = 0; j < LC_PER_LANE; ++j ) mag = mag + fabsf( x[j] );
    int start = 0, seq = 1;
    while( start < WAVE )
    {
      const uint32_t sb = (uint32_t)uni( (int)__float_as_uint( s ) );      // (every lane holds the same value: branch on the scalar unit)
      const int E = (int)( ( sb >> 23 ) & 255u );
      int b = start;                                   // first lane the grid does not carry
      if( E >= 64 && E <= 253 )                        // (below 2^-63 — where an addend could be a denormal worth half a grid step — one by one)
      {
        ++n_try;
        const float sgn = ( sb >> 31 ) ? -1.0f : 1.0f;                                 // |s| grows by -x when s < 0
        const float S = __uint_as_float( sb & 0x7fffffffu );
        const float C = __uint_as_float( ( (uint32_t)E << 23 ) | 0x400000u );          // 1.5 * 2^e
        const float quarter = __uint_as_float( (uint32_t)( E - 1 ) << 23 );            // 2^(e-1): the rounding trick's range
        const float u = __uint_as_float( (uint32_t)( E - 23 ) << 23 ), uh = __uint_as_float( (uint32_t)( E - 24 ) << 23 );      // ulp( S ), half of it
        const float lo_end = __uint_as_float( (uint32_t)E << 23 ) + u, hi_end = __uint_as_float( (uint32_t)( E + 1 ) << 23 ) - u;   // the binade, a grid step clear of its ends
        float p[LC_PER_LANE], run = 0.0f, half = 0.0f;      // half: the largest | x - rounded x | of the lane; u / 2 = an addend exactly half way between two grid points
#pragma unroll
        for( int j = 0; j < LC_PER_LANE; ++j )
        {
          const float r = __builtin_fmaf( x[j], sgn, C ) - C;                          // x (signed for |s|) on the grid, ties to even
          half = fmaxf( half, fabsf( __builtin_fmaf( x[j], sgn, -r ) ) );              // (exact: x and r differ by at most u / 2)
          run = run + r; p[j] = run;
        }
        const float incl = wave_scan_add_f32( run ), excl = incl - run;
        float lo = p[0], hi = p[0];
#pragma unroll
        for( int j = 1; j < LC_PER_LANE; ++j ) { lo = fminf( lo, p[j] ); hi = fmaxf( hi, p[j] ); }
        // (written so that a NaN anywhere — a lane after one that does not fit — reads as "does not fit")
        const bool fits = mag < quarter && half < uh && S + ( excl + lo ) >= lo_end && S + ( excl + hi ) <= hi_end;
        const unsigned long long badm = __builtin_amdgcn_ballot_w64( !fits && lane >= start );
        b = badm ? (int)__builtin_ctzll( badm ) : WAVE;
        if( b > start )
        {
          const float adv = rl( b < WAVE ? excl : incl, b < WAVE ? b : WAVE - 1 );
          s = sgn * ( S + adv );
        }
      }
      else if( ( sb << 1 ) == 0u )
      {
        // a sum of zero stays zero through addends that are all zero (the unmatched points a source may well begin with)
        const unsigned long long nzm = __builtin_amdgcn_ballot_w64( !( mag == 0.0f ) && lane >= start );
        b = nzm ? (int)__builtin_ctzll( nzm ) : WAVE;
      }
      if( b >= WAVE ) break;
      // lanes [b, e): their addends one after the other in fp32, as the reference adds them
      seq = ( b - start >= 2 ) ? 1 : min( 2 * seq, 4 );
      const int e = min( b + seq, WAVE );
      for( int l = b; l < e; ++l )
      {
#pragma unroll
        for( int j = 0; j < LC_PER_LANE; ++j ) s = s + rl( x[j], l );
      }
      n_seq += ( e - b ) * LC_PER_LANE;
      start = e;
      // the lanes done add nothing from here on (their prefix in the next attempt's scan is zero)
#pragma unroll
      for( int j = 0; j < LC_PER_LANE; ++j ) x[j] = lane < start ? 0.0f : x[j];
      mag = lane < start ? 0.0f : mag;
    }
  }
  if( more )
  {
#pragma unroll
    for( int k = 0; k < LC_DEPTH; ++k )
#pragma unroll
      for( int q = 0; q < LC_PER_LANE / 4; ++q ) X[k][q] = Y[k][q];
  }
  }
  if( dbg ) { dbg[0] = n_seq; dbg[1] = n_try; }
  return s;
}

// Launch 1, every CU: the fp64 moments of a quarter block (1 024 source points) per workgroup, and the seven addends of its points
// into the rows the walks read (B.addends).
__global__ __launch_bounds__( BLOCK ) void k_lane_moments_and_addends( IcpLaunch L, ChainBufs B )
{
  RS_CHAIN_SETPRIO();
  __shared__ ChainMomLds S;
  const int prob = blockIdx.y;
  if( L.active[prob] == 0 ) return;
  icp_bind( L, prob );
  ChainBufs Bq = B; Bq.n_seg = ( L.src.n + CH_SEG - 1 ) / CH_SEG; Bq.refresh = 0;
  chain_moments_block( L, Bq, prob, (int)blockIdx.x, S, L.rec + (size_t)L.pt_off * REC_F4, B.addends + lane_rows_base( L, prob ) );
}

// Launch 2: ONE workgroup per problem — seven waves walk the seven chains (from the addend rows), the eighth sums the moments' partials
// (k_icp_update_wide's work, in the order a 256-thread workgroup per moment added them: thread t takes partials t, t + 256, ...; a wave
// tree per 64; the four waves' sums in turn), a barrier, and the rest of the iteration (icp.h:253-295,455-493) centred on the chains'
// centroids.  (Until late in round 6 this was 35 + 2 workgroups per problem and a ticket for "who finishes last": every one of them a
// release fence, i.e. an L2 write-back — 19 000 of them per launch at 512 problems, 1.19 ms for walks that take 50 us; a barrier
// inside one workgroup needs none.)
constexpr int LW_WAVES = 8;
static_assert( CH_ROWS < LW_WAVES, "a wave per chain and one for the moments" );
__global__ __launch_bounds__( LW_WAVES * WAVE ) void k_lane_walk_and_update( IcpLaunch L, ChainBufs B )
{
  RS_CHAIN_SETPRIO();
  const int prob = blockIdx.x;
  if( L.active[prob] == 0 ) return;
  icp_bind( L, prob );
  const int lane = threadIdx.x & ( WAVE - 1 ), wib = uni( (int)threadIdx.x / WAVE );
  if( wib < CH_ROWS )
  {
    const int row = wib;
    int d[2] = { 0, 0 };
    const unsigned long long t0 = B.dbg ? wall_clock64() : 0ull;
    const float s = lane_chain_walk( B.addends + lane_rows_base( L, prob ) + (size_t)row * lane_row_stride( L.src.n ), L.src.n, lane, d );
    if( lane == 0 )
    {
      B.totals[( (size_t)prob * 3 + 1 ) * ICP_NMOM + row] = (double)s;
      if( B.resolved ) atomicAdd( B.resolved + prob, d[0] );
      // (RS_HIP_LANE_DEBUG) per (problem, chain): ticks of 10 ns the walk took, attempts, addends added one by one — of the last iteration
      if( B.dbg ) { int* o = B.dbg + ( (size_t)prob * CH_ROWS + row ) * 4; o[0] = (int)( wall_clock64() - t0 ); o[1] = d[1]; o[2] = d[0]; o[3] = L.src.n; }
    }
  }
  else if( wib == CH_ROWS )
  {
    for( int k = 0; k < ICP_NMOM; ++k )
    {
      const double* in = L.mom_part + ( (size_t)prob * ICP_NMOM + k ) * L.n_mom_blocks;
      double r[WAVES_PER_BLOCK];
#pragma unroll
      for( int w = 0; w < WAVES_PER_BLOCK; ++w )
      {
        double v = 0.0;
        for( int b = w * WAVE + lane; b < L.n_mom_blocks; b += BLOCK ) v += in[b];
        r[w] = wave_sum( v );
      }
      double t = 0.0;
#pragma unroll
      for( int w = 0; w < WAVES_PER_BLOCK; ++w ) t += r[w];
      if( lane == 0 ) L.res[(size_t)prob * ICP_NRES + k] = t;
    }
  }
  __syncthreads();                                       // (the sums and totals above: this workgroup's own global writes)
  icp_update_tail( L, prob );
}

void launch_icp_lane_chains( const IcpLaunch& L, const ChainBufs& B, hipStream_t st )
{
  hipLaunchKernelGGL( k_lane_moments_and_addends, dim3( L.n_mom_blocks, L.n_prob ), dim3( BLOCK ), 0, st, L, B );      // (L.n_mom_blocks == ceil( max_n / 1024 ))
  hipLaunchKernelGGL( k_lane_walk_and_update, dim3( L.n_prob ), dim3( LW_WAVES * WAVE ), 0, st, L, B );
}

// The step WITHOUT the chains: the fp64 moments from the searches' records, centred on their own fp64 centroids (icp_solve without the
// reference's) — what an early iteration of a scan-sized call runs (rs_api.hip: g_early_plain).  L.exact_centroids must be 0.
void launch_icp_plain_from_records( const IcpLaunch& L, const ChainBufs& B, hipStream_t st )
{
  ChainBufs Bq = B; Bq.refresh = 0;
  hipLaunchKernelGGL( k_chain_moments, dim3( B.n_blk * CH_QUARTERS, L.n_prob ), dim3( BLOCK ), 0, st, L, Bq );
  hipLaunchKernelGGL( k_icp_update_wide, dim3( ICP_NMOM, L.n_prob ), dim3( BLOCK ), 0, st, L, B.done );
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

} // namespace rs
