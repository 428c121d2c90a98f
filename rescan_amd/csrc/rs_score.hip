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

// one query's term of the sum (0 without a match: the reference `continue`s, :148)
__device__ __forceinline__ double score_of( const ScoreLaunch& L, bool found, float dot, float d2 )
{
  double s = 0.0;
  if( found )
  {
    const double angle = acos( (double)dot );                                     // :140
    const double normals_compat = exp( -( angle * angle ) / ( 2.0 * 0.5 * 0.5 ) ); // :149
    const double dist_compat = exp( -(double)d2 / ( 2.0 * L.sigma * L.sigma ) );   // :150, :36-40
    s = 0.05 * normals_compat + ( 1.0 - 0.05 ) * dist_compat;                      // :102-103,151
  }
  return s;
}

__device__ __forceinline__ void score_emit( const ScoreLaunch& L, int pose, int tile, bool active, int lane, const Match& m )
{
  double s = score_of( L, active && m.found, m.dot, m.d2 );
  s = wave_sum( s );
  if( lane == 0 ) L.part[(size_t)pose * L.obj.n_tiles + tile] = s;
}

// Object-space launch: small batches, brute-layout scenes.  grid = tiles x poses, one tile (wave) per workgroup.
// (7 waves per SIMD, 71 VGPRs, no scratch.  One tile per workgroup: a workgroup's slots are released when its LAST wave ends, and the four tiles
//  of a four-wave workgroup do not take equally long — 1.21 -> 1.09 ms for the 256-pose batch alone in round 2.)
#ifndef RS_SCORE_OCC
#define RS_SCORE_OCC 7
#endif
__global__ __launch_bounds__( WAVE, RS_SCORE_OCC ) void k_score( ScoreLaunch L )
{
  __shared__ WaveLds lds;
  const int pose = blockIdx.y;
  const int lane = threadIdx.x;
  EvalScope eval_scope( L.scene.evals, lds, lane );
  const int tile = blockIdx.x;
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
  int slog[16] = { 0 };
  const Match m = tile_search<true>( L.scene, active, qx, qy, qz, nx, ny, nz, (float)L.sigma, L.radius_sq, L.gate_tmin, L.K,
                                     lds, lane, L.solo_stages, &handoff, ( RS_DBG && L.hist ) ? slog : nullptr, no_match() );
  if( RS_DBG && L.hist && lane == 0 )
  {
    // diagnostic builds (RS_HIP_SCORE_HIST): candidates streamed by shell s for u unsettled lanes -> hist[s][u]; by the rank pass for u
    // lanes that need their rank -> hist[5][u]
    for( int sh = 0; sh < 5; ++sh ) if( slog[5 + 2 * sh] > 0 ) atomicAdd( L.hist + sh * 65 + min( slog[4 + 2 * sh], 64 ), (unsigned long long)slog[5 + 2 * sh] );
    if( slog[2] > 0 ) atomicAdd( L.hist + 5 * 65 + min( slog[15], 64 ), (unsigned long long)slog[2] );
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

// ------------------------------------------------------------------------------------------
// Scene-space batches.
//
// k_score gives a wave 64 OBJECT neighbours under one pose: their box is up to 0.25 m across wherever the pose puts it, the wave
// streams everything within the radius of that box past all 64 lanes (~840 candidates where ~160 lie in one query's ball), and a
// lane in empty space rides along through every shell its neighbours need.  Large batches are therefore re-tiled in the SCENE's
// frame: every (pose, point) query is transformed once and keyed by the scene-aligned block ("parent", edge ~ the radius) it falls
// in, the keys are radix-sorted, and a wave takes 64 consecutive entries — queries of one block, whatever pose they come from.
// Their candidate set is the same few cells, staged once; queries outside the scene's box (grown by the radius) sort to the end
// and retire without a search.  The search itself is tile_search unchanged — a lane's result never depended on its companions —
// and the sum is taken in k_score's order (per object tile over the lanes, then over the tiles), so the scores keep their bits.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t morton_2bit( int x, int y, int z )
{
  return (uint32_t)( ( x & 1 ) | ( ( y & 1 ) << 1 ) | ( ( z & 1 ) << 2 ) | ( ( x & 2 ) << 2 ) | ( ( y & 2 ) << 3 ) | ( ( z & 2 ) << 4 ) );
}

// One atomic per DISTINCT key of the wave: consecutive threads are neighbouring object points under one pose — they land in the same scene
// block, i.e. on a handful of counters (64 same-address atomics serialise at the memory side: the first form of this was slower than
// the radix sort it replaced).  A loop over the wave's distinct keys only forms the groups — lane masks, no memory operation — then the
// groups' leaders add their sizes in ONE atomic instruction (a returning atomic inside the loop made it a chain of 10-30 round trips
// per wave), and every lane takes its leader's result + its rank in the group.  COUNT_ONLY: nothing comes back.
template <bool COUNT_ONLY>
__device__ __forceinline__ uint32_t wave_key_slots( uint32_t* counters, uint32_t key, bool active )
{
  const int lane = (int)( threadIdx.x & ( WAVE - 1 ) );
  lanemask pending = RS_BALLOT( active );
  uint32_t rank = 0u, size = 0u;
  int leader = lane;
  while( pending != 0ull )
  {
    const int first = __builtin_ctzll( pending );
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane( (int)key, first );
    const lanemask same = RS_BALLOT( active && key == k0 );
    if( active && key == k0 ) { rank = (uint32_t)__popcll( same & ( ( 1ull << lane ) - 1ull ) ); size = (uint32_t)__popcll( same ); leader = first; }
    pending &= ~same;
  }
  if( COUNT_ONLY ) { if( active && lane == leader ) atomicAdd( counters + key, size ); return 0u; }
  uint32_t base = 0u;
  if( active && lane == leader ) base = atomicAdd( counters + key, size );
  base = (uint32_t)__shfl( (int)base, leader );
  return base + rank;
}

__global__ __launch_bounds__( BLOCK ) void k_score_keys( ScoreLaunch L )
{
  const int pose = blockIdx.y;
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if( i >= L.obj.n ) return;
  Xform X;
#pragma unroll
  for( int k = 0; k < 16; ++k ) X.m[k] = L.poses[pose * 16 + k];
  float qx, qy, qz, nx, ny, nz;
  score_query( L, X, i, true, qx, qy, qz, nx, ny, nz );
  // (the key only groups: what a query matches is decided by tile_search from its coordinates)
  const bool in = ( qx >= L.sq_lox ) & ( qx <= L.sq_hix ) & ( qy >= L.sq_loy ) & ( qy <= L.sq_hiy ) & ( qz >= L.sq_loz ) & ( qz <= L.sq_hiz );
  uint32_t key = (uint32_t)L.sq_n_parents << L.sq_fine_bits;
  if( in )
  {
    const int ix = min( max( (int)( ( qx - L.sq_ox ) * L.sq_inv_fine ), 0 ), 4 * L.sq_dpx - 1 );
    const int iy = min( max( (int)( ( qy - L.sq_oy ) * L.sq_inv_fine ), 0 ), 4 * L.sq_dpy - 1 );
    const int iz = min( max( (int)( ( qz - L.sq_oz ) * L.sq_inv_fine ), 0 ), 4 * L.sq_dpz - 1 );
    // low bits: the octant of the parent the query falls in, then the dominant axis and sign of its TRANSFORMED normal — a wave's
    // queries then also face the same way: what they can match, and how far away, is much the same for all 64 (16.4 M candidates
    // staged instead of 21.1 M)
    uint32_t fine = morton_2bit( ix & 3, iy & 3, iz & 3 ) >> 3;
    if( L.sq_nbin )
    {
      const float ax = fabsf( nx ), ay = fabsf( ny ), az = fabsf( nz );
      const uint32_t nb = ax >= ay && ax >= az ? ( nx < 0.0f ? 1u : 0u ) : ay >= az ? ( ny < 0.0f ? 3u : 2u ) : ( nz < 0.0f ? 5u : 4u );
      fine = ( fine << 3 ) | nb;
    }
    else fine = morton_2bit( ix & 3, iy & 3, iz & 3 );      // (the quarter-parent sub-cell instead)
    key = ( (uint32_t)( ( ( iz >> 2 ) * L.sq_dpy + ( iy >> 2 ) ) * L.sq_dpx + ( ix >> 2 ) ) << L.sq_fine_bits ) | fine;
  }
  const uint32_t j = (uint32_t)pose * (uint32_t)L.obj.n + (uint32_t)i;
  L.sq_key_a[j] = key;
  if( L.sq_hist ) (void)wave_key_slots<true>( L.sq_hist, key, true ); else L.sq_val_a[j] = j;
}

// Ordering by counting (round 6).  The keys only GROUP the queries (a lane's result never depended on its companions), so within a key any
// order will do: k_score_keys counts the keys (2.6 M adds onto ~10^5 occupied counters), one exclusive scan turns counts into offsets, and
// every query takes the next free place of its key — two passes over 10 MB of keys instead of a stable radix sort's three over 21 MB
// of pairs (0.15 ms of the 0.60 ms batch with its fills).  Which of two equal keys comes first now varies from run to run; the scores
// cannot (tests/test_gpu_parity.py::test_scores_scene_space_route_vs_golden: both routes, bit for bit).
__global__ __launch_bounds__( BLOCK ) void k_score_scatter( ScoreLaunch L, uint32_t items )
{
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  const bool have = j < items;
  const uint32_t key = have ? L.sq_key_a[j] : 0u;
  const uint32_t pos = wave_key_slots<false>( L.sq_hist, key, have );
  if( have ) { L.sq_key_b[pos] = key; L.sq_val_b[pos] = j; }
}

#ifndef RS_SCORE_SCENE_OCC
#define RS_SCORE_SCENE_OCC 7
#endif
template <bool CULL>
__global__ __launch_bounds__( WAVE, RS_SCORE_SCENE_OCC ) void k_score_scene( ScoreLaunch L, uint32_t items )
{
  __shared__ WaveLds lds;
  const int lane = threadIdx.x;
  EvalScope eval_scope( L.scene.evals, lds, lane );
  const uint32_t j = blockIdx.x * WAVE + lane;
  const bool have = j < items;
  const uint32_t key = have ? L.sq_key_b[j] : 0xffffffffu;
  const uint32_t val = have ? L.sq_val_b[j] : 0u;
  const uint32_t par = key >> L.sq_fine_bits;
  const bool valid = have && par < (uint32_t)L.sq_n_parents;
  float qx = 0.0f, qy = 0.0f, qz = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f;
  lanemask pending = RS_BALLOT( valid );
  if( pending != 0ull )
  {
    const uint32_t pose = val / (uint32_t)L.obj.n;
    const int i = (int)( val - pose * (uint32_t)L.obj.n );
    Xform X;
#pragma unroll
    for( int k = 0; k < 16; ++k ) X.m[k] = valid ? L.poses[pose * 16 + k] : 0.0f;
    score_query( L, X, i, valid, qx, qy, qz, nx, ny, nz );      // the same float operations as k_score's: same query, bit for bit
  }
  bool found = false; float r_dot = 0.0f, r_d2 = 0.0f;
  while( pending != 0ull )
  {
    // the entries of the first unserved lane's parent (sorted: a run of lanes)
    const int first = __builtin_ctzll( pending );
    const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane( (int)par, first );
    const bool act = valid && par == pk;
    int slog[16] = { 0 };
    if( RS_DBG ) slog[14] = __popcll( RS_BALLOT( act ) );
    const Match m = tile_search<true, false, false, CULL>( L.scene, act, qx, qy, qz, nx, ny, nz, (float)L.sigma, L.radius_sq, L.gate_tmin, L.K,
                                                           lds, lane, 0x7fffffff, nullptr, ( RS_DBG && L.hist ) ? slog : nullptr, no_match() );
    if( RS_DBG && L.hist && lane == 0 )
    {
      // diagnostic builds (RS_HIP_SCORE_HIST), as in k_score; row 6: searches by the number of lanes they serve
      for( int sh = 0; sh < 5; ++sh ) if( slog[5 + 2 * sh] > 0 ) atomicAdd( L.hist + sh * 65 + min( slog[4 + 2 * sh], 64 ), (unsigned long long)slog[5 + 2 * sh] );
      if( slog[2] > 0 ) atomicAdd( L.hist + 5 * 65 + min( slog[15], 64 ), (unsigned long long)slog[2] );
      atomicAdd( L.hist + 6 * 65 + slog[14], 1ull );
    }
    found = act ? m.found : found; r_dot = act ? m.dot : r_dot; r_d2 = act ? m.d2 : r_d2;
    pending &= ~RS_BALLOT( act );
  }
  if( have ) L.sq_pq[val] = score_of( L, found, r_dot, r_d2 );
}

// per pose: the per-point terms summed as k_score + k_score_final sum them — over the lanes of every object tile (wave_sum), then over
// the tiles in k_score_final's order, / n, narrowed to float (:156-157)
#define SCORE_GATHER_THREADS 1024
__global__ __launch_bounds__( SCORE_GATHER_THREADS ) void k_score_gather( ScoreLaunch L )
{
  __shared__ double red[BLOCK];
  const int pose = blockIdx.x;
  const int lane = threadIdx.x & ( WAVE - 1 ), wib = threadIdx.x / WAVE;
  const int n_tiles = L.obj.n_tiles;
  double* part = L.part + (size_t)pose * n_tiles;
  const double* pq = L.sq_pq + (size_t)pose * L.obj.n;
  for( int t = wib; t < n_tiles; t += SCORE_GATHER_THREADS / WAVE )
  {
    const int i = (int)L.obj.tiles[t] + lane;
    double s = i < (int)L.obj.tiles[t + 1] ? pq[i] : 0.0;
    s = wave_sum( s );
    if( lane == 0 ) part[t] = s;
  }
  __syncthreads();
  if( threadIdx.x < BLOCK )      // (k_score_final's order: BLOCK strided partial sums, then the tree)
  {
    double a = 0.0;
    for( int t = threadIdx.x; t < n_tiles; t += BLOCK ) a += part[t];
    red[threadIdx.x] = a;
  }
  __syncthreads();
  for( int s = BLOCK / 2; s > 0; s >>= 1 ) { if( (int)threadIdx.x < s ) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if( threadIdx.x == 0 ) L.scores[pose] = (float)( red[0] / (double)L.obj.n );
}

void launch_score( const ScoreLaunch& L, hipStream_t st )
{
  if( L.sq_key_a )
  {
    const uint32_t items = (uint32_t)L.n_poses * (uint32_t)L.obj.n;
    if( L.sq_hist )
    {
      const size_t n_bins = ( (size_t)1 << L.sq_bits ) + 1;
      (void)hipMemsetAsync( L.sq_hist, 0, n_bins * 4, st );
      hipLaunchKernelGGL( k_score_keys, dim3( ( L.obj.n + BLOCK - 1 ) / BLOCK, L.n_poses ), dim3( BLOCK ), 0, st, L );
      (void)build_exclusive_scan( L.sq_tmp, L.sq_tmp_bytes, L.sq_hist, L.sq_hist, n_bins, st );      // (in place: counts -> offsets)
      hipLaunchKernelGGL( k_score_scatter, dim3( ( items + BLOCK - 1 ) / BLOCK ), dim3( BLOCK ), 0, st, L, items );
    }
    else
    {
      hipLaunchKernelGGL( k_score_keys, dim3( ( L.obj.n + BLOCK - 1 ) / BLOCK, L.n_poses ), dim3( BLOCK ), 0, st, L );
      (void)build_sort_pairs( L.sq_tmp, L.sq_tmp_bytes, L.sq_key_a, L.sq_key_b, L.sq_val_a, L.sq_val_b, (int)items, L.sq_bits, st );
    }
    const dim3 sgrid( ( items + WAVE - 1 ) / WAVE );
    if( L.sq_cull ) hipLaunchKernelGGL( k_score_scene<true>, sgrid, dim3( WAVE ), 0, st, L, items );
    else            hipLaunchKernelGGL( k_score_scene<false>, sgrid, dim3( WAVE ), 0, st, L, items );
    hipLaunchKernelGGL( k_score_gather, dim3( L.n_poses ), dim3( SCORE_GATHER_THREADS ), 0, st, L );
    return;
  }
  (void)hipMemsetAsync( L.queue_count, 0, sizeof(int), st );
  hipLaunchKernelGGL( k_score, dim3( L.obj.n_tiles, L.n_poses ), dim3( WAVE ), 0, st, L );
  long long items = (long long)L.obj.n_tiles * L.n_poses;
  hipLaunchKernelGGL( k_score_coop, dim3( items < 4096 ? (int)( items > 0 ? items : 1 ) : 4096 ), dim3( COOP_BLOCK ), 0, st, L );
  hipLaunchKernelGGL( k_score_final, dim3( L.n_poses ), dim3( BLOCK ), 0, st, L );
}

} // namespace rs
