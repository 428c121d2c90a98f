// Unit test of chain_prefix (rs_icp_estimate.hip: the composition of a run of chain records as three integer prefix scans): random
// records per lane — valid ones, identities, never-records — against the sequential composition on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I rescan_amd/csrc tools/micro/chain_scan_test.hip -o tools/micro/chain_scan_test.bin
// (History: the first form of this scan composed (lo, hi, D) triples with __builtin_amdgcn_update_dpp moves; the compiler folded
//  the moves into v_subrev_u32_dpp / v_add_u32_dpp ... bound_ctrl and every row's first lane came out "never" on gfx950 — this
//  test found it.  The scans are inline assembly now, like wave_scan.)
#include "../../rescan_amd/csrc/rs_icp_estimate.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace rs;
__global__ void k_test( const int* in, int* out )
{
  ChainFn f; f.lo = in[3 * threadIdx.x]; f.hi = in[3 * threadIdx.x + 1]; f.D = in[3 * threadIdx.x + 2];
  f = chain_prefix( f, threadIdx.x );
  out[3 * threadIdx.x] = f.lo; out[3 * threadIdx.x + 1] = f.hi; out[3 * threadIdx.x + 2] = f.D;
}
static ChainFn then_host( ChainFn f, ChainFn g )
{
  ChainFn h; h.lo = std::max( f.lo, g.lo - f.D ); h.hi = std::min( f.hi, g.hi - f.D ); h.D = f.D + g.D;
  const bool never = f.lo > f.hi || g.lo > g.hi || h.lo > h.hi;
  if( never ) { h.lo = CH_M_HI; h.hi = CH_M_LO; h.D = 0; }
  return h;
}
int main()
{
  int *din, *dout; hipMalloc( &din, 64 * 12 ); hipMalloc( &dout, 64 * 12 );
  int bad = 0;
  for( int trial = 0; trial < 400; ++trial )
  {
    std::vector<int> in( 192 ), out( 192 );
    srand( trial );
    for( int l = 0; l < 64; ++l )
    {
      const int kind = rand() % 8;
      if( kind == 0 || ( trial < 20 && l < trial ) ) { in[3*l] = CH_M_LO; in[3*l+1] = CH_M_HI; in[3*l+2] = 0; }                 // identity
      else if( kind == 1 && trial % 3 == 0 ) { in[3*l] = CH_M_HI; in[3*l+1] = CH_M_LO; in[3*l+2] = 0; }                       // never
      else { const int d = ( rand() % 2001 - 1000 ) * ( trial % 5 == 4 ? 4000 : 1 ); in[3*l] = CH_M_LO + 1 + std::max( 0, -d ); in[3*l+1] = CH_M_HI - 1 - std::max( 0, d ) - rand() % 100000; in[3*l+2] = d; }
    }
    hipMemcpy( din, in.data(), 768, hipMemcpyHostToDevice );
    hipLaunchKernelGGL( k_test, dim3( 1 ), dim3( 64 ), 0, 0, din, dout );
    hipMemcpy( out.data(), dout, 768, hipMemcpyDeviceToHost );
    ChainFn run; run.lo = CH_M_LO; run.hi = CH_M_HI; run.D = 0;
    for( int l = 0; l < 64; ++l )
    {
      ChainFn g; g.lo = in[3*l]; g.hi = in[3*l+1]; g.D = in[3*l+2];
      run = then_host( run, g );
      if( out[3*l] != run.lo || out[3*l+1] != run.hi || out[3*l+2] != run.D )
      { if( bad < 10 ) printf( "trial %d lane %d: device (%d %d %d) host (%d %d %d)\n", trial, l, out[3*l], out[3*l+1], out[3*l+2], run.lo, run.hi, run.D ); ++bad; }
    }
  }
  printf( "mismatches: %d\n", bad );
  return bad != 0;
}
