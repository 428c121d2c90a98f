#include "../../rescan_amd/csrc/rs_kernels.hip"
#include <cstdio>
#include <vector>
using namespace rs;
__global__ void k_raw( const int* in, int* out )
{
  // raw moves: row_shr:1 with old = 111, bound_ctrl false / true
  const int v = in[threadIdx.x];
  out[threadIdx.x] = __builtin_amdgcn_update_dpp( 111, v, RS_DPP_ROW_SHR( 1 ), 0xf, 0xf, false );
  out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp( 111, v, RS_DPP_ROW_SHR( 1 ), 0xf, 0xf, true );
  out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp( 111, v, RS_DPP_BCAST15, 0xa, 0xf, false );
  out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp( 111, v, RS_DPP_BCAST31, 0xc, 0xf, false );
}
template <int STEPS>
__global__ void k_steps( const int* in, int* out )
{
  ChainFn f; f.lo = in[3 * threadIdx.x]; f.hi = in[3 * threadIdx.x + 1]; f.D = in[3 * threadIdx.x + 2];
  const ChainFn id = chain_identity();
  if( STEPS >= 1 ) f = chain_then( chain_dpp<RS_DPP_ROW_SHR( 1 ), 0xf>( id, f ), f );
  if( STEPS >= 2 ) f = chain_then( chain_dpp<RS_DPP_ROW_SHR( 2 ), 0xf>( id, f ), f );
  out[3 * threadIdx.x] = f.lo; out[3 * threadIdx.x + 1] = f.hi; out[3 * threadIdx.x + 2] = f.D;
}
int main()
{
  int *din, *dout; hipMalloc( &din, 64 * 12 ); hipMalloc( &dout, 256 * 4 );
  std::vector<int> in( 192 ), out( 256 );
  for( int l = 0; l < 64; ++l ) in[l] = 1000 + l;
  hipMemcpy( din, in.data(), 256, hipMemcpyHostToDevice );
  hipLaunchKernelGGL( k_raw, dim3( 1 ), dim3( 64 ), 0, 0, din, dout );
  hipMemcpy( out.data(), dout, 1024, hipMemcpyDeviceToHost );
  for( int k = 0; k < 4; ++k ) { printf( "raw %d:", k ); for( int l = 0; l < 64; ++l ) printf( " %d", out[64 * k + l] ); printf( "\n" ); }
  for( int l = 0; l < 64; ++l ) { in[3*l] = CH_M_LO + 1 + l; in[3*l+1] = CH_M_HI - 1 - l; in[3*l+2] = l + 1; }
  hipMemcpy( din, in.data(), 768, hipMemcpyHostToDevice );
  hipLaunchKernelGGL( k_steps<1>, dim3( 1 ), dim3( 64 ), 0, 0, din, dout );
  hipMemcpy( out.data(), dout, 768, hipMemcpyDeviceToHost );
  printf( "one step:" ); for( int l = 0; l < 20; ++l ) printf( " (%d %d %d)", out[3*l] - CH_M_LO, CH_M_HI - out[3*l+1], out[3*l+2] ); printf( "\n" );
  hipLaunchKernelGGL( k_steps<2>, dim3( 1 ), dim3( 64 ), 0, 0, din, dout );
  hipMemcpy( out.data(), dout, 768, hipMemcpyDeviceToHost );
  printf( "two steps:" ); for( int l = 0; l < 20; ++l ) printf( " (%d %d %d)", out[3*l] - CH_M_LO, CH_M_HI - out[3*l+1], out[3*l+2] ); printf( "\n" );
  return 0;
}
