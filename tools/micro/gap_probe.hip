// What separates two dependent kernels of one stream, and does a HIP graph change it?  A chain shaped like an ICP iteration — a wide
// kernel (~60 us), a narrower one (~35 us), a 512-block reduction (~20 us), a one-workgroup kernel (~14 us) — x 10, timed on the
// device with events: launched directly, and replayed from a captured graph.   hipcc --offload-arch=gfx950 -O2 gap_probe.hip -o gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void spin( float* p, int ticks ) { const long long t0 = wall_clock64(); while( wall_clock64() - t0 < ticks ) { } if( p && threadIdx.x == 0 && blockIdx.x == 0 ) p[0] += 1.0f; }
#define CK( x ) do { hipError_t e = ( x ); if( e != hipSuccess ) { printf( "%s: %s\n", #x, hipGetErrorString( e ) ); return 1; } } while( 0 )
static void chain( hipStream_t s, float* d, int iters )
{
  for( int i = 0; i < iters; ++i )
  {
    hipLaunchKernelGGL( spin, dim3( 4096 ), dim3( 64 ), 0, s, d, 1200 );     // 100 MHz clock: 12 us per wave, several rounds
    hipLaunchKernelGGL( spin, dim3( 300 ), dim3( 512 ), 0, s, d, 3500 );
    hipLaunchKernelGGL( spin, dim3( 512 ), dim3( 256 ), 0, s, d, 1900 );
    hipLaunchKernelGGL( spin, dim3( 1 ), dim3( 1024 ), 0, s, d, 1300 );
  }
}
int main()
{
  float* d; CK( hipMalloc( &d, 64 ) ); CK( hipMemset( d, 0, 64 ) );
  hipStream_t s; CK( hipStreamCreateWithFlags( &s, hipStreamNonBlocking ) );
  hipEvent_t a, b; CK( hipEventCreate( &a ) ); CK( hipEventCreate( &b ) );
  const int iters = 10;
  std::vector<float> direct, graphed;
  for( int r = 0; r < 12; ++r )
  {
    CK( hipEventRecord( a, s ) ); chain( s, d, iters ); CK( hipEventRecord( b, s ) ); CK( hipStreamSynchronize( s ) );
    float ms; CK( hipEventElapsedTime( &ms, a, b ) ); direct.push_back( ms );
  }
  hipGraph_t g; hipGraphExec_t ge;
  CK( hipStreamBeginCapture( s, hipStreamCaptureModeThreadLocal ) ); chain( s, d, iters ); CK( hipStreamEndCapture( s, &g ) );
  CK( hipGraphInstantiate( &ge, g, nullptr, nullptr, 0 ) );
  for( int r = 0; r < 12; ++r )
  {
    CK( hipEventRecord( a, s ) ); CK( hipGraphLaunch( ge, s ) ); CK( hipEventRecord( b, s ) ); CK( hipStreamSynchronize( s ) );
    float ms; CK( hipEventElapsedTime( &ms, a, b ) ); graphed.push_back( ms );
  }
  std::sort( direct.begin(), direct.end() ); std::sort( graphed.begin(), graphed.end() );
  const double ideal = iters * ( 12.0 * ( ( 4096 + 6143 ) / 6144 ) + 35.0 + 19.0 + 13.0 ) * 1e-3;     // (rough: one round of each kernel)
  printf( "chain of %d x 4 dependent kernels: direct launches median %.3f ms (min %.3f), graph replay median %.3f ms (min %.3f); kernels alone ~%.3f ms\n",
          iters, direct[direct.size() / 2], direct[0], graphed[graphed.size() / 2], graphed[0], ideal );
  return 0;
}
