// Issue rate of v_pk_add_f32 / v_pk_mul_f32 against pairs of v_add_f32 / v_mul_f32 on gfx950, at 1 and at 7 waves per SIMD: is the
// packed distance test of consider4 (rs_search.h) cheaper than the same arithmetic unpacked?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/pk_rate.hip -o tools/micro/pk_rate.bin && tools/micro/pk_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__( ( ext_vector_type( 2 ) ) );
#define REP16( x ) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__( 64 ) void k( float* out, int iters )
{
  f32x2 a0 = { (float)threadIdx.x, 1.0f }, a1 = { 2.0f, 3.0f }, a2 = { 4.0f, 5.0f }, a3 = { 6.0f, 7.0f }, q = { 0.5f, 0.25f };
  float s0 = threadIdx.x, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7, qs = 0.5f;
  for( int i = 0; i < iters; ++i )
  {
    if( MODE == 0 ) { REP16( asm volatile( "v_pk_add_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4" : "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ) : "v"( q ) ); ) }
    else { REP16( asm volatile( "v_add_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_add_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8"
                                : "+v"( s0 ), "+v"( s1 ), "+v"( s2 ), "+v"( s3 ), "+v"( s4 ), "+v"( s5 ), "+v"( s6 ), "+v"( s7 ) : "v"( qs ) ); ) }
  }
  out[blockIdx.x * 64 + threadIdx.x] = MODE == 0 ? a0.x + a1.x + a2.x + a3.x + a0.y + a1.y + a2.y + a3.y : s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
}
template <int MODE> static float run( int blocks, int iters, float* d )
{
  hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
  hipLaunchKernelGGL( k<MODE>, dim3( blocks ), dim3( 64 ), 0, 0, d, iters ); hipDeviceSynchronize();
  hipEventRecord( e0 ); hipLaunchKernelGGL( k<MODE>, dim3( blocks ), dim3( 64 ), 0, 0, d, iters ); hipEventRecord( e1 ); hipEventSynchronize( e1 );
  float ms; hipEventElapsedTime( &ms, e0, e1 ); return ms;
}
int main()
{
  float* d; hipMalloc( &d, 256 * 4 * 8 * 64 * 4 );
  const int iters = 20000;
  for( int waves_per_simd : { 1, 2, 4, 7 } )
  {
    const int blocks = 256 * 4 * waves_per_simd;
    const float pk = run<0>( blocks, iters, d ), sc = run<1>( blocks, iters, d );
    // per wave: iters x 16 x (4 packed | 8 plain) instructions = the same 128 flops x iters x 16 per lane
    printf( "%d waves per SIMD: packed %.3f ms (%.2f cycles per v_pk at 2.4 GHz per SIMD), plain %.3f ms (%.2f cycles per v_add/v_mul); packed / plain time %.2f\n", waves_per_simd,
            pk, pk * 1e-3 * 2.4e9 / ( (double)iters * 16 * 4 * waves_per_simd ), sc, sc * 1e-3 * 2.4e9 / ( (double)iters * 16 * 8 * waves_per_simd ), pk / sc );
  }
  return 0;
}
