// librs_benchaux.so — helpers of bench.py and of the diagnostics under tools/, NOT part of the product ABI (include/rescan_hip.h):
//   * host-side spin flags for a harness that issues independent operators from several threads and joins them thousands of
//     times per second (a thread that sleeps in a queue or on a condition variable pays the host scheduler's wake-up latency at
//     every hand-off — on a shared, busy host occasionally milliseconds, more than a whole step);
//   * a probe of where the hardware places the workgroups of a stream (which CUs a CU mask really selects).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>

extern "C" {

int rsb_spin_post( volatile int32_t* flag, int32_t value )
{
  if( !flag ) return -2;
  __atomic_store_n( (int32_t*)flag, value, __ATOMIC_RELEASE );
  return 0;
}

// returns once *flag >= at_least (acquire), busy-waiting; timeout_s > 0: -3 after that long
int rsb_spin_wait( const volatile int32_t* flag, int32_t at_least, double timeout_s )
{
  if( !flag ) return -2;
  const auto t0 = std::chrono::steady_clock::now();
  for( unsigned n = 0; ; ++n )
  {
    if( __atomic_load_n( (const int32_t*)flag, __ATOMIC_ACQUIRE ) >= at_least ) return 0;
    __builtin_ia32_pause();
    if( ( n & 0xffff ) == 0xffff && timeout_s > 0.0 &&
        std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count() > timeout_s ) return -3;
  }
}

} // extern "C"

// out[b] = XCC_ID | HW_ID << 8 of block b's first wave (HW_ID: CU_ID bits 11:8, SH_ID 12, SE_ID 15:13)
__global__ void k_probe_placement( uint32_t* out, int spin )
{
  uint32_t xcc, hw;
  asm volatile( "s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"( xcc ) );
  asm volatile( "s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"( hw ) );
  const long long t0 = wall_clock64();
  while( wall_clock64() - t0 < spin ) { }                 // hold the slot so that the blocks spread over everything allowed
  if( threadIdx.x == 0 ) out[blockIdx.x] = ( xcc & 15u ) | ( hw << 8 );
}

extern "C" int rsb_probe_placement( void* hip_stream, uint32_t* out_host, int32_t n_blocks )
{
  if( !out_host || n_blocks <= 0 ) return -2;
  hipStream_t st = (hipStream_t)hip_stream;
  uint32_t* d = nullptr;
  if( hipMalloc( (void**)&d, (size_t)n_blocks * 4 ) != hipSuccess ) return -3;
  hipLaunchKernelGGL( k_probe_placement, dim3( n_blocks ), dim3( 256 ), 0, st, d, 20000 );   // 100 MHz clock: 200 us
  hipError_t e = hipMemcpyAsync( out_host, d, (size_t)n_blocks * 4, hipMemcpyDeviceToHost, st );
  if( e == hipSuccess ) e = hipStreamSynchronize( st );
  (void)hipFree( d );
  return e == hipSuccess ? 0 : -3;
}
