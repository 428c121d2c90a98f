// librescan_hip host side: the C ABI of include/rescan_hip.h.
//
// Owns device clouds (cell-sorted 16-byte records + cell offset table), persistent
// workspaces, the ICP outer loop (searches and reductions on the GPU; the 6x6 LDLᵀ solve and
// the pose composition on the host in the reference's own arithmetic, lib/rs/icp.h:267-295)
// and the per-kernel HIP-event timing used by bench.py.  There is no CPU fallback: every
// compute entry point fails with RS_HIP_E_NODEVICE when no HIP device is usable.

#include "../../include/rescan_hip.h"
#include "rs_device.h"
#include "rs_math.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <array>
#include <atomic>

using namespace rs;

// ------------------------------------------------------------------------------------------
// runtime state
// ------------------------------------------------------------------------------------------

namespace {

char g_err[512] = "";
int g_device = -1;
bool g_ready = false;
// Every host thread that calls the library gets its own HIP stream and its own workspaces, so
// independent operators (an ICP chain, a score batch, a label transfer) issued from different
// threads overlap on the GPU: the tail of one kernel is filled by the others.  Clouds are
// immutable after creation and may be shared between threads.
thread_local hipStream_t g_own_stream = nullptr;
thread_local hipStream_t g_stream = nullptr;

void set_err( const char* fmt, ... )
{
  va_list ap; va_start( ap, fmt ); vsnprintf( g_err, sizeof(g_err), fmt, ap ); va_end( ap );
}

#define HIP_TRY( expr, code_on_fail )                                                        \
  do { hipError_t e_ = ( expr );                                                               \
       if( e_ != hipSuccess ) { set_err( "%s failed: %s (%s:%d)", #expr, hipGetErrorString( e_ ), __FILE__, __LINE__ ); \
                                return code_on_fail; } } while( 0 )

int ensure_ready()
{
  if( !g_ready ) { int rc = rs_hip_init( g_device >= 0 ? g_device : 0 ); if( rc ) return rc; }
  if( !g_own_stream )       // first call from this host thread
  {
    HIP_TRY( hipSetDevice( g_device ), RS_HIP_E_NODEVICE );
    HIP_TRY( hipStreamCreateWithFlags( &g_own_stream, hipStreamNonBlocking ), RS_HIP_E_NODEVICE );
  }
  if( !g_stream ) g_stream = g_own_stream;
  return RS_HIP_OK;
}

// growable device / pinned-host buffers, kept for the life of the process
struct DevBuf
{
  void* p = nullptr; size_t cap = 0;
  int ensure( size_t bytes )
  {
    if( bytes <= cap ) return RS_HIP_OK;
    if( p ) { HIP_TRY( hipFree( p ), RS_HIP_E_RUNTIME ); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    HIP_TRY( hipMalloc( &p, want ), RS_HIP_E_RUNTIME );
    cap = want;
    return RS_HIP_OK;
  }
  template <class T> T* as() { return (T*)p; }
};
struct PinBuf
{
  void* p = nullptr; size_t cap = 0;
  int ensure( size_t bytes )
  {
    if( bytes <= cap ) return RS_HIP_OK;
    if( p ) { HIP_TRY( hipHostFree( p ), RS_HIP_E_RUNTIME ); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    HIP_TRY( hipHostMalloc( &p, want, hipHostMallocDefault ), RS_HIP_E_RUNTIME );
    cap = want;
    return RS_HIP_OK;
  }
  template <class T> T* as() { return (T*)p; }
};

struct Workspace
{
  DevBuf state, slot, d2, dot, stat_acc, mom_part, res, wexp, queue, queue_count, multi;   // ICP (multi: the per-problem views of a multi-source batch)
  DevBuf poses, score_part, scores;                                             // score
  DevBuf sq_ka, sq_kb, sq_va, sq_vb, sq_pq, sq_tmp, sq_hist;                             // ... scene-space batches: keys / payloads (ping-pong), per-query terms, sort workspace
  DevBuf plc, labels, mind, fold_off, labels_o, mind_o, rows_o, ids_tab, ids_out, attr_in, attr_out;                 // labels (state in query order; *_o: input order)
  DevBuf q4, rd2, ridx, rnn, rows;                                              // rows / misc
  DevBuf enor, ecount, eoffset, e1, e2, ew;                                     // neighbourhood edges
  DevBuf cert_r, cert_dot, cert_slack;                                                    // ICP certificates
  DevBuf cov_bits, cov_plc, cov_agree;                                          // coverage scores
  DevBuf bld_pos, bld_nor, bld_k0, bld_k1, bld_v0, bld_v1, bld_v2, bld_small, bld_bits, bld_tmp;   // cloud construction
  DevBuf order_a, order_b;                                                      // ICP phase A slow-tile lists (ping-pong)
  DevBuf tmp_pos, tmp_pos2, tmp_nor2;                                           // estimate-only
  DevBuf lvl_pos, lvl_nor, lvl_cnt, lvl_within, lvl_offset, lvl_adj, lvl_state, lvl_misc, lvl_flags, lvl_scan, lvl_samples, lvl_tmp, lvl_cursor, lvl_work_a, lvl_work_b;   // level builder
  DevBuf faith;                                                                 // reference-order estimator: correspondences in source order
  DevBuf rp_segsum, rp_guess, rp_seg, rp_super, rp_totals, rp_redone;                     // ... its parallel form (replay)
  DevBuf ch_rec, ch_segsum, ch_prefix, ch_seg, ch_blk, ch_guess, ch_dbg, ch_done, ch_chk;                                    // the centroid chains of large sources (grid chains)
  PinBuf h_a, h_b, h_c, h_multi;      // h_c: the label entry points' placements; h_multi: a multi-source batch's problem views (an entry point that returns without a synchronisation — rs_hip_label_partial_device — may still be uploading from h_c)
};
thread_local Workspace g_ws;
// rs_hip_icp_align_traced: the calling thread's next single-problem alignment leaves its per-iteration errors here (the loop then
// reads its state after every iteration instead of every few)
thread_local float* g_icp_trace = nullptr;
thread_local int g_icp_trace_cap = 0;
DevBuf g_faith_redone;            // (a process-wide counter: rs_hip_icp_faith_redone)
std::mutex g_faith_redone_mu;

// ---- profiling ---------------------------------------------------------------------------
bool g_prof = false;
// The estimator of icp_estimate_rigid_xform_pt2pl (lib/rs/icp.h:210-298) by source size — round 6: the policy priced against the bar on every
// reference fixture (profiles/r06/estimator_policy.txt):
//   n <= g_ref_order_below (16 384)  the reference's own accumulation order and precisions (k_icp_faithful): its bits.  Every level-2 object of the
//                                    reference's call sites (2-10 k points); up to ~10 k points the sequential chains cost no more than the launches of
//                                    anything parallel (60-90 us per iteration either way), at 16 k 150 against 75.
//   n <= g_replay_below (0: off)     the same bits computed in parallel ("replay"): opt-in.
//   n <= g_lane_below (65 536)       LANE chains: the reference's 2.5 sigma cut and its seven centroid sums bit for bit by one wave per chain +
//                                    fp64 moments; any number of differently sized problems per launch (rs_hip_icp_align_multi) — every icp_align
//                                    call site of the reference (level-2 objects 2-10 k points, scene extracts up to ~50 k: SURVEY §8 a6).
//                                    <= 3e-6 from the reference's pose on its fixtures, equal iteration counts; 110 instead of 520 us per iteration at 50 k.
//   larger                           GRID chains: the same seven sums spread over the chip (k_chain_*) + fp64 moments; <= 1.5e-6 on ten 1 M-point
//                                    rooms, <= 9e-7 on the 24 scan-sized sweep runs.  A chain that hovers around zero gives the call up: again with
//                                    the sums by pass 2 of the replay.
// RS_HIP_EXACT_CENTROIDS=0: plain fp64 moments for everything above the first two thresholds (up to 2e-4 from the reference: NOT within the bar's
// margin — kept for measurements).  rs_hip_icp_reference_order_below( 65536 ) brings the reference's bits back for every call site.
// After a source's centroid chains gave a problem up, its next `g_chains_retry_after` calls go straight to the replay (rs_hip_icp_chains_retry_after).
std::atomic<int> g_chains_retry_after{ getenv( "RS_HIP_CHAINS_RETRY_AFTER" ) ? atoi( getenv( "RS_HIP_CHAINS_RETRY_AFTER" ) ) : 15 };
std::atomic<int> g_ref_order_below{ getenv( "RS_HIP_REF_ORDER_BELOW" ) ? atoi( getenv( "RS_HIP_REF_ORDER_BELOW" ) ) : 16384 };
std::atomic<int> g_chains_gave_up{ 0 };
std::atomic<int> g_faith_guess_permille{ getenv( "RS_HIP_FAITH_GUESS" ) ? atoi( getenv( "RS_HIP_FAITH_GUESS" ) ) : 1000 };
std::atomic<int> g_exact_centroids{ getenv( "RS_HIP_EXACT_CENTROIDS" ) ? atoi( getenv( "RS_HIP_EXACT_CENTROIDS" ) ) : 1 };
std::atomic<int> g_replay_below{ getenv( "RS_HIP_REPLAY_BELOW" ) ? atoi( getenv( "RS_HIP_REPLAY_BELOW" ) ) : 0 };
std::atomic<int> g_lane_below{ getenv( "RS_HIP_LANE_CHAINS_BELOW" ) ? atoi( getenv( "RS_HIP_LANE_CHAINS_BELOW" ) ) : 65536 };
// ... of a call with ONE problem, whose grid chains have the chip to themselves: the same sums — the same poses, bit for bit — faster from
// ~28 k points on (tools/lane_vs_grid.py, us per iteration lane | grid: 17 k 74 | 80, 25 k 80 | 80, 30 k 84 | 80, 40 k 101 | 90, 50 k 111 | 91,
// 65 k 130 | 97); batches keep the lane chains up to g_lane_below (a walk per problem side by side: eight 50 k-point refines 20 us per
// problem and iteration).  A threshold set beyond 65 536 (rs_hip_icp_lane_chains_below) means "the lane chains, whatever the call" and
// holds for single calls too; any other setting caps them at 28 672.
inline int lane_single_cap( int lane_below ) { return lane_below > 65536 ? lane_below : std::min( lane_below, 28672 ); }
inline bool icp_takes_lane_chains( int n_source, int n_problems )
{
  const int below = g_lane_below.load();
  return n_source <= ( n_problems > 1 ? below : lane_single_cap( below ) );
}
// The stop test's guard (round 6).  icp_align stops when |err - prev_err| < 1e-5 (icp.h:489).  The lane / grid chains follow the
// reference's errors to 1e-8 ... 6e-7 (their moments are exact where the reference rounds): when a decisive difference passes within
// that of 1e-5 the decision can fall the other way — one iteration more or less, 1e-4 ... 2e-4 in the pose; measured: 2 of 49
// object-sized runs, profiles/r06/stop_test_guard.txt.  A problem whose |delta err| comes within g_stop_guard of the threshold in any
// iteration that can stop (i > 5) is therefore RUN AGAIN in the reference's own order (sequential chains up to 65 536 points, their
// parallel form up to 262 144): the reference's bits.  ~20 % of stop-test calls at 1.5e-6; larger sources have no bit-exact estimator
// to fall back on (0.5 ms per iteration at 1 M points were never the default) and keep their result.  RS_HIP_STOP_GUARD=0: off.
std::atomic<float> g_stop_guard{ getenv( "RS_HIP_STOP_GUARD" ) ? (float)atof( getenv( "RS_HIP_STOP_GUARD" ) ) : 1.5e-6f };
std::atomic<long long> g_stop_guard_redone{ 0 };      // (diagnostics: rs_hip_icp_stop_guard_redone)
// PLAIN early iterations (round 6).  The reference's centroid chains decide where the iteration CONVERGES; an iteration far from the end only has
// to bring the pose near, and ICP forgets how it got there at its own contraction rate (measured on the headline, the chains in the last m
// iterations of ten: m = 10 / 7 / 5 / 3 / 1 -> 9.9e-7 / 6.9e-7 / 1.3e-6 / 2.5e-6 / 1.0e-5 from the reference's pose; profiles/r06/early_plain.txt).
// Rule, for scan-sized sources (more than 65 536 points): chain iterations precede any iteration whose result can be returned — TWO in a
// call of fixed length, which runs the plain step (fp64 moments centred on their own fp64 centroids: two launches instead of six) until two
// iterations before its end (m = 2 and m = 3 are the same few 1e-6 from the reference's pose on six rooms — <= 3.7e-6 / <= 2.6e-6, neither
// monotone in m — m = 1 is not: 1.1e-5, outside the 1e-5 this policy is held to); three in a call with the stop test (first decision at
// i = 6, icp.h:489), which runs plain in its first four iterations, and only where no bit-exact estimator guards the decision anyway
// (sources above 262 144 points: g_stop_guard).  A ten-iteration call at 1 M points: 2.52 -> 2.14 ms.  RS_HIP_EARLY_PLAIN=0: off;
// rs_hip_icp_early_plain().  RS_HIP_EARLY_TAIL=m: m chain iterations kept in both kinds of call (tools/early_plain_table.py).
std::atomic<int> g_early_plain{ getenv( "RS_HIP_EARLY_PLAIN" ) ? atoi( getenv( "RS_HIP_EARLY_PLAIN" ) ) : 1 };
inline int icp_plain_iterations( int n_source, int max_iter, bool fixed_iters )
{
  if( !g_early_plain.load() ) return 0;
  static const int forced = getenv( "RS_HIP_EARLY_TAIL" ) ? atoi( getenv( "RS_HIP_EARLY_TAIL" ) ) : 0;
  const int keep = forced > 0 ? forced : ( fixed_iters ? 2 : 3 );      // chain iterations before a result can be returned
  const int tail = std::max( 0, max_iter - keep );
  // (by SIZE, not by which kernels run the chains: an object refine contracts too slowly for this — rs_hip_icp_lane_chains_below( 0 ) puts
  //  50 k-point refines on the grid chains, and they keep them in every iteration)
  if( n_source <= 65536 ) return 0;
  if( fixed_iters ) return tail;
  return n_source > 262144 ? std::min( std::max( 0, 7 - keep ), tail ) : 0;
}
std::mutex g_prof_mutex;
struct ProfEntry { std::vector<std::pair<hipEvent_t, hipEvent_t>> spans; int64_t launches = 0; double ms = 0.0; };
std::map<std::string, ProfEntry> g_profmap;
unsigned long long* g_evals = nullptr;  // device counter of staged-and-evaluated candidates (all threads, all kernels) while profiling
std::vector<hipEvent_t> g_evpool;      // events are handed out in order and recycled by prof_collect
size_t g_evnext = 0;

// one event recorded on the calling thread's stream (null if profiling is off or events ran out)
hipEvent_t prof_event()
{
  hipEvent_t ev = nullptr;
  {
    std::lock_guard<std::mutex> lock( g_prof_mutex );
    if( g_evnext == g_evpool.size() ) { hipEvent_t x; if( hipEventCreate( &x ) != hipSuccess ) return nullptr; g_evpool.push_back( x ); }
    ev = g_evpool[g_evnext++];
  }
  (void)hipEventRecord( ev, g_stream );
  return ev;
}
void prof_span( const char* name, hipEvent_t a, hipEvent_t b )
{
  if( !a || !b ) return;
  std::lock_guard<std::mutex> lock( g_prof_mutex );
  g_profmap[name].spans.emplace_back( a, b );
}

struct ProfScope
{
  const char* name; hipEvent_t start = nullptr;
  ProfScope( const char* n ) : name( n ) { if( g_prof ) start = prof_event(); }
  ~ProfScope() { if( start ) prof_span( name, start, prof_event() ); }
};

// Consecutive segments of one stream sharing their boundary events: one event per boundary instead of
// two (an event between two dependent launches costs a few microseconds of their back-to-back dispatch,
// which matters for the ICP loop's chain of short kernels).
// The ICP loop may be sampled: one call in RS_HIP_PROF_EVERY (default 1: every call) carries the events and the averages
// per launch are taken over those — two events per iteration are ~12 us of a ~180 us iteration, 3-4 % of a bench step.
struct ProfChain
{
  const char* cur = nullptr; hipEvent_t at = nullptr;
  bool on = true;
  ProfChain()
  {
    static const unsigned every = getenv( "RS_HIP_PROF_EVERY" ) ? (unsigned)std::max( 1, atoi( getenv( "RS_HIP_PROF_EVERY" ) ) ) : 1u;
    thread_local unsigned tick = 0;
    if( g_prof ) on = ( tick++ % every ) == 0;
  }
  void mark( const char* name )
  {
    if( !g_prof || !on ) return;
    hipEvent_t ev = prof_event();
    if( cur ) prof_span( cur, at, ev );
    cur = name; at = ev;
  }
  ~ProfChain() { if( cur ) mark( nullptr ); }
};

void prof_collect()
{
  std::lock_guard<std::mutex> lock( g_prof_mutex );
  for( auto& kv : g_profmap )
  {
    ProfEntry& e = kv.second;
    for( auto& sp : e.spans )
    {
      float ms = 0.0f;
      (void)hipEventSynchronize( sp.second );
      if( hipEventElapsedTime( &ms, sp.first, sp.second ) == hipSuccess ) { e.ms += ms; e.launches++; }
    }
    e.spans.clear();
  }
  g_evnext = 0;
}

} // namespace

// ------------------------------------------------------------------------------------------
// clouds
// ------------------------------------------------------------------------------------------

struct rs_hip_cloud
{
  int32_t n = 0;
  bool has_nor = false;
  GridView view{};
  float4* d_pos = nullptr;
  float4* d_nor = nullptr;
  uint32_t* d_cell_start = nullptr;
  // query layout: the same points in Hilbert order, cut into tiles (one wave each)
  QueryView qview{};
  float4* d_qpos = nullptr;
  float4* d_qnor = nullptr;
  uint32_t* d_tiles = nullptr;
  int* d_qby_orig = nullptr;      // original index -> query slot (the reference-order estimator walks the source in its own order)
  std::vector<int32_t> qorder;    // query slot -> original index
  std::vector<int32_t> order;     // sorted slot -> original index
  std::vector<float> h_pos, h_nor;  // original-order host copies (AoS)
  float nor_max = 1.0f;             // max |normal| (bounds how fast a gate value can change with the query normal)
  float cell = 0.0f;
  int64_t bytes = 0;
  // as an ICP source whose centroid chains gave a problem up (sums that hover around zero: rs_hip_icp_align_batch): the next calls
  // this many go straight to pass 2 of the replay instead of paying for the attempt first
  mutable std::atomic<int> chains_wander{ 0 };
};

namespace {

// ---- query layout ---------------------------------------------------------------------------
// (Hilbert order + greedy tiling are computed on the device: rs_build.hip.)

// Tile extent limit from the cloud's own sampling density: a full tile of 64 surface samples
// spans about 8 sample spacings; allow half as much again, and never less than 0.25 m.
float query_extent_limit( int32_t n, size_t occupied_cells, float cell )
{
  if( n <= 0 || occupied_cells == 0 || !( cell > 0.0f ) ) return FLT_MAX;
  float per_cell = (float)n / (float)occupied_cells;
  float spacing = cell / std::sqrt( std::max( per_cell, 1.0f ) );
  // ... and never more than 0.3 m: a sparse cloud (a subsampled level, a random subset) searched in a denser one would
  // otherwise get tiles whose box is mostly the empty space between its points (1000 points against a 200 k scan:
  // 404 -> 60 us per ICP iteration with the cap)
  static const float cap = getenv( "RS_HIP_TILE_EXTENT_MAX" ) ? (float)atof( getenv( "RS_HIP_TILE_EXTENT_MAX" ) ) : 0.3f;
  return std::min( cap, std::max( 0.25f, 12.0f * spacing ) );
}

} // namespace

extern "C" {

const char* rs_hip_version( void ) { return "rescan_hip 0.1 gfx950"; }
const char* rs_hip_last_error( void ) { return g_err; }

int rs_hip_init( int device )
{
  int count = 0;
  hipError_t e = hipGetDeviceCount( &count );
  if( e != hipSuccess || count <= 0 )
  {
    set_err( "rs_hip_init: no HIP device (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString( e ) );
    return RS_HIP_E_NODEVICE;
  }
  if( device < 0 || device >= count ) { set_err( "rs_hip_init: device %d out of range (%d devices)", device, count ); return RS_HIP_E_ARG; }
  HIP_TRY( hipSetDevice( device ), RS_HIP_E_NODEVICE );
  // RS_HIP_SCHEDULE=spin: completion waits of THIS device busy-wait instead of sleeping — for callers that wait for small results
  // thousands of times per second (librescan_dropin.so asks for it: the unchanged pose_proposal issues ~37 k searches per run; so
  // does bench.py).  Opt-in: a library does not change a process's scheduling policy by itself (8 ranks x 3 busy-waiting threads on
  // a host with a CPU quota would starve each other).  Refused (harmlessly) when the process has set the device up already.
  {
    const char* sch = getenv( "RS_HIP_SCHEDULE" );
    if( sch && !std::strcmp( sch, "spin" ) ) (void)hipSetDeviceFlags( hipDeviceScheduleSpin );
    (void)hipGetLastError();
  }
  g_device = device;
  g_ready = true;
  return RS_HIP_OK;
}

int rs_hip_set_stream( void* s )
{
  int rc = ensure_ready(); if( rc ) return rc;
  g_stream = s ? (hipStream_t)s : g_own_stream;
  return RS_HIP_OK;
}

int rs_hip_stream_cu_mask( const uint32_t* mask, int32_t n_words )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( ( !mask ) != ( n_words <= 0 ) ) { set_err( "stream_cu_mask: bad arguments" ); return RS_HIP_E_ARG; }
  hipStream_t s = nullptr;
  if( mask ) HIP_TRY( hipExtStreamCreateWithCUMask( &s, (uint32_t)n_words, mask ), RS_HIP_E_RUNTIME );
  else       HIP_TRY( hipStreamCreateWithFlags( &s, hipStreamNonBlocking ), RS_HIP_E_RUNTIME );          // (null, 0): back to an unrestricted stream
  if( g_own_stream ) { (void)hipStreamSynchronize( g_own_stream ); (void)hipStreamDestroy( g_own_stream ); }
  const bool was_own = g_stream == g_own_stream || !g_stream;
  g_own_stream = s;
  if( was_own ) g_stream = s;
  return RS_HIP_OK;
}

void* rs_hip_get_stream( void )
{
  if( ensure_ready() ) return nullptr;
  return (void*)g_stream;
}

int rs_hip_synchronize( void )
{
  int rc = ensure_ready(); if( rc ) return rc;
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  return RS_HIP_OK;
}

int rs_hip_profile_enable( int on )
{
  if( on && !g_evals )
  {
    int rc = ensure_ready(); if( rc ) return rc;
    HIP_TRY( hipMalloc( (void**)&g_evals, EVAL_SHARDS * 64 ), RS_HIP_E_RUNTIME );
    HIP_TRY( hipMemset( g_evals, 0, EVAL_SHARDS * 64 ), RS_HIP_E_RUNTIME );
  }
  if( on )
  {
    // events are created ahead of the timed region (creating them one by one inside it shows up as stalls of a step)
    std::lock_guard<std::mutex> lock( g_prof_mutex );
    while( g_evpool.size() < 8192 ) { hipEvent_t x; if( hipEventCreate( &x ) != hipSuccess ) break; g_evpool.push_back( x ); }
  }
  g_prof = on != 0;
  return RS_HIP_OK;
}
// A named no-op on the calling thread's stream: where a caller's unit of work begins in a kernel trace (tools/pmc_summary.py --trace
// cuts a `rocprofv3 --kernel-trace` of bench.py at these; VERDICT r05, weak 6: a trace must show a step of the TIMED workload).
namespace rs { __global__ void k_step_marker() {} }
int rs_hip_profile_marker( void )
{
  int rc = ensure_ready(); if( rc ) return rc;
  hipLaunchKernelGGL( rs::k_step_marker, dim3( 1 ), dim3( 1 ), 0, g_stream );
  return RS_HIP_OK;
}
int rs_hip_profile_reset( void )
{
  prof_collect();
  std::lock_guard<std::mutex> lock( g_prof_mutex );
  for( auto& kv : g_profmap ) { kv.second.launches = 0; kv.second.ms = 0.0; }
  if( g_evals ) (void)hipMemset( g_evals, 0, EVAL_SHARDS * 64 );
  return RS_HIP_OK;
}
int rs_hip_profile_read( const char* name, int64_t* launches, double* total_ms )
{
  prof_collect();
  if( name && !std::strcmp( name, "candidates" ) )
  {
    // not a kernel: the number of candidates staged and evaluated since the last reset (each by the 64 lanes of its wave)
    unsigned long long v = 0;
    if( g_evals )
    {
      std::vector<unsigned long long> h( (size_t)EVAL_SHARDS * 8 );
      (void)hipMemcpy( h.data(), g_evals, h.size() * 8, hipMemcpyDeviceToHost );
      for( int k = 0; k < EVAL_SHARDS; ++k ) v += h[(size_t)k * 8];
    }
    if( launches ) *launches = (int64_t)v;
    if( total_ms ) *total_ms = 0.0;
    return RS_HIP_OK;
  }
  std::lock_guard<std::mutex> lock( g_prof_mutex );
  auto it = g_profmap.find( name ? name : "" );
  if( it == g_profmap.end() ) { if( launches ) *launches = 0; if( total_ms ) *total_ms = 0.0; return RS_HIP_OK; }
  if( launches ) *launches = it->second.launches;
  if( total_ms ) *total_ms = it->second.ms;
  return RS_HIP_OK;
}

// ---- cloud build -------------------------------------------------------------------------
// The caller's raw arrays are uploaded once and everything else happens on the device (rs_build.hip):
// bounds, the density-derived cell size, a stable sort by cell (input order inside a cell, like
// msh_hash_grid.h:511-532) with a dense offset table, and the Hilbert-ordered, tiled query layout.
// The host only takes the few decisions in between (cell size, table size, number of tiles).
// The index is built once per cloud level and reused by every search.
// pos / nor: packed xyz, host pointers — or device pointers when from_device (a level gathered on the device: the
// host copies the shim and the find_corrs entry point use are then downloaded instead of uploaded)
// (diagnostics, rs_hip_cloud_build_seconds: where a cloud's construction goes — host copy | upload + bounds | cell index | Hilbert order + tiles;
//  wall clock of the calling thread between the build's own synchronisations, summed over every cloud built by the process)
static std::mutex g_build_mu;
static double g_build_s[4] = { 0, 0, 0, 0 };
static long long g_build_n = 0;
struct BuildClock
{
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now(); double acc[4] = { 0, 0, 0, 0 };
  void lap( int k ) { const auto now = std::chrono::steady_clock::now(); acc[k] += std::chrono::duration<double>( now - t ).count(); t = now; }
  void commit() { std::lock_guard<std::mutex> lk( g_build_mu ); for( int k = 0; k < 4; ++k ) g_build_s[k] += acc[k]; ++g_build_n; }
};
static rs_hip_cloud_t* cloud_create_impl( const float* pos, const float* nor, int32_t n, float cell_size, bool from_device )
{
  if( ensure_ready() != RS_HIP_OK ) return nullptr;
  if( n < 0 || ( n > 0 && !pos ) ) { set_err( "rs_hip_cloud_create: bad arguments" ); return nullptr; }
  BuildClock clk;
  rs_hip_cloud* c = new rs_hip_cloud();
  c->n = n; c->has_nor = nor != nullptr;
  if( !from_device )
  {
    c->h_pos.assign( pos, pos + (size_t)3 * n );
    if( nor ) c->h_nor.assign( nor, nor + (size_t)3 * n );
  }
  else
  {
    c->h_pos.resize( (size_t)3 * n );
    if( nor ) c->h_nor.resize( (size_t)3 * n );
  }
  clk.lap( 0 );
  auto fail = [&]( hipError_t e ) { set_err( "rs_hip_cloud_create: %s", hipGetErrorString( e ) ); rs_hip_cloud_destroy( c ); return (rs_hip_cloud_t*)nullptr; };
  auto failrc = [&]( const char* what ) { set_err( "rs_hip_cloud_create: %s", what ); rs_hip_cloud_destroy( c ); return (rs_hip_cloud_t*)nullptr; };
#define CC( expr ) do { hipError_t e_ = ( expr ); if( e_ != hipSuccess ) return fail( e_ ); } while( 0 )
  const size_t nn = (size_t)std::max( 1, n );
  Workspace& W = g_ws;
  if( W.bld_pos.ensure( nn * 12 ) || ( nor && W.bld_nor.ensure( nn * 12 ) ) || W.bld_k0.ensure( ( nn + 1 ) * 4 ) || W.bld_k1.ensure( ( nn + 1 ) * 4 ) ||
      W.bld_v0.ensure( nn * 4 ) || W.bld_v1.ensure( nn * 4 ) || W.bld_v2.ensure( nn * 4 ) || W.bld_small.ensure( 64 ) )
    return failrc( "workspace allocation failed" );
  float* d_raw = W.bld_pos.as<float>(); float* d_rawn = nor ? W.bld_nor.as<float>() : nullptr;
  uint32_t *k0 = W.bld_k0.as<uint32_t>(), *k1 = W.bld_k1.as<uint32_t>(), *v0 = W.bld_v0.as<uint32_t>(), *v1 = W.bld_v1.as<uint32_t>(), *v2 = W.bld_v2.as<uint32_t>();
  unsigned* d_small = W.bld_small.as<unsigned>();
  if( n > 0 )
  {
    CC( hipMemcpyAsync( d_raw, pos, (size_t)n * 12, from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, g_stream ) );
    if( nor ) CC( hipMemcpyAsync( d_rawn, nor, (size_t)n * 12, from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, g_stream ) );
    if( from_device )
    {
      CC( hipMemcpyAsync( c->h_pos.data(), pos, (size_t)n * 12, hipMemcpyDeviceToHost, g_stream ) );
      if( nor ) CC( hipMemcpyAsync( c->h_nor.data(), nor, (size_t)n * 12, hipMemcpyDeviceToHost, g_stream ) );
    }
  }

  // bounds (+ the largest normal, which bounds how fast a gate value can change with the query normal)
  float mn[3] = { 0, 0, 0 }, mx[3] = { 0, 0, 0 };
  if( n > 0 )
  {
    const unsigned init[8] = { ~0u, ~0u, ~0u, 0u, 0u, 0u, 0u, 0u };
    unsigned got[8];
    CC( hipMemcpyAsync( d_small, init, 32, hipMemcpyHostToDevice, g_stream ) );
    launch_build_bounds( d_raw, d_rawn, n, d_small, g_stream );
    CC( hipMemcpyAsync( got, d_small, 32, hipMemcpyDeviceToHost, g_stream ) );
    CC( hipStreamSynchronize( g_stream ) );
    auto dec = []( unsigned u ) { unsigned b = ( u & 0x80000000u ) ? ( u ^ 0x80000000u ) : ~u; float f; std::memcpy( &f, &b, 4 ); return f; };
    for( int a = 0; a < 3; ++a ) { mn[a] = dec( got[a] ); mx[a] = dec( got[3 + a] ); }
    for( int a = 0; a < 3; ++a ) { if( !( mx[a] >= mn[a] ) || !std::isfinite( mn[a] ) || !std::isfinite( mx[a] ) ) { mn[a] = 0; mx[a] = 0; } }
    if( nor )
    {
      float n2; std::memcpy( &n2, &got[6], 4 );
      c->nor_max = ( got[7] || !std::isfinite( n2 ) ) ? INFINITY : std::sqrt( n2 ) * 1.00001f;
    }
  }

  clk.lap( 1 );
  // cell_size < 0: pick the cell from the cloud's own sampling density (about two sample
  // spacings: a surface patch then holds ~4 points per cell, the first search shell ~100-300).
  if( cell_size < 0.0f )
  {
    cell_size = 0.1f;
    if( n > 0 )
    {
      float ext = std::max( mx[0] - mn[0], std::max( mx[1] - mn[1], mx[2] - mn[2] ) );
      float c0 = std::min( 0.1f, std::max( ext / 64.0f, 1e-4f ) );
      for( int attempt = 0; attempt < 8; ++attempt, c0 *= 2.0f )
      {
        const float inv = 1.0f / c0;
        unsigned long long td[3]; double cells = 1.0;
        for( int a = 0; a < 3; ++a ) { td[a] = (unsigned long long)std::max( 0.0f, floorf( ( mx[a] - mn[a] ) * inv ) ) + 2ull; cells *= (double)td[a]; }
        if( cells > 1073741824.0 && attempt < 7 ) continue;          // trial grid too fine to count (huge, sparse extent): coarsen
        float per_cell = 4.0f;
        if( cells <= 1073741824.0 )
        {
          const size_t words = (size_t)( cells / 32.0 ) + 2;
          if( W.bld_bits.ensure( words * 4 ) ) return failrc( "workspace allocation failed" );
          CC( hipMemsetAsync( W.bld_bits.p, 0, words * 4, g_stream ) );
          CC( hipMemsetAsync( d_small, 0, 4, g_stream ) );
          launch_build_mark( d_raw, n, mn, inv, td[1], td[2], W.bld_bits.as<uint32_t>(), g_stream );
          launch_popcount( W.bld_bits.as<uint32_t>(), (int)words, (int*)d_small, g_stream );
          int occ = 0;
          CC( hipMemcpyAsync( &occ, d_small, 4, hipMemcpyDeviceToHost, g_stream ) );
          CC( hipStreamSynchronize( g_stream ) );
          per_cell = (float)n / (float)std::max( 1, occ );
        }
        if( per_cell >= 4.0f || attempt == 7 ) { cell_size = 2.0f * c0 / std::sqrt( std::max( per_cell, 1.0f ) ); break; }
      }
      static const float scale = getenv( "RS_HIP_AUTO_CELL_SCALE" ) ? (float)atof( getenv( "RS_HIP_AUTO_CELL_SCALE" ) ) : 1.0f;
      cell_size = std::min( std::max( cell_size * scale, 0.005f ), 2.0f );
    }
  }
  int dims[3] = { 1, 1, 1 };
  float inv_cell = 0.0f, cell = cell_size;
  if( cell_size > 0.0f && n > 0 )
  {
    for( ;; )
    {
      double cells = 1.0;
      for( int a = 0; a < 3; ++a ) { dims[a] = (int)std::floor( ( (double)mx[a] - (double)mn[a] ) / (double)cell ) + 1; cells *= dims[a]; }
      if( cells <= 48.0e6 ) break;
      cell *= 2.0f;                          // keep the dense offset table below ~200 MB
    }
    inv_cell = 1.0f / cell;
  }
  c->cell = ( cell_size > 0.0f ) ? cell : 0.0f;
  const size_t n_cells = (size_t)dims[0] * dims[1] * dims[2];

  const size_t pb = nn * sizeof(float4);
  CC( hipMalloc( (void**)&c->d_pos, pb ) );
  if( nor ) CC( hipMalloc( (void**)&c->d_nor, pb ) );
  CC( hipMalloc( (void**)&c->d_cell_start, ( n_cells + 1 ) * 4 ) );
  CC( hipMemsetAsync( c->d_cell_start, 0, ( n_cells + 1 ) * 4, g_stream ) );
  size_t occupied = 0;
  int key_bits = 1; while( ( 1ull << key_bits ) < n_cells && key_bits < 32 ) ++key_bits;
  const size_t tmp_bytes = std::max( std::max( build_sort_temp_bytes( (int)nn, 32 ), build_scan_temp_bytes( n_cells + 1 ) ), build_scan_temp_bytes( nn + 1 ) );
  if( W.bld_tmp.ensure( tmp_bytes + 256 ) ) return failrc( "workspace allocation failed" );
  if( n > 0 )
  {
    // cell ids + per-cell counts -> offsets; stable sort of the point indices by cell
    launch_build_cellids( d_raw, n, mn, inv_cell, dims, k0, v0, c->d_cell_start, g_stream );
    if( build_exclusive_scan( W.bld_tmp.p, tmp_bytes, c->d_cell_start, c->d_cell_start, n_cells + 1, g_stream ) ) return failrc( "device scan failed" );
    if( build_sort_pairs( W.bld_tmp.p, tmp_bytes, k0, k1, v0, v1, n, key_bits, g_stream ) ) return failrc( "device sort failed" );
    launch_build_gather( d_raw, d_rawn, v1, n, c->d_pos, c->d_nor, g_stream );
    CC( hipMemsetAsync( d_small, 0, 4, g_stream ) );
    launch_build_count_runs( k1, n, (int*)d_small, g_stream );
    int occ = 0;
    CC( hipMemcpyAsync( &occ, d_small, 4, hipMemcpyDeviceToHost, g_stream ) );
    c->order.resize( (size_t)n );
    CC( hipMemcpyAsync( c->order.data(), v1, (size_t)n * 4, hipMemcpyDeviceToHost, g_stream ) );
    CC( hipStreamSynchronize( g_stream ) );
    occupied = (size_t)occ;
  }
  c->bytes = (int64_t)( pb * ( nor ? 2 : 1 ) + ( n_cells + 1 ) * 4 );

  clk.lap( 2 );
  GridView& v = c->view;
  v.pos = c->d_pos; v.nor = c->d_nor; v.cell_start = c->d_cell_start;
  v.minx = mn[0]; v.miny = mn[1]; v.minz = mn[2]; v.inv_cell = inv_cell; v.cell = inv_cell > 0.0f ? cell : 0.0f;
  v.w = dims[0]; v.h = dims[1]; v.d = dims[2]; v.n = n;

  // query layout: Hilbert order (10 bits per axis over the largest extent), tiles of <= 64 points
  float lim_cell = cell_size > 0.0f ? cell : 0.1f;
  if( !( cell_size > 0.0f ) )
  {
    // brute layout has one cell: estimate the occupancy of a 0.1 m grid instead
    double vol_cells = 1.0; for( int a = 0; a < 3; ++a ) vol_cells *= std::floor( ( (double)mx[a] - mn[a] ) / 0.1 ) + 1.0;
    occupied = (size_t)std::max( 1.0, std::min( (double)n, std::pow( vol_cells, 2.0 / 3.0 ) ) );
  }
  CC( hipMalloc( (void**)&c->d_qpos, pb ) );
  if( nor ) CC( hipMalloc( (void**)&c->d_qnor, pb ) );
  int n_tiles = 0;
  if( n > 0 )
  {
    float ext = 0.0f;
    for( int a = 0; a < 3; ++a ) { float e = mx[a] - mn[a]; if( std::isfinite( e ) && e > ext ) ext = e; }
    const float scale = ext > 0.0f ? 1024.0f / ext : 0.0f;
    launch_build_hilbert( d_raw, n, mn, scale, k0, v0, g_stream );
    if( build_sort_pairs( W.bld_tmp.p, tmp_bytes, k0, k1, v0, v2, n, 30, g_stream ) ) return failrc( "device sort failed" );
    launch_build_gather( d_raw, d_rawn, v2, n, c->d_qpos, c->d_qnor, g_stream );
    CC( hipMalloc( (void**)&c->d_qby_orig, (size_t)n * 4 ) );
    launch_build_inverse( c->d_qpos, n, c->d_qby_orig, g_stream );
    // tile starts: flags -> exclusive scan -> scatter
    uint32_t* flags = k0; uint32_t* scanned = k1;
    CC( hipMemsetAsync( flags + n, 0, 4, g_stream ) );
    launch_build_tile_flags( c->d_qpos, n, query_extent_limit( n, occupied, lim_cell ), flags, v0, v1, g_stream );   // (order / v1 is on the host by now)
    if( build_exclusive_scan( W.bld_tmp.p, tmp_bytes, flags, scanned, (size_t)n + 1, g_stream ) ) return failrc( "device scan failed" );
    unsigned total = 0;
    CC( hipMemcpyAsync( &total, scanned + n, 4, hipMemcpyDeviceToHost, g_stream ) );
    c->qorder.resize( (size_t)n );
    CC( hipMemcpyAsync( c->qorder.data(), v2, (size_t)n * 4, hipMemcpyDeviceToHost, g_stream ) );
    CC( hipStreamSynchronize( g_stream ) );
    n_tiles = (int)total;
    CC( hipMalloc( (void**)&c->d_tiles, ( (size_t)n_tiles + 1 ) * 4 ) );
    launch_build_tile_scatter( flags, scanned, n, c->d_tiles, g_stream );
    const uint32_t last = (uint32_t)n;
    CC( hipMemcpyAsync( c->d_tiles + n_tiles, &last, 4, hipMemcpyHostToDevice, g_stream ) );
    CC( hipStreamSynchronize( g_stream ) );
  }
  else
  {
    CC( hipMalloc( (void**)&c->d_tiles, 4 ) );
    CC( hipMemsetAsync( c->d_tiles, 0, 4, g_stream ) );
    CC( hipStreamSynchronize( g_stream ) );
  }
  c->bytes += (int64_t)( pb * ( nor ? 2 : 1 ) + ( (size_t)n_tiles + 1 ) * 4 + (size_t)n * 4 );
  c->qview.pos = c->d_qpos; c->qview.nor = c->d_qnor; c->qview.tiles = c->d_tiles;
  c->qview.n = n; c->qview.n_tiles = n_tiles;
  clk.lap( 3 ); clk.commit();
#undef CC
  return c;
}

// The fills an ICP call begins with (queue lengths, tickets, statistics accumulators, the two lists of slow tiles, "no certificate") as ONE
// launch.  Each was a hipMemsetAsync — a kernel of the runtime's, 7-9 us apart on the stream: ten of them put 84 us in front of the first
// search of every call (profiles/r06/trace_one_serial_step.txt before this), 3 % of the headline's step.
namespace rs {
struct FillRanges { static constexpr int MAX = 16; uint32_t* p[MAX]; unsigned long long words[MAX]; uint32_t value[MAX]; int n; };
__global__ __launch_bounds__( 256 ) void k_fill_ranges( FillRanges R )
{
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  for( int r = 0; r < R.n; ++r )
    for( size_t i = t; i < R.words[r]; i += step ) R.p[r][i] = R.value[r];
}
}
namespace {
thread_local rs::FillRanges g_fills{};
int icp_fills_flush()
{
  if( g_fills.n == 0 ) return RS_HIP_OK;
  unsigned long long most = 0;
  for( int r = 0; r < g_fills.n; ++r ) most = std::max( most, g_fills.words[r] );
  const int blocks = (int)std::min<unsigned long long>( 1024ull, std::max<unsigned long long>( 1ull, ( most + 1023 ) / 1024 ) );
  hipLaunchKernelGGL( rs::k_fill_ranges, dim3( blocks ), dim3( 256 ), 0, g_stream, g_fills );
  g_fills.n = 0;
  return RS_HIP_OK;      // (a failed launch surfaces at the call's synchronisation, like every other kernel's)
}
// hipMemsetAsync( p, byte, bytes, g_stream ), deferred to the call's one fill launch (icp_fills_flush: before the first kernel that reads p)
int icp_fill( void* p, int byte, size_t bytes )
{
  if( bytes == 0 ) return RS_HIP_OK;
  if( ( bytes & 3 ) || ( (uintptr_t)p & 3 ) ) { HIP_TRY( hipMemsetAsync( p, byte, bytes, g_stream ), RS_HIP_E_RUNTIME ); return RS_HIP_OK; }
  if( g_fills.n == rs::FillRanges::MAX ) { if( int rc = icp_fills_flush() ) return rc; }
  const uint32_t b = (uint32_t)( byte & 0xff );
  g_fills.p[g_fills.n] = (uint32_t*)p; g_fills.words[g_fills.n] = bytes / 4; g_fills.value[g_fills.n] = b | ( b << 8 ) | ( b << 16 ) | ( b << 24 ); ++g_fills.n;
  return RS_HIP_OK;
}
}
extern "C" int64_t rs_hip_cloud_build_seconds( double out[4], int32_t reset )
{
  std::lock_guard<std::mutex> lk( g_build_mu );
  if( out ) for( int k = 0; k < 4; ++k ) out[k] = g_build_s[k];
  const long long n = g_build_n;
  if( reset ) { for( int k = 0; k < 4; ++k ) g_build_s[k] = 0.0; g_build_n = 0; }
  return n;
}

rs_hip_cloud_t* rs_hip_cloud_create( const float* pos, const float* nor, int32_t n, float cell_size )
{
  return cloud_create_impl( pos, nor, n, cell_size, false );
}

void rs_hip_cloud_destroy( rs_hip_cloud_t* c )
{
  if( !c ) return;
  if( c->d_pos ) (void)hipFree( c->d_pos );
  if( c->d_nor ) (void)hipFree( c->d_nor );
  if( c->d_cell_start ) (void)hipFree( c->d_cell_start );
  if( c->d_qpos ) (void)hipFree( c->d_qpos );
  if( c->d_qnor ) (void)hipFree( c->d_qnor );
  if( c->d_tiles ) (void)hipFree( c->d_tiles );
  if( c->d_qby_orig ) (void)hipFree( c->d_qby_orig );
  delete c;
}

int32_t rs_hip_cloud_size( const rs_hip_cloud_t* c ) { return c ? c->n : 0; }
int64_t rs_hip_cloud_bytes( const rs_hip_cloud_t* c ) { return c ? c->bytes : 0; }

void rs_hip_sincosf_model( const float* x, int64_t n, float* sin_out, float* cos_out )
{
  for( int64_t i = 0; i < n; ++i ) rs_sincosf_model( x[i], sin_out[i], cos_out[i] );
}
void rs_hip_mat4_inverse( const float* m, float* out ) { Mat4 a; std::memcpy( a.m, m, 64 ); Mat4 r = mat4_inverse( a ); std::memcpy( out, r.m, 64 ); }
void rs_hip_mat4_mul( const float* a_, const float* b_, float* out ) { Mat4 a, b; std::memcpy( a.m, a_, 64 ); std::memcpy( b.m, b_, 64 ); Mat4 r = mat4_mul( a, b ); std::memcpy( out, r.m, 64 ); }

} // extern "C"

// ------------------------------------------------------------------------------------------
// ICP
// ------------------------------------------------------------------------------------------

namespace {

// Hand-off policy of the two-phase search.  A lone wave that is still unsettled after streaming
// this many candidates sits in a cluttered neighbourhood and would need ~0.5 ms for the rest of
// its box; it is queued for the cooperative kernel instead.  That only pays when the launch has
// too few tiles to hide such stragglers behind other work (measured on MI355X: 1 M-query ICP,
// 15.6 k tiles: 0.60 -> 0.47 ms per search with hand-off; 256-pose score batch, 40 k tiles:
// 1.5 -> 3.0 ms, i.e. worse), so big launches keep everything in phase A.
// Hand-off rule of a search launch (rs_search.h: tile_search): low 16 bits = candidates a lone wave may
// stream while a lane is unsettled, high bits = the shell after which an unsettled tile is handed off whatever it
// streamed.  A wave that runs almost alone on its SIMD (few tiles in flight) is latency-bound, so the fewer tiles a
// launch has, the earlier the cooperative kernel takes over; launches with > 24 k tiles (score batches) keep
// everything in phase A — there the stragglers hide behind other tiles and the hand-off measured slower.
inline int handoff_threshold( long long total_tiles )
{
  static const int forced = getenv( "RS_HIP_SOLO_STAGES" ) ? atoi( getenv( "RS_HIP_SOLO_STAGES" ) ) : -1;
  static const int forced_k = getenv( "RS_HIP_HANDOFF_K" ) ? atoi( getenv( "RS_HIP_HANDOFF_K" ) ) : -1;
  if( forced > 0 ) return forced | ( std::max( 0, forced_k ) << 16 );
  if( total_tiles > 24000 ) return 0x7fffffff;
  const int k_always = forced_k >= 0 ? forced_k : 1;
  return ( total_tiles <= 12000 ? 128 : 192 ) | ( k_always << 16 );
}

inline float radius_sq_of( float r ) { return (float)( (double)r * (double)r ); }   // msh_hash_grid.h:1104,1111 + :828

struct IcpCtx
{
  IcpLaunch L{};
  int n_waves = 0;             // tiles of the (largest) source: the searches' grid
  size_t total_pts = 0;        // rows of the per-point arrays over all problems (one source: n_prob x n)
  size_t total_tiles = 0;      // likewise per tile
  size_t heavy_words = 0;      // words of one slow-tile buffer over all problems
};

// Device-resident loop state, one 4-byte word array after the other (n = n_prob):
//   T1 16n | active n | T1_prev 16n | iters n | err n | prev_err n | queued n | ticket 2n
constexpr size_t ICP_STATE_WORDS = 16 + 1 + 16 + 1 + 1 + 1 + 1 + 2;

// One source for all problems (src), or one per problem (srcs[n_prob], src == null: a multi-source batch — the kernels then bind
// their problem's view on the device, rs_icp.h: icp_bind).
int icp_prepare( IcpCtx& cx, const rs_hip_cloud_t* src, const rs_hip_cloud_t* tgt, int n_prob, const float* T2, const rs_hip_cloud_t* const* srcs = nullptr )
{
  if( n_prob <= 0 ) { set_err( "icp: empty batch" ); return RS_HIP_E_ARG; }
  if( !tgt || !tgt->has_nor || ( !src && !srcs ) ) { set_err( "icp: source and target clouds need normals" ); return RS_HIP_E_ARG; }
  for( int p = 0; p < ( srcs ? n_prob : 1 ); ++p )
  {
    const rs_hip_cloud_t* c = srcs ? srcs[p] : src;
    if( !c || !c->has_nor ) { set_err( "icp: source and target clouds need normals" ); return RS_HIP_E_ARG; }
  }
  IcpLaunch& L = cx.L;
  L.tgt = tgt->view; L.tgt.evals = g_prof ? g_evals : nullptr; L.n_prob = n_prob; L.K = 16;   // icp.h:330
  Mat4 t2; std::memcpy( t2.m, T2, 64 );
  Mat4 t2i = mat4_inverse( t2 );                                                                             // icp.h:329
  std::memcpy( L.T2i.m, t2i.m, 64 );
  const size_t np = (size_t)n_prob;
  int rc;
  L.multi = nullptr;
  if( srcs )
  {
    if( ( rc = g_ws.h_multi.ensure( np * sizeof( IcpProblem ) ) ) || ( rc = g_ws.multi.ensure( np * sizeof( IcpProblem ) ) ) ) return rc;
    // (pinned and only ever written here: every ICP entry point ends with a stream synchronisation, so the upload of the previous
    //  call's views is over before this one overwrites them)
    IcpProblem* P = g_ws.h_multi.as<IcpProblem>();
    size_t pts = 0, tiles = 0, heavy = 0; int max_n = 0, max_tiles = 0;
    for( size_t p = 0; p < np; ++p )
    {
      P[p].src = srcs[p]->qview; P[p].by_orig = srcs[p]->d_qby_orig;
      P[p].pt_off = (long long)pts; P[p].tile_off = (long long)tiles; P[p].heavy_off = (long long)heavy;
      pts += (size_t)srcs[p]->n; tiles += (size_t)srcs[p]->qview.n_tiles; heavy += heavy_stride( srcs[p]->qview.n_tiles );
      max_n = std::max( max_n, (int)srcs[p]->n ); max_tiles = std::max( max_tiles, srcs[p]->qview.n_tiles );
    }
    HIP_TRY( hipMemcpyAsync( g_ws.multi.p, P, np * sizeof( IcpProblem ), hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
    L.multi = (const IcpProblem*)g_ws.multi.p; L.src = QueryView{}; L.by_orig = nullptr;
    L.max_n = max_n; L.max_tiles = max_tiles;
    cx.total_pts = std::max<size_t>( 1, pts ); cx.total_tiles = std::max<size_t>( 1, tiles ); cx.heavy_words = heavy;
  }
  else
  {
    L.src = src->qview; L.by_orig = src->d_qby_orig; L.max_n = src->n; L.max_tiles = src->qview.n_tiles;
    cx.total_pts = np * std::max<size_t>( 1, (size_t)src->n ); cx.total_tiles = np * (size_t)std::max( 1, src->qview.n_tiles );
    cx.heavy_words = np * heavy_stride( src->qview.n_tiles );
  }
  cx.n_waves = L.max_tiles;
  L.n_mom_blocks = std::max( 1, std::min( 512, ( L.max_n + 255 ) / 256 ) );    // 512: gathers want more waves in flight than 256 give, the final tree fewer partials than 1024
  if( ( rc = g_ws.state.ensure( np * ICP_STATE_WORDS * 4 ) ) ||
      ( rc = g_ws.slot.ensure( cx.total_pts * 4 ) ) || ( rc = g_ws.d2.ensure( cx.total_pts * 4 ) ) || ( rc = g_ws.dot.ensure( cx.total_pts * 4 ) ) ||
      ( rc = g_ws.stat_acc.ensure( np * STAT_SHARDS * 4 * 8 ) ) ||
      ( rc = g_ws.mom_part.ensure( np * L.n_mom_blocks * ICP_NMOM * 8 ) ) || ( rc = g_ws.res.ensure( np * ICP_NRES * 8 ) ) ||
      ( rc = g_ws.h_a.ensure( np * ICP_NRES * 8 ) ) || ( rc = g_ws.h_b.ensure( np * ICP_STATE_WORDS * 4 ) ) ||
      ( rc = g_ws.queue.ensure( cx.total_tiles * 4 ) ) || ( rc = g_ws.queue_count.ensure( np * 4 ) ) )
    return rc;
  L.queue = g_ws.queue.as<int>(); L.queue_count = g_ws.queue_count.as<int>();
  L.solo_stages = handoff_threshold( (long long)cx.total_tiles );
  static const int heavy_streamed = getenv( "RS_HIP_HEAVY_STREAMED" ) ? atoi( getenv( "RS_HIP_HEAVY_STREAMED" ) ) : 400;
  static const int heavy_handoff = getenv( "RS_HIP_HEAVY_HANDOFF" ) ? atoi( getenv( "RS_HIP_HEAVY_HANDOFF" ) ) : 600;
  L.heavy_streamed = heavy_streamed; L.heavy_handoff = heavy_handoff;
  static const int heavy_total = getenv( "RS_HIP_HEAVY_TOTAL" ) ? atoi( getenv( "RS_HIP_HEAVY_TOTAL" ) ) : 400;
  L.heavy_total = heavy_total;
  float* w = g_ws.state.as<float>();
  L.T1 = w; L.active = (int*)( w + np * 16 ); L.T1_prev = w + np * 17;
  L.iters = (int*)( w + np * 33 ); L.err = w + np * 34; L.prev_err = w + np * 35; L.queued = (int*)( w + np * 36 ); L.ticket = (int*)( w + np * 37 );
  L.solve = 0; L.iter_index = 0; L.fixed_iters = 0;
  L.seed = getenv( "RS_HIP_NO_SEED" ) ? 0 : 1;
  L.by_rows = getenv( "RS_HIP_NO_BY_ROWS" ) ? 0 : 1;
  L.bounded_only = getenv( "RS_HIP_NO_BOUNDED_ONLY" ) ? 0 : 1;
  L.cert_r = nullptr; L.cert_dot = nullptr; L.cert_slack = nullptr; L.tgt_nor_max = tgt->nor_max;
  L.m_slot = g_ws.slot.as<int>(); L.m_d2 = g_ws.d2.as<float>(); L.m_dot = g_ws.dot.as<float>();
  L.stat_acc = nullptr;       // set by the align loop (fp64 estimator only)
  L.mom_part = g_ws.mom_part.as<double>(); L.res = g_ws.res.as<double>();
  L.w_explicit = nullptr;
  L.faith = nullptr;
  {
    // one counter for the process (every calling thread's kernels add to it; rs_hip_icp_faith_redone reads it from any thread)
    std::lock_guard<std::mutex> lk( g_faith_redone_mu );
    if( !g_faith_redone.p )
    {
      if( int rc = g_faith_redone.ensure( 64 ) ) return rc;
      HIP_TRY( hipMemset( g_faith_redone.p, 0, 64 ), RS_HIP_E_RUNTIME );
    }
  }
  L.faith_redone = g_faith_redone.as<int>();
  L.faith_guess_scale = (float)g_faith_guess_permille.load() / 1000.0f;
  g_fills.n = 0;      // (fills an earlier call queued and never launched — it failed before its first kernel — are not this call's)
  if( int rcf = icp_fill( g_ws.queue_count.p, 0, np * 4 ) ) return rcf;
  return RS_HIP_OK;
}

// Waves per queued tile in the cooperative kernel.  A short queue is bound by its heaviest tile (8 waves: from the third iteration on the
// certificates have emptied it; launches that skip phase A: while there are few tiles), a long one by throughput (4).  Batches of many
// problems (round 6): object refines keep a long queue in every iteration — the rim of every object has nothing to match — and the
// launch is throughput-bound however late: 512 50 k-point refines (410 k tiles) 33.8 ms per step with 8 waves per tile, 30.2 with 4,
// 28.9 with 2.
inline int icp_coop_waves( const IcpLaunch& L, long long total_tiles, int n_prob, int i )
{
  if( n_prob > 8 && total_tiles >= 131072 ) return 2;
  if( n_prob > 8 && total_tiles >= 32768 ) return 4;
  return L.coop_all ? ( total_tiles <= 1536 ? 8 : 4 ) : ( ( i >= 2 && L.cert_r ) ? 8 : 4 );
}

// buffers of the parallel reference-order estimator for n_prob problems of n_source points
int replay_prepare( ReplayBufs& B, int n_prob, int n_source )
{
  B.n_seg = replay_segments( n_source ); B.n_super = replay_superblocks( n_source );
  const size_t rows = (size_t)n_prob * ICP_NMOM * (size_t)B.n_seg, srows = (size_t)n_prob * ICP_NMOM * (size_t)B.n_super;
  int rc;
  if( ( rc = g_ws.rp_segsum.ensure( rows * 8 ) ) || ( rc = g_ws.rp_guess.ensure( 3 * rows * 8 ) ) || ( rc = g_ws.rp_seg.ensure( rows * replay_seg_bytes() ) ) ||
      ( rc = g_ws.rp_super.ensure( srows * replay_seg_bytes() ) ) ||
      ( rc = g_ws.rp_totals.ensure( (size_t)n_prob * 3 * ICP_NMOM * 8 ) ) || ( rc = g_ws.rp_redone.ensure( (size_t)n_prob * 4 + 64 ) ) ) return rc;
  B.segsum = g_ws.rp_segsum.as<double>(); B.guess = g_ws.rp_guess.as<double>(); B.seg = (ReplaySeg*)g_ws.rp_seg.p; B.super = (ReplaySeg*)g_ws.rp_super.p;
  B.totals = g_ws.rp_totals.as<double>(); B.redone = g_ws.rp_redone.as<int>();
  if( int rcf = icp_fill( g_ws.rp_redone.p, 0, (size_t)n_prob * 4 + 64 ) ) return rcf;
  return RS_HIP_OK;
}

// initial loop state (icp.h:441-442: errors start at 1e6)
int icp_upload_state( IcpCtx& cx, const float* T1s, size_t np )
{
  float* h = g_ws.h_b.as<float>();
  std::memset( h, 0, np * ICP_STATE_WORDS * 4 );
  int* hi = (int*)h;
  for( size_t p = 0; p < np; ++p )
  {
    std::memcpy( h + 16 * p, T1s + 16 * p, 64 ); hi[np * 16 + p] = 1;
    std::memcpy( h + np * 17 + 16 * p, T1s + 16 * p, 64 );
    h[np * 34 + p] = 1e6f; h[np * 35 + p] = 1e6f;
  }
  HIP_TRY( hipMemcpyAsync( g_ws.state.p, h, np * ICP_STATE_WORDS * 4, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  return RS_HIP_OK;
}

// Certificates on (rs_icp_search.hip: icp_certificate): two floats per (problem, source point).
int icp_enable_certificates( IcpCtx& cx )
{
  int rc;
  const size_t words = cx.total_pts;
  if( getenv( "RS_HIP_NO_CERT" ) || !std::isfinite( cx.L.tgt_nor_max ) ) return RS_HIP_OK;
  if( ( rc = g_ws.cert_r.ensure( words * 4 ) ) || ( rc = g_ws.cert_dot.ensure( words * 4 ) ) ) return rc;
  if( int rcf = icp_fill( g_ws.cert_r.p, 0xFF, words * 4 ) ) return rcf;      // NaN: no certificate
  cx.L.cert_r = g_ws.cert_r.as<float>(); cx.L.cert_dot = g_ws.cert_dot.as<float>();
  if( !getenv( "RS_HIP_NO_RANK_CERT" ) )            // read only next to a valid cert_r, written with every fresh one: no initialisation
  {
    if( ( rc = g_ws.cert_slack.ensure( words * 4 ) ) ) return rc;
    cx.L.cert_slack = g_ws.cert_slack.as<float>();
  }
  return RS_HIP_OK;
}

// ---- diagnostics (RS_HIP_DEBUG / RS_HIP_DEBUG_CYCLES): one iteration per chunk, device timers per tile ----
void icp_debug_before( IcpCtx& cx, int n )
{
  static DevBuf dbgbuf;
#ifndef RS_DBG
#define RS_DBG 0
#endif
  if( getenv( "RS_HIP_DEBUG_CYCLES" ) && !RS_DBG )
  { static bool told = false; if( !told ) fprintf( stderr, "[rs_hip] RS_HIP_DEBUG_CYCLES needs the diagnostic build: tools/variant.sh dbg -DRS_DBG=1, then RS_HIP_LIB=.../librescan_hip_dbg.so\n" ); told = true; }
  if( RS_DBG && getenv( "RS_HIP_DEBUG_CYCLES" ) && n == 1 )
  {
    if( dbgbuf.ensure( (size_t)cx.n_waves * 48 + 64 ) ) return;
    (void)hipMemsetAsync( (char*)dbgbuf.p + (size_t)cx.n_waves * 48, 0, 64, g_stream );
    cx.L.dbg = dbgbuf.as<unsigned long long>();
  }
}

void icp_debug_after( IcpCtx& cx, int n_src, int n, int i, float max_dist )
{
  (void)hipStreamSynchronize( g_stream );
  std::vector<int> qc( n ), ms( (size_t)n_src ); std::vector<float> cr( cx.L.cert_r ? (size_t)n_src : 0 );
  (void)hipMemcpy( qc.data(), cx.L.queue_count, (size_t)n * 4, hipMemcpyDeviceToHost );      // (reset when the iteration ends)
  (void)hipMemcpy( ms.data(), cx.L.m_slot, ms.size() * 4, hipMemcpyDeviceToHost );
  if( cx.L.cert_r ) (void)hipMemcpy( cr.data(), cx.L.cert_r, cr.size() * 4, hipMemcpyDeviceToHost );
  size_t unm = 0, cert = 0;
  for( int v : ms ) unm += v < 0;
  for( float v : cr ) cert += v > 0.0f;
  long long st_sum = 0, st_cnt = 0;
  if( cx.L.heavy_out )
  {
    std::vector<int> words( (size_t)cx.n_waves );
    (void)hipMemcpy( words.data(), cx.L.heavy_out + HEAVY_HDR + HEAVY_SLOTS, words.size() * 4, hipMemcpyDeviceToHost );
    for( int v : words ) if( (unsigned)v >> 2 ) { st_sum += (unsigned)v >> 2; ++st_cnt; }
    long long heavy = 0; for( int v : words ) heavy += ( v & 3 ) == 2;
    fprintf( stderr, "[rs_hip icp]   of the queued tiles, handed off as HEAVY bounded tiles (flag 2): %lld\n", heavy );
  }
  fprintf( stderr, "[rs_hip icp] it %d (max_dist %g, coop waves %d): prob 0 queued tiles %d of %d, unmatched %zu of %d, certificates %zu, candidates streamed per tile of phase A %lld\n",
           i, (double)max_dist, cx.L.coop_waves, qc[0], cx.n_waves, unm, n_src, cert, st_cnt ? st_sum / st_cnt : 0ll );
  if( !cx.L.dbg ) return;
  std::vector<unsigned long long> h( (size_t)cx.n_waves * 2 );
  (void)hipMemcpy( h.data(), cx.L.dbg, h.size() * 8, hipMemcpyDeviceToHost );
  struct Row { unsigned long long start, ticks; unsigned uns, handoff, stages, streamed, rank_streamed; };
  std::vector<Row> t;
  unsigned long long sum = 0, ho = 0, tiles_uns = 0, n_rank = 0, t0 = ~0ull, t1 = 0;
  for( int k = 0; k < cx.n_waves; ++k )
  {
    const unsigned long long v = h[2*k+1];
    Row r{ h[2*k], v & 0xfffff, (unsigned)( v >> 57 ), (unsigned)( ( v >> 20 ) & 1 ), (unsigned)( ( v >> 21 ) & 15 ), (unsigned)( ( v >> 25 ) & 0xffff ), (unsigned)( ( v >> 41 ) & 0xffff ) };
    t.push_back( r ); sum += r.ticks; ho += r.uns; tiles_uns += r.uns ? 1 : 0; n_rank += r.rank_streamed ? 1 : 0;
    t0 = std::min( t0, r.start ); t1 = std::max( t1, r.start + r.ticks );
  }
  fprintf( stderr, "[rs_hip dbg] phase A: %.1f us from first wave start to last wave end; unsettled lanes after shell 1: %llu in %llu tiles; rank pass in %llu tiles; mean tile %.1f us\n",
           ( t1 - t0 ) / 100.0, ho, tiles_uns, n_rank, (double)sum / cx.n_waves / 100.0 );
  {
    // waves in flight over time (10 slices) and the tiles that end last
    const double span = (double)( t1 - t0 );
    int live[10] = { 0 };
    for( const Row& r : t ) for( int b = 0; b < 10; ++b ) { const double at = t0 + span * ( b + 0.5 ) / 10; if( r.start <= at && at < r.start + r.ticks ) live[b]++; }
    fprintf( stderr, "[rs_hip dbg]   waves in flight at 5%%,15%%,..95%% of the span:" ); for( int b = 0; b < 10; ++b ) fprintf( stderr, " %d", live[b] ); fprintf( stderr, "\n" );
    std::vector<Row> byend = t; std::sort( byend.begin(), byend.end(), []( const Row& a, const Row& b ) { return a.start + a.ticks > b.start + b.ticks; } );
    for( int k = 0; k < 5 && k < (int)byend.size(); ++k )
      fprintf( stderr, "[rs_hip dbg]   late finisher: starts at %.1f us, runs %.1f us, shells %u, streamed %u, rank-pass %u, handoff %u\n",
               ( byend[k].start - t0 ) / 100.0, byend[k].ticks / 100.0, byend[k].stages, byend[k].streamed, byend[k].rank_streamed, byend[k].handoff );
  }
  std::sort( t.begin(), t.end(), []( const Row& a, const Row& b ) { return a.ticks < b.ticks; } );
  for( double q : { 0.5, 0.9, 0.99, 0.999, 1.0 } )
  {
    const Row& r = t[std::min( t.size() - 1, (size_t)( q * t.size() ) )];
    fprintf( stderr, "[rs_hip dbg]   phase A p%g: %.1f us, shells %u, streamed %u, rank-pass streamed %u, handoff %u\n", q * 100, r.ticks / 100.0, r.stages, r.streamed, r.rank_streamed, r.handoff );
  }
  std::vector<unsigned long long> c( (size_t)qc[0] * 4 );
  (void)hipMemcpy( c.data(), cx.L.dbg + 2 * (size_t)cx.n_waves, c.size() * 8, hipMemcpyDeviceToHost );
  std::vector<std::array<unsigned long long, 4>> rows;
  for( int b = 0; b < qc[0]; ++b ) rows.push_back( { c[4*b], c[4*b+1], c[4*b+2], c[4*b+3] } );
  std::sort( rows.begin(), rows.end() );
  auto pr = [&]( const char* tag, size_t k ) { if( rows.empty() ) return; k = std::min( k, rows.size() - 1 );
    fprintf( stderr, "[rs_hip dbg]   coop %s: %.1f us (setup %.1f, shell k=2 %.1f, full shell %.1f, rest %.1f), wave-0 streamed %llu cand, searching lanes %llu, unmatched %llu\n", tag, rows[k][0] / 100.0,
             ( rows[k][3] & 0xffff ) / 100.0, ( ( rows[k][3] >> 16 ) & 0xffff ) / 100.0, ( ( rows[k][3] >> 32 ) & 0xffff ) / 100.0, ( rows[k][3] >> 48 ) / 100.0,
             rows[k][1], rows[k][2] & 0xff, rows[k][2] >> 8 ); };
  unsigned long long cat[4]; (void)hipMemcpy( cat, cx.L.dbg + 6 * (size_t)cx.n_waves, 32, hipMemcpyDeviceToHost );
  if( RS_DBG >= 2 ) fprintf( stderr, "[rs_hip dbg]   unmatched lanes: skipped by certificate %llu, freshly certified %llu, rank-rejected %llu, loose-band only %llu\n", cat[0], cat[1], cat[2], cat[3] );
  pr( "p10", rows.size() / 10 ); pr( "p50", rows.size() / 2 ); pr( "p90", rows.size() * 9 / 10 ); pr( "p99", rows.size() * 99 / 100 ); pr( "max", rows.size() - 1 );
}

void icp_set_radius( IcpCtx& cx, float max_dist, float tmin )
{
  cx.L.radius = max_dist; cx.L.radius_sq = radius_sq_of( max_dist ); cx.L.gate_tmin = tmin;
  // fixed-point scales of the dist² statistics: r²·2^e1 and r⁴·2^e2 just below 2^36, so that a tile's sum (64 terms)
  // and the sum over 2^22 tiles stay below 2^64, with 36 bits below the radius
  const double r2 = cx.L.radius_sq;
  int e1 = 0, e2 = 0;
  if( r2 > 0.0 && std::isfinite( r2 ) ) { e1 = 35 - std::ilogb( r2 ); e2 = 35 - std::ilogb( r2 * r2 ); }
  e1 = std::max( -900, std::min( 900, e1 ) ); e2 = std::max( -900, std::min( 900, e2 ) );
  cx.L.stat_s1 = std::ldexp( 1.0, e1 ); cx.L.stat_i1 = std::ldexp( 1.0, -e1 );
  cx.L.stat_s2 = std::ldexp( 1.0, e2 ); cx.L.stat_i2 = std::ldexp( 1.0, -e2 );
}

} // namespace

extern "C" {

} // extern "C"

namespace {
// Buffers of the lane-chain estimator for n_prob problems with `rows` source points in all, the largest of max_n: the searches' records,
// the seven totals per problem, the moments' partials (one per 1 024 source points of the largest problem), the update's ticket.
std::atomic<long long> g_lane_seq_addends{ 0 }, g_lane_addends{ 0 };      // (diagnostics: rs_hip_icp_lane_chains_sequential)
int icp_lane_prepare( IcpCtx& cx, ChainBufs& CB, int n_prob, size_t rows, int max_n )
{
  int rc;
  const size_t np = (size_t)n_prob;
  CB.n_seg = chain_segments( max_n ); CB.n_blk = chain_blocks( max_n ); CB.refresh = 0;
  cx.L.n_mom_blocks = CB.n_blk * 4;
  if( ( rc = g_ws.ch_rec.ensure( std::max<size_t>( 1, rows ) * REC_F4 * 16 ) ) || ( rc = g_ws.rp_totals.ensure( np * 3 * ICP_NMOM * 8 ) ) ||
      ( rc = g_ws.rp_redone.ensure( np * 4 + 64 ) ) || ( rc = g_ws.ch_done.ensure( np * 8 ) ) ||
      ( rc = g_ws.mom_part.ensure( np * (size_t)cx.L.n_mom_blocks * ICP_NMOM * 8 ) ) ||
      ( rc = g_ws.ch_segsum.ensure( 8 * ( rows + 4 * np ) * 4 ) ) ) return rc;
  CB.addends = g_ws.ch_segsum.as<float>();      // (the grid chains' buffer: the two estimators never run in one call)
  if( getenv( "RS_HIP_LANE_DEBUG" ) )
  {
    if( ( rc = g_ws.ch_dbg.ensure( np * CH_ROWS * 16 ) ) ) return rc;
    HIP_TRY( hipMemsetAsync( g_ws.ch_dbg.p, 0, np * CH_ROWS * 16, g_stream ), RS_HIP_E_RUNTIME );
    CB.dbg = g_ws.ch_dbg.as<int>();
  }
  if( int rcf = icp_fill( g_ws.rp_redone.p, 0, np * 4 + 64 ) ) return rcf;
  if( int rcf = icp_fill( g_ws.ch_done.p, 0, np * 8 ) ) return rcf;
  CB.totals = g_ws.rp_totals.as<double>(); CB.resolved = g_ws.rp_redone.as<int>(); CB.done = g_ws.ch_done.as<int>(); CB.failed = nullptr;
  cx.L.rec = (float4*)g_ws.ch_rec.p; cx.L.mom_part = g_ws.mom_part.as<double>();
  g_lane_addends.fetch_add( (long long)rows * CH_ROWS );      // (per iteration; the diagnostic divides by what it counted the same way)
  return RS_HIP_OK;
}
// (diagnostics) addends the walks of this call added one by one, over all its iterations
void icp_lane_account( const ChainBufs& CB, int n_prob )
{
  std::vector<int> r( (size_t)n_prob );
  if( hipMemcpy( r.data(), CB.resolved, r.size() * 4, hipMemcpyDeviceToHost ) != hipSuccess ) return;
  long long t = 0; for( int v : r ) t += v;
  g_lane_seq_addends.fetch_add( t );
  if( CB.dbg )
  {
    std::vector<int> d( (size_t)n_prob * CH_ROWS * 4 );
    if( hipMemcpy( d.data(), CB.dbg, d.size() * 4, hipMemcpyDeviceToHost ) != hipSuccess ) return;
    for( int p = 0; p < std::min( n_prob, 4 ); ++p )
      for( int row = 0; row < CH_ROWS; ++row )
      {
        const int* o = d.data() + ( (size_t)p * CH_ROWS + row ) * 4;
        fprintf( stderr, "[rs_hip lane] problem %d chain %d: %d points, walk %.2f us, %d attempts (%d stretches), %d addends one by one\n", p, row, o[3], o[0] / 100.0, o[1], ( o[3] + 511 ) / 512, o[2] );
      }
  }
}

enum { ICP_CHAINS_GAVE_UP = -1000 };      // (internal) a centroid chain's walk gave the problem up: again, the seven sums by pass 2 of the replay

enum { ICP_STOP_EDGE = -1001 };           // (internal) a stop test of some problem came within the guard of its threshold: *edge lists them

int icp_align_batch_impl( const rs_hip_cloud_t* source, const rs_hip_cloud_t* target,
                          float* T1s, int32_t n, const float* T2, float max_dist, float max_angle,
                          int32_t max_iter, int32_t fixed_iters, float* errs, int32_t* iters, int centroid_mode,
                          bool force_bits = false /* the reference's own order (sequential / parallel) whatever the thresholds say */,
                          std::vector<int>* edge = nullptr /* out: problems whose stop test was decided inside the guard (their results are written all the same) */ )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !T1s || !T2 || !errs ) { set_err( "icp_align: null argument" ); return RS_HIP_E_ARG; }
  IcpCtx cx;
  if( ( rc = icp_prepare( cx, source, target, n, T2 ) ) ) return rc;
  const float tmin = icp_gate_threshold( max_angle );
  if( source->n == 0 ) { for( int p = 0; p < n; ++p ) { errs[p] = 1e6f; if( iters ) iters[p] = 1; } return RS_HIP_OK; }   // n_corrs == 0 on the first search
  if( ( rc = icp_enable_certificates( cx ) ) ) return rc;
  if( ( rc = icp_upload_state( cx, T1s, (size_t)n ) ) ) return rc;
  const bool ref_order = force_bits ? source->n <= 65536 : source->n <= g_ref_order_below.load();
  const bool replay = !ref_order && ( force_bits ? source->n <= 262144 : source->n <= g_replay_below.load() );
  const bool exact_centroids = !ref_order && !replay && centroid_mode != 0;
  ReplayBufs RB{};
  ChainBufs CB{};
  const bool lane = exact_centroids && icp_takes_lane_chains( source->n, n );      // (object-sized: one wave per chain, launch_icp_lane_chains)
  // the stop test's guard: only where the estimator is not the reference's order AND a bit-exact one exists to run the problem again with
  cx.L.stop_guard = ( exact_centroids && !fixed_iters && edge && source->n <= 262144 ) ? g_stop_guard.load() : 0.0f;
  const bool chains = exact_centroids && !lane && centroid_mode == 1;      // (2: the same sums through pass 2 of the replay — the cross-check, and what a problem the chains give up is run with)
  // (2 reads the searches' records like the chains do — RS_HIP_REPLAY2_GATHER=1: from k_icp_faith_gather's arrays, as up to round 3)
  static const bool replay2_gather = getenv( "RS_HIP_REPLAY2_GATHER" ) != nullptr;
  const bool from_records = exact_centroids && !chains && !lane && !replay2_gather;
  if( ref_order || replay || ( exact_centroids && !chains && !lane ) )
  {
    if( !from_records )
    {
      if( ( rc = g_ws.faith.ensure( (size_t)n * FAITH_REC * (size_t)source->n * 4 ) ) ) return rc;
      cx.L.faith = g_ws.faith.as<float>();
    }
    if( ( replay || exact_centroids ) && ( rc = replay_prepare( RB, n, source->n ) ) ) return rc;
  }
  if( chains )
  {
    if( ( rc = g_ws.rp_totals.ensure( (size_t)n * 3 * ICP_NMOM * 8 ) ) || ( rc = g_ws.rp_redone.ensure( (size_t)n * 4 + 64 ) ) ) return rc;
    RB.totals = g_ws.rp_totals.as<double>(); RB.redone = g_ws.rp_redone.as<int>();
    if( int rcf = icp_fill( g_ws.rp_redone.p, 0, (size_t)n * 4 + 64 ) ) return rcf;
  }
  if( lane )
  {
    if( ( rc = icp_lane_prepare( cx, CB, n, (size_t)n * (size_t)source->n, source->n ) ) ) return rc;
    RB.totals = CB.totals;
  }
  if( exact_centroids ) { cx.L.exact_centroids = 1; cx.L.centroid_totals = RB.totals; }
  if( chains )
  {
    CB.n_seg = chain_segments( source->n ); CB.n_blk = chain_blocks( source->n );
    const size_t rows = (size_t)n * CH_ROWS;
    if( ( rc = g_ws.ch_rec.ensure( (size_t)n * (size_t)source->n * REC_F4 * 16 ) ) || ( rc = g_ws.ch_segsum.ensure( rows * CB.n_seg * 8 ) ) ||
        ( rc = g_ws.ch_prefix.ensure( rows * CB.n_blk * 4 * 8 ) ) || ( rc = g_ws.ch_seg.ensure( rows * CB.n_seg * sizeof( ChainRec ) ) ) ||
        ( rc = g_ws.ch_blk.ensure( rows * CB.n_blk * sizeof( ChainRec ) ) ) ||
        ( rc = g_ws.ch_guess.ensure( rows * CB.n_seg * 4 ) ) ) return rc;
    CB.segsum = g_ws.ch_segsum.as<double>(); CB.blksum = g_ws.ch_prefix.as<double>(); CB.seg = (ChainRec*)g_ws.ch_seg.p; CB.blk = (ChainRec*)g_ws.ch_blk.p; CB.guess = g_ws.ch_guess.as<int>();
    CB.totals = RB.totals; CB.resolved = RB.redone;
    if( ( rc = g_ws.ch_done.ensure( (size_t)n * 8 ) ) ) return rc;
    if( int rcf = icp_fill( g_ws.ch_done.p, 0, (size_t)n * 8 ) ) return rcf;
    CB.done = g_ws.ch_done.as<int>(); CB.failed = CB.done + n;
    if( getenv( "RS_HIP_CHAIN_DEBUG" ) )
    {
      if( ( rc = g_ws.ch_dbg.ensure( rows * ( 4 + 64 * 8 ) * 4 ) ) ) return rc;
      CB.dbg = g_ws.ch_dbg.as<int>(); CB.dbg_reps = std::max( 1, atoi( getenv( "RS_HIP_CHAIN_DEBUG" ) ) );
      // the walk writes its mismatch marker only where it finds 0: a fresh (or an earlier call's) buffer must not speak for this one
      HIP_TRY( hipMemsetAsync( g_ws.ch_dbg.p, 0, rows * ( 4 + 64 * 8 ) * 4, g_stream ), RS_HIP_E_RUNTIME );
      if( !getenv( "RS_HIP_CHAIN_DEBUG_NO_SELFCHECK" ) )      // (the self-check redoes every step the slow way: without it the stamps are the walk's real times)
      {
        if( ( rc = g_ws.ch_chk.ensure( rows * ( 4 + 3 * 4096 ) * 4 ) ) ) return rc;
        HIP_TRY( hipMemsetAsync( g_ws.ch_chk.p, 0, rows * ( 4 + 3 * 4096 ) * 4, g_stream ), RS_HIP_E_RUNTIME );
        CB.chk = g_ws.ch_chk.as<int>();
      }
    }
    cx.L.rec = (float4*)g_ws.ch_rec.p;
    cx.L.n_mom_blocks = CB.n_blk * 4;              // k_chain_moments: one workgroup, one partial, per quarter block (1 024 source points)
    if( ( rc = g_ws.mom_part.ensure( (size_t)n * CB.n_blk * 4 * ICP_NMOM * 8 ) ) ) return rc;
    cx.L.mom_part = g_ws.mom_part.as<double>();
  }
  if( from_records )
  {
    // the searches' records, the chains' moment kernel and update (launch_icp_exact_centroids_from_records)
    CB.n_seg = chain_segments( source->n ); CB.n_blk = chain_blocks( source->n ); CB.refresh = 0;
    if( ( rc = g_ws.ch_rec.ensure( (size_t)n * (size_t)source->n * REC_F4 * 16 ) ) || ( rc = g_ws.ch_done.ensure( (size_t)n * 8 ) ) ||
        ( rc = g_ws.mom_part.ensure( (size_t)n * CB.n_blk * 4 * ICP_NMOM * 8 ) ) ) return rc;
    if( int rcf = icp_fill( g_ws.ch_done.p, 0, (size_t)n * 8 ) ) return rcf;
    CB.done = g_ws.ch_done.as<int>();
    cx.L.rec = (float4*)g_ws.ch_rec.p;
    cx.L.n_mom_blocks = CB.n_blk * 4;
    cx.L.mom_part = g_ws.mom_part.as<double>();
  }
  if( !ref_order && !replay )
  {
    if( int rcf = icp_fill( g_ws.stat_acc.p, 0, (size_t)n * STAT_SHARDS * 4 * 8 ) ) return rcf;
    cx.L.stat_acc = g_ws.stat_acc.as<unsigned long long>();
  }
  const size_t heavy_words = cx.heavy_words;
  const bool reorder = !getenv( "RS_HIP_NO_LPT" );
  if( reorder )
  {
    if( ( rc = g_ws.order_a.ensure( heavy_words * 4 ) ) || ( rc = g_ws.order_b.ensure( heavy_words * 4 ) ) ) return rc;
    if( int rcf = icp_fill( g_ws.order_a.p, 0, heavy_words * 4 ) ) return rcf;
    if( int rcf = icp_fill( g_ws.order_b.p, 0, heavy_words * 4 ) ) return rcf;
  }

  // The whole iteration runs on the device (search, statistics, moments, solve, pose update, stop
  // tests), so iterations are enqueued back to back and the host looks at the state only once per
  // chunk; problems that stopped inside a chunk turn the rest of its launches into no-ops
  // (every kernel returns at once for an inactive problem).
  const bool debug = getenv( "RS_HIP_DEBUG" ) || getenv( "RS_HIP_DEBUG_CYCLES" );
  static const int chunk_env = getenv( "RS_HIP_ICP_CHUNK" ) ? atoi( getenv( "RS_HIP_ICP_CHUNK" ) ) : 4;
  const size_t np = (size_t)n, state_bytes = np * ICP_STATE_WORDS * 4;
  float* hS = g_ws.h_b.as<float>();
  const int* hActive = (const int*)( hS + np * 16 );
  const long long total_tiles = (long long)cx.total_tiles;
  static const long long coop_all_below = getenv( "RS_HIP_COOP_ALL_BELOW" ) ? atoll( getenv( "RS_HIP_COOP_ALL_BELOW" ) ) : 4096;
  static const int coop_waves_forced = getenv( "RS_HIP_COOP_WAVES" ) ? atoi( getenv( "RS_HIP_COOP_WAVES" ) ) : 0;
  static const int chain_refresh = std::max( 0, getenv( "RS_HIP_CHAIN_REFRESH" ) ? atoi( getenv( "RS_HIP_CHAIN_REFRESH" ) ) : 0 );
  cx.L.solve = 1; cx.L.fixed_iters = fixed_iters ? 1 : 0;
  // (see g_early_plain.  Scan-sized sources only: the eight 50 k-point refines of bench.py --scaling strong end 2.2e-6 from the reference with the
  //  chains throughout, 8.7e-6 with three plain iterations, 2.0e-5 with seven or eight — an object refine contracts more slowly than a scan-to-scan fit)
  const int n_plain = ( chains || from_records ) ? icp_plain_iterations( source->n, max_iter, fixed_iters != 0 ) : 0;
  if( ( rc = icp_fills_flush() ) ) return rc;      // everything the call's first kernels expect zeroed, in one launch
  ProfChain prof;
  for( int i = 0; i < max_iter; )                                       // icp.h:444
  {
    // (the stop test looks at i > 5, icp.h:489: the first seven iterations go out in one piece, then chunk_env at a time — an iteration enqueued
    //  behind the one that stopped is six empty launches, a look at the state a copy and a synchronisation)
    const int chunk = ( debug || g_icp_trace ) ? 1 : ( fixed_iters ? max_iter - i : std::min( i == 0 ? std::max( 7, chunk_env ) : std::max( 1, chunk_env ), max_iter - i ) );
    for( int c = 0; c < chunk; ++c, ++i )
    {
      icp_set_radius( cx, max_dist, tmin );
      cx.L.iter_index = i;
      cx.L.warm = ( i > 0 && !getenv( "RS_HIP_NO_WARM" ) ) ? 1 : 0;   // every active problem wrote m_slot in iteration i-1
      // a short queue is bound by its heaviest tile: from the third iteration on the certificates have
      // emptied it (small batches: always)
      // launches of a few thousand tiles go straight to the cooperative kernel (launch_icp_corr); it is bound by its
      // heaviest tile while the list is short (8 waves per tile), by throughput beyond (4)
      cx.L.coop_all = total_tiles <= coop_all_below ? 1 : 0;
      cx.L.coop_waves = icp_coop_waves( cx.L, total_tiles, n, i );
      if( coop_waves_forced ) cx.L.coop_waves = coop_waves_forced;
      if( reorder )
      {
        cx.L.heavy_in = i == 0 ? nullptr : ( ( i & 1 ) ? g_ws.order_a.as<int>() : g_ws.order_b.as<int>() );
        cx.L.heavy_out = ( i & 1 ) ? g_ws.order_b.as<int>() : g_ws.order_a.as<int>();
      }
      if( debug ) icp_debug_before( cx, n );
      prof.mark( "nn_icp" ); launch_icp_corr( cx.L, g_stream );
      if( debug ) icp_debug_after( cx, source->n, n, i, max_dist );
      prof.mark( "icp_moments" );
      if( replay ) launch_icp_replay( cx.L, RB, g_stream );
      else if( chains )
      {
        // the binade guesses of the chains' records: made from the sums of the iteration before (k_chain_walk_and_moments) — in the first
        // iteration from its own, so the records and the walks wait for the moments there (RS_HIP_CHAIN_REFRESH=k: in every k-th as well)
        if( i < n_plain )
        {
          IcpLaunch Lp = cx.L; Lp.exact_centroids = 0;
          launch_icp_plain_from_records( Lp, CB, g_stream );
        }
        else
        {
          CB.refresh = ( i == n_plain || ( chain_refresh > 0 && ( i % chain_refresh ) == 0 ) ) ? 1 : 0;
          launch_icp_chain_centroids( cx.L, CB, g_stream );
        }
      }
      else if( lane ) launch_icp_lane_chains( cx.L, CB, g_stream );
      else if( from_records && i < n_plain ) { IcpLaunch Lp = cx.L; Lp.exact_centroids = 0; launch_icp_plain_from_records( Lp, CB, g_stream ); }
      else if( from_records ) launch_icp_exact_centroids_from_records( cx.L, RB, CB, g_stream );
      else if( exact_centroids ) launch_icp_exact_centroids( cx.L, RB, g_stream );
      else if( cx.L.faith ) launch_icp_faithful( cx.L, g_stream ); else launch_icp_moments( cx.L, g_stream );
      double nd = max_dist * 0.95;                                      // icp.h:493
      max_dist = (float)( nd > 0.05 ? nd : 0.05 );
    }
    prof.mark( nullptr );
    HIP_TRY( hipMemcpyAsync( hS, g_ws.state.p, state_bytes, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
    HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
    if( g_icp_trace && n == 1 && i >= 1 && i - 1 < g_icp_trace_cap ) g_icp_trace[i - 1] = hS[np * 34];      // the error after iteration i - 1
    int n_active = 0;
    for( int p = 0; p < n; ++p ) n_active += hActive[p] ? 1 : 0;
    if( n_active == 0 ) break;
  }
  if( lane ) icp_lane_account( CB, n );
  if( exact_centroids && getenv( "RS_HIP_DEBUG_TOTALS" ) )      // the seven centroid sums of the last iteration, as the estimator used them (bits)
  {
    std::vector<double> t( (size_t)n * 3 * ICP_NMOM );
    (void)hipMemcpy( t.data(), RB.totals, t.size() * 8, hipMemcpyDeviceToHost );
    for( int p = 0; p < n; ++p )
    {
      fprintf( stderr, "[rs_hip totals] mode %d problem %d:", centroid_mode, p );
      for( int r = 0; r < CH_ROWS; ++r ) { const float f = (float)t[( (size_t)p * 3 + 1 ) * ICP_NMOM + r]; uint32_t u; std::memcpy( &u, &f, 4 ); fprintf( stderr, " %08x", u ); }
      fprintf( stderr, "\n" );
    }
  }
  if( CB.dbg )      // RS_HIP_CHAIN_DEBUG: which segments the last iteration's chain walks added up addend by addend, and why their record did not fit
  {
    std::vector<int> d( (size_t)CH_ROWS * ( 4 + 64 * 8 ) );
    (void)hipMemcpy( d.data(), CB.dbg, d.size() * 4, hipMemcpyDeviceToHost );
    for( int r = 0; r < CH_ROWS; ++r )
    {
      const int* q = d.data() + (size_t)r * ( 4 + 64 * 8 );
      fprintf( stderr, "[rs_hip chains] chain %d: %d segments added one by one (%d with their addends fetched ahead), %d steps by one record, %d wave-wide scans%s; fetches done after %.2f us, walk after %.2f us\n", r, q[0] & 0xffff, q[2] & 0xffff, q[2] >> 16, ( q[0] >> 16 ) & 0x3fff,
               ( q[0] >> 30 ) ? " (STUCK)" : "", q[1] / 100.0, q[3] / 100.0 );
      fprintf( stderr, "   cut + block 0 after %.2f us, block forecasts after %.2f us, segment records + forecasts after %.2f us; %d shader cycles in the walk itself\n", q[4 + 63 * 8] / 100.0, q[4 + 63 * 8 + 1] / 100.0, q[4 + 63 * 8 + 2] / 100.0, q[4 + 63 * 8 + 3] );
      if( q[4 + 63 * 8 + 4] )
        fprintf( stderr, "   step %d taken by its record DIFFERS from the same step by scans: record es %x lo %d hi %d D %d tp %x | start %08x -> by scans %08x, by record %08x | kind type %d chunk %d block %d from %d to %d\n",
                 q[4 + 63 * 8 + 4] - 1, q[4 + 63 * 8 + 5], q[4 + 63 * 8 + 6], q[4 + 63 * 8 + 7], q[4 + 62 * 8], q[4 + 62 * 8 + 1], (unsigned)q[4 + 62 * 8 + 2], (unsigned)q[4 + 62 * 8 + 3], (unsigned)q[4 + 62 * 8 + 4],
                 q[4 + 62 * 8 + 5] & 3, ( q[4 + 62 * 8 + 5] >> 2 ) & 7, ( q[4 + 62 * 8 + 5] >> 5 ) & 63, ( q[4 + 62 * 8 + 5] >> 11 ) & 63, ( q[4 + 62 * 8 + 5] >> 17 ) & 127 );
      for( int k = 0; k < std::min( q[0] & 0xffff, 62 ); ++k )
      {
        const int* e = q + 4 + 8 * k; const unsigned sb = (unsigned)e[1];
        fprintf( stderr, "   segment %6d (at %6.2f us): value exp %3u mantissa %8u sign %u | %d cycles for the 64 adds | %s, %s\n", e[0], e[7] / 100.0, ( sb >> 23 ) & 255u,
                 ( sb & 0x7fffffu ) | 0x800000u, sb >> 31, e[3],
                 ( e[2] & 2 ) ? "block forecast" : "block NOT forecast", ( e[2] & 1 ) ? "addends ahead" : "addends fetched now" );
      }
    }
  }
  if( chains )
  {
    std::vector<int> failed( (size_t)n );
    HIP_TRY( hipMemcpy( failed.data(), CB.failed, (size_t)n * 4, hipMemcpyDeviceToHost ), RS_HIP_E_RUNTIME );
    for( int p = 0; p < n; ++p ) if( failed[p] ) { g_chains_gave_up.fetch_add( 1 ); return ICP_CHAINS_GAVE_UP; }      // (nothing written to the caller's arrays yet)
  }
  if( CB.chk )
  {
    std::vector<int> c( (size_t)CH_ROWS * ( 4 + 3 * 4096 ) );
    (void)hipMemcpy( c.data(), CB.chk, c.size() * 4, hipMemcpyDeviceToHost );
    for( int r = 0; r < CH_ROWS; ++r )
    {
      const int* q = c.data() + (size_t)r * ( 4 + 3 * 4096 );
      if( q[1] < 0 ) { fprintf( stderr, "[rs_hip chains] chain %d self-check: %d steps, all as the plain sum (%08x); %d segments end in another binade than they start in\n", r, q[0], (unsigned)q[3], q[2] ); continue; }
      const int k = q[1];
      auto show = [&]( int j ) { const int* e = q + 4 + 3 * j; const int kind = e[2]; fprintf( stderr, "      step %d: ends at segment %d with %08x; type %d chunk %d block %d from %d to %d slot %d, %s\n", j, e[0], (unsigned)e[1], kind & 3, ( kind >> 2 ) & 7, ( kind >> 5 ) & 63, ( kind >> 11 ) & 63, ( kind >> 17 ) & 127, ( kind >> 24 ) & 63, ( kind >> 30 ) & 1 ? "by its record" : "by scans / addend by addend" ); };
      fprintf( stderr, "[rs_hip chains] chain %d self-check: step %d of %d differs from the plain sum (%08x there)\n", r, k, q[0], (unsigned)q[2] );
      if( k > 0 ) show( k - 1 );
      show( k );
    }
  }
  const int* hIters = (const int*)( hS + np * 33 );
  for( int p = 0; p < n; ++p ) { std::memcpy( T1s + 16 * p, hS + 16 * p, 64 ); errs[p] = hS[np * 34 + p]; if( iters ) iters[p] = hIters[p]; }
  if( edge && cx.L.stop_guard > 0.0f )
  {
    const int* hEdge = (const int*)( hS + np * 37 );
    for( int p = 0; p < n; ++p ) if( hEdge[p] ) edge->push_back( p );
    if( !edge->empty() ) return ICP_STOP_EDGE;
  }
  return RS_HIP_OK;
}
} // namespace

extern "C" {

int rs_hip_icp_align_batch( const rs_hip_cloud_t* source, const rs_hip_cloud_t* target,
                            float* T1s, int32_t n, const float* T2, float max_dist, float max_angle,
                            int32_t max_iter, int32_t fixed_iters, float* errs, int32_t* iters )
{
  // Scan-sized sources: the reference's centroid sums by the grid chains — which give a problem up when a sum keeps changing
  // binade (coordinates that straddle the origin in a cancelling order: rs_icp_estimate.hip, chain_walk_row); the batch is then run
  // again with those sums by pass 2 of the replay: the same bits, 0.9 ms per iteration at a million points instead of 0.09.
  // The estimators of large sources keep per-point records per problem (48 B with the chains, 44 B + the replay's rows when they
  // give up): many start poses of a whole scan are run in slices of problems whose records stay below RS_HIP_ICP_BATCH_BYTES
  // (default 4 GB) — the problems are independent, so the results are those of the one batch.
  static const double cap = getenv( "RS_HIP_ICP_BATCH_BYTES" ) ? atof( getenv( "RS_HIP_ICP_BATCH_BYTES" ) ) : 4e9;
  const bool per_point_records = source && source->n > g_ref_order_below.load();
  // (96 B per point: the 48-byte records + what a slice that falls back to the replay adds — its segment rows, ~35 / 128 x ( 8 + 24 + sizeof( ReplaySeg ) ) ≈ 48 B per point)
  const int slice = per_point_records ? std::max( 1, (int)std::min<double>( (double)std::max( n, 1 ), cap / ( 96.0 * (double)std::max( source->n, 1 ) ) ) ) : std::max( n, 1 );
  for( int p0 = 0; p0 < std::max( n, 1 ); p0 += slice )
  {
    const int np = std::min( slice, n - p0 );
    float* T = T1s ? T1s + 16 * (size_t)p0 : nullptr; float* e = errs ? errs + p0 : nullptr; int32_t* it = iters ? iters + p0 : nullptr;
    // (a source whose chains gave up lately: the attempt — a cold search, the iterations up to the give-up, the launches that idle
    //  through the rest of the call: ~1.4 ms at a million points — is skipped; every 16th such call tries the chains again)
    int mode = g_exact_centroids.load();
    if( mode == 1 && per_point_records && g_chains_retry_after.load() > 0 && source->chains_wander.load() > 0 ) { source->chains_wander.fetch_sub( 1 ); mode = 2; }
    // (the start poses are kept: a problem whose stop test falls inside the guard is run again from its own)
    std::vector<float> T_in( T, T + 16 * (size_t)np );
    std::vector<int> edge;
    int rc = icp_align_batch_impl( source, target, T, np, T2, max_dist, max_angle, max_iter, fixed_iters, e, it, mode, false, &edge );
    // (a problem of the slice whose chains gave up: that slice again, its seven sums by pass 2 of the replay — nothing of it was written yet)
    if( rc == ICP_CHAINS_GAVE_UP )
    {
      source->chains_wander.store( g_chains_retry_after.load() );
      edge.clear();
      rc = icp_align_batch_impl( source, target, T, np, T2, max_dist, max_angle, max_iter, fixed_iters, e, it, 2, false, &edge );
    }
    if( rc == ICP_STOP_EDGE )
    {
      // those problems again, in the reference's own order: its decisions, its bits
      for( int q : edge )
      {
        std::memcpy( T + 16 * (size_t)q, T_in.data() + 16 * (size_t)q, 64 );
        int rc2 = icp_align_batch_impl( source, target, T + 16 * (size_t)q, 1, T2, max_dist, max_angle, max_iter, fixed_iters, e + q, it ? it + q : nullptr, mode, true );
        if( rc2 ) return rc2;
        g_stop_guard_redone.fetch_add( 1 );
      }
      rc = RS_HIP_OK;
    }
    if( rc ) return rc;
  }
  return RS_HIP_OK;
}

// The per-placement refine loop as one call (lib/rs/rs_database.h:220-230, apps/pose_proposal/main.cpp:190-202 with a different
// object per proposal): problem p aligns sources[p] to the target from T1s[p].  Sources within the reference-order estimator's
// range (rs_hip_icp_reference_order_below: every call site of the reference) run as ONE batch — grid.y = problem, the kernels bind
// their problem's source view on the device (rs_icp.h: icp_bind), so the sequential chains of all problems run side by side;
// every problem's result is what rs_hip_icp_align returns for it alone, bit for bit.  A batch with a larger source is run
// problem by problem (their estimators are built for one source per launch).
// which estimator rs_hip_icp_align would give a source of n points: 0 the reference's order, 1 the lane chains, 2 anything else
static int icp_estimator_class( int n )
{
  if( n <= g_ref_order_below.load() ) return 0;
  if( n <= g_replay_below.load() ) return 2;
  if( g_exact_centroids.load() != 0 && n <= g_lane_below.load() ) return 1;
  return 2;
}

static int icp_align_multi_group( const rs_hip_cloud_t* const* sources, const rs_hip_cloud_t* target, float* T1s, int32_t n, const float* T2,
                                  float max_dist, float max_angle, int32_t max_iter, int32_t fixed_iters, float* errs, int32_t* iters, bool lane );

int rs_hip_icp_align_multi( const rs_hip_cloud_t* const* sources, const rs_hip_cloud_t* target,
                            float* T1s, int32_t n, const float* T2, float max_dist, float max_angle,
                            int32_t max_iter, int32_t fixed_iters, float* errs, int32_t* iters )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !sources || !T1s || !T2 || !errs || n < 0 ) { set_err( "icp_align_multi: bad arguments" ); return RS_HIP_E_ARG; }
  if( n == 0 ) return RS_HIP_OK;
  // Every problem gets the estimator its own rs_hip_icp_align would (icp_estimator_class), so that its result is that call's bit for
  // bit: the problems within the reference-order range run as one batch (k_icp_faithful), those within the lane chains' range as
  // another (launch_icp_lane_chains), the rest — estimators built for one source per launch — one by one.
  std::vector<int> group[3];
  for( int p = 0; p < n; ++p )
  {
    if( !sources[p] ) { set_err( "icp_align_multi: null source" ); return RS_HIP_E_ARG; }
    group[sources[p]->n > 0 ? icp_estimator_class( sources[p]->n ) : 2].push_back( p );
  }
  for( int p : group[2] )
    if( ( rc = rs_hip_icp_align_batch( sources[p], target, T1s + 16 * p, 1, T2, max_dist, max_angle, max_iter, fixed_iters, errs + p, iters ? iters + p : nullptr ) ) ) return rc;
  for( int cls = 0; cls < 2; ++cls )
  {
    const std::vector<int>& g = group[cls];
    if( g.empty() ) continue;
    if( g.size() == 1 )
    {
      const int p = g[0];
      if( ( rc = rs_hip_icp_align_batch( sources[p], target, T1s + 16 * p, 1, T2, max_dist, max_angle, max_iter, fixed_iters, errs + p, iters ? iters + p : nullptr ) ) ) return rc;
      continue;
    }
    if( (int)g.size() == n )      // (the common case: one class, in place)
      return icp_align_multi_group( sources, target, T1s, n, T2, max_dist, max_angle, max_iter, fixed_iters, errs, iters, cls == 1 );
    std::vector<const rs_hip_cloud_t*> src( g.size() ); std::vector<float> T( 16 * g.size() ), e( g.size() ); std::vector<int32_t> it( g.size() );
    for( size_t k = 0; k < g.size(); ++k ) { src[k] = sources[g[k]]; std::memcpy( T.data() + 16 * k, T1s + 16 * g[k], 64 ); }
    if( ( rc = icp_align_multi_group( src.data(), target, T.data(), (int)g.size(), T2, max_dist, max_angle, max_iter, fixed_iters, e.data(), it.data(), cls == 1 ) ) ) return rc;
    for( size_t k = 0; k < g.size(); ++k ) { std::memcpy( T1s + 16 * g[k], T.data() + 16 * k, 64 ); errs[g[k]] = e[k]; if( iters ) iters[g[k]] = it[k]; }
  }
  return RS_HIP_OK;
}

// n >= 2 problems of ONE estimator class in the same launches: grid.y = problem, the kernels bind their problem's view on the device.
static int icp_align_multi_group( const rs_hip_cloud_t* const* sources, const rs_hip_cloud_t* target, float* T1s, int32_t n, const float* T2,
                                  float max_dist, float max_angle, int32_t max_iter, int32_t fixed_iters, float* errs, int32_t* iters, bool lane )
{
  int rc;
  IcpCtx cx;
  if( ( rc = icp_prepare( cx, nullptr, target, n, T2, sources ) ) ) return rc;
  const float tmin = icp_gate_threshold( max_angle );
  if( ( rc = icp_enable_certificates( cx ) ) ) return rc;
  if( ( rc = icp_upload_state( cx, T1s, (size_t)n ) ) ) return rc;
  ChainBufs CB{};
  std::vector<float> T_in;
  const float max_dist_in = max_dist;          // (the loop below shrinks max_dist, icp.h:493: a problem run again starts from the caller's)
  if( lane )
  {
    if( ( rc = icp_lane_prepare( cx, CB, n, cx.total_pts, cx.L.max_n ) ) ) return rc;
    cx.L.exact_centroids = 1; cx.L.centroid_totals = CB.totals;
    if( !fixed_iters ) { cx.L.stop_guard = g_stop_guard.load(); T_in.assign( T1s, T1s + 16 * (size_t)n ); }      // (see icp_align_batch_impl)
    if( int rcf = icp_fill( g_ws.stat_acc.p, 0, (size_t)n * STAT_SHARDS * 4 * 8 ) ) return rcf;
    cx.L.stat_acc = g_ws.stat_acc.as<unsigned long long>();
  }
  else
  {
    if( ( rc = g_ws.faith.ensure( (size_t)FAITH_REC * cx.total_pts * 4 ) ) ) return rc;
    cx.L.faith = g_ws.faith.as<float>();
  }
  const bool reorder = !getenv( "RS_HIP_NO_LPT" );
  if( reorder )
  {
    if( ( rc = g_ws.order_a.ensure( cx.heavy_words * 4 ) ) || ( rc = g_ws.order_b.ensure( cx.heavy_words * 4 ) ) ) return rc;
    if( int rcf = icp_fill( g_ws.order_a.p, 0, cx.heavy_words * 4 ) ) return rcf;
    if( int rcf = icp_fill( g_ws.order_b.p, 0, cx.heavy_words * 4 ) ) return rcf;
  }
  // the loop of icp_align_batch_impl, reference-order estimator only
  static const int chunk_env = getenv( "RS_HIP_ICP_CHUNK" ) ? atoi( getenv( "RS_HIP_ICP_CHUNK" ) ) : 4;
  static const long long coop_all_below = getenv( "RS_HIP_COOP_ALL_BELOW" ) ? atoll( getenv( "RS_HIP_COOP_ALL_BELOW" ) ) : 4096;
  static const int coop_waves_forced = getenv( "RS_HIP_COOP_WAVES" ) ? atoi( getenv( "RS_HIP_COOP_WAVES" ) ) : 0;
  const size_t np = (size_t)n, state_bytes = np * ICP_STATE_WORDS * 4;
  float* hS = g_ws.h_b.as<float>();
  const int* hActive = (const int*)( hS + np * 16 );
  const long long total_tiles = (long long)cx.total_tiles;
  cx.L.solve = 1; cx.L.fixed_iters = fixed_iters ? 1 : 0;
  if( ( rc = icp_fills_flush() ) ) return rc;
  ProfChain prof;
  for( int i = 0; i < max_iter; )                                       // icp.h:444
  {
    const int chunk = g_icp_trace ? 1 : ( fixed_iters ? max_iter - i : std::min( i == 0 ? std::max( 7, chunk_env ) : std::max( 1, chunk_env ), max_iter - i ) );
    for( int c = 0; c < chunk; ++c, ++i )
    {
      icp_set_radius( cx, max_dist, tmin );
      cx.L.iter_index = i;
      cx.L.warm = ( i > 0 && !getenv( "RS_HIP_NO_WARM" ) ) ? 1 : 0;
      cx.L.coop_all = total_tiles <= coop_all_below ? 1 : 0;
      cx.L.coop_waves = icp_coop_waves( cx.L, total_tiles, n, i );
      if( coop_waves_forced ) cx.L.coop_waves = coop_waves_forced;
      if( reorder )
      {
        cx.L.heavy_in = i == 0 ? nullptr : ( ( i & 1 ) ? g_ws.order_a.as<int>() : g_ws.order_b.as<int>() );
        cx.L.heavy_out = ( i & 1 ) ? g_ws.order_b.as<int>() : g_ws.order_a.as<int>();
      }
      prof.mark( "nn_icp" ); launch_icp_corr( cx.L, g_stream );
      prof.mark( "icp_moments" );
      if( lane ) launch_icp_lane_chains( cx.L, CB, g_stream ); else launch_icp_faithful( cx.L, g_stream );
      double nd = max_dist * 0.95;                                      // icp.h:493
      max_dist = (float)( nd > 0.05 ? nd : 0.05 );
    }
    prof.mark( nullptr );
    HIP_TRY( hipMemcpyAsync( hS, g_ws.state.p, state_bytes, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
    HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
    if( g_icp_trace && n == 1 && i >= 1 && i - 1 < g_icp_trace_cap ) g_icp_trace[i - 1] = hS[np * 34];      // the error after iteration i - 1
    int n_active = 0;
    for( int p = 0; p < n; ++p ) n_active += hActive[p] ? 1 : 0;
    if( n_active == 0 ) break;
  }
  if( lane ) icp_lane_account( CB, n );
  const int* hIters = (const int*)( hS + np * 33 );
  for( int p = 0; p < n; ++p ) { std::memcpy( T1s + 16 * p, hS + 16 * p, 64 ); errs[p] = hS[np * 34 + p]; if( iters ) iters[p] = hIters[p]; }
  if( lane && cx.L.stop_guard > 0.0f )
  {
    // the problems whose stop test was decided inside the guard: again, each alone, in the reference's own order
    std::vector<int> edge;
    { const int* hEdge = (const int*)( hS + np * 37 ); for( int p = 0; p < n; ++p ) if( hEdge[p] ) edge.push_back( p ); }
    for( int q : edge )
    {
      std::memcpy( T1s + 16 * (size_t)q, T_in.data() + 16 * (size_t)q, 64 );
      if( ( rc = icp_align_batch_impl( sources[q], target, T1s + 16 * (size_t)q, 1, T2, max_dist_in, max_angle, max_iter, fixed_iters, errs + q, iters ? iters + q : nullptr, 1, true ) ) ) return rc;
      g_stop_guard_redone.fetch_add( 1 );
    }
  }
  return RS_HIP_OK;
}

// (diagnostics) problems run again in the reference's order because a stop test of theirs was decided inside the guard, since rs_hip_init
int64_t rs_hip_icp_stop_guard_redone( void ) { return g_stop_guard_redone.load(); }
int32_t rs_hip_icp_early_plain( int32_t on )
{
  const int prev = g_early_plain.load();
  if( on >= 0 ) g_early_plain.store( on ? 1 : 0 );
  return prev;
}
float rs_hip_icp_stop_guard( float guard )
{
  const float prev = g_stop_guard.load();
  if( guard >= 0.0f ) g_stop_guard.store( guard );
  return prev;
}

int32_t rs_hip_icp_lane_chains_below( int32_t n_points )
{
  const int prev = g_lane_below.load();
  if( n_points >= 0 ) g_lane_below.store( n_points );
  return prev;
}

// (diagnostics) of all addends the lane chains have summed since rs_hip_init, how many were added one by one in fp32 (per mille)
int64_t rs_hip_icp_lane_chains_sequential( void ) { return g_lane_seq_addends.load(); }

int32_t rs_hip_icp_chains_gave_up( void ) { return g_chains_gave_up.load(); }

int32_t rs_hip_icp_chains_retry_after( int32_t calls )
{
  const int prev = g_chains_retry_after.load();
  if( calls >= 0 ) g_chains_retry_after.store( calls );
  return prev;
}

int32_t rs_hip_icp_reference_order_below( int32_t n_points )
{
  const int prev = g_ref_order_below.load();
  if( n_points >= 0 ) g_ref_order_below.store( n_points );
  return prev;
}

int32_t rs_hip_icp_replay_below( int32_t n_points )
{
  const int prev = g_replay_below.load();
  if( n_points >= 0 ) g_replay_below.store( n_points );
  return prev;
}

int32_t rs_hip_icp_exact_centroids( int32_t on )
{
  const int prev = g_exact_centroids.load();
  if( on >= 0 ) g_exact_centroids.store( on > 2 ? 1 : on );
  return prev;
}

int32_t rs_hip_icp_faith_guess( int32_t permille )
{
  const int prev = g_faith_guess_permille.load();
  if( permille >= 0 ) g_faith_guess_permille.store( permille );
  return prev;
}

int32_t rs_hip_icp_faith_redone( void )
{
  int v = 0;
#ifdef RS_FAITH_TIMING
  { int t[16]; (void)hipStreamSynchronize( g_stream ); if( g_faith_redone.p && hipMemcpy( t, g_faith_redone.p, sizeof t, hipMemcpyDeviceToHost ) == hipSuccess )
    for( int w = 0; w < 4; ++w ) fprintf( stderr, "[rs_hip faith timing] pass %d wave %d: %d cycles at work, %d at the barrier, %d chunks\n", RS_FAITH_TIMING, w, t[2 + 3 * w], t[3 + 3 * w], t[4 + 3 * w] ); }
#endif
  if( g_faith_redone.p && ( hipStreamSynchronize( g_stream ) != hipSuccess || hipMemcpy( &v, g_faith_redone.p, 4, hipMemcpyDeviceToHost ) != hipSuccess ) ) return -1;
  return v;
}

int32_t rs_hip_icp_replay_redone( void )
{
  int v = 0;
  if( g_ws.rp_redone.p && hipMemcpy( &v, g_ws.rp_redone.p, 4, hipMemcpyDeviceToHost ) != hipSuccess ) return -1;
  return v;
}

int rs_hip_icp_align_traced( const rs_hip_cloud_t* source, const rs_hip_cloud_t* target,
                             float* T1, const float* T2, float max_dist, float max_angle,
                             int32_t max_iter, int32_t fixed_iters, float* err, int32_t* n_iters, float* errs_per_iter )
{
  g_icp_trace = errs_per_iter; g_icp_trace_cap = errs_per_iter ? std::max( 0, max_iter ) : 0;
  const int rc = rs_hip_icp_align( source, target, T1, T2, max_dist, max_angle, max_iter, fixed_iters, err, n_iters );
  g_icp_trace = nullptr; g_icp_trace_cap = 0;
  return rc;
}

int rs_hip_icp_align( const rs_hip_cloud_t* source, const rs_hip_cloud_t* target,
                      float* T1, const float* T2, float max_dist, float max_angle,
                      int32_t max_iter, int32_t fixed_iters, float* err, int32_t* n_iters )
{
  float e = 1e6f; int32_t it = 0;
  int rc = rs_hip_icp_align_batch( source, target, T1, 1, T2, max_dist, max_angle, max_iter, fixed_iters, &e, &it );
  if( err ) *err = e;
  if( n_iters ) *n_iters = it;
  return rc;
}

int rs_hip_icp_find_corrs( const rs_hip_cloud_t* source, const rs_hip_cloud_t* target,
                           const float* T1, const float* T2, float max_dist, float max_angle,
                           float* corr_pts1, float* corr_nor1, float* corr_pts2, float* corr_nor2,
                           float* weights, int32_t* n_corrs )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !T1 || !T2 || !n_corrs ) { set_err( "icp_find_corrs: null argument" ); return RS_HIP_E_ARG; }
  IcpCtx cx;
  if( ( rc = icp_prepare( cx, source, target, 1, T2 ) ) ) return rc;
  *n_corrs = 0;
  const int nq = source->n;
  if( nq == 0 ) return RS_HIP_OK;
  std::vector<Mat4> T( 1 ); std::memcpy( T[0].m, T1, 64 );
  if( ( rc = icp_upload_state( cx, T1, 1 ) ) ) return rc;
  icp_set_radius( cx, max_dist, icp_gate_threshold( max_angle ) );
  if( ( rc = icp_fills_flush() ) ) return rc;
  { ProfScope ps( "nn_icp" ); launch_icp_corr( cx.L, g_stream ); }
  std::vector<int> slot( nq ); std::vector<float> d2( nq ), dot( nq );
  HIP_TRY( hipMemcpyAsync( slot.data(), cx.L.m_slot, (size_t)nq * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( d2.data(), cx.L.m_d2, (size_t)nq * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( dot.data(), cx.L.m_dot, (size_t)nq * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );

  // Back to source order and compact (icp.h:381-391).  The query positions/normals are
  // re-derived on the host with the same two mat·vec products the kernel used.
  std::vector<int> by_orig( nq );
  for( int s = 0; s < nq; ++s ) by_orig[source->qorder[s]] = s;
  Mat4 t2; std::memcpy( t2.m, T2, 64 ); Mat4 t2i = mat4_inverse( t2 );
  auto apply = []( const Mat4& M, const float* v, float w, float* o ) {
    o[0] = M.m[0] * v[0] + M.m[4] * v[1] + M.m[ 8] * v[2] + w * M.m[12];
    o[1] = M.m[1] * v[0] + M.m[5] * v[1] + M.m[ 9] * v[2] + w * M.m[13];
    o[2] = M.m[2] * v[0] + M.m[6] * v[1] + M.m[10] * v[2] + w * M.m[14]; };
  std::vector<float> dists; dists.reserve( nq );
  int ic = 0;
  for( int i = 0; i < nq; ++i )
  {
    const int s = by_orig[i];
    if( slot[s] < 0 ) continue;
    const int i2 = target->order[slot[s]];
    float t[3], qp[3], qn[3];
    apply( T[0], &source->h_pos[3*i], 1.0f, t ); apply( t2i, t, 1.0f, qp );
    apply( T[0], &source->h_nor[3*i], 0.0f, t ); apply( t2i, t, 0.0f, qn );
    if( corr_pts1 ) std::memcpy( corr_pts1 + 3*ic, qp, 12 );
    if( corr_nor1 ) std::memcpy( corr_nor1 + 3*ic, qn, 12 );
    if( corr_pts2 ) std::memcpy( corr_pts2 + 3*ic, &target->h_pos[3*i2], 12 );
    if( corr_nor2 ) std::memcpy( corr_nor2 + 3*ic, &target->h_nor[3*i2], 12 );
    if( weights ) weights[ic] = ( 1.0f - d2[s] / max_dist ) * dot[s];       // icp.h:387
    dists.push_back( d2[s] );
    ic++;
  }
  // icp.h:393-402 with msh_compute_mean / msh_compute_stddev in source order (msh_std.h:1800-1825)
  if( weights && ic > 0 )
  {
    float acc = 0; for( int i = 0; i < ic; ++i ) acc += dists[i];
    float mean = acc / (float)ic;
    float sq = 0.0f; for( int i = 0; i < ic; ++i ) sq += dists[i] * dists[i];
    float sd = (float)std::sqrt( sq / (float)ic - mean * mean );
    if( sd > 0.000001 ) for( int i = 0; i < ic; ++i ) if( dists[i] > 2.5f * sd ) weights[i] = 0.0;
  }
  *n_corrs = ic;
  return RS_HIP_OK;
}

int rs_hip_icp_estimate_pt2pl( const float* pts1, const float* pts2, const float* nor2,
                               const float* weights, int32_t n, float* T1, float* err )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !pts1 || !pts2 || !nor2 || !weights || !T1 || n <= 0 ) { set_err( "icp_estimate: bad arguments" ); return RS_HIP_E_ARG; }
  g_fills.n = 0;      // (see icp_prepare)
  // Present the correspondences to k_icp_moments as "query i matched slot i" with explicit weights.
  const size_t nn = (size_t)n;
  if( ( rc = g_ws.tmp_pos.ensure( nn * 16 ) ) || ( rc = g_ws.tmp_pos2.ensure( nn * 16 ) ) || ( rc = g_ws.tmp_nor2.ensure( nn * 16 ) ) ||
      ( rc = g_ws.wexp.ensure( nn * 4 ) ) || ( rc = g_ws.slot.ensure( nn * 4 ) ) || ( rc = g_ws.d2.ensure( nn * 4 ) ) || ( rc = g_ws.dot.ensure( nn * 4 ) ) ||
      ( rc = g_ws.state.ensure( ICP_STATE_WORDS * 4 ) ) ||
      ( rc = g_ws.mom_part.ensure( 256 * ICP_NMOM * 8 ) ) || ( rc = g_ws.res.ensure( ICP_NRES * 8 ) ) || ( rc = g_ws.h_a.ensure( ICP_NRES * 8 ) ) )
    return rc;
  std::vector<float4> a( nn ), b( nn ), c( nn ); std::vector<int> sl( nn );
  for( size_t i = 0; i < nn; ++i )
  {
    a[i] = make_float4( pts1[3*i], pts1[3*i+1], pts1[3*i+2], 0 );
    b[i] = make_float4( pts2[3*i], pts2[3*i+1], pts2[3*i+2], 0 );
    c[i] = make_float4( nor2[3*i], nor2[3*i+1], nor2[3*i+2], 0 );
    sl[i] = (int)i;
  }
  Mat4 I = mat4_identity(); double zeros[ICP_NRES]; std::memset( zeros, 0, sizeof(zeros) );
  float st_host[ICP_STATE_WORDS]; std::memset( st_host, 0, sizeof( st_host ) );       // identity pose, active, tickets zero
  std::memcpy( st_host, I.m, 64 ); { int one = 1; std::memcpy( st_host + 16, &one, 4 ); }
  HIP_TRY( hipMemcpyAsync( g_ws.tmp_pos.p, a.data(), nn * 16, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( g_ws.tmp_pos2.p, b.data(), nn * 16, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( g_ws.tmp_nor2.p, c.data(), nn * 16, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( g_ws.wexp.p, weights, nn * 4, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( g_ws.slot.p, sl.data(), nn * 4, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( g_ws.state.p, st_host, sizeof( st_host ), hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( g_ws.res.p, zeros, ICP_NRES * 8, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );     // host staging vectors go out of scope below
  IcpLaunch L{};
  L.tgt.pos = g_ws.tmp_pos2.as<float4>(); L.tgt.nor = g_ws.tmp_nor2.as<float4>(); L.tgt.n = n;
  L.src.pos = g_ws.tmp_pos.as<float4>(); L.src.nor = nullptr; L.src.tiles = nullptr; L.src.n = n; L.src.n_tiles = 0; L.n_prob = 1;
  L.multi = nullptr; L.max_n = n; L.max_tiles = 0;
  L.T1 = g_ws.state.as<float>(); L.active = (int*)( g_ws.state.as<float>() + 16 ); std::memcpy( L.T2i.m, I.m, 64 );
  L.ticket = (int*)( g_ws.state.as<float>() + 37 ); L.solve = 0;       // reductions only: the solve below runs on the host
  L.radius = 1.0f; L.m_slot = g_ws.slot.as<int>(); L.m_d2 = g_ws.d2.as<float>(); L.m_dot = g_ws.dot.as<float>();
  L.mom_part = g_ws.mom_part.as<double>(); L.res = g_ws.res.as<double>();
  L.n_mom_blocks = std::max( 1, std::min( 256, ( n + 255 ) / 256 ) ); L.w_explicit = g_ws.wexp.as<float>();
  // (a single estimator step on the caller's own correspondences: wherever the ICP loop would centre on the reference's chains — the lane
  //  chains' range — this entry point simply runs the reference's order: one call, its cost does not matter, its bits do)
  const int seq_below = std::max( g_ref_order_below.load(), g_exact_centroids.load() != 0 ? g_lane_below.load() : 0 );
  if( n <= std::max( seq_below, g_replay_below.load() ) )
  {
    // the reference's own accumulation order (k_icp_faithful, or its parallel form): solve and pose update on the device too
    if( ( rc = g_ws.faith.ensure( nn * FAITH_REC * 4 ) ) ) return rc;
    L.faith = g_ws.faith.as<float>(); L.by_orig = nullptr; L.err = g_ws.state.as<float>() + 34;
    // (the points are presented untransformed, so the state's pose stays the identity uploaded above and T1 is multiplied in afterwards)
    if( n <= seq_below || n > g_replay_below.load() ) { ProfScope ps( "icp_moments" ); launch_icp_faithful( L, g_stream ); }
    else
    {
      ReplayBufs RB{};
      if( ( rc = replay_prepare( RB, 1, n ) ) || ( rc = icp_fills_flush() ) ) return rc;
      ProfScope ps( "icp_moments" ); launch_icp_replay( L, RB, g_stream );
    }
    float out[ICP_STATE_WORDS];
    HIP_TRY( hipMemcpyAsync( out, g_ws.state.p, sizeof( out ), hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
    HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
    Mat4 step, T; std::memcpy( step.m, out, 64 ); std::memcpy( T.m, T1, 64 );
    T = mat4_mul( step, T );                      // icp.h:295: *T1 = T * *T1 (step = T * I, exact)
    std::memcpy( T1, T.m, 64 );
    if( err ) *err = out[34];
    return RS_HIP_OK;
  }
  { ProfScope ps( "icp_moments" ); launch_icp_moments( L, g_stream ); }
  double* hM = g_ws.h_a.as<double>();
  HIP_TRY( hipMemcpyAsync( hM, g_ws.res.p, ICP_NMOM * 8, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  Mat4 T; std::memcpy( T.m, T1, 64 );
  float e = 0.0f;
  // the reference has no Σw guard inside the estimator itself; keep T1 when the system is empty
  if( icp_solve( hM, T, e ) ) std::memcpy( T1, T.m, 64 );
  if( err ) *err = e;
  return RS_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// alignment score
// ------------------------------------------------------------------------------------------

std::atomic<long long> g_score_scene_from{ getenv( "RS_HIP_SCORE_SCENE_MIN" ) ? atoll( getenv( "RS_HIP_SCORE_SCENE_MIN" ) ) : 65536 };
int64_t rs_hip_score_scene_space_from( int64_t n_queries )
{
  const long long prev = g_score_scene_from.load();
  if( n_queries >= 0 ) g_score_scene_from.store( n_queries );
  return prev;
}

int rs_hip_alignment_scores( const rs_hip_cloud_t* object, const rs_hip_cloud_t* scene,
                             const float* poses, int32_t n_poses, float radius, int32_t max_n_neigh,
                             float* scores )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !object || !scene || !object->has_nor || !scene->has_nor || !poses || !scores || n_poses < 0 || max_n_neigh <= 0 )
  { set_err( "alignment_scores: bad arguments" ); return RS_HIP_E_ARG; }
  if( n_poses == 0 ) return RS_HIP_OK;
  if( object->n == 0 ) { for( int p = 0; p < n_poses; ++p ) scores[p] = NAN; return RS_HIP_OK; }   // 0/0, pose_proposal.cpp:156
  const int n_tiles = object->qview.n_tiles;
  if( ( rc = g_ws.poses.ensure( (size_t)n_poses * 64 ) ) || ( rc = g_ws.score_part.ensure( (size_t)n_poses * n_tiles * 8 ) ) ||
      ( rc = g_ws.scores.ensure( (size_t)n_poses * 4 ) ) ||
      ( rc = g_ws.queue.ensure( (size_t)std::min( n_poses, 65535 ) * n_tiles * 4 ) ) || ( rc = g_ws.queue_count.ensure( 4 ) ) )
    return rc;
  HIP_TRY( hipMemcpyAsync( g_ws.poses.p, poses, (size_t)n_poses * 64, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  ScoreLaunch L{};
  L.scene = scene->view; L.scene.evals = g_prof ? g_evals : nullptr; L.obj = object->qview;
  L.poses = g_ws.poses.as<float>(); L.radius_sq = radius_sq_of( radius ); L.gate_tmin = score_gate_threshold();
  L.K = max_n_neigh; L.sigma = (double)radius; L.part = g_ws.score_part.as<double>(); L.scores = g_ws.scores.as<float>();
  L.queue = g_ws.queue.as<int>(); L.queue_count = g_ws.queue_count.as<int>();
  L.solo_stages = handoff_threshold( (long long)n_tiles * n_poses );
  L.hist = nullptr;
  thread_local DevBuf histbuf;
  if( RS_DBG && getenv( "RS_HIP_SCORE_HIST" ) && !histbuf.ensure( 7 * 65 * 8 ) ) { (void)hipMemsetAsync( histbuf.p, 0, 7 * 65 * 8, g_stream ); L.hist = histbuf.as<unsigned long long>(); }
  // Scene-space route (rs_score.hip: k_score_scene) for batches on a cell grid that are worth a sort: every query keyed by the
  // scene-aligned block it lands in.  RS_HIP_SCORE_SCENE=0 turns it off, RS_HIP_SCORE_SCENE_MIN sets the batch size it starts at.
  static const int scene_route = getenv( "RS_HIP_SCORE_SCENE" ) ? atoi( getenv( "RS_HIP_SCORE_SCENE" ) ) : 1;
  const long long scene_min = g_score_scene_from.load();
  static const float parent_scale = getenv( "RS_HIP_SCORE_PARENT" ) ? (float)atof( getenv( "RS_HIP_SCORE_PARENT" ) ) : 1.0f;   // parent edge / radius
  int chunk_poses = 65535;      // the launch grid's y dimension is limited to 65535 poses per launch
  const bool scene_space = scene_route && L.scene.inv_cell > 0.0f &&
                           (long long)n_poses * object->n >= scene_min && object->n < ( 1 << 24 );
  if( scene_space )
  {
    const GridView& g = L.scene;
    static const int nbin = getenv( "RS_HIP_SCORE_NBIN" ) ? atoi( getenv( "RS_HIP_SCORE_NBIN" ) ) : 1;
    static const int cull = getenv( "RS_HIP_SCORE_CULL" ) ? atoi( getenv( "RS_HIP_SCORE_CULL" ) ) : 1;
    // the scene grid's box grown by the radius: a query outside has nothing to match ...
    float lo[3] = { g.minx - radius, g.miny - radius, g.minz - radius };
    float hi[3] = { g.minx + g.w * g.cell + radius, g.miny + g.h * g.cell + radius, g.minz + g.d * g.cell + radius };
    L.sq_lox = lo[0]; L.sq_hix = hi[0]; L.sq_loy = lo[1]; L.sq_hiy = hi[1]; L.sq_loz = lo[2]; L.sq_hiz = hi[2];
    // ... and the parent lattice only spans where queries CAN fall: the object's box under every pose, cut to that box (fewer key
    // bits: a radix pass less).  The object's own grid spans its bounding box; a brute-layout object falls back to the scene's box.
    const GridView& og = object->view;
    if( og.inv_cell > 0.0f )
    {
      float qlo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, qhi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
      const float omin[3] = { og.minx, og.miny, og.minz }, omax[3] = { og.minx + og.w * og.cell, og.miny + og.h * og.cell, og.minz + og.d * og.cell };
      bool finite = true;
      for( int p = 0; p < n_poses; ++p )
        for( int c = 0; c < 8; ++c )
        {
          const float x = ( c & 1 ) ? omax[0] : omin[0], y = ( c & 2 ) ? omax[1] : omin[1], z = ( c & 4 ) ? omax[2] : omin[2];
          const float* M = poses + (size_t)p * 16;
          for( int a = 0; a < 3; ++a )
          {
            const float v = M[a] * x + M[4 + a] * y + M[8 + a] * z + M[12 + a];
            if( !std::isfinite( v ) ) finite = false;
            qlo[a] = std::min( qlo[a], v ); qhi[a] = std::max( qhi[a], v );
          }
        }
      if( finite )
        for( int a = 0; a < 3; ++a )
        {
          const float pad = 1e-3f + 1e-5f * std::max( std::fabs( qlo[a] ), std::fabs( qhi[a] ) );      // (rounding of the corner transforms; the key only groups anyway)
          lo[a] = std::max( lo[a], qlo[a] - pad ); hi[a] = std::min( hi[a], qhi[a] + pad );
          if( hi[a] < lo[a] ) hi[a] = lo[a];
        }
    }
    // parent edge: the radius, as a whole number of cells (a coarser grid: its cell); doubled until the lattice has < 2^24 / 64 parents
    float parent = g.cell * std::max( 1.0f, roundf( parent_scale * radius / g.cell ) );
    const int fine_bits = 6;
    for( ;; parent *= 2.0f )
    {
      L.sq_dpx = (int)ceilf( ( hi[0] - lo[0] ) / parent ) + 1; L.sq_dpy = (int)ceilf( ( hi[1] - lo[1] ) / parent ) + 1; L.sq_dpz = (int)ceilf( ( hi[2] - lo[2] ) / parent ) + 1;
      const double total = (double)L.sq_dpx * L.sq_dpy * L.sq_dpz;
      if( total < (double)( 1 << 24 ) ) { L.sq_n_parents = (int)total; break; }
    }
    L.sq_ox = lo[0]; L.sq_oy = lo[1]; L.sq_oz = lo[2];
    int bits = 0;
    while( ( 1ll << bits ) <= ( (long long)L.sq_n_parents << fine_bits ) ) ++bits;      // the "nothing to match" key n_parents << fine_bits sorts last
    L.sq_bits = bits; L.sq_fine_bits = fine_bits; L.sq_inv_fine = 4.0f / parent; L.sq_nbin = nbin ? 1 : 0; L.sq_cull = cull;
    chunk_poses = (int)std::min<long long>( 65535, std::max<long long>( 1, ( 1ll << 30 ) / object->n ) );
    const size_t items = (size_t)std::min( chunk_poses, n_poses ) * object->n;
    // The route pays where MANY queries share a block — a proposal's verification, a refinement's neighbourhood: the bench's 256 poses put
    // ~650 queries into each of ~4 000 blocks.  The grid search of a pyramid level scatters its poses over the whole room (a dozen queries
    // per block): its waves would be no better filled with neighbours than object tiles are, and the sort — more key bits, millions of
    // items — costs more than the searches (level 2: 3.4 ms against 2.8; level 3: 2.7 against 1.85, tools/score_grid_timing.py).  Such
    // batches keep the object-space launch.  (A threshold of 0 queries, as the tests set it, takes the route whatever the density.)
    static const double density_min = getenv( "RS_HIP_SCORE_SCENE_DENSITY" ) ? atof( getenv( "RS_HIP_SCORE_SCENE_DENSITY" ) ) : 128.0;
    if( scene_min > 0 && (double)n_poses * object->n < density_min * (double)L.sq_n_parents )
    {
      L.sq_key_a = nullptr; chunk_poses = 65535;
    }
    else
    {
      // (counting instead of a radix sort while a table of 2^bits counters is small next to the batch: rs_score.hip, k_score_scatter; RS_HIP_SCORE_COUNTING=0: the radix sort)
      static const int counting_max_bits = getenv( "RS_HIP_SCORE_COUNTING" ) ? atoi( getenv( "RS_HIP_SCORE_COUNTING" ) ) : 22;
      const bool counting = bits <= counting_max_bits;
      const size_t n_bins = ( (size_t)1 << bits ) + 1;
      const size_t tmp_bytes = counting ? build_scan_temp_bytes( n_bins ) : build_sort_temp_bytes( (int)items, bits );
      if( ( rc = g_ws.sq_ka.ensure( items * 4 ) ) || ( rc = g_ws.sq_kb.ensure( items * 4 ) ) || ( rc = g_ws.sq_va.ensure( items * 4 ) ) ||
          ( rc = g_ws.sq_vb.ensure( items * 4 ) ) || ( rc = g_ws.sq_pq.ensure( items * 8 ) ) || ( rc = g_ws.sq_tmp.ensure( tmp_bytes ) ) ||
          ( counting && ( rc = g_ws.sq_hist.ensure( n_bins * 4 ) ) ) )
        return rc;
      L.sq_hist = counting ? g_ws.sq_hist.as<uint32_t>() : nullptr;
      L.sq_key_a = g_ws.sq_ka.as<uint32_t>(); L.sq_key_b = g_ws.sq_kb.as<uint32_t>(); L.sq_val_a = g_ws.sq_va.as<uint32_t>(); L.sq_val_b = g_ws.sq_vb.as<uint32_t>();
      L.sq_pq = g_ws.sq_pq.as<double>(); L.sq_tmp = g_ws.sq_tmp.p; L.sq_tmp_bytes = g_ws.sq_tmp.cap;
    }
  }
  for( int p0 = 0; p0 < n_poses; p0 += chunk_poses )
  {
    ScoreLaunch Lp = L;
    Lp.n_poses = std::min( chunk_poses, n_poses - p0 );
    Lp.poses = L.poses + (size_t)p0 * 16; Lp.part = L.part + (size_t)p0 * n_tiles; Lp.scores = L.scores + p0;
    ProfScope ps( "nn_score" );
    launch_score( Lp, g_stream );
  }
  HIP_TRY( hipMemcpyAsync( scores, g_ws.scores.p, (size_t)n_poses * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  if( L.hist && n_poses >= 256 )
  {
    std::vector<unsigned long long> h( 7 * 65 );
    (void)hipMemcpy( h.data(), L.hist, h.size() * 8, hipMemcpyDeviceToHost );
    unsigned long long all = 0; for( size_t k = 0; k < 6 * 65; ++k ) all += h[k];
    {
      unsigned long long ns = 0, nl = 0, b[5] = { 0, 0, 0, 0, 0 };
      for( int u = 0; u <= 64; ++u ) { ns += h[6 * 65 + u]; nl += h[6 * 65 + u] * u; b[u <= 4 ? 0 : u <= 8 ? 1 : u <= 16 ? 2 : u <= 32 ? 3 : 4] += h[6 * 65 + u]; }
      if( ns ) fprintf( stderr, "[rs_hip score hist] scene-space searches %llu, %.1f lanes each | by lanes 1-4: %4.1f %%  5-8: %4.1f %%  9-16: %4.1f %%  17-32: %4.1f %%  33-64: %4.1f %%\n", ns, (double)nl / ns,
                        100.0 * b[0] / ns, 100.0 * b[1] / ns, 100.0 * b[2] / ns, 100.0 * b[3] / ns, 100.0 * b[4] / ns );
    }
    fprintf( stderr, "[rs_hip score hist] candidates streamed %llu (rows 0-4: shells, by the lanes the shell is swept for; row 5: the rank pass, by the lanes that need their rank)\n", all );
    for( int sh = 0; sh < 6; ++sh )
    {
      unsigned long long t = 0, b[5] = { 0, 0, 0, 0, 0 };      // lanes the shell is swept for: 1-4, 5-8, 9-16, 17-32, 33-64
      for( int u = 0; u <= 64; ++u ) { t += h[sh * 65 + u]; b[u <= 4 ? 0 : u <= 8 ? 1 : u <= 16 ? 2 : u <= 32 ? 3 : 4] += h[sh * 65 + u]; }
      if( t ) fprintf( stderr, "[rs_hip score hist]   shell %d: %5.1f %% of all | by unsettled lanes 1-4: %4.1f %%  5-8: %4.1f %%  9-16: %4.1f %%  17-32: %4.1f %%  33-64: %4.1f %%\n", sh,
                       100.0 * (double)t / (double)all, 100.0 * b[0] / (double)t, 100.0 * b[1] / (double)t, 100.0 * b[2] / (double)t, 100.0 * b[3] / (double)t, 100.0 * b[4] / (double)t );
    }
  }
  return RS_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// label transfer
// ------------------------------------------------------------------------------------------

static int label_upload_placements( const rs_hip_placement_t* pl, int32_t n )
{
  int rc;
  if( ( rc = g_ws.plc.ensure( (size_t)n * sizeof(PlacementDev) ) ) || ( rc = g_ws.h_c.ensure( (size_t)n * sizeof(PlacementDev) ) ) ) return rc;
  PlacementDev* h = g_ws.h_c.as<PlacementDev>();
  for( int i = 0; i < n; ++i )
  {
    if( !pl[i].object || !pl[i].object->has_nor ) { set_err( "labels: placement %d has no object cloud with normals", i ); return RS_HIP_E_ARG; }
    Mat4 pose; std::memcpy( pose.m, pl[i].pose, 64 );
    Mat4 inv = mat4_inverse( pose ), nm = mat4_transpose( pose );       // rs_pointcloud_filters.cpp:750-751
    h[i].g = pl[i].object->view; h[i].g.evals = g_prof ? g_evals : nullptr;
    std::memcpy( h[i].inv.m, inv.m, 64 ); std::memcpy( h[i].nmat.m, nm.m, 64 );
    h[i].radius = pl[i].radius; h[i].radius_sq = radius_sq_of( pl[i].radius );
  }
  HIP_TRY( hipMemcpyAsync( g_ws.plc.p, h, (size_t)n * sizeof(PlacementDev), hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  return RS_HIP_OK;
}

// The label state of a scene lives on the device in the scene's QUERY order (g_ws.labels / g_ws.mind) while a call works on
// it; host arrays are in input order.  The two orders are exchanged by gathering (rs_rows.hip: k_label_to_*_order).
static int label_state_upload( const rs_hip_cloud_t* scene, const int8_t* labels, const float* min_dists )
{
  const size_t ns = (size_t)scene->n;
  int rc;
  if( ( rc = g_ws.labels.ensure( ns ) ) || ( rc = g_ws.mind.ensure( ns * 4 ) ) || ( rc = g_ws.labels_o.ensure( ns ) ) || ( rc = g_ws.mind_o.ensure( ns * 4 ) ) ) return rc;
  HIP_TRY( hipMemcpyAsync( g_ws.labels_o.p, labels, ns, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( g_ws.mind_o.p, min_dists, ns * 4, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  launch_label_to_query_order( scene->d_qpos, (long long)ns, g_ws.mind_o.as<float>(), g_ws.mind.as<float>(), g_ws.labels_o.as<int8_t>(), g_ws.labels.as<int8_t>(), g_stream );
  return RS_HIP_OK;
}
static int label_state_download( const rs_hip_cloud_t* scene, int8_t* labels, float* min_dists )
{
  const size_t ns = (size_t)scene->n;
  int rc;
  if( ( rc = g_ws.labels_o.ensure( ns ) ) || ( rc = g_ws.mind_o.ensure( ns * 4 ) ) ) return rc;
  launch_label_to_input_order( scene->d_qby_orig, (long long)ns, g_ws.mind.as<float>(), g_ws.mind_o.as<float>(), min_dists ? 1 : 0, g_ws.labels.as<int8_t>(), g_ws.labels_o.as<int8_t>(), g_stream );
  HIP_TRY( hipMemcpyAsync( labels, g_ws.labels_o.p, ns, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  if( min_dists ) HIP_TRY( hipMemcpyAsync( min_dists, g_ws.mind_o.p, ns * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  return RS_HIP_OK;
}
// one launch of the placement loop over the uploaded placements [first, first + n) (label_upload_placements) on the device-resident state
static int label_chain( const rs_hip_cloud_t* scene, int32_t first, int32_t n, int32_t label_base, bool fresh )
{
  int rc;
  const size_t ns = (size_t)scene->n;
  if( ( rc = g_ws.labels.ensure( ns ) ) || ( rc = g_ws.mind.ensure( ns * 4 ) ) ) return rc;
  LabelLaunch L{};
  L.scene = scene->qview; L.pl = g_ws.plc.as<PlacementDev>() + first; L.n_pl = n;
  L.label_base = label_base; L.gate_tmin = label_gate_threshold();
  L.labels = g_ws.labels.as<int8_t>(); L.min_d = g_ws.mind.as<float>(); L.rows = nullptr; L.fresh = fresh ? 1 : 0;
  { ProfScope ps( "nn_label" ); launch_label( L, g_stream ); }
  return RS_HIP_OK;
}

int rs_hip_assign_labels( const rs_hip_cloud_t* scene, const rs_hip_placement_t* placements,
                          int32_t n, int32_t label_base, int8_t* labels, float* min_dists )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !scene || !scene->has_nor || !labels || !min_dists || n < 0 || ( n > 0 && !placements ) ) { set_err( "assign_labels: bad arguments" ); return RS_HIP_E_ARG; }
  if( label_base + n > 127 ) { set_err( "assign_labels: more than 127 placements do not fit the reference's int8 labels" ); return RS_HIP_E_CAPACITY; }
  if( n == 0 || scene->n == 0 ) return RS_HIP_OK;
  if( ( rc = label_upload_placements( placements, n ) ) || ( rc = label_state_upload( scene, labels, min_dists ) ) || ( rc = label_chain( scene, 0, n, label_base, false ) ) ) return rc;
  return label_state_download( scene, labels, min_dists );
}

int rs_hip_label_rows( const rs_hip_cloud_t* scene, const rs_hip_placement_t* placements,
                       int32_t n, float* rows, int rows_device )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !scene || !scene->has_nor || !rows || n < 0 || ( n > 0 && !placements ) || rows_device < 0 || rows_device > 2 ) { set_err( "label_rows: bad arguments" ); return RS_HIP_E_ARG; }
  if( n == 0 || scene->n == 0 ) return RS_HIP_OK;
  const size_t ns = (size_t)scene->n;
  if( ( rc = label_upload_placements( placements, n ) ) ) return rc;
  // the kernel writes a row in the scene's query order (coalesced); rows_device = 2 hands that to the caller as it is
  float* d_rows = rows;
  if( rows_device != 2 ) { if( ( rc = g_ws.rows.ensure( (size_t)n * ns * 4 ) ) ) return rc; d_rows = g_ws.rows.as<float>(); }
  LabelLaunch L{};
  L.scene = scene->qview; L.pl = g_ws.plc.as<PlacementDev>(); L.n_pl = n;
  L.label_base = 0; L.gate_tmin = label_gate_threshold(); L.labels = nullptr; L.min_d = nullptr; L.rows = d_rows;
  { ProfScope ps( "nn_label" ); launch_label( L, g_stream ); }
  if( rows_device == 2 ) return RS_HIP_OK;
  float* d_out = rows;
  if( rows_device == 0 ) { if( ( rc = g_ws.rows_o.ensure( (size_t)n * ns * 4 ) ) ) return rc; d_out = g_ws.rows_o.as<float>(); }
  launch_label_to_input_order( scene->d_qby_orig, (long long)ns, d_rows, d_out, n, nullptr, nullptr, g_stream );
  if( rows_device == 0 )
  {
    HIP_TRY( hipMemcpyAsync( rows, d_out, (size_t)n * ns * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
    HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  }
  return RS_HIP_OK;
}

int rs_hip_label_partial_device( const rs_hip_cloud_t* scene, const rs_hip_placement_t* placements, int32_t n, int32_t label_base,
                                 float* min_dists_device, int8_t* labels_device )
{
  int rc = ensure_ready(); if( rc ) return rc;
  // normals are required even for an empty run: the kernel loads a lane's scene normal before it looks at the placement count
  if( !scene || !scene->has_nor || !min_dists_device || !labels_device || n < 0 || ( n > 0 && !placements ) ) { set_err( "label_partial_device: bad arguments" ); return RS_HIP_E_ARG; }
  if( label_base + n > 127 ) { set_err( "label_partial_device: more than 127 placements do not fit the reference's int8 labels" ); return RS_HIP_E_CAPACITY; }
  if( scene->n == 0 ) return RS_HIP_OK;
  if( n > 0 && ( rc = label_upload_placements( placements, n ) ) ) return rc;
  // the placement loop over this run, from the loop's initial state, its running (min_dist, label) written where the caller wants
  // them — in the scene cloud's query order, the order the kernel works in
  LabelLaunch L{};
  L.scene = scene->qview; L.pl = g_ws.plc.as<PlacementDev>(); L.n_pl = n;
  L.label_base = label_base; L.gate_tmin = label_gate_threshold(); L.labels = labels_device; L.min_d = min_dists_device; L.rows = nullptr; L.fresh = 1;
  { ProfScope ps( "nn_label" ); launch_label( L, g_stream ); }
  return RS_HIP_OK;
}

int rs_hip_fold_label_partials_device( const float* base_device, const int64_t* min_offsets, const int64_t* label_offsets, int32_t n_parts, int64_t scene_n,
                                       int8_t* labels, float* min_dists, const rs_hip_cloud_t* in_query_order_of )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( n_parts < 0 || scene_n < 0 || !labels || !min_dists || ( n_parts > 0 && ( !base_device || !min_offsets || !label_offsets ) ) ) { set_err( "fold_label_partials_device: bad arguments" ); return RS_HIP_E_ARG; }
  const rs_hip_cloud_t* qc = in_query_order_of;
  if( qc && qc->n != scene_n ) { set_err( "fold_label_partials_device: the cloud has %d points, the partials %lld", qc->n, (long long)scene_n ); return RS_HIP_E_ARG; }
  if( scene_n == 0 ) return RS_HIP_OK;
  const size_t ns = (size_t)scene_n;
  if( ( rc = g_ws.labels.ensure( ns ) ) || ( rc = g_ws.mind.ensure( ns * 4 ) ) || ( rc = g_ws.fold_off.ensure( (size_t)std::max( 1, n_parts ) * 16 ) ) ) return rc;
  std::vector<int64_t> off( (size_t)std::max( 1, n_parts ) * 2 );
  for( int r = 0; r < n_parts; ++r ) { off[r] = min_offsets[r]; off[(size_t)n_parts + r] = label_offsets[r]; }
  HIP_TRY( hipMemcpyAsync( g_ws.fold_off.p, off.data(), (size_t)std::max( 1, n_parts ) * 16, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );          // (off is a local)
  { ProfScope ps( "label_fold" );
    launch_label_fold_partials( base_device, g_ws.fold_off.as<long long>(), g_ws.fold_off.as<long long>() + n_parts, n_parts, (long long)scene_n,
                                g_ws.labels.as<int8_t>(), g_ws.mind.as<float>(), g_stream ); }
  if( qc ) return label_state_download( qc, labels, min_dists );
  HIP_TRY( hipMemcpyAsync( labels, g_ws.labels.p, ns, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( min_dists, g_ws.mind.p, ns * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  return RS_HIP_OK;
}

void rs_hip_combine_label_rows( const float* rows, int32_t n_rows, int64_t scene_n, int32_t label_base,
                                int8_t* labels, float* min_dists )
{
  for( int32_t k = 0; k < n_rows; ++k )
  {
    const float* r = rows + (size_t)k * scene_n;
    for( int64_t j = 0; j < scene_n; ++j )
      if( r[j] < min_dists[j] ) { min_dists[j] = r[j]; labels[j] = (int8_t)( label_base + k + 1 ); }   // rs_pointcloud_filters.cpp:763,772-773
  }
}

int rs_hip_fold_label_rows_device( const float* rows_device, const int64_t* row_offsets, int32_t n_rows, int64_t scene_n,
                                   int32_t label_base, int8_t* labels, float* min_dists, int32_t fresh,
                                   const rs_hip_cloud_t* rows_in_query_order_of )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( n_rows < 0 || scene_n < 0 || !labels || !min_dists || ( n_rows > 0 && ( !rows_device || !row_offsets ) ) ) { set_err( "fold_label_rows_device: bad arguments" ); return RS_HIP_E_ARG; }
  if( label_base + n_rows > 127 ) { set_err( "fold_label_rows_device: more than 127 placements do not fit the reference's int8 labels" ); return RS_HIP_E_CAPACITY; }
  const rs_hip_cloud_t* qc = rows_in_query_order_of;
  if( qc && qc->n != scene_n ) { set_err( "fold_label_rows_device: the cloud has %d points, the rows %lld", qc->n, (long long)scene_n ); return RS_HIP_E_ARG; }
  if( n_rows == 0 || scene_n == 0 ) return RS_HIP_OK;
  const size_t ns = (size_t)scene_n;
  if( ( rc = g_ws.labels.ensure( ns ) ) || ( rc = g_ws.mind.ensure( ns * 4 ) ) || ( rc = g_ws.fold_off.ensure( (size_t)n_rows * 8 ) ) ) return rc;
  if( !fresh )
  {
    if( qc ) { if( ( rc = label_state_upload( qc, labels, min_dists ) ) ) return rc; }
    else
    {
      HIP_TRY( hipMemcpyAsync( g_ws.labels.p, labels, ns, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
      HIP_TRY( hipMemcpyAsync( g_ws.mind.p, min_dists, ns * 4, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
    }
  }
  HIP_TRY( hipMemcpyAsync( g_ws.fold_off.p, row_offsets, (size_t)n_rows * 8, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  { ProfScope ps( "label_fold" );
    launch_label_fold( rows_device, g_ws.fold_off.as<long long>(), n_rows, (long long)scene_n, label_base, g_ws.labels.as<int8_t>(), g_ws.mind.as<float>(), fresh != 0, g_stream ); }
  if( qc ) return label_state_download( qc, labels, min_dists );
  HIP_TRY( hipMemcpyAsync( labels, g_ws.labels.p, ns, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( min_dists, g_ws.mind.p, ns * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  return RS_HIP_OK;
}

// rspf_arrangement_to_labels :780-848 on the device-resident state: order, the two passes.  Leaves the state in g_ws.labels /
// g_ws.mind (query order) and the sorted order in `ord`; *launched = 0 when there was nothing to launch (n = 0 or an empty scene).
static int arrangement_passes( const rs_hip_cloud_t* scene, const float* poses, const rs_hip_cloud_t* const* objects,
                               const int32_t* is_static, const int32_t* class_idx, int32_t n, float radius, int prioritize_static,
                               std::vector<int32_t>& ord, int* launched )
{
  int rc;
  *launched = 0;
  if( !scene || n < 0 || ( n > 0 && ( !poses || !objects || !is_static || !class_idx ) ) ) { set_err( "arrangement_to_labels: bad arguments" ); return RS_HIP_E_ARG; }
  for( int i = 0; i < n; ++i ) if( !objects[i] || !objects[i]->has_nor ) { set_err( "arrangement_to_labels: placement %d has no object cloud with normals", i ); return RS_HIP_E_ARG; }
  const int64_t ns = scene->n;
  if( n > 127 ) { set_err( "arrangement_to_labels: more than 127 placements do not fit the reference's int8 labels" ); return RS_HIP_E_CAPACITY; }
  if( !scene->has_nor && ns > 0 && n > 0 ) { set_err( "arrangement_to_labels: the scene cloud needs normals" ); return RS_HIP_E_ARG; }
  // :823-827 — qsort by (is_static << 10 | class_idx); glibc's qsort is a stable merge sort
  ord.resize( (size_t)n );
  for( int i = 0; i < n; ++i ) ord[i] = i;
  std::stable_sort( ord.begin(), ord.end(), [&]( int a, int b ) {
    return ( ( is_static[a] << 10 ) | class_idx[a] ) < ( ( is_static[b] << 10 ) | class_idx[b] ); } );
  int first_static = 0;                                                                      // :830-835
  for( int i = 0; i < n; ++i ) if( is_static[ord[i]] ) { first_static = i; break; }
  std::vector<rs_hip_placement_t> pl( n );
  for( int i = 0; i < n; ++i )
  {
    std::memcpy( pl[i].pose, poses + 16 * ord[i], 64 ); pl[i].object = objects[ord[i]];
    pl[i].radius = ( i < first_static ) ? radius : ( prioritize_static ? radius : 1.5f * radius );   // :837-848
  }
  if( ns == 0 || n == 0 ) return RS_HIP_OK;
  // Both passes run on the device-resident state (query order), which starts from (label 0, min_dist 1e9) inside the first
  // launch; only the final labels / min_dists come back (one gather to input order, one download).
  if( ( rc = label_upload_placements( pl.data(), n ) ) ) return rc;          // all of them once: the two launches read their parts
  bool have_state = false;
  if( first_static > 0 ) { if( ( rc = label_chain( scene, 0, first_static, 0, true ) ) ) return rc; have_state = true; }
  if( prioritize_static && have_state )                                                      // :841-844: min_dists start over, labels stay
  {
    const float big = 1e9f; uint32_t bits; std::memcpy( &bits, &big, 4 );
    HIP_TRY( hipMemsetD32Async( (hipDeviceptr_t)g_ws.mind.p, (int)bits, (size_t)ns, g_stream ), RS_HIP_E_RUNTIME );
  }
  if( ( rc = label_chain( scene, first_static, n - first_static, first_static, !have_state ) ) ) return rc;
  *launched = 1;
  return RS_HIP_OK;
}

int rs_hip_arrangement_to_labels( const rs_hip_cloud_t* scene,
                                  const float* poses, const rs_hip_cloud_t* const* objects,
                                  const int32_t* is_static, const int32_t* class_idx, int32_t n,
                                  float radius, int prioritize_static,
                                  int8_t* labels, float* min_dists, int32_t* sorted_order )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !labels ) { set_err( "arrangement_to_labels: bad arguments" ); return RS_HIP_E_ARG; }      // (min_dists may be null: not wanted)
  std::vector<int32_t> ord; int launched = 0;
  if( ( rc = arrangement_passes( scene, poses, objects, is_static, class_idx, n, radius, prioritize_static, ord, &launched ) ) ) return rc;
  if( sorted_order ) for( int i = 0; i < n; ++i ) sorted_order[i] = ord[i];
  if( !launched ) { for( int64_t j = 0; j < scene->n; ++j ) { labels[j] = 0; if( min_dists ) min_dists[j] = 1e9; } return RS_HIP_OK; }     // :799-802, :820
  return label_state_download( scene, labels, min_dists );
}

int rs_hip_arrangement_to_ids( const rs_hip_cloud_t* scene,
                               const float* poses, const rs_hip_cloud_t* const* objects,
                               const int32_t* is_static, const int32_t* class_idx, const int32_t* uidx, int32_t n,
                               float radius, int prioritize_static, int32_t unlabelled_class_idx,
                               int32_t* class_ids, int32_t* instance_ids, int8_t* labels, float* min_dists, int32_t* sorted_order )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !class_ids || !instance_ids || ( n > 0 && !uidx ) ) { set_err( "arrangement_to_ids: bad arguments" ); return RS_HIP_E_ARG; }
  std::vector<int32_t> ord; int launched = 0;
  if( ( rc = arrangement_passes( scene, poses, objects, is_static, class_idx, n, radius, prioritize_static, ord, &launched ) ) ) return rc;
  if( sorted_order ) for( int i = 0; i < n; ++i ) sorted_order[i] = ord[i];
  const size_t ns = (size_t)scene->n;
  if( !launched )
  {
    for( size_t j = 0; j < ns; ++j ) { class_ids[j] = unlabelled_class_idx; instance_ids[j] = 1024; if( labels ) labels[j] = 0; if( min_dists ) min_dists[j] = 1e9; }
    return RS_HIP_OK;
  }
  // :851-869 on the device, fused with the move from query order to input order
  std::vector<int32_t> tab( (size_t)2 * n );
  for( int i = 0; i < n; ++i ) { tab[i] = class_idx[ord[i]]; tab[n + i] = uidx[ord[i]]; }
  if( ( rc = g_ws.ids_tab.ensure( tab.size() * 4 ) ) || ( rc = g_ws.ids_out.ensure( ns * 8 ) ) || ( rc = g_ws.labels_o.ensure( ns ) ) || ( rc = g_ws.mind_o.ensure( ns * 4 ) ) ) return rc;
  HIP_TRY( hipMemcpyAsync( g_ws.ids_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  int* d_cls = g_ws.ids_out.as<int>(); int* d_inst = d_cls + ns;
  launch_label_ids_to_input_order( scene->d_qby_orig, (long long)ns, g_ws.labels.as<int8_t>(), g_ws.mind.as<float>(), g_ws.ids_tab.as<int>(), g_ws.ids_tab.as<int>() + n,
                                   unlabelled_class_idx, d_cls, d_inst, g_ws.labels_o.as<int8_t>(), g_ws.mind_o.as<float>(), g_stream );
  HIP_TRY( hipMemcpyAsync( class_ids, d_cls, ns * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( instance_ids, d_inst, ns * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  if( labels ) HIP_TRY( hipMemcpyAsync( labels, g_ws.labels_o.p, ns, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  if( min_dists ) HIP_TRY( hipMemcpyAsync( min_dists, g_ws.mind_o.p, ns * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );               // (`tab` is still alive here)
  return RS_HIP_OK;
}

// The attribute gathers that end rs_pointcloud__compute_level_poisson (lib/rs/rs_pointcloud.h:1090-1099): every per-point
// array of level 0 at the sample indices.  Arrays are host pointers (any of them may be NULL), `words` 32-bit words per point.
int rs_hip_gather_attributes( const int32_t* sample_idx, int32_t count, int32_t n_src,
                              const void* const* src, const int32_t* words, void* const* dst, int32_t n_arrays )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( count < 0 || n_src < 0 || n_arrays < 0 || ( count > 0 && !sample_idx ) || ( n_arrays > 0 && ( !src || !words || !dst ) ) ) { set_err( "gather_attributes: bad arguments" ); return RS_HIP_E_ARG; }
  if( count == 0 || n_arrays == 0 ) return RS_HIP_OK;
  for( int32_t i = 0; i < count; ++i ) if( sample_idx[i] < 0 || sample_idx[i] >= n_src ) { set_err( "gather_attributes: sample index %d out of range", sample_idx[i] ); return RS_HIP_E_ARG; }
  if( ( rc = g_ws.lvl_samples.ensure( (size_t)count * 4 ) ) ) return rc;
  HIP_TRY( hipMemcpyAsync( g_ws.lvl_samples.p, sample_idx, (size_t)count * 4, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  for( int a = 0; a < n_arrays; ++a )
  {
    if( !src[a] || !dst[a] ) continue;
    if( words[a] <= 0 ) { set_err( "gather_attributes: array %d has no width", a ); return RS_HIP_E_ARG; }
    const size_t in_b = (size_t)n_src * words[a] * 4, out_b = (size_t)count * words[a] * 4;
    if( ( rc = g_ws.attr_in.ensure( in_b ) ) || ( rc = g_ws.attr_out.ensure( out_b ) ) ) return rc;
    HIP_TRY( hipMemcpyAsync( g_ws.attr_in.p, src[a], in_b, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
    launch_gather_words( g_ws.attr_in.as<uint32_t>(), g_ws.lvl_samples.as<int>(), count, words[a], g_ws.attr_out.as<uint32_t>(), g_stream );
    HIP_TRY( hipMemcpyAsync( dst[a], g_ws.attr_out.p, out_b, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
    HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );      // the two staging buffers are reused by the next array
  }
  return RS_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// generic rows
// ------------------------------------------------------------------------------------------

int rs_hip_radius_search( const rs_hip_cloud_t* target, const float* query, int64_t n_query,
                          float radius, int32_t k, float* distances_sq, int32_t* indices, size_t* n_neighbors,
                          uint64_t* total )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !target || !query || !distances_sq || !indices || n_query < 0 || k <= 0 || !( radius > 0.0f ) ) { set_err( "radius_search: bad arguments" ); return RS_HIP_E_ARG; }
  if( n_query > 0x7fffffff / 2 ) { set_err( "radius_search: too many queries for one call" ); return RS_HIP_E_CAPACITY; }
  if( total ) *total = 0;
  if( n_query == 0 ) return RS_HIP_OK;
  const int nq = (int)n_query;
  const GridView& g = target->view;
  Workspace& W = g_ws;
  const size_t nn = (size_t)nq, nk = nn * k;
  // Long rows, or few queries: one wave per query in the caller's order (k_rows_wave) — one launch between the upload and
  // the download, which is what the reference's unchanged consumers need (mgs_compute_object_alignment_score calls this
  // tens of thousands of times with a few hundred queries and K = 64, pose_proposal.cpp:115-124).  Short rows of many
  // queries stay with the tiled successive-minima kernel below, whose cost grows with K but whose tiles share their sweeps.
  const int wave_path_max_k = 1024;
  const long long tiled_from = getenv( "RS_HIP_ROWS_TILED_FROM" ) ? atoll( getenv( "RS_HIP_ROWS_TILED_FROM" ) ) : 4096;      // (read per call: tests switch it)
  if( k <= wave_path_max_k && ( k >= 16 || nq < tiled_from ) && !getenv( "RS_HIP_NO_ROWS_WAVE" ) )
  {
    // One upload, one launch, one download, one synchronisation: the call is latency (the unchanged pose_proposal makes
    // ~35 k of them per run).  Queries and the overflow flag travel in one pinned block; counts, distances and indices
    // come back in one, and are handed to the caller's arrays on the host.
    const size_t up_words = nn * 3, down_words = ( nn + 1 ) + 2 * nk;
    if( ( rc = W.bld_pos.ensure( up_words * 4 ) ) || ( rc = W.rows.ensure( down_words * 4 ) ) ||
        ( rc = W.h_a.ensure( down_words * 4 ) ) || ( rc = W.h_b.ensure( up_words * 4 ) ) ) return rc;
    float* h_up = W.h_b.as<float>();
    std::memcpy( h_up, query, nn * 12 );
    int* h_nn = W.h_a.as<int>();
    // Small calls go without the two copies: the pinned blocks are device-visible, the kernel reads the queries from host memory
    // and stores counts and rows straight into it (a few tens of KB over PCIe), so a call is ONE launch and one synchronisation.
    static const size_t zero_copy_below = getenv( "RS_HIP_ROWS_ZERO_COPY_BELOW" ) ? (size_t)atoll( getenv( "RS_HIP_ROWS_ZERO_COPY_BELOW" ) ) : ( 256u << 10 );
    if( down_words * 4 <= zero_copy_below )
    {
      // (RS_HIP_ROWS_POLL=1, off by default: instead of synchronising the stream the host polls the counts — every wave stores its
      //  row, then, after a system-scope fence, the row's count, which the host pre-set to a sentinel.  3 us less per call, and not
      //  safe: in one of ~60 suite runs a row was read before it had arrived — writes to host memory over PCIe may pass each
      //  other, a fence on the device does not order their ARRIVAL; only the runtime's completion signal does.)
      static const bool poll = getenv( "RS_HIP_ROWS_POLL" ) != nullptr;
      const int pending = INT_MIN;
      if( poll ) for( int i = 0; i < nq; ++i ) ( (volatile int*)h_nn )[i] = pending;
      { ProfScope ps( "nn_rows" );
        launch_rows_wave( g, h_up, nq, k, radius, radius_sq_of( radius ), (float*)( h_nn + nn + 1 ), h_nn + nn + 1 + nk, h_nn, h_nn + nq, g_stream ); }
      bool done = false;
      if( poll )
      {
        const auto t0 = std::chrono::steady_clock::now();
        int i = 0;
        for( unsigned spins = 0; ; ++spins )
        {
          while( i < nq && ( (volatile int*)h_nn )[i] != pending ) ++i;
          if( i == nq ) { done = true; break; }
          __builtin_ia32_pause();
          if( ( spins & 1023 ) == 1023 && std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count() > 2e-3 ) break;
        }
        std::atomic_thread_fence( std::memory_order_acquire );
      }
      if( !done ) HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
    }
    else
    {
      float* d_up = W.bld_pos.as<float>();
      int* d_nn = W.rows.as<int>();                    // [nq] counts (-1: more than 1024 points within the radius), one spare word, then the rows
      float* d_d2 = W.rows.as<float>() + ( nn + 1 ); int* d_idx = W.rows.as<int>() + ( nn + 1 ) + nk;
      HIP_TRY( hipMemcpyAsync( d_up, h_up, up_words * 4, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
      { ProfScope ps( "nn_rows" );
        launch_rows_wave( g, d_up, nq, k, radius, radius_sq_of( radius ), d_d2, d_idx, d_nn, d_nn + nq, g_stream ); }
      HIP_TRY( hipMemcpyAsync( h_nn, d_nn, down_words * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
      HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
    }
    bool overflow = false;
    for( int i = 0; i < nq; ++i ) overflow |= h_nn[i] < 0;
    if( !overflow )
    {
      // rows are written up to their counts only: copy exactly those (the caller's arrays keep whatever else they held, like the reference's)
      const float* hd = (const float*)( h_nn + nn + 1 ); const int* hi = h_nn + nn + 1 + nk;
      uint64_t tot = 0;
      for( int i = 0; i < nq; ++i )
      {
        const size_t c = (size_t)h_nn[i];
        std::memcpy( distances_sq + (size_t)i * k, hd + (size_t)i * k, c * 4 );
        std::memcpy( indices + (size_t)i * k, hi + (size_t)i * k, c * 4 );
        tot += c; if( n_neighbors ) n_neighbors[i] = c;
      }
      if( total ) *total = tot;
      return RS_HIP_OK;
    }
    // some query has more than 1024 points within the radius: the storage-free kernel below takes the whole call
  }
  // order the queries along a Hilbert curve and tile them on the device, exactly like a cloud's query layout
  const size_t tmp_bytes = std::max( build_sort_temp_bytes( nq, 32 ), build_scan_temp_bytes( nn + 1 ) );
  if( ( rc = W.bld_pos.ensure( nn * 12 ) ) || ( rc = W.bld_k0.ensure( ( nn + 1 ) * 4 ) ) || ( rc = W.bld_k1.ensure( ( nn + 1 ) * 4 ) ) || ( rc = W.bld_v0.ensure( nn * 4 ) ) ||
      ( rc = W.bld_v1.ensure( nn * 4 ) ) || ( rc = W.bld_v2.ensure( nn * 4 ) ) || ( rc = W.bld_small.ensure( 64 ) ) || ( rc = W.bld_tmp.ensure( tmp_bytes + 256 ) ) ||
      ( rc = W.q4.ensure( nn * 16 ) ) || ( rc = W.tmp_pos.ensure( ( nn + 2 ) * 4 ) ) || ( rc = W.rd2.ensure( nk * 4 ) ) || ( rc = W.ridx.ensure( nk * 4 ) ) || ( rc = W.rnn.ensure( nn * 4 ) ) ) return rc;
  float* d_raw = W.bld_pos.as<float>();
  uint32_t *k0 = W.bld_k0.as<uint32_t>(), *k1 = W.bld_k1.as<uint32_t>(), *v0 = W.bld_v0.as<uint32_t>(), *v1 = W.bld_v1.as<uint32_t>(), *v2 = W.bld_v2.as<uint32_t>();
  unsigned* d_small = W.bld_small.as<unsigned>();
  HIP_TRY( hipMemcpyAsync( d_raw, query, nn * 12, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  const unsigned init[8] = { ~0u, ~0u, ~0u, 0u, 0u, 0u, 0u, 0u };
  unsigned got[8];
  HIP_TRY( hipMemcpyAsync( d_small, init, 32, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  launch_build_bounds( d_raw, nullptr, nq, d_small, g_stream );
  HIP_TRY( hipMemcpyAsync( got, d_small, 32, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  auto dec = []( unsigned u ) { unsigned b = ( u & 0x80000000u ) ? ( u ^ 0x80000000u ) : ~u; float f; std::memcpy( &f, &b, 4 ); return f; };
  float mn[3], ext = 0.0f;
  for( int a = 0; a < 3; ++a )
  {
    mn[a] = dec( got[a] ); const float hi = dec( got[3 + a] );
    if( !( hi >= mn[a] ) || !std::isfinite( mn[a] ) || !std::isfinite( hi ) ) mn[a] = 0.0f;
    else if( hi - mn[a] > ext ) ext = hi - mn[a];
  }
  launch_build_hilbert( d_raw, nq, mn, ext > 0.0f ? 1024.0f / ext : 0.0f, k0, v0, g_stream );
  if( build_sort_pairs( W.bld_tmp.p, tmp_bytes, k0, k1, v0, v2, nq, 30, g_stream ) ) { set_err( "radius_search: device sort failed" ); return RS_HIP_E_RUNTIME; }
  launch_build_gather( d_raw, nullptr, v2, nq, W.q4.as<float4>(), nullptr, g_stream );          // {x, y, z, bitcast(original query index)}
  uint32_t* flags = k0; uint32_t* scanned = k1;
  HIP_TRY( hipMemsetAsync( flags + nq, 0, 4, g_stream ), RS_HIP_E_RUNTIME );
  launch_build_tile_flags( W.q4.as<float4>(), nq, std::max( 0.25f, 4.0f * radius ), flags, v0, v1, g_stream );
  if( build_exclusive_scan( W.bld_tmp.p, tmp_bytes, flags, scanned, nn + 1, g_stream ) ) { set_err( "radius_search: device scan failed" ); return RS_HIP_E_RUNTIME; }
  unsigned n_tiles_u = 0;
  HIP_TRY( hipMemcpyAsync( &n_tiles_u, scanned + nq, 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  const int n_tiles = (int)n_tiles_u;
  launch_build_tile_scatter( flags, scanned, nq, W.tmp_pos.as<uint32_t>(), g_stream );
  const uint32_t last = (uint32_t)nq;
  HIP_TRY( hipMemcpyAsync( W.tmp_pos.as<uint32_t>() + n_tiles, &last, 4, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemsetAsync( g_ws.rd2.p, 0, nk * 4, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemsetAsync( g_ws.ridx.p, 0, nk * 4, g_stream ), RS_HIP_E_RUNTIME );
  RowsLaunch L{};
  L.tgt = g; L.q.pos = g_ws.q4.as<float4>(); L.q.nor = nullptr; L.q.tiles = g_ws.tmp_pos.as<uint32_t>(); L.q.n = nq; L.q.n_tiles = n_tiles;
  L.K = k; L.radius = radius; L.radius_sq = radius_sq_of( radius );
  L.d2 = g_ws.rd2.as<float>(); L.idx = g_ws.ridx.as<int>(); L.nn = g_ws.rnn.as<int>();
  { ProfScope ps( "nn_rows" ); launch_rows( L, g_stream ); }
  std::vector<int> counts( nq );
  HIP_TRY( hipMemcpyAsync( distances_sq, L.d2, nk * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( indices, L.idx, nk * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( counts.data(), L.nn, (size_t)nq * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  uint64_t tot = 0;
  for( int i = 0; i < nq; ++i ) { tot += (uint64_t)counts[i]; if( n_neighbors ) n_neighbors[i] = (size_t)counts[i]; }
  if( total ) *total = tot;
  return RS_HIP_OK;
}

} // extern "C"

// ------------------------------------------------------------------------------------------
// neighbourhood graph
// ------------------------------------------------------------------------------------------

extern "C" int rs_hip_compute_neighborhood( const rs_hip_cloud_t* cloud, int32_t max_nn, float radius_sq,
                                            float dist_exp, float angle_exp,
                                            int32_t* idx1, int32_t* idx2, float* weight, int64_t capacity, int64_t* n_edges )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !cloud || !cloud->has_nor || max_nn <= 0 || !( radius_sq > 0.0f ) || !idx1 || !idx2 || !weight || !n_edges )
  { set_err( "compute_neighborhood: bad arguments" ); return RS_HIP_E_ARG; }
  *n_edges = 0;
  const int n = cloud->n, K = max_nn;
  if( n == 0 ) return RS_HIP_OK;
  if( capacity < (int64_t)n * K ) { set_err( "compute_neighborhood: capacity must be >= n*max_nn" ); return RS_HIP_E_ARG; }
  const size_t nk = (size_t)n * K;
  if( ( rc = g_ws.rd2.ensure( nk * 4 ) ) || ( rc = g_ws.ridx.ensure( nk * 4 ) ) || ( rc = g_ws.rnn.ensure( (size_t)n * 4 ) ) ||
      ( rc = g_ws.enor.ensure( (size_t)n * 12 ) ) || ( rc = g_ws.ecount.ensure( (size_t)n * 4 ) ) || ( rc = g_ws.eoffset.ensure( ( (size_t)n + 1 ) * 4 ) ) ||
      ( rc = g_ws.e1.ensure( nk * 4 ) ) || ( rc = g_ws.e2.ensure( nk * 4 ) ) || ( rc = g_ws.ew.ensure( nk * 4 ) ) )
    return rc;
  // 1. self-search rows (rs_pointcloud_filters.cpp:685-693): radius = (float)sqrt(radius_sq), K = max_nn
  const float radius = (float)std::sqrt( (double)radius_sq );
  RowsLaunch R{};
  R.tgt = cloud->view; R.q = cloud->qview; R.K = K; R.radius = radius; R.radius_sq = radius_sq_of( radius );
  R.d2 = g_ws.rd2.as<float>(); R.idx = g_ws.ridx.as<int>(); R.nn = g_ws.rnn.as<int>();
  { ProfScope ps( "nn_rows" ); launch_rows( R, g_stream ); }
  // 2. rows -> unique weighted edges
  HIP_TRY( hipMemcpyAsync( g_ws.enor.p, cloud->h_nor.data(), (size_t)n * 12, hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
  EdgeLaunch E{};
  E.n = n; E.K = K; E.row_d2 = R.d2; E.row_idx = R.idx; E.row_nn = R.nn; E.nor = g_ws.enor.as<float>();
  E.radius_sq = radius_sq; E.dist_exp = dist_exp; E.angle_exp = angle_exp;
  auto small_int = []( float e ) { return ( e >= 0.0f && e <= 64.0f && e == std::floor( e ) ) ? (int)e : -1; };
  E.dist_int = small_int( dist_exp ); E.angle_int = small_int( angle_exp );
  E.count = g_ws.ecount.as<int>(); E.offset = g_ws.eoffset.as<unsigned>();
  E.e1 = g_ws.e1.as<int>(); E.e2 = g_ws.e2.as<int>(); E.ew = g_ws.ew.as<float>();
  { ProfScope ps( "edges" ); launch_edge_count( E, g_stream ); launch_edge_scan( E, g_stream ); launch_edge_write( E, g_stream ); }
  unsigned total = 0;
  HIP_TRY( hipMemcpyAsync( &total, E.offset + n, 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( idx1, E.e1, (size_t)total * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( idx2, E.e2, (size_t)total * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( weight, E.ew, (size_t)total * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  *n_edges = (int64_t)total;
  return RS_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// level builder (rs_rows.hip: k_level_*)
// ------------------------------------------------------------------------------------------

// the samples of a level, left on the device in g_ws.lvl_samples (increasing original indices)
static int level_samples_device( const rs_hip_cloud_t* cloud, float radius, int32_t max_n_neigh, int32_t* n_samples, int32_t* n_rounds )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !cloud || !( radius > 0.0f ) || max_n_neigh <= 0 || !n_samples ) { set_err( "level_samples: bad arguments" ); return RS_HIP_E_ARG; }
  *n_samples = 0; if( n_rounds ) *n_rounds = 0;
  const int n = cloud->n;
  if( n == 0 ) return RS_HIP_OK;
  const size_t n1 = (size_t)n + 1;
  const size_t tmp_bytes = build_scan_temp_bytes( n1 );
  if( ( rc = g_ws.lvl_cnt.ensure( n1 * 4 ) ) || ( rc = g_ws.lvl_within.ensure( n1 * 4 ) ) || ( rc = g_ws.lvl_offset.ensure( n1 * 4 ) ) ||
      ( rc = g_ws.lvl_state.ensure( n1 * 4 ) ) || ( rc = g_ws.lvl_misc.ensure( 64 ) ) || ( rc = g_ws.lvl_flags.ensure( n1 * 4 ) ) ||
      ( rc = g_ws.lvl_scan.ensure( n1 * 4 ) ) || ( rc = g_ws.lvl_samples.ensure( n1 * 4 ) ) || ( rc = g_ws.lvl_tmp.ensure( tmp_bytes ) ) ||
      ( rc = g_ws.lvl_cursor.ensure( n1 * 4 ) ) || ( rc = g_ws.lvl_work_a.ensure( n1 * 4 ) ) || ( rc = g_ws.lvl_work_b.ensure( n1 * 4 ) ) )
    return rc;
  LevelLaunch L{};
  L.tgt = cloud->view; L.q = cloud->qview; L.by_orig = cloud->d_qby_orig; L.n = n;
  L.radius = radius; L.radius_sq = radius_sq_of( radius ); L.max_n_neigh = max_n_neigh;
  L.n_earlier = g_ws.lvl_cnt.as<int>(); L.n_later = g_ws.lvl_within.as<int>(); L.offset = g_ws.lvl_offset.as<unsigned>();
  L.state = g_ws.lvl_state.as<int>(); L.word = g_ws.lvl_cursor.as<int>();
  L.flags = g_ws.lvl_flags.as<unsigned>(); L.flag_scan = g_ws.lvl_scan.as<unsigned>(); L.samples = g_ws.lvl_samples.as<int>();
  const int BATCH = 16;
  int* misc = g_ws.lvl_misc.as<int>();            // [0..BATCH] frontier lengths of a batch of steps (entry r: input of step r), [31] over-cap flag
  L.over_cap = misc + 31;
  HIP_TRY( hipMemsetAsync( misc, 0, 128, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemsetAsync( L.n_later + n, 0, 4, g_stream ), RS_HIP_E_RUNTIME );
  // 1. who is within the radius of whom: counts, row offsets, rows of later neighbours
  { ProfScope ps( "level_neighbours" ); launch_level_neighbours( L, false, g_stream ); }
  if( build_exclusive_scan( g_ws.lvl_tmp.p, tmp_bytes, (const uint32_t*)L.n_later, g_ws.lvl_offset.as<uint32_t>(), n1, g_stream ) )
  { set_err( "level_samples: device scan failed" ); return RS_HIP_E_RUNTIME; }
  unsigned total = 0; int over = 0;
  HIP_TRY( hipMemcpyAsync( &total, L.offset + n, 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipMemcpyAsync( &over, L.over_cap, 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  if( over ) { set_err( "level_samples: a point has more than max_n_neigh = %d points within the radius (a reference search would be truncated)", max_n_neigh ); return RS_HIP_E_CAPACITY; }
  if( ( rc = g_ws.lvl_adj.ensure( std::max<size_t>( 1, total ) * 4 ) ) ) return rc;
  L.adj = g_ws.lvl_adj.as<int>();
  { ProfScope ps( "level_neighbours" ); launch_level_neighbours( L, true, g_stream ); }
  // 2. propagate: one launch per frontier, the host looks at the frontier lengths once per batch of steps
  int* front[2] = { g_ws.lvl_work_a.as<int>(), g_ws.lvl_work_b.as<int>() };
  L.front_out = front[0]; L.front_count_out = misc;           // the first frontier: points without earlier neighbours
  launch_level_init( L, g_stream );
  // lanes per frontier item from the average row length
  const double avg_row = (double)total / (double)n;
  const int G = avg_row >= 24.0 ? 64 : ( avg_row >= 3.0 ? 8 : 1 );
  const int per_block = 256 / G;
  int rounds = 0, blocks = std::min( ( n + per_block - 1 ) / per_block, 16384 ); bool done = false, first = true;
  while( !done )
  {
    if( rounds > n + BATCH ) { set_err( "level_samples: no fixed point after %d steps", rounds ); return RS_HIP_E_RUNTIME; }
    if( !first ) HIP_TRY( hipMemcpyAsync( misc, misc + BATCH, 4, hipMemcpyDeviceToDevice, g_stream ), RS_HIP_E_RUNTIME );
    HIP_TRY( hipMemsetAsync( misc + 1, 0, BATCH * 4, g_stream ), RS_HIP_E_RUNTIME );
    { ProfScope ps( "level_rounds" );
      for( int b = 0; b < BATCH; ++b )
      {
        const int r = rounds + b;
        L.front_in = front[r & 1]; L.front_count_in = misc + b;
        L.front_out = front[( r + 1 ) & 1]; L.front_count_out = misc + b + 1;
        launch_level_frontier( L, G, blocks, g_stream );
      } }
    int len[BATCH + 1];
    HIP_TRY( hipMemcpyAsync( len, misc, ( BATCH + 1 ) * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
    HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
    int widest = 1;
    for( int b = 0; b <= BATCH && !done; ++b ) { if( len[b] == 0 ) done = true; else { if( b < BATCH ) rounds++; widest = std::max( widest, len[b] ); } }
    blocks = std::min( ( n + per_block - 1 ) / per_block, std::max( 256, 2 * ( ( widest + per_block - 1 ) / per_block ) ) );     // the next batch's frontiers are about as wide as this one's
    first = false;
  }
  // 3. the samples in increasing index order
  launch_level_flags( L, g_stream );
  if( build_exclusive_scan( g_ws.lvl_tmp.p, tmp_bytes, L.flags, g_ws.lvl_scan.as<uint32_t>(), n1, g_stream ) )
  { set_err( "level_samples: device scan failed" ); return RS_HIP_E_RUNTIME; }
  launch_level_scatter( L, g_stream );
  unsigned count = 0;
  HIP_TRY( hipMemcpyAsync( &count, L.flag_scan + n, 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  *n_samples = (int32_t)count;
  if( n_rounds ) *n_rounds = rounds;
  return RS_HIP_OK;
}

extern "C" int rs_hip_level_samples( const rs_hip_cloud_t* cloud, float radius, int32_t max_n_neigh,
                                     int32_t* sample_idx, int32_t* n_samples, int32_t* n_rounds )
{
  if( !sample_idx || !n_samples ) { set_err( "level_samples: bad arguments" ); return RS_HIP_E_ARG; }
  int rc = level_samples_device( cloud, radius, max_n_neigh, n_samples, n_rounds );
  if( rc || *n_samples == 0 ) return rc;
  HIP_TRY( hipMemcpyAsync( sample_idx, g_ws.lvl_samples.p, (size_t)*n_samples * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  return RS_HIP_OK;
}

// rs_pointcloud__compute_level_poisson + rs_pointcloud_compute_search_grid (rs_pointcloud.h:984-1106,849-863) without
// leaving the device: samples -> gather of the level's positions / normals -> index build.
extern "C" rs_hip_cloud_t* rs_hip_cloud_create_level( const rs_hip_cloud_t* base, float radius, int32_t max_n_neigh, float cell_size,
                                                      int32_t* sample_idx, int32_t* n_samples )
{
  int32_t count = 0;
  if( level_samples_device( base, radius, max_n_neigh, &count, nullptr ) ) return nullptr;
  if( n_samples ) *n_samples = count;
  const size_t nn = (size_t)std::max( 1, count );
  if( g_ws.lvl_pos.ensure( nn * 12 ) || ( base->has_nor && g_ws.lvl_nor.ensure( nn * 12 ) ) ) return nullptr;
  if( count > 0 )
  {
    launch_level_gather( g_ws.lvl_samples.as<int>(), count, base->d_qby_orig, base->d_qpos, base->has_nor ? base->d_qnor : nullptr,
                         g_ws.lvl_pos.as<float>(), base->has_nor ? g_ws.lvl_nor.as<float>() : nullptr, g_stream );
    if( sample_idx && hipMemcpyAsync( sample_idx, g_ws.lvl_samples.p, (size_t)count * 4, hipMemcpyDeviceToHost, g_stream ) != hipSuccess )
    { set_err( "cloud_create_level: copy of the sample indices failed" ); return nullptr; }
  }
  return cloud_create_impl( g_ws.lvl_pos.as<float>(), base->has_nor ? g_ws.lvl_nor.as<float>() : nullptr, count, cell_size, true );
}

// ------------------------------------------------------------------------------------------
// scene-coverage term
// ------------------------------------------------------------------------------------------

struct rs_hip_coverage
{
  VoxGrid grid{};
  float voxel_size = 0.0f, origin[3] = { 0, 0, 0 };
  int n_words = 0;
  long long valid = 0;
  uint32_t* d_bits = nullptr;
};

extern "C" rs_hip_coverage_t* rs_hip_coverage_create( const float bbox_min[3], const float bbox_max[3], float voxel_size,
                                                      const float* scene_pos, const float* scene_quality, int64_t n_scene,
                                                      float quality_threshold )
{
  if( ensure_ready() ) return nullptr;
  if( !bbox_min || !bbox_max || !( voxel_size > 0.0f ) || n_scene < 0 || ( n_scene > 0 && !scene_pos ) ) { set_err( "coverage_create: bad arguments" ); return nullptr; }
  rs_hip_coverage_t* c = new rs_hip_coverage_t();
  // isect_grid3d_init (lib/rs/intersect.h:59-75), same float operations
  const float fat = 0.3f;
  float mn[3], mx[3];
  for( int a = 0; a < 3; ++a ) { mn[a] = bbox_min[a] - fat; mx[a] = bbox_max[a] + fat; }
  const double cells = ( (double)std::ceil( ( mx[0] - mn[0] ) / voxel_size ) + 1 ) * ( (double)std::ceil( ( mx[1] - mn[1] ) / voxel_size ) + 1 ) *
                       ( (double)std::ceil( ( mx[2] - mn[2] ) / voxel_size ) + 1 );
  if( !( cells > 0 ) || cells > 2.0e9 ) { set_err( "coverage_create: %g voxels do not fit the reference's int32 cell index", cells ); delete c; return nullptr; }
  c->grid.x_res = (int)std::ceil( ( mx[0] - mn[0] ) / voxel_size ) + 1;
  c->grid.y_res = (int)std::ceil( ( mx[1] - mn[1] ) / voxel_size ) + 1;
  c->grid.z_res = (int)std::ceil( ( mx[2] - mn[2] ) / voxel_size ) + 1;
  c->grid.n_cells = c->grid.x_res * c->grid.y_res * c->grid.z_res;
  c->grid.ox = mn[0]; c->grid.oy = mn[1]; c->grid.oz = mn[2];
  c->grid.inv_voxel = 1.0f / voxel_size;                      // :100
  c->voxel_size = voxel_size; for( int a = 0; a < 3; ++a ) c->origin[a] = mn[a];
  c->n_words = ( c->grid.n_cells + 31 ) / 32;
  auto fail = [&]( const char* what ) { set_err( "coverage_create: %s", what ); if( c->d_bits ) (void)hipFree( c->d_bits ); delete c; return (rs_hip_coverage_t*)nullptr; };
  if( hipMalloc( &c->d_bits, (size_t)c->n_words * 4 ) != hipSuccess ) return fail( "hipMalloc failed" );
  if( hipMemsetAsync( c->d_bits, 0, (size_t)c->n_words * 4, g_stream ) != hipSuccess ) return fail( "memset failed" );
  float *d_pos = nullptr, *d_q = nullptr; int* d_cnt = nullptr;
  bool ok = hipMalloc( &d_cnt, 4 ) == hipSuccess && hipMemsetAsync( d_cnt, 0, 4, g_stream ) == hipSuccess;
  if( ok && n_scene > 0 )
  {
    ok = hipMalloc( &d_pos, (size_t)n_scene * 12 ) == hipSuccess &&
         hipMemcpyAsync( d_pos, scene_pos, (size_t)n_scene * 12, hipMemcpyHostToDevice, g_stream ) == hipSuccess;
    if( ok && scene_quality )
      ok = hipMalloc( &d_q, (size_t)n_scene * 4 ) == hipSuccess &&
           hipMemcpyAsync( d_q, scene_quality, (size_t)n_scene * 4, hipMemcpyHostToDevice, g_stream ) == hipSuccess;
    if( ok ) launch_voxel_mark( c->grid, d_pos, d_q, quality_threshold, n_scene, c->d_bits, g_stream );
  }
  int valid = 0;
  if( ok ) { launch_popcount( c->d_bits, c->n_words, d_cnt, g_stream ); ok = hipMemcpyAsync( &valid, d_cnt, 4, hipMemcpyDeviceToHost, g_stream ) == hipSuccess; }
  ok = ok && hipStreamSynchronize( g_stream ) == hipSuccess;
  if( d_pos ) (void)hipFree( d_pos );
  if( d_q ) (void)hipFree( d_q );
  if( d_cnt ) (void)hipFree( d_cnt );
  if( !ok ) return fail( "device step failed" );
  c->valid = valid;
  return c;
}

extern "C" void rs_hip_coverage_destroy( rs_hip_coverage_t* c )
{
  if( !c ) return;
  if( c->d_bits ) (void)hipFree( c->d_bits );
  delete c;
}

extern "C" int rs_hip_coverage_info( const rs_hip_coverage_t* c, int32_t res[3], float origin[3], int64_t* n_cells, int64_t* valid_cells )
{
  if( !c ) { set_err( "coverage_info: null handle" ); return RS_HIP_E_ARG; }
  if( res ) { res[0] = c->grid.x_res; res[1] = c->grid.y_res; res[2] = c->grid.z_res; }
  if( origin ) { origin[0] = c->origin[0]; origin[1] = c->origin[1]; origin[2] = c->origin[2]; }
  if( n_cells ) *n_cells = c->grid.n_cells;
  if( valid_cells ) *valid_cells = c->valid;
  return RS_HIP_OK;
}

extern "C" int rs_hip_coverage_scene_grid( const rs_hip_coverage_t* c, uint8_t* data )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !c || !data ) { set_err( "coverage_scene_grid: bad arguments" ); return RS_HIP_E_ARG; }
  std::vector<uint32_t> bits( (size_t)c->n_words );
  HIP_TRY( hipMemcpyAsync( bits.data(), c->d_bits, bits.size() * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );
  for( int i = 0; i < c->grid.n_cells; ++i ) data[i] = ( bits[(size_t)i >> 5] >> ( i & 31 ) ) & 1u;
  return RS_HIP_OK;
}

extern "C" int rs_hip_coverage_scores( rs_hip_coverage_t* c, const rs_hip_cloud_t* const* objects, const float* poses,
                                       const int32_t* is_static, const int32_t* first_placement, int32_t n_arr,
                                       float* scores, int32_t* agree )
{
  int rc = ensure_ready(); if( rc ) return rc;
  if( !c || n_arr < 0 || !scores || ( n_arr > 0 && !first_placement ) ) { set_err( "coverage_scores: bad arguments" ); return RS_HIP_E_ARG; }
  if( n_arr == 0 ) return RS_HIP_OK;
  const int n_plc_all = first_placement[n_arr];
  if( n_plc_all < 0 || ( n_plc_all > 0 && ( !objects || !poses || !is_static ) ) ) { set_err( "coverage_scores: bad placement lists" ); return RS_HIP_E_ARG; }
  if( (double)n_arr * c->n_words * 4 > 8.0e9 ) { set_err( "coverage_scores: batch of %d arrangements needs more than 8 GB of voxel bitmaps", n_arr ); return RS_HIP_E_CAPACITY; }
  std::vector<CoveragePlacement> h;
  int max_pts = 0;
  for( int a = 0; a < n_arr; ++a )
  {
    if( first_placement[a] > first_placement[a + 1] ) { set_err( "coverage_scores: first_placement must be non-decreasing" ); return RS_HIP_E_ARG; }
    for( int k = first_placement[a]; k < first_placement[a + 1]; ++k )
    {
      if( is_static[k] ) continue;                              // :1095-1096
      if( !objects[k] ) { set_err( "coverage_scores: placement %d has no object cloud", k ); return RS_HIP_E_ARG; }
      CoveragePlacement p{};
      p.pos = objects[k]->view.pos; p.n = objects[k]->n; p.arrangement = a;
      std::memcpy( p.pose.m, poses + 16 * (size_t)k, 64 );
      if( p.n > 0 ) { h.push_back( p ); max_pts = std::max( max_pts, p.n ); }
    }
  }
  const size_t bits_bytes = (size_t)n_arr * c->n_words * 4;
  if( ( rc = g_ws.cov_bits.ensure( bits_bytes ) ) || ( rc = g_ws.cov_agree.ensure( (size_t)n_arr * 4 ) ) ||
      ( rc = g_ws.cov_plc.ensure( std::max<size_t>( 1, h.size() ) * sizeof(CoveragePlacement) ) ) ) return rc;
  HIP_TRY( hipMemsetAsync( g_ws.cov_bits.p, 0, bits_bytes, g_stream ), RS_HIP_E_RUNTIME );      // :1090 memset
  HIP_TRY( hipMemsetAsync( g_ws.cov_agree.p, 0, (size_t)n_arr * 4, g_stream ), RS_HIP_E_RUNTIME );
  if( !h.empty() )
  {
    HIP_TRY( hipMemcpyAsync( g_ws.cov_plc.p, h.data(), h.size() * sizeof(CoveragePlacement), hipMemcpyHostToDevice, g_stream ), RS_HIP_E_RUNTIME );
    CoverageLaunch L{};
    L.grid = c->grid; L.scene_bits = c->d_bits; L.arr_bits = g_ws.cov_bits.as<uint32_t>(); L.n_words = c->n_words;
    L.plc = g_ws.cov_plc.as<CoveragePlacement>(); L.n_plc = (int)h.size(); L.max_pts = max_pts; L.agree = g_ws.cov_agree.as<int>();
    { ProfScope ps( "coverage" ); launch_coverage( L, g_stream ); }
  }
  std::vector<int> cnt( (size_t)n_arr );
  HIP_TRY( hipMemcpyAsync( cnt.data(), g_ws.cov_agree.p, (size_t)n_arr * 4, hipMemcpyDeviceToHost, g_stream ), RS_HIP_E_RUNTIME );
  HIP_TRY( hipStreamSynchronize( g_stream ), RS_HIP_E_RUNTIME );      // also keeps `h` alive until the upload is done
  for( int a = 0; a < n_arr; ++a )
  {
    float s = (float)cnt[a] / (float)c->valid;                   // :366
    if( c->valid == 0 ) s = 0.0f;                                // :367
    scores[a] = s;
    if( agree ) agree[a] = cnt[a];
  }
  return RS_HIP_OK;
}
