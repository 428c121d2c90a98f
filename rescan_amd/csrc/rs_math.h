// Small linear algebra with the reference's float arithmetic, operation for operation
// (column-major float[16], lib/msh/msh_vec_math.h): pose composition and the 6x6 solve of
// lib/rs/icp.h:267-295.  Compiled for the host (drop-in shim, estimate-only entry point) and, under
// hipcc, for the device as well: the ICP loop finishes every iteration on the GPU (the rs_*.hip kernel files:
// k_icp_update), so no host round trip sits between two searches.
#pragma once
#include <cmath>
#include <cstring>
#include <cstdint>

#if defined( __HIPCC__ )
#define RS_HD __host__ __device__
#else
#define RS_HD
#endif

namespace rs {

struct Mat4 { float m[16]; };

RS_HD inline Mat4 mat4_identity() { Mat4 r; for( int i = 0; i < 16; ++i ) r.m[i] = 0.0f; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }

// sinf/cosf as glibc >= 2.28 computes them (sysdeps/ieee754/flt-32/s_sincosf.h, the algorithm of Arm's optimized
// routines): a double-precision polynomial of the published coefficients, 0.56 ulp — NOT the correctly rounded value
// (it differs from (float)sin((double)x) for 0.85 % of the arguments), and the reference's poses carry exactly these
// bits (msh_rotate, msh_vec_math.h:2091-2092).  The fused steps are the ones of libm's FMA build, which every x86-64
// host of the last decade selects; pinned against the host's sinf/cosf on 6e8 arguments with zero mismatches
// (tests/test_capi_cpu.py checks a sample on every run).  |x| >= 120, inf and NaN (never an ICP step) fall back to
// the double-precision function rounded once.
RS_HD inline float rs_sincosf_poly( double x, double x2, bool cosine, bool negate_cos )
{
  // __sincosf_table[0]; table[1] is the same with c0..c4 negated
  const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  if( !cosine )
  {
    const double x3 = x * x2, t1 = fma( x2, s3, s2 ), x7 = x3 * x2, t = fma( x3, s1, x );
    return (float)fma( x7, t1, t );
  }
  const double g = negate_cos ? -1.0 : 1.0;
  const double x4 = x2 * x2, t2 = fma( x2, g * c4, g * c3 ), t1 = fma( x2, g * c1, g * c0 ), x6 = x4 * x2, t = fma( x4, g * c2, t1 );
  return (float)fma( x6, t2, t );
}
RS_HD inline void rs_sincosf_model( float a, float& s, float& c )
{
  uint32_t u; memcpy( &u, &a, 4 );
  const uint32_t top12 = ( u >> 20 ) & 0x7ff;
  const double x = a;
  if( top12 < 0x3f4 )                              // |a| < 0.75 (abstop12 compare against pi/4)
  {
    if( top12 < 0x398 ) { s = a; c = 1.0f; return; }   // |a| < 2^-12
    const double x2 = x * x;
    s = rs_sincosf_poly( x, x2, false, false ); c = rs_sincosf_poly( x, x2, true, false );
    return;
  }
  if( top12 < 0x42f )                              // |a| < 120: reduce_fast
  {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double r = x * hpi_inv;
    const int n = ( (int32_t)r + 0x800000 ) >> 24;
    const double xr = fma( -(double)n, hpi, x );
    const double sign = ( ( n & 3 ) == 1 || ( n & 3 ) == 2 ) ? -1.0 : 1.0;     // { 1, -1, -1, 1 }[n & 3]
    const double xs = xr * sign, x2 = xr * xr;
    const bool tab1 = ( n & 2 ) != 0;
    s = rs_sincosf_poly( xs, x2, ( n & 1 ) != 0, tab1 );
    c = rs_sincosf_poly( xs, x2, ( ( n ^ 1 ) & 1 ) != 0, tab1 );
    return;
  }
  s = (float)sin( x ); c = (float)cos( x );
}

// cosf/sinf of the host libm — by definition what the reference calls; on the device the model above.
RS_HD inline void rs_sincosf( float a, float& s, float& c )
{
#if defined( __HIP_DEVICE_COMPILE__ )
  rs_sincosf_model( a, s, c );
#else
  c = cosf( a ); s = sinf( a );
#endif
}

// msh_mat4_mul (msh_vec_math.h:1441-1476): element (row r, col c) = Σ_k b[k,c]·a[r,k], k ascending.
RS_HD inline Mat4 mat4_mul( const Mat4& a, const Mat4& b )
{
  Mat4 o;
  for( int c = 0; c < 4; ++c )
    for( int r = 0; r < 4; ++r )
      o.m[4*c+r] = b.m[4*c] * a.m[r] + b.m[4*c+1] * a.m[4+r] + b.m[4*c+2] * a.m[8+r] + b.m[4*c+3] * a.m[12+r];
  return o;
}

RS_HD inline Mat4 mat4_transpose( const Mat4& a )
{
  Mat4 o;
  for( int c = 0; c < 4; ++c ) for( int r = 0; r < 4; ++r ) o.m[4*c+r] = a.m[4*r+c];
  return o;
}

// msh_mat4_inverse (msh_vec_math.h:1818-1905): adjugate / determinant built from two banks of
// 2x2 minors.  The term tables below encode, per cofactor, (matrix element, minor, sign) in
// the order the reference adds them, so the float rounding sequence is the same.
inline Mat4 mat4_inverse( const Mat4& A )
{
  const float* m = A.m;
  // minors: bank 0 uses rows 2,3 of the transposed view, bank 1 rows 0,1
  static const int MI[2][6][4] = {
    { {10,15,14,11}, {6,11,10,7}, {2,7,6,3}, {6,15,14,7}, {2,11,10,3}, {2,15,14,3} },
    { {8,13,12,9},   {4,9,8,5},   {0,5,4,1}, {4,13,12,5}, {0,9,8,1},   {0,13,12,1} } };
  // cofactor k = s0*m[e0]*minor[d0] + s1*m[e1]*minor[d1] + s2*m[e2]*minor[d2] (first term always +)
  struct Term { int e, d; };
  static const Term CT[16][3] = {
    {{5,0},{9,3},{13,1}}, {{9,5},{1,0},{13,4}}, {{1,3},{5,5},{13,2}}, {{5,4},{9,2},{1,1}},
    {{8,3},{4,0},{12,1}}, {{0,0},{8,5},{12,4}}, {{4,5},{0,3},{12,2}}, {{0,1},{4,4},{8,2}},
    {{7,0},{11,3},{15,1}}, {{11,5},{3,0},{15,4}}, {{3,3},{7,5},{15,2}}, {{7,4},{3,1},{11,2}},
    {{10,3},{6,0},{14,1}}, {{2,0},{10,5},{14,4}}, {{6,5},{2,3},{14,2}}, {{2,1},{6,4},{10,2}} };
  // sign of the 2nd and 3rd terms (+1: add, -1: subtract)
  static const int CS[16][2] = {
    {-1,+1},{-1,-1},{-1,+1},{-1,-1}, {-1,-1},{-1,+1},{-1,-1},{-1,+1},
    {-1,+1},{-1,-1},{-1,+1},{-1,-1}, {-1,-1},{-1,+1},{-1,-1},{-1,+1} };
  float C[16];
  for( int bank = 0; bank < 2; ++bank )
  {
    float s[6];
    for( int i = 0; i < 6; ++i ) { const int* q = MI[bank][i]; s[i] = m[q[0]] * m[q[1]] - m[q[2]] * m[q[3]]; }
    for( int k = 8 * bank; k < 8 * bank + 8; ++k )
    {
      float t0 = m[CT[k][0].e] * s[CT[k][0].d];
      float t1 = m[CT[k][1].e] * s[CT[k][1].d];
      float t2 = m[CT[k][2].e] * s[CT[k][2].d];
      float acc = t0 - t1;                       // second term is subtracted in all 16 cofactors
      acc = ( CS[k][1] > 0 ) ? acc + t2 : acc - t2;
      C[k] = acc;
    }
  }
  float det = m[0] * C[0] + m[4] * C[1] + m[8] * C[2] + m[12] * C[3];
  float inv_det = 1.0f / det;
  Mat4 o;
  for( int i = 0; i < 16; ++i ) o.m[i] = inv_det * C[i];
  return o;
}

// msh_translate (msh_vec_math.h:2064-2074): col3 = (col0·tx + col1·ty) + (col2·tz + col3)
RS_HD inline Mat4 mat4_translate( const Mat4& a, float tx, float ty, float tz )
{
  Mat4 o = a;
  for( int r = 0; r < 4; ++r )
    o.m[12+r] = ( a.m[r] * tx + a.m[4+r] * ty ) + ( a.m[8+r] * tz + a.m[12+r] );
  return o;
}

// msh_rotate (msh_vec_math.h:2089-2132) about a coordinate axis (0 = x, 1 = y, 2 = z); the
// reference normalises the unit axis first, which is exact for these three.
RS_HD inline Mat4 mat4_rotate_axis( const Mat4& a, float angle, int axis )
{
  float c, s; rs_sincosf( angle, s, c );
  float t = 1.0f - c;
  float ax[3] = { 0.0f, 0.0f, 0.0f }; ax[axis] = 1.0f;
  float R[9];   // R[3*col + row]
  R[0] = c + ax[0] * ax[0] * t;  R[4] = c + ax[1] * ax[1] * t;  R[8] = c + ax[2] * ax[2] * t;
  float p = ax[0] * ax[1] * t, q = ax[2] * s;  R[1] = p + q;  R[3] = p - q;
  p = ax[0] * ax[2] * t;  q = ax[1] * s;        R[2] = p - q;  R[6] = p + q;
  p = ax[1] * ax[2] * t;  q = ax[0] * s;        R[5] = p + q;  R[7] = p - q;
  Mat4 o = a;
  for( int j = 0; j < 3; ++j )
    for( int r = 0; r < 4; ++r )
      o.m[4*j+r] = a.m[r] * R[3*j] + ( a.m[4+r] * R[3*j+1] + a.m[8+r] * R[3*j+2] );
  return o;
}

// Unpivoted LDLᵀ of a symmetric 6x6 (upper triangle read) and the two triangular solves, as
// trimesh's ldltdc/ldltsl instantiated at <double,6> (lib/rs/lineqn.h:153-218).  A zero pivot
// stops the factorisation; like the reference's caller (icp.h:276) we solve with whatever was
// produced.
RS_HD inline void ldlt6_solve( double A[6][6], const double b[6], double x[6] )
{
  // fully unrolled (fixed trip counts, no early exits from the loops) so that on the device every
  // array stays in registers; `ok` turns the remaining steps into no-ops after a zero pivot
  double rd[6] = { 0, 0, 0, 0, 0, 0 }, v[6] = { 0, 0, 0, 0, 0, 0 };
  bool ok = true;
#pragma unroll
  for( int i = 0; i < 6; ++i )
  {
#pragma unroll
    for( int k = 0; k < 6; ++k ) if( k < i ) v[k] = A[i][k] * rd[k];
#pragma unroll
    for( int j = 0; j < 6; ++j )
    {
      if( j < i || !ok ) continue;
      double sum = A[i][j];
#pragma unroll
      for( int k = 0; k < 6; ++k ) if( k < i ) sum -= v[k] * A[j][k];
      if( i == j ) { if( sum == 0 ) ok = false; else rd[i] = 1 / sum; }
      else A[j][i] = sum;
    }
  }
#pragma unroll
  for( int i = 0; i < 6; ++i )
  {
    double sum = b[i];
#pragma unroll
    for( int k = 0; k < 6; ++k ) if( k < i ) sum -= A[i][k] * x[k];
    x[i] = sum * rd[i];
  }
#pragma unroll
  for( int i = 5; i >= 0; --i )
  {
    double sum = 0;
#pragma unroll
    for( int k = 0; k < 6; ++k ) if( k > i ) sum += A[k][i] * x[k];
    x[i] -= sum * rd[i];
  }
}

// Finish lib/rs/icp.h:210-298 from the 35 uncentred fp64 moments (layout: rs_icp_estimate.hip, k_icp_moments).
// Returns false when the reference would have stopped before estimating (Σw <= 1e-7, icp.h:466).
// cen (may be null): the two weighted centroids c1, c2 to centre on (6 floats) instead of the moments' own Σw·p / Σw, Σw·q / Σw.
// The reference's centroids are sequential fp32 sums (icp.h:136-148) and carry a SYSTEMATIC rounding error: once the running
// sum's grid (its ulp) is coarser than the spread of the addends — Σw of 7 x 10^5 weights near 0.97 advances on a grid of
// 1/16 and every addend rounds UP to 1.0 — the sum drifts by parts in a thousand (Σw + 0.27 %, the centroids by a
// centimetre, their difference c1 - c2 by a millimetre at a far start pose), and the residuals d = (p - c1) - (q - c2) inherit
// c1 - c2.  An estimator that wants the reference's pose for a million-point source must centre where the reference centres.
RS_HD inline bool icp_solve( const double* M, Mat4& T1, float& err, const float* cen = nullptr )
{
  const double W = M[0];
  if( (float)W <= 1e-7 ) return false;
  const float c1f[3] = { cen ? cen[0] : (float)( M[1] / W ), cen ? cen[1] : (float)( M[2] / W ), cen ? cen[2] : (float)( M[3] / W ) };
  const float c2f[3] = { cen ? cen[3] : (float)( M[4] / W ), cen ? cen[4] : (float)( M[5] / W ), cen ? cen[5] : (float)( M[6] / W ) };
  const double c1[3] = { c1f[0], c1f[1], c1f[2] }, dl[3] = { (double)c1f[0] - c2f[0], (double)c1f[1] - c2f[1], (double)c1f[2] - c2f[2] };
  const double Maa[3][3] = { { M[7], M[8], M[9] }, { M[8], M[10], M[11] }, { M[9], M[11], M[12] } };
  double Man[3][3];
  for( int r = 0; r < 3; ++r ) for( int c = 0; c < 3; ++c ) Man[r][c] = M[13 + 3*r + c];
  const double Mnn[3][3] = { { M[22], M[23], M[24] }, { M[23], M[25], M[26] }, { M[24], M[26], M[27] } };
  const double vae[3] = { M[28], M[29], M[30] }, vne[3] = { M[31], M[32], M[33] };
  const double see = M[34];
  // X v = c1 × v
  const double X[3][3] = { { 0, -c1[2], c1[1] }, { c1[2], 0, -c1[0] }, { -c1[1], c1[0], 0 } };

  double XMnn[3][3], ManXt[3][3], XMnnXt[3][3];
  for( int r = 0; r < 3; ++r ) for( int c = 0; c < 3; ++c )
  {
    double s = 0; for( int k = 0; k < 3; ++k ) s += X[r][k] * Mnn[k][c];     XMnn[r][c] = s;       // X·Mnn
    s = 0;        for( int k = 0; k < 3; ++k ) s += Man[r][k] * X[c][k];     ManXt[r][c] = s;      // Man·Xᵀ
  }
  for( int r = 0; r < 3; ++r ) for( int c = 0; c < 3; ++c )
  { double s = 0; for( int k = 0; k < 3; ++k ) s += XMnn[r][k] * X[c][k]; XMnnXt[r][c] = s; }       // X·Mnn·Xᵀ
  double TL[3][3], TR[3][3];
  for( int r = 0; r < 3; ++r ) for( int c = 0; c < 3; ++c )
  {
    TL[r][c] = Maa[r][c] - ManXt[r][c] - ManXt[c][r] + XMnnXt[r][c];   // Σw·c cᵀ, c = a - X n
    TR[r][c] = Man[r][c] - XMnn[r][c];                                   // Σw·c nᵀ
  }
  double bc[3], bn[3], Mnn_d[3], Man_d[3], sum;
  for( int r = 0; r < 3; ++r ) { Mnn_d[r] = 0; Man_d[r] = 0; for( int k = 0; k < 3; ++k ) { Mnn_d[r] += Mnn[r][k] * dl[k]; Man_d[r] += Man[r][k] * dl[k]; } }
  for( int r = 0; r < 3; ++r )
  {
    double xv = 0, xm = 0;
    for( int k = 0; k < 3; ++k ) { xv += X[r][k] * vne[k]; xm += X[r][k] * Mnn_d[k]; }
    bc[r] = vae[r] - Man_d[r] - xv + xm;       // Σw·c·s,  s = e - δ·n
    bn[r] = vne[r] - Mnn_d[r];                 // Σw·n·s
  }
  sum = see - 2.0 * ( dl[0] * vne[0] + dl[1] * vne[1] + dl[2] * vne[2] ) + ( dl[0] * Mnn_d[0] + dl[1] * Mnn_d[1] + dl[2] * Mnn_d[2] );
  if( sum < 0.0 ) sum = 0.0;
  err = (float)sqrt( sum / W );                // icp.h:253

  double C[6][6], b[6], x[6] = { 0, 0, 0, 0, 0, 0 };    // icp.h:267-277
  for( int r = 0; r < 3; ++r ) for( int c = 0; c < 3; ++c )
  { C[r][c] = TL[r][c]; C[r][3+c] = TR[r][c]; C[3+r][c] = TR[c][r]; C[3+r][3+c] = Mnn[r][c]; }
  for( int r = 0; r < 3; ++r ) { b[r] = -bc[r]; b[3+r] = -bn[r]; }
  ldlt6_solve( C, b, x );

  Mat4 T = mat4_identity();                    // icp.h:280-295
  T = mat4_translate( T, c1f[0], c1f[1], c1f[2] );
  T = mat4_translate( T, (float)x[3], (float)x[4], (float)x[5] );
  T = mat4_rotate_axis( T, (float)x[0], 0 );
  T = mat4_rotate_axis( T, (float)x[1], 1 );
  T = mat4_rotate_axis( T, (float)x[2], 2 );
  T = mat4_translate( T, -c1f[0], -c1f[1], -c1f[2] );
  T1 = mat4_mul( T, T1 );
  return true;
}

// The same step from the reference's own accumulators (k_icp_faithful): A = TL[9] TR[9] BR[9] (column-major
// 3x3 floats, M[3*col+row]) and rhs[6], all fp32; sum = Σw·s² and tw = Σw in fp64; c1 = the fp32 centroid
// (icp.h:253-295).
RS_HD inline void icp_solve_ref_order( const float* A, double sum, double tw, const float c1f[3], Mat4& T1, float& err )
{
  const float* TL = A; const float* TR = A + 9; const float* BR = A + 18; const float* rhs = A + 27;
  err = (float)sqrt( sum / tw );
  double C[6][6], b[6], x[6] = { 0, 0, 0, 0, 0, 0 };    // icp.h:267-277
  for( int r = 0; r < 3; ++r ) for( int c = 0; c < 3; ++c )
  { C[r][c] = TL[3*c + r]; C[r][3+c] = TR[3*c + r]; C[3+r][c] = TR[3*r + c]; C[3+r][3+c] = BR[3*c + r]; }
  for( int i = 0; i < 6; ++i ) b[i] = -rhs[i];
  ldlt6_solve( C, b, x );
  Mat4 T = mat4_identity();                    // icp.h:280-295
  T = mat4_translate( T, c1f[0], c1f[1], c1f[2] );
  T = mat4_translate( T, (float)x[3], (float)x[4], (float)x[5] );
  T = mat4_rotate_axis( T, (float)x[0], 0 );
  T = mat4_rotate_axis( T, (float)x[1], 1 );
  T = mat4_rotate_axis( T, (float)x[2], 2 );
  T = mat4_translate( T, -c1f[0], -c1f[1], -c1f[2] );
  T1 = mat4_mul( T, T1 );
}

// ---- normal gates as thresholds on the (clamped) dot product ---------------------------
// The reference tests acos(dot) against an angle with the host libm.  acos is monotone, so
// each gate equals "dot >= t", t = the smallest float for which the host expression holds.
// The bisection below runs the *host's* acosf/acos, i.e. the same libm the reference app
// would call on this machine, and the kernels then only compare floats.

template <class Pred> inline float gate_threshold( Pred accept )
{
  // accept(x) is (assumed) monotone on [0,1]: false ... false true ... true
  if( !accept( 1.0f ) ) return 2.0f;            // nothing passes (dot <= 1 is enforced separately)
  if( accept( 0.0f ) ) return 0.0f;             // everything passes
  uint32_t lo = 0u, hi = 0x3f800000u;           // bit patterns of 0.0f and 1.0f; order-preserving for x >= 0
  while( hi - lo > 1u )
  {
    uint32_t mid = lo + ( hi - lo ) / 2u;
    float x; std::memcpy( &x, &mid, 4 );
    if( accept( x ) ) hi = mid; else lo = mid;
  }
  float t; std::memcpy( &t, &hi, 4 );
  return t;
}

// icp.h:372-374: acosf(max(dot,0)) < max_angle
inline float icp_gate_threshold( float max_angle ) { return gate_threshold( [=]( float x ) { return acosf( x ) < max_angle; } ); }
// pose_proposal.cpp:99,138-141: acos((double)max(dot,0)) - deg2rad(35) < 1e-6, msh_deg2rad(x) = x*0.005555555556*MSH_PI
inline double ref_deg2rad( double x ) { return x * 0.005555555556 * 3.1415926535897932384626433832; }
inline float score_gate_threshold() { const double a = ref_deg2rad( 35.0 ); return gate_threshold( [=]( float x ) { return acos( (double)x ) - a < 0.000001; } ); }
// rs_pointcloud_filters.cpp:769-770: (double)acosf(|dot|) < deg2rad(70)
inline float label_gate_threshold() { const double a = ref_deg2rad( 70.0 ); return gate_threshold( [=]( float x ) { return (double)acosf( x ) < a; } ); }

} // namespace rs
