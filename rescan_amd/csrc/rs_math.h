// Host-side small linear algebra with the reference's float arithmetic, operation for
// operation (column-major float[16], lib/msh/msh_vec_math.h).  These run on the CPU side of
// the ICP loop (pose composition, 6x6 solve) exactly as the reference does (lib/rs/icp.h:267-295);
// the GPU does the searches and the reductions.
#pragma once
#include <cmath>
#include <cstring>
#include <cstdint>

namespace rs {

struct Mat4 { float m[16]; };

inline Mat4 mat4_identity() { Mat4 r; std::memset( r.m, 0, sizeof(r.m) ); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }

// msh_mat4_mul (msh_vec_math.h:1441-1476): element (row r, col c) = Σ_k b[k,c]·a[r,k], k ascending.
inline Mat4 mat4_mul( const Mat4& a, const Mat4& b )
{
  Mat4 o;
  for( int c = 0; c < 4; ++c )
    for( int r = 0; r < 4; ++r )
      o.m[4*c+r] = b.m[4*c] * a.m[r] + b.m[4*c+1] * a.m[4+r] + b.m[4*c+2] * a.m[8+r] + b.m[4*c+3] * a.m[12+r];
  return o;
}

inline Mat4 mat4_transpose( const Mat4& a )
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
inline Mat4 mat4_translate( const Mat4& a, float tx, float ty, float tz )
{
  Mat4 o = a;
  for( int r = 0; r < 4; ++r )
    o.m[12+r] = ( a.m[r] * tx + a.m[4+r] * ty ) + ( a.m[8+r] * tz + a.m[12+r] );
  return o;
}

// msh_rotate (msh_vec_math.h:2089-2132) about a coordinate axis (0 = x, 1 = y, 2 = z); the
// reference normalises the unit axis first, which is exact for these three.
inline Mat4 mat4_rotate_axis( const Mat4& a, float angle, int axis )
{
  float c = cosf( angle ), s = sinf( angle ), t = 1.0f - c;
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
inline void ldlt6_solve( double A[6][6], const double b[6], double x[6] )
{
  double rd[6] = { 0, 0, 0, 0, 0, 0 }, v[5];
  bool ok = true;
  for( int i = 0; i < 6 && ok; ++i )
  {
    for( int k = 0; k < i; ++k ) v[k] = A[i][k] * rd[k];
    for( int j = i; j < 6; ++j )
    {
      double sum = A[i][j];
      for( int k = 0; k < i; ++k ) sum -= v[k] * A[j][k];
      if( i == j ) { if( sum == 0 ) { ok = false; break; } rd[i] = 1 / sum; }
      else A[j][i] = sum;
    }
  }
  for( int i = 0; i < 6; ++i )
  {
    double sum = b[i];
    for( int k = 0; k < i; ++k ) sum -= A[i][k] * x[k];
    x[i] = sum * rd[i];
  }
  for( int i = 5; i >= 0; --i )
  {
    double sum = 0;
    for( int k = i + 1; k < 6; ++k ) sum += A[k][i] * x[k];
    x[i] -= sum * rd[i];
  }
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
