// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Part of oracle/_ref (see ref_driver.cpp).
//
// The label-transfer normal gate of lib/rs/rs_pointcloud_filters.cpp:765-770.
// That file cannot be built here (it includes the un-vendored gco header), but
// whether its unqualified `acos(fabs(float))` resolves to the float or the
// double libm entry depends on its include preamble ("math.h", i.e. libstdc++'s
// C++ wrapper that pulls std::acos/std::fabs overloads into the global
// namespace).  This TU therefore uses the same preamble as
// rs_pointcloud_filters.cpp:1-14 (minus GCoptimization.h) and evaluates the gate
// with the reference's own msh_* primitives, so the compiler makes the same choice.
#include <algorithm>

#include "stdio.h"
#include "stdint.h"
#include "math.h"

#include "mg/hashtable.h"
#include "msh/msh_std.h"
#include "msh/msh_vec_math.h"
#include "msh/msh_geometry.h"
#include "msh/msh_hash_grid.h"

#include <string.h>

// n_scene is the scene normal *before* the normal_matrix multiply; returns the
// reference's accept decision for (pose, scene normal, matched object normal).
extern "C" int ref_label_gate( const float* pose, const float* n_scene, const float* n_obj )
{
  msh_mat4_t xform; memcpy( xform.data, pose, 64 );
  msh_mat4_t normal_matrix = msh_mat4_transpose( xform );
  msh_vec3_t n1 = msh_vec3( n_scene[0], n_scene[1], n_scene[2] );
  n1 = msh_mat4_vec3_mul( normal_matrix, n1, 0 );
  msh_vec3_t n2 = msh_vec3( n_obj[0], n_obj[1], n_obj[2] );
  float angle = acos( fabs( msh_vec3_dot( msh_vec3_normalize(n1), msh_vec3_normalize(n2) ) ) );
  return ( angle < msh_deg2rad(70.0) ) ? 1 : 0;
}

// The same expression on a raw |dot| value, for threshold pinning.
extern "C" int ref_label_gate_dot( float dot )
{
  float angle = acos( fabs( dot ) );
  return ( angle < msh_deg2rad(70.0) ) ? 1 : 0;
}

// The edge weight of rspf_compute_neighborhood (lib/rs/rs_pointcloud_filters.cpp:706-708), spelled
// as the reference spells it and compiled under the same preamble, so pow() resolves to the same
// overloads (double pow for the distance term, powf for the float/float normal term).
extern "C" float ref_edge_cost( float nn_dist, float dot_nm, float radius_sq, float dist_exp, float angle_exp )
{
  float dist_cost  = 1.0f - pow(nn_dist/(4.0*radius_sq), dist_exp);
  float norm_cost  = pow( msh_clamp(dot_nm, 0.0f, 1.0f), angle_exp);
  float cost = dist_cost * norm_cost;
  return cost;
}
