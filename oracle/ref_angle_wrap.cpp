// ref_angle_wrap.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" shim around the REFERENCE's own lidar_slam::Angle
// (/root/reference/L_SLAM/src/util/Angle.h, included from where it lies -- it needs <cmath> only;
// Twist.h / math_utils.h / transform_utils.h pull in PCL and Eigen through Vector3.h and cannot be
// built here).  Pins the pose-angle state handling of the oracle: the cached
// std::sin / std::cos of a float radian and `a += x` == Angle(a.rad() + x).
//
// Built by oracle/Makefile into oracle/_ref/libref_angle.so (git-ignored).
#include "Angle.h"

extern "C" {
// out = {rad, sin, cos} of Angle(rad), then of the same object after `+= add`
void ref_angle_state(float rad, float add, float out[6]) {
  lidar_slam::Angle a(rad);
  out[0] = a.rad();
  out[1] = a.sin();
  out[2] = a.cos();
  a += add;
  out[3] = a.rad();
  out[4] = a.sin();
  out[5] = a.cos();
}
}
