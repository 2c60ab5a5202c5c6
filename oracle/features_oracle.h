/*
 * features_oracle.h -- CPU ORACLE for the feature-extraction front end:
 * ScanRegistration::extractFeatures on a ring-sorted cloud.  TEST INFRASTRUCTURE, NOT PRODUCT CODE
 * (see lslam_oracle.h); citations and pin status ("parity unpinned") in features_oracle.c.
 */
#ifndef LSLAM_FEATURES_ORACLE_H
#define LSLAM_FEATURES_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int n_feature_regions;    /* 6 */
  int curvature_region;     /* 5 */
  int max_corner_sharp;     /* 2 */
  int max_surface_flat;     /* 4 */
  float less_flat_filter_size;        /* 0.2 */
  float surface_curvature_threshold;  /* 0.02 */
  float blind_threshold;              /* cos(deg2rad(0.5)) */
} oracle_reg_params;

void oracle_reg_default_params(oracle_reg_params *p);

/* pointClassify (ScanRegistration.cpp:557-687): -1 flat, 1 corner sharp, 5 one-side flat, 9 messy */
int oracle_point_classify(const float *cloud, size_t stride_floats, size_t idx, int curvature_region);

/* extractFeatures (ScanRegistration.cpp:190-425).  cloud: n_points x stride_floats, xyz first,
 * curvature_field = index of the float copied to the output intensity (toXYZI).  scan_ranges:
 * n_scans x {first, last} inclusive.  Outputs hold up to n_points {x,y,z,intensity} each;
 * counts = {sharp, less sharp, flat, less flat}.  Optional taps (may be NULL): per-point curvature,
 * _scanNeighborPicked right after setScanBuffersFor, final region label. */
void oracle_extract_features(const float *cloud, size_t n_points, size_t stride_floats, size_t curvature_field,
                             const int32_t *scan_ranges, size_t n_scans, const oracle_reg_params *cfg,
                             float *sharp, float *less_sharp, float *flat, float *less_flat, size_t counts[4],
                             float *curvature_out, int8_t *picked_out, int8_t *label_out);

/* MultiScanRegistration::process (MultiScanRegistration.cpp:94-190), no IMU: in = raw driver cloud
 * {x, y, z, ...}; out = ring-sorted {x, y, z, curvature = ring + relTime} in the registration's
 * swapped axes; ranges = n_rings x {first, last}; returns the number of points kept. */
size_t oracle_multiscan_register(const float *in, size_t n, size_t stride_floats, float lower_deg, float upper_deg,
                                 int n_rings, float scan_period, float *out, int32_t *ranges);

#ifdef __cplusplus
}
#endif
#endif
