/*
 * fmap_oracle.h -- CPU ORACLE for map maintenance: FeatureMap cube grid,
 * addFeatureCloud + per-cube VoxelGrid, active area, surround concatenation.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (see lslam_oracle.h).  Citations and the parity pin
 * status ("parity unpinned" for the PCL VoxelGrid part) are in fmap_oracle.c.
 * Clouds are {x, y, z, intensity} floats.
 */
#ifndef LSLAM_FMAP_ORACLE_H
#define LSLAM_FMAP_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* pcl::VoxelGrid<PointXYZI>::filter with a cubic leaf; out has room for n points; returns count. */
size_t oracle_voxel_grid(const float *in_xyzi, size_t n, size_t stride_floats, float leaf, float *out_xyzi);

typedef struct oracle_fmap oracle_fmap;
oracle_fmap *oracle_fmap_create(int cube_width, int cube_height, int cube_depth); /* FeatureMap.h:55-68 */
void oracle_fmap_free(oracle_fmap *f);
void oracle_fmap_setup_filter_size(oracle_fmap *f, float corner, float surf, float map); /* :72-76 */
void oracle_fmap_setup_cube_size(oracle_fmap *f, float s);                               /* :85-87 */
void oracle_fmap_setup_valid_distance(oracle_fmap *f, float d);                          /* :89-91 */
void oracle_fmap_origin(const oracle_fmap *f, int32_t out[3]);
void oracle_fmap_update(oracle_fmap *f, const float sensor_xyz[3]);                      /* :232-254 */
size_t oracle_fmap_valid_cubes(const oracle_fmap *f, int32_t *out, size_t cap);
/* :218-230; T = row-major 4x4 */
void oracle_fmap_add_feature_cloud(oracle_fmap *f, const float *corner, size_t nc, const float *surf,
                                   size_t ns, size_t stride_floats, const float T[16]);
/* :256-265; which = 0 corner, 1 surf; out may be NULL (count only) */
size_t oracle_fmap_get_surround(const oracle_fmap *f, int which, float *out_xyzi, size_t cap);
size_t oracle_fmap_cube_count(const oracle_fmap *f, int which, int cube);
size_t oracle_fmap_get_full_map(const oracle_fmap *f, float *out_xyzi, size_t cap);      /* :267-286 */

#ifdef __cplusplus
}
#endif
#endif
