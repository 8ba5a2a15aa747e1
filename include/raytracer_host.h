/*
 * raytracer_host.h -- C exports of libraytrace_host.so: the build-time half of the reference's host
 * API (TriangleMeshHost::readOBJ / rescale / buildBVH / bvhTreeToArray, optimized.cu:293-535, and the
 * PNG writer, cpu_launcher.cpp:719) for callers that are not C++ (tests, bench.py).  C++ callers use
 * include/raytracer.hpp directly.  No GPU code here; the render path is raytrace_hip.h.
 */
#ifndef RAYTRACER_HOST_H
#define RAYTRACER_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_mesh rth_mesh;

rth_mesh *rth_mesh_new(void);
void rth_mesh_free(rth_mesh *m);
/* readOBJ with the in-parser transform v*scale+offset; returns 0, or -1 when the file is missing
 * (mesh stays empty, cpu_launcher.cpp:322-325) */
int rth_mesh_read_obj(rth_mesh *m, const char *path, float scale, const float offset[3]);
/* geometry from arrays: nv*3 floats, nt*3 vertex indices (OBJ face order) */
void rth_mesh_set_arrays(rth_mesh *m, const float *verts_xyz, int nv, const int32_t *tri_vidx, int nt);
void rth_mesh_rescale(rth_mesh *m, float scale, const float offset[3]);   /* optimized.cu:297-301 */
/* buildBVH over all triangles + bvhTreeToArray; returns the node count */
int rth_mesh_build_bvh(rth_mesh *m);
int rth_mesh_num_vertices(const rth_mesh *m);
int rth_mesh_num_triangles(const rth_mesh *m);
int rth_mesh_num_nodes(const rth_mesh *m);
void rth_mesh_get_vertices(const rth_mesh *m, float *out_xyz);            /* nv*3 */
void rth_mesh_get_indices(const rth_mesh *m, int32_t *out_tri10);         /* nt*10, TriangleIndices layout */
void rth_mesh_get_bvh_array(const rth_mesh *m, float *out_arr10);         /* n_nodes*10 */
/* 8-bit RGB PNG; returns 0 on success */
int rth_write_png(const char *path, int W, int H, const uint8_t *rgb);

#ifdef __cplusplus
}
#endif
#endif
