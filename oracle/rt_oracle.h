/*
 * rt_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, gcc) of the render hot path of the reference
 * program cpu_launcher.cpp.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; nothing under
 * raytracinggpu_amd/ or include/ links, imports or calls it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 * against golden vectors produced by the reference itself, compiled from
 * /root/reference/cpu_launcher.cpp by oracle/Makefile into oracle/_ref/ and
 * dumped by oracle/make_golden.py into tests/golden/.
 *
 * Every function names the reference lines it follows (cpu_launcher.cpp:N).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct or_mesh or_mesh;
typedef struct or_scene or_scene;

/* per-render work counters (SURVEY 8d ray accounting / algorithmic bytes) */
typedef struct or_counters {
    uint64_t rays;        /* Scene::intersect_all calls                       */
    uint64_t mesh_rays;   /* rays whose root box test passed                  */
    uint64_t box_tests;   /* BoundingBox::intersect calls (root included)     */
    uint64_t nodes;       /* BVH nodes popped from the traversal stack        */
    uint64_t tri_tests;   /* moller_trumbore calls                            */
} or_counters;

typedef struct or_params {
    int32_t W, H;            /* image size (reference hard-codes 512x512, cpu:661) */
    int32_t num_rays;        /* argv[1], samples per pixel (cpu:659)               */
    int32_t num_bounce;      /* argv[2]; CPU convention: b => b+1 segments (cpu:567)*/
    int32_t row_begin, row_end; /* rows [row_begin,row_end) are rendered           */
    float   sigma;           /* Box-Muller jitter sigma (cpu:704 = 0)              */
    float   eps;             /* surface offset epsilon (cpu:575,582,610 = 1e-3)    */
    float   tri_tmin;        /* leaf accept t_cur > tri_tmin (cpu:301 = 1e-4f)     */
    float   fov;             /* alpha, horizontal FOV in radians (cpu:666)         */
    float   cam[3];          /* camera centre C (cpu:691)                          */
    uint32_t seed;           /* counter-RNG seed (DESIGN.md "RNG")                 */
    int32_t threads;         /* OpenMP threads, <=0 => runtime default             */
    int32_t rng_mode;        /* 0: counter RNG (parity with the HIP path)
                                1: std::mt19937(0) stream drawn sequentially, single thread,
                                   row-major -- replays oracle/ref_harness.cpp, i.e. the
                                   reference's own uniform() (cpu:531-536) with clock()==0   */
    int32_t stride;          /* render every stride-th row/column only (<=1: all);
                                the output is then dense ceil(rows/stride) x ceil(W/stride) */
    int32_t tile_rows;       /* with tile_step > 1: only rows of every tile_step-th tile of      */
    int32_t tile_step;       /* tile_rows rows (counted from row_begin) are rendered, densely
                                packed in the output (the interleaved tiling of SURVEY 8e)     */
    int32_t cam_mode;        /* 0: cpu_launcher's fixed camera (cpu:694-699)
                                1: realtime_render.cu's posed camera and per-sample averaging
                                   (KernelLaunch realtime:1100-1134, Camera realtime:803-861);
                                   parity UNPINNED: that program cannot be built here (CUDA + GL)  */
    float   yaw, pitch;      /* cam_mode 1: Camera::yaw / Camera::pitch                          */
} or_params;

/* ---- mesh (TriangleMesh, cpu:167-502) ---- */
or_mesh *or_mesh_new(void);
void     or_mesh_free(or_mesh *m);
/* readOBJ (cpu:315-493). xyz-only vertices get v*scale+offset inside the parser
 * (cpu:353-355 hard-codes 0.8 / (0,-10,0)).  Returns 0, or -1 if the file cannot
 * be opened (the reference prints "Error opening file!" and continues, cpu:322). */
int      or_mesh_read_obj(or_mesh *m, const char *path, float scale, const float offset[3]);
/* replace geometry by explicit arrays (already transformed vertices, OBJ order) */
void     or_mesh_set_arrays(or_mesh *m, const float *verts_xyz, int nv, const int32_t *tri_vidx, int nt);
/* smooth shading (parity unpinned: realtime_render.cu:221-245 get_smooth_normal): vertex normals and the
 * triangles' ni,nj,nk (3 per triangle, in the mesh's current triangle order); NULL switches back to flat shading */
void     or_mesh_set_normals(or_mesh *m, const float *normals_xyz, int n_normals, const int32_t *nidx);
/* TriangleMeshHost::rescale (optimized.cu:297-301) */
void     or_mesh_rescale(or_mesh *m, float scale, const float offset[3]);
/* the `transform` kernel of global_launcher.cu:340-365 on the vertices (rotation matrix row-major, then translation) */
void     or_mesh_transform(or_mesh *m, const float rotation[9], const float translation[3]);
/* keep the BVH's topology and triangle order, recompute every node's box (compute_bbox, cpu:180-188) */
void     or_mesh_refit(or_mesh *m);
/* buildBVH(&bvh, 0, indices.size()) (cpu:190-224) */
void     or_mesh_build_bvh(or_mesh *m);
/* a caller-supplied tree in the flat layout of bvhTreeToArray (10 floats per node, node 0 = root) + the triangle order its ranges refer to
 * (order[k] = current index of the triangle that moves to position k; NULL = keep): the traversal then walks THAT tree.  0 / -1 (malformed) */
int      or_mesh_set_bvh(or_mesh *m, const float *arr10, int n_nodes, const int32_t *order);
int      or_mesh_num_vertices(const or_mesh *m);
int      or_mesh_num_triangles(const or_mesh *m);
int      or_mesh_num_nodes(const or_mesh *m);
int      or_mesh_max_depth(const or_mesh *m);
void     or_mesh_get_vertices(const or_mesh *m, float *out_xyz);
void     or_mesh_get_triangles(const or_mesh *m, int32_t *out_vidx);
/* bvhTreeToArray layout (optimized.cu:512-534): 10 floats per node */
void     or_mesh_bvh_to_array(const or_mesh *m, float *out_arr10);
void     or_mesh_set_albedo(or_mesh *m, float r, float g, float b);
/* Geometry::mirror / in_refraction_index / out_refraction_index of the mesh (cpu:113-116; Geometry() cpu:110 = 0 / 1 / 1): Scene::getColor
 * branches on them for whichever object was hit (cpu:573-606).  Set before or_scene_add_mesh. */
void     or_mesh_set_material(or_mesh *m, int mirror, float n_in, float n_out);
/* TriangleMesh::intersect (cpu:238-313, ENABLE_BVH branch). returns hit flag */
int      or_mesh_intersect(const or_mesh *m, const float O[3], const float u[3], float tri_tmin,
                           float *t, float N[3], or_counters *cnt);

/* ---- primitives ---- */
/* Sphere::intersect (cpu:512-527) */
int or_sphere_intersect(const float C[3], float R, const float O[3], const float u[3], float *t, float N[3]);
/* BoundingBox::intersect (cpu:146-157) */
int or_box_intersect(const float mn[3], const float mx[3], const float O[3], const float u[3]);
/* TriangleMesh::moller_trumbore (cpu:226-236) */
int or_moller_trumbore(const float A[3], const float B[3], const float C[3],
                       const float O[3], const float u[3], float *t, float N[3]);

/* ---- scene (Scene, cpu:538-652) ---- */
or_scene *or_scene_new(void);
void      or_scene_free(or_scene *s);
/* addObject(new Sphere(C,R,albedo,mirror,n_in,n_out)) (cpu:507,540); returns id */
int  or_scene_add_sphere(or_scene *s, const float C[3], float R, const float albedo[3],
                         int mirror, float n_in, float n_out);
/* addObject(mesh) ; the scene borrows the mesh */
int  or_scene_add_mesh(or_scene *s, or_mesh *m);
void or_scene_set_light(or_scene *s, const float L[3], float intensity);
/* intersect_all (cpu:545-564) */
int  or_scene_intersect_all(const or_scene *s, const float O[3], const float u[3], float tri_tmin,
                            float P[3], float N[3], int *object_id, or_counters *cnt);
/* getColor (cpu:566-648) for one camera ray of pixel/sample; rng is the counter RNG */
void or_scene_get_color(const or_scene *s, const float O[3], const float u[3], int ray_depth,
                        float eps, float tri_tmin, uint32_t seed, uint32_t pixel, uint32_t sample,
                        float out_rgb[3], or_counters *cnt);

/* counter RNG, uniform in (0,1] (DESIGN.md "RNG"); replaces uniform() cpu:531-536 */
float or_uniform(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t dim);

/* pixel loop (cpu:693-718).  out_rgba: (row_end-row_begin)*W*4 floats, linear colour
 * average in .xyz and the number of rays traced for that pixel in .w.
 * out_rgb8 (may be NULL): gamma 1/2.2 + min(.,255) + truncation (cpu:714-716). */
int or_render(const or_scene *s, const or_params *p, float *out_rgba, uint8_t *out_rgb8,
              or_counters *cnt);
/* the same tonemap applied to an existing float framebuffer (cpu:714-716) */
void or_tonemap(const float *rgba, int npix, uint8_t *out_rgb8);

int or_max_threads(void);

/* ---- realtime_render.cu pieces (parity unpinned, see cam_mode) ---- */
/* Camera::rotate() (realtime:823-846): orthonormal basis from yaw and pitch */
void or_camera_basis(float yaw, float pitch, float bx[3], float by[3], float bz[3]);
/* WangHash (realtime:1190-1197): the per-frame RNG seed */
uint32_t or_wang_hash(uint32_t a);
/* accumbuffer += frame; display = accumbuffer / framenumber (cutil_math: a * (1.0f / s)); 8-bit image =
 * (unsigned char)min(powf(c, 1 / 2.2f), 255.) (realtime:1136-1147).  accum/frame/display: npix * 4 floats */
void or_progressive_accumulate(float *accum, const float *frame, int npix, int framenumber, float *display, uint8_t *out_rgb8);

#ifdef __cplusplus
}
#endif
#endif
