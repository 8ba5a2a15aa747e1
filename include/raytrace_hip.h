/*
 * raytrace_hip.h -- C-ABI of libraytrace_hip.so: the MI355X (gfx950) render path.
 *
 * Drop-in boundary for the per-pixel render of souhhcong/RaytracingGPU.  The
 * reference has no library interface: its render path is the CUDA sequence in
 * optimized.cu main() --
 *     cudaMalloc/cudaMemcpy of arr_bvh, indices, vertices      (optimized.cu:811-826)
 *     KernelLaunch<<<H*W/128,128,smem>>>(d_colors, W, H, num_rays, num_bounce,
 *                  d_indices, ni, d_vertices, nv, d_arr_bvh)   (optimized.cu:828-847)
 *     cudaDeviceSynchronize + cudaMemcpy D2H of the image      (optimized.cu:849-856)
 * -- and, on the CPU, the pixel loop of cpu_launcher.cpp:693-718.  Each entry point
 * below names the reference lines it replaces.  Plain pointers and sizes only; no
 * HIP, torch or C++ types cross this boundary.  Every function returns RT_OK (0) or
 * a negative rt_status, never exits and never throws (the reference's gpuErrchk
 * prints and calls exit(), optimized.cu:24-30).
 *
 * Result contract: for sigma == 0 the linear float colour written by rt_render* is
 * bit-identical to cpu_launcher.cpp's Scene::getColor average (color_avg, cpu:713)
 * when the reference's uniform() is replaced by the counter RNG of DESIGN.md; the
 * 8-bit image is cpu:714-716.  There is no CPU fallback in this library.
 */
#ifndef RAYTRACE_HIP_H
#define RAYTRACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 6
#define RT_MAX_SPHERES 16      /* reference: Geometry* objects[10], optimized.cu:663 */
#define RT_MAX_OBJECTS 16      /* spheres + meshes of one scene (Scene::objects, cpu:538-543)                     */
#define RT_MAX_SEGMENTS 16     /* reference: MAX_RAY_DEPTH 10, optimized.cu:22       */

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,       /* bad argument (message in rt_last_error)            */
    RT_ERR_NO_DEVICE = -2,     /* no gfx950 device / HIP runtime unusable            */
    RT_ERR_HIP = -3,           /* a HIP call failed (message in rt_last_error)       */
    RT_ERR_NO_SCENE = -4,      /* render before rt_scene_upload                      */
    RT_ERR_UNSUPPORTED = -5,   /* e.g. LDS variant that does not fit the mesh        */
    RT_ERR_INTERNAL = -6       /* an invariant of the library did not hold (a bug)   */
} rt_status;

/* kernel variants (BASELINE.json configs 3/4); all produce bit-identical results */
typedef enum rt_variant {
    RT_VARIANT_AUTO = 0,       /* the fastest measured variant: RT_VARIANT_WAVEFRONT_QUEUE; for a scene
                                  without a mesh (and no pose / smooth normals): RT_VARIANT_LOCKSTEP   */
    RT_VARIANT_GLOBAL = 1,     /* persistent lanes (micro-op scheduler); SoA nodes + packed
                                  triangles read from HBM through L2/L1                             */
    RT_VARIANT_LDS_VERTS = 2,  /* work-stack traversal (as 8) with the vertex array staged in LDS by a
                                  cooperative copy per workgroup (different-versions/
                                  optimized_vertices-in-shared.cu:681-686): a triangle test reads three
                                  indices + three LDS vertices; one workgroup per CU; RT_ERR_UNSUPPORTED
                                  when the vertices leave no room for a wave's carve in the 160 KB    */
    RT_VARIANT_LDS_TOP = 3,    /* work-stack traversal with the top BVH levels (breadth-first prefix of
                                  the node array, all of it if it fits) staged in LDS                */
    RT_VARIANT_LDS_ALL = 4,    /* both: vertices + as much of the top of the BVH as fits beside them  */
    RT_VARIANT_LOCKSTEP = 5,   /* one lane bound to one pixel for the whole frame, lock-step ray
                                  queries: the structure of KernelLaunch (optimized.cu:670-772);
                                  kept as the baseline the other variants are measured against      */
    RT_VARIANT_WAVEFRONT = 6,  /* uniform per-pixel shade/generate kernels alternating with a lean
                                  persistent traversal kernel; path state as float4 SoA in HBM;
                                  BVH nodes and triangles read from HBM through L2/L1               */
    RT_VARIANT_WAVEFRONT_LDS = 7, /* the same with every BVH node staged in LDS (one 1024-thread
                                  workgroup per CU shares the copy); RT_ERR_UNSUPPORTED when the
                                  nodes do not fit the 160 KB                                       */
    RT_VARIANT_WAVEFRONT_QUEUE = 8, /* wavefront pipeline whose traversal kernel keeps a per-wave LDS
                                  work stack of (ray slot, sibling pair) entries: every lane tests the two
                                  boxes of a pair, or two triangles, per step; no per-lane walk
                                  (rt_travq.hip.h).  The default (RT_VARIANT_AUTO)                  */
    RT_VARIANT_PATH = 9        /* the whole render in ONE persistent launch: a wave owns 64 paths (one per lane) from
                                  camera ray to framebuffer store; the work-stack traversal and the
                                  shading of ready paths alternate inside the wave, path state lives
                                  in LDS, nothing but the pixel leaves the CU (rt_path.hip.h)       */
} rt_variant;

typedef struct rt_ctx rt_ctx;

/* Sphere(C, R, albedo, mirror, n_in, n_out): cpu_launcher.cpp:505-511, Geometry cpu:106-118 */
typedef struct rt_sphere {
    float   center[3];
    float   radius;
    float   albedo[3];
    int32_t mirror;
    float   in_refraction_index;
    float   out_refraction_index;
} rt_sphere;

/* The mesh exactly as optimized.cu hands it to KernelLaunch (optimized.cu:670, 811-826):
 * Vector vertices[nv] (3 x f32), TriangleIndices indices[nt] in BVH order (only
 * vtxi,vtxj,vtxk are read, optimized.cu:271) and the bvhTreeToArray float[10] node array
 * (optimized.cu:512-534: [0]=left [1]=right(-1 leaf) [2..4]=mn [5..7]=mx [8]=tri_start [9]=tri_end). */
typedef struct rt_mesh {
    const float   *vertices;       /* n_vertices * 3                                          */
    int32_t        n_vertices;
    const int32_t *indices;        /* vtxi,vtxj,vtxk of triangle t at indices[t*index_stride] */
    int32_t        index_stride;   /* 3 = compact, 10 = sizeof(TriangleIndices)/4             */
    int32_t        n_triangles;
    const float   *bvh_arr10;      /* n_nodes * 10                                            */
    int32_t        n_nodes;
    float          albedo[3];      /* mesh_ptr->albedo, cpu:683                               */
    int32_t        object_slot;    /* position in Scene::objects (cpu_launcher: 6 = last,
                                      optimized.cu:690-700: 1); decides exact-tie order cpu:554 */
    /* ABI 6: the rest of Geometry (cpu:106-118), which a TriangleMesh inherits like a Sphere does and Scene::getColor reads
     * for WHICHEVER object was hit (cpu:573 objects[id]->mirror, cpu:580 the two indices).  A zero-initialised rt_mesh
     * (0 / 0 / 0) is the diffuse mesh of Geometry() (cpu:110: mirror 0, indices 1 / 1): equal indices take the diffuse branch. */
    int32_t        mirror;
    float          in_refraction_index;
    float          out_refraction_index;
} rt_mesh;

/* Scene::L / Scene::intensity (cpu:650-651), camera C and alpha (cpu:666,691) */
typedef struct rt_light  { float position[3]; float intensity; } rt_light;
typedef struct rt_camera { float position[3]; float fov; } rt_camera;

typedef struct rt_params {
    int32_t  width, height;        /* W,H (reference: 512, cpu:661-662)                       */
    int32_t  num_rays;             /* argv[1], cpu:659                                        */
    int32_t  num_bounce;           /* argv[2], cpu:659                                        */
    int32_t  depth_convention;     /* 0: cpu_launcher, b => b+1 segments (cpu:567)
                                      1: optimized.cu, b => b segments (optimized.cu:566)     */
    float    sigma;                /* anti-aliasing jitter: 0 (cpu:704) / 0.2 (optimized.cu:753) */
    float    eps;                  /* 1e-3 (cpu:575) / 1e-4 (optimized.cu:575)                */
    float    tri_tmin;             /* 1e-4f (cpu:301) / 0 (optimized.cu:275)                  */
    uint32_t seed;                 /* counter RNG seed; optimized.cu:745 uses 123456          */
    int32_t  variant;              /* rt_variant                                              */
} rt_params;

/* Which rows a call renders.  Local row r (0 <= r < n_rows) is image row
 *     row0 + (r / tile_rows) * tile_rows * tile_step + (r % tile_rows).
 * Contiguous range [a,b): {a, b-a, b-a, 1}.  Interleaved tiles of rank k of G
 * (SURVEY 8e): {k*R, n_local_rows, R, G}. */
typedef struct rt_rows {
    int32_t row0;
    int32_t n_rows;
    int32_t tile_rows;
    int32_t tile_step;
} rt_rows;

typedef struct rt_stats {
    float    kernel_ms;            /* HIP-event time of the last render kernel                */
    float    tonemap_ms;           /* of the last tonemap kernel (0 if none)                  */
    uint64_t pixels;               /* pixels of the last render call                          */
    int32_t  variant;              /* variant that actually ran                               */
    int32_t  lds_bytes;            /* dynamic + static LDS per workgroup                      */
    int32_t  block_threads;
    int32_t  grid_blocks;
    float    trav_ms;              /* wavefront variant: summed HIP-event time of the traversal kernel
                                      launches of the last sample of the last frame, and how many    */
    int32_t  trav_launches;
    int32_t  parts;                /* wavefront variant: concurrent sub-frames the call was cut into      */
    int32_t  adv_launches;         /* ... and the same for the uniform kernel (wf_advance, the launches after the first): */
    float    adv_ms;               /*     summed HIP-event time, launches, paths per launch (rt_stats_enable)             */
    int32_t  adv_paths;
    int32_t  travq_mode;           /* work-stack traversal kernel of the last render: 0 = sibling pairs (64-byte float nodes), 1 = 16-bit fixed-point pairs,
                                      2 = 4-wide fixed-point nodes (the default wherever the format fits: boxes nest, leaves of 1 .. 127 triangles, fewer than 2^21 nodes);
                                      -1 = another traversal kernel / no mesh */
    int32_t  reserved;
} rt_stats;

/* --- device / context -------------------------------------------------------- */
/* replaces the implicit CUDA device 0 of optimized.cu */
int rt_abi_version(void);
int rt_device_count(int *count);
int rt_ctx_create(rt_ctx **ctx, int device_id);
int rt_ctx_destroy(rt_ctx *ctx);
const char *rt_last_error(const rt_ctx *ctx);          /* ctx may be NULL: last global error */
int rt_device_name(const rt_ctx *ctx, char *buf, size_t buflen);

/* --- scene upload: replaces optimized.cu:811-826 (H2D of arr_bvh/indices/vertices)
 *     and the in-kernel scene construction optimized.cu:679-726 --------------------
 * All host arrays are copied; the caller keeps ownership.  mesh may be NULL
 * (spheres only: what the reference renders when the OBJ is missing, cpu:322-325). */
int rt_scene_upload(rt_ctx *ctx, const rt_sphere *spheres, int n_spheres, const rt_mesh *mesh,
                    const rt_light *light, const rt_camera *camera);
/* ... with any number of meshes (ABI 6): Scene::objects is a std::vector<Geometry*> scanned in insertion order with a strict '<'
 * (cpu:538-564; optimized.cu:663 `Geometry* objects[10]`), so a scene may hold several TriangleMesh objects at any positions.
 * meshes[k].object_slot are distinct positions in [0, n_spheres + n_meshes); the spheres fill the remaining positions in array
 * order.  At most RT_MAX_OBJECTS objects.  Each mesh keeps the tree its own buildBVH made (cpu:190-224) and its own root-box test
 * (cpu:279); inside the library the trees hang below synthetic nodes whose boxes are the unions of their children, the triangles
 * are stored mesh after mesh in object order, and one traversal finds the minimum over (t, object position, scan rank) -- the
 * result of the reference's loop over the objects.  A mesh without triangles stays an object that is never hit (missing OBJ,
 * cpu:322-325).  Every kernel variant renders such scenes.  With more than one mesh that has triangles the per-mesh operations -- rt_mesh_set_normals, rt_mesh_rebuild* --
 * are refused (RT_ERR_UNSUPPORTED); rt_mesh_transform moves them all. */
int rt_scene_upload_meshes(rt_ctx *ctx, const rt_sphere *spheres, int n_spheres, const rt_mesh *meshes, int n_meshes,
                           const rt_light *light, const rt_camera *camera);

/* --- render: replaces KernelLaunch + cudaDeviceSynchronize + D2H, optimized.cu:828-856,
 *     i.e. the pixel loop cpu:693-713.  Output: n_rows*width float4, .xyz = linear
 *     colour average (color_avg, cpu:713), .w = rays traced for the pixel. ------- */
int rt_render(rt_ctx *ctx, const rt_params *p, int row_begin, int row_end, float *out_rgba_host);
/* same, into device memory, asynchronous on `stream` (a hipStream_t, NULL = the
 * context's own stream); rows may be interleaved tiles (multi-GPU) */
int rt_render_device(rt_ctx *ctx, const rt_params *p, const rt_rows *rows, void *out_rgba_dev, void *stream);

/* --- a BATCH of frames in one launch chain (ABI 6).  A process that renders a small share of every frame -- one rank of eight on 1920x1080 owns
 *     0.26 Mpixel -- cannot fill the chip with the eleven dependent launches of ONE such frame, and a frame per stream runs out of hardware queues
 *     (four per process).  Here n_frames (<= RT_MAX_BATCH) frames of the same size and rows are traced as the items of ONE chain -- the machinery
 *     that traces the samples of a pixel as parallel items -- each with its OWN camera (position, fov: cpu:666, 691-699), its own seed and its own
 *     output buffer: a sequence of frames of a moving camera / a progressive render, not one frame repeated.  Frame k's buffer holds exactly what
 *     rt_render_device writes for the scene with that camera and p->seed = frames[k].seed (bit for bit: per-pixel arithmetic does not depend on
 *     what else is in the launch).  num_rays == 1; wavefront variants; p->seed is ignored; asynchronous on `stream`.  Throughput, not latency: the
 *     n frames finish together.  Replaces n x (KernelLaunch + sync, optimized.cu:828-849). */
#define RT_MAX_BATCH 16
typedef struct rt_frame_desc {
    rt_camera camera;              /* this frame's camera (Camera C / alpha, cpu:666,691)      */
    uint32_t  seed;                /* this frame's counter-RNG seed (rt_params.seed)           */
    uint32_t  reserved;
    void     *out_rgba_dev;        /* n_rows * width float4, as rt_render_device               */
} rt_frame_desc;
int rt_render_device_batch(rt_ctx *ctx, const rt_params *p, const rt_rows *rows, const rt_frame_desc *frames, int n_frames, void *stream);

/* --- tonemap: cpu:714-716 (gamma 1/2.2 in binary64, min 255, truncate) ------- */
int rt_tonemap_device(rt_ctx *ctx, const void *rgba_dev, int64_t n_pixels, void *rgb8_dev, void *stream);
/* render + tonemap + D2H of the interleaved RGB8 image (what stbi_write_png gets, cpu:719) */
int rt_render_rgb8(rt_ctx *ctx, const rt_params *p, int row_begin, int row_end, uint8_t *out_rgb8_host);

/* --- work accounting: renders rows [row_begin,row_end) with the counting instantiation of
 *     the kernel and returns the traversal work; 1 ray = 1 Scene::intersect_all call (cpu:545).  The
 *     algorithmic bytes of the roofline are 24 B*box_tests + 16 B*nodes + 48 B*tri_tests + 16 B*pixels. */
typedef struct rt_work {
    uint64_t rays;                 /* intersect_all calls (primary + shadow + bounce)         */
    uint64_t box_tests;            /* BoundingBox::intersect calls, root included             */
    uint64_t nodes;                /* BVH nodes whose box was hit (= nodes the reference pops) */
    uint64_t tri_tests;            /* moller_trumbore calls                                   */
    uint64_t box_literal;          /* of box_tests: decided by the literal divisions of cpu:147-152 (the error-bounded filter deferred) */
    uint64_t tri_literal;          /* of tri_tests: barycentrics by the literal divisions of cpu:232-233                                */
    uint64_t steps[12];            /* work-stack traversal kernel, summed over its waves and launches: loop iterations, refill passes,
                                    * refill rounds, queue fetches, TRI steps (128 triangle tests), BOX steps (64 sibling pairs),
                                    * literal-box fall-backs, serial drains; then blocks a step enters only when some lane needs them:
                                    * t-division blocks of the triangle tests (2 per TRI step at most), first and second leaf-queue
                                    * push of a BOX step; TRI steps in which a shadow ray stopped at a hit that certainly shades (any-hit; 0 for the
                                    * binary instantiation, which never stops early).  bench.py prices the vector-issue roofline with them. */
} rt_work;
/* The counters describe the REFERENCE-EQUIVALENT traversal (the binary instantiation of the kernel: every box the reference tests, cpu:284-293), whatever
 * kernel produces the frames: with the 16-bit fixed-point pairs (RT_TRAVQ_Q16) or the 4-wide BOX step
 * (RT_TRAVQ_QW) the production kernel enters a superset of the internal nodes and skips levels, stops a shadow ray at the first accepted triangle that certainly
 * shades and does not trace a shadow ray that a sphere shades already (any-hit, RT_TRAVQ_ANYHIT=0 turns both off: cpu:615 is monotone in the nearest hit's t, so the
 * frame is the same bit for bit), and its own visits are not what box_tests / nodes / tri_tests
 * report -- unless RT_TRAVQ_QW_COUNT=1 asks for the 4-wide kernel's own counting instantiation (experiments).
 * With several meshes (rt_scene_upload_meshes) the tree in use holds synthetic union nodes above the meshes' roots: box_tests / nodes count those too (one test of the
 * forest's root where the reference tests every mesh's root, cpu:279), tri_tests are the reference's. */
int rt_count_work(rt_ctx *ctx, const rt_params *p, int row_begin, int row_end, rt_work *out);

int rt_synchronize(rt_ctx *ctx);
int rt_get_stats(rt_ctx *ctx, rt_stats *stats);        /* waits for the last render to finish */
/* trav_ms / trav_launches are measured only on request (on != 0): production frames record no per-launch timing events */
int rt_stats_enable(rt_ctx *ctx, int on);
/* Frames in flight for a caller that renders frame after frame on ONE stream into ALTERNATING device buffers (rt_render_device,
 * rt_render_pose_device).  A frame is rendered as two sub-frames on two internal streams; normally both are forked from the caller's
 * stream when the call is made and joined back into it, so frame k+1 starts when ALL of frame k has finished and one internal stream
 * idles at every frame boundary (~5 % of a 1080p frame).  With pipelining on, a frame whose output buffer does not overlap the previous
 * frame's is ordered behind what was on the caller's stream when the PREVIOUS render call was made, and each of its sub-frames follows
 * the same sub-frame of the previous frame directly.  What the caller gives up: work submitted to the stream BETWEEN two render calls
 * is not waited for by the second call's kernels (work submitted before the first of the two is; consumers of a frame submitted after
 * its call still see it complete, the join into the caller's stream stays).  So: alternate between two (or more) output buffers, and
 * do not let anything the frame depends on -- a fill of its buffer, a wait for a reader on another stream -- be younger than the previous
 * call.  Same stream, same size and parameters' layout, else the frame falls back to the full fork (results never change, only overlap).
 * rt_render_async does this internally for its two slots (RT_ASYNC_PIPELINE=0 turns that off).  Off by default.
 * What the library can check it does: work it put on the stream itself since the previous render call (rt_tonemap_device) is remembered
 * with the buffers it reads and writes; a frame that would render into one of them takes the full fork instead (product build: results
 * and ordering stay right, only the overlap is lost) and is REFUSED with RT_ERR_INVALID by a -DRT_DEBUG build
 * (libraytrace_hip_debug.so), so that a test run shows the sequence breaks the rule.  Work the caller submits through HIP directly is
 * invisible to any library: that part of the rule stays the caller's. */
int rt_ctx_set_pipelining(rt_ctx *ctx, int on);

/* --- pipelined frames for a host caller.  optimized.cu renders, synchronises and then copies (optimized.cu:849-856), so the
 *     33 MB of a 1080p float frame cross PCIe strictly after the kernels.  rt_render_async renders the whole frame into one of
 *     two device buffers (slot 0 / 1) and starts its device-to-host copy on a separate copy stream; rt_wait(slot) blocks until that
 *     slot's frame is in out_host.  Calling rt_render_async(slot ^ 1) before rt_wait(slot) overlaps frame k's copy with frame
 *     k+1's kernels.  out_host: width*height float4 (rgb8 == 0) or width*height*3 bytes (rgb8 != 0: the tonemapped image of
 *     cpu:714-716); memory from rt_host_alloc makes the copy one DMA.  Re-using a slot whose frame has not been waited for is
 *     allowed: its kernels wait for the pending copy. */
int rt_render_async(rt_ctx *ctx, const rt_params *p, int slot, void *out_host, int rgb8);
int rt_wait(rt_ctx *ctx, int slot);
/* every device buffer of the context lives on the context's device (RT_ERR_INTERNAL otherwise): a context used from a thread
 * whose current device is another GPU must not allocate there (the CUDA programs of the reference only ever see device 0) */
int rt_ctx_selfcheck(rt_ctx *ctx);

/* --- pinned host memory for frame buffers.  optimized.cu copies its image into pageable memory (`new char[]`,
 *     optimized.cu:851-856); a buffer from rt_host_alloc lets the D2H copy of rt_render / rt_render_rgb8 /
 *     rt_render_multi run as one DMA at the PCIe rate instead of being staged by the runtime. ------------------ */
int rt_host_alloc(void **ptr, size_t bytes);
int rt_host_free(void *ptr);
/* device memory on the context's device for rt_render_device / rt_tonemap_device callers that have no HIP of their own
 * (the C++ host API stays free of hip_runtime.h) */
int rt_device_alloc(rt_ctx *ctx, void **ptr, size_t bytes);
int rt_device_free(void *ptr);
int rt_device_to_host(rt_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);   /* synchronous copy on the context's stream */

/* --- a batch of explicit rays through the PRODUCTION traversal launches.  TriangleMesh::intersect (cpu_launcher.cpp:238-313,
 *     optimized.cu:220-285) is callable with any ray; this entry writes the caller's rays into the traversal queue exactly as the
 *     render path's emitter does (root-box test of cpu:279 included) and runs the same kernel instantiation with the launch geometry
 *     of a frame: variant RT_VARIANT_WAVEFRONT_QUEUE (or AUTO: wf_travq, the default traversal), RT_VARIANT_WAVEFRONT (wf_trav) or
 *     RT_VARIANT_PATH (wf_path).  rays: n x 6 floats (O.xyz, u.xyz; u is used as given, the reference does not renormalise either);
 *     out: n x 5 floats (hit 0 / 1, t, N.xyz normalised as cpu:308; a miss leaves t = 1e9 = INF narrowed, cpu:283).
 *     tri_tmin: the leaf loop's t > tri_tmin (cpu:301: 1e-4; 0 = moller_trumbore's own t > 0). */
int rt_trace_rays(rt_ctx *ctx, const float *rays, int n, float tri_tmin, int variant, float *out);

/* --- known-answer entry points (test interface; same library, same device functions the render kernels inline).
 *     One lane per row.  Inputs/outputs use the layouts of tests/golden/kat.npz, which the reference's own functions
 *     produced (oracle/ref_harness.cpp):
 *       sphere   in C[3] R O[3] u[3]          out hit t N[3]      Sphere::intersect, cpu:512-527
 *       box      in mn[3] mx[3] O[3] u[3]     out hit             BoundingBox::intersect, cpu:146-157
 *                route 0 literal divisions, 1 slab_filtered (root-box pre-test, stackless walks), 2 qbox_filter + literal
 *                fall-back (work-stack kernels)
 *       triangle in A[3] B[3] C[3] O[3] u[3]  out hit t N[3]      moller_trumbore, cpu:226-236 (N = e1 x e2, unnormalised)
 *       mesh     in O[3] u[3]                 out hit t N[3]      TriangleMesh::intersect, cpu:238-313, on the uploaded mesh
 *                route 0 the work-stack kernels' primitives, 1 the stackless mesh_intersect of the lock-step kernels
 *     counts: how many tests the error-bounded filters decided and how many fell back to the literal divisions. --- */
typedef struct rt_kat_counts {
    uint64_t n;
    uint64_t box_decided, box_literal;
    uint64_t tri_decided, tri_literal;
} rt_kat_counts;
int rt_kat_sphere(rt_ctx *ctx, const float *in, int n, float *out);
int rt_kat_sqrt(rt_ctx *ctx, const float *in, int n, float *out);   /* out[i] = the device's correctly rounded square root (every sqrt of cpu_launcher.cpp: Vector::norm, Sphere::intersect, getColor) */
int rt_kat_box(rt_ctx *ctx, const float *in, int n, int route, float *out, rt_kat_counts *counts);
int rt_kat_triangle(rt_ctx *ctx, const float *in, int n, float *out, rt_kat_counts *counts);
int rt_kat_mesh(rt_ctx *ctx, const float *in, int n, float tri_tmin, int route, float *out, rt_kat_counts *counts);
/* 64-bit hashes of the device-side node layouts the upload (or a refit / rebuild) derived for the traversal kernels: [0] float sibling pairs, [1] 16-bit fixed-point
 * pairs, [2] 4-wide quads (which four nodes a quad holds is chosen by a surface-area DP on the device), [3] leaf boxes by triangle; 0 = not in use.  Two uploads of one
 * tree give equal hashes: the layouts -- and with them the work a frame does -- are a function of the tree alone. */
int rt_kat_layout_hash(rt_ctx *ctx, uint64_t out[4]);

/* --- device-side mesh transform: the `transform` kernel of global_launcher.cu:340-365 / transformMesh
 *     (realtime_render.cu:1151-1166) applied to the uploaded vertices -- v' = R v (row-major 3x3), then += translation --
 *     followed, on the device, by the triangle precompute and a REFIT of the BVH: same tree, same triangle order, every
 *     node's box recomputed as compute_bbox (cpu_launcher.cpp:180-188) of its range.  (The reference never refits: to get
 *     the tree buildBVH would build for the moved mesh, rebuild on the host and call rt_scene_upload.) ------------------ */
int rt_mesh_transform(rt_ctx *ctx, const float rotation[9], const float translation[3]);

/* --- device-side BVH BUILD: TriangleMesh::buildBVH (cpu_launcher.cpp:190-224; the reference's device twin is the
 *     one-thread recursive buildBVH of global_launcher.cu:298-331, launched by KernelInit :848-881) over the uploaded triangles
 *     with the vertices as they are on the device now (i.e. after rt_mesh_transform): level by level, one workgroup per node --
 *     box of the range, longest axis, midpoint split, the reference's in-place partition, its stop rule -- then the numbering
 *     of bvhTreeToArray (optimized.cu:512-534).  The tree equals the host builder's bit for bit: boxes, node indices, and the
 *     order the partition leaves the triangles in.  The library then re-lays the mesh out for its kernels as rt_scene_upload
 *     does; the uploaded order becomes the new BVH order (the reference partitions `indices` in place too).
 *     Optional outputs: bvh_arr10_out (capacity (2 * n_triangles + 2) * 10 floats), tri_order_out[n_triangles] (position ->
 *     index of the triangle in the order before this call), n_nodes_out.
 *     Failure: an error during the BUILD leaves the scene in use untouched.  An allocation failure during the re-layout that
 *     follows (RT_ERR_HIP: out of device memory) leaves the context WITHOUT a scene (RT_ERR_NO_SCENE from the render calls):
 *     upload again.  Cost model: one workgroup per node and one blocking read-back per level, so the top levels of a mesh far
 *     larger than the cat's 3 954 triangles run on a single CU each (the build is a step before the hot path, not part of it). */
int rt_mesh_rebuild(rt_ctx *ctx, float *bvh_arr10_out, int32_t *tri_order_out, int32_t *n_nodes_out);   /* = rt_mesh_rebuild_mode(RT_BVH_REFERENCE) */

/* --- ... or a DIFFERENT, better tree, built in parallel (SURVEY 8f3 "and a GPU LBVH build"; opt-in, RT_BVH_REFERENCE stays the default
 *     everywhere).  The reference's stop rule (cpu_launcher.cpp:217) leaves leaves that grow with the mesh -- 2 M triangles: 357 triangle
 *     tests per ray -- and its device builder is ONE thread (global_launcher.cu:298-331, :848-881).  RT_BVH_LBVH: Morton codes of the
 *     triangle centroids (63 bits), radix sort, the binary radix tree over the sorted codes with one thread per node (Karras 2012),
 *     boxes bottom-up by min / max of the vertex coordinates (the values compute_bbox, cpu:180-188, folds for the node's range), and for
 *     every node the surface-area heuristic's choice between ONE leaf (at most 32 triangles) and its subtree -- the reference's traversal
 *     never prunes by distance, so a ray pays for every box it pierces and every triangle of every leaf it enters: the cost the SAH models.  Same outputs in the same formats -- the flat `float[10]` tree of bvhTreeToArray and the order the
 *     triangles are left in -- so every kernel variant runs on it unchanged and a CPU checker can be handed the very same tree
 *     (oracle: or_mesh_set_bvh).  What changes against the reference tree: WHICH triangles a ray tests (far fewer), never what a
 *     test returns; the image differs from the reference tree's only where two triangles are hit at bit-equal t (shared edges: the
 *     scan order breaks the tie, cpu:301 / SURVEY H5).  A mesh of at most four triangles is a single leaf in either mode (the reference builder runs).
 *     rt_mesh_build_stats: what the last rebuild did (device_build_ms = the builder's kernels, HIP events; install_ms = the
 *     re-layout for the render kernels that follows, host side). */
typedef enum rt_bvh_mode { RT_BVH_REFERENCE = 0, RT_BVH_LBVH = 1 } rt_bvh_mode;
typedef struct rt_build_stats {
    int32_t mode;                  /* the mode that ran                                        */
    int32_t n_triangles, n_nodes;
    int32_t n_leaves, max_leaf_tris, max_depth;   /* RT_BVH_LBVH only (0 otherwise)            */
    float   device_build_ms;       /* builder kernels + sort, HIP events on the context's stream */
    float   install_ms;            /* the kernels' formats: on the device for RT_BVH_LBVH (closed forms over the builder's arrays), else
                                      read-back + re-layout on the host + upload; wall clock                                    */
    int32_t install_on_device;
    int32_t reserved;
} rt_build_stats;
int rt_mesh_rebuild_mode(rt_ctx *ctx, int mode, float *bvh_arr10_out, int32_t *tri_order_out, int32_t *n_nodes_out);
int rt_mesh_build_stats(const rt_ctx *ctx, rt_build_stats *out);

/* --- smooth (interpolated) normals: get_smooth_normal of realtime_render.cu:221-245 / global_launcher.cu:207-231
 *     -- beta, gamma by the literal divisions, alpha = 1 - beta - gamma, N = normalize(alpha Na + beta Nb + gamma Nc) --
 *     replaces the flat normal of the winning triangle.  normals_xyz: n_normals * 3; nidx: TriangleIndices::ni,nj,nk of
 *     triangle t at nidx[t * index_stride .. +2], triangles in the order of rt_mesh.indices (for a TriangleIndices array
 *     pass &indices[0].ni and stride 10).  Call after rt_scene_upload (a new upload drops them); NULL = flat again.
 *     rt_mesh_transform then moves the normals the way the reference's kernel does.  Wavefront variants only. ---------- */
int rt_mesh_set_normals(rt_ctx *ctx, const float *normals_xyz, int n_normals, const int32_t *nidx, int index_stride, int n_triangles);

/* --- posed camera + progressive accumulation: the headless form of realtime_render.cu.  Camera
 *     {C, yaw, pitch} with Camera::rotate() (realtime_render.cu:803-861); ray generation and per-sample averaging of its
 *     KernelLaunch (:1100-1134: u_center = C + bz*z + bx*X + by*Y, outcolor += color * (1./num_rays)); accumulation and
 *     display of :1136-1147 (accumbuffer += frame; display = accumbuffer / framenumber; powf(c, 1/2.2f)); the frame's RNG
 *     seed is WangHash(framenumber) (:1190-1197, :1268).  Wavefront variants only.  Parity: checked against the oracle's
 *     restatement; the reference program itself (CUDA + GL, cuRAND) cannot be run, so this row is unpinned. ----------- */
typedef struct rt_camera_pose { float position[3]; float yaw; float pitch; float fov; } rt_camera_pose;
int rt_camera_basis(const rt_camera_pose *pose, float bx[3], float by[3], float bz[3]);   /* Camera::rotate(), host */
/* one frame with the posed camera (no accumulation): full frame to host / rows to device memory */
int rt_render_pose(rt_ctx *ctx, const rt_params *p, const rt_camera_pose *pose, float *out_rgba_host);
int rt_render_pose_device(rt_ctx *ctx, const rt_params *p, const rt_camera_pose *pose, const rt_rows *rows, void *out_rgba_dev, void *stream);
/* disp(): buffer_reset / frames++ / KernelLaunch / display.  Outputs may be NULL.  display: height*width float4
 * (.xyz = accumulated colour / frames, .w = rays traced so far); rgb8: interleaved RGB8 */
int rt_progressive_reset(rt_ctx *ctx);
int rt_progressive_frame(rt_ctx *ctx, const rt_params *p, const rt_camera_pose *pose, float *display_rgba_host, uint8_t *rgb8_host);
int rt_progressive_frames(const rt_ctx *ctx, int *frames);

/* --- one host process, several devices (SURVEY 8b rt_render_multi; the reference uses the implicit device 0,
 *     optimized.cu:828-856).  The frame is cut into RT_MULTI_TILE_ROWS-row tiles, tile k -> device k mod n
 *     (interleaved); the scene is replicated; every device renders its tiles; each peer pushes them over
 *     xGMI into the root device (device_ids[0]); one kernel on the root restores row order.  The result is bitwise
 *     the single-device frame.  Device ids may repeat (several contexts on one device).  The one-process-per-GPU
 *     path (rt_render_device + an RCCL gather) is the other way to use several GPUs. ------------- */
#define RT_MAX_DEVICES 16
#define RT_MULTI_TILE_ROWS 8
typedef struct rt_multi rt_multi;
typedef struct rt_multi_stats {
    int32_t  n_devices;
    int32_t  device_id[RT_MAX_DEVICES];
    float    kernel_ms[RT_MAX_DEVICES];   /* HIP-event time of each device's render kernels                  */
    float    gather_ms;                   /* root: end of its own render -> frame assembled (waits for peers) */
    float    frame_ms;                    /* host wall clock of the whole call                                */
    uint64_t rays;                        /* rays traced for the frame (sum of the .w channel)                */
    uint64_t gather_bytes;                /* bytes the peers moved into the root device for the last frame   */
    int32_t  peer_access[RT_MAX_DEVICES]; /* 1: device k writes the root's memory directly (peer access over xGMI),
                                             0: no peer path, the runtime stages the copy; -1: same device as the root */
    float    submit_ms;                   /* host: call entry -> every device's launches, events and peer copy issued (one submit
                                             thread per device; frame_ms - submit_ms is spent waiting for the devices)          */
} rt_multi_stats;
int rt_multi_create(rt_multi **m, const int *device_ids, int n_devices);
int rt_multi_destroy(rt_multi *m);
const char *rt_multi_last_error(const rt_multi *m);   /* m may be NULL: last global error */
int rt_multi_scene_upload(rt_multi *m, const rt_sphere *spheres, int n_spheres, const rt_mesh *mesh,
                          const rt_light *light, const rt_camera *camera);
int rt_multi_scene_upload_meshes(rt_multi *m, const rt_sphere *spheres, int n_spheres, const rt_mesh *meshes, int n_meshes,
                                 const rt_light *light, const rt_camera *camera);   /* as rt_scene_upload_meshes, on every device */
/* full frame, height*width float4, to host memory / to memory of the root device */
int rt_render_multi(rt_multi *m, const rt_params *p, float *out_rgba_host);
int rt_render_multi_device(rt_multi *m, const rt_params *p, void *out_rgba_dev_on_root);
/* the PNG path: every device tonemaps its tiles (cpu:714-716) and the exchange moves the 8-bit image, 3 bytes per pixel
 * instead of 16 (the reference copies its 8-bit image off the device too, optimized.cu:856); height*width*3 bytes, the
 * bytes rt_render_rgb8 gives on one device */
int rt_render_multi_rgb8(rt_multi *m, const rt_params *p, uint8_t *out_rgb8_host);
int rt_multi_get_stats(rt_multi *m, rt_multi_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACE_HIP_H */
