/*
 * rt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rt_oracle.h).
 *
 * Plain-C restatement of the render path of /root/reference/cpu_launcher.cpp.
 * Build: gcc -O3 -fopenmp -ffp-contract=off (oracle/Makefile).  -ffp-contract=off
 * because the reference binary (g++ -O3 on baseline x86-64, Makefile:38) contains
 * no fused multiply-adds: every + - * / sqrt below is a single IEEE-754 binary32
 * (or, where the reference mixes in double literals, binary64) operation in the
 * reference's source order.
 *
 * "cpu:N" = /root/reference/cpu_launcher.cpp line N,
 * "opt:N" = /root/reference/optimized.cu line N.
 *
 * Deliberate, documented departures from the literal source:
 *  - uniform() (cpu:531-536, mt19937 seeded by clock()) is replaced by a counter
 *    RNG keyed (seed,pixel,sample,depth,dim); the reference is not reproducible
 *    for num_bounce>=1, so stochastic parity is defined on this RNG.
 *  - cpu:288-292 reads uninitialised t_left/t_right (UB); the Makefile's -O3 build
 *    behaves as "push every child whose box is hit"  and that is what
 *    is restated; the golden image pins it.
 *  - W, H, camera, light, epsilons, OBJ scale/offset are parameters instead of
 *    literals so that the BASELINE configs (1920x1080 ...) can be rendered.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* cpu:31-34 */
#define OR_PI 3.14159265358979323846
#define OR_INF (1e9 + 9)

/* ------------------------------------------------------------------ Vector */
/* cpu:45-96 */
typedef struct { float d[3]; } vec;

static inline vec V(float x, float y, float z) { vec r = {{x, y, z}}; return r; }
static inline vec vadd(vec a, vec b) { return V(a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]); }
static inline vec vsub(vec a, vec b) { return V(a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]); }
static inline vec vneg(vec a) { return V(-a.d[0], -a.d[1], -a.d[2]); }
static inline vec smul(float a, vec b) { return V(a * b.d[0], a * b.d[1], a * b.d[2]); }   /* cpu:78 */
static inline vec vmuls(vec a, float b) { return V(a.d[0] * b, a.d[1] * b, a.d[2] * b); }  /* cpu:81 */
static inline vec vmul(vec a, vec b) { return V(a.d[0] * b.d[0], a.d[1] * b.d[1], a.d[2] * b.d[2]); }
static inline vec vdivs(vec a, float b) { return V(a.d[0] / b, a.d[1] / b, a.d[2] / b); }  /* cpu:88 */
static inline float dot(vec a, vec b) { return a.d[0] * b.d[0] + a.d[1] * b.d[1] + a.d[2] * b.d[2]; }
static inline vec cross(vec a, vec b) {
    return V(a.d[1] * b.d[2] - a.d[2] * b.d[1], a.d[2] * b.d[0] - a.d[0] * b.d[2], a.d[0] * b.d[1] - a.d[1] * b.d[0]);
}
static inline float norm2(vec a) { return a.d[0] * a.d[0] + a.d[1] * a.d[1] + a.d[2] * a.d[2]; }
/* cpu:55-57: sqrt() of a float argument; binary32 result either way */
static inline float norm(vec a) { return sqrtf(norm2(a)); }
static inline vec normalize(vec a) { float n = norm(a); return V(a.d[0] / n, a.d[1] / n, a.d[2] / n); } /* cpu:58-63 */
#define SQR(X) ((X) * (X))

typedef struct { vec O, u; float refraction_index; } ray;   /* cpu:98-104 */
static inline ray R(vec O, vec u, float n) { ray r; r.O = O; r.u = u; r.refraction_index = n; return r; }

static inline void cnt_add(or_counters *c, const or_counters *d) {
    c->rays += d->rays; c->mesh_rays += d->mesh_rays; c->box_tests += d->box_tests;
    c->nodes += d->nodes; c->tri_tests += d->tri_tests;
}

/* ------------------------------------------------------------- BoundingBox */
/* cpu:131-158 */
typedef struct { vec mn, mx; } bbox;

static inline bbox bbox_empty(void) {
    bbox b;
    b.mn = V((float)OR_INF, (float)OR_INF, (float)OR_INF);        /* cpu:135 */
    b.mx = V((float)-OR_INF, (float)-OR_INF, (float)-OR_INF);
    return b;
}
/* std::min(a,b) = (b<a)?b:a ; std::max(a,b) = (a<b)?b:a */
static inline float stdmin(float a, float b) { return (b < a) ? b : a; }
static inline float stdmax(float a, float b) { return (a < b) ? b : a; }
static inline void bbox_update(bbox *b, vec v) {                  /* cpu:137-144 */
    for (int k = 0; k < 3; k++) {
        b->mn.d[k] = stdmin(b->mn.d[k], v.d[k]);
        b->mx.d[k] = stdmax(b->mx.d[k], v.d[k]);
    }
}
/* cpu:146-157.  Never writes t; no tmax>0 test; strict '>' (SURVEY H7).
 * std::min({a,b,c}) is min_element: m=a; if(b<m)m=b; if(c<m)m=c;
 * std::max({a,b,c}) is max_element: m=a; if(m<b)m=b; if(m<c)m=c. */
static inline int bbox_intersect(const bbox *b, const ray *r) {
    float t0x = (b->mn.d[0] - r->O.d[0]) / r->u.d[0];
    float t0y = (b->mn.d[1] - r->O.d[1]) / r->u.d[1];
    float t0z = (b->mn.d[2] - r->O.d[2]) / r->u.d[2];
    float t1x = (b->mx.d[0] - r->O.d[0]) / r->u.d[0];
    float t1y = (b->mx.d[1] - r->O.d[1]) / r->u.d[1];
    float t1z = (b->mx.d[2] - r->O.d[2]) / r->u.d[2];
    float tmp;
    if (t0x > t1x) { tmp = t0x; t0x = t1x; t1x = tmp; }
    if (t0y > t1y) { tmp = t0y; t0y = t1y; t1y = tmp; }
    if (t0z > t1z) { tmp = t0z; t0z = t1z; t1z = tmp; }
    float mn = t1x; if (t1y < mn) mn = t1y; if (t1z < mn) mn = t1z;
    float mx = t0x; if (mx < t0y) mx = t0y; if (mx < t0z) mx = t0z;
    return mn > mx;
}

/* ------------------------------------------------------------------- BVH */
/* cpu:160-165 */
typedef struct bvh_node {
    struct bvh_node *left, *right;
    bbox bb;
    int triangle_start, triangle_end;
} bvh_node;

/* cpu:121-129: only vtxi,vtxj,vtxk are read on the render path */
typedef struct { int vtxi, vtxj, vtxk; int ni, nj, nk; } tri_idx;   /* ni,nj,nk: smooth shading only (realtime_render.cu:221-245) */

struct or_mesh {
    vec *vertices; int nv, cap_v;
    tri_idx *indices; int nt, cap_t;
    int n_normals, n_uvs;          /* counts only: negative-index resolution, cpu:385-390 */
    bvh_node *bvh;                 /* root; NULL until built */
    int n_nodes, max_depth;
    vec albedo;
    int mirror; float in_refraction_index, out_refraction_index;   /* Geometry's other fields (cpu:106-118): a TriangleMesh inherits them like a Sphere does */
    vec *normals; int n_shading_normals;   /* smooth shading; NULL = flat, as in cpu_launcher.cpp */
};

or_mesh *or_mesh_new(void) {
    or_mesh *m = (or_mesh *)calloc(1, sizeof(or_mesh));
    m->albedo = V(0, 0, 0);
    m->mirror = 0; m->in_refraction_index = 1; m->out_refraction_index = 1;   /* Geometry(), cpu:110 */
    return m;
}
static void bvh_free(bvh_node *n) {
    if (!n) return;
    bvh_free(n->left); bvh_free(n->right); free(n);
}
void or_mesh_free(or_mesh *m) {
    if (!m) return;
    bvh_free(m->bvh); free(m->vertices); free(m->indices); free(m->normals); free(m);
}
static void push_vertex(or_mesh *m, vec v) {
    if (m->nv == m->cap_v) { m->cap_v = m->cap_v ? 2 * m->cap_v : 1024; m->vertices = (vec *)realloc(m->vertices, sizeof(vec) * m->cap_v); }
    m->vertices[m->nv++] = v;
}
static void push_tri(or_mesh *m, int i, int j, int k) {
    if (m->nt == m->cap_t) { m->cap_t = m->cap_t ? 2 * m->cap_t : 1024; m->indices = (tri_idx *)realloc(m->indices, sizeof(tri_idx) * m->cap_t); }
    tri_idx t = {i, j, k, -1, -1, -1};
    m->indices[m->nt++] = t;
}
void or_mesh_set_albedo(or_mesh *m, float r, float g, float b) { m->albedo = V(r, g, b); }
/* mesh_ptr->mirror / in_refraction_index / out_refraction_index: public members of Geometry (cpu:113-116) that getColor reads for ANY object hit (cpu:573, 580) */
void or_mesh_set_material(or_mesh *m, int mirror, float n_in, float n_out) { m->mirror = mirror; m->in_refraction_index = n_in; m->out_refraction_index = n_out; }

/* vertex index resolution used all over cpu:382-479: 1-based, negative = relative */
static inline int vidx(const or_mesh *m, int i) { return (i < 0) ? m->nv + i : i - 1; }

/* readOBJ, cpu:315-493.  Only what feeds the render path is kept (vertex
 * positions and vtx indices); vn/vt are counted because the parser's control
 * flow (which sscanf pattern matches) does not depend on their values. */
int or_mesh_read_obj(or_mesh *m, const char *path, float scale, const float offset[3]) {
    FILE *f = fopen(path, "r");
    if (f == NULL) {                      /* cpu:322-325 */
        printf("Error opening file!\n"); fflush(stdout);
        return -1;
    }
    char line[255];
    while (!feof(f)) {
        if (!fgets(line, 255, f)) break;  /* cpu:329 */
        /* cpu:331-333: erase after the last char not in " \r\t" ('\n' is not in the set) */
        {
            int len = (int)strlen(line), last = -1;
            for (int q = len - 1; q >= 0; q--) {
                if (line[q] != ' ' && line[q] != '\r' && line[q] != '\t') { last = q; break; }
            }
            line[last + 1] = '\0';
        }
        if (line[0] == 'v' && line[1] == ' ') {            /* cpu:340-358 */
            vec v = V(0, 0, 0); float c0, c1, c2;
            if (sscanf(line, "v %f %f %f %f %f %f\n", &v.d[0], &v.d[1], &v.d[2], &c0, &c1, &c2) == 6) {
                push_vertex(m, v);                         /* coloured vertices are not transformed */
            } else {
                sscanf(line, "v %f %f %f\n", &v.d[0], &v.d[1], &v.d[2]);
                v = vadd(vmuls(v, scale), V(offset[0], offset[1], offset[2]));   /* cpu:354 */
                push_vertex(m, v);
            }
        }
        if (line[0] == 'v' && line[1] == 'n') m->n_normals++;   /* cpu:359-363 */
        if (line[0] == 'v' && line[1] == 't') m->n_uvs++;       /* cpu:364-368 */
        if (line[0] == 'f') {                                   /* cpu:369-488 */
            int i0 = 0, i1 = 0, i2 = 0, i3 = 0, j0, j1, j2, j3, k0, k1, k2, k3, nn, offset_c = 0;
            char *consumed = line + 1;
            nn = sscanf(consumed, "%u/%u/%u %u/%u/%u %u/%u/%u%n", (unsigned *)&i0, (unsigned *)&j0, (unsigned *)&k0,
                        (unsigned *)&i1, (unsigned *)&j1, (unsigned *)&k1, (unsigned *)&i2, (unsigned *)&j2, (unsigned *)&k2, &offset_c);
            if (nn == 9) {
                push_tri(m, vidx(m, i0), vidx(m, i1), vidx(m, i2));
            } else {
                nn = sscanf(consumed, "%u/%u %u/%u %u/%u%n", (unsigned *)&i0, (unsigned *)&j0, (unsigned *)&i1, (unsigned *)&j1,
                            (unsigned *)&i2, (unsigned *)&j2, &offset_c);
                if (nn == 6) {
                    push_tri(m, vidx(m, i0), vidx(m, i1), vidx(m, i2));
                } else {
                    nn = sscanf(consumed, "%u %u %u%n", (unsigned *)&i0, (unsigned *)&i1, (unsigned *)&i2, &offset_c);
                    if (nn == 3) {
                        push_tri(m, vidx(m, i0), vidx(m, i1), vidx(m, i2));
                    } else {
                        nn = sscanf(consumed, "%u//%u %u//%u %u//%u%n", (unsigned *)&i0, (unsigned *)&k0, (unsigned *)&i1,
                                    (unsigned *)&k1, (unsigned *)&i2, (unsigned *)&k2, &offset_c);
                        push_tri(m, vidx(m, i0), vidx(m, i1), vidx(m, i2));   /* cpu:410-417, unconditional */
                    }
                }
            }
            consumed += offset_c;
            for (;;) {                                          /* fan triangulation, cpu:424-486 */
                if (consumed[0] == '\n') break;
                if (consumed[0] == '\0') break;
                nn = sscanf(consumed, "%u/%u/%u%n", (unsigned *)&i3, (unsigned *)&j3, (unsigned *)&k3, &offset_c);
                if (nn == 3) {
                    push_tri(m, vidx(m, i0), vidx(m, i2), vidx(m, i3));
                    consumed += offset_c; i2 = i3;
                } else {
                    nn = sscanf(consumed, "%u/%u%n", (unsigned *)&i3, (unsigned *)&j3, &offset_c);
                    if (nn == 2) {
                        push_tri(m, vidx(m, i0), vidx(m, i2), vidx(m, i3));
                        consumed += offset_c; i2 = i3;
                    } else {
                        nn = sscanf(consumed, "%u//%u%n", (unsigned *)&i3, (unsigned *)&k3, &offset_c);
                        if (nn == 2) {
                            push_tri(m, vidx(m, i0), vidx(m, i2), vidx(m, i3));
                            consumed += offset_c; i2 = i3;
                        } else {
                            nn = sscanf(consumed, "%u%n", (unsigned *)&i3, &offset_c);
                            if (nn == 1) {
                                push_tri(m, vidx(m, i0), vidx(m, i2), vidx(m, i3));
                                consumed += offset_c; i2 = i3;
                            } else {
                                consumed += 1;
                            }
                        }
                    }
                }
            }
        }
    }
    fclose(f);
    return 0;
}

void or_mesh_set_arrays(or_mesh *m, const float *verts_xyz, int nv, const int32_t *tri_vidx, int nt) {
    m->nv = 0; m->nt = 0;
    for (int i = 0; i < nv; i++) push_vertex(m, V(verts_xyz[3 * i], verts_xyz[3 * i + 1], verts_xyz[3 * i + 2]));
    for (int i = 0; i < nt; i++) push_tri(m, tri_vidx[3 * i], tri_vidx[3 * i + 1], tri_vidx[3 * i + 2]);
    bvh_free(m->bvh); m->bvh = NULL; m->n_nodes = 0; m->max_depth = 0;
}

/* vertex normals + per-triangle normal indices (TriangleIndices::ni,nj,nk) in the CURRENT triangle order; NULL = flat */
void or_mesh_set_normals(or_mesh *m, const float *normals_xyz, int n, const int32_t *nidx) {
    free(m->normals); m->normals = NULL; m->n_shading_normals = 0;
    if (!normals_xyz || !nidx) return;
    m->normals = (vec *)malloc(sizeof(vec) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) m->normals[i] = V(normals_xyz[3 * i], normals_xyz[3 * i + 1], normals_xyz[3 * i + 2]);
    m->n_shading_normals = n;
    for (int i = 0; i < m->nt; i++) { m->indices[i].ni = nidx[3 * i]; m->indices[i].nj = nidx[3 * i + 1]; m->indices[i].nk = nidx[3 * i + 2]; }
}

/* opt:297-301 */
void or_mesh_rescale(or_mesh *m, float scale, const float offset[3]) {
    for (int i = 0; i < m->nv; i++) m->vertices[i] = vadd(vmuls(m->vertices[i], scale), V(offset[0], offset[1], offset[2]));
}

/* cpu:180-188 */
static bbox compute_bbox(const or_mesh *m, int ts, int te) {
    bbox bb = bbox_empty();
    for (int i = ts; i < te; i++) {
        bbox_update(&bb, m->vertices[m->indices[i].vtxi]);
        bbox_update(&bb, m->vertices[m->indices[i].vtxj]);
        bbox_update(&bb, m->vertices[m->indices[i].vtxk]);
    }
    return bb;
}

/* cpu:190-224 */
/* rotate + transform kernels, global_launcher.cu:340-365 (realtime_render.cu:1151-1166 launches the same):
 * v' = (R[0]*v0 + R[1]*v1 + R[2]*v2, ...) then += translation.  Vertices only (this restatement carries no normals). */
void or_mesh_transform(or_mesh *m, const float R[9], const float t[3]) {
    for (int i = 0; i < m->nv; i++) {
        const vec v = m->vertices[i];
        vec r = V(R[0] * v.d[0] + R[1] * v.d[1] + R[2] * v.d[2],
                  R[3] * v.d[0] + R[4] * v.d[1] + R[5] * v.d[2],
                  R[6] * v.d[0] + R[7] * v.d[1] + R[8] * v.d[2]);
        r.d[0] += t[0]; r.d[1] += t[1]; r.d[2] += t[2];
        m->vertices[i] = r;
    }
    for (int i = 0; i < m->n_shading_normals; i++) {   /* the kernel also ADDS the translation to the normals (global_launcher.cu:357-363) */
        const vec v = m->normals[i];
        vec r = V(R[0] * v.d[0] + R[1] * v.d[1] + R[2] * v.d[2],
                  R[3] * v.d[0] + R[4] * v.d[1] + R[5] * v.d[2],
                  R[6] * v.d[0] + R[7] * v.d[1] + R[8] * v.d[2]);
        r.d[0] += t[0]; r.d[1] += t[1]; r.d[2] += t[2];
        m->normals[i] = r;
    }
}
/* keep the tree, recompute every node's box from its triangle range exactly as buildBVH does (cpu:193 compute_bbox) */
static void refit_node(or_mesh *m, bvh_node *n) {
    if (!n) return;
    n->bb = compute_bbox(m, n->triangle_start, n->triangle_end);
    refit_node(m, n->left); refit_node(m, n->right);
}
void or_mesh_refit(or_mesh *m) { refit_node(m, m->bvh); }

static void build_bvh(or_mesh *m, bvh_node *cur, int ts, int te, int depth) {
    m->n_nodes++;
    if (depth > m->max_depth) m->max_depth = depth;
    cur->triangle_start = ts;
    cur->triangle_end = te;
    cur->left = NULL;
    cur->right = NULL;
    cur->bb = compute_bbox(m, ts, te);

    vec diag = vsub(cur->bb.mx, cur->bb.mn);
    int max_axis;
    if (diag.d[0] >= diag.d[1] && diag.d[0] >= diag.d[2]) max_axis = 0;
    else if (diag.d[1] >= diag.d[0] && diag.d[1] >= diag.d[2]) max_axis = 1;
    else max_axis = 2;

    int pivot = ts;
    float split = (cur->bb.mn.d[max_axis] + cur->bb.mx.d[max_axis]) / 2;
    for (int i = ts; i < te; i++) {
        float cen = (m->vertices[m->indices[i].vtxi].d[max_axis] + m->vertices[m->indices[i].vtxj].d[max_axis] +
                     m->vertices[m->indices[i].vtxk].d[max_axis]) / 3;
        if (cen < split) {
            tri_idx tmp = m->indices[i]; m->indices[i] = m->indices[pivot]; m->indices[pivot] = tmp;
            pivot++;
        }
    }
    if (pivot <= ts || pivot >= te - 1 || te - ts < 5) return;     /* cpu:217 */
    cur->left = (bvh_node *)calloc(1, sizeof(bvh_node));
    cur->right = (bvh_node *)calloc(1, sizeof(bvh_node));
    build_bvh(m, cur->left, ts, pivot, depth + 1);
    build_bvh(m, cur->right, pivot, te, depth + 1);
}

void or_mesh_build_bvh(or_mesh *m) {
    bvh_free(m->bvh);
    m->bvh = (bvh_node *)calloc(1, sizeof(bvh_node));
    m->n_nodes = 0; m->max_depth = 0;
    build_bvh(m, m->bvh, 0, m->nt, 0);                              /* cpu:684 */
}

/* A caller-supplied tree (test infrastructure for SURVEY 8f3's LBVH: parity is "HIP == oracle ON THE SAME TREE").  arr10 is the flat layout
 * of bvhTreeToArray (optimized.cu:512-534): per node [left, right, mn.xyz, mx.xyz, triangle_start, triangle_end), -1 = no child, node 0 the
 * root, any numbering; order[k] = index (in the CURRENT triangle array) of the triangle that moves to position k -- the ranges refer to the
 * new positions, as the reference's ranges refer to `indices` after its in-place partition.  The traversal (mesh_intersect) is unchanged:
 * it walks whatever tree hangs off m->bvh.  Returns 0, or -1 for a malformed tree (nothing is changed then). */
static bvh_node *bvh_from_array(const float *arr, int n_nodes, int idx, int depth, int *count, int *maxd, int nt, int *ok) {
    if (idx < 0 || idx >= n_nodes || *count >= n_nodes || depth > 4096) { *ok = 0; return NULL; }
    const float *a = arr + (size_t)idx * 10;
    bvh_node *n = (bvh_node *)calloc(1, sizeof(bvh_node));
    (*count)++;
    if (depth > *maxd) *maxd = depth;
    n->bb.mn = V(a[2], a[3], a[4]); n->bb.mx = V(a[5], a[6], a[7]);
    n->triangle_start = (int)a[8]; n->triangle_end = (int)a[9];
    if (n->triangle_start < 0 || n->triangle_end < n->triangle_start || n->triangle_end > nt) *ok = 0;
    const int l = (int)a[0], r = (int)a[1];
    if ((l < 0) != (r < 0)) *ok = 0;
    if (l >= 0 && r >= 0 && *ok) {
        n->left = bvh_from_array(arr, n_nodes, l, depth + 1, count, maxd, nt, ok);
        n->right = bvh_from_array(arr, n_nodes, r, depth + 1, count, maxd, nt, ok);
    }
    return n;
}
int or_mesh_set_bvh(or_mesh *m, const float *arr10, int n_nodes, const int32_t *order) {
    if (!m || !arr10 || n_nodes < 1) return -1;
    if (order) {
        char *seen = (char *)calloc((size_t)(m->nt > 0 ? m->nt : 1), 1);
        for (int k = 0; k < m->nt; k++) {
            if (order[k] < 0 || order[k] >= m->nt || seen[order[k]]) { free(seen); return -1; }   /* a permutation, nothing else */
            seen[order[k]] = 1;
        }
        free(seen);
    }
    int count = 0, maxd = 0, ok = 1;
    bvh_node *root = bvh_from_array(arr10, n_nodes, 0, 0, &count, &maxd, m->nt, &ok);
    if (!ok || count != n_nodes) { bvh_free(root); return -1; }
    if (order) {
        tri_idx *nw = (tri_idx *)malloc(sizeof(tri_idx) * (size_t)(m->nt > 0 ? m->nt : 1));
        for (int k = 0; k < m->nt; k++) nw[k] = m->indices[order[k]];
        memcpy(m->indices, nw, sizeof(tri_idx) * (size_t)m->nt);
        free(nw);
    }
    bvh_free(m->bvh);
    m->bvh = root; m->n_nodes = n_nodes; m->max_depth = maxd;
    return 0;
}

int or_mesh_num_vertices(const or_mesh *m) { return m->nv; }
int or_mesh_num_triangles(const or_mesh *m) { return m->nt; }
int or_mesh_num_nodes(const or_mesh *m) { return m->n_nodes; }
int or_mesh_max_depth(const or_mesh *m) { return m->max_depth; }
void or_mesh_get_vertices(const or_mesh *m, float *o) {
    for (int i = 0; i < m->nv; i++) { o[3 * i] = m->vertices[i].d[0]; o[3 * i + 1] = m->vertices[i].d[1]; o[3 * i + 2] = m->vertices[i].d[2]; }
}
void or_mesh_get_triangles(const or_mesh *m, int32_t *o) {
    for (int i = 0; i < m->nt; i++) { o[3 * i] = m->indices[i].vtxi; o[3 * i + 1] = m->indices[i].vtxj; o[3 * i + 2] = m->indices[i].vtxk; }
}

/* opt:512-534 */
static void tree_to_array(const bvh_node *cur, float *arr, size_t *arr_size, size_t idx) {
    arr[idx * 10 + 2] = cur->bb.mn.d[0]; arr[idx * 10 + 3] = cur->bb.mn.d[1]; arr[idx * 10 + 4] = cur->bb.mn.d[2];
    arr[idx * 10 + 5] = cur->bb.mx.d[0]; arr[idx * 10 + 6] = cur->bb.mx.d[1]; arr[idx * 10 + 7] = cur->bb.mx.d[2];
    arr[idx * 10 + 8] = (float)cur->triangle_start;
    arr[idx * 10 + 9] = (float)cur->triangle_end;
    if (cur->left) {
        arr[idx * 10 + 0] = (float)((*arr_size)++);
        tree_to_array(cur->left, arr, arr_size, (size_t)arr[idx * 10 + 0]);
    } else arr[idx * 10 + 0] = -1;
    if (cur->right) {
        arr[idx * 10 + 1] = (float)((*arr_size)++);
        tree_to_array(cur->right, arr, arr_size, (size_t)arr[idx * 10 + 1]);
    } else arr[idx * 10 + 1] = -1;
}
void or_mesh_bvh_to_array(const or_mesh *m, float *out) {
    size_t n = 1;                                                   /* opt:812 */
    if (m->bvh) tree_to_array(m->bvh, out, &n, 0);
}

/* cpu:226-236 */
static inline int moller_trumbore(vec A, vec B, vec C, vec *N, const ray *r, float *t) {
    vec e1 = vsub(B, A);
    vec e2 = vsub(C, A);
    *N = cross(e1, e2);
    if (dot(r->u, *N) == 0) return 0;
    float beta = dot(e2, cross(vsub(A, r->O), r->u)) / dot(r->u, *N);
    float gamma = -dot(e1, cross(vsub(A, r->O), r->u)) / dot(r->u, *N);
    if (!(0 <= beta && beta <= 1) || !(0 <= gamma && gamma <= 1)) return 0;
    *t = dot(vsub(A, r->O), *N) / dot(r->u, *N);
    return beta + gamma <= 1 && *t > 0;
}

#define OR_STACK_MAX 256

/* cpu:277-311 (ENABLE_BVH).  "Hit" is restated as "some triangle accepted":
 * the reference returns t_min != INF with INF a double (always true),
 * which Scene::intersect_all neutralises through t < t_min (cpu:554). */
static int mesh_intersect(const or_mesh *m, const ray *r, float tri_tmin, float *t, vec *N, or_counters *cnt) {
    if (!m->bvh) return 0;
    cnt->box_tests++;
    if (!bbox_intersect(&m->bvh->bb, r)) return 0;                 /* cpu:279 */
    cnt->mesh_rays++;
    const bvh_node *stack[OR_STACK_MAX];
    int sp = 0;
    stack[sp++] = m->bvh;

    float t_min = (float)OR_INF;                                    /* cpu:283 */
    int any = 0, idx_min = -1;
    vec Nbest = V(0, 0, 0);
    while (sp) {
        const bvh_node *cur = stack[--sp];
        cnt->nodes++;
        if (cur->left) {
            cnt->box_tests += 2;
            int ok_left = bbox_intersect(&cur->left->bb, r);
            int ok_right = bbox_intersect(&cur->right->bb, r);
            /* cpu:291-292 with the -O3 behaviour of the uninitialised t_left/t_right */
            if (ok_left) stack[sp++] = cur->left;
            if (ok_right) stack[sp++] = cur->right;
        } else {
            for (int i = cur->triangle_start; i < cur->triangle_end; i++) {
                float t_cur;
                vec A = m->vertices[m->indices[i].vtxi], B = m->vertices[m->indices[i].vtxj], C = m->vertices[m->indices[i].vtxk];
                vec N_triangle;
                cnt->tri_tests++;
                int inter = moller_trumbore(A, B, C, &N_triangle, r, &t_cur);
                if (!inter) continue;
                if (t_cur > tri_tmin && t_cur < t_min) {            /* cpu:301 */
                    t_min = t_cur;
                    Nbest = N_triangle;
                    any = 1; idx_min = i;
                }
            }
        }
    }
    if (!any) return 0;
    *N = normalize(Nbest);                                          /* cpu:308 */
    if (m->normals && idx_min >= 0) {                               /* get_smooth_normal, realtime_render.cu:221-245 */
        const tri_idx tid = m->indices[idx_min];
        const vec A = m->vertices[tid.vtxi], B = m->vertices[tid.vtxj], C = m->vertices[tid.vtxk];
        const vec e1 = vsub(B, A), e2 = vsub(C, A);
        const vec Nt = cross(e1, e2);
        const float beta = dot(e2, cross(vsub(A, r->O), r->u)) / dot(r->u, Nt);
        const float gamma = -dot(e1, cross(vsub(A, r->O), r->u)) / dot(r->u, Nt);
        const float alpha = 1 - beta - gamma;
        const vec Na = m->normals[tid.ni], Nb = m->normals[tid.nj], Nc = m->normals[tid.nk];
        *N = normalize(vadd(vadd(smul(alpha, Na), smul(beta, Nb)), smul(gamma, Nc)));
    }
    *t = t_min;
    return 1;
}

int or_mesh_intersect(const or_mesh *m, const float O[3], const float u[3], float tri_tmin, float *t, float N[3], or_counters *cnt) {
    or_counters local = {0, 0, 0, 0, 0};
    ray r = R(V(O[0], O[1], O[2]), V(u[0], u[1], u[2]), 1.f);
    vec n = V(0, 0, 0); float tt = 0;
    int hit = mesh_intersect(m, &r, tri_tmin, &tt, &n, &local);
    if (hit) { *t = tt; N[0] = n.d[0]; N[1] = n.d[1]; N[2] = n.d[2]; }
    if (cnt) cnt_add(cnt, &local);
    return hit;
}

/* ---------------------------------------------------------------- Sphere */
/* cpu:512-527 */
static inline int sphere_intersect(vec C, float Rr, const ray *r, float *t, vec *N) {
    float delta = SQR(dot(r->u, vsub(r->O, C))) - (norm2(vsub(r->O, C)) - Rr * Rr);
    if (delta < 0) return 0;
    float t1 = dot(r->u, vsub(C, r->O)) - sqrtf(delta);
    float t2 = dot(r->u, vsub(C, r->O)) + sqrtf(delta);
    if (t2 < 0) return 0;
    *t = t1 < 0 ? t2 : t1;
    *N = normalize(vsub(vadd(r->O, smul(*t, r->u)), C));
    return 1;
}

int or_sphere_intersect(const float C[3], float Rr, const float O[3], const float u[3], float *t, float N[3]) {
    ray r = R(V(O[0], O[1], O[2]), V(u[0], u[1], u[2]), 1.f);
    vec n; float tt;
    int hit = sphere_intersect(V(C[0], C[1], C[2]), Rr, &r, &tt, &n);
    if (hit) { *t = tt; N[0] = n.d[0]; N[1] = n.d[1]; N[2] = n.d[2]; }
    return hit;
}
int or_box_intersect(const float mn[3], const float mx[3], const float O[3], const float u[3]) {
    bbox b; b.mn = V(mn[0], mn[1], mn[2]); b.mx = V(mx[0], mx[1], mx[2]);
    ray r = R(V(O[0], O[1], O[2]), V(u[0], u[1], u[2]), 1.f);
    return bbox_intersect(&b, &r);
}
int or_moller_trumbore(const float A[3], const float B[3], const float C[3], const float O[3], const float u[3], float *t, float N[3]) {
    ray r = R(V(O[0], O[1], O[2]), V(u[0], u[1], u[2]), 1.f);
    vec n; float tt = 0;
    int hit = moller_trumbore(V(A[0], A[1], A[2]), V(B[0], B[1], B[2]), V(C[0], C[1], C[2]), &n, &r, &tt);
    N[0] = n.d[0]; N[1] = n.d[1]; N[2] = n.d[2];
    if (hit) *t = tt;
    return hit;
}

/* ----------------------------------------------------------------- Scene */
/* Geometry, cpu:106-118 ; Scene, cpu:538-652 */
#define OR_MAX_OBJECTS 64
typedef struct {
    int is_mesh;
    vec C; float Rr;              /* sphere */
    const or_mesh *mesh;          /* mesh   */
    vec albedo; int id; int mirror; float in_refraction_index, out_refraction_index;
} geometry;

struct or_scene {
    geometry objects[OR_MAX_OBJECTS];
    int n;
    float intensity;              /* cpu:650 */
    vec L;                        /* cpu:651 */
};

or_scene *or_scene_new(void) {
    or_scene *s = (or_scene *)calloc(1, sizeof(or_scene));
    s->intensity = 3e10f;
    s->L = V(-10.f, 20.f, 40.f);
    return s;
}
void or_scene_free(or_scene *s) { free(s); }
int or_scene_add_sphere(or_scene *s, const float C[3], float Rr, const float albedo[3], int mirror, float n_in, float n_out) {
    if (s->n >= OR_MAX_OBJECTS) return -1;
    geometry *g = &s->objects[s->n];
    memset(g, 0, sizeof(*g));
    g->C = V(C[0], C[1], C[2]); g->Rr = Rr; g->albedo = V(albedo[0], albedo[1], albedo[2]);
    g->mirror = mirror; g->in_refraction_index = n_in; g->out_refraction_index = n_out;
    g->id = s->n;                                                   /* cpu:541 */
    return s->n++;
}
int or_scene_add_mesh(or_scene *s, or_mesh *m) {
    if (s->n >= OR_MAX_OBJECTS) return -1;
    geometry *g = &s->objects[s->n];
    memset(g, 0, sizeof(*g));
    g->is_mesh = 1; g->mesh = m; g->albedo = m->albedo;
    g->mirror = m->mirror; g->in_refraction_index = m->in_refraction_index; g->out_refraction_index = m->out_refraction_index;   /* cpu:110 unless the caller set them */
    g->id = s->n;
    return s->n++;
}
void or_scene_set_light(or_scene *s, const float L[3], float intensity) { s->L = V(L[0], L[1], L[2]); s->intensity = intensity; }

/* cpu:545-564 */
static int intersect_all(const or_scene *s, const ray *r, float tri_tmin, vec *P, vec *N, int *objectId, or_counters *cnt) {
    float t_min = (float)OR_INF;
    int id_min = -1;
    vec N_min = V(0, 0, 0);
    cnt->rays++;
    for (int k = 0; k < s->n; k++) {
        const geometry *g = &s->objects[k];
        float t = 0;
        vec N_tmp = V(0, 0, 0);
        int ok = g->is_mesh ? mesh_intersect(g->mesh, r, tri_tmin, &t, &N_tmp, cnt)
                            : sphere_intersect(g->C, g->Rr, r, &t, &N_tmp);
        if (ok && t < t_min) {
            t_min = t;
            id_min = g->id;
            N_min = N_tmp;
        }
    }
    *P = vadd(r->O, smul(t_min, r->u));                             /* cpu:560, also on a miss */
    *objectId = id_min;
    *N = N_min;
    return id_min != -1;
}

int or_scene_intersect_all(const or_scene *s, const float O[3], const float u[3], float tri_tmin, float P[3], float N[3], int *object_id, or_counters *cnt) {
    or_counters local = {0, 0, 0, 0, 0};
    ray r = R(V(O[0], O[1], O[2]), V(u[0], u[1], u[2]), 1.f);
    vec p, n; int id;
    int hit = intersect_all(s, &r, tri_tmin, &p, &n, &id, &local);
    for (int k = 0; k < 3; k++) { P[k] = p.d[k]; N[k] = n.d[k]; }
    *object_id = id;
    if (cnt) cnt_add(cnt, &local);
    return hit;
}

/* ------------------------------------------------------------------- RNG */
/* Counter RNG (DESIGN.md "RNG"): three rounds of the lowbias32 integer
 * finaliser over (seed, pixel, sample, depth*4+dim); 24 random bits mapped to
 * (0,1] so that log(r1) (cpu:707) is finite.  Replaces uniform(), cpu:531-536. */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
float or_uniform(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t dim) {
    uint32_t h = mix32(pixel ^ mix32(seed));
    h = mix32(h ^ (sample * 0x9E3779B1U));
    h = mix32(h ^ ((depth * 4U + dim) * 0x85EBCA77U));
    return (float)((h >> 8) + 1U) * 0x1p-24f;
}

/* std::mt19937 (32-bit Mersenne twister, default seed handling of libstdc++) and
 * std::uniform_real_distribution<float>(0,1) as libstdc++ implements it:
 * generate_canonical<float,24> draws ONE 32-bit word, returns float(x)/2^32 and
 * replaces a result that rounded up to 1.0f by nextafterf(1,0).  cpu:531-536. */
typedef struct { uint32_t mt[624]; int idx; } mt19937;
static void mt_seed(mt19937 *g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253U * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(mt19937 *g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->mt[i] & 0x80000000U) | (g->mt[(i + 1) % 624] & 0x7fffffffU);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680U; y ^= (y << 15) & 0xefc60000U; y ^= y >> 18;
    return y;
}
static float mt_uniform(mt19937 *g) {
    float r = (float)mt_next(g) / 4294967296.0f;
    if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
    return r;
}

typedef struct { uint32_t seed, pixel, sample; float eps, tri_tmin; or_counters *cnt; mt19937 *mt; } trace_ctx;
static inline float rng(const trace_ctx *c, uint32_t depth, uint32_t dim) {
    if (c->mt) return mt_uniform(c->mt);
    return or_uniform(c->seed, c->pixel, c->sample, depth, dim);
}

/* cpu:566-648.  depth_index counts segments from the camera (0,1,...) and keys the RNG. */
static vec get_color(const or_scene *s, ray in, int ray_depth, uint32_t depth_index, const trace_ctx *c) {
    if (ray_depth < 0) return V(0.f, 0.f, 0.f);                     /* cpu:567 */
    vec P, N;
    int sphere_id = -1;
    int inter = intersect_all(s, &in, c->tri_tmin, &P, &N, &sphere_id, c->cnt);
    vec color = V(0, 0, 0);
    if (inter) {
        const geometry *g = &s->objects[sphere_id];
        if (g->mirror) {                                            /* cpu:573-579 */
            float epsilon = c->eps;
            vec P_adjusted = vadd(P, smul(epsilon, N));
            vec new_direction = vsub(in.u, smul(2 * dot(in.u, N), N));
            return get_color(s, R(P_adjusted, new_direction, in.refraction_index), ray_depth - 1, depth_index + 1, c);
        } else if (g->in_refraction_index != g->out_refraction_index) {   /* cpu:580-604 */
            float epsilon = c->eps;
            float refract_ratio;
            int out2in = in.refraction_index == g->out_refraction_index;
            if (out2in) {
                refract_ratio = g->out_refraction_index / g->in_refraction_index;
            } else {
                refract_ratio = g->in_refraction_index / g->out_refraction_index;
                N = vneg(N);
            }
            if (((out2in && in.refraction_index > g->in_refraction_index) ||
                 (!out2in && in.refraction_index > g->out_refraction_index)) &&
                SQR(refract_ratio) * (1 - SQR(dot(in.u, N))) > 1) {
                return get_color(s, R(vadd(P, smul(epsilon, N)), vsub(in.u, smul(2 * dot(in.u, N), N)), in.refraction_index),
                                 ray_depth - 1, depth_index + 1, c);
            }
            vec P_adjusted = vsub(P, smul(epsilon, N));
            vec N_component = smul(-sqrtf(1 - SQR(refract_ratio) * (1 - SQR(dot(in.u, N)))), N);
            vec T_component = smul(refract_ratio, vsub(in.u, smul(dot(in.u, N), N)));
            vec new_direction = vadd(N_component, T_component);
            if (out2in) return get_color(s, R(P_adjusted, new_direction, g->in_refraction_index), ray_depth - 1, depth_index + 1, c);
            else        return get_color(s, R(P_adjusted, new_direction, g->out_refraction_index), ray_depth - 1, depth_index + 1, c);
        } else {                                                    /* cpu:605-645 */
            vec P_prime, N_prime;
            int sphere_id_shadow;
            float epsilon = c->eps;
            vec P_adjusted = vadd(P, smul(epsilon, N));
            vec direct_color, indirect_color;
            vec to_light = vsub(s->L, P_adjusted);
            ray shadow = R(P_adjusted, vdivs(to_light, norm(to_light)), 1.f);   /* NORMED_VEC, cpu:30,614 */
            (void)intersect_all(s, &shadow, c->tri_tmin, &P_prime, &N_prime, &sphere_id_shadow, c->cnt);
            if (norm2(vsub(P_prime, P_adjusted)) <= norm2(vsub(s->L, P_adjusted))) {
                direct_color = V(0, 0, 0);
            } else {
                vec wlight = normalize(vsub(s->L, P));
                /* cpu:623: PI is a double literal => evaluated in binary64, narrowed to float l */
                float mx = stdmax(dot(N, wlight), 0.f);
                float l = (float)((double)s->intensity / (4 * OR_PI * (double)norm2(vsub(s->L, P))) * (double)mx);
                /* cpu:624: (l*albedo) / float(PI) */
                direct_color = vdivs(smul(l, g->albedo), (float)OR_PI);
            }
            float r1 = rng(c, depth_index, 0);                      /* cpu:628-629 */
            float r2 = rng(c, depth_index, 1);
            /* cpu:630-632: cos/sin in binary64, sqrt(1-r2) in binary32, product narrowed */
            float x = (float)(cos(2 * OR_PI * (double)r1) * (double)sqrtf(1 - r2));
            float y = (float)(sin(2 * OR_PI * (double)r1) * (double)sqrtf(1 - r2));
            float z = sqrtf(r2);
            vec T1;
            if (fabsf(N.d[1]) != 0 && fabsf(N.d[0]) != 0) T1 = V(-N.d[1], N.d[0], 0);
            else T1 = V(-N.d[2], 0, N.d[0]);
            T1 = normalize(T1);
            vec T2 = cross(N, T1);
            vec random_direction = vadd(vadd(smul(x, T1), smul(y, T2)), smul(z, N));
            indirect_color = vmul(g->albedo, get_color(s, R(P_adjusted, random_direction, 1.f), ray_depth - 1, depth_index + 1, c));
            color = vadd(direct_color, indirect_color);
        }
    }
    return color;
}

void or_scene_get_color(const or_scene *s, const float O[3], const float u[3], int ray_depth, float eps, float tri_tmin,
                        uint32_t seed, uint32_t pixel, uint32_t sample, float out_rgb[3], or_counters *cnt) {
    or_counters local = {0, 0, 0, 0, 0};
    trace_ctx c = {seed, pixel, sample, eps, tri_tmin, &local, NULL};
    vec col = get_color(s, R(V(O[0], O[1], O[2]), V(u[0], u[1], u[2]), 1.f), ray_depth, 0, &c);
    out_rgb[0] = col.d[0]; out_rgb[1] = col.d[1]; out_rgb[2] = col.d[2];
    if (cnt) cnt_add(cnt, &local);
}

/* cpu:714-716: std::min(std::pow(c, 1./2.2), 255.) converted to unsigned char */
static inline uint8_t tonemap1(float c) {
    double v = pow((double)c, 1. / 2.2);
    if (255. < v) v = 255.;            /* std::min(v,255.) = (255.<v)?255.:v ; NaN stays NaN */
    if (!(v == v)) return 0;           /* double->uchar of NaN is UB in the reference; pick 0 */
    return (uint8_t)v;
}
void or_tonemap(const float *rgba, int npix, uint8_t *out) {
    for (int p = 0; p < npix; p++)
        for (int k = 0; k < 3; k++) out[3 * p + k] = tonemap1(rgba[4 * p + k]);
}

int or_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Camera::rotate(), realtime_render.cu:823-846 (host code: float cos/sin/sqrt overloads) */
void or_camera_basis(float yaw, float pitch, float obx[3], float oby[3], float obz[3]) {
    vec bx = V(1, 0, 0), by = V(0, 1, 0), bz = V(0, 0, -1);
    const float cy = cosf(yaw), sy = sinf(yaw);
    bx = vadd(vmuls(bx, cy), vmuls(bz, sy));
    bz = cross(by, bx);
    const float cp = cosf(pitch), sp = sinf(pitch);
    by = vsub(vmuls(by, cp), vmuls(bz, sp));
    bz = cross(bx, by);
    bx = normalize(bx); by = normalize(by); bz = normalize(bz);
    for (int k = 0; k < 3; k++) { obx[k] = bx.d[k]; oby[k] = by.d[k]; obz[k] = bz.d[k]; }
}

/* WangHash, realtime_render.cu:1190-1197 */
uint32_t or_wang_hash(uint32_t a) {
    a = (a ^ 61u) ^ (a >> 16);
    a = a + (a << 3);
    a = a ^ (a >> 4);
    a = a * 0x27d4eb2du;
    a = a ^ (a >> 15);
    return a;
}

/* realtime_render.cu:1136-1147 */
void or_progressive_accumulate(float *accum, const float *frame, int npix, int framenumber, float *display, uint8_t *out_rgb8) {
    const float inv = 1.0f / (float)framenumber;                      /* cutil_math operator/(float3, float) */
    for (int p = 0; p < npix; p++) {
        for (int k = 0; k < 3; k++) {
            accum[4 * p + k] += frame[4 * p + k];
            const float c = accum[4 * p + k] * inv;
            if (display) display[4 * p + k] = c;
            if (out_rgb8) {
                double v = (double)powf(c, 1 / 2.2f);
                if (!(v < 255.)) v = 255.;
                out_rgb8[3 * p + k] = (uint8_t)v;
            }
        }
        accum[4 * p + 3] += frame[4 * p + 3];                         /* rays traced so far */
        if (display) display[4 * p + 3] = accum[4 * p + 3];
    }
}

/* main's pixel loop, cpu:693-718 */
int or_render(const or_scene *s, const or_params *p, float *out_rgba, uint8_t *out_rgb8, or_counters *cnt) {
    const int W = p->W, H = p->H;
    if (W <= 0 || H <= 0 || p->num_rays <= 0 || p->row_begin < 0 || p->row_end > H || p->row_begin > p->row_end) return -1;
    const int stride = p->stride > 1 ? p->stride : 1;
    const int ncols = (W + stride - 1) / stride;
    /* list of image rows to render */
    int *rowlist = (int *)malloc(sizeof(int) * (size_t)(p->row_end - p->row_begin + 1));
    int nrows = 0;
    for (int r = p->row_begin; r < p->row_end; r += stride) {
        if (p->tile_step > 1 && p->tile_rows > 0 && ((r - p->row_begin) / p->tile_rows) % p->tile_step != 0) continue;
        rowlist[nrows++] = r;
    }
    const float alpha = p->fov;
    /* cpu:694 `-W / (2 * tan(alpha/2))`: alpha is a compile-time constant in the reference, so g++ -O3
     * folds tan(float) with MPFR => the CORRECTLY ROUNDED binary32 tangent (0x1.279a74p-1 for pi/3/2),
     * whereas glibc's run-time tanf returns 0x1.279a76p-1.  binary64 tan narrowed to binary32 reproduces
     * the folded value (pinned by tests/golden/ref_render.npz; DESIGN.md hazard H12). */
    const float z = -W / (2 * (float)tan((double)(alpha / 2)));
    const vec Cc = V(p->cam[0], p->cam[1], p->cam[2]);
    float cbx[3] = {1, 0, 0}, cby[3] = {0, 1, 0}, cbz[3] = {0, 0, -1};
    if (p->cam_mode == 1) or_camera_basis(p->yaw, p->pitch, cbx, cby, cbz);
    const vec Bx = V(cbx[0], cbx[1], cbx[2]), By = V(cby[0], cby[1], cby[2]), Bz = V(cbz[0], cbz[1], cbz[2]);
    or_counters total = {0, 0, 0, 0, 0};
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = p->threads > 0 ? p->threads : omp_get_max_threads();
#endif
    mt19937 gen;
    mt19937 *mt = NULL;
    if (p->rng_mode == 1) { mt_seed(&gen, 0); mt = &gen; nthreads = 1; }
#pragma omp parallel num_threads(nthreads)
    {
        or_counters local = {0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 1)
        for (int ii = 0; ii < nrows; ii++) {
            const int i = rowlist[ii];
            for (int jj = 0; jj < ncols; jj++) {
                const int j = jj * stride;
                /* cpu:699: the +0.5 / -0.5 are double literals, narrowed by Vector(float,...) */
                vec u_center = V((float)((double)((float)j - (float)W / 2) + 0.5),
                                 (float)((double)((float)H / 2 - (float)i) - 0.5), z);
                if (p->cam_mode == 1)   /* realtime:1115: cam.C + cam.bz * z + cam.bx * (x - W/2 + 0.5) + cam.by * (H/2 - y - 0.5) */
                    u_center = vadd(vadd(vadd(Cc, vmuls(Bz, z)), vmuls(Bx, u_center.d[0])), vmuls(By, u_center.d[1]));
                const float inv_n = (float)(1. / p->num_rays);           /* realtime:1131 color * (1./num_rays) */
                vec color_total = V(0, 0, 0);
                uint64_t rays_before = local.rays;
                uint32_t pixel = (uint32_t)i * (uint32_t)W + (uint32_t)j;
                for (int t = 0; t < p->num_rays; t++) {
                    trace_ctx c = {p->seed, pixel, (uint32_t)t, p->eps, p->tri_tmin, &local, mt};
                    float sigma = p->sigma;
                    float r1 = rng(&c, 0, 2);                                   /* cpu:705-706 */
                    float r2 = rng(&c, 0, 3);
                    /* cpu:707: sigma*sqrt(-2*log(r1)) in float, cos/sin(2*PI*r2) in double */
                    float bm = sigma * sqrtf(-2 * logf(r1));
                    vec jit = V((float)((double)bm * cos(2 * OR_PI * (double)r2)),
                                (float)((double)bm * sin(2 * OR_PI * (double)r2)), 0);
                    vec u = normalize(vadd(u_center, jit));
                    vec color = get_color(s, R(Cc, u, 1.f), p->num_bounce, 0, &c);
                    color_total = vadd(color_total, p->cam_mode == 1 ? vmuls(color, inv_n) : color);
                }
                vec color_avg = p->cam_mode == 1 ? color_total : vdivs(color_total, (float)p->num_rays);   /* cpu:713 */
                size_t o = (size_t)ii * ncols + jj;
                if (out_rgba) {
                    out_rgba[4 * o + 0] = color_avg.d[0]; out_rgba[4 * o + 1] = color_avg.d[1];
                    out_rgba[4 * o + 2] = color_avg.d[2]; out_rgba[4 * o + 3] = (float)(local.rays - rays_before);
                }
                if (out_rgb8) {
                    out_rgb8[3 * o + 0] = tonemap1(color_avg.d[0]);
                    out_rgb8[3 * o + 1] = tonemap1(color_avg.d[1]);
                    out_rgb8[3 * o + 2] = tonemap1(color_avg.d[2]);
                }
            }
        }
#pragma omp critical
        cnt_add(&total, &local);
    }
    free(rowlist);
    if (cnt) *cnt = total;
    return 0;
}
