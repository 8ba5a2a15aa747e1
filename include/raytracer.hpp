// raytracer.hpp -- C++17 host API of the MI355X render path (header-only).
//
// Mirrors the host-side classes a user of souhhcong/RaytracingGPU writes main() with --
// Vector, Ray, Geometry, Sphere, TriangleIndices, BoundingBox, BVH, TriangleMesh /
// TriangleMeshHost, Scene (cpu_launcher.cpp:45-165, 167-224, 315-502, 505-511, 538-543,
// 649-651; optimized.cu:293-535) -- with the same member names and semantics, so the
// reference's scene set-up code compiles against this header.  What it does NOT contain is
// the per-pixel render: Renderer hands the scene to libraytrace_hip.so through the C-ABI of
// raytrace_hip.h (no HIP headers above that boundary).
//
// All float arithmetic that feeds the device (OBJ transform, bounding boxes, centroids) is
// written one IEEE operation per source operator; compile host code with -ffp-contract=off.
#ifndef RAYTRACER_HPP
#define RAYTRACER_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "raytrace_hip.h"
#include "raytrace_rccl.h"

namespace raytracer {

constexpr double kPi = 3.14159265358979323846;   // PI, cpu:31-33
constexpr double kInf = 1e9 + 9;                 // INF, cpu:34

// ---- Vector, cpu:45-96 -------------------------------------------------------------------
class Vector {
public:
    explicit Vector(float x = 0, float y = 0, float z = 0) { data[0] = x; data[1] = y; data[2] = z; }
    float norm2() const { return data[0] * data[0] + data[1] * data[1] + data[2] * data[2]; }
    float norm() const { return std::sqrt(norm2()); }
    void normalize() { float n = norm(); data[0] /= n; data[1] /= n; data[2] /= n; }
    float operator[](int i) const { return data[i]; }
    float &operator[](int i) { return data[i]; }
    float data[3];
};
inline Vector operator+(const Vector &a, const Vector &b) { return Vector(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vector operator-(const Vector &a, const Vector &b) { return Vector(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vector operator-(const Vector &a) { return Vector(-a[0], -a[1], -a[2]); }
inline Vector operator*(float a, const Vector &b) { return Vector(a * b[0], a * b[1], a * b[2]); }
inline Vector operator*(const Vector &a, float b) { return Vector(a[0] * b, a[1] * b, a[2] * b); }
inline Vector operator*(const Vector &a, const Vector &b) { return Vector(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
inline Vector operator/(const Vector &a, float b) { return Vector(a[0] / b, a[1] / b, a[2] / b); }
inline float dot(const Vector &a, const Vector &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Vector cross(const Vector &a, const Vector &b) {
    return Vector(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

// ---- Ray, cpu:98-104 ---------------------------------------------------------------------
class Ray {
public:
    Ray(const Vector &O, const Vector &u, float refraction_index = 1.f) : O(O), u(u), refraction_index(refraction_index) {}
    Vector O, u;
    float refraction_index;
};

// ---- Geometry, cpu:106-118.  Intersection happens on the device; the host object carries
// the material and its slot in Scene::objects. ------------------------------------------------
class Geometry {
public:
    Geometry(const Vector &albedo, int id, bool mirror, float in_refraction_index, float out_refraction_index)
        : albedo(albedo), id(id), mirror(mirror), in_refraction_index(in_refraction_index), out_refraction_index(out_refraction_index) {}
    Geometry() : id(-1), mirror(false), in_refraction_index(1), out_refraction_index(1) {}
    virtual ~Geometry() = default;
    virtual bool is_mesh() const { return false; }
    Vector albedo;
    int id;
    bool mirror;
    float in_refraction_index;
    float out_refraction_index;
};

// ---- Sphere, cpu:505-511 -----------------------------------------------------------------
class Sphere : public Geometry {
public:
    Sphere(const Vector &C, float R, const Vector &albedo, bool mirror = false, float in_refraction_index = 1.f,
           float out_refraction_index = 1.f)
        : Geometry(albedo, -1, mirror, in_refraction_index, out_refraction_index), C(C), R(R) {}
    Vector C;
    float R;
};

// ---- TriangleIndices, cpu:121-129 (same 10 x int32 layout the CUDA programs copy to the device)
class TriangleIndices {
public:
    TriangleIndices(int vtxi = -1, int vtxj = -1, int vtxk = -1, int ni = -1, int nj = -1, int nk = -1, int uvi = -1,
                    int uvj = -1, int uvk = -1, int group = -1)
        : vtxi(vtxi), vtxj(vtxj), vtxk(vtxk), uvi(uvi), uvj(uvj), uvk(uvk), ni(ni), nj(nj), nk(nk), group(group) {}
    int vtxi, vtxj, vtxk;
    int uvi, uvj, uvk;
    int ni, nj, nk;
    int group;
};
static_assert(sizeof(TriangleIndices) == 40, "TriangleIndices must stay 10 x int32 (rt_mesh.index_stride = 10)");

// ---- BoundingBox, cpu:131-144 ------------------------------------------------------------
class BoundingBox {
public:
    Vector mn, mx;
    BoundingBox() : mn(Vector((float)kInf, (float)kInf, (float)kInf)), mx(Vector((float)-kInf, (float)-kInf, (float)-kInf)) {}
    void update(const Vector &v) {
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], v[k]); mx[k] = std::max(mx[k], v[k]); }
    }
};

// ---- BVH, cpu:160-165 --------------------------------------------------------------------
class BVH {
public:
    BVH() : left(nullptr), right(nullptr), triangle_start(0), triangle_end(0) {}
    ~BVH() { delete left; delete right; }
    BVH(const BVH &) = delete;
    BVH &operator=(const BVH &) = delete;
    BVH *left, *right;
    BoundingBox bb;
    int triangle_start, triangle_end;
};

// ---- TriangleMesh (cpu:167-502) == TriangleMeshHost (optimized.cu:293-535) ------------------
class TriangleMesh : public Geometry {
public:
    TriangleMesh() {}
    bool is_mesh() const override { return true; }

    // The transform readOBJ applies to xyz-only vertices while parsing (cpu:353-355 hard-codes these).
    float obj_scale = 0.8f;
    Vector obj_offset = Vector(0, -10, 0);

    // readOBJ, cpu:315-493 / optimized.cu:303-454.  Accepts `v x y z [r g b]`, `vn`, `vt`, `usemtl`
    // and `f` with v, v/vt, v/vt/vn, v//vn corners, negative (relative) indices and polygons, which
    // are fan-triangulated around the first corner.  A missing file prints "Error opening file!"
    // and leaves the mesh empty (cpu:322-325): the scene then renders without it.
    void readOBJ(const char *obj) {
        FILE *f = std::fopen(obj, "r");
        if (!f) { std::printf("Error opening file!\n"); return; }
        int curGroup = -1;
        std::string line;
        char buf[4096];
        while (std::fgets(buf, sizeof(buf), f)) {
            line.assign(buf);
            while (!line.empty() && std::strchr(" \r\t\n", line.back())) line.pop_back();
            if (line.size() < 2) continue;
            const char *s = line.c_str();
            if (s[0] == 'u' && s[1] == 's') { ++curGroup; continue; }
            if (s[0] == 'v' && s[1] == ' ') {
                float c[6];
                int n = std::sscanf(s + 2, "%f %f %f %f %f %f", &c[0], &c[1], &c[2], &c[3], &c[4], &c[5]);
                Vector v(c[0], c[1], c[2]);
                if (n == 6) {   // coloured vertex: kept as is (cpu:344-350)
                    vertexcolors.push_back(Vector(clamp01(c[3]), clamp01(c[4]), clamp01(c[5])));
                } else {
                    v = v * obj_scale + obj_offset;
                }
                vertices.push_back(v);
                bb.update(v);
            } else if (s[0] == 'v' && s[1] == 'n') {
                Vector v; std::sscanf(s + 2, "%f %f %f", &v[0], &v[1], &v[2]); normals.push_back(v);
            } else if (s[0] == 'v' && s[1] == 't') {
                Vector v; std::sscanf(s + 2, "%f %f", &v[0], &v[1]); uvs.push_back(v);
            } else if (s[0] == 'f') {
                parse_face(s + 1, curGroup);
            }
        }
        std::fclose(f);
    }

    // optimized.cu:297-301
    void rescale(float scale, Vector offset) {
        for (auto &v : vertices) v = v * scale + offset;
    }

    // cpu:180-188
    BoundingBox compute_bbox(int triangle_start, int triangle_end) const {
        BoundingBox b;
        for (int i = triangle_start; i < triangle_end; ++i) {
            b.update(vertices[indices[i].vtxi]); b.update(vertices[indices[i].vtxj]); b.update(vertices[indices[i].vtxk]);
        }
        return b;
    }

    // buildBVH, cpu:190-224 / optimized.cu:476-510: top-down, split the longest axis of the node's
    // box at its midpoint, partition `indices` in place by triangle centroid, stop when one side
    // would be empty / a single triangle or fewer than five triangles remain.
    void buildBVH(BVH *cur, int triangle_start, int triangle_end) {
        ++n_bvhs;
        cur->triangle_start = triangle_start;
        cur->triangle_end = triangle_end;
        delete cur->left; delete cur->right;
        cur->left = cur->right = nullptr;
        cur->bb = compute_bbox(triangle_start, triangle_end);
        const Vector diag = cur->bb.mx - cur->bb.mn;
        int axis = 2;
        if (diag[0] >= diag[1] && diag[0] >= diag[2]) axis = 0;
        else if (diag[1] >= diag[0] && diag[1] >= diag[2]) axis = 1;
        const float split = (cur->bb.mn[axis] + cur->bb.mx[axis]) / 2;
        int pivot = triangle_start;
        for (int i = triangle_start; i < triangle_end; ++i) {
            const TriangleIndices &t = indices[i];
            const float cen = (vertices[t.vtxi][axis] + vertices[t.vtxj][axis] + vertices[t.vtxk][axis]) / 3;
            if (cen < split) { std::swap(indices[i], indices[pivot]); ++pivot; }
        }
        if (pivot <= triangle_start || pivot >= triangle_end - 1 || triangle_end - triangle_start < 5) return;
        cur->left = new BVH;
        cur->right = new BVH;
        buildBVH(cur->left, triangle_start, pivot);
        buildBVH(cur->right, pivot, triangle_end);
    }

    // bvhTreeToArray, optimized.cu:512-534: 10 floats per node, pre-order, child index reserved
    // before descending; call with arr_size = 1 for the root at index 0.
    void bvhTreeToArray(const BVH *cur, float *arr_bvh, size_t &arr_size, size_t arr_idx = 0) const {
        float *a = arr_bvh + arr_idx * 10;
        for (int k = 0; k < 3; ++k) { a[2 + k] = cur->bb.mn[k]; a[5 + k] = cur->bb.mx[k]; }
        a[8] = (float)cur->triangle_start;
        a[9] = (float)cur->triangle_end;
        a[0] = a[1] = -1;
        if (cur->left) { const size_t l = arr_size++; a[0] = (float)l; bvhTreeToArray(cur->left, arr_bvh, arr_size, l); }
        if (cur->right) { const size_t r = arr_size++; a[1] = (float)r; bvhTreeToArray(cur->right, arr_bvh, arr_size, r); }
    }

    // convenience: (re)build the tree over all triangles and flatten it
    std::vector<float> buildFlatBVH() {
        n_bvhs = 0;
        buildBVH(&bvh, 0, (int)indices.size());
        std::vector<float> arr(n_bvhs * 10);
        size_t n = 1;
        bvhTreeToArray(&bvh, arr.data(), n);
        return arr;
    }

    std::vector<TriangleIndices> indices;
    std::vector<Vector> vertices, normals, uvs, vertexcolors;
    BoundingBox bb;
    BVH bvh;
    size_t n_bvhs = 0;

private:
    static float clamp01(float x) { return std::min(1.f, std::max(0.f, x)); }
    int resolve(long i, size_t n) const { return i < 0 ? (int)((long)n + i) : (int)(i - 1); }

    // one face line: corners "v", "v/vt", "v/vt/vn", "v//vn"
    void parse_face(const char *s, int group) {
        struct Corner { long v = 0, t = 0, n = 0; bool has_t = false, has_n = false; };
        std::vector<Corner> cs;
        while (*s) {
            while (*s == ' ' || *s == '\t') ++s;
            if (!*s) break;
            char *end;
            Corner c;
            c.v = std::strtol(s, &end, 10);
            if (end == s) { ++s; continue; }
            s = end;
            if (*s == '/') {
                ++s;
                if (*s != '/') { c.t = std::strtol(s, &end, 10); c.has_t = end != s; s = end; }
                if (*s == '/') { ++s; c.n = std::strtol(s, &end, 10); c.has_n = end != s; s = end; }
            }
            cs.push_back(c);
        }
        for (size_t k = 2; k < cs.size(); ++k) {   // fan around corner 0 (cpu:424-486)
            const Corner &a = cs[0], &b = cs[k - 1], &c = cs[k];
            TriangleIndices t;
            t.group = group;
            t.vtxi = resolve(a.v, vertices.size()); t.vtxj = resolve(b.v, vertices.size()); t.vtxk = resolve(c.v, vertices.size());
            if (a.has_t && b.has_t && c.has_t) { t.uvi = resolve(a.t, uvs.size()); t.uvj = resolve(b.t, uvs.size()); t.uvk = resolve(c.t, uvs.size()); }
            if (a.has_n && b.has_n && c.has_n) { t.ni = resolve(a.n, normals.size()); t.nj = resolve(b.n, normals.size()); t.nk = resolve(c.n, normals.size()); }
            indices.push_back(t);
        }
    }
};
using TriangleMeshHost = TriangleMesh;   // optimized.cu's name for the same host object

// ---- Scene, cpu:538-543, 649-651 -----------------------------------------------------------
class Scene {
public:
    void addObject(Geometry *s) { s->id = (int)objects.size(); objects.push_back(s); }
    std::vector<Geometry *> objects;
    float intensity = 3e10f;
    Vector L = Vector(-10.f, 20.f, 40.f);
};

struct Camera {
    Vector C = Vector(0, 0, 55);          // cpu:691
    float alpha = (float)(kPi / 3);       // cpu:666
};

// what cpu_launcher.cpp and optimized.cu hard-code next to the scene
struct RenderSettings {
    int W = 512, H = 512;                 // cpu:661-662
    int num_rays = 1, num_bounce = 0;
    int depth_convention = 0;             // 0 cpu_launcher (b+1 segments), 1 optimized.cu (b segments)
    float sigma = 0.f, eps = 1e-3f, tri_tmin = 1e-4f;
    uint32_t seed = 123456;
    int variant = RT_VARIANT_AUTO;
    static RenderSettings cpu_launcher() { return RenderSettings(); }
    static RenderSettings optimized_cu() { RenderSettings s; s.depth_convention = 1; s.sigma = 0.2f; s.eps = 1e-4f; s.tri_tmin = 0.f; return s; }
};

class Error : public std::runtime_error {
public:
    Error(int code, const std::string &what) : std::runtime_error(what), code(code) {}
    int code;
};

// ---- Renderer: Scene -> libraytrace_hip.so.  Replaces the CUDA block of optimized.cu main()
// (:794-857) and the pixel loop of cpu_launcher.cpp (:693-718). ---------------------------------
// The reference's Scene flattened into what rt_scene_upload / rt_multi_scene_upload take (the mesh as optimized.cu hands it
// to its kernel: Vector[], TriangleIndices[] in BVH order, bvhTreeToArray's float[10] nodes).
struct SceneArrays {
    std::vector<rt_sphere> sph;
    std::vector<std::vector<float>> arrs, vertss;                   // per mesh: bvhTreeToArray's nodes, the vertices as 3 floats each
    std::vector<rt_mesh> meshes;                                    // every TriangleMesh of Scene::objects, with its position (object_slot) and Geometry's fields
    rt_light lt{};
    rt_camera cm{};
    SceneArrays(const Scene &scene, const Camera &cam) {
        if (scene.objects.size() > RT_MAX_OBJECTS) throw Error(RT_ERR_INVALID, "at most " + std::to_string(RT_MAX_OBJECTS) + " objects per Scene");
        for (size_t i = 0; i < scene.objects.size(); ++i) {
            const Geometry *g = scene.objects[i];
            if (g->is_mesh()) {
                const TriangleMesh *mesh = static_cast<const TriangleMesh *>(g);
                size_t nodes = 0;
                count_nodes(&mesh->bvh, nodes);
                arrs.emplace_back(nodes * 10);
                size_t n = 1;
                mesh->bvhTreeToArray(&mesh->bvh, arrs.back().data(), n);
                vertss.emplace_back(mesh->vertices.size() * 3);
                for (size_t v = 0; v < mesh->vertices.size(); ++v)
                    for (int k = 0; k < 3; ++k) vertss.back()[3 * v + k] = mesh->vertices[v][k];
                rt_mesh m{};
                m.n_vertices = (int)mesh->vertices.size();
                m.indices = mesh->indices.empty() ? nullptr : &mesh->indices[0].vtxi;
                m.index_stride = (int)(sizeof(TriangleIndices) / sizeof(int32_t));
                m.n_triangles = (int)mesh->indices.size();
                m.n_nodes = (int)nodes;
                for (int k = 0; k < 3; ++k) m.albedo[k] = mesh->albedo[k];
                m.object_slot = (int)i;
                // Geometry's other members (cpu:113-116): getColor reads them for whichever object was hit (cpu:573-606), a mesh included
                m.mirror = mesh->mirror ? 1 : 0;
                m.in_refraction_index = mesh->in_refraction_index; m.out_refraction_index = mesh->out_refraction_index;
                meshes.push_back(m);
                continue;
            }
            const Sphere *s = static_cast<const Sphere *>(g);
            rt_sphere r;
            for (int k = 0; k < 3; ++k) { r.center[k] = s->C[k]; r.albedo[k] = s->albedo[k]; }
            r.radius = s->R; r.mirror = s->mirror ? 1 : 0;
            r.in_refraction_index = s->in_refraction_index; r.out_refraction_index = s->out_refraction_index;
            sph.push_back(r);
        }
        for (size_t k = 0; k < meshes.size(); ++k) {                // (the vectors have stopped growing: their storage stays where it is)
            meshes[k].vertices = vertss[k].data();
            meshes[k].bvh_arr10 = arrs[k].data();
        }
        lt = rt_light{{scene.L[0], scene.L[1], scene.L[2]}, scene.intensity};
        cm = rt_camera{{cam.C[0], cam.C[1], cam.C[2]}, cam.alpha};
    }
    static void count_nodes(const BVH *b, size_t &n) { ++n; if (b->left) count_nodes(b->left, n); if (b->right) count_nodes(b->right, n); }
};

// The RCCL transport of the one-process-per-GPU path (raytrace_rccl.h): every rank of the node makes one TileComm from the same
// 128-byte id (rank 0 creates it with TileComm::make_id and hands it to the others -- a file, the environment, MPI).  The library
// is loaded on first use, so a single-GPU program never loads librccl.  Two ranks of a communicator cannot share a device.
class TileComm {
public:
    struct Api {
        void *so = nullptr;
        int (*id_create)(unsigned char *) = nullptr;
        int (*create)(rt_comm **, int, int, int, const unsigned char *) = nullptr;
        int (*destroy)(rt_comm *) = nullptr;
        const char *(*last_error)(const rt_comm *) = nullptr;
        void *(*stream)(const rt_comm *) = nullptr;
        int (*gather_tiles)(rt_comm *, const void *, int, int, int, int, int, void *, void *) = nullptr;
        uint64_t (*last_bytes)(const rt_comm *) = nullptr;
        int (*sync)(rt_comm *) = nullptr;
    };
    static const Api &api() {
        static Api a = [] {
            Api x;
            const char *path = std::getenv("RT_RCCL_LIB");            // default: next to the program / on the loader's path
            x.so = dlopen(path ? path : "libraytrace_rccl.so", RTLD_NOW | RTLD_LOCAL);
            if (!x.so) throw Error(RT_ERR_UNSUPPORTED, std::string("libraytrace_rccl.so: ") + dlerror());
            auto sym = [&](const char *n) { void *f = dlsym(x.so, n); if (!f) throw Error(RT_ERR_UNSUPPORTED, std::string("libraytrace_rccl.so lacks ") + n); return f; };
            x.id_create = reinterpret_cast<decltype(x.id_create)>(sym("rt_comm_id_create"));
            x.create = reinterpret_cast<decltype(x.create)>(sym("rt_comm_create"));
            x.destroy = reinterpret_cast<decltype(x.destroy)>(sym("rt_comm_destroy"));
            x.last_error = reinterpret_cast<decltype(x.last_error)>(sym("rt_comm_last_error"));
            x.stream = reinterpret_cast<decltype(x.stream)>(sym("rt_comm_stream"));
            x.gather_tiles = reinterpret_cast<decltype(x.gather_tiles)>(sym("rt_comm_gather_tiles"));
            x.last_bytes = reinterpret_cast<decltype(x.last_bytes)>(sym("rt_comm_last_bytes"));
            x.sync = reinterpret_cast<decltype(x.sync)>(sym("rt_comm_sync"));
            return x;
        }();
        return a;
    }
    static std::vector<unsigned char> make_id() {
        std::vector<unsigned char> id(RT_COMM_ID_BYTES);
        if (int rc = api().id_create(id.data()); rc != RT_COMM_OK) throw Error(RT_ERR_HIP, std::string("rt_comm_id_create: ") + api().last_error(nullptr));
        return id;
    }
    TileComm(int device, int rank, int world, const std::vector<unsigned char> &id) : rank_(rank), world_(world) {
        if (id.size() != RT_COMM_ID_BYTES) throw Error(RT_ERR_INVALID, "a communicator id is RT_COMM_ID_BYTES bytes");
        if (int rc = api().create(&c_, device, rank, world, id.data()); rc != RT_COMM_OK) throw Error(RT_ERR_HIP, std::string("rt_comm_create: ") + api().last_error(nullptr));
    }
    ~TileComm() { if (c_) api().destroy(c_); }
    TileComm(const TileComm &) = delete;
    TileComm &operator=(const TileComm &) = delete;
    int rank() const { return rank_; }
    int world() const { return world_; }
    void *stream() const { return api().stream(c_); }
    void gather_tiles(const void *tiles_dev, int W, int H, int bytes_per_pixel, void *frame_dev_on_root, int root = 0, int tile_rows = RT_MULTI_TILE_ROWS) {
        if (int rc = api().gather_tiles(c_, tiles_dev, W, H, bytes_per_pixel, tile_rows, root, frame_dev_on_root, nullptr); rc != RT_COMM_OK)
            throw Error(RT_ERR_HIP, std::string("rt_comm_gather_tiles: ") + api().last_error(c_));
    }
    void sync() { if (int rc = api().sync(c_); rc != RT_COMM_OK) throw Error(RT_ERR_HIP, std::string("rt_comm_sync: ") + api().last_error(c_)); }
    uint64_t last_bytes() const { return api().last_bytes(c_); }

private:
    rt_comm *c_ = nullptr;
    int rank_ = 0, world_ = 1;
};

class Renderer {
public:
    explicit Renderer(int device = 0) {
        int rc = rt_ctx_create(&ctx_, device);
        if (rc != RT_OK) throw Error(rc, std::string("rt_ctx_create: ") + rt_last_error(nullptr));
    }
    ~Renderer() { rt_ctx_destroy(ctx_); }
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;

    // Copies the scene to the GPU.  Every TriangleMesh in the scene must have its BVH built (buildBVH); spheres and meshes in any number
    // and order up to RT_MAX_OBJECTS (Scene::objects, cpu:538-543), each with Geometry's material fields.
    void upload(const Scene &scene, const Camera &cam = Camera()) {
        SceneArrays a(scene, cam);
        check(rt_scene_upload_meshes(ctx_, a.sph.data(), (int)a.sph.size(), a.meshes.data(), (int)a.meshes.size(), &a.lt, &a.cm), "rt_scene_upload_meshes");
    }

    // 8-bit interleaved RGB image, W*H*3 bytes, exactly what cpu:714-716 stores
    std::vector<unsigned char> render_rgb8(const RenderSettings &s) {
        rt_params p = params(s);
        std::vector<unsigned char> image((size_t)s.W * s.H * 3);
        check(rt_render_rgb8(ctx_, &p, 0, s.H, image.data()), "rt_render_rgb8");
        return image;
    }
    // linear float4 framebuffer (colour average, rays traced)
    std::vector<float> render_float(const RenderSettings &s) {
        rt_params p = params(s);
        std::vector<float> fb((size_t)s.W * s.H * 4);
        check(rt_render(ctx_, &p, 0, s.H, fb.data()), "rt_render");
        return fb;
    }
    rt_stats stats() { rt_stats st{}; check(rt_get_stats(ctx_, &st), "rt_get_stats"); return st; }
    rt_ctx *handle() { return ctx_; }

    // One process per GPU: rank `rank` of `world` owns the 8-row tiles rank, rank + world, ... of the frame.
    // tile_rows_of() lists the image rows of its dense tile buffer; render_tiles_rgb8() renders exactly those rows
    // (rt_render_device with an interleaved rt_rows, tonemap on the device) and returns them as RGB8.  The exchange between
    // the processes is the caller's (an RCCL gather over xGMI, MPI, files: INTEGRATION.md); assemble_tiles() is its inverse.
    static std::vector<int> tile_rows_of(int H, int rank, int world, int tile_rows = RT_MULTI_TILE_ROWS) {
        std::vector<int> rows;
        for (int t = rank; t * tile_rows < H; t += world)
            for (int r = t * tile_rows; r < std::min(H, (t + 1) * tile_rows); ++r) rows.push_back(r);
        return rows;
    }
    std::vector<unsigned char> render_tiles_rgb8(const RenderSettings &s, int rank, int world, int tile_rows = RT_MULTI_TILE_ROWS) {
        rt_params p = params(s);
        const int n = (int)tile_rows_of(s.H, rank, world, tile_rows).size();
        std::vector<unsigned char> img((size_t)n * s.W * 3);
        if (n == 0) return img;
        DeviceBuffer rgba(ctx_, (size_t)n * s.W * 16), rgb8(ctx_, (size_t)n * s.W * 3 + 16);
        rt_rows rows{rank * tile_rows, n, tile_rows, world};
        check(rt_render_device(ctx_, &p, &rows, rgba.p, nullptr), "rt_render_device");
        check(rt_tonemap_device(ctx_, rgba.p, (int64_t)n * s.W, rgb8.p, nullptr), "rt_tonemap_device");
        check(rt_synchronize(ctx_), "rt_synchronize");
        check(rt_device_to_host(ctx_, img.data(), rgb8.p, img.size()), "rt_device_to_host");
        return img;
    }
    // The same with the exchange done here: this rank's tiles are rendered and tone-mapped on the communicator's stream, one RCCL
    // gather moves every rank's tiles into the root's frame in device memory (each tile straight into its place), and the root
    // returns the H x W RGB8 image -- the bytes render_rgb8 gives on one GPU.  Other ranks return an empty vector.  Collective.
    std::vector<unsigned char> render_gather_rgb8(const RenderSettings &s, TileComm &comm, int root = 0, int tile_rows = RT_MULTI_TILE_ROWS) {
        rt_params p = params(s);
        const int rank = comm.rank(), world = comm.world();
        const int n = (int)tile_rows_of(s.H, rank, world, tile_rows).size();
        DeviceBuffer rgba(ctx_, (size_t)std::max(n, 1) * s.W * 16), rgb8(ctx_, (size_t)std::max(n, 1) * s.W * 3 + 16);
        std::unique_ptr<DeviceBuffer> frame;
        if (rank == root) frame = std::make_unique<DeviceBuffer>(ctx_, (size_t)s.H * s.W * 3 + 16);
        if (n > 0) {
            rt_rows rows{rank * tile_rows, n, tile_rows, world};
            check(rt_render_device(ctx_, &p, &rows, rgba.p, comm.stream()), "rt_render_device");
            check(rt_tonemap_device(ctx_, rgba.p, (int64_t)n * s.W, rgb8.p, comm.stream()), "rt_tonemap_device");
        }
        comm.gather_tiles(rgb8.p, s.W, s.H, 3, frame ? frame->p : nullptr, root, tile_rows);
        comm.sync();
        std::vector<unsigned char> img;
        if (rank == root) {
            img.resize((size_t)s.H * s.W * 3);
            check(rt_device_to_host(ctx_, img.data(), frame->p, img.size()), "rt_device_to_host");
        }
        return img;
    }
    static void assemble_tiles(std::vector<unsigned char> &frame, const std::vector<unsigned char> &tiles, int W, int H, int rank, int world,
                               int tile_rows = RT_MULTI_TILE_ROWS) {
        const std::vector<int> rows = tile_rows_of(H, rank, world, tile_rows);
        for (size_t k = 0; k < rows.size(); ++k) std::copy(tiles.begin() + k * (size_t)W * 3, tiles.begin() + (k + 1) * (size_t)W * 3, frame.begin() + (size_t)rows[k] * W * 3);
    }

    // --- the pieces of realtime_render.cu / global_launcher.cu behind the same boundary ---
    // transformMesh (global_launcher.cu:932-946) on the uploaded mesh, then triangle precompute + BVH refit on the device
    void transform_mesh(const float rotation[9], const Vector &translation) {
        const float t[3] = {translation[0], translation[1], translation[2]};
        check(rt_mesh_transform(ctx_, rotation, t), "rt_mesh_transform");
    }
    // get_smooth_normal (realtime_render.cu:221-245): the mesh's `normals` and its TriangleIndices::ni,nj,nk (all >= 0)
    void use_smooth_normals(const TriangleMesh &mesh) {
        std::vector<float> n(mesh.normals.size() * 3);
        for (size_t i = 0; i < mesh.normals.size(); ++i)
            for (int k = 0; k < 3; ++k) n[3 * i + k] = mesh.normals[i][k];
        check(rt_mesh_set_normals(ctx_, n.data(), (int)mesh.normals.size(), mesh.indices.empty() ? nullptr : &mesh.indices[0].ni,
                                  (int)(sizeof(TriangleIndices) / sizeof(int32_t)), (int)mesh.indices.size()), "rt_mesh_set_normals");
    }
    void use_flat_normals() { check(rt_mesh_set_normals(ctx_, nullptr, 0, nullptr, 3, 0), "rt_mesh_set_normals"); }
    // Camera{C, yaw, pitch} (realtime_render.cu:803-861) and disp() (:1243-1290) without the window: one accumulated frame
    std::vector<unsigned char> progressive_frame(const RenderSettings &s, const rt_camera_pose &pose, std::vector<float> *display = nullptr) {
        rt_params p = params(s);
        std::vector<unsigned char> image((size_t)s.W * s.H * 3);
        if (display) display->resize((size_t)s.W * s.H * 4);
        check(rt_progressive_frame(ctx_, &p, &pose, display ? display->data() : nullptr, image.data()), "rt_progressive_frame");
        return image;
    }
    void progressive_reset() { check(rt_progressive_reset(ctx_), "rt_progressive_reset"); }   // buffer_reset

    static rt_params params(const RenderSettings &s) {
        rt_params p{};
        p.width = s.W; p.height = s.H; p.num_rays = s.num_rays; p.num_bounce = s.num_bounce;
        p.depth_convention = s.depth_convention; p.sigma = s.sigma; p.eps = s.eps; p.tri_tmin = s.tri_tmin;
        p.seed = s.seed; p.variant = s.variant;
        return p;
    }

private:
    struct DeviceBuffer {                  // device memory through the C-ABI (no HIP above it)
        void *p = nullptr;
        DeviceBuffer(rt_ctx *c, size_t bytes) { if (rt_device_alloc(c, &p, bytes) != RT_OK) throw Error(RT_ERR_HIP, std::string("rt_device_alloc: ") + rt_last_error(c)); }
        ~DeviceBuffer() { rt_device_free(p); }
        DeviceBuffer(const DeviceBuffer &) = delete;
        DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    };
    void check(int rc, const char *what) { if (rc != RT_OK) throw Error(rc, std::string(what) + ": " + rt_last_error(ctx_)); }
    rt_ctx *ctx_ = nullptr;
};

// Several GPUs from ONE host process (rt_multi_*): interleaved 8-row tiles, tile k -> devices[k mod n], the frame assembled
// on devices[0].  Same bits as Renderer.
class MultiRenderer {
public:
    explicit MultiRenderer(const std::vector<int> &devices) {
        int rc = rt_multi_create(&m_, devices.data(), (int)devices.size());
        if (rc != RT_OK) throw Error(rc, std::string("rt_multi_create: ") + rt_multi_last_error(nullptr));
    }
    ~MultiRenderer() { rt_multi_destroy(m_); }
    MultiRenderer(const MultiRenderer &) = delete;
    MultiRenderer &operator=(const MultiRenderer &) = delete;
    void upload(const Scene &scene, const Camera &cam = Camera()) {
        SceneArrays a(scene, cam);
        check(rt_multi_scene_upload_meshes(m_, a.sph.data(), (int)a.sph.size(), a.meshes.data(), (int)a.meshes.size(), &a.lt, &a.cm), "rt_multi_scene_upload_meshes");
    }
    std::vector<float> render_float(const RenderSettings &s) {
        rt_params p = Renderer::params(s);
        std::vector<float> fb((size_t)s.W * s.H * 4);
        check(rt_render_multi(m_, &p, fb.data()), "rt_render_multi");
        return fb;
    }
    // the PNG path: tonemap on every device, 8-bit tiles exchanged (what stbi_write_png gets, cpu:719)
    std::vector<unsigned char> render_rgb8(const RenderSettings &s) {
        rt_params p = Renderer::params(s);
        std::vector<unsigned char> img((size_t)s.W * s.H * 3);
        check(rt_render_multi_rgb8(m_, &p, img.data()), "rt_render_multi_rgb8");
        return img;
    }
    rt_multi_stats stats() { rt_multi_stats st{}; check(rt_multi_get_stats(m_, &st), "rt_multi_get_stats"); return st; }
private:
    void check(int rc, const char *what) { if (rc != RT_OK) throw Error(rc, std::string(what) + ": " + rt_multi_last_error(m_)); }
    rt_multi *m_ = nullptr;
};

// ---- PNG output: 8-bit RGB, what stbi_write_png(name, W, H, 3, data, 0) produces for the
// reference (cpu:719).  zlib-deflated, filter 0 on every row.  Returns false on I/O error. --------
bool write_png(const char *path, int W, int H, const unsigned char *rgb);

}  // namespace raytracer
#endif
