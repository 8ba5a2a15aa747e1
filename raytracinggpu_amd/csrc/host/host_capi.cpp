// host_capi.cpp -- C exports of include/raytracer_host.h over the C++ host API (raytracer.hpp).
#include "../../../include/raytracer_host.h"
#include "../../../include/raytracer.hpp"

using namespace raytracer;

struct rth_mesh {
    TriangleMesh mesh;
    std::vector<float> arr;
};

extern "C" {

rth_mesh *rth_mesh_new(void) { return new (std::nothrow) rth_mesh(); }
void rth_mesh_free(rth_mesh *m) { delete m; }

int rth_mesh_read_obj(rth_mesh *m, const char *path, float scale, const float offset[3]) {
    FILE *f = std::fopen(path, "r");
    const bool ok = f != nullptr;
    if (f) std::fclose(f);
    m->mesh.obj_scale = scale;
    m->mesh.obj_offset = Vector(offset[0], offset[1], offset[2]);
    m->mesh.readOBJ(path);
    std::fflush(stdout);
    return ok ? 0 : -1;
}

void rth_mesh_set_arrays(rth_mesh *m, const float *v, int nv, const int32_t *t, int nt) {
    m->mesh.vertices.clear(); m->mesh.indices.clear();
    for (int i = 0; i < nv; ++i) m->mesh.vertices.push_back(Vector(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
    for (int i = 0; i < nt; ++i) m->mesh.indices.push_back(TriangleIndices(t[3 * i], t[3 * i + 1], t[3 * i + 2]));
    m->arr.clear();
}

void rth_mesh_rescale(rth_mesh *m, float scale, const float offset[3]) {
    m->mesh.rescale(scale, Vector(offset[0], offset[1], offset[2]));
}

int rth_mesh_build_bvh(rth_mesh *m) {
    m->arr = m->mesh.buildFlatBVH();
    return (int)m->mesh.n_bvhs;
}

int rth_mesh_num_vertices(const rth_mesh *m) { return (int)m->mesh.vertices.size(); }
int rth_mesh_num_triangles(const rth_mesh *m) { return (int)m->mesh.indices.size(); }
int rth_mesh_num_nodes(const rth_mesh *m) { return (int)(m->arr.size() / 10); }
void rth_mesh_get_vertices(const rth_mesh *m, float *o) {
    for (size_t i = 0; i < m->mesh.vertices.size(); ++i)
        for (int k = 0; k < 3; ++k) o[3 * i + k] = m->mesh.vertices[i][k];
}
void rth_mesh_get_indices(const rth_mesh *m, int32_t *o) {
    if (!m->mesh.indices.empty()) std::memcpy(o, &m->mesh.indices[0], m->mesh.indices.size() * sizeof(TriangleIndices));
}
void rth_mesh_get_bvh_array(const rth_mesh *m, float *o) {
    if (!m->arr.empty()) std::memcpy(o, m->arr.data(), m->arr.size() * sizeof(float));
}
int rth_write_png(const char *path, int W, int H, const uint8_t *rgb) { return write_png(path, W, H, rgb) ? 0 : -1; }

}  // extern "C"
