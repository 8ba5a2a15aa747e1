/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE (fixture generator), NOT PRODUCT CODE.
 *
 * Compiles the REAL reference translation unit /root/reference/cpu_launcher.cpp
 * (included below by path, never copied) and calls its own classes --
 * Sphere::intersect, BoundingBox::intersect, TriangleMesh::{readOBJ,buildBVH,
 * moller_trumbore,intersect}, Scene::{intersect_all,getColor} -- to dump golden
 * vectors as raw little-endian arrays.  oracle/make_golden.py packs them into
 * tests/golden/*.npz.  Built only where /root/reference exists
 * (oracle/Makefile target "ref"), output in oracle/_ref/.
 *
 * Determinism shim: the reference seeds its thread-local std::mt19937 with
 * clock()+thread (cpu_launcher.cpp:533).  All system headers that declare
 * clock() are included first, then clock() is defined to 0 for the reference's
 * text only, so a single-threaded run draws the mt19937(0) stream.  The oracle's
 * "mt" RNG mode replays that stream, which pins the stochastic (num_bounce>=1)
 * branches of Scene::getColor bit-for-bit against the reference.
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <random>
#include <stack>
#include <string>
#include <thread>
#include <vector>
#include <math.h>
#include <stdio.h>
#include <time.h>
#include <omp.h>

#define clock() ((clock_t)0)
#define main ref_main
#include "cpu_launcher.cpp"   /* -I/root/reference */
#undef main
#undef clock

static void dump(const std::string &path, const void *p, size_t bytes) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    if (bytes && fwrite(p, 1, bytes, f) != bytes) { perror("fwrite"); exit(2); }
    fclose(f);
}

/* flatten the reference's pointer BVH in the node order and 10-float layout of
 * optimized.cu:512-534 (root 0, index taken before descending) */
static void flatten(const BVH *cur, std::vector<float> &arr, size_t &n, size_t idx) {
    if (arr.size() < (idx + 1) * 10) arr.resize((idx + 1) * 10);
    arr[idx * 10 + 2] = cur->bb.mn[0]; arr[idx * 10 + 3] = cur->bb.mn[1]; arr[idx * 10 + 4] = cur->bb.mn[2];
    arr[idx * 10 + 5] = cur->bb.mx[0]; arr[idx * 10 + 6] = cur->bb.mx[1]; arr[idx * 10 + 7] = cur->bb.mx[2];
    arr[idx * 10 + 8] = cur->triangle_start; arr[idx * 10 + 9] = cur->triangle_end;
    if (cur->left) { size_t l = n++; arr.resize(std::max(arr.size(), n * 10)); arr[idx * 10 + 0] = l; flatten(cur->left, arr, n, l); }
    else arr[idx * 10 + 0] = -1;
    if (cur->right) { size_t r = n++; arr.resize(std::max(arr.size(), n * 10)); arr[idx * 10 + 1] = r; flatten(cur->right, arr, n, r); }
    else arr[idx * 10 + 1] = -1;
}

static const char *OBJ_PATH = "cadnav.com_model/Models_F0202A090/cat.obj";   /* cpu_launcher.cpp:681 */

/* the six wall spheres of cpu_launcher.cpp:673-678 as data: centre, radius, albedo, in the reference's order */
static const float WALLS[6][7] = {{0, 0, -1000, 940, 0, 1, 0}, {0, -1000, 0, 990, 0, 0, 1}, {0, 1000, 0, 940, 1, 0, 0},
                                  {-1000, 0, 0, 940, 0, 1, 1}, {1000, 0, 0, 940, 1, 1, 0}, {0, 0, 1000, 940, 1, 0, 1}};
static void add_wall(Scene &s, int k) {
    const float *w = WALLS[k];
    s.addObject(new Sphere(Vector(w[0], w[1], w[2]), w[3], Vector(w[4], w[5], w[6])));
}
static void add_walls(Scene &s, int first = 0, int last = 6) { for (int k = first; k < last; ++k) add_wall(s, k); }
static void add_demo(Scene &s) {    /* the commented-out objects, cpu_launcher.cpp:669-672 */
    s.addObject(new Sphere(Vector(0, 0, 0), 10, Vector(0., 0., 0.), 0, 1.5, 1));
    s.addObject(new Sphere(Vector(-20, 0, 0), 10, Vector(0., 0., 0.), 1));
    s.addObject(new Sphere(Vector(20, 0, 0), 9, Vector(0., 0., 0.), 0, 1, 1.5));
    s.addObject(new Sphere(Vector(20, 0, 0), 10, Vector(0., 0., 0.), 0, 1.5, 1));
}
static TriangleMesh *load_cat() {   /* cpu_launcher.cpp:680-684 */
    TriangleMesh *mesh_ptr = new TriangleMesh();
    mesh_ptr->readOBJ(OBJ_PATH);
    mesh_ptr->albedo = Vector(0.25, 0.25, 0.25);
    mesh_ptr->buildBVH(&(mesh_ptr->bvh), 0, mesh_ptr->indices.size());
    return mesh_ptr;
}

/* Scenes the reference's CLASSES accept but its main() never builds: a TriangleMesh is a Geometry, so its public mirror /
 * refraction members are read by getColor like a sphere's (cpu_launcher.cpp:573-606), and Scene::objects takes any number of meshes at any
 * positions (cpu_launcher.cpp:538-564).  The second cat is the first one at half size, moved in front of it: v * 0.5f + (16, -5, 20), applied
 * to the vertices readOBJ left (float operators of the reference's Vector), before its own buildBVH. */
static TriangleMesh *load_cat2() {
    TriangleMesh *m = new TriangleMesh();
    m->readOBJ(OBJ_PATH);
    for (auto &v : m->vertices) v = v * 0.5f + Vector(16, -5, 20);
    m->albedo = Vector(0.6, 0.3, 0.1);
    m->buildBVH(&(m->bvh), 0, m->indices.size());
    return m;
}
static bool add_material_scene(Scene &s, const std::string &scene) {
    if (scene == "cpu_mirror") { add_walls(s); TriangleMesh *m = load_cat(); m->mirror = 1; s.addObject(m); return true; }
    if (scene == "cpu_glass") { add_walls(s); TriangleMesh *m = load_cat(); m->in_refraction_index = 1.5; m->out_refraction_index = 1; s.addObject(m); return true; }
    if (scene == "two_cats") {          /* objects: 3 walls, cat (diffuse), 3 walls, second cat (mirror) -- a mesh in the MIDDLE of the order and one at the end */
        add_walls(s, 0, 3);
        s.addObject(load_cat());
        add_walls(s, 3, 6);
        TriangleMesh *b = load_cat2(); b->mirror = 1; s.addObject(b);
        return true;
    }
    if (scene == "two_cats_diffuse") {  /* both diffuse, the SAME geometry twice at positions 0 and 7: every triangle hit is an exact tie, the earlier object wins (cpu:554) */
        TriangleMesh *a = load_cat(); a->albedo = Vector(0.9, 0.1, 0.1); s.addObject(a);
        add_walls(s);
        TriangleMesh *b = load_cat(); b->albedo = Vector(0.1, 0.9, 0.1); s.addObject(b);
        return true;
    }
    return false;
}

static int cmd_mesh(const std::string &out) {
    TriangleMesh *m = new TriangleMesh();
    m->readOBJ(OBJ_PATH);
    std::vector<float> v; std::vector<int> t;
    for (auto &x : m->vertices) { v.push_back(x[0]); v.push_back(x[1]); v.push_back(x[2]); }
    for (auto &x : m->indices) { t.push_back(x.vtxi); t.push_back(x.vtxj); t.push_back(x.vtxk); }
    dump(out + "/vertices.f32", v.data(), v.size() * 4);
    dump(out + "/tri_obj_order.i32", t.data(), t.size() * 4);
    m->buildBVH(&(m->bvh), 0, m->indices.size());
    t.clear();
    for (auto &x : m->indices) { t.push_back(x.vtxi); t.push_back(x.vtxj); t.push_back(x.vtxk); }
    dump(out + "/tri_bvh_order.i32", t.data(), t.size() * 4);
    std::vector<float> arr; size_t n = 1;
    flatten(&m->bvh, arr, n, 0);
    arr.resize(n * 10);
    dump(out + "/bvh_arr10.f32", arr.data(), arr.size() * 4);
    printf("mesh: %zu vertices %zu triangles %zu nodes\n", m->vertices.size(), m->indices.size(), n);
    return 0;
}

/* primitive known-answer tests on inputs read from files written by make_golden.py */
static std::vector<float> slurp(const std::string &p) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) { perror(p.c_str()); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<float> v(n / 4);
    if (n && fread(v.data(), 1, n, f) != (size_t)n) { perror("fread"); exit(2); }
    fclose(f);
    return v;
}

static int cmd_kat(const std::string &dir) {
    {   /* spheres: rows of [C(3) R O(3) u(3)] -> [hit t N(3)] */
        std::vector<float> in = slurp(dir + "/kat_sphere_in.f32"), out;
        for (size_t i = 0; i + 10 <= in.size(); i += 10) {
            Sphere s(Vector(in[i], in[i + 1], in[i + 2]), in[i + 3], Vector(1, 1, 1));
            Ray r(Vector(in[i + 4], in[i + 5], in[i + 6]), Vector(in[i + 7], in[i + 8], in[i + 9]));
            float t = 0; Vector N(0, 0, 0);
            bool hit = s.intersect(r, t, N);
            out.push_back(hit); out.push_back(hit ? t : 0);
            for (int k = 0; k < 3; k++) out.push_back(hit ? N[k] : 0);
        }
        dump(dir + "/kat_sphere_out.f32", out.data(), out.size() * 4);
    }
    {   /* boxes: rows of [mn(3) mx(3) O(3) u(3)] -> [hit] */
        std::vector<float> in = slurp(dir + "/kat_box_in.f32"), out;
        for (size_t i = 0; i + 12 <= in.size(); i += 12) {
            BoundingBox b; b.mn = Vector(in[i], in[i + 1], in[i + 2]); b.mx = Vector(in[i + 3], in[i + 4], in[i + 5]);
            Ray r(Vector(in[i + 6], in[i + 7], in[i + 8]), Vector(in[i + 9], in[i + 10], in[i + 11]));
            float t = 0;
            out.push_back(b.intersect(r, t));
        }
        dump(dir + "/kat_box_out.f32", out.data(), out.size() * 4);
    }
    {   /* triangles: rows of [A(3) B(3) C(3) O(3) u(3)] -> [hit t N(3)] (N unnormalised, always written) */
        std::vector<float> in = slurp(dir + "/kat_tri_in.f32"), out;
        TriangleMesh m;
        for (size_t i = 0; i + 15 <= in.size(); i += 15) {
            Vector A(in[i], in[i + 1], in[i + 2]), B(in[i + 3], in[i + 4], in[i + 5]), C(in[i + 6], in[i + 7], in[i + 8]);
            Ray r(Vector(in[i + 9], in[i + 10], in[i + 11]), Vector(in[i + 12], in[i + 13], in[i + 14]));
            float t = 0; Vector N(0, 0, 0);
            bool hit = m.moller_trumbore(A, B, C, N, r, t);
            out.push_back(hit); out.push_back(hit ? t : 0);
            for (int k = 0; k < 3; k++) out.push_back(N[k]);
        }
        dump(dir + "/kat_tri_out.f32", out.data(), out.size() * 4);
    }
    {   /* whole-mesh intersect on the cat: rows of [O(3) u(3)] -> [hit t N(3)];
           hit is restated as t<1e9f (the reference's own flag is always true) */
        std::vector<float> in = slurp(dir + "/kat_mesh_in.f32"), out;
        TriangleMesh *m = load_cat();
        for (size_t i = 0; i + 6 <= in.size(); i += 6) {
            Ray r(Vector(in[i], in[i + 1], in[i + 2]), Vector(in[i + 3], in[i + 4], in[i + 5]));
            float t = 0; Vector N(0, 0, 0);
            bool ok = m->intersect(r, t, N);
            bool hit = ok && t < 1e9f;
            out.push_back(hit); out.push_back(hit ? t : 0);
            for (int k = 0; k < 3; k++) out.push_back(hit ? N[k] : 0);
        }
        dump(dir + "/kat_mesh_out.f32", out.data(), out.size() * 4);
    }
    return 0;
}

/* Single-threaded copy of main's pixel loop (cpu_launcher.cpp:693-713) for an
 * arbitrary W,H and pixel stride, calling the reference's Scene::getColor.
 * Dumps the linear colour average (before gamma) and the primary hit record. */
static int cmd_render(const std::string &scene, int W, int H, int num_rays, int num_bounce, int stride, const std::string &out) {
    float alpha = PI / 3;
    Scene s;
    if (scene == "demo10") { add_demo(s); add_walls(s); }
    else if (scene == "spheres") { add_walls(s); }
    else if (scene == "cpu") { add_walls(s); s.addObject(load_cat()); }
    else if (add_material_scene(s, scene)) { }
    else { fprintf(stderr, "unknown scene\n"); return 2; }
    Vector C(0, 0, 55);
    float z = -W / (2 * tan(alpha / 2));
    std::vector<float> col, hitrec;
    for (int i = 0; i < H; i += stride) {
        for (int j = 0; j < W; j += stride) {
            unsigned int seed = omp_get_thread_num();
            Vector u_center((float)j - (float)W / 2 + 0.5, (float)H / 2 - i - 0.5, z);
            {   /* primary hit record, no RNG involved */
                Vector u = u_center; u.normalize();
                Vector P, N; int id = -1;
                bool inter = s.intersect_all(Ray(C, u), P, N, id);
                hitrec.push_back(inter ? id : -1);
                for (int k = 0; k < 3; k++) hitrec.push_back(P[k]);
                for (int k = 0; k < 3; k++) hitrec.push_back(inter ? N[k] : 0);
            }
            Vector color_total(0, 0, 0);
            for (int t = 0; t < num_rays; t++) {
                float sigma = 0;
                float r1 = uniform(seed);
                float r2 = uniform(seed);
                Vector u = u_center + Vector(sigma * sqrt(-2 * log(r1)) * cos(2 * PI * r2), sigma * sqrt(-2 * log(r1)) * sin(2 * PI * r2), 0);
                u.normalize();
                Ray r(C, u);
                Vector color = s.getColor(r, num_bounce);
                color_total = color_total + color;
            }
            Vector color_avg = color_total / num_rays;
            for (int k = 0; k < 3; k++) col.push_back(color_avg[k]);
        }
    }
    dump(out + ".color.f32", col.data(), col.size() * 4);
    dump(out + ".hit.f32", hitrec.data(), hitrec.size() * 4);
    return 0;
}

/* multi-threaded mean image of the stochastic estimator (statistical golden).
 * Uses the reference's own RNG (not reproducible sample-by-sample; mean and per-pixel
 * standard error are what is kept). */
static int cmd_stat(int W, int H, int num_rays, int num_bounce, int stride, const std::string &out) {
    float alpha = PI / 3;
    Scene s; add_walls(s); s.addObject(load_cat());
    Vector C(0, 0, 55);
    float z = -W / (2 * tan(alpha / 2));
    int nh = (H + stride - 1) / stride, nw = (W + stride - 1) / stride;
    std::vector<float> mean(nh * nw * 3), sem(nh * nw * 3);
    #pragma omp parallel for schedule(dynamic, 1)
    for (int ii = 0; ii < nh; ii++) {
        for (int jj = 0; jj < nw; jj++) {
            int i = ii * stride, j = jj * stride;
            unsigned int seed = omp_get_thread_num();
            Vector u_center((float)j - (float)W / 2 + 0.5, (float)H / 2 - i - 0.5, z);
            double sum[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
            for (int t = 0; t < num_rays; t++) {
                Vector u = u_center; u.normalize();
                Vector color = s.getColor(Ray(C, u), num_bounce);
                for (int k = 0; k < 3; k++) { sum[k] += color[k]; sq[k] += (double)color[k] * color[k]; }
            }
            for (int k = 0; k < 3; k++) {
                double m = sum[k] / num_rays, var = sq[k] / num_rays - m * m;
                mean[(ii * nw + jj) * 3 + k] = m;
                sem[(ii * nw + jj) * 3 + k] = std::sqrt(std::max(var, 0.0) / num_rays);
            }
        }
    }
    dump(out + ".mean.f32", mean.data(), mean.size() * 4);
    dump(out + ".sem.f32", sem.data(), sem.size() * 4);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_harness mesh|kat|render|stat ...\n"); return 2; }
    std::string cmd = argv[1];
    if (cmd == "mesh" && argc == 3) return cmd_mesh(argv[2]);
    if (cmd == "kat" && argc == 3) return cmd_kat(argv[2]);
    if (cmd == "render" && argc == 9)
        return cmd_render(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), argv[8]);
    if (cmd == "stat" && argc == 8)
        return cmd_stat(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), argv[7]);
    fprintf(stderr, "bad arguments\n");
    return 2;
}
