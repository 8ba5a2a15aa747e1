// launcher_main.cpp -- `rt_launcher num_rays num_bounce`: the reference's CLI (cpu_launcher.cpp:654-725,
// optimized.cu:774-884) on the MI355X render path.  Same two positional arguments, same usage message and
// exit code on a wrong argument count, same hard-coded scene, 512x512, `Rendering time: X s` on stdout and
// an 8-bit RGB PNG in the working directory.  Everything the reference fixes at compile time is an
// optional flag after the two positionals:
//   --program cpu|optimized   scene/constants of cpu_launcher.cpp (default) or optimized.cu
//   --scene cat|spheres|demo10   --width W --height H --out FILE --obj FILE --device N --variant N
//   --devices 0,1,2,...          several GPUs from this one process (interleaved row tiles, rt_render_multi)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <iostream>
#include <sstream>
#include <string>

#include "../../../include/raytracer.hpp"

using namespace raytracer;

int main(int argc, char *argv[]) {
    if (argc < 3 || argv[1][0] == '-' || argv[2][0] == '-' || (argc - 3) % 2 != 0) {
        std::cout << "Invalid number of arguments!\nThe first argument is number of rays and the second argument is number of bounces.";
        return 0;   // cpu:655-658
    }
    const int num_rays = atoi(argv[1]), num_bounce = atoi(argv[2]);
    auto start_time = std::chrono::system_clock::now();
    std::string program = "cpu", scene_name = "cat", out, obj = "cadnav.com_model/Models_F0202A090/cat.obj";
    int W = 512, H = 512, device = 0, variant = RT_VARIANT_AUTO;
    std::vector<int> devices;
    for (int i = 3; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--program") program = v;
        else if (k == "--scene") scene_name = v;
        else if (k == "--width") W = atoi(v.c_str());
        else if (k == "--height") H = atoi(v.c_str());
        else if (k == "--out") out = v;
        else if (k == "--obj") obj = v;
        else if (k == "--device") device = atoi(v.c_str());
        else if (k == "--variant") variant = atoi(v.c_str());
        else if (k == "--devices") { std::stringstream ss(v); std::string tok; while (std::getline(ss, tok, ',')) devices.push_back(atoi(tok.c_str())); }
        else { std::cerr << "unknown option " << k << "\n"; return 2; }
    }
    const bool optimized = program == "optimized";
    if (out.empty()) out = optimized ? "image_optimized.png" : "image.png";   // opt:862 / cpu:719

    try {
        Scene s;
        TriangleMesh *mesh_ptr = nullptr;
        if (scene_name == "cat") {
            mesh_ptr = new TriangleMesh();
            mesh_ptr->readOBJ(obj.c_str());                               // cpu:682
            if (optimized) mesh_ptr->rescale(0.6f, Vector(0.f, -4.f, 0.f));   // opt:804
            mesh_ptr->albedo = Vector(0.25, 0.25, 0.25);                  // cpu:683
            mesh_ptr->buildBVH(&(mesh_ptr->bvh), 0, (int)mesh_ptr->indices.size());   // cpu:684
        }
        if (scene_name == "demo10") {                                     // the commented objects, cpu:669-672
            s.addObject(new Sphere(Vector(0, 0, 0), 10, Vector(0., 0., 0.), 0, 1.5, 1));
            s.addObject(new Sphere(Vector(-20, 0, 0), 10, Vector(0., 0., 0.), 1));
            s.addObject(new Sphere(Vector(20, 0, 0), 9, Vector(0., 0., 0.), 0, 1, 1.5));
            s.addObject(new Sphere(Vector(20, 0, 0), 10, Vector(0., 0., 0.), 0, 1.5, 1));
        }
        s.addObject(new Sphere(Vector(0, 0, -1000), 940, Vector(0., 1., 0.)));     // cpu:673
        if (mesh_ptr && optimized) s.addObject(mesh_ptr);                          // opt:690-700: mesh is object 1
        s.addObject(new Sphere(Vector(0, -1000, 0), 990, Vector(0., 0., 1.)));
        s.addObject(new Sphere(Vector(0, 1000, 0), 940, Vector(1., 0., 0.)));
        s.addObject(new Sphere(Vector(-1000, 0, 0), 940, Vector(0., 1., 1.)));
        s.addObject(new Sphere(Vector(1000, 0, 0), 940, Vector(1., 1., 0.)));
        s.addObject(new Sphere(Vector(0, 0, 1000), 940, Vector(1., 0., 1.)));
        if (mesh_ptr && !optimized) s.addObject(mesh_ptr);                         // cpu:685: mesh is object 6

        RenderSettings rs = optimized ? RenderSettings::optimized_cu() : RenderSettings::cpu_launcher();
        rs.W = W; rs.H = H; rs.num_rays = num_rays; rs.num_bounce = num_bounce; rs.variant = variant;
        std::vector<unsigned char> image;
        if (!devices.empty()) {                                       // several GPUs, one process
            MultiRenderer renderer(devices);
            renderer.upload(s);
            const std::vector<float> fb = renderer.render_float(rs);
            image.resize((size_t)W * H * 3);
            for (size_t px = 0; px < (size_t)W * H; ++px)
                for (int k = 0; k < 3; ++k) image[3 * px + k] = (unsigned char)std::min(std::pow((double)fb[4 * px + k], 1. / 2.2), 255.);   // cpu:714-716
            if (!write_png(out.c_str(), W, H, image.data())) { std::cerr << "cannot write " << out << "\n"; return 1; }
            const rt_multi_stats st = renderer.stats();
            std::chrono::duration<float> run_time = std::chrono::system_clock::now() - start_time;
            std::cout << "Rendering time: " << run_time.count() << " s\n";
            std::cerr << st.n_devices << " devices, frame " << st.frame_ms << " ms, gather " << st.gather_ms << " ms, " << st.rays << " rays\n";
            return 0;
        }
        Renderer renderer(device);
        renderer.upload(s);
        image = renderer.render_rgb8(rs);
        if (!write_png(out.c_str(), W, H, image.data())) { std::cerr << "cannot write " << out << "\n"; return 1; }
        const rt_stats st = renderer.stats();
        auto end_time = std::chrono::system_clock::now();
        std::chrono::duration<float> run_time = end_time - start_time;
        std::cout << "Rendering time: " << run_time.count() << " s\n";
        std::cerr << "kernel " << st.kernel_ms << " ms, tonemap " << st.tonemap_ms << " ms, variant " << st.variant << "\n";
    } catch (const Error &e) {
        std::cerr << "error " << e.code << ": " << e.what() << "\n";
        return 1;
    }
    return 0;
}
