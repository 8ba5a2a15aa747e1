// launcher_main.cpp -- `rt_launcher num_rays num_bounce`: the reference's CLI (cpu_launcher.cpp:654-725,
// optimized.cu:774-884) on the MI355X render path.  Same two positional arguments, same usage message and
// exit code on a wrong argument count, same hard-coded scene, 512x512, `Rendering time: X s` on stdout and
// an 8-bit RGB PNG in the working directory.  Everything the reference fixes at compile time is an
// optional flag after the two positionals:
//   --program cpu|optimized   scene/constants of cpu_launcher.cpp (default) or optimized.cu
//   --scene cat|spheres|demo10   --width W --height H --out FILE --obj FILE --device N --variant N
//   --mesh-material mirror|glass   --second-cat 1      (scenes the reference's classes accept and its main() never builds: cpu:106-118, 538-606)
//   --devices 0,1,2,...          several GPUs from this one process (interleaved row tiles, rt_render_multi_rgb8)
//   --tile-rank R --tile-world G --tiles FILE   one process per GPU: render only rank R's interleaved 8-row tiles and write
//                                them (raw RGB8) to FILE; the processes' exchange is the caller's (RCCL / MPI / files)
//   --tile-rank R --tile-world G --rccl-id FILE   one process per GPU with the exchange done here: rank 0 writes a fresh RCCL
//                                communicator id to FILE (the others wait for it), every rank renders its tiles on --device and
//                                one RCCL gather over xGMI puts them into rank 0's frame; rank 0 writes the PNG
//       [--rccl-nonce S]         what ties FILE to THIS launch (default: the parent process id; env RT_RCCL_NONCE): a file left by another
//                                run is never accepted; rank 0 removes FILE before it writes and after the communicator is up
//       [--rccl-timeout SEC]     give up (exit code 3) if the communicator is not up after SEC seconds (default 120)
//   --assemble F0,F1,...         put the tile files of ranks 0..G-1 together and write the PNG (no GPU needed)
//   --timing 1                   where the program's wall time goes, phase by phase, on stderr (the reference times all of main, cpu:660,721-723): the time before
//                                main (loader), OBJ + BVH, context creation, upload, the frame (the library adds its own phases: RT_TIMING), the PNG
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <memory>
#include <unistd.h>
#include <time.h>
#include <cstdio>
#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>

#include "../../../include/raytracer.hpp"

using namespace raytracer;

int main(int argc, char *argv[]) {
    if (argc < 3 || argv[1][0] == '-' || argv[2][0] == '-' || (argc - 3) % 2 != 0) {
        std::cout << "Invalid number of arguments!\nThe first argument is number of rays and the second argument is number of bounces.";
        return 0;   // cpu:655-658
    }
    const int num_rays = atoi(argv[1]), num_bounce = atoi(argv[2]);
    auto start_time = std::chrono::system_clock::now();
    auto lap_t = std::chrono::steady_clock::now();
    bool timing = false;
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "timing: %-44s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(n - lap_t).count());
        lap_t = n;
    };
    std::string program = "cpu", scene_name = "cat", out, obj = "cadnav.com_model/Models_F0202A090/cat.obj";
    int W = 512, H = 512, device = 0, variant = RT_VARIANT_AUTO;
    std::vector<int> devices;
    int tile_rank = -1, tile_world = 0;
    std::string tiles_file, rccl_id_file;
    std::string rccl_nonce = getenv("RT_RCCL_NONCE") ? getenv("RT_RCCL_NONCE") : std::to_string((long)getppid());   // one launch = one nonce (the ranks' common parent by default)
    int rccl_timeout_s = 120;
    std::vector<std::string> assemble;
    std::string mesh_material;
    bool second_cat = false;
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
        else if (k == "--mesh-material") mesh_material = v;
        else if (k == "--second-cat") second_cat = atoi(v.c_str()) != 0;
        else if (k == "--tile-rank") tile_rank = atoi(v.c_str());
        else if (k == "--tile-world") tile_world = atoi(v.c_str());
        else if (k == "--tiles") tiles_file = v;
        else if (k == "--rccl-id") rccl_id_file = v;
        else if (k == "--rccl-nonce") rccl_nonce = v;
        else if (k == "--rccl-timeout") rccl_timeout_s = std::max(1, atoi(v.c_str()));
        else if (k == "--timing") { timing = atoi(v.c_str()) != 0; if (timing) setenv("RT_TIMING", "1", 1); }
        else if (k == "--assemble") { std::stringstream ss(v); std::string tok; while (std::getline(ss, tok, ',')) assemble.push_back(tok); }
        else if (k == "--devices") { std::stringstream ss(v); std::string tok; while (std::getline(ss, tok, ',')) devices.push_back(atoi(tok.c_str())); }
        else { std::cerr << "unknown option " << k << "\n"; return 2; }
    }
    if (timing) {                                                       // process start -> main: /proc/self/stat field 22 (start time in clock ticks since boot) against the boot clock
        std::ifstream st("/proc/self/stat");
        std::string all((std::istreambuf_iterator<char>(st)), std::istreambuf_iterator<char>());
        const size_t rp = all.rfind(')');
        std::stringstream ss(rp == std::string::npos ? std::string() : all.substr(rp + 2));
        std::string tok; double ticks = -1;
        for (int f = 3; f <= 22 && (ss >> tok); ++f) if (f == 22) ticks = atof(tok.c_str());
        timespec bt{}; clock_gettime(CLOCK_BOOTTIME, &bt);
        if (ticks >= 0) fprintf(stderr, "timing: %-44s %9.3f ms (10 ms resolution)\n", "process start -> main (loader, libamdhip64)", (bt.tv_sec + bt.tv_nsec * 1e-9 - ticks / sysconf(_SC_CLK_TCK)) * 1e3);
        lap_t = std::chrono::steady_clock::now();
    }
    const bool optimized = program == "optimized";
    if (out.empty()) out = optimized ? "image_optimized.png" : "image.png";   // opt:862 / cpu:719

    if (!assemble.empty()) {                                           // the root of a process-per-GPU run: tiles -> frame -> PNG
        std::vector<unsigned char> frame((size_t)W * H * 3);
        const int G = (int)assemble.size();
        for (int r = 0; r < G; ++r) {
            std::ifstream f(assemble[r], std::ios::binary);
            std::vector<unsigned char> tiles((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            if (tiles.size() != Renderer::tile_rows_of(H, r, G).size() * (size_t)W * 3) { std::cerr << assemble[r] << ": wrong size for rank " << r << " of " << G << "\n"; return 1; }
            Renderer::assemble_tiles(frame, tiles, W, H, r, G);
        }
        if (!write_png(out.c_str(), W, H, frame.data())) { std::cerr << "cannot write " << out << "\n"; return 1; }
        return 0;
    }
    if (tile_rank >= 0 && (tile_world < 1 || tile_rank >= tile_world || (tiles_file.empty() && rccl_id_file.empty()))) { std::cerr << "--tile-rank needs --tile-world and --tiles or --rccl-id\n"; return 2; }

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
        // --mesh-material mirror | glass: Geometry's public members on the mesh (cpu:113-116), as a user of the reference's classes would set them
        if (mesh_ptr && mesh_material == "mirror") mesh_ptr->mirror = true;
        if (mesh_ptr && mesh_material == "glass") { mesh_ptr->in_refraction_index = 1.5f; mesh_ptr->out_refraction_index = 1.f; }
        // --second-cat 1: a second TriangleMesh in the same Scene::objects (cpu:538-543), the cat at half size in front of the first: v * 0.5 + (16, -5, 20), mirror, added last
        TriangleMesh *mesh2_ptr = nullptr;
        if (mesh_ptr && second_cat) {
            mesh2_ptr = new TriangleMesh();
            mesh2_ptr->readOBJ(obj.c_str());
            for (auto &v : mesh2_ptr->vertices) v = v * 0.5f + Vector(16, -5, 20);
            mesh2_ptr->albedo = Vector(0.6f, 0.3f, 0.1f);
            mesh2_ptr->mirror = true;
            mesh2_ptr->buildBVH(&(mesh2_ptr->bvh), 0, (int)mesh2_ptr->indices.size());
        }
        lap("readOBJ + buildBVH (host)");
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
        if (mesh2_ptr) s.addObject(mesh2_ptr);

        RenderSettings rs = optimized ? RenderSettings::optimized_cu() : RenderSettings::cpu_launcher();
        rs.W = W; rs.H = H; rs.num_rays = num_rays; rs.num_bounce = num_bounce; rs.variant = variant;
        std::vector<unsigned char> image;
        if (!devices.empty()) {                                       // several GPUs, one process
            MultiRenderer renderer(devices);
            renderer.upload(s);
            image = renderer.render_rgb8(rs);                          // tonemap (cpu:714-716) on every device, 8-bit tiles exchanged
            if (!write_png(out.c_str(), W, H, image.data())) { std::cerr << "cannot write " << out << "\n"; return 1; }
            const rt_multi_stats st = renderer.stats();
            std::chrono::duration<float> run_time = std::chrono::system_clock::now() - start_time;
            std::cout << "Rendering time: " << run_time.count() << " s\n";
            std::cerr << st.n_devices << " devices, frame " << st.frame_ms << " ms, gather " << st.gather_ms << " ms (" << st.gather_bytes << " bytes), " << st.rays << " rays\n";
            return 0;
        }
        if (tile_rank >= 0 && !rccl_id_file.empty()) {                // one process per GPU, tiles gathered over RCCL into rank 0
            // Everything that can fail on the host side happens BEFORE any rank enters the collective communicator set-up: a rank that
            // throws here never publishes / consumes an id, and the others give up after their bounded wait below.
            Renderer renderer(device);
            renderer.upload(s);
            // The id travels through a file.  A file left by an earlier or aborted run must never be taken for this run's: the file holds
            // the id AND the launch's nonce (--rccl-nonce, default: the parent process id, which the ranks of one launcher script share),
            // rank 0 removes whatever is at the path before it creates its id and removes its own file once the communicator exists (the
            // set-up is collective: by then every rank has read it), and the other ranks accept only a file that carries their nonce.
            const std::string tag = "rtid:" + rccl_nonce + ":";
            std::vector<unsigned char> id;
            if (tile_rank == 0) {
                std::remove(rccl_id_file.c_str());
                id = TileComm::make_id();
                const std::string tmp = rccl_id_file + ".tmp";
                {
                    std::ofstream f(tmp, std::ios::binary);
                    f.write(tag.data(), (std::streamsize)tag.size());
                    f.write(reinterpret_cast<const char *>(id.data()), (std::streamsize)id.size());
                    if (!f) { std::cerr << "cannot write " << tmp << "\n"; return 1; }
                }
                if (std::rename(tmp.c_str(), rccl_id_file.c_str()) != 0) { std::cerr << "cannot rename " << tmp << "\n"; return 1; }   // written whole before it gets its name
            } else {
                for (int tries = 0; tries < rccl_timeout_s * 10 && id.empty(); ++tries) {   // bounded: --rccl-timeout
                    std::ifstream f(rccl_id_file, std::ios::binary);
                    std::vector<unsigned char> raw;
                    if (f) raw.assign((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
                    if (raw.size() == tag.size() + RT_COMM_ID_BYTES && std::equal(tag.begin(), tag.end(), raw.begin())) id.assign(raw.begin() + (long)tag.size(), raw.end());
                    else std::this_thread::sleep_for(std::chrono::milliseconds(100));   // absent, half-written, or another launch's file
                }
                if (id.empty()) { std::cerr << "no communicator id of launch " << rccl_nonce << " in " << rccl_id_file << "\n"; return 1; }
            }
            // ncclCommInitRank blocks until every rank of the id has arrived.  If one never does (it failed above, or was never started) a
            // process that has initialised the GPU must not sit there for ever: a watchdog ends it after --rccl-timeout seconds.
            std::atomic<bool> comm_up{false};
            std::thread watchdog([&comm_up, rccl_timeout_s, tile_rank] {
                for (int k = 0; k < rccl_timeout_s * 10 && !comm_up.load(); ++k) std::this_thread::sleep_for(std::chrono::milliseconds(100));
                if (!comm_up.load()) { std::cerr << "rank " << tile_rank << ": the RCCL communicator did not come up within " << rccl_timeout_s << " s (a rank missing or holding another id); giving up\n"; std::_Exit(3); }
            });
            std::unique_ptr<TileComm> comm;
            try { comm.reset(new TileComm(device, tile_rank, tile_world, id)); }
            catch (...) { comm_up.store(true); watchdog.join(); if (tile_rank == 0) std::remove(rccl_id_file.c_str()); throw; }
            comm_up.store(true);
            watchdog.join();
            if (tile_rank == 0) std::remove(rccl_id_file.c_str());      // every rank has read it: nothing stale stays behind
            image = renderer.render_gather_rgb8(rs, *comm);
            if (tile_rank == 0 && !write_png(out.c_str(), W, H, image.data())) { std::cerr << "cannot write " << out << "\n"; return 1; }
            std::chrono::duration<float> run_time = std::chrono::system_clock::now() - start_time;
            std::cout << "Rendering time: " << run_time.count() << " s\n";
            std::cerr << "rank " << tile_rank << " of " << tile_world << ": " << comm->last_bytes() << " bytes over RCCL\n";
            return 0;
        }
        if (tile_rank >= 0) {                                         // one process per GPU: this rank's tiles only
            Renderer renderer(device);
            renderer.upload(s);
            const std::vector<unsigned char> tiles = renderer.render_tiles_rgb8(rs, tile_rank, tile_world);
            std::ofstream f(tiles_file, std::ios::binary);
            f.write(reinterpret_cast<const char *>(tiles.data()), (std::streamsize)tiles.size());
            if (!f) { std::cerr << "cannot write " << tiles_file << "\n"; return 1; }
            std::chrono::duration<float> run_time = std::chrono::system_clock::now() - start_time;
            std::cout << "Rendering time: " << run_time.count() << " s\n";
            return 0;
        }
        Renderer renderer(device);
        lap("Renderer (rt_ctx_create), total");
        renderer.upload(s);
        lap("upload (flatten + rt_scene_upload), total");
        image = renderer.render_rgb8(rs);
        lap("render_rgb8, total");
        if (!write_png(out.c_str(), W, H, image.data())) { std::cerr << "cannot write " << out << "\n"; return 1; }
        lap("write_png");
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
