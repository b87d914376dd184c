// `./raytrace scene_file` -- the reference's command line (src/main.cpp:15-98) on top of
// the MI355X trace loop.  Same positional argument, same stdout lines, same exit codes,
// same output-name rule; optional flags may follow the scene path:
//   --ref-compat        film_resolution W H -> W x W, as the reference parser does
//   --accel auto|brute|lbvh
//   --seed N            seed of the jitter sequence (sample_rate > 1)
//   --film W H          override the film size      --spp N   override sample_rate
//   --stats             print ray counts / Mrays/s after the trace
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "raytrace_host.h"

using namespace rtamd;

static void displayElapsed(double sec) {
    int days, hours, minutes;
    days = (int)(sec / 86400.0);
    sec -= days * 86400.0;
    hours = (int)(sec / 3600.0);
    sec -= hours * 3600.0;
    minutes = (int)(sec / 60.0);
    sec -= minutes * 60.0;
    std::printf("%dd %dh %dm %fs\n", days, hours, minutes, sec);
}

int main(int argc, char* argv[]) {
    argv++, argc--;
    if (argc == 0) {
        std::printf("No Scene File Specified\n");
        std::exit(-1);
    }
    char* scenename = argv[0];
    int flags = 0, accel = RT_ACCEL_AUTO, film_w = 0, film_h = 0, spp = 0;
    bool stats = false, have_seed = false, stray = false;
    unsigned seed = 1;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--ref-compat") flags |= RT_PARSE_REF_COMPAT;
        else if (a == "--stats") stats = true;
        else if (a == "--accel" && i + 1 < argc) {
            std::string v = argv[++i];
            accel = v == "brute" ? RT_ACCEL_BRUTE : v == "lbvh" ? RT_ACCEL_LBVH : RT_ACCEL_AUTO;
        } else if (a == "--seed" && i + 1 < argc) { seed = (unsigned)std::strtoul(argv[++i], nullptr, 10); have_seed = true; }
        else if (a == "--film" && i + 2 < argc) { film_w = std::atoi(argv[++i]); film_h = std::atoi(argv[++i]); }
        else if (a == "--spp" && i + 1 < argc) spp = std::atoi(argv[++i]);
        else stray = true;
    }
    if (stray) std::printf("Too many arguments. Defaulting to first one\n");

    Scene* data = new Scene();
    auto start = std::chrono::system_clock::now();
    if (data->parseScene(scenename, flags) < 0) {
        std::printf("Scene Parse Failed\n");
        std::exit(-1);
    }
    std::chrono::duration<double> elapsed = std::chrono::system_clock::now() - start;
    std::printf("Scene parsed in ");
    displayElapsed(elapsed.count());
    if (have_seed) data->setSeed(seed);
    if (film_w > 0 && film_h > 0) data->setFilm(film_w, film_h);
    if (spp > 0) data->setSampleRate(spp);

    RayTrace t(accel);
    start = std::chrono::system_clock::now();
    Image* img = t.rayTrace(*data);
    elapsed = std::chrono::system_clock::now() - start;
    if (!img) {
        std::printf("Scene Trace Failed\n");
        std::exit(-1);
    }
    std::printf("Scene traced in ");
    displayElapsed(elapsed.count());
    if (stats) {
        const rt_stats& s = t.stats();
        double rays = (double)(s.rays_primary + s.rays_secondary + s.rays_shadow);
        std::printf("rays primary %llu secondary %llu shadow %llu | %.3f ms on device | %.1f Mrays/s | accel %s\n",
                    (unsigned long long)s.rays_primary, (unsigned long long)s.rays_secondary,
                    (unsigned long long)s.rays_shadow, s.ms_render, rays / (s.ms_render * 1e3),
                    s.accel_used == RT_ACCEL_LBVH ? "lbvh" : "brute");
    }
    char out[4096];
    rt_output_name(data->getImage().getFileName().c_str(), out, sizeof(out));
    delete data;
    img->Write(out);
    delete img;
    return 0;
}
