// `./raytrace scene_file` -- the reference's command line (src/main.cpp:15-98) on top of
// the MI355X trace loop.  Same positional argument, same stdout lines, same exit codes,
// same output-name rule, and by default the same film as the reference renders:
// `film_resolution W H` gives W x W (src/parse.cpp:119).  Optional flags may follow the
// scene path:
//   --film-as-written   honour the H of `film_resolution W H` (the reference does not)
//   --ref-compat        accepted for symmetry: the default behaviour
//   --accel auto|brute|lbvh
//   --gpus N            split the frame over devices 0..N-1 (one process, a host thread each)
//   --seed N            seed of the jitter sequence (sample_rate > 1)
//   --film W H          override the film size      --spp N   override sample_rate
//   --stats             after the trace, one JSON line: rays by class, per-stage device ms,
//                       Mrays/s, node / primitive fetches and tests per ray; and after the image is written a
//                       second one with the wall-clock split of the run (parse, contexts, scene upload + hierarchies,
//                       frame, pixel conversion, BMP file)
//   --stats-file PATH   the same JSON into a file instead of stdout
//   --calibrate         time both hierarchies and every walk on a sample of the frame first and keep the
//                       fastest per ray generation (what a library user gets by default; here one frame is
//                       rendered per process, and the timing costs two to three frames)
//   --debug             what the reference's DEBUG build prints: scene dump, view-plane corners,
//                       output file name (src/main.cpp:50-53, :74-76; src/raytrace.cpp:45-47, :78-82)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "raytrace_host.h"

using namespace rtamd;

static void displayElapsed(double sec) {
    const int days = (int)(sec / 86400.0);
    sec -= days * 86400.0;
    const int hours = (int)(sec / 3600.0);
    sec -= hours * 3600.0;
    const int minutes = (int)(sec / 60.0);
    sec -= minutes * 60.0;
    std::printf("%dd %dh %dm %fs\n", days, hours, minutes, sec);
}

static std::string stats_json(const rt_stats& s, int gpus, double wall_s) {
    const double rays = (double)(s.rays_primary + s.rays_secondary + s.rays_shadow);
    const double closest = (double)(s.rays_primary + s.rays_secondary), any = (double)s.rays_shadow;
    auto per = [](unsigned long long v, double n) { return n > 0 ? (double)v / n : 0.0; };
    char b[2048];
    std::snprintf(b, sizeof(b),
        "{\"rays\": {\"primary\": %llu, \"secondary\": %llu, \"shadow\": %llu}, "
        "\"ms\": {\"render\": %.4f, \"extend\": %.4f, \"shadow\": %.4f, \"shade\": %.4f, \"other\": %.4f, \"wall\": %.4f}, "
        "\"mrays_per_s\": %.3f, \"gpus\": %d, \"accel\": \"%s\", \"generations\": %d, \"batches\": %d, \"retries\": %d, "
        "\"launches\": {\"extend\": %d, \"shadow\": %d}, "
        "\"fetches_per_ray\": {\"closest_hit\": {\"nodes\": %.4f, \"spheres\": %.4f, \"triangles\": %.4f}, "
        "\"any_hit\": {\"nodes\": %.4f, \"spheres\": %.4f, \"triangles\": %.4f}}, "
        "\"tests_per_ray\": {\"closest_hit\": {\"box_pairs\": %.4f, \"primitives\": %.4f}, "
        "\"any_hit\": {\"box_pairs\": %.4f, \"primitives\": %.4f}}}",
        (unsigned long long)s.rays_primary, (unsigned long long)s.rays_secondary, (unsigned long long)s.rays_shadow,
        s.ms_render, s.ms_extend, s.ms_shadow, s.ms_shade, s.ms_other, wall_s * 1e3,
        s.ms_render > 0 ? rays / (s.ms_render * 1e3) : 0.0, gpus, s.accel_used == RT_ACCEL_LBVH ? "lbvh" : "brute",
        s.generations, s.batches, s.retries, s.launches_extend, s.launches_shadow,
        per(s.nodes_visited[0], closest), per(s.sphere_tests[0], closest), per(s.tri_tests[0], closest),
        per(s.nodes_visited[1], any), per(s.sphere_tests[1], any), per(s.tri_tests[1], any),
        per(s.lane_node_tests[0], closest), per(s.lane_prim_tests[0], closest),
        per(s.lane_node_tests[1], any), per(s.lane_prim_tests[1], any));
    return b;
}

int main(int argc, char* argv[]) {
    argv++, argc--;
    if (argc == 0) {
        std::printf("No Scene File Specified\n");
        std::exit(-1);
    }
    char* scenename = argv[0];
    int flags = RT_PARSE_REF_COMPAT, accel = RT_ACCEL_AUTO, film_w = 0, film_h = 0, spp = 0, gpus = 1;
    bool stats = false, have_seed = false, stray = false, debug = false, calibrate = false;
    std::string stats_file;
    unsigned seed = 1;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--ref-compat") flags |= RT_PARSE_REF_COMPAT;
        else if (a == "--film-as-written") flags &= ~RT_PARSE_REF_COMPAT;
        else if (a == "--stats") stats = true;
        else if (a == "--stats-file" && i + 1 < argc) { stats = true; stats_file = argv[++i]; }
        else if (a == "--debug") debug = true;
        else if (a == "--calibrate") calibrate = true;
        else if (a == "--accel" && i + 1 < argc) {
            std::string v = argv[++i];
            accel = v == "brute" ? RT_ACCEL_BRUTE : v == "lbvh" ? RT_ACCEL_LBVH : RT_ACCEL_AUTO;
        } else if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) { seed = (unsigned)std::strtoul(argv[++i], nullptr, 10); have_seed = true; }
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
    const double parse_ms = elapsed.count() * 1e3;
    std::printf("Scene parsed in ");
    displayElapsed(elapsed.count());
    if (have_seed) data->setSeed(seed);
    if (film_w > 0 && film_h > 0) data->setFilm(film_w, film_h);
    if (spp > 0) data->setSampleRate(spp);
    if (debug) {
        std::printf("Printing\n");
        printScene(data);
    }

    RayTrace t(accel, gpus);
    t.setDebug(debug);
    t.setDetailedStats(stats);
    t.setCalibration(calibrate);
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
        const std::string js = stats_json(t.stats(), t.gpus(), elapsed.count());
        if (stats_file.empty()) std::printf("%s\n", js.c_str());
        else if (FILE* f = std::fopen(stats_file.c_str(), "w")) { std::fprintf(f, "%s\n", js.c_str()); std::fclose(f); }
        else std::printf("cannot write %s\n", stats_file.c_str());
    }
    char out[4096];
    rt_output_name(data->getImage().getFileName().c_str(), out, sizeof(out));
    delete data;
    start = std::chrono::system_clock::now();
    img->Write(out);
    elapsed = std::chrono::system_clock::now() - start;
    delete img;
    if (stats) {
        const RayTrace::WallTimes& w = t.wallTimes();
        char b[512];
        std::snprintf(b, sizeof(b), "{\"wall_ms\": {\"parse\": %.3f, \"create\": %.3f, \"load\": %.3f, \"render\": %.3f, "
                      "\"convert\": %.3f, \"write_bmp\": %.3f}}", parse_ms, w.create, w.load, w.render, w.convert, elapsed.count() * 1e3);
        if (stats_file.empty()) std::printf("%s\n", b);
        else if (FILE* f = std::fopen(stats_file.c_str(), "a")) { std::fprintf(f, "%s\n", b); std::fclose(f); }
    }
    if (debug) std::printf("Outputed image to file: %s\n", out);
    return 0;
}
