#include "raytrace_host.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace rtamd {

Image::Image(int w, int h) : pixels((size_t)w * h, Pixel{255, 255, 255, 0}), width(w), height(h) {}

void Image::Write(char* fname) {
    std::vector<uint8_t> rgb((size_t)width * height * 3);
    for (size_t i = 0; i < pixels.size(); i++) {
        rgb[3 * i] = pixels[i].r; rgb[3 * i + 1] = pixels[i].g; rgb[3 * i + 2] = pixels[i].b;
    }
    int rc = rt_write_bmp24(fname, rgb.data(), width, height);
    if (rc != RT_OK) std::printf("EasyBMP Error: Cannot open file %s for output.\n", fname);
}

Scene::Scene() {}
Scene::~Scene() { rt_free_scene(d); }

int Scene::parseScene(char* file, int flags) {
    rt_free_scene(d);
    d = nullptr;
    char name[4096];
    int rc = rt_parse_scene_file(file, flags, &d, name, sizeof(name));
    if (rc != RT_OK) {
        // the reference prints its own message before returning -1 (src/parse.cpp:97, :206)
        std::printf("%s\n", rt_last_error());
        return -1;
    }
    edited = *d;
    img.filename = name;
    img.width = d->width;
    img.height = d->height;
    return 1;
}

int Scene::getDepth() const { return edited.max_depth; }
int Scene::getSampleRate() const { return edited.spp; }
void Scene::setSeed(uint32_t seed) { edited.seed = seed; }
void Scene::setFilm(int w, int h) { edited.width = w; edited.height = h; img.width = w; img.height = h; }
void Scene::setSampleRate(int spp) { edited.spp = spp; }

// ---------------------------------------------------------------- scene dump
// Line for line what the reference's DEBUG build prints (src/parse.cpp:791-953): camera, spheres,
// film, background, lights, ambient light, depth, projection, triangles; every vector as
// "(%f %f %f)".
namespace {
void vec(const char* label, const float* v) { std::printf("%s(%f %f %f)\n", label, v[0], v[1], v[2]); }
void material(const float* m) {
    std::printf("Material:\n");
    vec("Ambient Color: ", m);
    vec("Diffuse Color: ", m + 3);
    vec("Specular Color: ", m + 6);
    std::printf("Cosin Power: %f\n", m[9]);
    vec("Transmissive Color: ", m + 10);
    std::printf("ior: %f\n", m[13]);
}
void light(int type, const float* color, const float* pos, const float* dir, float a1, float a2) {
    std::printf("Type: ");
    if (type == RT_LIGHT_DIRECTIONAL) { std::printf("Directional\n"); vec("Direction: ", dir); }
    else if (type == RT_LIGHT_POINT) { std::printf("Point\n"); vec("Position: ", pos); }
    else if (type == RT_LIGHT_SPOT) {
        std::printf("Spot\n");
        vec("Position: ", pos);
        vec("Direction: ", dir);
        std::printf("Angle: (%f %f)\n", a1, a2);
    } else std::printf("Ambient\n");
    vec("Color: ", color);
}
}  // namespace

void printScene(Scene* scn) {
    const rt_scene_desc* s = scn->desc();
    if (!s) return;
    std::printf("\nCamera:\n");
    vec("Position: ", s->cam_pos);
    vec("Direction: ", s->cam_dir);
    vec("Up Vector: ", s->cam_up);
    std::printf("Half Angle: %f\n\n", s->cam_half_angle);
    std::printf("Spheres:\n");
    for (int i = 0; i < s->n_spheres; i++) {
        vec("Position: ", s->spheres + 4 * (size_t)i);
        std::printf("Radius: %f\n", s->spheres[4 * (size_t)i + 3]);
        material(s->materials + 14 * (size_t)s->sphere_mat[i]);
    }
    if (s->n_spheres == 0) std::printf("No Spheres\n");
    std::printf("\nImage:\nFile Name: %s\nWidth: %d\nHeight: %d\n\n", scn->getImage().getFileName().c_str(), s->width, s->height);
    vec("Background Color: ", s->background);
    std::printf("\nLights:\n");
    for (int i = 0; i < s->n_lights; i++) {
        const rt_light& L = s->lights[i];
        light(L.type, L.color, L.pos, L.dir, L.angle1, L.angle2);
    }
    if (s->n_lights == 0) std::printf("No Lights\n");
    std::printf("\nAmbient Light:\n");
    const float zero[3] = {0, 0, 0};
    light(RT_LIGHT_AMBIENT, s->ambient, zero, zero, 0, 0);
    std::printf("\nDepth: %d\n", s->max_depth);
    std::printf("Projection Type: %s\n\n", s->proj == RT_PROJ_PERSP ? "Perspective" : "Orthographic");
    std::printf("Triangles:\n");
    for (int i = 0; i < s->n_triangles; i++) {
        const float* v = s->tri_verts + 9 * (size_t)i;
        const float* n = s->tri_normals + 9 * (size_t)i;
        vec("Vertex1: ", v); vec("Vertex2: ", v + 3); vec("Vertex3: ", v + 6);
        vec("Normal: ", n);
        if (s->tri_flags[i]) { vec("Normal2: ", n + 3); vec("Normal3: ", n + 6); }
        material(s->materials + 14 * (size_t)s->tri_mat[i]);
    }
    if (s->n_triangles == 0) std::printf("No Triangles\n");
    std::printf("\n");
}

// ---------------------------------------------------------------- trace
RayTrace::RayTrace() {}
RayTrace::RayTrace(int accel_, int gpus_) : accel(accel_), ngpus(gpus_ < 1 ? 1 : gpus_) {}
RayTrace::~RayTrace() { for (rt_ctx* c : ctx) rt_destroy(c); }

namespace {
// The reference prints `0%`, `10%`, ... from inside its pixel loop, one line whenever the running
// pixel index is a multiple of (int)(.1 * width * height) (src/raytrace.cpp:83-85, :97-103), and
// `100%` after the loop (:125).  Here a line is printed as soon as the renderer reports that many
// pixels done, so the lines come out while the frame is traced, batch by batch.
struct Progress {
    long long total = 0, tenper = 0, lines = 0, printed = 0;
    static void on_batch(void* user, uint64_t done, uint64_t shard_total) {
        Progress* p = (Progress*)user;
        // a shard's share of its own pixels stands for the same share of the film
        const double frac = shard_total ? (double)done / (double)shard_total : 1.0;
        while (p->printed < p->lines && (double)(p->printed * p->tenper) <= frac * (double)p->total) {
            std::printf("%lld%%\n", 10 * p->printed);
            p->printed++;
        }
        std::fflush(stdout);
    }
};
}  // namespace

Image* RayTrace::rayTrace(Scene& scn) {
    if (!scn.d) return nullptr;
    const rt_scene_desc& s = scn.edited;
    if (debug) std::printf("Beginning Trace\n");
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    auto t0 = clk::now();
    while ((int)ctx.size() < ngpus) {
        rt_options opt{};
        opt.accel = accel;
        opt.device = (int)ctx.size();
        rt_ctx* c = nullptr;
        int rc = rt_create(&c, &opt);
        if (rc != RT_OK) { std::printf("%s: %s\n", rt_strerror(rc), rt_last_error()); return nullptr; }
        ctx.push_back(c);
    }
    wall.create = ms_since(t0);
    t0 = clk::now();
    for (rt_ctx* c : ctx) {
        int rc = rt_load_scene(c, &s);
        if (rc != RT_OK) { std::printf("%s: %s\n", rt_strerror(rc), rt_last_error()); return nullptr; }
        rt_set_stage_times(c, detailed ? 1 : 0);
        rt_set_work_counters(c, detailed ? 1 : 0);
        rt_set_calibration(c, calibrate ? 1 : 0);      // one frame per process: timing the walks first costs more than it saves
    }
    wall.load = ms_since(t0);
    if (debug) {
        // P1 / P2 / Plane Dist of the DEBUG build (src/raytrace.cpp:78-82): Scene::getPlaneDist and
        // getExtremePoints restated on the host (src/parse.cpp:275-277, src/raytrace.cpp:635-647)
        const float pd = (float)(s.height / (2.0 * std::tan((double)s.cam_half_angle)));
        const float* D = s.cam_dir; const float* U = s.cam_up; const float* P = s.cam_pos;
        const float right[3] = {D[1] * U[2] - D[2] * U[1], D[2] * U[0] - D[0] * U[2], D[0] * U[1] - D[1] * U[0]};
        const float hw = (float)(s.width / 2.0), hh = (float)(s.height / 2.0);
        float p1[3], p2[3];
        for (int k = 0; k < 3; k++) {
            const float p0 = P[k] + D[k] * pd;
            p1[k] = (p0 + right[k] * hw) + U[k] * hh;
            p2[k] = (p0 - right[k] * hw) - U[k] * hh;
        }
        std::printf("P1: (%f %f %f)\n", p1[0], p1[1], p1[2]);
        std::printf("P2: (%f %f %f)\n", p2[0], p2[1], p2[2]);
        std::printf("Plane Dist: %f\n", (double)pd);
    }
    Progress prog;
    prog.total = (long long)s.width * s.height;
    prog.tenper = (long long)(int)(.1 * (double)(int)prog.total);
    prog.lines = prog.tenper > 0 ? (prog.total + prog.tenper - 1) / prog.tenper : 1;
    rt_set_progress_callback(ctx[0], &Progress::on_batch, &prog);
    std::vector<uint8_t> rgb((size_t)s.width * s.height * 3);
    t0 = clk::now();
    int rc = ngpus == 1 ? rt_render(ctx[0], rgb.data(), nullptr, &last)
                        : rt_render_multi(ctx.data(), ngpus, rgb.data(), &last);
    rt_set_progress_callback(ctx[0], nullptr, nullptr);
    if (rc != RT_OK) { std::printf("%s: %s\n", rt_strerror(rc), rt_last_error()); return nullptr; }
    wall.render = ms_since(t0);
    std::printf("100%%\n");
    t0 = clk::now();
    Image* dest = new Image(s.width, s.height);
    for (size_t i = 0; i < dest->pixels.size(); i++) {
        dest->pixels[i].r = rgb[3 * i]; dest->pixels[i].g = rgb[3 * i + 1]; dest->pixels[i].b = rgb[3 * i + 2];
    }
    wall.convert = ms_since(t0);
    return dest;
}

}  // namespace rtamd
