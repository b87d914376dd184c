#include "raytrace_host.h"

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
    use_edited = true;
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

RayTrace::RayTrace() {}
RayTrace::RayTrace(int accel_) : accel(accel_) {}
RayTrace::~RayTrace() { rt_destroy(ctx); }

Image* RayTrace::rayTrace(Scene& scn) {
    if (!scn.d) return nullptr;
    if (!ctx) {
        rt_options opt{};
        opt.accel = accel;
        int rc = rt_create(&ctx, &opt);
        if (rc != RT_OK) { std::printf("%s: %s\n", rt_strerror(rc), rt_last_error()); return nullptr; }
    }
    const rt_scene_desc& s = scn.edited;
    int rc = rt_load_scene(ctx, &s);
    if (rc != RT_OK) { std::printf("%s: %s\n", rt_strerror(rc), rt_last_error()); return nullptr; }
    // progress lines as the reference prints them while tracing (src/raytrace.cpp:97-103, :125)
    std::vector<uint8_t> rgb((size_t)s.width * s.height * 3);
    for (int p = 0; p < 100; p += 10) std::printf("%d%%\n", p);
    rc = rt_render(ctx, rgb.data(), nullptr, &last);
    if (rc != RT_OK) { std::printf("%s: %s\n", rt_strerror(rc), rt_last_error()); return nullptr; }
    std::printf("100%%\n");
    Image* dest = new Image(s.width, s.height);
    for (size_t i = 0; i < dest->pixels.size(); i++) {
        dest->pixels[i].r = rgb[3 * i]; dest->pixels[i].g = rgb[3 * i + 1]; dest->pixels[i].b = rgb[3 * i + 2];
    }
    return dest;
}

}  // namespace rtamd
