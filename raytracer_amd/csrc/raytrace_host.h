// Host-side mirror of the reference's interface for the trace path, above the C-ABI
// (include/rt_api.h).  Same class and method names, argument meaning and error
// behaviour as the reference, so `main` reads like reference src/main.cpp:
//   Scene::parseScene(char*)        -> 1 on success, -1 on failure  (src/parse.cpp:92-271)
//   RayTrace::rayTrace(Scene&)      -> new Image*, caller deletes   (src/raytrace.cpp:44-129)
//   Image::Write(char*)             -> 24-bit BMP                   (src/image.cpp:109-124)
// Nothing here computes a pixel on the CPU: rayTrace() forwards to rt_render and
// returns nullptr (after printing the library's error) when no GPU path exists.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_api.h"

namespace rtamd {

struct Pixel { uint8_t r, g, b, a; };

class Image {
  public:
    Image(int width, int height);
    int Width() const { return width; }
    int Height() const { return height; }
    Pixel& GetPixel(int x, int y) { return pixels[(size_t)y * width + x]; }   // src/image.h:61
    void Write(char* fname);
    std::vector<Pixel> pixels;
    int width, height;
};

namespace rt {
class Image {   // the film description held by the scene (reference rt::Image, src/parse.h:220-239)
  public:
    std::string getFileName() const { return filename; }
    int getWidth() const { return width; }
    int getHeight() const { return height; }
    std::string filename = "raytraced.bmp";
    int width = 640, height = 480;
};
}  // namespace rt

class Scene {
  public:
    Scene();
    ~Scene();
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    // flags: RT_PARSE_REF_COMPAT to reproduce film_resolution W H -> W x W
    int parseScene(char* file, int flags = 0);
    rt::Image getImage() const { return img; }
    int getDepth() const;
    int getSampleRate() const;
    void setSeed(uint32_t seed);
    void setFilm(int w, int h);
    void setSampleRate(int spp);
    const rt_scene_desc* desc() const { return d; }
  private:
    rt_scene_desc* d = nullptr;   // owned by the library (rt_free_scene)
    rt_scene_desc edited{};       // copy with caller overrides applied
    bool use_edited = false;
    rt::Image img;
    friend class RayTrace;
};

class RayTrace {
  public:
    RayTrace();
    explicit RayTrace(int accel);
    ~RayTrace();
    Image* rayTrace(Scene& scn);
    const rt_stats& stats() const { return last; }
  private:
    rt_ctx* ctx = nullptr;
    int accel = RT_ACCEL_AUTO;
    rt_stats last{};
};

}  // namespace rtamd
