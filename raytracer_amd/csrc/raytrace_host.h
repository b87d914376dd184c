// Host-side mirror of the reference's interface for the trace path, above the C-ABI
// (include/rt_api.h).  Same class and method names, argument meaning and error
// behaviour as the reference, so `main` reads like reference src/main.cpp:
//   Scene::parseScene(char*)        -> 1 on success, -1 on failure  (src/parse.cpp:92-271)
//   RayTrace::rayTrace(Scene&)      -> new Image*, caller deletes   (src/raytrace.cpp:44-129)
//   Image::Write(char*)             -> 24-bit BMP                   (src/image.cpp:109-124)
//   printScene(Scene*)              -> the DEBUG build's scene dump (src/parse.cpp:791-953)
// Nothing here computes a pixel on the CPU: rayTrace() forwards to rt_render (or, for several
// GPUs, rt_render_multi) and returns nullptr (after printing the library's error) when no GPU
// path exists.
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
    // flags: RT_PARSE_REF_COMPAT reproduces film_resolution W H -> W x W (src/parse.cpp:119)
    int parseScene(char* file, int flags = RT_PARSE_REF_COMPAT);
    rt::Image getImage() const { return img; }
    int getDepth() const;
    int getSampleRate() const;
    void setSeed(uint32_t seed);
    void setFilm(int w, int h);
    void setSampleRate(int spp);
    const rt_scene_desc* desc() const { return d ? &edited : nullptr; }
  private:
    rt_scene_desc* d = nullptr;   // owned by the library (rt_free_scene)
    rt_scene_desc edited{};       // copy with caller overrides applied
    rt::Image img;
    friend class RayTrace;
};

// the scene dump of the reference's DEBUG build (src/main.cpp:50-53 -> printScene, src/parse.cpp:791-953)
void printScene(Scene* scn);

class RayTrace {
  public:
    RayTrace();
    // accel: RT_ACCEL_*; gpus: devices 0 .. gpus-1, one context and one host thread each
    explicit RayTrace(int accel, int gpus = 1);
    ~RayTrace();
    RayTrace(const RayTrace&) = delete;
    RayTrace& operator=(const RayTrace&) = delete;
    Image* rayTrace(Scene& scn);
    const rt_stats& stats() const { return last; }
    // wall-clock milliseconds of the last rayTrace(): contexts created, scene uploaded + hierarchies built, frame
    // rendered and copied out (includes the first allocation of the ray pool), pixels converted into the Image
    struct WallTimes { double create = 0, load = 0, render = 0, convert = 0; };
    const WallTimes& wallTimes() const { return wall; }
    void setDebug(bool on) { debug = on; }           // the DEBUG build's lines of rayTrace (src/raytrace.cpp:45-47, :78-82)
    void setCalibration(bool on) { calibrate = on; }   // time hierarchies and walks before the (one) frame; off by default here
    void setDetailedStats(bool on) { detailed = on; } // per-stage times and traversal work counts in stats()
    int gpus() const { return ngpus; }
  private:
    std::vector<rt_ctx*> ctx;
    int accel = RT_ACCEL_AUTO;
    int ngpus = 1;
    bool debug = false, detailed = false, calibrate = false;
    rt_stats last{};
    WallTimes wall{};
};

}  // namespace rtamd
