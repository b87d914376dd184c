// C-ABI of the MI355X trace loop: context, scene upload, batch scheduler.
// See include/rt_api.h for the contract of every entry point.
//
// There is no CPU path: every entry point that computes needs a HIP device and
// fails with RT_ERR_NO_DEVICE / RT_ERR_HIP otherwise.
#include "../../include/rt_api.h"
#include "rt_device.h"
#include "rt_kernels.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static_assert(RT_TILE == RT_TILE_PX, "tile size of the public header and the kernels differ");

// ---------------------------------------------------------------- errors
static thread_local std::string g_last_error;
void rt_set_error(const std::string& s) { g_last_error = s; }

extern "C" const char* rt_last_error(void) { return g_last_error.c_str(); }

extern "C" const char* rt_strerror(int status) {
    switch (status) {
        case RT_OK: return "ok";
        case RT_ERR_INVALID: return "invalid argument";
        case RT_ERR_IO: return "file could not be opened or written";
        case RT_ERR_PARSE: return "scene parse failed";
        case RT_ERR_NO_DEVICE: return "no HIP device (there is no CPU fallback)";
        case RT_ERR_HIP: return "HIP runtime error";
        case RT_ERR_NOMEM: return "out of memory";
        case RT_ERR_UNSUPPORTED: return "unsupported configuration";
        case RT_ERR_NO_SCENE: return "no scene loaded";
        default: return "unknown status";
    }
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            rt_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                  \
            return (e_ == hipErrorOutOfMemory) ? RT_ERR_NOMEM : RT_ERR_HIP;                   \
        }                                                                                     \
    } while (0)

// ---------------------------------------------------------------- allocation-free scratch
// The one buffer rt_assemble_shards_noalloc needs beyond its inputs and outputs: the tile-row table.  A
// fixed array in device memory, one copy per GPU the library runs on (held until the process ends), cut into
// RT_NOALLOC_SLOTS slices of RT_NOALLOC_MAX_ROWS entries: every context holds one slice of its GPU's array from
// rt_create to rt_destroy, so contexts (and graphs captured from different contexts) never share table entries.
// `scratch` is only a host-side handle of the symbol: the array's address on a device comes from
// hipGetSymbolAddress with that device current, is looked up once per device (rt_create) and is what every
// kernel argument uses.
static __device__ __align__(256) uint32_t scratch[RT_NOALLOC_SLOTS * RT_NOALLOC_MAX_ROWS];

#define RT_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static uint32_t* g_scratch_ptr[RT_MAX_DEVICES];      // per device ordinal; nullptr = not looked up yet
static uint32_t g_scratch_used[RT_MAX_DEVICES];      // per device: bit k set = slice k belongs to a living context

// Address of `scratch` on `device` (which must be the current device of the calling thread) and a free slice of it
// (*slot = -1 when every slice is taken: that context's allocation-free entry point then reports it).
static int scratch_on_device(int device, uint32_t** out, int* slot) {
    *out = nullptr;
    *slot = -1;
    if (device < 0 || device >= RT_MAX_DEVICES) { rt_set_error("device ordinal beyond the scratch table"); return RT_ERR_UNSUPPORTED; }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        uint32_t* p = nullptr;
        hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            rt_set_error(std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
            return RT_ERR_HIP;
        }
        g_scratch_ptr[device] = p;
    }
    for (int k = 0; k < RT_NOALLOC_SLOTS; k++)
        if (!(g_scratch_used[device] & (1u << k))) { g_scratch_used[device] |= 1u << k; *slot = k; break; }
    *out = g_scratch_ptr[device] + (size_t)std::max(0, *slot) * RT_NOALLOC_MAX_ROWS;
    return RT_OK;
}

static void scratch_release(int device, int slot) {
    if (device < 0 || device >= RT_MAX_DEVICES || slot < 0) return;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    g_scratch_used[device] &= ~(1u << slot);
}

// ---------------------------------------------------------------- streams
// Streams this library creates are never destroyed: a context returns them to a per-device list and the next
// context takes them from there.  Reason: the HIP runtime registers completion handlers for the marker commands of a
// stream (event records, waits on events of other streams, synchronisations), run later by a thread of the HSA
// runtime.  In the runtime of ROCm 7.0 (the libamdhip64.so that PyTorch 2.10+rocm7.0 bundles, which a process
// that imports torch first runs on) such a handler holds a plain pointer to the stream's queue object, and
// hipStreamDestroy frees that object without waiting for handlers still pending: a handler that runs late -- the
// handler thread gets no CPU while other threads of the process spin -- then decrements a counter and clears a word
// in memory that has been handed out again.  (The runtime of ROCm 7.2 holds a reference on the queue object for
// every pending handler.)  A stream that lives as long as the process has no such window.
static std::mutex g_stream_mutex;
static std::vector<hipStream_t> g_spare_streams[RT_MAX_DEVICES];

// A non-blocking stream on `device` (which must be the current device of the calling thread).
static hipError_t acquire_stream(int device, hipStream_t* out) {
    *out = nullptr;
    if (device >= 0 && device < RT_MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        if (!g_spare_streams[device].empty()) {
            *out = g_spare_streams[device].back();
            g_spare_streams[device].pop_back();
            return hipSuccess;
        }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

// Gives a stream of acquire_stream back, idle (the caller has synchronised it).
static void recycle_stream(int device, hipStream_t s) {
    if (!s) return;
    if (device >= 0 && device < RT_MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        g_spare_streams[device].push_back(s);
    }
    // (a device ordinal beyond the table: the stream is left alone rather than destroyed)
}

// Graphs with parallel branches run their side branches on streams the runtime creates with the executable graph and
// destroys with it: the same window.  Such graphs are only used on runtimes that keep the queue object alive for
// pending handlers (ROCm 7.2 on: HIP_VERSION 7.2).
static bool runtime_keeps_queues_alive() {
    static const bool ok = [] {
        int v = 0;
        if (hipRuntimeGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); return false; }
        return v >= 70200000;
    }();
    return ok;
}

// ---------------------------------------------------------------- context
enum { RT_TREE_KARRAS = 0, RT_TREE_PLOC = 1, RT_TREE_SAH = 2, RT_TREES = 3 };
static const char* const kTreeName[RT_TREES] = {"Karras", "PLOC", "SAH"};

struct rt_ctx {
    int device = 0;
    int num_cu = 256;
    int max_blocks = 2048;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    rt_options opt{};

    // scene
    bool has_scene = false;
    int accel = RT_ACCEL_BRUTE;
    // LBVH walk per generation: 2 = per-lane, 3 = wave packet, 4 = per-lane on the 4-wide collapse.  Chosen by calibrate() from
    // measured kernel times on a sample of the frame (coherent rays favour the packet walk,
    // incoherent ones the per-lane walk); RT_PACKET_GENS / RT_PACKET_SHADOW_GENS force it.
    int8_t mode_extend[RT_MAX_GEN + 1], mode_shadow[RT_MAX_GEN + 1];
    bool tuned = false, forced_modes = false;
    bool calibration = true;     // rt_set_calibration
    bool shade_shadowed = false; // the last batch ran without shadow launches
    bool inline_shadows = false; // brute-force scene whose primitives fit the shade kernel's LDS (RT_INLINE_SHADOWS=0 turns it off)
    bool team_walks = true;      // RT_TEAM_WALKS=0: calibration leaves the team walk out
    uint32_t tiles_cap = 0;    // batches of this scene that overflowed a pool that could not grow were split: keep to that size
    bool count_work = false;   // traversal kernels count fetches and tests (rt_set_work_counters)
    int stage_times = 0;       // HIP events between the stage launches (rt_set_stage_times / RT_STAGE_TIMES): 1 every stage, 2 only the extend launches, 3 only the shadow launches
    rt_progress_fn progress = nullptr; void* progress_user = nullptr;
    uint64_t grown_capacity = 0;   // pool size a single-tile batch needed beyond an explicit rt_options.pool_rays
    RtkLaunchCache launch_cache{}; // resident-block counts of the traversal kernels on THIS device
    std::vector<hipEvent_t> event_pool;   // timing events of the stage times, reused from frame to frame
    bool two_children = false;
    // Overlapped pipeline for thin batches: shadow + lighting of generation g run on a second stream while
    // the first traces generation g + 1.  The per-generation hit records and the hit list are double-buffered
    // for it (generation parity), events order the two streams.
    uint64_t overlap_rays = 8u << 20;      // generations expected to hold at most this many rays are overlapped (0: never)
    int reported_first_ov = -1;            // RT_VERBOSE: the last overlap start that was printed
    // rays per generation of a batch: what the last frame's batches held (the calibration sample, scaled, before
    // any frame); 0 = unknown.  Only decides WHERE the overlap starts, never a result.
    uint64_t gen_rays[RT_MAX_GEN + 2] = {0}, gen_rays_frame[RT_MAX_GEN + 2] = {0};
    bool gen_rays_known = false;
    hipStream_t stream2 = nullptr;
    float4 *hp2 = nullptr, *hn2 = nullptr, *hd2 = nullptr; uint32_t* hitlist2 = nullptr; uint64_t alt_capacity = 0;
    std::vector<hipEvent_t> ev_scan, ev_light; // some material can send out a reflection AND a refraction ray: resolve passes needed
    DScene S{};
    DCamera cam{};
    float4* d_sph = nullptr; int* d_sph_mat = nullptr;
    float4* d_tri = nullptr; float4* d_tri_n = nullptr;
    float4* d_mats = nullptr; DLight* d_lights = nullptr;
    int* d_prim_idx = nullptr; float4* d_leaf = nullptr;
    // The hierarchies built over the same leaf records at load time: Karras (always), PLOC, binned surface-area build
    // (where built).  The first render keeps one (calibrate / adopt_tree) and frees the others; tree_in_use = -1 until then.
    float4* t_nodes[RT_TREES] = {nullptr, nullptr, nullptr};
    float4* t_wide[RT_TREES] = {nullptr, nullptr, nullptr};      // their 4-wide collapses (per-lane walks only); nullptr where not built
    int t_depth[RT_TREES] = {0, 0, 0};
    double t_cost[RT_TREES][2] = {{0, 0}, {0, 0}, {0, 0}};      // surface-area expectation: node visits, primitive tests (rtl_tree_cost)
    int tree_in_use = -1;
    float region[6] = {0, 0, 0, 0, 0, 0};  // box of every point a ray of a render can start from (camera origins + primitives): what the hierarchies are conservative for

    // ray pool
    DPool P{};
    DCounters* d_counters = nullptr;
    DCounters* h_counters = nullptr;   // pinned: where the counters of a batch are read back to (see copy_out)
    uint64_t pool_capacity = 0;
    uint64_t scol_capacity = 0;
    uint64_t occl_capacity = 0;

    // partition
    int part_rank = -1, part_world = -1, part_w = -1, part_h = -1;
    std::vector<uint32_t> tiles;
    uint32_t* d_tiles = nullptr;
    uint32_t* d_rowbase = nullptr; int rowbase_world = -1, rowbase_w = -1, rowbase_h = -1;
    uint32_t* scratch = nullptr;    // this context's slice of its device's allocation-free scratch array
    int scratch_slot = -1;          // which slice (-1: all were taken when the context was created)

    // staging
    uint8_t* d_rgb = nullptr; float* d_rgbf = nullptr; size_t rgb_pixels = 0; size_t rgbf_pixels = 0;
    uint8_t* d_shard = nullptr; size_t shard_cap = 0;      // rt_render_multi: this context's shard
    uint8_t* d_all = nullptr; size_t all_cap = 0;          // rt_render_multi, context 0: every shard
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // rt_render's way out: the frame goes to pinned staging memory in chunks and host threads move the chunks
    // on into the caller's pageable buffer as they arrive (RT_FAST_COPY=0: one plain copy)
    uint8_t* h_stage = nullptr; size_t h_stage_cap = 0; bool fast_copy = true;
    std::vector<hipEvent_t> copy_events;

    // Thin batches replayed as HIP graphs: a frame of a few hundred microseconds is a chain of dozens of launches, and
    // queueing them one by one costs the host about as long as the device needs to run them.  A batch whose launches
    // would be the same as those of an earlier batch (same arguments by value: scene, pool, camera, tiles, walks,
    // stream split) is captured the second time it comes and replayed from then on.  RT_GRAPHS=0 turns it off,
    // RT_GRAPH_SAMPLES=n moves the size limit (batches of more samples are queued directly: their enqueue time is hidden).
    struct BatchGraph { std::vector<uint8_t> key; int seen = 0; bool failed = false; hipGraphExec_t exec = nullptr; };
    std::vector<BatchGraph> graphs;
    bool graphs_on = true;
    uint64_t graph_max_samples = 16u << 20;
    uint64_t graph_replays = 0;    // batches of this context that ran as a graph launch
};

static void drop_graphs(rt_ctx* c) {
    for (auto& g : c->graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    c->graphs.clear();
}

static void free_scene(rt_ctx* c) {
    drop_graphs(c);
    hipFree(c->d_sph); hipFree(c->d_sph_mat); hipFree(c->d_tri); hipFree(c->d_tri_n);
    hipFree(c->d_mats); hipFree(c->d_lights); hipFree(c->d_prim_idx); hipFree(c->d_leaf);
    for (int t = 0; t < RT_TREES; t++) {
        hipFree(c->t_nodes[t]); hipFree(c->t_wide[t]);
        c->t_nodes[t] = c->t_wide[t] = nullptr; c->t_depth[t] = 0; c->t_cost[t][0] = c->t_cost[t][1] = 0;
    }
    c->tree_in_use = -1;
    c->d_sph = nullptr; c->d_sph_mat = nullptr; c->d_tri = nullptr; c->d_tri_n = nullptr;
    c->d_mats = nullptr; c->d_lights = nullptr; c->d_prim_idx = nullptr; c->d_leaf = nullptr;
    c->has_scene = false;
}
static void free_pool(rt_ctx* c) {
    hipFree(c->P.o); hipFree(c->P.d); hipFree(c->P.hp); hipFree(c->P.hn); hipFree(c->P.hd); hipFree(c->P.loc);
    hipFree(c->P.cre); hipFree(c->P.cra); hipFree(c->P.link); hipFree(c->P.wstats); hipFree(c->P.occl); hipFree(c->P.scol);
    hipFree(c->P.mask_hit); hipFree(c->P.mask_refl); hipFree(c->P.mask_refr);
    hipFree(c->P.child_base); hipFree(c->P.scan_state); hipFree(c->P.hitlist); hipFree(c->P.rflag);   // (scan_ticket and wq lie in scan_state's allocation)
    hipFree(c->hp2); hipFree(c->hn2); hipFree(c->hd2); hipFree(c->hitlist2);
    c->hp2 = c->hn2 = c->hd2 = nullptr; c->hitlist2 = nullptr; c->alt_capacity = 0;
    c->P = DPool{};
    c->pool_capacity = c->scol_capacity = c->occl_capacity = 0;
}

extern "C" int rt_create(rt_ctx** out, const rt_options* opt) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        rt_set_error("no HIP device is visible; this library has no CPU fallback");
        return RT_ERR_NO_DEVICE;
    }
    // owned until the last step succeeds: any early return below releases what was created so far
    struct Guard { rt_ctx* c; ~Guard() { if (c) rt_destroy(c); } } guard{new rt_ctx()};
    rt_ctx* c = guard.c;
    if (opt) c->opt = *opt;
    for (int g = 0; g <= RT_MAX_GEN; g++) { c->mode_extend[g] = 3; c->mode_shadow[g] = 3; }
    if (const char* e = getenv("RT_COUNT_WORK")) c->count_work = atoi(e) != 0;
    if (const char* e = getenv("RT_STAGE_TIMES")) c->stage_times = (atoi(e) == 2 || atoi(e) == 3) ? atoi(e) : 1;
    if (const char* e = getenv("RT_OVERLAP_RAYS")) c->overlap_rays = strtoull(e, nullptr, 10);
    if (const char* e = getenv("RT_TEAM_WALKS")) c->team_walks = atoi(e) != 0;
    if (const char* e = getenv("RT_FAST_COPY")) c->fast_copy = atoi(e) != 0;
    if (const char* e = getenv("RT_GRAPHS")) c->graphs_on = atoi(e) != 0;
    if (const char* e = getenv("RT_GRAPH_SAMPLES")) c->graph_max_samples = strtoull(e, nullptr, 10);
    if (const char* e = getenv("RT_PACKET_GENS")) {
        c->forced_modes = true;
        // RT_LANE_WIDE: the other generations walk the 4-wide tree per lane (1) or per team of lanes (2)
        const int lane_mode = getenv("RT_LANE_WIDE") ? (atoi(getenv("RT_LANE_WIDE")) == 2 ? 5 : 4) : 2;
        for (int g = 0; g <= RT_MAX_GEN; g++) c->mode_extend[g] = g < atoi(e) ? 3 : lane_mode;
    }
    if (const char* e = getenv("RT_PACKET_SHADOW_GENS")) {
        c->forced_modes = true;
        const int lane_mode = getenv("RT_LANE_WIDE") ? (atoi(getenv("RT_LANE_WIDE")) == 2 ? 5 : 4) : 2;
        for (int g = 0; g <= RT_MAX_GEN; g++) c->mode_shadow[g] = g < atoi(e) ? 3 : lane_mode;
    }
    c->device = c->opt.device;
    if (c->device < 0 || c->device >= ndev) { c->device = 0; rt_set_error("device ordinal out of range"); return RT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->max_blocks = c->num_cu * 8;
    // test hook: a small grid makes films of a few thousand rays take the paths of large frames (several chunks per
    // wave in the per-lane walks, tiles shared by ticket in the scan)
    if (const char* e = getenv("RT_MAX_BLOCKS")) c->max_blocks = std::max(8, atoi(e));
    if (c->opt.stream) { c->stream = (hipStream_t)c->opt.stream; c->own_stream = false; }
    else { HIP_TRY(acquire_stream(c->device, &c->stream)); c->own_stream = true; }
    HIP_TRY(hipMalloc(&c->d_counters, sizeof(DCounters)));
    HIP_TRY(hipHostMalloc((void**)&c->h_counters, sizeof(DCounters), hipHostMallocDefault));
    {
        int src = scratch_on_device(c->device, &c->scratch, &c->scratch_slot);     // the device is current here
        if (src != RT_OK) return src;
    }
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    guard.c = nullptr;
    *out = c;
    return RT_OK;
}

extern "C" void rt_destroy(rt_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    // both streams idle before anything they may still use goes away
    if (c->stream2) hipStreamSynchronize(c->stream2);
    if (c->stream) hipStreamSynchronize(c->stream);
    drop_graphs(c);
    free_scene(c);
    free_pool(c);
    if (c->h_counters) hipHostFree(c->h_counters);
    hipFree(c->d_counters); hipFree(c->d_tiles); hipFree(c->d_rowbase); hipFree(c->d_rgb); hipFree(c->d_rgbf);
    hipFree(c->d_shard); hipFree(c->d_all);
    if (c->h_stage) hipHostFree(c->h_stage);
    // the streams go back to the process-wide list (see acquire_stream: they are never destroyed); events last
    recycle_stream(c->device, c->stream2);
    if (c->own_stream) recycle_stream(c->device, c->stream);
    scratch_release(c->device, c->scratch_slot);
    for (auto e : c->event_pool) hipEventDestroy(e);
    for (auto e : c->ev_scan) hipEventDestroy(e);
    for (auto e : c->ev_light) hipEventDestroy(e);
    for (auto e : c->copy_events) hipEventDestroy(e);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    (void)hipGetLastError();
    delete c;
}

// ---------------------------------------------------------------- camera
// Frame constants exactly as the reference derives them on the host:
// Scene::getPlaneDist (src/parse.cpp:275-277), RayTrace::getExtremePoints
// (src/raytrace.cpp:635-647), and the loop-invariant head of RayTrace::getRay
// (src/raytrace.cpp:443-445, :457-462).
static void compute_camera(const rt_scene_desc& s, DCamera& cam) {
    F3 pos = f3(s.cam_pos[0], s.cam_pos[1], s.cam_pos[2]);
    F3 dir = f3(s.cam_dir[0], s.cam_dir[1], s.cam_dir[2]);
    F3 up = f3(s.cam_up[0], s.cam_up[1], s.cam_up[2]);
    float pdf = (float)(s.height / (2.0 * std::tan((double)s.cam_half_angle)));
    double d = pdf, w = s.width, h = s.height;
    F3 right = cross3(dir, up);
    F3 p0 = pos + (dir * (float)d);
    F3 p1 = p0 + (right * (float)(w / 2.0));
    F3 p2 = p0 - (right * (float)(w / 2.0));
    p1 = p1 + (up * (float)(h / 2.0));
    p2 = p2 - (up * (float)(h / 2.0));
    F3 rightn = norm3(cross3(dir, up));
    F3 down = norm3(up * -1.0f);
    F3 ptop = p2 - p1;
    cam.pos = pos; cam.dir = dir; cam.p1 = p1;
    cam.urv = rightn * dot3(ptop, rightn);
    cam.blv = down * dot3(ptop, down);
    cam.width = s.width; cam.height = s.height; cam.spp = s.spp; cam.proj = s.proj;
    cam.seed = s.seed;
    cam.inv_w = 1.0 / (double)s.width; cam.inv_h = 1.0 / (double)s.height;
    cam.spp_shift = -1;
    for (int k = 0; k < 31; k++) if (s.spp == (1 << k)) cam.spp_shift = k;
    // jitter counter = ((y*W + x)*spp + j)*2 + k, largest value below W*H*spp*2
    cam.counter32 = ((double)s.width * (double)s.height * (double)s.spp * 2.0 <= 4294967296.0) ? 1u : 0u;
    cam.jitter_hi0 = rt_lowbias32(s.seed * 0x9E3779B9U + 0x85EBCA6BU);
}

// ---------------------------------------------------------------- scene upload
template <typename T>
static int upload(T** dst, const std::vector<T>& src, hipStream_t st) {
    *dst = nullptr;
    size_t n = src.size() ? src.size() : 1;
    HIP_TRY(hipMalloc((void**)dst, n * sizeof(T)));
    if (src.size()) HIP_TRY(hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return RT_OK;
}

static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

extern "C" int rt_validate_scene(const rt_scene_desc* s) {
    if (!s) return RT_ERR_INVALID;
    if (s->width <= 0 || s->height <= 0 || s->spp <= 0 || s->n_spheres < 0 || s->n_triangles < 0 ||
        s->n_lights < 0 || s->n_materials < 0) {
        rt_set_error("scene descriptor: non-positive film size / sample rate or negative count");
        return RT_ERR_INVALID;
    }
    if ((s->n_spheres > 0 && (!s->spheres || !s->sphere_mat)) ||
        (s->n_triangles > 0 && (!s->tri_verts || !s->tri_normals || !s->tri_mat || !s->tri_flags)) ||
        (s->n_materials > 0 && !s->materials) || (s->n_lights > 0 && !s->lights)) {
        rt_set_error("scene descriptor: a count is positive but its array is NULL");
        return RT_ERR_INVALID;
    }
    if (s->max_depth >= RT_MAX_GEN) { rt_set_error("max_depth beyond the supported 63"); return RT_ERR_UNSUPPORTED; }
    if ((uint64_t)s->width > 65535u * RT_TILE || (uint64_t)s->height > 65535u * RT_TILE) {
        rt_set_error("film beyond 65535 tiles a side");
        return RT_ERR_UNSUPPORTED;
    }
    // node and leaf offsets are 32-bit byte offsets of 64-byte records; a hit's record packs material << 2
    if ((uint64_t)s->n_spheres + (uint64_t)s->n_triangles > (1ull << 26)) {
        rt_set_error("more than 2^26 primitives");
        return RT_ERR_UNSUPPORTED;
    }
    if ((uint64_t)s->n_materials >= (1ull << 29)) { rt_set_error("2^29 or more materials"); return RT_ERR_UNSUPPORTED; }
    for (int i = 0; i < s->n_spheres; i++)
        if (s->sphere_mat[i] < 0 || s->sphere_mat[i] >= s->n_materials) { rt_set_error("sphere material index out of range"); return RT_ERR_INVALID; }
    for (int i = 0; i < s->n_triangles; i++)
        if (s->tri_mat[i] < 0 || s->tri_mat[i] >= s->n_materials) { rt_set_error("triangle material index out of range"); return RT_ERR_INVALID; }
    return RT_OK;
}

extern "C" int rt_load_scene(rt_ctx* c, const rt_scene_desc* s) {
    if (!c || !s) return RT_ERR_INVALID;
    int vrc = rt_validate_scene(s);
    if (vrc != RT_OK) return vrc;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_scene(c);

    std::vector<float4> sph(s->n_spheres);
    std::vector<int> sph_mat(s->n_spheres);
    for (int i = 0; i < s->n_spheres; i++) {
        const float* p = s->spheres + 4 * (size_t)i;
        sph[i] = make_float4(p[0], p[1], p[2], p[3]);
        sph_mat[i] = s->sphere_mat[i];
    }
    // triangle record: a = v0, ab = v0 - v1, ac = v0 - v2 -- the float subtractions
    // the reference performs per test (src/raytrace.cpp:316-317), done once.
    std::vector<float4> tri(3 * (size_t)s->n_triangles), trn(3 * (size_t)s->n_triangles);
    for (int i = 0; i < s->n_triangles; i++) {
        const float* v = s->tri_verts + 9 * (size_t)i;
        const float* n = s->tri_normals + 9 * (size_t)i;
        F3 v0 = f3(v[0], v[1], v[2]), v1 = f3(v[3], v[4], v[5]), v2 = f3(v[6], v[7], v[8]);
        F3 ab = v0 - v1, ac = v0 - v2;
        tri[3 * (size_t)i + 0] = make_float4(v0.x, v0.y, v0.z, ab.x);
        tri[3 * (size_t)i + 1] = make_float4(ab.y, ab.z, ac.x, ac.y);
        tri[3 * (size_t)i + 2] = make_float4(ac.z, u2f((uint32_t)s->tri_mat[i]), u2f(s->tri_flags[i] ? 1u : 0u), 0.0f);
        for (int k = 0; k < 3; k++) trn[3 * (size_t)i + k] = make_float4(n[3 * k], n[3 * k + 1], n[3 * k + 2], 0.0f);
    }
    std::vector<float4> mats(4 * (size_t)s->n_materials);
    for (int i = 0; i < s->n_materials; i++) {
        const float* m = s->materials + 14 * (size_t)i;
        mats[4 * (size_t)i + 0] = make_float4(m[0], m[1], m[2], m[9]);     // ambient, ns
        mats[4 * (size_t)i + 1] = make_float4(m[3], m[4], m[5], m[13]);    // diffuse, ior
        mats[4 * (size_t)i + 2] = make_float4(m[6], m[7], m[8], 0.0f);     // specular
        mats[4 * (size_t)i + 3] = make_float4(m[10], m[11], m[12], 0.0f);  // transmissive
    }
    // A hit spawns a reflection ray iff every specular component is non-zero and a refraction ray iff
    // every transmissive one is (src/raytrace.cpp:589, :600): only a material with both can give a ray
    // two children, and only such rays need the bottom-up resolve passes.
    c->two_children = false;
    for (int i = 0; i < s->n_materials; i++) {
        const float* m = s->materials + 14 * (size_t)i;
        if (m[6] != 0 && m[7] != 0 && m[8] != 0 && m[10] != 0 && m[11] != 0 && m[12] != 0) c->two_children = true;
    }
    std::vector<DLight> lights(s->n_lights);
    for (int i = 0; i < s->n_lights; i++) {
        const rt_light& L = s->lights[i];
        DLight& D = lights[i];
        D.type = L.type;
        D.cr = L.color[0]; D.cg = L.color[1]; D.cb = L.color[2];
        D.px = L.pos[0]; D.py = L.pos[1]; D.pz = L.pos[2];
        D.dx = L.dir[0]; D.dy = L.dir[1]; D.dz = L.dir[2];
        D.a1 = L.angle1; D.a2 = L.angle2;
    }
    int rc;
    if ((rc = upload(&c->d_sph, sph, c->stream)) != RT_OK) return rc;
    if ((rc = upload(&c->d_sph_mat, sph_mat, c->stream)) != RT_OK) return rc;
    if ((rc = upload(&c->d_tri, tri, c->stream)) != RT_OK) return rc;
    if ((rc = upload(&c->d_tri_n, trn, c->stream)) != RT_OK) return rc;
    if ((rc = upload(&c->d_mats, mats, c->stream)) != RT_OK) return rc;
    if ((rc = upload(&c->d_lights, lights, c->stream)) != RT_OK) return rc;

    DScene& S = c->S;
    S = DScene{};
    S.sph = c->d_sph; S.sph_mat = c->d_sph_mat; S.tri = c->d_tri; S.tri_n = c->d_tri_n;
    S.mats = c->d_mats; S.lights = c->d_lights;
    S.ns = s->n_spheres; S.nt = s->n_triangles; S.nl = s->n_lights; S.max_depth = s->max_depth; S.nm = s->n_materials;
    S.bg = f3(s->background[0], s->background[1], s->background[2]);
    S.ambient = f3(s->ambient[0], s->ambient[1], s->ambient[2]);
    S.n_prims = S.ns + S.nt;
    S.profile = getenv("RT_PROFILE") != nullptr;
    compute_camera(*s, c->cam);

    // brute force <-> LBVH: the caller's threshold, else the scene file's bvh_threshold (the switch the
    // reference intended, src/raytrace.cpp:61-69, src/parse.cpp:252-256), else 64 primitives
    int thr = c->opt.lbvh_threshold > 0 ? c->opt.lbvh_threshold : (s->bvh_threshold > 0 ? s->bvh_threshold : 64);
    int accel = c->opt.accel;
    if (accel == RT_ACCEL_AUTO) accel = (S.n_prims > thr) ? RT_ACCEL_LBVH : RT_ACCEL_BRUTE;
    // the LBVH is also built when brute force renders, if rt_query_rays may ask for it
    if (S.n_prims >= 2) {
        int radius = 16;                                     // RT_PLOC_RADIUS=0: Karras hierarchy only
        if (const char* e = getenv("RT_PLOC_RADIUS")) radius = std::max(0, std::min(256, atoi(e)));
        // 4-wide nodes with float boxes (128 B) while the tree fits the L2 caches (32 MB of them at 2^19 primitives):
        // there a walk's step is latency-bound and decoding bytes only lengthens it (K4: 2.2 ms against 2.4); beyond,
        // with 8-bit boxes (64 B): the walks are bound by the bytes they move (K5: 68 ms against 73)
        S.wide_quant = S.n_prims > (1 << 19) ? 1 : 0;
        if (const char* e = getenv("RT_WIDE_QUANT")) S.wide_quant = atoi(e) != 0;
        // where the camera's rays start: the eye, or the film parallelogram of an orthographic camera (src/raytrace.cpp:441-485)
        float origin_box[6];
        {
            const DCamera& cam = c->cam;
            F3 lo = cam.pos, hi = cam.pos;
            if (cam.proj != RT_PROJ_PERSP) {
                const F3 corner[4] = {cam.p1, cam.p1 + cam.urv, cam.p1 + cam.blv, (cam.p1 + cam.urv) + cam.blv};
                lo = hi = corner[0];
                for (int k = 1; k < 4; k++) {
                    lo = f3(std::min(lo.x, corner[k].x), std::min(lo.y, corner[k].y), std::min(lo.z, corner[k].z));
                    hi = f3(std::max(hi.x, corner[k].x), std::max(hi.y, corner[k].y), std::max(hi.z, corner[k].z));
                }
            }
            origin_box[0] = lo.x; origin_box[1] = lo.y; origin_box[2] = lo.z;
            origin_box[3] = hi.x; origin_box[4] = hi.y; origin_box[5] = hi.z;
        }
        // the binned surface-area build pays for scenes whose frames are bound by tests per ray, not by memory: up to 2^18
        // primitives by default (a wave per node: the top levels of a larger scene would take longer than they save)
        int sah_max = 1 << 18;
        if (const char* e = getenv("RT_SAH_MAX_PRIMS")) sah_max = std::max(0, atoi(e));
        hipError_t e = (hipError_t)rtl_build(c->stream, c->d_sph, S.ns, c->d_tri, S.nt, &c->t_nodes[RT_TREE_KARRAS], &c->d_prim_idx, &c->d_leaf,
                                             &c->t_depth[RT_TREE_KARRAS], radius, &c->t_nodes[RT_TREE_PLOC], &c->t_depth[RT_TREE_PLOC],
                                             &c->t_wide[RT_TREE_KARRAS], &c->t_wide[RT_TREE_PLOC], S.wide_quant, origin_box, c->region,
                                             sah_max, &c->t_nodes[RT_TREE_SAH], &c->t_depth[RT_TREE_SAH], &c->t_wide[RT_TREE_SAH]);
        if (e != hipSuccess) { rt_set_error(std::string("LBVH build: ") + hipGetErrorString(e)); return RT_ERR_HIP; }
        for (int t = 0; t < RT_TREES; t++)
            if (c->t_nodes[t] && rtl_tree_cost(c->stream, c->t_nodes[t], S.n_prims - 1, c->t_cost[t]) != (int)hipSuccess) (void)hipGetLastError();
        S.nodes = c->t_nodes[RT_TREE_KARRAS]; S.nodes4 = c->t_wide[RT_TREE_KARRAS]; S.prim_idx = c->d_prim_idx; S.leaf = c->d_leaf; S.root = 0;
        S.depth = c->t_depth[RT_TREE_KARRAS];
        if (getenv("RT_VERBOSE")) {
            fprintf(stderr, "[rt bvh] %d primitives;", S.n_prims);
            for (int t = 0; t < RT_TREES; t++) {
                if (c->t_nodes[t]) fprintf(stderr, " %s depth %d, expected node visits %.2f + primitive tests %.2f per line through the root box;",
                                           kTreeName[t], c->t_depth[t], c->t_cost[t][0], c->t_cost[t][1]);
                else fprintf(stderr, " %s not built;", kTreeName[t]);
            }
            fprintf(stderr, " PLOC radius %d\n", radius);
        }
    } else {
        accel = RT_ACCEL_BRUTE;
    }
    c->accel = accel;
    c->inline_shadows = rtk_shade_can_shadow(S.ns, S.nt) && !(getenv("RT_INLINE_SHADOWS") && atoi(getenv("RT_INLINE_SHADOWS")) == 0);
    c->tuned = false;
    c->gen_rays_known = false;
    for (int g = 0; g <= RT_MAX_GEN; g++) c->gen_rays[g] = c->gen_rays_frame[g] = 0;
    c->tiles_cap = 0;
    c->grown_capacity = 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->has_scene = true;
    return RT_OK;
}

// ---------------------------------------------------------------- partition
static inline int tile_owner(int tx, int ty, int world) { return (tx + 3 * ty) % world; }

extern "C" int rt_shard_tiles(int width, int height, int rank, int world) {
    if (width <= 0 || height <= 0 || world <= 0 || rank < 0 || rank >= world) return 0;
    int ntx = (width + RT_TILE - 1) / RT_TILE, nty = (height + RT_TILE - 1) / RT_TILE;
    int n = 0;
    for (int ty = 0; ty < nty; ty++)
        for (int tx = 0; tx < ntx; tx++)
            if (tile_owner(tx, ty, world) == rank) n++;
    return n;
}

extern "C" size_t rt_shard_bytes(int width, int height, int world) {
    int m = 0;
    for (int r = 0; r < world; r++) m = std::max(m, rt_shard_tiles(width, height, r, world));
    return (size_t)m * RT_TILE * RT_TILE * 3;
}

static int set_partition(rt_ctx* c, int rank, int world) {
    int w = c->cam.width, h = c->cam.height;
    if (c->part_rank == rank && c->part_world == world && c->part_w == w && c->part_h == h) return RT_OK;
    int ntx = (w + RT_TILE - 1) / RT_TILE, nty = (h + RT_TILE - 1) / RT_TILE;
    c->tiles.clear();
    for (int ty = 0; ty < nty; ty++)
        for (int tx = 0; tx < ntx; tx++)
            if (tile_owner(tx, ty, world) == rank) c->tiles.push_back((uint32_t)tx | ((uint32_t)ty << 16));
    hipFree(c->d_tiles); c->d_tiles = nullptr;
    HIP_TRY(hipMalloc(&c->d_tiles, std::max<size_t>(1, c->tiles.size()) * sizeof(uint32_t)));
    if (!c->tiles.empty())
        HIP_TRY(hipMemcpyAsync(c->d_tiles, c->tiles.data(), c->tiles.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->part_rank = rank; c->part_world = world; c->part_w = w; c->part_h = h;
    return RT_OK;
}

// ---------------------------------------------------------------- pool
static int ensure_pool(rt_ctx* c, uint64_t capacity, uint64_t samples) {
    uint64_t occl = capacity * (uint64_t)std::max(1, c->S.nl);
    if (c->pool_capacity >= capacity && c->scol_capacity >= samples && c->occl_capacity >= occl) {
        c->P.capacity = (uint32_t)capacity;
        return RT_OK;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_pool(c);
    if (capacity > 0x7FFFFFF0ull) { rt_set_error("ray pool beyond 2^31 records"); return RT_ERR_INVALID; }
    const auto pool_t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipMalloc(&c->P.o, capacity * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->P.d, capacity * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->P.hp, capacity * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->P.hn, capacity * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->P.hd, capacity * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->P.loc, capacity * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->P.cre, capacity * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->P.cra, capacity * sizeof(float4)));
    uint64_t groups = capacity / 64 + RT_MAX_GEN + 4, sblocks = groups / 256 + 2;
    HIP_TRY(hipMalloc(&c->P.mask_hit, groups * 8));
    HIP_TRY(hipMalloc(&c->P.mask_refl, groups * 8));
    HIP_TRY(hipMalloc(&c->P.mask_refr, groups * 8));
    HIP_TRY(hipMalloc(&c->P.child_base, groups * 4));
    // everything a batch needs zeroed lies in ONE allocation, cleared by one fill per batch: the scan's state words
    // (generation g starts at word (groups before it >> 8) + g), its tickets, then the work-queue counters of
    // the traversal launches (two launches per generation; only the generations in use are cleared)
    c->P.scan_words = sblocks + RT_MAX_GEN + 8;
    const size_t ticket_bytes = ((RT_MAX_GEN + 1) * sizeof(uint32_t) + 127) / 128 * 128;
    HIP_TRY(hipMalloc(&c->P.scan_state, c->P.scan_words * sizeof(unsigned long long) + ticket_bytes +
                                            (size_t)2 * (RT_MAX_GEN + 1) * RT_WQ_LAUNCH * sizeof(uint32_t)));
    c->P.scan_ticket = (uint32_t*)(c->P.scan_state + c->P.scan_words);
    c->P.wq = (uint32_t*)((char*)c->P.scan_ticket + ticket_bytes);
    HIP_TRY(hipMalloc(&c->P.hitlist, capacity * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&c->P.rflag, (capacity + 63) / 64 * 64));
    HIP_TRY(hipMalloc(&c->P.occl, occl));
    HIP_TRY(hipMalloc(&c->P.scol, samples * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->P.link, capacity * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&c->P.wstats, (size_t)2 * RT_MAX_WAVES * 8 * sizeof(unsigned long long)));
    c->P.capacity = (uint32_t)capacity;
    c->pool_capacity = capacity; c->scol_capacity = samples; c->occl_capacity = occl;
    if (getenv("RT_VERBOSE"))
        fprintf(stderr, "[rt pool] %llu rays, %llu samples: %.2f GB allocated in %.2f ms\n", (unsigned long long)capacity,
                (unsigned long long)samples, ((double)capacity * (132.0 + 8.0 / 64 * 3 + 1 + std::max(1, c->S.nl)) + (double)samples * 16.0) / 1e9,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - pool_t0).count());
    return RT_OK;
}

// ---------------------------------------------------------------- batch scheduler
struct BatchResult {
    bool overflow = false;
    uint64_t secondary = 0, shadow = 0, nodes[2] = {0, 0}, sph[2] = {0, 0}, tri[2] = {0, 0}, lane_nodes[2] = {0, 0}, lane_prims[2] = {0, 0};
    int generations = 0;
    double stage_b[4] = {0, 0, 0, 0};      // stage times of the second stream (overlapped pipeline)
    double host_enqueue_ms = 0;            // wall time of queueing the launches
};

struct StageTimes {
    bool on = false;
    std::vector<hipEvent_t> ev;      // boundaries
    std::vector<int> cat;            // category of the span that ENDS at ev[i+1]
    hipStream_t st = nullptr;
    std::vector<hipEvent_t>* pool = nullptr;   // the context's spare events: creating one costs microseconds, a frame uses dozens
    int only = -1;                   // >= 0: only spans of this category are timed (an event before and one after each: an event costs
                                     // the stream a few microseconds, which a frame of a few hundred notices)
    StageTimes() = default;
    StageTimes(const StageTimes&) = delete;
    StageTimes& operator=(const StageTimes&) = delete;
    ~StageTimes() { release(); }     // every exit path of the scheduler gives the events back
    void release() {
        for (auto e : ev) {
            if (pool) pool->push_back(e);
            else hipEventDestroy(e);
        }
        ev.clear(); cat.clear();
    }
    void begin(int category) {       // start of a span of `category` (needed only when the spans before it are not timed)
        if (on && only >= 0 && only == category) record(-1);
    }
    void mark(int category) {
        if (!on || (only >= 0 && category != only)) return;
        record(category);
    }
    void record(int category) {
        hipEvent_t e = nullptr;
        if (pool && !pool->empty()) { e = pool->back(); pool->pop_back(); }
        if ((!e && hipEventCreate(&e) != hipSuccess) || hipEventRecord(e, st) != hipSuccess) {
            // no stage times for this frame rather than times with a hole in them
            if (e) hipEventDestroy(e);
            (void)hipGetLastError();
            on = false;
            release();
            return;
        }
        ev.push_back(e); cat.push_back(category);
    }
    void collect(double out[4]) {
        if (!on) return;
        for (size_t i = 1; i < ev.size(); i++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev[i - 1], ev[i]) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (cat[i] >= 0) out[cat[i]] += ms;          // category -1: a wait for the other stream, nobody's time
        }
        release();
    }
};

// What the overlapped pipeline needs beyond the pool: a second stream, one event pair per generation and
// the second copy of the per-generation buffers.
static int ensure_overlap(rt_ctx* c) {
    if (!c->stream2) HIP_TRY(acquire_stream(c->device, &c->stream2));
    while (c->ev_scan.size() < (size_t)RT_MAX_GEN + 1) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        c->ev_scan.push_back(a);
        HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        c->ev_light.push_back(b);
    }
    if (c->alt_capacity < c->pool_capacity) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->hp2); hipFree(c->hn2); hipFree(c->hd2); hipFree(c->hitlist2);
        c->hp2 = c->hn2 = c->hd2 = nullptr; c->hitlist2 = nullptr; c->alt_capacity = 0;
        HIP_TRY(hipMalloc(&c->hp2, c->pool_capacity * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->hn2, c->pool_capacity * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->hd2, c->pool_capacity * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->hitlist2, c->pool_capacity * sizeof(uint32_t)));
        c->alt_capacity = c->pool_capacity;
    }
    return RT_OK;
}

// Queues the launches of one batch on the context's streams (or, while stream A is being captured, into the graph).
// On an error nothing is left split: work already queued on the second stream is joined back into the first, so that
// the caller can wait for the first stream alone and a capture in progress can end.
static int enqueue_batch(rt_ctx* c, uint32_t tile0, uint32_t ntiles, int shard_mode, uint8_t* d_out, float* d_outf,
                         int first_ov, StageTimes& tm, StageTimes& tmB) {
    const DScene& S = c->S;
    const uint32_t npixels = ntiles * RT_TILE_AREA;
    const uint32_t nsamples = (uint32_t)((uint64_t)npixels * (uint32_t)c->cam.spp);
    hipStream_t st = c->stream;
    hipStream_t B = c->stream2;
    const uint64_t cap = c->P.capacity;
    const bool any_ov = first_ov <= S.max_depth;
    int rc = RT_OK;
    int last_b = -1;               // last generation whose second-stream work was queued (-1: none yet)
    // A launch that the runtime rejects (bad configuration, missing code object) is reported by the next
    // hipGetLastError, not by the launch: ask after every launch group and name the stage, so a failure does
    // not surface as an anonymous error at the final synchronisation.
    auto failed = [&](const char* stage, int g) {
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) return false;
        char b[200];
        std::snprintf(b, sizeof(b), "launch of %s (generation %d): %s", stage, g, hipGetErrorString(e));
        rt_set_error(b);
        rc = RT_ERR_HIP;
        return true;
    };
    tm.mark(3);
    rtk_init_counters(st, c->d_counters, nsamples, c->P, S.max_depth + 1, c->count_work || S.profile);
    tm.mark(3);
    if (S.max_depth < 0) rtk_fill_background(st, S, c->P, nsamples, c->max_blocks);   // nothing is traced at all
    if (failed("init", -1)) return rc;
    uint64_t bound = nsamples;
    // The generations of the batch.  Fat ones run one after the other on the context's stream (A): extend ->
    // scan -> shadow -> shade.  From the first generation expected to be thin on (few rays: launches that last
    // as long as their longest walk and leave most of the chip idle) the work is split over two streams:
    //   A: extend(g) -> scan(g) -> spawn(g) -> extend(g+1) ...     B: miss(g) -> shadow(g) -> light(g)
    // B starts a generation when its scan is done.  The colour of a ray is final only when its parent's
    // record exists, so everything that hands colours on (the misses of generation g, the leaves among its
    // hits) runs on B, in generation order, after the lighting of g - 1.  Buffers written per generation by A
    // and read by B (hit records, hit list) alternate between two copies; A reuses a copy once the lighting
    // of the generation before last has read it.
    for (int g = 0; g <= S.max_depth && rc == RT_OK; g++) {
        // LBVH: generations whose rays are still coherent walk the tree as wave packets
        int acc_e = c->accel, acc_s = c->accel;
        if (c->accel == RT_ACCEL_LBVH) { acc_e = c->mode_extend[g]; acc_s = c->mode_shadow[g]; }
        if (acc_e >= 4 && !S.nodes4) acc_e = 2;
        if (acc_s >= 4 && !S.nodes4) acc_s = 2;
        const bool ov = g >= first_ov;
        DPool P = c->P;
        if (any_ov && (g & 1)) { P.hp = c->hp2; P.hn = c->hn2; P.hd = c->hd2; P.hitlist = c->hitlist2; }
        if (ov && g - 2 >= first_ov) { hipStreamWaitEvent(st, c->ev_light[g - 2], 0); tm.mark(-1); }
        tm.begin(0);
        rtk_extend(st, acc_e, S, P, c->d_counters, c->cam, c->d_tiles, tile0, g, bound, c->max_blocks, c->count_work, c->launch_cache, ov);
        if (failed("extend", g)) break;
        tm.mark(0);
        rtk_compact(st, P, c->d_counters, g, bound, c->max_blocks, acc_e);
        if (failed("compact", g)) break;
        tm.mark(3);
        const uint64_t items = bound * (uint64_t)std::max(1, S.nl);
        if (!ov) {
            // a brute-force scene small enough to sit in LDS: the shade launch answers its own shadow queries
            // (not while the work counters are on: they count the shadow launch's tests)
            const bool shade_shadows = c->accel == RT_ACCEL_BRUTE && c->inline_shadows && !c->count_work && !S.profile;
            c->shade_shadowed = shade_shadows;
            if (!shade_shadows) {
                tm.begin(1);
                rtk_shadow(st, acc_s, S, P, c->d_counters, g, items, c->max_blocks, c->count_work, c->launch_cache);
                if (failed("shadow", g)) break;
                tm.mark(1);
            }
            rtk_shade(st, S, P, c->d_counters, g, bound, c->max_blocks, shade_shadows ? 3 : 0);
            if (failed("shade", g)) break;
            tm.mark(2);
        } else {
            hipEventRecord(c->ev_scan[g], st);
            if (g < S.max_depth) {                      // the last generation sends out no rays
                rtk_shade(st, S, P, c->d_counters, g, bound, c->max_blocks, 1);
                tm.mark(2);
            }
            hipStreamWaitEvent(B, c->ev_scan[g], 0);
            last_b = g;
            tmB.mark(-1);
            if (g >= 1) rtk_miss(B, S, P, c->d_counters, g, bound, c->max_blocks);
            tmB.mark(3);
            tmB.begin(1);
            rtk_shadow(B, acc_s, S, P, c->d_counters, g, items, c->max_blocks, c->count_work, c->launch_cache);
            tmB.mark(1);
            rtk_shade(B, S, P, c->d_counters, g, bound, c->max_blocks, 2);
            tmB.mark(2);
            hipEventRecord(c->ev_light[g], B);
            if (failed("spawn / miss / shadow / light (two streams)", g)) break;
        }
        bound = std::min<uint64_t>(bound * 2, cap);
    }
    if (rc != RT_OK) {
        // join: everything queued on B so far precedes whatever follows on A (the record of this generation may be
        // the very call that failed, so a fresh one is made)
        if (last_b >= 0) {
            hipEventRecord(c->ev_light[last_b], B);
            hipStreamWaitEvent(st, c->ev_light[last_b], 0);
            (void)hipGetLastError();
        }
        return rc;
    }
    if (any_ov) { hipStreamWaitEvent(st, c->ev_light[S.max_depth], 0); tm.mark(-1); }
    for (int g = S.max_depth - 1; g >= 0 && c->two_children; g--)
        rtk_resolve(st, S, c->P, c->d_counters, g, std::min<uint64_t>((uint64_t)nsamples << std::min(g, 20), cap), c->max_blocks);
    rtk_finish(st, c->cam, c->P, c->d_tiles, tile0, npixels, shard_mode, d_out, d_outf, c->max_blocks);
    if (c->count_work) rtk_reduce_stats(st, c->P, c->d_counters);
    if (failed("resolve / finish", -1)) return rc;
    tm.mark(3);
    return RT_OK;
}

// Everything the launches of a batch take by value, as bytes: two batches with the same bytes queue the same launches.
static void batch_key(const rt_ctx* c, uint32_t tile0, uint32_t ntiles, int shard_mode, const void* d_out, const void* d_outf,
                      int first_ov, std::vector<uint8_t>& key) {
    key.clear();
    auto put = [&key](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; key.insert(key.end(), b, b + n); };
    const void* ptrs[] = {d_out, d_outf, c->d_tiles, c->d_counters, c->hp2, c->hn2, c->hd2, c->hitlist2, c->stream, c->stream2};
    const int ints[] = {(int)tile0, (int)ntiles, shard_mode, first_ov, c->accel, c->max_blocks, c->inline_shadows ? 1 : 0,
                        c->count_work ? 1 : 0, c->two_children ? 1 : 0};
    put(ptrs, sizeof(ptrs)); put(ints, sizeof(ints));
    put(c->mode_extend, sizeof(c->mode_extend)); put(c->mode_shadow, sizeof(c->mode_shadow));
    put(&c->S, sizeof(DScene)); put(&c->P, sizeof(DPool)); put(&c->cam, sizeof(DCamera));
}

// Enqueues one batch (tiles [tile0, tile0+ntiles) of the current partition) and
// waits for it.  All launches are asynchronous; the only host sync is the final
// read of the counters (overflow flag + ray statistics).
static int run_batch(rt_ctx* c, uint32_t tile0, uint32_t ntiles, int shard_mode, uint8_t* d_out, float* d_outf,
                     BatchResult& res, StageTimes& tm) {
    const DScene& S = c->S;
    const uint32_t spp = (uint32_t)c->cam.spp;
    const uint32_t npixels = ntiles * RT_TILE_AREA;
    const uint64_t nsamples64 = (uint64_t)npixels * spp;
    hipStream_t st = c->stream;
    const uint64_t cap = c->P.capacity;
    const auto host_t0 = std::chrono::steady_clock::now();
    int first_ov = S.max_depth + 1;
    // (brute force launches are short: the extra launches of the split would cost more than the overlap gives)
    if (c->overlap_rays && c->accel == RT_ACCEL_LBVH && c->gen_rays_known && c->stream2 && c->alt_capacity >= cap && S.max_depth >= 1) {
        while (first_ov > 0 && c->gen_rays[first_ov - 1] <= c->overlap_rays) first_ov--;
        if (first_ov == S.max_depth && S.max_depth > 0) first_ov = S.max_depth + 1;     // a lone last generation has nothing to overlap with
    }
    const bool any_ov = first_ov <= S.max_depth;
    if (c->reported_first_ov != first_ov && getenv("RT_VERBOSE")) {
        if (any_ov) fprintf(stderr, "[rt overlap] generations %d..%d on two streams (expected rays there: %llu and fewer)\n", first_ov,
                            S.max_depth, (unsigned long long)c->gen_rays[first_ov]);
        else fprintf(stderr, "[rt overlap] none\n");
        c->reported_first_ov = first_ov;
    }
    StageTimes tmB;
    tmB.st = c->stream2; tmB.on = tm.on && any_ov; tmB.pool = &c->event_pool; tmB.only = tm.only;
    // Thin batches run as a graph from the second time the same launches come (see rt_ctx::graphs).  Not while stage
    // times are taken (events recorded under capture carry no time stamps) nor with the tuning aids on.
    // A batch split over two streams becomes a graph with two branches, whose second branch the runtime runs on a
    // stream of its own that is destroyed with the executable graph: only on runtimes where that is safe (see
    // runtime_keeps_queues_alive); elsewhere such batches are queued launch by launch.
    rt_ctx::BatchGraph* bg = nullptr;
    if (c->graphs_on && !tm.on && !S.profile && nsamples64 <= c->graph_max_samples && (!any_ov || runtime_keeps_queues_alive())) {
        std::vector<uint8_t> key;
        batch_key(c, tile0, ntiles, shard_mode, d_out, d_outf, first_ov, key);
        for (auto& g : c->graphs)
            if (g.key == key) { bg = &g; break; }
        if (!bg) {
            if (c->graphs.size() >= 16) drop_graphs(c);       // (nothing of this context is in flight between batches)
            c->graphs.emplace_back();
            bg = &c->graphs.back();
            bg->key.swap(key);
        }
        bg->seen++;
    }
    bool queued = false;
    if (bg && bg->exec) {
        if (first_ov > 0) c->shade_shadowed = c->accel == RT_ACCEL_BRUTE && c->inline_shadows && !c->count_work && !S.profile;
        HIP_TRY(hipGraphLaunch(bg->exec, st));
        c->graph_replays++;
        queued = true;
    } else if (bg && bg->seen >= 2 && !bg->failed) {
        // capture this batch's launches (both streams: the second joins through the events) instead of running them
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int rc = enqueue_batch(c, tile0, ntiles, shard_mode, d_out, d_outf, first_ov, tm, tmB);
            const hipError_t ee = hipStreamEndCapture(st, &graph);
            if (rc == RT_OK && ee == hipSuccess && graph && hipGraphInstantiate(&bg->exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                if (hipGraphLaunch(bg->exec, st) == hipSuccess) { c->graph_replays++; queued = true; }
                else { hipGraphExecDestroy(bg->exec); bg->exec = nullptr; }
            }
            if (graph) hipGraphDestroy(graph);
        }
        if (!queued) { (void)hipGetLastError(); bg->failed = true; bg->exec = nullptr; }    // queue it directly, now and from now on
    }
    if (!queued) {
        const int rc = enqueue_batch(c, tile0, ntiles, shard_mode, d_out, d_outf, first_ov, tm, tmB);
        if (rc != RT_OK) {
            // nothing of the batch is in flight when the caller sees the error (it may free the buffers next)
            const std::string why = rt_last_error();
            hipStreamSynchronize(st);
            if (c->stream2) hipStreamSynchronize(c->stream2);
            (void)hipGetLastError();
            rt_set_error(why);
            return rc;
        }
    }
    res.host_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    HIP_TRY(hipMemcpyAsync(c->h_counters, c->d_counters, sizeof(DCounters), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const DCounters hc = *c->h_counters;
    res.overflow = hc.overflow != 0;
    tmB.collect(res.stage_b);
    if (S.profile)
        fprintf(stderr, "[rt profile] batch tiles %u: closest-hit walks %llu steps %llu walk-cycles %llu (%.0f cyc/step, %.1f steps/walk) wave-cycles %llu | any-hit walks %llu steps %llu walk-cycles %llu (%.0f cyc/step)\n",
                ntiles, hc.prof_walks[0], hc.prof_steps[0], hc.prof_walk_cycles[0],
                hc.prof_steps[0] ? (double)hc.prof_walk_cycles[0] / hc.prof_steps[0] : 0.0,
                hc.prof_walks[0] ? (double)hc.prof_steps[0] / hc.prof_walks[0] : 0.0, hc.prof_wave_cycles[0],
                hc.prof_walks[1], hc.prof_steps[1], hc.prof_walk_cycles[1],
                hc.prof_steps[1] ? (double)hc.prof_walk_cycles[1] / hc.prof_steps[1] : 0.0);
    uint64_t used = 0;
    for (int g = 0; g <= S.max_depth + 1 && g <= RT_MAX_GEN; g++) used += hc.gen_count[g];
    if (used > cap) res.overflow = true;
    if (!res.overflow) {
        for (int g = 0; g <= S.max_depth; g++) {
            c->gen_rays_frame[g] = std::max<uint64_t>(c->gen_rays_frame[g], hc.gen_count[g]);
            if (g > 0) res.secondary += hc.gen_count[g];
            res.shadow += (uint64_t)hc.hit_count[g] * (uint64_t)S.nl;
            if (hc.gen_count[g]) res.generations = g + 1;
        }
        for (int k = 0; k < 2; k++) {
            res.nodes[k] = hc.nodes[k]; res.sph[k] = hc.sph_tests[k]; res.tri[k] = hc.tri_tests[k];
            res.lane_nodes[k] = hc.lane_nodes[k]; res.lane_prims[k] = hc.lane_prims[k];
        }
    }
    return RT_OK;
}

// Keeps hierarchy `which` of those built at load time (Karras if it does not exist) and frees the others.
static void adopt_tree(rt_ctx* c, int which) {
    if (c->tree_in_use >= 0) return;
    if (which < 0 || which >= RT_TREES || !c->t_nodes[which]) which = RT_TREE_KARRAS;
    for (int t = 0; t < RT_TREES; t++) {
        if (t == which) continue;
        hipFree(c->t_nodes[t]); hipFree(c->t_wide[t]);
        c->t_nodes[t] = c->t_wide[t] = nullptr;
    }
    c->tree_in_use = which;
    c->S.nodes = c->t_nodes[which];
    c->S.nodes4 = c->t_wide[which];
    c->S.depth = c->t_depth[which];
    c->S.root = 0;
}

// The choice when nothing is measured: the binned surface-area build where it exists, else PLOC on meshes of up to
// 2^20 primitives, Karras on huge uniform clouds; RT_TREE=karras|ploc|sah overrides.
static int default_tree(const rt_ctx* c) {
    int t = c->t_nodes[RT_TREE_SAH] ? RT_TREE_SAH : (c->S.n_prims <= (1 << 20) && c->t_nodes[RT_TREE_PLOC]) ? RT_TREE_PLOC : RT_TREE_KARRAS;
    if (const char* e = getenv("RT_TREE")) {
        for (int k = 0; k < RT_TREES; k++)
            if (!strcasecmp(e, kTreeName[k]) && c->t_nodes[k]) t = k;
    }
    return t;
}

// Picks the hierarchy (Karras or PLOC, when both were built) and, per ray generation, the
// fastest of the walks (2 per-lane, 3 wave packet, 4 per-lane on the 4-wide collapse) for the
// extend and the shadow kernel, by timing them on a
// sample of the frame: tiles worth up to 32 Mi samples (at most one batch), evenly spaced over
// the rank's share of the film, all their samples.  Every combination gives bit-identical results, so
// only speed is at stake; the sample's pixels are discarded.  The sample is a whole batch
// on purpose: a small one measures latency, not throughput, and picks the per-lane walk
// where the packet walk wins at scale.  Once per loaded scene.
static int calibrate(rt_ctx* c, uint32_t tiles_per_batch) {
    c->tuned = true;
    if (c->accel != RT_ACCEL_LBVH || c->forced_modes || !c->calibration || c->tiles.empty()) {
        adopt_tree(c, default_tree(c));
        return RT_OK;
    }
    DScene& S = c->S;
    const uint32_t spp = (uint32_t)c->cam.spp;
    const uint64_t per_tile = (uint64_t)RT_TILE_AREA * spp;
    const uint32_t total = (uint32_t)c->tiles.size();
    const int G = S.max_depth + 1;
    // candidates: every hierarchy that was built -- except Karras where the surface-area build exists (on scenes that
    // small it never won a timing, and every candidate costs a trial frame)
    // (RT_TREE=karras|ploc|sah: that hierarchy and no other; RT_CALIBRATE_ALL=1: all of them)
    int cand[RT_TREES], ntrees = 0;
    for (int t = 0; t < RT_TREES; t++)
        if (c->t_nodes[t] && !(t == RT_TREE_KARRAS && c->t_nodes[RT_TREE_SAH] && !getenv("RT_CALIBRATE_ALL"))) cand[ntrees++] = t;
    if (const char* e = getenv("RT_TREE"))
        for (int t = 0; t < RT_TREES; t++)
            if (c->t_nodes[t] && !strcasecmp(e, kTreeName[t])) { cand[0] = t; ntrees = 1; }
    const float4* tree_nodes[RT_TREES]; const float4* tree_wide[RT_TREES]; int tree_depth[RT_TREES];
    for (int k = 0; k < ntrees; k++) { tree_nodes[k] = c->t_nodes[cand[k]]; tree_wide[k] = c->t_wide[cand[k]]; tree_depth[k] = c->t_depth[cand[k]]; }
    std::vector<hipEvent_t> ev((size_t)G * 4, nullptr);
    uint32_t* d_pick = nullptr;
    auto cleanup = [&]() {
        for (auto& e : ev) if (e) hipEventDestroy(e);
        hipFree(d_pick);
        S.nodes = tree_nodes[0]; S.nodes4 = tree_wide[0]; S.depth = tree_depth[0]; S.root = 0;
    };
#define CAL_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rt_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); cleanup(); return (e_ == hipErrorOutOfMemory) ? RT_ERR_NOMEM : RT_ERR_HIP; } } while (0)
    for (auto& e : ev) CAL_TRY(hipEventCreate(&e));
    // [tree][walk mode][generation], milliseconds
    struct Times { float v[RT_TREES][6][RT_MAX_GEN + 1]; };
    std::vector<Times> times(2);
    auto& te = times[0].v;
    auto& tsh = times[1].v;
    uint32_t sample_rays[RT_MAX_GEN + 1] = {0};      // rays per generation of the sample (known once the first trial has run)
    bool team_timed[RT_MAX_GEN + 1] = {false};
    bool ok = false;
    // at most 32 Mi samples: enough to be in the throughput regime, and about one extra K3 frame in all
    uint32_t n = std::min(total, std::max(1u, std::min<uint32_t>(tiles_per_batch, (uint32_t)std::max<uint64_t>(1, (32ull << 20) / per_tile))));
    for (; !ok && n >= 1; n /= 2) {
        std::vector<uint32_t> pick(n);
        for (uint32_t k = 0; k < n; k++) pick[k] = c->tiles[(uint64_t)k * total / n];
        hipFree(d_pick); d_pick = nullptr;
        CAL_TRY(hipMalloc(&d_pick, n * sizeof(uint32_t)));
        CAL_TRY(hipMemcpyAsync(d_pick, pick.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        CAL_TRY(hipStreamSynchronize(c->stream));      // pick goes out of scope at the end of this round
        const uint32_t nsamples = (uint32_t)(n * per_tile);
        ok = true;
        for (int t = 0; t < ntrees && ok; t++) {
            S.nodes = tree_nodes[t]; S.nodes4 = tree_wide[t]; S.depth = tree_depth[t]; S.root = 0;
            for (int mode = 2; mode <= 5 && ok; mode++) {
                // the per-lane walk on the binary tree is timed only where no 4-wide collapse exists: wherever
                // both were timed the 4-wide one won, and the binary one is the most expensive trial of all
                if ((mode >= 4 && !tree_wide[t]) || (mode == 2 && tree_wide[t]) || (mode == 5 && !c->team_walks)) {
                    for (int g = 0; g < G; g++) te[t][mode][g] = tsh[t][mode][g] = 1e30f;
                    continue;
                }
                // (timing small samples three times and keeping the fastest was tried: the choice still flips between runs
                // on the thinnest frames -- the two walks are within a few per cent there -- and the first frame trebles)
                const int reps = 1;
                for (int g = 0; g < G; g++) te[t][mode][g] = tsh[t][mode][g] = 1e30f;
                if (mode == 5) {
                    bool any_thin = false;
                    for (int g = 0; g < G; g++) any_thin = any_thin || sample_rays[g] <= (1u << 20);
                    if (!any_thin) continue;
                }
                for (int rep = 0; rep < reps && ok; rep++) {
                    hipStream_t st = c->stream;
                    rtk_init_counters(st, c->d_counters, nsamples, c->P, G, false);
                    uint64_t bound = nsamples;
                    for (int g = 0; g < G; g++) {
                        // team walks are for generations of few rays (every lane-step of theirs costs a full wave
                        // instruction): a generation the packet trial counted above a million rays is not timed
                        // with them, the packet walk runs in its place
                        const int m = (mode == 5 && sample_rays[g] > (1u << 20)) ? 3 : mode;
                        team_timed[g] = m == 5;
                        hipEventRecord(ev[4 * g + 0], st);
                        rtk_extend(st, m, S, c->P, c->d_counters, c->cam, d_pick, 0, g, bound, c->max_blocks, false, c->launch_cache);
                        hipEventRecord(ev[4 * g + 1], st);
                        rtk_compact(st, c->P, c->d_counters, g, bound, c->max_blocks, m);
                        hipEventRecord(ev[4 * g + 2], st);
                        rtk_shadow(st, m, S, c->P, c->d_counters, g, bound * (uint64_t)std::max(1, S.nl), c->max_blocks, false, c->launch_cache);
                        hipEventRecord(ev[4 * g + 3], st);
                        rtk_shade(st, S, c->P, c->d_counters, g, bound, c->max_blocks);
                        CAL_TRY(hipGetLastError());        // a rejected launch of this generation, named by the expression
                        bound = std::min<uint64_t>(bound * 2, c->P.capacity);
                    }
                    if (hipMemcpyAsync(c->h_counters, c->d_counters, sizeof(DCounters), hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess) { ok = false; n = 0; break; }
                    const DCounters hc = *c->h_counters;
                    if (hc.overflow) { ok = false; break; }      // sample does not fit the pool: halve it
                    for (int g = 0; g < G; g++) {
                        float a = 1e30f, b = 1e30f;
                        sample_rays[g] = hc.gen_count[g];
                        hipEventElapsedTime(&a, ev[4 * g + 0], ev[4 * g + 1]);
                        hipEventElapsedTime(&b, ev[4 * g + 2], ev[4 * g + 3]);
                        if (mode == 5 && !team_timed[g]) continue;
                        te[t][mode][g] = std::min(te[t][mode][g], a);
                        tsh[t][mode][g] = std::min(tsh[t][mode][g], b);
                    }
                }
            }
        }
        if (ok) break;
    }
    cleanup();
#undef CAL_TRY
    if (ok) {
        float best_total[RT_TREES] = {0, 0, 0};
        for (int t = 0; t < ntrees; t++)
            for (int g = 0; g < G; g++)
                best_total[t] += std::min(std::min(te[t][2][g], te[t][5][g]), std::min(te[t][3][g], te[t][4][g])) +
                                 std::min(std::min(tsh[t][2][g], tsh[t][5][g]), std::min(tsh[t][3][g], tsh[t][4][g]));
        int t = 0;
        for (int k = 1; k < ntrees; k++) if (best_total[k] < best_total[t]) t = k;
        adopt_tree(c, cand[t]);
        // first guess of the rays per generation of a whole batch, until a frame has been counted
        for (int g = 0; g < G; g++)
            c->gen_rays[g] = (uint64_t)((double)sample_rays[g] * ((double)tiles_per_batch * (double)per_tile) / ((double)n * (double)per_tile));
        c->gen_rays_known = true;
        for (int g = 0; g < G; g++) {
            // within 10 % the packet walk is kept: thin generations time within noise of each other, and a
            // choice that flips from run to run would make the work counters (fetches per ray) flip with it
            auto best_of = [](float lane, float packet, float wide, float team) {
                if (packet <= 1.1f * lane && packet <= 1.1f * wide && packet <= 1.1f * team) return 3;
                if (team < lane && team < wide) return 5;
                return wide < lane ? 4 : 2;
            };
            c->mode_extend[g] = (int8_t)best_of(te[t][2][g], te[t][3][g], te[t][4][g], te[t][5][g]);
            c->mode_shadow[g] = (int8_t)best_of(tsh[t][2][g], tsh[t][3][g], tsh[t][4][g], tsh[t][5][g]);
        }
        if (getenv("RT_VERBOSE")) {
            fprintf(stderr, "[rt calibrate] %u tiles; traversal ms", n);
            for (int k = 0; k < ntrees; k++) fprintf(stderr, "%s %s %.3f", k ? "," : "", kTreeName[cand[k]], best_total[k]);
            const char* tag = "?-LPWT";
            fprintf(stderr, " -> %s; extend (lane/packet/wide/team ms):", kTreeName[cand[t]]);
            for (int g = 0; g < G; g++) fprintf(stderr, " g%d %.3f/%.3f/%.3f/%.3f->%c", g, te[t][2][g] > 1e29f ? 0.0f : te[t][2][g], te[t][3][g], te[t][4][g] > 1e29f ? 0.0f : te[t][4][g], te[t][5][g] > 1e29f ? 0.0f : te[t][5][g], tag[c->mode_extend[g]]);
            fprintf(stderr, "; rays:");
            for (int g = 0; g < G; g++) fprintf(stderr, " g%d %u", g, sample_rays[g]);
            fprintf(stderr, "; shadow:");
            for (int g = 0; g < G; g++) fprintf(stderr, " g%d %.3f/%.3f/%.3f/%.3f->%c", g, tsh[t][2][g] > 1e29f ? 0.0f : tsh[t][2][g], tsh[t][3][g], tsh[t][4][g] > 1e29f ? 0.0f : tsh[t][4][g], tsh[t][5][g] > 1e29f ? 0.0f : tsh[t][5][g], tag[c->mode_shadow[g]]);
            fprintf(stderr, "\n");
        }
    } else {
        adopt_tree(c, default_tree(c));
        if (getenv("RT_VERBOSE"))
            fprintf(stderr, "[rt calibrate] no sample fitted the ray pool: keeping the packet walk everywhere\n");
    }
    return RT_OK;
}

static int render_internal(rt_ctx* c, int rank, int world, int shard_mode, uint8_t* d_out, float* d_outf,
                           rt_stats* stats) {
    if (!c) return RT_ERR_INVALID;
    if (!c->has_scene) { rt_set_error("rt_load_scene has not succeeded on this context"); return RT_ERR_NO_SCENE; }
    HIP_TRY(hipSetDevice(c->device));
    int rc = set_partition(c, rank, world);
    if (rc != RT_OK) return rc;
    const uint32_t spp = (uint32_t)c->cam.spp;
    // Batch = as many samples as one pass of the pipeline takes.  Large on purpose: the late, thin
    // generations of a batch cannot run faster than their longest walks, so their launches have a floor
    // of ~0.1 ms each whatever they hold; one 128 Mi batch (about 75 GB of pool, a quarter of this GPU's
    // HBM) renders the K3 frame in 16 ms where four 32 Mi batches take 20.
    uint64_t want_samples = c->opt.batch_samples > 0 ? (uint64_t)c->opt.batch_samples : (128ull << 20);
    if (const char* e = getenv("RT_BATCH_SAMPLES")) {
        uint64_t v = strtoull(e, nullptr, 10);
        if (v > 0) want_samples = v;
    }
    want_samples = std::min<uint64_t>(want_samples, 1ull << 30);      // sample indices of a batch are 32-bit
    uint64_t per_tile = (uint64_t)RT_TILE_AREA * spp;
    if (per_tile > 0x7FFFFFFFull) { rt_set_error("sample_rate too large for one tile per batch"); return RT_ERR_UNSUPPORTED; }
    uint32_t total_tiles = (uint32_t)c->tiles.size();
    uint32_t tiles_per_batch = 1;
    for (;;) {
        tiles_per_batch = (uint32_t)std::max<uint64_t>(1, want_samples / per_tile);
        tiles_per_batch = std::min(tiles_per_batch, std::max(1u, total_tiles));
        if (c->tiles_cap) tiles_per_batch = std::min(tiles_per_batch, c->tiles_cap);
        uint64_t batch_samples = (uint64_t)tiles_per_batch * per_tile;
        uint64_t capacity = c->opt.pool_rays > 0 ? (uint64_t)c->opt.pool_rays : 4 * batch_samples;
        capacity = std::max(capacity, batch_samples);
        capacity = std::max(capacity, c->grown_capacity);   // what a single tile turned out to need
        if (c->opt.pool_rays <= 0) capacity = std::max(capacity, c->pool_capacity);   // keep what earlier frames grew
        capacity = std::min<uint64_t>(capacity, 0x7FFFFFF0ull);
        rc = ensure_pool(c, capacity, batch_samples);
        if (rc == RT_OK) break;
        // not enough device memory for this batch size: halve it (the image does not depend on it)
        if (rc != RT_ERR_NOMEM || want_samples <= (4ull << 20) || tiles_per_batch <= 1) return rc;
        (void)hipGetLastError();
        free_pool(c);
        want_samples /= 2;
    }
    if (!c->tuned && (rc = calibrate(c, tiles_per_batch)) != RT_OK) return rc;
    // the second copy of the per-generation buffers is worth having, not worth failing for
    bool want_overlap = c->overlap_rays > 0 && c->S.max_depth >= 1 && c->accel == RT_ACCEL_LBVH;
    auto try_overlap = [&]() {
        if (!want_overlap) return;
        if (ensure_overlap(c) != RT_OK) {
            (void)hipGetLastError();
            hipFree(c->hp2); hipFree(c->hn2); hipFree(c->hd2); hipFree(c->hitlist2);
            c->hp2 = c->hn2 = c->hd2 = nullptr; c->hitlist2 = nullptr; c->alt_capacity = 0;
            want_overlap = false;
        }
    };
    try_overlap();

    StageTimes tm;
    tm.st = c->stream;
    tm.pool = &c->event_pool;
    tm.on = stats != nullptr && c->stage_times != 0;
    tm.only = c->stage_times == 2 ? 0 : (c->stage_times == 3 ? 1 : -1);
    rt_stats acc{};
    double stage_b[4] = {0, 0, 0, 0};
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    // work list of (tile0, ntiles); an overflowing batch is split in two and retried
    std::vector<std::pair<uint32_t, uint32_t>> work;
    for (uint32_t t = total_tiles; t > 0;) {
        uint32_t n = std::min(tiles_per_batch, t);
        // push in reverse so the vector pops in ascending tile order
        work.push_back({t - n, n});
        t -= n;
    }
    // progress in film pixels: tiles at the right and bottom edge are cut by the film
    auto tile_pixels = [&](uint32_t t) {
        const int tx = (int)(t & 0xFFFF), ty = (int)(t >> 16);
        return (uint64_t)std::min(RT_TILE, c->cam.width - tx * RT_TILE) * (uint64_t)std::min(RT_TILE, c->cam.height - ty * RT_TILE);
    };
    uint64_t pixels_total = 0, pixels_done = 0;
    for (uint32_t t : c->tiles) pixels_total += tile_pixels(t);
    if (c->progress) c->progress(c->progress_user, 0, pixels_total);
    while (!work.empty()) {
        auto [t0, n] = work.back();
        work.pop_back();
        BatchResult br;
        rc = run_batch(c, t0, n, shard_mode, d_out, d_outf, br, tm);
        if (rc != RT_OK) return rc;
        if (br.overflow) {
            acc.retries++;
            // Prefer a bigger pool (it stays for the following batches and frames) to a
            // smaller batch; split only when the device cannot hold a pool twice as big.
            uint64_t bigger = (uint64_t)c->P.capacity * 2;      // twice what this batch ran with (the allocation may be larger)
            size_t free_b = 0, total_b = 0;
            // (an explicit rt_options.pool_rays is a hard cap: then batches are split)
            bool can_grow = c->opt.pool_rays <= 0 && bigger < 0xFFFFFFF0ull && hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
                            (double)bigger * (140.0 + std::max(1, c->S.nl)) < 0.5 * ((double)free_b + (double)c->pool_capacity * 140.0);
            if (can_grow || n == 1) {
                if (bigger > 0x7FFFFFF0ull) { rt_set_error("one tile needs a ray pool beyond 2^31 records"); return RT_ERR_UNSUPPORTED; }
                rc = ensure_pool(c, bigger, c->scol_capacity);
                if (rc != RT_OK) return rc;
                // under an explicit pool_rays the next frame would start from the small pool again and overflow on
                // the same tile: remember what it took
                if (c->opt.pool_rays > 0) c->grown_capacity = std::max(c->grown_capacity, bigger);
                try_overlap();
                work.push_back({t0, n});
            } else {
                uint32_t h = n / 2;
                work.push_back({t0 + h, n - h});
                work.push_back({t0, h});
                c->tiles_cap = c->tiles_cap ? std::min(c->tiles_cap, n - h) : n - h;   // later batches and frames start there
            }
            continue;
        }
        acc.batches++;
        acc.ms_host_enqueue += br.host_enqueue_ms;
        if (c->progress) {
            for (uint32_t t = t0; t < t0 + n; t++) pixels_done += tile_pixels(c->tiles[t]);
            c->progress(c->progress_user, pixels_done, pixels_total);
        }
        for (int k = 0; k < 4; k++) stage_b[k] += br.stage_b[k];
        acc.rays_secondary += br.secondary;
        acc.rays_shadow += br.shadow;
        for (int k = 0; k < 2; k++) {
            acc.nodes_visited[k] += br.nodes[k]; acc.sphere_tests[k] += br.sph[k]; acc.tri_tests[k] += br.tri[k];
            acc.lane_node_tests[k] += br.lane_nodes[k]; acc.lane_prim_tests[k] += br.lane_prims[k];
        }
        acc.launches_extend += c->S.max_depth + 1;
        acc.launches_shadow += (c->S.nl && !c->shade_shadowed) ? c->S.max_depth + 1 : 0;
        acc.generations = std::max(acc.generations, br.generations);
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    for (int g = 0; g <= RT_MAX_GEN; g++) { c->gen_rays[g] = c->gen_rays_frame[g]; c->gen_rays_frame[g] = 0; }
    c->gen_rays_known = true;
    if (stats) {
        float ms = 0;
        hipEventElapsedTime(&ms, c->ev0, c->ev1);
        // primary rays: one per sample of every film pixel this rank owns
        uint64_t pix = 0;
        for (uint32_t t : c->tiles) {
            int tx = (int)(t & 0xFFFF), ty = (int)(t >> 16);
            int w = std::min(RT_TILE, c->cam.width - tx * RT_TILE), h = std::min(RT_TILE, c->cam.height - ty * RT_TILE);
            pix += (uint64_t)w * h;
        }
        acc.rays_primary = pix * spp;
        acc.ms_render = ms;
        double t4[4] = {0, 0, 0, 0};
        tm.collect(t4);
        for (int k = 0; k < 4; k++) t4[k] += stage_b[k];
        acc.ms_extend = t4[0]; acc.ms_shadow = t4[1]; acc.ms_shade = t4[2]; acc.ms_other = t4[3];
        acc.accel_used = c->accel;
        *stats = acc;
    }
    return RT_OK;
}

extern "C" int rt_render_device(rt_ctx* c, void* d_rgb, rt_stats* stats) {
    if (!d_rgb) return RT_ERR_INVALID;
    return render_internal(c, 0, 1, 0, (uint8_t*)d_rgb, nullptr, stats);
}

// Device-to-host copy of a finished frame into the caller's (pageable) buffer, queued on the context's stream and
// complete on return.  A bandwidth choice: a plain copy into pageable memory is staged by the runtime at about 8 GB/s
// (3 ms for the 25 MB of a 3840x2160 frame).  The frame goes to pinned staging memory of the context instead -- 2 MiB
// chunks at the link's rate, an event after each -- and the calling thread plus up to three helpers copy the chunks
// on as their events complete (one chunk: no events, no helpers).
static int copy_out(rt_ctx* c, void* dst, const void* src, size_t bytes) {
    const size_t CH = (size_t)2 << 20;
    if (bytes == 0) return RT_OK;
    bool staged = c->fast_copy;
    if (staged && c->h_stage_cap < bytes) {
        if (c->h_stage) hipHostFree(c->h_stage);
        c->h_stage = nullptr; c->h_stage_cap = 0;
        const size_t want = std::max(bytes, (size_t)1 << 20);
        if (hipHostMalloc((void**)&c->h_stage, want, hipHostMallocDefault) == hipSuccess) c->h_stage_cap = want;
        else { (void)hipGetLastError(); c->h_stage = nullptr; staged = false; }
    }
    const size_t nchunks = (bytes + CH - 1) / CH;
    while (staged && nchunks > 1 && c->copy_events.size() < nchunks) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); staged = false; break; }
        c->copy_events.push_back(e);
    }
    if (!staged) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return RT_OK;
    }
    if (nchunks == 1) {
        HIP_TRY(hipMemcpyAsync(c->h_stage, src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::memcpy(dst, c->h_stage, bytes);
        return RT_OK;
    }
    for (size_t i = 0; i < nchunks; i++) {
        const size_t off = i * CH, len = std::min(CH, bytes - off);
        HIP_TRY(hipMemcpyAsync(c->h_stage + off, (const uint8_t*)src + off, len, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->copy_events[i], c->stream));
    }
    const int nthreads = (int)std::min<size_t>(4, nchunks);
    std::atomic<int> failed{0};
    auto work = [&](int k) {
        if (hipSetDevice(c->device) != hipSuccess) { failed = 1; return; }
        for (size_t i = (size_t)k; i < nchunks; i += (size_t)nthreads) {
            if (hipEventSynchronize(c->copy_events[i]) != hipSuccess) { failed = 1; return; }
            const size_t off = i * CH, len = std::min(CH, bytes - off);
            std::memcpy((uint8_t*)dst + off, c->h_stage + off, len);
        }
    };
    std::vector<std::thread> helpers;
    for (int k = 1; k < nthreads; k++) helpers.emplace_back(work, k);
    work(0);
    for (auto& t : helpers) t.join();
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (failed) { rt_set_error("device-to-host copy of the frame failed"); return RT_ERR_HIP; }
    return RT_OK;
}

extern "C" int rt_render(rt_ctx* c, uint8_t* rgb_out, float* rgbf_out, rt_stats* stats) {
    if (!c || (!rgb_out && !rgbf_out)) return RT_ERR_INVALID;
    if (!c->has_scene) { rt_set_error("rt_load_scene has not succeeded on this context"); return RT_ERR_NO_SCENE; }
    HIP_TRY(hipSetDevice(c->device));
    size_t px = (size_t)c->cam.width * c->cam.height;
    if (c->rgb_pixels < px) {
        hipFree(c->d_rgb); c->d_rgb = nullptr;
        HIP_TRY(hipMalloc(&c->d_rgb, px * 3));
        c->rgb_pixels = px;
    }
    if (rgbf_out && c->rgbf_pixels < px) {
        hipFree(c->d_rgbf); c->d_rgbf = nullptr;
        HIP_TRY(hipMalloc(&c->d_rgbf, px * 3 * sizeof(float)));
        c->rgbf_pixels = px;
    }
    int rc = render_internal(c, 0, 1, 0, c->d_rgb, rgbf_out ? c->d_rgbf : nullptr, stats);
    if (rc != RT_OK) return rc;
    if (rgb_out) { rc = copy_out(c, rgb_out, c->d_rgb, px * 3); if (rc != RT_OK) return rc; }
    if (rgbf_out) { rc = copy_out(c, rgbf_out, c->d_rgbf, px * 3 * sizeof(float)); if (rc != RT_OK) return rc; }
    return RT_OK;
}

extern "C" int rt_render_shard(rt_ctx* c, int rank, int world, void* d_shard, rt_stats* stats) {
    if (!c || !d_shard || world <= 0 || rank < 0 || rank >= world) return RT_ERR_INVALID;
    return render_internal(c, rank, world, 1, (uint8_t*)d_shard, nullptr, stats);
}

extern "C" int rt_assemble_shards(rt_ctx* c, const void* d_all, int world, void* d_rgb) {
    if (!c || !d_all || !d_rgb || world <= 0) return RT_ERR_INVALID;
    if (!c->has_scene) return RT_ERR_NO_SCENE;
    HIP_TRY(hipSetDevice(c->device));
    int w = c->cam.width, h = c->cam.height;
    int ntx = (w + RT_TILE - 1) / RT_TILE, nty = (h + RT_TILE - 1) / RT_TILE;
    if (c->rowbase_world != world || c->rowbase_w != w || c->rowbase_h != h) {
        std::vector<uint32_t> rb((size_t)world * nty);
        for (int r = 0; r < world; r++) {
            uint32_t acc = 0;
            for (int ty = 0; ty < nty; ty++) {
                rb[(size_t)r * nty + ty] = acc;
                int first = ((r - 3 * ty) % world + world) % world;
                if (first < ntx) acc += (uint32_t)((ntx - first + world - 1) / world);
            }
        }
        hipFree(c->d_rowbase); c->d_rowbase = nullptr;
        HIP_TRY(hipMalloc(&c->d_rowbase, rb.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(c->d_rowbase, rb.data(), rb.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->rowbase_world = world; c->rowbase_w = w; c->rowbase_h = h;
    }
    rtk_assemble(c->stream, (const uint8_t*)d_all, rt_shard_bytes(w, h, world), world, w, h, ntx, nty, c->d_rowbase, (uint8_t*)d_rgb);
    HIP_TRY(hipGetLastError());
    // on the context's own stream the call is complete on return; on a stream the caller gave (rt_options.stream) the
    // kernel is queued there and ordering is the caller's, like everything else the caller puts on that stream
    if (c->own_stream) HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

// The same de-interleave without any allocation: the table is rebuilt in this context's slice of the GPU's scratch
// array by a kernel (every entry the assemble kernel reads is rewritten, so nothing of an earlier call survives),
// then the assemble kernel reads it from there.  No hipMalloc / hipFree / host-to-device copy.
extern "C" int rt_assemble_shards_noalloc(rt_ctx* c, const void* d_all, int world, void* d_rgb) {
    if (!c || !d_all || !d_rgb || world <= 0) return RT_ERR_INVALID;
    if (!c->has_scene) return RT_ERR_NO_SCENE;
    const int w = c->cam.width, h = c->cam.height;
    const int ntx = (w + RT_TILE - 1) / RT_TILE, nty = (h + RT_TILE - 1) / RT_TILE;
    if ((uint64_t)world * (uint64_t)nty > (uint64_t)RT_NOALLOC_MAX_ROWS) {
        char b[160];
        std::snprintf(b, sizeof(b), "rt_assemble_shards_noalloc: %d ranks x %d tile rows exceed the %d entries of the fixed table",
                      world, nty, RT_NOALLOC_MAX_ROWS);
        rt_set_error(b);
        return RT_ERR_UNSUPPORTED;
    }
    if (!c->scratch) { rt_set_error("scratch array was not resolved on this device"); return RT_ERR_HIP; }
    if (c->scratch_slot < 0) {
        rt_set_error("rt_assemble_shards_noalloc: all slices of this GPU's fixed table were held by other contexts when this one was created");
        return RT_ERR_UNSUPPORTED;
    }
    HIP_TRY(hipSetDevice(c->device));
    rtk_rowbase(c->stream, world, ntx, nty, c->scratch);
    rtk_assemble(c->stream, (const uint8_t*)d_all, rt_shard_bytes(w, h, world), world, w, h, ntx, nty, c->scratch, (uint8_t*)d_rgb);
    HIP_TRY(hipGetLastError());
    // a stream under graph capture must not be waited for; the captured work runs when the graph does
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    if (cs == hipStreamCaptureStatusNone && c->own_stream) HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

// ---------------------------------------------------------------- one process, several GPUs
// Every context renders its shard on a host thread of its own (a context's calls are serialised by its owner;
// the threads share nothing but the read-only arguments), then the shards travel to the first context's device.
extern "C" int rt_render_multi(rt_ctx* const* ctxs, int n, uint8_t* rgb_out, rt_stats* stats) {
    if (!ctxs || n <= 0 || !rgb_out) return RT_ERR_INVALID;
    for (int r = 0; r < n; r++) {
        if (!ctxs[r]) return RT_ERR_INVALID;
        if (!ctxs[r]->has_scene) { rt_set_error("rt_load_scene has not succeeded on every context"); return RT_ERR_NO_SCENE; }
        for (int q = 0; q < r; q++)
            if (ctxs[q] == ctxs[r]) { rt_set_error("the same context given twice"); return RT_ERR_INVALID; }
        if (ctxs[r]->cam.width != ctxs[0]->cam.width || ctxs[r]->cam.height != ctxs[0]->cam.height ||
            ctxs[r]->cam.spp != ctxs[0]->cam.spp || ctxs[r]->S.n_prims != ctxs[0]->S.n_prims) {
            rt_set_error("the contexts hold different scenes");
            return RT_ERR_INVALID;
        }
    }
    rt_ctx* c0 = ctxs[0];
    const int w = c0->cam.width, h = c0->cam.height;
    const size_t sb = rt_shard_bytes(w, h, n), px = (size_t)w * h;
    for (int r = 0; r < n; r++) {
        rt_ctx* c = ctxs[r];
        HIP_TRY(hipSetDevice(c->device));
        if (c->shard_cap < sb) {
            hipFree(c->d_shard); c->d_shard = nullptr; c->shard_cap = 0;
            HIP_TRY(hipMalloc(&c->d_shard, sb));
            c->shard_cap = sb;
        }
    }
    HIP_TRY(hipSetDevice(c0->device));
    if (c0->all_cap < sb * (size_t)n) {
        hipFree(c0->d_all); c0->d_all = nullptr; c0->all_cap = 0;
        HIP_TRY(hipMalloc(&c0->d_all, sb * (size_t)n));
        c0->all_cap = sb * (size_t)n;
    }
    if (c0->rgb_pixels < px) {
        hipFree(c0->d_rgb); c0->d_rgb = nullptr; c0->rgb_pixels = 0;
        HIP_TRY(hipMalloc(&c0->d_rgb, px * 3));
        c0->rgb_pixels = px;
    }
    std::vector<rt_stats> st((size_t)n);
    std::vector<int> rcs((size_t)n, RT_OK);
    std::vector<std::string> errs((size_t)n);
    {
        std::vector<std::thread> threads;
        for (int r = 1; r < n; r++)
            threads.emplace_back([&, r]() {
                rcs[r] = rt_render_shard(ctxs[r], r, n, ctxs[r]->d_shard, &st[r]);
                if (rcs[r] != RT_OK) errs[r] = rt_last_error();      // the message is per thread: carry it over
            });
        rcs[0] = rt_render_shard(c0, 0, n, c0->d_shard, &st[0]);
        if (rcs[0] != RT_OK) errs[0] = rt_last_error();
        for (auto& t : threads) t.join();
    }
    for (int r = 0; r < n; r++)
        if (rcs[r] != RT_OK) {
            rt_set_error("shard " + std::to_string(r) + " of " + std::to_string(n) + ": " + errs[r]);
            return rcs[r];
        }
    // the one exchange step of the frame: every shard to the first device, each over its own link
    HIP_TRY(hipSetDevice(c0->device));
    for (int r = 0; r < n; r++) {
        if (ctxs[r]->device == c0->device)
            HIP_TRY(hipMemcpyAsync(c0->d_all + (size_t)r * sb, ctxs[r]->d_shard, sb, hipMemcpyDeviceToDevice, c0->stream));
        else
            HIP_TRY(hipMemcpyPeerAsync(c0->d_all + (size_t)r * sb, c0->device, ctxs[r]->d_shard, ctxs[r]->device, sb, c0->stream));
    }
    int rc = rt_assemble_shards(c0, c0->d_all, n, c0->d_rgb);
    if (rc != RT_OK) return rc;
    rc = copy_out(c0, rgb_out, c0->d_rgb, px * 3);
    if (rc != RT_OK) return rc;
    if (stats) {
        rt_stats sum = st[0];
        for (int r = 1; r < n; r++) {
            sum.rays_primary += st[r].rays_primary; sum.rays_secondary += st[r].rays_secondary; sum.rays_shadow += st[r].rays_shadow;
            for (int k = 0; k < 2; k++) {
                sum.nodes_visited[k] += st[r].nodes_visited[k]; sum.sphere_tests[k] += st[r].sphere_tests[k];
                sum.tri_tests[k] += st[r].tri_tests[k]; sum.lane_node_tests[k] += st[r].lane_node_tests[k];
                sum.lane_prim_tests[k] += st[r].lane_prim_tests[k];
            }
            sum.ms_render = std::max(sum.ms_render, st[r].ms_render);
            sum.ms_extend = std::max(sum.ms_extend, st[r].ms_extend); sum.ms_shadow = std::max(sum.ms_shadow, st[r].ms_shadow);
            sum.ms_shade = std::max(sum.ms_shade, st[r].ms_shade); sum.ms_other = std::max(sum.ms_other, st[r].ms_other);
            sum.launches_extend += st[r].launches_extend; sum.launches_shadow += st[r].launches_shadow;
            sum.batches += st[r].batches; sum.retries += st[r].retries;
            sum.generations = std::max(sum.generations, st[r].generations);
            sum.ms_host_enqueue = std::max(sum.ms_host_enqueue, st[r].ms_host_enqueue);
        }
        *stats = sum;
    }
    return RT_OK;
}

// ---------------------------------------------------------------- queries
extern "C" int rt_query_rays(rt_ctx* c, int64_t n, const float* o, const float* d, const float* dmax,
                             int any_hit, int accel, int32_t* prim_out, float* dist2_out) {
    if (!c || n < 0 || !o || !d || !prim_out) return RT_ERR_INVALID;
    if (!c->has_scene) return RT_ERR_NO_SCENE;
    // a hierarchy other than the one in use (or, before the first render, than Karras): held until the first render
    const int other = (accel & RT_ACCEL_ALT_TREE) ? RT_TREE_PLOC : (accel & RT_ACCEL_SAH_TREE) ? RT_TREE_SAH : -1;
    const bool alt = other >= 0;
    accel &= ~(RT_ACCEL_ALT_TREE | RT_ACCEL_SAH_TREE);
    if (accel != RT_ACCEL_BRUTE && accel != RT_ACCEL_LBVH && accel != RT_ACCEL_LBVH_PACKET && accel != RT_ACCEL_LBVH_WIDE &&
        accel != RT_ACCEL_LBVH_TEAM) return RT_ERR_INVALID;
    if (alt && (accel == RT_ACCEL_BRUTE || !c->t_nodes[other] || (c->tree_in_use >= 0 && c->tree_in_use != other))) {
        rt_set_error("that hierarchy is not held (it was not built, or a render has already chosen another)");
        return RT_ERR_UNSUPPORTED;
    }
    if (accel != RT_ACCEL_BRUTE && !c->S.nodes) { rt_set_error("scene has fewer than 2 primitives: no LBVH"); return RT_ERR_UNSUPPORTED; }
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(c->device));
    // A hierarchy is conservative for rays that start where a render's rays can start (the box of the camera's
    // origins and all primitives): the exact tests accept more the farther away a ray starts.  Queries from
    // outside that box are answered by brute force, which is the definition of the result.
    std::vector<int64_t> outside;
    if (accel != RT_ACCEL_BRUTE)
        for (int64_t i = 0; i < n; i++) {
            bool in = true;
            for (int k = 0; k < 3; k++) in = in && o[3 * i + k] >= c->region[k] && o[3 * i + k] <= c->region[3 + k];
            if (!in) outside.push_back(i);
        }
    float *d_o = nullptr, *d_d = nullptr, *d_m = nullptr, *d_d2 = nullptr; int* d_p = nullptr;
    int rc = RT_OK;
    auto cleanup = [&]() { hipFree(d_o); hipFree(d_d); hipFree(d_m); hipFree(d_d2); hipFree(d_p); };
#define Q_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rt_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); cleanup(); return RT_ERR_HIP; } } while (0)
    Q_TRY(hipMalloc(&d_o, n * 3 * sizeof(float)));
    Q_TRY(hipMalloc(&d_d, n * 3 * sizeof(float)));
    Q_TRY(hipMalloc(&d_p, n * sizeof(int)));
    Q_TRY(hipMalloc(&d_d2, n * sizeof(float)));
    if (dmax) Q_TRY(hipMalloc(&d_m, n * sizeof(float)));
    Q_TRY(hipMemcpyAsync(d_o, o, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    Q_TRY(hipMemcpyAsync(d_d, d, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (dmax) Q_TRY(hipMemcpyAsync(d_m, dmax, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    DScene Sq = c->S;
    if (alt) { Sq.nodes = c->t_nodes[other]; Sq.nodes4 = c->t_wide[other]; Sq.depth = c->t_depth[other]; Sq.root = 0; }
    if ((accel == RT_ACCEL_LBVH_WIDE || accel == RT_ACCEL_LBVH_TEAM) && !Sq.nodes4) {
        cleanup();
        rt_set_error("no 4-wide collapse of this hierarchy was built (tree deeper than 64)");
        return RT_ERR_UNSUPPORTED;
    }
    rtk_query(c->stream, accel, any_hit, Sq, n, d_o, d_d, d_m, d_p, d_d2);
    Q_TRY(hipGetLastError());
    Q_TRY(hipMemcpyAsync(prim_out, d_p, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (dist2_out) Q_TRY(hipMemcpyAsync(dist2_out, d_d2, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    Q_TRY(hipStreamSynchronize(c->stream));
#undef Q_TRY
    cleanup();
    if (!outside.empty()) {
        const int64_t m = (int64_t)outside.size();
        std::vector<float> o2(3 * m), d2(3 * m), m2(dmax ? m : 0), r2(m);
        std::vector<int32_t> p2(m);
        for (int64_t j = 0; j < m; j++) {
            const int64_t i = outside[j];
            for (int k = 0; k < 3; k++) { o2[3 * j + k] = o[3 * i + k]; d2[3 * j + k] = d[3 * i + k]; }
            if (dmax) m2[j] = dmax[i];
        }
        rc = rt_query_rays(c, m, o2.data(), d2.data(), dmax ? m2.data() : nullptr, any_hit, RT_ACCEL_BRUTE, p2.data(), r2.data());
        if (rc != RT_OK) return rc;
        for (int64_t j = 0; j < m; j++) {
            prim_out[outside[j]] = p2[j];
            if (dist2_out) dist2_out[outside[j]] = r2[j];
        }
    }
    return rc;
}

extern "C" int rt_set_progress_callback(rt_ctx* c, rt_progress_fn fn, void* user) {
    if (!c) return RT_ERR_INVALID;
    c->progress = fn;
    c->progress_user = user;
    return RT_OK;
}

extern "C" int rt_set_stage_times(rt_ctx* c, int on) {
    if (!c) return RT_ERR_INVALID;
    c->stage_times = (on == 2 || on == 3) ? on : (on != 0 ? 1 : 0);
    return RT_OK;
}

extern "C" int rt_set_graphs(rt_ctx* c, int on, uint64_t* replays) {
    if (!c) return RT_ERR_INVALID;
    if (on >= 0) {
        c->graphs_on = on != 0;
        if (!c->graphs_on) {
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipStreamSynchronize(c->stream));
            drop_graphs(c);
        }
    }
    if (replays) *replays = c->graph_replays;
    return RT_OK;
}

extern "C" int rt_get_tree_stats(rt_ctx* c, rt_tree_stats* out) {
    if (!c || !out) return RT_ERR_INVALID;
    if (!c->has_scene) return RT_ERR_NO_SCENE;
    *out = rt_tree_stats{};
    out->primitives = c->S.n_prims;
    out->in_use = c->tree_in_use;
    for (int t = 0; t < RT_TREES; t++) {
        out->depth[t] = c->t_nodes[t] ? c->t_depth[t] : 0;
        out->node_visits[t] = c->t_cost[t][0];
        out->prim_tests[t] = c->t_cost[t][1];
    }
    return RT_OK;
}

extern "C" int rt_set_calibration(rt_ctx* c, int on) {
    if (!c) return RT_ERR_INVALID;
    c->calibration = on != 0;
    return RT_OK;
}

extern "C" int rt_set_work_counters(rt_ctx* c, int on) {
    if (!c) return RT_ERR_INVALID;
    c->count_work = on != 0;
    return RT_OK;
}

extern "C" int rt_primary_rays(rt_ctx* c, int64_t n, const int32_t* x, const int32_t* y, const int32_t* j,
                               float* o_out, float* d_out) {
    if (!c || n < 0 || !x || !y || !j || !o_out || !d_out) return RT_ERR_INVALID;
    if (!c->has_scene) return RT_ERR_NO_SCENE;
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(c->device));
    int *dx = nullptr, *dy = nullptr, *dj = nullptr; float *po = nullptr, *pd = nullptr;
    auto cleanup = [&]() { hipFree(dx); hipFree(dy); hipFree(dj); hipFree(po); hipFree(pd); };
#define Q_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rt_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); cleanup(); return RT_ERR_HIP; } } while (0)
    Q_TRY(hipMalloc(&dx, n * sizeof(int))); Q_TRY(hipMalloc(&dy, n * sizeof(int))); Q_TRY(hipMalloc(&dj, n * sizeof(int)));
    Q_TRY(hipMalloc(&po, n * 3 * sizeof(float))); Q_TRY(hipMalloc(&pd, n * 3 * sizeof(float)));
    Q_TRY(hipMemcpyAsync(dx, x, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    Q_TRY(hipMemcpyAsync(dy, y, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    Q_TRY(hipMemcpyAsync(dj, j, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    rtk_primary_rays(c->stream, c->cam, n, dx, dy, dj, po, pd);
    Q_TRY(hipGetLastError());
    Q_TRY(hipMemcpyAsync(o_out, po, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    Q_TRY(hipMemcpyAsync(d_out, pd, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    Q_TRY(hipStreamSynchronize(c->stream));
#undef Q_TRY
    cleanup();
    return RT_OK;
}

extern "C" uint32_t rt_jitter_hash(uint32_t seed, uint64_t counter) { return rt_jitter(seed, counter); }
