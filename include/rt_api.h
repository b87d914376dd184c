/* rt_api.h -- C-ABI of the MI355X-native trace loop.
 *
 * Drop-in replacement for ONE path of LastingLuck/Raytracer: the per-pixel trace
 * loop behind `Image* RayTrace::rayTrace(Scene& scn)` (reference
 * src/raytrace.h:69, src/raytrace.cpp:44-129, called from src/main.cpp:56-58).
 * The reference has no plugin/FFI layer; this header is the seam a maintainer
 * would bind in its place: plain pointers and sizes, no C++ types, no
 * exceptions, no exit() inside the library.
 *
 * All arithmetic that decides a pixel follows the reference's brute-force
 * branch operation for operation (float/double placement, evaluation order, no
 * fused multiply-add), so images are reproduced bit-for-bit except where the
 * device libm (pow, acos) rounds a double differently from glibc.
 *
 * Threading: calls on one rt_ctx must be serialised by the caller; different
 * contexts (one per GPU / per process) are independent.
 */
#ifndef RT_API_H_
#define RT_API_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status */
enum {
    RT_OK = 0,
    RT_ERR_INVALID = 1,      /* bad argument / inconsistent descriptor        */
    RT_ERR_IO = 2,           /* file could not be opened / written            */
    RT_ERR_PARSE = 3,        /* scene file rejected (bad vertex/normal index) */
    RT_ERR_NO_DEVICE = 4,    /* no usable HIP device: there is NO CPU fallback */
    RT_ERR_HIP = 5,          /* a HIP runtime call failed (see rt_last_error) */
    RT_ERR_NOMEM = 6,
    RT_ERR_UNSUPPORTED = 7,  /* e.g. max_depth beyond the supported limit     */
    RT_ERR_NO_SCENE = 8
};

/* Light types: numeric values of the reference's enum LightType (src/parse.h:18-23). */
enum { RT_LIGHT_AMBIENT = 0, RT_LIGHT_DIRECTIONAL = 1, RT_LIGHT_POINT = 2, RT_LIGHT_SPOT = 3 };
/* Projection: numeric values of the reference's enum ProjType (src/parse.h:13-16). */
enum { RT_PROJ_ORTHO = 0, RT_PROJ_PERSP = 1 };
/* Acceleration structure selection. */
enum { RT_ACCEL_AUTO = 0, RT_ACCEL_BRUTE = 1, RT_ACCEL_LBVH = 2,
       RT_ACCEL_LBVH_PACKET = 3 /* rt_query_rays only: force the wave-packet walk of the LBVH */,
       RT_ACCEL_LBVH_WIDE = 4   /* rt_query_rays only: force the per-lane walk of the 4-wide collapse */,
       RT_ACCEL_LBVH_TEAM = 5   /* rt_query_rays only: force the walk of the 4-wide collapse by a team of lanes per ray */,
       RT_ACCEL_ALT_TREE = 16   /* rt_query_rays only, OR-ed to an LBVH walk: use the second
                                   hierarchy (PLOC) that rt_load_scene built beside the Karras
                                   one; available until the first render has chosen between them */,
       RT_ACCEL_SAH_TREE = 32   /* the same for the third hierarchy (binned surface-area build), where
                                   rt_load_scene built it */ };

/* One light: the fields of the reference's class Light (src/parse.h:188-195). */
typedef struct rt_light {
    int32_t type;
    float color[3];
    float pos[3];
    float dir[3];
    float angle1, angle2; /* spot angles, stored raw as the parser does (src/parse.cpp:160-163) */
} rt_light;

/* Everything RayTrace::rayTrace reads from `Scene&` through its by-value getters
 * (src/parse.h:311-327), flattened.  Borrowed for the duration of the call only.
 * materials[i] = amb3 dif3 spec3 ns trans3 ior, the `material` directive's
 * order (src/parse.cpp:139-149).  tri_normals holds three normals per triangle:
 * for a flat `triangle` all three are norm(cross(v1-v0, v2-v0))
 * (src/parse.cpp:462-469), for `normal_triangle` the file's normals
 * (src/parse.cpp:482-491) and tri_flags[i] = 1. */
typedef struct rt_scene_desc {
    int32_t width, height;  /* film size (Scene::img)                   */
    int32_t spp;            /* sample_rate; > 1 enables jitter          */
    int32_t max_depth;      /* Scene::maxDepth                          */
    int32_t proj;           /* RT_PROJ_*                                */
    uint32_t seed;          /* seed of the shared jitter sequence       */
    float cam_pos[3], cam_dir[3], cam_up[3]; /* as given, NOT normalised */
    float cam_half_angle;   /* radians                                  */
    float background[3];
    float ambient[3];       /* ambient light colour                     */
    int32_t n_spheres;
    int32_t n_triangles;
    int32_t n_materials;
    int32_t n_lights;
    const float* spheres;      /* [n_spheres][4]  cx cy cz r */
    const int32_t* sphere_mat; /* [n_spheres]                */
    const float* tri_verts;    /* [n_triangles][9]           */
    const float* tri_normals;  /* [n_triangles][9]           */
    const int32_t* tri_flags;  /* [n_triangles]              */
    const int32_t* tri_mat;    /* [n_triangles]              */
    const float* materials;    /* [n_materials][14]          */
    const rt_light* lights;    /* [n_lights]                 */
    /* `bvh_threshold n` / `bvh_depth n` of the scene file (src/parse.cpp:252-261), 0 = not given.
     * The reference meant the threshold as its brute-force <-> BVH switch (`getNumObjects() >
     * getBVHThreshold()`, the commented-out block at src/raytrace.cpp:61-69); with RT_ACCEL_AUTO
     * and rt_options.lbvh_threshold == 0 it plays that role here.  bvh_depth is carried but
     * unused: the LBVH has no depth cap. */
    int32_t bvh_threshold;
    int32_t bvh_depth;
} rt_scene_desc;

typedef struct rt_options {
    int32_t device;            /* HIP device ordinal                                    */
    int32_t accel;             /* RT_ACCEL_*; AUTO = brute force (LDS-staged) for small
                                  scenes, LBVH above lbvh_threshold primitives          */
    int32_t lbvh_threshold;    /* 0 = the scene's bvh_threshold if it gives one, else 64 */
    int32_t batch_samples;     /* pixel-samples per wavefront batch, 0 = default (128 Mi,
                                  halved while the pool does not fit in device memory)   */
    int64_t pool_rays;         /* ray-pool capacity per batch, 0 = default (4 x batch)  */
    void* stream;              /* hipStream_t to run on, NULL = a stream the ctx owns.
                                * Work queued on it before a call is seen by the call, and every call
                                * returns with its work on it complete.  Thin ray generations also use a
                                * second, internal stream that forks from and joins this one inside the call. */
} rt_options;

typedef struct rt_stats {
    uint64_t rays_primary;     /* rays that reach an intersection query, by class --     */
    uint64_t rays_secondary;   /* the reference's two `intersect` call sites             */
    uint64_t rays_shadow;      /* (src/raytrace.cpp:139 and :536)                        */
    /* traversal work of the closest-hit [0] and any-hit [1] kernels; counted only while
     * rt_set_work_counters(ctx, 1) is in force (counting costs ~8 % of traversal time), else 0 */
    uint64_t nodes_visited[2]; /* LBVH inner-node records fetched (64 B each): one per ray
                                  in the per-lane walk, one per 64-ray wave in the packet walk */
    uint64_t sphere_tests[2];  /* sphere records fetched (16 B each), same convention    */
    uint64_t tri_tests[2];     /* triangle records fetched (36 B each), same convention  */
    uint64_t lane_node_tests[2]; /* box-pair tests computed, counted per ray             */
    uint64_t lane_prim_tests[2]; /* exact primitive tests computed, counted per ray      */
    double ms_render;          /* device time of the whole call (HIP events)             */
    /* per-stage device time, filled only when the environment variable
       RT_STAGE_TIMES is set (HIP events between the stage launches) */
    double ms_extend;          /* closest-hit kernels                                    */
    double ms_shadow;          /* any-hit kernels                                        */
    double ms_shade;           /* shading + spawn + compaction kernels                   */
    double ms_other;           /* generate, resolve, finish                              */
    int32_t launches_extend;   /* kernel launches behind ms_extend / ms_shadow           */
    int32_t launches_shadow;
    int32_t batches;
    int32_t retries;           /* batches re-run at half size after a ray-pool overflow  */
    int32_t accel_used;        /* RT_ACCEL_BRUTE or RT_ACCEL_LBVH                        */
    int32_t generations;       /* deepest ray generation that held any ray, +1           */
    double ms_host_enqueue;    /* host time spent queueing the frame's launches (wall clock,
                                  before the wait for the device)                        */
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* ------------------------------------------------------------ life cycle */
/* Replaces `RayTrace t = RayTrace();` (src/main.cpp:56, src/raytrace.cpp:33-36). */
int rt_create(rt_ctx** out, const rt_options* opt);
void rt_destroy(rt_ctx* ctx);
const char* rt_strerror(int status);
/* Text of the last failure on this thread (HIP error string, parse message ...). */
const char* rt_last_error(void);

/* The host-side checks rt_load_scene applies to a descriptor before anything touches the
 * device (needs no GPU): RT_ERR_INVALID for non-positive film / sample rate, negative counts, a
 * positive count with a NULL array, a material index out of range; RT_ERR_UNSUPPORTED for
 * max_depth > 63, more than 2^26 primitives, 2^29 or more materials, a film beyond 65535 tiles a side. */
int rt_validate_scene(const rt_scene_desc* scene);

/* Replaces the by-value reads of `Scene&` (src/parse.h:311-327): packs the scene
 * as SoA in HBM and, when selected, builds the LBVH that stands where the
 * reference's BVH::make stood (src/raytrace.cpp:51-58, :652-788). */
int rt_load_scene(rt_ctx* ctx, const rt_scene_desc* scene);

/* Replaces `Image* RayTrace::rayTrace(Scene&)` (src/raytrace.cpp:44-129).
 * rgb_out: caller-allocated width*height*3 bytes, row-major, top row first, R,G,B
 * -- the reference's `pixels[y*width + x]` order (src/image.h:61) without the
 * alpha byte.  rgbf_out (may be NULL): width*height*3 floats, the averaged colour
 * before quantisation.  Both are HOST pointers. */
int rt_render(rt_ctx* ctx, uint8_t* rgb_out, float* rgbf_out, rt_stats* stats);

/* Progress of rt_render / rt_render_device / rt_render_shard: fn(user, pixels_done, pixels_total)
 * is called on the calling thread before the first batch (pixels_done = 0) and after every batch
 * of the frame.  Stands where the reference prints its `0%` ... `90%` lines from inside the pixel
 * loop (src/raytrace.cpp:97-103).  fn = NULL removes it. */
typedef void (*rt_progress_fn)(void* user, uint64_t pixels_done, uint64_t pixels_total);
int rt_set_progress_callback(rt_ctx* ctx, rt_progress_fn fn, void* user);

/* Same frame, but the result stays in device memory (d_rgb: width*height*3 bytes,
 * device pointer).  This is what bench.py times. */
int rt_render_device(rt_ctx* ctx, void* d_rgb, rt_stats* stats);

/* --------------------------------------------------- multi-GPU tile shards */
/* The image is cut into RT_TILE x RT_TILE tiles; tile (tx,ty) belongs to rank
 * (tx + 3*ty) mod world.  Each rank renders only its tiles into a packed buffer
 * (tile-major, RT_TILE*RT_TILE*3 bytes per tile, in the rank's tile order); the
 * caller gathers the buffers (RCCL) and rank 0 assembles the frame.  The
 * reference's counterpart is its OpenMP pixel loop (src/raytrace.cpp:92-96). */
#define RT_TILE 32
/* number of tiles rank owns for a width x height film */
int rt_shard_tiles(int width, int height, int rank, int world);
/* bytes of the packed buffer every rank must provide (same for all ranks: sized
 * for the rank with most tiles, so an all-gather of equal slices works) */
size_t rt_shard_bytes(int width, int height, int world);
int rt_render_shard(rt_ctx* ctx, int rank, int world, void* d_shard, rt_stats* stats);
/* d_all: world consecutive shard buffers of rt_shard_bytes each (device);
 * d_rgb: width*height*3 bytes (device).  On a context with a stream of its own the frame is complete on return; on a
 * context that runs on the caller's stream (rt_options.stream) the kernel is queued on that stream and returns at once
 * -- what precedes it there (the gather) and what follows (a copy, a synchronisation) is ordered by the stream, so a
 * frame costs one host synchronisation (the ray counters of rt_render_shard) instead of three.  The same holds for
 * rt_assemble_shards_noalloc. */
int rt_assemble_shards(rt_ctx* ctx, const void* d_all, int world, void* d_rgb);

/* One frame on n GPUs from ONE process: ctxs[r] (any devices, the same scene loaded into each)
 * renders shard r of n on a host thread of its own, the shards are copied to ctxs[0]'s device
 * (peer copies) and assembled there; rgb_out is a HOST buffer of width*height*3 bytes.  stats (may
 * be NULL): ray counts summed over the shards, ms_render = the slowest shard.  The in-process
 * counterpart of the reference's OpenMP pixel loop (src/raytrace.cpp:92-96); byte-identical to
 * rt_render for every n. */
int rt_render_multi(rt_ctx* const* ctxs, int n, uint8_t* rgb_out, rt_stats* stats);

/* rt_assemble_shards for callers that must not allocate (HIP graph capture, serving loops,
 * harnesses that forbid hipMalloc): no counterpart in the reference.  Same arguments, same
 * result, but the tile-row table the kernel needs lives in a fixed array in device memory that the library
 * holds per GPU (64 KiB) and is rebuilt by a small kernel on every call, so the call performs no
 * hipMalloc / hipFree.  The array is cut into RT_NOALLOC_SLOTS slices; a context owns one slice of its GPU's array
 * from rt_create to rt_destroy, so different contexts -- and graphs captured from them, replayed at the same time --
 * never share table entries.  Limits: world * ceil(height / RT_TILE) <= RT_NOALLOC_MAX_ROWS, and the context must
 * be one of the first RT_NOALLOC_SLOTS living on its GPU; else RT_ERR_UNSUPPORTED.  The work is queued on the
 * context's stream; the call waits for it unless that stream is being captured.  Calls on ONE context are
 * serialised by its owner as all calls are; two graphs captured from the same context must not run at once. */
#define RT_NOALLOC_MAX_ROWS 2048
#define RT_NOALLOC_SLOTS 8
int rt_assemble_shards_noalloc(rt_ctx* ctx, const void* d_all, int world, void* d_rgb);

/* -------------------------------------------------------- query utilities */
/* Closest-hit / any-hit queries on caller-supplied rays (host pointers), used to
 * validate the LBVH against brute force.  o,d: [n][3].  dmax: [n] squared
 * distances (NULL = +inf).  accel: RT_ACCEL_BRUTE, or RT_ACCEL_LBVH / _PACKET / _WIDE / _TEAM
 * (a walk of the hierarchy), optionally | RT_ACCEL_ALT_TREE.
 * prim_out[n]: -1 miss, else sphere index, or n_spheres + triangle index.
 * dist2_out[n] (may be NULL): squared distance of the hit.
 * The hierarchies are built to give the brute-force answer for rays that start where the rays of
 * a render can start -- inside the box that holds the camera's ray origins and every primitive:
 * what the reference's float tests accept beyond the ideal primitive grows with the distance to
 * the ray's origin (src/raytrace.cpp:232-241), and the boxes are padded for that region.  A query
 * whose origin lies outside it is answered by brute force whatever `accel` says. */
int rt_query_rays(rt_ctx* ctx, int64_t n, const float* o, const float* d, const float* dmax,
                  int any_hit, int accel, int32_t* prim_out, float* dist2_out);
/* Measurement hook (no counterpart in the reference): on != 0 makes the traversal kernels of
 * later renders count node / primitive fetches and tests into rt_stats; off by default.
 * The image does not depend on it. */
int rt_set_work_counters(rt_ctx* ctx, int on);
/* Measurement hook (no counterpart in the reference; its role is the reference's BVH::make, src/raytrace.cpp:652-788):
 * the hierarchies rt_load_scene built over the scene and which one renders.  For each -- [0] Karras (Morton order),
 * [1] PLOC, [2] binned surface-area build; depth 0 = not built or freed after the first render chose another --
 * the depth and the surface-area expectation of the work of a line through the root box: node visits (box-pair
 * tests) and exact primitive tests.  in_use: -1 before the first render of the scene, else the index kept. */
typedef struct rt_tree_stats {
    int32_t primitives;
    int32_t in_use;
    int32_t depth[3];
    int32_t reserved;
    double node_visits[3];
    double prim_tests[3];
} rt_tree_stats;
int rt_get_tree_stats(rt_ctx* ctx, rt_tree_stats* out);
/* on = 0: the next rt_load_scene's first render does not time the hierarchies and walks on a sample of the
 * frame (two to three extra frame times) but takes the choice that is right for most scenes -- for a caller
 * that renders ONE frame per scene, as the reference's main does (src/main.cpp:56-58).  Default on.
 * The image does not depend on it. */
int rt_set_calibration(rt_ctx* ctx, int on);
/* Replay of thin batches as HIP graphs.  A batch of at most 16 Mi samples whose launches repeat those of an earlier
 * batch of the context (same scene, film, partition, output buffer and choice of walks) is captured into a graph the
 * second time and launched as one from then on: such a frame is a chain of dozens of launches that takes the host
 * about as long to queue as the device to run.  on = 0 turns it off (and drops the graphs), 1 on (the default),
 * negative leaves it as it is; replays, if not NULL, receives the number of batches of this context that ran as a
 * graph launch so far.  Not used while stage times are taken.  The image does not depend on it.
 * No counterpart in the reference (its frame loop has no launches to queue, src/raytrace.cpp:92-122). */
int rt_set_graphs(rt_ctx* ctx, int on, uint64_t* replays);
/* Measurement hook: on = 1 makes later renders fill rt_stats.ms_extend / ms_shadow / ms_shade /
 * ms_other (HIP events between the stage launches); on = 2 only ms_extend, on = 3 only ms_shadow (an
 * event before and after every launch of that kind: a third of the events, for frames short enough to
 * notice them).  The environment variable RT_STAGE_TIMES=1|2|3 does the same for every context. */
int rt_set_stage_times(rt_ctx* ctx, int on);
/* Primary ray of (x[i], y[i], sample j[i]) -- RayTrace::getRay (src/raytrace.cpp:441-485). */
int rt_primary_rays(rt_ctx* ctx, int64_t n, const int32_t* x, const int32_t* y, const int32_t* j,
                    float* o_out, float* d_out);
/* value of the shared jitter sequence standing in for rand(): 0 .. 2^31-1 */
uint32_t rt_jitter_hash(uint32_t seed, uint64_t counter);

/* ----------------------------------------------- scene file and BMP output */
/* Parser for the reference's text scene format (Scene::parseScene,
 * src/parse.cpp:92-271).  flags: RT_PARSE_REF_COMPAT reproduces
 * `film_resolution W H` -> W x W (src/parse.cpp:119); default honours H.
 * out_name receives the `output_image` token (default "raytraced.bmp"). */
enum { RT_PARSE_REF_COMPAT = 1 };
int rt_parse_scene_file(const char* path, int flags, rt_scene_desc** out, char* out_name,
                        size_t out_name_cap);
void rt_free_scene(rt_scene_desc* scene);
/* file name rule of src/main.cpp:65: name.substr(0, name.find(".bmp") + 4) */
int rt_output_name(const char* image_name, char* out, size_t cap);
/* 24-bit BMP exactly as Image::Write -> BMP::WriteToFile emits it
 * (src/image.cpp:109-124, src/EasyBMP.cpp:422-598). */
int rt_write_bmp24(const char* path, const uint8_t* rgb, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H_ */
