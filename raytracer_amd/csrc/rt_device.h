// Device-side records and the reference-exact arithmetic of the trace loop.
//
// Arithmetic contract (reference src/parse.h:25-75, src/parse.cpp:301-396): a
// vector is three floats; vector/vector and vector/scalar operators work in
// float, a double scalar is narrowed to float at the call; dot / cross /
// magnitudeSq / lengthSq are float sums evaluated left to right; magnitude() is
// (float)sqrt((double)sumsq) -- identical to a correctly rounded sqrtf; norm()
// is three float divisions.  The translation unit is compiled with
// -ffp-contract=off so no product and sum is ever fused; the only fused
// operations are the explicit __builtin_fmaf calls of the LBVH box test, which
// is conservative and never decides a hit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_MAX_GEN 64          // ray generations held per batch (max_depth + 1 <= RT_MAX_GEN)
#define RT_BLOCK 256
#define RT_TILE_PX 32          // must equal RT_TILE of the public header
#define RT_TILE_AREA (RT_TILE_PX * RT_TILE_PX)

struct F3 {
    float x, y, z;
};
__host__ __device__ inline F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__host__ __device__ inline F3 f3(const float4& v) { return f3(v.x, v.y, v.z); }
__host__ __device__ inline F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ inline F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ inline F3 operator*(F3 a, float c) { return f3(a.x * c, a.y * c, a.z * c); }
__host__ __device__ inline F3 operator/(F3 a, float c) { return f3(a.x / c, a.y / c, a.z / c); }
__host__ __device__ inline float dot3(F3 u, F3 v) { return (u.x * v.x) + (u.y * v.y) + (u.z * v.z); }
__host__ __device__ inline F3 cross3(F3 u, F3 v) {
    return f3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
__host__ __device__ inline float magsq3(F3 a) { return (a.x * a.x) + (a.y * a.y) + (a.z * a.z); }
__host__ __device__ inline float mag3(F3 a) { return __builtin_sqrtf(magsq3(a)); }
__host__ __device__ inline F3 norm3(F3 a) { float m = mag3(a); return f3(a.x / m, a.y / m, a.z / m); }
__host__ __device__ inline F3 neg3(F3 a) { return f3(-a.x, -a.y, -a.z); }   // == a * -1.0f exactly

// ---------------------------------------------------------------- scene (SoA in HBM)
struct DLight {
    int type;
    float cr, cg, cb;
    float px, py, pz;
    float dx, dy, dz;
    float a1, a2;
};

// Frame constants: computed once per frame on the host exactly as the reference
// does (Scene::getPlaneDist src/parse.cpp:275-277, RayTrace::getExtremePoints
// src/raytrace.cpp:635-647, the loop-invariant part of getRay :443-445,:457-462).
struct DCamera {
    F3 pos, dir;      // camera position, (un-normalised) direction
    F3 p1;            // top-left corner of the view plane
    F3 urv, blv;      // right^ * dot(p2-p1, right^),  down^ * dot(p2-p1, down^)
    int width, height, spp, proj;
    uint32_t seed;
    double inv_w, inv_h;   // 1.0 / width, 1.0 / height, correctly rounded (see rt_cdiv)
    // frame constants that shorten the per-sample integer work: log2(spp) when spp is a power of two (else
    // -1), and, when the jitter counter of the whole film fits 32 bits, the hash of its (zero) high word
    int spp_shift;
    uint32_t counter32, jitter_hi0;
};

// a / b, correctly rounded, for a divisor whose reciprocal y = RN(1 / b) is at hand: two
// fused residual corrections of a * y (Markstein's division; exact unless b's significand is
// all ones or the quotient leaves the normal range, neither of which a film dimension or
// 2^31 - 1 can cause).  Five multiply-adds instead of the ~12-instruction hardware sequence
// with its quarter-rate reciprocal.
__host__ __device__ inline double rt_cdiv(double a, double b, double y) {
    double q = a * y;
    q = __builtin_fma(__builtin_fma(-q, b, a), y, q);
    return __builtin_fma(__builtin_fma(-q, b, a), y, q);
}

struct DScene {
    const float4* sph;       // [ns]  cx cy cz r
    const int* sph_mat;      // [ns]
    const float4* tri;       // [nt][3]  (a.xyz, ab.x) (ab.yz, ac.xy) (ac.z, mat, flag, -)
    const float4* tri_n;     // [nt][3]  n0 n1 n2 (w unused)
    const float4* mats;      // [nm][4]  (amb, ns) (dif, ior) (spec, -) (trans, -)
    const DLight* lights;    // [nl]
    int ns, nt, nl, max_depth;
    int nm;                  // materials
    F3 bg, ambient;
    // LBVH (valid when nodes != nullptr)
    const float4* nodes;     // [n_internal][4], see rt_lbvh.hip
    const float4* nodes4;    // 4-wide collapse of the same tree (nullptr if not built); root = node 0 in both.  wide_quant != 0:
                             // one 64-byte record [4] per node with 8-bit child boxes; else [8] with float boxes
    int wide_quant;
    const int* prim_idx;     // [ns+nt] sorted primitive ids (sphere i, or ns + triangle i)
    const float4* leaf;      // [ns+nt][3] primitive records in tree order, see rt_lbvh.hip
    int root;                // root reference
    int depth;               // most inner nodes above any leaf (bounds the traversal stacks)
    int n_prims;
    int profile;             // 1: collect the cycle counters of DCounters (tuning aid)
};

// ---------------------------------------------------------------- per-batch ray pool
// One record per traced ray, all generations of a batch kept until the bottom-up
// colour resolve.  SoA, 16-byte elements so every access is a coalesced dwordx4.
// Work distribution of the traversal kernels: RT_NQ counters per launch, each on a cache
// line of its own (a line sustains only ~90 returning atomics per microsecond), queue q
// handing out the 64-ray groups q, q + RT_NQ, q + 2 RT_NQ, ...
#define RT_MAX_WAVES 8192                                 // wave slots of the statistics array (>= waves of the largest traversal grid)
#define RT_NQ 256
#define RT_WQ_STRIDE 32                                  // uint32 per counter = 128 bytes
#define RT_WQ_LAUNCH (RT_NQ * RT_WQ_STRIDE)              // uint32 per launch
#ifndef RT_WQ_PROBES
#define RT_WQ_PROBES 2                                   // neighbours tried once the home queue is empty
#endif

struct DPool {
    // per ray, all generations of the batch
    uint32_t* wq;     // work-queue counters: launch slot 2g (extend of generation g), 2g+1 (shadow); zeroed per batch
    float4* o;        // origin xyz, w = parent code (bits): gen 0 -> sample index; else parent ray index
    float4* d;        // direction xyz, w = child slot (0 reflection, 1 refraction)
    float4* loc;      // rays that spawned children only: local colour rgb, w = (mat << 2 | refl | refr<<1) (bits)
    float4* cre;      // colour returned by the reflection child
    float4* cra;      // colour returned by the refraction child
    unsigned long long* wstats;   // [2][RT_MAX_WAVES][8] traversal work counters per wave slot (closest-hit, any-hit), see flush_stats
    uint32_t* link;   // generations >= 1: parent ray | child slot << 31, written with the ray (dense, in ray order)
    // per ray of the CURRENT generation (index = ray - generation start), rewritten every generation
    float4* hp;       // hit point xyz, w = primitive id (bits)           [hit rays only]
    float4* hn;       // shading normal xyz, w = material id | child slot << 31 (bits)
    float4* hd;       // direction of the ray that hit, w = parent code (as in o.w)
    // per 64-ray group: wave ballots, kept for every generation of the batch (index gbase[g] + group:
    // the bottom-up resolve pass finds the rays with pending children in them) ...
    unsigned long long* mask_hit;    // ray hit something
    unsigned long long* mask_refl;   // hit spawns a reflection ray
    unsigned long long* mask_refr;   // hit spawns a refraction ray
    // ... and their prefix sums, for the current generation only (index = group)
    uint32_t* child_base;  // next-generation rays before the group
    // single-pass scan: one state word per 256 groups of every generation of the batch (zeroed per batch), and
    // a block ticket per generation; scan_ticket lies right behind the scan_words state words and wq behind it
    // (one allocation, cleared by one fill per batch)
    unsigned long long* scan_state;
    uint32_t* scan_ticket;
    uint64_t scan_words;
    uint32_t* hitlist;     // compacted: h -> ray index within the generation, in ray order
    uint8_t* rflag;        // per ray of the current generation, written by extend launches made of team walks instead of the masks: 1 hit | 2 reflects | 4 refracts
    uint8_t* occl;    // [h * nl + light] 1 = in shadow
    float4* scol;     // per sample colour of the batch
    uint32_t capacity;     // rays
};

struct DCounters {
    uint32_t gen_count[RT_MAX_GEN + 1];   // rays per generation (appended by the shade kernel of g-1)
    uint32_t gbase[RT_MAX_GEN + 2];       // first 64-ray group of each generation in the mask arrays (groups never straddle generations)
    uint32_t hit_count[RT_MAX_GEN + 1];   // hits per generation
    uint32_t overflow;                    // set when a spawn did not fit the pool
    uint32_t pad;
    // traversal work, closest-hit [0] and any-hit [1] kernels: inner nodes fetched,
    // sphere tests, triangle tests
    unsigned long long nodes[2];
    unsigned long long sph_tests[2];
    unsigned long long tri_tests[2];
    unsigned long long lane_nodes[2];     // per-lane box-pair tests / primitive tests actually computed
    unsigned long long lane_prims[2];
    // tuning aid (RT_PROFILE=1): shader cycles inside packet walks, walk steps, and wave lifetimes
    unsigned long long prof_walk_cycles[2], prof_steps[2], prof_wave_cycles[2], prof_walks[2];
};

// ---------------------------------------------------------------- jitter
__host__ __device__ inline uint32_t rt_lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
// Stand-in for rand() (reference src/raytrace.cpp:448-449): a value in 0 .. 2^31-1
// that depends only on (seed, ((y*W+x)*spp+j)*2+k), so every device, the oracle
// and the reference draw the same jitter for the same sample.
__host__ __device__ inline uint32_t rt_jitter(uint32_t seed, uint64_t counter) {
    uint32_t hi = rt_lowbias32((uint32_t)(counter >> 32) + seed * 0x9E3779B9U + 0x85EBCA6BU);
    return rt_lowbias32((uint32_t)counter ^ hi) >> 1;
}
