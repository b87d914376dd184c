// Wavefront pipeline of the trace loop for gfx950 (MI355X).
//
//   generate -> [ extend -> shadow -> shade/spawn ] per ray generation -> resolve -> finish
//
// The reference recursion (evaluateRayTree -> getColor -> evaluateRayTree,
// src/raytrace.cpp:131-151, :488-632) is unrolled generation by generation: all
// rays of recursion depth g form one contiguous slice of a ray pool; the shade
// kernel of depth g appends the reflection / refraction rays of depth g+1 at the
// offsets a prefix sum over wave64 ballots assigned (no atomics).  Colours flow
// back bottom-up: a ray adds spec*child and trans*child to its local colour in
// the reference's order (:594-596, :625-627) and hands the sum to its parent's
// slot, so every float operation of the recursion happens exactly once and in
// the same order -- the image does not depend on batch size, queue order or
// the number of GPUs.
//
// No kernel needs a host round-trip for its size: every launch is a fixed
// persistent grid that reads the generation's ray count from device memory and
// strides over it, so a whole batch is enqueued asynchronously.
#include "rt_trace.h"
#include "rt_kernels.h"
#include "rt_packet_asm.h"

// k_shade: the compiler settles for 3 waves per SIMD (150 VGPRs); capped at 128 registers it spills about
// twenty dwords, and the kernel waits on memory most of the time, so the fourth wave more than pays for them
// (a fifth does not: measured).
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 4
#endif
#ifdef RT_WAVES
#define RT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(RT_WAVES, RT_WAVES)))
#else
// the per-lane walks hide memory latency with waves: 5 per SIMD at the price of a few spilled dwords;
// the packet walks fit 6 without spilling -- and MUST NOT spill: their traversal stack lives in the lanes of
// a vector register written with v_writelane, and a spill of it under a partial EXEC mask would lose entries.
// Their counting variants carry more live values, so they settle for 5 waves (measurement only); the build
// fails if any packet kernel uses scratch memory (tools/isa_stats.py --check).
#define RT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu((ACCEL == 2 || ACCEL == 4) ? RT_LANE_WAVES : ACCEL == 5 ? RT_TEAM_WAVES : (ACCEL == 3 ? (COUNT ? RT_PACKET_WAVES - 1 : RT_PACKET_WAVES) : 1), 8)))
#endif
#ifndef RT_LANE_WAVES
#define RT_LANE_WAVES 5
#endif
#ifndef RT_TEAM_WAVES
#define RT_TEAM_WAVES 5
#endif
#ifndef RT_PACKET_WAVES
#define RT_PACKET_WAVES 6
#endif

#include <algorithm>

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ uint32_t gen_start(const DCounters* __restrict__ c, int g) {
    uint32_t s = 0;
    for (int i = 0; i < g; i++) s += c->gen_count[i];
    return s;
}
__device__ __forceinline__ uint32_t gen_size(const DCounters* __restrict__ c, int g, uint32_t start,
                                    uint32_t capacity) {
    uint32_t n = c->gen_count[g];
    if (start >= capacity) return 0;
    uint32_t room = capacity - start;
    return n < room ? n : room;
}

// Dynamic distribution of 64-item groups over the waves of a persistent grid.  A static
// stride leaves waves with cheap rays (sky) idle while the ones that drew expensive rays
// finish: measured, the average wave of the closest-hit kernel was alive for only 65 % of
// the launch.  Here every wave draws the next group of its home queue with one atomic
// (issued BEFORE the current group is processed, so its latency is hidden), and once
// the home queue is empty it helps RT_WQ_PROBES neighbours (more probes only lengthen the tail).  The group a wave processes
// only selects WHICH rays it works on; results land at positions fixed by the ray index.
__device__ __forceinline__ uint32_t wq_grab(uint32_t* counter, uint32_t lane) {
    uint32_t k = 0;
    if (lane == 0) k = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return k;
}
template <class F>
__device__ __forceinline__ void for_each_group(uint32_t* wq, uint64_t ngroups, F&& body) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    // Fewer groups than waves (the thin late generations): one group per wave, no queue.  With the
    // queue a wave draws its second group before it starts the first, so half of the waves would
    // walk two groups back to back while the other half found the queues empty and left.
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const bool one_each = ngroups <= (uint64_t)nwaves;
    if (one_each && wave >= ngroups) return;
    // every queue must be some wave's home queue (a queue that is nobody's home and no neighbour's would keep its
    // groups): a grid of fewer waves than queues uses as many queues as it has waves
    const uint32_t nq = nwaves < RT_NQ ? nwaves : RT_NQ;
    uint32_t q = wave % nq;
    int probes = 0;
    uint32_t k = 0;
    if (!one_each) k = (uint32_t)__builtin_amdgcn_readfirstlane((int)wq_grab(wq + q * RT_WQ_STRIDE, lane));
    while (true) {
        uint64_t grp = wave;
        uint32_t kn = 0;
        if (!one_each) {
            grp = (uint64_t)k * nq + q;
            if (grp >= ngroups) {
                if (++probes > RT_WQ_PROBES) break;
                q = (q + 1) % nq;
                k = (uint32_t)__builtin_amdgcn_readfirstlane((int)wq_grab(wq + q * RT_WQ_STRIDE, lane));
                continue;
            }
            kn = wq_grab(wq + q * RT_WQ_STRIDE, lane);          // next group, in flight during body()
        }
        body(grp);
        if (one_each) break;
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)kn);
    }
}

// The same hand-out one draw at a time, for a wave that asks for its next chunk of work itself: next() gives a
// group of the launch that nobody else gets, or false when the home queue and its neighbours are empty.  The draw
// for the following call is issued at once, so its latency hides behind the work on the group just handed out.
struct GroupSource {
    uint32_t* wq;
    uint64_t ngroups;
    uint32_t q, nq, lane, ticket;
    int probes;
    __device__ __forceinline__ void init(uint32_t* wq_, uint64_t ngroups_) {
        wq = wq_; ngroups = ngroups_;
        lane = threadIdx.x & 63u;
        const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
        const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
        nq = nwaves < RT_NQ ? nwaves : RT_NQ;           // (as in for_each_group: no queue without a home wave)
        q = wave % nq;
        probes = 0;
        ticket = wq_grab(wq + q * RT_WQ_STRIDE, lane);
    }
    __device__ __forceinline__ bool next(uint64_t& grp) {
        while (true) {
            const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
            grp = (uint64_t)k * nq + q;
            if (grp < ngroups) { ticket = wq_grab(wq + q * RT_WQ_STRIDE, lane); return true; }
            if (++probes > RT_WQ_PROBES) return false;
            q = (q + 1) % nq;
            ticket = wq_grab(wq + q * RT_WQ_STRIDE, lane);
        }
    }
};

// A ray of generation g whose colour is final hands it to its parent (`parent` = the parent's pool
// index, or the sample index for g == 0; `slot` = 0 for a reflection ray, 1 for a refraction ray).
// A parent with ONE child is finished by that very colour: local + spec (or trans) * child, the
// reference's own operation (src/raytrace.cpp:594-596, :625-627), so the walk goes on to the
// grandparent, and so on up to the sample.  A parent with TWO children needs both colours, added
// in a fixed order: the child parks its colour in the parent's cre / cra slot and k_resolve
// finishes that parent later.  The parent's record (`loc`: local colour + material and spawn
// bits) was written by the shade launch of the generation before.
// `l` = the parent's loc record when the caller has fetched it already (g > 0), see deliver below.
__device__ __forceinline__ void deliver_from(const DScene& S, const DPool& P, int g, uint32_t parent, uint32_t slot, F3 c, float4 l) {
    while (g > 0) {
        const uint32_t code = fbits(l.w);
        if ((code & 3u) == 3u) {
            const float4 v = make_float4(c.x, c.y, c.z, 0.0f);
            if (slot == 0) P.cre[parent] = v;
            else P.cra[parent] = v;
            return;
        }
        const float4 k = S.mats[4 * (size_t)(code >> 2) + 2 + slot];     // specular for slot 0, transmissive for slot 1
        float rval = l.x, gval = l.y, bval = l.z;
        rval += k.x * c.x; gval += k.y * c.y; bval += k.z * c.z;
        c = f3(rval, gval, bval);
        g--;
        if (g > 0) {
            const uint32_t link = P.link[parent];
            parent = link & 0x7FFFFFFFu;
            slot = link >> 31;
            l = P.loc[parent];
        }
    }
    P.scol[parent] = make_float4(c.x, c.y, c.z, 0.0f);
}
__device__ __forceinline__ void deliver(const DScene& S, const DPool& P, int g, uint32_t parent, uint32_t slot, F3 c) {
    float4 l = make_float4(0, 0, 0, 0);
    if (g > 0) l = P.loc[parent];
    deliver_from(S, P, g, parent, slot, c, l);
}

// ---------------------------------------------------------------- brute force (LDS staged)
// Spheres, then triangles, in index order with a strict `<`, which reproduces
// the reference's tie rules: lowest index wins inside a list (:250, :361) and a
// triangle replaces a sphere only when strictly closer (:172).  Primitives are
// staged through LDS in chunks so scenes of any size stream once per block, not
// once per ray.  All threads of the block must call (barriers inside).
#define BRUTE_CHUNK 768   // float4 slots: 768 spheres or 256 triangles per chunk (12 KB)

// fetch counts (records read: one per lane in the per-lane walk, one per WAVE in
// the packet walk) and per-lane test counts (arithmetic actually performed)
struct TraceStats {
    uint32_t nodes = 0, sph = 0, tri = 0, lane_nodes = 0, lane_prims = 0;   // per lane, per launch
    unsigned long long walk_cycles = 0, steps = 0, walks = 0;                // tuning aid, lane 0 only
};

template <bool ANY>
__device__ __forceinline__ void brute_trace(const DScene& S, float4* lds, bool active, F3 o, F3 d, float dmax,
                                   int& best, float& best_d2, HitUV& uv, TraceStats& ts) {
    double dd = (double)dot3(d, d);
    best = -1;
    best_d2 = dmax;
    bool done = !active;
    for (int base = 0; base < S.ns; base += BRUTE_CHUNK) {
        int n = min(BRUTE_CHUNK, S.ns - base);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = S.sph[base + i];
        __syncthreads();
        if (!done) {
            for (int i = 0; i < n; i++) {
                float d2, tf;
                if (sphere_test(o, d, dd, lds[i], dmax, d2, tf)) {
                    if (best < 0 || d2 < best_d2) { best = base + i; best_d2 = d2; uv.a = tf; }
                    if (ANY) { done = true; break; }
                }
            }
            ts.sph += n; ts.lane_prims += n;
        }
    }
    for (int base = 0; base < S.nt; base += BRUTE_CHUNK / 3) {
        int n = min(BRUTE_CHUNK / 3, S.nt - base);
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) lds[i] = S.tri[3 * (size_t)base + i];
        __syncthreads();
        if (!done) {
            for (int i = 0; i < n; i++) {
                F3 a, ab, ac;
                tri_load(lds + 3 * i, a, ab, ac);
                TriHit h;
                if (tri_test(o, d, a, ab, ac, dmax, h)) {
                    if (best < 0 || h.d2 < best_d2) { best = S.ns + base + i; best_d2 = h.d2; uv.a = (float)h.beta; uv.b = (float)h.gamma; }
                    if (ANY) { done = true; break; }
                }
            }
            ts.tri += n; ts.lane_prims += n;
        }
    }
}

// ---------------------------------------------------------------- LBVH traversal
// Node i = 16 floats: child 0 box as (lo.x, hi.x, lo.y, hi.y, lo.z, hi.z), child 1 box
// the same way, then (ref0, ref1, -, -).  ref >= 0: inner node; ref < 0: leaf holding
// sorted primitive ~ref.  Boxes are padded at build time by what the exact tests can
// accept beyond the ideal primitive (rt_lbvh.hip: prim_box) and the slab test is
// widened, so a primitive the exact test would accept is never culled; the exact
// tests and the (distance, primitive id) lexicographic minimum then give the
// brute-force answer whatever the visiting order.
#define LBVH_STACK 96   // >= deepest possible tree: 63 Morton bits + 32 index tie-break bits

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4f* cv4p;

// Per-ray constants of the slab test t = plane * inv - o * inv (one fused multiply-add
// per plane, two planes per packed instruction).  A zero direction component would
// make inv infinite and the difference an inf - inf; a 1e-20 slope keeps the
// arithmetic finite and means the same thing for any box of a scene (the ray stays
// inside or outside that slab).  The box test only has to be conservative -- boxes
// are padded by at least 1e-4 of their coordinates, far above this rounding -- it never
// decides a hit.
struct SlabRay {
    F3 inv, noi;   // 1 / d and -(o / d)
    float inv_dd;  // 1 / dot(d, d)
};
__device__ __forceinline__ SlabRay slab_ray(F3 o, F3 d) {
    F3 ds = f3(__builtin_fabsf(d.x) < 1e-20f ? __builtin_copysignf(1e-20f, d.x) : d.x,
               __builtin_fabsf(d.y) < 1e-20f ? __builtin_copysignf(1e-20f, d.y) : d.y,
               __builtin_fabsf(d.z) < 1e-20f ? __builtin_copysignf(1e-20f, d.z) : d.z);
    SlabRay r;
    // hardware reciprocals (1 ulp): the slab test is conservative by far more than that
    r.inv = f3(__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y), __builtin_amdgcn_rcpf(ds.z));
    r.noi = f3(-(o.x * r.inv.x), -(o.y * r.inv.y), -(o.z * r.inv.z));
    r.inv_dd = __builtin_amdgcn_rcpf(dot3(d, d));
    return r;
}
// Ray-parameter bound equivalent to a squared-distance bound (d2 = t^2 dot(d, d)), widened
// by 1e-4 -- three orders of magnitude above the error of the approximate reciprocal and
// square root -- so it only ever culls boxes that lie wholly beyond the bound.  An
// infinite (or NaN) bound culls nothing.
__device__ __forceinline__ float slab_tmax(const SlabRay& r, float d2) {
    return __builtin_amdgcn_sqrtf(d2 * r.inv_dd) * 1.0001f;
}

// One box from its three (t at lo plane, t at hi plane) pairs.  OCT < 8: the signs of
// the direction are known (bit k set = negative along axis k), so near and far planes
// are picked at compile time; OCT == 8: any direction, min / max per axis.
template <int OCT>
__device__ __forceinline__ bool slab_hit(v2f x, v2f y, v2f z, float tmax, float& tnear) {
    float nx, fx, ny, fy, nz, fz;
    if (OCT == 8) {
        nx = fminf(x.x, x.y); fx = fmaxf(x.x, x.y);
        ny = fminf(y.x, y.y); fy = fmaxf(y.x, y.y);
        nz = fminf(z.x, z.y); fz = fmaxf(z.x, z.y);
    } else {
        nx = (OCT & 1) ? x.y : x.x; fx = (OCT & 1) ? x.x : x.y;
        ny = (OCT & 2) ? y.y : y.x; fy = (OCT & 2) ? y.x : y.y;
        nz = (OCT & 4) ? z.y : z.x; fz = (OCT & 4) ? z.x : z.y;
    }
    float tn = fmaxf(fmaxf(fmaxf(nx, ny), nz), 0.0f);
    float tf = fminf(fminf(fx, fy), fz);
    tnear = tn;
    return tn <= fminf(tf * 1.0000005f, tmax);
}
template <int OCT>
__device__ __forceinline__ void box_pair(v4f q0, v4f q1, v4f q2, const SlabRay& R, float tmax, bool& h0, bool& h1,
                                float& tn0, float& tn1) {
    const v2f ix = {R.inv.x, R.inv.x}, iy = {R.inv.y, R.inv.y}, iz = {R.inv.z, R.inv.z};
    const v2f ox = {R.noi.x, R.noi.x}, oy = {R.noi.y, R.noi.y}, oz = {R.noi.z, R.noi.z};
    v2f x0 = __builtin_elementwise_fma((v2f){q0.x, q0.y}, ix, ox);
    v2f y0 = __builtin_elementwise_fma((v2f){q0.z, q0.w}, iy, oy);
    v2f z0 = __builtin_elementwise_fma((v2f){q1.x, q1.y}, iz, oz);
    v2f x1 = __builtin_elementwise_fma((v2f){q1.z, q1.w}, ix, ox);
    v2f y1 = __builtin_elementwise_fma((v2f){q2.x, q2.y}, iy, oy);
    v2f z1 = __builtin_elementwise_fma((v2f){q2.z, q2.w}, iz, oz);
    h0 = slab_hit<OCT>(x0, y0, z0, tmax, tn0);
    h1 = slab_hit<OCT>(x1, y1, z1, tmax, tn1);
}

// exact test of one tree-ordered leaf record (see rt_lbvh.hip: k_leaves)
__device__ __forceinline__ bool leaf_test(F3 o, F3 d, double dd, float4 l0, float4 l1, float4 l2, float dmax,
                                 float& d2, HitUV& uv) {
    if (fbits(l2.z) != 0u) return sphere_test(o, d, dd, l0, dmax, d2, uv.a);
    TriHit h;
    bool hit = tri_test(o, d, f3(l0.x, l0.y, l0.z), f3(l0.w, l1.x, l1.y), f3(l1.z, l1.w, l2.x), dmax, h);
    d2 = h.d2;
    uv.a = (float)h.beta; uv.b = (float)h.gamma;
    return hit;
}

// One inner-node step of a per-lane walk: tests the children of node `cur`, continues with the
// nearest one that is hit and leaves the others on the lane's stack, nearest on top; with no
// child hit the next reference comes off the stack.  Returns the next reference (>= 0 inner
// node, < 0 leaf) or LANE_DONE when nothing is left.  WIDE: the 4-wide collapse of the tree (8
// float4 per node: four boxes, four references), which halves the number of dependent fetches.
#define LANE_DONE ((int)0x80000000)   // ~(2^31 - 1): never a leaf slot; also the reference of an empty wide slot

__device__ __forceinline__ void order2(float& ka, int& ra, float& kb, int& rb) {   // ascending by key
    const bool sw = kb < ka;
    const float k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const int r0 = sw ? rb : ra, r1 = sw ? ra : rb;
    ka = k0; kb = k1; ra = r0; rb = r1;
}

// Traversal stack of one lane, two ways.  PrivateStack: an array in the lane's private (scratch) memory, as deep
// as any tree can need.  LdsStack: entry k of the block's thread t at lds[k * RT_BLOCK + t] (a wave reads one row:
// no bank conflicts beyond the two halves of the wave), LANE_LDS_STACK deep; a walk that would go deeper sets
// `over` and its caller redoes that one ray with a PrivateStack.  On a scene far larger than the caches the pushes
// of the private stacks are HBM traffic -- the stacks of all resident waves together (130 MB) are many times the
// L2, measured 5.8 GB written per 10 M-ray shadow launch of the 11 M-primitive scene -- which LDS takes away.
#ifndef LANE_LDS_STACK
#define LANE_LDS_STACK 24
#endif
struct PrivateStack {      // (the array itself is a local of the caller: a struct that held it would drag sp into memory with it)
    int* e;                // [LBVH_STACK]
    int sp;
    __device__ __forceinline__ void reset() { sp = 0; }
    __device__ __forceinline__ void push(int v) { if (sp < LBVH_STACK) e[sp++] = v; }
    __device__ __forceinline__ void push_if(int v, bool c) { if (c) push(v); }
    __device__ __forceinline__ void settle() {}
    __device__ __forceinline__ bool empty() const { return sp <= 0; }
    __device__ __forceinline__ int pop() { return e[--sp]; }
};
struct LdsStack {
    int* col;          // this lane's column of the block's LDS area
    int sp;
    bool over;
    __device__ __forceinline__ void init(int* block_lds) { col = block_lds + threadIdx.x; sp = 0; over = false; }
    __device__ __forceinline__ void reset() { sp = 0; over = false; }
    __device__ __forceinline__ void push(int v) {
        if (sp < LANE_LDS_STACK) col[sp * RT_BLOCK] = v;
        else over = true;
        sp++;
    }
    // push without a branch: the store always happens -- to the row the entry would take, or to the spare row past the
    // last one when the stack is full -- and only the depth depends on c; settle() after a step's pushes notes a
    // stack that outgrew its rows
    __device__ __forceinline__ void push_if(int v, bool c) {
        col[(sp < LANE_LDS_STACK ? sp : LANE_LDS_STACK) * RT_BLOCK] = v;
        sp += c ? 1 : 0;
    }
    __device__ __forceinline__ void settle() { if (sp > LANE_LDS_STACK) over = true; }
    __device__ __forceinline__ bool empty() const { return sp <= 0; }
    __device__ __forceinline__ int pop() { --sp; return col[(sp < LANE_LDS_STACK ? sp : LANE_LDS_STACK - 1) * RT_BLOCK]; }
};
#define LANE_STACK_F4 ((LANE_LDS_STACK + 1) * RT_BLOCK / 4)  // float4 slots of LDS a block of walkers with LdsStacks needs (one spare row, see push_if)

template <bool WIDE, bool ANY = false, class STK>
__device__ __forceinline__ int lane_inner_step(const DScene& S, int cur, const SlabRay& R, float tmax, STK& stk, TraceStats& ts) {
    // counted in 64-byte records fetched and box pairs tested: a wide node is two pairs, and one record when its
    // boxes are quantised, two when they are floats
    ts.nodes += (WIDE && !S.wide_quant) ? 2 : 1; ts.lane_nodes += WIDE ? 2 : 1;
    if (!WIDE) {
        const v4f* nd = (const v4f*)S.nodes + 4 * (size_t)cur;
        v4f q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
        float tn0, tn1;
        bool h0, h1;
        box_pair<8>(q0, q1, q2, R, tmax, h0, h1, tn0, tn1);
        const int c0 = (int)fbits(q3.x), c1 = (int)fbits(q3.y);
        if (h0 && h1) {
            const bool swap = tn1 < tn0;
            stk.push(swap ? c0 : c1);
            return swap ? c1 : c0;
        }
        if (h0) return c0;
        if (h1) return c1;
        return stk.empty() ? LANE_DONE : stk.pop();
    } else {
        float k0, k1, k2, k3;
        bool h0, h1, h2, h3;
        int r0, r1, r2, r3;
        if (S.wide_quant) {
            // one 64-byte record: origin + step exponents, the child boxes as bytes on that grid, four references
            const v4f* nd = (const v4f*)S.nodes4 + 4 * (size_t)cur;
            const v4f q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
            const uint32_t eb = fbits(q0.w);
            // plane distance t = (origin + q * step) * inv - o * inv = q * (step * inv) + (origin * inv + noi)
            const float ax = bitsf((eb & 0xFFu) << 23) * R.inv.x, ay = bitsf(((eb >> 8) & 0xFFu) << 23) * R.inv.y,
                        az = bitsf(((eb >> 16) & 0xFFu) << 23) * R.inv.z;
            const float bx = __builtin_fmaf(q0.x, R.inv.x, R.noi.x), by = __builtin_fmaf(q0.y, R.inv.y, R.noi.y),
                        bz = __builtin_fmaf(q0.z, R.inv.z, R.noi.z);
            // The plane a ray enters a slab by depends only on the sign of its slope: the near / far byte of all four
            // children is picked per axis with one select on the packed word, so the box tests need no min / max.
            const bool sx = R.inv.x < 0.0f, sy = R.inv.y < 0.0f, sz = R.inv.z < 0.0f;
            const uint32_t lx = fbits(q1.x), ly = fbits(q1.y), lz = fbits(q1.z), hx = fbits(q1.w), hy = fbits(q2.x), hz = fbits(q2.y);
            const uint32_t nx = sx ? hx : lx, fx = sx ? lx : hx, ny = sy ? hy : ly, fy = sy ? ly : hy, nz = sz ? hz : lz, fz = sz ? lz : hz;
            float kk[4]; bool hh[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                // (near, far) plane of an axis in one packed multiply-add
                const v2f tx = __builtin_elementwise_fma((v2f){(float)((nx >> (8 * c)) & 0xFFu), (float)((fx >> (8 * c)) & 0xFFu)}, (v2f){ax, ax}, (v2f){bx, bx});
                const v2f ty = __builtin_elementwise_fma((v2f){(float)((ny >> (8 * c)) & 0xFFu), (float)((fy >> (8 * c)) & 0xFFu)}, (v2f){ay, ay}, (v2f){by, by});
                const v2f tz = __builtin_elementwise_fma((v2f){(float)((nz >> (8 * c)) & 0xFFu), (float)((fz >> (8 * c)) & 0xFFu)}, (v2f){az, az}, (v2f){bz, bz});
                hh[c] = slab_hit<0>(tx, ty, tz, tmax, kk[c]);
            }
            k0 = kk[0]; k1 = kk[1]; k2 = kk[2]; k3 = kk[3];
            h0 = hh[0]; h1 = hh[1]; h2 = hh[2]; h3 = hh[3];
            r0 = (int)fbits(q2.z); r1 = (int)fbits(q2.w); r2 = (int)fbits(q3.x); r3 = (int)fbits(q3.y);
        } else {
            // 128 bytes: four float boxes, then the four references
            const v4f* nd = (const v4f*)S.nodes4 + 8 * (size_t)cur;
            v4f q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3], q4 = nd[4], q5 = nd[5], q6 = nd[6];
            box_pair<8>(q0, q1, q2, R, tmax, h0, h1, k0, k1);
            box_pair<8>(q3, q4, q5, R, tmax, h2, h3, k2, k3);
            r0 = (int)fbits(q6.x); r1 = (int)fbits(q6.y); r2 = (int)fbits(q6.z); r3 = (int)fbits(q6.w);
        }
        const float inf = __builtin_inff();
        // a missed or empty slot sorts last (an empty slot's inverted box reads as an infinite slab in the min / max test)
        k0 = (h0 && r0 != LANE_DONE) ? k0 : inf; k1 = (h1 && r1 != LANE_DONE) ? k1 : inf;
        k2 = (h2 && r2 != LANE_DONE) ? k2 : inf; k3 = (h3 && r3 != LANE_DONE) ? k3 : inf;
        if (ANY) {
            // a shadow query ends with ANY hit: the nearest child goes first, the other hit children wait in slot
            // order -- no sorting network (25 of a step's 150 vector instructions)
            const float m = fminf(fminf(fminf(k0, k1), k2), k3);
            const bool s0 = k0 == m, s1 = !s0 && k1 == m, s2 = !s0 && !s1 && k2 == m, s3 = !s0 && !s1 && !s2;
            stk.push_if(r3, k3 < inf && !s3);
            stk.push_if(r2, k2 < inf && !s2);
            stk.push_if(r1, k1 < inf && !s1);
            stk.push_if(r0, k0 < inf && !s0);
            stk.settle();
            if (m < inf) return s0 ? r0 : s1 ? r1 : s2 ? r2 : r3;
            return stk.empty() ? LANE_DONE : stk.pop();
        }
        order2(k0, r0, k1, r1); order2(k2, r2, k3, r3);      // sorting network for four
        order2(k0, r0, k2, r2); order2(k1, r1, k3, r3);
        order2(k1, r1, k2, r2);
        stk.push_if(r3, k3 < inf);    // farthest first: the nearer ones come off first
        stk.push_if(r2, k2 < inf);
        stk.push_if(r1, k1 < inf);
        stk.settle();
        if (k0 < inf) return r0;
        return stk.empty() ? LANE_DONE : stk.pop();
    }
}

// Per-lane while-while traversal with a private stack: every lane walks its own path.
template <bool ANY, bool WIDE = false>
__device__ __forceinline__ void lbvh_trace(const DScene& S, bool active, F3 o, F3 d, float dmax, int& best,
                                  float& best_d2, HitUV& uv, TraceStats& ts) {
    best = -1;
    best_d2 = dmax;
    if (!active) return;
    double dd = (double)dot3(d, d);
    const SlabRay R = slab_ray(o, d);
    float tmax = slab_tmax(R, dmax);
    int stack_mem[LBVH_STACK];
    PrivateStack stk{stack_mem, 0};
    int cur = 0;                                   // both trees keep their root at node 0
    while (cur != LANE_DONE) {
        if (cur >= 0) {
            cur = lane_inner_step<WIDE, ANY>(S, cur, R, tmax, stk, ts);
        } else {
            const float4* lf = S.leaf + 3 * (size_t)(~cur);
            float4 l0 = lf[0], l1 = lf[1], l2 = lf[2];
            int prim = (int)fbits(l2.y);
            if (fbits(l2.z) != 0u) ts.sph++; else ts.tri++;
            ts.lane_prims++;
            float d2 = 0;
            HitUV cand;
            if (leaf_test(o, d, dd, l0, l1, l2, dmax, d2, cand)) {
                if (ANY) { best = prim; best_d2 = d2; return; }
                if (best < 0 || d2 < best_d2 || (d2 == best_d2 && prim < best)) {
                    best = prim;
                    best_d2 = d2;
                    uv = cand;
                    tmax = slab_tmax(R, d2);
                }
            }
            cur = stk.empty() ? LANE_DONE : stk.pop();
        }
    }
}

// Packet traversal: the 64 rays of a wave walk the tree TOGETHER.  The current
// node is wave-uniform, so node and leaf records come through the scalar data
// path (one fetch per wave, operands in SGPRs), the traversal stack lives in the
// lanes of two VGPRs (entry k in lane k: v_writelane / v_readlane), and there is
// no divergence inside the loop.  A child is entered when ANY lane's ray hits its
// box; the nearer child by majority vote goes first.  Every lane still applies its
// own distance bound and the exact tests, so the result per ray is the same as
// the per-lane walk (and as brute force); only the visiting order is shared.
// When all rays of the wave point into the same octant -- primary rays, shadow
// rays of a directional light -- the descent runs a variant compiled for that
// octant, which needs no per-axis min / max.  All 64 lanes must call.
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// one entry of the wave-shared stack: lane `sel` of v takes the wave-uniform value
__device__ __forceinline__ void lane_write(int& v, int value, int sel) {
    // the lane select goes through m0 (saved and restored): gfx9 VALU encodings may name
    // only one SGPR besides it
    int keep;
    value = __builtin_amdgcn_readfirstlane(value);
    sel = __builtin_amdgcn_readfirstlane(sel);
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 1\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                 : "+v"(v), "=&s"(keep) : "s"(value), "s"(sel));
}

#define PACKET_DONE ((int)0x80000000)   // ~(2^31 - 1): never a leaf slot

struct PacketState {
    int sp;                        // wave-uniform stack depth
    int vs0, vs1;                  // entries 0..63 / 64..127, one per lane
    unsigned long long alive;      // lanes still looking for a hit
    uint32_t n_nodes, n_lane_nodes;
};

// Walks inner nodes from `cur` until a leaf comes up (returned, < 0) or the stack is
// empty (PACKET_DONE).
template <int OCT>
__device__ __forceinline__ int packet_descend(cv4p nodes, int cur, const SlabRay& R, float tmax, PacketState& st) {
    while (cur >= 0) {
        cv4p nd = nodes + 4 * (size_t)cur;
        v4f q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
        float tn0, tn1;
        bool h0, h1;
        box_pair<OCT>(q0, q1, q2, R, tmax, h0, h1, tn0, tn1);
        const unsigned long long m0 = ballot64(h0) & st.alive, m1 = ballot64(h1) & st.alive;
        st.n_nodes++;
        st.n_lane_nodes += (uint32_t)__popcll(st.alive);
        const int c0 = (int)fbits(q3.x), c1 = (int)fbits(q3.y);
        if (m0 != 0ull && m1 != 0ull) {
            // lanes that hit both boxes vote with their nearer one, the others with the one they hit
            const unsigned long long le = ballot64(tn0 <= tn1);
            const int v0 = __popcll(m0 & (~m1 | le));
            const int v1 = __popcll(m0 | m1) - v0;
            const bool swap = v1 > v0;
            const int nearc = swap ? c1 : c0, farc = swap ? c0 : c1;
            if (st.sp < 64) lane_write(st.vs0, farc, st.sp);
            else lane_write(st.vs1, farc, st.sp - 64);
            st.sp++;
            cur = nearc;
        } else if (m0 != 0ull) cur = c0;
        else if (m1 != 0ull) cur = c1;
        else {
            if (st.sp == 0) return PACKET_DONE;
            st.sp--;
            cur = (st.sp < 64) ? __builtin_amdgcn_readlane(st.vs0, st.sp) : __builtin_amdgcn_readlane(st.vs1, st.sp - 64);
        }
    }
    return cur;
}

template <bool ANY, bool FAST>
__device__ __forceinline__ void lbvh_trace_packet_impl(const DScene& S, bool active, F3 o, F3 d, float dmax, int& best,
                                              float& best_d2, HitUV& uv, TraceStats& ts) {
    best = -1;
    best_d2 = dmax;
    PacketState st;
    st.alive = ballot64(active);
    if (st.alive == 0ull) return;
    const int lane = (int)(threadIdx.x & 63);
    double dd = (double)dot3(d, d);
    const SlabRay R = slab_ray(o, d);
    float tmax = slab_tmax(R, dmax);
    // octant shared by every live ray of the wave, or 8
    int oct = 0;
    {
        const unsigned long long sx = ballot64(R.inv.x < 0.0f) & st.alive, sy = ballot64(R.inv.y < 0.0f) & st.alive,
                                 sz = ballot64(R.inv.z < 0.0f) & st.alive;
        const bool uniform = (sx == 0ull || sx == st.alive) && (sy == 0ull || sy == st.alive) && (sz == 0ull || sz == st.alive);
        oct = uniform ? ((sx ? 1 : 0) | (sy ? 2 : 0) | (sz ? 4 : 0)) : 8;
    }
    // The tree is read-only for the whole launch and the record address is
    // wave-uniform: reading through the constant address space makes the compiler
    // fetch a record with scalar loads (s_load_dwordx16) instead of 64 identical
    // per-lane loads.
    cv4p nodes = (cv4p)S.nodes;
    cv4p leaf = (cv4p)S.leaf;
    st.sp = 0; st.vs0 = 0; st.vs1 = 0; st.n_nodes = 0; st.n_lane_nodes = 0;
    uint32_t n_sph = 0, n_tri = 0, n_lane_prims = 0;
    int cur = S.root;
    unsigned long long t0 = S.profile ? __builtin_readcyclecounter() : 0ull;
    const pka_v2f pa = {R.inv.x, R.inv.y}, pb = {R.inv.z, R.noi.x}, pc = {R.noi.y, R.noi.z};
    while (true) {
        if (FAST) {
            int steps = 0;
            switch (oct) {
                case 0: pka_descend_0(S.nodes, st.alive, pa, pb, pc, tmax, cur, st.sp, steps, st.vs0); break;
                case 1: pka_descend_1(S.nodes, st.alive, pa, pb, pc, tmax, cur, st.sp, steps, st.vs0); break;
                case 2: pka_descend_2(S.nodes, st.alive, pa, pb, pc, tmax, cur, st.sp, steps, st.vs0); break;
                case 3: pka_descend_3(S.nodes, st.alive, pa, pb, pc, tmax, cur, st.sp, steps, st.vs0); break;
                case 4: pka_descend_4(S.nodes, st.alive, pa, pb, pc, tmax, cur, st.sp, steps, st.vs0); break;
                case 5: pka_descend_5(S.nodes, st.alive, pa, pb, pc, tmax, cur, st.sp, steps, st.vs0); break;
                case 6: pka_descend_6(S.nodes, st.alive, pa, pb, pc, tmax, cur, st.sp, steps, st.vs0); break;
                case 7: pka_descend_7(S.nodes, st.alive, pa, pb, pc, tmax, cur, st.sp, steps, st.vs0); break;
                default: pka_descend_any(S.nodes, st.alive, pa, pb, pc, tmax, cur, st.sp, steps, st.vs0); break;
            }
            st.n_nodes += (uint32_t)steps;
            st.n_lane_nodes += (uint32_t)steps * (uint32_t)__popcll(st.alive);
        } else {
            cur = packet_descend<8>(nodes, cur, R, tmax, st);
        }
        cur = __builtin_amdgcn_readfirstlane(cur);
        if (cur == PACKET_DONE) break;
        cv4p lf = leaf + 3 * (size_t)(~cur);
        v4f r0 = lf[0], r1 = lf[1], r2 = lf[2];
        float4 l0 = make_float4(r0.x, r0.y, r0.z, r0.w), l1 = make_float4(r1.x, r1.y, r1.z, r1.w),
               l2 = make_float4(r2.x, r2.y, r2.z, r2.w);
        int prim = (int)fbits(l2.y);
        if (fbits(l2.z) != 0u) n_sph++; else n_tri++;
        n_lane_prims += (uint32_t)__popcll(st.alive);
        bool found = false;
        if ((st.alive >> lane) & 1ull) {
            float d2 = 0;
            HitUV cand;
            if (leaf_test(o, d, dd, l0, l1, l2, dmax, d2, cand)) {
                if (ANY) { best = prim; best_d2 = d2; found = true; }
                else if (best < 0 || d2 < best_d2 || (d2 == best_d2 && prim < best)) {
                    best = prim;
                    best_d2 = d2;
                    uv = cand;
                    tmax = slab_tmax(R, d2);
                }
            }
        }
        if (ANY) {
            st.alive &= ~ballot64(found);
            if (st.alive == 0ull) break;
        }
        if (st.sp == 0) break;
        st.sp--;
        cur = (st.sp < 64) ? __builtin_amdgcn_readlane(st.vs0, st.sp) : __builtin_amdgcn_readlane(st.vs1, st.sp - 64);
    }
    if (lane == 0) {
        ts.nodes += st.n_nodes; ts.sph += n_sph; ts.tri += n_tri;
        ts.lane_nodes += st.n_lane_nodes; ts.lane_prims += n_lane_prims;
        if (S.profile) { ts.walk_cycles += __builtin_readcyclecounter() - t0; ts.steps += st.n_nodes + n_sph + n_tri; ts.walks++; }
    }
}

// Trees at most 64 deep (all but degenerate scenes) take the hand-scheduled descent, whose
// stack is one register wide; deeper ones the compiled one.
template <bool ANY>
__device__ __forceinline__ void lbvh_trace_packet(const DScene& S, bool active, F3 o, F3 d, float dmax, int& best,
                                         float& best_d2, HitUV& uv, TraceStats& ts) {
    if (S.depth <= 64) lbvh_trace_packet_impl<ANY, true>(S, active, o, d, dmax, best, best_d2, uv, ts);
    else lbvh_trace_packet_impl<ANY, false>(S, active, o, d, dmax, best, best_d2, uv, ts);
}

template <int ACCEL, bool ANY>
__device__ __forceinline__ void trace(const DScene& S, float4* lds, bool active, F3 o, F3 d, float dmax,
                             int& best, float& best_d2, HitUV& uv, TraceStats& ts) {
    uv.a = 0; uv.b = 0;
    if (ACCEL == 1) brute_trace<ANY>(S, lds, active, o, d, dmax, best, best_d2, uv, ts);
    else if (ACCEL == 2) lbvh_trace<ANY, false>(S, active, o, d, dmax, best, best_d2, uv, ts);
    else if (ACCEL == 4) lbvh_trace<ANY, true>(S, active, o, d, dmax, best, best_d2, uv, ts);
    else lbvh_trace_packet<ANY>(S, active, o, d, dmax, best, best_d2, uv, ts);
}

// ---------------------------------------------------------------- team walk (the lanes of a wave share out its rays)
// A launch with few rays ends when its LONGEST walk does: a ray that grazes through a hedge visits hundreds
// of nodes, one dependent fetch after the other, while the chip stands all but empty.  Here a wave holds NR rays
// (records and traversal stacks in LDS) and its 64 lanes test 16 nodes of the 4-wide tree per round, one
// child box per lane; the 16 node slots are shared out evenly among the rays that still have entries on their
// stacks, so a ray gets 16 / NR nodes per round while all are alive and all 16 once it is the last one.  A
// child that is hit and is a leaf gets its exact test at once, in the same lane; one that is an inner node goes
// on its ray's stack -- entries in the order a depth-first walk would visit them (the children of the entry that
// was on top end up on top, nearest last).  The lanes look at boxes a depth-first walk would have culled after a
// nearer hit; the hit reported is the same (distance, primitive id) lexicographic minimum over the same exact
// tests: candidates meet in a 64-bit atomic minimum of (distance bits, primitive) in the ray's record.  A stack
// that would outgrow its LDS slice (TEAM_STACK entries; near the limit a ray takes one entry per round, which is a
// depth-first walk) hands the ray to the private-stack walk.
#ifndef TEAM_NR
#define TEAM_NR 8           // rays per wave, closest-hit walks
#endif
#ifndef TEAM_NRS
#define TEAM_NRS 8          // rays per wave, any-hit walks
#endif
#define TEAM_NR_MAX (TEAM_NR > TEAM_NRS ? TEAM_NR : TEAM_NRS)
#define TEAM_STACK 96
#define TEAM_REC_F4 9       // float4 per ray record (6 used; 36 words apart: the records of a wave start in different banks)
#ifndef TEAM_FLUSH
#define TEAM_FLUSH 16       // leaves waiting for their exact test before the wave stops walking to test them
#endif
#define TEAM_QUEUE 96       // >= TEAM_FLUSH - 1 + 64: a round adds at most one leaf per lane
#define TEAM_LDS_F4 (TEAM_NR_MAX * (TEAM_REC_F4 + TEAM_STACK / 4) + TEAM_QUEUE / 4)   // per wave: records, stacks, leaf queue

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {            // v of the lane CTRL names (quad_perm)
    return bitsf((uint32_t)__builtin_amdgcn_update_dpp(0, (int)fbits(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
#define DPP_QUAD_XOR1 0xB1      // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E      // quad_perm [2,3,0,1]
#define DPP_QUAD_XOR3 0x1B      // quad_perm [3,2,1,0]

// Record of ray r in LDS, float4 rec[r * TEAM_REC_F4 + ...]:
//   0: o.xyz, dmax      1: d.xyz, 1/dot(d,d)      2: 1/d.xyz, dd low word      3: -(o/d).xyz, dd high word
//   4: best primitive, best distance (bits; the two words are ONE 64-bit key, ~0 = no hit yet), u, v of that hit
//   5: stack depth, any-hit found, redo with the private-stack walk, -
// All 64 lanes call; lanes r < NR pass ray r of the wave (active = there is one) and get its result back.
template <bool ANY, int NR>
__device__ __forceinline__ void team_walk(const DScene& S, float4* wave_lds, bool active, F3 o, F3 d, float dmax, int& best,
                                          float& best_d2, HitUV& uv, TraceStats& ts) {
    static_assert(NR == 4 || NR == 8 || NR == 16, "16 node slots per round are shared by at most 16 rays");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = lane >> 2;
    const int c = (int)(lane & 3u);
    float4* rec = wave_lds;
    int* stacks = (int*)(wave_lds + NR * TEAM_REC_F4);
    uint32_t* queue = (uint32_t*)(wave_lds + TEAM_NR_MAX * (TEAM_REC_F4 + TEAM_STACK / 4));
    int qn = 0;                                                       // wave-uniform: leaves in the queue
    const double dd0 = (double)dot3(d, d);
    if (lane < (uint32_t)NR) {
        const SlabRay R = slab_ray(o, d);
        float4* r = rec + lane * TEAM_REC_F4;
        r[0] = make_float4(o.x, o.y, o.z, dmax);
        r[1] = make_float4(d.x, d.y, d.z, R.inv_dd);
        r[2] = make_float4(R.inv.x, R.inv.y, R.inv.z, __int_as_float(__double2loint(dd0)));
        r[3] = make_float4(R.noi.x, R.noi.y, R.noi.z, __int_as_float(__double2hiint(dd0)));
        r[4] = make_float4(bitsf(0xFFFFFFFFu), bitsf(0xFFFFFFFFu), 0.0f, 0.0f);
        r[5] = make_float4(__int_as_float(active ? 1 : 0), __int_as_float(0), __int_as_float(0), 0.0f);
        stacks[lane * TEAM_STACK] = 0;                            // both trees keep their root at node 0
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t alive = (uint32_t)ballot64(lane < (uint32_t)NR && active);
    const unsigned long long t0 = S.profile ? __builtin_readcyclecounter() : 0ull;
    uint32_t rounds = 0;
    while (alive != 0u) {
        rounds++;
        // wave-uniform part (scalar unit): how many rays are alive, how many of the 16 node slots each gets, and
        // the list of their numbers, four bits each
        const int n = __popc(alive);
        // node slots per ray this round = 16 / n, and ceil(256 / that) for the lanes' division by it, from tables
        // (n = 1 .. 8 a nibble / a byte each; nine or more rays alive: one slot each)
        const int share = n <= 8 ? (int)((0x1112347Full >> (4 * (n - 1))) & 15ull) + 1 : 1;
        const uint32_t recip = n <= 8 ? (uint32_t)((0x8080805640342010ull >> (8 * (n - 1))) & 255ull) : 256u;
        unsigned long long order = 0ull;
        {
            uint32_t a = alive;
            for (int i = 0; i < n; i++) { order |= (unsigned long long)__builtin_ctz(a) << (4 * i); a &= a - 1u; }
        }
        const int k = (int)((slot * recip) >> 8);                   // = slot / share: this lane works for the k-th ray alive,
        const int e = (int)slot - k * share;                        //   entry e of its share
        const bool mine = k < n;
        const int ray = mine ? (int)((order >> (4 * k)) & 15ull) : 0;
        const float4* rr = rec + ray * TEAM_REC_F4;
        int* stk = stacks + ray * TEAM_STACK;
        const int sp = mine ? __float_as_int(rr[5].x) : 0;
        // a popped node may push four: never more pops than the free part of the stack can take the children of
        const int room = (TEAM_STACK - sp) / 3;
        const int want = share < room ? share : (room < 1 ? 1 : room);
        const int npop = sp < want ? sp : want;
        const bool has = e < npop;
        const int node = has ? stk[sp - 1 - e] : 0;
        bool hit = false;
        float tn = 0;
        int ref = LANE_DONE;
        if (has) {
            const float4 ri = rr[2], rn = rr[3], rk = rr[4];
            const uint32_t kd2 = fbits(rk.y);
            // bound of the box test: the ray's own bound until it has a hit, then the distance of the best one
            float tmax;
            { SlabRay Rt; Rt.inv_dd = rr[1].w; tmax = slab_tmax(Rt, (ANY || kd2 == 0xFFFFFFFFu) ? rr[0].w : bitsf(kd2)); }
            v2f tx, ty, tz;
            if (S.wide_quant) {
                const v4f* nd = (const v4f*)S.nodes4 + 4 * (size_t)node;
                const v4f q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
                const uint32_t eb = fbits(q0.w);
                const float ax = bitsf((eb & 0xFFu) << 23) * ri.x, ay = bitsf(((eb >> 8) & 0xFFu) << 23) * ri.y,
                            az = bitsf(((eb >> 16) & 0xFFu) << 23) * ri.z;
                const float bx = __builtin_fmaf(q0.x, ri.x, rn.x), by = __builtin_fmaf(q0.y, ri.y, rn.y),
                            bz = __builtin_fmaf(q0.z, ri.z, rn.z);
                const int sh = 8 * c;
                tx = (v2f){__builtin_fmaf((float)((fbits(q1.x) >> sh) & 0xFFu), ax, bx), __builtin_fmaf((float)((fbits(q1.w) >> sh) & 0xFFu), ax, bx)};
                ty = (v2f){__builtin_fmaf((float)((fbits(q1.y) >> sh) & 0xFFu), ay, by), __builtin_fmaf((float)((fbits(q2.x) >> sh) & 0xFFu), ay, by)};
                tz = (v2f){__builtin_fmaf((float)((fbits(q1.z) >> sh) & 0xFFu), az, bz), __builtin_fmaf((float)((fbits(q2.y) >> sh) & 0xFFu), az, bz)};
                ref = (int)fbits(c == 0 ? q2.z : c == 1 ? q2.w : c == 2 ? q3.x : q3.y);
                if (c == 0) ts.nodes += 1;
            } else {
                const float* nf = (const float*)S.nodes4 + 32 * (size_t)node;
                const float2* bx = (const float2*)(nf + 6 * c);
                const float2 x = bx[0], y = bx[1], z = bx[2];
                ref = __float_as_int(nf[24 + c]);
                tx = (v2f){__builtin_fmaf(x.x, ri.x, rn.x), __builtin_fmaf(x.y, ri.x, rn.x)};
                ty = (v2f){__builtin_fmaf(y.x, ri.y, rn.y), __builtin_fmaf(y.y, ri.y, rn.y)};
                tz = (v2f){__builtin_fmaf(z.x, ri.z, rn.z), __builtin_fmaf(z.y, ri.z, rn.z)};
                if (c == 0) ts.nodes += 2;
            }
            if (c == 0) ts.lane_nodes += 2;
            hit = slab_hit<8>(tx, ty, tz, tmax, tn) && ref != LANE_DONE;
        }
        // children that are leaves: their exact tests -- several times the arithmetic of the rest of a round, wanted by
        // a handful of lanes per round -- wait in a queue of (ray, leaf) pairs until TEAM_FLUSH of them can run together
        const bool at_leaf = hit && ref < 0;
        {
            const unsigned long long lm = ballot64(at_leaf);
            if (at_leaf) queue[qn + __popcll(lm & ((1ull << lane) - 1ull))] = ((uint32_t)ray << 28) | (uint32_t)(~ref);
            qn += __popcll(lm);
        }
        // children that are inner nodes go on their ray's stack
        const bool push = hit && ref >= 0;
        const unsigned long long pm = ballot64(push);
        const int width = 4 * share;                                      // lanes of this ray: [width * k, + width)
        const unsigned long long wmask = width == 64 ? ~0ull : ((1ull << (width & 63)) - 1ull);
        const int base = (width * k) & 63;
        {
            const unsigned long long tb = (pm >> base) & wmask;
            const uint32_t quad = (uint32_t)(tb >> (4 * e)) & 0xFu;
            const int above = __popcll((tb >> (4 * e)) >> 4);                  // pushed by entries deeper in the stack: they go first
            const float t1 = dpp_f<DPP_QUAD_XOR1>(tn), t2 = dpp_f<DPP_QUAD_XOR2>(tn), t3 = dpp_f<DPP_QUAD_XOR3>(tn);
            const int c1 = c ^ 1, c2 = c ^ 2, c3 = c ^ 3;
            int farther = 0;                                                   // of my own node's children: farther ones first
            farther += ((quad >> c1) & 1u) && (t1 > tn || (t1 == tn && c1 > c));
            farther += ((quad >> c2) & 1u) && (t2 > tn || (t2 == tn && c2 > c));
            farther += ((quad >> c3) & 1u) && (t3 > tn || (t3 == tn && c3 > c));
            const int pos = sp - npop + above + farther;
            bool over = false;
            if (push) {
                if (pos < TEAM_STACK) stk[pos] = ref;
                else over = true;
            }
            const unsigned long long ov = ballot64(over);
            const bool ray_over = ((ov >> base) & wmask) != 0ull;
            int nsp = sp - npop + __popcll(tb);
            if (ray_over) nsp = 0;
            const bool first = mine && e == 0 && c == 0;                       // the ray's first lane keeps its books
            if (first) {
                float* book = (float*)(rec + ray * TEAM_REC_F4 + 5);     // (integers kept as float bit patterns: the record is read as float4)
                book[0] = __int_as_float(nsp);
                if (ray_over) book[2] = __int_as_float(1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        alive = (uint32_t)ballot64(lane < (uint32_t)NR && __float_as_int(rec[(lane < (uint32_t)NR ? lane : 0u) * TEAM_REC_F4 + 5].x) > 0);
        // the queued exact tests, a lane each: when enough wait, and before the wave runs out of nodes to walk
        if (qn >= TEAM_FLUSH || (alive == 0u && qn > 0)) {
            while (qn > 0) {
                const int m = qn < 64 ? qn : 64;
                qn -= m;
                unsigned long long key = ~0ull;
                float ca = 0, cb = 0;
                bool found = false;
                int fray = 0;
                if ((int)lane < m) {
                    const uint32_t en = queue[qn + lane];
                    fray = (int)(en >> 28);
                    const float4* rq = rec + fray * TEAM_REC_F4;
                    if (!(ANY && __float_as_int(rq[5].y) != 0)) {          // (a shadow ray that has its answer needs no more tests)
                        const float4* lf = S.leaf + 3 * (size_t)(en & 0x0FFFFFFFu);
                        const float4 l0 = lf[0], l1 = lf[1], l2 = lf[2];
                        const float4 ro = rq[0], rd = rq[1];
                        const double dd = __hiloint2double(__float_as_int(rq[3].w), __float_as_int(rq[2].w));
                        const int prim = (int)fbits(l2.y);
                        if (fbits(l2.z) != 0u) ts.sph++; else ts.tri++;
                        ts.lane_prims++;
                        float d2 = 0;
                        HitUV cand;
                        if (leaf_test(f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), dd, l0, l1, l2, ro.w, d2, cand)) {
                            found = true;
                            key = ((unsigned long long)fbits(d2) << 32) | (uint32_t)prim;
                            ca = cand.a; cb = cand.b;
                        }
                    }
                }
                const unsigned long long fm = ballot64(found);
                if (fm != 0ull) {
                    if (ANY) {
                        // shadowed: nothing more to find out, the ray's stack is dropped
                        if (found) { float* book = (float*)(rec + fray * TEAM_REC_F4 + 5); book[0] = __int_as_float(0); book[1] = __int_as_float(1); }
                    } else {
                        // candidates meet in the ray's record: smallest (distance, primitive) wins, and the winner leaves its u, v
                        if (found) atomicMin((unsigned long long*)(rec + fray * TEAM_REC_F4 + 4), key);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        if (found && *(volatile unsigned long long*)(rec + fray * TEAM_REC_F4 + 4) == key) {
                            float* w = (float*)(rec + fray * TEAM_REC_F4 + 4);
                            w[2] = ca; w[3] = cb;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            if (ANY) alive = (uint32_t)ballot64(lane < (uint32_t)NR && __float_as_int(rec[(lane < (uint32_t)NR ? lane : 0u) * TEAM_REC_F4 + 5].x) > 0);
        }
    }
    if (S.profile && lane == 0) { ts.walk_cycles += __builtin_readcyclecounter() - t0; ts.steps += rounds; ts.walks++; }
    best = -1; best_d2 = dmax; uv.a = 0; uv.b = 0;
    bool redo = false;
    if (lane < (uint32_t)NR && active) {
        const float4 rk = rec[lane * TEAM_REC_F4 + 4], bk = rec[lane * TEAM_REC_F4 + 5];
        redo = __float_as_int(bk.z) != 0;
        if (ANY) { if (__float_as_int(bk.y) != 0) { best = 0; best_d2 = 0; } }
        else if (fbits(rk.y) != 0xFFFFFFFFu || fbits(rk.x) != 0xFFFFFFFFu) { best = (int)fbits(rk.x); best_d2 = rk.y; uv.a = rk.z; uv.b = rk.w; }
    }
    if (ballot64(redo) != 0ull) {
        // a stack outgrew its slice (rare): that ray again, stack in private memory
        int b; float bd; HitUV buv;
        lbvh_trace<ANY, true>(S, redo, o, d, dmax, b, bd, buv, ts);
        if (redo) { best = b; best_d2 = bd; uv = buv; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                     // the next call rewrites the records
}

// Work counters of a traversal launch.  Every wave adds its totals to a slot of its own
// (plain loads and stores: the slot belongs to this wave id for the launch, and launches
// of a batch follow one another), and one small kernel sums the slots at the end of the
// batch.  Adding them to shared counters with atomics instead costs a fifth of the frame:
// thousands of waves finish together and their atomics queue up on one cache line.
__device__ __forceinline__ void flush_stats(const DPool& P, DCounters* C, int which, const TraceStats& ts) {
    unsigned long long v[5] = {ts.nodes, ts.sph, ts.tri, ts.lane_nodes, ts.lane_prims};
    for (int k = 0; k < 5; k++)
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    if ((threadIdx.x & 63) == 0) {
        const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        if (wave < RT_MAX_WAVES) {
            unsigned long long* slot = P.wstats + ((size_t)which * RT_MAX_WAVES + wave) * 8;
            for (int k = 0; k < 5; k++)
                if (v[k]) slot[k] += v[k];
        }
        if (ts.walks) {   // tuning aid (RT_PROFILE) only
            atomicAdd(&C->prof_walk_cycles[which], ts.walk_cycles); atomicAdd(&C->prof_steps[which], ts.steps);
            atomicAdd(&C->prof_walks[which], ts.walks);
        }
    }
}

__global__ void k_reduce_stats(DPool P, DCounters* C) {   // one block per counter: blockIdx.x = which * 5 + k
    __shared__ unsigned long long red[256];
    const int which = blockIdx.x / 5, k = blockIdx.x % 5;
    unsigned long long sum = 0;
    for (uint32_t w = threadIdx.x; w < RT_MAX_WAVES; w += blockDim.x)
        sum += P.wstats[((size_t)which * RT_MAX_WAVES + w) * 8 + k];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        unsigned long long* dst = k == 0 ? C->nodes : k == 1 ? C->sph_tests : k == 2 ? C->tri_tests
                                : k == 3 ? C->lane_nodes : C->lane_prims;
        dst[which] = red[0];
    }
}

// ---------------------------------------------------------------- generate
// Pixel of local pixel index lp: tiles of RT_TILE_PX^2 pixels, 8x8 sub-blocks inside
// a tile so that the 64 lanes of a wave cover a compact screen region.
__device__ __forceinline__ void pixel_of(const uint32_t* __restrict__ tiles, uint32_t tile0, uint32_t lp,
                                int& x, int& y, int& px, int& py, uint32_t& tile) {
    tile = tile0 + (lp >> 10);
    uint32_t pin = lp & 1023u;
    uint32_t blk = pin >> 6, w = pin & 63u;
    px = (int)((blk & 3u) * 8u + (w & 7u));
    py = (int)((blk >> 2) * 8u + (w >> 3));
    uint32_t t = tiles[tile];
    x = (int)(t & 0xFFFFu) * RT_TILE_PX + px;
    y = (int)(t >> 16) * RT_TILE_PX + py;
}

// RayTrace::getRay (src/raytrace.cpp:441-485) for one sample.
__device__ __forceinline__ void primary_ray(const DCamera& cam, int x, int y, int j, F3& o, F3& d) {
    double r1, r2;
    if (cam.spp > 1) {
        uint32_t j1, j2;
        if (cam.counter32) {
            // the counter of every sample of this film fits 32 bits: its high word is zero and that word's
            // hash a frame constant (rt_jitter with the same result, four quarter-rate multiplies fewer)
            const uint32_t pix = (uint32_t)y * (uint32_t)cam.width + (uint32_t)x;
            const uint32_t c32 = ((cam.spp_shift >= 0 ? pix << cam.spp_shift : pix * (uint32_t)cam.spp) + (uint32_t)j) << 1;
            j1 = rt_lowbias32(c32 ^ cam.jitter_hi0) >> 1;
            j2 = rt_lowbias32((c32 + 1u) ^ cam.jitter_hi0) >> 1;
        } else {
            unsigned long long c = (((unsigned long long)y * (unsigned long long)cam.width +
                                     (unsigned long long)x) * (unsigned long long)cam.spp +
                                    (unsigned long long)j) * 2ull;
            j1 = rt_jitter(cam.seed, c);
            j2 = rt_jitter(cam.seed, c + 1);
        }
        r1 = rt_cdiv((double)(int)j1, 2147483647.0, 1.0 / 2147483647.0);
        r2 = rt_cdiv((double)(int)j2, 2147483647.0, 1.0 / 2147483647.0);
    } else {
        r1 = r2 = .5;
    }
    const double w = (double)cam.width, h = (double)cam.height;
    F3 ur = cam.urv * (float)(rt_cdiv((double)x, w, cam.inv_w) + rt_cdiv(r1, w, cam.inv_w));
    F3 bl = cam.blv * (float)(rt_cdiv((double)y, h, cam.inv_h) + rt_cdiv(r2, h, cam.inv_h));
    F3 p = (ur + bl) + cam.p1;
    if (cam.proj == 1) { o = cam.pos; d = norm3(p); }
    else { o = p; d = cam.dir; }
}

// Ray record of slot i of generation g.  Generation 0 is never stored: its rays
// are recomputed from the sample index wherever they are needed (the extend and
// shade kernels), which saves writing and re-reading 32 bytes per sample.
struct RayRec {
    F3 o, d;
    uint32_t parent;   // gen 0: sample index; else index of the parent ray
    uint32_t slot;     // 0 reflection child, 1 refraction child
    bool valid;        // false for generation-0 slots of pixels outside the film
};
struct GenSrc {        // what generation 0 is generated from
    DCamera cam;
    const uint32_t* tiles;
    uint32_t tile0;
    int defer_miss;    // 1: misses of generations >= 1 are delivered later by k_miss (overlapped pipeline)
};
__device__ __forceinline__ RayRec load_ray(const DPool& P, const GenSrc& G, int g, uint32_t i, uint32_t r) {
    RayRec R;
    if (g == 0) {
        uint32_t lp; int j;
        if (G.cam.spp_shift >= 0) { lp = i >> G.cam.spp_shift; j = (int)(i & ((1u << G.cam.spp_shift) - 1u)); }
        else { lp = i / (uint32_t)G.cam.spp; j = (int)(i - lp * (uint32_t)G.cam.spp); }
        int x, y, px, py; uint32_t tile;
        pixel_of(G.tiles, G.tile0, lp, x, y, px, py, tile);
        R.valid = x < G.cam.width && y < G.cam.height;
        R.o = f3(0, 0, 0); R.d = f3(0, 0, 1);
        if (R.valid) primary_ray(G.cam, x, y, j, R.o, R.d);
        R.parent = i; R.slot = 0;
    } else {
        float4 qo = P.o[r], qd = P.d[r];
        R.o = f3(qo); R.d = f3(qd);
        R.parent = fbits(qo.w); R.slot = fbits(qd.w);
        R.valid = true;
    }
    return R;
}

__global__ void k_primary_rays(DCamera cam, int64_t n, const int* x, const int* y, const int* j,
                               float* o_out, float* d_out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 o, d;
    primary_ray(cam, x[i], y[i], j[i], o, d);
    o_out[3 * i] = o.x; o_out[3 * i + 1] = o.y; o_out[3 * i + 2] = o.z;
    d_out[3 * i] = d.x; d_out[3 * i + 1] = d.y; d_out[3 * i + 2] = d.z;
}

__global__ void k_init_counters(DCounters* C, uint32_t nsamples) {
    int i = threadIdx.x;
    if (i <= RT_MAX_GEN) { C->gen_count[i] = (i == 0) ? nsamples : 0; C->hit_count[i] = 0; C->gbase[i] = 0; }
    if (i == 0) { C->overflow = 0; C->pad = 0; }
    if (i < 2) {
        C->nodes[i] = 0; C->sph_tests[i] = 0; C->tri_tests[i] = 0; C->lane_nodes[i] = 0; C->lane_prims[i] = 0;
        C->prof_walk_cycles[i] = 0; C->prof_steps[i] = 0; C->prof_wave_cycles[i] = 0; C->prof_walks[i] = 0;
    }
}

// ---------------------------------------------------------------- spawn decision
// Which secondary rays a hit sends out (src/raytrace.cpp:589, :600-615): reflection
// iff every specular component is non-zero, refraction iff every transmissive
// component is non-zero and there is no total internal reflection.  Used by the
// extend kernel (to size the next generation) and by the shade kernel (to build
// the rays); same inputs, same arithmetic, same answer.
struct Spawn {
    bool refl, refr;
    double dni, ior, tir;
};
__device__ __forceinline__ Spawn spawn_of(F3 spec, F3 trans, float iorf, F3 direct, F3 normal) {
    Spawn s;
    s.refl = (spec.x != 0 && spec.y != 0 && spec.z != 0);
    s.refr = false;
    s.dni = 0; s.ior = 0; s.tir = 0;
    if (trans.x != 0 && trans.y != 0 && trans.z != 0) {
        F3 irdir = direct * -1.0f;
        s.dni = (double)dot3(irdir, normal);
        s.ior = (s.dni <= 0) ? (double)iorf : 1.0 / (double)iorf;
        s.tir = 1.0 - (s.ior * s.ior) * (1.0 - (s.dni * s.dni));
        s.refr = s.tir >= 0;
    }
    return s;
}

// ---------------------------------------------------------------- extend (closest hit)
// evaluateRayTree's intersect + miss branch (src/raytrace.cpp:139-150): a miss
// returns the background to the parent at once; a hit leaves its record (point,
// normal) at the ray's slot.  Queue compaction is wave64 ballots + a prefix sum:
// each wave publishes three 64-bit masks for its 64 rays -- hit, spawns a
// reflection ray, spawns a refraction ray -- and the scan kernels below turn the
// popcounts into the compacted hit list and the write offsets of the next
// generation.  No atomics: a single hot counter sustains only ~90 returning
// atomics per microsecond on this chip, which one atomic per wave would exceed
// many times over; the scan also keeps hits and children in ray order, which
// preserves the coherence of neighbouring rays.
// Every lane of the wave must call (ballots inside).
template <int ACCEL>
__device__ __forceinline__ void extend_one(const DScene& S, const DPool& P, const GenSrc& G, int g, uint32_t start,
                                  uint32_t count, uint32_t gbase, uint32_t i, float4* lds, TraceStats& ts) {
    bool inrange = i < count;
    uint32_t r = start + (inrange ? i : 0);
    RayRec R = load_ray(P, G, g, inrange ? i : 0, r);
    bool active = inrange && R.valid;
    F3 o = R.o, d = R.d;
    int best; float best_d2; HitUV uv;
    trace<ACCEL, false>(S, lds, active, o, d, __builtin_inff(), best, best_d2, uv, ts);
    bool hit = active && best >= 0;
    bool refl = false, refr = false;
    // a miss returns the background to the parent: its record is asked for now, so that the fetch
    // travels while the hit lanes of the wave build their records
    const bool miss = active && !hit;
    float4 ploc = make_float4(0, 0, 0, 0);
    const bool deliver_now = miss && !(G.defer_miss && g > 0);
    if (deliver_now && g > 0) ploc = P.loc[R.parent];
    if (hit) {
        F3 point, normal; int mat;
        make_hit_uv(S, o, d, best, uv, point, normal, mat);
        P.hp[i] = make_float4(point.x, point.y, point.z, bitsf((uint32_t)best));
        P.hn[i] = make_float4(normal.x, normal.y, normal.z, bitsf((uint32_t)mat | (R.slot << 31)));
        P.hd[i] = make_float4(d.x, d.y, d.z, bitsf(R.parent));
        if (g + 1 <= S.max_depth) {   // deeper children are never traced (src/raytrace.cpp:132-135)
            const float4* M = S.mats + 4 * (size_t)mat;
            Spawn sp = spawn_of(f3(M[2]), f3(M[3]), M[1].w, d, normal);
            refl = sp.refl; refr = sp.refr;
        }
    }
    if (deliver_now) deliver_from(S, P, g, R.parent, R.slot, S.bg, ploc);
    if (ACCEL == 2 || ACCEL == 4) {
        // launches of per-lane walks leave a flag byte per ray (their chunk walks finish rays out of order, see
        // extend_chunks); the scan builds the masks
        if (inrange) P.rflag[i] = (uint8_t)((hit ? 1u : 0u) | (refl ? 2u : 0u) | (refr ? 4u : 0u));
        return;
    }
    unsigned long long mh = __ballot(hit), m1 = __ballot(refl), m2 = __ballot(refr);
    if ((threadIdx.x & 63) == 0 && inrange) {
        uint32_t grp = gbase + (i >> 6);
        P.mask_hit[grp] = mh; P.mask_refl[grp] = m1; P.mask_refr[grp] = m2;
    }
}

// extend_one for a launch whose walks are team walks: the wave owns the NR consecutive rays of `item`, lane r
// the r-th of them.  The three ballot masks of a 64-ray group cannot be put together by one wave here (a group
// is spread over 64 / NR waves): each ray leaves a flag byte (P.rflag: 1 hit, 2 spawns a reflection ray,
// 4 a refraction ray) and the scan kernel builds the masks from the bytes.
template <int NR>
__device__ __forceinline__ void extend_team(const DScene& S, const DPool& P, const GenSrc& G, int g, uint32_t start,
                                            uint32_t count, uint64_t item, float4* wave_lds, TraceStats& ts) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t i64 = item * (uint64_t)NR + lane;
    const bool inrange = lane < (uint32_t)NR && i64 < (uint64_t)count;
    const uint32_t i = inrange ? (uint32_t)i64 : 0u;
    RayRec R;
    R.o = f3(0, 0, 0); R.d = f3(0, 0, 1); R.parent = 0; R.slot = 0; R.valid = false;
    if (inrange) R = load_ray(P, G, g, i, start + i);
    const bool active = inrange && R.valid;
    int best; float best_d2; HitUV uv;
    team_walk<false, NR>(S, wave_lds, active, R.o, R.d, __builtin_inff(), best, best_d2, uv, ts);
    if (!inrange) return;
    uint32_t bits = 0;
    if (active && best >= 0) {
        F3 point, normal; int mat;
        make_hit_uv(S, R.o, R.d, best, uv, point, normal, mat);
        P.hp[i] = make_float4(point.x, point.y, point.z, bitsf((uint32_t)best));
        P.hn[i] = make_float4(normal.x, normal.y, normal.z, bitsf((uint32_t)mat | (R.slot << 31)));
        P.hd[i] = make_float4(R.d.x, R.d.y, R.d.z, bitsf(R.parent));
        bits = 1;
        if (g + 1 <= S.max_depth) {   // deeper children are never traced (src/raytrace.cpp:132-135)
            const float4* M = S.mats + 4 * (size_t)mat;
            Spawn sp = spawn_of(f3(M[2]), f3(M[3]), M[1].w, R.d, normal);
            bits |= (sp.refl ? 2u : 0u) | (sp.refr ? 4u : 0u);
        }
    } else if (active && !(G.defer_miss && g > 0)) {
        deliver(S, P, g, R.parent, R.slot, S.bg);
    }
    P.rflag[i] = (uint8_t)bits;
}

#ifndef REFILL_GROUPS
#define REFILL_GROUPS 2
#endif
#ifndef REFILL_MIN
#define REFILL_MIN 16
#endif
#ifndef LEAF_MIN
#define LEAF_MIN 20
#endif


// Closest-hit walk with lane refill (see shadow_chunks for the idea): a wave draws chunks of `span` consecutive
// rays of the generation (span a multiple of 64) one after the other.  A lane whose walk has ended waits until
// REFILL_MIN lanes have, then those lanes write their hit (or miss) records together and take the next rays of the
// chunk -- or of the wave's next chunk when this one has none left, so the longest rays of a chunk never walk alone.
// Rays are finished out of order and a 64-ray group may be split between two refills: every ray leaves a flag
// byte (P.rflag, as the team walks do) and the scan builds the three ballot masks of each group from the bytes.
template <bool WIDE>
__device__ __forceinline__ void extend_chunks(const DScene& S, const DPool& P, const GenSrc& G, int g, uint32_t start,
                                     uint32_t count, GroupSource& src, uint32_t span, int* lds_stack, TraceStats& ts) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    enum { IDLE = 0, WALK = 1, FIN = 2 };
    uint32_t first = 0, n = 0, next = 0;     // wave-uniform: the chunk being handed out, and how far
    bool more = true;                        // wave-uniform: the launch may still have chunks for this wave
    uint32_t i = 0;                          // this lane's ray (index inside the generation)
    int state = IDLE;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 1);
    uint32_t parent = 0, slot = 0;
    bool valid = false;
    float tmax = 0, best_d2 = 0;
    double dd = 1;
    SlabRay R = slab_ray(o, d);
    int cur = LANE_DONE, best = -1;
    HitUV uv; uv.a = 0; uv.b = 0;
    LdsStack stk;
    stk.init(lds_stack);
    auto begin = [&](uint32_t ray) {
        i = ray;
        RayRec rr = load_ray(P, G, g, ray, start + ray);
        o = rr.o; d = rr.d; parent = rr.parent; slot = rr.slot; valid = rr.valid;
        best = -1; best_d2 = __builtin_inff(); uv.a = 0; uv.b = 0;
        dd = (double)dot3(d, d);
        R = slab_ray(o, d);
        tmax = __builtin_inff();
        cur = 0; stk.reset();                       // both trees keep their root at node 0
        state = valid ? WALK : FIN;
    };
    while (true) {
        if (ballot64(state != IDLE) == 0ull && !more) break;
        const bool at_node = state == WALK && cur >= 0;
        if (at_node) {
            cur = lane_inner_step<WIDE>(S, cur, R, tmax, stk, ts);
            if (cur == LANE_DONE) state = FIN;
        }
        // lanes at a leaf wait until LEAF_MIN of them are (see shadow_chunks)
        const bool at_leaf = state == WALK && !at_node;
        const unsigned long long leafm = ballot64(at_leaf);
        if (leafm != 0ull && (__popcll(leafm) >= LEAF_MIN || ballot64(at_node) == 0ull)) {
            if (at_leaf) {
                const float4* lf = S.leaf + 3 * (size_t)(~cur);
                float4 l0 = lf[0], l1 = lf[1], l2 = lf[2];
                int prim = (int)fbits(l2.y);
                if (fbits(l2.z) != 0u) ts.sph++; else ts.tri++;
                ts.lane_prims++;
                float d2 = 0;
                HitUV cand;
                if (leaf_test(o, d, dd, l0, l1, l2, __builtin_inff(), d2, cand)) {
                    if (best < 0 || d2 < best_d2 || (d2 == best_d2 && prim < best)) {
                        best = prim; best_d2 = d2; uv = cand;
                        tmax = slab_tmax(R, d2);
                    }
                }
                if (!stk.empty()) cur = stk.pop();
                else state = FIN;
            }
        }
        if (state == WALK && stk.over) {
            // deeper than the LDS stack holds (rare): this one ray again, on a stack in private memory
            lbvh_trace<false, WIDE>(S, true, o, d, __builtin_inff(), best, best_d2, uv, ts);
            stk.reset();
            state = FIN;
        }
        const unsigned long long fin = ballot64(state == FIN);
        if (__popcll(fin) >= REFILL_MIN || ballot64(state == WALK) == 0ull) {
            if (state == FIN) {
                uint32_t bits = 0;
                if (valid && best >= 0) {
                    F3 point, normal; int mat;
                    make_hit_uv(S, o, d, best, uv, point, normal, mat);
                    P.hp[i] = make_float4(point.x, point.y, point.z, bitsf((uint32_t)best));
                    P.hn[i] = make_float4(normal.x, normal.y, normal.z, bitsf((uint32_t)mat | (slot << 31)));
                    P.hd[i] = make_float4(d.x, d.y, d.z, bitsf(parent));
                    bits = 1;
                    if (g + 1 <= S.max_depth) {
                        const float4* M = S.mats + 4 * (size_t)mat;
                        Spawn spn = spawn_of(f3(M[2]), f3(M[3]), M[1].w, d, normal);
                        bits |= (spn.refl ? 2u : 0u) | (spn.refr ? 4u : 0u);
                    }
                } else {
                    if (valid && !(G.defer_miss && g > 0)) deliver(S, P, g, parent, slot, S.bg);
                }
                P.rflag[i] = (uint8_t)bits;
                state = IDLE;
            }
            // idle lanes take the chunk's next rays; what the chunk cannot serve comes from the wave's next chunk
            for (int round = 0; round < 2 && more; round++) {
                if (next >= n) {
                    uint64_t c;
                    if (!src.next(c)) { more = false; break; }
                    first = (uint32_t)c * span;
                    n = min(span, count - first);
                    next = 0;
                }
                const unsigned long long idle = ballot64(state == IDLE);
                if (idle == 0ull) break;
                if (state == IDLE) {
                    const uint32_t rel = next + (uint32_t)__popcll(idle & below);
                    if (rel < n) begin(first + rel);
                }
                next += (uint32_t)__popcll(idle);
                if (next <= n) break;            // every idle lane was served
            }
        }
    }
}

template <int ACCEL, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) RT_WAVES_ATTR void k_extend(DScene S, DPool P, DCounters* C, GenSrc G, int g) {
    __shared__ float4 lds[ACCEL == 1 ? BRUTE_CHUNK : (ACCEL == 2 || ACCEL == 4) ? LANE_STACK_F4 : ACCEL == 5 ? TEAM_LDS_F4 * (RT_BLOCK / 64) : 1];
    if (C->overflow) return;   // ray slots past an overflow were never written: the host re-runs the batch
    const uint32_t start = gen_start(C, g);
    const uint32_t count = gen_size(C, g, start, P.capacity);
    const uint32_t gbase = C->gbase[g];
    TraceStats ts;
    unsigned long long tw = S.profile ? __builtin_readcyclecounter() : 0ull;
    if (ACCEL == 1) {   // block-wide LDS staging: the whole block moves together
        for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += gridDim.x * blockDim.x)
            extend_one<ACCEL>(S, P, G, g, start, count, gbase, base + threadIdx.x, lds, ts);
    } else if (ACCEL == 2 || ACCEL == 4) {
        const uint64_t groups = ((uint64_t)count + 63) >> 6;
        const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
        uint64_t per = groups / (waves * 4);
        per = per < 1 ? 1 : (per > REFILL_GROUPS ? REFILL_GROUPS : per);
        const uint32_t span = (uint32_t)per << 6;
        if (groups <= waves) {      // a group per wave at most, nothing to refill from: plain 64-ray groups
            for_each_group(P.wq + (size_t)(2 * g) * RT_WQ_LAUNCH, groups, [&](uint64_t grp) {
                extend_one<ACCEL>(S, P, G, g, start, count, gbase, (uint32_t)(grp << 6) + (threadIdx.x & 63u), lds, ts);
            });
        } else {
            GroupSource src;
            src.init(P.wq + (size_t)(2 * g) * RT_WQ_LAUNCH, ((uint64_t)count + span - 1) / span);
            extend_chunks<ACCEL == 4>(S, P, G, g, start, count, src, span, (int*)lds, ts);
        }
    } else if (ACCEL == 5) {
        for_each_group(P.wq + (size_t)(2 * g) * RT_WQ_LAUNCH, ((uint64_t)count + TEAM_NR - 1) / TEAM_NR, [&](uint64_t item) {
            extend_team<TEAM_NR>(S, P, G, g, start, count, item, lds + (threadIdx.x >> 6) * TEAM_LDS_F4, ts);
        });
    } else {
        for_each_group(P.wq + (size_t)(2 * g) * RT_WQ_LAUNCH, ((uint64_t)count + 63) >> 6, [&](uint64_t grp) {
            extend_one<ACCEL>(S, P, G, g, start, count, gbase, (uint32_t)(grp << 6) + (threadIdx.x & 63u), lds, ts);
        });
    }
    if (S.profile && (threadIdx.x & 63) == 0) atomicAdd(&C->prof_wave_cycles[0], __builtin_readcyclecounter() - tw);
    if (COUNT) flush_stats(P, C, 0, ts);   // never flushed = never counted: the compiler drops the counting above (profiling launches the counting variant)
}

// ---------------------------------------------------------------- queue compaction (prefix sums)
// One thread per 64-ray group, ONE launch per generation: a single-pass scan with decoupled look-back
// (Merrill & Garland 2016).  A block takes the next 256 groups by ticket (so every block with a smaller
// number is already running and nobody waits for a block that has not started), scans its (hits, children)
// counts in LDS, publishes its total, and its first wave then reads the totals of the blocks before it --
// 64 at a time -- back to the nearest one that has published its inclusive prefix.  With the prefix the block
// writes the child offsets of its groups and its part of the compacted hit list; the last block publishes
// the generation totals.  No atomics on the ray data, hits and children stay in ray order.
// Everything one block tells another travels INSIDE the state word (flag + both sums in 64 bits), so the
// words are read and written with relaxed device-scope atomics and no fence: a release / acquire pair at
// device scope would write back and invalidate the whole L2 of the XCD on every publication.
// (Three launches -- block sums, scan of the sums, offsets -- took 45 us per generation on a thin frame, most
// of it between the launches; this is one.)
#define SCAN_BLOCK 256
#define SCAN_AGG 1ull        // state word = flag << 62 | hits << 31 | children (children saturate at 2^31 - 1:
#define SCAN_PREFIX 2ull     // more than any pool holds, so a saturated sum always raises the overflow flag)
#define SCAN_KIDS_MAX 0x7FFFFFFFull

__device__ __forceinline__ unsigned long long scan_pack(unsigned long long flag, uint32_t hits, unsigned long long kids) {
    return (flag << 62) | ((unsigned long long)hits << 31) | (kids > SCAN_KIDS_MAX ? SCAN_KIDS_MAX : kids);
}

// block-wide exclusive scan of a uint2 (both components), returns the block total
__device__ __forceinline__ uint2 block_scan(uint2 v, uint2* lds /*[SCAN_BLOCK]*/, uint2& total) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        uint2 a = lds[t];
        uint2 b = (t >= off) ? lds[t - off] : make_uint2(0, 0);
        __syncthreads();
        lds[t] = make_uint2(a.x + b.x, a.y + b.y);
        __syncthreads();
    }
    uint2 incl = lds[t];
    total = lds[SCAN_BLOCK - 1];
    __syncthreads();
    return make_uint2(incl.x - v.x, incl.y - v.y);
}

// A block's tile is 256 x items groups, items = 1 .. 8 chosen per generation: thin generations get many small
// tiles (parallelism; their launch is over when the last tile is), fat ones few large tiles (the look-back of
// thousands of tiles that run at once would walk through thousands of unfinished neighbours).
#define SCAN_ITEMS 8                                  // most groups per thread
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)           // most groups per block and state word: 2048 = 131072 rays
#define SCAN_TILES_WANTED 1024u

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan(DPool P, DCounters* C, int g, int from_flags) {
    __shared__ uint2 lds[SCAN_BLOCK];
    __shared__ uint32_t hb[SCAN_TILE];
    __shared__ unsigned long long hm[SCAN_TILE];
    __shared__ uint32_t s_bid;
    __shared__ unsigned long long s_prefix[2];        // hits, children before this block
    if (C->overflow) return;
    const uint32_t start = gen_start(C, g);
    const uint32_t count = gen_size(C, g, start, P.capacity);
    const uint32_t ngroups = (count + 63) >> 6;
    uint32_t items = (ngroups + SCAN_BLOCK * SCAN_TILES_WANTED - 1) / (SCAN_BLOCK * SCAN_TILES_WANTED);
    items = items < 1u ? 1u : (items > (uint32_t)SCAN_ITEMS ? (uint32_t)SCAN_ITEMS : items);
    const uint32_t tile = SCAN_BLOCK * items;
    const uint32_t nblocks = ngroups ? (ngroups + tile - 1) / tile : 1;   // an empty generation still publishes its totals
    // the grid is sized for the most rays the generation could hold: blocks beyond the tiles there are leave at once
    // (a ticket each would queue thousands of atomics on one counter); the others share the tiles by ticket
    if (blockIdx.x >= nblocks) return;
    const uint32_t gbase = C->gbase[g];
    // this generation's state words: generations never share one (gbase counts the groups before it)
    unsigned long long* state = P.scan_state + (gbase >> 8) + (uint32_t)g;
    const uint32_t lane = threadIdx.x & 63u;
    while (true) {
        if (threadIdx.x == 0) s_bid = atomicAdd(&P.scan_ticket[g], 1u);
        __syncthreads();
        const uint32_t bid = s_bid;
        if (bid >= nblocks) break;
        const uint32_t first = bid * tile;                                          // first group of the block
        const uint32_t ng = ngroups > first ? min(tile, ngroups - first) : 0u;
        // thread t owns groups first + t * items .. + items - 1; the masks go to LDS through coalesced loads
        for (uint32_t k = threadIdx.x; k < tile; k += SCAN_BLOCK) {
            unsigned long long mh = 0ull;
            uint32_t kids = 0;
            if (k < ng && !from_flags) {
                mh = P.mask_hit[gbase + first + k];
                kids = (uint32_t)(__popcll(P.mask_refl[gbase + first + k]) + __popcll(P.mask_refr[gbase + first + k]));
            }
            if (!from_flags || k >= ng) {                 // (with flags the slots of real groups are written below)
                hm[k] = mh;
                hb[k] = kids;
            }
        }
        if (from_flags) {
            // The extend launch left one flag byte per ray (team walks), not masks: build the three masks of every
            // group here and keep them for the shade kernels.  A lane takes 16 consecutive bytes (one load), four
            // neighbouring lanes make a group: bit b of four bytes at once is (x >> b & 0x01010101) * 0x10204080
            // >> 28, the quarters meet through quad shuffles.  Coalesced: a wave reads 1 KiB = 16 groups per load.
            const uint32_t q = threadIdx.x & 3u;
            for (uint32_t k4 = threadIdx.x; k4 < tile * 4u; k4 += SCAN_BLOCK) {
                const uint32_t k = k4 >> 2;
                const bool in = k < ng;
                uint4 w = make_uint4(0, 0, 0, 0);
                if (in) w = ((const uint4*)(P.rflag + ((size_t)(first + k) << 6)))[q];
                const uint32_t x[4] = {w.x, w.y, w.z, w.w};
                uint32_t p0 = 0, p1 = 0, p2 = 0;                 // 16 bits each: this lane's quarter of the masks
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    p0 |= (((x[t] & 0x01010101u) * 0x10204080u) >> 28) << (4 * t);
                    p1 |= ((((x[t] >> 1) & 0x01010101u) * 0x10204080u) >> 28) << (4 * t);
                    p2 |= ((((x[t] >> 2) & 0x01010101u) * 0x10204080u) >> 28) << (4 * t);
                }
                // halves of the 64-bit masks: quarters 0, 1 -> low word, 2, 3 -> high word
                uint32_t lo0 = q < 2u ? p0 << (16 * q) : 0u, hi0 = q >= 2u ? p0 << (16 * (q - 2u)) : 0u;
                uint32_t lo1 = q < 2u ? p1 << (16 * q) : 0u, hi1 = q >= 2u ? p1 << (16 * (q - 2u)) : 0u;
                uint32_t lo2 = q < 2u ? p2 << (16 * q) : 0u, hi2 = q >= 2u ? p2 << (16 * (q - 2u)) : 0u;
                lo0 |= (uint32_t)dpp_i<DPP_QUAD_XOR1>((int)lo0); hi0 |= (uint32_t)dpp_i<DPP_QUAD_XOR1>((int)hi0);
                lo1 |= (uint32_t)dpp_i<DPP_QUAD_XOR1>((int)lo1); hi1 |= (uint32_t)dpp_i<DPP_QUAD_XOR1>((int)hi1);
                lo2 |= (uint32_t)dpp_i<DPP_QUAD_XOR1>((int)lo2); hi2 |= (uint32_t)dpp_i<DPP_QUAD_XOR1>((int)hi2);
                lo0 |= (uint32_t)dpp_i<DPP_QUAD_XOR2>((int)lo0); hi0 |= (uint32_t)dpp_i<DPP_QUAD_XOR2>((int)hi0);
                lo1 |= (uint32_t)dpp_i<DPP_QUAD_XOR2>((int)lo1); hi1 |= (uint32_t)dpp_i<DPP_QUAD_XOR2>((int)hi1);
                lo2 |= (uint32_t)dpp_i<DPP_QUAD_XOR2>((int)lo2); hi2 |= (uint32_t)dpp_i<DPP_QUAD_XOR2>((int)hi2);
                if (in && q == 0u) {
                    unsigned long long mh = ((unsigned long long)hi0 << 32) | lo0, m1 = ((unsigned long long)hi1 << 32) | lo1,
                                       m2 = ((unsigned long long)hi2 << 32) | lo2;
                    const uint32_t left = count - ((first + k) << 6);      // rays of the group (the last one may be short:
                    if (left < 64u) {                                      //  nobody wrote the bytes past the generation's end)
                        const unsigned long long keep = (1ull << left) - 1ull;
                        mh &= keep; m1 &= keep; m2 &= keep;
                    }
                    P.mask_hit[gbase + first + k] = mh; P.mask_refl[gbase + first + k] = m1; P.mask_refr[gbase + first + k] = m2;
                    hm[k] = mh;
                    hb[k] = (uint32_t)(__popcll(m1) + __popcll(m2));
                }
            }
        }
        __syncthreads();
        uint32_t eh[SCAN_ITEMS], ek[SCAN_ITEMS];
        uint2 v = make_uint2(0, 0);
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS; q++) {
            eh[q] = v.x; ek[q] = v.y;
            if ((uint32_t)q < items) {
                const uint32_t k = threadIdx.x * items + q;
                v.x += (uint32_t)__popcll(hm[k]); v.y += hb[k];
            }
        }
        uint2 total;
        const uint2 ex = block_scan(v, lds, total);
        if (threadIdx.x < 64) {                    // wave 0: publish, look back
            unsigned long long ph = 0ull, pk = 0ull;
            if (bid == 0) {
                if (lane == 0) __hip_atomic_store(&state[0], scan_pack(SCAN_PREFIX, total.x, total.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (lane == 0) __hip_atomic_store(&state[bid], scan_pack(SCAN_AGG, total.x, total.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int top = (int)bid - 1;            // nearest block not yet added
                while (true) {
                    const int j = top - (int)lane;
                    unsigned long long sw = scan_pack(SCAN_PREFIX, 0, 0);          // before block 0: an empty prefix
                    if (j >= 0) {
                        do { sw = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((sw >> 62) == 0ull);
                    }
                    const unsigned long long pfx = ballot64((sw >> 62) == SCAN_PREFIX);
                    const int stop = pfx ? __builtin_ctzll(pfx) : 64;              // lanes 0 .. stop take part (stop = 64: all)
                    unsigned long long h = ((int)lane <= stop) ? ((sw >> 31) & 0x7FFFFFFFull) : 0ull;
                    unsigned long long k = ((int)lane <= stop) ? (sw & SCAN_KIDS_MAX) : 0ull;
                    for (int off = 32; off > 0; off >>= 1) { h += __shfl_down(h, off); k += __shfl_down(k, off); }
                    ph += __shfl(h, 0); pk += __shfl(k, 0);
                    if (pfx) break;
                    top -= 64;
                }
                if (lane == 0) __hip_atomic_store(&state[bid], scan_pack(SCAN_PREFIX, (uint32_t)(ph + total.x), pk + total.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) { s_prefix[0] = ph; s_prefix[1] = pk; }
        }
        __syncthreads();
        const unsigned long long bh = s_prefix[0], bk = s_prefix[1];
        const uint32_t kbase = (uint32_t)(bk > SCAN_KIDS_MAX ? SCAN_KIDS_MAX : bk);
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS; q++) {
            if ((uint32_t)q < items) {
                const uint32_t k = threadIdx.x * items + q;
                if (k < ng) P.child_base[first + k] = kbase + ex.y + ek[q];
                hb[k] = (uint32_t)bh + ex.x + eh[q];   // (every thread rewrites only its own slots: the counts it read from them are in registers)
            }
        }
        __syncthreads();
        // compacted hit list of the block's groups: a wave per group, one thread per ray, coalesced
        for (uint32_t k = threadIdx.x; k < ng * 64u; k += SCAN_BLOCK) {
            const uint32_t gl = k >> 6, bit = k & 63u;
            const unsigned long long m = hm[gl];
            if ((m >> bit) & 1ull)
                P.hitlist[hb[gl] + (uint32_t)__popcll(m & ((1ull << bit) - 1ull))] = ((first + gl) << 6) + bit;
        }
        if (bid == nblocks - 1 && threadIdx.x == 0) {
            const unsigned long long kids = bk + total.y;
            C->hit_count[g] = (uint32_t)bh + total.x;
            C->gen_count[g + 1] = (uint32_t)(kids > SCAN_KIDS_MAX ? SCAN_KIDS_MAX : kids);
            C->gbase[g + 1] = gbase + ngroups;
            if ((unsigned long long)start + C->gen_count[g] + kids > (unsigned long long)P.capacity) C->overflow = 1;
        }
        __syncthreads();                           // hb / hm / s_bid are rewritten by the next round
    }
}

// ---------------------------------------------------------------- per-lane any-hit walk with lane refill
// Incoherent shadow rays differ wildly in length: one hits the first box it enters, its
// neighbour crosses the whole scene.  Walking 64 of them to the end leaves most lanes idle
// most of the time.  Here a wave draws chunks of up to REFILL_GROUPS x 64 consecutive work
// items from the launch's queues; a lane that has its answer stores it and, once REFILL_MIN
// lanes are idle, every idle lane takes the next item of the chunk (ballot + prefix count, no
// atomics) -- and of the wave's NEXT chunk when this one has run out, so the wave stays full
// until the launch runs out of work.  (Walking every chunk to its end first was measured: the
// longest ray of a chunk takes about as long as the rest of it, VALU lane utilisation 0.37
// against 0.47 chained, the 11 M-primitive scene's frame 64.8 -> 54.2 ms.)  One loop iteration
// = one node or one leaf per lane, exactly the step of lbvh_trace<true>.
template <bool WIDE>
__device__ __forceinline__ void shadow_chunks(const DScene& S, const DPool& P, int* lds_stack, GroupSource& src, uint64_t span,
                                      uint64_t items, TraceStats& ts) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t nl = (uint32_t)S.nl;
    uint64_t first = 0;                  // wave-uniform: the chunk being handed out to the lanes, [first, first + count)
    uint32_t count = 0, next = 0;        // ... and the first item of it not handed out yet
    bool more = true;                    // wave-uniform: the launch may still have chunks for this wave
    uint64_t item = 0;                   // this lane's item
    bool alive = false;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 1);
    float dmax = 0, tmax = 0;
    double dd = 1;
    SlabRay R = slab_ray(o, d);
    int cur = LANE_DONE;
    LdsStack stk;
    stk.init(lds_stack);
    auto start = [&](uint64_t it) {
        item = it;
        uint32_t h = (uint32_t)(it / nl);
        int j = (int)(it - (uint64_t)h * nl);
        o = f3(P.hp[P.hitlist[h]]);
        light_dir(S.lights[j], o, d, dmax);
        dd = (double)dot3(d, d);
        R = slab_ray(o, d);
        tmax = slab_tmax(R, dmax);
        cur = 0;                                    // both trees keep their root at node 0
        stk.reset();
        alive = true;
    };
    while (true) {
        // idle lanes take the next items of the chunk; a chunk that has run out is followed by the wave's next one at
        // once, so the lanes still walking the old chunk's longest rays have company (walking every chunk to its end
        // on its own left two thirds of the lane slots empty: a chunk's longest ray takes as long as the rest of it)
        const unsigned long long idle = ~ballot64(alive);
        if (more && __popcll(idle) >= REFILL_MIN) {
            for (int round = 0; round < 2 && more; round++) {      // what the chunk cannot serve comes from the next one
                if (next >= count) {
                    uint64_t c;
                    if (!src.next(c)) { more = false; break; }
                    first = c * span;
                    count = (uint32_t)(first + span < items ? span : items - first);
                    next = 0;
                }
                const unsigned long long free_lanes = ~ballot64(alive);
                if (free_lanes == 0ull) break;
                if (!alive) {
                    const uint32_t rel = next + (uint32_t)__popcll(free_lanes & below);
                    if (rel < count) start(first + rel);
                }
                next += (uint32_t)__popcll(free_lanes);
                if (next <= count) break;            // every idle lane was served
            }
        }
        if (ballot64(alive) == 0ull) {
            if (!more) break;
            continue;
        }
        int verdict = -1;                // 0: reaches the light, 1: occluded
        const bool at_node = alive && cur >= 0;
        if (at_node) {
            cur = lane_inner_step<WIDE, true>(S, cur, R, tmax, stk, ts);
            if (cur == LANE_DONE) verdict = 0;
        }
        // The exact primitive test is four times the cost of a node step and only a few lanes
        // want it at any time: lanes that have reached a leaf wait until LEAF_MIN have (or
        // nobody is descending any more), then test together.
        const bool at_leaf = alive && !at_node;
        const unsigned long long leafm = ballot64(at_leaf);
        if (leafm != 0ull && (__popcll(leafm) >= LEAF_MIN || ballot64(at_node) == 0ull)) {
            if (at_leaf) {
                const float4* lf = S.leaf + 3 * (size_t)(~cur);
                float4 l0 = lf[0], l1 = lf[1], l2 = lf[2];
                if (fbits(l2.z) != 0u) ts.sph++; else ts.tri++;
                ts.lane_prims++;
                float d2 = 0;
                HitUV cand;
                if (leaf_test(o, d, dd, l0, l1, l2, dmax, d2, cand)) verdict = 1;
                else if (!stk.empty()) cur = stk.pop();
                else verdict = 0;
            }
        }
        if (alive && stk.over) {
            // deeper than the LDS stack holds (rare): this one ray again, on a stack in private memory
            int b; float bd; HitUV buv;
            lbvh_trace<true, WIDE>(S, true, o, d, dmax, b, bd, buv, ts);
            verdict = b >= 0 ? 1 : 0;
            stk.reset();
        }
        if (verdict >= 0) {
            P.occl[item] = (uint8_t)verdict;
            alive = false;
        }
    }
}

// ---------------------------------------------------------------- shadow (any hit)
// The shadow query of getColor (src/raytrace.cpp:523-539) for every
// (hit, light) pair of the generation: one work item per pair, hit-major (item = hit * lights + light).
// (Light-major numbering -- a wave holding neighbouring hit points and ONE light -- was measured: the 11 M
// primitive scene's shadow launches took 61 ms instead of 51; the three rays of a hit share the nodes
// around their common origin.)
template <int ACCEL, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) RT_WAVES_ATTR void k_shadow(DScene S, DPool P, DCounters* C, int g) {
    __shared__ float4 lds[ACCEL == 1 ? BRUTE_CHUNK : (ACCEL == 2 || ACCEL == 4) ? LANE_STACK_F4 : ACCEL == 5 ? TEAM_LDS_F4 * (RT_BLOCK / 64) : 1];
    if (C->overflow) return;
    const uint32_t nh = C->hit_count[g];
    const uint64_t items = (uint64_t)nh * (uint64_t)S.nl;
    TraceStats ts;
    auto one = [&](uint64_t k) {
        bool active = k < items;
        uint32_t h = active ? (uint32_t)(k / (uint32_t)S.nl) : 0;
        int j = active ? (int)(k - (uint64_t)h * (uint32_t)S.nl) : 0;
        F3 point = f3(P.hp[P.hitlist[h]]);
        F3 I; float dist2;
        light_dir(S.lights[j], point, I, dist2);
        int best; float best_d2; HitUV uv;
        trace<ACCEL, true>(S, lds, active, point, I, dist2, best, best_d2, uv, ts);
        if (active) P.occl[k] = (best >= 0) ? 1 : 0;
    };
    if (ACCEL == 1) {
        for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < items; base += (uint64_t)gridDim.x * blockDim.x)
            one(base + threadIdx.x);
    } else if (ACCEL == 2 || ACCEL == 4) {
        // chunks of several 64-item groups when there is enough work for every wave to get a few
        const uint64_t groups = (items + 63) >> 6;
        const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
        uint64_t per = groups / (waves * 4);
        per = per < 1 ? 1 : (per > REFILL_GROUPS ? REFILL_GROUPS : per);
        const uint64_t span = per << 6;
        if (groups <= waves) {      // a group per wave at most, nothing to refill from: plain 64-item groups
            for_each_group(P.wq + (size_t)(2 * g + 1) * RT_WQ_LAUNCH, groups,
                           [&](uint64_t grp) { one((grp << 6) + (threadIdx.x & 63u)); });
        } else {
            GroupSource src;
            src.init(P.wq + (size_t)(2 * g + 1) * RT_WQ_LAUNCH, (items + span - 1) / span);
            shadow_chunks<ACCEL == 4>(S, P, (int*)lds, src, span, items, ts);
        }
    } else if (ACCEL == 5) {
        // team walks: the wave owns TEAM_NRS consecutive items, lane r the r-th
        const uint32_t lane = threadIdx.x & 63u;
        for_each_group(P.wq + (size_t)(2 * g + 1) * RT_WQ_LAUNCH, (items + TEAM_NRS - 1) / TEAM_NRS, [&](uint64_t pk) {
            const uint64_t k = pk * TEAM_NRS + lane;
            const bool active = lane < TEAM_NRS && k < items;
            F3 point = f3(0, 0, 0), I = f3(0, 0, 1); float dist2 = 0;
            if (active) {
                const uint32_t h = (uint32_t)(k / (uint32_t)S.nl);
                const int j = (int)(k - (uint64_t)h * (uint32_t)S.nl);
                point = f3(P.hp[P.hitlist[h]]);
                light_dir(S.lights[j], point, I, dist2);
            }
            int best; float best_d2; HitUV uv;
            team_walk<true, TEAM_NRS>(S, lds + (threadIdx.x >> 6) * TEAM_LDS_F4, active, point, I, dist2, best, best_d2, uv, ts);
            if (active) P.occl[k] = (best >= 0) ? 1 : 0;
        });
    } else {
        for_each_group(P.wq + (size_t)(2 * g + 1) * RT_WQ_LAUNCH, (items + 63) >> 6,
                       [&](uint64_t grp) { one((grp << 6) + (threadIdx.x & 63u)); });
    }
    if (COUNT) flush_stats(P, C, 1, ts);
}

// One unoccluded light's share of getColor (src/raytrace.cpp:541-587): diffuse and specular terms added to
// (rval, gval, bval) in the reference's order and precision.
__device__ __forceinline__ void add_light(const DLight& L, F3 point, F3 normal, F3 direct, F3 dif, F3 spec, float ns,
                                          float& rval, float& gval, float& bval) {
    F3 I; float dist2f;
    light_dir(L, point, I, dist2f);
    double dist = (double)dist2f;
    double dotni = (double)dot3(normal, I);
    double cosAlpha = (dotni > 0) ? dotni : 0;
    F3 lc = f3(L.cr, L.cg, L.cb);
    F3 illum = lc;
    if (L.type == 2) {
        illum = illum * (float)(1.0 / dist);
    } else if (L.type == 3) {
        double alpha = acos(cosAlpha);
        if (alpha < (double)L.a1) illum = lc * (float)(1.0 / dist);
        else if (alpha > (double)L.a2) illum = f3(0, 0, 0);
        else {
            double amt = 1 - ((alpha - (double)L.a1) / (double)(L.a2 - L.a1));
            illum = lc * (float)((1.0 / dist) * amt);
        }
    }
    rval = (float)((double)rval + (double)(dif.x * illum.x) * cosAlpha);
    gval = (float)((double)gval + (double)(dif.y * illum.y) * cosAlpha);
    bval = (float)((double)bval + (double)(dif.z * illum.z) * cosAlpha);
    F3 ref = (normal * (float)(2.0 * dotni)) - I;
    F3 Vv = direct * -1.0f;
    double powval = (double)dot3(norm3(Vv), norm3(ref));
    double pspec = (powval < 0.0) ? 0.0 : pow(powval, (double)ns);
    rval = (float)((double)rval + ((double)spec.x * pspec) * (double)illum.x);
    gval = (float)((double)gval + ((double)spec.y * pspec) * (double)illum.y);
    bval = (float)((double)bval + ((double)spec.z * pspec) * (double)illum.z);
}
// Directions of the secondary rays of a hit (src/raytrace.cpp:589-593, :600-624); irdir = -direction of the ray that hit.
__device__ __forceinline__ F3 reflect_dir(F3 normal, F3 irdir) { return (normal * (2.0f * dot3(normal, irdir))) - irdir; }
__device__ __forceinline__ F3 refract_dir(const Spawn& sp, F3 normal, F3 irdir) {
    F3 refdir;
    if (sp.dni >= 0) refdir = (normal * (float)(sp.ior * sp.dni - __builtin_sqrt(sp.tir))) - (irdir * (float)sp.ior);
    else             refdir = (normal * (float)(sp.ior * sp.dni + __builtin_sqrt(sp.tir))) - (irdir * (float)sp.ior);
    return norm3(refdir);
}

// Any-hit query against primitives that sit in LDS for the whole launch (scenes of a handful of primitives):
// brute_trace<true> without the staging, same order, same tests.
#define SHADE_LDS_MATS 128        // materials the shade kernels keep in LDS (8 KB); others are fetched from memory
#define SHADE_MAX_SPH 256
#define SHADE_MAX_TRI 128
__device__ __forceinline__ bool resident_any_hit(const float4* sph, int ns, const float4* tri, int nt, F3 o, F3 d, float dmax) {
    const double dd = (double)dot3(d, d);
    for (int i = 0; i < ns; i++) {
        float d2, tf;
        if (sphere_test(o, d, dd, sph[i], dmax, d2, tf)) return true;
    }
    for (int i = 0; i < nt; i++) {
        F3 a, ab, ac;
        tri_load(tri + 3 * i, a, ab, ac);
        TriHit h;
        if (tri_test(o, d, a, ab, ac, dmax, h)) return true;
    }
    return false;
}

// ---------------------------------------------------------------- shade + spawn
// getColor (src/raytrace.cpp:488-632) without the recursion: local illumination
// with the shadow results, then the reflection / refraction rays of the next
// generation are written at the offsets the scan assigned (or, past max_depth,
// the background is added directly as evaluateRayTree :132-135 returns it
// without tracing).
// PHASE 0: the whole of getColor; 1: only the spawn (child rays; no lights, so few registers: 8 waves per
// SIMD); 2: only the colour (lights, hand-over).  1 and 2 together do what 0 does, for the overlapped pipeline.
// 3: as 0, and the shadow query of every light is answered here, against primitives kept in LDS: a scene of a
// handful of primitives has no use for a shadow launch of its own and the round trip of its flags through memory.
template <int PHASE>
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(PHASE == 1 ? 1 : RT_SHADE_WAVES, PHASE == 1 ? 8 : RT_SHADE_WAVES))) void k_shade(DScene S, DPool P, DCounters* C, int g) {
    constexpr int phase = PHASE == 3 ? 0 : PHASE;
    __shared__ float4 l_sph[PHASE == 3 ? SHADE_MAX_SPH : 1];
    __shared__ float4 l_tri[PHASE == 3 ? 3 * SHADE_MAX_TRI : 1];
    // the materials of most scenes are a handful: kept in LDS, their fetch -- the link of the chain hit index ->
    // hit record -> material that every hit waits for -- costs an LDS access instead of a trip to the L2
    __shared__ float4 l_mats[PHASE == 1 ? 1 : 4 * SHADE_LDS_MATS];
    const int cached_mats = PHASE == 1 ? 0 : (S.nm < SHADE_LDS_MATS ? S.nm : SHADE_LDS_MATS);
    if (PHASE != 1) {
        for (int i = threadIdx.x; i < 4 * cached_mats; i += blockDim.x) l_mats[i] = S.mats[i];
        if (PHASE != 3) __syncthreads();
    }
    if (PHASE == 3) {
        for (int i = threadIdx.x; i < S.ns; i += blockDim.x) l_sph[i] = S.sph[i];
        for (int i = threadIdx.x; i < 3 * S.nt; i += blockDim.x) l_tri[i] = S.tri[i];
        __syncthreads();
    }
    if (C->overflow) return;
    const uint32_t start = gen_start(C, g);
    const uint32_t nh = C->hit_count[g];
    const uint32_t next_start = start + C->gen_count[g];
    const uint32_t gbase = C->gbase[g];
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i_next = (h < nh) ? P.hitlist[h] : 0u;
    for (; h < nh; h += stride) {
        const uint32_t i = i_next;
        if (h + stride < nh) i_next = P.hitlist[h + stride];      // the next hit's index travels while this one is shaded
        const uint32_t r = start + i;
        float4 qp = P.hp[i], qn = P.hn[i], qd = P.hd[i];
        // what the spawn step needs from the scan, fetched with the hit record instead of after the
        // lights: one dependent memory round trip less per hit (the kernel waits on memory 78 % of the time)
        const uint32_t grp = i >> 6;
        const uint32_t cbase = P.child_base[grp];
        const unsigned long long mrefl = P.mask_refl[gbase + grp], mrefr = P.mask_refr[gbase + grp];
        const uint32_t parent = fbits(qd.w), slot = fbits(qn.w) >> 31;
        F3 direct = f3(qd);
        F3 point = f3(qp);
        F3 normal = f3(qn);
        int mat = (int)(fbits(qn.w) & 0x7FFFFFFFu);
        float4 m0, m1, m2, m3;
        if (mat < cached_mats) { m0 = l_mats[4 * mat]; m1 = l_mats[4 * mat + 1]; m2 = l_mats[4 * mat + 2]; m3 = l_mats[4 * mat + 3]; }
        else { const float4* M = S.mats + 4 * (size_t)mat; m0 = M[0]; m1 = M[1]; m2 = M[2]; m3 = M[3]; }
        F3 dif = f3(m1), spec = f3(m2), trans = f3(m3);
        float ns = m0.w, iorf = m1.w;
        float rval = m0.x, gval = m0.y, bval = m0.z;
        rval *= S.ambient.x; gval *= S.ambient.y; bval *= S.ambient.z;
        for (int j = 0; j < (phase == 1 ? 0 : S.nl); j++) {
            const DLight L = S.lights[j];
            if (PHASE == 3) {
                F3 sI; float sdist2;
                light_dir(L, point, sI, sdist2);
                if (resident_any_hit(l_sph, S.ns, l_tri, S.nt, point, sI, sdist2)) continue;
            } else if (P.occl[(uint64_t)h * (uint32_t)S.nl + j]) continue;
            add_light(L, point, normal, direct, dif, spec, ns, rval, gval, bval);
        }
        F3 irdir = direct * -1.0f;
        Spawn sp = spawn_of(spec, trans, iorf, direct, normal);
        bool leaf = !(sp.refl || sp.refr);
        if (!leaf && g + 1 > S.max_depth) {
            // children sit past max_depth: each returns the background untraced
            if (sp.refl) { rval += spec.x * S.bg.x; gval += spec.y * S.bg.y; bval += spec.z * S.bg.z; }
            if (sp.refr) { rval += trans.x * S.bg.x; gval += trans.y * S.bg.y; bval += trans.z * S.bg.z; }
            leaf = true;
        }
        if (leaf) {
            if (phase != 1) deliver(S, P, g, parent, slot, f3(rval, gval, bval));
            continue;
        }
        uint32_t code = ((uint32_t)mat << 2) | (sp.refl ? 1u : 0u) | (sp.refr ? 2u : 0u);
        if (phase != 1) P.loc[r] = make_float4(rval, gval, bval, bitsf(code));
        if (phase == 2) continue;
        // where the scan put this ray's children: rays of the group before it first
        const unsigned long long below = (1ull << (i & 63u)) - 1ull;
        uint64_t at = (uint64_t)next_start + cbase + (uint32_t)__popcll(mrefl & below) + (uint32_t)__popcll(mrefr & below);
        if (at + (sp.refl ? 1u : 0u) + (sp.refr ? 1u : 0u) > (uint64_t)P.capacity) continue;   // overflow, flagged by the scan
        if (sp.refl) {
            F3 rfd = reflect_dir(normal, irdir);
            P.o[at] = make_float4(point.x, point.y, point.z, bitsf(r));
            P.d[at] = make_float4(rfd.x, rfd.y, rfd.z, bitsf(0u));
            P.link[at] = r;
            at++;
        }
        if (sp.refr) {
            F3 rad = refract_dir(sp, normal, irdir);
            P.o[at] = make_float4(point.x, point.y, point.z, bitsf(r));
            P.d[at] = make_float4(rad.x, rad.y, rad.z, bitsf(1u));
            P.link[at] = r | 0x80000000u;
        }
    }
}

// ---------------------------------------------------------------- deferred misses
// The background for the rays of generation g >= 1 that hit nothing, when the extend kernel was told
// to leave them (overlapped pipeline: the parents' records may not be written yet while it runs).
__global__ __launch_bounds__(RT_BLOCK) void k_miss(DScene S, DPool P, DCounters* C, int g) {
    if (C->overflow) return;
    const uint32_t start = gen_start(C, g);
    const uint32_t count = gen_size(C, g, start, P.capacity);
    const uint32_t gbase = C->gbase[g];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        if ((P.mask_hit[gbase + (i >> 6)] >> (i & 63u)) & 1ull) continue;
        const uint32_t link = P.link[start + i];
        deliver(S, P, g, link & 0x7FFFFFFFu, link >> 31, S.bg);
    }
}

// ---------------------------------------------------------------- resolve (bottom-up)
// Rays that sent out BOTH a reflection and a refraction ray: adds the children's colours in the
// reference's order -- reflection first (src/raytrace.cpp:594-596), then refraction (:625-627) --
// and hands the sum on (deliver).  Launched per generation, deepest first, and only for scenes
// that have a material which can spawn both.
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(DScene S, DPool P, DCounters* C, int g) {
    if (C->overflow) return;
    const uint32_t start = gen_start(C, g);
    const uint32_t count = gen_size(C, g, start, P.capacity);
    const uint32_t gbase = C->gbase[g];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint32_t grp = gbase + (i >> 6);
        const unsigned long long both = P.mask_refl[grp] & P.mask_refr[grp];
        if (!((both >> (i & 63u)) & 1ull)) continue;
        uint32_t r = start + i;
        float4 l = P.loc[r];
        const float4* M = S.mats + 4 * (size_t)(fbits(l.w) >> 2);
        float rval = l.x, gval = l.y, bval = l.z;
        float4 s = M[2], c = P.cre[r];
        rval += s.x * c.x; gval += s.y * c.y; bval += s.z * c.z;
        float4 t = M[3], ca = P.cra[r];
        rval += t.x * ca.x; gval += t.y * ca.y; bval += t.z * ca.z;
        if (g == 0) deliver(S, P, 0, r, 0, f3(rval, gval, bval));
        else {
            const uint32_t link = P.link[r];
            deliver(S, P, g, link & 0x7FFFFFFFu, link >> 31, f3(rval, gval, bval));
        }
    }
}

// max_depth < 0: the primary ray itself sits past max_depth, so evaluateRayTree returns the
// background for every sample without tracing anything (src/raytrace.cpp:132-135).
__global__ void k_fill_background(DScene S, DPool P, uint32_t nsamples) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nsamples; i += gridDim.x * blockDim.x)
        P.scol[i] = make_float4(S.bg.x, S.bg.y, S.bg.z, 0.0f);
}

// ---------------------------------------------------------------- finish
// ave (src/raytrace.cpp:1079-1089) then Pixel::SetClamp(c*255.0)
// (src/raytrace.cpp:112, src/pixel.cpp:38-43): (int) truncation as x86-64
// performs it (NaN and out-of-range become INT_MIN), then clamp to 0..255.
__device__ __forceinline__ uint8_t quantise(float c) {
    double v = (double)c * 255.0;
    int i;
    if (!(v == v) || v >= 2147483648.0 || v <= -2147483649.0) i = (-2147483647 - 1);
    else i = (int)v;
    return (uint8_t)((i < 0) ? 0 : (i > 255) ? 255 : i);
}

// One pixel per thread, but a pixel's samples lie 16 * spp bytes apart from its
// neighbour's: read per thread, every load touches 64 cache lines.  The block therefore
// stages its 256 pixels through LDS, 8 samples at a time (25 KB: six blocks per CU; 16 at a time
// leaves three and is 19 % slower): coalesced loads (lane l takes
// record l of the block's contiguous run), then each thread adds up its own pixel in
// sample order.  25 floats per pixel row keeps the per-thread reads off each other's banks.
#define FIN_CHUNK 8
#define FIN_ROW (3 * FIN_CHUNK + 1)
__global__ __launch_bounds__(RT_BLOCK) void k_finish(DCamera cam, DPool P, const uint32_t* tiles,
                                                     uint32_t tile0, uint32_t npixels, int shard_mode,
                                                     uint8_t* out, float* outf) {
    __shared__ float stage[RT_BLOCK * FIN_ROW];
    const uint32_t spp = (uint32_t)cam.spp;
    for (uint32_t base = blockIdx.x * blockDim.x; base < npixels; base += gridDim.x * blockDim.x) {
        const uint32_t lp = base + threadIdx.x;
        const uint32_t npx = min((uint32_t)RT_BLOCK, npixels - base);
        F3 u = f3(0, 0, 0);
        for (uint32_t j0 = 0; j0 < spp; j0 += FIN_CHUNK) {
            const uint32_t cnt = min((uint32_t)FIN_CHUNK, spp - j0);
            __syncthreads();
            for (uint32_t e = threadIdx.x; e < npx * cnt; e += RT_BLOCK) {
                uint32_t p = e / cnt, j = e - p * cnt;
                float4 c = P.scol[(size_t)(base + p) * spp + j0 + j];
                float* dst = stage + p * FIN_ROW + 3 * j;
                dst[0] = c.x; dst[1] = c.y; dst[2] = c.z;
            }
            __syncthreads();
            if (lp < npixels) {
                const float* src = stage + threadIdx.x * FIN_ROW;
                uint32_t j = 0;
                if (j0 == 0) { u = f3(src[0], src[1], src[2]); j = 1; }   // ave starts from the first sample, not from zero
                for (; j < cnt; j++) u = u + f3(src[3 * j], src[3 * j + 1], src[3 * j + 2]);
            }
        }
        if (lp >= npixels) continue;
        int x, y, px, py; uint32_t tile;
        pixel_of(tiles, tile0, lp, x, y, px, py, tile);
        if (x >= cam.width || y >= cam.height) continue;
        if (spp > 1) u = u / (float)cam.spp;
        size_t at = shard_mode ? ((size_t)tile * RT_TILE_AREA + (size_t)py * RT_TILE_PX + px)
                               : ((size_t)y * cam.width + x);
        if (out) {
            out[3 * at] = quantise(u.x); out[3 * at + 1] = quantise(u.y); out[3 * at + 2] = quantise(u.z);
        }
        if (outf) { outf[3 * at] = u.x; outf[3 * at + 1] = u.y; outf[3 * at + 2] = u.z; }
    }
}

// Packed shard buffers of all ranks -> row-major frame.  rowbase[rank*nty + ty] =
// index (in the rank's tile order) of the first tile the rank owns in tile row ty.
__global__ void k_assemble(const uint8_t* all, size_t shard_bytes, int world, int width, int height,
                           int ntx, int nty, const uint32_t* rowbase, uint8_t* out) {
    size_t n = (size_t)width * height;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int y = (int)(i / width), x = (int)(i - (size_t)y * width);
        int tx = x / RT_TILE_PX, ty = y / RT_TILE_PX;
        int rank = (tx + 3 * ty) % world;
        int first = ((rank - 3 * ty) % world + world) % world;   // first tx owned in this row
        uint32_t local = rowbase[rank * nty + ty] + (uint32_t)((tx - first) / world);
        size_t src = (size_t)rank * shard_bytes +
                     3 * ((size_t)local * RT_TILE_AREA + (size_t)(y % RT_TILE_PX) * RT_TILE_PX + (x % RT_TILE_PX));
        out[3 * i] = all[src]; out[3 * i + 1] = all[src + 1]; out[3 * i + 2] = all[src + 2];
    }
}

// rowbase[rank * nty + ty] of k_assemble, computed on the device (one thread per rank): tiles of tile row ty
// that belong to `rank` are tx = first, first + world, ... with first = (rank - 3 ty) mod world.
__global__ void k_rowbase(int world, int ntx, int nty, uint32_t* rowbase) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= world) return;
    uint32_t acc = 0;
    for (int ty = 0; ty < nty; ty++) {
        rowbase[(size_t)r * nty + ty] = acc;
        const int first = ((r - 3 * ty) % world + world) % world;
        if (first < ntx) acc += (uint32_t)((ntx - first + world - 1) / world);
    }
}

// ---------------------------------------------------------------- ray queries (validation)
template <int ACCEL, bool ANY>
__global__ __launch_bounds__(RT_BLOCK) void k_query(DScene S, int64_t n, const float* o, const float* d,
                                                    const float* dmax, int* prim_out, float* d2_out) {
    __shared__ float4 lds[ACCEL == 1 ? BRUTE_CHUNK : ACCEL == 5 ? TEAM_LDS_F4 * (RT_BLOCK / 64) : 1];
    TraceStats ts;
    if (ACCEL == 5) {
        // team walks: a wave takes its 64 rays NR at a time, lane r the r-th
        constexpr int NR = ANY ? TEAM_NRS : TEAM_NR;
        const int lane = (int)(threadIdx.x & 63u);
        for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / 64 * 64; base < n; base += (int64_t)gridDim.x * blockDim.x) {
            for (int pass = 0; pass < 64 / NR; pass++) {
                const int64_t i = base + (int64_t)pass * NR + lane;
                const bool active = lane < NR && i < n;
                const int64_t ii = active ? i : 0;
                F3 ro = f3(o[3 * ii], o[3 * ii + 1], o[3 * ii + 2]);
                F3 rd = f3(d[3 * ii], d[3 * ii + 1], d[3 * ii + 2]);
                float dm = dmax ? dmax[ii] : __builtin_inff();
                int best; float best_d2; HitUV uv;
                team_walk<ANY, NR>(S, lds + (threadIdx.x >> 6) * TEAM_LDS_F4, active, ro, rd, dm, best, best_d2, uv, ts);
                if (active) {
                    prim_out[i] = best;
                    if (d2_out) d2_out[i] = best >= 0 ? best_d2 : __builtin_inff();
                }
            }
        }
        return;
    }
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = base + threadIdx.x;
        bool active = i < n;
        int64_t ii = active ? i : 0;
        F3 ro = f3(o[3 * ii], o[3 * ii + 1], o[3 * ii + 2]);
        F3 rd = f3(d[3 * ii], d[3 * ii + 1], d[3 * ii + 2]);
        float dm = dmax ? dmax[ii] : __builtin_inff();
        int best; float best_d2; HitUV uv;
        trace<ACCEL, ANY>(S, lds, active, ro, rd, dm, best, best_d2, uv, ts);
        if (active) {
            prim_out[i] = best;
            if (d2_out) d2_out[i] = best >= 0 ? best_d2 : __builtin_inff();
        }
    }
}

// ---------------------------------------------------------------- launchers
static inline int grid_for(uint64_t items, int max_blocks) {
    uint64_t b = (items + RT_BLOCK - 1) / RT_BLOCK;
    if (b < 1) b = 1;
    if (b > (uint64_t)max_blocks) b = max_blocks;
    return (int)b;
}

void rtk_init_counters(hipStream_t st, DCounters* C, uint32_t nsamples, const DPool& P, int generations, bool count_work) {
    // scan state words, scan tickets and the work-queue counters of the generations in use: one contiguous run
    const size_t zero_bytes = (size_t)((char*)(P.wq + (size_t)2 * (generations > 0 ? generations : 1) * RT_WQ_LAUNCH) - (char*)P.scan_state);
    hipMemsetAsync(P.scan_state, 0, zero_bytes, st);
    if (count_work) hipMemsetAsync(P.wstats, 0, (size_t)2 * RT_MAX_WAVES * 8 * sizeof(unsigned long long), st);
    hipLaunchKernelGGL(k_init_counters, dim3(1), dim3(128), 0, st, C, nsamples);
}
// Traversal launches are persistent grids sized to what is RESIDENT at once: with
// more blocks than fit, the surplus waves start late and run alone at the end of
// the launch (each wave owns a fixed share of the rays), which costs more than it
// hides.  Blocks per CU come from the occupancy query, once per kernel.
template <typename K>
static int resident_blocks(K kernel, int max_blocks) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, RT_BLOCK, 0) != hipSuccess || per_cu < 1) return max_blocks;
    int cus = max_blocks / 8;   // max_blocks is 8 x CU count
    return std::min(max_blocks, std::max(1, per_cu) * cus);
}

template <bool COUNT>
static void launch_extend(hipStream_t st, int accel, const DScene& S, const DPool& P, DCounters* C, const GenSrc& G,
                          int g, uint64_t bound, int max_blocks, int* res) {
    if (!res[accel]) res[accel] = accel == 1 ? resident_blocks(k_extend<1, COUNT>, max_blocks)
                                 : accel == 2 ? resident_blocks(k_extend<2, COUNT>, max_blocks)
                                 : accel == 4 ? resident_blocks(k_extend<4, COUNT>, max_blocks)
                                 : accel == 5 ? resident_blocks(k_extend<5, COUNT>, max_blocks)
                                              : resident_blocks(k_extend<3, COUNT>, max_blocks);
    dim3 grid(grid_for(accel == 5 ? bound * (64 / TEAM_NR) : bound, res[accel])), block(RT_BLOCK);
    if (accel == 1) hipLaunchKernelGGL((k_extend<1, COUNT>), grid, block, 0, st, S, P, C, G, g);
    else if (accel == 2) hipLaunchKernelGGL((k_extend<2, COUNT>), grid, block, 0, st, S, P, C, G, g);
    else if (accel == 4) hipLaunchKernelGGL((k_extend<4, COUNT>), grid, block, 0, st, S, P, C, G, g);
    else if (accel == 5) hipLaunchKernelGGL((k_extend<5, COUNT>), grid, block, 0, st, S, P, C, G, g);
    else hipLaunchKernelGGL((k_extend<3, COUNT>), grid, block, 0, st, S, P, C, G, g);
}
void rtk_extend(hipStream_t st, int accel, const DScene& S, const DPool& P, DCounters* C, const DCamera& cam,
                const uint32_t* tiles, uint32_t tile0, int g, uint64_t bound, int max_blocks, bool count,
                RtkLaunchCache& cache, bool defer_miss) {
    GenSrc G{cam, tiles, tile0, defer_miss ? 1 : 0};
    if (count || S.profile) launch_extend<true>(st, accel, S, P, C, G, g, bound, max_blocks, cache.extend[1]);
    else launch_extend<false>(st, accel, S, P, C, G, g, bound, max_blocks, cache.extend[0]);
}
void rtk_compact(hipStream_t st, const DPool& P, DCounters* C, int g, uint64_t bound, int max_blocks, int extend_accel) {
    uint64_t groups = (bound + 63) / 64;
    int blocks = (int)std::min<uint64_t>((groups + SCAN_BLOCK - 1) / SCAN_BLOCK, (uint64_t)std::min<int>(max_blocks, (int)SCAN_TILES_WANTED));
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_scan, dim3(blocks), dim3(SCAN_BLOCK), 0, st, P, C, g, (extend_accel == 5 || extend_accel == 2 || extend_accel == 4) ? 1 : 0);
}
template <bool COUNT>
static void launch_shadow(hipStream_t st, int accel, const DScene& S, const DPool& P, DCounters* C, int g,
                          uint64_t bound, int max_blocks, int* res) {
    if (!res[accel]) res[accel] = accel == 1 ? resident_blocks(k_shadow<1, COUNT>, max_blocks)
                                 : accel == 2 ? resident_blocks(k_shadow<2, COUNT>, max_blocks)
                                 : accel == 4 ? resident_blocks(k_shadow<4, COUNT>, max_blocks)
                                 : accel == 5 ? resident_blocks(k_shadow<5, COUNT>, max_blocks)
                                              : resident_blocks(k_shadow<3, COUNT>, max_blocks);
    dim3 grid(grid_for(accel == 5 ? bound * (64 / TEAM_NRS) : bound, res[accel])), block(RT_BLOCK);
    if (accel == 1) hipLaunchKernelGGL((k_shadow<1, COUNT>), grid, block, 0, st, S, P, C, g);
    else if (accel == 2) hipLaunchKernelGGL((k_shadow<2, COUNT>), grid, block, 0, st, S, P, C, g);
    else if (accel == 4) hipLaunchKernelGGL((k_shadow<4, COUNT>), grid, block, 0, st, S, P, C, g);
    else if (accel == 5) hipLaunchKernelGGL((k_shadow<5, COUNT>), grid, block, 0, st, S, P, C, g);
    else hipLaunchKernelGGL((k_shadow<3, COUNT>), grid, block, 0, st, S, P, C, g);
}
void rtk_shadow(hipStream_t st, int accel, const DScene& S, const DPool& P, DCounters* C, int g,
                uint64_t bound, int max_blocks, bool count, RtkLaunchCache& cache) {
    if (S.nl == 0) return;
    if (count || S.profile) launch_shadow<true>(st, accel, S, P, C, g, bound, max_blocks, cache.shadow[1]);
    else launch_shadow<false>(st, accel, S, P, C, g, bound, max_blocks, cache.shadow[0]);
}
bool rtk_shade_can_shadow(int ns, int nt) { return ns <= SHADE_MAX_SPH && nt <= SHADE_MAX_TRI; }
void rtk_shade(hipStream_t st, const DScene& S, const DPool& P, DCounters* C, int g, uint64_t bound,
               int max_blocks, int phase) {
    dim3 grid(grid_for(bound, max_blocks)), block(RT_BLOCK);
    if (phase == 1) hipLaunchKernelGGL(k_shade<1>, grid, block, 0, st, S, P, C, g);
    else if (phase == 2) hipLaunchKernelGGL(k_shade<2>, grid, block, 0, st, S, P, C, g);
    else if (phase == 3) hipLaunchKernelGGL(k_shade<3>, grid, block, 0, st, S, P, C, g);
    else hipLaunchKernelGGL(k_shade<0>, grid, block, 0, st, S, P, C, g);
}
void rtk_miss(hipStream_t st, const DScene& S, const DPool& P, DCounters* C, int g, uint64_t bound, int max_blocks) {
    hipLaunchKernelGGL(k_miss, dim3(grid_for(bound, max_blocks)), dim3(RT_BLOCK), 0, st, S, P, C, g);
}
void rtk_reduce_stats(hipStream_t st, const DPool& P, DCounters* C) {
    hipLaunchKernelGGL(k_reduce_stats, dim3(10), dim3(256), 0, st, P, C);
}
void rtk_resolve(hipStream_t st, const DScene& S, const DPool& P, DCounters* C, int g, uint64_t bound,
                 int max_blocks) {
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(bound, max_blocks)), dim3(RT_BLOCK), 0, st, S, P, C, g);
}
void rtk_fill_background(hipStream_t st, const DScene& S, const DPool& P, uint32_t nsamples, int max_blocks) {
    if (nsamples == 0) return;
    hipLaunchKernelGGL(k_fill_background, dim3(grid_for(nsamples, max_blocks)), dim3(RT_BLOCK), 0, st, S, P, nsamples);
}
void rtk_finish(hipStream_t st, const DCamera& cam, const DPool& P, const uint32_t* tiles, uint32_t tile0,
                uint32_t npixels, int shard_mode, uint8_t* out, float* outf, int max_blocks) {
    hipLaunchKernelGGL(k_finish, dim3(grid_for(npixels, max_blocks)), dim3(RT_BLOCK), 0, st, cam, P, tiles,
                       tile0, npixels, shard_mode, out, outf);
}
void rtk_assemble(hipStream_t st, const uint8_t* all, size_t shard_bytes, int world, int width, int height,
                  int ntx, int nty, const uint32_t* rowbase, uint8_t* out) {
    hipLaunchKernelGGL(k_assemble, dim3(grid_for((uint64_t)width * height, 4096)), dim3(RT_BLOCK), 0, st, all,
                       shard_bytes, world, width, height, ntx, nty, rowbase, out);
}
void rtk_rowbase(hipStream_t st, int world, int ntx, int nty, uint32_t* rowbase) {
    hipLaunchKernelGGL(k_rowbase, dim3((unsigned)((world + 63) / 64)), dim3(64), 0, st, world, ntx, nty, rowbase);
}
void rtk_query(hipStream_t st, int accel, int any_hit, const DScene& S, int64_t n, const float* o,
               const float* d, const float* dmax, int* prim_out, float* d2_out) {
    dim3 grid(grid_for((uint64_t)n, 2048)), block(RT_BLOCK);
    if (accel == 1) {
        if (any_hit) hipLaunchKernelGGL((k_query<1, true>), grid, block, 0, st, S, n, o, d, dmax, prim_out, d2_out);
        else hipLaunchKernelGGL((k_query<1, false>), grid, block, 0, st, S, n, o, d, dmax, prim_out, d2_out);
    } else if (accel == 2) {
        if (any_hit) hipLaunchKernelGGL((k_query<2, true>), grid, block, 0, st, S, n, o, d, dmax, prim_out, d2_out);
        else hipLaunchKernelGGL((k_query<2, false>), grid, block, 0, st, S, n, o, d, dmax, prim_out, d2_out);
    } else if (accel == 4) {
        if (any_hit) hipLaunchKernelGGL((k_query<4, true>), grid, block, 0, st, S, n, o, d, dmax, prim_out, d2_out);
        else hipLaunchKernelGGL((k_query<4, false>), grid, block, 0, st, S, n, o, d, dmax, prim_out, d2_out);
    } else if (accel == 5) {
        if (any_hit) hipLaunchKernelGGL((k_query<5, true>), grid, block, 0, st, S, n, o, d, dmax, prim_out, d2_out);
        else hipLaunchKernelGGL((k_query<5, false>), grid, block, 0, st, S, n, o, d, dmax, prim_out, d2_out);
    } else {
        if (any_hit) hipLaunchKernelGGL((k_query<3, true>), grid, block, 0, st, S, n, o, d, dmax, prim_out, d2_out);
        else hipLaunchKernelGGL((k_query<3, false>), grid, block, 0, st, S, n, o, d, dmax, prim_out, d2_out);
    }
}
void rtk_primary_rays(hipStream_t st, const DCamera& cam, int64_t n, const int* x, const int* y,
                      const int* j, float* o_out, float* d_out) {
    hipLaunchKernelGGL(k_primary_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cam, n, x, y, j,
                       o_out, d_out);
}
