// GPU LBVH over spheres + triangles for gfx950.
//
// Stands where the reference's BVH::make stood (src/raytrace.cpp:652-788), which
// is incomplete (it drops large triangles) and is not reproduced.  Built fresh:
//   1. per-primitive padded AABB + scene bounds (block reduce + float atomics)
//   2. 63-bit Morton code of the box centre (21 bits per axis)
//   3. radix sort of (code, primitive id) pairs            [rt_sort.h: eight 8-bit passes, wave64 ballots]
//   4. Karras 2012: one thread per inner node finds its key range and split
//   5. bottom-up refit: the second thread to arrive at a node merges the boxes
// The tree only has to be CONSERVATIVE: traversal re-runs the reference's exact
// primitive tests, so the image equals brute force however the tree is shaped.
//
// Node i = 4 float4: child 0 box (lo.x, hi.x, lo.y, hi.y, lo.z, hi.z), child 1 box likewise, (ref0, ref1, -, -)
// ref >= 0: inner node, ref < 0: leaf = sorted primitive slot ~ref.
#include "rt_device.h"
#include "rt_kernels.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "rt_sort.h"

struct Box { float lo[3], hi[3]; };

// Triangle record (a.xyz, ab.x)(ab.yz, ac.xy)(ac.z, ...) -> the three corners the
// exact test spans: a, a - ab, a - ac.
__device__ inline Box raw_box(const float4* sph, int ns, const float4* tri, int p) {
    Box b;
    if (p < ns) {
        float4 s = sph[p];
        float r = fabsf(s.w);
        b.lo[0] = s.x - r; b.lo[1] = s.y - r; b.lo[2] = s.z - r;
        b.hi[0] = s.x + r; b.hi[1] = s.y + r; b.hi[2] = s.z + r;
    } else {
        const float4* t = tri + 3 * (size_t)(p - ns);
        float4 q0 = t[0], q1 = t[1], q2 = t[2];
        float a[3] = {q0.x, q0.y, q0.z}, ab[3] = {q0.w, q1.x, q1.y}, ac[3] = {q1.z, q1.w, q2.x};
        for (int k = 0; k < 3; k++) {
            float v1 = a[k] - ab[k], v2 = a[k] - ac[k];
            b.lo[k] = fminf(a[k], fminf(v1, v2));
            b.hi[k] = fmaxf(a[k], fmaxf(v1, v2));
        }
    }
    return b;
}

// order-preserving float <-> uint map so atomicMin/Max on uints reduce floats
__host__ __device__ inline uint32_t f2ord(float f) {
#ifdef __HIP_DEVICE_COMPILE__
    uint32_t u = __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4);
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t u) {
    uint32_t v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(v);
#else
    float f; memcpy(&f, &v, 4); return f;
#endif
}

// The box a walk may cull a primitive by.  It has to hold every ray the reference's exact test can ACCEPT, and
// those tests are float arithmetic on the vector from the ray's origin to the primitive, so what they accept
// grows with the distance E to the origin:
//   sphere (src/raytrace.cpp:232-241): det = (d.e)^2 - |d|^2 (|e|^2 - r^2) with d.e, |e|^2, r^2 and the difference
//     rounded to float.  With u = 2^-24 the rounding errors sum to at most 13 u |d|^2 |e|^2 + 5 u |d|^2 r^2, so
//     det >= 0 can hold for a line that passes the centre at any distance rho with
//     rho^2 <= r^2 (1 + 6u) + 8.4e-7 |e|^2: a sphere of radius 0.01 seen from 10 away is "hit" up to 0.013 from
//     its centre.  The box is that of the radius sqrt(r^2 (1 + 2e-6) + 1e-6 E^2).
//   triangle (:314-334): float determinants whose errors are about 8 u |e| |edge| |edge'|: the ray may pass an
//     edge at a distance of 8 u E / sin(angle to the edge) = 5e-7 E / sin on the wrong side.  Padded by 2e-6 E
//     (rays within 14 degrees of the edge they graze, and slivers, stay outside the guarantee: the bound is a
//     worst case that sums every rounding error with the same sign; in a directed test of ten million rays
//     aimed within a few roundings of edges from up to 3 000 units away, an eighth of this pad was enough).
// E = distance from the primitive to the farthest point a ray can start from: the region spanned by the camera's
// origins and all primitives (secondary and shadow rays start on a primitive).  The 1e-4 of the coordinates'
// magnitude covers what the tests round in the primitive's own data.
#ifndef RT_TRI_PAD_PER_DISTANCE
#define RT_TRI_PAD_PER_DISTANCE 2e-6f
#endif
__device__ inline Box prim_box(const float4* sph, int ns, const float4* tri, int p, const uint32_t* region) {
    Box b = raw_box(sph, ns, tri, p);
    float m = 0.0f, e2 = 0.0f, h2 = 0.0f;
    for (int k = 0; k < 3; k++) {
        m = fmaxf(m, fmaxf(fabsf(b.lo[k]), fabsf(b.hi[k])));
        float c = 0.5f * (b.lo[k] + b.hi[k]), half = 0.5f * (b.hi[k] - b.lo[k]);
        float far = fmaxf(c - ord2f(region[k]), ord2f(region[3 + k]) - c);
        e2 += far * far; h2 += half * half;
    }
    const float E = sqrtf(e2) + sqrtf(h2);
    float pad = m * 1e-4f + 1e-30f;
    if (p < ns) {
        float r = fabsf(sph[p].w);
        pad += fmaxf(sqrtf(r * r * 1.000002f + 1e-6f * E * E) - r, 0.0f);
    } else {
        pad += RT_TRI_PAD_PER_DISTANCE * E;
    }
    for (int k = 0; k < 3; k++) { b.lo[k] -= pad; b.hi[k] += pad; }
    return b;
}

// region[0..5] (ordered uints, initialised with the box of the camera's ray origins) grows to hold every primitive
__global__ void k_region(const float4* sph, int ns, const float4* tri, int n, uint32_t* region) {
    __shared__ uint32_t red[6];
    if (threadIdx.x < 6) red[threadIdx.x] = (threadIdx.x < 3) ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        Box b = raw_box(sph, ns, tri, p);
        for (int k = 0; k < 3; k++) {
            if (b.lo[k] == b.lo[k]) atomicMin(&red[k], f2ord(b.lo[k]));
            if (b.hi[k] == b.hi[k]) atomicMax(&red[3 + k], f2ord(b.hi[k]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&region[threadIdx.x], red[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&region[threadIdx.x], red[threadIdx.x]);
}

__global__ void k_bounds(const float4* sph, int ns, const float4* tri, int n, const uint32_t* region, Box* boxes,
                         uint32_t* scene /*[6]*/) {
    __shared__ uint32_t red[6];
    if (threadIdx.x < 6) red[threadIdx.x] = (threadIdx.x < 3) ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        Box b = prim_box(sph, ns, tri, p, region);
        boxes[p] = b;
        for (int k = 0; k < 3; k++) {
            float c = 0.5f * (b.lo[k] + b.hi[k]);
            if (c == c) { atomicMin(&red[k], f2ord(c)); atomicMax(&red[3 + k], f2ord(c)); }
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&scene[threadIdx.x], red[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&scene[threadIdx.x], red[threadIdx.x]);
}

__device__ inline unsigned long long spread21(unsigned long long x) {
    x &= 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void k_morton(const Box* boxes, int n, const uint32_t* scene, unsigned long long* keys, uint32_t* vals) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    unsigned long long code = 0;
    for (int k = 0; k < 3; k++) {
        float lo = ord2f(scene[k]), hi = ord2f(scene[3 + k]);
        float c = 0.5f * (boxes[p].lo[k] + boxes[p].hi[k]);
        float ext = hi - lo;
        float u = (ext > 0.0f) ? (c - lo) / ext : 0.0f;
        if (!(u >= 0.0f)) u = 0.0f;
        if (u > 1.0f) u = 1.0f;
        unsigned long long q = (unsigned long long)fminf(u * 2097152.0f, 2097151.0f);
        code |= spread21(q) << (2 - k);
    }
    keys[p] = code;
    vals[p] = (uint32_t)p;
}

// Karras 2012, "Maximizing Parallelism in the Construction of BVHs, Octrees, and
// k-d Trees": common-prefix length of sorted keys i and j, ties broken by index.
__device__ inline int delta(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

__global__ void k_hierarchy(const unsigned long long* keys, int n, int* child /*[2*(n-1)]*/,
                            int* parent_inner /*[n-1]*/, int* parent_leaf /*[n]*/) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    int c0 = (lo == gamma) ? ~gamma : gamma;
    int c1 = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    child[2 * i] = c0;
    child[2 * i + 1] = c1;
    if (c0 >= 0) parent_inner[c0] = 2 * i; else parent_leaf[~c0] = 2 * i;
    if (c1 >= 0) parent_inner[c1] = 2 * i + 1; else parent_leaf[~c1] = 2 * i + 1;
    if (i == 0) parent_inner[0] = -1;
}

__device__ inline void store_child_box(float4* nodes, int node, int side, const Box& b) {
    float* f = (float*)(nodes + 4 * (size_t)node);
    float* dst = f + 6 * side;   // floats 0..5 = box 0, 6..11 = box 1, each (lo.x, hi.x, lo.y, hi.y, lo.z, hi.z)
    dst[0] = b.lo[0]; dst[1] = b.hi[0]; dst[2] = b.lo[1];
    dst[3] = b.hi[1]; dst[4] = b.lo[2]; dst[5] = b.hi[2];
}

__global__ void k_refit(const Box* boxes, const uint32_t* sorted, int n, const int* child, const int* parent_inner,
                        const int* parent_leaf, uint32_t* flags, float4* nodes) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    Box b = boxes[sorted[j]];
    int link = parent_leaf[j];
    while (link >= 0) {
        int node = link >> 1, side = link & 1;
        store_child_box(nodes, node, side, b);
        __threadfence();
        uint32_t old = atomicAdd(&flags[node], 1u);
        if (old == 0) return;                 // first arrival: the sibling's thread continues
        __threadfence();
        // second arrival: both child boxes are in place; read the sibling's through L2
        const float* f = (const float*)(nodes + 4 * (size_t)node);
        const float* sib = f + 6 * (1 - side);
        for (int k = 0; k < 3; k++) {
            float slo = __hip_atomic_load(&sib[2 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float shi = __hip_atomic_load(&sib[2 * k + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b.lo[k] = fminf(b.lo[k], slo);
            b.hi[k] = fmaxf(b.hi[k], shi);
        }
        // child references live in the last float4 of the node
        int* refs = (int*)(nodes + 4 * (size_t)node + 3);
        refs[0] = child[2 * node]; refs[1] = child[2 * node + 1]; refs[2] = 0; refs[3] = 0;
        link = parent_inner[node];
    }
}

// Leaf records in Morton order, so traversal reads a primitive without going
// through prim_idx: 3 float4 per leaf,
//   triangle: (a.xyz, ab.x) (ab.yz, ac.xy) (ac.z, prim id, 0, -)
//   sphere:   (c.xyz, r)    (-)            (-,    prim id, 1, -)
__global__ void k_leaves(const float4* sph, int ns, const float4* tri, const uint32_t* sorted, int n, float4* leaf) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t p = sorted[j];
    float4 q0, q1 = make_float4(0, 0, 0, 0), q2;
    if ((int)p < ns) {
        q0 = sph[p];
        q2 = make_float4(0.0f, __uint_as_float(p), __uint_as_float(1u), 0.0f);
    } else {
        const float4* t = tri + 3 * (size_t)(p - ns);
        q0 = t[0]; q1 = t[1];
        q2 = make_float4(t[2].x, __uint_as_float(p), __uint_as_float(0u), 0.0f);
    }
    leaf[3 * (size_t)j] = q0; leaf[3 * (size_t)j + 1] = q1; leaf[3 * (size_t)j + 2] = q2;
}

// Depth of the tree = the largest number of inner nodes above a leaf: bounds the
// traversal stacks.
__global__ void k_depth(const int* parent_inner, const int* parent_leaf, int n, uint32_t* out) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int dpt = 0;
    if (j < n)
        for (int link = parent_leaf[j]; link >= 0; link = parent_inner[link >> 1]) dpt++;
    for (int off = 32; off > 0; off >>= 1) dpt = max(dpt, __shfl_down(dpt, off));
    if ((threadIdx.x & 63) == 0 && dpt > 0) atomicMax(out, (uint32_t)dpt);
}

// ---------------------------------------------------------------- PLOC
// Parallel locally-ordered clustering (Meister & Bittner 2018): bottom-up agglomeration
// over the Morton-sorted sequence.  Every round, each cluster looks RADIUS places to either
// side for the neighbour whose union with it has the smallest surface area; two clusters
// that pick each other merge into a new inner node, which takes the place of the left
// one; the sequence is compacted and the round repeats until one cluster is left.
// The trees come out close to a surface-area-heuristic build (fewer node visits per ray
// than the Karras hierarchy over the same codes) at the cost of ~40 small rounds.
// Ties go to the lower index, which makes the globally cheapest pair mutual, so every
// round merges at least once.  Node ids follow the order of creation (prefix sums, no
// atomics): deterministic, and nodes of one round are contiguous in Morton order.
// A cluster is two float4: (lo.xyz, ref) and (hi.xyz, depth of its subtree).
__global__ void k_ploc_init(const Box* boxes, const uint32_t* sorted, int n, float4* lo, float4* hi) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    Box b = boxes[sorted[j]];
    lo[j] = make_float4(b.lo[0], b.lo[1], b.lo[2], __int_as_float(~j));
    hi[j] = make_float4(b.hi[0], b.hi[1], b.hi[2], __int_as_float(0));
}

__global__ void k_ploc_nn(const float4* __restrict__ lo, const float4* __restrict__ hi, int n, int radius, int* nn) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 l = lo[i], h = hi[i];
    unsigned long long best = ~0ull;
    const int j0 = max(0, i - radius), j1 = min(n - 1, i + radius);
    for (int j = j0; j <= j1; j++) {
        if (j == i) continue;
        const float4 a = lo[j], b = hi[j];
        float dx = fmaxf(h.x, b.x) - fminf(l.x, a.x);
        float dy = fmaxf(h.y, b.y) - fminf(l.y, a.y);
        float dz = fmaxf(h.z, b.z) - fminf(l.z, a.z);
        float area = dx * dy + dy * dz + dz * dx;       // >= 0 (or NaN): its bit pattern orders like the value
        unsigned long long key = ((unsigned long long)__float_as_uint(area) << 32) | (unsigned)j;
        best = key < best ? key : best;
    }
    nn[i] = (int)(unsigned)best;
}

// flags[i] = (survives the round) | (is the left partner of a merge) << 32
__global__ void k_ploc_flag(const int* nn, int n, unsigned long long* flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int j = nn[i];
    bool mutual = nn[j] == i;
    unsigned long long valid = !(mutual && i > j), merge = (mutual && i < j);
    flags[i] = valid | (merge << 32);
}

__global__ void k_ploc_merge(const float4* lo, const float4* hi, const int* nn, const unsigned long long* flags,
                             const unsigned long long* pref, int n, int node_base, float4* nodes,
                             int* parent, int* inner, float4* lo2, float4* hi2) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long f = flags[i];
    if (!(f & 1ull)) return;
    const unsigned long long p = pref[i];
    const int pos = (int)(unsigned)p;
    float4 l = lo[i], h = hi[i];
    if (f >> 32) {
        const int k = node_base + (int)(p >> 32);
        const int j = nn[i];
        const float4 a = lo[j], b = hi[j];
        float4* nd = nodes + 4 * (size_t)k;
        nd[0] = make_float4(l.x, h.x, l.y, h.y);
        nd[1] = make_float4(l.z, h.z, a.x, b.x);
        nd[2] = make_float4(a.y, b.y, a.z, b.z);
        nd[3] = make_float4(l.w, a.w, 0.0f, 0.0f);                       // the two child references
        const int c0 = __float_as_int(l.w), c1 = __float_as_int(a.w);
        int below = 1;                                                    // inner nodes of this subtree
        if (c0 >= 0) { parent[c0] = 2 * k; below += inner[c0]; }
        if (c1 >= 0) { parent[c1] = 2 * k + 1; below += inner[c1]; }
        inner[k] = below;
        parent[k] = -1;                                                   // until a later round adopts it
        int depth = max(__float_as_int(h.w), __float_as_int(b.w)) + 1;
        l = make_float4(fminf(l.x, a.x), fminf(l.y, a.y), fminf(l.z, a.z), __int_as_float(k));
        h = make_float4(fmaxf(h.x, b.x), fmaxf(h.y, b.y), fmaxf(h.z, b.z), __int_as_float(depth));
    }
    lo2[pos] = l;
    hi2[pos] = h;
}

// Depth-first (pre-order) position of every node: root 0, a left child right after its
// parent, a right child after the whole left subtree.  A walk that descends reads forward
// through memory, which is what keeps a tree far larger than the caches fast; the ids the
// rounds hand out put the nodes of one subtree a round apart instead.
__global__ void k_ploc_dfs(const float4* nodes, const int* parent, const int* inner, int m, int* pos) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m) return;
    int at = 0;
    for (int link = parent[v]; link >= 0; ) {
        const int a = link >> 1;
        at += 1;
        if (link & 1) {                                   // came up from the right: skip the left sibling's subtree
            const int left = __float_as_int(nodes[4 * (size_t)a + 3].x);
            if (left >= 0) at += inner[left];
        }
        link = parent[a];
    }
    pos[v] = at;
}
__global__ void k_ploc_place(const float4* nodes, const int* pos, int m, float4* out) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m) return;
    const float4* src = nodes + 4 * (size_t)v;
    float4 refs = src[3];
    int c0 = __float_as_int(refs.x), c1 = __float_as_int(refs.y);
    if (c0 >= 0) c0 = pos[c0];
    if (c1 >= 0) c1 = pos[c1];
    float4* dst = out + 4 * (size_t)pos[v];
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    dst[3] = make_float4(__int_as_float(c0), __int_as_float(c1), 0.0f, 0.0f);
}

// boxes / sorted: padded primitive boxes and the Morton order.  Fills nodes[n-1] (root = node 0), depth.
static int ploc_build(hipStream_t st, const Box* boxes, const uint32_t* sorted, int n, int radius, float4* nodes_out,
                      int* root, int* depth) {
    float4 *lo[2] = {nullptr, nullptr}, *hi[2] = {nullptr, nullptr}, *nodes = nullptr;
    int *nn = nullptr, *parent = nullptr, *inner = nullptr, *pos = nullptr;
    unsigned long long *flags = nullptr, *pref = nullptr; void* tmp = nullptr;
    auto cleanup = [&]() {
        void* ptrs[] = {lo[0], lo[1], hi[0], hi[1], nodes, nn, parent, inner, pos, flags, pref, tmp};
        for (void* p : ptrs) (void)hipFree(p);
    };
#define P_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return (int)e_; } } while (0)
    for (int k = 0; k < 2; k++) { P_TRY(hipMalloc(&lo[k], (size_t)n * sizeof(float4))); P_TRY(hipMalloc(&hi[k], (size_t)n * sizeof(float4))); }
    P_TRY(hipMalloc(&nn, (size_t)n * sizeof(int)));
    P_TRY(hipMalloc(&nodes, (size_t)(n - 1) * 4 * sizeof(float4)));
    P_TRY(hipMalloc(&parent, (size_t)n * sizeof(int))); P_TRY(hipMalloc(&inner, (size_t)n * sizeof(int)));
    P_TRY(hipMalloc(&pos, (size_t)n * sizeof(int)));
    P_TRY(hipMalloc(&flags, (size_t)n * 8)); P_TRY(hipMalloc(&pref, (size_t)n * 8));
    P_TRY(hipMalloc(&tmp, rts_scan_temp_bytes<unsigned long long>((size_t)n)));
    hipLaunchKernelGGL(k_ploc_init, dim3((n + 255) / 256), dim3(256), 0, st, boxes, sorted, n, lo[0], hi[0]);
    int count = n, node_base = 0, cur = 0, rounds = 0;
    while (count > 1) {
        if (++rounds > 4096) { cleanup(); return (int)hipErrorUnknown; }
        dim3 grid((count + 255) / 256), block(256);
        hipLaunchKernelGGL(k_ploc_nn, grid, block, 0, st, lo[cur], hi[cur], count, radius, nn);
        hipLaunchKernelGGL(k_ploc_flag, grid, block, 0, st, nn, count, flags);
        P_TRY(rts_exclusive_scan<unsigned long long>(tmp, flags, pref, (size_t)count, st));
        hipLaunchKernelGGL(k_ploc_merge, grid, block, 0, st, lo[cur], hi[cur], nn, flags, pref, count, node_base, nodes,
                           parent, inner, lo[cur ^ 1], hi[cur ^ 1]);
        unsigned long long last[2];
        P_TRY(hipMemcpyAsync(&last[0], flags + (count - 1), 8, hipMemcpyDeviceToHost, st));
        P_TRY(hipMemcpyAsync(&last[1], pref + (count - 1), 8, hipMemcpyDeviceToHost, st));
        P_TRY(hipStreamSynchronize(st));
        unsigned long long tot = last[0] + last[1];
        int survivors = (int)(unsigned)tot, merges = (int)(tot >> 32);
        if (merges < 1 || survivors != count - merges) { cleanup(); return (int)hipErrorUnknown; }
        node_base += merges;
        count = survivors;
        cur ^= 1;
    }
    float4 top[2];
    P_TRY(hipMemcpyAsync(&top[0], lo[cur], sizeof(float4), hipMemcpyDeviceToHost, st));
    P_TRY(hipMemcpyAsync(&top[1], hi[cur], sizeof(float4), hipMemcpyDeviceToHost, st));
    P_TRY(hipStreamSynchronize(st));
    int r, dpt;
    memcpy(&r, &top[0].w, 4); memcpy(&dpt, &top[1].w, 4);
    if (node_base != n - 1 || r != n - 2) { cleanup(); return (int)hipErrorUnknown; }   // the last node made is the root
    hipLaunchKernelGGL(k_ploc_dfs, dim3((n - 1 + 255) / 256), dim3(256), 0, st, nodes, parent, inner, n - 1, pos);
    hipLaunchKernelGGL(k_ploc_place, dim3((n - 1 + 255) / 256), dim3(256), 0, st, nodes, pos, n - 1, nodes_out);
    P_TRY(hipGetLastError());
    P_TRY(hipStreamSynchronize(st));
    *root = 0; *depth = dpt;
    cleanup();
#undef P_TRY
    return (int)hipSuccess;
}

// ---------------------------------------------------------------- binned surface-area build
// Top-down build by the surface-area heuristic with 16 bins per axis (Wald 2007), for scenes small enough that a
// few milliseconds at load time buy the cheapest tree to walk: every frame of a scene of 10^4 .. 10^5 primitives is
// bound by the number of node and primitive tests per ray, and a greedy top-down split chooses exactly for that.
// One launch per level of the tree; a WAVE per node: its lanes stride over the node's primitives (Morton slots, held
// in two index arrays that alternate by level), bin their box centres along the three axes (boxes and counts per bin
// in LDS, atomics), 45 lanes price the 3 x 15 candidate planes, and the node's primitives are partitioned by the
// cheapest one (ballot + prefix counts).  Leaves hold one primitive, as in the other hierarchies, and share their
// leaf records (reference = ~Morton slot).  Node ids are pre-order positions -- a node of k primitives has k - 1
// inner nodes below it, so left child = id + 1, right child = id + (primitives on the left) -- which is also the
// memory order a descending walk wants.  Child boxes are unions of the padded primitive boxes: conservative like
// every other tree here.  Centres that all fall into one bin (coincident primitives) are split by index.
#define SAH_BINS 16
struct SahNode { int b, e, id; };

__device__ __forceinline__ float wave_min_f(float v) { for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off)); return v; }
__device__ __forceinline__ float wave_max_f(float v) { for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off)); return v; }
__device__ __forceinline__ int wave_max_i(int v) { for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off)); return v; }

__global__ __launch_bounds__(256) void k_sah_level(const Box* __restrict__ sbox /* boxes in Morton-slot order */, const uint32_t* __restrict__ idx_in,
                                                   uint32_t* __restrict__ idx_out, const SahNode* __restrict__ todo, int ntodo,
                                                   SahNode* __restrict__ next, int* next_count, float4* __restrict__ nodes) {
    __shared__ uint32_t s_box[4][3][SAH_BINS][6];
    __shared__ uint32_t s_cnt[4][3][SAH_BINS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    uint32_t (*bbox)[SAH_BINS][6] = s_box[wv];
    uint32_t (*bcnt)[SAH_BINS] = s_cnt[wv];
    for (int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; t < ntodo; t += nwaves) {
        const SahNode nd = todo[t];
        const int b = nd.b, e = nd.e, cnt = nd.e - nd.b;
        // ---- bounds of the box centres
        float cmin[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, cmax[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        for (int i = b + lane; i < e; i += 64) {
            const Box bx = sbox[idx_in[i]];
            for (int k = 0; k < 3; k++) {
                const float c = 0.5f * (bx.lo[k] + bx.hi[k]);
                if (c == c) { cmin[k] = fminf(cmin[k], c); cmax[k] = fmaxf(cmax[k], c); }
            }
        }
        float scale[3];
        for (int k = 0; k < 3; k++) {
            cmin[k] = wave_min_f(cmin[k]); cmax[k] = wave_max_f(cmax[k]);
            const float ext = cmax[k] - cmin[k];
            scale[k] = (ext > 0.0f && ext < __builtin_inff()) ? (float)SAH_BINS * 0.99999f / ext : 0.0f;
        }
        // ---- binning
        for (int q = lane; q < 3 * SAH_BINS; q += 64) {
            uint32_t* bb = bbox[q / SAH_BINS][q % SAH_BINS];
            bb[0] = bb[1] = bb[2] = 0xFFFFFFFFu; bb[3] = bb[4] = bb[5] = 0u;
            bcnt[q / SAH_BINS][q % SAH_BINS] = 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        auto bin_of = [&](const Box& bx, int k) {
            const float c = 0.5f * (bx.lo[k] + bx.hi[k]);
            int q = (c == c) ? (int)((c - cmin[k]) * scale[k]) : 0;
            return q < 0 ? 0 : (q > SAH_BINS - 1 ? SAH_BINS - 1 : q);
        };
        for (int i = b + lane; i < e; i += 64) {
            const Box bx = sbox[idx_in[i]];
            for (int k = 0; k < 3; k++) {
                const int q = bin_of(bx, k);
                uint32_t* bb = bbox[k][q];
                for (int a = 0; a < 3; a++) {
                    if (bx.lo[a] == bx.lo[a]) atomicMin(&bb[a], f2ord(bx.lo[a]));
                    if (bx.hi[a] == bx.hi[a]) atomicMax(&bb[3 + a], f2ord(bx.hi[a]));
                }
                atomicAdd(&bcnt[k][q], 1u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- price the 3 x 15 planes: lane = axis * 15 + plane (left = bins 0 .. plane)
        float cost = __builtin_inff();
        Box L, R;
        int nL = 0;
        for (int a = 0; a < 3; a++) { L.lo[a] = R.lo[a] = __builtin_inff(); L.hi[a] = R.hi[a] = -__builtin_inff(); }
        if (lane < 3 * (SAH_BINS - 1)) {
            const int k = lane / (SAH_BINS - 1), pl = lane % (SAH_BINS - 1);
            int nR = 0;
            for (int q = 0; q < SAH_BINS; q++) {
                const uint32_t c = bcnt[k][q];
                if (!c) continue;
                Box& T = q <= pl ? L : R;
                (q <= pl ? nL : nR) += (int)c;
                for (int a = 0; a < 3; a++) {
                    if (bbox[k][q][a] != 0xFFFFFFFFu) T.lo[a] = fminf(T.lo[a], ord2f(bbox[k][q][a]));
                    if (bbox[k][q][3 + a] != 0u) T.hi[a] = fmaxf(T.hi[a], ord2f(bbox[k][q][3 + a]));
                }
            }
            auto area = [](const Box& T) {
                const float dx = T.hi[0] - T.lo[0], dy = T.hi[1] - T.lo[1], dz = T.hi[2] - T.lo[2];
                return (dx >= 0.0f && dy >= 0.0f && dz >= 0.0f) ? dx * dy + dy * dz + dz * dx : 0.0f;
            };
            if (nL > 0 && nR > 0) {
                cost = area(L) * (float)nL + area(R) * (float)nR;
                if (!(cost == cost)) cost = 3.0e38f;               // boxes with infinite faces: any plane that separates does
            }
        }
        // cheapest plane, ties to the lowest lane
        unsigned long long key = ((unsigned long long)__float_as_uint(cost) << 32) | (unsigned)lane;    // cost >= 0: bit order = value order
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(key, off); key = o < key ? o : key; }
        const int win = (int)(unsigned)key;
        const bool by_index = !(__uint_as_float((uint32_t)(key >> 32)) < __builtin_inff());    // no plane separates the centres
        const int axis = by_index ? 0 : win / (SAH_BINS - 1), plane = by_index ? 0 : win % (SAH_BINS - 1);
        int nl = by_index ? cnt / 2 : __shfl(nL, win);
        // ---- partition; the winner's child boxes travel from its lane, or are gathered again for a split by index
        int offL = 0, offR = 0, firstL = -1, firstR = -1;
        Box UL, UR;
        for (int a = 0; a < 3; a++) { UL.lo[a] = UR.lo[a] = __builtin_inff(); UL.hi[a] = UR.hi[a] = -__builtin_inff(); }
        for (int base = b; base < e; base += 64) {
            const int i = base + lane;
            const bool in = i < e;
            uint32_t slot = 0; bool left = false;
            Box bx;
            if (in) {
                slot = idx_in[i];
                bx = sbox[slot];
                left = by_index ? (i - b) < nl : bin_of(bx, axis) <= plane;
            }
            const unsigned long long ml = __builtin_amdgcn_ballot_w64(in && left), mr = __builtin_amdgcn_ballot_w64(in && !left);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (in) {
                const int dst = left ? b + offL + __popcll(ml & below) : b + nl + offR + __popcll(mr & below);
                idx_out[dst] = slot;
                if (dst == b) firstL = (int)slot;
                if (dst == b + nl) firstR = (int)slot;
                if (by_index) {
                    Box& T = left ? UL : UR;
                    for (int a = 0; a < 3; a++) { T.lo[a] = fminf(T.lo[a], bx.lo[a]); T.hi[a] = fmaxf(T.hi[a], bx.hi[a]); }
                }
            }
            offL += __popcll(ml); offR += __popcll(mr);
        }
        firstL = wave_max_i(firstL); firstR = wave_max_i(firstR);
        float f[12];
        for (int a = 0; a < 3; a++) {
            if (by_index) {
                f[2 * a] = wave_min_f(UL.lo[a]); f[2 * a + 1] = wave_max_f(UL.hi[a]);
                f[6 + 2 * a] = wave_min_f(UR.lo[a]); f[6 + 2 * a + 1] = wave_max_f(UR.hi[a]);
            } else {
                f[2 * a] = __shfl(L.lo[a], win); f[2 * a + 1] = __shfl(L.hi[a], win);
                f[6 + 2 * a] = __shfl(R.lo[a], win); f[6 + 2 * a + 1] = __shfl(R.hi[a], win);
            }
        }
        if (lane == 0) {
            const int nr = cnt - nl;
            const int refL = nl == 1 ? ~firstL : nd.id + 1, refR = nr == 1 ? ~firstR : nd.id + nl;
            float4* dst = nodes + 4 * (size_t)nd.id;
            dst[0] = make_float4(f[0], f[1], f[2], f[3]);
            dst[1] = make_float4(f[4], f[5], f[6], f[7]);
            dst[2] = make_float4(f[8], f[9], f[10], f[11]);
            dst[3] = make_float4(__int_as_float(refL), __int_as_float(refR), 0.0f, 0.0f);
            const int kids = (nl > 1 ? 1 : 0) + (nr > 1 ? 1 : 0);
            if (kids) {
                int at = atomicAdd(next_count, kids);
                if (nl > 1) next[at++] = SahNode{b, b + nl, nd.id + 1};
                if (nr > 1) next[at] = SahNode{b + nl, e, nd.id + nl};
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // the bins are rewritten for the wave's next node
    }
}

__global__ void k_sah_gather(const Box* boxes, const uint32_t* sorted, int n, Box* sbox, uint32_t* idx) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    sbox[j] = boxes[sorted[j]];
    idx[j] = (uint32_t)j;
}

// boxes / sorted: padded primitive boxes and the Morton order.  Fills nodes_out[n-1] (root = node 0), *depth = levels.
// Returns a hipError_t as int; hipErrorUnknown for a tree deeper than max_depth (the caller then does without it).
static int sah_build(hipStream_t st, const Box* boxes, const uint32_t* sorted, int n, int max_depth, float4* nodes_out, int* depth) {
    Box* sbox = nullptr; uint32_t* idx[2] = {nullptr, nullptr}; SahNode* list[2] = {nullptr, nullptr}; int* counter = nullptr;
    auto cleanup = [&]() { void* ptrs[] = {sbox, idx[0], idx[1], list[0], list[1], counter}; for (void* p : ptrs) (void)hipFree(p); };
#define S_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return (int)e_; } } while (0)
    S_TRY(hipMalloc(&sbox, (size_t)n * sizeof(Box)));
    for (int k = 0; k < 2; k++) { S_TRY(hipMalloc(&idx[k], (size_t)n * 4)); S_TRY(hipMalloc(&list[k], (size_t)n * sizeof(SahNode))); }
    S_TRY(hipMalloc(&counter, 4));
    hipLaunchKernelGGL(k_sah_gather, dim3((n + 255) / 256), dim3(256), 0, st, boxes, sorted, n, sbox, idx[0]);
    const SahNode root{0, n, 0};
    S_TRY(hipMemcpyAsync(list[0], &root, sizeof(root), hipMemcpyHostToDevice, st));
    S_TRY(hipStreamSynchronize(st));                         // root is a local
    int todo = 1, level = 0;
    while (todo > 0) {
        if (level >= max_depth) { cleanup(); return (int)hipErrorUnknown; }
        S_TRY(hipMemsetAsync(counter, 0, 4, st));
        const int blocks = std::min(1024, (todo + 3) / 4);
        hipLaunchKernelGGL(k_sah_level, dim3(blocks), dim3(256), 0, st, (const Box*)sbox, (const uint32_t*)idx[level & 1], idx[(level + 1) & 1],
                           (const SahNode*)list[level & 1], todo, list[(level + 1) & 1], counter, nodes_out);
        S_TRY(hipGetLastError());
        int got = 0;
        S_TRY(hipMemcpyAsync(&got, counter, 4, hipMemcpyDeviceToHost, st));
        S_TRY(hipStreamSynchronize(st));
        if (got < 0 || got > n) { cleanup(); return (int)hipErrorUnknown; }
        todo = got;
        level++;
    }
    *depth = level;
    cleanup();
#undef S_TRY
    return (int)hipSuccess;
}

// ---------------------------------------------------------------- 4-wide collapse
// A per-lane walk is a chain of dependent fetches; folding every second level of the binary
// tree into its parent halves the chain.  Inner nodes at even depth survive; each takes the
// children of its (odd-depth) inner children, or the child itself where that is a leaf: up
// to four entries.  Ids are the binary ids compacted (prefix sum), so the layout keeps its locality.
//
// A wide node is ONE 64-byte record (4 float4): the four child boxes are stored as 8-bit offsets on a
// grid laid over the node's own box -- origin = the low corner of the union of the children, one
// power-of-two step per axis -- so a step of the walk moves 64 bytes instead of 128 (the walks of scenes
// far larger than the caches are bound by exactly those bytes):
//   q0 = (origin.x, origin.y, origin.z, bits: biased exponent of the x | y << 8 | z << 16 step)
//   q1 = bits (lo.x of the 4 children, one byte each) (lo.y) (lo.z) (hi.x)
//   q2 = bits (hi.y) (hi.z), ref0, ref1          q3 = ref2, ref3, -, -
// ref >= 0 wide node, < 0 leaf ~slot, RT_REF_NONE empty slot (its box is inverted: lo 255, hi 0).
// Quantisation only ever GROWS a box: lo is rounded down and moved one more step down, hi up and one more
// up (clamped to the grid, whose ends are the union's own faces or beyond), which also covers the
// rounding of the walk's decode; the exact primitive tests decide every hit, so the image cannot change.
#define RT_REF_NONE ((int)0x80000000)

__global__ void k_w4_parent(const float4* nodes, int m, int* parent) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m) return;
    float4 refs = nodes[4 * (size_t)v + 3];
    int c0 = __float_as_int(refs.x), c1 = __float_as_int(refs.y);
    if (c0 >= 0) parent[c0] = v;
    if (c1 >= 0) parent[c1] = v;
}
__global__ void k_w4_flag(const int* parent, int m, int root, uint32_t* flag) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m) return;
    int d = 0;
    for (int a = v; a != root; a = parent[a]) d++;
    flag[v] = (d & 1) ? 0u : 1u;
}
__device__ inline void w4_entry(float* dst, int* refs, int k, const float* box, int ref) {
    for (int t = 0; t < 6; t++) dst[6 * k + t] = box[t];
    refs[k] = ref;
}
// The plain form of a wide node, for trees that fit the caches (there a step of the walk is bound by latency
// and the arithmetic of decoding bytes only lengthens it): 8 float4 = the 4 boxes as (lo.x, hi.x, lo.y, hi.y,
// lo.z, hi.z), then the 4 references.
__global__ void k_w4_collapse_exact(const float4* nodes, int m, const uint32_t* flag, const uint32_t* idx, float4* out) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m || !flag[v]) return;
    const float* src = (const float*)(nodes + 4 * (size_t)v);
    float box[24]; int refs[4];
    int k = 0;
    for (int s = 0; s < 2; s++) {
        int c = __float_as_int(src[12 + s]);
        if (c < 0) { w4_entry(box, refs, k++, src + 6 * s, c); continue; }
        const float* cs = (const float*)(nodes + 4 * (size_t)c);
        for (int t = 0; t < 2; t++) {
            int g = __float_as_int(cs[12 + t]);
            w4_entry(box, refs, k++, cs + 6 * t, g < 0 ? g : (int)idx[g]);
        }
    }
    const float none[6] = {__builtin_inff(), -__builtin_inff(), __builtin_inff(), -__builtin_inff(), __builtin_inff(), -__builtin_inff()};
    for (; k < 4; k++) w4_entry(box, refs, k, none, RT_REF_NONE);
    float4* dst = out + 8 * (size_t)idx[v];
    for (int q = 0; q < 6; q++) dst[q] = make_float4(box[4 * q], box[4 * q + 1], box[4 * q + 2], box[4 * q + 3]);
    dst[6] = make_float4(__int_as_float(refs[0]), __int_as_float(refs[1]), __int_as_float(refs[2]), __int_as_float(refs[3]));
    dst[7] = make_float4(0, 0, 0, 0);
}

__global__ void k_w4_collapse(const float4* nodes, int m, const uint32_t* flag, const uint32_t* idx, float4* out) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m || !flag[v]) return;
    const float* src = (const float*)(nodes + 4 * (size_t)v);
    float box[24]; int refs[4];       // per child (lo.x, hi.x, lo.y, hi.y, lo.z, hi.z)
    int k = 0;
    for (int s = 0; s < 2; s++) {
        int c = __float_as_int(src[12 + s]);
        if (c < 0) { w4_entry(box, refs, k++, src + 6 * s, c); continue; }
        const float* cs = (const float*)(nodes + 4 * (size_t)c);
        for (int t = 0; t < 2; t++) {
            int g = __float_as_int(cs[12 + t]);
            w4_entry(box, refs, k++, cs + 6 * t, g < 0 ? g : (int)idx[g]);
        }
    }
    const int nvalid = k;
    for (; k < 4; k++) refs[k] = RT_REF_NONE;
    uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0}, ebits = 0;
    float origin[3];
    for (int a = 0; a < 3; a++) {
        float lo = box[2 * a], hi = box[2 * a + 1];
        for (int c = 1; c < nvalid; c++) { lo = fminf(lo, box[6 * c + 2 * a]); hi = fmaxf(hi, box[6 * c + 2 * a + 1]); }
        // step = the smallest power of two with 253 steps >= the extent (two steps of the 255 are the margin), but
        // never below 2^-20 of the coordinates' magnitude: a flat box keeps a margin well above float rounding
        int ex = 0;
        const float mant = frexpf((hi - lo) / 253.0f, &ex);            // extent / 253 = mant * 2^ex, mant in [0.5, 1)
        (void)mant;
        int em = 0;
        (void)frexpf(fmaxf(fabsf(lo), fabsf(hi)), &em);
        if (hi - lo == 0.0f) ex = -100;                                  // (frexp of zero reports exponent 0)
        ex = max(ex, em - 20);
        ex = max(-100, min(100, ex));
        if (!(hi - lo >= 0.0f) || !(fabsf(lo) < __builtin_inff()) || !(fabsf(hi) < __builtin_inff())) ex = 100;   // NaN / infinite boxes: everything maps to the grid's ends
        const float step = ldexpf(1.0f, ex);
        origin[a] = lo;
        ebits |= (uint32_t)(ex + 127) << (8 * a);
        for (int c = 0; c < 4; c++) {
            uint32_t l = 255u, h = 0u;                                   // empty slot: inverted
            if (c < nvalid) {
                const float fl = floorf((box[6 * c + 2 * a] - lo) / step) - 1.0f;
                const float fh = ceilf((box[6 * c + 2 * a + 1] - lo) / step) + 1.0f;
                l = (uint32_t)fminf(fmaxf(fl, 0.0f), 255.0f);
                h = (uint32_t)fminf(fmaxf(fh, 0.0f), 255.0f);
                if (!(fl == fl)) l = 0u;                                 // NaN coordinates: the whole grid
                if (!(fh == fh)) h = 255u;
            }
            qlo[a] |= l << (8 * c);
            qhi[a] |= h << (8 * c);
        }
    }
    float4* dst = out + 4 * (size_t)idx[v];
    dst[0] = make_float4(origin[0], origin[1], origin[2], __uint_as_float(ebits));
    dst[1] = make_float4(__uint_as_float(qlo[0]), __uint_as_float(qlo[1]), __uint_as_float(qlo[2]), __uint_as_float(qhi[0]));
    dst[2] = make_float4(__uint_as_float(qhi[1]), __uint_as_float(qhi[2]), __int_as_float(refs[0]), __int_as_float(refs[1]));
    dst[3] = make_float4(__int_as_float(refs[2]), __int_as_float(refs[3]), 0.0f, 0.0f);
}

// nodes: binary tree of m inner nodes, root 0.  -> *out (root 0; 4 float4 per wide node when quantised, else 8), or nullptr on failure.
static void collapse4(hipStream_t st, const float4* nodes, int m, bool quantised, float4** out) {
    *out = nullptr;
    int* parent = nullptr; uint32_t *flag = nullptr, *idx = nullptr; void* tmp = nullptr; float4* wide = nullptr;
    auto cleanup = [&]() { (void)hipFree(parent); (void)hipFree(flag); (void)hipFree(idx); (void)hipFree(tmp); };
    auto fail = [&]() { cleanup(); (void)hipFree(wide); (void)hipGetLastError(); };
    if (hipMalloc(&parent, (size_t)m * 4) != hipSuccess || hipMalloc(&flag, (size_t)m * 4) != hipSuccess ||
        hipMalloc(&idx, (size_t)m * 4) != hipSuccess) return fail();
    dim3 grid((m + 255) / 256), block(256);
    hipLaunchKernelGGL(k_w4_parent, grid, block, 0, st, nodes, m, parent);
    hipLaunchKernelGGL(k_w4_flag, grid, block, 0, st, parent, m, 0, flag);
    if (hipMalloc(&tmp, rts_scan_temp_bytes<uint32_t>((size_t)m)) != hipSuccess) return fail();
    if (rts_exclusive_scan<uint32_t>(tmp, flag, idx, (size_t)m, st) != hipSuccess) return fail();
    uint32_t last[2] = {0, 0};
    if (hipMemcpyAsync(&last[0], flag + (m - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&last[1], idx + (m - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return fail();
    const size_t m4 = (size_t)last[0] + last[1];
    if (m4 < 1 || hipMalloc(&wide, m4 * (quantised ? 4 : 8) * sizeof(float4)) != hipSuccess) return fail();
    if (quantised) hipLaunchKernelGGL(k_w4_collapse, grid, block, 0, st, nodes, m, flag, idx, wide);
    else hipLaunchKernelGGL(k_w4_collapse_exact, grid, block, 0, st, nodes, m, flag, idx, wide);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail();
    cleanup();
    *out = wide;
}

// ---------------------------------------------------------------- tree quality
// Surface-area cost of a binary hierarchy: a ray that crosses the root box crosses a convex box inside it with
// probability area(box) / area(root) (for uniformly distributed lines), so
//   out[0] = sum over inner nodes of area / area(root)   = expected node visits (box-pair tests) of such a ray,
//   out[1] = sum over leaves of area / area(root)        = expected exact primitive tests,
//   out[2] = area(root).
// The two sums are what a better tree lowers; walks that stop early (any-hit, a shrinking distance bound) see less.
__device__ inline double box_area6(const float* b) {      // (lo.x, hi.x, lo.y, hi.y, lo.z, hi.z)
    const double dx = (double)b[1] - (double)b[0], dy = (double)b[3] - (double)b[2], dz = (double)b[5] - (double)b[4];
    if (!(dx >= 0.0) || !(dy >= 0.0) || !(dz >= 0.0)) return 0.0;
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}
__global__ void k_tree_cost(const float4* nodes, int m, double* acc /*[3]: inner, leaf, root (root written by node 0)*/) {
    __shared__ double red[2][256];
    double inner = 0.0, leaf = 0.0;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < m; v += gridDim.x * blockDim.x) {
        const float* f = (const float*)(nodes + 4 * (size_t)v);
        for (int s = 0; s < 2; s++) {
            const double a = box_area6(f + 6 * s);
            if (__float_as_int(f[12 + s]) >= 0) inner += a; else leaf += a;
        }
        if (v == 0) {
            float u[6];
            for (int k = 0; k < 3; k++) { u[2 * k] = fminf(f[2 * k], f[6 + 2 * k]); u[2 * k + 1] = fmaxf(f[2 * k + 1], f[6 + 2 * k + 1]); }
            const double a = box_area6(u);
            acc[2] = a;
            inner += a;
        }
    }
    red[0][threadIdx.x] = inner; red[1][threadIdx.x] = leaf;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(&acc[0], red[0][0]); atomicAdd(&acc[1], red[1][0]); }
}

// nodes: binary tree of m inner nodes, root 0.  out[0] expected node visits, out[1] expected primitive tests (see above).
int rtl_tree_cost(hipStream_t st, const float4* nodes, int m, double out[2]) {
    out[0] = out[1] = 0.0;
    if (!nodes || m < 1) return (int)hipSuccess;
    double* acc = nullptr;
    hipError_t e = hipMalloc(&acc, 3 * sizeof(double));
    if (e != hipSuccess) return (int)e;
    double h[3] = {0, 0, 0};
    if ((e = hipMemsetAsync(acc, 0, 3 * sizeof(double), st)) == hipSuccess) {
        const int blocks = (m + 255) / 256;
        hipLaunchKernelGGL(k_tree_cost, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, st, nodes, m, acc);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h, acc, sizeof(h), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(acc);
    if (e == hipSuccess && h[2] > 0.0) { out[0] = h[0] / h[2]; out[1] = h[1] / h[2]; }
    return (int)e;
}

#define L_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); (void)hipFree(nodes); (void)hipFree(vals2); (void)hipFree(leaf); return (int)e_; } } while (0)

// Builds the Karras hierarchy (nodes_out, at most 95 deep; wide_out = its 4-wide collapse or nullptr),
// when ploc_radius > 0 a PLOC hierarchy over the same Morton order (ploc_out; nullptr if it failed or came out deeper
// than the 64-entry stacks of the fast walks allow) and, for scenes of at most sah_max_prims primitives, a binned
// surface-area build (sah_out, same conditions).  All share the leaf records and have
// their root at node 0; which one traverses faster depends on the scene and is measured
// by the caller.
int rtl_build(hipStream_t st, const float4* sph, int ns, const float4* tri, int nt, float4** nodes_out,
              int** prim_idx_out, float4** leaf_out, int* depth, int ploc_radius, float4** ploc_out,
              int* ploc_depth, float4** wide_out, float4** ploc_wide_out, int wide_quantised, const float* origin_box,
              float* region_out, int sah_max_prims, float4** sah_out, int* sah_depth, float4** sah_wide_out) {
    int n = ns + nt;
    const bool verbose = getenv("RT_VERBOSE") != nullptr;
    auto now_ms = [t_start = std::chrono::steady_clock::now()]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    double t_mark = now_ms();
    auto lap = [&](const char* what) { if (verbose) { const double t = now_ms(); fprintf(stderr, "[rt bvh build] %s %.2f ms\n", what, t - t_mark); t_mark = t; } };
    *nodes_out = nullptr; *prim_idx_out = nullptr; *leaf_out = nullptr; *ploc_out = nullptr; *ploc_depth = 0;
    *wide_out = nullptr; *ploc_wide_out = nullptr;
    *sah_out = nullptr; *sah_depth = 0; *sah_wide_out = nullptr;
    if (n < 2) return (int)hipErrorInvalidValue;
    Box* boxes = nullptr; uint32_t* scene = nullptr;   // scene[0..5] centre bounds, [6] tree depth, [8..13] region of ray origins
    unsigned long long *keys = nullptr, *keys2 = nullptr; uint32_t *vals = nullptr, *vals2 = nullptr;
    int *child = nullptr, *pin = nullptr, *plf = nullptr; uint32_t* flags = nullptr; void* tmp = nullptr;
    float4* nodes = nullptr; float4* leaf = nullptr;
    auto cleanup = [&]() {
        void* ptrs[] = {boxes, scene, keys, keys2, vals, child, pin, plf, flags, tmp};
        for (void* p : ptrs) (void)hipFree(p);
    };
    L_TRY(hipMalloc(&boxes, (size_t)n * sizeof(Box)));
    L_TRY(hipMalloc(&scene, 16 * sizeof(uint32_t)));
    L_TRY(hipMalloc(&keys, (size_t)n * 8)); L_TRY(hipMalloc(&keys2, (size_t)n * 8));
    L_TRY(hipMalloc(&vals, (size_t)n * 4)); L_TRY(hipMalloc(&vals2, (size_t)n * 4));
    L_TRY(hipMalloc(&child, (size_t)(n - 1) * 2 * sizeof(int)));
    L_TRY(hipMalloc(&pin, (size_t)(n - 1) * sizeof(int)));
    L_TRY(hipMalloc(&plf, (size_t)n * sizeof(int)));
    L_TRY(hipMalloc(&flags, (size_t)(n - 1) * sizeof(uint32_t)));
    L_TRY(hipMalloc(&nodes, (size_t)(n - 1) * 4 * sizeof(float4)));
    L_TRY(hipMalloc(&leaf, (size_t)n * 3 * sizeof(float4)));
    uint32_t init[16] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
    for (int k = 0; k < 6 && origin_box; k++)
        if (origin_box[k] == origin_box[k]) init[8 + k] = f2ord(origin_box[k]);
    L_TRY(hipMemcpyAsync(scene, init, sizeof(init), hipMemcpyHostToDevice, st));
    L_TRY(hipStreamSynchronize(st));                                                // init is a local
    L_TRY(hipMemsetAsync(flags, 0, (size_t)(n - 1) * sizeof(uint32_t), st));
    int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_region, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, st, sph, ns, tri, n, scene + 8);
    hipLaunchKernelGGL(k_bounds, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, st, sph, ns, tri, n, scene + 8, boxes, scene);
    hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(256), 0, st, boxes, n, scene, keys, vals);
    L_TRY(hipMalloc(&tmp, rts_sort_temp_bytes((size_t)n)));
    {
        bool in_alt = false;
        L_TRY(rts_sort_pairs(tmp, keys, keys2, vals, vals2, (size_t)n, 63, st, &in_alt));
        if (!in_alt) { std::swap(keys, keys2); std::swap(vals, vals2); }      // keys2 / vals2 = the sorted pairs from here on
        // A hierarchy over keys that are not sorted can contain cycles, and a walk through a cycle never ends:
        // check the order and that the values are still the ids 0 .. n-1 before anything is built on them.
        unsigned long long* chk = (unsigned long long*)tmp;
        L_TRY(hipMemsetAsync(chk, 0, 16, st));
        hipLaunchKernelGGL(k_rts_check_sorted, dim3(blocks), dim3(256), 0, st, keys2, vals2, (size_t)n, chk);
        unsigned long long got[2] = {1, 0};
        L_TRY(hipMemcpyAsync(got, chk, 16, hipMemcpyDeviceToHost, st));
        L_TRY(hipStreamSynchronize(st));
        if (got[0] != 0 || got[1] != (unsigned long long)n * (unsigned long long)(n - 1) / 2) L_TRY(hipErrorUnknown);
    }
    hipLaunchKernelGGL(k_leaves, dim3(blocks), dim3(256), 0, st, sph, ns, tri, vals2, n, leaf);
    hipLaunchKernelGGL(k_hierarchy, dim3((n - 1 + 255) / 256), dim3(256), 0, st, keys2, n, child, pin, plf);
    hipLaunchKernelGGL(k_refit, dim3(blocks), dim3(256), 0, st, boxes, vals2, n, child, pin, plf, flags, nodes);
    hipLaunchKernelGGL(k_depth, dim3(blocks), dim3(256), 0, st, pin, plf, n, scene + 6);
    uint32_t dpt = 0, reg[6] = {0, 0, 0, 0, 0, 0};
    L_TRY(hipMemcpyAsync(&dpt, scene + 6, sizeof(dpt), hipMemcpyDeviceToHost, st));
    L_TRY(hipMemcpyAsync(reg, scene + 8, sizeof(reg), hipMemcpyDeviceToHost, st));
    L_TRY(hipStreamSynchronize(st));
    *depth = (int)dpt;
    for (int k = 0; k < 6 && region_out; k++) region_out[k] = ord2f(reg[k]);
    lap("boxes + sort + Karras hierarchy");
    if (ploc_radius > 0) {
        float4* pn = nullptr;
        int r = 0, d = 0;
        if (hipMalloc(&pn, (size_t)(n - 1) * 4 * sizeof(float4)) == hipSuccess &&
            ploc_build(st, boxes, vals2, n, ploc_radius, pn, &r, &d) == (int)hipSuccess && d <= 64) {
            *ploc_out = pn; *ploc_depth = d;
            collapse4(st, pn, n - 1, wide_quantised != 0, ploc_wide_out);
        } else {
            (void)hipFree(pn);
            (void)hipGetLastError();
        }
        lap("PLOC + its 4-wide collapse");
    }
    if (n <= sah_max_prims) {
        float4* sn = nullptr;
        int d = 0;
        if (hipMalloc(&sn, (size_t)(n - 1) * 4 * sizeof(float4)) == hipSuccess &&
            sah_build(st, boxes, vals2, n, 64, sn, &d) == (int)hipSuccess) {
            *sah_out = sn; *sah_depth = d;
            collapse4(st, sn, n - 1, wide_quantised != 0, sah_wide_out);
        } else {
            (void)hipFree(sn);
            (void)hipGetLastError();
        }
        lap("binned surface-area build + its 4-wide collapse");
    }
    // 4-wide versions for the per-lane walks (their stacks hold up to 3 entries per wide level: depth <= 64)
    if (*depth <= 64) collapse4(st, nodes, n - 1, wide_quantised != 0, wide_out);
    L_TRY(hipGetLastError());
    L_TRY(hipStreamSynchronize(st));
    lap("4-wide collapse of the Karras hierarchy");
    cleanup();
    *nodes_out = nodes;
    *prim_idx_out = (int*)vals2;   // sorted primitive ids, kept by the caller
    *leaf_out = leaf;
    return (int)hipSuccess;
}
