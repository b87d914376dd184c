// Host-callable launchers of the wavefront kernels (rt_kernels.hip) and of the
// LBVH builder (rt_lbvh.hip).  `bound` is an upper bound on the work items, used
// only to size the persistent grid; the kernels read the true counts on device.
#pragma once

#include "rt_device.h"

// Blocks of each traversal kernel that are resident at once on the context's device, by (counting variant,
// walk): filled on first use from the occupancy query, kept per context because it is a property of the device.
struct RtkLaunchCache {
    int extend[2][6] = {{0}};
    int shadow[2][6] = {{0}};
};

void rtk_init_counters(hipStream_t st, DCounters* C, uint32_t nsamples, const DPool& P, int generations, bool count_work);
// generation 0 is generated on the fly from (cam, tiles, tile0); deeper generations read the pool
void rtk_extend(hipStream_t st, int accel, const DScene& S, const DPool& P, DCounters* C, const DCamera& cam,
                const uint32_t* tiles, uint32_t tile0, int g, uint64_t bound, int max_blocks, bool count,
                RtkLaunchCache& cache, bool defer_miss = false);
// prefix sums over the wave masks of generation g: hit list, hit / next-generation counts, child offsets
void rtk_reduce_stats(hipStream_t st, const DPool& P, DCounters* C);
// extend_accel: the walk of the extend launch before it (2, 4, 5 = per-lane and team walks leave flag bytes, not masks)
void rtk_compact(hipStream_t st, const DPool& P, DCounters* C, int g, uint64_t bound, int max_blocks, int extend_accel);
void rtk_shadow(hipStream_t st, int accel, const DScene& S, const DPool& P, DCounters* C, int g,
                uint64_t bound, int max_blocks, bool count, RtkLaunchCache& cache);
// phase 0: lights + hand-over + spawn; 1: spawn only; 2: lights + hand-over only (overlapped pipeline);
// 3: as 0 with the shadow queries answered inside (no shadow launch before it), for scenes rtk_shade_can_shadow accepts
bool rtk_shade_can_shadow(int ns, int nt);
void rtk_shade(hipStream_t st, const DScene& S, const DPool& P, DCounters* C, int g, uint64_t bound, int max_blocks,
               int phase = 0);
// background for the missed rays of generation g >= 1 that an extend launch with defer_miss left alone
void rtk_miss(hipStream_t st, const DScene& S, const DPool& P, DCounters* C, int g, uint64_t bound, int max_blocks);
void rtk_resolve(hipStream_t st, const DScene& S, const DPool& P, DCounters* C, int g, uint64_t bound,
                 int max_blocks);
void rtk_fill_background(hipStream_t st, const DScene& S, const DPool& P, uint32_t nsamples, int max_blocks);
void rtk_finish(hipStream_t st, const DCamera& cam, const DPool& P, const uint32_t* tiles, uint32_t tile0,
                uint32_t npixels, int shard_mode, uint8_t* out, float* outf, int max_blocks);
void rtk_assemble(hipStream_t st, const uint8_t* all, size_t shard_bytes, int world, int width, int height,
                  int ntx, int nty, const uint32_t* rowbase, uint8_t* out);
// the tile-row table of rtk_assemble, written by a kernel into `rowbase` ([world * nty] uint32, device memory)
void rtk_rowbase(hipStream_t st, int world, int ntx, int nty, uint32_t* rowbase);
void rtk_query(hipStream_t st, int accel, int any_hit, const DScene& S, int64_t n, const float* o,
               const float* d, const float* dmax, int* prim_out, float* d2_out);
void rtk_primary_rays(hipStream_t st, const DCamera& cam, int64_t n, const int* x, const int* y,
                      const int* j, float* o_out, float* d_out);

// LBVH over spheres + triangles (Morton codes -> radix sort -> Karras hierarchy ->
// bottom-up refit).  Allocates *nodes ([n_prims-1][4] float4), *prim_idx ([n_prims])
// and *leaf ([n_prims][3] float4, primitives in tree order) with hipMalloc; returns a hipError_t as int.  n_prims >= 2.
// origin_box (lo.xyz, hi.xyz; may be null): where the camera's rays start.  The leaf boxes are padded by what the
// exact tests can accept beyond the ideal primitive, which grows with the distance to a ray's origin; region_out
// (may be null) receives the box of origin_box and all primitives: the hierarchy is conservative for rays that
// start inside it, and only for those.
int rtl_build(hipStream_t st, const float4* sph, int ns, const float4* tri, int nt, float4** nodes,
              int** prim_idx, float4** leaf, int* depth, int ploc_radius, float4** ploc_nodes, int* ploc_depth,
              float4** wide_nodes, float4** ploc_wide_nodes, int wide_quantised, const float* origin_box,
              float* region_out, int sah_max_prims, float4** sah_nodes, int* sah_depth, float4** sah_wide_nodes);
// Surface-area cost of a binary hierarchy of m inner nodes (root 0): out[0] = expected node visits, out[1] = expected
// exact primitive tests of a random line through the root box.  Returns a hipError_t as int.
int rtl_tree_cost(hipStream_t st, const float4* nodes, int m, double out[2]);
