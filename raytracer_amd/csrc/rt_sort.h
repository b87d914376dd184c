// Device-wide exclusive scan and least-significant-digit radix sort of (64-bit key, 32-bit value) pairs
// for gfx950: what the tree builders need, written for wave64 (ballots over 64 lanes, LDS histograms).
//
// Scan: three launches -- tile sums, one block scanning the sums, tile scans with their offset.
// Sort: eight passes of eight bits; per pass a histogram launch (256 bins per 4096-key tile, LDS atomics),
// a scan of the bin-major histogram table, and a scatter launch that ranks the keys of a tile round by
// round: inside a wave by matching digits with eight ballots (rank = lanes below with the same digit),
// across the four waves of the block through a small LDS table.  Rounds and lanes are visited in index
// order, so every pass is stable, which is what makes digit-by-digit sorting correct.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTS_BLOCK 256
#define RTS_SCAN_ITEMS 8
#define RTS_SCAN_TILE (RTS_BLOCK * RTS_SCAN_ITEMS)
#define RTS_SORT_ROUNDS 16
#define RTS_SORT_TILE (RTS_BLOCK * RTS_SORT_ROUNDS)

// ---------------------------------------------------------------- scan
template <typename T>
__device__ __forceinline__ T rts_block_exclusive(T v, T* lds /*[RTS_BLOCK]*/, T& total) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < RTS_BLOCK; off <<= 1) {
        const T a = lds[t];
        const T b = (t >= off) ? lds[t - off] : (T)0;
        __syncthreads();
        lds[t] = a + b;
        __syncthreads();
    }
    const T incl = lds[t];
    total = lds[RTS_BLOCK - 1];
    __syncthreads();
    return incl - v;
}

template <typename T>
__global__ __launch_bounds__(RTS_BLOCK) void k_rts_tile_sums(const T* in, size_t n, T* sums) {
    __shared__ T lds[RTS_BLOCK];
    const size_t base = (size_t)blockIdx.x * RTS_SCAN_TILE + (size_t)threadIdx.x * RTS_SCAN_ITEMS;
    T v = 0;
    for (int q = 0; q < RTS_SCAN_ITEMS; q++)
        if (base + q < n) v += in[base + q];
    T total;
    rts_block_exclusive(v, lds, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

template <typename T>
__global__ __launch_bounds__(RTS_BLOCK) void k_rts_scan_sums(T* sums, size_t nb) {   // one block, in place
    __shared__ T lds[RTS_BLOCK];
    T carry = 0;
    for (size_t base = 0; base < nb; base += RTS_BLOCK) {
        const size_t i = base + threadIdx.x;
        const T v = i < nb ? sums[i] : (T)0;
        T total;
        const T ex = rts_block_exclusive(v, lds, total);
        if (i < nb) sums[i] = carry + ex;
        carry += total;
    }
}

template <typename T>
__global__ __launch_bounds__(RTS_BLOCK) void k_rts_tile_scan(const T* in, T* out, size_t n, const T* sums) {
    __shared__ T lds[RTS_BLOCK];
    const size_t base = (size_t)blockIdx.x * RTS_SCAN_TILE + (size_t)threadIdx.x * RTS_SCAN_ITEMS;
    T item[RTS_SCAN_ITEMS];
    T v = 0;
    for (int q = 0; q < RTS_SCAN_ITEMS; q++) {
        item[q] = base + q < n ? in[base + q] : (T)0;
        v += item[q];
    }
    T total;
    T run = rts_block_exclusive(v, lds, total) + sums[blockIdx.x];
    for (int q = 0; q < RTS_SCAN_ITEMS; q++) {
        if (base + q < n) out[base + q] = run;
        run += item[q];
    }
}

template <typename T>
static size_t rts_scan_temp_bytes(size_t n) {
    return ((n + RTS_SCAN_TILE - 1) / RTS_SCAN_TILE + 1) * sizeof(T);
}

// out[i] = in[0] + ... + in[i-1].  in and out may be the same array.  temp: rts_scan_temp_bytes<T>(n) bytes.
template <typename T>
static hipError_t rts_exclusive_scan(void* temp, const T* in, T* out, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const size_t nb = (n + RTS_SCAN_TILE - 1) / RTS_SCAN_TILE;
    T* sums = (T*)temp;
    hipLaunchKernelGGL(k_rts_tile_sums<T>, dim3((unsigned)nb), dim3(RTS_BLOCK), 0, st, in, n, sums);
    hipLaunchKernelGGL(k_rts_scan_sums<T>, dim3(1), dim3(RTS_BLOCK), 0, st, sums, nb);
    hipLaunchKernelGGL(k_rts_tile_scan<T>, dim3((unsigned)nb), dim3(RTS_BLOCK), 0, st, in, out, n, (const T*)sums);
    return hipGetLastError();
}

// ---------------------------------------------------------------- radix sort
// hist[bin * nblocks + block]: bin-major, so that one exclusive scan of the table gives, for every (bin, block),
// the number of keys with a smaller digit anywhere plus the keys with this digit in earlier blocks.
__global__ __launch_bounds__(RTS_BLOCK) void k_rts_hist(const unsigned long long* keys, size_t n, int shift, uint32_t nblocks,
                                                        uint32_t* hist) {
    __shared__ uint32_t bins[256];
    bins[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RTS_SORT_TILE;
    for (int r = 0; r < RTS_SORT_ROUNDS; r++) {
        const size_t i = base + (size_t)r * RTS_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&bins[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = bins[threadIdx.x];
}

__global__ __launch_bounds__(RTS_BLOCK) void k_rts_scatter(const unsigned long long* keys, const uint32_t* vals, size_t n, int shift,
                                                           uint32_t nblocks, const uint32_t* offsets, unsigned long long* keys_out,
                                                           uint32_t* vals_out) {
    __shared__ uint32_t base[256];                 // next output slot of every digit for this block
    __shared__ uint32_t cnt[RTS_BLOCK / 64][256];  // keys of every digit in every wave, this round
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    base[threadIdx.x] = offsets[(size_t)threadIdx.x * nblocks + blockIdx.x];
    const size_t first = (size_t)blockIdx.x * RTS_SORT_TILE;
    for (int r = 0; r < RTS_SORT_ROUNDS; r++) {
        for (int w = 0; w < RTS_BLOCK / 64; w++) cnt[w][threadIdx.x] = 0;
        __syncthreads();
        const size_t i = first + (size_t)r * RTS_BLOCK + threadIdx.x;
        const bool valid = i < n;
        unsigned long long key = 0;
        uint32_t val = 0, digit = 0;
        if (valid) { key = keys[i]; val = vals[i]; digit = (uint32_t)(key >> shift) & 255u; }
        // lanes of this wave that hold the same digit
        unsigned long long same = __builtin_amdgcn_ballot_w64(valid);
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(same & below);
        if (valid && rank == 0) cnt[wave][digit] = (uint32_t)__popcll(same);
        __syncthreads();
        if (valid) {
            uint32_t at = base[digit] + rank;
            for (uint32_t w = 0; w < wave; w++) at += cnt[w][digit];
            keys_out[at] = key;
            vals_out[at] = val;
        }
        __syncthreads();
        uint32_t add = 0;
        for (int w = 0; w < RTS_BLOCK / 64; w++) add += cnt[w][threadIdx.x];
        base[threadIdx.x] += add;
        __syncthreads();
    }
}

__global__ void k_rts_check_sorted(const unsigned long long* keys, const uint32_t* vals, size_t n, unsigned long long* out /*[2]*/) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0, sum = 0;
    if (i < n) {
        sum = vals[i];
        if (i + 1 < n && keys[i] > keys[i + 1]) bad = 1;
    }
    for (int off = 32; off > 0; off >>= 1) { bad += __shfl_down(bad, off); sum += __shfl_down(sum, off); }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicAdd(&out[0], bad);
        atomicAdd(&out[1], sum);
    }
}

static size_t rts_sort_temp_bytes(size_t n) {
    const size_t nblocks = (n + RTS_SORT_TILE - 1) / RTS_SORT_TILE;
    const size_t table = 256 * nblocks;
    return table * sizeof(uint32_t) + rts_scan_temp_bytes<uint32_t>(table) + 64;
}

// Sorts the n pairs by the low `bits` bits of the key (rounded up to whole bytes), stable.  The pairs bounce
// between (keys, vals) and (keys_alt, vals_alt); *in_alt tells where they end up.
static hipError_t rts_sort_pairs(void* temp, unsigned long long* keys, unsigned long long* keys_alt, uint32_t* vals, uint32_t* vals_alt,
                                 size_t n, int bits, hipStream_t st, bool* in_alt) {
    *in_alt = false;
    if (n == 0) return hipSuccess;
    const uint32_t nblocks = (uint32_t)((n + RTS_SORT_TILE - 1) / RTS_SORT_TILE);
    const size_t table = (size_t)256 * nblocks;
    uint32_t* hist = (uint32_t*)temp;
    void* scan_temp = (void*)(hist + table);
    unsigned long long* kin = keys; unsigned long long* kout = keys_alt;
    uint32_t* vin = vals; uint32_t* vout = vals_alt;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(k_rts_hist, dim3(nblocks), dim3(RTS_BLOCK), 0, st, kin, n, shift, nblocks, hist);
        hipError_t e = rts_exclusive_scan<uint32_t>(scan_temp, hist, hist, table, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_rts_scatter, dim3(nblocks), dim3(RTS_BLOCK), 0, st, kin, vin, n, shift, nblocks, hist, kout, vout);
        unsigned long long* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
        *in_alt = !*in_alt;
    }
    return hipGetLastError();
}
