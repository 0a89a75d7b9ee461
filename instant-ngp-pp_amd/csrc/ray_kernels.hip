// Ray-side kernels: AABB / sphere intersection, morton + packbits, occupancy-grid helpers,
// the training marcher (count -> scan -> expand) and the test-time marcher.
//
// Built with -ffp-contract=off: every expression keeps the reference's operation order
// (raymarching.cu / intersection.cu) with one rounding per operation, so results are
// bit-identical to oracle/ngp_oracle.c.
#include "common.h"

#define SQRT3F 1.73205080757f

namespace {

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// ------------------------------------------------------------------ intersections (R1)
// One lane per ray, serial loop over voxels (V is 1 on the hot path, rendering.py:29).
// Hits are kept in registers for max_hits==1 and in the output row otherwise, then the
// row is insertion-sorted ascending on t1 with unused (-1) slots first — the layout the
// reference's torch::sort + gather produces (intersection.cu:94-97).
template <bool SPHERE>
__global__ void intersect_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                 const float* __restrict__ centers, const float* __restrict__ sizes,
                                 int n_rays, int n_prims, int max_hits,
                                 int32_t* __restrict__ hit_cnt, float* __restrict__ hits_t,
                                 int64_t* __restrict__ hits_idx)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    float* ht = hits_t + (size_t)r * max_hits * 2;
    int64_t* hi = hits_idx + (size_t)r * max_hits;
    for (int k = 0; k < max_hits; k++) { ht[2 * k] = -1.0f; ht[2 * k + 1] = -1.0f; hi[k] = -1; }
    int cnt = 0;
    for (int v = 0; v < n_prims; v++) {
        float t1, t2;
        if (SPHERE) {
            const float cx = ox - centers[3 * v], cy = oy - centers[3 * v + 1], cz = oz - centers[3 * v + 2];
            const float rad = sizes[v];
            const float a = dx * dx + dy * dy + dz * dz;
            const float half_b = dx * cx + dy * cy + dz * cz;
            const float c = cx * cx + cy * cy + cz * cz - rad * rad;
            const float disc = half_b * half_b - a * c;
            t1 = -1.0f; t2 = -1.0f;
            if (!(disc < 0)) {
                const float sq = sqrtf(disc);
                t1 = (-half_b - sq) / a; t2 = (-half_b + sq) / a;
            }
        } else {
            const float cx = centers[3 * v], cy = centers[3 * v + 1], cz = centers[3 * v + 2];
            const float hx = sizes[3 * v], hy = sizes[3 * v + 1], hz = sizes[3 * v + 2];
            const float ax = (cx - hx - ox) * ix, bx = (cx + hx - ox) * ix;
            const float ay = (cy - hy - oy) * iy, by = (cy + hy - oy) * iy;
            const float az = (cz - hz - oz) * iz, bz = (cz + hz - oz) * iz;
            t1 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
            t2 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
            if (t1 > t2) { t1 = -1.0f; t2 = -1.0f; }
        }
        if (t2 > 0) {
            if (cnt < max_hits) { ht[2 * cnt] = fmaxf(t1, 0.0f); ht[2 * cnt + 1] = t2; hi[cnt] = v; }
            cnt++;
        }
    }
    hit_cnt[r] = cnt;
    for (int i = 1; i < max_hits; i++) {
        const float a = ht[2 * i], b = ht[2 * i + 1];
        const int64_t v = hi[i];
        int j = i - 1;
        while (j >= 0 && ht[2 * j] > a) {
            ht[2 * j + 2] = ht[2 * j]; ht[2 * j + 3] = ht[2 * j + 1]; hi[j + 1] = hi[j];
            j--;
        }
        ht[2 * j + 2] = a; ht[2 * j + 3] = b; hi[j + 1] = v;
    }
}

__global__ void clamp_near_kernel(float* __restrict__ hits_t, int n_rays, int max_hits, float near_d)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    float* p = hits_t + (size_t)r * max_hits * 2;
    const float t1 = p[0];
    if (t1 >= 0 && t1 < near_d) p[0] = near_d;
}

// ------------------------------------------------------------------ morton / packbits (O1)
__device__ __forceinline__ uint32_t spread3(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton_enc(uint32_t x, uint32_t y, uint32_t z)
{
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}
__device__ __forceinline__ uint32_t compact3(uint32_t x)
{
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

__global__ void morton_kernel(const int32_t* __restrict__ coords, int n, int32_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = (int32_t)morton_enc((uint32_t)coords[3 * i], (uint32_t)coords[3 * i + 1], (uint32_t)coords[3 * i + 2]);
}

__global__ void morton_invert_kernel(const int32_t* __restrict__ idx, int n, int32_t* __restrict__ coords)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t v = idx[i];
    coords[3 * i] = (int32_t)compact3((uint32_t)(v >> 0));
    coords[3 * i + 1] = (int32_t)compact3((uint32_t)(v >> 1));
    coords[3 * i + 2] = (int32_t)compact3((uint32_t)(v >> 2));
}

// one lane per output byte; the 8 floats of a byte are two 16-byte loads
__global__ void packbits_kernel(const float4* __restrict__ grid, int n_bytes, float thr_host,
                                const float* __restrict__ thr_dev, uint8_t* __restrict__ bits)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_bytes) return;
    const float thr = thr_dev ? *thr_dev : thr_host;
    const float4 a = grid[2 * (size_t)n], b = grid[2 * (size_t)n + 1];
    uint32_t m = 0;
    m |= (a.x > thr) ? 1u : 0u;   m |= (a.y > thr) ? 2u : 0u;
    m |= (a.z > thr) ? 4u : 0u;   m |= (a.w > thr) ? 8u : 0u;
    m |= (b.x > thr) ? 16u : 0u;  m |= (b.y > thr) ? 32u : 0u;
    m |= (b.z > thr) ? 64u : 0u;  m |= (b.w > thr) ? 128u : 0u;
    bits[n] = (uint8_t)m;
}

__global__ void cell_points_kernel(const int32_t* __restrict__ coords, const float* __restrict__ noise, int n3,
                                   int G, float s, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    const float hgs = s / G;
    const float c = (float)coords[i] / (float)(G - 1) * 2 - 1;
    out[i] = c * (s - hgs) + (noise[i] * 2 - 1) * hgs;
}

// ---- sampled occupancy update, fused (networks.py:308-333, 388-405) ---------------------------------------
// M uniformly random cells + M cells drawn uniformly from the occupied ones, their jittered world points, a
// Morton bucket order (gather locality of the density evaluation that follows), scatter of the densities back
// and EMA + mean-of-positive-cells + threshold: 8 launches instead of ~70 torch ones.  All randomness comes
// from a counter-based hash of (seed, sample id, draw), so the result does not depend on launch order and is
// the same on every data-parallel rank.
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint32_t id, uint32_t draw)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)id * 8u + draw + 1u);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}
__device__ __forceinline__ uint32_t rand_below(uint64_t seed, uint32_t id, uint32_t draw, uint32_t n)
{
    return (uint32_t)(((uint64_t)hash_u32(seed, id, draw) * n) >> 32);
}

// cells per block of the occupancy count / compaction kernels: 256 lanes x 4 cells
#define OCC_BLOCK 1024

__device__ __forceinline__ int occ_mask4(const float* __restrict__ grid, int g3, int base, float thr)
{
    int m = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (base + j < g3 && grid[base + j] > thr) m |= 1 << j;
    return m;
}

__global__ void __launch_bounds__(256) occ_count_kernel(const float* __restrict__ grid, int g3, float thr,
                                                        int32_t* __restrict__ block_counts)
{
    __shared__ int part[4];
    const int base = blockIdx.x * OCC_BLOCK + threadIdx.x * 4;
    int c = __popc(occ_mask4(grid, g3, base, thr));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of `n` counts by ONE 1024-thread block: offsets[i] = sum of counts[0..i), total[0] = the sum;
// also clears `clear_n` ints at `clear` (the bucket cursors of the next pass)
__global__ void __launch_bounds__(1024) scan_counts_kernel(const int32_t* __restrict__ counts, int n,
                                                           int32_t* __restrict__ offsets, int32_t* __restrict__ total,
                                                           int32_t* __restrict__ clear, int clear_n)
{
    __shared__ int wsum[16];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    int s = 0;
    for (int i = lo; i < hi; i++) s += counts[i];
    // exclusive scan of the 1024 thread sums: inside each wave by shuffles, across the 16 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    int run = woff + inc - s;
    for (int i = lo; i < hi; i++) { offsets[i] = run; run += counts[i]; }
    if (threadIdx.x == 1023 && total) total[0] = woff + inc;
    for (int i = threadIdx.x; i < clear_n; i += 1024) clear[i] = 0;
}

__global__ void __launch_bounds__(256) occ_compact_kernel(const float* __restrict__ grid, int g3, float thr,
                                                          const int32_t* __restrict__ block_offsets,
                                                          int32_t* __restrict__ occ_list)
{
    __shared__ int wsum[4];
    const int base = blockIdx.x * OCC_BLOCK + threadIdx.x * 4;
    const int m = occ_mask4(grid, g3, base, thr);
    const int c = __popc(m);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int pos = block_offsets[blockIdx.x] + inc - c;
    for (int w = 0; w < wave; w++) pos += wsum[w];
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (m & (1 << j)) occ_list[pos++] = base + j;     // ascending cell (= Morton) index
}

// Morton order of the samples matters: the density evaluation that follows gathers 128 hash-table rows per
// point, and neighbouring points share them.  A two-level counting sort: 256 top-level buckets (the key's high
// bits; per-block LDS histograms keep the global atomics at one per bucket and block), then one workgroup per
// bucket orders it by the next 13 bits in LDS.
#define SAMPLE_TOP_BITS 8
#define SAMPLE_TOP (1 << SAMPLE_TOP_BITS)
#define SAMPLE_SUB_BITS 13
#define SAMPLE_SUB (1 << SAMPLE_SUB_BITS)
#define SAMPLE_CHUNK 4096          // samples per 256-thread block in the histogram / scatter passes

// sample i < m: a uniformly random cell; sample m + j: the k-th occupied cell, k uniform in [0, n_occ) — with no
// occupied cell at all (the reference then samples the m uniform ones only, networks.py:326-329) it repeats
// sample j exactly, which adds nothing.  keys = Morton index, sids = the id that seeds the sample's jitter.
__global__ void __launch_bounds__(256) sample_cells_kernel(const int32_t* __restrict__ occ_list,
                                                           const int32_t* __restrict__ n_occ_p, int G, int m,
                                                           uint64_t seed, int shift, int32_t* __restrict__ keys,
                                                           int32_t* __restrict__ sids, int32_t* __restrict__ bucket_counts)
{
    __shared__ int hist[SAMPLE_TOP];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int n_occ = *n_occ_p;
    const int lo = blockIdx.x * SAMPLE_CHUNK;
    for (int i = lo + threadIdx.x; i < lo + SAMPLE_CHUNK && i < 2 * m; i += 256) {
        int sid = i;
        uint32_t key;
        if (i >= m && n_occ > 0) {
            key = (uint32_t)occ_list[rand_below(seed, (uint32_t)i, 3, (uint32_t)n_occ)];
        } else {
            if (i >= m) sid = i - m;
            key = morton_enc(rand_below(seed, (uint32_t)sid, 0, (uint32_t)G), rand_below(seed, (uint32_t)sid, 1, (uint32_t)G),
                             rand_below(seed, (uint32_t)sid, 2, (uint32_t)G));
        }
        keys[i] = (int32_t)key; sids[i] = sid;
        atomicAdd(&hist[key >> shift], 1);
    }
    __syncthreads();
    if (hist[threadIdx.x]) atomicAdd(bucket_counts + threadIdx.x, hist[threadIdx.x]);
}

// scatter into the top-level buckets: one reservation per (block, bucket), positions inside it by LDS atomics
__global__ void __launch_bounds__(256) bucket_scatter_kernel(const int32_t* __restrict__ keys, const int32_t* __restrict__ sids,
                                                             int n, int shift, const int32_t* __restrict__ bucket_offsets,
                                                             int32_t* __restrict__ bucket_cursor,
                                                             int32_t* __restrict__ keys_b, int32_t* __restrict__ sids_b)
{
    __shared__ int hist[SAMPLE_TOP], base[SAMPLE_TOP];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int lo = blockIdx.x * SAMPLE_CHUNK;
    for (int i = lo + threadIdx.x; i < lo + SAMPLE_CHUNK && i < n; i += 256) atomicAdd(&hist[(uint32_t)keys[i] >> shift], 1);
    __syncthreads();
    const int c = hist[threadIdx.x];
    base[threadIdx.x] = bucket_offsets[threadIdx.x] + (c ? atomicAdd(bucket_cursor + threadIdx.x, c) : 0);
    __syncthreads();
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = lo + threadIdx.x; i < lo + SAMPLE_CHUNK && i < n; i += 256) {
        const uint32_t key = (uint32_t)keys[i];
        const uint32_t b = key >> shift;
        const int pos = base[b] + atomicAdd(&hist[b], 1);
        keys_b[pos] = (int32_t)key; sids_b[pos] = sids[i];
    }
}

// one workgroup per top-level bucket: counting sort by the next SAMPLE_SUB_BITS key bits in LDS, then the
// jittered world point of every sample: x_w = (coord/(G-1)*2-1)*(s - s/G) + (u*2-1)*s/G, u in [0,1)  (networks.py:391-395)
__global__ void __launch_bounds__(1024) bucket_sort_points_kernel(const int32_t* __restrict__ keys_b, const int32_t* __restrict__ sids_b,
                                                                  const int32_t* __restrict__ bucket_offsets, int n, int sub_shift,
                                                                  int G, float s, uint64_t seed,
                                                                  int32_t* __restrict__ out_idx, float* __restrict__ out_xyz)
{
    __shared__ int bins[SAMPLE_SUB];
    __shared__ int wsum[16];
    const int b = blockIdx.x;
    const int lo = bucket_offsets[b], hi = b + 1 < SAMPLE_TOP ? bucket_offsets[b + 1] : n;
    for (int i = threadIdx.x; i < SAMPLE_SUB; i += 1024) bins[i] = 0;
    __syncthreads();
    for (int i = lo + threadIdx.x; i < hi; i += 1024) atomicAdd(&bins[((uint32_t)keys_b[i] >> sub_shift) & (SAMPLE_SUB - 1)], 1);
    __syncthreads();
    // exclusive scan of the bins: 8 per thread, wave scan, 16 wave totals
    constexpr int PER = SAMPLE_SUB / 1024;
    int loc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { loc[k] = bins[threadIdx.x * PER + k]; sum += loc[k]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int run = lo + inc - sum;
    for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int k = 0; k < PER; k++) { bins[threadIdx.x * PER + k] = run; run += loc[k]; }
    __syncthreads();
    const float hgs = s / G;
    for (int i = lo + threadIdx.x; i < hi; i += 1024) {
        const uint32_t key = (uint32_t)keys_b[i];
        const uint32_t sid = (uint32_t)sids_b[i];
        const int pos = atomicAdd(&bins[(key >> sub_shift) & (SAMPLE_SUB - 1)], 1);
        out_idx[pos] = (int32_t)key;
        const uint32_t c[3] = { compact3(key), compact3(key >> 1), compact3(key >> 2) };
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float u = (float)(hash_u32(seed, sid, 4 + k) >> 8) * (1.0f / 16777216.0f);
            const float cc = (float)c[k] / (float)(G - 1) * 2 - 1;
            out_xyz[3 * (size_t)pos + k] = cc * (s - hgs) + (u * 2 - 1) * hgs;
        }
    }
}

// tmp[idx[i]] = max(tmp[idx[i]], sigma[i]) — densities are positive (Softplus), so the int compare orders them;
// with several samples in one cell the reference keeps whichever index_put wrote last: the largest is one of them
__global__ void __launch_bounds__(256) grid_scatter_max_kernel(float* __restrict__ tmp, const int32_t* __restrict__ idx,
                                                               const float* __restrict__ sigma, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = sigma[i];
    if (v > 0.0f) atomicMax(reinterpret_cast<int*>(tmp) + idx[i], __float_as_int(v));
}

// EMA (networks.py:400-403) + per-block sum / count of the positive cells, in a fixed order (deterministic)
__global__ void __launch_bounds__(256) grid_ema_stats_kernel(float* __restrict__ grid, const float* __restrict__ tmp, int n,
                                                             float decay, float* __restrict__ partials)
{
    __shared__ float ps[4], pc[4];
    float sum = 0.0f, cnt = 0.0f;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        float g = grid[i];
        if (!(g < 0)) { g = fmaxf(g * decay, tmp[i]); grid[i] = g; }
        if (g > 0) { sum += g; cnt += 1.0f; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); cnt += __shfl_xor(cnt, o, 64); }
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = sum; pc[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        partials[2 * blockIdx.x + 1] = (pc[0] + pc[1]) + (pc[2] + pc[3]);
    }
}

// threshold = min(mean of the positive cells, density_threshold)  (networks.py:405-407); no positive cell: 0
__global__ void __launch_bounds__(256) grid_threshold_kernel(const float* __restrict__ partials, int n_blocks,
                                                             float density_threshold, float* __restrict__ thr_out)
{
    __shared__ double ps[4], pc[4];
    double sum = 0.0, cnt = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += 256) { sum += partials[2 * i]; cnt += partials[2 * i + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); cnt += __shfl_xor(cnt, o, 64); }
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = sum; pc[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = (ps[0] + ps[1]) + (ps[2] + ps[3]), c = (pc[0] + pc[1]) + (pc[2] + pc[3]);
        const float mean = c > 0 ? (float)(s / c) : 0.0f;
        thr_out[0] = fminf(mean, density_threshold);
        thr_out[1] = mean;
    }
}

__global__ void grid_ema_kernel(float* __restrict__ grid, const float* __restrict__ tmp, int n, float decay)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = grid[i];
    if (!(g < 0)) grid[i] = fmaxf(g * decay, tmp[i]);
}

// ------------------------------------------------------------------ marcher (R3 / T1)
__device__ __forceinline__ float step_dt(float t, float esf, int max_samples, int G, float scale)
{
    return clampf(t * esf, SQRT3F / max_samples, SQRT3F * 2 * scale / G);
}
__device__ __forceinline__ int mip_of_pos(float x, float y, float z, int cascades)
{
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e; frexpf(mx, &e);
    return min(cascades - 1, max(0, e + 1));
}
__device__ __forceinline__ int mip_of_dt(float dt, int G, int cascades)
{
    int e; frexpf(dt * G, &e);
    return min(cascades - 1, max(0, e));
}

struct MarchRay {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;
};

// One DDA decision at t (raymarching.cu:205-233).  Returns true when the cell is occupied
// (t untouched, dt = the step to take), else advances t past the empty cell.
__device__ __forceinline__ bool march_probe(const MarchRay& c, const uint8_t* __restrict__ bits, int cascades,
                                            int G, uint32_t G3, float Ginv, float scale, float dt_scale, float esf,
                                            int max_samples, float& t, float& x, float& y, float& z, float& dt)
{
    const float tt = t;
    x = c.ox + tt * c.dx; y = c.oy + tt * c.dy; z = c.oz + tt * c.dz;
    dt = step_dt(tt, esf, max_samples, G, dt_scale);
    const int mip = max(mip_of_pos(x, y, z, cascades), mip_of_dt(dt, G, cascades));
    const float bound = fminf(scalbnf(1.0f, mip - 1), scale);
    const float binv = 1 / bound;
    const int nx = (int)clampf(0.5f * (x * binv + 1) * G, 0.0f, G - 1.0f);
    const int ny = (int)clampf(0.5f * (y * binv + 1) * G, 0.0f, G - 1.0f);
    const int nz = (int)clampf(0.5f * (z * binv + 1) * G, 0.0f, G - 1.0f);
    const uint32_t idx = (uint32_t)mip * G3 + morton_enc((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    const bool occ = bits[idx >> 3] & (1u << (idx & 7u));
    if (occ) return true;
    const float tx = (((nx + 0.5f + 0.5f * copysignf(1.0f, c.dx)) * Ginv * 2 - 1) * bound - x) * c.ix;
    const float ty = (((ny + 0.5f + 0.5f * copysignf(1.0f, c.dy)) * Ginv * 2 - 1) * bound - y) * c.iy;
    const float tz = (((nz + 0.5f + 0.5f * copysignf(1.0f, c.dz)) * Ginv * 2 - 1) * bound - z) * c.iz;
    const float t_target = tt + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    float tn = tt;
    do { tn += step_dt(tn, esf, max_samples, G, dt_scale); } while (tn < t_target);
    t = tn;
    return false;
}

__device__ __forceinline__ MarchRay load_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, size_t r)
{
    MarchRay c;
    c.ox = rays_o[3 * r]; c.oy = rays_o[3 * r + 1]; c.oz = rays_o[3 * r + 2];
    c.dx = rays_d[3 * r]; c.dy = rays_d[3 * r + 1]; c.dz = rays_d[3 * r + 2];
    c.ix = 1.0f / c.dx; c.iy = 1.0f / c.dy; c.iz = 1.0f / c.dz;
    return c;
}

#ifdef NGP_AB_VARIANTS   // A/B build only (NGP_MARCH_LANE_PER_RAY=1): the serial lane-per-ray walk the wave kernel replaced
// Pass 1: one lane per ray walks the occupancy bitfield ONCE, parks each accepted sample's t
// in t_scratch[r*max_samples + k] (a lane's consecutive stores fall in the same L2 lines and
// are merged there) and records the count.  The reference walks every ray twice.
__global__ void march_count_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                   const float* __restrict__ hits_t, const uint8_t* __restrict__ bits,
                                   int cascades, float scale, float esf, const float* __restrict__ noise,
                                   int G, int max_samples, int n_rays,
                                   float* __restrict__ t_scratch, int32_t* __restrict__ counts)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const MarchRay c = load_ray(rays_o, rays_d, r);
    const uint32_t G3 = (uint32_t)G * G * G;
    const float Ginv = 1.0f / G;
    float t1 = hits_t[2 * r];
    const float t2 = hits_t[2 * r + 1];
    if (t1 >= 0) {
        const float dt0 = step_dt(t1, esf, max_samples, G, scale);
        t1 += dt0 * noise[r];
    }
    float t = t1; int n = 0;
    float* ts = t_scratch + (size_t)r * max_samples;
    while (0 <= t && t < t2 && n < max_samples) {
        float x, y, z, dt;
        const float tcur = t;
        if (march_probe(c, bits, cascades, G, G3, Ginv, scale, scale, esf, max_samples, t, x, y, z, dt)) {
            ts[n] = tcur;
            t += dt; n++;
        }
    }
    counts[r] = n;
}
#endif  // NGP_AB_VARIANTS

// Pass 1, wavefront-packed: one WAVE per ray.  Whatever the occupancy says, the marcher only ever
// stands on elements of one chain t_{k+1} = t_k + dt(t_k) (an accepted sample advances by dt, a skip
// advances by dt until it has passed the empty cell, raymarching.cu:224-233), and that chain does not
// depend on the bitfield.  So: lane 0 lays down a segment of the chain in LDS (the additions must be
// sequential to stay bit-identical to the serial walk); the 64 lanes then probe 64 consecutive
// elements at once — position, mip level, Morton bit test, and for an empty cell the index of the first
// chain element behind it (binary search in the segment) — and the serial walk over the window
// collapses to a few v_readlane hops through those per-lane successors.  Visited-and-occupied lanes
// are the samples; their t's are written compacted, in order.  Same results as the serial lane-per-ray walk
// (march_count_kernel, A/B build), bit for bit; ~20x shorter because the bitfield latency is paid once per 64 elements, not per step.
__global__ void __launch_bounds__(256) march_wave_kernel(const float* __restrict__ rays_o,
                                                         const float* __restrict__ rays_d,
                                                         const float* __restrict__ hits_t,
                                                         const uint8_t* __restrict__ bits, int cascades, float scale,
                                                         float esf, const float* __restrict__ noise, int G,
                                                         int max_samples, int n_rays, float* __restrict__ t_scratch,
                                                         int32_t* __restrict__ counts)
{
    constexpr int SEG = 1024;
    __shared__ float chain_s[4][SEG + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= n_rays) return;                                  // wave-uniform
    float* chain = chain_s[wave];
    const MarchRay c = load_ray(rays_o, rays_d, r);
    const uint32_t G3 = (uint32_t)G * G * G;
    const float Ginv = 1.0f / G;
    float t1 = hits_t[2 * r];
    const float t2 = hits_t[2 * r + 1];
    if (t1 >= 0) t1 += step_dt(t1, esf, max_samples, G, scale) * noise[r];
    float* ts = t_scratch + (size_t)r * max_samples;
    float t = t1;               // next chain element to stand on (wave-uniform)
    float pending = 0.0f;       // a skip target that lies beyond the previous segment
    bool has_pending = false;
    int n = 0;
    while (0 <= t && t < t2 && n < max_samples) {
        // ---- lay down the next segment: chain[0..len) < t2, chain[len] = the element after it
        int len = 0;
        if (lane == 0) {
            float tc = t;
            while (len < SEG && tc < t2) { chain[len++] = tc; tc += step_dt(tc, esf, max_samples, G, scale); }
            chain[len] = tc;
        }
        len = __builtin_amdgcn_readfirstlane(len);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int k0 = 0;
        if (has_pending) {      // first element that is not before the pending skip target
            int lo = 0, hi = len;                     // invariant: answer in [lo, hi]
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (chain[mid] < pending) lo = mid + 1; else hi = mid; }
            k0 = lo;
            if (k0 < len || !(chain[len] < pending)) has_pending = false;
        }
        while (k0 < len && n < max_samples) {
            const int k = k0 + lane;
            const bool valid = k < len;
            int nxt = k + 1;            // successor index inside the segment (len = leave the segment)
            bool occ = false, beyond = false;
            float tk = 0.0f, t_target = 0.0f;
            if (valid) {
                tk = chain[k];
                const float x = c.ox + tk * c.dx, y = c.oy + tk * c.dy, z = c.oz + tk * c.dz;
                const float dt = step_dt(tk, esf, max_samples, G, scale);
                const int mip = max(mip_of_pos(x, y, z, cascades), mip_of_dt(dt, G, cascades));
                const float bound = fminf(scalbnf(1.0f, mip - 1), scale);
                const float binv = 1 / bound;
                const int nx = (int)clampf(0.5f * (x * binv + 1) * G, 0.0f, G - 1.0f);
                const int ny = (int)clampf(0.5f * (y * binv + 1) * G, 0.0f, G - 1.0f);
                const int nz = (int)clampf(0.5f * (z * binv + 1) * G, 0.0f, G - 1.0f);
                const uint32_t idx = (uint32_t)mip * G3 + morton_enc((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
                occ = bits[idx >> 3] & (1u << (idx & 7u));
                if (!occ) {
                    const float tx = (((nx + 0.5f + 0.5f * copysignf(1.0f, c.dx)) * Ginv * 2 - 1) * bound - x) * c.ix;
                    const float ty = (((ny + 0.5f + 0.5f * copysignf(1.0f, c.dy)) * Ginv * 2 - 1) * bound - y) * c.iy;
                    const float tz = (((nz + 0.5f + 0.5f * copysignf(1.0f, c.dz)) * Ginv * 2 - 1) * bound - z) * c.iz;
                    t_target = tk + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
                    // do { t += dt } while (t < t_target): first j >= k+1 with !(chain[j] < t_target)
                    int lo = k + 1, hi = len;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (chain[mid] < t_target) lo = mid + 1; else hi = mid; }
                    nxt = lo;
                    beyond = lo == len && chain[len] < t_target;
                }
            }
            // ---- the serial walk over this window: hop through the successors (wave-uniform)
            uint64_t visited = 0;
            int cur = k0;
            const int wend = min(k0 + 64, len);
            int last = -1;
            while (cur < wend) {
                last = cur - k0;
                visited |= 1ull << last;
                cur = __builtin_amdgcn_readlane(nxt, last);
            }
            // accepted samples = visited and occupied, in chain order, capped at max_samples
            const uint64_t acc = visited & __ballot(occ);
            const int rank = __popcll(acc & ((1ull << lane) - 1ull));
            const int room = max_samples - n;
            if (((acc >> lane) & 1ull) && rank < room) ts[n + rank] = tk;
            const int cnt = __popcll(acc);
            n += cnt < room ? cnt : room;
            // a skip out of the segment whose target lies behind the sentinel has to be finished in the next one
            if (cur >= len && last >= 0) {
                const int b = __builtin_amdgcn_readlane((int)beyond, last);
                if (b) { has_pending = true; pending = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_target), last)); }
            }
            k0 = cur;
        }
        t = chain[len];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) counts[r] = n;
}

// Pass 2: single workgroup, exclusive scan of the per-ray counts -> rays_a rows (ray order)
// and the {total, n_rays} counter.
__global__ void __launch_bounds__(1024) march_scan_kernel(const int32_t* __restrict__ counts, int n_rays,
                                                          int64_t* __restrict__ rays_a, int32_t* __restrict__ counter)
{
    __shared__ int32_t wave_tot[16];
    __shared__ int64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_rays; base += 1024) {
        const int r = base + tid;
        const int32_t c = r < n_rays ? counts[r] : 0;
        int32_t v = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t u = __shfl_up(v, o, 64);
            if (lane >= o) v += u;
        }
        if (lane == 63) wave_tot[wid] = v;
        __syncthreads();
        int32_t woff = 0;
        for (int w = 0; w < wid; w++) woff += wave_tot[w];
        const int64_t carry = carry_s;
        if (r < n_rays) {
            rays_a[3 * (size_t)r] = r;
            rays_a[3 * (size_t)r + 1] = carry + woff + (v - c);
            rays_a[3 * (size_t)r + 2] = c;
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + v;
        __syncthreads();
    }
    if (tid == 0) { counter[0] = (int32_t)carry_s; counter[1] = n_rays; }
}

// Pass 3: one wave per ray, lanes stride over the ray's samples: coalesced reads of the
// parked t's and contiguous xyz/dir/delta/t writes.
__global__ void march_expand_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                    const float* __restrict__ t_scratch, const int64_t* __restrict__ rays_a,
                                    float esf, int G, float scale, int max_samples, int n_rays,
                                    float* __restrict__ xyzs, float* __restrict__ dirs,
                                    float* __restrict__ deltas, float* __restrict__ ts)
{
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (r >= n_rays) return;
    const int64_t start = rays_a[3 * (size_t)r + 1];
    const int n = (int)rays_a[3 * (size_t)r + 2];
    if (n == 0) return;
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    const float* tsrc = t_scratch + (size_t)r * max_samples;
    for (int k = lane; k < n; k += 64) {
        const float t = tsrc[k];
        const int64_t s = start + k;
        xyzs[3 * s] = ox + t * dx; xyzs[3 * s + 1] = oy + t * dy; xyzs[3 * s + 2] = oz + t * dz;
        dirs[3 * s] = dx; dirs[3 * s + 1] = dy; dirs[3 * s + 2] = dz;
        ts[s] = t;
        deltas[s] = step_dt(t, esf, max_samples, G, scale);
    }
}

// optional: zero rows [counter[0], capacity) like the reference's torch::zeros outputs
__global__ void march_zero_tail_kernel(const int32_t* __restrict__ counter, int64_t capacity,
                                       float* __restrict__ xyzs, float* __restrict__ dirs,
                                       float* __restrict__ deltas, float* __restrict__ ts)
{
    const int64_t first = counter[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = first * 3 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity * 3; i += stride) {
        xyzs[i] = 0.0f; dirs[i] = 0.0f;
    }
    for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += stride) {
        deltas[i] = 0.0f; ts[i] = 0.0f;
    }
}

// Test-time marcher (raymarching.cu:353-403): one lane per alive ray, at most n_samples steps.
// calc_dt receives `cascades` where `scale` is expected (raymarching.cu:370,399) — kept.
__global__ void march_test_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                  float* __restrict__ hits_t, const int64_t* __restrict__ alive,
                                  const uint8_t* __restrict__ bits, int cascades, float scale, float esf,
                                  int G, int max_samples, int n_samples, int n_alive,
                                  float* __restrict__ xyzs, float* __restrict__ dirs,
                                  float* __restrict__ deltas, float* __restrict__ ts, int32_t* __restrict__ n_eff,
                                  int32_t* __restrict__ state)
{
    if (state) {   // device-driven rounds (ngp_test_round_begin): the round's sizes live on the device
        if (state[3]) return;
        n_alive = state[0]; n_samples = state[1];
    }
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const size_t r = (size_t)alive[n];
    const MarchRay c = load_ray(rays_o, rays_d, r);
    const uint32_t G3 = (uint32_t)G * G * G;
    const float Ginv = 1.0f / G;
    float t = hits_t[2 * r];
    const float t2 = hits_t[2 * r + 1];
    int s = 0;
    float t_resume = t; // t right after the last accepted sample (raymarching.cu:390)
    while (t < t2 && s < n_samples) {
        float x, y, z, dt;
        const float tcur = t;
        if (march_probe(c, bits, cascades, G, G3, Ginv, scale, (float)cascades, esf, max_samples, t, x, y, z, dt)) {
            const size_t o = (size_t)n * n_samples + s;
            xyzs[3 * o] = x; xyzs[3 * o + 1] = y; xyzs[3 * o + 2] = z;
            dirs[3 * o] = c.dx; dirs[3 * o + 1] = c.dy; dirs[3 * o + 2] = c.dz;
            ts[o] = tcur; deltas[o] = dt;
            t += dt; s++;
            t_resume = t;
        }
    }
    if (s > 0) hits_t[2 * r] = t_resume;
    n_eff[n] = s;
    if (state && s > 0)   // total_samples += N_eff_samples.sum() (rendering.py:77)
        atomicAdd(reinterpret_cast<unsigned long long*>(state + 6), (unsigned long long)s);
}

// ---- device-driven test-time rounds (rendering.py:46-133 without a host round trip per round) ----------------------
// state (int32[8]): [0] n_alive  [1] samples per ray of this round  [2] sum of [1] over the rounds so far  [3] done
//                   [4] rounds begun  [5] n_alive after the last compaction  [6:8] int64 total samples.
// The loop head `while samples < max_samples: if N_alive == 0: break; N_samples = max(min(N_rays // N_alive, 64), min)`
__global__ void test_round_begin_kernel(int32_t* __restrict__ state, int n_rays, int min_samples, int max_total)
{
    if (threadIdx.x != 0 || blockIdx.x != 0 || state[3]) return;
    const int n_alive = state[4] == 0 ? state[0] : state[5];   // [5]: the count ngp_alive_compact left behind the last round
    state[0] = n_alive;
    if (n_alive <= 0 || state[2] >= max_total) { state[3] = 1; state[1] = 0; return; }
    int ns = n_rays / n_alive;
    ns = ns < 64 ? ns : 64;
    ns = ns > min_samples ? ns : min_samples;
    state[1] = ns;
    state[2] += ns;
    state[4] += 1;
}

// alive_out = alive_in[alive_in >= 0] over the first state[0] entries, order kept (rendering.py:115); two launches:
// survivors per block of 1024, then every block sums the counts in front of it and writes its own survivors.
__global__ void __launch_bounds__(1024) alive_count_kernel(const int64_t* __restrict__ alive, const int32_t* __restrict__ state,
                                                            int32_t* __restrict__ counts)
{
    __shared__ int part[16];
    const int n_alive = state[3] ? 0 : state[0];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const bool keep = i < n_alive && alive[i] >= 0;
    const int c = __popcll(__ballot(keep));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int q = 0; q < 16; q++) t += part[q];
        counts[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(1024) alive_write_kernel(const int64_t* __restrict__ alive, int32_t* __restrict__ state,
                                                            const int32_t* __restrict__ counts, int n_blocks,
                                                            int64_t* __restrict__ alive_out)
{
    __shared__ int part[16];
    __shared__ int base_s;
    if (state[3]) return;
    const int n_alive = state[0];
    // survivors in front of this block (n_blocks <= a few hundred: one strided pass)
    int before = 0, total = 0;
    for (int b = threadIdx.x; b < n_blocks; b += 1024) {
        const int c = counts[b];
        total += c;
        if (b < (int)blockIdx.x) before += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o, 64); total += __shfl_xor(total, o, 64); }
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = before; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int q = 0; q < 16; q++) t += part[q];
        base_s = t;
    }
    __syncthreads();
    // (total over all waves, for the new n_alive: same reduction once more)
    __shared__ int tot_part[16];
    if ((threadIdx.x & 63) == 0) tot_part[threadIdx.x >> 6] = total;
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const bool keep = i < n_alive && alive[i] >= 0;
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int wcount[16];
    if (lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int q = 0; q < wave; q++) off += wcount[q];
    if (keep) alive_out[off + __popcll(m & ((1ull << lane) - 1ull))] = alive[i];
    if (blockIdx.x == gridDim.x - 1) {
        __syncthreads();    // (every thread of the block reaches this: no early exit above for this block)
        if (threadIdx.x == 0) {
            int t = 0;
            for (int q = 0; q < 16; q++) t += tot_part[q];
            state[5] = t;   // NOT state[0]: blocks of this launch that start later still read the old count from it;
                            // ngp_test_round_begin adopts [5] at the head of the next round
        }
    }
}

} // namespace

// ====================================================================== C ABI
extern "C" {

int ngp_ray_aabb_intersect(const float* rays_o, const float* rays_d, const float* centers,
                           const float* half_sizes, int n_rays, int n_voxels, int max_hits,
                           int32_t* hit_cnt, float* hits_t, int64_t* hits_idx, void* stream)
{
    if (n_rays < 0 || n_voxels < 0 || max_hits < 1) return NGP_EINVAL;
    if (n_rays == 0) return NGP_OK;
    if (!rays_o || !rays_d || !hit_cnt || !hits_t || !hits_idx || (n_voxels && (!centers || !half_sizes))) return NGP_EINVAL;
    hipLaunchKernelGGL(intersect_kernel<false>, dim3(ngp_blocks(n_rays, 256)), dim3(256), 0, (hipStream_t)stream,
                       rays_o, rays_d, centers, half_sizes, n_rays, n_voxels, max_hits, hit_cnt, hits_t, hits_idx);
    return ngp_check_launch();
}

int ngp_ray_sphere_intersect(const float* rays_o, const float* rays_d, const float* centers,
                             const float* radii, int n_rays, int n_spheres, int max_hits,
                             int32_t* hit_cnt, float* hits_t, int64_t* hits_idx, void* stream)
{
    if (n_rays < 0 || n_spheres < 0 || max_hits < 1) return NGP_EINVAL;
    if (n_rays == 0) return NGP_OK;
    if (!rays_o || !rays_d || !hit_cnt || !hits_t || !hits_idx || (n_spheres && (!centers || !radii))) return NGP_EINVAL;
    hipLaunchKernelGGL(intersect_kernel<true>, dim3(ngp_blocks(n_rays, 256)), dim3(256), 0, (hipStream_t)stream,
                       rays_o, rays_d, centers, radii, n_rays, n_spheres, max_hits, hit_cnt, hits_t, hits_idx);
    return ngp_check_launch();
}

int ngp_clamp_near(float* hits_t, int n_rays, int max_hits, float near_distance, void* stream)
{
    if (n_rays < 0 || max_hits < 1) return NGP_EINVAL;
    if (n_rays == 0) return NGP_OK;
    if (!hits_t) return NGP_EINVAL;
    hipLaunchKernelGGL(clamp_near_kernel, dim3(ngp_blocks(n_rays, 256)), dim3(256), 0, (hipStream_t)stream,
                       hits_t, n_rays, max_hits, near_distance);
    return ngp_check_launch();
}

int ngp_morton3D(const int32_t* coords, int n, int32_t* indices, void* stream)
{
    if (n < 0) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!coords || !indices) return NGP_EINVAL;
    hipLaunchKernelGGL(morton_kernel, dim3(ngp_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, coords, n, indices);
    return ngp_check_launch();
}

int ngp_morton3D_invert(const int32_t* indices, int n, int32_t* coords, void* stream)
{
    if (n < 0) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!coords || !indices) return NGP_EINVAL;
    hipLaunchKernelGGL(morton_invert_kernel, dim3(ngp_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, indices, n, coords);
    return ngp_check_launch();
}

int ngp_packbits(const float* density_grid, int n_bytes, float threshold, const float* threshold_dev,
                 uint8_t* density_bitfield, void* stream)
{
    if (n_bytes < 0) return NGP_EINVAL;
    if (n_bytes == 0) return NGP_OK;
    if (!density_grid || !density_bitfield || ((uintptr_t)density_grid & 15)) return NGP_EINVAL;
    hipLaunchKernelGGL(packbits_kernel, dim3(ngp_blocks(n_bytes, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)density_grid, n_bytes, threshold, threshold_dev, density_bitfield);
    return ngp_check_launch();
}

int ngp_grid_cell_points(const int32_t* coords, const float* noise, int n, int grid_size, float s,
                         float* xyzs_w, void* stream)
{
    if (n < 0 || grid_size < 2) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!coords || !noise || !xyzs_w) return NGP_EINVAL;
    hipLaunchKernelGGL(cell_points_kernel, dim3(ngp_blocks((int64_t)n * 3, 256)), dim3(256), 0, (hipStream_t)stream,
                       coords, noise, n * 3, grid_size, s, xyzs_w);
    return ngp_check_launch();
}

static int key_bits(int g3)
{
    int bits = 0;
    while ((1ll << bits) < (long long)g3) bits++;
    return bits;
}

int64_t ngp_grid_sample_workspace(int grid_size, int m)
{
    if (grid_size < 1 || grid_size > 1024 || m < 1) return NGP_EINVAL;
    const int64_t g3 = (int64_t)grid_size * grid_size * grid_size;
    const int64_t nb = (g3 + OCC_BLOCK - 1) / OCC_BLOCK;
    return 2 * nb + 4 + g3 + 8 * (int64_t)m + 3 * SAMPLE_TOP;
}

int ngp_grid_sample_cells(const float* density_grid_c, int grid_size, float density_threshold, int m, int64_t seed,
                          float s, int32_t* workspace, int32_t* indices, float* xyzs_w, void* stream)
{
    if (grid_size < 2 || grid_size > 1024 || m < 1) return NGP_EINVAL;
    if (!density_grid_c || !workspace || !indices || !xyzs_w) return NGP_EINVAL;
    const int64_t g3l = (int64_t)grid_size * grid_size * grid_size;
    if (g3l > (1ll << 30) || 2 * (int64_t)m > (1ll << 30)) return NGP_EINVAL;
    const int g3 = (int)g3l;
    const int nb = (g3 + OCC_BLOCK - 1) / OCC_BLOCK;
    const int bits = key_bits(g3);
    const int shift = bits > SAMPLE_TOP_BITS ? bits - SAMPLE_TOP_BITS : 0;          // key >> shift = top-level bucket
    const int sub_shift = shift > SAMPLE_SUB_BITS ? shift - SAMPLE_SUB_BITS : 0;    // next 13 bits order a bucket
    int32_t* block_counts = workspace;
    int32_t* block_offsets = block_counts + nb;
    int32_t* n_occ = block_offsets + nb;            // 4 ints (alignment)
    int32_t* occ_list = n_occ + 4;
    int32_t* keys = occ_list + g3;
    int32_t* sids = keys + 2 * (int64_t)m;
    int32_t* keys_b = sids + 2 * (int64_t)m;
    int32_t* sids_b = keys_b + 2 * (int64_t)m;
    int32_t* bucket_counts = sids_b + 2 * (int64_t)m;
    int32_t* bucket_offsets = bucket_counts + SAMPLE_TOP;
    int32_t* bucket_cursor = bucket_offsets + SAMPLE_TOP;
    const int n = 2 * m;
    const unsigned chunks = ngp_blocks(n, SAMPLE_CHUNK);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(occ_count_kernel, dim3(nb), dim3(256), 0, st, density_grid_c, g3, density_threshold, block_counts);
    // the scan also clears the bucket histogram of the sampling pass
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, st, block_counts, nb, block_offsets, n_occ,
                       bucket_counts, SAMPLE_TOP);
    hipLaunchKernelGGL(occ_compact_kernel, dim3(nb), dim3(256), 0, st, density_grid_c, g3, density_threshold,
                       block_offsets, occ_list);
    hipLaunchKernelGGL(sample_cells_kernel, dim3(chunks), dim3(256), 0, st, occ_list, n_occ, grid_size, m,
                       (uint64_t)seed, shift, keys, sids, bucket_counts);
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, st, bucket_counts, SAMPLE_TOP, bucket_offsets,
                       (int32_t*)nullptr, bucket_cursor, SAMPLE_TOP);
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(chunks), dim3(256), 0, st, keys, sids, n, shift, bucket_offsets,
                       bucket_cursor, keys_b, sids_b);
    hipLaunchKernelGGL(bucket_sort_points_kernel, dim3(SAMPLE_TOP), dim3(1024), 0, st, keys_b, sids_b, bucket_offsets, n,
                       sub_shift, grid_size, s, (uint64_t)seed, indices, xyzs_w);
    return ngp_check_launch();
}

int ngp_density_grid_scatter_max(float* density_grid_tmp_c, const int32_t* indices, const float* sigmas, int n, void* stream)
{
    if (n < 0) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!density_grid_tmp_c || !indices || !sigmas) return NGP_EINVAL;
    hipLaunchKernelGGL(grid_scatter_max_kernel, dim3(ngp_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       density_grid_tmp_c, indices, sigmas, n);
    return ngp_check_launch();
}

int ngp_density_grid_ema_threshold(float* density_grid, const float* density_grid_tmp, int n, float decay,
                                   float density_threshold, float* partials, float* threshold_out, void* stream)
{
    if (n < 0) return NGP_EINVAL;
    if (!threshold_out || !partials) return NGP_EINVAL;
    if (n > 0 && (!density_grid || !density_grid_tmp)) return NGP_EINVAL;
    const int blocks = 512;   // partials: 2 * 512 floats
    hipLaunchKernelGGL(grid_ema_stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, density_grid,
                       density_grid_tmp, n, decay, partials);
    hipLaunchKernelGGL(grid_threshold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, blocks,
                       density_threshold, threshold_out);
    return ngp_check_launch();
}

int ngp_density_grid_ema(float* density_grid, const float* density_grid_tmp, int n, float decay, void* stream)
{
    if (n < 0) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!density_grid || !density_grid_tmp) return NGP_EINVAL;
    hipLaunchKernelGGL(grid_ema_kernel, dim3(ngp_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       density_grid, density_grid_tmp, n, decay);
    return ngp_check_launch();
}

int ngp_raymarching_train(const float* rays_o, const float* rays_d, const float* hits_t,
                          const uint8_t* density_bitfield, int cascades, float scale,
                          float exp_step_factor, const float* noise, int grid_size,
                          int max_samples, int n_rays,
                          float* t_scratch, int32_t* ray_counts,
                          int64_t* rays_a, float* xyzs, float* dirs, float* deltas, float* ts,
                          int32_t* counter, int64_t sample_capacity, int zero_tail, void* stream)
{
    if (n_rays < 0 || cascades < 1 || grid_size < 1 || grid_size > 1024 || max_samples < 1) return NGP_EINVAL;
    if (!counter) return NGP_EINVAL;
    if (n_rays > 0 && (!rays_o || !rays_d || !hits_t || !density_bitfield || !noise || !t_scratch || !ray_counts ||
                       !rays_a || !xyzs || !dirs || !deltas || !ts)) return NGP_EINVAL;
    if (sample_capacity < (int64_t)n_rays * max_samples) return NGP_EINVAL; // worst case must fit
    hipStream_t st = (hipStream_t)stream;
#ifdef NGP_AB_VARIANTS
    static const bool lane_per_ray = getenv("NGP_MARCH_LANE_PER_RAY") != nullptr;   // A/B: the serial walk
    if (n_rays > 0 && lane_per_ray)
        hipLaunchKernelGGL(march_count_kernel, dim3(ngp_blocks(n_rays, 64)), dim3(64), 0, st,
                           rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise,
                           grid_size, max_samples, n_rays, t_scratch, ray_counts);
    else
#endif
    if (n_rays > 0)
        hipLaunchKernelGGL(march_wave_kernel, dim3(ngp_blocks(n_rays, 4)), dim3(256), 0, st,
                           rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise,
                           grid_size, max_samples, n_rays, t_scratch, ray_counts);
    hipLaunchKernelGGL(march_scan_kernel, dim3(1), dim3(1024), 0, st, ray_counts, n_rays, rays_a, counter);
    if (n_rays > 0)
        hipLaunchKernelGGL(march_expand_kernel, dim3(ngp_blocks((int64_t)n_rays * 64, 256)), dim3(256), 0, st,
                           rays_o, rays_d, t_scratch, rays_a, exp_step_factor, grid_size, scale, max_samples, n_rays,
                           xyzs, dirs, deltas, ts);
    if (zero_tail && sample_capacity > 0)
        hipLaunchKernelGGL(march_zero_tail_kernel, dim3(2048), dim3(256), 0, st, counter, sample_capacity,
                           xyzs, dirs, deltas, ts);
    return ngp_check_launch();
}

int ngp_raymarching_test(const float* rays_o, const float* rays_d, float* hits_t,
                         const int64_t* alive_indices, const uint8_t* density_bitfield,
                         int cascades, float scale, float exp_step_factor, int grid_size,
                         int max_samples, int n_samples, int n_alive,
                         float* xyzs, float* dirs, float* deltas, float* ts,
                         int32_t* n_eff_samples, void* stream)
{
    if (n_alive < 0 || cascades < 1 || grid_size < 1 || grid_size > 1024 || n_samples < 1) return NGP_EINVAL;
    if (n_alive == 0) return NGP_OK;
    if (!rays_o || !rays_d || !hits_t || !alive_indices || !density_bitfield || !xyzs || !dirs || !deltas || !ts ||
        !n_eff_samples) return NGP_EINVAL;
    hipLaunchKernelGGL(march_test_kernel, dim3(ngp_blocks(n_alive, 64)), dim3(64), 0, (hipStream_t)stream,
                       rays_o, rays_d, hits_t, alive_indices, density_bitfield, cascades, scale, exp_step_factor,
                       grid_size, max_samples, n_samples, n_alive, xyzs, dirs, deltas, ts, n_eff_samples,
                       (int32_t*)nullptr);
    return ngp_check_launch();
}

int ngp_test_round_begin(int32_t* state, int n_rays, int min_samples, int max_samples_total, void* stream)
{
    if (!state || n_rays < 1 || min_samples < 1 || max_samples_total < 1) return NGP_EINVAL;
    hipLaunchKernelGGL(test_round_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, n_rays, min_samples,
                       max_samples_total);
    return ngp_check_launch();
}

int ngp_raymarching_test_rounds(const float* rays_o, const float* rays_d, float* hits_t, const int64_t* alive_indices,
                                const uint8_t* density_bitfield, int cascades, float scale, float exp_step_factor,
                                int grid_size, int max_samples, int32_t* state, int n_alive_bound, float* xyzs,
                                float* dirs, float* deltas, float* ts, int32_t* n_eff_samples, void* stream)
{
    if (n_alive_bound < 0 || cascades < 1 || grid_size < 1 || grid_size > 1024) return NGP_EINVAL;
    if (n_alive_bound == 0) return NGP_OK;
    if (!state || !rays_o || !rays_d || !hits_t || !alive_indices || !density_bitfield || !xyzs || !dirs || !deltas || !ts ||
        !n_eff_samples) return NGP_EINVAL;
    hipLaunchKernelGGL(march_test_kernel, dim3(ngp_blocks(n_alive_bound, 64)), dim3(64), 0, (hipStream_t)stream,
                       rays_o, rays_d, hits_t, alive_indices, density_bitfield, cascades, scale, exp_step_factor,
                       grid_size, max_samples, 0, 0, xyzs, dirs, deltas, ts, n_eff_samples, state);
    return ngp_check_launch();
}

int ngp_alive_compact(const int64_t* alive_in, int32_t* state, int n_alive_bound, int32_t* block_counts,
                      int64_t* alive_out, void* stream)
{
    if (n_alive_bound < 0) return NGP_EINVAL;
    if (n_alive_bound == 0) return NGP_OK;
    if (!state || !alive_in || !block_counts || !alive_out) return NGP_EINVAL;
    const int blocks = (int)ngp_blocks(n_alive_bound, 1024);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(alive_count_kernel, dim3(blocks), dim3(1024), 0, st, alive_in, state, block_counts);
    hipLaunchKernelGGL(alive_write_kernel, dim3(blocks), dim3(1024), 0, st, alive_in, state, block_counts, blocks, alive_out);
    return ngp_check_launch();
}

} // extern "C"
