// Multiresolution hash grid (tcnn Grid/Hash encoding semantics) and the
// spherical-harmonics direction encoding.
//
// Work decomposition (MI355X-first, not tcnn's level-per-blockIdx.y):
//   an ITEM is one (sample, level) pair, items are numbered sample-major so the encoded output
//   (n, L*F) is item-contiguous: item i owns floats [i*F, (i+1)*F).
//   forward / input-grad: LPI = F/4 lanes per item, each lane moves one 16-byte quarter/half
//     row per corner (`global_load_dwordx4`), so a wave's output store is one contiguous
//     1 KiB run and every table access is a whole 16/32-byte row segment.
//   param-grad: F lanes per item, one fp32 atomic per lane per corner, so each
//     `global_atomic_add_f32` wave-instruction covers whole 32-byte rows (8 rows per
//     instruction at F=8) — the shape the memory-side atomic units want (MI355X_MICROARCH
//     "Global float atomics": one lane per row is ~17x slower than contiguous segments).
//     Lanes whose upstream gradient is exactly zero (samples behind the early-termination
//     point, volumerendering.cu:111) issue no atomics.
#include "common.h"
#include <stdlib.h>

namespace {

struct GridMeta {
    uint32_t n_levels, n_features;
    uint32_t offset[NGP_MAX_LEVELS];
    uint32_t size[NGP_MAX_LEVELS];   // rows in the level
    uint32_t res[NGP_MAX_LEVELS];
    uint32_t flags[NGP_MAX_LEVELS];  // bit0: hashed, bit1: size is a power of two
    float scale[NGP_MAX_LEVELS];
};

struct LevelInfo {
    uint32_t offset, size, res, flags;
    float scale;
};

__device__ __forceinline__ LevelInfo level_info(const GridMeta& m, uint32_t l)
{
    LevelInfo li;
    li.offset = m.offset[l]; li.size = m.size[l]; li.res = m.res[l]; li.flags = m.flags[l]; li.scale = m.scale[l];
    return li;
}

__device__ __forceinline__ uint32_t row_index(const LevelInfo& li, uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t idx;
    if (li.flags & 1u) {
        idx = x ^ (y * 2654435761u) ^ (z * 805459861u);
        idx = (li.flags & 2u) ? (idx & (li.size - 1u)) : (idx % li.size);
    } else {
        idx = x + y * li.res + z * li.res * li.res;
        if (idx >= li.size) idx %= li.size;
    }
    return li.offset + idx;
}

struct Cell {
    uint32_t g[3];
    float w[3];
};

__device__ __forceinline__ Cell cell_of(const float* __restrict__ x, int64_t sample, float scale)
{
    Cell c;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float p = fmaf(scale, x[3 * sample + k], 0.5f);
        const float fl = floorf(p);
        c.g[k] = (uint32_t)(int)fl;
        c.w[k] = p - fl;
    }
    return c;
}

template <int V> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float2 T; };
template <> struct VecT<4> { typedef float4 T; };

template <int V> __device__ __forceinline__ void vec_fma(float (&acc)[V], float w, const typename VecT<V>::T& v);
template <> __device__ __forceinline__ void vec_fma<1>(float (&acc)[1], float w, const float& v) { acc[0] = fmaf(w, v, acc[0]); }
template <> __device__ __forceinline__ void vec_fma<2>(float (&acc)[2], float w, const float2& v)
{
    acc[0] = fmaf(w, v.x, acc[0]); acc[1] = fmaf(w, v.y, acc[1]);
}
template <> __device__ __forceinline__ void vec_fma<4>(float (&acc)[4], float w, const float4& v)
{
    acc[0] = fmaf(w, v.x, acc[0]); acc[1] = fmaf(w, v.y, acc[1]);
    acc[2] = fmaf(w, v.z, acc[2]); acc[3] = fmaf(w, v.w, acc[3]);
}
template <int V> __device__ __forceinline__ float vec_dot(const float (&g)[V], const typename VecT<V>::T& a, const typename VecT<V>::T& b);
template <> __device__ __forceinline__ float vec_dot<1>(const float (&g)[1], const float& a, const float& b) { return g[0] * (a - b); }
template <> __device__ __forceinline__ float vec_dot<2>(const float (&g)[2], const float2& a, const float2& b)
{
    return g[0] * (a.x - b.x) + g[1] * (a.y - b.y);
}
template <> __device__ __forceinline__ float vec_dot<4>(const float (&g)[4], const float4& a, const float4& b)
{
    return g[0] * (a.x - b.x) + g[1] * (a.y - b.y) + g[2] * (a.z - b.z) + g[3] * (a.w - b.w);
}

// ------------------------------------------------------------------ forward (H1)
template <int F, int EXP = 0>   // EXP (A/B build): 1 all rows from the first 1,024 of the table (cache hits), 2 no output store
__global__ void __launch_bounds__(256) grid_fwd_kernel(GridMeta meta, const float* __restrict__ table,
                                                       const float* __restrict__ x, int64_t n_items,
                                                       float* __restrict__ y, int64_t ldy)
{
    constexpr int V = F >= 4 ? 4 : F;   // floats per lane
    constexpr int LPI = F / V;          // lanes per item
    typedef typename VecT<V>::T vec_t;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t item = tid / LPI;
    const int sub = (int)(tid % LPI);
    if (item >= n_items) return;
    const uint32_t L = meta.n_levels;
    const int64_t sample = item / L;
    const uint32_t level = (uint32_t)(item - sample * L);
    const LevelInfo li = level_info(meta, level);
    const Cell c = cell_of(x, sample, li.scale);

    uint32_t rows[8];
    float wts[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t cx = k & 1, cy = (k >> 1) & 1, cz = (k >> 2) & 1;
        rows[k] = row_index(li, c.g[0] + cx, c.g[1] + cy, c.g[2] + cz);
        if (EXP & 1) rows[k] &= 1023u;
        wts[k] = (cx ? c.w[0] : 1 - c.w[0]) * (cy ? c.w[1] : 1 - c.w[1]) * (cz ? c.w[2] : 1 - c.w[2]);
    }
    vec_t vals[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
        vals[k] = *reinterpret_cast<const vec_t*>(table + (size_t)rows[k] * F + sub * V);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; j++) acc[j] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) vec_fma<V>(acc, wts[k], vals[k]);
    vec_t out;
    float* o = reinterpret_cast<float*>(&out);
#pragma unroll
    for (int j = 0; j < V; j++) o[j] = acc[j];
    if ((EXP & 2) && acc[0] != 123.456f) return;
    *reinterpret_cast<vec_t*>(y + sample * ldy + level * F + sub * V) = out;
}

// ------------------------------------------------------------------ ray-coherent tile gathers (H1, H3; F = 8)
// Samples arrive in ray order (the marcher's), and consecutive samples of a ray stand in the same cell on the
// coarse levels (step sqrt(3)/1024 against cells of 1/16 .. 1/1024): the item-per-(sample, level) kernel above
// fetches those rows again for every sample.  Here a WAVE owns S = 32 CONSECUTIVE samples x 4 levels (the four
// waves of a workgroup cover the 16 levels of one sample tile), and per level
//   phase 1  lane = (sample, half row): cell of the sample; a lane whose cell differs from the previous sample's
//            opens a run; ballot + popcount number the runs, the run leaders write their cell into a list (LDS);
//   phase 2  lane = (unique cell, corner, half row): ONE 16-byte load per (unique cell, corner, half) instead of
//            one per (sample, corner, half), committed to a wave-private LDS image [cell][half][corner] (padded:
//            conflict-free for the reads of phase 3);
//   phase 3  lane = (sample, half row): the sample's eight corner pieces come back from LDS (lanes of one run read
//            the same address: broadcast), trilinear blend in the forward kernel's operation order (bit-identical).
// The loads of level i+1 are in flight under phase 3 of level i.  The tile's outputs (32 samples x 4 levels x 8
// floats) leave through the same LDS image transposed, so that every store instruction writes whole 128-byte
// segments of the (n, L*F) rows.
namespace tile {
constexpr int F = 8;
constexpr int LV = 4;                 // levels per wave
constexpr int S = 32;                 // samples per wave

#ifdef NGP_AB_VARIANTS   // the LDS-staged variant (measured, not shipped: DESIGN.md section 4) lives in the A/B build only
constexpr int CELL = 68;              // dwords per staged cell: [half 2][corner 8][4] + 4 (bank spread, see phase 3)
constexpr int LIST = 4 * S;           // dwords per cell list: (gx, gy, gz, -) per run
constexpr int WAVE_LDS = 2 * LIST + S * CELL;
struct Run {
    float w0, w1, w2;
    uint32_t u;                       // index of the lane's run (= unique cell) among the wave's
    uint32_t U;                       // number of runs, wave-uniform
};

__device__ __forceinline__ Run phase1(const LevelInfo& li, float px, float py, float pz, int lane, uint32_t* list)
{
    const float p0 = fmaf(li.scale, px, 0.5f), p1 = fmaf(li.scale, py, 0.5f), p2 = fmaf(li.scale, pz, 0.5f);
    const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
    const int g0 = (int)f0, g1 = (int)f1, g2 = (int)f2;
    const int q0 = __shfl_up(g0, 2), q1 = __shfl_up(g1, 2), q2 = __shfl_up(g2, 2);
    const bool lead = !(lane & 1) && (lane < 2 || g0 != q0 || g1 != q1 || g2 != q2);
    const unsigned long long m = __ballot(lead);
    Run r;
    r.w0 = p0 - f0; r.w1 = p1 - f1; r.w2 = p2 - f2;
    r.u = (uint32_t)__popcll(m & ((2ull << lane) - 1ull)) - 1u;   // leaders at or below this lane (lane 63: mask = ~0)
    r.U = (uint32_t)__popcll(m);
    if (lead) *reinterpret_cast<uint4*>(list + 4 * r.u) = make_uint4((uint32_t)g0, (uint32_t)g1, (uint32_t)g2, 0u);
    return r;
}

#endif

// row of a corner for the two kinds of level the tile kernels take (the launcher sends layouts with a hashed level whose
// size is no power of two, or tables of 4 GiB and more, to the item-per-(sample, level) kernels): byte offset into the table
template <bool HASHED>
__device__ __forceinline__ uint32_t row_offset(const LevelInfo& li, uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t idx;
    if (HASHED)
        idx = (x ^ (y * 2654435761u) ^ (z * 805459861u)) & (li.size - 1u);
    else {
        idx = x + li.res * (y + li.res * z);        // positions in [0, 1]: < 2 res^3 <= 2 size, one subtraction is the modulo
        if (idx >= li.size) {
            idx -= li.size;
            if (idx >= li.size) idx %= li.size;     // a position outside the unit cube (any caller of the C ABI): stay inside the level
        }
    }
    return (li.offset + idx) * (uint32_t)(F * sizeof(float));
}

#ifdef NGP_AB_VARIANTS
// phase 2a: the loads of one level (at most 8 per lane: 32 cells x 16 pieces / 64 lanes), into registers
template <bool HASHED>
__device__ __forceinline__ void issue_impl(const LevelInfo& li, const char* __restrict__ table, const uint32_t* list,
                                           uint32_t U, int lane, float4 (&R)[8])
{
    const uint32_t cq = (uint32_t)lane >> 4, corner = ((uint32_t)lane >> 1) & 7u, half = (uint32_t)lane & 1u;
    const uint32_t cx = corner & 1u, cy = (corner >> 1) & 1u, cz = corner >> 2;
    const uint32_t* mine = list + 4 * cq;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     // (a conditional store into R[it] would send R to scratch)
        if ((uint32_t)(it * 4) < U) {                       // wave-uniform
            if (it * 4 + cq < U) {
                const uint32_t gx = mine[16 * it], gy = mine[16 * it + 1], gz = mine[16 * it + 2];
                const uint32_t off = row_offset<HASHED>(li, gx + cx, gy + cy, gz + cz) + half * 16u;
                v = *reinterpret_cast<const float4*>(table + off);
            }
        }
        R[it] = v;
    }
}

__device__ __forceinline__ void issue(const LevelInfo& li, const float* __restrict__ table, const uint32_t* list,
                                      uint32_t U, int lane, float4 (&R)[8])
{
    if (li.flags & 1u) issue_impl<true>(li, reinterpret_cast<const char*>(table), list, U, lane, R);
    else issue_impl<false>(li, reinterpret_cast<const char*>(table), list, U, lane, R);
}

// phase 2b: registers -> LDS image
__device__ __forceinline__ void commit(float* cells, uint32_t U, int lane, const float4 (&R)[8])
{
    const uint32_t cq = (uint32_t)lane >> 4, corner = ((uint32_t)lane >> 1) & 7u, half = (uint32_t)lane & 1u;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        if ((uint32_t)(it * 4) < U) {
            const uint32_t c = it * 4 + cq;
            if (c < U) *reinterpret_cast<float4*>(cells + c * CELL + half * 32 + corner * 4) = R[it];
        }
    }
}

#endif

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace tile

#ifdef NGP_AB_VARIANTS
__global__ void __launch_bounds__(256, 4) grid_fwd_tile_kernel(GridMeta meta, const float* __restrict__ table,
                                                            const float* __restrict__ x, int64_t n,
                                                            float* __restrict__ y, int64_t ldy)
{
    using namespace tile;
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][WAVE_LDS];
    uint32_t* W = lds[threadIdx.x >> 6];
    float* cells = reinterpret_cast<float*>(W + 2 * LIST);
    const uint32_t L = meta.n_levels;
    const uint32_t wpc = (L + LV - 1) / LV;                 // waves per sample tile
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t chunk = wave_global / wpc;
    const uint32_t lg = (uint32_t)(wave_global - chunk * wpc);   // scalar: the level records come by s_load
    const int lane = threadIdx.x & 63;
    const int64_t s0 = chunk * S;
    if (s0 >= n) return;
    const int j = lane >> 1, half = lane & 1;
    const int64_t sj = s0 + j < n ? s0 + j : n - 1;         // tail lanes repeat the last sample (same run, no extra loads)
    const float px = x[3 * sj], py = x[3 * sj + 1], pz = x[3 * sj + 2];
    const uint32_t nlv = L - lg * LV < (uint32_t)LV ? L - lg * LV : (uint32_t)LV;

    float4 R[8];
    float out[LV][4];
    LevelInfo li = level_info(meta, lg * LV);
    Run cur = phase1(li, px, py, pz, lane, W);
    wave_sync();
    issue(li, table, W, cur.U, lane, R);
#pragma unroll
    for (int i = 0; i < LV; i++) {
        if ((uint32_t)i < nlv) {                            // wave-uniform
            commit(cells, cur.U, lane, R);
            Run nxt = cur;
            if ((uint32_t)(i + 1) < nlv) {
                li = level_info(meta, lg * LV + i + 1);
                uint32_t* list = W + ((i + 1) & 1) * LIST;
                nxt = phase1(li, px, py, pz, lane, list);
                wave_sync();
                issue(li, table, list, nxt.U, lane, R);
            } else
                wave_sync();
            const float* cb = cells + cur.u * CELL + half * 32;
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float4 v = *reinterpret_cast<const float4*>(cb + k * 4);
                const float wt = ((k & 1) ? cur.w0 : 1 - cur.w0) * ((k & 2) ? cur.w1 : 1 - cur.w1) * ((k & 4) ? cur.w2 : 1 - cur.w2);
                vec_fma<4>(acc, wt, v);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) out[i][q] = acc[q];
            cur = nxt;
            wave_sync();                                    // phase 3's reads before the next commit
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) out[i][q] = 0.0f;
        }
    }
    // transposition: lane (sample j, half) -> lane (sample it*8 + lane/8, 16-byte piece lane%8 of the 128 bytes)
#pragma unroll
    for (int i = 0; i < LV; i++)
        *reinterpret_cast<float4*>(cells + j * 32 + i * 8 + half * 4) = make_float4(out[i][0], out[i][1], out[i][2], out[i][3]);
    wave_sync();
    const int piece = lane & 7;
    if ((lg * LV + (piece >> 1)) < L) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int js = it * 8 + (lane >> 3);
            if (s0 + js < n)
                *reinterpret_cast<float4*>(y + (s0 + js) * ldy + lg * (LV * F) + piece * 4) =
                    *reinterpret_cast<const float4*>(cells + js * 32 + piece * 4);
        }
    }
}

// input gradient on the same tiles: phase 3 forms the three partial derivatives from the staged corners, the four
// level groups of a sample tile meet in LDS and the tile's (32, 3) block of dL_dx leaves as one contiguous store.
#ifndef NGP_TILE_BI_WAVES
#define NGP_TILE_BI_WAVES 4
#endif
__global__ void __launch_bounds__(256, NGP_TILE_BI_WAVES) grid_bwd_input_tile_kernel(GridMeta meta, const float* __restrict__ table,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ dL_dy, int64_t lddy,
                                                                  int64_t n, float* __restrict__ dL_dx)
{
    using namespace tile;
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][WAVE_LDS];
    uint32_t* W = lds[threadIdx.x >> 6];
    float* cells = reinterpret_cast<float*>(W + 2 * LIST);
    float* red = reinterpret_cast<float*>(W);               // the wave's (32, 3) partial sums, over its dead cell lists
    const uint32_t L = meta.n_levels;
    const uint32_t wpc = (L + LV - 1) / LV;                 // 1, 2 or 4 (checked by the launcher)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + wv;
    const int64_t chunk = wave_global / wpc;
    const uint32_t lg = (uint32_t)(wave_global - chunk * wpc);
    const int lane = threadIdx.x & 63;
    const int64_t s0 = chunk * S;
    const bool active = s0 < n;                             // wave-uniform; no early exit: the block meets at a barrier
    const int j = lane >> 1, half = lane & 1;
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
    if (active) {
        const int64_t sj = s0 + j < n ? s0 + j : n - 1;
        const float px = x[3 * sj], py = x[3 * sj + 1], pz = x[3 * sj + 2];
        const uint32_t nlv = L - lg * LV < (uint32_t)LV ? L - lg * LV : (uint32_t)LV;
        // the tile's upstream gradients: whole 128-byte segments in, (sample, half) pieces out of LDS
        float4 go[LV];
        {
            const int piece = lane & 7;
            const bool have = (lg * LV + (piece >> 1)) < L;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int js = it * 8 + (lane >> 3);
                const int64_t sg = s0 + js < n ? s0 + js : n - 1;
                const float4 v = have ? *reinterpret_cast<const float4*>(dL_dy + sg * lddy + lg * (LV * F) + piece * 4)
                                      : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                *reinterpret_cast<float4*>(cells + js * 32 + piece * 4) = v;
            }
            wave_sync();
#pragma unroll
            for (int i = 0; i < LV; i++) go[i] = *reinterpret_cast<const float4*>(cells + j * 32 + i * 8 + half * 4);
            wave_sync();
        }
        float4 R[8];
        LevelInfo li = level_info(meta, lg * LV);
        Run cur = phase1(li, px, py, pz, lane, W);
        wave_sync();
        issue(li, table, W, cur.U, lane, R);
#pragma unroll
        for (int i = 0; i < LV; i++) {
            if ((uint32_t)i < nlv) {
                const float scale = li.scale;
                commit(cells, cur.U, lane, R);
                Run nxt = cur;
                if ((uint32_t)(i + 1) < nlv) {
                    li = level_info(meta, lg * LV + i + 1);
                    uint32_t* list = W + ((i + 1) & 1) * LIST;
                    nxt = phase1(li, px, py, pz, lane, list);
                    wave_sync();
                    issue(li, table, list, nxt.U, lane, R);
                } else
                    wave_sync();
                const float* cb = cells + cur.u * CELL + half * 32;
                const float g4[4] = {go[i].x, go[i].y, go[i].z, go[i].w};
                const float wx0 = 1 - cur.w0, wx1 = cur.w0, wy0 = 1 - cur.w1, wy1 = cur.w1, wz0 = 1 - cur.w2, wz1 = cur.w2;
                // one z plane of corners at a time (16 registers of corner data instead of 32): d/dx and d/dy inside the
                // plane, d/dz as the plane z1 arrives and replaces z0 corner by corner
                float4 v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const float4*>(cb + k * 4);
                float dx = wz0 * (wy0 * vec_dot<4>(g4, v[1], v[0]) + wy1 * vec_dot<4>(g4, v[3], v[2]));
                float dy = wz0 * (wx0 * vec_dot<4>(g4, v[2], v[0]) + wx1 * vec_dot<4>(g4, v[3], v[1]));
                float dz = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float4 t = *reinterpret_cast<const float4*>(cb + (k + 4) * 4);
                    dz = fmaf(((k & 1) ? wx1 : wx0) * ((k & 2) ? wy1 : wy0), vec_dot<4>(g4, t, v[k]), dz);
                    v[k] = t;
                }
                dx = fmaf(wz1, wy0 * vec_dot<4>(g4, v[1], v[0]) + wy1 * vec_dot<4>(g4, v[3], v[2]), dx);
                dy = fmaf(wz1, wx0 * vec_dot<4>(g4, v[2], v[0]) + wx1 * vec_dot<4>(g4, v[3], v[1]), dy);
                gx = fmaf(dx, scale, gx); gy = fmaf(dy, scale, gy); gz = fmaf(dz, scale, gz);
                cur = nxt;
                wave_sync();
            }
        }
        gx += __shfl_xor(gx, 1); gy += __shfl_xor(gy, 1); gz += __shfl_xor(gz, 1);
        if (!half) { red[3 * j] = gx; red[3 * j + 1] = gy; red[3 * j + 2] = gz; }
    }
    __syncthreads();
    // thread t: tile t / 96 of the block, float t % 96 of its (32, 3) block
    const int tiles = 4 / (int)wpc;
    for (int t = threadIdx.x; t < tiles * S * 3; t += 256) {
        const int tl = t / (S * 3), r = t - tl * (S * 3);
        const int64_t c0 = ((int64_t)blockIdx.x * 4 + tl * (int)wpc) / wpc * S;   // first sample of that tile
        if (c0 < n && c0 + r / 3 < n) {
            float sum = 0.0f;
            for (uint32_t w = 0; w < wpc; w++) sum += reinterpret_cast<const float*>(lds[tl * wpc + w])[r];
            dL_dx[3 * c0 + r] = sum;
        }
    }
}

#endif   // NGP_AB_VARIANTS (staged variant)

// ------------------------------------------------------------------ run-leader gathers (H1, H3; F = 8), no cell staging
// Same tiles (a wave = 32 consecutive samples x 4 levels, lane = (sample, half row)), but the unique cells are not
// compacted: only the lanes that OPEN a run (first sample of a run of equal cells) load their eight corner pieces,
// and the other lanes of the run fetch them from the leader's registers with ds_bpermute (32 dwords per level; none
// when every sample opens its own run, the usual case on the finest levels).  No LDS image of the cells, so the
// kernel keeps 8 waves per SIMD and as many loads in flight as the item-per-(sample, level) kernel, with the
// duplicate requests of a run removed.
namespace run {
constexpr int F = 8, LV = 4, S = 32;

struct Lead {
    float w0, w1, w2;
    int g0, g1, g2;
    bool lead;                         // this lane's sample opens a run
    int src;                           // lane that holds the lane's corner pieces (the run leader's lane of the same half)
    bool all;                          // every sample opens its own run (wave-uniform)
};

__device__ __forceinline__ Lead phase1(float scale, float px, float py, float pz, int lane)
{
    Lead r;
    const float p0 = fmaf(scale, px, 0.5f), p1 = fmaf(scale, py, 0.5f), p2 = fmaf(scale, pz, 0.5f);
    const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
    r.g0 = (int)f0; r.g1 = (int)f1; r.g2 = (int)f2;
    r.w0 = p0 - f0; r.w1 = p1 - f1; r.w2 = p2 - f2;
    const int q0 = __shfl_up(r.g0, 2), q1 = __shfl_up(r.g1, 2), q2 = __shfl_up(r.g2, 2);
    r.lead = lane < 2 || r.g0 != q0 || r.g1 != q1 || r.g2 != q2;
    const unsigned long long m = __ballot(r.lead);            // both halves of a sample agree: bits come in pairs
    r.all = m == ~0ull;
    const unsigned long long even = m & 0x5555555555555555ull & ((2ull << lane) - 1ull);   // leaders' even lanes <= lane
    r.src = (63 - __clzll((long long)even)) | (lane & 1);
    return r;
}

template <bool HASHED>
__device__ __forceinline__ void load8(const LevelInfo& li, const char* __restrict__ table, const Lead& c, int half, float4 (&R)[8])
{
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t off = tile::row_offset<HASHED>(li, (uint32_t)c.g0 + (k & 1), (uint32_t)c.g1 + ((k >> 1) & 1),
                                                       (uint32_t)c.g2 + (k >> 2)) + (uint32_t)half * 16u;
        R[k] = *reinterpret_cast<const float4*>(table + off);
    }
}

// the eight corner pieces of the lane's cell: loaded by run leaders, handed to the rest of the run
__device__ __forceinline__ void corners(const LevelInfo& li, const float* __restrict__ table, const Lead& c, int half,
                                        float4 (&R)[8])
{
#pragma unroll
    for (int k = 0; k < 8; k++) R[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (c.lead) {
        if (li.flags & 1u) load8<true>(li, reinterpret_cast<const char*>(table), c, half, R);
        else load8<false>(li, reinterpret_cast<const char*>(table), c, half, R);
    }
    if (!c.all) {                                            // wave-uniform
        const int a = c.src << 2;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            R[k].x = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(R[k].x)));
            R[k].y = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(R[k].y)));
            R[k].z = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(R[k].z)));
            R[k].w = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(R[k].w)));
        }
    }
}
}  // namespace run

__global__ void __launch_bounds__(256) grid_fwd_run_kernel(GridMeta meta, const float* __restrict__ table,
                                                           const float* __restrict__ x, int64_t n,
                                                           float* __restrict__ y, int64_t ldy)
{
    using namespace run;
    __shared__ __attribute__((aligned(16))) float ost[4][S * 32];   // per wave: the tile's (32, 32) outputs, transposed out
    const uint32_t L = meta.n_levels;
    const uint32_t wpc = (L + LV - 1) / LV;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + wv;
    // sample-major: the four waves of a workgroup are the four level groups of one tile.  (Level-group-major order — all
    // tiles for levels 0-3, then 4-7, ... so that a level's rows are re-touched within microseconds — was measured and is
    // no faster: the gathers are bound by the number of 64-byte L1 misses in flight per CU, not by where they are served.)
    const int64_t chunk = wave_global / wpc;
    const uint32_t lg = (uint32_t)(wave_global - chunk * wpc);
    const int lane = threadIdx.x & 63;
    const int64_t s0 = chunk * S;
    if (s0 >= n) return;
    const int j = lane >> 1, half = lane & 1;
    const int64_t sj = s0 + j < n ? s0 + j : n - 1;
    const float px = x[3 * sj], py = x[3 * sj + 1], pz = x[3 * sj + 2];
    float* st = ost[wv];
#pragma unroll
    for (int i = 0; i < LV; i++) {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (lg * LV + i < L) {                                // wave-uniform
            const LevelInfo li = level_info(meta, lg * LV + i);
            const Lead c = phase1(li.scale, px, py, pz, lane);
            float4 R[8];
            corners(li, table, c, half, R);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float wt = ((k & 1) ? c.w0 : 1 - c.w0) * ((k & 2) ? c.w1 : 1 - c.w1) * ((k & 4) ? c.w2 : 1 - c.w2);
                vec_fma<4>(acc, wt, R[k]);
            }
        }
        *reinterpret_cast<float4*>(st + j * 32 + i * 8 + half * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    tile::wave_sync();
    const int piece = lane & 7;
    if ((lg * LV + (piece >> 1)) < L) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int js = it * 8 + (lane >> 3);
            if (s0 + js < n)
                *reinterpret_cast<float4*>(y + (s0 + js) * ldy + lg * (LV * F) + piece * 4) =
                    *reinterpret_cast<const float4*>(st + js * 32 + piece * 4);
        }
    }
}

__global__ void __launch_bounds__(256) grid_bwd_input_run_kernel(GridMeta meta, const float* __restrict__ table,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ dL_dy, int64_t lddy,
                                                                 int64_t n, float* __restrict__ dL_dx)
{
    using namespace run;
    __shared__ __attribute__((aligned(16))) float gst[4][S * 32];   // per wave: the tile's (32, 32) upstream gradients
    __shared__ float red[4][S * 3];
    const uint32_t L = meta.n_levels;
    const uint32_t wpc = (L + LV - 1) / LV;                   // 1, 2 or 4 (checked by the launcher)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + wv;
    const int64_t chunk = wave_global / wpc;
    const uint32_t lg = (uint32_t)(wave_global - chunk * wpc);
    const int lane = threadIdx.x & 63;
    const int64_t s0 = chunk * S;
    const bool active = s0 < n;                               // wave-uniform; the block meets at a barrier below
    const int j = lane >> 1, half = lane & 1;
    if (active) {
        float* st = gst[wv];
        const int64_t sj = s0 + j < n ? s0 + j : n - 1;
        const float px = x[3 * sj], py = x[3 * sj + 1], pz = x[3 * sj + 2];
        {
            const int piece = lane & 7;
            const bool have = (lg * LV + (piece >> 1)) < L;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int js = it * 8 + (lane >> 3);
                const int64_t sg = s0 + js < n ? s0 + js : n - 1;
                const float4 v = have ? *reinterpret_cast<const float4*>(dL_dy + sg * lddy + lg * (LV * F) + piece * 4)
                                      : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                *reinterpret_cast<float4*>(st + js * 32 + piece * 4) = v;
            }
            tile::wave_sync();
        }
        float gx = 0.0f, gy = 0.0f, gz = 0.0f;
#pragma unroll
        for (int i = 0; i < LV; i++) {
            if (lg * LV + i < L) {
                const LevelInfo li = level_info(meta, lg * LV + i);
                const Lead c = phase1(li.scale, px, py, pz, lane);
                float4 v[8];
                corners(li, table, c, half, v);
                const float4 go = *reinterpret_cast<const float4*>(st + j * 32 + i * 8 + half * 4);
                const float g4[4] = {go.x, go.y, go.z, go.w};
                const float wx0 = 1 - c.w0, wx1 = c.w0, wy0 = 1 - c.w1, wy1 = c.w1, wz0 = 1 - c.w2, wz1 = c.w2;
                const float dx = wy0 * wz0 * vec_dot<4>(g4, v[1], v[0]) + wy1 * wz0 * vec_dot<4>(g4, v[3], v[2]) +
                                 wy0 * wz1 * vec_dot<4>(g4, v[5], v[4]) + wy1 * wz1 * vec_dot<4>(g4, v[7], v[6]);
                const float dy = wx0 * wz0 * vec_dot<4>(g4, v[2], v[0]) + wx1 * wz0 * vec_dot<4>(g4, v[3], v[1]) +
                                 wx0 * wz1 * vec_dot<4>(g4, v[6], v[4]) + wx1 * wz1 * vec_dot<4>(g4, v[7], v[5]);
                const float dz = wx0 * wy0 * vec_dot<4>(g4, v[4], v[0]) + wx1 * wy0 * vec_dot<4>(g4, v[5], v[1]) +
                                 wx0 * wy1 * vec_dot<4>(g4, v[6], v[2]) + wx1 * wy1 * vec_dot<4>(g4, v[7], v[3]);
                gx = fmaf(dx, li.scale, gx); gy = fmaf(dy, li.scale, gy); gz = fmaf(dz, li.scale, gz);
            }
        }
        gx += __shfl_xor(gx, 1); gy += __shfl_xor(gy, 1); gz += __shfl_xor(gz, 1);
        if (!half) { red[wv][3 * j] = gx; red[wv][3 * j + 1] = gy; red[wv][3 * j + 2] = gz; }
    }
    __syncthreads();
    // thread t: tile t / 96 of the block, float t % 96 of its (32, 3) block of dL_dx
    const int tiles = 4 / (int)wpc;
    for (int t = threadIdx.x; t < tiles * S * 3; t += 256) {
        const int tl = t / (S * 3), r = t - tl * (S * 3);
        const int64_t c0 = ((int64_t)blockIdx.x * 4 + tl * (int)wpc) / wpc * S;
        if (c0 < n && c0 + r / 3 < n) {
            float sum = 0.0f;
            for (uint32_t w = 0; w < wpc; w++) sum += red[tl * wpc + w][r];
            dL_dx[3 * c0 + r] = sum;
        }
    }
}

// ------------------------------------------------------------------ param gradient (H2)
// The three kernels below are superseded by grid_bwd_param_slide_kernel and are compiled only into the A/B build
// (NGP_AB_VARIANTS=1 python -m instant-ngp-pp_amd.build; tools/grid_microbench.py times them against the product kernel).
#ifdef NGP_AB_VARIANTS
template <int F>
__global__ void __launch_bounds__(256) grid_bwd_param_kernel(GridMeta meta, const float* __restrict__ x,
                                                             const float* __restrict__ dL_dy, int64_t lddy,
                                                             int64_t n_items, float* __restrict__ dtable)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t item = tid / F;
    const int f = (int)(tid % F);
    if (item >= n_items) return;
    const uint32_t L = meta.n_levels;
    const int64_t sample = item / L;
    const uint32_t level = (uint32_t)(item - sample * L);
    const float g = dL_dy[sample * lddy + level * F + f];
    if (g == 0.0f) return; // adds nothing: skip the 8 atomics
    const LevelInfo li = level_info(meta, level);
    const Cell c = cell_of(x, sample, li.scale);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t cx = k & 1, cy = (k >> 1) & 1, cz = (k >> 2) & 1;
        const uint32_t row = row_index(li, c.g[0] + cx, c.g[1] + cy, c.g[2] + cz);
        const float w = (cx ? c.w[0] : 1 - c.w[0]) * (cy ? c.w[1] : 1 - c.w[1]) * (cz ? c.w[2] : 1 - c.w[2]);
        atomicAdd(dtable + (size_t)row * F + f, w * g);
    }
}

// Param gradient with run merging.  A wave owns CHUNK consecutive samples (ray order) for
// 64/F consecutive levels: lane = (level_in_wave, feature).  Each lane walks the chunk serially,
// keeping the 8 corner sums of its feature in registers while the sample stays in the same grid
// cell (consecutive samples of a ray are sqrt(3)/1024 apart, so at coarse levels tens of samples
// share a cell), and flushes them with one atomic per corner when the cell changes.  This cuts
// the number of memory-side atomic requests — the resource that bounds this kernel — by the
// run length, and removes the same-address pile-up on the small coarse levels.
template <int F, int CHUNK>
__global__ void __launch_bounds__(256) grid_bwd_param_merge_kernel(GridMeta meta, const float* __restrict__ x,
                                                                   const float* __restrict__ dL_dy, int64_t lddy,
                                                                   int64_t n, float* __restrict__ dtable)
{
    constexpr int LV = 64 / F;       // levels per wave
    constexpr int SUB = 8;           // samples whose loads are issued together
    const uint32_t L = meta.n_levels;
    const uint32_t waves_per_chunk = (L + LV - 1) / LV;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t chunk = __builtin_amdgcn_readfirstlane((int)(wave_global / waves_per_chunk));
    const uint32_t lg = __builtin_amdgcn_readfirstlane((int)(wave_global % waves_per_chunk));
    const int lane = threadIdx.x & 63;
    const uint32_t level = lg * LV + lane / F;
    const int f = lane % F;
    const int64_t s0 = chunk * CHUNK;
    if (s0 >= n) return;
    const int64_t s1 = s0 + CHUNK < n ? s0 + CHUNK : n;
    const bool active = level < L;
    const LevelInfo li = level_info(meta, active ? level : 0);
    const size_t ld = (size_t)lddy;

    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = 0.0f;
    uint32_t b0 = 0, b1 = 0, b2 = 0;
    bool have = false;

    auto flush = [&]() {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (acc[k] != 0.0f) {
                const uint32_t row = row_index(li, b0 + (k & 1), b1 + ((k >> 1) & 1), b2 + ((k >> 2) & 1));
                atomicAdd(dtable + (size_t)row * F + f, acc[k]);
            }
            acc[k] = 0.0f;
        }
    };

    for (int64_t sb = s0; sb < s1; sb += SUB) {
        float g[SUB], px[SUB], py[SUB], pz[SUB];
#pragma unroll
        for (int j = 0; j < SUB; j++) {
            const int64_t s = sb + j < s1 ? sb + j : s1 - 1;
            g[j] = active ? dL_dy[(size_t)s * ld + level * F + f] : 0.0f;
            px[j] = x[3 * s]; py[j] = x[3 * s + 1]; pz[j] = x[3 * s + 2];
        }
#pragma unroll
        for (int j = 0; j < SUB; j++) {
            if (sb + j >= s1) break;
            const float p0 = fmaf(li.scale, px[j], 0.5f), p1 = fmaf(li.scale, py[j], 0.5f), p2 = fmaf(li.scale, pz[j], 0.5f);
            const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
            const uint32_t g0 = (uint32_t)(int)f0, g1 = (uint32_t)(int)f1, g2 = (uint32_t)(int)f2;
            const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
            if (!(have && g0 == b0 && g1 == b1 && g2 == b2)) {
                if (have) flush();
                b0 = g0; b1 = g1; b2 = g2; have = true;
            }
            const float gv = g[j];
            const float x0 = (1 - w0) * gv, x1 = w0 * gv;
            const float y0 = 1 - w1, y1 = w1, z0 = 1 - w2, z1 = w2;
            acc[0] = fmaf(x0 * y0, z0, acc[0]); acc[1] = fmaf(x1 * y0, z0, acc[1]);
            acc[2] = fmaf(x0 * y1, z0, acc[2]); acc[3] = fmaf(x1 * y1, z0, acc[3]);
            acc[4] = fmaf(x0 * y0, z1, acc[4]); acc[5] = fmaf(x1 * y0, z1, acc[5]);
            acc[6] = fmaf(x0 * y1, z1, acc[6]); acc[7] = fmaf(x1 * y1, z1, acc[7]);
        }
    }
    if (have) flush();
}

// Variant of the run-merging scatter in which 16 lanes serve one level: lane = (level, x-corner,
// feature).  The two x-neighbours of a corner pair are adjacent table rows whenever the base
// index is even (dense levels: idx = x + ...; hashed levels: x enters the hash as x ^ ..., so
// x|1 flips only bit 0), and the atomic unit works on 64-byte blocks: issuing both rows from
// the same wave-instruction lets one memory-side request carry 64 B instead of 32 B.
template <int F, int CHUNK>
__global__ void __launch_bounds__(256) grid_bwd_param_merge2_kernel(GridMeta meta, const float* __restrict__ x,
                                                                    const float* __restrict__ dL_dy, int64_t lddy,
                                                                    int64_t n, float* __restrict__ dtable)
{
    constexpr int LV = 64 / (2 * F);  // levels per wave
    constexpr int SUB = 8;
    const uint32_t L = meta.n_levels;
    const uint32_t waves_per_chunk = (L + LV - 1) / LV;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t chunk = __builtin_amdgcn_readfirstlane((int)(wave_global / waves_per_chunk));
    const uint32_t lg = __builtin_amdgcn_readfirstlane((int)(wave_global % waves_per_chunk));
    const int lane = threadIdx.x & 63;
    const uint32_t level = lg * LV + lane / (2 * F);
    const uint32_t xb = (lane / F) & 1;
    const int f = lane % F;
    const int64_t s0 = chunk * CHUNK;
    if (s0 >= n) return;
    const int64_t s1 = s0 + CHUNK < n ? s0 + CHUNK : n;
    const bool active = level < L;
    const LevelInfo li = level_info(meta, active ? level : 0);
    const size_t ld = (size_t)lddy;

    float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    uint32_t b0 = 0, b1 = 0, b2 = 0;
    bool have = false;

    auto flush = [&]() {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (acc[k] != 0.0f) {
                const uint32_t row = row_index(li, b0 + xb, b1 + (k & 1), b2 + ((k >> 1) & 1));
                atomicAdd(dtable + (size_t)row * F + f, acc[k]);
            }
            acc[k] = 0.0f;
        }
    };

    for (int64_t sb = s0; sb < s1; sb += SUB) {
        float g[SUB], px[SUB], py[SUB], pz[SUB];
#pragma unroll
        for (int j = 0; j < SUB; j++) {
            const int64_t s = sb + j < s1 ? sb + j : s1 - 1;
            g[j] = active ? dL_dy[(size_t)s * ld + level * F + f] : 0.0f;
            px[j] = x[3 * s]; py[j] = x[3 * s + 1]; pz[j] = x[3 * s + 2];
        }
#pragma unroll
        for (int j = 0; j < SUB; j++) {
            if (sb + j >= s1) break;
            const float p0 = fmaf(li.scale, px[j], 0.5f), p1 = fmaf(li.scale, py[j], 0.5f), p2 = fmaf(li.scale, pz[j], 0.5f);
            const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
            const uint32_t g0 = (uint32_t)(int)f0, g1 = (uint32_t)(int)f1, g2 = (uint32_t)(int)f2;
            const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
            if (!(have && g0 == b0 && g1 == b1 && g2 == b2)) {
                if (have) flush();
                b0 = g0; b1 = g1; b2 = g2; have = true;
            }
            const float xw = (xb ? w0 : 1 - w0) * g[j];
            const float y0 = 1 - w1, y1 = w1, z0 = 1 - w2, z1 = w2;
            acc[0] = fmaf(xw * y0, z0, acc[0]); acc[1] = fmaf(xw * y1, z0, acc[1]);
            acc[2] = fmaf(xw * y0, z1, acc[2]); acc[3] = fmaf(xw * y1, z1, acc[3]);
        }
    }
    if (have) flush();
}

#endif  // NGP_AB_VARIANTS

// Run merging with corner-level carry-over.  A plain run-merging kernel (keep the 8 corner sums in registers while
// the sample stays in its cell) flushes all corners whenever the CELL changes; but a sample that moves to a neighbouring cell keeps half (face move), a
// quarter (edge move) or one (diagonal move) of its eight corners.  Here every lane keeps its four
// (y,z) accumulators addressed relative to a sliding 2x2x2 window: on a move of at most one cell
// per axis only the corners that LEAVE the window are flushed, the others slide to their new slot
// (y/z moves: inside the lane; x moves: between the two x-corner lanes of the level, lane ^ F).
// Each visited corner is then flushed once per visit — the fewest atomics run merging can issue.
template <int F, int CHUNK>
__global__ void __launch_bounds__(256) grid_bwd_param_slide_kernel(GridMeta meta, const float* __restrict__ x,
                                                                   const float* __restrict__ dL_dy, int64_t lddy,
                                                                   const float* __restrict__ row_scale,
                                                                   int64_t n, float* __restrict__ dtable)
{
    constexpr int LV = 64 / (2 * F);  // levels per wave
    constexpr int SUB = 8;
    const uint32_t L = meta.n_levels;
    const uint32_t waves_per_chunk = (L + LV - 1) / LV;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t chunk = __builtin_amdgcn_readfirstlane((int)(wave_global / waves_per_chunk));
    const uint32_t lg = __builtin_amdgcn_readfirstlane((int)(wave_global % waves_per_chunk));
    const int lane = threadIdx.x & 63;
    const uint32_t level = lg * LV + lane / (2 * F);
    const uint32_t xb = (lane / F) & 1;
    const int f = lane % F;
    const int64_t s0 = chunk * CHUNK;
    if (s0 >= n) return;
    const int64_t s1 = s0 + CHUNK < n ? s0 + CHUNK : n;
    const bool active = level < L;
    const LevelInfo li = level_info(meta, active ? level : 0);
    const size_t ld = (size_t)lddy;

    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;   // slots (cy,cz) = (0,0) (1,0) (0,1) (1,1)
    int b0 = 0, b1 = 0, b2 = 0;
    bool have = false;

    auto put = [&](float& a, int gx, int gy, int gz) {
        if (a != 0.0f) {
            const uint32_t row = row_index(li, (uint32_t)gx, (uint32_t)gy, (uint32_t)gz);
            atomicAdd(dtable + (size_t)row * F + f, a);
        }
        a = 0.0f;
    };

    for (int64_t sb = s0; sb < s1; sb += SUB) {
        float g[SUB], px[SUB], py[SUB], pz[SUB];
#pragma unroll
        for (int j = 0; j < SUB; j++) {
            const int64_t s = sb + j < s1 ? sb + j : s1 - 1;
            g[j] = active ? dL_dy[(size_t)s * ld + level * F + f] * (row_scale ? row_scale[s] : 1.0f) : 0.0f;
            px[j] = x[3 * s]; py[j] = x[3 * s + 1]; pz[j] = x[3 * s + 2];
        }
#pragma unroll
        for (int j = 0; j < SUB; j++) {
            if (sb + j >= s1) break;
            const float p0 = fmaf(li.scale, px[j], 0.5f), p1 = fmaf(li.scale, py[j], 0.5f), p2 = fmaf(li.scale, pz[j], 0.5f);
            const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
            const int g0 = (int)f0, g1 = (int)f1, g2 = (int)f2;
            const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
            const int dx = g0 - b0, dy = g1 - b1, dz = g2 - b2;
            if (have && (dx | dy | dz) != 0) {
                const int X = b0 + (int)xb;
                const bool near_move = dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
                if (!near_move) {
                    put(a0, X, b1, b2); put(a1, X, b1 + 1, b2); put(a2, X, b1, b2 + 1); put(a3, X, b1 + 1, b2 + 1);
                } else {
                    // corners leaving the window in y / z
                    const bool fy0 = dy == 1, fy1 = dy == -1, fz0 = dz == 1, fz1 = dz == -1;
                    if (fy0 || fz0) put(a0, X, b1, b2);
                    if (fy1 || fz0) put(a1, X, b1 + 1, b2);
                    if (fy0 || fz1) put(a2, X, b1, b2 + 1);
                    if (fy1 || fz1) put(a3, X, b1 + 1, b2 + 1);
                    // slide the survivors (flushed slots are zero)
                    if (dy == 1) { a0 = a1; a1 = 0.0f; a2 = a3; a3 = 0.0f; }
                    else if (dy == -1) { a1 = a0; a0 = 0.0f; a3 = a2; a2 = 0.0f; }
                    if (dz == 1) { a0 = a2; a2 = 0.0f; a1 = a3; a3 = 0.0f; }
                    else if (dz == -1) { a2 = a0; a0 = 0.0f; a3 = a1; a1 = 0.0f; }
                    if (dx != 0) {
                        // the x-corner that stays in the window changes lanes: the lane whose old X
                        // leaves flushes (at the NEW y/z base), then takes over its partner's sums
                        const bool leaving = (dx == 1) ? (xb == 0) : (xb == 1);
                        if (leaving) {
                            put(a0, X, g1, g2); put(a1, X, g1 + 1, g2); put(a2, X, g1, g2 + 1); put(a3, X, g1 + 1, g2 + 1);
                        }
                        const float q0 = __shfl_xor(a0, F, 64), q1 = __shfl_xor(a1, F, 64);
                        const float q2 = __shfl_xor(a2, F, 64), q3 = __shfl_xor(a3, F, 64);
                        if (leaving) { a0 = q0; a1 = q1; a2 = q2; a3 = q3; }
                        else { a0 = 0.0f; a1 = 0.0f; a2 = 0.0f; a3 = 0.0f; }
                    }
                }
            }
            b0 = g0; b1 = g1; b2 = g2; have = true;
            const float xw = (xb ? w0 : 1 - w0) * g[j];
            const float y0 = 1 - w1, y1 = w1, z0 = 1 - w2, z1 = w2;
            a0 = fmaf(xw * y0, z0, a0); a1 = fmaf(xw * y1, z0, a1);
            a2 = fmaf(xw * y0, z1, a2); a3 = fmaf(xw * y1, z1, a3);
        }
    }
    if (have) {
        const int X = b0 + (int)xb;
        put(a0, X, b1, b2); put(a1, X, b1 + 1, b2); put(a2, X, b1, b2 + 1); put(a3, X, b1 + 1, b2 + 1);
    }
}

// Superseded by grid_bwd_param_tag_kernel below (round 2): compiled only into the A/B build.
#ifdef NGP_AB_VARIANTS
// Line-aligned sliding window (F = 8: a table row is 32 bytes, a 64-byte atomic request covers the two
// rows (2k, 2k+1) of a level).  Memory-side float atomics are limited by REQUESTS, one per 64-byte line a
// wave-instruction touches (tools/atomic_shapes.hip: 20 G requests/s whether a request carries 4 or 64
// bytes), so the unit of accumulation here is the LINE, not the row: per (y,z) corner slot a lane pair
// keeps the sums of line A (the line holding row x0) and line B (the next line, used when x0 is the odd
// row of its pair) and a line is flushed ONCE, whole, when the sample's 2x2x2 window has left it —
// an x-step inside a line flushes nothing, where the row-window kernel above sends out a lone 32-byte
// row.  Which x start a line (even or odd) is a property of the (y,z) row: even for hashed levels
// (x enters the hash by xor, so x ^ 1 is the neighbour), parity of res * (y + z) for dense levels.
// No cross-lane traffic: lane = (level, row-in-line, feature) owns its sums from first add to flush.
// tools/scatter_model.py: 26.0 -> 22.7 requests per sample on a captured training batch.
template <int CHUNK, int MODE = 0>   // MODE bit 0 (A/B build): no atomics, the adds go to a per-lane sink (the kernel's ALU time)
__global__ void __launch_bounds__(256) grid_bwd_param_line_kernel(GridMeta meta, const float* __restrict__ x,
                                                                  const float* __restrict__ dL_dy, int64_t lddy,
                                                                  const float* __restrict__ row_scale,
                                                                  int64_t n, float* __restrict__ dtable)
{
    constexpr int F = 8;
    constexpr int LV = 4;             // levels per wave: 16 lanes each
    constexpr int SUB = 8;
    const uint32_t L = meta.n_levels;
    const uint32_t waves_per_chunk = (L + LV - 1) / LV;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t chunk = __builtin_amdgcn_readfirstlane((int)(wave_global / waves_per_chunk));
    const uint32_t lg = __builtin_amdgcn_readfirstlane((int)(wave_global % waves_per_chunk));
    const int lane = threadIdx.x & 63;
    const uint32_t level = lg * LV + lane / (2 * F);
    const int xb = (lane / F) & 1;    // this lane's row inside a line
    const int f = lane % F;
    const int64_t s0 = chunk * CHUNK;
    if (s0 >= n) return;
    const int64_t s1 = s0 + CHUNK < n ? s0 + CHUNK : n;
    const bool active = level < L;
    const LevelInfo li = level_info(meta, active ? level : 0);
    const size_t ld = (size_t)lddy;
    // dense level of odd resolution: the line pairing alternates with y + z
    const int odd = (!(li.flags & 1u) && (li.res & 1u)) ? 1 : 0;

    // slots (cy,cz) = (0,0) (1,0) (0,1) (1,1); A = line holding x = base, B = the line after it
    float A0 = 0.0f, A1 = 0.0f, A2 = 0.0f, A3 = 0.0f, B0 = 0.0f, B1 = 0.0f, B2 = 0.0f, B3 = 0.0f;
    int b0 = 0, b1 = 0, b2 = 0;
    bool have = false;
    float sink = 0.0f;   // MODE 1 only

    // (macros, not lambdas taking references: the sums must stay in registers)
#define NGP_PUT(a_, gx_, gy_, gz_)                                                                      \
    do {                                                                                                \
        if ((a_) != 0.0f) {                                                                             \
            const uint32_t row_ = row_index(li, (uint32_t)(gx_), (uint32_t)(gy_), (uint32_t)(gz_));     \
            if (MODE & 1) sink += (a_) * (float)(row_ & 1023u);                                         \
            else atomicAdd(dtable + (size_t)row_ * F + f, (a_));                                        \
        }                                                                                               \
        (a_) = 0.0f;                                                                                    \
    } while (0)
    // first x of the line that holds x0 in the (y,z) row: x0 - ((x0 - parity) & 1)
#define NGP_LINE_X(x0_, y_, z_) ((x0_) - (((x0_) - (odd & ((y_) + (z_)))) & 1))
#define NGP_FLUSH_SLOT(a_, b_, x0_, y_, z_)                        \
    do {                                                           \
        const int xa_ = NGP_LINE_X(x0_, y_, z_) + xb;              \
        NGP_PUT(a_, xa_, y_, z_); NGP_PUT(b_, xa_ + 2, y_, z_);    \
    } while (0)
    // the window's x base moved from xo to xn (|xn - xo| <= 1) in the (y,z) row of one slot
#define NGP_STEP_X(a_, b_, xo_, xn_, y_, z_)                                             \
    do {                                                                                 \
        const int lo_ = NGP_LINE_X(xo_, y_, z_), ln_ = NGP_LINE_X(xn_, y_, z_);          \
        if (ln_ > lo_) { NGP_PUT(a_, lo_ + xb, y_, z_); (a_) = (b_); (b_) = 0.0f; }      \
        else if (ln_ < lo_) { NGP_PUT(b_, lo_ + 2 + xb, y_, z_); (b_) = (a_); (a_) = 0.0f; } \
    } while (0)

    for (int64_t sb = s0; sb < s1; sb += SUB) {
        float g[SUB], px[SUB], py[SUB], pz[SUB];
#pragma unroll
        for (int j = 0; j < SUB; j++) {
            const int64_t s = sb + j < s1 ? sb + j : s1 - 1;
            g[j] = active ? dL_dy[(size_t)s * ld + level * F + f] * (row_scale ? row_scale[s] : 1.0f) : 0.0f;
            px[j] = x[3 * s]; py[j] = x[3 * s + 1]; pz[j] = x[3 * s + 2];
        }
#pragma unroll
        for (int j = 0; j < SUB; j++) {
            if (sb + j >= s1) break;
            const float p0 = fmaf(li.scale, px[j], 0.5f), p1 = fmaf(li.scale, py[j], 0.5f), p2 = fmaf(li.scale, pz[j], 0.5f);
            const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
            const int g0 = (int)f0, g1 = (int)f1, g2 = (int)f2;
            const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
            const int dx = g0 - b0, dy = g1 - b1, dz = g2 - b2;
            if (have && (dx | dy | dz) != 0) {
                const bool near_move = dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
                if (!near_move) {
                    NGP_FLUSH_SLOT(A0, B0, b0, b1, b2); NGP_FLUSH_SLOT(A1, B1, b0, b1 + 1, b2);
                    NGP_FLUSH_SLOT(A2, B2, b0, b1, b2 + 1); NGP_FLUSH_SLOT(A3, B3, b0, b1 + 1, b2 + 1);
                } else {
                    // (y,z) rows leaving the window go out whole; the others slide to their new slot
                    const bool fy0 = dy == 1, fy1 = dy == -1, fz0 = dz == 1, fz1 = dz == -1;
                    if (fy0 || fz0) NGP_FLUSH_SLOT(A0, B0, b0, b1, b2);
                    if (fy1 || fz0) NGP_FLUSH_SLOT(A1, B1, b0, b1 + 1, b2);
                    if (fy0 || fz1) NGP_FLUSH_SLOT(A2, B2, b0, b1, b2 + 1);
                    if (fy1 || fz1) NGP_FLUSH_SLOT(A3, B3, b0, b1 + 1, b2 + 1);
                    if (dy == 1) { A0 = A1; B0 = B1; A1 = 0.0f; B1 = 0.0f; A2 = A3; B2 = B3; A3 = 0.0f; B3 = 0.0f; }
                    else if (dy == -1) { A1 = A0; B1 = B0; A0 = 0.0f; B0 = 0.0f; A3 = A2; B3 = B2; A2 = 0.0f; B2 = 0.0f; }
                    if (dz == 1) { A0 = A2; B0 = B2; A2 = 0.0f; B2 = 0.0f; A1 = A3; B1 = B3; A3 = 0.0f; B3 = 0.0f; }
                    else if (dz == -1) { A2 = A0; B2 = B0; A0 = 0.0f; B0 = 0.0f; A3 = A1; B3 = B1; A1 = 0.0f; B1 = 0.0f; }
                    if (dx != 0) {
                        NGP_STEP_X(A0, B0, b0, g0, g1, g2); NGP_STEP_X(A1, B1, b0, g0, g1 + 1, g2);
                        NGP_STEP_X(A2, B2, b0, g0, g1, g2 + 1); NGP_STEP_X(A3, B3, b0, g0, g1 + 1, g2 + 1);
                    }
                }
            }
            b0 = g0; b1 = g1; b2 = g2; have = true;
            // row x0 is row q of its line (q = 0: both corners in line A; q = 1: x0 closes line A, x0+1 opens line B)
            const float gv = g[j];
            const float xw0 = (1 - w0) * gv, xw1 = w0 * gv;
            const float y0 = 1 - w1, y1 = w1, z0 = 1 - w2, z1 = w2;
#define NGP_LINE_ACC(A_, B_, Y_, Z_, WY_, WZ_)                                    \
            {                                                                     \
                const int q = (g0 - (odd & ((Y_) + (Z_)))) & 1;                   \
                const float va = q ? (xb ? xw0 : 0.0f) : (xb ? xw1 : xw0);        \
                const float vb = (q && !xb) ? xw1 : 0.0f;                         \
                A_ = fmaf(va * (WY_), (WZ_), A_);                                 \
                B_ = fmaf(vb * (WY_), (WZ_), B_);                                 \
            }
            NGP_LINE_ACC(A0, B0, g1, g2, y0, z0)
            NGP_LINE_ACC(A1, B1, g1 + 1, g2, y1, z0)
            NGP_LINE_ACC(A2, B2, g1, g2 + 1, y0, z1)
            NGP_LINE_ACC(A3, B3, g1 + 1, g2 + 1, y1, z1)
#undef NGP_LINE_ACC
        }
    }
    if (have) {
        NGP_FLUSH_SLOT(A0, B0, b0, b1, b2); NGP_FLUSH_SLOT(A1, B1, b0, b1 + 1, b2);
        NGP_FLUSH_SLOT(A2, B2, b0, b1, b2 + 1); NGP_FLUSH_SLOT(A3, B3, b0, b1 + 1, b2 + 1);
    }
    if ((MODE & 1) && sink == 123.456f) dtable[lane] = sink;
}

// Experiment: the same walk with the per-sample body NOT unrolled (the product kernel above unrolls 8 samples
// with every flush path: ~70 KB of code against a 64 KB instruction cache shared by two CUs).  The gradient
// values run PF samples ahead in a register ring (rotation, no dynamic register index).
// MODE bit 0: no atomics (the flush arithmetic stays, the adds go to a per-lane sink) — the kernel's ALU time.
template <int CHUNK, int MODE>
__global__ void __launch_bounds__(256) grid_bwd_param_line_rolled_kernel(GridMeta meta, const float* __restrict__ x,
                                                                         const float* __restrict__ dL_dy, int64_t lddy,
                                                                         const float* __restrict__ row_scale,
                                                                         int64_t n, float* __restrict__ dtable)
{
    constexpr int F = 8;
    constexpr int LV = 4;
    const uint32_t L = meta.n_levels;
    const uint32_t waves_per_chunk = (L + LV - 1) / LV;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t chunk = __builtin_amdgcn_readfirstlane((int)(wave_global / waves_per_chunk));
    const uint32_t lg = __builtin_amdgcn_readfirstlane((int)(wave_global % waves_per_chunk));
    const int lane = threadIdx.x & 63;
    const uint32_t level = lg * LV + lane / (2 * F);
    const int xb = (lane / F) & 1;
    const int f = lane % F;
    const int64_t s0 = chunk * CHUNK;
    if (s0 >= n) return;
    const int64_t s1 = s0 + CHUNK < n ? s0 + CHUNK : n;
    const bool active = level < L;
    const LevelInfo li = level_info(meta, active ? level : 0);
    const size_t ld = (size_t)lddy;
    const int odd = (!(li.flags & 1u) && (li.res & 1u)) ? 1 : 0;
    float sink = 0.0f;

    float A0 = 0.0f, A1 = 0.0f, A2 = 0.0f, A3 = 0.0f, B0 = 0.0f, B1 = 0.0f, B2 = 0.0f, B3 = 0.0f;
    int b0 = 0, b1 = 0, b2 = 0;
    bool have = false;

    const float* gp = dL_dy + (size_t)(active ? level : 0) * F + f;
    const int64_t last = s1 - 1;
#define NGP_LOADG(s_) gp[(size_t)((s_) < last ? (s_) : last) * ld]
#define NGP_LOADS(s_) (row_scale ? row_scale[(s_) < last ? (s_) : last] : 1.0f)
    float gr0 = NGP_LOADG(s0), gr1 = NGP_LOADG(s0 + 1), gr2 = NGP_LOADG(s0 + 2), gr3 = NGP_LOADG(s0 + 3);
    float sc0 = NGP_LOADS(s0), sc1 = NGP_LOADS(s0 + 1), sc2 = NGP_LOADS(s0 + 2), sc3 = NGP_LOADS(s0 + 3);
    float qx = x[3 * s0], qy = x[3 * s0 + 1], qz = x[3 * s0 + 2];
#pragma unroll 1
    for (int64_t s = s0; s < s1; s++) {
        const float gv = active ? gr0 * sc0 : 0.0f;
        gr0 = gr1; gr1 = gr2; gr2 = gr3; gr3 = NGP_LOADG(s + 4);
        sc0 = sc1; sc1 = sc2; sc2 = sc3; sc3 = NGP_LOADS(s + 4);
        const float pxj = qx, pyj = qy, pzj = qz;
        {
            const int64_t sn = s + 1 < last ? s + 1 : last;
            qx = x[3 * sn]; qy = x[3 * sn + 1]; qz = x[3 * sn + 2];
        }
        const float p0 = fmaf(li.scale, pxj, 0.5f), p1 = fmaf(li.scale, pyj, 0.5f), p2 = fmaf(li.scale, pzj, 0.5f);
        const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
        const int g0 = (int)f0, g1 = (int)f1, g2 = (int)f2;
        const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
        const int dx = g0 - b0, dy = g1 - b1, dz = g2 - b2;
        if (have && (dx | dy | dz) != 0) {
            const bool near_move = dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
            if (!near_move) {
                NGP_FLUSH_SLOT(A0, B0, b0, b1, b2); NGP_FLUSH_SLOT(A1, B1, b0, b1 + 1, b2);
                NGP_FLUSH_SLOT(A2, B2, b0, b1, b2 + 1); NGP_FLUSH_SLOT(A3, B3, b0, b1 + 1, b2 + 1);
            } else {
                const bool fy0 = dy == 1, fy1 = dy == -1, fz0 = dz == 1, fz1 = dz == -1;
                if (fy0 || fz0) NGP_FLUSH_SLOT(A0, B0, b0, b1, b2);
                if (fy1 || fz0) NGP_FLUSH_SLOT(A1, B1, b0, b1 + 1, b2);
                if (fy0 || fz1) NGP_FLUSH_SLOT(A2, B2, b0, b1, b2 + 1);
                if (fy1 || fz1) NGP_FLUSH_SLOT(A3, B3, b0, b1 + 1, b2 + 1);
                if (dy == 1) { A0 = A1; B0 = B1; A1 = 0.0f; B1 = 0.0f; A2 = A3; B2 = B3; A3 = 0.0f; B3 = 0.0f; }
                else if (dy == -1) { A1 = A0; B1 = B0; A0 = 0.0f; B0 = 0.0f; A3 = A2; B3 = B2; A2 = 0.0f; B2 = 0.0f; }
                if (dz == 1) { A0 = A2; B0 = B2; A2 = 0.0f; B2 = 0.0f; A1 = A3; B1 = B3; A3 = 0.0f; B3 = 0.0f; }
                else if (dz == -1) { A2 = A0; B2 = B0; A0 = 0.0f; B0 = 0.0f; A3 = A1; B3 = B1; A1 = 0.0f; B1 = 0.0f; }
                if (dx != 0) {
                    NGP_STEP_X(A0, B0, b0, g0, g1, g2); NGP_STEP_X(A1, B1, b0, g0, g1 + 1, g2);
                    NGP_STEP_X(A2, B2, b0, g0, g1, g2 + 1); NGP_STEP_X(A3, B3, b0, g0, g1 + 1, g2 + 1);
                }
            }
        }
        b0 = g0; b1 = g1; b2 = g2; have = true;
        const float xw0 = (1 - w0) * gv, xw1 = w0 * gv;
        const float y0 = 1 - w1, y1 = w1, z0 = 1 - w2, z1 = w2;
#define NGP_LINE_ACC(A_, B_, Y_, Z_, WY_, WZ_)                                    \
        {                                                                     \
            const int q = (g0 - (odd & ((Y_) + (Z_)))) & 1;                   \
            const float va = q ? (xb ? xw0 : 0.0f) : (xb ? xw1 : xw0);        \
            const float vb = (q && !xb) ? xw1 : 0.0f;                         \
            A_ = fmaf(va * (WY_), (WZ_), A_);                                 \
            B_ = fmaf(vb * (WY_), (WZ_), B_);                                 \
        }
        NGP_LINE_ACC(A0, B0, g1, g2, y0, z0)
        NGP_LINE_ACC(A1, B1, g1 + 1, g2, y1, z0)
        NGP_LINE_ACC(A2, B2, g1, g2 + 1, y0, z1)
        NGP_LINE_ACC(A3, B3, g1 + 1, g2 + 1, y1, z1)
#undef NGP_LINE_ACC
    }
#undef NGP_LOADG
#undef NGP_LOADS
    if (have) {
        NGP_FLUSH_SLOT(A0, B0, b0, b1, b2); NGP_FLUSH_SLOT(A1, B1, b0, b1 + 1, b2);
        NGP_FLUSH_SLOT(A2, B2, b0, b1, b2 + 1); NGP_FLUSH_SLOT(A3, B3, b0, b1 + 1, b2 + 1);
    }
    if ((MODE & 1) && sink == 123.456f) dtable[lane] = sink;
}
#undef NGP_PUT
#undef NGP_LINE_X
#undef NGP_FLUSH_SLOT
#undef NGP_STEP_X
#endif  // NGP_AB_VARIANTS

// Two-phase line scatter (F = 8).  The sliding-window kernel above spends its time in VALU work, not in atomic
// requests (measured: 0.58 ms with the atomics compiled out against 0.70 ms with them, 237 VALU instructions per
// wave and sample): the 16 lanes (row-in-line, feature) of a level all redo the same floor / delta / hash / window
// arithmetic.  Here that arithmetic is done ONCE per (sample, level), in parallel:
//   phase 1  lane = (level, sample-in-round): the cell of the sample on its level, the rows of its 8 corners, and per
//            (y,z) corner slot the 64-byte LINES the two x-corners fall into (line id = row >> 1: rows 2k, 2k+1 share a
//            request) with the trilinear weight each row-in-line takes — written to a wave-private LDS record
//            together with the round's 16 x 32 gradient values;
//   phase 2  lane = (level, row-in-line, feature), serial over the round's samples: per slot two running sums A / B
//            tagged with their line ids.  A sum whose line is one of the new sample's two lines of the slot is kept
//            (moved between A and B if the window stepped across a line), any other non-zero sum goes out as ONE
//            atomic wave-instruction per line.  Slots are addressed by the ABSOLUTE parity of (y, z), so a window
//            step in y or z moves no data: the rows that stay keep their slot, the ones that leave fail the tag
//            compare.  No geometry cases (near / far moves, line parity of odd dense resolutions, hash or not): two
//            corners that land in the same line are merged because they ARE the same line.
// ~80 VALU instructions per wave and sample instead of 237; what remains is the memory-side request rate.
// MODE bit 0 (A/B build): no atomics, the adds go to a per-lane sink.
template <int CHUNK, int MODE = 0>
__global__ void __launch_bounds__(256) grid_bwd_param_tag_kernel(GridMeta meta, const float* __restrict__ x,
                                                                 const float* __restrict__ dL_dy, int64_t lddy,
                                                                 const float* __restrict__ row_scale,
                                                                 int64_t n, float* __restrict__ dtable)
{
    constexpr int F = 8;
    constexpr int LV = 4;             // levels per wave
    constexpr int R = 16;             // samples per round: 64 lanes = 16 samples x 4 levels in phase 1
    constexpr int REC = 132;          // dwords per sample record (128 + 4: phase-1 stores of 8 lanes hit 8 x 4 banks)
    constexpr uint32_t INVALID = 0xFFFFFFFFu;
    // record: [0,16) tagA[lv][slot]  [16,32) tagB[lv][slot]  [32,64) wA[lv][xb][slot]  [64,96) wB[lv][xb][slot]
    //         [96,128) g[lv][f]
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][R * REC];
    uint32_t* W = lds[threadIdx.x >> 6];
    const uint32_t L = meta.n_levels;
    const uint32_t waves_per_chunk = (L + LV - 1) / LV;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t chunk = __builtin_amdgcn_readfirstlane((int)(wave_global / waves_per_chunk));
    const uint32_t lg = __builtin_amdgcn_readfirstlane((int)(wave_global % waves_per_chunk));
    const int lane = threadIdx.x & 63;
    const int64_t s0 = chunk * CHUNK;
    if (s0 >= n) return;
    const int64_t s1 = s0 + CHUNK < n ? s0 + CHUNK : n;
    const size_t ld = (size_t)lddy;

    // phase-1 identity
    const int lv1 = lane >> 4, j1 = lane & 15;
    const uint32_t level1 = lg * LV + lv1;
    const LevelInfo li = level_info(meta, level1 < L ? level1 : 0);
    // gradient staging identity: lane -> (sample lane >> 2, 8 floats (lane & 3) * 8 .. + 7 of the wave's 32)
    const int gs = lane >> 2, gq = lane & 3;
    const bool g_ok = (lg * LV * F + gq * 8 + 8) <= L * F;   // the 8 floats belong to existing levels
    // phase-2 identity
    const int lv2 = lane >> 4, xb = (lane >> 3) & 1, f = lane & 7;
    const uint32_t lane_off = (uint32_t)(xb * F + f);

    float A[4] = {0.0f, 0.0f, 0.0f, 0.0f}, B[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t tA[4] = {INVALID, INVALID, INVALID, INVALID}, tB[4] = {INVALID, INVALID, INVALID, INVALID};
    float sink = 0.0f;

#define NGP_TAG_PUT(a_, t_)                                                            \
    do {                                                                               \
        if (MODE & 1) sink += (a_) * (float)((t_) & 1023u);                            \
        else atomicAdd(dtable + (((t_) << 4) + lane_off), (a_));                       \
    } while (0)

    // prefetch of the first round
    auto clampS = [&](int64_t s) { return s < s1 ? s : s1 - 1; };
    float px, py, pz;
    float4 ga, gb;
    float gsc;
    {
        const int64_t s = clampS(s0 + j1);
        px = x[3 * s]; py = x[3 * s + 1]; pz = x[3 * s + 2];
        const int64_t sg = clampS(s0 + gs);
        const float* gp = dL_dy + (size_t)sg * ld + lg * LV * F + gq * 8;
        ga = g_ok ? *reinterpret_cast<const float4*>(gp) : make_float4(0, 0, 0, 0);
        gb = g_ok ? *reinterpret_cast<const float4*>(gp + 4) : make_float4(0, 0, 0, 0);
        gsc = row_scale ? row_scale[sg] : 1.0f;
    }
    for (int64_t sb = s0; sb < s1; sb += R) {
        // ---------------- phase 1: one (sample, level) per lane
        {
            const float p0 = fmaf(li.scale, px, 0.5f), p1 = fmaf(li.scale, py, 0.5f), p2 = fmaf(li.scale, pz, 0.5f);
            const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
            const int g0 = (int)f0, g1 = (int)f1, g2 = (int)f2;
            const float w0 = p0 - f0, w1 = p1 - f1, w2 = p2 - f2;
            uint32_t ta[4], tb[4];
            float wa0[4], wa1[4], wb0[4], wb1[4];
#pragma unroll
            for (int slot = 0; slot < 4; slot++) {
                const int cy = ((slot & 1) ^ g1) & 1, cz = ((slot >> 1) ^ g2) & 1;   // slot = absolute parity of (y, z)
                const uint32_t r0 = row_index(li, (uint32_t)g0, (uint32_t)(g1 + cy), (uint32_t)(g2 + cz));
                const uint32_t r1 = row_index(li, (uint32_t)(g0 + 1), (uint32_t)(g1 + cy), (uint32_t)(g2 + cz));
                const float wyz = (cy ? w1 : 1 - w1) * (cz ? w2 : 1 - w2);
                const float a0 = (1 - w0) * wyz, a1 = w0 * wyz;
                const uint32_t l0 = r0 >> 1, l1 = r1 >> 1;
                const bool same = l0 == l1;
                const uint32_t b0 = r0 & 1u, b1 = r1 & 1u;
                ta[slot] = l0;
                tb[slot] = same ? INVALID : l1;
                // weight of row-in-line 0 / 1 of line A, of line B
                wa0[slot] = (b0 == 0 ? a0 : 0.0f) + ((same && b1 == 0) ? a1 : 0.0f);
                wa1[slot] = (b0 == 1 ? a0 : 0.0f) + ((same && b1 == 1) ? a1 : 0.0f);
                wb0[slot] = (!same && b1 == 0) ? a1 : 0.0f;
                wb1[slot] = (!same && b1 == 1) ? a1 : 0.0f;
            }
            uint32_t* rec = W + j1 * REC;
            *reinterpret_cast<uint4*>(rec + lv1 * 4) = make_uint4(ta[0], ta[1], ta[2], ta[3]);
            *reinterpret_cast<uint4*>(rec + 16 + lv1 * 4) = make_uint4(tb[0], tb[1], tb[2], tb[3]);
            *reinterpret_cast<float4*>(rec + 32 + lv1 * 8) = make_float4(wa0[0], wa0[1], wa0[2], wa0[3]);
            *reinterpret_cast<float4*>(rec + 32 + lv1 * 8 + 4) = make_float4(wa1[0], wa1[1], wa1[2], wa1[3]);
            *reinterpret_cast<float4*>(rec + 64 + lv1 * 8) = make_float4(wb0[0], wb0[1], wb0[2], wb0[3]);
            *reinterpret_cast<float4*>(rec + 64 + lv1 * 8 + 4) = make_float4(wb1[0], wb1[1], wb1[2], wb1[3]);
            float* grec = reinterpret_cast<float*>(W + gs * REC + 96 + gq * 8);
            *reinterpret_cast<float4*>(grec) = make_float4(ga.x * gsc, ga.y * gsc, ga.z * gsc, ga.w * gsc);
            *reinterpret_cast<float4*>(grec + 4) = make_float4(gb.x * gsc, gb.y * gsc, gb.z * gsc, gb.w * gsc);
        }
        // the next round's operands travel under phase 2
        if (sb + R < s1) {
            const int64_t s = clampS(sb + R + j1);
            px = x[3 * s]; py = x[3 * s + 1]; pz = x[3 * s + 2];
            const int64_t sg = clampS(sb + R + gs);
            const float* gp = dL_dy + (size_t)sg * ld + lg * LV * F + gq * 8;
            if (g_ok) { ga = *reinterpret_cast<const float4*>(gp); gb = *reinterpret_cast<const float4*>(gp + 4); }
            gsc = row_scale ? row_scale[sg] : 1.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase 2: serial over the round's samples
        const int jmax = (int)((s1 - sb) < R ? (s1 - sb) : R);
        const uint32_t* base = W + lv2 * 4;
#pragma unroll 2
        for (int j = 0; j < jmax; j++) {
            const uint32_t* rec = base + j * REC;
            const uint4 nA4 = *reinterpret_cast<const uint4*>(rec);
            const uint4 nB4 = *reinterpret_cast<const uint4*>(rec + 16);
            const float4 wA4 = *reinterpret_cast<const float4*>(rec + 32 + lv2 * 4 + xb * 4);
            const float4 wB4 = *reinterpret_cast<const float4*>(rec + 64 + lv2 * 4 + xb * 4);
            const float gv = *reinterpret_cast<const float*>(rec + 96 + lv2 * 4 + f);
            const uint32_t nA[4] = {nA4.x, nA4.y, nA4.z, nA4.w}, nB[4] = {nB4.x, nB4.y, nB4.z, nB4.w};
            const float wA[4] = {wA4.x, wA4.y, wA4.z, wA4.w}, wB[4] = {wB4.x, wB4.y, wB4.z, wB4.w};
#pragma unroll
            for (int slot = 0; slot < 4; slot++) {
                const bool mAA = tA[slot] == nA[slot], mBA = tB[slot] == nA[slot];
                const bool mAB = tA[slot] == nB[slot], mBB = tB[slot] == nB[slot];
                if (!(mAA || mAB) && A[slot] != 0.0f) NGP_TAG_PUT(A[slot], tA[slot]);
                if (!(mBA || mBB) && B[slot] != 0.0f) NGP_TAG_PUT(B[slot], tB[slot]);
                const float keepA = mAA ? A[slot] : (mBA ? B[slot] : 0.0f);
                const float keepB = mBB ? B[slot] : (mAB ? A[slot] : 0.0f);
                A[slot] = fmaf(wA[slot], gv, keepA);
                // a B sum without a line (both x-corners in line A: tag INVALID, weight 0) must stay exactly 0 — with a
                // non-finite gradient 0 * gv is NaN, NaN != 0 would flush it, and INVALID << 4 is not an address of the table
                B[slot] = nB[slot] == INVALID ? 0.0f : fmaf(wB[slot], gv, keepB);
                tA[slot] = nA[slot];
                tB[slot] = nB[slot];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int slot = 0; slot < 4; slot++) {
        if (A[slot] != 0.0f) NGP_TAG_PUT(A[slot], tA[slot]);
        if (B[slot] != 0.0f) NGP_TAG_PUT(B[slot], tB[slot]);
    }
#undef NGP_TAG_PUT
    if ((MODE & 1) && sink == 123.456f) dtable[lane] = sink;
}

// ------------------------------------------------------------------ input gradient (H3)
// GROUP = lanes that belong to one sample (L * LPI, a power of two <= 64): their partial
// (dx,dy,dz) are summed with xor-shuffles and lane 0 of the group stores the result.
template <int F, int GROUP>
__global__ void __launch_bounds__(256) grid_bwd_input_kernel(GridMeta meta, const float* __restrict__ table,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ dL_dy, int64_t lddy,
                                                             int64_t n_items, float* __restrict__ dL_dx)
{
    constexpr int V = F >= 4 ? 4 : F;
    constexpr int LPI = F / V;
    typedef typename VecT<V>::T vec_t;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t item = tid / LPI;
    const int sub = (int)(tid % LPI);
    const bool live = item < n_items;
    const uint32_t L = meta.n_levels;
    const int64_t sample = live ? item / L : 0;
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
    if (live) {
        const uint32_t level = (uint32_t)(item - sample * L);
        const LevelInfo li = level_info(meta, level);
        const Cell c = cell_of(x, sample, li.scale);
        float go[V];
        {
            const vec_t gv = *reinterpret_cast<const vec_t*>(dL_dy + sample * lddy + level * F + sub * V);
            const float* gp = reinterpret_cast<const float*>(&gv);
#pragma unroll
            for (int j = 0; j < V; j++) go[j] = gp[j];
        }
        vec_t vals[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t cx = k & 1, cy = (k >> 1) & 1, cz = (k >> 2) & 1;
            const uint32_t row = row_index(li, c.g[0] + cx, c.g[1] + cy, c.g[2] + cz);
            vals[k] = *reinterpret_cast<const vec_t*>(table + (size_t)row * F + sub * V);
        }
        const float wx0 = 1 - c.w[0], wx1 = c.w[0], wy0 = 1 - c.w[1], wy1 = c.w[1], wz0 = 1 - c.w[2], wz1 = c.w[2];
        // d/dx: pairs (k, k|1) over (y,z); d/dy: pairs (k, k|2) over (x,z); d/dz: pairs (k, k|4) over (x,y)
        gx = wy0 * wz0 * vec_dot<V>(go, vals[1], vals[0]) + wy1 * wz0 * vec_dot<V>(go, vals[3], vals[2]) +
             wy0 * wz1 * vec_dot<V>(go, vals[5], vals[4]) + wy1 * wz1 * vec_dot<V>(go, vals[7], vals[6]);
        gy = wx0 * wz0 * vec_dot<V>(go, vals[2], vals[0]) + wx1 * wz0 * vec_dot<V>(go, vals[3], vals[1]) +
             wx0 * wz1 * vec_dot<V>(go, vals[6], vals[4]) + wx1 * wz1 * vec_dot<V>(go, vals[7], vals[5]);
        gz = wx0 * wy0 * vec_dot<V>(go, vals[4], vals[0]) + wx1 * wy0 * vec_dot<V>(go, vals[5], vals[1]) +
             wx0 * wy1 * vec_dot<V>(go, vals[6], vals[2]) + wx1 * wy1 * vec_dot<V>(go, vals[7], vals[3]);
        gx *= li.scale; gy *= li.scale; gz *= li.scale;
    }
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) {
        gx += __shfl_xor(gx, o, GROUP);
        gy += __shfl_xor(gy, o, GROUP);
        gz += __shfl_xor(gz, o, GROUP);
    }
    if (live && (threadIdx.x & (GROUP - 1)) == 0) {
        dL_dx[3 * sample] = gx; dL_dx[3 * sample + 1] = gy; dL_dx[3 * sample + 2] = gz;
    }
}

// Fallback for level counts whose group is not a power of two: one lane per sample.
template <int F>
__global__ void grid_bwd_input_serial_kernel(GridMeta meta, const float* __restrict__ table,
                                             const float* __restrict__ x, const float* __restrict__ dL_dy,
                                             int64_t lddy, int64_t n, float* __restrict__ dL_dx)
{
    const int64_t sample = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sample >= n) return;
    float acc[3] = { 0, 0, 0 };
    for (uint32_t level = 0; level < meta.n_levels; level++) {
        const LevelInfo li = level_info(meta, level);
        const Cell c = cell_of(x, sample, li.scale);
        const float* go = dL_dy + sample * lddy + level * F;
        for (int gd = 0; gd < 3; gd++) {
            const int a = (gd + 1) % 3, b = (gd + 2) % 3;
            for (int cc = 0; cc < 4; cc++) {
                uint32_t q0[3], q1[3];
                const int ca = cc & 1, cb = (cc >> 1) & 1;
                const float wt = li.scale * (ca ? c.w[a] : 1 - c.w[a]) * (cb ? c.w[b] : 1 - c.w[b]);
                q0[a] = q1[a] = c.g[a] + ca; q0[b] = q1[b] = c.g[b] + cb;
                q0[gd] = c.g[gd]; q1[gd] = c.g[gd] + 1;
                const float* r0 = table + (size_t)row_index(li, q0[0], q0[1], q0[2]) * F;
                const float* r1 = table + (size_t)row_index(li, q1[0], q1[1], q1[2]) * F;
                float dot = 0;
                for (int f = 0; f < F; f++) dot += go[f] * (r1[f] - r0[f]);
                acc[gd] += wt * dot;
            }
        }
    }
    dL_dx[3 * sample] = acc[0]; dL_dx[3 * sample + 1] = acc[1]; dL_dx[3 * sample + 2] = acc[2];
}

// ------------------------------------------------------------------ double backward (H4)
// F lanes per item.  v = dLoss/d(dL_dx) (n,3).
//   dtable[row(c)] += sum_d v_d * scale * (+/- prod_{d'!=d} w~) * dL_dy      (atomic)
//   dL_ddLdy[f]     = sum_d v_d * scale * sum_c (+/- prod w~) * table[row(c)][f]
template <int F>
__global__ void __launch_bounds__(256) grid_bwd_bwd_input_kernel(GridMeta meta, const float* __restrict__ table,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ dL_dy, int64_t lddy,
                                                                 const float* __restrict__ v, int64_t n_items,
                                                                 float* __restrict__ dtable,
                                                                 float* __restrict__ dL_ddLdy)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t item = tid / F;
    const int f = (int)(tid % F);
    if (item >= n_items) return;
    const uint32_t L = meta.n_levels;
    const int64_t sample = item / L;
    const uint32_t level = (uint32_t)(item - sample * L);
    const LevelInfo li = level_info(meta, level);
    const Cell c = cell_of(x, sample, li.scale);
    const float vx = v[3 * sample] * li.scale, vy = v[3 * sample + 1] * li.scale, vz = v[3 * sample + 2] * li.scale;
    const float g = dL_dy[sample * lddy + level * F + f];
    float ddy = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t cx = k & 1, cy = (k >> 1) & 1, cz = (k >> 2) & 1;
        const float wx = cx ? c.w[0] : 1 - c.w[0], wy = cy ? c.w[1] : 1 - c.w[1], wz = cz ? c.w[2] : 1 - c.w[2];
        // derivative of the trilinear weight of corner k w.r.t. each coordinate
        const float coef = vx * (cx ? 1.0f : -1.0f) * wy * wz + vy * (cy ? 1.0f : -1.0f) * wx * wz +
                           vz * (cz ? 1.0f : -1.0f) * wx * wy;
        const uint32_t row = row_index(li, c.g[0] + cx, c.g[1] + cy, c.g[2] + cz);
        if (dL_ddLdy) ddy = fmaf(coef, table[(size_t)row * F + f], ddy);
        if (dtable && g != 0.0f && coef != 0.0f) atomicAdd(dtable + (size_t)row * F + f, coef * g);
    }
    if (dL_ddLdy) dL_ddLdy[item * F + f] = ddy;
}

// ------------------------------------------------------------------ spherical harmonics (H5)
__device__ __forceinline__ void sh_eval(float x, float y, float z, int degree, float* o)
{
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    if (degree <= 1) return;
    o[1] = -0.48860251190291987f * y; o[2] = 0.48860251190291987f * z; o[3] = -0.48860251190291987f * x;
    if (degree <= 2) return;
    o[4] = 1.0925484305920792f * xy; o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz; o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    if (degree <= 3) return;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

template <int DEG>
__global__ void sh_fwd_kernel(const float* __restrict__ xin, int64_t n, float* __restrict__ y, int64_t ldy)
{
    constexpr int D = DEG * DEG;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[16];
    sh_eval(xin[3 * i] * 2 - 1, xin[3 * i + 1] * 2 - 1, xin[3 * i + 2] * 2 - 1, DEG, o);
#pragma unroll
    for (int k = 0; k < D; k++) y[i * ldy + k] = o[k];
}

// SH of a raw direction: d -> (normalize(d, eps 1e-6) + 1) / 2 -> basis, i.e. networks.py:198,222
// (F.normalize, the [0,1] remap and the encoder) in one launch
template <int DEG>
__global__ void sh_fwd_dirs_kernel(const float* __restrict__ d, int64_t n, float* __restrict__ y, int64_t ldy)
{
    constexpr int D = DEG * DEG;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dx = d[3 * i], dy = d[3 * i + 1], dz = d[3 * i + 2];
    const float inv = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f);
    const float ux = (dx * inv + 1.0f) * 0.5f, uy = (dy * inv + 1.0f) * 0.5f, uz = (dz * inv + 1.0f) * 0.5f;
    float o[16];
    sh_eval(ux * 2 - 1, uy * 2 - 1, uz * 2 - 1, DEG, o);
#pragma unroll
    for (int k = 0; k < D; k++) y[i * ldy + k] = o[k];
}

// dL_dx of the SH basis (includes the factor 2 of the [0,1] -> [-1,1] remap)
template <int DEG>
__global__ void sh_bwd_kernel(const float* __restrict__ xin, const float* __restrict__ dL_dy, int64_t n,
                              float* __restrict__ dL_dx)
{
    constexpr int D = DEG * DEG;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xin[3 * i] * 2 - 1, y = xin[3 * i + 1] * 2 - 1, z = xin[3 * i + 2] * 2 - 1;
    float g[16];
#pragma unroll
    for (int k = 0; k < 16; k++) g[k] = k < D ? dL_dy[i * D + k] : 0.0f;
    float dx = 0, dy = 0, dz = 0;
    if (DEG >= 2) {
        dy += -0.48860251190291987f * g[1]; dz += 0.48860251190291987f * g[2]; dx += -0.48860251190291987f * g[3];
    }
    if (DEG >= 3) {
        dx += 1.0925484305920792f * y * g[4];  dy += 1.0925484305920792f * x * g[4];
        dy += -1.0925484305920792f * z * g[5]; dz += -1.0925484305920792f * y * g[5];
        dz += 2 * 0.94617469575755997f * z * g[6];
        dx += -1.0925484305920792f * z * g[7]; dz += -1.0925484305920792f * x * g[7];
        dx += 2 * 0.54627421529603959f * x * g[8]; dy += -2 * 0.54627421529603959f * y * g[8];
    }
    if (DEG >= 4) {
        const float x2 = x * x, y2 = y * y, z2 = z * z;
        dx += 0.59004358992664352f * (-6.0f * x * y) * g[9];  dy += 0.59004358992664352f * (-3.0f * x2 + 3.0f * y2) * g[9];
        dx += 2.8906114426405538f * y * z * g[10]; dy += 2.8906114426405538f * x * z * g[10]; dz += 2.8906114426405538f * x * y * g[10];
        dy += 0.45704579946446572f * (1.0f - 5.0f * z2) * g[11]; dz += 0.45704579946446572f * y * (-10.0f * z) * g[11];
        dz += 0.3731763325901154f * (15.0f * z2 - 3.0f) * g[12];
        dx += 0.45704579946446572f * (1.0f - 5.0f * z2) * g[13]; dz += 0.45704579946446572f * x * (-10.0f * z) * g[13];
        dx += 1.4453057213202769f * z * 2 * x * g[14]; dy += -1.4453057213202769f * z * 2 * y * g[14];
        dz += 1.4453057213202769f * (x2 - y2) * g[14];
        dx += 0.59004358992664352f * (-3.0f * x2 + 3.0f * y2) * g[15]; dy += 0.59004358992664352f * 6.0f * x * y * g[15];
    }
    dL_dx[3 * i] = 2 * dx; dL_dx[3 * i + 1] = 2 * dy; dL_dx[3 * i + 2] = 2 * dz;
}

bool make_meta(const ngp_grid_desc* d, GridMeta& m)
{
    if (!d || d->n_levels < 1 || d->n_levels > NGP_MAX_LEVELS) return false;
    const uint32_t F = d->n_features;
    if (!(F == 1 || F == 2 || F == 4 || F == 8)) return false;
    m.n_levels = d->n_levels; m.n_features = F;
    for (uint32_t l = 0; l < NGP_MAX_LEVELS; l++) {
        m.offset[l] = 0; m.size[l] = 1; m.res[l] = 1; m.flags[l] = 0; m.scale[l] = 0;
    }
    for (uint32_t l = 0; l < d->n_levels; l++) {
        const uint32_t size = d->offsets[l + 1] - d->offsets[l], res = d->resolution[l];
        if (size == 0) return false;
        // tcnn's index loop: accumulate dims while stride <= size; hashed iff size < final stride
        uint64_t stride = 1;
        for (int k = 0; k < 3 && stride <= size; k++) stride *= res;
        uint32_t flags = 0;
        if (size < stride) flags |= 1u;
        if ((size & (size - 1)) == 0) flags |= 2u;
        m.offset[l] = d->offsets[l]; m.size[l] = size; m.res[l] = res; m.flags[l] = flags; m.scale[l] = d->scale[l];
    }
    return true;
}

template <int F>
void launch_bwd_input(const GridMeta& m, const float* table, const float* x, const float* dL_dy, int64_t lddy,
                      int64_t n, float* dL_dx, hipStream_t st)
{
    constexpr int LPI = F >= 4 ? F / 4 : 1;
    const int64_t n_items = n * m.n_levels;
    const int group = (int)m.n_levels * LPI;
    const dim3 grid(ngp_blocks(n_items * LPI, 256));
#define BWD_IN(G) hipLaunchKernelGGL((grid_bwd_input_kernel<F, G>), grid, dim3(256), 0, st, m, table, x, dL_dy, lddy, n_items, dL_dx)
    switch (group) {
        case 1: BWD_IN(1); break;
        case 2: BWD_IN(2); break;
        case 4: BWD_IN(4); break;
        case 8: BWD_IN(8); break;
        case 16: BWD_IN(16); break;
        case 32: BWD_IN(32); break;
        case 64: BWD_IN(64); break;
        default:
            hipLaunchKernelGGL(grid_bwd_input_serial_kernel<F>, dim3(ngp_blocks(n, 256)), dim3(256), 0, st, m, table, x,
                               dL_dy, lddy, n, dL_dx);
    }
#undef BWD_IN
}

} // namespace

extern "C" {

int64_t ngp_grid_layout(int n_levels, int n_features, int log2_hashmap_size, int base_resolution,
                        double per_level_scale, ngp_grid_desc* d)
{
    if (!d || n_levels < 1 || n_levels > NGP_MAX_LEVELS || log2_hashmap_size < 1 || log2_hashmap_size > 31 ||
        base_resolution < 1) return NGP_EINVAL;
    if (!(n_features == 1 || n_features == 2 || n_features == 4 || n_features == 8)) return NGP_EINVAL;
    d->n_levels = (uint32_t)n_levels; d->n_features = (uint32_t)n_features;
    const float l2 = log2f((float)per_level_scale);
    uint32_t off = 0;
    for (int l = 0; l < n_levels; l++) {
        const float sc = exp2f(l * l2) * base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(sc) + 1;
        const uint32_t cap = 1u << log2_hashmap_size;
        const uint64_t dense = (uint64_t)res * res * res;
        uint32_t p = dense > (uint64_t)0xFFFFFFF0u ? 0xFFFFFFF0u : (uint32_t)dense;
        p = (p + 7u) / 8u * 8u;
        if (p > cap) p = cap;
        d->scale[l] = sc; d->resolution[l] = res; d->offsets[l] = off;
        off += p;
    }
    d->offsets[n_levels] = off;
    for (int l = n_levels; l < NGP_MAX_LEVELS; l++) { d->scale[l] = 0; d->resolution[l] = 0; d->offsets[l + 1] = off; }
    return (int64_t)off * n_features;
}

#define GRID_DISPATCH_F(F_, CALL)            \
    switch (F_) {                            \
        case 1: { constexpr int F = 1; CALL; } break; \
        case 2: { constexpr int F = 2; CALL; } break; \
        case 4: { constexpr int F = 4; CALL; } break; \
        case 8: { constexpr int F = 8; CALL; } break; \
        default: return NGP_EINVAL;          \
    }

// A/B build: NGP_GRID_GATHER_OLD=1 (looked up per call, so that one process can time both) selects the
// item-per-(sample, level) gathers; the product build always takes the tiles for F = 8.
// the tile kernels take F = 8 layouts whose hashed levels have power-of-two sizes (tcnn's always do) and address
// the table with 32-bit byte offsets
static bool tile_layout_ok(const GridMeta& m, const ngp_grid_desc* d)
{
    if (m.n_features != 8) return false;
    if ((uint64_t)d->offsets[m.n_levels] * 32u >= (1ull << 32)) return false;
    for (uint32_t l = 0; l < m.n_levels; l++) {
        if ((m.flags[l] & 1u) && !(m.flags[l] & 2u)) return false;
        if (!(m.flags[l] & 1u) && (uint64_t)m.res[l] * m.res[l] * m.res[l] > m.size[l]) return false;
    }
    return true;
}

static inline int gather_variant()   // 0: item-per-(sample, level), 1: run leaders (product), 2: staged unique cells
{
#ifdef NGP_AB_VARIANTS
    const char* e = getenv("NGP_GRID_GATHER_OLD");
    return e ? (e[0] == '1' ? 0 : e[0] == '2' ? 2 : 1) : 1;
#else
    return 1;
#endif
}
static inline bool tile_gathers() { return gather_variant() != 0; }

static bool ld_ok(const GridMeta& m, int64_t ld, const void* p)
{
    const int64_t w = (int64_t)m.n_levels * m.n_features;
    if (ld < w) return false;
    if (m.n_features >= 4) return (ld % 4 == 0) && (((uintptr_t)p & 15) == 0);
    if (m.n_features == 2) return (ld % 2 == 0) && (((uintptr_t)p & 7) == 0);
    return true;
}

int ngp_grid_fwd(const ngp_grid_desc* desc, const float* table, const float* x, int64_t n, float* y, int64_t ldy,
                 void* stream)
{
    GridMeta m;
    if (!make_meta(desc, m) || n < 0) return NGP_EINVAL;
    if (n > 0 && !ld_ok(m, ldy, y)) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!table || !x || !y) return NGP_EINVAL;
    const int64_t n_items = n * m.n_levels;
    hipStream_t st = (hipStream_t)stream;
    GRID_DISPATCH_F(m.n_features, {
        constexpr int LPI = F >= 4 ? F / 4 : 1;
#ifdef NGP_AB_VARIANTS
        static const int fwd_exp = getenv("NGP_GRID_FWD_EXP") ? atoi(getenv("NGP_GRID_FWD_EXP")) : 0;
        if (F == 8 && fwd_exp) {
            const dim3 grid(ngp_blocks(n_items * LPI, 256));
            if (fwd_exp == 1) hipLaunchKernelGGL((grid_fwd_kernel<8, 1>), grid, dim3(256), 0, st, m, table, x, n_items, y, ldy);
            else if (fwd_exp == 2) hipLaunchKernelGGL((grid_fwd_kernel<8, 2>), grid, dim3(256), 0, st, m, table, x, n_items, y, ldy);
            else hipLaunchKernelGGL((grid_fwd_kernel<8, 3>), grid, dim3(256), 0, st, m, table, x, n_items, y, ldy);
            return ngp_check_launch();
        }
#endif
        if (F == 8 && tile_layout_ok(m, desc) && tile_gathers()) {   // the reference's tables: ray-coherent tiles, one load per unique cell corner
            const int64_t waves = ((n + tile::S - 1) / tile::S) * ((m.n_levels + tile::LV - 1) / tile::LV);
#ifdef NGP_AB_VARIANTS
            if (gather_variant() == 2) {
                hipLaunchKernelGGL(grid_fwd_tile_kernel, dim3(ngp_blocks(waves * 64, 256)), dim3(256), 0, st, m, table, x, n,
                                   y, ldy);
                return ngp_check_launch();
            }
#endif
            hipLaunchKernelGGL(grid_fwd_run_kernel, dim3(ngp_blocks(waves * 64, 256)), dim3(256), 0, st, m, table, x, n,
                                   y, ldy);
        } else
            hipLaunchKernelGGL(grid_fwd_kernel<F>, dim3(ngp_blocks(n_items * LPI, 256)), dim3(256), 0, st, m, table, x,
                               n_items, y, ldy);
    });
    return ngp_check_launch();
}

int ngp_grid_bwd_param(const ngp_grid_desc* desc, const float* x, const float* dL_dy, int64_t lddy, int64_t n,
                       float* dtable, void* stream)
{
    return ngp_grid_bwd_param_scaled(desc, x, dL_dy, lddy, nullptr, n, dtable, stream);
}

int ngp_grid_bwd_param_scaled(const ngp_grid_desc* desc, const float* x, const float* dL_dy, int64_t lddy,
                              const float* row_scale, int64_t n, float* dtable, void* stream)
{
    GridMeta m;
    if (!make_meta(desc, m) || n < 0) return NGP_EINVAL;
    if (n > 0 && lddy < (int64_t)m.n_levels * m.n_features) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!x || !dL_dy || !dtable) return NGP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    GRID_DISPATCH_F(m.n_features, {
        constexpr int CHUNK = 32;
        constexpr int LV2 = 64 / (2 * F) > 0 ? 64 / (2 * F) : 1;
        const int64_t waves2 = ((n + CHUNK - 1) / CHUNK) * ((m.n_levels + LV2 - 1) / LV2);
#ifdef NGP_AB_VARIANTS
        // A/B build only: the superseded variants, selected once per process
        static const int variant = getenv("NGP_GRID_BWD_SIMPLE") ? 1 : getenv("NGP_GRID_BWD_NOPAIR") ? 2
                                   : getenv("NGP_GRID_BWD_NOSLIDE") ? 3 : getenv("NGP_GRID_BWD_NOLINE") ? 4 : 0;
        static const int lds_pad = getenv("NGP_SCATTER_LDS") ? atoi(getenv("NGP_SCATTER_LDS")) : 0;
        constexpr int LV = 64 / F;
        const int64_t waves = ((n + CHUNK - 1) / CHUNK) * ((m.n_levels + LV - 1) / LV);
        if (variant != 0 && variant != 4 && row_scale) return NGP_EINVAL;   // the superseded variants take no row scale
        if (variant == 1) {
            const int64_t n_items = n * m.n_levels;
            hipLaunchKernelGGL(grid_bwd_param_kernel<F>, dim3(ngp_blocks(n_items * F, 256)), dim3(256), 0, st, m, x,
                               dL_dy, lddy, n_items, dtable);
            return ngp_check_launch();
        }
        if (variant == 2) {
            hipLaunchKernelGGL((grid_bwd_param_merge_kernel<F, CHUNK>), dim3(ngp_blocks(waves * 64, 256)), dim3(256),
                               lds_pad, st, m, x, dL_dy, lddy, n, dtable);
            return ngp_check_launch();
        }
        if (variant == 3) {
            hipLaunchKernelGGL((grid_bwd_param_merge2_kernel<F, CHUNK>), dim3(ngp_blocks(waves2 * 64, 256)), dim3(256),
                               lds_pad, st, m, x, dL_dy, lddy, n, dtable);
            return ngp_check_launch();
        }
        // superseded sliding-window line kernel: 8 as it was, 1 without atomics, 2 / 3 rolled (with / without atomics);
        // two-phase kernel: 4 with 32-sample chunks, 5 without atomics, 6 = product (64), 7 with 128-sample chunks
        static const int line_mode = getenv("NGP_SCATTER_MODE") ? atoi(getenv("NGP_SCATTER_MODE")) : 0;
        if (F == 8 && variant == 0 && line_mode != 0) {
            const dim3 grid(ngp_blocks(waves2 * 64, 256));
            if (line_mode == 8)
                hipLaunchKernelGGL((grid_bwd_param_line_kernel<CHUNK, 0>), grid, dim3(256), 0, st, m, x, dL_dy, lddy, row_scale, n, dtable);
            else if (line_mode == 1)
                hipLaunchKernelGGL((grid_bwd_param_line_kernel<CHUNK, 1>), grid, dim3(256), 0, st, m, x, dL_dy, lddy, row_scale, n, dtable);
            else if (line_mode == 2)
                hipLaunchKernelGGL((grid_bwd_param_line_rolled_kernel<CHUNK, 0>), grid, dim3(256), 0, st, m, x, dL_dy, lddy, row_scale, n, dtable);
            else if (line_mode == 3)
                hipLaunchKernelGGL((grid_bwd_param_line_rolled_kernel<CHUNK, 1>), grid, dim3(256), 0, st, m, x, dL_dy, lddy, row_scale, n, dtable);
            else if (line_mode == 4)
                hipLaunchKernelGGL((grid_bwd_param_tag_kernel<CHUNK, 0>), grid, dim3(256), 0, st, m, x, dL_dy, lddy, row_scale, n, dtable);
            else if (line_mode == 5)
                hipLaunchKernelGGL((grid_bwd_param_tag_kernel<CHUNK, 1>), grid, dim3(256), 0, st, m, x, dL_dy, lddy, row_scale, n, dtable);
            else if (line_mode == 6) {
                const int64_t w = ((n + 63) / 64) * ((m.n_levels + 3) / 4);
                hipLaunchKernelGGL((grid_bwd_param_tag_kernel<64, 0>), dim3(ngp_blocks(w * 64, 256)), dim3(256), 0, st, m, x, dL_dy, lddy, row_scale, n, dtable);
            } else {
                const int64_t w = ((n + 127) / 128) * ((m.n_levels + 3) / 4);
                hipLaunchKernelGGL((grid_bwd_param_tag_kernel<128, 0>), dim3(ngp_blocks(w * 64, 256)), dim3(256), 0, st, m, x, dL_dy, lddy, row_scale, n, dtable);
            }
            return ngp_check_launch();
        }
        static const int line_chunk = getenv("NGP_SCATTER_CHUNK") ? atoi(getenv("NGP_SCATTER_CHUNK")) : 32;
        if (F == 8 && variant == 0 && line_chunk != 32) {
            const int64_t w = ((n + line_chunk - 1) / line_chunk) * ((m.n_levels + 3) / 4);
            if (line_chunk == 64)
                hipLaunchKernelGGL(grid_bwd_param_line_kernel<64>, dim3(ngp_blocks(w * 64, 256)), dim3(256), 0, st, m, x,
                                   dL_dy, lddy, row_scale, n, dtable);
            else
                hipLaunchKernelGGL(grid_bwd_param_line_kernel<128>, dim3(ngp_blocks(w * 64, 256)), dim3(256), 0, st, m, x,
                                   dL_dy, lddy, row_scale, n, dtable);
            return ngp_check_launch();
        }
        const bool line = F == 8 && variant == 0 && lddy % 4 == 0 && ((uintptr_t)dL_dy & 15) == 0 &&
                          (uint64_t)desc->offsets[m.n_levels] * 8u < (1ull << 32);
#else
        // the two-phase kernel stages the gradient rows with 16-byte loads (row stride and base must allow them) and
        // addresses the table with 32-bit element indices (below 2^32 floats: 2^29 rows)
        const bool line = F == 8 && lddy % 4 == 0 && ((uintptr_t)dL_dy & 15) == 0 &&
                          (uint64_t)desc->offsets[m.n_levels] * 8u < (1ull << 32);
#endif
        if (line) {  // F = 8 (the reference's tables): accumulate per 64-byte line, two-phase kernel, 64-sample chunks
            constexpr int TCHUNK = 64;
            const int64_t wt = ((n + TCHUNK - 1) / TCHUNK) * ((m.n_levels + 3) / 4);
            hipLaunchKernelGGL((grid_bwd_param_tag_kernel<TCHUNK, 0>), dim3(ngp_blocks(wt * 64, 256)), dim3(256), 0, st,
                               m, x, dL_dy, lddy, row_scale, n, dtable);
        } else
            hipLaunchKernelGGL((grid_bwd_param_slide_kernel<F, CHUNK>), dim3(ngp_blocks(waves2 * 64, 256)), dim3(256),
                               0, st, m, x, dL_dy, lddy, row_scale, n, dtable);
    });
    return ngp_check_launch();
}

int ngp_grid_bwd_input(const ngp_grid_desc* desc, const float* table, const float* x, const float* dL_dy,
                       int64_t lddy, int64_t n, float* dL_dx, void* stream)
{
    GridMeta m;
    if (!make_meta(desc, m) || n < 0) return NGP_EINVAL;
    if (n > 0 && !ld_ok(m, lddy, dL_dy)) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!table || !x || !dL_dy || !dL_dx) return NGP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t wpc = (m.n_levels + tile::LV - 1) / tile::LV;
    if (4 % wpc == 0 && tile_layout_ok(m, desc) && tile_gathers()) {
        const int64_t waves = ((n + tile::S - 1) / tile::S) * wpc;
#ifdef NGP_AB_VARIANTS
        if (gather_variant() == 2) {
            hipLaunchKernelGGL(grid_bwd_input_tile_kernel, dim3(ngp_blocks(waves, 4)), dim3(256), 0, st, m, table, x, dL_dy,
                               lddy, n, dL_dx);
            return ngp_check_launch();
        }
#endif
        hipLaunchKernelGGL(grid_bwd_input_run_kernel, dim3(ngp_blocks(waves, 4)), dim3(256), 0, st, m, table, x, dL_dy,
                               lddy, n, dL_dx);
        return ngp_check_launch();
    }
    GRID_DISPATCH_F(m.n_features, { launch_bwd_input<F>(m, table, x, dL_dy, lddy, n, dL_dx, st); });
    return ngp_check_launch();
}

int ngp_grid_bwd_bwd_input(const ngp_grid_desc* desc, const float* table, const float* x, const float* dL_dy,
                           int64_t lddy, const float* dL_ddLdx, int64_t n, float* dtable, float* dL_ddLdy,
                           void* stream)
{
    GridMeta m;
    if (!make_meta(desc, m) || n < 0) return NGP_EINVAL;
    if (n > 0 && lddy < (int64_t)m.n_levels * m.n_features) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!table || !x || !dL_dy || !dL_ddLdx) return NGP_EINVAL;
    const int64_t n_items = n * m.n_levels;
    hipStream_t st = (hipStream_t)stream;
    GRID_DISPATCH_F(m.n_features, {
        hipLaunchKernelGGL(grid_bwd_bwd_input_kernel<F>, dim3(ngp_blocks(n_items * F, 256)), dim3(256), 0, st, m,
                           table, x, dL_dy, lddy, dL_ddLdx, n_items, dtable, dL_ddLdy);
    });
    return ngp_check_launch();
}

int ngp_sh_fwd(const float* x, int64_t n, int degree, float* y, int64_t ldy, void* stream)
{
    if (n < 0 || degree < 1 || degree > 4 || ldy < degree * degree) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!x || !y) return NGP_EINVAL;
    const dim3 grid(ngp_blocks(n, 256));
    hipStream_t st = (hipStream_t)stream;
    switch (degree) {
        case 1: hipLaunchKernelGGL(sh_fwd_kernel<1>, grid, dim3(256), 0, st, x, n, y, ldy); break;
        case 2: hipLaunchKernelGGL(sh_fwd_kernel<2>, grid, dim3(256), 0, st, x, n, y, ldy); break;
        case 3: hipLaunchKernelGGL(sh_fwd_kernel<3>, grid, dim3(256), 0, st, x, n, y, ldy); break;
        default: hipLaunchKernelGGL(sh_fwd_kernel<4>, grid, dim3(256), 0, st, x, n, y, ldy); break;
    }
    return ngp_check_launch();
}

int ngp_sh_fwd_dirs(const float* d, int64_t n, int degree, float* y, int64_t ldy, void* stream)
{
    if (n < 0 || degree < 1 || degree > 4 || ldy < degree * degree) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!d || !y) return NGP_EINVAL;
    const dim3 grid(ngp_blocks(n, 256));
    hipStream_t st = (hipStream_t)stream;
    switch (degree) {
        case 1: hipLaunchKernelGGL(sh_fwd_dirs_kernel<1>, grid, dim3(256), 0, st, d, n, y, ldy); break;
        case 2: hipLaunchKernelGGL(sh_fwd_dirs_kernel<2>, grid, dim3(256), 0, st, d, n, y, ldy); break;
        case 3: hipLaunchKernelGGL(sh_fwd_dirs_kernel<3>, grid, dim3(256), 0, st, d, n, y, ldy); break;
        default: hipLaunchKernelGGL(sh_fwd_dirs_kernel<4>, grid, dim3(256), 0, st, d, n, y, ldy); break;
    }
    return ngp_check_launch();
}

int ngp_sh_bwd_input(const float* x, const float* dL_dy, int64_t n, int degree, float* dL_dx, void* stream)
{
    if (n < 0 || degree < 1 || degree > 4) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!x || !dL_dy || !dL_dx) return NGP_EINVAL;
    const dim3 grid(ngp_blocks(n, 256));
    hipStream_t st = (hipStream_t)stream;
    switch (degree) {
        case 1: hipLaunchKernelGGL(sh_bwd_kernel<1>, grid, dim3(256), 0, st, x, dL_dy, n, dL_dx); break;
        case 2: hipLaunchKernelGGL(sh_bwd_kernel<2>, grid, dim3(256), 0, st, x, dL_dy, n, dL_dx); break;
        case 3: hipLaunchKernelGGL(sh_bwd_kernel<3>, grid, dim3(256), 0, st, x, dL_dy, n, dL_dx); break;
        default: hipLaunchKernelGGL(sh_bwd_kernel<4>, grid, dim3(256), 0, st, x, dL_dy, n, dL_dx); break;
    }
    return ngp_check_launch();
}

} // extern "C"
