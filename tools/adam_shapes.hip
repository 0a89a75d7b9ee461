// Microbenchmark: achieved HBM rate of the clip + Adam sweep (p, g, m, v read; p, m, v written; g zeroed where it
// is not zero already) as a function of launch shape and cache policy, 154 M parameters (the colour table).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/adam_shapes tools/adam_shapes.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Hyper { float step_size, beta1, beta2, eps, bc2_sqrt; };

__device__ __forceinline__ void upd(float4& pp, const float4& gg, float4& mm, float4& vv, const Hyper h, float gs)
{
    float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float gr = ga[j] * gs;
        ma[j] = ma[j] + (gr - ma[j]) * (1.0f - h.beta1);
        va[j] = va[j] * h.beta2 + (1.0f - h.beta2) * gr * gr;
        const float denom = sqrtf(va[j]) / h.bc2_sqrt + h.eps;
        pa[j] = pa[j] - h.step_size * (ma[j] / denom);
    }
}

__device__ __forceinline__ bool nz(const float4& q) { return q.x != 0.0f || q.y != 0.0f || q.z != 0.0f || q.w != 0.0f; }

typedef float vf4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 ld(const float4* p)
{
    if (NT) {
        const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(float4* p, const float4& v)
{
    if (NT) {
        vf4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(p));
    } else *p = v;
}

// UNROLL independent 16-byte quads per lane and trip, grid-stride
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) adam_stride(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                                   float4* __restrict__ v, int64_t n4, Hyper h, const float* __restrict__ gsp)
{
    const float gs = *gsp;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 P[UNROLL], G[UNROLL], M[UNROLL], V[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t k = i + u * stride;
            P[u] = ld<NT>(p + k); G[u] = ld<NT>(g + k); M[u] = ld<NT>(m + k); V[u] = ld<NT>(v + k);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t k = i + u * stride;
            upd(P[u], G[u], M[u], V[u], h, gs);
            st<NT>(p + k, P[u]); st<NT>(m + k, M[u]); st<NT>(v + k, V[u]);
            if (nz(G[u])) st<NT>(g + k, zero4);
        }
    }
    for (; i < n4; i += stride) {
        float4 P = p[i], G = g[i], M = m[i], V = v[i];
        upd(P, G, M, V, h, gs);
        p[i] = P; m[i] = M; v[i] = V;
        if (nz(G)) g[i] = zero4;
    }
}

// every workgroup owns one contiguous span of each stream (DRAM page locality) and walks it in 256 x 16 B steps
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) adam_span(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                                 float4* __restrict__ v, int64_t n4, Hyper h, const float* __restrict__ gsp)
{
    const float gs = *gsp;
    const int64_t span = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * span, hi = lo + span < n4 ? lo + span : n4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t i = lo + threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < hi; i += UNROLL * 256) {
        float4 P[UNROLL], G[UNROLL], M[UNROLL], V[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t k = i + u * 256;
            P[u] = ld<NT>(p + k); G[u] = ld<NT>(g + k); M[u] = ld<NT>(m + k); V[u] = ld<NT>(v + k);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t k = i + u * 256;
            upd(P[u], G[u], M[u], V[u], h, gs);
            st<NT>(p + k, P[u]); st<NT>(m + k, M[u]); st<NT>(v + k, V[u]);
            if (nz(G[u])) st<NT>(g + k, zero4);
        }
    }
    for (; i < hi; i += 256) {
        float4 P = p[i], G = g[i], M = m[i], V = v[i];
        upd(P, G, M, V, h, gs);
        p[i] = P; m[i] = M; v[i] = V;
        if (nz(G)) g[i] = zero4;
    }
}

// m and v interleaved per 16-byte quad ([m4 | v4] = 32 contiguous bytes): three read and three write streams
// instead of four and four
template <int UNROLL>
__global__ void __launch_bounds__(256) adam_mv(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ mv,
                                               float4* __restrict__ unused, int64_t n4, Hyper h, const float* __restrict__ gsp)
{
    const float gs = *gsp;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 P[UNROLL], G[UNROLL], M[UNROLL], V[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t k = i + u * stride;
            P[u] = p[k]; G[u] = g[k]; M[u] = mv[2 * k]; V[u] = mv[2 * k + 1];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t k = i + u * stride;
            upd(P[u], G[u], M[u], V[u], h, gs);
            p[k] = P[u]; mv[2 * k] = M[u]; mv[2 * k + 1] = V[u];
            if (nz(G[u])) g[k] = zero4;
        }
    }
}

// everything in one array of [p4 | g4 | m4 | v4] records: one read and one write stream (not a layout the tables
// can have — the gathers and the scatter need p and g as they are — but the ceiling of "fewer streams")
template <int UNROLL>
__global__ void __launch_bounds__(256) adam_aos(float4* __restrict__ a, float4* __restrict__ u1, float4* __restrict__ u2,
                                                float4* __restrict__ u3, int64_t n4, Hyper h, const float* __restrict__ gsp)
{
    const float gs = *gsp;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 P[UNROLL], G[UNROLL], M[UNROLL], V[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t k = (i + u * stride) * 4;
            P[u] = a[k]; G[u] = a[k + 1]; M[u] = a[k + 2]; V[u] = a[k + 3];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t k = (i + u * stride) * 4;
            upd(P[u], G[u], M[u], V[u], h, gs);
            a[k] = P[u]; a[k + 2] = M[u]; a[k + 3] = V[u];
            if (nz(G[u])) a[k + 1] = zero4;
        }
    }
}

// the sweep without its arithmetic and without the gradient's zero-fill: 16 B read, 12 B written per parameter
__global__ void __launch_bounds__(256) copy_4r3w(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                                 float4* __restrict__ v, int64_t n4, Hyper h, const float* __restrict__ gsp)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + stride < n4; i += 2 * stride) {
        const int64_t k = i + stride;
        float4 a = p[i], b = g[i], c = m[i], d = v[i], e = p[k], f = g[k], q = m[k], r = v[k];
        a.x += b.x; e.x += f.x;
        p[i] = a; m[i] = d; v[i] = c; p[k] = e; m[k] = r; v[k] = q;
    }
}

__global__ void fill_grad(float4* g, int64_t n4, float frac_nz)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        uint32_t r = (uint32_t)i * 2654435761u; r ^= r >> 15; r *= 0x2c1b3c6du; r ^= r >> 12;
        const bool on = (r & 0xffff) < (uint32_t)(frac_nz * 65536.0f);
        g[i] = on ? make_float4(1e-3f, -2e-3f, 3e-3f, 1e-4f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

int main()
{
    const int64_t n = 154107392, n4 = n / 4;
    float4 *p, *g, *m, *v; float* gs;
    CHECK(hipMalloc(&p, n * 4)); CHECK(hipMalloc(&g, n * 4)); CHECK(hipMalloc(&m, n * 4)); CHECK(hipMalloc(&v, n * 4));
    CHECK(hipMalloc(&gs, 4));
    CHECK(hipMemset(p, 0, n * 4)); CHECK(hipMemset(m, 0, n * 4)); CHECK(hipMemset(v, 0, n * 4));
    const float one = 1.0f;
    CHECK(hipMemcpy(gs, &one, 4, hipMemcpyHostToDevice));
    const Hyper h = { 1e-2f, 0.9f, 0.999f, 1e-8f, 0.5f };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double bytes = 28.0 * n;
#define RUN(label, KERNEL, BLOCKS)                                                                   \
    {                                                                                                \
        float best = 1e9f;                                                                           \
        for (int rep = 0; rep < 4; rep++) {                                                          \
            hipLaunchKernelGGL(fill_grad, dim3(2048), dim3(256), 0, 0, g, n4, 0.75f);                \
            CHECK(hipDeviceSynchronize());                                                           \
            CHECK(hipEventRecord(e0));                                                               \
            hipLaunchKernelGGL(KERNEL, dim3(BLOCKS), dim3(256), 0, 0, p, g, m, v, n4, h, gs);        \
            CHECK(hipEventRecord(e1));                                                               \
            CHECK(hipEventSynchronize(e1));                                                          \
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));                                       \
            if (rep > 0 && ms < best) best = ms;                                                     \
        }                                                                                            \
        printf("%-44s blocks %5d  %7.3f ms  %6.2f TB/s (28 B/param)\n", label, BLOCKS, best, bytes / best / 1e9); \
    }
    RUN("stride x2 (shipped shape)", (adam_stride<2, false>), 512);
    RUN("copy 4 read / 3 write streams, no arithmetic", copy_4r3w, 512);
    RUN("copy 4 read / 3 write streams, no arithmetic", copy_4r3w, 1024);
    {   // m / v interleaved: the m buffer holds both (v is free for the duration)
        float4* mv; CHECK(hipMalloc(&mv, n * 8)); CHECK(hipMemset(mv, 0, n * 8));
        float4* keep = m; m = mv;
        RUN("m / v interleaved x2", (adam_mv<2>), 512);
        RUN("m / v interleaved x2", (adam_mv<2>), 1024);
        RUN("m / v interleaved x4", (adam_mv<4>), 512);
        m = keep; CHECK(hipFree(mv));
    }
    {   // one array of records
        float4* a; CHECK(hipMalloc(&a, n * 16)); CHECK(hipMemset(a, 0, n * 16));
        float4* keep = p; p = a;
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL((adam_aos<2>), dim3(512), dim3(256), 0, 0, p, g, m, v, n4, h, gs);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%-44s blocks %5d  %7.3f ms  %6.2f TB/s (28 B/param; gradient all zero: no zero-fill)\n", "array of [p g m v] records x2", 512, best, bytes / best / 1e9);
        p = keep; CHECK(hipFree(a));
    }
    RUN("stride x2", (adam_stride<2, false>), 1024);
    RUN("stride x2", (adam_stride<2, false>), 2048);
    RUN("stride x2", (adam_stride<2, false>), 8192);
    RUN("stride x4", (adam_stride<4, false>), 512);
    RUN("stride x4", (adam_stride<4, false>), 1024);
    RUN("stride x2 nontemporal", (adam_stride<2, true>), 512);
    RUN("stride x2 nontemporal", (adam_stride<2, true>), 1024);
    RUN("stride x2 nontemporal", (adam_stride<2, true>), 2048);
    RUN("stride x4 nontemporal", (adam_stride<4, true>), 512);
    RUN("stride x4 nontemporal", (adam_stride<4, true>), 1024);
    RUN("span x2", (adam_span<2, false>), 512);
    RUN("span x2", (adam_span<2, false>), 1024);
    RUN("span x2", (adam_span<2, false>), 2048);
    RUN("span x4", (adam_span<4, false>), 1024);
    RUN("span x2 nontemporal", (adam_span<2, true>), 512);
    RUN("span x2 nontemporal", (adam_span<2, true>), 1024);
    RUN("span x2 nontemporal", (adam_span<2, true>), 2048);
    RUN("span x4 nontemporal", (adam_span<4, true>), 1024);
    return 0;
}
