// Microbenchmark: rate of no-return global float atomics as a function of the ACCESS SHAPE of one
// wave-instruction (64 lanes x 4 bytes), for the shapes a hash-grid scatter can produce with F = 8
// (32-byte rows).  Fills the hole in MI355X_MICROARCH.md "Global float atomics" (segments of 8-64 B
// unmeasured).  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_shapes tools/atomic_shapes.hip
// Run under rocprofv3 --pmc TCC_EA0_ATOMIC_sum WRITE_SIZE to get requests / bytes per launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

// SEG = contiguous floats per segment (64, 32, 16, 8, 4, 1); MIS = 1: segments start 8 floats (32 B) off a
// 64-byte boundary, so a 16-float pair straddles two lines
template <int SEG, int MIS>
__global__ void __launch_bounds__(256) atomic_shape_kernel(float* __restrict__ tbl, uint32_t n_units /* table size / SEG units */,
                                                           int iters, uint32_t seed)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t seg = lane / SEG, off = lane % SEG;
    for (int it = 0; it < iters; it++) {
        const uint32_t r = mix(seed ^ mix(wave * 1315423911u + it * 2654435761u + seg * 40503u));
        uint32_t unit = r % (n_units - 2);
        size_t idx = (size_t)unit * SEG + off + (MIS ? 8 : 0);
        __hip_atomic_fetch_add(tbl + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// pairs: 4 pairs of 32-byte rows per instruction; PAIRED = 1: rows (2k, 2k+1) of a random line (one 64-byte line),
// PAIRED = 0: the two rows are independent random rows (8 distinct lines)
template <int PAIRED>
__global__ void __launch_bounds__(256) atomic_rows_kernel(float* __restrict__ tbl, uint32_t n_rows, int iters, uint32_t seed)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t grp = lane >> 4, xb = (lane >> 3) & 1, f = lane & 7;
    for (int it = 0; it < iters; it++) {
        uint32_t row;
        if (PAIRED) {
            const uint32_t r = mix(seed ^ mix(wave * 1315423911u + it * 2654435761u + grp * 40503u));
            row = ((r % (n_rows / 2)) << 1) | xb;
        } else {
            const uint32_t r = mix(seed ^ mix(wave * 1315423911u + it * 2654435761u + (grp * 2 + xb) * 40503u));
            row = r % n_rows;
        }
        __hip_atomic_fetch_add(tbl + (size_t)row * 8 + f, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename K>
static void run(const char* name, K launch, double bytes_per_instr, int segs, int iters, int blocks)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch(2);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)blocks * 4 * iters;
    printf("%-34s %8.3f ms  %7.1f GB/s added  %6.2f G segments/s  %6.2f G wave-instr/s\n", name, ms,
           instr * bytes_per_instr / ms / 1e6, instr * segs / ms / 1e6, instr / ms / 1e6);
}

int main(int argc, char** argv)
{
    const size_t n_floats = (size_t)128 << 20;   // 512 MB table
    float* tbl;
    CHECK(hipMalloc(&tbl, n_floats * 4));
    CHECK(hipMemset(tbl, 0, n_floats * 4));
    const int blocks = 4096, iters = 256;
#define SHAPE(SEG, MIS, label) run(label, [&](uint32_t s) { hipLaunchKernelGGL((atomic_shape_kernel<SEG, MIS>), dim3(blocks), dim3(256), 0, 0, tbl, (uint32_t)(n_floats / SEG), iters, s); }, 256.0, 64 / SEG, iters, blocks)
    SHAPE(64, 0, "1 x 256 B contiguous");
    SHAPE(32, 0, "2 x 128 B");
    SHAPE(16, 0, "4 x 64 B (line aligned)");
    SHAPE(16, 1, "4 x 64 B (straddling two lines)");
    SHAPE(8, 0, "8 x 32 B rows");
    SHAPE(4, 0, "16 x 16 B");
    SHAPE(1, 0, "64 x 4 B");
    run("4 row pairs, same line (2k,2k+1)", [&](uint32_t s) { hipLaunchKernelGGL(atomic_rows_kernel<1>, dim3(blocks), dim3(256), 0, 0, tbl, (uint32_t)(n_floats / 8), iters, s); }, 256.0, 8, iters, blocks);
    run("8 independent rows", [&](uint32_t s) { hipLaunchKernelGGL(atomic_rows_kernel<0>, dim3(blocks), dim3(256), 0, 0, tbl, (uint32_t)(n_floats / 8), iters, s); }, 256.0, 8, iters, blocks);
    CHECK(hipFree(tbl));
    return 0;
}
