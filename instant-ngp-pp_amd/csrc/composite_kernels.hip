// Ray-segment kernels: alpha compositing (train fw/bw, alpha-only, test), Ref-NeRF normal
// regularisers, distortion loss, segment_csr.
//
// Layout: one 32-lane half-wave per rays_a row.  The lanes of a half-wave take 32
// consecutive samples of the ray (coalesced loads), the transmittance T_k = prod(1-a_j) is a
// multiplicative scan across the lanes, early termination is a ballot on T <= T_threshold,
// and per-ray outputs are one cross-lane reduction at the end.  The reference walks each ray
// serially in one thread (volumerendering.cu:84-114 etc.).
//
// Differences from the serial walk are confined to fp32 summation/product association.
#include "common.h"

namespace {

struct Seg {
    int64_t ray, start;
    int n;
};

__device__ __forceinline__ bool seg_load(const int64_t* __restrict__ rays_a, int n_rays, Seg& s, int& lane32)
{
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = gtid >> 5;
    lane32 = threadIdx.x & 31;
    if (row >= n_rays) return false;
    s.ray = rays_a[3 * (size_t)row];
    s.start = rays_a[3 * (size_t)row + 1];
    s.n = (int)rays_a[3 * (size_t)row + 2];
    return true;
}

// Per-chunk transmittance bookkeeping shared by all compositing kernels.
struct Chunk {
    float a, T_before, T_after;
    bool valid, active;
    int first; // lane of the stopping sample inside this chunk, or -1
};

__device__ __forceinline__ Chunk chunk_alpha(const float* __restrict__ sigmas, const float* __restrict__ deltas,
                                             int64_t s, bool valid, float T_run, float T_thr, int lane32)
{
    Chunk c;
    c.valid = valid;
    const float sig = valid ? sigmas[s] : 0.0f;
    const float dl = valid ? deltas[s] : 0.0f;
    c.a = valid ? 1.0f - __expf(-sig * dl) : 0.0f;
    const float om = 1.0f - c.a;
    const float pin = half_incl_scan_mul(om, lane32);
    float pex = __shfl_up(pin, 1, 32);
    if (lane32 == 0) pex = 1.0f;
    c.T_before = T_run * pex;
    c.T_after = T_run * pin;
    const bool stopped = valid && (c.T_after <= T_thr);
    c.first = half_first(__ballot(stopped), threadIdx.x & 63);
    c.active = valid && (c.first < 0 || lane32 <= c.first);
    return c;
}

// ------------------------------------------------------------------ composite_alpha_fw
__global__ void composite_alpha_fw_kernel(const float* __restrict__ sigmas, const float* __restrict__ deltas,
                                          const int64_t* __restrict__ rays_a, float T_thr, int n_rays,
                                          float* __restrict__ alphas, float* __restrict__ ws)
{
    Seg sg; int lane;
    if (!seg_load(rays_a, n_rays, sg, lane)) return;
    float T_run = 1.0f;
    int k0 = 0;
    for (; k0 < sg.n; k0 += 32) {
        const int k = k0 + lane;
        const int64_t s = sg.start + k;
        const Chunk c = chunk_alpha(sigmas, deltas, s, k < sg.n, T_run, T_thr, lane);
        if (c.valid) { alphas[s] = c.active ? c.a : 0.0f; ws[s] = c.active ? c.a * c.T_before : 0.0f; }
        if (c.first >= 0) { k0 += 32; break; }
        T_run = __shfl(c.T_after, 31, 32);
    }
    for (; k0 < sg.n; k0 += 32) {
        const int k = k0 + lane;
        if (k < sg.n) { alphas[sg.start + k] = 0.0f; ws[sg.start + k] = 0.0f; }
    }
}

// ------------------------------------------------------------------ composite_train_fw (V1)
template <int CMAX>
__global__ void composite_train_fw_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                          const float* __restrict__ normals, const float* __restrict__ sems,
                                          const float* __restrict__ deltas, const float* __restrict__ ts,
                                          const int64_t* __restrict__ rays_a, float T_thr, int classes, int n_rays,
                                          int64_t* __restrict__ total_samples, float* __restrict__ opacity,
                                          float* __restrict__ depth, float* __restrict__ rgb,
                                          float* __restrict__ normal, float* __restrict__ sem, float* __restrict__ ws)
{
    Seg sg; int lane;
    if (!seg_load(rays_a, n_rays, sg, lane)) return;
    float T_run = 1.0f;
    float aO = 0, aD = 0, aR = 0, aG = 0, aB = 0, aNx = 0, aNy = 0, aNz = 0;
    float aS[CMAX > 0 ? CMAX : 1];
#pragma unroll
    for (int c = 0; c < CMAX; c++) aS[c] = 0;
    int stop = -1;
    int k0 = 0;
    for (; k0 < sg.n; k0 += 32) {
        const int k = k0 + lane;
        const int64_t s = sg.start + k;
        const Chunk c = chunk_alpha(sigmas, deltas, s, k < sg.n, T_run, T_thr, lane);
        const float w = c.active ? c.a * c.T_before : 0.0f;
        if (c.valid) {
            ws[s] = w;
            aO += w;
            aD += w * ts[s];
            aR += w * rgbs[3 * s]; aG += w * rgbs[3 * s + 1]; aB += w * rgbs[3 * s + 2];
            aNx += w * normals[3 * s]; aNy += w * normals[3 * s + 1]; aNz += w * normals[3 * s + 2];
#pragma unroll
            for (int cc = 0; cc < CMAX; cc++)
                if (cc < classes) aS[cc] += w * sems[s * classes + cc];
        }
        if (c.first >= 0) { stop = k0 + c.first; k0 += 32; break; }
        T_run = __shfl(c.T_after, 31, 32);
    }
    for (; k0 < sg.n; k0 += 32) { // samples behind the stop: zero weight
        const int k = k0 + lane;
        if (k < sg.n) ws[sg.start + k] = 0.0f;
    }
    aO = half_sum(aO); aD = half_sum(aD);
    aR = half_sum(aR); aG = half_sum(aG); aB = half_sum(aB);
    aNx = half_sum(aNx); aNy = half_sum(aNy); aNz = half_sum(aNz);
#pragma unroll
    for (int cc = 0; cc < CMAX; cc++)
        if (cc < classes) aS[cc] = half_sum(aS[cc]);
    if (lane == 0) {
        const size_t r = (size_t)sg.ray;
        total_samples[r] = stop >= 0 ? stop : sg.n;
        opacity[r] = aO; depth[r] = aD;
        rgb[3 * r] = aR; rgb[3 * r + 1] = aG; rgb[3 * r + 2] = aB;
        normal[3 * r] = aNx; normal[3 * r + 1] = aNy; normal[3 * r + 2] = aNz;
#pragma unroll
        for (int cc = 0; cc < CMAX; cc++)
            if (cc < classes) sem[r * classes + cc] = aS[cc];
    }
}

// ------------------------------------------------------------------ composite_train_bw (V2)
template <int CMAX>
__global__ void composite_train_bw_kernel(const float* __restrict__ dL_dopacity, const float* __restrict__ dL_ddepth,
                                          const float* __restrict__ dL_drgb, const float* __restrict__ dL_dnormal,
                                          const float* __restrict__ dL_dsem, const float* __restrict__ dL_dws,
                                          const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                          const float* __restrict__ ws, const float* __restrict__ deltas,
                                          const float* __restrict__ ts, const int64_t* __restrict__ rays_a,
                                          const float* __restrict__ opacity, const float* __restrict__ depth,
                                          const float* __restrict__ rgb, float T_thr, int classes, int n_rays,
                                          float* __restrict__ dL_dsigmas, float* __restrict__ dL_drgbs,
                                          float* __restrict__ dL_dnormals, float* __restrict__ dL_dsems)
{
    Seg sg; int lane;
    if (!seg_load(rays_a, n_rays, sg, lane)) return;
    if (sg.n <= 0) return;
    const size_t r = (size_t)sg.ray;
    const float R = rgb[3 * r], G = rgb[3 * r + 1], B = rgb[3 * r + 2];
    const float O = opacity[r], D = depth[r];
    // a NULL upstream gradient is an all-zero one (outputs the loss never touched)
    float gR = 0.0f, gG = 0.0f, gB = 0.0f;
    if (dL_drgb) { gR = dL_drgb[3 * r]; gG = dL_drgb[3 * r + 1]; gB = dL_drgb[3 * r + 2]; }
    const float gO = dL_dopacity ? dL_dopacity[r] : 0.0f, gD = dL_ddepth ? dL_ddepth[r] : 0.0f;
    float gNx = 0.0f, gNy = 0.0f, gNz = 0.0f;
    if (dL_dnormals) { gNx = dL_dnormal[3 * r]; gNy = dL_dnormal[3 * r + 1]; gNz = dL_dnormal[3 * r + 2]; }
    float gS[CMAX > 0 ? CMAX : 1];
#pragma unroll
    for (int cc = 0; cc < CMAX; cc++) gS[cc] = (dL_dsems && cc < classes) ? dL_dsem[r * classes + cc] : 0.0f;

    // total of dL_dws*ws over the whole segment (volumerendering.cu:206-210)
    float tot = 0.0f;
    if (dL_dws) {
        for (int k = lane; k < sg.n; k += 32) tot += dL_dws[sg.start + k] * ws[sg.start + k];
        tot = half_sum(tot);
    }

    float T_run = 1.0f, r_run = 0, g_run = 0, b_run = 0, d_run = 0, p_run = 0;
    int k0 = 0;
    for (; k0 < sg.n; k0 += 32) {
        const int k = k0 + lane;
        const int64_t s = sg.start + k;
        const Chunk c = chunk_alpha(sigmas, deltas, s, k < sg.n, T_run, T_thr, lane);
        const float w = c.valid ? c.a * c.T_before : 0.0f;
        float cr = 0, cg = 0, cb = 0, tt = 0, dws = 0, dl = 0, wsv = 0;
        if (c.valid) {
            cr = rgbs[3 * s]; cg = rgbs[3 * s + 1]; cb = rgbs[3 * s + 2];
            tt = ts[s]; dws = dL_dws ? dL_dws[s] : 0.0f; dl = deltas[s]; wsv = ws[s];
        }
        const float ri = r_run + half_incl_scan_add(w * cr, lane);
        const float gi = g_run + half_incl_scan_add(w * cg, lane);
        const float bi = b_run + half_incl_scan_add(w * cb, lane);
        const float di = d_run + half_incl_scan_add(w * tt, lane);
        const float pi = p_run + half_incl_scan_add(dws * wsv, lane);
        if (c.valid) {
            const float wa = c.active ? w : 0.0f;
            dL_drgbs[3 * s] = gR * wa; dL_drgbs[3 * s + 1] = gG * wa; dL_drgbs[3 * s + 2] = gB * wa;
            if (dL_dnormals) {
                dL_dnormals[3 * s] = gNx * wa; dL_dnormals[3 * s + 1] = gNy * wa; dL_dnormals[3 * s + 2] = gNz * wa;
            }
            if (dL_dsems) {
#pragma unroll
                for (int cc = 0; cc < CMAX; cc++)
                    if (cc < classes) dL_dsems[s * classes + cc] = gS[cc] * wa;
            }
            const float T = c.T_after;
            const float v = dl * (gR * (cr * T - (R - ri)) +
                                  gG * (cg * T - (G - gi)) +
                                  gB * (cb * T - (B - bi)) +
                                  gO * (1 - O) +
                                  gD * (tt * T - (D - di)) +
                                  T * dws - (tot - pi));
            dL_dsigmas[s] = c.active ? v : 0.0f;
        }
        if (c.first >= 0) { k0 += 32; break; }
        T_run = __shfl(c.T_after, 31, 32);
        r_run = __shfl(ri, 31, 32); g_run = __shfl(gi, 31, 32); b_run = __shfl(bi, 31, 32);
        d_run = __shfl(di, 31, 32); p_run = __shfl(pi, 31, 32);
    }
    for (; k0 < sg.n; k0 += 32) {
        const int k = k0 + lane;
        if (k < sg.n) {
            const int64_t s = sg.start + k;
            dL_dsigmas[s] = 0.0f;
            dL_drgbs[3 * s] = 0.0f; dL_drgbs[3 * s + 1] = 0.0f; dL_drgbs[3 * s + 2] = 0.0f;
            if (dL_dnormals) { dL_dnormals[3 * s] = 0.0f; dL_dnormals[3 * s + 1] = 0.0f; dL_dnormals[3 * s + 2] = 0.0f; }
            if (dL_dsems) for (int cc = 0; cc < classes; cc++) dL_dsems[s * classes + cc] = 0.0f;
        }
    }
}

// Generic-classes fallbacks (classes > 32): semantic channels handled by a second sweep that
// re-derives w from the ws written by the first sweep.
__global__ void composite_sem_fw_kernel(const float* __restrict__ ws, const float* __restrict__ sems,
                                        const int64_t* __restrict__ rays_a, int classes, int n_rays,
                                        float* __restrict__ sem)
{
    Seg sg; int lane;
    if (!seg_load(rays_a, n_rays, sg, lane)) return;
    for (int cc = 0; cc < classes; cc++) {
        float a = 0.0f;
        for (int k = lane; k < sg.n; k += 32) a += ws[sg.start + k] * sems[(sg.start + k) * classes + cc];
        a = half_sum(a);
        if (lane == 0) sem[(size_t)sg.ray * classes + cc] = a;
    }
}

__global__ void composite_sem_bw_kernel(const float* __restrict__ dL_dsem, const float* __restrict__ w_eff,
                                        const int64_t* __restrict__ rays_a, int classes, int n_rays,
                                        float* __restrict__ dL_dsems)
{
    Seg sg; int lane;
    if (!seg_load(rays_a, n_rays, sg, lane)) return;
    for (int k = lane; k < sg.n; k += 32) {
        const int64_t s = sg.start + k;
        const float w = w_eff[s];
        for (int cc = 0; cc < classes; cc++) dL_dsems[s * classes + cc] = dL_dsem[(size_t)sg.ray * classes + cc] * w;
    }
}

// ------------------------------------------------------------------ composite_test_fw (T1)
// One lane per alive ray (<= 64 samples per call, volumerendering.cu:335-373).
__global__ void composite_test_fw_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                         const float* __restrict__ normals, const float* __restrict__ normals_raw,
                                         const float* __restrict__ sems, const float* __restrict__ deltas,
                                         const float* __restrict__ ts, int64_t* __restrict__ alive, float T_thr,
                                         int classes, const int32_t* __restrict__ n_eff, int n_alive, int n_samples,
                                         float* __restrict__ opacity, float* __restrict__ depth,
                                         float* __restrict__ rgb, float* __restrict__ normal,
                                         float* __restrict__ normal_raw, float* __restrict__ sem,
                                         const int32_t* __restrict__ state)
{
    if (state) {   // device-driven rounds: sizes from ngp_test_round_begin's state
        if (state[3]) return;
        n_alive = state[0]; n_samples = state[1];
    }
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int ne = n_eff[n];
    if (ne == 0) { alive[n] = -1; return; }
    const size_t r = (size_t)alive[n];
    // accumulators start from the running per-ray values so that the additions happen in the
    // reference's order (volumerendering.cu:352-365 adds sample by sample into the outputs)
    float aO = opacity[r], aD = depth[r];
    float T = 1 - aO;
    float aR = rgb[3 * r], aG = rgb[3 * r + 1], aB = rgb[3 * r + 2];
    float nx = normal[3 * r], ny = normal[3 * r + 1], nz = normal[3 * r + 2];
    float rx = normal_raw[3 * r], ry = normal_raw[3 * r + 1], rz = normal_raw[3 * r + 2];
    int s = 0;
    bool dead = false;
    while (s < ne) {
        const size_t o = (size_t)n * n_samples + s;
        const float a = 1.0f - __expf(-sigmas[o] * deltas[o]);
        const float w = a * T;
        aR += w * rgbs[3 * o]; aG += w * rgbs[3 * o + 1]; aB += w * rgbs[3 * o + 2];
        aD += w * ts[o]; aO += w;
        nx += w * normals[3 * o]; ny += w * normals[3 * o + 1]; nz += w * normals[3 * o + 2];
        rx += w * normals_raw[3 * o]; ry += w * normals_raw[3 * o + 1]; rz += w * normals_raw[3 * o + 2];
        for (int c = 0; c < classes; c++) sem[r * classes + c] += w * sems[o * classes + c];
        T *= 1.0f - a;
        if (T <= T_thr) { dead = true; break; }
        s++;
    }
    rgb[3 * r] = aR; rgb[3 * r + 1] = aG; rgb[3 * r + 2] = aB;
    depth[r] = aD; opacity[r] = aO;
    normal[3 * r] = nx; normal[3 * r + 1] = ny; normal[3 * r + 2] = nz;
    normal_raw[3 * r] = rx; normal_raw[3 * r + 1] = ry; normal_raw[3 * r + 2] = rz;
    if (dead) alive[n] = -1;
}

// ------------------------------------------------------------------ RefLoss (V3)
__global__ void refloss_fw_kernel(const float* __restrict__ sigmas, const float* __restrict__ ndiff,
                                  const float* __restrict__ nori, const float* __restrict__ deltas,
                                  const int64_t* __restrict__ rays_a, float T_thr, int n_rays,
                                  float* __restrict__ loss_o, float* __restrict__ loss_p)
{
    Seg sg; int lane;
    if (!seg_load(rays_a, n_rays, sg, lane)) return;
    float T_run = 1.0f, ao = 0, ax = 0, ay = 0, az = 0;
    for (int k0 = 0; k0 < sg.n; k0 += 32) {
        const int k = k0 + lane;
        const int64_t s = sg.start + k;
        const Chunk c = chunk_alpha(sigmas, deltas, s, k < sg.n, T_run, T_thr, lane);
        if (c.active) {
            const float w = c.a * c.T_before;
            ax += w * ndiff[3 * s]; ay += w * ndiff[3 * s + 1]; az += w * ndiff[3 * s + 2];
            ao += w * nori[s];
        }
        if (c.first >= 0) break;
        T_run = __shfl(c.T_after, 31, 32);
    }
    ao = half_sum(ao); ax = half_sum(ax); ay = half_sum(ay); az = half_sum(az);
    if (lane == 0) {
        const size_t r = (size_t)sg.ray;
        loss_o[r] = ao; loss_p[3 * r] = ax; loss_p[3 * r + 1] = ay; loss_p[3 * r + 2] = az;
    }
}

__global__ void refloss_bw_kernel(const float* __restrict__ dL_dlo, const float* __restrict__ dL_dlp,
                                  const float* __restrict__ sigmas, const float* __restrict__ ndiff,
                                  const float* __restrict__ nori, const float* __restrict__ deltas,
                                  const int64_t* __restrict__ rays_a, const float* __restrict__ loss_o,
                                  const float* __restrict__ loss_p, float T_thr, int n_rays,
                                  float* __restrict__ dL_dsigmas, float* __restrict__ dL_dndiff,
                                  float* __restrict__ dL_dnori)
{
    Seg sg; int lane;
    if (!seg_load(rays_a, n_rays, sg, lane)) return;
    const size_t r = (size_t)sg.ray;
    const float X = loss_p[3 * r], Y = loss_p[3 * r + 1], Z = loss_p[3 * r + 2], O = loss_o[r];
    const float gx = dL_dlp[3 * r], gy = dL_dlp[3 * r + 1], gz = dL_dlp[3 * r + 2], go = dL_dlo[r];
    float T_run = 1.0f, x_run = 0, y_run = 0, z_run = 0, o_run = 0;
    int k0 = 0;
    for (; k0 < sg.n; k0 += 32) {
        const int k = k0 + lane;
        const int64_t s = sg.start + k;
        const Chunk c = chunk_alpha(sigmas, deltas, s, k < sg.n, T_run, T_thr, lane);
        const float w = c.valid ? c.a * c.T_before : 0.0f;
        float dx = 0, dy = 0, dz = 0, no = 0, dl = 0;
        if (c.valid) { dx = ndiff[3 * s]; dy = ndiff[3 * s + 1]; dz = ndiff[3 * s + 2]; no = nori[s]; dl = deltas[s]; }
        const float xi = x_run + half_incl_scan_add(w * dx, lane);
        const float yi = y_run + half_incl_scan_add(w * dy, lane);
        const float zi = z_run + half_incl_scan_add(w * dz, lane);
        const float oi = o_run + half_incl_scan_add(w * no, lane);
        if (c.valid) {
            const float wa = c.active ? w : 0.0f;
            dL_dndiff[3 * s] = gx * wa; dL_dndiff[3 * s + 1] = gy * wa; dL_dndiff[3 * s + 2] = gz * wa;
            dL_dnori[s] = go * wa;
            const float T = c.T_after;
            const float v = dl * (gx * (dx * T - (X - xi)) + gy * (dy * T - (Y - yi)) + gz * (dz * T - (Z - zi)) +
                                  go * (no * T - (O - oi)));
            dL_dsigmas[s] = c.active ? v : 0.0f;
        }
        if (c.first >= 0) { k0 += 32; break; }
        T_run = __shfl(c.T_after, 31, 32);
        x_run = __shfl(xi, 31, 32); y_run = __shfl(yi, 31, 32); z_run = __shfl(zi, 31, 32); o_run = __shfl(oi, 31, 32);
    }
    for (; k0 < sg.n; k0 += 32) {
        const int k = k0 + lane;
        if (k < sg.n) {
            const int64_t s = sg.start + k;
            dL_dsigmas[s] = 0.0f; dL_dnori[s] = 0.0f;
            dL_dndiff[3 * s] = 0.0f; dL_dndiff[3 * s + 1] = 0.0f; dL_dndiff[3 * s + 2] = 0.0f;
        }
    }
}

// ------------------------------------------------------------------ distortion loss (D1)
__global__ void distortion_fw_kernel(const float* __restrict__ ws, const float* __restrict__ deltas,
                                     const float* __restrict__ ts, const int64_t* __restrict__ rays_a, int n_rays,
                                     float* __restrict__ loss, float* __restrict__ ws_inc, float* __restrict__ wts_inc)
{
    Seg sg; int lane;
    if (!seg_load(rays_a, n_rays, sg, lane)) return;
    float w_run = 0, wt_run = 0, acc = 0;
    for (int k0 = 0; k0 < sg.n; k0 += 32) {
        const int k = k0 + lane;
        const int64_t s = sg.start + k;
        const bool valid = k < sg.n;
        const float w = valid ? ws[s] : 0.0f;
        const float wt = valid ? w * ts[s] : 0.0f;
        const float wi = w_run + half_incl_scan_add(w, lane);
        const float wti = wt_run + half_incl_scan_add(wt, lane);
        float we = __shfl_up(wi, 1, 32), wte = __shfl_up(wti, 1, 32);
        if (lane == 0) { we = w_run; wte = wt_run; }
        if (valid) {
            ws_inc[s] = wi; wts_inc[s] = wti;
            acc += 2 * (wti * we - wi * wte) + 1.0f / 3 * w * w * deltas[s];
        }
        w_run = __shfl(wi, 31, 32); wt_run = __shfl(wti, 31, 32);
    }
    acc = half_sum(acc);
    if (lane == 0) loss[(size_t)sg.ray] = acc;
}

__global__ void distortion_bw_kernel(const float* __restrict__ dL_dloss, const float* __restrict__ ws_inc,
                                     const float* __restrict__ wts_inc, const float* __restrict__ ws,
                                     const float* __restrict__ deltas, const float* __restrict__ ts,
                                     const int64_t* __restrict__ rays_a, int n_rays, float* __restrict__ dL_dws)
{
    Seg sg; int lane;
    if (!seg_load(rays_a, n_rays, sg, lane)) return;
    if (sg.n <= 0) return; // the reference reads start-1 here (losses.cu:125-128)
    const int64_t end = sg.start + sg.n - 1;
    const float w_sum = ws_inc[end], wt_sum = wts_inc[end];
    const float g = dL_dloss[(size_t)sg.ray];
    for (int k = lane; k < sg.n; k += 32) {
        const int64_t s = sg.start + k;
        const float t = ts[s];
        const float left = (k == 0) ? 0.0f : (t * ws_inc[s - 1] - wts_inc[s - 1]);
        float v = g * 2 * (left + (wt_sum - wts_inc[s] - t * (w_sum - ws_inc[s])));
        v += g * 2.0f / 3 * ws[s] * deltas[s];
        dL_dws[s] = v;
    }
}

// ------------------------------------------------------------------ segment_csr sum (R4)
__global__ void segment_csr_kernel(const float* __restrict__ src, const int64_t* __restrict__ indptr, int n_seg,
                                   int width, float* __restrict__ out)
{
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int seg = gtid >> 5, lane = threadIdx.x & 31;
    if (seg >= n_seg) return;
    const int64_t b = indptr[seg], e = indptr[seg + 1];
    for (int c = 0; c < width; c++) {
        float a = 0.0f;
        for (int64_t k = b + lane; k < e; k += 32) a += src[k * width + c];
        a = half_sum(a);
        if (lane == 0) out[(size_t)seg * width + c] = a;
    }
}

// ------------------------------------------------------------------ fused loss / glue kernels
// NeRFLoss default terms (losses.py:96-105, reduced as train.py:307 does: sum of term means) with the
// gradients of the rgb / opacity terms, one pass over the rays:
//   terms[1] += mean (rgb-gt)^2                    d_rgb     = 2 (rgb-gt) / (3 n)
//   terms[2] += lambda_o mean(-o log o), o=op+1e-10 d_opacity = lambda_o (-log o - 1) / n
//   terms[3] += lambda_d mean(dist)                (dist = per-ray distortion loss, may be NULL)
//   terms[0] += their sum
__global__ void __launch_bounds__(256) nerf_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ gt,
                                                        const float* __restrict__ opacity,
                                                        const float* __restrict__ dist, int n_rays, float g_rgb,
                                                        float g_op, float g_dist, float* __restrict__ terms,
                                                        float* __restrict__ d_rgb, float* __restrict__ d_opacity)
{
    __shared__ float part[3][4];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    if (r < n_rays) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float e = rgb[3 * r + c] - gt[3 * r + c];
            s0 += e * e;
            d_rgb[3 * r + c] = g_rgb * 2.0f * e;
        }
        const float o = opacity[r] + 1e-10f;
        const float lg = logf(o);
        s1 = -o * lg;
        d_opacity[r] = g_op * (-lg - 1.0f);
        if (dist) s2 = dist[r];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = s0; part[1][threadIdx.x >> 6] = s1; part[2][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t_rgb = (part[0][0] + part[0][1] + part[0][2] + part[0][3]) * g_rgb;
        const float t_op = (part[1][0] + part[1][1] + part[1][2] + part[1][3]) * g_op;
        const float t_dist = (part[2][0] + part[2][1] + part[2][2] + part[2][3]) * g_dist;
        atomicAdd(terms, t_rgb + t_op + t_dist);
        atomicAdd(terms + 1, t_rgb);
        atomicAdd(terms + 2, t_op);
        atomicAdd(terms + 3, t_dist);
    }
}

// ---- W-lane groups (W = 32: a half-wave, W = 64: a whole wave per ray) for the fused kernel below ------------------
template <int W> __device__ __forceinline__ float grp_sum(float v)
{
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, W);
    return v;
}
template <int W> __device__ __forceinline__ float grp_scan_add(float v, int lane)
{
#pragma unroll
    for (int o = 1; o < W; o <<= 1) {
        const float u = __shfl_up(v, o, W);
        if (lane >= o) v += u;
    }
    return v;
}
template <int W> __device__ __forceinline__ float grp_scan_mul(float v, int lane)
{
#pragma unroll
    for (int o = 1; o < W; o <<= 1) {
        const float u = __shfl_up(v, o, W);
        if (lane >= o) v *= u;
    }
    return v;
}
template <int W> __device__ __forceinline__ int grp_first(unsigned long long ballot, int lane64)
{
    if (W == 64) return ballot ? (int)__builtin_ctzll(ballot) : -1;
    const unsigned m = (unsigned)(ballot >> (lane64 & 32));
    return m ? (__ffs((int)m) - 1) : -1;
}
template <int W> __device__ __forceinline__ bool seg_load_w(const int64_t* __restrict__ rays_a, int n_rays, Seg& s, int& lane)
{
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = gtid / W;
    lane = threadIdx.x & (W - 1);
    if (row >= n_rays) return false;
    s.ray = rays_a[3 * (size_t)row];
    s.start = rays_a[3 * (size_t)row + 1];
    s.n = (int)rays_a[3 * (size_t)row + 2];
    return true;
}
template <int W> __device__ __forceinline__ Chunk chunk_alpha_w(const float* __restrict__ sigmas, const float* __restrict__ deltas,
                                                                int64_t s, bool valid, float T_run, float T_thr, int lane)
{
    Chunk c;
    c.valid = valid;
    const float sig = valid ? sigmas[s] : 0.0f;
    const float dl = valid ? deltas[s] : 0.0f;
    c.a = valid ? 1.0f - __expf(-sig * dl) : 0.0f;
    const float om = 1.0f - c.a;
    const float pin = grp_scan_mul<W>(om, lane);
    float pex = __shfl_up(pin, 1, W);
    if (lane == 0) pex = 1.0f;
    c.T_before = T_run * pex;
    c.T_after = T_run * pin;
    const bool stopped = valid && (c.T_after <= T_thr);
    c.first = grp_first<W>(__ballot(stopped), threadIdx.x & 63);
    c.active = valid && (c.first < 0 || lane <= c.first);
    return c;
}

// ------------------------------------------------------------------ fused render + default loss + its gradients
// Everything between the field's raw outputs and the field's backward on the default recipe (rendering.py:221-249,
// losses.py:96-105, train.py:307), per ray, in ONE launch (it replaces -normalize x2, softmax, composite_train_fw,
// the RefLoss inputs + forward, distortion fw/bw, the loss and composite_train_bw: 12 launches on a serial chain):
//   pass A  compositing with early stop: ws, opacity, depth, rgb, normal_pred, semantic, total_samples, Ro, Rp, and the
//           distortion loss of the ray (running inclusive scans of w and w t);
//           per-sample normals (-normalize of d sigma/dx and of the head's output) and softmax live in registers only;
//   seeds   d_rgb = 2 (rgb - gt) / (3 R), d_opacity = lambda_o (-log o - 1) / R, d_dist = lambda_d / R; loss terms;
//   pass B  tot = sum_s dL_dws[s] ws[s] (distortion backward in closed form, losses.cu:130-139);
//   pass C  composite_train_bw's scans: dL_dsigmas, dL_drgbs (volumerendering.cu:193-245; normal / semantic terms do not
//           enter dL_dsigma there, and their own gradients are zero on this recipe).
// Every lane re-reads only what the SAME lane wrote (ws), the scans are recomputed: no cross-lane traffic through memory.
struct RenderLossArgs {
    const float *sigmas, *rgbs, *dsig_dx, *np_raw, *sem_logits, *dirs, *deltas, *ts, *gt, *scale3;
    const float* bg;   // optional (3): background colour, rgb += bg (1 - opacity) (rendering.py:236-241)
    const int64_t* rays_a;
    int64_t ld_np, ld_sem;
    float T_thr, g_rgb, g_op, g_dist;
    int classes, n_rays;
    int64_t *total_samples, *vr_samples;
    float *opacity, *depth, *rgb, *normal, *sem, *ws, *Ro, *Rp, *terms, *d_sigmas, *d_rgbs;
};

template <int CMAX, int W>
__global__ void __launch_bounds__(256) render_loss_fused_kernel(RenderLossArgs p)
{
    constexpr int RPB = 256 / W;          // rays per block
    __shared__ float part[3][RPB];
    __shared__ unsigned long long part_n[RPB];
    Seg sg; int lane;
    const bool have = seg_load_w<W>(p.rays_a, p.n_rays, sg, lane);
    float s_rgb = 0.0f, s_op = 0.0f, s_dist = 0.0f;
    unsigned long long n_used = 0;
    if (have) {
        const size_t r = (size_t)sg.ray;
        const float sc0 = p.scale3 ? p.scale3[0] : 1.0f, sc1 = p.scale3 ? p.scale3[1] : 1.0f, sc2 = p.scale3 ? p.scale3[2] : 1.0f;
        // ---------------- pass A
        float T_run = 1.0f;
        float aO = 0, aD = 0, aR = 0, aG = 0, aB = 0, aNx = 0, aNy = 0, aNz = 0, aRo = 0, aPx = 0, aPy = 0, aPz = 0;
        float aS[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; c++) aS[c] = 0;
        float w_run = 0, wt_run = 0, dacc = 0;
        int stop = -1;
        int k0 = 0;
        for (; k0 < sg.n; k0 += W) {
            const int k = k0 + lane;
            const int64_t s = sg.start + k;
            const Chunk c = chunk_alpha_w<W>(p.sigmas, p.deltas, s, k < sg.n, T_run, p.T_thr, lane);
            const float w = c.active ? c.a * c.T_before : 0.0f;
            float tt = 0.0f, dl = 0.0f;
            if (c.valid) {
                p.ws[s] = w;
                tt = p.ts[s]; dl = p.deltas[s];
            }
            if (c.active) {   // samples behind the stop carry no weight: their normals / classes are not needed
                aO += w;
                aD += w * tt;
                aR += w * p.rgbs[3 * s]; aG += w * p.rgbs[3 * s + 1]; aB += w * p.rgbs[3 * s + 2];
                // normals_raw = -normalize(d sigma/dx * scale3), normals_pred = -normalize(head), eps 1e-6 (F.normalize)
                float gx = p.dsig_dx[3 * s] * sc0, gy = p.dsig_dx[3 * s + 1] * sc1, gz = p.dsig_dx[3 * s + 2] * sc2;
                float inv = -1.0f / fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-6f);
                gx *= inv; gy *= inv; gz *= inv;
                float hx = p.np_raw[s * p.ld_np], hy = p.np_raw[s * p.ld_np + 1], hz = p.np_raw[s * p.ld_np + 2];
                inv = -1.0f / fmaxf(sqrtf(hx * hx + hy * hy + hz * hz), 1e-6f);
                hx *= inv; hy *= inv; hz *= inv;
                aNx += w * hx; aNy += w * hy; aNz += w * hz;
                // Ref-NeRF regularisers (rendering.py:243-245)
                const float ex = gx - hx, ey = gy - hy, ez = gz - hz;
                aPx += w * (ex * ex); aPy += w * (ey * ey); aPz += w * (ez * ez);
                const float dx = p.dirs[3 * s], dy = p.dirs[3 * s + 1], dz = p.dirs[3 * s + 2];
                const float dinv = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f);
                const float dot = fmaxf((gx * dx + gy * dy + gz * dz) * dinv, 0.0f);
                aRo += w * (dot * dot);
                // softmax over the class logits
                float lg[CMAX], mx = -INFINITY;
#pragma unroll
                for (int cc = 0; cc < CMAX; cc++) {
                    lg[cc] = cc < p.classes ? p.sem_logits[s * p.ld_sem + cc] : -INFINITY;
                    mx = fmaxf(mx, lg[cc]);
                }
                float den = 0.0f;
#pragma unroll
                for (int cc = 0; cc < CMAX; cc++) { lg[cc] = cc < p.classes ? __expf(lg[cc] - mx) : 0.0f; den += lg[cc]; }
                const float winv = w / den;
#pragma unroll
                for (int cc = 0; cc < CMAX; cc++) aS[cc] += winv * lg[cc];
            }
            // distortion loss (losses.cu:8-59): inclusive scans of w and w t over the ray
            const float wt = w * tt;
            const float wi = w_run + grp_scan_add<W>(w, lane);
            const float wti = wt_run + grp_scan_add<W>(wt, lane);
            float we = __shfl_up(wi, 1, W), wte = __shfl_up(wti, 1, W);
            if (lane == 0) { we = w_run; wte = wt_run; }
            if (c.valid) dacc += 2 * (wti * we - wi * wte) + 1.0f / 3 * w * w * dl;
            w_run = __shfl(wi, W - 1, W); wt_run = __shfl(wti, W - 1, W);
            if (c.first >= 0) { stop = k0 + c.first; k0 += W; break; }
            T_run = __shfl(c.T_after, W - 1, W);
        }
        for (; k0 < sg.n; k0 += W) {   // behind the stop: zero weight, zero gradients
            const int k = k0 + lane;
            if (k < sg.n) {
                const int64_t s = sg.start + k;
                p.ws[s] = 0.0f; p.d_sigmas[s] = 0.0f;
                p.d_rgbs[3 * s] = 0.0f; p.d_rgbs[3 * s + 1] = 0.0f; p.d_rgbs[3 * s + 2] = 0.0f;
            }
        }
        const int n_live = stop >= 0 ? (stop / W + 1) * W : sg.n;   // passes B and C stop behind the stop sample's chunk
        aO = grp_sum<W>(aO); aD = grp_sum<W>(aD); aR = grp_sum<W>(aR); aG = grp_sum<W>(aG); aB = grp_sum<W>(aB);
        aNx = grp_sum<W>(aNx); aNy = grp_sum<W>(aNy); aNz = grp_sum<W>(aNz);
        aRo = grp_sum<W>(aRo); aPx = grp_sum<W>(aPx); aPy = grp_sum<W>(aPy); aPz = grp_sum<W>(aPz);
        dacc = grp_sum<W>(dacc);
#pragma unroll
        for (int cc = 0; cc < CMAX; cc++) aS[cc] = grp_sum<W>(aS[cc]);
        // ---------------- loss terms and gradient seeds of the ray
        // the image colour: composited colour over the background (black when bg == NULL)
        float fR = aR, fG = aG, fB = aB, bg0 = 0.0f, bg1 = 0.0f, bg2 = 0.0f;
        if (p.bg) {
            bg0 = p.bg[0]; bg1 = p.bg[1]; bg2 = p.bg[2];
            const float rest = 1.0f - aO;
            fR = aR + bg0 * rest; fG = aG + bg1 * rest; fB = aB + bg2 * rest;
        }
        const float e0 = fR - p.gt[3 * r], e1 = fG - p.gt[3 * r + 1], e2 = fB - p.gt[3 * r + 2];
        const float gR = p.g_rgb * 2.0f * e0, gG = p.g_rgb * 2.0f * e1, gB = p.g_rgb * 2.0f * e2;
        const float oo = aO + 1e-10f;
        const float lgo = logf(oo);
        // d loss / d opacity: the entropy term, and through the background's weight (1 - opacity)
        const float gO = p.g_op * (-lgo - 1.0f) - (gR * bg0 + gG * bg1 + gB * bg2);
        const float gd = p.g_dist;
        if (lane == 0) {
            p.total_samples[r] = stop >= 0 ? stop : sg.n;
            p.opacity[r] = aO; p.depth[r] = aD;
            p.rgb[3 * r] = fR; p.rgb[3 * r + 1] = fG; p.rgb[3 * r + 2] = fB;
            p.normal[3 * r] = aNx; p.normal[3 * r + 1] = aNy; p.normal[3 * r + 2] = aNz;
            p.Ro[r] = aRo; p.Rp[3 * r] = aPx; p.Rp[3 * r + 1] = aPy; p.Rp[3 * r + 2] = aPz;
#pragma unroll
            for (int cc = 0; cc < CMAX; cc++)
                if (cc < p.classes) p.sem[r * p.classes + cc] = aS[cc];
            s_rgb = e0 * e0 + e1 * e1 + e2 * e2; s_op = -oo * lgo; s_dist = dacc;
            n_used = (unsigned long long)(stop >= 0 ? stop : sg.n);
        }
        const float w_sum = w_run, wt_sum = wt_run;
        // ---------------- pass B: tot = sum dL_dws ws  (dL_dws of the distortion term, closed form)
        float tot = 0.0f;
        if (gd != 0.0f) {
            float wr = 0, wtr = 0;
            for (int q0 = 0; q0 < n_live; q0 += W) {
                const int k = q0 + lane;
                const int64_t s = sg.start + k;
                const bool valid = k < sg.n;
                const float w = valid ? p.ws[s] : 0.0f, tt = valid ? p.ts[s] : 0.0f, dl = valid ? p.deltas[s] : 0.0f;
                const float wi = wr + grp_scan_add<W>(w, lane);
                const float wti = wtr + grp_scan_add<W>(w * tt, lane);
                float we = __shfl_up(wi, 1, W), wte = __shfl_up(wti, 1, W);
                if (lane == 0) { we = wr; wte = wtr; }
                const float dws = gd * 2 * ((tt * we - wte) + (wt_sum - wti - tt * (w_sum - wi))) + gd * 2.0f / 3 * w * dl;
                tot += dws * w;
                wr = __shfl(wi, W - 1, W); wtr = __shfl(wti, W - 1, W);
            }
            tot = grp_sum<W>(tot);
        }
        // ---------------- pass C: composite_train_bw
        {
            float T2 = 1.0f, r_run = 0, g_run = 0, b_run = 0, p_run = 0, wr = 0, wtr = 0;
            for (int q0 = 0; q0 < n_live; q0 += W) {
                const int k = q0 + lane;
                const int64_t s = sg.start + k;
                const Chunk c = chunk_alpha_w<W>(p.sigmas, p.deltas, s, k < sg.n, T2, p.T_thr, lane);
                const float w = c.valid ? c.a * c.T_before : 0.0f;
                float cr = 0, cg = 0, cb = 0, tt = 0, dl = 0, wsv = 0;
                if (c.valid) {
                    cr = p.rgbs[3 * s]; cg = p.rgbs[3 * s + 1]; cb = p.rgbs[3 * s + 2];
                    tt = p.ts[s]; dl = p.deltas[s]; wsv = p.ws[s];
                }
                const float wi = wr + grp_scan_add<W>(wsv, lane);
                const float wti = wtr + grp_scan_add<W>(wsv * tt, lane);
                float we = __shfl_up(wi, 1, W), wte = __shfl_up(wti, 1, W);
                if (lane == 0) { we = wr; wte = wtr; }
                const float dws = gd != 0.0f ? gd * 2 * ((tt * we - wte) + (wt_sum - wti - tt * (w_sum - wi))) + gd * 2.0f / 3 * wsv * dl
                                             : 0.0f;
                const float ri = r_run + grp_scan_add<W>(w * cr, lane);
                const float gi = g_run + grp_scan_add<W>(w * cg, lane);
                const float bi = b_run + grp_scan_add<W>(w * cb, lane);
                const float pi = p_run + grp_scan_add<W>(dws * wsv, lane);
                if (c.valid) {
                    const float wa = c.active ? w : 0.0f;
                    p.d_rgbs[3 * s] = gR * wa; p.d_rgbs[3 * s + 1] = gG * wa; p.d_rgbs[3 * s + 2] = gB * wa;
                    const float T = c.T_after;
                    const float v = dl * (gR * (cr * T - (aR - ri)) + gG * (cg * T - (aG - gi)) + gB * (cb * T - (aB - bi)) +
                                          gO * (1 - aO) + T * dws - (tot - pi));
                    p.d_sigmas[s] = c.active ? v : 0.0f;
                }
                if (c.first >= 0) break;
                T2 = __shfl(c.T_after, W - 1, W);
                r_run = __shfl(ri, W - 1, W); g_run = __shfl(gi, W - 1, W); b_run = __shfl(bi, W - 1, W); p_run = __shfl(pi, W - 1, W);
                wr = __shfl(wi, W - 1, W); wtr = __shfl(wti, W - 1, W);
            }
        }
    }
    // ---------------- loss terms of the block's rays: one set of atomics per block
    const int hw = threadIdx.x / W;
    if ((threadIdx.x & (W - 1)) == 0) { part[0][hw] = s_rgb; part[1][hw] = s_op; part[2][hw] = s_dist; part_n[hw] = n_used; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t0 = 0, t1 = 0, t2 = 0;
        unsigned long long tn = 0;
        for (int q = 0; q < RPB; q++) { t0 += part[0][q]; t1 += part[1][q]; t2 += part[2][q]; tn += part_n[q]; }
        t0 *= p.g_rgb; t1 *= p.g_op; t2 *= p.g_dist;
        atomicAdd(p.terms, t0 + t1 + t2);
        atomicAdd(p.terms + 1, t0);
        atomicAdd(p.terms + 2, t1);
        atomicAdd(p.terms + 3, t2);
        atomicAdd(reinterpret_cast<unsigned long long*>(p.vr_samples), tn);
    }
}

// Inputs of RefLoss (rendering.py:243-245): normals_diff = (n_raw - n_pred)^2,
// normals_ori = max(<n_raw, normalize(dir)>, 0)^2
__global__ void refloss_inputs_kernel(const float* __restrict__ n_raw, const float* __restrict__ n_pred,
                                      const float* __restrict__ dirs, int64_t n, float* __restrict__ ndiff,
                                      float* __restrict__ nori)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float rx = n_raw[3 * i], ry = n_raw[3 * i + 1], rz = n_raw[3 * i + 2];
    const float ex = rx - n_pred[3 * i], ey = ry - n_pred[3 * i + 1], ez = rz - n_pred[3 * i + 2];
    ndiff[3 * i] = ex * ex; ndiff[3 * i + 1] = ey * ey; ndiff[3 * i + 2] = ez * ez;
    const float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
    const float inv = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f);
    const float dot = fmaxf((rx * dx + ry * dy + rz * dz) * inv, 0.0f);
    nori[i] = dot * dot;
}

// y = -normalize(x * scale, eps=1e-6)  (F.normalize semantics: x / max(|x|, eps)); rows of 3
__global__ void neg_normalize_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ scale3,
                                     int64_t n, float* __restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i * ldx], b = x[i * ldx + 1], c = x[i * ldx + 2];
    if (scale3) { a *= scale3[0]; b *= scale3[1]; c *= scale3[2]; }
    const float inv = -1.0f / fmaxf(sqrtf(a * a + b * b + c * c), 1e-6f);
    y[3 * i] = a * inv; y[3 * i + 1] = b * inv; y[3 * i + 2] = c * inv;
}

// backward of y = -x/max(|x|,eps):  dx = -(g - u <u,g>)/|x| with u = x/|x|   (|x| > eps)
__global__ void neg_normalize_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ g,
                                         int64_t n, float* __restrict__ dx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[i * ldx], b = x[i * ldx + 1], c = x[i * ldx + 2];
    const float ga = g[3 * i], gb = g[3 * i + 1], gc = g[3 * i + 2];
    const float nrm = sqrtf(a * a + b * b + c * c);
    if (nrm > 1e-6f) {
        const float inv = 1.0f / nrm;
        const float ua = a * inv, ub = b * inv, uc = c * inv;
        const float dot = ua * ga + ub * gb + uc * gc;
        dx[3 * i] = -(ga - ua * dot) * inv; dx[3 * i + 1] = -(gb - ub * dot) * inv; dx[3 * i + 2] = -(gc - uc * dot) * inv;
    } else {
        dx[3 * i] = -ga * 1e6f; dx[3 * i + 1] = -gb * 1e6f; dx[3 * i + 2] = -gc * 1e6f;
    }
}

inline dim3 seg_grid(int n_rays) { return dim3(ngp_blocks((int64_t)n_rays * 32, 256)); }

} // namespace

// ------------------------------------------------------------------ live samples of a marched batch
// A ray's samples behind its early-termination point (T <= T_threshold, volumerendering.cu:111) carry zero weight
// and receive zero gradient; the density head has to see them (the stop depends on their predecessors' sigma), the
// colour branch and its backward do not.  Three launches turn (sigma, delta, rays_a) into the ascending list of the
// rows that DO take part, with the very bookkeeping the compositing kernels use (chunk_alpha: same scan, same
// comparison, same bits), so the compositor never reads a colour the list left out.
__global__ void live_count_kernel(const float* __restrict__ sigmas, const float* __restrict__ deltas,
                                  const int64_t* __restrict__ rays_a, float T_thr, int n_rays, int32_t* __restrict__ counts)
{
    Seg sg; int lane;
    if (!seg_load(rays_a, n_rays, sg, lane)) return;
    float T_run = 1.0f;
    int live = sg.n;
    for (int k0 = 0; k0 < sg.n; k0 += 32) {
        const int k = k0 + lane;
        const Chunk c = chunk_alpha(sigmas, deltas, sg.start + k, k < sg.n, T_run, T_thr, lane);
        if (c.first >= 0) { live = k0 + c.first + 1; break; }
        T_run = __shfl(c.T_after, 31, 32);
    }
    if (lane == 0) counts[(blockIdx.x * blockDim.x + threadIdx.x) >> 5] = live;
}

// exclusive scan of the per-ray counts in place (one workgroup, fixed order), total to *n_live
__global__ void __launch_bounds__(1024) live_scan_kernel(int32_t* __restrict__ counts, int n_rays, int32_t* __restrict__ n_live)
{
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n_rays; base += 1024) {
        const int i = base + threadIdx.x;
        const int32_t v = i < n_rays ? counts[i] : 0;
        int32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int32_t before = carry_s;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (i < n_rays) counts[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_live = carry_s;
}

// (also moves up to two (N,3) row blocks — the positions and directions the colour branch starts from — into the
// compacted order: a ray's live rows are one contiguous run on both sides)
__global__ void live_fill_kernel(const int64_t* __restrict__ rays_a, int n_rays, const int32_t* __restrict__ offsets,
                                 const int32_t* __restrict__ n_live_total, int32_t* __restrict__ live_idx,
                                 int32_t* __restrict__ inv_idx, const float* __restrict__ a3, float* __restrict__ a3_c,
                                 const float* __restrict__ b3, float* __restrict__ b3_c)
{
    Seg sg; int lane;
    if (!seg_load(rays_a, n_rays, sg, lane)) return;
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int32_t off = offsets[row];
    const int32_t end = row + 1 < n_rays ? offsets[row + 1] : *n_live_total;
    const int live = end - off;
    for (int k = lane; k < sg.n; k += 32) {
        const int64_t s = sg.start + k;
        if (k < live) { live_idx[off + k] = (int32_t)s; inv_idx[s] = off + k; }
        else inv_idx[s] = -1;
    }
    for (int e = lane; e < 3 * live; e += 32) {
        if (a3) a3_c[3 * (int64_t)off + e] = a3[3 * sg.start + e];
        if (b3) b3_c[3 * (int64_t)off + e] = b3[3 * sg.start + e];
    }
}

// up to three row blocks at once (the colour branch's outputs back in sample order)
__global__ void spread_rows3_kernel(const float* __restrict__ s0, int c0, float* __restrict__ d0,
                                    const float* __restrict__ s1, int c1, float* __restrict__ d1,
                                    const float* __restrict__ s2, int c2, float* __restrict__ d2,
                                    const int32_t* __restrict__ inv, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t j = inv[i];
        for (int c = 0; c < c0; c++) d0[i * c0 + c] = j >= 0 ? s0[j * c0 + c] : 0.0f;
        for (int c = 0; c < c1; c++) d1[i * c1 + c] = j >= 0 ? s1[j * c1 + c] : 0.0f;
        for (int c = 0; c < c2; c++) d2[i * c2 + c] = j >= 0 ? s2[j * c2 + c] : 0.0f;
    }
}

// dst[j, 0:cols] = src[idx[j], 0:cols]
__global__ void gather_rows_kernel(const float* __restrict__ src, int64_t ld_src, int cols, const int32_t* __restrict__ idx,
                                   int64_t n_out, float* __restrict__ dst, int64_t ld_dst)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_out * cols; e += stride) {
        const int64_t j = e / cols;
        const int c = (int)(e - j * cols);
        dst[j * ld_dst + c] = src[(int64_t)idx[j] * ld_src + c];
    }
}

// dst[i, 0:cols] = inv[i] >= 0 ? src[inv[i], 0:cols] : 0
__global__ void spread_rows_kernel(const float* __restrict__ src, int64_t ld_src, int cols, const int32_t* __restrict__ inv,
                                   int64_t n, float* __restrict__ dst, int64_t ld_dst)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * cols; e += stride) {
        const int64_t i = e / cols;
        const int c = (int)(e - i * cols);
        const int32_t j = inv[i];
        dst[i * ld_dst + c] = j >= 0 ? src[(int64_t)j * ld_src + c] : 0.0f;
    }
}

extern "C" {

int ngp_composite_alpha_fw(const float* sigmas, const float* deltas, const int64_t* rays_a, float T_threshold,
                           int n_rays, float* alphas, float* ws, void* stream)
{
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return NGP_OK;
    if (!sigmas || !deltas || !rays_a || !alphas || !ws) return NGP_EINVAL;
    hipLaunchKernelGGL(composite_alpha_fw_kernel, seg_grid(n_rays), dim3(256), 0, (hipStream_t)stream,
                       sigmas, deltas, rays_a, T_threshold, n_rays, alphas, ws);
    return ngp_check_launch();
}

int ngp_composite_train_fw(const float* sigmas, const float* rgbs, const float* normals_pred, const float* sems,
                           const float* deltas, const float* ts, const int64_t* rays_a, float T_threshold,
                           int classes, int n_rays, int64_t* total_samples, float* opacity, float* depth,
                           float* rgb, float* normal_pred, float* sem, float* ws, void* stream)
{
    if (n_rays < 0 || classes < 0) return NGP_EINVAL;
    if (n_rays == 0) return NGP_OK;
    if (!rays_a || !total_samples || !opacity || !depth || !rgb || !normal_pred || (classes && !sem)) return NGP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_FW(CM, CL)                                                                                          \
    hipLaunchKernelGGL(composite_train_fw_kernel<CM>, seg_grid(n_rays), dim3(256), 0, st, sigmas, rgbs,            \
                       normals_pred, sems, deltas, ts, rays_a, T_threshold, CL, n_rays, total_samples, opacity,    \
                       depth, rgb, normal_pred, sem, ws)
    if (classes == 0) LAUNCH_FW(0, 0);
    else if (classes <= 8) LAUNCH_FW(8, classes);
    else if (classes <= 16) LAUNCH_FW(16, classes);
    else if (classes <= 32) LAUNCH_FW(32, classes);
    else {
        LAUNCH_FW(0, 0);
        hipLaunchKernelGGL(composite_sem_fw_kernel, seg_grid(n_rays), dim3(256), 0, st, ws, sems, rays_a, classes,
                           n_rays, sem);
    }
#undef LAUNCH_FW
    return ngp_check_launch();
}

int ngp_composite_train_bw(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb,
                           const float* dL_dnormal_pred, const float* dL_dsem, const float* dL_dws,
                           const float* sigmas, const float* rgbs, const float* normals_pred, const float* ws,
                           const float* deltas, const float* ts, const int64_t* rays_a, const float* opacity,
                           const float* depth, const float* rgb, const float* normal_pred, float T_threshold,
                           int classes, int n_rays, float* dL_dsigmas, float* dL_drgbs, float* dL_dnormals_pred,
                           float* dL_dsems, void* stream)
{
    (void)normals_pred; (void)normal_pred;
    if (n_rays < 0 || classes < 0) return NGP_EINVAL;
    if (n_rays == 0) return NGP_OK;
    if (!rays_a || !dL_dsigmas || !dL_drgbs) return NGP_EINVAL;
    if ((dL_dnormals_pred && !dL_dnormal_pred) || (dL_dsems && classes && !dL_dsem)) return NGP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BW(CM, CL)                                                                                          \
    hipLaunchKernelGGL(composite_train_bw_kernel<CM>, seg_grid(n_rays), dim3(256), 0, st, dL_dopacity, dL_ddepth,  \
                       dL_drgb, dL_dnormal_pred, dL_dsem, dL_dws, sigmas, rgbs, ws, deltas, ts, rays_a, opacity,   \
                       depth, rgb, T_threshold, CL, n_rays, dL_dsigmas, dL_drgbs, dL_dnormals_pred, dL_dsems)
    if (classes == 0) LAUNCH_BW(0, 0);
    else if (classes <= 8) LAUNCH_BW(8, classes);
    else if (classes <= 16) LAUNCH_BW(16, classes);
    else if (classes <= 32) LAUNCH_BW(32, classes);
    else {
        // ws saved by the forward is exactly the effective weight (zero behind the stop)
        LAUNCH_BW(0, 0);
        if (dL_dsems)
            hipLaunchKernelGGL(composite_sem_bw_kernel, seg_grid(n_rays), dim3(256), 0, st, dL_dsem, ws, rays_a, classes,
                               n_rays, dL_dsems);
    }
#undef LAUNCH_BW
    return ngp_check_launch();
}

int ngp_composite_test_fw(const float* sigmas, const float* rgbs, const float* normals, const float* normals_raw,
                          const float* sems, const float* deltas, const float* ts, const float* hits_t,
                          int64_t* alive_indices, float T_threshold, int classes, const int32_t* n_eff_samples,
                          int n_alive, int n_samples, float* opacity, float* depth, float* rgb, float* normal,
                          float* normal_raw, float* sem, void* stream)
{
    (void)hits_t;
    if (n_alive < 0 || classes < 0 || n_samples < 1) return NGP_EINVAL;
    if (n_alive == 0) return NGP_OK;
    if (!sigmas || !rgbs || !normals || !normals_raw || !deltas || !ts || !alive_indices || !n_eff_samples ||
        !opacity || !depth || !rgb || !normal || !normal_raw || (classes && (!sem || !sems))) return NGP_EINVAL;
    hipLaunchKernelGGL(composite_test_fw_kernel, dim3(ngp_blocks(n_alive, 64)), dim3(64), 0, (hipStream_t)stream,
                       sigmas, rgbs, normals, normals_raw, sems, deltas, ts, alive_indices, T_threshold, classes,
                       n_eff_samples, n_alive, n_samples, opacity, depth, rgb, normal, normal_raw, sem,
                       (const int32_t*)nullptr);
    return ngp_check_launch();
}

int ngp_composite_test_fw_rounds(const float* sigmas, const float* rgbs, const float* normals, const float* normals_raw,
                                 const float* sems, const float* deltas, const float* ts, int64_t* alive_indices,
                                 float T_threshold, int classes, const int32_t* n_eff_samples, const int32_t* state,
                                 int n_alive_bound, float* opacity, float* depth, float* rgb, float* normal,
                                 float* normal_raw, float* sem, void* stream)
{
    if (n_alive_bound < 0 || classes < 0) return NGP_EINVAL;
    if (n_alive_bound == 0) return NGP_OK;
    if (!state || !sigmas || !rgbs || !normals || !normals_raw || !deltas || !ts || !alive_indices || !n_eff_samples ||
        !opacity || !depth || !rgb || !normal || !normal_raw || (classes && (!sem || !sems))) return NGP_EINVAL;
    hipLaunchKernelGGL(composite_test_fw_kernel, dim3(ngp_blocks(n_alive_bound, 64)), dim3(64), 0, (hipStream_t)stream,
                       sigmas, rgbs, normals, normals_raw, sems, deltas, ts, alive_indices, T_threshold, classes,
                       n_eff_samples, 0, 0, opacity, depth, rgb, normal, normal_raw, sem, state);
    return ngp_check_launch();
}

int ngp_composite_refloss_fw(const float* sigmas, const float* normals_diff, const float* normals_ori,
                             const float* deltas, const float* ts, const int64_t* rays_a, float T_threshold,
                             int n_rays, float* loss_o, float* loss_p, void* stream)
{
    (void)ts;
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return NGP_OK;
    if (!rays_a || !loss_o || !loss_p) return NGP_EINVAL;
    hipLaunchKernelGGL(refloss_fw_kernel, seg_grid(n_rays), dim3(256), 0, (hipStream_t)stream, sigmas, normals_diff,
                       normals_ori, deltas, rays_a, T_threshold, n_rays, loss_o, loss_p);
    return ngp_check_launch();
}

int ngp_composite_refloss_bw(const float* dL_dloss_o, const float* dL_dloss_p, const float* sigmas,
                             const float* normals_diff, const float* normals_ori, const float* deltas,
                             const float* ts, const int64_t* rays_a, const float* loss_o, const float* loss_p,
                             float T_threshold, int n_rays, float* dL_dsigmas, float* dL_dnormals_diff,
                             float* dL_dnormals_ori, void* stream)
{
    (void)ts;
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return NGP_OK;
    if (!rays_a || !dL_dsigmas || !dL_dnormals_diff || !dL_dnormals_ori) return NGP_EINVAL;
    hipLaunchKernelGGL(refloss_bw_kernel, seg_grid(n_rays), dim3(256), 0, (hipStream_t)stream, dL_dloss_o, dL_dloss_p,
                       sigmas, normals_diff, normals_ori, deltas, rays_a, loss_o, loss_p, T_threshold, n_rays,
                       dL_dsigmas, dL_dnormals_diff, dL_dnormals_ori);
    return ngp_check_launch();
}

int ngp_distortion_loss_fw(const float* ws, const float* deltas, const float* ts, const int64_t* rays_a, int n_rays,
                           float* loss, float* ws_inclusive_scan, float* wts_inclusive_scan, void* stream)
{
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return NGP_OK;
    if (!rays_a || !loss || !ws_inclusive_scan || !wts_inclusive_scan) return NGP_EINVAL;
    hipLaunchKernelGGL(distortion_fw_kernel, seg_grid(n_rays), dim3(256), 0, (hipStream_t)stream, ws, deltas, ts,
                       rays_a, n_rays, loss, ws_inclusive_scan, wts_inclusive_scan);
    return ngp_check_launch();
}

int ngp_distortion_loss_bw(const float* dL_dloss, const float* ws_inclusive_scan, const float* wts_inclusive_scan,
                           const float* ws, const float* deltas, const float* ts, const int64_t* rays_a, int n_rays,
                           float* dL_dws, void* stream)
{
    if (n_rays < 0) return NGP_EINVAL;
    if (n_rays == 0) return NGP_OK;
    if (!rays_a || !dL_dloss || !dL_dws) return NGP_EINVAL;
    hipLaunchKernelGGL(distortion_bw_kernel, seg_grid(n_rays), dim3(256), 0, (hipStream_t)stream, dL_dloss,
                       ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a, n_rays, dL_dws);
    return ngp_check_launch();
}

int ngp_nerf_loss(const float* rgb, const float* target_rgb, const float* opacity, const float* distortion,
                  int n_rays, float lambda_opacity, float lambda_distortion, float* terms, float* d_rgb,
                  float* d_opacity, void* stream)
{
    if (n_rays < 1 || !rgb || !target_rgb || !opacity || !terms || !d_rgb || !d_opacity) return NGP_EINVAL;
    hipLaunchKernelGGL(nerf_loss_kernel, dim3(ngp_blocks(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, rgb,
                       target_rgb, opacity, distortion, n_rays, 1.0f / (3.0f * n_rays), lambda_opacity / n_rays,
                       lambda_distortion / n_rays, terms, d_rgb, d_opacity);
    return ngp_check_launch();
}

int ngp_render_loss_fused(const float* sigmas, const float* rgbs, const float* dsigma_dx, const float* scale3,
                          const float* normal_head, int64_t ld_normal, const float* sem_logits, int64_t ld_sem,
                          const float* dirs, const float* deltas, const float* ts, const int64_t* rays_a,
                          const float* target_rgb, const float* rgb_bg, float T_threshold, int classes, int n_rays, float lambda_opacity,
                          float lambda_distortion, int64_t* total_samples, int64_t* vr_samples, float* opacity,
                          float* depth, float* rgb, float* normal_pred, float* sem, float* ws, float* loss_o,
                          float* loss_p, float* terms, float* dL_dsigmas, float* dL_drgbs, void* stream)
{
    if (n_rays < 0 || classes < 0 || classes > 8 || ld_normal < 3 || ld_sem < classes) return NGP_EINVAL;
    if (n_rays == 0) return NGP_OK;
    if (!rays_a || !target_rgb || !total_samples || !vr_samples || !opacity || !depth || !rgb || !normal_pred ||
        (classes && !sem) || !loss_o || !loss_p || !terms) return NGP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (reinterpret_cast<char*>(vr_samples) == reinterpret_cast<char*>(terms) + 4 * sizeof(float)) {   // adjacent: one fill
        if (hipMemsetAsync(terms, 0, 4 * sizeof(float) + sizeof(int64_t), st) != hipSuccess) return NGP_ELAUNCH;
    } else {
        if (hipMemsetAsync(terms, 0, 4 * sizeof(float), st) != hipSuccess) return NGP_ELAUNCH;
        if (hipMemsetAsync(vr_samples, 0, sizeof(int64_t), st) != hipSuccess) return NGP_ELAUNCH;
    }
    RenderLossArgs a;
    a.sigmas = sigmas; a.rgbs = rgbs; a.dsig_dx = dsigma_dx; a.np_raw = normal_head; a.sem_logits = sem_logits;
    a.dirs = dirs; a.deltas = deltas; a.ts = ts; a.gt = target_rgb; a.scale3 = scale3; a.rays_a = rays_a; a.bg = rgb_bg;
    a.ld_np = ld_normal; a.ld_sem = ld_sem; a.T_thr = T_threshold;
    a.g_rgb = 1.0f / (3.0f * n_rays); a.g_op = lambda_opacity / n_rays; a.g_dist = lambda_distortion / n_rays;
    a.classes = classes; a.n_rays = n_rays; a.total_samples = total_samples; a.vr_samples = vr_samples;
    a.opacity = opacity; a.depth = depth; a.rgb = rgb; a.normal = normal_pred; a.sem = sem; a.ws = ws;
    a.Ro = loss_o; a.Rp = loss_p; a.terms = terms; a.d_sigmas = dL_dsigmas; a.d_rgbs = dL_drgbs;
    // a 32-lane half-wave per ray (W = 64, a whole wave per ray, was measured: 144 us per launch in the step against 85 —
    // 83 VGPRs leave 5 waves per SIMD, so 8192 wave-sized rays no longer fit the chip at once)
    hipLaunchKernelGGL((render_loss_fused_kernel<8, 32>), seg_grid(n_rays), dim3(256), 0, st, a);
    return ngp_check_launch();
}

int ngp_refloss_inputs(const float* normals_raw, const float* normals_pred, const float* dirs, int64_t n,
                       float* normals_diff, float* normals_ori, void* stream)
{
    if (n < 0) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!normals_raw || !normals_pred || !dirs || !normals_diff || !normals_ori) return NGP_EINVAL;
    hipLaunchKernelGGL(refloss_inputs_kernel, dim3(ngp_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, normals_raw,
                       normals_pred, dirs, n, normals_diff, normals_ori);
    return ngp_check_launch();
}

int ngp_neg_normalize(const float* x, int64_t ldx, const float* scale3, int64_t n, float* y, void* stream)
{
    if (n < 0 || ldx < 3) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!x || !y) return NGP_EINVAL;
    hipLaunchKernelGGL(neg_normalize_kernel, dim3(ngp_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       scale3, n, y);
    return ngp_check_launch();
}

int ngp_neg_normalize_bwd(const float* x, int64_t ldx, const float* dL_dy, int64_t n, float* dL_dx, void* stream)
{
    if (n < 0 || ldx < 3) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!x || !dL_dy || !dL_dx) return NGP_EINVAL;
    hipLaunchKernelGGL(neg_normalize_bwd_kernel, dim3(ngp_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       dL_dy, n, dL_dx);
    return ngp_check_launch();
}

int ngp_segment_csr_sum(const float* src, const int64_t* indptr, int n_seg, int width, float* out, void* stream)
{
    if (n_seg < 0 || width < 1) return NGP_EINVAL;
    if (n_seg == 0) return NGP_OK;
    if (!indptr || !out) return NGP_EINVAL;
    hipLaunchKernelGGL(segment_csr_kernel, seg_grid(n_seg), dim3(256), 0, (hipStream_t)stream, src, indptr, n_seg,
                       width, out);
    return ngp_check_launch();
}

int ngp_live_rows(const float* sigmas, const float* deltas, const int64_t* rays_a, float T_threshold, int64_t n_rays,
                  int32_t* offsets, int32_t* live_idx, int32_t* inv_idx, int32_t* n_live, const float* a3, float* a3_c,
                  const float* b3, float* b3_c, void* stream)
{
    if ((a3 && !a3_c) || (b3 && !b3_c)) return NGP_EINVAL;
    if (n_rays < 0 || n_rays > 0x7fffff00) return NGP_EINVAL;
    if (!n_live) return NGP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n_rays == 0) return hipMemsetAsync(n_live, 0, sizeof(int32_t), st) == hipSuccess ? NGP_OK : NGP_ELAUNCH;
    if (!sigmas || !deltas || !rays_a || !offsets || !live_idx || !inv_idx) return NGP_EINVAL;
    hipLaunchKernelGGL(live_count_kernel, seg_grid((int)n_rays), dim3(256), 0, st, sigmas, deltas, rays_a, T_threshold,
                       (int)n_rays, offsets);
    hipLaunchKernelGGL(live_scan_kernel, dim3(1), dim3(1024), 0, st, offsets, (int)n_rays, n_live);
    hipLaunchKernelGGL(live_fill_kernel, seg_grid((int)n_rays), dim3(256), 0, st, rays_a, (int)n_rays, offsets, n_live,
                       live_idx, inv_idx, a3, a3_c, b3, b3_c);
    return ngp_check_launch();
}

int ngp_spread_rows3(const float* src0, int cols0, float* dst0, const float* src1, int cols1, float* dst1,
                     const float* src2, int cols2, float* dst2, const int32_t* inv_idx, int64_t n, void* stream)
{
    if (n < 0 || cols0 < 0 || cols1 < 0 || cols2 < 0) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!inv_idx || (cols0 && !dst0) || (cols1 && !dst1) || (cols2 && !dst2)) return NGP_EINVAL;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(spread_rows3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src0, cols0, dst0,
                       src1, cols1, dst1, src2, cols2, dst2, inv_idx, n);
    return ngp_check_launch();
}

int ngp_gather_rows(const float* src, int64_t ld_src, int cols, const int32_t* idx, int64_t n_out, float* dst,
                    int64_t ld_dst, void* stream)
{
    if (n_out < 0 || cols < 1 || ld_src < cols || ld_dst < cols) return NGP_EINVAL;
    if (n_out == 0) return NGP_OK;
    if (!src || !idx || !dst) return NGP_EINVAL;
    int64_t blocks = (n_out * cols + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, cols,
                       idx, n_out, dst, ld_dst);
    return ngp_check_launch();
}

int ngp_spread_rows(const float* src, int64_t ld_src, int cols, const int32_t* inv_idx, int64_t n, float* dst,
                    int64_t ld_dst, void* stream)
{
    if (n < 0 || cols < 1 || ld_src < cols || ld_dst < cols) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!inv_idx || !dst) return NGP_EINVAL;   // src may be NULL when no row is live (every inv is -1)
    int64_t blocks = (n * cols + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(spread_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, cols,
                       inv_idx, n, dst, ld_dst);
    return ngp_check_launch();
}

} // extern "C"
