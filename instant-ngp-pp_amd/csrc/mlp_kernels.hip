// Small-MLP layers on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains),
// activation backward, fused Adam, sum of squares.
//
// One LDS-tiled GEMM template serves the three products a dense layer needs:
//   FWD   y  (n, n_out) = act(x (n, n_in) . W^T + b)       A = x  [m][k],  B = W [n][k]
//   DGRAD dx (n, n_in)  = dz (n, n_out) . W                A = dz [m][k],  B = W [k][n]
//   WGRAD dW (n_out,n_in) += dz^T . x  (split over samples, fp32 atomics)
//                                                          A = dz [k][m],  B = x [k][n]
// Both operands are staged k-major in LDS (As[k][m], Bs[k][n]) so that an MFMA fragment read
// is one conflict-free ds_read_b32 per lane: lane l of the 32x32x2 MFMA holds A[i=l&31][k=l>>5]
// and B[k=l>>5][j=l&31]; C/D is col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5).
// Layers with n_out <= 4 (the density head, networks.py:57) use VALU kernels instead.
#include "common.h"
#include <stdlib.h>

// Tuning switches for A/B timing exist in the A/B build only (-DNGP_AB_VARIANTS); the product build reads no environment
// variable and keeps no other hidden state.
#ifdef NGP_AB_VARIANTS
static inline bool ab_flag(const char* name) { return getenv(name) != nullptr; }
static inline long ab_long(const char* name, long dflt) { const char* e = getenv(name); const long v = e ? atol(e) : 0; return v > 0 ? v : dflt; }
#else
static inline constexpr bool ab_flag(const char*) { return false; }
static inline constexpr long ab_long(const char*, long dflt) { return dflt; }
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };

struct GemmArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int64_t M, N, K;      // logical GEMM sizes
    const float* bias;    // FWD only (may be null)
    float* z_pre;         // FWD only: pre-activation copy, ld = N (may be null)
    int act;
    int64_t k_chunk;      // WGRAD: K range per blockIdx.z
    int vecA, vecB;       // 16-byte vector loads legal for the operand
    int accumulate;       // DGRAD: C += result
    float* bias_grad;     // WGRAD: db[m] += sum_k dz[k][m] (may be null)
    // XF kernels (DGRAD / WGRAD of the FIRST layer of a 2-layer MLP): A points at the saved hidden
    // activations and the operand dz1 = act1'(hidden) * (dz2 . W2) is formed while the tile is
    // staged, so dz1 never exists in HBM.  dz2 (rows, xf_nout) already carries act2'.
    // F2 kernels (FWD of a 2-layer MLP whose hidden width fits one column tile): the 1..4-wide second
    // layer out = act2(hidden . W2^T + b2) is applied to the activated tile in the epilogue
    const float* f2_W2; int64_t f2_ldw2; const float* f2_b2; float* f2_out; int64_t f2_ldo; int f2_nout, f2_act;
    float* f2_dact;   // optional (n, n_out): act2'(z2) expressed through the output (what ngp_act_bwd gives for a unit gradient)
    const float* xf_dz2; int64_t xf_lddz2;
    const float* xf_W2; int64_t xf_ldw2;
    int xf_nout, xf_act;
    float* xf_dW2; int64_t xf_lddw2; float* xf_db2;   // XF WGRAD: also dW2 += dz2^T . hidden, db2 += colsum(dz2)
#ifdef NGP_AB_VARIANTS
    int exp;   // diagnostic switches of the streaming weight-gradient kernel (NGP_WGRAD_EXP): 1 no hand-over / atomics,
               // 4 no operand transform (dz1 := hidden)
#endif
};

// softplus(v) = log(1+e^v) with the hardware exp/log (v_exp_f32 / v_log_f32, ~1e-6 relative):
// v > 20 -> v (torch's threshold); v < -15 -> e^v (1+e^v would round to 1); else log(1+e^v).
__device__ __forceinline__ float softplus_fast(float v)
{
    // raw v_exp_f32 / v_log_f32 (base 2): __expf / __logf wrap them in denormal-range scaling (compare, select, ldexp: ~8 more
    // vector instructions per element) that a softplus never needs — below v = -87 the result is < 1e-38 either way
    const float e = __builtin_amdgcn_exp2f(v * 1.4426950408889634f);
    float l = 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + e);
    // the logarithm is computed unconditionally and SELECTED: left inside the conditional, hipcc wraps every element of an
    // epilogue tile in its own exec-mask branch (64 s_and_saveexec / s_cbranch / s_or per tile and wave, with their s_nop padding)
    asm volatile("" : "+v"(l));
    const float r = v < -15.0f ? e : l;
    return v > 20.0f ? v : r;
}

__device__ __forceinline__ float act_fwd(float v, int act)
{
    switch (act) {
        case NGP_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case NGP_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        case NGP_ACT_SOFTPLUS: return softplus_fast(v);
        case NGP_ACT_EXP: return __expf(v);
        default: return v;
    }
}

// derivative of an activation expressed through its OUTPUT y, for a unit upstream gradient (bitwise what act_bwd_kernel writes)
__device__ __forceinline__ float act_dout(float y, int act)
{
    switch (act) {
        case NGP_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case NGP_ACT_SIGMOID: return y * (1.0f - y);
        case NGP_ACT_SOFTPLUS: return -expm1f(-y);
        case NGP_ACT_EXP: return y;
        default: return 1.0f;
    }
}

__device__ __forceinline__ float act_grad_from_output(float y, int act)
{
    switch (act) {
        case NGP_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case NGP_ACT_SIGMOID: return y * (1.0f - y);
        case NGP_ACT_SOFTPLUS: return -expm1f(-y);
        case NGP_ACT_EXP: return y;
        default: return 1.0f;
    }
}

// derivative through the OUTPUT for the two hidden activations of the model, cheap enough for the
// staging path of a GEMM (softplus' = 1 - exp(-y), v_exp_f32)
__device__ __forceinline__ float act_grad_fast(float y, int act)
{
    if (act == NGP_ACT_SOFTPLUS) { // 1 - exp(-y) cancels for tiny y: two Taylor terms there (relative error < 2e-7)
        // raw v_exp_f32 (base 2): y >= 0, so exp(-y) never needs __expf's denormal-range scaling (~5 more instructions)
        const float e = __builtin_amdgcn_exp2f(y * -1.4426950408889634f), t = y * (1.0f - 0.5f * y);   // both sides evaluated: a select, not a branch
        return y < 1e-3f ? t : 1.0f - e;
    }
    return act == NGP_ACT_RELU ? (y > 0.0f ? 1.0f : 0.0f) : act_grad_from_output(y, act);
}

constexpr int BK = 32;
constexpr int XF_OMAX = 4;   // widest second layer the fused operand transform handles

// Global -> register fetch and register -> LDS commit are separate so that the fetch of K-tile
// t+1 is in flight while the MFMAs of tile t run (register double buffering; one LDS buffer).

// rows x BK tile of a [row][k]-contiguous matrix; each thread owns float4 pieces along k
template <int ROWS>
struct TileT {
    static constexpr int KQ = BK / 4;                       // float4 pieces per row
    static constexpr int RPP = 256 / KQ;                    // rows per pass
    static constexpr int PASSES = (ROWS + RPP - 1) / RPP;
    float v[PASSES][4];

    __device__ __forceinline__ void fetch(const float* __restrict__ src, int64_t ld, int64_t row0, int64_t rows,
                                          int64_t k0, int64_t kend, bool vec)
    {
        const int t = threadIdx.x;
#pragma unroll
        for (int pass = 0; pass < PASSES; pass++) {
            const int row = t / KQ + pass * RPP;
            const int kq = (t % KQ) * 4;
            const int64_t gr = row0 + row, gk = k0 + kq;
#pragma unroll
            for (int j = 0; j < 4; j++) v[pass][j] = 0.0f;
            if (row < ROWS && gr < rows) {
                const float* p = src + gr * ld + gk;
                if (vec && gk + 3 < kend) {
                    const float4 q = *reinterpret_cast<const float4*>(p);
                    v[pass][0] = q.x; v[pass][1] = q.y; v[pass][2] = q.z; v[pass][3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) if (gk + j < kend) v[pass][j] = p[j];
                }
            }
        }
    }
    // XF: v holds hidden[row][k..k+3]; dz1 = act1'(hidden) * sum_o dz2[row][o] W2[o][k].
    // W2s [OM][128] (W2's columns k of this product) and D2s [ROWS][4] (dz2 of the block's rows) live
    // in LDS, so nothing of the transform stays in registers across the MFMA loop.
    template <int OM>
    __device__ __forceinline__ void xform(int act, const float* __restrict__ W2s, const float* __restrict__ D2s,
                                          int k_local, float (&)[XF_OMAX][4], float (&)[XF_OMAX], bool)
    {
        const int t = threadIdx.x;
        const int kq = k_local + (t % KQ) * 4;
        float w2[OM][4];
#pragma unroll
        for (int o = 0; o < OM; o++) {
            const float4 q = *reinterpret_cast<const float4*>(W2s + o * 128 + kq);
            w2[o][0] = q.x; w2[o][1] = q.y; w2[o][2] = q.z; w2[o][3] = q.w;
        }
#pragma unroll
        for (int pass = 0; pass < PASSES; pass++) {
            const int row = t / KQ + pass * RPP;
            if (row < ROWS) {
                const float4 dq = *reinterpret_cast<const float4*>(D2s + row * 4);
                const float d[4] = {dq.x, dq.y, dq.z, dq.w};
                float sv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int o = 0; o < OM; o++)
#pragma unroll
                    for (int j = 0; j < 4; j++) sv[j] = fmaf(d[o], w2[o][j], sv[j]);
#pragma unroll
                for (int j = 0; j < 4; j++) v[pass][j] = sv[j] * act_grad_fast(v[pass][j], act);
            }
        }
    }
    // dst[k][row], k-major with leading dimension LD
    template <int LD>
    __device__ __forceinline__ void commit(float* __restrict__ dst) const
    {
        const int t = threadIdx.x;
#pragma unroll
        for (int pass = 0; pass < PASSES; pass++) {
            const int row = t / KQ + pass * RPP;
            const int kq = (t % KQ) * 4;
            if (row < ROWS) {
#pragma unroll
                for (int j = 0; j < 4; j++) dst[(kq + j) * LD + row] = v[pass][j];
            }
        }
    }
};

// BK x COLS tile of a [k][col]-contiguous matrix
template <int COLS>
struct TileD {
    static constexpr int TPR = COLS / 4;                    // threads per k-row
    static constexpr int RPP = 256 / TPR;                   // k-rows per pass
    static constexpr int PASSES = (BK + RPP - 1) / RPP;
    float v[PASSES][4];

    __device__ __forceinline__ void fetch(const float* __restrict__ src, int64_t ld, int64_t col0, int64_t cols,
                                          int64_t k0, int64_t kend, bool vec)
    {
        const int t = threadIdx.x;
#pragma unroll
        for (int pass = 0; pass < PASSES; pass++) {
            const int kr = t / TPR + pass * RPP;
            const int c4 = (t % TPR) * 4;
            const int64_t gk = k0 + kr, gc = col0 + c4;
#pragma unroll
            for (int j = 0; j < 4; j++) v[pass][j] = 0.0f;
            if (kr < BK && gk < kend) {
                const float* p = src + gk * ld + gc;
                if (vec && gc + 3 < cols) {
                    const float4 q = *reinterpret_cast<const float4*>(p);
                    v[pass][0] = q.x; v[pass][1] = q.y; v[pass][2] = q.z; v[pass][3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) if (gc + j < cols) v[pass][j] = p[j];
                }
            }
        }
    }
    // XF: v holds hidden[k][col..col+3] (k = sample row).  W2s [OM][128] holds W2's columns of this
    // block's m range, D2s [BK][4] the dz2 rows of the CURRENT K tile (double buffered by the caller).
    // With W2G the second layer's weight gradient is accumulated from the raw tile on the way:
    // gw[o][j] += dz2[k][o] * hidden[k][col+j], gb[o] += dz2[k][o].
    template <int OM>
    __device__ __forceinline__ void xform(int act, const float* __restrict__ W2s, const float* __restrict__ D2s,
                                          int, float (&gw)[XF_OMAX][4], float (&gb)[XF_OMAX], bool w2g)
    {
        const int t = threadIdx.x;
        const int c4 = (t % TPR) * 4;
        float w2[OM][4];
#pragma unroll
        for (int o = 0; o < OM; o++) {
            const float4 q = *reinterpret_cast<const float4*>(W2s + o * 128 + c4);
            w2[o][0] = q.x; w2[o][1] = q.y; w2[o][2] = q.z; w2[o][3] = q.w;
        }
#pragma unroll
        for (int pass = 0; pass < PASSES; pass++) {
            const int kr = t / TPR + pass * RPP;
            if (kr < BK) {
                const float4 dq = *reinterpret_cast<const float4*>(D2s + kr * 4);
                const float d[4] = {dq.x, dq.y, dq.z, dq.w};
                float sv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int o = 0; o < OM; o++) {
                    if (w2g) {
                        gb[o] += d[o];
#pragma unroll
                        for (int j = 0; j < 4; j++) gw[o][j] = fmaf(d[o], v[pass][j], gw[o][j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) sv[j] = fmaf(d[o], w2[o][j], sv[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) v[pass][j] = sv[j] * act_grad_fast(v[pass][j], act);
            }
        }
    }
    template <int LD>
    __device__ __forceinline__ void commit(float* __restrict__ dst) const
    {
        const int t = threadIdx.x;
#pragma unroll
        for (int pass = 0; pass < PASSES; pass++) {
            const int kr = t / TPR + pass * RPP;
            const int c4 = (t % TPR) * 4;
            if (kr < BK) {
#pragma unroll
                for (int j = 0; j < 4; j++) dst[kr * LD + c4 + j] = v[pass][j];
            }
        }
    }
};

template <bool TRANSPOSED, int EXT> struct TileSel { typedef TileT<EXT> type; };
template <int EXT> struct TileSel<false, EXT> { typedef TileD<EXT> type; };

template <int MODE, int WM, int WN, int TM, int TN, int XFW, int F2 = 0>   // XFW / F2 = width of the fused second layer (0: plain); XFW + 16: WGRAD also leaves dW2 / db2
__device__ __forceinline__ void gemm_body(const GemmArgs& p)
{
    constexpr int XF = XFW & 15;
    constexpr bool W2G = (XFW & 16) != 0;
    static_assert(!XF || MODE != MODE_FWD, "the operand transform exists for DGRAD / WGRAD only");
    static_assert(!F2 || MODE == MODE_FWD, "the second-layer epilogue exists for FWD only");
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int LDA = BM + 2, LDB = BN + 2;
    __shared__ float As[BK * LDA];
    __shared__ float Bs[BK * LDB];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BM, n0 = (int64_t)blockIdx.y * BN;
    int64_t kbeg = 0, kend = p.K;
    if (MODE == MODE_WGRAD) {
        kbeg = (int64_t)blockIdx.z * p.k_chunk;
        kend = kbeg + p.k_chunk < p.K ? kbeg + p.k_chunk : p.K;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    // A is [m][k] (transposed staging) except for WGRAD where it is dz [k][m];
    // B is W [n][k] for FWD (transposed staging), W [k][n] / x [k][n] otherwise.
    typename TileSel<MODE != MODE_WGRAD, BM>::type ta;
    typename TileSel<MODE == MODE_FWD, BN>::type tb;
    float bias_acc = 0.0f; // WGRAD: column sums of dz for db (blockIdx.y == 0 only)
    const bool want_db = MODE == MODE_WGRAD && p.bias_grad != nullptr && blockIdx.y == 0;
    const bool want_w2 = W2G && XF != 0 && MODE == MODE_WGRAD && p.xf_dW2 != nullptr && blockIdx.y == 0;
    float gw2[XF_OMAX][4], gb2[XF_OMAX];
#pragma unroll
    for (int o = 0; o < XF_OMAX; o++) {
        gb2[o] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) gw2[o][j] = 0.0f;
    }

    ta.fetch(p.A, p.lda, m0, p.M, kbeg, kend, p.vecA);
    tb.fetch(p.B, p.ldb, n0, p.N, kbeg, kend, p.vecB);

    // XF staging in LDS: W2's columns for this block, and dz2 — for DGRAD the block's rows (fixed),
    // for WGRAD the rows of one K tile at a time, double buffered and prefetched like the tile itself
    __shared__ float W2s[XF ? XF * 128 : 1];
    __shared__ float D2s[XF ? (MODE == MODE_WGRAD ? 2 * BK * 4 : 128 * 4) : 1];
    float d2pre[XF ? XF : 1];
    auto d2_fetch = [&](int64_t k0) {      // WGRAD: dz2 rows k0 .. k0+BK-1, one row per thread
        if (threadIdx.x < BK) {
            const int64_t gk = k0 + threadIdx.x;
#pragma unroll
            for (int o = 0; o < XF; o++)
                d2pre[o] = (gk < kend && o < p.xf_nout) ? p.xf_dz2[gk * p.xf_lddz2 + o] : 0.0f;
        }
    };
    auto d2_commit = [&](int buf) {
        if (threadIdx.x < BK) {
#pragma unroll
            for (int o = 0; o < 4; o++) D2s[(buf * BK + threadIdx.x) * 4 + o] = o < XF ? d2pre[o < XF ? o : 0] : 0.0f;
        }
    };
    if constexpr (XF != 0) {
        for (int idx = threadIdx.x; idx < XF * 128; idx += 256) {
            const int o = idx >> 7, c = idx & 127;
            const int64_t col = (MODE == MODE_WGRAD ? m0 : 0) + c;
            const int64_t lim = MODE == MODE_WGRAD ? p.M : p.K;
            W2s[idx] = (o < p.xf_nout && col < lim) ? p.xf_W2[o * p.xf_ldw2 + col] : 0.0f;
        }
        if (MODE == MODE_WGRAD) {
            d2_fetch(kbeg);
            d2_commit(0);
        } else {
            for (int idx = threadIdx.x; idx < 128 * 4; idx += 256) {
                const int row = idx >> 2, o = idx & 3;
                const int64_t gr = m0 + row;
                D2s[idx] = (row < BM && gr < p.M && o < p.xf_nout && o < XF) ? p.xf_dz2[gr * p.xf_lddz2 + o] : 0.0f;
            }
        }
        __syncthreads();
    }
    int it = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK, it++) {
        if constexpr (XF != 0)   // hidden -> dz1 in registers, on its way to LDS (+ dW2 / db2 partials for WGRAD)
            ta.template xform<XF>(p.xf_act, W2s, D2s + (MODE == MODE_WGRAD ? (it & 1) * BK * 4 : 0), (int)(k0 - kbeg),
                                  gw2, gb2, want_w2);
        ta.template commit<LDA>(As);
        tb.template commit<LDB>(Bs);
        __syncthreads();
        if (k0 + BK < kend) {
            ta.fetch(p.A, p.lda, m0, p.M, k0 + BK, kend, p.vecA);
            if constexpr (XF != 0 && MODE == MODE_WGRAD) d2_fetch(k0 + BK);
            tb.fetch(p.B, p.ldb, n0, p.N, k0 + BK, kend, p.vecB);
        }
        if (want_db && threadIdx.x < BM) {
#pragma unroll
            for (int k = 0; k < BK; k++) bias_acc += As[k * LDA + threadIdx.x];
        }
#pragma unroll
        for (int s = 0; s < BK / 2; s++) {
            float a[TM], b[TN];
#pragma unroll
            for (int tm = 0; tm < TM; tm++) a[tm] = As[(2 * s + lh) * LDA + (wm * TM + tm) * 32 + li];
#pragma unroll
            for (int tn = 0; tn < TN; tn++) b[tn] = Bs[(2 * s + lh) * LDB + (wn * TN + tn) * 32 + li];
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
        if constexpr (XF != 0 && MODE == MODE_WGRAD) { if (k0 + BK < kend) d2_commit((it + 1) & 1); }
        __syncthreads();
    }
    if (want_db && threadIdx.x < BM && m0 + threadIdx.x < p.M) atomicAdd(p.bias_grad + m0 + threadIdx.x, bias_acc);
    if constexpr (W2G && XF != 0 && MODE == MODE_WGRAD) {
        // dW2 / db2 partials: threads with the same column quad (t % TPR) meet in LDS (the staging
        // tiles are free: the K loop ended with a barrier), one atomic per output element and block
        constexpr int TPR = BM / 4;             // threads per k-row of the A tile
        constexpr int GRP = 256 / TPR;          // k-rows per pass = partials per column
        __shared__ float red[(XF ? XF : 1) * (1024 + 32)];   // [GRP][XF][BM] (= 1024 XF floats for every BM) + [GRP][XF]
        if (want_w2) {
            const int t = threadIdx.x, grp = t / TPR, c4 = (t % TPR) * 4;
#pragma unroll
            for (int o = 0; o < XF; o++) {
#pragma unroll
                for (int j = 0; j < 4; j++) red[(grp * XF + o) * BM + c4 + j] = gw2[o][j];
                if (c4 == 0) red[GRP * XF * BM + grp * XF + o] = gb2[o];
            }
        }
        __syncthreads();
        if (want_w2) {
            for (int idx = threadIdx.x; idx < XF * BM; idx += 256) {
                const int o = idx / BM, c = idx - o * BM;
                if (o < p.xf_nout && m0 + c < p.M) {
                    float sum = 0.0f;
#pragma unroll
                    for (int g = 0; g < GRP; g++) sum += red[(g * XF + o) * BM + c];
                    atomicAdd(p.xf_dW2 + o * p.xf_lddw2 + m0 + c, sum);
                }
            }
            if (p.xf_db2 && blockIdx.x == 0 && threadIdx.x < XF && threadIdx.x < p.xf_nout) {
                float sum = 0.0f;
#pragma unroll
                for (int g = 0; g < GRP; g++) sum += red[GRP * XF * BM + g * XF + threadIdx.x];
                atomicAdd(p.xf_db2 + threadIdx.x, sum);
            }
        }
    }

    if constexpr (F2 != 0) {
        // hidden tile: bias + activation + store as usual; second layer: every lane forms the partial
        // dot products of its TN columns for its 16 rows, a transposing butterfly over the 32 lanes
        // of the half-wave (16 shuffles per output instead of 16 x 5) leaves one row total per
        // lane pair, the WN column halves meet in LDS (the staging tiles are free by now).
        float* red = As;   // [WN][BM][F2]
        constexpr int OGW = F2 < 4 ? F2 : 4;            // outputs per pass over the tile
        constexpr int OGN = (F2 + OGW - 1) / OGW;       // passes (8 outputs: two passes of four keep 64 partial sums)
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
#pragma unroll
            for (int og = 0; og < OGN; og++) {
                float part[OGW][16];
#pragma unroll
                for (int o = 0; o < OGW; o++)
#pragma unroll
                    for (int r = 0; r < 16; r++) part[o][r] = 0.0f;
#pragma unroll
                for (int tn = 0; tn < TN; tn++) {
                    const int64_t n = n0 + (wn * TN + tn) * 32 + li;
                    const bool ncol = n < p.N;
                    const float bv = (ncol && p.bias) ? p.bias[n] : 0.0f;
                    float w2[OGW];
#pragma unroll
                    for (int o = 0; o < OGW; o++)
                        w2[o] = (ncol && og * OGW + o < p.f2_nout) ? p.f2_W2[(og * OGW + o) * p.f2_ldw2 + n] : 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int64_t m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float v = act_fwd(acc[tm][tn][r] + bv, p.act);
                        if (og == 0 && ncol && m < p.M) p.C[m * p.ldc + n] = v;
#pragma unroll
                        for (int o = 0; o < OGW; o++) part[o][r] = fmaf(v, w2[o], part[o][r]);
                    }
                }
#pragma unroll
                for (int o = 0; o < OGW; o++) {
                    float* v = part[o];
#pragma unroll
                    for (int half = 8; half >= 1; half >>= 1) {
                        const int mask = half * 2;   // 16, 8, 4, 2
                        const bool up = (li & mask) != 0;
#pragma unroll
                        for (int j = 0; j < half; j++) {
                            float lo = v[j], hi = v[j + half];
                            if constexpr (OGW == 1) asm volatile("" : "+v"(lo), "+v"(hi));   // plain selects (see mlp_stream_fwd_kernel's butterfly)
                            const float keep = up ? hi : lo;
                            const float send = up ? lo : hi;
                            v[j] = keep + __shfl_xor(send, mask, 64);
                        }
                    }
                    const float tot = v[0] + __shfl_xor(v[0], 1, 64);
                    const int rr = ((li >> 4) & 1) * 8 + ((li >> 3) & 1) * 4 + ((li >> 2) & 1) * 2 + ((li >> 1) & 1);
                    const int row = (wm * TM + tm) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                    if ((li & 1) == 0) red[(wn * BM + row) * F2 + og * OGW + o] = tot;
                }
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < BM * F2; idx += 256) {
            const int row = idx / F2, o = idx - row * F2;
            const int64_t m = m0 + row;
            if (o < p.f2_nout && m < p.M) {
                float sum = p.f2_b2 ? p.f2_b2[o] : 0.0f;
#pragma unroll
                for (int w = 0; w < WN; w++) sum += red[(w * BM + row) * F2 + o];
                const float yv = act_fwd(sum, p.f2_act);
                p.f2_out[m * p.f2_ldo + o] = yv;
                if (p.f2_dact) p.f2_dact[m * p.f2_ldo + o] = act_dout(yv, p.f2_act);
            }
        }
        return;
    }

#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
            const int64_t n = n0 + (wn * TN + tn) * 32 + li;
            if (n >= p.N) continue;
            const float bv = (MODE == MODE_FWD && p.bias) ? p.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                float v = acc[tm][tn][r];
                if (MODE == MODE_FWD) {
                    v += bv;
                    if (p.z_pre) p.z_pre[m * p.N + n] = v;
                    p.C[m * p.ldc + n] = act_fwd(v, p.act);
                } else if (MODE == MODE_DGRAD) {
                    if (p.accumulate) v += p.C[m * p.ldc + n];
                    p.C[m * p.ldc + n] = v;
                } else {
                    atomicAdd(p.C + m * p.ldc + n, v);
                }
            }
        }
}

template <int MODE, int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) gemm_kernel(GemmArgs p)
{
    gemm_body<MODE, WM, WN, TM, TN, 0>(p);
}

// 2-layer forward: hidden = act1(x W1^T + b1) stored, out = act2(hidden W2^T + b2) from the epilogue
template <int WM, int WN, int TM, int TN, int F2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) gemm_fwd2_kernel(GemmArgs p)
{
    gemm_body<MODE_FWD, WM, WN, TM, TN, 0, F2>(p);
}

// operand-transform variant: the extra staging registers must not cost the third workgroup per CU
template <int MODE, int WM, int WN, int TM, int TN, int XF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) gemm_xf_kernel(GemmArgs p)
{
    gemm_body<MODE, WM, WN, TM, TN, XF>(p);
}

// ---------------------------------------------------------------- streaming 2-layer forward (H = 32 TN)
// The weight matrix of a 128-wide layer is 64-80 KB: it fits the LDS of a CU next to little else,
// and then nothing but the sample rows has to move.  One 512-thread workgroup per CU keeps W1 as
// [n][k] (k contiguous, row stride K+4 floats: a lane's 16-byte slot is (n (K/4+1) + const) mod 16 with
// K/4+1 odd, a permutation over each 16-lane group of ds_read_b128 — conflict-free), W2 and the biases in
// LDS for its whole life; each of its 8 waves streams 32-row tiles on its own:
//   A operand  : lane (row li, half lh) loads float4 x[row][8q+4lh .. +3] straight from global memory
//                into registers — the reduction index may be permuted freely as long as A and B agree,
//                so the 4 floats feed 4 consecutive 32x32x2 MFMA steps; no LDS staging, no transposition;
//   B operand  : lane (column li, half lh) reads float4 W1[n][8q+4lh .. +3] from LDS, one K step ahead;
//   the KQ float4 of the NEXT tile replace the current ones one by one as they are consumed (rolling
//   prefetch: a full tile of MFMA time, ~7 us, covers the load latency);
// so the K loop has no barrier and no LDS write at all.  Epilogue: the accumulators start at b1, the
// activated tile leaves through a per-wave [32][32] LDS transposition as 16-byte stores, the second
// layer is per-lane partial products and a transposing butterfly over the 32 lanes (as in
// gemm_fwd2_kernel, without the cross-wave combine: a wave owns all H columns of its rows).
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The row prefetch is written as inline assembly with hand-placed s_waitcnt: vmcnt counts loads AND
// stores in issue order, and the compiler's own bookkeeping answers a load that is older than the
// stores of the previous tile's epilogue with a full drain at every tile start.  The piece loaded at
// step q of tile t is read at step q of tile t+1 with exactly KQ-1 loads and all epilogue stores
// (4 TN hidden stores + the outputs for a full tile) issued after it, so vmcnt(STREAM_VMCNT) — "all but the STREAM_VMCNT youngest
// are done" — is always enough and never waits for more than a few of the oldest stores.
template <int OFF>
__device__ __forceinline__ void stream_load(f32x4& dst, const float* ptr)
{
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void stream_wait(f32x4& v)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}

template <int F2, int KQ, int ACT1, int TN>   // KQ = K / 8: float4 pieces per lane and tile; TN = H / 32 column blocks
__global__ void __launch_bounds__(512) mlp_stream_fwd_kernel(GemmArgs p, int n_tiles)
{
    extern __shared__ float smem[];
    constexpr int K = KQ * 8;
    constexpr int LDW = K + 4;
    constexpr int STREAM_VMCNT = KQ - 1 + 4 * TN - 4;   // a full tile leaves 4 TN hidden stores + the outputs behind
    constexpr int H = 32 * TN;
    float* Ws = smem;                       // [H][LDW]
    float* W2s = Ws + H * LDW;              // [F2][H]
    float* b1s = W2s + F2 * H;              // [H]
    float* b2s = b1s + H;                   // [F2]  (a global load in the epilogue would drain every older
                                            //        load and store of the wave: vmcnt counts in order)
    float* stage = b2s + 8;                 // [waves][32][32]
    const int nthr = blockDim.x, nw = nthr >> 6;   // 8 waves (one workgroup fills a CU's registers) or 4
    for (int idx = threadIdx.x; idx < H * (K / 4); idx += nthr) {
        const int n = idx / (K / 4), k4 = (idx - n * (K / 4)) * 4;
        const float4 q = *reinterpret_cast<const float4*>(p.B + (int64_t)n * p.ldb + k4);
        *reinterpret_cast<float4*>(Ws + n * LDW + k4) = q;
    }
    for (int idx = threadIdx.x; idx < F2 * H; idx += nthr) {
        const int o = idx / H, n = idx - o * H;
        W2s[idx] = o < p.f2_nout ? p.f2_W2[o * p.f2_ldw2 + n] : 0.0f;
    }
    if (threadIdx.x < H) b1s[threadIdx.x] = p.bias ? p.bias[threadIdx.x] : 0.0f;
    if (threadIdx.x < F2) b2s[threadIdx.x] = (p.f2_b2 && threadIdx.x < p.f2_nout) ? p.f2_b2[threadIdx.x] : 0.0f;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int stride = gridDim.x * nw;
    int tile = blockIdx.x * nw + wave;
    if (tile >= n_tiles) return;
    f32x4 a[KQ];
    auto row_ptr = [&](int t) {
        int64_t r = (int64_t)t * 32 + li;
        if (r >= p.M) r = p.M - 1;
        return p.A + r * p.lda + 4 * lh;
    };
    {
        const float* src = row_ptr(tile);
#pragma unroll
        for (int q = 0; q < KQ; q++) a[q] = *reinterpret_cast<const f32x4*>(src + 8 * q);   // compiler-tracked: waited for below
    }
    const int f2_act = p.f2_act, nout = p.f2_nout;
    for (; tile < n_tiles; tile += stride) {
        const int next = tile + stride;
        const float* nsrc = row_ptr(next < n_tiles ? next : tile);
        f32x16 acc[TN];
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {   // the accumulators start at the bias: f32 MFMAs and VALU work share the
            const float bv = b1s[tn * 32 + li];   // SIMD's multipliers, so every VALU instruction saved is MFMA time
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tn][r] = bv;
        }
        // B operands one K step ahead of the MFMAs that use them; the scheduling barriers keep the
        // compiler from hoisting all 64+ LDS reads of the tile to the top (256 registers, spills)
        float4 bcur[TN], bnxt[TN];
#pragma unroll
        for (int tn = 0; tn < TN; tn++) bcur[tn] = *reinterpret_cast<const float4*>(Ws + (tn * 32 + li) * LDW + 4 * lh);
#pragma unroll
        for (int q = 0; q < KQ; q++) {
            stream_wait<STREAM_VMCNT>(a[q]);
            const f32x4 av = a[q];
            if (q + 1 < KQ) {
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
                    bnxt[tn] = *reinterpret_cast<const float4*>(Ws + (tn * 32 + li) * LDW + 8 * (q + 1) + 4 * lh);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int tn = 0; tn < TN; tn++) {
                    const float bf[4] = {bcur[tn].x, bcur[tn].y, bcur[tn].z, bcur[tn].w};
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bf[j], acc[tn], 0, 0, 0);
                }
            }
            switch (q) {   // the immediate offset must be a literal
#define STREAM_CASE(Q) case Q: stream_load<32 * Q>(a[Q < KQ ? Q : 0], nsrc); break;
                STREAM_CASE(0) STREAM_CASE(1) STREAM_CASE(2) STREAM_CASE(3) STREAM_CASE(4) STREAM_CASE(5) STREAM_CASE(6)
                STREAM_CASE(7) STREAM_CASE(8) STREAM_CASE(9) STREAM_CASE(10) STREAM_CASE(11) STREAM_CASE(12) STREAM_CASE(13)
                STREAM_CASE(14) STREAM_CASE(15) STREAM_CASE(16) STREAM_CASE(17) STREAM_CASE(18) STREAM_CASE(19)
#undef STREAM_CASE
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);    // this step: the 4 LDS reads of the next step first,
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);   // then the 16 MFMAs (a whole step covers the LDS latency)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tn = 0; tn < TN; tn++) bcur[tn] = bnxt[tn];
        }
        // ---- epilogue
        const int64_t m0 = (int64_t)tile * 32;
        float* stg = stage + wave * 1024;       // this wave's [32][32] transposition tile
        constexpr int OGW = F2 < 4 ? F2 : 4;
        constexpr int OGN = (F2 + OGW - 1) / OGW;
#pragma unroll
        for (int og = 0; og < OGN; og++) {
            float part[OGW][16];
#pragma unroll
            for (int o = 0; o < OGW; o++)
#pragma unroll
                for (int r = 0; r < 16; r++) part[o][r] = 0.0f;
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                const int n = tn * 32 + li;
                float w2[OGW];
#pragma unroll
                for (int o = 0; o < OGW; o++) w2[o] = W2s[(og * OGW + o) * H + n];
                if (og == 0) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const float z = acc[tn][r];
                        acc[tn][r] = ACT1 == NGP_ACT_RELU ? fmaxf(z, 0.0f) : (ACT1 == NGP_ACT_SOFTPLUS ? softplus_fast(z) : z);
                    }
                    // hidden store: 64 dword stores per lane and tile would overrun the 64 vector-memory
                    // operations a wave may have in flight; through LDS the 32x32 block leaves as 4 x 16 bytes
                    // per lane (rows of 128 contiguous bytes).  Both access patterns are conflict-free on an
                    // unpadded [32][32] tile; LDS operations of one wave execute in order, so no barrier.
#pragma unroll
                    for (int r = 0; r < 16; r++) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[tn][r];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int f = lane + 64 * i, row = f >> 3, c4 = (f & 7) * 4;
                        const float4 v = *reinterpret_cast<const float4*>(stg + row * 32 + c4);
                        if (m0 + row < p.M) *reinterpret_cast<float4*>(p.C + (m0 + row) * p.ldc + tn * 32 + c4) = v;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; r++)
#pragma unroll
                    for (int o = 0; o < OGW; o++) part[o][r] = fmaf(acc[tn][r], w2[o], part[o][r]);
            }
#pragma unroll
            for (int o = 0; o < OGW; o++) {
                float* v = part[o];
#pragma unroll
                for (int half = 8; half >= 1; half >>= 1) {
                    const int mask = half * 2;   // 16, 8, 4, 2
                    const bool up = (li & mask) != 0;
#pragma unroll
                    for (int j = 0; j < half; j++) {
                        float lo = v[j], hi = v[j + half];
                        // two plain selects: without the barrier hipcc turns `up ? v[j + half] : v[j]` into an extract with a
                        // lane-dependent index, i.e. a 15-compare select chain per read (930 vector instructions per tile)
                        // (one output only: with 4 the compiler emits the plain selects by itself and the barrier costs 5 %)
                        if constexpr (OGW == 1) asm volatile("" : "+v"(lo), "+v"(hi));
                        const float keep = up ? hi : lo;
                        const float send = up ? lo : hi;
                        v[j] = keep + __shfl_xor(send, mask, 64);
                    }
                }
                const float tot = v[0] + __shfl_xor(v[0], 1, 64);
                const int rr = ((li >> 4) & 1) * 8 + ((li >> 3) & 1) * 4 + ((li >> 2) & 1) * 2 + ((li >> 1) & 1);
                const int64_t m = m0 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                const int oo = og * OGW + o;
                if ((li & 1) == 0 && oo < nout && m < p.M) {
                    const float yv = act_fwd(tot + b2s[oo], f2_act);
                    p.f2_out[m * p.f2_ldo + oo] = yv;
                    if (p.f2_dact) p.f2_dact[m * p.f2_ldo + oo] = act_dout(yv, f2_act);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- streaming data gradient (H = n_in = 128)
// dx = dz1 . W1 with dz1 = act1'(hidden) * (dz2 . W2) formed in registers: the same organisation as
// mlp_stream_fwd_kernel.  W1 sits in LDS transposed ([input column][hidden unit], the reduction index
// contiguous), a lane loads 16-byte pieces of its row of `hidden` and turns them into dz1 on the spot
// (W2's pieces are LDS broadcasts, the row's dz2 values travel with the prefetch), the tile leaves
// through the per-wave LDS transposition as 16-byte stores.  No epilogue arithmetic at all.
template <int OFF>
__device__ __forceinline__ void stream_load1(float& dst, const float* ptr)
{
    asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void stream_wait1(float& v)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}


template <int XF, int ACT1>
__global__ void __launch_bounds__(512) mlp_stream_dgrad_kernel(GemmArgs p, int n_tiles)
{
    extern __shared__ float smem[];
    constexpr int KQ = 16, LDW = 132;
    constexpr int STREAM_VMCNT = KQ - 1 + 12;
    float* Wt = smem;                       // [128 input columns][LDW]: Wt[j][n] = W1[n][j]
    float* W2s = Wt + 128 * LDW;            // [XF][128]
    float* stage = W2s + XF * 128;          // [waves][32][32]
    const int nthr = blockDim.x, nw = nthr >> 6;
    for (int idx = threadIdx.x; idx < 128 * 128; idx += nthr) {
        const int n = idx >> 7, j = idx & 127;
        Wt[j * LDW + n] = p.B[(int64_t)n * p.ldb + j];
    }
    for (int idx = threadIdx.x; idx < XF * 128; idx += nthr) {
        const int o = idx >> 7, n = idx & 127;
        W2s[idx] = o < p.xf_nout ? p.xf_W2[o * p.xf_ldw2 + n] : 0.0f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int stride = gridDim.x * nw;
    int tile = blockIdx.x * nw + wave;
    if (tile >= n_tiles) return;
    f32x4 a[KQ];
    float d2[XF];
    auto row_of = [&](int t) {
        int64_t r = (int64_t)t * 32 + li;
        return r < p.M ? r : p.M - 1;
    };
    {
        const int64_t r = row_of(tile);
        const float* src = p.A + r * p.lda + 4 * lh;
#pragma unroll
        for (int q = 0; q < KQ; q++) a[q] = *reinterpret_cast<const f32x4*>(src + 8 * q);
#pragma unroll
        for (int o = 0; o < XF; o++) d2[o] = o < p.xf_nout ? p.xf_dz2[r * p.xf_lddz2 + o] : 0.0f;
    }
    const int nout = p.xf_nout;
    float* stg = stage + wave * 1024;
    for (; tile < n_tiles; tile += stride) {
        const int next = tile + stride;
        const int64_t nr = row_of(next < n_tiles ? next : tile);
        const float* nsrc = p.A + nr * p.lda + 4 * lh;
        const float* ndz2 = p.xf_dz2 + nr * p.xf_lddz2;
        f32x16 acc[4];
#pragma unroll
        for (int tn = 0; tn < 4; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tn][r] = 0.0f;
        float dcur[XF];
#pragma unroll
        for (int o = 0; o < XF; o++) {
            stream_wait1<STREAM_VMCNT>(d2[o]);
            dcur[o] = d2[o];
        }
        float4 bcur[4], bnxt[4];
#pragma unroll
        for (int tn = 0; tn < 4; tn++) bcur[tn] = *reinterpret_cast<const float4*>(Wt + (tn * 32 + li) * LDW + 4 * lh);
#pragma unroll
        for (int q = 0; q < KQ; q++) {
            stream_wait<STREAM_VMCNT>(a[q]);
            const f32x4 hv = a[q];
            if (q + 1 < KQ) {
#pragma unroll
                for (int tn = 0; tn < 4; tn++)
                    bnxt[tn] = *reinterpret_cast<const float4*>(Wt + (tn * 32 + li) * LDW + 8 * (q + 1) + 4 * lh);
            }
            // dz1 for this lane's 4 hidden units
            float sv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int o = 0; o < XF; o++) {
                const float4 w = *reinterpret_cast<const float4*>(W2s + o * 128 + 8 * q + 4 * lh);
                sv[0] = fmaf(dcur[o], w.x, sv[0]); sv[1] = fmaf(dcur[o], w.y, sv[1]);
                sv[2] = fmaf(dcur[o], w.z, sv[2]); sv[3] = fmaf(dcur[o], w.w, sv[3]);
            }
            float av[4];
#pragma unroll
            for (int j = 0; j < 4; j++) av[j] = sv[j] * act_grad_fast(hv[j], ACT1);
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int tn = 0; tn < 4; tn++) {
                    const float bf[4] = {bcur[tn].x, bcur[tn].y, bcur[tn].z, bcur[tn].w};
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bf[j], acc[tn], 0, 0, 0);
                }
            }
            switch (q) {   // the immediate offset must be a literal
#define STREAM_CASE(Q) case Q: stream_load<32 * Q>(a[Q], nsrc); break;
                STREAM_CASE(0) STREAM_CASE(1) STREAM_CASE(2) STREAM_CASE(3) STREAM_CASE(4) STREAM_CASE(5) STREAM_CASE(6)
                STREAM_CASE(7) STREAM_CASE(8) STREAM_CASE(9) STREAM_CASE(10) STREAM_CASE(11) STREAM_CASE(12) STREAM_CASE(13)
                STREAM_CASE(14) STREAM_CASE(15)
#undef STREAM_CASE
            }
            if (q == 0) {   // the next tile's dz2 values ride behind its first piece
                if (XF > 0) stream_load1<0>(d2[0], ndz2);
                if (XF > 1) stream_load1<4>(d2[XF > 1 ? 1 : 0], ndz2);
                if (XF > 2) stream_load1<8>(d2[XF > 2 ? 2 : 0], ndz2);
                if (XF > 3) stream_load1<12>(d2[XF > 3 ? 3 : 0], ndz2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tn = 0; tn < 4; tn++) bcur[tn] = bnxt[tn];
        }
        // ---- store the tile through the LDS transposition (see mlp_stream_fwd_kernel)
        const int64_t m0 = (int64_t)tile * 32;
#pragma unroll
        for (int tn = 0; tn < 4; tn++) {
#pragma unroll
            for (int r = 0; r < 16; r++) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[tn][r];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int f = lane + 64 * i, row = f >> 3, c4 = (f & 7) * 4;
                const float4 v = *reinterpret_cast<const float4*>(stg + row * 32 + c4);
                if (m0 + row < p.M) *reinterpret_cast<float4*>(p.C + (m0 + row) * p.ldc + tn * 32 + c4) = v;
            }
        }
    }
    (void)nout;
}

// ---------------------------------------------------------------- streaming weight gradient (H = 128)
// dW1 (128 x n_in) += dz1^T . x over a chunk of samples per workgroup, dz1 = act1'(hidden) * (dz2 . W2).
// Both operands are [sample][column] matrices and the reduction runs over samples, so the MFMA's
// register layout IS the memory layout: for the step that covers samples s and s+1, lane (li, lh)
// needs hidden[s+lh][h0+li] and x[s+lh][32 tn + li] — plain dword loads, two 128-byte row segments
// per instruction, no LDS, no barrier, no transposition.  A wave owns 32 hidden units and ALL input
// columns (TN = 4 or 5 accumulator blocks), so every dz1 element is formed exactly once per
// workgroup; W2's column for the lane's hidden unit stays in registers.  The x loads run D steps ahead
// in a register ring; hidden, dz2 and the remainder columns come in once per 16 samples as 16-byte pieces
// and reach the lanes through a wave-private LDS tile (see "Operand traffic" below).  Samples past the end
// of the chunk are read from a clamped row with dz2 = 0, which zeroes dz1 and with it every contribution.
// dW2 / db2 / db1 ride along as per-lane sums.
template <int XF, int ACT1, int TN, bool W2G>
__global__ void __launch_bounds__(512) mlp_stream_wgrad_kernel(GemmArgs p, int64_t chunk)
{
    // 8 waves: two groups of 4 (one per 32 hidden units) split the workgroup's chunk of samples in halves and
    // meet in LDS before the atomics — the 128 x n_in atomic adds per workgroup are a fixed cost (~55 us
    // per launch with two 4-wave workgroups per CU), one 8-wave workgroup per CU halves it at the same occupancy
    extern __shared__ float lds[];
#ifndef NGP_WGRAD_D
#define NGP_WGRAD_D 8
#endif
    constexpr int D = NGP_WGRAD_D;     // steps (of two samples) the operand loads run ahead of the MFMAs
    // (wave and group numbers as scalars: the chunk bounds, the loop counter and the base pointers then live in SGPRs)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) & 3)),
              grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int li = lane & 31, lh = lane >> 5;
    const bool two = blockDim.x == 512;                  // launched with 4 waves instead: one group, no hand-over
    const int64_t half = two ? chunk / 2 : chunk;        // chunk is a multiple of 32
    const int64_t cbeg = (int64_t)blockIdx.x * chunk;    // the workgroup's chunk is never empty (launch geometry)
    const int64_t cend = cbeg + chunk < p.K ? cbeg + chunk : p.K;
    const int64_t kbeg = cbeg + grp * half < cend ? cbeg + grp * half : cend;
    const int64_t kend = grp == 0 ? (cbeg + half < cend ? cbeg + half : cend) : cend;
    const int h = wave * 32 + li;
    const int N = (int)p.N;
    float w2[XF];
#pragma unroll
    for (int o = 0; o < XF; o++) w2[o] = p.xf_W2[o * p.xf_ldw2 + h];
    // input columns of the lane: 4 li .. 4 li + 3 for the first four accumulator blocks (ONE 16-byte load per
    // sample instead of four dwords; which column a lane's accumulator stands for is free to choose),
    // 128 + li for the fifth (columns past n_in read zeros and feed accumulator columns that are never stored)
    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tn][r] = 0.0f;
    float gw[XF], gb2[XF], gb1 = 0.0f;
#pragma unroll
    for (int o = 0; o < XF; o++) { gw[o] = 0.0f; gb2[o] = 0.0f; }

    // Operand traffic.  A wave-instruction costs the CU's address unit about the same whether it moves 256 bytes or
    // 1 KB, and with 8 waves per CU three to six small loads per step (a dword of `hidden` = 2 x 128 bytes, XF
    // broadcast dwords of dz2, a dword of the remainder columns) bound this loop, not the MFMAs (measured: 0.325 ->
    // 0.179 ms with the loop's loads compiled out, nothing from dropping the transform or the hand-over).  So only x
    // keeps a load per step (16 bytes per lane, 1 KB per instruction); `hidden`, the remainder columns and dz2 are
    // fetched once per TURN of 16 samples as 16-byte pieces (one or two instructions each), parked in registers
    // for a turn, then laid down in a wave-private LDS tile from which each step picks its operands with
    // ds_read_b32 (LDS operations of one wave execute in order: no barrier).
    static_assert(D == 8, "a turn of the staging tile is 16 samples = 8 steps");
    float* stg = lds + (two ? (threadIdx.x >> 6) * 1152 : wave * 2048);
    float* sH = stg;             // [16][32]  hidden[s][h0 + c]
    float* sX5 = stg + 512;      // [16][32]  x[s][128 + c]
    float* sD = stg + 1024;      // [16][4]   dz2[s][o] (0 past the end of the chunk), then 64 dummy words
    const int hs = lane >> 3, hp = lane & 7;   // hidden piece: sample (two half-turns of 8), 16-byte piece of the 32 units
    const int xs = lane >> 2, xp = lane & 3;   // remainder piece: sample, 16-byte piece of 16 columns
    const int x5_second = (TN > 4 && N > 144) ? 16 : 0;   // 32 remainder columns (n_in = 160) or 16 (the piece twice)
    // Prefetch by whole turns.  All loads of turn t+1 (8 x pieces, the hidden / remainder / dz2 pieces) are issued
    // at the top of turn t and nothing of them is touched before the top of turn t+1, 8 steps (~4,000 cycles of
    // MFMA work of the SIMD's two waves) later — so the one wait per turn the compiler places there (vmcnt(0): every
    // load it waits for is a turn old) costs next to nothing, and no load is in flight across the loop's back-edge.
    // A ROLLING ring (x piece of step d reloaded at step d, as this kernel used to do) looks cheaper in registers
    // but cannot be expressed: the compiler's wait insertion merges the loop's two entry states and drains the whole
    // ring at the first step of every turn, and inline-assembly loads with hand-placed counts are not safe either
    // (the register allocator renames the ring across the back-edge and copies registers whose loads have not
    // landed).  The loop body is one basic block (softplus' is a select, every lane takes part in every load and
    // store); scheduling barriers keep the loads at the top of the turn and each step's MFMAs in their step.
    f32x4 th0, th1, tx0, tx1;
    float td;
    const bool d_lane = lane < 16 * XF;
    const int d_s = d_lane ? lane / XF : 0, d_o = d_lane ? lane % XF : 0;
    float* d_slot = sD + (d_lane ? d_s * 4 + d_o : 64 + lane);   // lanes without a dz2 element store to a dummy word
    f32x4 xn[D], xc[D];
    // Addresses.  Every load of a turn reads row (t0 + k) clamped to the group's last row.  Written with 64-bit rows
    // (clamp, row * ld, base + 4 * ...) that is ~10 vector instructions per load, three of them quarter-rate integer
    // multiplies: ~110 of the loop's ~250 vector instructions per turn beside 32 MFMAs.  Instead: scalar base pointers at
    // the group's first row, 32-bit BYTE offsets relative to them (the launcher keeps chunk * ld * 4 below 2^31), a
    // per-lane constant part, a scalar part that advances by 16 rows per turn, and the clamp as ONE v_min_u32 against
    // the lane's offset in the last row (offsets grow with the row for a fixed column).
    const int64_t row0 = kbeg < kend ? kbeg : kend - 1;                  // (an empty group reads row kend - 1 >= 0, never uses it)
    const uint32_t rmax = (uint32_t)(kend - 1 - row0);                   // last row, relative
    const char* const Bb = reinterpret_cast<const char*>(p.B + row0 * p.ldb);
    const char* const Ab = reinterpret_cast<const char*>(p.A + row0 * p.lda + wave * 32);
    const char* const Db = reinterpret_cast<const char*>(p.xf_dz2 + row0 * p.xf_lddz2);
    const uint32_t ldb4 = (uint32_t)p.ldb * 4u, lda4 = (uint32_t)p.lda * 4u, ldd4 = (uint32_t)p.xf_lddz2 * 4u;
    const uint32_t xo0 = (uint32_t)lh * ldb4 + 16u * li, xmax = rmax * ldb4 + 16u * li;
    const uint32_t ho0 = (uint32_t)hs * lda4 + 16u * hp, hmax = rmax * lda4 + 16u * hp;
    const uint32_t x5o = (uint32_t)xs * ldb4 + 512u + 16u * xp, x5max = rmax * ldb4 + 512u + 16u * xp;
    const uint32_t dof = (uint32_t)d_s * ldd4 + 4u * d_o, dmax = rmax * ldd4 + 4u * d_o;
    auto umin = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
    auto issue_turn = [&](int64_t t0) {
        const uint32_t rel = (uint32_t)(t0 - row0);                      // scalar: rows in front of this turn
        const uint32_t tx = rel * ldb4, th = rel * lda4, tdz = rel * ldd4;
#pragma unroll
        for (int d = 0; d < D; d++)
            xn[d] = *reinterpret_cast<const f32x4*>(Bb + umin(tx + (uint32_t)(2 * d) * ldb4 + xo0, xmax));
        th0 = *reinterpret_cast<const f32x4*>(Ab + umin(th + ho0, hmax));
        th1 = *reinterpret_cast<const f32x4*>(Ab + umin(th + 8u * lda4 + ho0, hmax));
        if (TN > 4) {
            const char* xr = Bb + umin(tx + x5o, x5max);
            tx0 = *reinterpret_cast<const f32x4*>(xr);
            tx1 = *reinterpret_cast<const f32x4*>(xr + 4 * x5_second);   // n_in = 144: the same piece again
        }
        td = *reinterpret_cast<const float*>(Db + umin(tdz + dof, dmax));
    };
    auto commit_turn = [&](int64_t t0) {
#pragma unroll
        for (int d = 0; d < D; d++) xc[d] = xn[d];
        *reinterpret_cast<f32x4*>(sH + hs * 32 + 4 * hp) = th0;
        *reinterpret_cast<f32x4*>(sH + (8 + hs) * 32 + 4 * hp) = th1;
        if (TN > 4) {
            *reinterpret_cast<f32x4*>(sX5 + xs * 32 + 4 * xp) = tx0;
            *reinterpret_cast<f32x4*>(sX5 + xs * 32 + 16 + 4 * xp) = tx1;   // n_in = 144: columns that are never stored
        }
        *d_slot = (t0 + d_s < kend) ? td : 0.0f;
    };
#ifdef NGP_AB_VARIANTS
    const int EXP = p.exp;
#else
    constexpr int EXP = 0;
#endif
    issue_turn(kbeg);   // (an empty group reads the clamped row kend - 1 >= 0 and never uses it)
    for (int64_t s0 = kbeg; s0 < kend; s0 += 2 * D) {
        commit_turn(s0);
        __builtin_amdgcn_sched_barrier(0);
        issue_turn(s0 + 2 * D);
        __builtin_amdgcn_sched_barrier(0);
        // LDS operands one step ahead of the MFMAs that use them
        float hidN = sH[lh * 32 + li], b5N = TN > 4 ? sX5[lh * 32 + li] : 0.0f, dzN[XF];
#pragma unroll
        for (int o = 0; o < XF; o++) dzN[o] = sD[lh * 4 + o];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const float hid = hidN, b5 = b5N;
            float b[TN], dz[XF];
#pragma unroll
            for (int o = 0; o < XF; o++) dz[o] = dzN[o];
            if (d + 1 < D) {
                const int j = 2 * (d + 1) + lh;
                hidN = sH[j * 32 + li];
                if (TN > 4) b5N = sX5[j * 32 + li];
#pragma unroll
                for (int o = 0; o < XF; o++) dzN[o] = sD[j * 4 + o];
            }
#pragma unroll
            for (int tn = 0; tn < 4; tn++) b[tn] = xc[d][tn];
            if (TN > 4) b[TN - 1] = b5;
            float sum = 0.0f;
#pragma unroll
            for (int o = 0; o < XF; o++) sum = fmaf(dz[o], w2[o], sum);
            float a = sum * act_grad_fast(hid, ACT1);
#ifdef NGP_AB_VARIANTS
            a = (EXP & 4) ? hid : a;
#endif
            if (W2G) {
#pragma unroll
                for (int o = 0; o < XF; o++) {
                    gw[o] = fmaf(dz[o], hid, gw[o]);
                    gb2[o] += dz[o];
                }
            }
            gb1 += a;
#pragma unroll
            for (int tn = 0; tn < TN; tn++) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[tn], acc[tn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();   // the staging tiles lie inside the regions the hand-over below writes
    if (EXP & 1) {   // diagnostic: the loop alone (one store keeps the sums alive)
        float t = gb1;
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) t += acc[tn][r];
#pragma unroll
        for (int o = 0; o < XF; o++) t += gw[o] + gb2[o];
        if (t == 123.456f) p.C[lane] = t;
        return;
    }
    // ---- the second group hands its sums to the first, lane for lane
    if (two) {
        float* comb = lds + (wave * (TN * 16 + 2 * XF + 1)) * 64 + lane;
        if (grp == 1) {
#pragma unroll
            for (int tn = 0; tn < TN; tn++)
#pragma unroll
                for (int r = 0; r < 16; r++) comb[(tn * 16 + r) * 64] = acc[tn][r];
#pragma unroll
            for (int o = 0; o < XF; o++) { comb[(TN * 16 + o) * 64] = gw[o]; comb[(TN * 16 + XF + o) * 64] = gb2[o]; }
            comb[(TN * 16 + 2 * XF) * 64] = gb1;
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tn][r] += comb[(tn * 16 + r) * 64];
#pragma unroll
        for (int o = 0; o < XF; o++) { gw[o] += comb[(TN * 16 + o) * 64]; gb2[o] += comb[(TN * 16 + XF + o) * 64]; }
        gb1 += comb[(TN * 16 + 2 * XF) * 64];
    }
    // The lane's four accumulator blocks stand for ADJACENT columns: straight atomics would touch 16-byte
    // pieces 64 bytes apart.  Half a tile at a time (16 rows) goes through LDS and leaves as atomics on
    // 256 contiguous bytes per instruction.  (A wave only touches its own slice of `lds` from here on, and
    // that slice lies inside the region it alone read above.)
    float (*wstage)[128] = reinterpret_cast<float (*)[128]>(lds + wave * (two ? (TN * 16 + 2 * XF + 1) * 64 : 2048));
#pragma unroll
    for (int half2 = 0; half2 < 2; half2++) {
#pragma unroll
        for (int rr = 0; rr < 8; rr++) {
            const int r = half2 * 8 + rr;
            const int row16 = (rr & 3) + 8 * (rr >> 2) + 4 * lh;        // row inside the 16-row half
            const float4 v = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            *reinterpret_cast<float4*>(&wstage[row16][4 * li]) = v;
        }
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int flat = i * 64 + lane, row16 = flat >> 7, col = flat & 127;
            atomicAdd(p.C + (int64_t)(wave * 32 + half2 * 16 + row16) * p.ldc + col, wstage[row16][col]);
        }
    }
    if (TN > 4) {
        const int col = 128 + li;
        if (col < N) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                atomicAdd(p.C + (int64_t)row * p.ldc + col, acc[TN - 1][r]);
            }
        }
    }
    // per-lane sums: the two sample parities (lh) meet, lanes 0..31 carry hidden unit h
    gb1 += __shfl_xor(gb1, 32, 64);
    if (p.bias_grad && lh == 0) atomicAdd(p.bias_grad + h, gb1);
    if (W2G && p.xf_dW2) {
#pragma unroll
        for (int o = 0; o < XF; o++) {
            const float t = gw[o] + __shfl_xor(gw[o], 32, 64);
            if (lh == 0 && o < p.xf_nout) atomicAdd(p.xf_dW2 + o * p.xf_lddw2 + h, t);
            const float t2 = gb2[o] + __shfl_xor(gb2[o], 32, 64);   // every lane of a half holds the same sum
            if (p.xf_db2 && wave == 0 && lane == 0 && o < p.xf_nout) atomicAdd(p.xf_db2 + o, t2);
        }
    }
}

// ---------------------------------------------------------------- skinny layers (n_out <= 4)
// forward: one half-wave per sample row, each lane owns float4 pieces of the row (16-byte loads),
// four rows per half-wave in flight, dot products reduced by xor-shuffles.
__global__ void __launch_bounds__(256) skinny_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ b, int64_t n, int n_in, int n_out,
                                                         int act, float* __restrict__ y, int64_t ldy,
                                                         float* __restrict__ z_pre, int vec)
{
    const int lane = threadIdx.x & 31;
    const int64_t hw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int64_t row0 = hw * 4;
    if (row0 >= n) return;
    float acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int o = 0; o < 4; o++) acc[u][o] = 0.0f;
    if (vec) {
        for (int k = lane * 4; k < n_in; k += 128) {
            float4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t r = row0 + u < n ? row0 + u : n - 1;
                xv[u] = *reinterpret_cast<const float4*>(x + r * ldx + k);
            }
#pragma unroll
            for (int o = 0; o < 4; o++) {
                if (o < n_out) {
                    const float4 w = *reinterpret_cast<const float4*>(W + o * ldw + k);
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        acc[u][o] += xv[u].x * w.x + xv[u].y * w.y + xv[u].z * w.z + xv[u].w * w.w;
                }
            }
        }
    } else {
        for (int k = lane; k < n_in; k += 32) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t r = row0 + u < n ? row0 + u : n - 1;
                const float xv = x[r * ldx + k];
#pragma unroll
                for (int o = 0; o < 4; o++) if (o < n_out) acc[u][o] = fmaf(xv, W[o * ldw + k], acc[u][o]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int o = 0; o < 4; o++) {
            if (o >= n_out) continue;
            float v = half_sum(acc[u][o]);
            if (lane == 0 && row0 + u < n) {
                if (b) v += b[o];
                if (z_pre) z_pre[(row0 + u) * n_out + o] = v;
                y[(row0 + u) * ldy + o] = act_fwd(v, act);
            }
        }
}

__global__ void skinny_dgrad_kernel(const float* __restrict__ dz, int64_t lddz, const float* __restrict__ W, int64_t ldw,
                                    int64_t n, int n_in, int n_out, float* __restrict__ dx, int64_t lddx,
                                    int accumulate)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * n_in) return;
    const int64_t row = i / n_in;
    const int k = (int)(i - row * n_in);
    float v = accumulate ? dx[row * lddx + k] : 0.0f;
    for (int o = 0; o < n_out; o++) v = fmaf(dz[row * lddz + o], W[o * ldw + k], v);
    dx[row * lddx + k] = v;
}

// dW[o][k] += sum_rows dz[row][o]*x[row][k] (n_out <= 4), db[o] += sum_rows dz[row][o].
// 256 threads = (256/CW) row lanes x CW column lanes; rows are walked four at a time so that four
// independent x loads are in flight per thread; LDS tree over the row lanes; one atomic per output.
__global__ void __launch_bounds__(256) skinny_wgrad_kernel(const float* __restrict__ dz, int64_t lddz,
                                                           const float* __restrict__ x, int64_t ldx, int64_t n,
                                                           int n_in, int n_out, int cw, int rows_per_block,
                                                           float* __restrict__ dW, int64_t ldw, float* __restrict__ db)
{
    __shared__ float part[4][256];
    const int k = threadIdx.x % cw, rl = threadIdx.x / cw, nrl = 256 / cw;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    float acc[4] = { 0, 0, 0, 0 }, bsum[4] = { 0, 0, 0, 0 };
    if (k < n_in) {
        for (int64_t r = r0 + rl; r < r1; r += 4 * nrl) {
            float xv[4], dv[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t rr = r + u * nrl;
                const bool ok = rr < r1;
                xv[u] = ok ? x[rr * ldx + k] : 0.0f;
#pragma unroll
                for (int o = 0; o < 4; o++) dv[u][o] = (ok && o < n_out) ? dz[rr * lddz + o] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int o = 0; o < 4; o++) { acc[o] = fmaf(dv[u][o], xv[u], acc[o]); bsum[o] += dv[u][o]; }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; o++) part[o][threadIdx.x] = acc[o];
    __syncthreads();
    for (int s = nrl / 2; s > 0; s >>= 1) {
        if (rl < s) {
#pragma unroll
            for (int o = 0; o < 4; o++) part[o][threadIdx.x] += part[o][threadIdx.x + s * cw];
        }
        __syncthreads();
    }
    if (rl == 0 && k < n_in) {
#pragma unroll
        for (int o = 0; o < 4; o++) if (o < n_out) atomicAdd(dW + o * ldw + k, part[o][threadIdx.x]);
    }
    if (db) { // column 0's lanes hold the dz sums: reduce them over the row lanes as well
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 4; o++) part[o][threadIdx.x] = (k == 0) ? bsum[o] : 0.0f;
        __syncthreads();
        for (int s = nrl / 2; s > 0; s >>= 1) {
            if (rl < s && k == 0) {
#pragma unroll
                for (int o = 0; o < 4; o++) part[o][threadIdx.x] += part[o][threadIdx.x + s * cw];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
#pragma unroll
            for (int o = 0; o < 4; o++) if (o < n_out) atomicAdd(db + o, part[o][0]);
        }
    }
}

// db[j] += sum_rows dz[row][j].  256 threads = (256/CW) row lanes x CW columns (CW = n_out rounded
// up to a power of two <= 128), LDS tree over the row lanes, one atomic per column per block.
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ dz, int64_t lddz, int64_t n, int n_out,
                                                     int cw, int rows_per_block, float* __restrict__ db)
{
    __shared__ float part[256];
    const int j = threadIdx.x % cw, rl = threadIdx.x / cw, nrl = 256 / cw;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    float a = 0.0f;
    if (j < n_out)
        for (int64_t r = r0 + rl; r < r1; r += nrl) a += dz[r * lddz + j];
    part[threadIdx.x] = a;
    __syncthreads();
    for (int s = nrl / 2; s > 0; s >>= 1) {
        if (rl < s) part[threadIdx.x] += part[threadIdx.x + s * cw];
        __syncthreads();
    }
    if (rl == 0 && j < n_out) atomicAdd(db + j, part[threadIdx.x]);
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yz, int64_t count, int act,
                               float* __restrict__ dz)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float g = dy ? dy[i] : 1.0f;
        float v;
        switch (act) {
            case NGP_ACT_RELU: v = yz[i] > 0.0f ? g : 0.0f; break;
            case NGP_ACT_SIGMOID: { const float s = yz[i]; v = g * s * (1.0f - s); } break;
            case NGP_ACT_SOFTPLUS: v = g * -expm1f(-yz[i]); break; // sigmoid(z) = 1 - exp(-softplus(z))
            case NGP_ACT_EXP: v = g * yz[i]; break;
            default: v = g;
        }
        dz[i] = v;
    }
}

// act_bwd over (n, cols <= 4) rows that also accumulates sum_rows ||dz[row, :]||_2 into *norm_acc (the norm-bound sums of
// ngp_clip_decide come out of the pass that forms dz2 instead of a launch of their own)
__global__ void __launch_bounds__(256) act_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ yz,
                                                           int64_t n, int cols, int act, float* __restrict__ dz,
                                                           float* __restrict__ norm_acc)
{
    __shared__ float part[4];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float acc = 0.0f;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        float q = 0.0f;
        for (int c = 0; c < cols; c++) {
            const int64_t i = r * cols + c;
            const float g = dy ? dy[i] : 1.0f;
            float v;
            switch (act) {
                case NGP_ACT_RELU: v = yz[i] > 0.0f ? g : 0.0f; break;
                case NGP_ACT_SIGMOID: { const float s = yz[i]; v = g * s * (1.0f - s); } break;
                case NGP_ACT_SOFTPLUS: v = g * -expm1f(-yz[i]); break;
                case NGP_ACT_EXP: v = g * yz[i]; break;
                default: v = g;
            }
            dz[i] = v;
            q = fmaf(v, v, q);
        }
        acc += sqrtf(q);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(norm_acc, part[0] + part[1] + part[2] + part[3]);
}

// derivative of an activation expressed through its OUTPUT y

// Hidden-layer backward of a 2-layer MLP whose output layer is narrow (n_out <= 16):
//   dz2[n][o] = dOut[n][o] * act2'(out[n][o])          (written to dz2, row stride lddz2)
//   dz1[n][j] = (sum_o dz2[n][o] * W2[o][j]) * act1'(hidden[n][j])
// Thread = hidden column j, W2's column in registers, rows strided over the grid.  Replaces an
// MFMA dgrad with K <= 16 plus two elementwise passes; purely bandwidth bound.
// dOut == nullptr means "all ones" (the d(sigma)/dx pass of the density head).
template <int OMAX, bool W2G>
__global__ void __launch_bounds__(256) mlp_hidden_bwd_kernel(const float* __restrict__ dOut, int64_t lddo,
                                                             const float* __restrict__ out, int64_t ldo, int act2,
                                                             const float* __restrict__ W2, int64_t ldw2,
                                                             const float* __restrict__ hidden, int64_t ldh, int act1,
                                                             int64_t n, int H, int n_out,
                                                             float* __restrict__ dz2, int64_t lddz2,
                                                             float* __restrict__ dz1, int64_t lddz1,
                                                             float* __restrict__ dW2, int64_t lddw2,
                                                             float* __restrict__ db2)
{
    // lane = (row lane, 4 consecutive hidden columns): 16-byte loads/stores of hidden / dz1
    const int lpr = H / 4;                 // lanes per row
    const int rows_per_block = 256 / lpr;
    const int j4 = (threadIdx.x % lpr) * 4;
    const int rsub = threadIdx.x / lpr;
    float w[OMAX][4];
    float gw[W2G ? OMAX : 1][4], gb[W2G ? OMAX : 1];   // W2G: dW2 / db2 partials of this thread
#pragma unroll
    for (int o = 0; o < OMAX; o++)
#pragma unroll
        for (int c = 0; c < 4; c++) w[o][c] = o < n_out ? W2[o * ldw2 + j4 + c] : 0.0f;
#pragma unroll
    for (int o = 0; o < (W2G ? OMAX : 1); o++) {
        gb[o] = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) gw[o][c] = 0.0f;
    }
    const int64_t stride = (int64_t)gridDim.x * rows_per_block;
    for (int64_t row0 = (int64_t)blockIdx.x * rows_per_block + rsub; row0 < n; row0 += 2 * stride) {
        float4 hv[2];
        float acc[2][4];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int64_t row = row0 + u * stride;
            hv[u] = row < n ? *reinterpret_cast<const float4*>(hidden + row * ldh + j4) : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; c++) acc[u][c] = 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int64_t row = row0 + u * stride;
            if (row >= n) continue;
#pragma unroll
            for (int o = 0; o < OMAX; o++) {
                if (o < n_out) {
                    const float g = dOut ? dOut[row * lddo + o] : 1.0f;
                    const float d = g * act_grad_from_output(out[row * ldo + o], act2);
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[u][c] = fmaf(d, w[o][c], acc[u][c]);
                    if (dz2 && j4 == 0) dz2[row * lddz2 + o] = d;
                    if (W2G) {
                        gb[o] += d;
                        gw[o][0] = fmaf(d, hv[u].x, gw[o][0]); gw[o][1] = fmaf(d, hv[u].y, gw[o][1]);
                        gw[o][2] = fmaf(d, hv[u].z, gw[o][2]); gw[o][3] = fmaf(d, hv[u].w, gw[o][3]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int64_t row = row0 + u * stride;
            if (row >= n) continue;
            float4 r;
            r.x = acc[u][0] * act_grad_from_output(hv[u].x, act1);
            r.y = acc[u][1] * act_grad_from_output(hv[u].y, act1);
            r.z = acc[u][2] * act_grad_from_output(hv[u].z, act1);
            r.w = acc[u][3] * act_grad_from_output(hv[u].w, act1);
            *reinterpret_cast<float4*>(dz1 + row * lddz1 + j4) = r;
        }
    }
    if (W2G) {
        // threads that own the same four columns (same threadIdx % lpr) meet in LDS; one atomic per
        // output element and block (the launch is capped at 1024 blocks for that reason)
        __shared__ float red[256 * 4 * OMAX + 256 * OMAX];
#pragma unroll
        for (int o = 0; o < OMAX; o++) {
#pragma unroll
            for (int c = 0; c < 4; c++) red[(rsub * OMAX + o) * H + j4 + c] = gw[o][c];   // [rsub][o][H]: 256*4*OMAX floats
            if (j4 == 0) red[256 * 4 * OMAX + rsub * OMAX + o] = gb[o];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < OMAX * H; idx += 256) {
            const int o = idx / H, c = idx - o * H;
            if (o < n_out) {
                float sum = 0.0f;
                for (int r = 0; r < rows_per_block; r++) sum += red[(r * OMAX + o) * H + c];
                atomicAdd(dW2 + o * lddw2 + c, sum);
            }
        }
        if (db2 && threadIdx.x < OMAX && threadIdx.x < n_out) {
            float sum = 0.0f;
            for (int r = 0; r < rows_per_block; r++) sum += red[256 * 4 * OMAX + r * OMAX + threadIdx.x];
            atomicAdd(db2 + threadIdx.x, sum);
        }
    }
}

// ---------------------------------------------------------------- optimizer
template <int U>   // 16-byte quads per lane and trip
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float step_size, float beta1, float beta2,
                                                   float eps, float bc2_sqrt, float weight_decay,
                                                   const float* __restrict__ grad_scale, int zero_grad, bool sparse_zero)
{
    const float gs = grad_scale ? *grad_scale : 1.0f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p); float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
    auto update = [&](float4& pp, const float4& gg, float4& mm, float4& vv) {
        float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float gr = ga[j] * gs;
            if (weight_decay != 0.0f) gr += weight_decay * pa[j];
            ma[j] = ma[j] + (gr - ma[j]) * (1.0f - beta1);
            va[j] = va[j] * beta2 + (1.0f - beta2) * gr * gr;
            const float denom = sqrtf(va[j]) / bc2_sqrt + eps;
            pa[j] = pa[j] - step_size * (ma[j] / denom);
        }
    };
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto nz = [](const float4& q) { return q.x != 0.0f || q.y != 0.0f || q.z != 0.0f || q.w != 0.0f; };
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // U independent 16-B quads per lane and trip: the launch is capped well below full occupancy
    // (other streams' kernels must be able to get wave slots and registers), so the bytes in flight come from
    // the unroll instead of from more waves
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 pa[U], ga[U], ma[U], va[U];
#pragma unroll
        for (int u = 0; u < U; u++) { pa[u] = p4[i + u * stride]; ga[u] = g4[i + u * stride]; ma[u] = m4[i + u * stride]; va[u] = v4[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            update(pa[u], ga[u], ma[u], va[u]);
            p4[i + u * stride] = pa[u]; m4[i + u * stride] = ma[u]; v4[i + u * stride] = va[u];
            // untouched table rows (no sample near them this step) already hold zeros: do not write them again
            if (zero_grad && (sparse_zero ? nz(ga[u]) : true)) g4[i + u * stride] = zero4;
        }
    }
    for (; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        update(pp, gg, mm, vv);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (zero_grad && (sparse_zero ? nz(gg) : true)) g4[i] = zero4;
    }
    // tail
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gr = g[i] * gs;
        if (weight_decay != 0.0f) gr += weight_decay * p[i];
        const float mn = m[i] + (gr - m[i]) * (1.0f - beta1);
        const float vn = v[i] * beta2 + (1.0f - beta2) * gr * gr;
        m[i] = mn; v[i] = vn;
        p[i] = p[i] - step_size * (mn / (sqrtf(vn) / bc2_sqrt + eps));
        if (zero_grad) g[i] = 0.0f;
    }
}

__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out,
                                                    const int32_t* __restrict__ flag)
{
    if (flag && *flag == 0) return;   // the clip decision did not need the exact norm

    __shared__ float part[4];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float a = 0.0f, b = 0.0f;
    // 16-byte loads, two independent quads per trip (x is 16-byte aligned: checked by the caller)
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int64_t n4 = n >> 2;
    int64_t i = tid;
    for (; i + stride < n4; i += 2 * stride) {
        const float4 p = x4[i], q = x4[i + stride];
        a = fmaf(p.x, p.x, a); a = fmaf(p.y, p.y, a); a = fmaf(p.z, p.z, a); a = fmaf(p.w, p.w, a);
        b = fmaf(q.x, q.x, b); b = fmaf(q.y, q.y, b); b = fmaf(q.z, q.z, b); b = fmaf(q.w, q.w, b);
    }
    for (; i < n4; i += stride) {
        const float4 p = x4[i];
        a = fmaf(p.x, p.x, a); a = fmaf(p.y, p.y, a); a = fmaf(p.z, p.z, a); a = fmaf(p.w, p.w, a);
    }
    for (int64_t j = (n4 << 2) + tid; j < n; j += stride) a = fmaf(x[j], x[j], a);
    a += b;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

__global__ void __launch_bounds__(256) sumsq_scalar_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out,
                                                           const int32_t* __restrict__ flag)
{
    if (flag && *flag == 0) return;

    __shared__ float part[4];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float a = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a = fmaf(x[i], x[i], a);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// coef = extra_scale * min(1, max_norm / (sqrt(sumsq) + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float extra_scale,
                                 float* __restrict__ coef)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float norm = sqrtf(*sumsq) * extra_scale;
        float c = max_norm / (norm + 1e-6f);
        if (c > 1.0f) c = 1.0f;
        *coef = c * extra_scale;
    }
}

// sum over rows of the Euclidean norm of the first `cols` entries (cols <= 16): one lane per row
__global__ void __launch_bounds__(256) row_norm_sum_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int cols,
                                                           float* __restrict__ out)
{
    __shared__ float part[4];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float a = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float q = 0.0f;
        for (int c = 0; c < cols; c++) { const float v = x[i * ldx + c]; q = fmaf(v, v, q); }
        a += sqrtf(q);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// clipping settled from an upper bound of the gradient norm (see ngp_clip_decide in the header)
__global__ void __launch_bounds__(1024) clip_decide_kernel(const float* __restrict__ sums,
                                                           const float* __restrict__ w1a, int64_t n1a,
                                                           const float* __restrict__ w2a, int64_t n2a,
                                                           const float* __restrict__ w1b, int64_t n1b,
                                                           const float* __restrict__ w2b, int64_t n2b,
                                                           const float* __restrict__ sumsq_rest, float max_norm,
                                                           float extra_scale, float* __restrict__ coef,
                                                           int32_t* __restrict__ need_exact,
                                                           const float* __restrict__ rest, int64_t n_rest,
                                                           float* __restrict__ sumsq_out)
{
    __shared__ float ws[5][16];
    const float* ptr[5] = {w1a, w2a, w1b, w2b, rest};
    const int64_t cnt[5] = {n1a, n2a, n1b, n2b, rest ? n_rest : 0};
    for (int k = 0; k < 5; k++) {
        float q = 0.0f;
        // 16-byte pieces, several in flight per thread (one block reads ~80 k floats: the loads' latency is the launch's length)
        const bool vec = (((uintptr_t)ptr[k]) & 15) == 0;
        const int64_t n4 = vec ? cnt[k] / 4 : 0;
        const float4* p4 = reinterpret_cast<const float4*>(ptr[k]);
        float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
#pragma unroll 4
        for (int64_t i = threadIdx.x; i < n4; i += 1024) {
            const float4 v = p4[i];
            q0 = fmaf(v.x, v.x, q0); q1 = fmaf(v.y, v.y, q1); q2 = fmaf(v.z, v.z, q2); q3 = fmaf(v.w, v.w, q3);
        }
        q = (q0 + q1) + (q2 + q3);
        for (int64_t i = 4 * n4 + threadIdx.x; i < cnt[k]; i += 1024) q = fmaf(ptr[k][i], ptr[k][i], q);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        if ((threadIdx.x & 63) == 0) ws[k][threadIdx.x >> 6] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float f[4];
        for (int k = 0; k < 4; k++) {
            float t = 0.0f;
            for (int w = 0; w < 16; w++) t += ws[k][w];
            f[k] = sqrtf(t);
        }
        float rest_sq = sumsq_rest ? *sumsq_rest : 0.0f;
        if (rest) {                       // the exact part summed here (ngp_clip_decide_rest): no launch of its own
            for (int w = 0; w < 16; w++) rest_sq += ws[4][w];
            *sumsq_out = rest_sq;         // the conditional exact route (ngp_sumsq_if) continues from it
        }
        const float ba = f[0] * f[1] * sums[0], bb = f[2] * f[3] * sums[1];
        const float bound = sqrtf(ba * ba + bb * bb + rest_sq) * extra_scale;
        const bool safe = bound * 1.001f + 1e-6f < max_norm;     // false for NaN / inf
        *need_exact = safe ? 0 : 1;
        if (safe) *coef = extra_scale;
    }
}

__global__ void clip_coef_if_kernel(const float* __restrict__ sumsq, float max_norm, float extra_scale,
                                    float* __restrict__ coef, const int32_t* __restrict__ flag)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && *flag != 0) {
        const float norm = sqrtf(*sumsq) * extra_scale;
        float c = max_norm / (norm + 1e-6f);
        if (c > 1.0f) c = 1.0f;
        *coef = c * extra_scale;
    }
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

#define LAUNCH_GEMM(MODE, WM, WN, TM, TN)                                                                   \
    do {                                                                                                   \
        if (XF == 0) hipLaunchKernelGGL((gemm_kernel<MODE, WM, WN, TM, TN>), grid, dim3(256), 0, st, p);      \
        else if (p.xf_dW2) hipLaunchKernelGGL((gemm_xf_kernel<MODE, WM, WN, TM, TN, (XF ? XF : 1) + (MODE == MODE_WGRAD ? 16 : 0)>), grid, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((gemm_xf_kernel<MODE, WM, WN, TM, TN, (XF ? XF : 1)>), grid, dim3(256), 0, st, p); \
    } while (0)

// C (M, N) = A (M, K) . B (K, N): 128-wide tiles when the output is wider than 32 columns
template <int XF>
void launch_dgrad(const GemmArgs& p, hipStream_t st)
{
    if (p.N > 32) {
        dim3 grid(ngp_blocks(p.M, 128), ngp_blocks(p.N, 128));
        LAUNCH_GEMM(MODE_DGRAD, 2, 2, 2, 2);
    } else {
        dim3 grid(ngp_blocks(p.M, 128), 1);
        LAUNCH_GEMM(MODE_DGRAD, 4, 1, 1, 1);
    }
}

// C (M, N) += A^T . B over K = samples, split so that one full wave of workgroups (3 per CU) is in flight
template <int XF>
void launch_wgrad(GemmArgs p, hipStream_t st)
{
    const int64_t n = p.K;
    const bool big_m = p.M > 32, big_n = p.N > 32;
    const int64_t tiles = (int64_t)ngp_blocks(p.M, big_m ? 128 : 32) * ngp_blocks(p.N, big_n ? 128 : 32);
    static const int target_blocks = (int)ab_long("NGP_WGRAD_BLOCKS", 768); // 3 workgroups per CU x 256 CUs
    int64_t splits = target_blocks / (tiles > 0 ? tiles : 1);
    if (splits < 1) splits = 1;
    int64_t chunk = (n + splits - 1) / splits;
    chunk = (chunk + BK - 1) / BK * BK;
    if (chunk < 256) chunk = 256;
    splits = (n + chunk - 1) / chunk;
    p.k_chunk = chunk;
    if (big_m && big_n) {
        dim3 grid(ngp_blocks(p.M, 128), ngp_blocks(p.N, 128), (unsigned)splits);
        LAUNCH_GEMM(MODE_WGRAD, 2, 2, 2, 2);
    } else if (big_m) { // tall: 128 x 32 tiles
        dim3 grid(ngp_blocks(p.M, 128), ngp_blocks(p.N, 32), (unsigned)splits);
        LAUNCH_GEMM(MODE_WGRAD, 4, 1, 1, 1);
    } else if (big_n) { // wide: 32 x 128 tiles
        dim3 grid(ngp_blocks(p.M, 32), ngp_blocks(p.N, 128), (unsigned)splits);
        LAUNCH_GEMM(MODE_WGRAD, 1, 4, 1, 1);
    } else {
        dim3 grid(ngp_blocks(p.M, 64), ngp_blocks(p.N, 64), (unsigned)splits);
        LAUNCH_GEMM(MODE_WGRAD, 2, 2, 1, 1);
    }
}

inline bool xf_args_ok(const float* dz2, int64_t lddz2, const float* W2, int64_t ldw2, const float* hidden,
                       int64_t ldh, int H, int n_out)
{
    return dz2 && W2 && hidden && n_out >= 1 && n_out <= XF_OMAX && lddz2 >= n_out && ldw2 >= H && ldh >= H && H >= 8;
}

} // namespace


extern "C" {

int ngp_linear_fwd(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* b, int64_t n, int n_in,
                   int n_out, int activation, float* y, int64_t ldy, float* z_pre, void* stream)
{
    if (n < 0 || n_in < 1 || n_out < 1 || ldx < n_in || ldw < n_in || ldy < n_out) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!x || !W || !y) return NGP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n_out <= 4) {
        const int vec = aligned16(x) && aligned16(W) && (ldx % 4 == 0) && (ldw % 4 == 0) && (n_in % 4 == 0);
        hipLaunchKernelGGL(skinny_fwd_kernel, dim3(ngp_blocks((n + 3) / 4 * 32, 256)), dim3(256), 0, st, x, ldx, W, ldw,
                           b, n, n_in, n_out, activation, y, ldy, z_pre, vec);
        return ngp_check_launch();
    }
    GemmArgs p{};
    p.A = x; p.lda = ldx; p.B = W; p.ldb = ldw; p.C = y; p.ldc = ldy;
    p.M = n; p.N = n_out; p.K = n_in; p.bias = b; p.z_pre = z_pre; p.act = activation; p.k_chunk = 0;
    p.vecA = aligned16(x) && (ldx % 4 == 0); p.vecB = aligned16(W) && (ldw % 4 == 0);
    if (n_out > 32) {
        dim3 grid(ngp_blocks(n, 128), ngp_blocks(n_out, 128));
        hipLaunchKernelGGL((gemm_kernel<MODE_FWD, 2, 2, 2, 2>), grid, dim3(256), 0, st, p);
    } else {
        dim3 grid(ngp_blocks(n, 128), 1);
        hipLaunchKernelGGL((gemm_kernel<MODE_FWD, 4, 1, 1, 1>), grid, dim3(256), 0, st, p);
    }
    return ngp_check_launch();
}

static int mlp2_fwd_impl(const float* x, int64_t ldx, const float* W1, int64_t ldw1, const float* b1, int act1,
                         const float* W2, int64_t ldw2, const float* b2, int act2, int64_t n, int n_in, int H, int n_out,
                         float* hidden, int64_t ldh, float* out, int64_t ldo, float* dact, void* stream);

int ngp_mlp2_fwd(const float* x, int64_t ldx, const float* W1, int64_t ldw1, const float* b1, int act1,
                 const float* W2, int64_t ldw2, const float* b2, int act2, int64_t n, int n_in, int H, int n_out,
                 float* hidden, int64_t ldh, float* out, int64_t ldo, void* stream)
{
    return mlp2_fwd_impl(x, ldx, W1, ldw1, b1, act1, W2, ldw2, b2, act2, n, n_in, H, n_out, hidden, ldh, out, ldo, nullptr,
                         stream);
}

int ngp_mlp2_fwd_dact(const float* x, int64_t ldx, const float* W1, int64_t ldw1, const float* b1, int act1,
                      const float* W2, int64_t ldw2, const float* b2, int act2, int64_t n, int n_in, int H, int n_out,
                      float* hidden, int64_t ldh, float* out, int64_t ldo, float* dact_out, void* stream)
{
    if (n > 0 && !dact_out) return NGP_EINVAL;
    return mlp2_fwd_impl(x, ldx, W1, ldw1, b1, act1, W2, ldw2, b2, act2, n, n_in, H, n_out, hidden, ldh, out, ldo, dact_out,
                         stream);
}

static int mlp2_fwd_impl(const float* x, int64_t ldx, const float* W1, int64_t ldw1, const float* b1, int act1,
                         const float* W2, int64_t ldw2, const float* b2, int act2, int64_t n, int n_in, int H, int n_out,
                         float* hidden, int64_t ldh, float* out, int64_t ldo, float* dact, void* stream)
{
    if (n < 0 || n_in < 1 || H < 1 || H > 128 || n_out < 1 || n_out > 8 || ldx < n_in || ldw1 < n_in || ldh < H ||
        ldw2 < H || ldo < n_out)
        return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!x || !W1 || !W2 || !hidden || !out) return NGP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    GemmArgs p{};
    p.A = x; p.lda = ldx; p.B = W1; p.ldb = ldw1; p.C = hidden; p.ldc = ldh;
    p.M = n; p.N = H; p.K = n_in; p.bias = b1; p.act = act1;
    p.vecA = aligned16(x) && (ldx % 4 == 0); p.vecB = aligned16(W1) && (ldw1 % 4 == 0);
    p.f2_W2 = W2; p.f2_ldw2 = ldw2; p.f2_b2 = b2; p.f2_out = out; p.f2_ldo = ldo; p.f2_nout = n_out; p.f2_act = act2;
    p.f2_dact = dact;
    dim3 grid(ngp_blocks(n, 128), 1);
    static const bool stream_ok = !ab_flag("NGP_MLP_NO_STREAM");
    const bool wide_ok = H == 128 && (n_in == 128 || n_in == 144 || n_in == 160);
    // the 32-wide heads are HBM-bound either way (0.062 ms streaming, 0.060 ms tiled): opt-in only
    static const bool stream_heads = ab_flag("NGP_MLP_STREAM_HEADS");
    const bool head_ok = stream_heads && H == 32 && n_in == 128;
    if (stream_ok && (wide_ok || head_ok) && p.vecA && p.vecB &&
        (act1 == NGP_ACT_RELU || act1 == NGP_ACT_SOFTPLUS) && aligned16(hidden) && ldh % 4 == 0 &&
        n <= (int64_t)32 * 0x7fffff00) {
        // streaming kernel: W1 resident in LDS, workgroups of 8 waves, 32-row tiles per wave.
        // Chosen by shape only, never by n: a row's result must not depend on the size of the batch it
        // sits in (the test-time renderer's two loops are compared bit for bit)
        const int n_tiles = (int)((n + 31) / 32);
        const int f2 = n_out == 1 ? 1 : (n_out <= 4 ? 4 : 8);
        static const int nw = (int)ab_long("NGP_MLP_NW_FWD", ab_long("NGP_MLP_NW", 8));   // waves per workgroup (A/B: 4)
        const size_t lds = (size_t)(H * (n_in + 4) + f2 * H + H + 8 + nw * 1024) * sizeof(float);
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return NGP_ELAUNCH;
            n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        // one workgroup per CU for the 128-wide layers (W1 takes half the LDS), two for the 32-wide heads
        static const int wg_rounds_f = (int)ab_long("NGP_MLP_WG_ROUNDS_FWD", ab_long("NGP_MLP_WG_ROUNDS", 1));   // A/B: k workgroups per CU, one after the other
        const int max_blocks = n_cu * (H == 128 ? 1 : 2) * wg_rounds_f;
        const int blocks = (n_tiles + nw - 1) / nw < max_blocks ? (n_tiles + nw - 1) / nw : max_blocks;
#define LAUNCH_STREAM(F2V, KQV, ACTV, TNV)                                                                              \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_stream_fwd_kernel<F2V, KQV, ACTV, TNV>),         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)              \
                return NGP_ELAUNCH;                                                                                     \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((mlp_stream_fwd_kernel<F2V, KQV, ACTV, TNV>), dim3(blocks), dim3(64 * nw), lds, st, p, n_tiles); \
    } while (0)
#define LAUNCH_STREAM_A(F2V, KQV, TNV)                                                                                  \
    do {                                                                                                                \
        if (act1 == NGP_ACT_RELU) LAUNCH_STREAM(F2V, KQV, NGP_ACT_RELU, TNV);                                           \
        else LAUNCH_STREAM(F2V, KQV, NGP_ACT_SOFTPLUS, TNV);                                                            \
    } while (0)
#define LAUNCH_STREAM_K(F2V)                                                                                            \
    do {                                                                                                                \
        if (H == 32) LAUNCH_STREAM_A(F2V, 16, 1);                                                                       \
        else if (n_in == 128) LAUNCH_STREAM_A(F2V, 16, 4);                                                              \
        else if (n_in == 144) LAUNCH_STREAM_A(F2V, 18, 4);                                                              \
        else LAUNCH_STREAM_A(F2V, 20, 4);                                                                               \
    } while (0)
        if (f2 == 1) LAUNCH_STREAM_K(1);
        else if (f2 == 4) LAUNCH_STREAM_K(4);
        else LAUNCH_STREAM_K(8);
#undef LAUNCH_STREAM_K
#undef LAUNCH_STREAM_A
#undef LAUNCH_STREAM
        return ngp_check_launch();
    }
    if (H > 32) {
        if (n_out == 1) hipLaunchKernelGGL((gemm_fwd2_kernel<2, 2, 2, 2, 1>), grid, dim3(256), 0, st, p);
        else if (n_out <= 4) hipLaunchKernelGGL((gemm_fwd2_kernel<2, 2, 2, 2, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_fwd2_kernel<2, 2, 2, 2, 8>), grid, dim3(256), 0, st, p);
    } else {
        if (n_out == 1) hipLaunchKernelGGL((gemm_fwd2_kernel<4, 1, 1, 1, 1>), grid, dim3(256), 0, st, p);
        else if (n_out <= 4) hipLaunchKernelGGL((gemm_fwd2_kernel<4, 1, 1, 1, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_fwd2_kernel<4, 1, 1, 1, 8>), grid, dim3(256), 0, st, p);
    }
    return ngp_check_launch();
}

int ngp_linear_bwd_input(const float* dz, int64_t lddz, const float* W, int64_t ldw, int64_t n, int n_in, int n_out,
                         float* dx, int64_t lddx, int accumulate, void* stream)
{
    if (n < 0 || n_in < 1 || n_out < 1 || lddz < n_out || ldw < n_in || lddx < n_in) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!dz || !W || !dx) return NGP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n_out <= 4) {
        hipLaunchKernelGGL(skinny_dgrad_kernel, dim3(ngp_blocks(n * n_in, 256)), dim3(256), 0, st, dz, lddz, W, ldw, n,
                           n_in, n_out, dx, lddx, accumulate);
        return ngp_check_launch();
    }
    GemmArgs p{};
    p.A = dz; p.lda = lddz; p.B = W; p.ldb = ldw; p.C = dx; p.ldc = lddx;
    p.M = n; p.N = n_in; p.K = n_out; p.act = 0; p.k_chunk = 0; p.accumulate = accumulate;
    p.vecA = aligned16(dz) && (lddz % 4 == 0); p.vecB = aligned16(W) && (ldw % 4 == 0);
    launch_dgrad<0>(p, st);
    return ngp_check_launch();
}

int ngp_linear_bwd_weight(const float* dz, int64_t lddz, const float* x, int64_t ldx, int64_t n, int n_in, int n_out,
                          float* dW, int64_t ldw, float* db, void* stream)
{
    if (n < 0 || n_in < 1 || n_out < 1 || lddz < n_out || ldx < n_in || ldw < n_in) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!dz || !x || !dW) return NGP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n_out <= 4) {
        if (n_in > 256) return NGP_EINVAL;
        int cw = 1;
        while (cw < n_in) cw <<= 1;
        const int rpb = 512;
        hipLaunchKernelGGL(skinny_wgrad_kernel, dim3(ngp_blocks(n, rpb)), dim3(256), 0, st, dz, lddz, x, ldx, n, n_in,
                           n_out, cw, rpb, dW, ldw, db);
        return ngp_check_launch();
    }
    GemmArgs p{};
    p.A = dz; p.lda = lddz; p.B = x; p.ldb = ldx; p.C = dW; p.ldc = ldw;
    p.M = n_out; p.N = n_in; p.K = n; p.act = 0; p.bias_grad = db;
    p.vecA = aligned16(dz) && (lddz % 4 == 0); p.vecB = aligned16(x) && (ldx % 4 == 0);
    launch_wgrad<0>(p, st);
    return ngp_check_launch();
}

int ngp_act_bwd(const float* dy, const float* y_or_z, int64_t count, int activation, float* dz, void* stream)
{
    if (count < 0) return NGP_EINVAL;
    if (count == 0) return NGP_OK;
    if (!dz || (activation != NGP_ACT_NONE && !y_or_z)) return NGP_EINVAL;   // dy == NULL: all ones
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, y_or_z, count,
                       activation, dz);
    return ngp_check_launch();
}

int ngp_act_bwd_rows(const float* dy, const float* y_or_z, int64_t n, int cols, int activation, float* dz,
                     float* row_norm_acc, void* stream)
{
    if (n < 0 || cols < 1 || cols > 4) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!dz || !row_norm_acc || (activation != NGP_ACT_NONE && !y_or_z)) return NGP_EINVAL;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;      // one atomic per block
    hipLaunchKernelGGL(act_bwd_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, y_or_z, n, cols,
                       activation, dz, row_norm_acc);
    return ngp_check_launch();
}

int ngp_mlp_hidden_bwd(const float* dOut, int64_t lddo, const float* out, int64_t ldo, int act2, const float* W2,
                       int64_t ldw2, const float* hidden, int64_t ldh, int act1, int64_t n, int H, int n_out,
                       float* dz2, int64_t lddz2, float* dz1, int64_t lddz1, float* dW2, int64_t lddw2, float* db2,
                       void* stream)
{
    if (n < 0 || n_out < 1 || n_out > 16 || !(H == 32 || H == 64 || H == 128)) return NGP_EINVAL;
    if ((dW2 && (lddw2 < H || n_out > 4)) || (db2 && !dW2)) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!out || !W2 || !hidden || !dz1) return NGP_EINVAL;
    if (((uintptr_t)hidden & 15) || ((uintptr_t)dz1 & 15) || (ldh % 4) || (lddz1 % 4)) return NGP_EINVAL;
    const int rows_per_block = 256 / (H / 4);
    int64_t blocks = (n + rows_per_block - 1) / rows_per_block;
    if (blocks > 8192) blocks = 8192;
    if (dW2 && blocks > 1024) blocks = 1024;   // every block ends with one atomic per dW2 element
    hipStream_t st = (hipStream_t)stream;
    if (dW2)
        hipLaunchKernelGGL((mlp_hidden_bwd_kernel<4, true>), dim3((unsigned)blocks), dim3(256), 0, st, dOut, lddo, out,
                           ldo, act2, W2, ldw2, hidden, ldh, act1, n, H, n_out, dz2, lddz2, dz1, lddz1, dW2, lddw2, db2);
    else if (n_out <= 4)
        hipLaunchKernelGGL((mlp_hidden_bwd_kernel<4, false>), dim3((unsigned)blocks), dim3(256), 0, st, dOut, lddo, out,
                           ldo, act2, W2, ldw2, hidden, ldh, act1, n, H, n_out, dz2, lddz2, dz1, lddz1, nullptr, 0,
                           nullptr);
    else
        hipLaunchKernelGGL((mlp_hidden_bwd_kernel<16, false>), dim3((unsigned)blocks), dim3(256), 0, st, dOut, lddo, out,
                           ldo, act2, W2, ldw2, hidden, ldh, act1, n, H, n_out, dz2, lddz2, dz1, lddz1, nullptr, 0,
                           nullptr);
    return ngp_check_launch();
}

int ngp_mlp_bwd_input(const float* dz2, int64_t lddz2, const float* W2, int64_t ldw2, const float* hidden,
                      int64_t ldh, int act1, const float* W1, int64_t ldw1, int64_t n, int n_in, int H, int n_out,
                      float* dx, int64_t lddx, int accumulate, void* stream)
{
    if (n < 0 || n_in < 1 || n_out < 1 || n_out > XF_OMAX || H < 8 || ldw1 < n_in || lddx < n_in) return NGP_EINVAL;
    if (n == 0) return NGP_OK;   // empty batches carry null pointers
    if (!xf_args_ok(dz2, lddz2, W2, ldw2, hidden, ldh, H, n_out) || !W1 || !dx) return NGP_EINVAL;
    GemmArgs p{};
    p.A = hidden; p.lda = ldh; p.B = W1; p.ldb = ldw1; p.C = dx; p.ldc = lddx;
    p.M = n; p.N = n_in; p.K = H; p.accumulate = accumulate;
    p.vecA = aligned16(hidden) && (ldh % 4 == 0); p.vecB = aligned16(W1) && (ldw1 % 4 == 0);
    p.xf_dz2 = dz2; p.xf_lddz2 = lddz2; p.xf_W2 = W2; p.xf_ldw2 = ldw2; p.xf_nout = n_out; p.xf_act = act1;
    static const bool stream_ok = !ab_flag("NGP_MLP_NO_STREAM");
    if (stream_ok && H == 128 && n_in == 128 && !accumulate && p.vecA && aligned16(dx) && lddx % 4 == 0 &&
        (act1 == NGP_ACT_RELU || act1 == NGP_ACT_SOFTPLUS) && n <= (int64_t)32 * 0x7fffff00) {
        // streaming kernel (chosen by shape only, see ngp_mlp2_fwd); out-of-range xf_nout columns read as zero,
        // but the dz2 prefetch reads XF floats per row: stay inside the row
        hipStream_t st = (hipStream_t)stream;
        const int n_tiles = (int)((n + 31) / 32);
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return NGP_ELAUNCH;
            n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        static const int nw = (int)ab_long("NGP_MLP_NW_DGRAD", ab_long("NGP_MLP_NW", 8));
        static const int wg_rounds_d = (int)ab_long("NGP_MLP_WG_ROUNDS_DGRAD", ab_long("NGP_MLP_WG_ROUNDS", 1));
        const int cap_d = n_cu * wg_rounds_d;
        const int blocks = (n_tiles + nw - 1) / nw < cap_d ? (n_tiles + nw - 1) / nw : cap_d;
#define LAUNCH_SD(XFV, ACTV)                                                                                            \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        const size_t lds = (size_t)(128 * 132 + XFV * 128 + nw * 1024) * sizeof(float);                                 \
        if (!attr_set) {                                                                                                \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_stream_dgrad_kernel<XFV, ACTV>),                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)              \
                return NGP_ELAUNCH;                                                                                     \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((mlp_stream_dgrad_kernel<XFV, ACTV>), dim3(blocks), dim3(64 * nw), lds, st, p, n_tiles);    \
    } while (0)
#define LAUNCH_SD_A(XFV)                                                                                                \
    do {                                                                                                                \
        if (act1 == NGP_ACT_RELU) LAUNCH_SD(XFV, NGP_ACT_RELU);                                                         \
        else LAUNCH_SD(XFV, NGP_ACT_SOFTPLUS);                                                                          \
    } while (0)
        if (n_out == 1) LAUNCH_SD_A(1);
        else if (n_out == 2) LAUNCH_SD_A(2);
        else if (n_out == 3) LAUNCH_SD_A(3);
        else LAUNCH_SD_A(4);
#undef LAUNCH_SD_A
#undef LAUNCH_SD
        return ngp_check_launch();
    }
    if (n_out == 1) launch_dgrad<1>(p, (hipStream_t)stream);
    else if (n_out <= 3) launch_dgrad<3>(p, (hipStream_t)stream);
    else launch_dgrad<XF_OMAX>(p, (hipStream_t)stream);
    return ngp_check_launch();
}

int ngp_mlp_bwd_weight(const float* dz2, int64_t lddz2, const float* W2, int64_t ldw2, const float* hidden,
                       int64_t ldh, int act1, const float* x, int64_t ldx, int64_t n, int n_in, int H, int n_out,
                       float* dW1, int64_t ldw, float* db1, float* dW2, int64_t lddw2, float* db2, void* stream)
{
    if (n < 0 || n_in < 1 || n_out < 1 || n_out > XF_OMAX || H < 8 || ldx < n_in || ldw < n_in ||
        (dW2 && lddw2 < H) || (db2 && !dW2))
        return NGP_EINVAL;
    if (n == 0) return NGP_OK;   // empty batches carry null pointers
    if (!xf_args_ok(dz2, lddz2, W2, ldw2, hidden, ldh, H, n_out) || !x || !dW1) return NGP_EINVAL;
    GemmArgs p{};
    p.A = hidden; p.lda = ldh; p.B = x; p.ldb = ldx; p.C = dW1; p.ldc = ldw;
    p.M = H; p.N = n_in; p.K = n; p.bias_grad = db1;
    p.vecA = aligned16(hidden) && (ldh % 4 == 0); p.vecB = aligned16(x) && (ldx % 4 == 0);
    p.xf_dz2 = dz2; p.xf_lddz2 = lddz2; p.xf_W2 = W2; p.xf_ldw2 = ldw2; p.xf_nout = n_out; p.xf_act = act1;
    p.xf_dW2 = dW2; p.xf_lddw2 = lddw2; p.xf_db2 = db2;
#ifdef NGP_AB_VARIANTS
    static const int wgrad_exp = getenv("NGP_WGRAD_EXP") ? atoi(getenv("NGP_WGRAD_EXP")) : 0;
    p.exp = wgrad_exp;
#endif
    static const bool stream_ok = !ab_flag("NGP_MLP_NO_STREAM") && !ab_flag("NGP_MLP_NO_STREAM_WGRAD");
    if (stream_ok && H == 128 && (n_in == 128 || n_in == 144 || n_in == 160) && p.vecA && p.vecB &&
        (act1 == NGP_ACT_RELU || act1 == NGP_ACT_SOFTPLUS)) {
        hipStream_t st = (hipStream_t)stream;
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return NGP_ELAUNCH;
            n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        // The kernel takes workgroups of 8 waves (two groups meeting in LDS, half the atomics; alone the faster shape
        // for the 128-column case: 0.236 -> 0.225 ms) or of 4; a group's share of the chunk is a multiple of the 16
        // samples one turn of the prefetch ring covers
        // ONE 4-wave workgroup per CU (a wave per SIMD): alone the kernel is ~10 % slower than with 8 waves per CU
        // (0.38 -> 0.41-0.43 ms in the step), but it runs beside the gradient scatter, and what the scatter's waves
        // get of the CU is worth more to the step (-2.5 %, profiles/r03_occupancy_shaping.txt); NGP_WGRAD_WPC=9: the
        // round's earlier shape (8 waves per CU) for the A/B
        static const int wpc_env = (int)ab_long("NGP_WGRAD_WPC", 1);
        const int wpc = wpc_env == 9 ? 0 : wpc_env;
        const int threads = (n_in == 128 && !wpc) ? 512 : 256;
        int64_t blocks = (int64_t)n_cu * (wpc ? wpc : (threads == 512 ? 1 : 2));
        int64_t chunk = ((n + blocks - 1) / blocks + 31) / 32 * 32;
        if (chunk < 128) chunk = 128;
        {   // the kernel addresses a chunk with 32-bit byte offsets: keep (chunk + 32) rows of the widest operand below 2^31
            int64_t ldmax = ldx > ldh ? ldx : ldh;
            if (lddz2 > ldmax) ldmax = lddz2;
            const int64_t cap = (((int64_t)1 << 31) / (4 * ldmax) - 64) / 32 * 32;
            if (cap < 128) return NGP_EINVAL;
            if (chunk > cap) chunk = cap;
        }
        blocks = (n + chunk - 1) / chunk;
#define LAUNCH_SW2(XFV, ACTV, TNV, W2GV)                                                                                \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        const size_t lds = (size_t)4 * (threads == 512 ? (TNV * 16 + 2 * XFV + 1) * 64 : 2048) * sizeof(float);         \
        if (!attr_set) {                                                                                                \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_stream_wgrad_kernel<XFV, ACTV, TNV, W2GV>),      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)               \
                return NGP_ELAUNCH;                                                                                     \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((mlp_stream_wgrad_kernel<XFV, ACTV, TNV, W2GV>), dim3((unsigned)blocks), dim3(threads), lds, st, \
                           p, chunk);                                                                                   \
    } while (0)
#define LAUNCH_SW(XFV, ACTV, TNV)                                                                                       \
    do {                                                                                                                \
        if (dW2) LAUNCH_SW2(XFV, ACTV, TNV, true);                                                                      \
        else LAUNCH_SW2(XFV, ACTV, TNV, false);                                                                         \
    } while (0)
#define LAUNCH_SW_T(XFV, ACTV)                                                                                          \
    do {                                                                                                                \
        if (n_in == 128) LAUNCH_SW(XFV, ACTV, 4);                                                                       \
        else LAUNCH_SW(XFV, ACTV, 5);                                                                                   \
    } while (0)
#define LAUNCH_SW_A(XFV)                                                                                                \
    do {                                                                                                                \
        if (act1 == NGP_ACT_RELU) LAUNCH_SW_T(XFV, NGP_ACT_RELU);                                                       \
        else LAUNCH_SW_T(XFV, NGP_ACT_SOFTPLUS);                                                                        \
    } while (0)
        if (n_out == 1) LAUNCH_SW_A(1);
        else if (n_out == 2) LAUNCH_SW_A(2);
        else if (n_out == 3) LAUNCH_SW_A(3);
        else LAUNCH_SW_A(4);
#undef LAUNCH_SW_A
#undef LAUNCH_SW_T
#undef LAUNCH_SW
#undef LAUNCH_SW2
        return ngp_check_launch();
    }
    if (n_out == 1) launch_wgrad<1>(p, (hipStream_t)stream);
    else if (n_out <= 3) launch_wgrad<3>(p, (hipStream_t)stream);
    else launch_wgrad<XF_OMAX>(p, (hipStream_t)stream);
    return ngp_check_launch();
}

int ngp_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, const float* grad_scale, int zero_grad,
                  void* stream)
{
    return ngp_adam_step_width(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                               zero_grad, 0, stream);
}

int ngp_adam_step_width(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int64_t step, const float* grad_scale, int zero_grad,
                        int workgroups, void* stream)
{
    if (n < 0 || step < 1 || workgroups < 0) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return NGP_EINVAL;
    if (!aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq)) return NGP_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    // Default 2 workgroups per CU (a wave per SIMD each, 64 registers): 5.2 TB/s alone (2048 workgroups: 4.5) and the sweep ends
    // soonest.  ONE workgroup per CU lets the 8-wave MLP workgroups of the next step's density path in beside it (the sweep then
    // takes 1.5-1.7 instead of 1.1-1.4 ms, the colour chain follows alone).  Which is faster per step depends on the loop and
    // the box (+-3 %: profiles/r03_occupancy_shaping.txt (1), (9)), so the caller may name the width:
    // NGPTrainer times both in its own loop and keeps the faster one.
    static const int64_t cap_default = ab_long("NGP_ADAM_BLOCKS", 512);
    const int64_t cap = workgroups > 0 ? workgroups : cap_default;
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > cap) blocks = cap;
    static const bool sparse_zero = !ab_flag("NGP_ADAM_DENSE_ZERO");
    static const int unroll = (int)ab_long("NGP_ADAM_UNROLL", 2);
#define LAUNCH_ADAM(U)                                                                                                  \
    hipLaunchKernelGGL(adam_kernel<U>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, \
                       exp_avg_sq, n, step_size, beta1, beta2, eps, bc2_sqrt, weight_decay, grad_scale, zero_grad,      \
                       sparse_zero)
#ifdef NGP_AB_VARIANTS
    if (unroll == 4) LAUNCH_ADAM(4);
    else if (unroll == 3) LAUNCH_ADAM(3);
    else
#endif
        LAUNCH_ADAM(2);
    (void)unroll;
#undef LAUNCH_ADAM
    return ngp_check_launch();
}

int ngp_sumsq(const float* x, int64_t n, float* out, void* stream)
{
    return ngp_sumsq_if(x, n, out, nullptr, stream);
}

int ngp_sumsq_if(const float* x, int64_t n, float* out, const int32_t* flag, void* stream)
{
    if (n < 0 || !out) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!x) return NGP_EINVAL;
    if (aligned16(x)) {
        int64_t blocks = ((n >> 2) + 255) / 256;
        if (blocks < 1) blocks = 1;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, out, flag);
    } else {
        int64_t blocks = (n + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(sumsq_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, out, flag);
    }
    return ngp_check_launch();
}

int ngp_clip_coef(const float* sumsq, float max_norm, float extra_scale, float* coef, void* stream)
{
    if (!sumsq || !coef) return NGP_EINVAL;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sumsq, max_norm, extra_scale, coef);
    return ngp_check_launch();
}

int ngp_row_norm_sum(const float* x, int64_t ldx, int64_t n, int cols, float* out, void* stream)
{
    if (n < 0 || cols < 1 || cols > 16 || ldx < cols || !out) return NGP_EINVAL;
    if (n == 0) return NGP_OK;
    if (!x) return NGP_EINVAL;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(row_norm_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, n, cols, out);
    return ngp_check_launch();
}

int ngp_clip_decide(const float* row_norm_sums, const float* w1_a, int64_t n1_a, const float* w2_a, int64_t n2_a,
                    const float* w1_b, int64_t n1_b, const float* w2_b, int64_t n2_b, const float* sumsq_rest,
                    float max_norm, float extra_scale, float* coef, int32_t* need_exact, void* stream)
{
    if (!row_norm_sums || !w1_a || !w2_a || !w1_b || !w2_b || n1_a < 1 || n2_a < 1 || n1_b < 1 || n2_b < 1 || !coef ||
        !need_exact)
        return NGP_EINVAL;
    hipLaunchKernelGGL(clip_decide_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_norm_sums, w1_a, n1_a, w2_a,
                       n2_a, w1_b, n1_b, w2_b, n2_b, sumsq_rest, max_norm, extra_scale, coef, need_exact,
                       (const float*)nullptr, (int64_t)0, (float*)nullptr);
    return ngp_check_launch();
}

int ngp_clip_decide_rest(const float* row_norm_sums, const float* w1_a, int64_t n1_a, const float* w2_a, int64_t n2_a,
                         const float* w1_b, int64_t n1_b, const float* w2_b, int64_t n2_b, const float* rest,
                         int64_t n_rest, float* sumsq, float max_norm, float extra_scale, float* coef,
                         int32_t* need_exact, void* stream)
{
    if (!row_norm_sums || !w1_a || !w2_a || !w1_b || !w2_b || n1_a < 1 || n2_a < 1 || n1_b < 1 || n2_b < 1 || !coef ||
        !need_exact || !rest || n_rest < 0 || !sumsq)
        return NGP_EINVAL;
    hipLaunchKernelGGL(clip_decide_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_norm_sums, w1_a, n1_a, w2_a,
                       n2_a, w1_b, n1_b, w2_b, n2_b, (const float*)sumsq, max_norm, extra_scale, coef, need_exact, rest,
                       n_rest, sumsq);
    return ngp_check_launch();
}

int ngp_clip_coef_if(const float* sumsq, float max_norm, float extra_scale, float* coef, const int32_t* flag,
                     void* stream)
{
    if (!sumsq || !coef || !flag) return NGP_EINVAL;
    hipLaunchKernelGGL(clip_coef_if_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sumsq, max_norm, extra_scale,
                       coef, flag);
    return ngp_check_launch();
}

const char* ngp_version(void) { return "ngp_hip 0.2 gfx950"; }

} // extern "C"
