// MFMA GEMM with fused operand prologue / epilogue for gfx950.
//
//   C[M,N] = act( A'[M,K] . W[N,K]^T * exp(*exp_scale) + bias ) (+ R)      (see include/cvcl_hip.h)
//
// Replaces the implicit ATen/cuDNN calls behind: torchvision Bottleneck 1x1 convolutions (reference
// call site multimodal/multimodal.py:101), nn.Linear fc/head (:190-192), the ViT linears
// (multimodal/vision_transformer_dino_mugs.py:92-94,113-115) and image_features @ text_features.T
// (multimodal/multimodal.py:755).
//
// Design (CDNA4):
//  * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA 32x32
//    tiles); K tile = 128 bytes per row (64 bf16 / 32 fp32) staged through LDS with a 144-byte row
//    pitch (9 x 16-B slots: ds_read_b128 fragment reads of 16 different rows hit 16 different slots).
//  * bf16 storage: v_mfma_f32_32x32x16_bf16; fp32 storage: v_mfma_f32_32x32x2_f32 (exact fp32, the
//    parity mode).  Operands are "swapped" (MFMA-A = weight rows, MFMA-B = activation rows) so each
//    lane ends up with 4 consecutive output channels of one output row.
//  * software pipeline over the flattened (M tile, K tile) sequence: while a tile is multiplied out of
//    LDS, the next tile's global loads are already in flight in registers -- also across M-tile
//    boundaries, so the epilogue of one tile overlaps the loads of the next.
//  * workgroups are persistent over M tiles (grid.x = grid_m <= what is co-resident, grid.y = N tiles):
//    the per-channel BatchNorm statistics of the output are accumulated in registers across tiles and
//    written once per workgroup as a partial row (deterministic, no atomics).  Workgroups (i, j) and
//    (i, j+1) have linear ids i and i+grid_m (grid_m % 8 == 0) -> same XCD -> the A tile they share is
//    an L2 hit.
//  * the epilogue goes through LDS so that C is written as full 128-byte row segments (16 B per lane).
//  * the BatchNorm(+ReLU) of the *producer* layer is applied to A while it is staged (per-K scale /
//    shift), so a normalised activation tensor is never written to HBM.
#include <cstdlib>
#include <type_traits>

#include "cvcl_common.h"

namespace {

struct GemmDev {
    const void* A; const void* W; void* C;
    int M, N, K, lda, ldw, ldc;
    const float* a_scale; const float* a_shift; int a_relu;
    int g_ho, g_wo, g_hi, g_wi, g_s;
    const float* exp_scale; const float* bias; int act;
    const void* R; int ldr;
    const float* c_scale; const float* c_shift; const float* r_scale; const float* r_shift;   // BN-apply epilogue
    float* stats;
    int stats_acc;         // stats is an int64 accumulator [8][2][N] (CVCL_STATS_ACCUMULATE), not partial rows
    const float* centre;   // storage centre of a raw convolution output (NULL = 0): accumulators start at -centre[n]
    int vec_in;    // A/W rows are 16-byte aligned and K is a whole number of chunks
    int vec_out;   // C/R rows are 16-byte aligned
    int num_m_tiles;
    void* C2;      // EPI 5: second output (the GELU pre-activation)
    float* a_rowsum;   // TR & 1 (fp32): a_rowsum[m] = sum_k A'[m][k], written by the workgroups of column tile 0
};

template <typename T> struct FragOps;
template <> struct FragOps<bf16_t> {
    using Frag = bf16x8;
    __device__ static inline void mma(f32x16& acc, const Frag& w, const Frag& a) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, acc, 0, 0, 0);
    }
};
template <> struct FragOps<float> {
    using Frag = f32x4;
    // The contraction index may be permuted freely as long as both operands agree: lane-half h owns
    // k = 4*(2g+h)..+3 of the 8-wide group and feeds element s to the s-th 32x32x2 step.
    __device__ static inline void mma(f32x16& acc, const Frag& w, const Frag& a) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s], a[s], acc, 0, 0, 0);
    }
};

__device__ inline float apply_act(float v, int act) {
    if (act == CVCL_ACT_RELU) return fmaxf(v, 0.f);
    if (act == CVCL_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    return v;
}

// bf16 epilogues: gelu_bf16out (cvcl_common.h: 9 VALU instructions on v_exp_f32 / v_rcp_f32, |error| <= 2.6e-5): the library
// erff costs ~5x the instructions (fc1 + GELU of ViT-B: 424 us vs 330 us without the activation).  The fp32 parity kernels keep erff.
__device__ inline float apply_act_bf16(float v, int act) {
    if (act == CVCL_ACT_RELU) return fmaxf(v, 0.f);
    if (act == CVCL_ACT_GELU) return gelu_bf16out(v);
    return v;
}

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 144;   // LDS row pitch in bytes (128 B of K + 16 B pad)

template <typename T> constexpr int stage_rowb() { return (64 + (sizeof(T) == 2 ? 8 : 4)) * (int)sizeof(T); }
template <typename T> constexpr int gemm_lds_bytes() {
    return (BM + BN) * ROWB > 4 * 64 * stage_rowb<T>() ? (BM + BN) * ROWB : 4 * 64 * stage_rowb<T>();
}

// Operand tile held in registers between its global load and its LDS write (4 x 16 B per operand per thread).
template <typename T> struct TileRegs {
    Chunk<T> a[4], w[4];
};

// PRO: 0 = plain A, 1 = A*scale+shift, 2 = relu(A*scale+shift).
// LEAN: the convolution fast path -- 16-byte aligned operands, N a multiple of 128, no bias / activation /
// output scale / residual: the epilogue carries no per-element predicates.  !LEAN handles everything.
// TR (fp32 tail GEMMs, round 5): bit 0 = the A operand is given K-major (element (m, k) at A[k * lda + m]), bit 1 = the same for
// W -- the gradient GEMMs of nn.Linear (dW = dY^T X, dX = dY W) and of the similarity logits read their operands as they lie,
// no transposed copies.  A K-major tile is fetched as 32 k-rows x 128 contiguous elements (full 128-byte segments per k-row) and
// scattered into the same [row][k] LDS image the fragment reads expect (ds_write_b32, 2-way banked = free).
// TR bit 2 (round 5) = SPLIT arithmetic for fp32 operands: every element is split into two bf16 parts on its way into LDS
// (hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits between them) and the product is formed as hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_bf16 -- three MFMAs at 16x the fp32 instruction's rate, fp32 accumulation, products good to ~2^-16 relative
// (the dropped lo.lo term and the split's remainder).  The trainable tail's linears in the bf16 configurations (text transformer,
// reference multimodal/multimodal.py:553-573: under Lightning's bf16 autocast these are plain bf16 GEMMs; here they keep fp32
// operands and ~fp32 results at a quarter of the exact kernel's time).  The exact mode stays the parity mode.
// hi / lo parts of the split: hi = bf16(x), lo = bf16(x - hi).  A FINITE x beyond the bf16 range (3.39e38 < |x| <= fp32 max) would
// round hi to infinity -- x - inf makes lo = -inf, and inf times the OTHER operand's lo part (a small number of either sign) gives
// -inf or NaN next to the hi.hi product's +inf: NaN where the exact fp32 GEMM gives a finite number or inf.  hi is clamped to the
// largest finite bf16 there (lo = x - hi is then ~1e36, representable), so the product stays what the operands make it; an inf / NaN
// operand keeps hi = inf / NaN with lo = 0 (its products are non-finite in either arithmetic).
// (selects only: a branch in the staging code of a kernel with loads in flight makes the compiler drain them)
__device__ __forceinline__ void split_hi_lo(float x, bf16_t& hi, bf16_t& lo) {
    const bf16_t h0 = (bf16_t)x;
    const bool hfin = __builtin_isfinite((float)h0), xfin = __builtin_isfinite(x);
    const bf16_t hmax = __builtin_bit_cast(bf16_t, (unsigned short)(x < 0.f ? 0xFF7F : 0x7F7F));              // +-3.3895e38
    const bf16_t h = (!hfin && xfin) ? hmax : h0;
    hi = h;
    lo = (bf16_t)(xfin ? x - (float)h : 0.f);
}

template <typename T, int PRO, bool LEAN, int MINW, int TR = 0>
__global__ __launch_bounds__(256, MINW) void gemm_kernel(GemmDev p) {
    static_assert(TR == 0 || (sizeof(T) == 4 && PRO == 0 && !LEAN), "K-major operands / split arithmetic: fp32, no prologue");
    constexpr bool SPLIT = (TR & 4) != 0;
    constexpr int EPC = ElemTraits<T>::kPerChunk;   // elements per 16-B chunk
    constexpr int BK = 8 * EPC;
    constexpr int SROW = stage_rowb<T>();
    using Frag = typename FragOps<T>::Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sW = smem + BM * ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ W = (const T*)p.W;
    T* __restrict__ C = (T*)p.C;
    const T* __restrict__ R = (const T*)p.R;

    const int kc = tid & 7, r0 = tid >> 3;          // staging role: chunk kc of rows r0 + 32 j
                                                    // (K-major operand: k-rows kc + 8 j, elements 4 r0 .. 4 r0 + 3)
    const float out_scale = (!LEAN && p.exp_scale) ? expf(*p.exp_scale) : 1.f;
    const int ktiles = (p.K + BK - 1) / BK;

    float st_sum[8], st_sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { st_sum[e] = 0.f; st_sq[e] = 0.f; }

    // ---- load side of the pipeline: position (l_mt, l_kt) of the tile whose loads are in flight ----------
    int l_mt = blockIdx.x, l_kt = 0;
    unsigned a_off[4];                              // element offsets (every operand here is < 2^31 elements)
    bool a_ok[4];
    unsigned w_off[4];
    bool w_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + r0 + 32 * j;
        w_ok[j] = n < p.N;
        w_off[j] = (unsigned)n * (unsigned)p.ldw;
    }
    auto set_rows = [&](int mt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mt * BM + r0 + 32 * j;
            a_ok[j] = m < p.M;
            long row = m;
            if (p.g_s > 1 && a_ok[j]) {
                const int hw = p.g_ho * p.g_wo;
                const int b = m / hw, r = m - b * hw;
                const int oy = r / p.g_wo, ox = r - oy * p.g_wo;
                row = ((long)b * p.g_hi + (long)oy * p.g_s) * p.g_wi + (long)ox * p.g_s;
            }
            a_off[j] = (unsigned)(row * p.lda);
        }
    };
    TileRegs<T> t;
    float sc[EPC], sh[EPC];
    auto issue = [&]() {                            // global -> registers for tile (l_mt, l_kt); no waiting here
        const int k = l_kt * BK + kc * EPC;
        // K-major operand: 4 consecutive rows (m | n) of k-row kk per 16-byte load; row-major operand: 4 | 8 consecutive k of a row
        auto load_kmajor = [&](Chunk<T>& dst, const T* base, int ld, int kk, int r, int rows) __attribute__((always_inline)) {
            if (kk < p.K && r + 3 < rows && p.vec_in) dst.load(base + (long)kk * ld + r);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) dst.set(e, (kk < p.K && r + e < rows) ? ElemTraits<T>::to_f(base[(long)kk * ld + r + e]) : 0.f);
            }
        };
        if constexpr ((TR & 1) != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) load_kmajor(t.a[j], A, p.lda, l_kt * BK + kc + 8 * j, l_mt * BM + r0 * 4, p.M);
        }
        if constexpr ((TR & 2) != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) load_kmajor(t.w[j], W, p.ldw, l_kt * BK + kc + 8 * j, n0 + r0 * 4, p.N);
        }
        if (LEAN || p.vec_in) {
            const bool k_ok = k < p.K;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr ((TR & 1) == 0) { if (a_ok[j] && k_ok) t.a[j].load(A + a_off[j] + k); else t.a[j].zero(); }
                if constexpr ((TR & 2) == 0) { if (w_ok[j] && k_ok) t.w[j].load(W + w_off[j] + k); else t.w[j].zero(); }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const bool ok = (k + e) < p.K;
                    if constexpr ((TR & 1) == 0) t.a[j].set(e, (a_ok[j] && ok) ? ElemTraits<T>::to_f(A[a_off[j] + k + e]) : 0.f);
                    if constexpr ((TR & 2) == 0) t.w[j].set(e, (w_ok[j] && ok) ? ElemTraits<T>::to_f(W[w_off[j] + k + e]) : 0.f);
                }
        }
        if constexpr (PRO != 0) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const bool ok = (k + e) < p.K;
                sc[e] = ok ? p.a_scale[k + e] : 0.f;
                sh[e] = ok ? p.a_shift[k + e] : 0.f;
            }
        }
    };
    bool l_live = l_mt < p.num_m_tiles;
    if (l_live) { set_rows(l_mt); issue(); }

    // centred storage (convolution epilogues only): the accumulators of column n start at -centre[n]; this lane's columns are
    // n0 + wn*64 + nt*32 + 8g + 4h + e
    float cinit[2][16];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = n0 + wn * 64 + nt * 32 + 8 * (e >> 2) + 4 * h + (e & 3);
            cinit[nt][e] = (p.centre && n < p.N) ? -p.centre[n] : 0.f;
        }

    for (int cm = blockIdx.x; cm < p.num_m_tiles; cm += gridDim.x) {
        const int m0 = cm * BM;
        [[maybe_unused]] float rsum[4] = {0.f, 0.f, 0.f, 0.f};       // K-major A: this thread's share of sum_k A'[m0 + 4 r0 + e][k]
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = cinit[i][e];

        for (int kt = 0; kt < ktiles; ++kt) {
            // registers hold tile (cm, kt): BatchNorm(+ReLU) of the producer on the fly, then into LDS
            if constexpr (PRO != 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        float v = fmaf(t.a[j].get(e), sc[e], sh[e]);
                        if constexpr (PRO == 2) v = fmaxf(v, 0.f);
                        t.a[j].set(e, v);
                    }
            }
            __syncthreads();                      // LDS free: previous tile consumed / epilogue staging drained
            if constexpr (SPLIT) {
                // LDS row image (128 of the 144 bytes): [hi of k 0..31 as bf16 | lo of k 0..31 as bf16]
                auto split_store = [&](char* base, const Chunk<T>& c, bool kmajor, int j) __attribute__((always_inline)) {
                    bf16_t hi[4], lo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = c.get(e);
                        split_hi_lo(x, hi[e], lo[e]);
                    }
                    if (!kmajor) {
                        char* row = base + (r0 + 32 * j) * ROWB + kc * 8;
                        *reinterpret_cast<bf16x4*>(row) = bf16x4{hi[0], hi[1], hi[2], hi[3]};
                        *reinterpret_cast<bf16x4*>(row + 64) = bf16x4{lo[0], lo[1], lo[2], lo[3]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            char* el = base + (r0 * 4 + e) * ROWB + (kc + 8 * j) * 2;
                            *reinterpret_cast<bf16_t*>(el) = hi[e];
                            *reinterpret_cast<bf16_t*>(el + 64) = lo[e];
                        }
                    }
                };
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    split_store(sA, t.a[j], (TR & 1) != 0, j);
                    split_store(sW, t.w[j], (TR & 2) != 0, j);
                    if constexpr ((TR & 1) != 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) rsum[e] += t.a[j].get(e);
                    }
                }
            } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr ((TR & 1) == 0) t.a[j].store((T*)(sA + (r0 + 32 * j) * ROWB + kc * 16));
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        *(float*)(sA + (r0 * 4 + e) * ROWB + (kc + 8 * j) * 4) = t.a[j].get(e);
                        rsum[e] += t.a[j].get(e);                  // (dead unless a_rowsum is asked for: see below)
                    }
                }
                if constexpr ((TR & 2) == 0) t.w[j].store((T*)(sW + (r0 + 32 * j) * ROWB + kc * 16));
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) *(float*)(sW + (r0 * 4 + e) * ROWB + (kc + 8 * j) * 4) = t.w[j].get(e);
                }
            }
            }
            __syncthreads();
            // advance the load position and put the next tile's loads in flight under the MFMAs (and the epilogue)
            if (++l_kt == ktiles) {
                l_kt = 0;
                l_mt += gridDim.x;
                l_live = l_mt < p.num_m_tiles;
                if (l_live) set_rows(l_mt);
            }
            if (l_live) issue();
            if constexpr (SPLIT) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {                   // 16 k per step: chunk g2 * 2 + h of the hi / lo half-rows
                    bf16x8 wh[2], wl[2], ah[2], al[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const char* wr = sW + (wn * 64 + q * 32 + l31) * ROWB + (g2 * 2 + h) * 16;
                        const char* ar = sA + (wm * 64 + q * 32 + l31) * ROWB + (g2 * 2 + h) * 16;
                        wh[q] = *reinterpret_cast<const bf16x8*>(wr); wl[q] = *reinterpret_cast<const bf16x8*>(wr + 64);
                        ah[q] = *reinterpret_cast<const bf16x8*>(ar); al[q] = *reinterpret_cast<const bf16x8*>(ar + 64);
                    }
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {           // small terms first
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[nt], ah[mt], acc[nt][mt], 0, 0, 0);
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nt], al[mt], acc[nt][mt], 0, 0, 0);
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nt], ah[mt], acc[nt][mt], 0, 0, 0);
                        }
                }
            } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                Frag fw[2], fa[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    fw[q] = *reinterpret_cast<const Frag*>(sW + (wn * 64 + q * 32 + l31) * ROWB + (g * 2 + h) * 16);
                    fa[q] = *reinterpret_cast<const Frag*>(sA + (wm * 64 + q * 32 + l31) * ROWB + (g * 2 + h) * 16);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) FragOps<T>::mma(acc[nt][mt], fw[nt], fa[mt]);
            }
            }
        }

        if constexpr ((TR & 1) != 0) {
            // bias gradient of nn.Linear fused into its weight-gradient GEMM: row sums of A' = dY^T, i.e. column sums of dY.  The
            // eight threads kc = 0..7 of a row group are adjacent lanes: fixed-order xor tree, deterministic
            if (p.a_rowsum && blockIdx.y == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = rsum[e];
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
                    if (kc == 0 && m0 + r0 * 4 + e < p.M) p.a_rowsum[m0 + r0 * 4 + e] = v;
                }
            }
        }
        // ---- epilogue: registers -> (scale, bias, act, round) -> LDS -> full-row stores ----------
        __syncthreads();                          // every wave is done reading the operand tiles
        char* stg = smem + wave * 64 * SROW;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n_local = nt * 32 + 8 * g + 4 * h;
                    const int n_glob = n0 + wn * 64 + n_local;
                    T q[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (LEAN) {
                            q[e] = ElemTraits<T>::from_f(acc[nt][mt][4 * g + e]);
                        } else {
                            float v = acc[nt][mt][4 * g + e] * out_scale;
                            if (p.bias && n_glob + e < p.N) v += p.bias[n_glob + e];
                            q[e] = ElemTraits<T>::from_f(apply_act(v, p.act));
                        }
                    }
                    T* dst = (T*)(stg + (mt * 32 + l31) * SROW) + n_local;
                    if constexpr (sizeof(T) == 2) {
                        *reinterpret_cast<bf16x4*>(dst) = bf16x4{q[0], q[1], q[2], q[3]};
                    } else {
                        *reinterpret_cast<f32x4*>(dst) = f32x4{q[0], q[1], q[2], q[3]};
                    }
                }
        // (wave-private staging region: LDS ops of one wave complete in order, no barrier needed)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = j * 8 + (lane >> 3), colc = (lane & 7) * 8;
            const int m = m0 + wm * 64 + row, n = n0 + wn * 64 + colc;
            const T* src = (const T*)(stg + row * SROW) + colc;
            float v[8];
            if constexpr (sizeof(T) == 2) {
                bf16x8 tt = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)tt[e];
            } else {
                f32x4 t0 = *reinterpret_cast<const f32x4*>(src), t1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = t0[e]; v[4 + e] = t1[e]; }
            }
            if constexpr (LEAN) {
                if (m < p.M) {
                    // v[] already holds the values as stored (converted from the staged T)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        st_sum[e] += v[e];
                        st_sq[e] = fmaf(v[e], v[e], st_sq[e]);
                    }
                    T* dst = C + (long)m * p.ldc + n;
                    if constexpr (sizeof(T) == 2) {
                        *reinterpret_cast<bf16x8*>(dst) = *reinterpret_cast<const bf16x8*>(src);
                    } else {
                        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    }
                }
            } else
            if (m < p.M && n < p.N) {
                const bool full = (n + 8 <= p.N) && p.vec_out;
                if (R) {
                    if (full) {
                        if constexpr (sizeof(T) == 2) {
                            bf16x8 tt = *reinterpret_cast<const bf16x8*>(R + (long)m * p.ldr + n);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += (float)tt[e];
                        } else {
                            f32x4 t0 = *reinterpret_cast<const f32x4*>(R + (long)m * p.ldr + n);
                            f32x4 t1 = *reinterpret_cast<const f32x4*>(R + (long)m * p.ldr + n + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] += t0[e]; v[4 + e] += t1[e]; }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (n + e < p.N) v[e] += ElemTraits<T>::to_f(R[(long)m * p.ldr + n + e]);
                    }
                }
                T o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = ElemTraits<T>::from_f(v[e]);
                    if (n + e < p.N) {                       // statistics of the tensor as stored
                        const float sv = ElemTraits<T>::to_f(o[e]);
                        st_sum[e] += sv;
                        st_sq[e] = fmaf(sv, sv, st_sq[e]);
                    }
                }
                T* dst = C + (long)m * p.ldc + n;
                if (full) {
                    if constexpr (sizeof(T) == 2) {
                        *reinterpret_cast<bf16x8*>(dst) = bf16x8{o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
                    } else {
                        *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
                        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < p.N) dst[e] = o[e];
                }
            }
        }
    }

    if (p.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = 8; o <= 32; o <<= 1) {
                st_sum[e] += __shfl_xor(st_sum[e], o, 64);
                st_sq[e] += __shfl_xor(st_sq[e], o, 64);
            }
        }
        __syncthreads();
        float* red = (float*)smem;                  // [4 waves][2][64]
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 2 + 0) * 64 + lane * 8 + e] = st_sum[e];
                red[(wave * 2 + 1) * 64 + lane * 8 + e] = st_sq[e];
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int wn_ = tid >> 6, c = tid & 63, n = n0 + tid;
            if (n < p.N) {
                const float s = red[((0 + 2 * wn_) * 2 + 0) * 64 + c] + red[((1 + 2 * wn_) * 2 + 0) * 64 + c];
                const float q = red[((0 + 2 * wn_) * 2 + 1) * 64 + c] + red[((1 + 2 * wn_) * 2 + 1) * 64 + c];
                cvcl_bn_stats_out(p.stats, p.stats_acc, blockIdx.x, p.N, n, s, q);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Direct-to-LDS variant for the convolution fast path (bf16, no operand prologue, K % 64 == 0, N % 128 == 0).
//
// Both operand tiles are fetched with global_load_lds_dwordx4 (no VGPR round trip, no ds_write): each wave
// instruction lands 64 x 16 B = 8 consecutive 128-byte rows.  The destination must be lane-linear, so LDS rows are
// unpadded and bank conflicts are avoided by an XOR swizzle applied on the *source* address: the 16-byte chunk c of
// row r lives at chunk position c ^ ((r >> 1) & 7), which makes the 16 rows touched by a ds_read_b128 lane group
// hit 16 different 16-byte slots.  Two LDS buffers; per K tile: wait for this wave's loads, one barrier, issue the
// next tile's loads into the other buffer, multiply the current one.  The epilogue stages C through the buffer
// that was just consumed (same swizzle idea) and writes full 128-byte rows.
// ------------------------------------------------------------------------------------------------
constexpr int GL_OPER = 128 * 128;          // bytes of one operand tile (128 rows x 64 bf16)
constexpr int GL_BUF = 2 * GL_OPER;         // A tile + W tile
constexpr int GL_LDS = 2 * GL_BUF;          // double buffered: 64 KiB

__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// EPI = 0: convolution epilogue (round, BN partial sums; C may be NULL = statistics only).
// EPI = 1 / 3 / 4: + bias, activation (none / ReLU / GELU), residual (the ViT / nn.Linear epilogue:
//          out = round(round(act(acc + bias)) + R)).
// EPI = 2: Bottleneck tail: out = relu(round(acc) * c_scale[n] + c_shift[n] + (R | R * r_scale[n] + r_shift[n])) --
//          BatchNorm of this conv's (rounded) output + identity / normalised downsample branch + ReLU, no statistics.
// EPI = 5: linear + GELU for training: C2 = u = round(acc + bias) (kept for the backward), C = round(gelu(u)).
// EPI = 6: data-gradient GEMM through a GELU: C = round(round(acc) * gelu'(R)), R = the saved pre-activation tile.
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(GemmDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // EPI 1 / 3 / 4 = linear epilogue with activation none / ReLU / GELU compiled in (a run-time switch per element cost
    // two scalar branches per value and inlined erff 64 times behind them)
    constexpr bool LIN = EPI == 1 || EPI == 3 || EPI == 4 || EPI == 5 || EPI == 6;
    constexpr int ACT = EPI == 3 ? CVCL_ACT_RELU : (EPI == 4 ? CVCL_ACT_GELU : CVCL_ACT_NONE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const bf16_t* __restrict__ A = (const bf16_t*)p.A;
    const bf16_t* __restrict__ W = (const bf16_t*)p.W;
    bf16_t* __restrict__ C = (bf16_t*)p.C;
    const bf16_t* __restrict__ R = (const bf16_t*)p.R;
    const int ktiles = p.K / 64;

    // staging role: wave w issues instructions j = 0..3 per operand; instruction (w, j) covers tile rows
    // (4w + j) * 8 .. + 7; this lane supplies row (4w + j) * 8 + lane / 8, chunk position lane % 8
    int s_row[4];
    unsigned w_off[4], a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);          // logical chunk that belongs at this position
        s_row[j] = r;
        w_off[j] = (unsigned)(n0 + r) * (unsigned)p.ldw + c * 8;
    }
    auto set_rows = [&](int mt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = mt * BM + s_row[j];
            if (m >= p.M) m = p.M - 1;                         // tail rows: any valid row, masked at the store
            long row = m;
            if (p.g_s > 1) {
                const int hw = p.g_ho * p.g_wo;
                const int b = m / hw, r = m - b * hw;
                const int oy = r / p.g_wo, ox = r - oy * p.g_wo;
                row = ((long)b * p.g_hi + (long)oy * p.g_s) * p.g_wi + (long)ox * p.g_s;
            }
            const int c = (lane & 7) ^ ((s_row[j] >> 1) & 7);
            a_off[j] = (unsigned)(row * p.lda) + c * 8;
        }
    };
    auto issue = [&](int buf, int kt) {
        char* base = smem + buf * GL_BUF + wave * 4 * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(A + a_off[j] + kt * 64, base + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(W + w_off[j] + kt * 64, base + GL_OPER + j * 1024);
    };

    float st_sum[8], st_sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { st_sum[e] = 0.f; st_sq[e] = 0.f; }
    // EPI 2: this lane always handles the same 8 output channels in the read-back phase: load their affines once
    float cs[8], cb[8], rs[8], rb[8];
    if constexpr (EPI == 2) {
        const int nn = n0 + wn * 64 + (lane & 7) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cs[e] = p.c_scale[nn + e]; cb[e] = p.c_shift[nn + e];
            rs[e] = p.r_scale ? p.r_scale[nn + e] : 1.f; rb[e] = p.r_scale ? p.r_shift[nn + e] : 0.f;
        }
    }

    // EPI 1: the column tile is fixed per workgroup, so the bias values this lane adds (n = n0 + wn*64 + nt*32 + 8g + 4h + e)
    // are loaded once.  Loading them per tile put global loads behind the next tile's in-flight global_load_lds in the
    // in-order vmcnt queue: every tile's epilogue then waited for a full operand fetch (qkv GEMM of ViT-B: 357 vs 292 us;
    // same reason the residual rows are fetched before the K loop).
    f32x4 bias_r[2][4];
    if constexpr (LIN) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bias_r[nt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias) bias_r[nt][g] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 64 + nt * 32 + 8 * g + 4 * h);
            }
    }

    // centred storage (EPI 0 / 2): the accumulators of column n start at -centre[n] (this lane's 32 columns, fixed per workgroup)
    f32x4 cinit[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            cinit[nt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == 0 || EPI == 2) {
                if (p.centre) {
                    const f32x4 cv = *reinterpret_cast<const f32x4*>(p.centre + n0 + wn * 64 + nt * 32 + 8 * g + 4 * h);
                    cinit[nt][g] = f32x4{-cv[0], -cv[1], -cv[2], -cv[3]};
                }
            }
        }

    // fragment read offsets inside a buffer (rows fixed per lane, swizzle per row)
    int fw_off[2], fa_off[2], fw_sw[2], fa_sw[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rw = wn * 64 + q * 32 + l31, ra = wm * 64 + q * 32 + l31;
        fw_off[q] = GL_OPER + rw * 128; fw_sw[q] = (rw >> 1) & 7;
        fa_off[q] = ra * 128;           fa_sw[q] = (ra >> 1) & 7;
    }

    int l_mt = blockIdx.x, l_kt = 0, buf = 0;
    bool l_live = l_mt < p.num_m_tiles;
    if (l_live) { set_rows(l_mt); issue(0, 0); }

    // true when the previous output tile of this workgroup issued exactly 8 stores per wave (all 128 rows inside M, C given)
    bool counted_wait = false;
    for (int cm = blockIdx.x; cm < p.num_m_tiles; cm += gridDim.x) {
        const int m0 = cm * BM;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = cinit[i][e >> 2][e & 3];

        // EPI 2: the residual rows this lane will need in the read-back phase are fetched now, so that their latency
        // hides behind the K loop instead of sitting between the last MFMA and the stores
        bf16x8 rpre[8];
        if (EPI == 2 || (LIN && R)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int m = m0 + wm * 64 + j * 8 + (lane >> 3);
                if (m >= p.M) m = p.M - 1;
                rpre[j] = *reinterpret_cast<const bf16x8*>(R + (long)m * p.ldr + n0 + wn * 64 + (lane & 7) * 8);
            }
        }

        for (int kt = 0; kt < ktiles; ++kt) {
            // this wave's part of the current K tile has landed.  vmcnt retires in issue order (loads and stores alike
            // on gfx9-family parts), so at the first K tile of an output tile only the operand loads -- issued BEFORE
            // the previous tile's 8 epilogue stores (and this tile's 8 residual-row loads) -- have to be complete:
            // a plain vmcnt(0) also waited for the store acknowledgements of the previous tile, once per output tile.
            if (kt == 0 && counted_wait) {
                if (EPI == 2 || (LIN && R)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                        // ... and everybody's; buffer buf^1 is free again
            if (++l_kt == ktiles) {
                l_kt = 0;
                l_mt += gridDim.x;
                l_live = l_mt < p.num_m_tiles;
                if (l_live) set_rows(l_mt);
            }
            if (l_live) issue(buf ^ 1, l_kt);
            const char* cur = smem + buf * GL_BUF;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = g * 2 + h;
                bf16x8 fw[2], fa[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    fw[q] = *reinterpret_cast<const bf16x8*>(cur + fw_off[q] + ((c ^ fw_sw[q]) << 4));
                    fa[q] = *reinterpret_cast<const bf16x8*>(cur + fa_off[q] + ((c ^ fa_sw[q]) << 4));
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
            }
            buf ^= 1;
        }

        // ---- epilogue: stage C through the buffer just consumed (buf ^ 1), 8 KiB per wave, rows of 128 B with the
        // 16-byte chunks XOR-swizzled by the row; then full-row 16-byte stores + BN partial sums
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // every wave finished reading that buffer
        char* stg = smem + (buf ^ 1) * GL_BUF + wave * 8192;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = mt * 32 + l31;
                    const int chunk = nt * 4 + g;                // 8 channels per 16-B chunk; this lane owns half h
                    bf16x4 q;
                    if constexpr (LIN) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) q[e] = (bf16_t)apply_act_bf16(acc[nt][mt][4 * g + e] + bias_r[nt][g][e], ACT);
                    } else {
                        q = bf16x4{(bf16_t)acc[nt][mt][4 * g + 0], (bf16_t)acc[nt][mt][4 * g + 1],
                                   (bf16_t)acc[nt][mt][4 * g + 2], (bf16_t)acc[nt][mt][4 * g + 3]};
                    }
                    *reinterpret_cast<bf16x4*>(stg + row * 128 + ((chunk ^ (row & 7)) << 4) + h * 8) = q;
                }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = j * 8 + (lane >> 3), chunk = lane & 7;
            const int m = m0 + wm * 64 + row, n = n0 + wn * 64 + chunk * 8;
            bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((chunk ^ (row & 7)) << 4));
            if (m < p.M) {
                if constexpr (EPI == 5) {
                    *reinterpret_cast<bf16x8*>((bf16_t*)p.C2 + (long)m * p.ldc + n) = v;          // the pre-activation, as stored
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)gelu_bf16out((float)v[e]);
                } else if constexpr (EPI == 6) {
                    const bf16x8 r = rpre[j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] * gelu_grad_fast((float)r[e]));
                } else if constexpr (LIN) {
                    if (R) {
                        const bf16x8 r = rpre[j];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)r[e]);
                    }
                } else if constexpr (EPI == 2) {
                    const bf16x8 r = rpre[j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float y = fmaf((float)v[e], cs[e], cb[e]);
                        const float idv = p.r_scale ? fmaf((float)r[e], rs[e], rb[e]) : (float)r[e];
                        v[e] = (bf16_t)fmaxf(y + idv, 0.f);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)v[e];
                        st_sum[e] += f;
                        st_sq[e] = fmaf(f, f, st_sq[e]);
                    }
                }
                if (EPI != 0 || C) stream_store(v, reinterpret_cast<bf16x8*>(C + (long)m * p.ldc + n));   // streamed out: keep L2 for the operands
            }
        }
        // the next loop iteration's vmcnt wait + barrier orders these staging reads before the buffer is refilled
        counted_wait = EPI != 5 && (EPI != 0 || C != nullptr) && (m0 + BM <= p.M);   // (EPI 5 issues 16 stores per wave: plain wait)
    }

    if (EPI == 0 && p.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = 8; o <= 32; o <<= 1) {
                st_sum[e] += __shfl_xor(st_sum[e], o, 64);
                st_sq[e] += __shfl_xor(st_sq[e], o, 64);
            }
        }
        __syncthreads();
        float* red = (float*)smem;                  // [4 waves][2][64]
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 2 + 0) * 64 + lane * 8 + e] = st_sum[e];
                red[(wave * 2 + 1) * 64 + lane * 8 + e] = st_sq[e];
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int wn_ = tid >> 6, c = tid & 63, n = n0 + tid;
            const float sv = red[((0 + 2 * wn_) * 2 + 0) * 64 + c] + red[((1 + 2 * wn_) * 2 + 0) * 64 + c];
            const float qv = red[((0 + 2 * wn_) * 2 + 1) * 64 + c] + red[((1 + 2 * wn_) * 2 + 1) * 64 + c];
            cvcl_bn_stats_out(p.stats, p.stats_acc, blockIdx.x, p.N, n, sv, qv);
        }
    }
}

__global__ void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = by + i, c = bx + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? in[(long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int c = bx + i, r = by + threadIdx.x;
        if (c < cols && r < rows) out[(long)c * rows + r] = tile[threadIdx.x][i];
    }
}

// d_bias[n] = sum_m dY[m][n]; one workgroup per 16 columns, 64 row slices reduced through LDS in a fixed order
// (the first version used 64 columns x 4 slices: N / 64 workgroups walking M / 4 dependent loads each -- 60 us for a
// 1280 x 2048 gradient)
__global__ __launch_bounds__(1024) void colsum_f32_kernel(const float* __restrict__ dY, float* __restrict__ out, int M, int N) {
    __shared__ float part[64][16];
    const int c = threadIdx.x & 15, s = threadIdx.x >> 4, n = blockIdx.x * 16 + c;
    float acc = 0.f;
    if (n < N && s < M) acc = ordered_sum<8, float>((M - s + 63) / 64, [&](int j) { return dY[(long)(s + 64 * j) * N + n]; });
    part[s][c] = acc;
    __syncthreads();
    if (s == 0 && n < N) {
        float t = 0.f;
        for (int i = 0; i < 64; ++i) t += part[i][c];
        out[n] = t;
    }
}

// Resident workgroups per CU of each kernel variant (queried once); the persistent grid never exceeds what is
// co-resident, so there is no second "wave" of workgroups and no tail.
// bf16: the register allocator is asked for 2 or 3 workgroups per CU ($CVCL_GEMM_MINW, default 2: at 3 the
// 168-VGPR budget spills ~25 registers to scratch); fp32 (parity mode) runs one workgroup per CU.
template <typename T> int min_waves() {
    if (sizeof(T) != 2) return 1;
    static int v = 0;
    if (!v) v = cvcl_lab_int("CVCL_GEMM_MINW", 2) == 3 ? 3 : 2;
    return v;
}

template <typename T, int PRO, bool LEAN, int MINW>
int resident_per_cu_v() {
    static int cached = 0;
    if (cached) return cached;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gemm_kernel<T, PRO, LEAN, MINW>, 256,
                                                     gemm_lds_bytes<T>()) != hipSuccess || n < 1)
        n = 1;
    cached = n > 4 ? 4 : n;
    return cached;
}
template <typename T, int PRO, bool LEAN>
int resident_per_cu() {
    if constexpr (sizeof(T) == 2)
        return min_waves<T>() == 3 ? resident_per_cu_v<T, PRO, LEAN, 3>() : resident_per_cu_v<T, PRO, LEAN, 2>();
    else
        return resident_per_cu_v<T, PRO, LEAN, 1>();
}

int num_cus() {
    static int cached = 0;
    if (cached) return cached;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
        n = 256;
    cached = n;
    return cached;
}

// persistent grid.x: minimise the tiles per workgroup; prefer multiples of 8 so that the workgroups (i, j) and
// (i, j+1) -- linear ids i and i + grid_m -- share an XCD and therefore the A tile in L2
int grid_m_for(int M, int N, int capacity) {
    const int tiles = cvcl_div_up(M, BM), ntn = cvcl_div_up(N, BN);
    int cap = capacity / ntn;
    if (cap < 1) cap = 1;
    if (tiles <= cap) return tiles;
    int best = cap, best_cost = cvcl_div_up(tiles, cap);
    for (int g = cap & ~7; g >= 8 && g > cap / 2; g -= 8) {
        const int cost = cvcl_div_up(tiles, g);
        if (cost < best_cost || (cost == best_cost && (best & 7) != 0)) { best = g; best_cost = cost; }
    }
    return best;
}

template <typename T> int grid_m_query(int M, int N);

template <typename T, int PRO, bool LEAN, int MINW, int TR = 0>
int launch_gemm_w(const cvcl_gemm_args* a, GemmDev& d, hipStream_t stream) {
    const int gm = grid_m_query<T>(a->M, a->N);      // same capacity for every variant (see grid_m_query)
    if (a->stats) CVCL_CHECK_ARG(a->stats_rows == CVCL_STATS_ACCUMULATE || a->stats_rows >= gm, "cvcl_gemm: stats_rows %d < grid_m %d", a->stats_rows, gm);
    static CvclLdsAttr attr_set;
    constexpr int lds = gemm_lds_bytes<T>();
    if (!attr_set.ready()) {
        if (hipFuncSetAttribute((const void*)gemm_kernel<T, PRO, LEAN, MINW, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            cvcl_set_error("cvcl_gemm: cannot raise dynamic LDS limit to %d", lds);
            return CVCL_ELAUNCH;
        }
        attr_set.mark();
    }
    dim3 grid(gm, cvcl_div_up(a->N, BN));
    CvclProfScope prof(stream, sizeof(T) == 2 ? CVCL_K_GEMM : CVCL_K_GEMM_F32);
    hipLaunchKernelGGL((gemm_kernel<T, PRO, LEAN, MINW, TR>), grid, dim3(256), lds, stream, d);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

template <typename T, int PRO, bool LEAN>
int launch_gemm_v(const cvcl_gemm_args* a, GemmDev& d, hipStream_t stream) {
    if constexpr (sizeof(T) == 2)
        return min_waves<T>() == 3 ? launch_gemm_w<T, PRO, LEAN, 3>(a, d, stream) : launch_gemm_w<T, PRO, LEAN, 2>(a, d, stream);
    else
        return launch_gemm_w<T, PRO, LEAN, 1>(a, d, stream);
}

// ------------------------------------------------------------------------------------------------
// Small fp32 GEMMs (fc 2048->E on 256 pooled rows, the B x B similarity logits and their gradients): a 128 x 128 MFMA
// tiling would leave a handful of workgroups walking the whole K (8 workgroups for the fc layer = 160 us).  Here a
// workgroup owns a 16 x 16 output tile and splits K over 16 thread slices (float4 loads straight from L2, 4 x 4
// register micro-tiles), then reduces the 16 partial tiles through LDS in a fixed order -- deterministic, no scratch.
template <int TR>
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                             float* __restrict__ C, int M, int N, int K, int lda, int ldw,
                                                             int ldc, const float* __restrict__ exp_scale,
                                                             const float* __restrict__ bias) {
    __shared__ float red[16][16 * 16 + 4];
    const int tid = threadIdx.x, kl = tid & 15, tc = (tid >> 4) & 3, tr = tid >> 6;
    const int m0 = blockIdx.y * 16 + tr * 4, n0 = blockIdx.x * 16 + tc * 4;
    const float* ap[4];
    const float* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ap[i] = A + (long)min(m0 + i, M - 1) * lda;          // rows past the edge are clamped; their results are not stored
        wp[i] = W + (long)min(n0 + i, N - 1) * ldw;
    }
    // K-major operand (TR bit 0: A, bit 1: W; see gemm_kernel): the 4 x 4 block (4 rows x 4 k) is fetched as 4 k-rows of 4
    // consecutive rows; rows past the edge read as zero (their results are not stored)
    auto kmajor = [&](const float* base, int ld, int k, int r, int rows, f32x4 (&out)[4]) __attribute__((always_inline)) {
        f32x4 t[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (r + 3 < rows) t[kk] = *reinterpret_cast<const f32x4*>(base + (long)(k + kk) * ld + r);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) t[kk][e] = r + e < rows ? base[(long)(k + kk) * ld + r + e] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = f32x4{t[0][i], t[1][i], t[2][i], t[3][i]};
    };
    float acc[4][4] = {};
    for (int k = kl * 4; k < K; k += 64) {
        f32x4 a[4], w[4];
        if constexpr ((TR & 1) != 0) kmajor(A, lda, k, m0, M, a);
        if constexpr ((TR & 2) != 0) kmajor(W, ldw, k, n0, N, w);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr ((TR & 1) == 0) a[i] = *reinterpret_cast<const f32x4*>(ap[i] + k);
            if constexpr ((TR & 2) == 0) w[i] = *reinterpret_cast<const f32x4*>(wp[i] + k);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j] = fmaf(a[i][e], w[j][e], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[kl][(tr * 4 + i) * 16 + tc * 4 + j] = acc[i][j];
    __syncthreads();
    const int r = tid >> 4, c = tid & 15;
    float v = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) v += red[l][r * 16 + c];
    const int m = blockIdx.y * 16 + r, n = blockIdx.x * 16 + c;
    if (m < M && n < N) {
        if (exp_scale) v *= expf(*exp_scale);
        if (bias) v += bias[n];
        C[(long)m * ldc + n] = v;
    }
}


// ------------------------------------------------------------------------------------------------
// The trainable tail's fp32 GEMMs with split arithmetic (round 5): 64 x 64 output tile per 256-thread workgroup, one tile per
// workgroup (no persistence), K step 32.  The products of the text transformer (reference multimodal/multimodal.py:553-573) are
// 1280 x {512..2048} x {512..2048}: 128 x 128 tiles gave 40-160 workgroups, each walking its K tiles alone on a CU at ~2 us per
// tile (load -> LDS -> multiply, one wave per SIMD: profiles/r05_tail_c4_kernel_stats.csv, 47-98 us per launch whatever the
// arithmetic).  Here a launch is 256-640 small workgroups, 18 KiB of LDS and < 128 registers each, so several share a CU and hide each
// other's load -> LDS -> multiply round trips; operands row-major or K-major (TR bits as gemm_kernel), fp32 in, hi / lo bf16 parts
// in LDS, hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16, fp32 out (+ bias, ReLU), row sums of a K-major A for the bias gradient.
constexpr int TS = 64;                                   // tile side
template <int TR>
__global__ __launch_bounds__(256, 2) void gemm_f32_split_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                                                int M, int N, int K, int lda, int ldw, int ldc, const float* __restrict__ bias,
                                                                int act, float* __restrict__ a_rowsum, int vec) {
    __shared__ __attribute__((aligned(16))) char smem[2 * TS * ROWB];
    char* sA = smem;
    char* sW = smem + TS * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * TS, n0 = blockIdx.y * TS;
    // staging roles.  row-major operand: chunk kc (4 k) of rows r0, r0 + 32; K-major operand: 4 rows 4 mc .. of k-rows kr, kr + 16
    const int kc = tid & 7, r0 = tid >> 3, mc = tid & 15, kr = tid >> 4;
    // K tiles are requested PF tiles ahead into a ring of register sets: with one workgroup per CU (dW of a 2048 x 512 weight is
    // exactly 256 tiles) nothing else hides the ~1 us load round trip of a tile whose multiply takes 0.1 us -- the first version,
    // one tile ahead, ran 50 us per launch at 1.3 us per K step
    constexpr int PF = 4;
    f32x4 ra[PF][2], rw[PF][2];
    // FAST (a workgroup-uniform property: 16-byte aligned operands, the tile inside M x N, K a multiple of 128): unconditional
    // 16-byte loads -- a branch between a load and its use makes the compiler wait for ALL outstanding loads (vmcnt(0)) and the ring
    // of requests collapses to one tile in flight
    auto load_tile = [&](f32x4 (&qa)[2], f32x4 (&qw)[2], int k0, auto FAST) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            auto fetch = [&](const float* base, int ld, bool kmajor, int row_base, int rows) -> f32x4 {
                if constexpr (decltype(FAST)::value) {
                    if (!kmajor) return *reinterpret_cast<const f32x4*>(base + (long)(row_base + r0 + 32 * j) * ld + k0 + kc * 4);
                    return *reinterpret_cast<const f32x4*>(base + (long)(k0 + kr + 16 * j) * ld + row_base + mc * 4);
                }
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (!kmajor) {
                    const int r = row_base + r0 + 32 * j, k = k0 + kc * 4;
                    if (r < rows) {
                        const float* p = base + (long)r * ld + k;
                        if (vec && k + 3 < K) v = *reinterpret_cast<const f32x4*>(p);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = k + e < K ? p[e] : 0.f;
                        }
                    }
                } else {
                    const int kk = k0 + kr + 16 * j, r = row_base + mc * 4;
                    if (kk < K) {
                        const float* p = base + (long)kk * ld + r;
                        if (vec && r + 3 < rows) v = *reinterpret_cast<const f32x4*>(p);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = r + e < rows ? p[e] : 0.f;
                        }
                    }
                }
                return v;
            };
            qa[j] = fetch(A, lda, (TR & 1) != 0, m0, M);
            qw[j] = fetch(W, ldw, (TR & 2) != 0, n0, N);
        }
    };
    float rsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto store_tile = [&](const f32x4 (&qa)[2], const f32x4 (&qw)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            auto put = [&](char* base, const f32x4& v, bool kmajor) {
                bf16_t hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split_hi_lo(v[e], hi[e], lo[e]);
                if (!kmajor) {
                    char* row = base + (r0 + 32 * j) * ROWB + kc * 8;
                    *reinterpret_cast<bf16x4*>(row) = bf16x4{hi[0], hi[1], hi[2], hi[3]};
                    *reinterpret_cast<bf16x4*>(row + 64) = bf16x4{lo[0], lo[1], lo[2], lo[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        char* el = base + (mc * 4 + e) * ROWB + (kr + 16 * j) * 2;
                        *reinterpret_cast<bf16_t*>(el) = hi[e];
                        *reinterpret_cast<bf16_t*>(el + 64) = lo[e];
                    }
                }
            };
            put(sA, qa[j], (TR & 1) != 0);
            put(sW, qw[j], (TR & 2) != 0);
            if constexpr ((TR & 1) != 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) rsum[e] += qa[j][e];
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int ktiles = (K + 31) / 32;
    auto k_loop = [&](auto FAST) __attribute__((always_inline)) {
        // (the ring's tail: requests past the last tile re-read the last one -- harmless, never stored -- so that no branch sits
        // between a request and its use on the fast path)
#pragma unroll
        for (int u = 0; u < PF; ++u) load_tile(ra[u], rw[u], min(u, ktiles - 1) * 32, FAST);
        for (int kt0 = 0; kt0 < ktiles; kt0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int kt = kt0 + u;
                // (fast path: K is a multiple of 32 PF, the body is ONE basic block -- with a branch per step the compiler's wait
                // insertion lost count across the back edge and drained the ring, vmcnt(0), at every fourth step)
                if (decltype(FAST)::value || kt < ktiles) {     // (uniform)
                    __syncthreads();                           // the previous tile's fragment reads are done
                    store_tile(ra[u], rw[u]);
                    __syncthreads();
                    load_tile(ra[u], rw[u], min(kt + PF, ktiles - 1) * 32, FAST);     // PF tiles ahead, in flight under the next multiplies
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const char* wr = sW + (wn * 32 + l31) * ROWB + (g2 * 2 + h) * 16;
                        const char* ar = sA + (wm * 32 + l31) * ROWB + (g2 * 2 + h) * 16;
                        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wr), wl = *reinterpret_cast<const bf16x8*>(wr + 64);
                        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ar), al = *reinterpret_cast<const bf16x8*>(ar + 64);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc, 0, 0, 0);      // small terms first
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc, 0, 0, 0);
                    }
                }
            }
        }
    };
    if (vec && m0 + TS <= M && n0 + TS <= N && (K & (32 * PF - 1)) == 0) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    // this lane: output row m = m0 + wm * 32 + l31, columns n0 + wn * 32 + 8 g + 4 h + e
    const int m = m0 + wm * 32 + l31;
    if (m < M) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * 32 + 8 * g + 4 * h;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[4 * g + e] + ((bias && n + e < N) ? bias[n + e] : 0.f);
                if (act == CVCL_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
            }
            float* dst = C + (long)m * ldc + n;
            if (n + 3 < N && (ldc & 3) == 0 && vec) *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) dst[e] = v[e];
            }
        }
    }
    if constexpr ((TR & 1) != 0) {
        // row sums of the K-major A (the bias gradient beside dW = dY^T X): the 16 k-row slices of a row group through LDS, fixed order
        if (a_rowsum && blockIdx.y == 0) {
            __syncthreads();
            float* part = reinterpret_cast<float*>(smem);          // [16][64]
#pragma unroll
            for (int e = 0; e < 4; ++e) part[kr * 64 + mc * 4 + e] = rsum[e];
            __syncthreads();
            if (tid < 64 && m0 + tid < M) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) t += part[i * 64 + tid];
                a_rowsum[m0 + tid] = t;
            }
        }
    }
}

}  // namespace
extern "C" int cvcl_gemm_pro(const cvcl_gemm_args* a, void* stream);
extern "C" int cvcl_gemm_pro_supported(const cvcl_gemm_args* a);
extern "C" int cvcl_gemm_pro_stats_rows(int M, int N);
extern "C" int cvcl_gemm8w(int epi, const cvcl_gemm_args* a, void* stream);
extern "C" int cvcl_gemm8w_supported(int M, int N, int K, int lda, int ldw, int ldc);
extern "C" int cvcl_gemm8w_stats_rows(int M, int N);
namespace {

// Policy for the 8-wave 256 (224) x 256 kernel (gemm8w.hip): the MFMA-bound shapes -- K >= 256, N a multiple of 256, enough
// 256-row tiles to occupy the chip at one workgroup per CU, plain operands (no BN prologue, no output
// scale, no Bottleneck-tail / GELU-backward epilogue) -- and, when BN statistics are requested, a statistics buffer sized by
// cvcl_gemm_stats_rows.  Measured on MI355X (tools/gemm_lab, profiles/r02_gemm_lab.txt): ResNeXt layer-3/4 1x1 convolutions
// and ViT-B linears 10-25 % faster than the 128 x 128 kernel below; $CVCL_GEMM8W=0 switches it off.
// Returns -1 (not selected) or the epilogue id.
inline int pick_gemm8w(int dtype, const cvcl_gemm_args* a) {
    static const bool on = cvcl_env_on("CVCL_GEMM8W");
    if (!on || dtype != CVCL_BF16) return -1;
    if (!cvcl_gemm8w_supported(a->M, a->N, a->K, a->lda, a->ldw, a->ldc) || a->K < 256) return -1;
    if (a->a_scale || a->exp_scale || a->c_scale || a->C_pre || a->G) return -1;
    if (a->ln_stats && (a->R || !a->ln_colsum || !a->bias || a->row_part)) return -1;
    if (a->row_part && (!a->R || a->act != CVCL_ACT_NONE)) return -1;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!al16(a->A) || !al16(a->W) || !al16(a->C) || !al16(a->R) || !al16(a->bias) || (a->R && a->ldr % 8)) return -1;
    long a_rows = a->M;
    if (a->gather_stride > 1) {                              // strided 1x1 convolution (the downsample branch of blocks 2.0 / 3.0 / 4.0)
        if (a->gather_ho <= 0 || a->gather_wo <= 0 || a->M % (a->gather_ho * a->gather_wo)) return -1;
        a_rows = (long)(a->M / (a->gather_ho * a->gather_wo)) * a->gather_hi * a->gather_wi;
    }
    if (a_rows * a->lda >= (1L << 31) || (long)a->N * a->ldw >= (1L << 31)) return -1;
    if ((long)cvcl_div_up(a->M, 256) * (a->N / 256) < 96) return -1;
    // bandwidth-bound shapes stay with the 128 x 128 kernel (two workgroups per CU keep more bytes in flight): layer-2 block-0
    // conv1, M 802816 x N 256 x K 256, measured 176 us there vs 191-197 us here; N K / (N + K) = flop per byte of A + C traffic
    static const long min_intensity = cvcl_lab_int("CVCL_G8_MIN_INTENSITY", 170);
    if ((long)a->N * a->K < min_intensity * (a->N + a->K)) return -1;
    const bool plain = !a->bias && !a->R && a->act == CVCL_ACT_NONE;
    if (a->stats) {
        if (!plain || (a->stats_rows != CVCL_STATS_ACCUMULATE && a->stats_rows < cvcl_gemm8w_stats_rows(a->M, a->N))) return -1;
        return 0;
    }
    if (plain) return 0;
    if (!a->C || (a->R && a->act != CVCL_ACT_NONE)) return -1;     // (activation + residual together: the 128 x 128 kernel)
    return 1;
}

// Bandwidth-bound 1x1 convolution with the producer's BN + ReLU on its input (gemm_pro.hip): conv3 of ResNeXt layers 1-2.
// $CVCL_GEMM_PRO=0 refuses it (callers then have to normalise the operand themselves).
inline bool pick_gemm_pro(int dtype, const cvcl_gemm_args* a) {
    static const bool on = cvcl_env_on("CVCL_GEMM_PRO");
    if (!on || dtype != CVCL_BF16 || !cvcl_gemm_pro_supported(a)) return false;
    // a plain operand takes this kernel only where it is a byte stream: M >= 2^17 rows of K <= 256 (conv1 of layer2.0 at B >= 64);
    // smaller products stay on the tiled kernels
    if (!a->a_scale && a->M < (1 << 17)) return false;
    return !a->stats || a->stats_rows == CVCL_STATS_ACCUMULATE || a->stats_rows >= cvcl_gemm_pro_stats_rows(a->M, a->N);
}

inline bool is_lean(const cvcl_gemm_args* a, const GemmDev& d) {
    return d.vec_in && d.vec_out && !a->bias && !a->exp_scale && !a->R && !a->G && !a->C_pre && a->act == CVCL_ACT_NONE && (a->N % BN) == 0;
}
inline int pro_kind(const cvcl_gemm_args* a) { return !a->a_scale ? 0 : (a->a_relu ? 2 : 1); }

template <int EPI>
int launch_gemm_glds(const cvcl_gemm_args* a, GemmDev& d, hipStream_t stream) {
    const int gm = grid_m_query<bf16_t>(a->M, a->N);
    if (a->stats) CVCL_CHECK_ARG(a->stats_rows == CVCL_STATS_ACCUMULATE || a->stats_rows >= gm, "cvcl_gemm: stats_rows %d < grid_m %d", a->stats_rows, gm);
    static CvclLdsAttr attr_set;
    if (!attr_set.ready()) {
        if (hipFuncSetAttribute((const void*)gemm_glds_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, GL_LDS) != hipSuccess) {
            cvcl_set_error("cvcl_gemm: cannot raise dynamic LDS limit to %d", GL_LDS);
            return CVCL_ELAUNCH;
        }
        attr_set.mark();
    }
    dim3 grid(gm, a->N / BN);
    CvclProfScope prof(stream, CVCL_K_GEMM);
    hipLaunchKernelGGL(gemm_glds_kernel<EPI>, grid, dim3(256), GL_LDS, stream, d);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

template <typename T>
int launch_gemm(const cvcl_gemm_args* a, hipStream_t stream) {
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    GemmDev d;
    d.A = a->A; d.W = a->W; d.C = a->C;
    d.M = a->M; d.N = a->N; d.K = a->K; d.lda = a->lda; d.ldw = a->ldw; d.ldc = a->ldc;
    d.a_scale = a->a_scale; d.a_shift = a->a_shift; d.a_relu = a->a_relu;
    d.g_ho = a->gather_ho; d.g_wo = a->gather_wo; d.g_hi = a->gather_hi; d.g_wi = a->gather_wi;
    d.g_s = a->gather_stride;
    d.exp_scale = a->exp_scale; d.bias = a->bias; d.act = a->act;
    d.R = a->R; d.ldr = a->ldr; d.stats = a->stats; d.centre = a->centre;
    d.stats_acc = a->stats && a->stats_rows == CVCL_STATS_ACCUMULATE;
    d.c_scale = a->c_scale; d.c_shift = a->c_shift; d.r_scale = a->r_scale; d.r_shift = a->r_shift;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    d.vec_in = (a->K % EPC == 0) && (a->lda % EPC == 0) && (a->ldw % EPC == 0) && al16(a->A) && al16(a->W);
    d.vec_out = (a->ldc % EPC == 0) && al16(a->C) && (!a->R || ((a->ldr % EPC == 0) && al16(a->R)));
    d.num_m_tiles = cvcl_div_up(a->M, BM);
    d.C2 = nullptr;
    d.a_rowsum = a->a_rowsum;
    const int tr = (a->a_trans ? 1 : 0) | (a->w_trans ? 2 : 0);
    const bool split = a->f32_split != 0;
    if (tr || a->a_rowsum || split) {
        // K-major operands / fused row sums: the fp32 gradient GEMMs of the trainable tail (no prologue, gather, statistics or BN tail)
        CVCL_CHECK_ARG(sizeof(T) == 4, "cvcl_gemm: a_trans / w_trans / a_rowsum / f32_split are fp32 options");
        CVCL_CHECK_ARG(!a->a_rowsum || a->a_trans, "cvcl_gemm: a_rowsum goes with a_trans (the bias gradient beside dW = dY^T X)");
        CVCL_CHECK_ARG(!a->a_scale && !(a->gather_stride > 1) && !a->stats && !a->centre && !a->c_scale && !a->C_pre && !a->G && a->C,
                       "cvcl_gemm: K-major operands / split arithmetic take no prologue / gather / statistics / BN-tail options");
    }
    const bool lean = is_lean(a, d);
    if constexpr (sizeof(T) == 2) {
        if (pick_gemm_pro(CVCL_BF16, a)) return cvcl_gemm_pro(a, stream);
        const int e8 = pick_gemm8w(CVCL_BF16, a);
        if (e8 >= 0) return cvcl_gemm8w(e8, a, stream);
        static const bool use_glds = cvcl_lab_int("CVCL_GEMM_GLDS", 1) != 0;
        if (use_glds && a->c_scale) {                // Bottleneck tail epilogue: only the direct-to-LDS kernel implements it
            CVCL_CHECK_ARG(d.vec_in && d.vec_out && pro_kind(a) == 0 && a->K % 64 == 0 && a->N % BN == 0 && a->R && a->c_shift &&
                               !a->bias && !a->exp_scale && !a->stats && (a->r_scale == nullptr) == (a->r_shift == nullptr),
                           "cvcl_gemm: the c_scale epilogue needs bf16, K %% 64 == 0, N %% 128 == 0, a residual and no bias/stats");
            return launch_gemm_glds<2>(a, d, stream);
        }
        if (use_glds && lean && pro_kind(a) == 0 && a->K % 64 == 0) return launch_gemm_glds<0>(a, d, stream);
        // ViT / nn.Linear shapes: bias, activation, residual, no statistics
        const bool al = ((uintptr_t)a->bias & 15) == 0;
        const bool lin_ok = use_glds && d.vec_in && d.vec_out && pro_kind(a) == 0 && a->K % 64 == 0 && a->N % BN == 0 && !a->exp_scale &&
                            !a->stats && al && !(a->gather_stride > 1);
        if (a->C_pre || a->G) {                          // training epilogues: only this kernel implements them
            CVCL_CHECK_ARG(lin_ok, "cvcl_gemm: the C_pre / G epilogues need bf16, K %% 64 == 0, N %% 128 == 0 and 16-byte aligned rows");
            if (a->C_pre) {
                CVCL_CHECK_ARG(a->act == CVCL_ACT_GELU && !a->R && !a->G && ((uintptr_t)a->C_pre & 15) == 0,
                               "cvcl_gemm: C_pre goes with act = GELU and no residual");
                d.C2 = a->C_pre;
                return launch_gemm_glds<5>(a, d, stream);
            }
            CVCL_CHECK_ARG(a->act == CVCL_ACT_NONE && !a->R && !a->bias && a->ldg % 8 == 0 && ((uintptr_t)a->G & 15) == 0,
                           "cvcl_gemm: G (GELU-backward epilogue) takes no bias / activation / residual");
            d.R = a->G; d.ldr = a->ldg;
            return launch_gemm_glds<6>(a, d, stream);
        }
        if (lin_ok)
            return a->act == CVCL_ACT_GELU ? launch_gemm_glds<4>(a, d, stream)
                 : a->act == CVCL_ACT_RELU ? launch_gemm_glds<3>(a, d, stream) : launch_gemm_glds<1>(a, d, stream);
    }
    CVCL_CHECK_ARG(a->C && !a->c_scale, "cvcl_gemm: statistics-only / BN-tail epilogues need the direct-to-LDS bf16 path");
    if constexpr (sizeof(T) == 4) {
        auto al16p = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
        // measured cost models (us, MI355X): the split-K VALU kernel runs ~13.4 GMAC/s-per-us of work on any shape; the 128-tile
        // fp32 MFMA kernel needs ~4.6 us per 64-deep K step per round of <= 256 tiles, whatever M and N are (4.2 with split
        // arithmetic: its K step is bound by the serial load -> LDS -> multiply structure at one wave per SIMD, not by the matrix
        // pipe); the 64 x 64 split kernel ~1 us per (tile, 32-deep K step) with ~4 workgroups per CU overlapping
        static const bool split64_on = cvcl_lab_int("CVCL_SPLIT64", 1) != 0;
        const bool split64_ok = split64_on && split && pro_kind(a) == 0 && !(a->gather_stride > 1) && !a->stats && !a->R && !a->centre && !a->exp_scale &&
                                (a->act == CVCL_ACT_NONE || a->act == CVCL_ACT_RELU);
        const double t_small = (double)a->M * a->N * a->K / 13.4e6 + 5.0;
        const double t_mfma128 = 12.0 + (a->K / 64.0) * (split ? 4.2 : 4.6) * cvcl_div_up((long)cvcl_div_up(a->M, BM) * cvcl_div_up(a->N, BN), 256);
        const double t_split64 = split64_ok ? 6.0 + (double)cvcl_div_up(a->M, TS) * cvcl_div_up(a->N, TS) * cvcl_div_up(a->K, 32) / 1024.0 +
                                              0.25 * cvcl_div_up(a->K, 32) : 1e30;
        if (pro_kind(a) == 0 && !(a->gather_stride > 1) && !a->stats && !a->R && !a->centre && a->act == CVCL_ACT_NONE && a->K % 4 == 0 &&
            a->lda % 4 == 0 && a->ldw % 4 == 0 && al16p(a->A) && al16p(a->W) && !a->a_rowsum && t_small < t_mfma128 && t_small < t_split64) {
            CvclProfScope prof(stream, CVCL_K_GEMM_F32);
            const dim3 grid(cvcl_div_up(a->N, 16), cvcl_div_up(a->M, 16));
#define CVCL_SMALL(TR_) hipLaunchKernelGGL(gemm_f32_small_kernel<TR_>, grid, dim3(256), 0, stream, (const float*)a->A, (const float*)a->W, \
                                           (float*)a->C, a->M, a->N, a->K, a->lda, a->ldw, a->ldc, a->exp_scale, a->bias)
            switch (tr) { case 0: CVCL_SMALL(0); break; case 1: CVCL_SMALL(1); break; case 2: CVCL_SMALL(2); break; default: CVCL_SMALL(3); }
#undef CVCL_SMALL
            CVCL_LAUNCH_CHECK();
            return CVCL_OK;
        }
        if (split64_ok && t_split64 < t_mfma128) {
            // the tail's products: many 64 x 64 workgroups (gemm_f32_split_kernel)
            const int vec = (a->lda % 4 == 0) && (a->ldw % 4 == 0) && al16p(a->A) && al16p(a->W) && al16p(a->C);
            CvclProfScope prof(stream, CVCL_K_GEMM_F32);
            const dim3 grid(cvcl_div_up(a->M, TS), cvcl_div_up(a->N, TS));
#define CVCL_SPLIT(TR_) hipLaunchKernelGGL(gemm_f32_split_kernel<TR_>, grid, dim3(256), 0, stream, (const float*)a->A, (const float*)a->W, \
                                           (float*)a->C, a->M, a->N, a->K, a->lda, a->ldw, a->ldc, a->bias, a->act, a->a_rowsum, vec)
            switch (tr) { case 0: CVCL_SPLIT(0); break; case 1: CVCL_SPLIT(1); break; case 2: CVCL_SPLIT(2); break; default: CVCL_SPLIT(3); }
#undef CVCL_SPLIT
            CVCL_LAUNCH_CHECK();
            return CVCL_OK;
        }
        if (tr || split) {
            CVCL_CHECK_ARG(pro_kind(a) == 0, "cvcl_gemm: K-major operands take no prologue");
            switch (tr | (split ? 4 : 0)) {
                case 1: return launch_gemm_w<float, 0, false, 1, 1>(a, d, stream);
                case 2: return launch_gemm_w<float, 0, false, 1, 2>(a, d, stream);
                case 3: return launch_gemm_w<float, 0, false, 1, 3>(a, d, stream);
                case 4: return launch_gemm_w<float, 0, false, 1, 4>(a, d, stream);
                case 5: return launch_gemm_w<float, 0, false, 1, 5>(a, d, stream);
                case 6: return launch_gemm_w<float, 0, false, 1, 6>(a, d, stream);
                default: return launch_gemm_w<float, 0, false, 1, 7>(a, d, stream);
            }
        }
    }
    switch (pro_kind(a)) {
        case 0: return lean ? launch_gemm_v<T, 0, true>(a, d, stream) : launch_gemm_v<T, 0, false>(a, d, stream);
        case 1: return lean ? launch_gemm_v<T, 1, true>(a, d, stream) : launch_gemm_v<T, 1, false>(a, d, stream);
        default: return lean ? launch_gemm_v<T, 2, true>(a, d, stream) : launch_gemm_v<T, 2, false>(a, d, stream);
    }
}

// every variant of one dtype is compiled for the same occupancy target, so one query stands for all of them
template <typename T>
int grid_m_query(int M, int N) {
    return grid_m_for(M, N, resident_per_cu<T, 0, true>() * num_cus());
}

}  // namespace

extern "C" int cvcl_gemm_grid_m(int dtype, int M, int N, int has_prologue) {
    (void)has_prologue;
    return dtype == CVCL_BF16 ? grid_m_query<bf16_t>(M, N) : grid_m_query<float>(M, N);
}

// exact number of BN-statistics rows cvcl_gemm will write for these arguments (a->stats / a->stats_rows need not be set:
// the answer assumes a buffer of that many rows will be passed)
extern "C" int cvcl_gemm_stats_rows(int dtype, const cvcl_gemm_args* a) {
    if (!a) return 0;
    cvcl_gemm_args t = *a;
    static float dummy;
    t.stats = &dummy;
    t.stats_rows = 1 << 30;
    if (pick_gemm_pro(dtype, &t)) return cvcl_gemm_pro_stats_rows(a->M, a->N);
    if (pick_gemm8w(dtype, &t) == 0) return cvcl_gemm8w_stats_rows(a->M, a->N);
    return cvcl_gemm_grid_m(dtype, a->M, a->N, 0);
}

extern "C" int cvcl_gemm_ln_supported(const cvcl_gemm_args* a) { return a && pick_gemm8w(CVCL_BF16, a) == 1; }

extern "C" int cvcl_gemm(int dtype, const cvcl_gemm_args* a, void* stream) {
    CVCL_CHECK_ARG(a && a->A && a->W && (a->C || a->stats), "cvcl_gemm: null operand");
    if ((a->ln_stats || a->ln_colsum || a->row_part) && !(dtype == CVCL_BF16 && pick_gemm8w(dtype, a) == 1)) {
        cvcl_set_error("cvcl_gemm: ln_stats / row_part (LayerNorm folded into the linear) exist in the 8-wave bf16 kernel only; these "
                       "arguments do not select it (M %d N %d K %d) -- ask cvcl_gemm_ln_supported first", a->M, a->N, a->K);
        return CVCL_EUNSUPPORTED;
    }
    CVCL_CHECK_ARG(!a->c_scale || dtype == CVCL_BF16, "cvcl_gemm: the c_scale epilogue exists for bf16 only");
    CVCL_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "cvcl_gemm: bad shape %d %d %d", a->M, a->N, a->K);
    CVCL_CHECK_ARG((a->a_scale == nullptr) == (a->a_shift == nullptr), "cvcl_gemm: a_scale/a_shift must come together");
    // centred storage belongs to the convolution epilogues (plain / statistics / Bottleneck tail); nn.Linear epilogues have no BN behind them
    CVCL_CHECK_ARG(!a->centre || (((uintptr_t)a->centre & 15) == 0 && !a->bias && !a->exp_scale && !a->C_pre && !a->G &&
                                  (a->c_scale || (!a->R && a->act == CVCL_ACT_NONE))),
                   "cvcl_gemm: centre goes with the convolution epilogues only (16-byte aligned, no bias / activation / residual)");
    if (dtype == CVCL_F32) return launch_gemm<float>(a, (hipStream_t)stream);
    if (dtype == CVCL_BF16) return launch_gemm<bf16_t>(a, (hipStream_t)stream);
    cvcl_set_error("cvcl_gemm: unknown dtype %d", dtype);
    return CVCL_EINVAL;
}

extern "C" int cvcl_transpose_f32(const float* in, float* out, int rows, int cols, void* stream) {
    CVCL_CHECK_ARG(in && out && rows > 0 && cols > 0, "cvcl_transpose_f32: bad args");
    dim3 grid(cvcl_div_up(cols, 32), cvcl_div_up(rows, 32));
    CvclProfScope prof(stream, CVCL_K_OTHER);
    hipLaunchKernelGGL(transpose_f32_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, in, out, rows, cols);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_colsum_f32(const float* dY, float* d_bias, int M, int N, void* stream) {
    CVCL_CHECK_ARG(dY && d_bias && M > 0 && N > 0, "cvcl_colsum_f32: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(cvcl_div_up(N, 16)), dim3(1024), 0, (hipStream_t)stream, dY, d_bias, M, N);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}
