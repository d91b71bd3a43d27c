// ResNeXt-50 32x4d image-encoder blocks for gfx950 (reference: torchvision.models.resnext50_32x4d
// reached at multimodal/multimodal.py:101 through VisionEncoder.forward; factory :155-158, utils.py:207-209).
//
// Activations are NHWC.  "raw" tensors are convolution outputs *before* BatchNorm: every convolution
// kernel also emits per-channel sum / sum-of-squares partial rows of what it stored, bn_finalize turns
// them into a per-channel (scale, shift) and updates the running statistics, and the *consumer* of
// the raw tensor applies scale/shift(+ReLU) while loading.  A normalised tensor is materialised only
// where the network needs it twice (block outputs, max-pool output).
//
//   stem 7x7/2 (3->64)   : bf16: MFMA 16x16x32, B-operand gathered straight from an LDS image patch
//                          (k ordered (c, ky, kx padded to 8): 8 consecutive k = 8 consecutive pixels)
//   grouped 3x3 (32 grps): bf16: MFMA 16x16x32 on 16-channel units (block-diagonal weights for 4/8
//                          channels per group, one tap x 32 channels per step for 32 per group),
//                          input band staged in LDS with BN+ReLU applied on the way in
//   fp32 (parity mode)   : direct VALU kernels with identical semantics + a column-statistics pass
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "cvcl_common.h"

namespace {

constexpr int kMaxStatsRows = 1024;
constexpr size_t kTrunkAccLayer = (size_t)kBnAccRows * 2 * 2048;     // int64 slots of one layer's BatchNorm accumulators [8][2][2048]

// ------------------------------------------------------------------------------------------------
// BatchNorm bookkeeping
// ------------------------------------------------------------------------------------------------
// stats [rows][2][C] partial sums -> scale = gamma / sqrt(var + eps), shift = beta - mean * scale;
// running_mean/var EMA with the unbiased variance (nn.BatchNorm2d train mode), num_batches_tracked += 1.
// the running-statistics update r <- (1 - m) r + m x: one spelling, shared by the in-place and the deferred form (identical bits)
// (bn_ema lives in cvcl_common.h: shared with the finalize-on-load consumers)

// moments != NULL: the batch mean / unbiased variance are written there ([2][C]) and the running statistics are left alone --
// `bn_apply_moments_kernel` applies them later (cvcl_resnext50_fwd_deferred_stats: passes pipelined on two streams).
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ stats, int rows, double count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           int64_t* __restrict__ nbt, float momentum, float eps,
                                                           float* __restrict__ scale, float* __restrict__ shift, int C,
                                                           float* __restrict__ moments, int moments_ld,
                                                           const float* __restrict__ centre) {
    // 16 channels x 64 row slices per workgroup (short dependent-load chains: the early layers have 512 partial rows and
    // only 64-256 channels); slices combined in a fixed order (deterministic)
    constexpr int NS = 64, NC = 16;
    __shared__ double ps[NS][NC], pq[NS][NC];
    const int c = threadIdx.x & (NC - 1), sl = threadIdx.x / NC, ch = blockIdx.x * NC + c;
    double s = 0.0, q = 0.0;
    if (ch < C) {
        // eight rows' loads are issued before the first is added (same summation order): one wait per eight rows instead of a
        // dependent L2 round trip per row -- the kernel is nothing but latency
        constexpr int UN = 8;
        for (int r0 = sl; r0 < rows; r0 += NS * UN) {
            float a[UN], b[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int r = min(r0 + NS * u, rows - 1);                       // clamped: always a valid address, no branch
                a[u] = stats[((long)r * 2 + 0) * C + ch];
                b[u] = stats[((long)r * 2 + 1) * C + ch];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (r0 + NS * u < rows) { s += (double)a[u]; q += (double)b[u]; }
        }
    }
    ps[sl][c] = s;
    pq[sl][c] = q;
    __syncthreads();
    if (sl == 0 && ch < C) {
        s = 0.0; q = 0.0;
        for (int i = 0; i < NS; ++i) { s += ps[i][c]; q += pq[i][c]; }
        // the statistics are those of the STORED tensor y - c (centred storage): (scale, shift) apply to it as it lies in memory;
        // the mean of y itself, for the running statistics, is mean + c
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float sc = gamma[ch] / sqrtf((float)var + eps);
        scale[ch] = sc;
        shift[ch] = beta[ch] - (float)mean * sc;
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        const float true_mean = centre ? (float)(mean + (double)centre[ch]) : (float)mean;
        if (moments) {
            moments[ch] = true_mean;
            moments[moments_ld + ch] = (float)unbiased;
        } else if (running_mean) {
            running_mean[ch] = bn_ema(running_mean[ch], true_mean, momentum);
            running_var[ch] = bn_ema(running_var[ch], (float)unbiased, momentum);
        }
    }
    if (!moments && nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

// ---- finalize-on-load (round 6; BnSrc / bn_slice_affine in cvcl_common.h) ---------------------------------------------------------------
// bn_finalize is a 5 us kernel of dependent L2 round trips between a convolution and the consumer of its output: 45 launches per
// pass, each with two kernel boundaries on the trunk's dependent chain.  Where the consumer's workgroups normalise a FIXED set of
// channels -- the grouped 3x3's 64-channel slabs (BN1 of all 16 Bottlenecks), the Gram launch's K <= 256 operand columns (BN2 of
// layers 1-2) -- the producing convolution ACCUMULATES its partial sums instead of writing partial rows (int64 fixed point, one row
// per XCD: order-independent = deterministic) and the consumer forms its channels' (scale, shift) in its prologue: one round trip of
// 16 independent loads per channel, no launch; the workgroup with ``publish`` set also leaves the batch moments / running statistics
// (and the affine, for later readers) behind.  61 -> 38 launches of the bn_finalize class per pass; one pass in flight 5.66 -> 5.46 ms
// per C2 step, two passes in flight unchanged (the other pass already ran in those gaps) -- profiles/r06_fol_ab.txt, which also holds
// what was measured and dropped: partial ROWS reduced redundantly in every consumer workgroup (bit-identical to bn_finalize, but 3-8 us
// of dependent loads and an ordered fp64 chain inside every consumer: no gain), and channel-sliced 1024-thread forms of the
// elementwise passes of layers 3-4 finalizing BN2 / BN3 on load (18 launches left, but 2-3 us more per pass than the launch saved).

// deferred running-statistics update of the whole trunk: all 53 layers in one launch, from the moments a pass left behind
struct ApplyMomentsAll {
    float* rm[53]; float* rv[53]; int64_t* nbt[53];
    int C[53];
};
__global__ __launch_bounds__(256) void bn_apply_moments_kernel(ApplyMomentsAll t, const float* __restrict__ moments, float momentum) {
    const int l = blockIdx.x;
    const float* m = moments + (size_t)l * 4096;
    for (int ch = threadIdx.x; ch < t.C[l]; ch += 256) {
        t.rm[l][ch] = bn_ema(t.rm[l][ch], m[ch], momentum);
        t.rv[l][ch] = bn_ema(t.rv[l][ch], m[2048 + ch], momentum);
    }
    if (threadIdx.x == 0 && t.nbt[l]) *t.nbt[l] += 1;
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift, const float* __restrict__ centre, int C) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < C) {
        const float sc = gamma[ch] / sqrtf(rv[ch] + eps);
        scale[ch] = sc;
        shift[ch] = beta[ch] - (centre ? rm[ch] - centre[ch] : rm[ch]) * sc;      // affine of the tensor stored as y - c
    }
}

// eval mode of the whole trunk: all 53 (scale, shift) pairs from the running statistics in ONE launch (the layer table
// travels by value in the kernel argument block), instead of 53 dependent 4-us launches on the latency path
struct EvalAffineAll {
    const float* gamma[53]; const float* beta[53]; const float* rm[53]; const float* rv[53];
    int C[53];
};
// centres: the storage centres the pass will use ([53][2048], "Centred storage" in cvcl_hip.h); NULL with centres_out: the
// running means themselves, which this kernel then writes to centres_out for the convolutions to read (shift = beta);
// both NULL: plain storage
__global__ __launch_bounds__(256) void bn_eval_affine_all_kernel(EvalAffineAll t, float eps, float* __restrict__ affine,
                                                                 const float* __restrict__ centres, float* __restrict__ centres_out) {
    const int l = blockIdx.x;
    for (int ch = threadIdx.x; ch < t.C[l]; ch += 256) {
        const float sc = t.gamma[l][ch] / sqrtf(t.rv[l][ch] + eps);
        const float rm = t.rm[l][ch];
        affine[(size_t)l * 4096 + ch] = sc;
        if (centres) {
            affine[(size_t)l * 4096 + 2048 + ch] = t.beta[l][ch] - (rm - centres[(size_t)l * 2048 + ch]) * sc;
        } else if (centres_out) {
            affine[(size_t)l * 4096 + 2048 + ch] = t.beta[l][ch];
            centres_out[(size_t)l * 2048 + ch] = rm;
        } else {
            affine[(size_t)l * 4096 + 2048 + ch] = t.beta[l][ch] - rm * sc;
        }
    }
}

// per-column sum / sumsq partial rows of a stored [rows, C] tensor (fp32 parity path + tests)
template <typename T>
__global__ __launch_bounds__(256) void col_stats_kernel(const T* __restrict__ x, long rows, int C, float* __restrict__ stats) {
    __shared__ float ps[4][64], pq[4][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6, ch = blockIdx.y * 64 + c;
    float s = 0.f, q = 0.f;
    if (ch < C) {
        for (long r = (long)blockIdx.x * 4 + sl; r < rows; r += (long)gridDim.x * 4) {
            const float v = ElemTraits<T>::to_f(x[r * C + ch]);
            s += v;
            q = fmaf(v, v, q);
        }
    }
    ps[sl][c] = s;
    pq[sl][c] = q;
    __syncthreads();
    if (sl == 0 && ch < C) {
        stats[((long)blockIdx.x * 2 + 0) * C + ch] = (ps[0][c] + ps[1][c]) + (ps[2][c] + ps[3][c]);
        stats[((long)blockIdx.x * 2 + 1) * C + ch] = (pq[0][c] + pq[1][c]) + (pq[2][c] + pq[3][c]);
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing (reference layout OIHW fp32 -> kernel layout)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = ElemTraits<T>::from_f(in[i]);
}

// stem: [64][3][7][7] -> bf16 [4 ntile][6 kstep][16 n][32 k], k = kblock*8 + kx, r = 4*kstep + kblock = c*7 + ky
__global__ void pack_stem_kernel(const float* __restrict__ w, bf16_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * 6 * 16 * 32) return;
    const int k = i & 31, n = (i >> 5) & 15, ks = (i >> 9) % 6, nt = i / (6 * 512);
    const int r = 4 * ks + (k >> 3), kx = k & 7;
    float v = 0.f;
    if (r < 21 && kx < 7) {
        const int c = r / 7, ky = r % 7;
        v = w[((nt * 16 + n) * 3 + c) * 49 + ky * 7 + kx];
    }
    out[i] = (bf16_t)v;
}

// grouped 3x3: [C][cg][3][3] -> bf16 units of [KS][16 n][32 k]
//   cg <= 16: unit = 16-channel chunk, KS = 5, k = tapsel*16 + ci, tap = 2*ks + tapsel (tap 9 = zero),
//             block-diagonal: zero where input channel and output channel are in different groups
//   cg == 32: unit = (group, ntile), KS = 9 (= tap), k = ci
__global__ void pack_gconv_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int C, int cg) {
    const int KS = cg == 32 ? 9 : 5;
    const long total = (long)(C / 16) * KS * 512;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i & 31), n = (int)((i >> 5) & 15), ks = (int)((i >> 9) % KS), unit = (int)(i / ((long)KS * 512));
    const int co = unit * 16 + n;
    float v = 0.f;
    if (cg == 32) {
        v = w[((long)co * 32 + k) * 9 + ks];
    } else {
        const int tap = 2 * ks + (k >> 4), ci_abs = unit * 16 + (k & 15);
        if (tap < 9 && ci_abs / cg == co / cg) v = w[((long)co * cg + (ci_abs % cg)) * 9 + tap];
    }
    out[i] = (bf16_t)v;
}

// ------------------------------------------------------------------------------------------------
// stem 7x7 stride 2 pad 3, NCHW fp32 image -> NHWC raw                       (bf16, MFMA)
// ------------------------------------------------------------------------------------------------
constexpr int STEM_TH = 4;                 // output rows per work item
constexpr int STEM_ROWS = 2 * STEM_TH + 5; // input rows per channel
constexpr int STEM_PITCH = 144;            // dwords per patch row (288 bf16 >= 224 + 6 + 8 slack); 144 % 32 == 16
constexpr int STEM_OPITCH = 144;           // bytes per pixel of a wave's output slot (128 + 16 pad)

__global__ __launch_bounds__(256, 2) void stem_mfma_kernel(const float* __restrict__ x, const bf16_t* __restrict__ wp,
                                                        bf16_t* __restrict__ y, float* __restrict__ stats,
                                                        const float* __restrict__ centre, int B, int Hin, int Win) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* patch = (unsigned int*)smem;                 // [3*STEM_ROWS][STEM_PITCH] dwords (2 bf16 each)
    const int Ho = Hin / 2, Wo = Win / 2;
    const int bands = cvcl_div_up(Ho, STEM_TH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pix = lane & 15, kb = lane >> 4;

    bf16x8 wf[4][6];                                           // all 64 output channels' weights, register resident
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
            wf[nt][ks] = *reinterpret_cast<const bf16x8*>(wp + ((nt * 6 + ks) * 16 + pix) * 32 + kb * 8);

    f32x2 ssum[4][2], ssq[4][2];                              // (channel pairs: packed fp32)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { ssum[a][b] = f32x2{0.f, 0.f}; ssq[a][b] = f32x2{0.f, 0.f}; }

    const int mtiles_per_row = cvcl_div_up(Wo, 16);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // per-lane constants of the multiply loop: patch row (dwords) of k step ks -- k = 4 ks + kb -> (channel, ky), padded k at 20
    int koff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        const int r = min(4 * ks + kb, 20);
        const int c = r / 7, ky = r - c * 7;
        koff[ks] = (c * STEM_ROWS + ky) * STEM_PITCH;
    }
    for (int i = tid; i < 3 * STEM_ROWS * STEM_PITCH; i += 256) patch[i] = 0u;       // pad columns stay zero for good
    char* wst = smem + 3 * STEM_ROWS * STEM_PITCH * 4 + wave * (16 * STEM_OPITCH);  // this wave's output slot: 16 pixels
    // centred storage: the accumulators start at -centre[channel] (64 floats in LDS behind the output slots, re-read per tile:
    // the weights already hold 96 registers)
    float* cen = reinterpret_cast<float*>(smem + 3 * STEM_ROWS * STEM_PITCH * 4 + 4 * 16 * STEM_OPITCH);
    if (tid < 64) cen[tid] = centre ? -centre[tid] : 0.f;                           // (visible after the first item's barrier)
    // (XCD-major item order, as in gconv_mfma_kernel: adjacent bands of an image share 5 of their 13 input rows -- 0.247 GB read for
    // a 0.154 GB input in the plain order; time-neutral for this kernel, the traffic is what it saves)
    const int bx = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    for (int item = bx; item < B * bands; item += gridDim.x) {
        const int b = item / bands, band = item - b * bands;
        const int oy0 = band * STEM_TH;
        __syncthreads();
        // stage: rows (c, iy) <- x[b][c][2*oy0 - 3 + iy][*] as bf16 at columns x + 3 (the 3 + slack pad columns on either side were
        // zeroed once and are never written).  Slot s of a wave is patch row 4 s + wave, a lane is one 16-byte vector of that row
        // (Win / 4 <= 64 of them): the row, its channel, input row and validity are wave-uniform scalars -- no per-thread index
        // arithmetic (the flat index form spent two runtime divisions per slot and phase).  All of a batch's loads are issued
        // before the first LDS write; rows outside the image are staged as zeros.
        {
            constexpr int NROW = 3 * STEM_ROWS, NV = 5;          // two batches of 5 slots: 10 x 4 rows >= 39
            const int vec_per_row = Win / 4, jl = min(lane, vec_per_row - 1);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 v[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int r = min((half * NV + i) * 4 + wave_u, NROW - 1);
                    const int c = r / STEM_ROWS, iy = r - c * STEM_ROWS;
                    const int yin = min(max(2 * oy0 - 3 + iy, 0), Hin - 1);
                    v[i] = *reinterpret_cast<const f32x4*>(x + (((long)b * 3 + c) * Hin + yin) * Win + 4 * jl);
                }
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int r = (half * NV + i) * 4 + wave_u;
                    const int iy = r % STEM_ROWS, yin = 2 * oy0 - 3 + iy;
                    if (r < NROW && lane < vec_per_row) {
                        const bool ok = yin >= 0 && yin < Hin;
                        const unsigned u01 = ok ? round2(f32x2{v[i][0], v[i][1]}) : 0u, u23 = ok ? round2(f32x2{v[i][2], v[i][3]}) : 0u;
                        // four pixels from padded column 4 lane + 3 (odd): 2 + 4 + 2 bytes
                        char* dst = reinterpret_cast<char*>(patch + r * STEM_PITCH) + 8 * lane + 6;
                        *reinterpret_cast<unsigned short*>(dst) = (unsigned short)u01;
                        *reinterpret_cast<unsigned*>(dst + 2) = (u01 >> 16) | (u23 << 16);
                        *reinterpret_cast<unsigned short*>(dst + 6) = (unsigned short)(u23 >> 16);
                    }
                }
            }
        }
        __syncthreads();
        // m-tiles of the band in row-major order, wave w takes w, w + 4, ...: (row, tile in row) advance as scalars
        int ty = 0, tx = wave_u;
        while (tx >= mtiles_per_row) { tx -= mtiles_per_row; ++ty; }
        while (ty < STEM_TH) {
            const int ox0 = tx * 16;
            const int oy = oy0 + ty, ox = ox0 + pix;
            f32x4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = *reinterpret_cast<const f32x4*>(cen + nt * 16 + kb * 4);
            const unsigned int* row0 = patch + 2 * ty * STEM_PITCH + (ox < Wo ? ox : 0);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const unsigned int* src = row0 + koff[ks];
                u32x4 raw = {src[0], src[1], src[2], src[3]};      // 8 consecutive input pixels (kx = 0..7)
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], bfrag, acc[nt], 0, 0, 0);
            }
            // the tile (16 pixels x 64 channels) goes through a wave-private LDS slot so that it leaves as full 128-byte pixel rows
            // (16 B per lane) instead of 8-byte pieces in the MFMA layout
            const bool inside = oy < Ho && ox < Wo;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const u32x2 o = {round2(f32x2{acc[nt][0], acc[nt][1]}), round2(f32x2{acc[nt][2], acc[nt][3]})};
                *reinterpret_cast<u32x2*>(wst + pix * STEM_OPITCH + nt * 32 + kb * 8) = o;
                if (inside) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x2 sv = widen2(o[e]);
                        ssum[nt][e] += sv;
                        ssq[nt][e] = __builtin_elementwise_fma(sv, sv, ssq[nt][e]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            if (oy < Ho && y) {                                   // (y == NULL: statistics only -- the fused stem below recomputes the tile)
                bf16_t* drow = y + (((long)b * Ho + oy) * Wo + ox0) * 64 + (lane & 7) * 8;
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) {
                    const int px = (lane >> 3) + 8 * rd;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(wst + px * STEM_OPITCH + (lane & 7) * 16);
                    if (ox0 + px < Wo) *reinterpret_cast<u32x4*>(drow + (long)px * 64) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            tx += 4;
            while (tx >= mtiles_per_row) { tx -= mtiles_per_row; ++ty; }
        }
    }
    // statistics: reduce over the 16 pixel lanes, then over the 4 waves; channel = nt*16 + kb*4 + e
    __syncthreads();
    float* red = (float*)smem;                                    // [4 waves][2][64]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = ssum[nt][e >> 1][e & 1], q = ssq[nt][e >> 1][e & 1];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
            if (pix == 0) {
                red[(wave * 2 + 0) * 64 + nt * 16 + kb * 4 + e] = s;
                red[(wave * 2 + 1) * 64 + nt * 16 + kb * 4 + e] = q;
            }
        }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        const float v = (red[(0 * 2 + which) * 64 + c] + red[(1 * 2 + which) * 64 + c]) +
                        (red[(2 * 2 + which) * 64 + c] + red[(3 * 2 + which) * 64 + c]);
        stats[((long)blockIdx.x * 2 + which) * 64 + c] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// stem 7x7 stride 2 + BatchNorm + ReLU + maxpool 3x3 stride 2 pad 1 in ONE pass (round 5; torchvision ResNet.forward
// conv1 -> bn1 -> relu -> maxpool, reached from multimodal/multimodal.py:101): NCHW fp32 image -> NHWC bf16 [B, H/4, W/4, 64].
// The raw stem output [B, 112, 112, 64] (411 MB at B = 256) is never written: train mode runs stem_mfma_kernel with y = NULL first
// (statistics only), bn_finalize, then this kernel RECOMPUTES the convolution (it is 30 GFLOP) and pools it out of LDS --
// 0.57 + 0.72 GB of traffic become 0.15 + 0.26 GB.  Bit-identical to the two-pass form: the same MFMA sequence per output, the same
// rounding of the raw value to bf16 before the affine, and round(max) == max(round) on the non-negative post-ReLU values.
// One 512-thread workgroup per CU; an item = STEMP_TP pooled rows of one image = 2 TP + 1 convolution rows (the first one shared
// with the previous item: 25 % recomputation at TP = 2) = 4 TP + 7 input rows per channel.  Convolution tiles (16 pixels x 64
// channels, 24 MFMAs) are dealt to the 8 waves and land, normalised, in an LDS band [conv row][pixel + 1][64 channels]; after a
// barrier every thread takes (pooled pixel, 8 channels) units: nine 16-byte LDS reads, packed integer maxima, one 16-byte store.
constexpr int STEMP_TP = 2;
constexpr int STEMP_CR = 2 * STEMP_TP + 1;             // convolution rows per item
constexpr int STEMP_ROWS = 2 * STEMP_CR + 5;           // input rows per channel
constexpr int STEMP_BW = 128;                          // band row width in pixels (column 0 = the left padding column)
__global__ __launch_bounds__(512, 1) void stem_pool_mfma_kernel(const float* __restrict__ x, const bf16_t* __restrict__ wp,
                                                                bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ centre,
                                                                int B, int Hin, int Win) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* patch = (unsigned int*)smem;                 // [3 * STEMP_ROWS][STEM_PITCH] dwords (2 bf16 each)
    char* band = smem + 3 * STEMP_ROWS * STEM_PITCH * 4;       // [STEMP_CR][STEMP_BW][64] bf16
    float* cen = reinterpret_cast<float*>(band + STEMP_CR * STEMP_BW * 128);      // [64] -centre | [64] scale | [64] shift
    const int Ho = Hin / 2, Wo = Win / 2;
    const int Hp = (Ho - 1) / 2 + 1, Wp = (Wo - 1) / 2 + 1;
    const int bands = cvcl_div_up(Hp, STEMP_TP);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pix = lane & 15, kb = lane >> 4;

    bf16x8 wf[4][6];                                           // all 64 output channels' weights, register resident
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
            wf[nt][ks] = *reinterpret_cast<const bf16x8*>(wp + ((nt * 6 + ks) * 16 + pix) * 32 + kb * 8);
    const int mtiles_per_row = cvcl_div_up(Wo, 16);
    int koff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        const int r = min(4 * ks + kb, 20);
        const int c = r / 7, ky = r - c * 7;
        koff[ks] = (c * STEMP_ROWS + ky) * STEM_PITCH;
    }
    for (int i = tid; i < 3 * STEMP_ROWS * STEM_PITCH; i += 512) patch[i] = 0u;       // pad columns stay zero for good
    for (int i = tid; i < STEMP_CR * STEMP_BW * 32; i += 512) reinterpret_cast<unsigned*>(band)[i] = 0u;   // (padding columns: never rewritten)
    if (tid < 64) {
        cen[tid] = centre ? -centre[tid] : 0.f;
        cen[64 + tid] = scale[tid];
        cen[128 + tid] = shift[tid];
    }
    const int bx = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    for (int item = bx; item < B * bands; item += gridDim.x) {
        const int b = item / bands, bnd = item - b * bands;
        const int p0 = bnd * STEMP_TP;
        const int oy_first = 2 * p0 - 1;                      // convolution row of band row 0 (-1 for the first item: padding)
        // ---- stage: rows (c, iy) <- x[b][c][2 oy_first - 3 + iy][*] as bf16 at columns x + 3 (see stem_mfma_kernel) ----
        {
            constexpr int NROW = 3 * STEMP_ROWS, NV = (NROW + 15) / 16;      // 8 waves x 2 batches x NV slots >= NROW
            const int vec_per_row = Win / 4, jl = min(lane, vec_per_row - 1);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 v[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int r = min((half * NV + i) * 8 + wave_u, NROW - 1);
                    const int c = r / STEMP_ROWS, iy = r - c * STEMP_ROWS;
                    const int yin = min(max(2 * oy_first - 3 + iy, 0), Hin - 1);
                    v[i] = *reinterpret_cast<const f32x4*>(x + (((long)b * 3 + c) * Hin + yin) * Win + 4 * jl);
                }
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int r = (half * NV + i) * 8 + wave_u;
                    const int iy = r % STEMP_ROWS, yin = 2 * oy_first - 3 + iy;
                    if (r < NROW && lane < vec_per_row) {
                        const bool ok = yin >= 0 && yin < Hin;
                        const unsigned u01 = ok ? round2(f32x2{v[i][0], v[i][1]}) : 0u, u23 = ok ? round2(f32x2{v[i][2], v[i][3]}) : 0u;
                        char* dst = reinterpret_cast<char*>(patch + r * STEM_PITCH) + 8 * lane + 6;
                        *reinterpret_cast<unsigned short*>(dst) = (unsigned short)u01;
                        *reinterpret_cast<unsigned*>(dst + 2) = (u01 >> 16) | (u23 << 16);
                        *reinterpret_cast<unsigned short*>(dst + 6) = (unsigned short)(u23 >> 16);
                    }
                }
            }
        }
        __syncthreads();
        // ---- convolution tiles of the band's STEMP_CR rows: wave w takes tiles w, w + 8, ... (row-major) ----
        for (int t = wave_u; t < STEMP_CR * mtiles_per_row; t += 8) {
            const int ty = t / mtiles_per_row, tx = t - ty * mtiles_per_row;
            const int ox0 = tx * 16, ox = ox0 + pix;
            const int oy = oy_first + ty;
            char* brow = band + (ty * STEMP_BW + ox + 1) * 128 + kb * 8;
            if (oy < 0 || oy >= Ho) {                          // a padding row of the pooling window: zeros (post-ReLU domain)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<u32x2*>(brow + nt * 32) = u32x2{0u, 0u};
                continue;
            }
            f32x4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = *reinterpret_cast<const f32x4*>(cen + nt * 16 + kb * 4);
            const unsigned int* row0 = patch + 2 * ty * STEM_PITCH + (ox < Wo ? ox : 0);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const unsigned int* src = row0 + koff[ks];
                u32x4 raw = {src[0], src[1], src[2], src[3]};
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], bfrag, acc[nt], 0, 0, 0);
            }
            const bool inside = ox < Wo;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                // what the two-pass form stores, then what bn_relu_maxpool computes from it: relu(raw_bf16 * scale + shift)
                const u32x2 o = {round2(f32x2{acc[nt][0], acc[nt][1]}), round2(f32x2{acc[nt][2], acc[nt][3]})};
                const f32x4 sc4 = *reinterpret_cast<const f32x4*>(cen + 64 + nt * 16 + kb * 4);
                const f32x4 sh4 = *reinterpret_cast<const f32x4*>(cen + 128 + nt * 16 + kb * 4);
                const f32x2 r01 = widen2(o[0]), r23 = widen2(o[1]);
                const f32x2 v01 = {fmaxf(fmaf(r01[0], sc4[0], sh4[0]), 0.f), fmaxf(fmaf(r01[1], sc4[1], sh4[1]), 0.f)};
                const f32x2 v23 = {fmaxf(fmaf(r23[0], sc4[2], sh4[2]), 0.f), fmaxf(fmaf(r23[1], sc4[3], sh4[3]), 0.f)};
                *reinterpret_cast<u32x2*>(brow + nt * 32) = inside ? u32x2{round2(v01), round2(v23)} : u32x2{0u, 0u};
            }
        }
        __syncthreads();
        // ---- pool: (pooled row pr, pooled pixel pc, 8-channel chunk c8) units; band row 2 pr + dy, band column 2 pc + dx ----
        typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
        for (int u = tid; u < STEMP_TP * Wp * 8; u += 512) {
            const int c8 = u & 7, q = u >> 3;
            const int pr = q / Wp, pc = q - pr * Wp;
            if (p0 + pr >= Hp) continue;
            const char* src = band + ((2 * pr) * STEMP_BW + 2 * pc) * 128 + c8 * 16;
            u16x8 m = *reinterpret_cast<const u16x8*>(src);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    // (2 pc + dx <= Wo + 1 < STEMP_BW; the band's right padding columns are zero)
                    m = __builtin_elementwise_max(m, *reinterpret_cast<const u16x8*>(src + (dy * STEMP_BW + dx) * 128));
                }
            *reinterpret_cast<u16x8*>(y + (((long)b * Hp + p0 + pr) * Wp + pc) * 64 + c8 * 8) = m;
        }
        // (the next item's staging touches only the patch, which nobody reads after the barrier above; its convolution phase
        // rewrites the band behind the staging barrier, which every thread reaches with its pooling done)
    }
}

// fp32 parity path: direct convolution, one thread per (pixel, output channel); weights in OIHW
__global__ __launch_bounds__(256) void stem_direct_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ y, const float* __restrict__ centre,
                                                              int B, int Hin, int Win) {
    const int Ho = Hin / 2, Wo = Win / 2;
    const long total = (long)B * Ho * Wo * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i & 63);
        const long p = i >> 6;
        const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), b = (int)(p / ((long)Wo * Ho));
        float acc = centre ? -centre[co] : 0.f;
        for (int c = 0; c < 3; ++c)
            for (int ky = 0; ky < 7; ++ky) {
                const int yin = 2 * oy - 3 + ky;
                if (yin < 0 || yin >= Hin) continue;
                for (int kx = 0; kx < 7; ++kx) {
                    const int xin = 2 * ox - 3 + kx;
                    if (xin < 0 || xin >= Win) continue;
                    acc = fmaf(x[(((long)b * 3 + c) * Hin + yin) * Win + xin], w[((co * 3 + c) * 7 + ky) * 7 + kx], acc);
                }
            }
        y[i] = acc;
    }
}

// fp32 parity path, LDS-tiled (the direct kernel above stays as the fallback for shapes whose band does not fit): one workgroup =
// one band of TH output rows of one image.  The band's input rows (zero padding included) are staged per channel in LDS, the
// weights transposed to [tap][co]; a wave then takes (16-channel chunk, 64 output pixels) units: lane = pixel, 16 accumulators,
// per tap one LDS read of the input value and 16 FMAs against wave-uniform weights (broadcast 16-byte LDS reads).
__global__ __launch_bounds__(256) void stem_tiled_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ y, const float* __restrict__ centre,
                                                             int B, int Hin, int Win, int TH, int bands) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int Ho = Hin / 2, Wo = Win / 2;
    const int Wp = Win + 6, rows_in = 2 * TH + 5, plane = rows_in * Wp;
    float* ws = smf;                                  // [147][64]
    float* cs = ws + 147 * 64;                        // [64]  -centre
    float* xs = cs + 64;                              // [3][plane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 147 * 64; e += 256) {
        const int co = e / 147, tap = e - co * 147;   // (contiguous reads of the OIHW weights)
        ws[tap * 64 + co] = w[e];
    }
    if (tid < 64) cs[tid] = centre ? -centre[tid] : 0.f;
    for (int item = blockIdx.x; item < B * bands; item += gridDim.x) {
        const int b = item / bands, band = item - b * bands;
        const int oy0 = band * TH, iy0 = 2 * oy0 - 3;
        __syncthreads();                              // the previous item's readers are done (first item: weights visible)
        for (int e = tid; e < 3 * plane; e += 256) {
            const int c = e / plane, r = e - c * plane, ry = r / Wp, rx = r - ry * Wp;
            const int yin = iy0 + ry, xin = rx - 3;
            float v = 0.f;
            if ((unsigned)yin < (unsigned)Hin && (unsigned)xin < (unsigned)Win) v = x[(((long)b * 3 + c) * Hin + yin) * Win + xin];
            xs[e] = v;
        }
        __syncthreads();
        const int n_out = min(TH, Ho - oy0) * Wo, nblk = (n_out + 63) / 64;
        for (int u = wave; u < 4 * nblk; u += 4) {
            const int chunk = u / nblk, blk = u - chunk * nblk;
            const int q = blk * 64 + lane;
            const bool ok = q < n_out;
            const int qq = ok ? q : 0, ty = qq / Wo, ox = qq - ty * Wo;
            float acc[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = cs[chunk * 16 + o];
            const float* xb = xs + (2 * ty) * Wp + 2 * ox;
            const float* wb = ws + chunk * 16;
            for (int c = 0; c < 3; ++c)
                for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) {
                        const float xv = xb[c * plane + ky * Wp + kx];
                        const float* wt = wb + ((c * 7 + ky) * 7 + kx) * 64;
#pragma unroll
                        for (int o4 = 0; o4 < 4; ++o4) {
                            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wt + o4 * 4);
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc[o4 * 4 + k] = fmaf(xv, w4[k], acc[o4 * 4 + k]);
                        }
                    }
                }
            if (ok) {
                float* dst = y + (((long)b * Ho + oy0 + ty) * Wo + ox) * 64 + chunk * 16;
#pragma unroll
                for (int o4 = 0; o4 < 4; ++o4)
                    *reinterpret_cast<f32x4*>(dst + o4 * 4) = f32x4{acc[o4 * 4], acc[o4 * 4 + 1], acc[o4 * 4 + 2], acc[o4 * 4 + 3]};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BN + ReLU + maxpool 3x3 stride 2 pad 1 (NHWC), 8 channels per thread
// ------------------------------------------------------------------------------------------------
// output rows per thread (round 5): R vertically adjacent outputs share input rows -- 3 (2 R + 1) taps for R outputs instead of 9 R, and
// (2 R + 1) / R instead of 3 input rows fetched per output row across the L2s (the counters had FETCH_SIZE at 1.5 x the input:
// consecutive output rows belong to workgroups on different XCDs)
#ifndef CVCL_POOL_ROWS
#define CVCL_POOL_ROWS 2
#endif
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, T* __restrict__ y,
                                                              int B, int H, int W, int C) {
    constexpr int R = CVCL_POOL_ROWS, NR = 2 * R + 1, NTAP = 3 * NR;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, CC = C / 8;
    const int Hq = (Ho + R - 1) / R;
    const long total = (long)B * Hq * Wo * CC;
    // (measured in round 3: an XCD-major block order -- consecutive 32-pixel blocks on one XCD so that the input row two output
    // rows share is fetched by one L2 instead of two -- made this kernel 7 % SLOWER (0.132 -> 0.142 ms); plain order kept)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % CC);
        const long p = i / CC;
        const int ox = (int)(p % Wo), oq = (int)((p / Wo) % Hq), b = (int)(p / ((long)Wo * Hq));
        const int oy = R * oq;
        float sc[8], sh[8], m[R][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = scale[cc * 8 + e]; sh[e] = shift[cc * 8 + e];
#pragma unroll
            for (int r = 0; r < R; ++r) m[r][e] = 0.f;                           // relu output >= 0
        }
        // all taps (input rows 2 oy - 1 .. 2 oy + 2 R - 1) are loaded from clamped (always valid) addresses before any is used; a
        // clamped tap repeats a pixel of the window it belongs to, which leaves the maximum unchanged -- no branch around a load
        // (branches made every tap wait for the previous one).  Output r's window is rows 2 r .. 2 r + 2 of the NR; where an output
        // does not exist (Ho not a multiple of R) its rows clamp into the image and its result is not stored
        constexpr int EPC = ElemTraits<T>::kPerChunk, NC = 8 / EPC;              // 8 channels = 1 (bf16) or 2 (fp32) chunks
        Chunk<T> tap[NTAP][NC];
        const T* src[NTAP];
#pragma unroll
        for (int ky = 0; ky < NR; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                // only row 0 can fall above the image (-> row 0, inside output 0's window); rows below it clamp to H - 1, which lies
                // inside the window of the last existing output
                const int yin = min(max(2 * oy - 1 + ky, 0), H - 1);
                const int xin = min(max(2 * ox - 1 + kx, 0), W - 1);
                src[ky * 3 + kx] = x + (((long)b * H + yin) * W + xin) * C + cc * 8;
            }
        // (the scheduler had sunk every tap's load next to its use -- nine s_waitcnt vmcnt(0), nine dependent round trips per
        // output, in the ISA -- although the source issued them together; the barriers pin "all addresses, all loads, then math")
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
            for (int q = 0; q < NC; ++q) tap[t][q].load(src[t] + q * EPC);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = fmaf(tap[t][e / EPC].get(e % EPC), sc[e], sh[e]);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (t >= 6 * r && t < 6 * r + 9) m[r][e] = fmaxf(m[r][e], v);
            }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (oy + r < Ho) {
                T* dst = y + (((long)b * Ho + oy + r) * Wo + ox) * C + cc * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[e] = ElemTraits<T>::from_f(m[r][e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// grouped 3x3 convolution, pad 1, stride 1|2, NHWC raw in (BN+ReLU applied on load) -> NHWC raw out
// ------------------------------------------------------------------------------------------------
constexpr int GC_CS = 64;                  // channels per workgroup slab
#ifndef CVCL_GC_PIXB
#define CVCL_GC_PIXB 144
#endif
constexpr int GC_PIXB = CVCL_GC_PIXB;      // LDS bytes per staged pixel (128 B of channels + pad; an odd number of 16-byte slots)

struct GconvDev {
    const void* x; const float* a_scale; const float* a_shift; const void* w; void* y; float* stats;
    const float* centre;            // storage centre of the output (NULL = 0): the accumulators start at -centre[channel]
    int B, H, W, C, cg, stride, Ho, Wo, TH, bands, rows_in;
    float act_floor; // 0 = ReLU after the affine; -inf = none (a_scale == NULL: plain convolution of x, used by the data gradient)
    BnSrc src;       // src.acc != NULL: the input's BatchNorm affine is formed here from its producer's accumulators (finalize-on-load)
    int stats_acc;   // stats is an int64 accumulator [8][2][C] (cvcl_common.h), not partial rows
};
// Phase ablation for timing studies is a BUILD option (no run-time flag in the loop: the per-slot tests it needed cost branches in
// every work item): -DCVCL_GCONV_ABLATE=<bits>, 1 skip the BN math, 2 skip the MFMA loop (and the stores), 4 skip the stores,
// 8 skip the LDS staging writes -- e.g. CVCL_EXTRA_FLAGS=-DCVCL_GCONV_ABLATE=2 CVCL_LIB_SUFFIX=_abl2 python build.py.
#ifndef CVCL_GCONV_ABLATE
#define CVCL_GCONV_ABLATE 0
#endif

template <bool WIDE, int NPF, int NMT>
__global__ __launch_bounds__(256, WIDE ? 2 : 3) void gconv_mfma_kernel(GconvDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bf16_t* __restrict__ x = (const bf16_t*)p.x;
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pix = lane & 15, kb = lane >> 4;
    const int slab = blockIdx.y, c0 = slab * GC_CS;
    const int Wp = p.W + 2;                                       // staged row width incl. halo columns
    constexpr bool wide = WIDE;                                   // 32 channels per group (layer4) vs 4/8/16
    constexpr int ABL = CVCL_GCONV_ABLATE;
    constexpr int KS = WIDE ? 9 : 5;
    // this wave's unit: 16 output channels [c0 + 16*wave, +16); for cg == 32 the unit's inputs are the
    // 32 channels of its group, else the same 16 channels (block-diagonal weights)
    const int unit = c0 / 16 + wave;
    const bf16_t* wu = (const bf16_t*)p.w + (long)unit * KS * 512;
    bf16x8 wf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(wu + (ks * 16 + pix) * 32 + kb * 8);
    const int in_ch_off = wide ? ((wave >> 1) * 32 + kb * 8) : (wave * 16 + (kb & 1) * 8);   // within the slab

    // staging role: 8 chunks (of 8 channels) per pixel, 32 pixels per pass
    const int s_chunk = tid & 7, s_pix0 = tid >> 3;
    f32x2 sc[4], sh[4];                                           // (channel pairs: the staging math runs on packed fp32)
    if (p.src.acc) {
        // finalize-on-load: this slab's 64 input channels from the producing convolution's accumulators (the band buffers are not
        // in use yet); the workgroups of grid column 0 publish the moments
        float* aff = reinterpret_cast<float*>(smem);
        bn_slice_affine<GC_CS>(p.src, c0, blockIdx.x == 0, aff, aff + 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cl = s_chunk * 8 + 2 * e;
            sc[e] = f32x2{aff[cl], aff[cl + 1]};
            sh[e] = f32x2{aff[64 + cl], aff[64 + cl + 1]};
        }
        __syncthreads();                                          // (the band staging below overwrites the scratch)
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = c0 + s_chunk * 8 + 2 * e;
            sc[e] = p.a_scale ? f32x2{p.a_scale[ch], p.a_scale[ch + 1]} : f32x2{1.f, 1.f};
            sh[e] = p.a_scale ? f32x2{p.a_shift[ch], p.a_shift[ch + 1]} : f32x2{0.f, 0.f};
        }
    }
    const bool relu_in = p.act_floor == 0.f;                      // (-inf: no activation -- the data-gradient use)
    f32x2 ssum[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}, ssq[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};     // (channel pairs: packed fp32)

    const int npix_in = p.rows_in * Wp;
    const int n_out = p.TH * p.Wo, n_mt = cvcl_div_up(n_out, 16);
    // per-lane constants of the compute loop: LDS byte offset of each K step's tap (+ this lane's channel block),
    // first pixel's (row, column), output channel
    int tap_off[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        int tap = wide ? ks : 2 * ks + (kb >> 1);
        if (tap > 8) tap = 8;                                          // padded tap: its weights are zero
        const int ky = tap / 3, kx = tap - ky * 3;
        tap_off[ks] = (ky * Wp + kx) * GC_PIXB + in_ch_off * 2;
    }
    // the m-tiles (16 output pixels each) of a band are the same for every work item: per lane, the LDS byte offset of the tile's
    // first tap (< 64 KiB), its band row and whether it exists, packed offset | row << 16 | exists << 24 (NMT = 4 | 8 tiles:
    // gconv_plan keeps TH * Wo <= 128)
    int mtab[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        const int q = pix + 16 * mt;
        const bool ok = mt < n_mt && q < n_out;
        const int ty = q / p.Wo, ox = q - ty * p.Wo;
        mtab[mt] = ok ? ((((ty * p.stride) * Wp + ox * p.stride) * GC_PIXB) | (ty << 16) | (1 << 24)) : 0;
    }
    char* s_out = smem + npix_in * GC_PIXB;                            // output band [TH * Wo pixels][GC_PIXB]
    // software pipeline over work items: the next band's pixels are loaded into registers (raw, no waiting) before
    // the current band is multiplied out of LDS; BN+ReLU and the LDS write happen one iteration later.
    // NPF = ceil(staged pixels / 32) rounded up to an even count (template parameter: every slot is loaded, see below)
    u32x4 pf[NPF];
    bool pf_in[NPF];
    // Every slot issues its load unconditionally from a clamped (always valid) address and the predicate only decides later whether
    // the value or the zero padding is staged.  With the loads inside divergent `if`s the compiler put an s_waitcnt vmcnt(0) in
    // front of each of them: ten serialized memory round trips per work item -- the kernel's phases simply added up (179 us for
    // layer 1, of which 86 were this chain).
    // (row, column) of this thread's staging slots inside the band are the same for every work item, so everything about a slot
    // that does not depend on the item is computed once (the kernel is bound by VALU issue): its band row, the element offset of
    // its (clamped) input column, and whether it is inside the band and not horizontal padding.  A slot that stages padding
    // still loads -- from the clamped neighbour pixel, a line the band reads anyway (a far-away dummy address showed up as +39 %
    // HBM reads in the PMC pass)
    int slot_y[NPF], slot_xoff[NPF];
    bool slot_ok[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int pi = s_pix0 + 32 * i;
        const int pc = min(pi, npix_in - 1);                          // (slots past the band re-read its last pixel)
        const int ry = pc / Wp, xin = pc - ry * Wp - 1;
        slot_y[i] = ry;
        slot_ok[i] = pi < npix_in && xin >= 0 && xin < p.W;
        slot_xoff[i] = min(max(xin, 0), p.W - 1) * p.C;
    }
    const int row_elems = p.W * p.C;
    auto prefetch = [&](int item) {
        const int b = item / p.bands, band = item - b * p.bands;
        const int iy0 = band * p.TH * p.stride - 1;
        const bf16_t* xb = x + (long)b * p.H * p.W * p.C + c0 + s_chunk * 8;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int yin = iy0 + slot_y[i];
            pf_in[i] = slot_ok[i] && (unsigned)yin < (unsigned)p.H;
            const int yc = min(max(yin, 0), p.H - 1);
            pf[i] = *reinterpret_cast<const u32x4*>(xb + (yc * row_elems + slot_xoff[i]));
        }
    };
    f32x4 acc_init = {0.f, 0.f, 0.f, 0.f};                             // output channel c0 + wave*16 + kb*4 + e
    if (p.centre) {
        const f32x4 cv = *reinterpret_cast<const f32x4*>(p.centre + c0 + wave * 16 + kb * 4);
        acc_init = f32x4{-cv[0], -cv[1], -cv[2], -cv[3]};
    }
    const int n_items = p.B * p.bands;
    const int st_off = (tid >> 3) * p.C + (tid & 7) * 8;             // store phase: this thread's chunk inside a band's output
    // the weight fragments (loaded above) are ready from here on: without this the compiler, conservative across the loop's back
    // edge, waits on vmcnt before the first MFMAs of every work item -- i.e. for the next item's prefetch just issued
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0) only
    // Work items in XCD-major order: block (x, y) has linear id x + y * gridDim.x and runs on XCD (linear id) % 8, every XCD with
    // an L2 of its own; consecutive items are adjacent bands of one image and share their halo rows.  With gridDim.x a multiple
    // of 8 the XCD is x % 8, and XCD j takes the contiguous item range [j * gridDim.x / 8, (j + 1) * gridDim.x / 8) of every round
    // (plain order: a band's neighbours sat on other XCDs and the halo rows were fetched twice: 2.12 GB read for 1.95 in the PMC pass)
    const int bx = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    if (bx < n_items) prefetch(bx);
    for (int item = bx; item < n_items; item += gridDim.x) {
        const int b = item / p.bands, band = item - b * p.bands;
        const int oy0 = band * p.TH;
        __syncthreads();
        // staging without per-slot branches: every slot is transformed (relu(round(x * scale + shift)) on channel pairs:
        // v_pk_fma_f32, v_cvt_pk_bf16_f32, v_pk_max_i16), padding slots are then zeroed with a mask (zero padding lives in the
        // post-activation domain), and only the write itself is predicated (the last slot may lie past the band)
        auto stage = [&](auto RELU) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const int pi = s_pix0 + 32 * i;
                const unsigned keep = pf_in[i] ? 0xffffffffu : 0u;
                u32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned y = (ABL & 1) ? pf[i][e] : round2(__builtin_elementwise_fma(widen2(pf[i][e]), sc[e], sh[e]));
                    if constexpr (decltype(RELU)::value) y = relu2(y);
                    v[e] = y & keep;
                }
                if (!(ABL & 8) && pi < npix_in) *reinterpret_cast<u32x4*>(smem + pi * GC_PIXB + s_chunk * 16) = v;
            }
        };
        if (relu_in) stage(std::true_type{});
        else stage(std::false_type{});
        __syncthreads();
        if (item + (int)gridDim.x < n_items) prefetch(item + gridDim.x);
        if (ABL & 2) continue;
        // two independent m-tiles in flight per wave: their LDS reads and MFMA chains interleave
        const int rows_left = p.Ho - oy0;                            // (a last band may be partial)
#pragma unroll
        for (int mt = 0; mt < NMT; mt += 2) {
            if (mt >= n_mt) break;
            const int e0 = mtab[mt], e1 = mtab[mt + 1];
            const int base0 = e0 & 0xffff, base1 = e1 & 0xffff;
            const int q0 = pix + 16 * mt, q1 = q0 + 16;
            const bool wr0 = (e0 >> 24) && ((e0 >> 16) & 0xff) < rows_left, wr1 = (e1 >> 24) && ((e1 >> 16) & 0xff) < rows_left;
            f32x4 acc0 = acc_init, acc1 = acc_init;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(smem + base0 + tap_off[ks]);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(smem + base1 + tap_off[ks]);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], a0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], a1, acc1, 0, 0, 0);
            }
            if (wr0) {
                const u32x2 o = {round2(f32x2{acc0[0], acc0[1]}), round2(f32x2{acc0[2], acc0[3]})};
                *reinterpret_cast<u32x2*>(s_out + q0 * GC_PIXB + wave * 32 + kb * 8) = o;
#pragma unroll
                for (int e = 0; e < 2; ++e) { const f32x2 sv = widen2(o[e]); ssum[e] += sv; ssq[e] = __builtin_elementwise_fma(sv, sv, ssq[e]); }
            }
            if (wr1) {
                const u32x2 o = {round2(f32x2{acc1[0], acc1[1]}), round2(f32x2{acc1[2], acc1[3]})};
                *reinterpret_cast<u32x2*>(s_out + q1 * GC_PIXB + wave * 32 + kb * 8) = o;
#pragma unroll
                for (int e = 0; e < 2; ++e) { const f32x2 sv = widen2(o[e]); ssum[e] += sv; ssq[e] = __builtin_elementwise_fma(sv, sv, ssq[e]); }
            }
        }
        // the band's output sits in LDS as [pixel][64 channels]: write it out as full 128-byte pixel rows
        // (16 B per lane).  Scattered 8-byte stores straight from the MFMA layout cost more than the whole
        // load + compute phases together (ablation in DESIGN.md).
        __syncthreads();
        if (!(ABL & 4)) {
            const int valid_rows = min(p.TH, p.Ho - oy0);
            const int n_chunks = valid_rows * p.Wo * 8;
            // chunk tid + 256 k of the band (pixel (tid >> 3) + 32 k, channels 8 (tid & 7) ..): the element offset inside the band's
            // output rows is item-invariant up to the stride 32 C per k; only the band's base pointer changes per item
            bf16_t* yb = y + ((long)b * p.Ho + oy0) * p.Wo * p.C + c0 + st_off;
            const char* so = s_out + (tid >> 3) * GC_PIXB + (tid & 7) * 16;
#pragma unroll
            for (int k = 0; k < NMT / 2; ++k) {                   // NMT * 16 pixels * 8 chunks / 256 threads
                if (tid + 256 * k < n_chunks)
                    *reinterpret_cast<bf16x8*>(yb + (long)k * 32 * p.C) = *reinterpret_cast<const bf16x8*>(so + k * 32 * GC_PIXB);
            }
        }
    }
    // per-channel partial sums: reduce over the 16 pixel lanes; channel = c0 + wave*16 + kb*4 + e
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float s = ssum[e >> 1][e & 1], q = ssq[e >> 1][e & 1];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (pix == 0 && p.stats) {
            const int ch = c0 + wave * 16 + kb * 4 + e;
            cvcl_bn_stats_out(p.stats, p.stats_acc, blockIdx.x, p.C, ch, s, q);
        }
    }
}

// fp32 parity path: direct grouped convolution, weights in the reference OIHW layout [C][cg][3][3]
__global__ __launch_bounds__(256) void gconv_direct_f32_kernel(const float* __restrict__ x, const float* __restrict__ a_scale,
                                                               const float* __restrict__ a_shift, const float* __restrict__ w,
                                                               float* __restrict__ y, const float* __restrict__ centre,
                                                               int B, int H, int W, int C, int cg,
                                                               int stride, int Ho, int Wo, float act_floor) {
    const long total = (long)B * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % C);
        const long pp = i / C;
        const int ox = (int)(pp % Wo), oy = (int)((pp / Wo) % Ho), b = (int)(pp / ((long)Wo * Ho));
        const int g0 = (co / cg) * cg;
        float acc = centre ? -centre[co] : 0.f;
        for (int ci = 0; ci < cg; ++ci) {
            const float sc = a_scale ? a_scale[g0 + ci] : 1.f, sh = a_scale ? a_shift[g0 + ci] : 0.f;
            for (int ky = 0; ky < 3; ++ky) {
                const int yin = oy * stride - 1 + ky;
                if (yin < 0 || yin >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int xin = ox * stride - 1 + kx;
                    if (xin < 0 || xin >= W) continue;
                    const float v = fmaxf(fmaf(x[(((long)b * H + yin) * W + xin) * C + g0 + ci], sc, sh), act_floor);
                    acc = fmaf(v, w[((long)co * cg + ci) * 9 + ky * 3 + kx], acc);
                }
            }
        }
        y[i] = acc;
    }
}

// fp32 parity path, LDS-tiled, for the trunk's shapes (C % 64 == 0, 4 | 8 | 16 | 32 channels per group; anything else and bands that
// do not fit take the direct kernel above): one workgroup = one band of TH output rows of one image x one 64-channel slab.  The
// band's input (producer BN + ReLU applied, zero padding after it) is staged channel-major in LDS, the slab's weights transposed to
// [group][tap][ci][co]; a wave takes (group, 64 output pixels) units: lane = pixel, CG accumulators, per (tap, ci) one LDS read of
// the input value and CG FMAs against wave-uniform weights (broadcast 16-byte LDS reads).  ~60x the direct kernel at B = 256.
template <int CG>
__global__ __launch_bounds__(256) void gconv_tiled_f32_kernel(const float* __restrict__ x, const float* __restrict__ a_scale,
                                                              const float* __restrict__ a_shift, const float* __restrict__ w,
                                                              float* __restrict__ y, const float* __restrict__ centre,
                                                              int B, int H, int W, int C, int stride, int Ho, int Wo, int TH,
                                                              int bands, int plane_p, float act_floor) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    constexpr int NG = 64 / CG;
    const int Wp = W + 2, rows_in = (TH - 1) * stride + 3, plane = rows_in * Wp;
    float* ws = smf;                                  // [NG][9][CG ci][CG co]
    float* cs = ws + 64 * 9 * CG;                     // [64]  -centre
    float* xs = cs + 64;                              // [64][plane_p]  (plane_p = plane padded to 1 mod 8: staging writes spread over banks)
    const int c0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 64 * CG * 9; e += 256) {    // OIHW [C][CG][3][3], read contiguously
        const int co = e / (CG * 9), r = e - co * (CG * 9), ci = r / 9, tap = r - ci * 9;
        const int g = co / CG, o = co - g * CG;
        ws[((g * 9 + tap) * CG + ci) * CG + o] = w[(long)c0 * CG * 9 + e];
    }
    if (tid < 64) cs[tid] = centre ? -centre[c0 + tid] : 0.f;
    const int ch4 = tid & 15;                         // staging role: 4 channels of a pixel
    f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
    if (a_scale) {
        sc4 = *reinterpret_cast<const f32x4*>(a_scale + c0 + ch4 * 4);
        sh4 = *reinterpret_cast<const f32x4*>(a_shift + c0 + ch4 * 4);
    }
    for (int item = blockIdx.x; item < B * bands; item += gridDim.x) {
        const int b = item / bands, band = item - b * bands;
        const int oy0 = band * TH, iy0 = oy0 * stride - 1;
        __syncthreads();                              // the previous item's readers are done (first item: weights visible)
        for (int pix = tid >> 4; pix < plane; pix += 16) {
            const int ry = pix / Wp, rx = pix - ry * Wp;
            const int yin = iy0 + ry, xin = rx - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};           // zero padding lives in the post-activation domain
            if ((unsigned)yin < (unsigned)H && (unsigned)xin < (unsigned)W) {
                v = *reinterpret_cast<const f32x4*>(x + (((long)b * H + yin) * W + xin) * C + c0 + ch4 * 4);
                if (a_scale) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(v[k], sc4[k], sh4[k]), act_floor);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) xs[(ch4 * 4 + k) * plane_p + pix] = v[k];
        }
        __syncthreads();
        const int n_out = min(TH, Ho - oy0) * Wo, nblk = (n_out + 63) / 64;
        for (int u = wave; u < NG * nblk; u += 4) {
            const int g = u / nblk, blk = u - g * nblk;
            const int q = blk * 64 + lane;
            const bool ok = q < n_out;
            const int qq = ok ? q : 0, ty = qq / Wo, ox = qq - ty * Wo;
            float acc[CG];
#pragma unroll
            for (int o = 0; o < CG; ++o) acc[o] = cs[g * CG + o];
            const float* xb = xs + (g * CG) * plane_p + (ty * stride) * Wp + ox * stride;
            const float* wb = ws + g * 9 * CG * CG;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll 4
                for (int ci = 0; ci < CG; ++ci) {
                    const float xv = xb[ci * plane_p + ky * Wp + kx];
                    const float* wt = wb + (tap * CG + ci) * CG;
#pragma unroll
                    for (int o4 = 0; o4 < CG / 4; ++o4) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wt + o4 * 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[o4 * 4 + k] = fmaf(xv, w4[k], acc[o4 * 4 + k]);
                    }
                }
            }
            if (ok) {
                float* dst = y + (((long)b * Ho + oy0 + ty) * Wo + ox) * C + c0 + g * CG;
#pragma unroll
                for (int o4 = 0; o4 < CG / 4; ++o4)
                    *reinterpret_cast<f32x4*>(dst + o4 * 4) = f32x4{acc[o4 * 4], acc[o4 * 4 + 1], acc[o4 * 4 + 2], acc[o4 * 4 + 3]};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out = relu(raw * scale + shift + identity),  identity = idn  or  idn * idn_scale + idn_shift
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_add_relu_kernel(const T* __restrict__ raw, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const T* __restrict__ idn,
                                                          const float* __restrict__ idn_scale,
                                                          const float* __restrict__ idn_shift, T* __restrict__ out,
                                                          long rows, int C) {
    // The grid stride (gridDim * 256 chunks) is a multiple of the chunks per row (C / EPC divides 256 for every
    // ResNeXt width), so a thread always lands on the same channel chunk: the per-channel affine is loaded once.
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    const int CC = C / EPC;
    const long total = rows * CC;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(i0 % CC) * EPC;
    // per-channel affines as 16-byte loads from unconditional addresses (a `cond ? p[i] : 1.f` per element made the compiler emit
    // one s_waitcnt vmcnt(0) per channel: eight dependent L2 round trips at the start of every workgroup)
    float sc[EPC], sh[EPC], isc[EPC], ish[EPC];
    const float* isp = idn_scale ? idn_scale : scale;
    const float* ihp = idn_scale ? idn_shift : shift;
#pragma unroll
    for (int v4 = 0; v4 < EPC / 4; ++v4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(scale + c + 4 * v4), b = *reinterpret_cast<const f32x4*>(shift + c + 4 * v4);
        const f32x4 ia = *reinterpret_cast<const f32x4*>(isp + c + 4 * v4), ib = *reinterpret_cast<const f32x4*>(ihp + c + 4 * v4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[4 * v4 + e] = a[e];
            sh[4 * v4 + e] = b[e];
            isc[4 * v4 + e] = idn_scale ? ia[e] : 1.f;
            ish[4 * v4 + e] = idn_scale ? ib[e] : 0.f;
        }
    }
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = i0; i < total; i += 2 * stride) {          // two independent 16-B chunks in flight per thread
        const long j = i + stride;
        const bool has_j = j < total;
        Chunk<T> a0, d0, a1, d1, o;
        a0.load(raw + i * EPC);
        d0.load(idn + i * EPC);
        if (has_j) { a1.load(raw + j * EPC); d1.load(idn + j * EPC); }
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float v = fmaf(a0.get(e), sc[e], sh[e]);
            const float r = idn_scale ? fmaf(d0.get(e), isc[e], ish[e]) : d0.get(e);
            o.set(e, fmaxf(v + r, 0.f));
        }
        o.store(out + i * EPC);
        if (has_j) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float v = fmaf(a1.get(e), sc[e], sh[e]);
                const float r = idn_scale ? fmaf(d1.get(e), isc[e], ish[e]) : d1.get(e);
                o.set(e, fmaxf(v + r, 0.f));
            }
            o.store(out + j * EPC);
        }
    }
}

// y = relu(x * scale + shift) over [rows, C] (may run in place).  Used ahead of conv3: applying BatchNorm in the
// GEMM's operand load costs the affine once per 128-column tile of the output (2..16x redundant VALU work that
// also stalls the MFMA pipe), while this pass touches the narrow tensor exactly once.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, T* __restrict__ y,
                                                            long rows, int C) {
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    const int CC = C / EPC;
    const long total = rows * CC;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(i0 % CC) * EPC;                 // fixed per thread: (grid * 256) % CC == 0 (see launcher)
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int v4 = 0; v4 < EPC / 4; ++v4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(scale + c + 4 * v4), b = *reinterpret_cast<const f32x4*>(shift + c + 4 * v4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[4 * v4 + e] = a[e]; sh[4 * v4 + e] = b[e]; }
    }
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = i0; i < total; i += 2 * stride) {
        const long j = i + stride;
        const bool has_j = j < total;
        Chunk<T> a0, a1, o;
        a0.load(x + i * EPC);
        if (has_j) a1.load(x + j * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.set(e, fmaxf(fmaf(a0.get(e), sc[e], sh[e]), 0.f));
        o.store(y + i * EPC);
        if (has_j) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.set(e, fmaxf(fmaf(a1.get(e), sc[e], sh[e]), 0.f));
            o.store(y + j * EPC);
        }
    }
}

// global average pool: [B, HW, C] -> [B, C] fp32
template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, float* __restrict__ out, int B, int HW, int C) {
    const long total = (long)B * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long b = i / C;
        const float acc = ordered_sum<8, float>(HW, [&](int p) { return ElemTraits<T>::to_f(x[(b * HW + p) * C + c]); });
        out[i] = acc / (float)HW;
    }
}

int grid_for(long total, int per_block = 256, int cap = 4096) {
    long g = (total + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
static int bn_finalize_launch(const float* stats, int rows, long count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                              float eps, float* scale, float* shift, int C, float* moments, int moments_ld, const float* centre,
                              void* stream) {
    CVCL_CHECK_ARG(stats && gamma && beta && scale && shift && rows > 0 && count > 0 && C > 0, "cvcl_bn_finalize: bad args");
    CvclProfScope prof(stream, CVCL_K_BN_FINALIZE);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cvcl_div_up(C, 16)), dim3(1024), 0, (hipStream_t)stream, stats, rows,
                       (double)count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale,
                       shift, C, moments, moments_ld, centre);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_bn_finalize(const float* stats, int rows, long count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                float eps, float* scale, float* shift, const float* centre, int C, void* stream) {
    return bn_finalize_launch(stats, rows, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                              scale, shift, C, nullptr, 0, centre, stream);
}

extern "C" int cvcl_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* scale, float* shift, const float* centre, int C,
                                   void* stream) {
    CVCL_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && C > 0, "cvcl_bn_eval_affine: bad args");
    CvclProfScope prof(stream, CVCL_K_BN_FINALIZE);
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cvcl_div_up(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, scale, shift, centre, C);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_col_stats_rows(long rows) {
    long g = (rows + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > 256 ? 256 : g);
}

extern "C" int cvcl_col_stats(int dtype, const void* x, long rows, int C, float* stats, int stats_rows, void* stream) {
    CVCL_CHECK_ARG(x && stats && rows > 0 && C > 0, "cvcl_col_stats: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    const int g = cvcl_col_stats_rows(rows);
    CVCL_CHECK_ARG(stats_rows >= g, "cvcl_col_stats: stats_rows %d < %d", stats_rows, g);
    dim3 grid(g, cvcl_div_up(C, 64));
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(col_stats_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, rows, C, stats);
    else
        hipLaunchKernelGGL(col_stats_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, rows, C, stats);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" size_t cvcl_packed_weight_bytes(int dtype, int kind, int cout, int cin_per_group, int k) {
    const size_t es = dtype == CVCL_BF16 ? 2 : 4;
    if (dtype == CVCL_F32 || kind == CVCL_PACK_DENSE) return (size_t)cout * cin_per_group * k * k * es;
    if (kind == CVCL_PACK_STEM7) return (size_t)4 * 6 * 512 * 2;
    if (kind == CVCL_PACK_GCONV3) return (size_t)(cout / 16) * (cin_per_group == 32 ? 9 : 5) * 512 * 2;
    return 0;
}

extern "C" int cvcl_pack_conv_weight(int dtype, int kind, const float* w_oihw, void* out, int cout, int cin_per_group,
                                     int k, void* stream) {
    CVCL_CHECK_ARG(w_oihw && out && cout > 0 && cin_per_group > 0, "cvcl_pack_conv_weight: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)cout * cin_per_group * k * k;
    if (dtype == CVCL_F32) {                       // parity mode keeps the reference layout
        hipLaunchKernelGGL(cast_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, w_oihw, (float*)out, n);
    } else if (kind == CVCL_PACK_DENSE) {
        hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, w_oihw, (bf16_t*)out, n);
    } else if (kind == CVCL_PACK_STEM7) {
        CVCL_CHECK_ARG(cout == 64 && cin_per_group == 3 && k == 7, "cvcl_pack_conv_weight: stem must be 64x3x7x7");
        hipLaunchKernelGGL(pack_stem_kernel, dim3(cvcl_div_up(4 * 6 * 512, 256)), dim3(256), 0, s, w_oihw, (bf16_t*)out);
    } else if (kind == CVCL_PACK_GCONV3) {
        CVCL_CHECK_ARG(k == 3 && (cin_per_group == 4 || cin_per_group == 8 || cin_per_group == 16 || cin_per_group == 32) &&
                           cout % 64 == 0, "cvcl_pack_conv_weight: unsupported grouped conv %d/%d", cout, cin_per_group);
        const long total = (long)(cout / 16) * (cin_per_group == 32 ? 9 : 5) * 512;
        hipLaunchKernelGGL(pack_gconv_kernel, dim3(cvcl_div_up(total, 256)), dim3(256), 0, s, w_oihw, (bf16_t*)out, cout,
                           cin_per_group);
    } else {
        cvcl_set_error("cvcl_pack_conv_weight: unknown kind %d", kind);
        return CVCL_EINVAL;
    }
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

static int stem_grid(int B, int Hin) {
    const int items = B * cvcl_div_up(Hin / 2, STEM_TH);
    return items < 512 ? items : 512;
}

extern "C" int cvcl_stem_conv_stats_rows(int dtype, int B, int H, int W) {
    if (dtype == CVCL_BF16) return stem_grid(B, H);
    return cvcl_col_stats_rows((long)B * (H / 2) * (W / 2));
}

extern "C" int cvcl_stem_conv7x7(int dtype, const float* x_nchw, const void* w_packed, void* y_nhwc, float* stats,
                                 int stats_rows, const float* centre, int B, int H, int W, void* stream) {
    CVCL_CHECK_ARG(x_nchw && w_packed && (y_nhwc || (dtype == CVCL_BF16 && stats)) && B > 0 && H % 2 == 0 && W % 2 == 0,
                   "cvcl_stem_conv7x7: bad args");                 // (bf16: y_nhwc NULL = statistics only)
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CVCL_BF16) {
        CVCL_CHECK_ARG(W + 6 + 8 <= 2 * STEM_PITCH && W % 4 == 0 && W / 4 <= 64,
                       "cvcl_stem_conv7x7: width %d not supported by the staged patch", W);
        const int g = stem_grid(B, H);
        CVCL_CHECK_ARG(!stats || stats_rows >= g, "cvcl_stem_conv7x7: stats_rows %d < %d", stats_rows, g);
        const size_t lds = (size_t)3 * STEM_ROWS * STEM_PITCH * 4 + 4 * 16 * STEM_OPITCH + 64 * 4;
        float* st = stats;
        CVCL_CHECK_ARG(st, "cvcl_stem_conv7x7: the bf16 kernel always emits statistics; pass a buffer");
        CvclProfScope prof(stream, CVCL_K_STEM);
        hipLaunchKernelGGL(stem_mfma_kernel, dim3(g), dim3(256), lds, s, x_nchw, (const bf16_t*)w_packed, (bf16_t*)y_nhwc,
                           st, centre, B, H, W);
        CVCL_LAUNCH_CHECK();
        return CVCL_OK;
    }
    const long total = (long)B * (H / 2) * (W / 2) * 64;
    {
        CvclProfScope prof(stream, CVCL_K_STEM);
        // LDS-tiled kernel when a band of >= 1 output row fits (weights 37 KB + 3 x (2 TH + 5) x (W + 6) floats)
        const int Ho = H / 2;
        auto lds_of = [&](int th) { return (size_t)(147 * 64 + 64 + 3 * (2 * th + 5) * (W + 6)) * 4; };
        int TH = Ho < 8 ? Ho : 8;
        while (TH > 1 && lds_of(TH) > 150 * 1024) --TH;
        static const bool tiled_on = cvcl_env_on("CVCL_F32_TILED");
        if (tiled_on && lds_of(TH) <= 150 * 1024) {
            static CvclLdsAttr attr;
            if (!attr.ready()) {
                if (hipFuncSetAttribute((const void*)stem_tiled_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                    cvcl_set_error("cvcl_stem_conv7x7: cannot raise the dynamic LDS limit");
                    return CVCL_ELAUNCH;
                }
                attr.mark();
            }
            const int bands = cvcl_div_up(Ho, TH);
            const long items = (long)B * bands;
            hipLaunchKernelGGL(stem_tiled_f32_kernel, dim3((unsigned)(items < 1024 ? items : 1024)), dim3(256), lds_of(TH), s, x_nchw,
                               (const float*)w_packed, (float*)y_nhwc, centre, B, H, W, TH, bands);
        } else {
            hipLaunchKernelGGL(stem_direct_f32_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, s, x_nchw,
                               (const float*)w_packed, (float*)y_nhwc, centre, B, H, W);
        }
    }
    CVCL_LAUNCH_CHECK();
    if (stats) return cvcl_col_stats(CVCL_F32, y_nhwc, (long)B * (H / 2) * (W / 2), 64, stats, stats_rows, stream);
    return CVCL_OK;
}

// conv1 7x7/2 + bn1 + relu + maxpool 3x3/2 in one pass (stem_pool_mfma_kernel; bf16): x NCHW fp32 -> y NHWC bf16 [B, H/4, W/4, 64]
extern "C" int cvcl_stem_pool_supported(int dtype, int H, int W) {
    return dtype == CVCL_BF16 && H % 2 == 0 && W % 4 == 0 && W / 4 <= 64 && W + 6 + 8 <= 2 * STEM_PITCH && W / 2 + 2 <= STEMP_BW;
}

extern "C" int cvcl_stem_pool(int dtype, const float* x_nchw, const void* w_packed, const float* scale, const float* shift,
                              const float* centre, void* y_nhwc, int B, int H, int W, void* stream) {
    CVCL_CHECK_ARG(x_nchw && w_packed && scale && shift && y_nhwc && B > 0, "cvcl_stem_pool: bad args");
    CVCL_CHECK_ARG(cvcl_stem_pool_supported(dtype, H, W), "cvcl_stem_pool: bf16 and a width <= %d only (got dtype %d, %d x %d)",
                   2 * (STEMP_BW - 2), dtype, H, W);
    const size_t lds = (size_t)3 * STEMP_ROWS * STEM_PITCH * 4 + (size_t)STEMP_CR * STEMP_BW * 128 + 3 * 64 * 4;
    static CvclLdsAttr attr;
    if (!attr.ready()) {
        if (hipFuncSetAttribute((const void*)stem_pool_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            cvcl_set_error("cvcl_stem_pool: cannot raise the dynamic LDS limit");
            return CVCL_ELAUNCH;
        }
        attr.mark();
    }
    const int Hp = (H / 2 - 1) / 2 + 1;
    const int items = B * cvcl_div_up(Hp, STEMP_TP);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int grid = items < cus ? items : cus;
    CvclProfScope prof(stream, CVCL_K_STEM);
    hipLaunchKernelGGL(stem_pool_mfma_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, x_nchw, (const bf16_t*)w_packed,
                       (bf16_t*)y_nhwc, scale, shift, centre, B, H, W);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_bn_relu_maxpool(int dtype, const void* x, const float* scale, const float* shift, void* y, int B,
                                    int H, int W, int C, void* stream) {
    CVCL_CHECK_ARG(x && scale && shift && y && C % 8 == 0, "cvcl_bn_relu_maxpool: bad args");
    CvclProfScope prof(stream, CVCL_K_MAXPOOL);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * ((Ho + CVCL_POOL_ROWS - 1) / CVCL_POOL_ROWS) * Wo * (C / 8);          // a thread owns CVCL_POOL_ROWS output rows
    const int grid = grid_for(total, 256, 8192);
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(bn_relu_maxpool_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, scale, shift, (float*)y, B, H, W, C);
    else
        hipLaunchKernelGGL(bn_relu_maxpool_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, scale, shift, (bf16_t*)y, B, H, W, C);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

namespace {
struct GconvPlan { int TH, bands, rows_in, grid_x; size_t lds; };
GconvPlan gconv_plan(int B, int H, int W, int C, int stride) {
    GconvPlan g;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1, Wp = W + 2;
    // output rows per work item: staged input band + output band within ~52 KiB (3 workgroups per CU) and the
    // input pixel count within the 10 x 32 register-prefetch slots of the kernel
    auto bytes = [&](int th) { return (size_t)(((th - 1) * stride + 3) * Wp + th * Wo) * GC_PIXB; };
    static const int lds_kb = cvcl_lab_int("CVCL_GCONV_LDS_KB", 52);
    int TH = Ho;
    while (TH > 1 && (bytes(TH) > (size_t)lds_kb * 1024 || ((TH - 1) * stride + 3) * Wp > 10 * 32 || TH * Wo > 128)) TH = (TH + 1) / 2;
    g.TH = TH;
    g.bands = cvcl_div_up(Ho, TH);
    g.rows_in = (TH - 1) * stride + 3;
    g.lds = bytes(TH);
    // persistent grid = what is co-resident (no second round of workgroups): LDS- and register-limited to 3 per CU.
    // (Round 2 built a pipelined form -- one 512-thread workgroup per CU, double-buffered input / output bands, one barrier per
    // band, the two wave halves running store / stage / load and MFMA in opposite order: correct on every test and SLOWER,
    // 1.47 vs 1.05 ms per step.  Three small workgroups per CU hide each other's phases better than one pipelined one; removed.)
    int per_cu = (int)((160 * 1024) / g.lds);
    const int reg_limit = (C / 32 == 32) ? 2 : 3;                     // launch bounds of the two kernel variants
    if (per_cu > reg_limit) per_cu = reg_limit;
    if (per_cu < 1) per_cu = 1;
    const int slabs = C / GC_CS;
    int gx = per_cu * 256 / slabs;
    if (gx < 1) gx = 1;
    const int items = B * g.bands;
    g.grid_x = items < gx ? items : gx;
    return g;
}
}  // namespace

extern "C" int cvcl_gconv3x3_stats_rows(int dtype, int B, int H, int W, int C, int stride) {
    if (dtype == CVCL_BF16) return gconv_plan(B, H, W, C, stride).grid_x;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    return cvcl_col_stats_rows((long)B * Ho * Wo);
}

// src != NULL (bf16 only, internal: the trunk's launch sequence): the input's BatchNorm affine comes from its producer's partial
// rows inside the kernel (finalize-on-load) instead of from a_scale / a_shift
static int gconv3x3_impl(int dtype, const void* x, const float* a_scale, const float* a_shift, const BnSrc* src, const void* w_packed,
                         void* y, float* stats, int stats_rows, const float* centre, int B, int H, int W, int C, int groups,
                         int stride, void* stream) {
    CVCL_CHECK_ARG(x && w_packed && y && (!a_scale == !a_shift), "cvcl_gconv3x3: null pointer");
    CVCL_CHECK_ARG(!src || (dtype == CVCL_BF16 && src->acc && src->gamma && src->beta && src->C == C && src->count > 0 &&
                            (const void*)src->acc != (const void*)stats),
                   "cvcl_gconv3x3: finalize-on-load source");
    CVCL_CHECK_ARG(stats_rows != CVCL_STATS_ACCUMULATE || dtype == CVCL_BF16, "cvcl_gconv3x3: accumulators exist for bf16 only");
    const float act_floor = (a_scale || src) ? 0.f : -INFINITY;
    CVCL_CHECK_ARG(B > 0 && (stride == 1 || stride == 2) && groups > 0 && C % groups == 0, "cvcl_gconv3x3: bad shape");
    const int cg = C / groups;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CVCL_BF16) {
        CVCL_CHECK_ARG((cg == 4 || cg == 8 || cg == 16 || cg == 32) && C % GC_CS == 0,
                       "cvcl_gconv3x3: unsupported channels-per-group %d (C=%d)", cg, C);
        const GconvPlan g = gconv_plan(B, H, W, C, stride);
        CVCL_CHECK_ARG(g.lds <= 160 * 1024 && g.rows_in * (W + 2) <= 10 * 32 && g.TH * ((W - 1) / stride + 1) <= 128,
                       "cvcl_gconv3x3: feature map too wide for one staged band (%zu B, %d pixels)", g.lds, g.rows_in * (W + 2));
        CVCL_CHECK_ARG(!stats || stats_rows == CVCL_STATS_ACCUMULATE || stats_rows >= g.grid_x, "cvcl_gconv3x3: stats_rows %d < %d", stats_rows, g.grid_x);
        GconvDev d;
        d.x = x; d.a_scale = a_scale; d.a_shift = a_shift; d.w = w_packed; d.y = y; d.stats = stats; d.centre = centre;
        d.B = B; d.H = H; d.W = W; d.C = C; d.cg = cg; d.stride = stride; d.Ho = Ho; d.Wo = Wo;
        d.TH = g.TH; d.bands = g.bands; d.rows_in = g.rows_in;
        d.act_floor = act_floor;
        d.src = BnSrc{};
        d.stats_acc = stats && stats_rows == CVCL_STATS_ACCUMULATE;
        if (src) {
            CVCL_CHECK_ARG(g.lds >= 2048, "cvcl_gconv3x3: band buffers smaller than the finalize-on-load scratch");
            d.src = *src;
        }
        CvclProfScope prof(stream, CVCL_K_GCONV);
        int rc;
        const int slots = cvcl_div_up(g.rows_in * (W + 2), 32);
        static CvclLdsAttr attr_set[2][2][11];                 // per instantiation (wide, short m-tile table, slots)
        auto launch = [&](auto kern) -> int {
            CvclLdsAttr& done = attr_set[cg == 32][g.TH * Wo <= 64][slots <= 4 ? 4 : slots <= 6 ? 6 : slots <= 8 ? 8 : 10];
            if (!done.ready()) {
                if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                    cvcl_set_error("cvcl_gconv3x3: cannot raise the dynamic LDS limit");
                    return CVCL_ELAUNCH;
                }
                done.mark();
            }
            hipLaunchKernelGGL(kern, dim3(g.grid_x, C / GC_CS), dim3(256), g.lds, s, d);
            return CVCL_OK;
        };
        const bool few = g.TH * Wo <= 64;                    // <= 4 m-tiles per band: the short m-tile table
#define CVCL_GCONV_PICK(W_) \
        (slots <= 4 ? (few ? launch(gconv_mfma_kernel<W_, 4, 4>) : launch(gconv_mfma_kernel<W_, 4, 8>)) \
         : slots <= 6 ? (few ? launch(gconv_mfma_kernel<W_, 6, 4>) : launch(gconv_mfma_kernel<W_, 6, 8>)) \
         : slots <= 8 ? (few ? launch(gconv_mfma_kernel<W_, 8, 4>) : launch(gconv_mfma_kernel<W_, 8, 8>)) \
                      : (few ? launch(gconv_mfma_kernel<W_, 10, 4>) : launch(gconv_mfma_kernel<W_, 10, 8>)))
        rc = cg == 32 ? CVCL_GCONV_PICK(true) : CVCL_GCONV_PICK(false);
#undef CVCL_GCONV_PICK
        if (rc) return rc;
        CVCL_LAUNCH_CHECK();
        return CVCL_OK;
    }
    const long total = (long)B * Ho * Wo * C;
    {
        CvclProfScope prof(stream, CVCL_K_GCONV);
        // LDS-tiled kernel for the trunk's shapes when a band fits: weights 64 x 9 x cg floats + 64 channel planes of the band
        auto plane_p_of = [&](int th) { const int pl = ((th - 1) * stride + 3) * (W + 2); return pl + ((9 - (pl & 7)) & 7); };   // = 1 mod 8
        auto lds_of = [&](int th) { return (size_t)(64 * 9 * cg + 64 + 64 * plane_p_of(th)) * 4; };
        int TH = Ho;
        while (TH > 1 && lds_of(TH) > 150 * 1024) --TH;
        static const bool tiled_on = cvcl_env_on("CVCL_F32_TILED");
        const bool tiled = tiled_on && C % 64 == 0 && (cg == 4 || cg == 8 || cg == 16 || cg == 32) && lds_of(TH) <= 150 * 1024 &&
                           ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)a_scale & 15) == 0 && ((uintptr_t)a_shift & 15) == 0;
        if (tiled) {
            const int bands = cvcl_div_up(Ho, TH);
            const long items = (long)B * bands;
            const int slabs = C / 64;
            long gx = 1024 / slabs;                          // ~4 workgroups per CU in total; each stages its weights once
            if (gx < 1) gx = 1;
            if (gx > items) gx = items;
            static CvclLdsAttr attr[4];
            auto launch = [&](auto kern, int slot) -> int {
                if (!attr[slot].ready()) {
                    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                        cvcl_set_error("cvcl_gconv3x3: cannot raise the dynamic LDS limit");
                        return CVCL_ELAUNCH;
                    }
                    attr[slot].mark();
                }
                hipLaunchKernelGGL(kern, dim3((unsigned)gx, slabs), dim3(256), lds_of(TH), s, (const float*)x, a_scale, a_shift,
                                   (const float*)w_packed, (float*)y, centre, B, H, W, C, stride, Ho, Wo, TH, bands, plane_p_of(TH), act_floor);
                return CVCL_OK;
            };
            const int rc = cg == 4 ? launch(gconv_tiled_f32_kernel<4>, 0) : cg == 8 ? launch(gconv_tiled_f32_kernel<8>, 1)
                         : cg == 16 ? launch(gconv_tiled_f32_kernel<16>, 2) : launch(gconv_tiled_f32_kernel<32>, 3);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL(gconv_direct_f32_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, s, (const float*)x, a_scale,
                               a_shift, (const float*)w_packed, (float*)y, centre, B, H, W, C, cg, stride, Ho, Wo, act_floor);
        }
    }
    CVCL_LAUNCH_CHECK();
    if (stats) return cvcl_col_stats(CVCL_F32, y, (long)B * Ho * Wo, C, stats, stats_rows, stream);
    return CVCL_OK;
}

extern "C" int cvcl_gconv3x3(int dtype, const void* x, const float* a_scale, const float* a_shift, const void* w_packed,
                             void* y, float* stats, int stats_rows, const float* centre, int B, int H, int W, int C, int groups,
                             int stride, void* stream) {
    return gconv3x3_impl(dtype, x, a_scale, a_shift, nullptr, w_packed, y, stats, stats_rows, centre, B, H, W, C, groups, stride, stream);
}

extern "C" int cvcl_bn_add_relu(int dtype, const void* raw, const float* scale, const float* shift, const void* idn,
                                const float* idn_scale, const float* idn_shift, void* out, long rows, int C, void* stream) {
    CVCL_CHECK_ARG(raw && scale && shift && idn && out && rows > 0 && C % 8 == 0, "cvcl_bn_add_relu: bad args");
    CvclProfScope prof(stream, CVCL_K_BN_ADD_RELU);
    CVCL_CHECK_ARG((idn_scale == nullptr) == (idn_shift == nullptr), "cvcl_bn_add_relu: idn_scale/idn_shift pair");
    const int epc = dtype == CVCL_F32 ? 4 : 8;
    CVCL_CHECK_ARG(C % epc == 0, "cvcl_bn_add_relu: C must be a multiple of %d", epc);
    hipStream_t s = (hipStream_t)stream;
    // the kernel needs (grid * 256) % (C / epc) == 0 so that a thread keeps its channel chunk across iterations
    const int cc = C / epc;
    int g256 = cc, t256 = 256;
    while (t256) { const int r = g256 % t256; g256 = t256; t256 = r; }      // gcd(cc, 256)
    const int mult = cc / g256;
    int grid = grid_for(rows * (long)cc, 256 * 4, 16384);
    grid = (grid + mult - 1) / mult * mult;
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(bn_add_relu_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)raw,
                           scale, shift, (const float*)idn, idn_scale, idn_shift, (float*)out, rows, C);
    else
        hipLaunchKernelGGL(bn_add_relu_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)raw,
                           scale, shift, (const bf16_t*)idn, idn_scale, idn_shift, (bf16_t*)out, rows, C);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_bn_relu_apply(int dtype, const void* x, const float* scale, const float* shift, void* y, long rows,
                                  int C, void* stream) {
    CVCL_CHECK_ARG(x && scale && shift && y && rows > 0, "cvcl_bn_relu_apply: bad args");
    CvclProfScope prof(stream, CVCL_K_BN_APPLY);
    const int epc = dtype == CVCL_F32 ? 4 : 8;
    CVCL_CHECK_ARG(C % epc == 0, "cvcl_bn_relu_apply: C must be a multiple of %d", epc);
    const int cc = C / epc;
    int g256 = cc, t256 = 256;
    while (t256) { const int r = g256 % t256; g256 = t256; t256 = r; }
    const int mult = cc / g256;
    int grid = grid_for(rows * (long)cc, 256 * 4, 16384);
    grid = (grid + mult - 1) / mult * mult;
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(bn_relu_apply_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, scale,
                           shift, (float*)y, rows, C);
    else
        hipLaunchKernelGGL(bn_relu_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           scale, shift, (bf16_t*)y, rows, C);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_avgpool(int dtype, const void* x, float* out, int B, int HW, int C, void* stream) {
    CVCL_CHECK_ARG(x && out && B > 0 && HW > 0 && C > 0, "cvcl_avgpool: bad args");
    CvclProfScope prof(stream, CVCL_K_AVGPOOL);
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(avgpool_kernel<float>, dim3(grid_for((long)B * C)), dim3(256), 0, (hipStream_t)stream, (const float*)x, out, B, HW, C);
    else
        hipLaunchKernelGGL(avgpool_kernel<bf16_t>, dim3(grid_for((long)B * C)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, B, HW, C);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

// ------------------------------------------------------------------------------------------------
// whole-network forward (one C call enqueues every kernel of the trunk on the caller's stream)
// ------------------------------------------------------------------------------------------------
namespace {
struct Spec { int cin, cout, k, stride, groups; };
constexpr int kLayers[4] = {3, 4, 6, 3};
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline size_t act_elems(int B, int H, int W) {
    // largest activation: stem output [B, H/2, W/2, 64] == layer1 tensors [B, H/4, W/4, 256]
    return (size_t)B * (H / 2) * (W / 2) * 64;
}
}  // namespace

extern "C" size_t cvcl_resnext50_workspace_bytes(int dtype, int B, int H, int W) {
    const size_t es = dtype == CVCL_BF16 ? 2 : 4;
    return 5 * al256(act_elems(B, H, W) * es) + al256((size_t)kMaxStatsRows * 2 * 2048 * 4) + al256((size_t)53 * 2 * 2048 * 4) +
           al256((size_t)53 * 2048 * 4) + al256(cvcl_conv1x1_gram_workspace_bytes(256)) + al256(53 * kTrunkAccLayer * 8);
}

extern "C" size_t cvcl_resnext50_centres_floats(void) { return (size_t)53 * 2048; }

// ------------------------------------------------------------------------------------------------
// One Bottleneck (torchvision Bottleneck.forward: conv1-bn1-relu, conv2(grouped 3x3, stride)-bn2-relu, conv3-bn3,
// + identity / downsample, relu) as the launch sequence the whole-trunk call uses.  L = this block's layers
// (conv1, conv2, conv3[, downsample]); aff = their (scale, shift) scratch, 4096 floats per layer; mom = where the
// deferred-statistics pass leaves the batch moments (or NULL: running statistics updated in place).
// ------------------------------------------------------------------------------------------------
namespace {
struct BlockCtx {
    int dtype, B, training;
    float momentum, eps;
    float* stats;
    void* gram_ws;                   // cvcl_conv1x1_gram workspace (K = 256)
    void* stream;
    long long* acc;                  // BatchNorm accumulators of the block's layers [n_layers][8][2][2048], zeroed for this pass; or NULL
};
constexpr size_t kAccLayer = kTrunkAccLayer;

// cen = the block's storage centres ([n_layers][2048] floats, "Centred storage" in cvcl_hip.h) or NULL
int bottleneck_fwd(const BlockCtx& c, int stage, bool first, int h, int wd, const char* X, char* R1, char* R2, char* R3, char* RD,
                   char* dst, const cvcl_convbn_params* L, float* aff, float* mom, const float* cen) {
    const int dtype = c.dtype, B = c.B, training = c.training;
    float* stats = c.stats;
    void* stream = c.stream;
    const int planes = 64 << stage, width = planes * 2, outc = planes * 4;
    const int inplanes = first ? (stage == 0 ? 64 : outc / 2) : outc;
    const int stride = (stage > 0 && first) ? 2 : 1;
    const int ho = h / stride, wo = wd / stride;
    const long m_in = (long)B * h * wd, m_out = (long)B * ho * wo;
    const int l1 = 0, l2 = 1, l3 = 2, ld = 3;
    int rc;
    auto scale_of = [&](int l) { return aff + (size_t)l * 4096; };
    auto shift_of = [&](int l) { return aff + (size_t)l * 4096 + 2048; };
    auto centre_of = [&](int l) -> const float* { return cen ? cen + (size_t)l * 2048 : nullptr; };
    // [lab: CVCL_SKIP_FINALIZE_AFTER=n -- upper bound on what removing the finalize chain could buy: after n finalize calls the
    //  launches are skipped and the consumers read the (scale, shift) of an earlier pass; only meaningful on a repeated batch]
    static const int skip_after = cvcl_lab_int("CVCL_SKIP_FINALIZE_AFTER", 0);
    static long finalize_calls = 0;
    const int stats_cap = kMaxStatsRows;
    auto finalize = [&](int l, int rows, long count, int C) -> int {
        if (skip_after > 0 && ++finalize_calls > skip_after) return CVCL_OK;
        if (training)
            return bn_finalize_launch(stats, rows, count, L[l].gamma, L[l].beta, L[l].running_mean, L[l].running_var,
                                      L[l].num_batches_tracked, c.momentum, c.eps, scale_of(l), shift_of(l), C,
                                      mom ? mom + (size_t)l * 4096 : nullptr, 2048, centre_of(l), stream);
        return CVCL_OK;                                   // eval mode: every layer's affine was produced up front
    };
    // finalize-on-load (csrc/resnext.hip "finalize-on-load"): layer l's convolution ACCUMULATES its statistics into acc_of(l) and the
    // consumer of its raw output forms the affine itself -- no cvcl_bn_finalize launch for that layer
    auto acc_of = [&](int l) { return c.acc + (size_t)l * kAccLayer; };
    auto bn_src = [&](int l, long count, int C) {
        BnSrc b;
        b.acc = acc_of(l); b.C = C; b.count = (double)count;
        b.gamma = L[l].gamma; b.beta = L[l].beta; b.centre = centre_of(l);
        b.running_mean = L[l].running_mean; b.running_var = L[l].running_var; b.nbt = L[l].num_batches_tracked;
        b.momentum = c.momentum; b.eps = c.eps;
        b.moments = mom ? mom + (size_t)l * 4096 : nullptr; b.moments_ld = 2048;
        b.scale_out = scale_of(l); b.shift_out = shift_of(l);
        return b;
    };
    static const bool fol_on = cvcl_env_on("CVCL_FINALIZE_ON_LOAD");               // (0: partial rows + a bn_finalize launch per layer)
    const bool use_acc = fol_on && c.acc && training && dtype == CVCL_BF16 && skip_after == 0;
    // (which forms this block takes -- the explanations sit at the launches below)
    static const int pro_stages = cvcl_lab_int("CVCL_CONV3_PRO_STAGES", 2);
    const bool pro = dtype == CVCL_BF16 && stage < pro_stages && (width == 128 || width == 256);
    static const int fused_stages = cvcl_lab_int("CVCL_FUSED_TAIL_STAGES", 2);
    const bool fused_tail = dtype == CVCL_BF16 && (stage < fused_stages || !training);
    static const bool ds_recompute_on = cvcl_lab_int("CVCL_DS_RECOMPUTE", 1) != 0 && cvcl_env_on("CVCL_GEMM_PRO");
    const bool ds_recompute = ds_recompute_on && first && stride == 1 && inplanes == 64 && fused_tail && pro && width == 128;
    static const bool gram_on = cvcl_lab_int("CVCL_BN_GRAM", 1) != 0;
    auto gram_stats = [&](int l, const void* A, int K, const float* a_scale, const float* a_shift, int a_relu, const BnSrc* src = nullptr) -> int {
        if (skip_after > 0 && ++finalize_calls > skip_after) return CVCL_OK;          // [lab: the Gram launches go as well]
        const double* g = nullptr;
        int r = cvcl_conv1x1_gram_src(A, K, m_out, K, src ? nullptr : a_scale, src ? nullptr : a_shift, src, a_relu, c.gram_ws,
                                      cvcl_conv1x1_gram_workspace_bytes(256), &g, stream);
        if (r) return r;
        return cvcl_bn_from_gram(g, K, m_out, L[l].w, K, outc, L[l].gamma, L[l].beta, L[l].running_mean, L[l].running_var,
                                 L[l].num_batches_tracked, c.momentum, c.eps, scale_of(l), shift_of(l), mom ? mom + (size_t)l * 4096 : nullptr,
                                 2048, centre_of(l), stream);
    };
    // The downsample branch of a stage's first block runs FIRST: its operand X was just written by the previous kernel and still
    // sits in the Infinity Cache (after conv1 / conv2 / conv3 it no longer does).
    // layer1.0 (bf16, fused tail, BN-prologue kernel): the branch is a K = 64 product of the block input, recomputed inside the tail
    // pass (gemm_pro.hip PRO_TAIL_DS) instead of being written to HBM (411 MB at B = 256) and read back; its own launch shrinks to a
    // Gram launch for its BN statistics (train mode) or disappears (eval mode).  $CVCL_DS_RECOMPUTE=0: the stored form.
    const bool ds_gram = ds_recompute && training && gram_on;
    // BN2 of layers 1-2: formed by the Gram launch (the first reader of relu(bn2(.)); it publishes the affine for the tail pass behind
    // it).  (Layers 3-4 keep partial rows + cvcl_bn_finalize for BN2 / BN3 / the downsample BatchNorm: their consumers are elementwise
    // passes whose workgroups touch every channel.  Re-blocked into channel-sliced 1024-thread workgroups that finalize on load they
    // cost 2-3 us more per launch than the launch they save once two passes overlap: profiles/r06_fol_ab.txt.)
    static const int fol_gram_lab = cvcl_lab_int("CVCL_FOL_GRAM", 1);                // [lab: 0 = BN2 of layers 1-2 keeps its finalize launch]
    const bool fol2 = use_acc && fused_tail && pro && gram_on && fol_gram_lab != 0;
    if (first) {
        if (ds_gram) {
            if ((rc = gram_stats(ld, X, inplanes, nullptr, nullptr, 0))) return rc;
        } else {
            // downsample 1x1 stride s: X -> RD [m_out, outc]
            cvcl_gemm_args a = {};
            a.A = X; a.W = L[ld].w; a.C = ds_recompute ? nullptr : RD;
            a.M = (int)m_out; a.N = outc; a.K = inplanes; a.lda = inplanes; a.ldw = inplanes; a.ldc = outc;
            if (stride > 1) { a.gather_ho = ho; a.gather_wo = wo; a.gather_hi = h; a.gather_wi = wd; a.gather_stride = stride; }
            a.stats = training ? stats : nullptr; a.stats_rows = stats_cap;
            a.centre = centre_of(ld);
            if (a.C || a.stats) { if ((rc = cvcl_gemm(dtype, &a, stream))) return rc; }
            if ((rc = finalize(ld, a.stats ? cvcl_gemm_stats_rows(dtype, &a) : 0, m_out, outc))) return rc;
        }
    }
    // conv1 1x1: X [m_in, inplanes] -> R1 [m_in, width]
    int rows1 = 0;
    static const int fol1_lab = cvcl_lab_int("CVCL_FOL_CONV1", 1);                   // [lab: 0 = BN1 keeps its finalize launch]
    const bool fol1 = use_acc && fol1_lab != 0;
    {
        cvcl_gemm_args a = {};
        a.A = X; a.W = L[l1].w; a.C = R1;
        a.M = (int)m_in; a.N = width; a.K = inplanes; a.lda = inplanes; a.ldw = inplanes; a.ldc = width;
        a.stats = training ? (fol1 ? (float*)acc_of(l1) : stats) : nullptr;
        a.stats_rows = fol1 ? CVCL_STATS_ACCUMULATE : stats_cap;
        a.centre = centre_of(l1);
        if ((rc = cvcl_gemm(dtype, &a, stream))) return rc;
        rows1 = training ? cvcl_gemm_stats_rows(dtype, &a) : 0;
    }
    // BN1's statistics -> affine: inside conv2's prologue (every workgroup of the grouped convolution normalises one fixed slab of
    // 64 channels: 16 accumulator loads per channel)
    if (!fol1 && (rc = finalize(l1, rows1, m_in, width))) return rc;
    // conv2 grouped 3x3 (stride here): R1 -> R2 [m_out, width], BN1+ReLU fused into the load
    {
        const BnSrc src1 = fol1 ? bn_src(l1, m_in, width) : BnSrc{};
        if ((rc = gconv3x3_impl(dtype, R1, fol1 ? nullptr : scale_of(l1), fol1 ? nullptr : shift_of(l1), fol1 ? &src1 : nullptr, L[l2].w, R2,
                                training ? (fol2 ? (float*)acc_of(l2) : stats) : nullptr, fol2 ? CVCL_STATS_ACCUMULATE : stats_cap,
                                centre_of(l2), B, h, wd, width, 32, stride, stream))) return rc;
    }
    if (!fol2 && (rc = finalize(l2, training ? cvcl_gconv3x3_stats_rows(dtype, B, h, wd, width, stride) : 0, m_out, width))) return rc;
    // conv3 1x1: relu(bn2(R2)) -> R3 [m_out, outc].  Layers 1-2 (K = width <= 256, bandwidth-bound): BN2 + ReLU rides conv3's
    // operand load (gemm_pro.hip: applied once per element, W resident in registers) -- no pass of its own over the tensor.
    // Layers 3-4 (MFMA-bound, 4-8 column-tile workgroups per A tile): BN2 + ReLU is applied in place first (one pass over the
    // narrow tensor), which is cheaper than repeating it in every column tile's operand path.
    if (!pro) {
        if ((rc = cvcl_bn_relu_apply(dtype, R2, scale_of(l2), shift_of(l2), R2, m_out, width, stream))) return rc;
    }
    // Layers 1-2 in bf16 ($CVCL_FUSED_TAIL_STAGES leading stages, default 2): conv3 is HBM-bound and cheap there, so it runs
    // twice -- a statistics-only pass (reads only the narrow operand), then a pass whose epilogue applies BN3 + identity /
    // normalised downsample + ReLU and writes the block output -- instead of materialising raw3 and re-reading it in
    // bn_add_relu (saves one write and one read of the wide tensor; results are bit-identical).  Both passes apply BN2 + ReLU
    // on the operand load (gemm_pro.hip).  Measured per step at B = 256 (round 2, with gemm_pro): 1 stage 5.47 ms, 2 stages
    // 5.46 ms and 1.2 GB less HBM traffic; layers 3-4 are MFMA-bound and keep the materialised form.
    // In eval mode there is no statistics pass at all, so the fused tail is used in every stage.
    auto conv3_args = [&]() {
        cvcl_gemm_args a = {};
        a.A = R2; a.W = L[l3].w;
        a.M = (int)m_out; a.N = outc; a.K = width; a.lda = width; a.ldw = width; a.ldc = outc;
        if (pro) { a.a_scale = scale_of(l2); a.a_shift = shift_of(l2); a.a_relu = 1; }
        a.centre = centre_of(l3);
        return a;
    };
    // Train mode with the fused tail: BN3's batch statistics are needed before the product exists.  They come from the Gram matrix
    // of the operand (bn_gram.hip: sum y = w.s, sum y^2 = w^T G w -- one read of the narrow tensor, K <= 256 <= N / 2) instead of
    // a statistics-only run of the whole GEMM.  [lab: CVCL_BN_GRAM=0 the statistics-only pass]
    if (fused_tail && training && pro && gram_on) {
        const BnSrc s2 = fol2 ? bn_src(l2, m_out, width) : BnSrc{};
        if ((rc = gram_stats(l3, R2, width, scale_of(l2), shift_of(l2), 1, fol2 ? &s2 : nullptr))) return rc;
    } else if (!fused_tail || training) {
        cvcl_gemm_args a = conv3_args();
        a.C = fused_tail ? nullptr : R3;
        a.stats = training ? stats : nullptr; a.stats_rows = stats_cap;
        if (a.C || a.stats) { if ((rc = cvcl_gemm(dtype, &a, stream))) return rc; }
        if ((rc = finalize(l3, a.stats ? cvcl_gemm_stats_rows(dtype, &a) : 0, m_out, outc))) return rc;
    } else {
        if ((rc = finalize(l3, 0, m_out, outc))) return rc;              // eval mode: affine from the running stats
    }
    if (fused_tail) {
        cvcl_gemm_args a = conv3_args();
        a.C = dst; a.act = CVCL_ACT_RELU;
        a.c_scale = scale_of(l3); a.c_shift = shift_of(l3);
        if (ds_recompute) {
            a.A2 = X; a.W2 = L[ld].w; a.K2 = inplanes; a.lda2 = inplanes; a.ldw2 = inplanes; a.centre2 = centre_of(ld);
        } else {
            a.R = first ? RD : X; a.ldr = outc;
        }
        if (first) { a.r_scale = scale_of(ld); a.r_shift = shift_of(ld); }
        if ((rc = cvcl_gemm(dtype, &a, stream))) return rc;
    } else if (first) {
        if ((rc = cvcl_bn_add_relu(dtype, R3, scale_of(l3), shift_of(l3), RD, scale_of(ld), shift_of(ld), dst, m_out,
                                   outc, stream))) return rc;
    } else {
        if ((rc = cvcl_bn_add_relu(dtype, R3, scale_of(l3), shift_of(l3), X, nullptr, nullptr, dst, m_out, outc,
                                   stream))) return rc;
    }
    return CVCL_OK;
}

// block-level scratch: R1 / R2 / R3 / RD, statistics rows, 4 affines
inline size_t block_act_bytes(int dtype, int B, int h, int w, int stage) {
    const size_t es = dtype == CVCL_BF16 ? 2 : 4;
    const size_t outc = (size_t)256 << stage;
    return al256((size_t)B * h * w * outc * es);          // >= every intermediate of the block (R1 is [B,h,w,outc/2])
}
}  // namespace

extern "C" size_t cvcl_resnext50_block_workspace_bytes(int dtype, int B, int h, int w, int stage) {
    return 3 * block_act_bytes(dtype, B, h, w, stage) + al256((size_t)kMaxStatsRows * 2 * 2048 * 4) + al256((size_t)4 * 4096 * 4) +
           al256(cvcl_conv1x1_gram_workspace_bytes(256)) + al256(4 * kAccLayer * 8);
}

extern "C" int cvcl_resnext50_block_fwd(int dtype, int B, int h, int w, int stage, int first, int training, const void* x_nhwc,
                                        const cvcl_convbn_params* layers, int n_layers, void* workspace, size_t workspace_bytes,
                                        void* out_nhwc, float momentum, float eps, const float* centres, void* stream) {
    CVCL_CHECK_ARG(x_nhwc && layers && workspace && out_nhwc, "cvcl_resnext50_block_fwd: null pointer");
    CVCL_CHECK_ARG(((uintptr_t)centres & 15) == 0, "cvcl_resnext50_block_fwd: centres must be 16-byte aligned");
    CVCL_CHECK_ARG(stage >= 0 && stage < 4 && n_layers == (first ? 4 : 3), "cvcl_resnext50_block_fwd: stage %d with %d layers", stage, n_layers);
    CVCL_CHECK_ARG(B > 0 && h > 0 && w > 0 && (!(stage > 0 && first) || (h % 2 == 0 && w % 2 == 0)), "cvcl_resnext50_block_fwd: bad shape");
    if (workspace_bytes < cvcl_resnext50_block_workspace_bytes(dtype, B, h, w, stage)) {
        cvcl_set_error("cvcl_resnext50_block_fwd: workspace too small");
        return CVCL_EWORKSPACE;
    }
    char* p = (char*)workspace;
    const size_t ab = block_act_bytes(dtype, B, h, w, stage);
    char* R1 = p; char* R2 = p + ab; char* RD = p + 2 * ab; p += 3 * ab;
    float* stats = (float*)p; p += al256((size_t)kMaxStatsRows * 2 * 2048 * 4);
    float* aff = (float*)p; p += al256((size_t)4 * 4096 * 4);
    void* gram_ws = p; p += al256(cvcl_conv1x1_gram_workspace_bytes(256));
    long long* acc = (long long*)p;
    if (training && dtype == CVCL_BF16 &&
        hipMemsetAsync(acc, 0, (size_t)n_layers * kAccLayer * 8, (hipStream_t)stream) != hipSuccess) {
        cvcl_set_error("cvcl_resnext50_block_fwd: cannot clear the BatchNorm accumulators");
        return CVCL_ELAUNCH;
    }
    if (!training) {                                      // eval mode: affines from the running statistics
        const int planes = 64 << stage;
        const int Cs[4] = {planes * 2, planes * 2, planes * 4, planes * 4};
        for (int l = 0; l < n_layers; ++l) {
            int rc = cvcl_bn_eval_affine(layers[l].gamma, layers[l].beta, layers[l].running_mean, layers[l].running_var, eps,
                                         aff + (size_t)l * 4096, aff + (size_t)l * 4096 + 2048,
                                         centres ? centres + (size_t)l * 2048 : nullptr, Cs[l], stream);
            if (rc) return rc;
        }
    }
    BlockCtx ctx = {dtype, B, training, momentum, eps, stats, gram_ws, stream, (training && dtype == CVCL_BF16) ? acc : nullptr};
    return bottleneck_fwd(ctx, stage, first != 0, h, w, (const char*)x_nhwc, R1, R2, R1, RD, (char*)out_nhwc, layers, aff, nullptr,
                          centres);
}

static int resnext50_fwd_impl(int dtype, int B, int H, int W, int training, const float* x_nchw,
                              const cvcl_convbn_params* layers, int n_layers, void* workspace, size_t workspace_bytes,
                              void* layer4_out_nhwc, float* pooled, float momentum, float eps, float* moments,
                              const float* centres, void* stream) {
    CVCL_CHECK_ARG(x_nchw && layers && workspace && layer4_out_nhwc && pooled, "cvcl_resnext50_fwd: null pointer");
    CVCL_CHECK_ARG(((uintptr_t)centres & 15) == 0, "cvcl_resnext50_fwd: centres must be 16-byte aligned");
    CVCL_CHECK_ARG(n_layers == 53, "cvcl_resnext50_fwd: expected 53 conv+bn layers, got %d", n_layers);
    CVCL_CHECK_ARG(B > 0 && H % 32 == 0 && W % 32 == 0, "cvcl_resnext50_fwd: H, W must be multiples of 32");
    if (workspace_bytes < cvcl_resnext50_workspace_bytes(dtype, B, H, W)) {
        cvcl_set_error("cvcl_resnext50_fwd: workspace too small");
        return CVCL_EWORKSPACE;
    }
    const size_t es = dtype == CVCL_BF16 ? 2 : 4;
    char* w = (char*)workspace;
    char* buf[5];
    for (int i = 0; i < 5; ++i) { buf[i] = w; w += al256(act_elems(B, H, W) * es); }
    float* stats = (float*)w; w += al256((size_t)kMaxStatsRows * 2 * 2048 * 4);
    float* affine = (float*)w; w += al256((size_t)53 * 2 * 2048 * 4);
    float* eval_centres = (float*)w; w += al256((size_t)53 * 2048 * 4);   // eval mode without caller centres: the running means (see below)
    void* gram_ws = w; w += al256(cvcl_conv1x1_gram_workspace_bytes(256));
    // BatchNorm accumulators of the 53 layers (bf16 train mode: finalize-on-load), cleared once per pass ahead of the stem
    long long* acc = (training && dtype == CVCL_BF16) ? (long long*)w : nullptr;
    if (acc && hipMemsetAsync(acc, 0, 53 * kTrunkAccLayer * 8, (hipStream_t)stream) != hipSuccess) {
        cvcl_set_error("cvcl_resnext50_fwd: cannot clear the BatchNorm accumulators");
        return CVCL_ELAUNCH;
    }
    int rc, li = 0;

    // (scale, shift) of layer l live at affine + l * 4096
    auto scale_of = [&](int l) { return affine + (size_t)l * 4096; };
    auto shift_of = [&](int l) { return affine + (size_t)l * 4096 + 2048; };
    auto finalize = [&](int l, int rows, long count, int C) -> int {
        const cvcl_convbn_params& L = layers[l];
        if (training)
            return bn_finalize_launch(stats, rows, count, L.gamma, L.beta, L.running_mean, L.running_var, L.num_batches_tracked,
                                      momentum, eps, scale_of(l), shift_of(l), C, moments ? moments + (size_t)l * 4096 : nullptr, 2048,
                                      centres ? centres + (size_t)l * 2048 : nullptr, stream);
        return CVCL_OK;                                   // eval mode: every layer's affine was produced up front
    };
    if (!training) {
        EvalAffineAll t;
        int l = 0;
        auto put = [&](int C) {
            t.gamma[l] = layers[l].gamma; t.beta[l] = layers[l].beta; t.rm[l] = layers[l].running_mean; t.rv[l] = layers[l].running_var;
            t.C[l] = C; ++l;
        };
        put(64);
        for (int st = 0; st < 4; ++st)
            for (int b = 0; b < kLayers[st]; ++b) {
                const int pl = 64 << st;
                put(pl * 2); put(pl * 2); put(pl * 4);
                if (b == 0) put(pl * 4);
            }
        for (int i = 0; i < 53; ++i)
            CVCL_CHECK_ARG(t.gamma[i] && t.beta[i] && t.rm[i] && t.rv[i], "cvcl_resnext50_fwd: layer %d lacks BatchNorm tensors", i);
        CvclProfScope prof(stream, CVCL_K_BN_FINALIZE);
        // eval mode in bf16 stores every raw conv output as y - running_mean unless the caller brings its own centres
        // ($CVCL_CENTRED_STORAGE=0: plain storage; fp32 keeps plain storage -- nothing to gain there)
        static const bool centred = cvcl_env_on("CVCL_CENTRED_STORAGE");
        float* own = (!centres && centred && dtype == CVCL_BF16) ? eval_centres : nullptr;
        hipLaunchKernelGGL(bn_eval_affine_all_kernel, dim3(53), dim3(256), 0, (hipStream_t)stream, t, eps, affine, centres, own);
        CVCL_LAUNCH_CHECK();
        if (own) centres = own;
    }

    // ---- stem ----
    int h = H / 2, wd = W / 2;
    char* RAW = buf[2];
    const int srows = cvcl_stem_conv_stats_rows(dtype, B, H, W);
    char* X = buf[0];
    char* OUT = buf[1];
    // [lab: CVCL_STEM_POOL=1] the fused stem (cvcl_stem_pool: statistics-only pass, then convolution + bn1 + relu + maxpool in one
    // kernel; bit-identical, -1.0 GB of traffic per step at B = 256) -- measured and NOT the default (profiles/r05_ab_stem.txt, same
    // box: C2 5.13 -> 5.22 ms): the stem convolution itself runs at 116 us for 30 GFLOP (LDS-gather-bound), so recomputing it costs
    // more than the 411 MB it stops writing; the API and its bit-identity test stay for when the convolution gets faster.
    static const bool stem_pool_on = cvcl_lab_int("CVCL_STEM_POOL", 0) != 0;
    if (stem_pool_on && cvcl_stem_pool_supported(dtype, H, W)) {
        if (training && (rc = cvcl_stem_conv7x7(dtype, x_nchw, layers[0].w, nullptr, stats, kMaxStatsRows, centres, B, H, W, stream))) return rc;
        if ((rc = finalize(0, srows, (long)B * h * wd, 64))) return rc;
        if ((rc = cvcl_stem_pool(dtype, x_nchw, layers[0].w, scale_of(0), shift_of(0), centres, X, B, H, W, stream))) return rc;
    } else {
        if ((rc = cvcl_stem_conv7x7(dtype, x_nchw, layers[0].w, RAW, stats, kMaxStatsRows, centres, B, H, W, stream))) return rc;
        if ((rc = finalize(0, srows, (long)B * h * wd, 64))) return rc;
        if ((rc = cvcl_bn_relu_maxpool(dtype, RAW, scale_of(0), shift_of(0), X, B, h, wd, 64, stream))) return rc;
    }
    h /= 2; wd /= 2;
    li = 1;
    BlockCtx ctx = {dtype, B, training, momentum, eps, stats, gram_ws, stream, nullptr};
    for (int stage = 0; stage < 4; ++stage) {
        for (int bi = 0; bi < kLayers[stage]; ++bi) {
            const int stride = (stage > 0 && bi == 0) ? 2 : 1;
            const bool last = (stage == 3 && bi == kLayers[3] - 1);
            char* dst = last ? (char*)layer4_out_nhwc : OUT;
            ctx.acc = acc ? acc + (size_t)li * kTrunkAccLayer : nullptr;
            if ((rc = bottleneck_fwd(ctx, stage, bi == 0, h, wd, X, buf[2], buf[3], buf[2], buf[4], dst, layers + li,
                                     affine + (size_t)li * 4096, moments ? moments + (size_t)li * 4096 : nullptr,
                                     centres ? centres + (size_t)li * 2048 : nullptr))) return rc;
            li += bi == 0 ? 4 : 3;
            char* t = X; X = dst; OUT = (t == (char*)layer4_out_nhwc) ? OUT : t;
            h /= stride; wd /= stride;
        }
    }
    return cvcl_avgpool(dtype, layer4_out_nhwc, pooled, B, h * wd, 2048, stream);
}

extern "C" int cvcl_resnext50_fwd(int dtype, int B, int H, int W, int training, const float* x_nchw,
                                  const cvcl_convbn_params* layers, int n_layers, void* workspace, size_t workspace_bytes,
                                  void* layer4_out_nhwc, float* pooled, float momentum, float eps, const float* centres,
                                  void* stream) {
    return resnext50_fwd_impl(dtype, B, H, W, training, x_nchw, layers, n_layers, workspace, workspace_bytes, layer4_out_nhwc, pooled,
                              momentum, eps, nullptr, centres, stream);
}

// ------------------------------------------------------------------------------------------------
// Train-mode pass with the running-statistics update split off, for passes pipelined on two streams: a frozen trunk's
// consecutive passes are independent except for the BatchNorm running statistics (each pass normalises with its own batch
// statistics), so pass k+1 may run beside pass k -- each fills the other's tail rounds, dependent-launch gaps and
// MFMA-bound phases -- as long as the 53 EMA updates are applied in pass order.  The pass leaves (mean, unbiased variance) of
// every layer in `moments` ([53][2][2048] floats) and touches no BatchNorm buffer; cvcl_resnext50_apply_moments, enqueued by
// the caller behind the previous pass's apply (one event), performs the updates of cvcl_resnext50_fwd bit for bit.
extern "C" size_t cvcl_resnext50_moments_floats(void) { return (size_t)53 * 2 * 2048; }

extern "C" int cvcl_resnext50_fwd_deferred_stats(int dtype, int B, int H, int W, const float* x_nchw,
                                                 const cvcl_convbn_params* layers, int n_layers, void* workspace,
                                                 size_t workspace_bytes, void* layer4_out_nhwc, float* pooled, float eps,
                                                 float* moments, const float* centres, void* stream) {
    CVCL_CHECK_ARG(moments, "cvcl_resnext50_fwd_deferred_stats: moments is NULL");
    return resnext50_fwd_impl(dtype, B, H, W, 1, x_nchw, layers, n_layers, workspace, workspace_bytes, layer4_out_nhwc, pooled,
                              0.f, eps, moments, centres, stream);
}

extern "C" int cvcl_resnext50_apply_moments(const cvcl_convbn_params* layers, int n_layers, const float* moments, float momentum,
                                            void* stream) {
    CVCL_CHECK_ARG(layers && moments && n_layers == 53, "cvcl_resnext50_apply_moments: expected 53 conv+bn layers and their moments");
    ApplyMomentsAll t;
    int l = 0;
    auto put = [&](int C) {
        t.rm[l] = layers[l].running_mean; t.rv[l] = layers[l].running_var; t.nbt[l] = layers[l].num_batches_tracked;
        t.C[l] = C; ++l;
    };
    put(64);
    for (int st = 0; st < 4; ++st)
        for (int b = 0; b < kLayers[st]; ++b) {
            const int pl = 64 << st;
            put(pl * 2); put(pl * 2); put(pl * 4);
            if (b == 0) put(pl * 4);
        }
    for (int i = 0; i < 53; ++i)
        CVCL_CHECK_ARG(t.rm[i] && t.rv[i], "cvcl_resnext50_apply_moments: layer %d lacks running statistics", i);
    CvclProfScope prof(stream, CVCL_K_BN_FINALIZE);
    hipLaunchKernelGGL(bn_apply_moments_kernel, dim3(53), dim3(256), 0, (hipStream_t)stream, t, moments, momentum);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}
