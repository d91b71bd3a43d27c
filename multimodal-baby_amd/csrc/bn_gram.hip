// BatchNorm statistics of a 1x1 convolution's output WITHOUT computing the output (round 4): for y = a' W^T with a' = the
// (BatchNorm + ReLU'd, bf16-rounded) operand rows the convolution multiplies,
//
//      sum_m y[m][n]   = w_n . s                     s = sum_m a'[m]            (K values)
//      sum_m y[m][n]^2 = w_n^T G w_n                 G = sum_m a'[m] a'[m]^T    (K x K, symmetric)
//
// torchvision Bottleneck.forward, reached from multimodal/multimodal.py:101: bn3(conv3(relu(bn2(.)))) in layers 1-2 and the
// downsample branch bn_d(conv_d(x)) of layer1.0, whose convolutions run FUSED with their consumer (gemm_pro.hip: the Bottleneck
// tail applies BN3 in its epilogue), so BN3's batch statistics are needed before the product is ever formed.  Rounds 2-3 ran the
// whole GEMM a first time for them ("statistics-only pass": 52.6 GFLOP at 720-800 TFLOP/s per launch, 7 launches = 0.51 ms of a
// 6 ms trunk pass, at 1.4-2.8 TB/s of HBM).  The Gram matrix needs M K^2 MACs instead of M K N (K <= 256 <= N / 2) and no
// per-element epilogue, so this pass is bound by reading the operand once.
//
//   gram_pro_kernel<NT>   K = 32 NT (64 | 128 | 256).  One workgroup per CU, persistent over 64-row tiles, waves specialised:
//                         waves 4 .. (8 of them at K <= 128, 4 at K = 256) PRODUCE -- rows arrive as 16-byte chunks through registers (64 KiB per CU in flight), are
//                         transformed exactly as gemm_pro.hip transforms them (scale, shift, ReLU, round to bf16; rows past M
//                         contribute zeros), summed per column, and written row-major into a double-buffered LDS tile (pitch
//                         K * 2 + 64 bytes: wgrad.hip's bank layout); waves 0-3 CONSUME -- the MFMA fragments (8 consecutive rows of
//                         one column, the transpose of the image) come from ds_read_b64_tr_b16, only the upper-triangle 32 x 32 tiles
//                         of G are computed (wave c owns tile rows I0 = c % (NT / 2) and I1 = NT - 1 - I0: NT + 1 tiles, balanced),
//                         accumulators live in registers for the workgroup's whole life.  One barrier per tile: tile t + 1 is staged
//                         while tile t is multiplied (every SIMD holds one wave of each kind).  A first version whose eight waves
//                         all staged, met, and multiplied ran its phases back to back: 57.8 us for layer 1's conv3 operand against
//                         33 us with the math ablated.  Output: one fp32 partial per workgroup (tiles + column sums), fixed order
//                         everywhere (deterministic).
//   gram_reduce_kernel    partials -> G, s in fp64.
//   bn_from_gram_kernel   per output channel: mean = w.s / M, var = w^T G w / M - mean^2 in fp64, then exactly what bn_finalize does
//                         with them (scale / shift of the STORED tensor y - centre, running statistics or deferred moments).
// The statistics are those of the fp32-accumulated product; the tail stores round(y - centre): the difference is the mean of the
// bf16 rounding errors (~2^-9 |y| / sqrt(M) on the mean), far below the tolerance of any consumer (tests/test_bn_gram_gpu.py).
#include <cstdlib>

#include "cvcl_common.h"

namespace {

// phase ablation for timing studies (build option, as in resnext.hip's grouped convolution): 1 no BN math, 2 no MFMA loop, 4 no column
// sums, 8 no LDS staging writes
#ifndef CVCL_GRAM_ABLATE
#define CVCL_GRAM_ABLATE 0
#endif
constexpr int GP_ABL = CVCL_GRAM_ABLATE;
constexpr int GP_PM = 64;                  // rows per tile
constexpr int GP_MAXG = 256;               // workgroups (= partials)

struct GramDev {
    const bf16_t* A; const float* a_scale; const float* a_shift;
    BnSrc src;                             // src.acc != NULL: the operand's BatchNorm affine is formed here (finalize-on-load), not read
    float* P; float* CS;                   // partials: P[grid][T][32][32], CS[grid][K]
    long M; int lda, relu, tiles;
};

typedef __bf16 gp_tr4 __attribute__((__vector_size__(4 * sizeof(__bf16))));
__device__ inline bf16x4 gp_tr_read(const char* p) {
    auto lp = reinterpret_cast<__attribute__((address_space(3))) gp_tr4*>((__attribute__((address_space(3))) char*)(p));
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16(lp));
}

__host__ __device__ constexpr int gp_tile_index(int NT, int I, int J) { return I * NT - I * (I - 1) / 2 + (J - I); }   // I <= J
template <int NT> constexpr int gp_pitch() { return NT * 64 + 64; }
template <int NT> constexpr int gp_lds_bytes() { return 2 * GP_PM * gp_pitch<NT>(); }      // (>= 16 KiB: reused for the final exchanges)

// producer waves per workgroup: the staging arithmetic of ONE wave per SIMD is a dependent chain that issues at well under one
// instruction per four cycles (PMC, K = 128: 18 us of VALU issue in a 44.7 us kernel whose matrix pipe is busy for 6.5) -- two
// producer waves per SIMD interleave.  K = 256 keeps one (its nine accumulator tiles leave no registers for twelve waves)
template <int NT> constexpr int gp_producer_threads() { return NT == 8 ? 256 : 512; }

// The workgroup-wide barrier protocol of gram_pro_kernel.  The producer and the consumer waves run different code and must meet at
// exactly the same barriers, in this order: rounds * NPF step barriers (one per staged tile), gp_exchange_barriers<NT>() for the
// consumers' row-step exchange (two per accumulator tile when a tile's four row steps are shared by several waves), then two for the
// producers' column sums.  Both roles take the middle count from here; the consumer's exchange loop is checked against it at compile time.
template <int NT> __host__ __device__ constexpr int gp_exchange_barriers() {
    constexpr int NPAIR = NT / 2, KG = 4 / NPAIR;
    return KG > 1 ? 2 * (NT + 1) : 0;
}

// ---- producer waves (waves 4 ..): chunk s_c of rows s_r + RPP * i of every tile; a thread's 8 columns never change
template <int NT>
__device__ __forceinline__ void gram_producer(const GramDev& p, char* smem, const int tid, const int rounds, const float* a_scale,
                                              const float* a_shift) {
    constexpr int PT = gp_producer_threads<NT>();
    constexpr int K = NT * 32;
    constexpr int PITCH = gp_pitch<NT>();
    constexpr int CPR = K / 8;                          // 16-byte chunks per row (8 | 16 | 32)
    constexpr int RPP = PT / CPR;                       // rows per staging pass (64 | 32 | 8)
    constexpr int NCH = GP_PM / RPP;                    // chunks per thread per tile (1 | 2 | 8)
    constexpr int ABUF = GP_PM * PITCH;
    constexpr int NPF = 16 / NT;                        // tiles in flight (8 | 4 | 2): 64 KiB per CU
    const int s_c = tid % CPR, s_r = tid / CPR;
    const bool plain = a_scale == nullptr;              // (downsample branch: the operand is the block input as stored)
    f32x2 sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sc[e] = plain ? f32x2{1.f, 1.f} : f32x2{a_scale[s_c * 8 + 2 * e], a_scale[s_c * 8 + 2 * e + 1]};
        sh[e] = plain ? f32x2{0.f, 0.f} : f32x2{a_shift[s_c * 8 + 2 * e], a_shift[s_c * 8 + 2 * e + 1]};
    }
    const short fl = (!plain && p.relu) ? (short)0 : (short)-32768;           // ReLU on rounded bf16 pairs = packed int16 max (relu2)
    const s16x2 floor2 = s16x2{fl, fl};
    u32x4 araw[NPF][NCH];
    // every slot always loads (rows past M clamp to the last row) and there is no branch between a load and its use (the plain
    // operand runs through scale 1, shift 0, floor -32768): behind a condition the compiler cannot count the loads in flight and
    // waits for ALL of them before each staging -- one memory round trip per tile
    auto load_a = [&](int tile, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            long m = (long)tile * GP_PM + s_r + RPP * i;
            if (m >= p.M) m = p.M - 1;
            araw[slot][i] = *reinterpret_cast<const u32x4*>(p.A + m * p.lda + s_c * 8);
        }
    };
    f32x2 csum[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) csum[e] = f32x2{0.f, 0.f};
    auto stage_a = [&](int tile, int buf, int slot) __attribute__((always_inline)) {
        char* dst = smem + buf * ABUF;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = s_r + RPP * i;
            const unsigned keep = (long)tile * GP_PM + r < p.M ? 0xffffffffu : 0u;         // (a tile past the end stages zeros)
            u32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned y = araw[slot][i][e];
                if constexpr (!(GP_ABL & 1)) {
                    y = round2(__builtin_elementwise_fma(widen2(y), sc[e], sh[e]));
                    y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, y), floor2));
                }
                y &= keep;
                v[e] = y;
                if constexpr (!(GP_ABL & 4)) csum[e] += widen2(y);
            }
            if constexpr (!(GP_ABL & 8)) *reinterpret_cast<u32x4*>(dst + r * PITCH + s_c * 16) = v;
            else if (v[0] == 0x12345678u && v[1] == 0x9abcdef0u) *reinterpret_cast<u32x4*>(dst + r * PITCH + s_c * 16) = v;
        }
    };
    const int first = blockIdx.x, G = gridDim.x;
#pragma unroll
    for (int s = 0; s < NPF; ++s) load_a(first + s * G, s);
    stage_a(first, 0, 0);
    load_a(first + NPF * G, 0);
    // step t (whole rounds of NPF, no exits in between): barrier (tile t is staged, the consumers are done with tile t - 1), then
    // tile t + 1 goes into the buffer tile t - 1 occupied while the consumers multiply tile t
    int tile = first + G;                               // the tile staged in this step
    for (int rd = 0; rd < rounds; ++rd) {
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int slot = (j + 1) % NPF;
            __syncthreads();
            stage_a(tile, (j + 1) & 1, slot);           // (NPF is even: the buffer of step rd * NPF + j + 1)
            load_a(tile + NPF * G, slot);
            tile += G;
        }
    }
    // ---- epilogue: the consumers' exchange barriers (gp_exchange_barriers: the one definition both roles count from), then the
    // column sums: the RPP staging rows of a column in a fixed order
#pragma unroll
    for (int t = 0; t < gp_exchange_barriers<NT>(); ++t) __syncthreads();
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);        // [RPP][K] floats (<= 16 KiB)
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[s_r * K + s_c * 8 + 2 * e] = csum[e][0]; red[s_r * K + s_c * 8 + 2 * e + 1] = csum[e][1]; }
    __syncthreads();                                    // (all eight waves meet it)
    if (tid < K) {
        float a = 0.f;
        for (int l = 0; l < RPP; ++l) a += red[l * K + tid];
        p.CS[(long)blockIdx.x * K + tid] = a;
    }
}

// ---- consumer wave C (waves 0-3): tile-row pair I0 = C % (NT / 2), I1 = NT - 1 - I0 (NT + 1 tiles of the upper triangle, balanced)
// and, where there are fewer pairs than waves, a share of the tile's four 16-row steps (NT = 4: two steps, NT = 2: one)
template <int NT, int C>
__device__ __forceinline__ void gram_consumer(const GramDev& p, char* smem, const int lane, const int rounds, const int my_tiles) {
    constexpr int PITCH = gp_pitch<NT>();
    constexpr int ABUF = GP_PM * PITCH;
    constexpr int T = NT * (NT + 1) / 2;
    constexpr int NPAIR = NT / 2, KG = 4 / NPAIR, KS = 4 / KG;          // row steps per wave (4 | 2 | 1)
    constexpr int I0 = C % NPAIR, I1 = NT - 1 - I0, kg = C / NPAIR;
    constexpr int NACC = NT + 1;
    constexpr int NPF = 16 / NT;
    static_assert(NACC * (KG > 1 ? 2 : 0) == gp_exchange_barriers<NT>(), "the exchange loop below and the producers' barrier count");
    // fragment addressing (wgrad.hip): 16-lane group g4 reads a [4 m][16 col] block; lane q supplies row q / 4, cols 4 (q % 4) .. + 3
    const int g4 = lane >> 4, q = lane & 15;
    const int frag_off = ((g4 >> 1) * 8 + (q >> 2)) * PITCH + ((g4 & 1) * 16 + (q & 3) * 4) * 2;
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const int n = rounds * NPF;
    for (int t = 0; t < n; ++t) {
        __syncthreads();
        if (t < my_tiles && !(GP_ABL & 2)) {
            const char* tb = smem + (t & 1) * ABUF + frag_off;
#pragma unroll
            for (int k2 = 0; k2 < KS; ++k2) {
                const char* rb = tb + (kg * KS + k2) * 16 * PITCH;            // 16 rows of the tile
                bf16x8 fr[NT];                               // the column-block fragments this wave needs: J = I0 .. NT - 1
#pragma unroll
                for (int J = I0; J < NT; ++J) {
                    const bf16x4 v0 = gp_tr_read(rb + J * 64), v1 = gp_tr_read(rb + 4 * PITCH + J * 64);
                    fr[J] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                // tiles (I0, J), J = I0 .. NT - 1 -> acc[J - I0];  tiles (I1, J), J = I1 .. NT - 1 -> acc[NT - I0 + J - I1]
#pragma unroll
                for (int J = I0; J < NT; ++J)
                    acc[J - I0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[I0], fr[J], acc[J - I0], 0, 0, 0);
#pragma unroll
                for (int J = I1; J < NT; ++J)
                    acc[NT - I0 + J - I1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[I1], fr[J], acc[NT - I0 + J - I1], 0, 0, 0);
            }
        }
    }
    // ---- this workgroup's partial: the row-step shares of a tile (waves kg = 1 .. KG - 1) are added through LDS in a fixed order, one
    // tile slot at a time
    float* xch = reinterpret_cast<float*>(smem);             // [(KG - 1) NPAIR waves][1024] floats (<= 12 KiB)
    float* outp = p.P + (long)blockIdx.x * T * 1024;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        if constexpr (KG > 1) {
            __syncthreads();
            if constexpr (kg > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[((kg - 1) * NPAIR + I0) * 1024 + r * 64 + lane] = acc[t][r];
            }
            __syncthreads();
        }
        if constexpr (kg == 0) {
            const int I = t < NT - I0 ? I0 : I1, J = t < NT - I0 ? I0 + t : I1 + (t - (NT - I0));
            float* o = outp + gp_tile_index(NT, I, J) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3), col = lane & 31;          // i (block I), j (block J)
                float v = acc[t][r];
#pragma unroll
                for (int g = 1; g < KG; ++g) v += xch[((g - 1) * NPAIR + I0) * 1024 + r * 64 + lane];
                o[row * 32 + col] = v;
            }
        }
    }
    __syncthreads();                                         // (the producers' column-sum exchange: LDS is theirs after this one ...
    __syncthreads();                                         //  ... and summed after this one)
}

template <int NT>
__global__ __launch_bounds__(256 + gp_producer_threads<NT>()) void gram_pro_kernel(GramDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NPF = 16 / NT;                             // the producers' tiles in flight: steps come in whole rounds of NPF
    const int first = blockIdx.x, G = gridDim.x;
    const int my_tiles = first < p.tiles ? (p.tiles - first + G - 1) / G : 0;
    const int rounds = (my_tiles + NPF - 1) / NPF;
    // finalize-on-load (cvcl_common.h): the operand's BatchNorm affine (BN2 of a layer-1/2 Bottleneck) from the grouped convolution's
    // accumulators, by every workgroup for all K channels (16 loads per channel); workgroup 0 publishes it -- the tail pass that
    // follows reads (scale, shift) from there -- with the batch moments / running statistics
    __shared__ float gp_aff[2][NT * 32];
    const float* a_scale = p.a_scale;
    const float* a_shift = p.a_shift;
    if (p.src.acc) {
        bn_slice_affine<NT * 32>(p.src, 0, blockIdx.x == 0, gp_aff[0], gp_aff[1]);
        a_scale = gp_aff[0];
        a_shift = gp_aff[1];
    }
    switch (wave) {                                          // (every wave meets the same barriers in its own specialisation)
        case 0: gram_consumer<NT, 0>(p, smem, tid & 63, rounds, my_tiles); break;
        case 1: gram_consumer<NT, 1>(p, smem, tid & 63, rounds, my_tiles); break;
        case 2: gram_consumer<NT, 2>(p, smem, tid & 63, rounds, my_tiles); break;
        case 3: gram_consumer<NT, 3>(p, smem, tid & 63, rounds, my_tiles); break;
        default: gram_producer<NT>(p, smem, tid - 256, rounds, a_scale, a_shift); break;
    }
}

// partials -> fp64: the FULL symmetric G [K][K] (row-major: both triangles, so that a consumer streams its row) and s [K] behind it.
// 64 elements x 4 partial groups per block: every thread adds parts / 4 partials with 16 loads in flight (the plain one-thread-per-
// element loop was a chain of `parts` dependent-latency loads), the groups are combined in a fixed order.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ P, const float* __restrict__ CS, int parts, int NT,
                                                          double* __restrict__ out) {
    const int K = NT * 32, n_tile_elems = NT * (NT + 1) / 2 * 1024;
    __shared__ double sh[4][64];
    const int e = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;
    const bool is_tile = i < n_tile_elems, is_cs = !is_tile && i < n_tile_elems + K;
    const float* src = is_tile ? P + i : CS + (is_cs ? i - n_tile_elems : 0);
    const long stride = is_tile ? n_tile_elems : K;
    double a = 0.0;
    if (is_tile || is_cs) {
        int g = pg;
        for (; g + 60 < parts; g += 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[(long)(g + 4 * u) * stride];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += (double)v[u];
        }
        for (; g < parts; g += 4) a += (double)src[(long)g * stride];
    }
    sh[pg][e] = a;
    __syncthreads();
    if (pg == 0 && (is_tile || is_cs)) {
        const double v = (sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e]);
        if (is_cs) {
            out[(long)K * K + (i - n_tile_elems)] = v;
        } else {                                             // tile t = (I, J), element (r, c) -> G[I 32 + r][J 32 + c] and its mirror
            const int t = i >> 10, r = (i >> 5) & 31, c = i & 31;
            int I = 0, base = 0;
            while (t >= base + (NT - I)) { base += NT - I; ++I; }
            const int J = I + (t - base);
            out[(long)(I * 32 + r) * K + J * 32 + c] = v;
            if (I != J) out[(long)(J * 32 + c) * K + I * 32 + r] = v;
        }
    }
}

__device__ inline float gram_ema(float r, float x, float m) { return fmaf(m, x, (1.f - m) * r); }

// 4 output channels per workgroup, thread k = row k of G: r_k = sum_l G[k][l] w[l] (G is symmetric: thread k walks COLUMN k, so that a
// wave's load is 64 consecutive doubles; 16 loads in flight), then sum y^2 = sum_k w[k] r_k, sum y = sum_k w[k] s[k]; everything in
// fp64, where var = E[y^2] - mean^2 loses nothing that matters (2^-53 mean^2 / var).  The sums over k: butterflies inside a wave, the
// sixteen waves in a fixed order.
constexpr int GF_CH = 4;
__device__ inline double gf_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int K>
__global__ __launch_bounds__(1024) void bn_from_gram_kernel(const double* __restrict__ Gd, double inv_count, double bessel,
                                                            const bf16_t* __restrict__ W,
                                                            int ldw, int N, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var,
                                                            int64_t* __restrict__ nbt, float momentum, float eps, float* __restrict__ scale,
                                                            float* __restrict__ shift, float* __restrict__ moments, int moments_ld,
                                                            const float* __restrict__ centre) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double* s = Gd + (long)K * K;
    __shared__ __attribute__((aligned(16))) double wsh[256][GF_CH];
    __shared__ double red[16][2 * GF_CH];
    // thread (kk, lq): rows kk + 64 i (i < K / 64) of G against sixteenth lq of the columns.  Several rows per thread because the
    // weights are wave-uniform LDS reads (8 clocks per 16-byte broadcast): one row per thread was bound by them (15.6 us at K = 256)
    const int kk = threadIdx.x & 63, lq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n0 = blockIdx.x * GF_CH;
    const int k = threadIdx.x & 255;
    constexpr int R = K / 64, NL = K / 16;                               // rows per thread (1 | 2 | 4), columns per thread (4 | 8 | 16)
    constexpr int LB = NL < 32 / R ? NL : 32 / R;                        // columns per batch of loads (<= 32 in flight)
    // (measured and dropped: requesting the first batch of G, the column sums and the channel's BatchNorm tensors before the weights
    // are in LDS -- no change at K <= 128, spills at K = 256.  What is left, 9 us at K <= 128 and 15 us at K = 256 / N = 512, is
    // mostly the price of any small kernel with a few dependent memory round trips on this part: bn_finalize takes 5 us)
    if (threadIdx.x < 256) {
#pragma unroll
        for (int c = 0; c < GF_CH; ++c) wsh[k][c] = (k < K && n0 + c < N) ? (double)(float)W[(long)(n0 + c) * ldw + k] : 0.0;
    }
    __syncthreads();
    double r[R][GF_CH];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int c = 0; c < GF_CH; ++c) r[i][c] = 0.0;
    const int lb = lq * NL;
    const double* col = Gd + (long)lb * K + kk;                           // (G is symmetric: row k = column k, coalesced)
#pragma unroll 1
    for (int b = 0; b < NL / LB; ++b) {
        const int l0 = lb + b * LB;
        double g[LB][R];
#pragma unroll
        for (int u = 0; u < LB; ++u)
#pragma unroll
            for (int i = 0; i < R; ++i) g[u][i] = col[u * K + 64 * i];
        col += LB * K;
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const d2 w01 = *reinterpret_cast<const d2*>(&wsh[l0 + u][0]), w23 = *reinterpret_cast<const d2*>(&wsh[l0 + u][2]);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                r[i][0] += g[u][i] * w01[0]; r[i][1] += g[u][i] * w01[1]; r[i][2] += g[u][i] * w23[0]; r[i][3] += g[u][i] * w23[1];
            }
        }
    }
    double sk[R];
#pragma unroll
    for (int i = 0; i < R; ++i) sk[i] = lq == 0 ? s[kk + 64 * i] : 0.0;
    const bool fin = threadIdx.x < GF_CH && n0 + threadIdx.x < N;
    const int ch = fin ? n0 + threadIdx.x : 0;
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < GF_CH; ++c) {
        double q = 0.0, m = 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            q += wsh[kk + 64 * i][c] * r[i][c];
            m += wsh[kk + 64 * i][c] * sk[i];
        }
        q = gf_wave_sum(q); m = gf_wave_sum(m);
        if (kk == 0) { red[wv][c] = q; red[wv][GF_CH + c] = m; }
    }
    __syncthreads();
    if (fin) {
        double sum_sq = 0.0, sum = 0.0;
        for (int w = 0; w < 16; ++w) { sum_sq += red[w][threadIdx.x]; sum += red[w][GF_CH + threadIdx.x]; }
        const double mean_y = sum * inv_count;
        double var = sum_sq * inv_count - mean_y * mean_y;
        if (var < 0.0) var = 0.0;
        // the consumers normalise the STORED tensor y - centre (bn_finalize's convention): its mean is mean_y - centre
        const double mean = mean_y - (centre ? (double)centre[ch] : 0.0);
        const float sc = gamma[ch] / sqrtf((float)var + eps);
        scale[ch] = sc;
        shift[ch] = beta[ch] - (float)mean * sc;
        const double unbiased = var * bessel;
        if (moments) {
            moments[ch] = (float)mean_y;
            moments[moments_ld + ch] = (float)unbiased;
        } else if (running_mean) {
            running_mean[ch] = gram_ema(running_mean[ch], (float)mean_y, momentum);
            running_var[ch] = gram_ema(running_var[ch], (float)unbiased, momentum);
        }
    }
    if (!moments && nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

// (one workgroup per CU also at K = 256, where a partial is 147 KiB: 128 workgroups measured 48.9 us against 37.7 for gram + reduce)
int gram_grid(long tiles) {
    int g = GP_MAXG;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 && n < g) g = n;
    return tiles < g ? (int)tiles : g;
}

template <int NT>
int gram_launch(const GramDev& d, int grid, hipStream_t st) {
    static CvclLdsAttr attr;
    if (!attr.ready()) {
        if (hipFuncSetAttribute((const void*)gram_pro_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes<NT>()) != hipSuccess) {
            cvcl_set_error("cvcl_conv1x1_gram: cannot raise the dynamic LDS limit");
            return CVCL_ELAUNCH;
        }
        attr.mark();
    }
    hipLaunchKernelGGL(gram_pro_kernel<NT>, dim3(grid), dim3(256 + gp_producer_threads<NT>()), gp_lds_bytes<NT>(), st, d);
    return CVCL_OK;
}

}  // namespace

// workspace: [GP_MAXG partials of tiles + column sums (fp32)] [G + s in fp64]
extern "C" size_t cvcl_conv1x1_gram_workspace_bytes(int K) {
    if (K != 64 && K != 128 && K != 256) return 0;
    const size_t NT = K / 32, T = NT * (NT + 1) / 2;
    return (size_t)GP_MAXG * (T * 1024 + K) * 4 + ((size_t)K * K + K) * 8 + 256;
}

// G (fp64 [K][K], symmetric, both triangles) and s (fp64 [K], behind it) of a' = relu?(A * a_scale + a_shift) rounded
// to bf16 (a_scale NULL: A as stored); -> gram_out inside the workspace (returned through *gram_out)
extern "C" int cvcl_conv1x1_gram(const void* A, int lda, long M, int K, const float* a_scale, const float* a_shift, int a_relu, void* workspace,
                                 size_t workspace_bytes, const double** gram_out, void* stream) {
    return cvcl_conv1x1_gram_src(A, lda, M, K, a_scale, a_shift, nullptr, a_relu, workspace, workspace_bytes, gram_out, stream);
}

// src != NULL (internal: the trunk's launch sequence): the operand's affine comes from its producer's accumulators inside the kernel
int cvcl_conv1x1_gram_src(const void* A, int lda, long M, int K, const float* a_scale, const float* a_shift, const BnSrc* src, int a_relu,
                          void* workspace, size_t workspace_bytes, const double** gram_out, void* stream) {
    CVCL_CHECK_ARG(A && workspace && M > 0 && (K == 64 || K == 128 || K == 256) && lda % 8 == 0 && lda >= K && ((uintptr_t)A & 15) == 0 &&
                       (a_scale == nullptr) == (a_shift == nullptr) && ((uintptr_t)workspace & 255) == 0,
                   "cvcl_conv1x1_gram: bf16 rows with K = 64 | 128 | 256, 16-byte aligned (K %d)", K);
    CVCL_CHECK_ARG(!src || (src->acc && src->C == K && src->gamma && src->beta && src->count > 0 && !a_scale), "cvcl_conv1x1_gram: finalize-on-load source");
    if (workspace_bytes < cvcl_conv1x1_gram_workspace_bytes(K)) {
        cvcl_set_error("cvcl_conv1x1_gram: workspace too small");
        return CVCL_EWORKSPACE;
    }
    const int NT = K / 32, T = NT * (NT + 1) / 2;
    GramDev d;
    d.A = (const bf16_t*)A; d.a_scale = a_scale; d.a_shift = a_shift; d.relu = a_relu; d.M = M; d.lda = lda;
    d.src = src ? *src : BnSrc{};
    d.tiles = cvcl_div_up(M, GP_PM);
    const int grid = gram_grid(d.tiles);
    d.P = (float*)workspace;
    d.CS = d.P + (size_t)GP_MAXG * T * 1024;
    double* out = (double*)(((uintptr_t)(d.CS + (size_t)GP_MAXG * K) + 15) & ~(uintptr_t)15);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // [lab: upper bounds -- after n calls the Gram launch / the reduce launch is skipped and the consumers read an earlier pass's G;
    //  only meaningful on a repeated batch]
    static const int skip_pro = cvcl_lab_int("CVCL_SKIP_GRAM_PRO_AFTER", 0), skip_red = cvcl_lab_int("CVCL_SKIP_GRAM_REDUCE_AFTER", 0);
    static long calls = 0;
    ++calls;
    if (!(skip_pro > 0 && calls > skip_pro)) {
        CvclProfScope prof(stream, CVCL_K_GEMM_PRO);
        rc = NT == 8 ? gram_launch<8>(d, grid, st) : NT == 4 ? gram_launch<4>(d, grid, st) : gram_launch<2>(d, grid, st);
        if (rc) return rc;
        CVCL_LAUNCH_CHECK();
    }
    if (!(skip_red > 0 && calls > skip_red)) {
        CvclProfScope prof(stream, CVCL_K_BN_FINALIZE);
        const int n = T * 1024 + K;
        hipLaunchKernelGGL(gram_reduce_kernel, dim3(cvcl_div_up(n, 64)), dim3(256), 0, st, d.P, d.CS, grid, NT, out);
        CVCL_LAUNCH_CHECK();
    }
    if (gram_out) *gram_out = out;
    return CVCL_OK;
}

// BatchNorm (train mode) of y = a' W^T from the Gram data of a': (scale, shift) of the stored tensor y - centre, running statistics
// (or, with moments != NULL, the batch mean / unbiased variance at moments[ch], moments[moments_ld + ch]) -- cvcl_bn_finalize's outputs
extern "C" int cvcl_bn_from_gram(const double* gram, int K, long count, const void* W, int ldw, int N, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                 float* scale, float* shift, float* moments, int moments_ld, const float* centre, void* stream) {
    CVCL_CHECK_ARG(gram && W && gamma && beta && scale && shift && count > 0 && N > 0 && (K == 64 || K == 128 || K == 256) && ldw >= K,
                   "cvcl_bn_from_gram: bad args");
    static const int skip_fg = cvcl_lab_int("CVCL_SKIP_FROM_GRAM_AFTER", 0);       // [lab: as above, the bn_from_gram launch]
    static long fg_calls = 0;
    if (skip_fg > 0 && ++fg_calls > skip_fg) return CVCL_OK;
    CvclProfScope prof(stream, CVCL_K_BN_FINALIZE);
#define CVCL_FROM_GRAM(KK)                                                                                                                  \
    hipLaunchKernelGGL(bn_from_gram_kernel<KK>, dim3(cvcl_div_up(N, GF_CH)), dim3(1024), 0, (hipStream_t)stream, gram, 1.0 / (double)count, \
                       count > 1 ? (double)count / (double)(count - 1) : 1.0,                                                              \
                       (const bf16_t*)W, ldw, N, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift, \
                       moments, moments_ld, centre)
    if (K == 256) CVCL_FROM_GRAM(256); else if (K == 128) CVCL_FROM_GRAM(128); else CVCL_FROM_GRAM(64);
#undef CVCL_FROM_GRAM
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}
