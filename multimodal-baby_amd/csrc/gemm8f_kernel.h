// 256 (192) x 256 e4m3 GEMM tile for the ViT linears of BASELINE configs[4]: the 8-wave pipeline of gemm8w_kernel.h on
// v_mfma_scale_f32_32x32x64_f8f6f4 (round 4; replaces the 128 x 128 two-buffer kernel of gemm_fp8.hip on the large shapes).
//
//      C[M,N] = act( (A8[M,K] . W8[N,K]^T) * sa[m] * sw[n] + bias[n] ) (+ R)      (A8, W8 e4m3 bytes, K contiguous; N % 256 == 0,
//                                                                                  K % 128 == 0)
// Reference call sites: the four nn.Linear of a ViT block, multimodal/vision_transformer_dino_mugs.py:92-94 (Mlp.fc1 / fc2),
// :113-115 (Attention.qkv / proj); the fp8 storage points are this repo's (configs[4]), emulated by tests/test_encoders_gpu.py.
//
// Why: the 128 x 128 kernel waits vmcnt(0) + barrier per 128-byte K tile with one tile of prefetch and runs its epilogue with the
// matrix pipe idle -- MFMA pipe 20-28 % busy (profiles/r04_pmc_c5_summary.txt).  Here, as in the bf16 kernel: one 512-thread
// workgroup per CU, a wave owns (32 MT) x 64 of the tile (MT x 2 accumulator tiles of 32 x 32 = 128 registers at MT = 4), a
// 4-stage LDS ring of 32 KiB stages (64 BYTES = 64 k per operand row) filled by global_load_lds_dwordx4, fragments
// double-buffered in registers, ONE barrier and ONE counted vmcnt wait per stage, the fragment reads of stage g+1 and the loads
// of stage g+4 interleaved with the second half of stage g's MFMAs.  Per stage and wave: 2 MT MFMAs of 64 cycles against
// 2 (2 + MT) ds_read_b128.
//
// LDS image of a stage: row r of an operand at r * 64, its 16-byte chunk c at chunk position c ^ ((r >> 2) & 3) (swizzle on the
// SOURCE address of the DMA).  A lane's fragment for one MFMA = chunks h and h + 2 of row lane & 31 (h = lane >> 5: the k
// order the scaled MFMA uses, tools/probes/mx_scale_probe_b.hip); the 16-lane groups of a ds_read_b128 then cover the 16 slots
// of a 256-byte bank row exactly once.
//
// Per-tile epilogue operands (weight-row scales, bias, activation-row scales: 3 x 1 KiB) arrive by DMA with the tile's first
// stage into a parity slot; MX input: the e8m0 block scales of a 128-k tile (256 rows x 4 bytes) arrive with its first stage
// into a 4-slot ring.  These extra loads sit in single waves' queues and only make a few counted waits conservative.
#pragma once
#include <type_traits>

#include "cvcl_common.h"

namespace g8f {

typedef int v8i __attribute__((ext_vector_type(8)));

constexpr int BN = 256;
constexpr int BKB = 64;                            // bytes (= k) per operand row and stage
constexpr int NSTAGE = 4;
constexpr int A_BYTES = 16384;                     // 256 rows x 64 B
constexpr int STAGE_BYTES = 2 * A_BYTES;
constexpr int STG_BYTES = 2048;                    // per-wave epilogue staging: 32 rows x 32 columns bf16
constexpr int OPS_SLOT = 5120;                                     // per parity: sw 256 f32 | bias 256 f32 | sa 256 f32 -- or, LayerNorm folded
                                                                   // (KIND 3 / 4): sw | b' | column sums s | 256 x (rstd, -mean rstd)
constexpr int OPS_OFF = NSTAGE * STAGE_BYTES + 8 * STG_BYTES;      // [2 parities][OPS_SLOT] = 10 KiB
constexpr int MXS_OFF = OPS_OFF + 2 * OPS_SLOT;                    // [4 K-tile slots][256 rows x 4 scale bytes] = 4 KiB
static_assert(MXS_OFF + 4096 <= 160 * 1024, "LDS budget");
constexpr int LDS_BYTES = 160 * 1024;

// KIND 0: per-row activation scales sa, bf16 output (+ bias, optional ReLU / GELU)                      (qkv)
// KIND 1: MX input (e8m0 per 32 k of A, tiled [K/128][M][4]), bf16 output + bias + residual               (proj, fc2)
// KIND 2: per-row activation scales, bias + activation, MX OUTPUT (e4m3 + e8m0 per 32 columns)           (fc1)
// Round 5 -- nn.LayerNorm folded into the e4m3 linear it feeds (reference vision_transformer_dino_mugs.py:136-149; the bf16 form is
// gemm8w_kernel.h's LNF): A = the RAW residual rows, MX-quantised (e8m0 per 32 k) by the epilogue of the linear that produced
// them; W = e4m3(W diag(gamma)) with per-row scales sw; ln_colsum[n] = sw[n] sum_k W8[n][k]; bias = b + W beta;
// ln_stats[m] = (rstd, -mean rstd):   y = rstd (A8 . W8^T) sw - mean rstd ln_colsum + bias  = LayerNorm(x) W^T + b
// KIND 3: MX input + that affine, bf16 output                                                              (qkv)
// KIND 4: MX input + that affine + activation, MX OUTPUT                                                   (fc1)
struct Dev {
    const unsigned char* A; const unsigned char* W; bf16_t* C; const bf16_t* R;
    const float* sa; const float* sw; const float* bias;
    const float* ln_stats; const float* ln_colsum;
    const unsigned char* a_bs; unsigned char* C8; unsigned char* c_bs;
    int M, N, K, lda, ldw, ldc, ldr, ldc8, act;
    int tiles_m, ncol;
};

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MT, int KIND, int ACT>
__global__ __launch_bounds__(512, 2) void gemm8f_kernel(Dev p) {
    constexpr int BM = MT * 64;
    constexpr bool MXA = KIND == 1 || KIND >= 3, MXOUT = KIND == 2 || KIND == 4, RES = KIND == 1, LNF = KIND >= 3;
    constexpr int ESTORES = MT * 4 * (MXOUT ? 2 : 1);      // global stores per lane per full tile epilogue (MX: + the scale bytes)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                // waves w and w + 4 (one SIMD) own the two row halves of a column strip
    const int l31 = lane & 31, h = lane >> 5;

    // ---- this workgroup's tiles: the column-fastest tile list in XCD-major rank order (as the bf16 kernel's linear epilogue) ----
    const int b = blockIdx.x, G = gridDim.x;
    const int q0 = (b & 7) * (G >> 3) + (b >> 3);
    const int total = p.tiles_m * p.ncol;
    const int nt = q0 < total ? (total - q0 + G - 1) / G : 0;
    const int ti = q0 / p.ncol, tj = q0 - ti * p.ncol;
    const int KS = p.K / BKB;
    const int S = nt * KS;
    if (S == 0) return;
    const int step_i = G / p.ncol, step_j = G - step_i * p.ncol;

    // ---- staging: wave w lands row blocks 2w, 2w+1 (16 rows x 64 B each) of both operands per stage ----
    const int srow = lane >> 2;
    const int slog = (lane & 3) ^ ((lane >> 4) & 3);        // logical chunk fetched by this lane (it lands at chunk position lane & 3)
    unsigned w_off[2], a_raw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        w_off[j] = (unsigned)(tj * BN + (wave * 2 + j) * 16 + srow) * (unsigned)p.ldw + slog * 16;
        int r = (wave * 2 + j) * 16 + srow;
        if (r >= BM) r = BM - 1;
        a_raw[j] = (unsigned)(ti * BM + r) * (unsigned)p.lda + slog * 16;
    }
    const unsigned a_unit = (unsigned)BM * (unsigned)p.lda, w_unit = (unsigned)BN * (unsigned)p.ldw;
    const unsigned a_lim = (unsigned)(p.M - 1) * (unsigned)p.lda + 48;
    int l_t = 0, l_ks = 0, l_j = tj, l_i = ti;
    // MX scale ring: the slot of a 128-k tile = its running number & 3, derived from (tile, k stage) on both sides (separate
    // running counters ended up in scratch: the closures' by-reference captures + the epilogue's compiler barriers, and a scratch
    // reload sits behind an s_waitcnt vmcnt(0) -- INSIDE the K loop it drained the stage ring every other stage)
    const int KS2 = KS >> 1;
    auto issue = [&](int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE_BYTES + wave * 2048;
        const int k0 = l_ks * BKB;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(p.A + min(a_raw[j], a_lim) + k0, base + j * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(p.W + w_off[j] + k0, base + A_BYTES + j * 1024);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        // with the first stage of a tile, its epilogue operands (waves 0-2: sw / bias / sa of the tile, 1 KiB each, parity slot)
        if (l_ks == 0 && wave < (LNF ? 5 : MXA ? 2 : 3)) {
            char* dst = smem + OPS_OFF + (l_t & 1) * OPS_SLOT + wave * 1024;
            // (ragged last tile: whole 4-row granules past M are clamped to the granule that holds row M - 1 -- their rows are
            // masked at the store; that granule itself may reach up to 3 floats past sa[M - 1]: cvcl_hip.h asks for the padding)
            const float* src = wave == 0 ? p.sw + l_j * BN + lane * 4 : wave == 1 ? p.bias + l_j * BN + lane * 4
                                                                                 : p.sa + min(l_i * BM + lane * 4, (p.M - 1) & ~3);
            if constexpr (LNF) {                   // waves 2: the column sums; 3 / 4: (rstd, -mean rstd) of the upper / lower 128 rows
                if (wave == 2) src = p.ln_colsum + l_j * BN + lane * 4;
                else if (wave >= 3) src = p.ln_stats + (long)min(l_i * BM + (wave - 3) * 128 + lane * 2, (p.M - 1) & ~1) * 2;
            }
            if (wave == 1 && !p.bias) *reinterpret_cast<f32x4*>(dst + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};     // no bias: zeros
            else glds16(src, dst);
        }
        // MX input: with the first stage of a 128-k tile, the 4 block-scale bytes of the tile's 256 rows (1 KiB; one wave per K tile)
        if constexpr (MXA) {
            if ((l_ks & 1) == 0) {
                const int l_slot = (l_t * KS2 + (l_ks >> 1)) & 3;
                if (wave == (LNF ? 4 : 2) + l_slot)
                    glds16(p.a_bs + ((long)(l_ks >> 1) * p.M + min(l_i * BM + lane * 4, (p.M - 1) & ~3)) * 4,     // (as sa: granules of 4 rows)
                           smem + MXS_OFF + l_slot * 1024);
            }
        }
        l_ks = __builtin_amdgcn_readfirstlane(l_ks + 1);
        if (l_ks == KS) {
            l_ks = 0;
            if (l_t + 1 < nt) {
                l_t = __builtin_amdgcn_readfirstlane(l_t + 1);
                int di = step_i, dj = step_j;
                if (l_j + dj >= p.ncol) { dj -= p.ncol; ++di; }
                l_j = __builtin_amdgcn_readfirstlane(l_j + dj);
                l_i = __builtin_amdgcn_readfirstlane(l_i + di);
                const unsigned da = (unsigned)di * a_unit, dw = (unsigned)dj * w_unit;
#pragma unroll
                for (int j = 0; j < 2; ++j) { a_raw[j] += da; w_off[j] += dw; }
            }
        }
    };

    // ---- fragment addressing: lane -> row lane & 31 of a 32-row tile, chunks h and h + 2 ----
    const int f_sw = (l31 >> 2) & 3;
    const int f_lo = l31 * 64 + ((h ^ f_sw) << 4), f_hi = l31 * 64 + (((h + 2) ^ f_sw) << 4);
    const int a_base = wm * (BM / 2) * 64, w_base = A_BYTES + wn * 64 * 64;

    v8i fa[2][MT], fw[2][2];
    f32x16 acc[2][MT];
    unsigned aw[MT];                                         // MX input: the 4 scale bytes of this lane's rows for the current 128-k tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) aw[mt] = 0x7f7f7f7fu;
    auto read_one = [&](const char* sb, int off) __attribute__((always_inline)) -> v8i {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(sb + off + f_lo), hi = *reinterpret_cast<const u32x4*>(sb + off + f_hi);
        return v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };
    auto read_frags = [&](int buf, auto P) __attribute__((always_inline)) {
        constexpr int q = decltype(P)::value;
        const char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) fw[q][n2] = read_one(sb, w_base + n2 * 2048);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[q][mt] = read_one(sb, a_base + mt * 2048);
    };
    auto mma_half = [&](auto P, auto HALF, int ks) __attribute__((always_inline)) {
        constexpr int q = decltype(P)::value, hh = decltype(HALF)::value;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            acc[hh][mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                fw[q][hh], fa[q][mt], acc[hh][mt], 0, 0, 0, 0x7f7f7f7f, 0,
                MXA ? (int)(aw[mt] >> (8 * (2 * (ks & 1) + h))) : 0x7f7f7f7f);     // byte 0 = the scale of k block 2 (ks & 1) + h
    };
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n2][mt][e] = 0.f;

    char* stg = smem + NSTAGE * STAGE_BYTES + wave * STG_BYTES;
    const int r_chunk = lane & 7, r_row0 = lane >> 3;        // read-back of a 16-row x 64-column block: row 8 j + (lane >> 3), 16-byte chunk lane & 7
    const int e_row = lane & 15, e_half = (lane >> 4) & 1;   // staging: this lane's row l31 = 16 e_half + e_row

    // -> a lower bound on the VMEM instructions this call issued (exact for a full tile)
    auto epilogue = [&](int m0, int n0, int parity) __attribute__((always_inline)) -> int {
        const bool full = m0 + BM <= p.M;
        const float* ops = reinterpret_cast<const float*>(smem + OPS_OFF + parity * OPS_SLOT);
        bf16x8 rr[2];
        auto load_res = [&](int blk) __attribute__((always_inline)) {       // blk = mt * 2 + half: 16 rows x 64 columns
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int m = m0 + wm * (BM / 2) + blk * 16 + j * 8 + r_row0;
                if (m >= p.M) m = p.M - 1;
                rr[j] = *reinterpret_cast<const bf16x8*>(p.R + (long)m * p.ldr + n0 + wn * 64 + r_chunk * 8);
            }
        };
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float sa_r = 1.f;
            if constexpr (!MXA) sa_r = ops[512 + wm * (BM / 2) + mt * 32 + l31];
            f32x2 rs = {1.f, 0.f};                 // LayerNorm folded: this lane's row (rstd, -mean rstd)
            if constexpr (LNF) rs = *reinterpret_cast<const f32x2*>(ops + 768 + (wm * (BM / 2) + mt * 32 + l31) * 2);
            // scales, bias, activation -> bf16: this lane's 32 values of row l31 (8 pieces of 4 consecutive columns)
            bf16x4 qv[2][4];
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cb = wn * 64 + n2 * 32 + 8 * g + 4 * h;
                    const f32x4 sw_r = *reinterpret_cast<const f32x4*>(ops + cb);
                    const f32x4 bias_r = *reinterpret_cast<const f32x4*>(ops + 256 + cb);
                    f32x4 cs_r = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (LNF) cs_r = *reinterpret_cast<const f32x4*>(ops + 512 + cb);
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {                       // pairs: packed fp32 arithmetic (bit-identical per element)
                        f32x2 v = f32x2{acc[n2][mt][4 * g + e], acc[n2][mt][4 * g + e + 1]};
                        if constexpr (!MXA) v = v * f32x2{sa_r, sa_r};
                        if constexpr (LNF)     // rstd (acc sw) + (-mean rstd) s + b'
                            v = __builtin_elementwise_fma(v * f32x2{sw_r[e], sw_r[e + 1]}, f32x2{rs[0], rs[0]},
                                                          __builtin_elementwise_fma(f32x2{rs[1], rs[1]}, f32x2{cs_r[e], cs_r[e + 1]},
                                                                                    f32x2{bias_r[e], bias_r[e + 1]}));
                        else
                        v = __builtin_elementwise_fma(v, f32x2{sw_r[e], sw_r[e + 1]}, f32x2{bias_r[e], bias_r[e + 1]});
                        if (ACT == CVCL_ACT_RELU) v = f32x2{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
                        if (ACT == CVCL_ACT_GELU) v = gelu_bf16out2(v);
                        qv[n2][g][e] = (bf16_t)v[0];
                        qv[n2][g][e + 1] = (bf16_t)v[1];
                        acc[n2][mt][4 * g + e] = 0.f;
                        acc[n2][mt][4 * g + e + 1] = 0.f;
                    }
                }
            // 16 rows at a time through the wave's 2 KiB staging block (16 rows x 128 B, chunk ^ (row & 7)): the lanes of the other
            // row half sit out the write; the read-back gives every lane 16 bytes of a full 128-byte row segment
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if constexpr (RES) load_res(mt * 2 + half);
                // (compiler barriers: the staging block is exchanged BETWEEN LANES; without them the compiler reasons per thread --
                // "a lane that skipped the write re-reads what it read last time" -- and moved a read into the masked region)
                asm volatile("" ::: "memory");
                if (e_half == half) {
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<bf16x4*>(stg + e_row * 128 + (((n2 * 4 + g) ^ (e_row & 7)) << 4) + h * 8) = qv[n2][g];
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = j * 8 + r_row0;
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((r_chunk ^ (row & 7)) << 4));
                    const int m = m0 + wm * (BM / 2) + mt * 32 + half * 16 + row, n = n0 + wn * 64 + r_chunk * 8;
                    if constexpr (!MXOUT) {
                        if (full || m < p.M) {
                            if constexpr (RES) {
                                const bf16x8 r = rr[j];
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)r[e]);
                            }
                            stream_store(v, reinterpret_cast<bf16x8*>(p.C + (long)m * p.ldc + n));
                        }
                    } else {
                        // e4m3 + one e8m0 scale per 32 columns of the row: the block = 4 adjacent lanes (chunks 4b .. 4b+3)
                        const u32x4 vw = __builtin_bit_cast(u32x4, v);
                        unsigned mb = bf16x8_absmax_bits(vw);
                        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xB1, 0xf, 0xf, true));
                        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x4E, 0xf, 0xf, true));
                        const unsigned sb = mx_scale_byte(__uint_as_float(mb << 16));
                        if (full || m < p.M) {
                            const u32x2 w8 = bf16x8_to_fp8_scaled(vw, __uint_as_float(sb << 23));
                            stream_store(w8, reinterpret_cast<u32x2*>(p.C8 + (long)m * p.ldc8 + n));
                            if ((r_chunk & 3) == 0) p.c_bs[((long)(n >> 7) * p.M + m) * 4 + ((n >> 5) & 3)] = (unsigned char)sb;
                        }
                    }
                }
            }
        }
        return full ? ESTORES : 0;
    };

    // ---- prologue: stages 0..3 in flight, stage 0 landed and in registers ----
    issue(0); advance(); issue(1); advance(); issue(2); advance(); issue(3); advance();
    wait_vm<12>();
    __builtin_amdgcn_s_barrier();
    read_frags(0, std::integral_constant<int, 0>{});
    const int mxs_row = (wm * (BM / 2) + l31) * 4;
    auto read_scales = [&](int kt_running) __attribute__((always_inline)) {   // the scales of that 128-k tile (running number) -> aw
        if constexpr (MXA) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                aw[mt] = *reinterpret_cast<const unsigned*>(smem + MXS_OFF + (kt_running & 3) * 1024 + mxs_row + mt * 128);
        }
    };
    read_scales(0);

    int after_epi = 0, epi_ops = 0;
    int c_ks = 0, c_i = ti, c_j = tj, c_t = 0;
    auto step = [&](int g, auto P) __attribute__((always_inline)) {
        constexpr int q = decltype(P)::value;
        mma_half(P, std::integral_constant<int, 0>{}, c_ks);
        if (after_epi > 0 && epi_ops == ESTORES) wait_vm<8 + ESTORES>();
        else wait_vm<8>();
        if (after_epi > 0) --after_epi;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // The LAST stage of a tile reads the next tile's first fragments only AFTER the epilogue: held across it they are 8 (2 + MT)
        // registers the epilogue needs (the MT = 4 instantiations spilled 32-116 bytes per lane); the reads are then not hidden
        // behind MFMAs once per tile (K / 64 >= 4 stages).
        if (c_ks + 1 == KS) {
            issue(g & 3);
            mma_half(P, std::integral_constant<int, 1>{}, c_ks);
            advance();
            c_ks = 0;
            epi_ops = epilogue(c_i * BM, c_j * BN, c_t & 1);
            c_t = __builtin_amdgcn_readfirstlane(c_t + 1);
            after_epi = 3;
            c_i += step_i;
            c_j += step_j;
            if (c_j >= p.ncol) { c_j -= p.ncol; ++c_i; }
            c_i = __builtin_amdgcn_readfirstlane(c_i);
            c_j = __builtin_amdgcn_readfirstlane(c_j);
            __builtin_amdgcn_sched_barrier(0);
            read_frags((g + 1) & 3, std::integral_constant<int, 1 - q>{});
            if constexpr (MXA) read_scales(c_t * KS2);           // (KS is even: the next stage opens a K tile; c_t = the next tile already)
        } else {
            read_frags((g + 1) & 3, std::integral_constant<int, 1 - q>{});
            issue(g & 3);                                       // stage g+4 into the buffer stage g occupied
            mma_half(P, std::integral_constant<int, 1>{}, c_ks);
            // one MFMA between any two of the 2 (2 + MT) reads / 4 loads of this half (MT MFMAs: the rest follows them)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            advance();
            // the next stage's MX scales (its K tile's slot was filled >= 2 stages ago, behind a counted wait and this barrier)
            if constexpr (MXA) {
                if (c_ks & 1) { __builtin_amdgcn_sched_barrier(0); read_scales(c_t * KS2 + ((c_ks + 1) >> 1)); }   // (the next stage opens a K tile)
            }
            c_ks = __builtin_amdgcn_readfirstlane(c_ks + 1);
        }
    };
    for (int g = 0; g < S; g += 2) {
        step(g, std::integral_constant<int, 0>{});
        step(g + 1, std::integral_constant<int, 1>{});
    }
    wait_vm<0>();                                            // the re-loads past the end must not outlive the workgroup
}

}  // namespace g8f
