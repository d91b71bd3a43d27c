// 256 x 256 bf16 GEMM tile, FOUR waves per workgroup -- one per SIMD -- each owning a 128 x 128 block of the output.
//
//      C[M,N] = A[M,K] . W[N,K]^T          (A, W row-major, K contiguous; N % 256 == 0, K % 128 == 0)
//
// Why (next to gemm8w_kernel.h): the 8-wave kernel's wave tile is (16 MI) x 64: 12 fragment reads per 32 MFMAs (0.375 per MFMA),
// 96 KB of ds_read_b128 traffic per 32-deep stage per CU, and its ablation (profiles/r02_gemm_lab.txt) puts the loss in exactly
// that traffic.  A 128 x 128 wave tile reads 16 fragments per 64 MFMAs (0.25 per MFMA, 64 KB per stage per CU).  Its 256
// accumulator registers do not fit beside a second wave on the SIMD: the kernel is built for ONE wave per SIMD
// (__launch_bounds__(256) + amdgpu_waves_per_eu(1, 1): the wave owns the SIMD's whole 512-entry unified register file --
// 256 accumulators + 128 fragment registers, double buffered, + addresses).  With nobody else on the SIMD every stall is
// exposed, so the wave is its own latency hiding: the fragment reads of stage g+1 and the global->LDS loads of stage g+4 are
// interleaved one by one with the second half of stage g's 64 MFMAs.
//
// Pipeline (same shape as gemm8w): "stage" = 32 k = 64 bytes per operand row; 4-stage LDS ring of 32 KiB stages filled by
// global_load_lds_dwordx4 (16 rows x 64 B per wave instruction, 16-byte chunk c of row r at chunk position c ^ swz((r >> 2) & 3),
// swizzle applied on the SOURCE address); one barrier per stage; counted vmcnt waits (loads and stores retire in issue order).
// Epilogue per 16-row block through 4 KiB of wave-private LDS (rows of 256 B, chunk ^ (row & 15)): full 256-byte row segments.
#pragma once
#include <type_traits>

#include "../../multimodal-baby_amd/csrc/cvcl_common.h"

namespace g4w {

constexpr int BN = 256;
constexpr int BK = 32;
constexpr int NSTAGE = 4;
constexpr int A_BYTES = 16384;                    // 256 rows x 64 B
constexpr int STAGE_BYTES = 2 * A_BYTES;          // + W: 256 rows x 64 B
constexpr int STG_BYTES = 4096;                   // per-wave epilogue staging: 16 rows x 256 B
constexpr int ACC_OFF = NSTAGE * STAGE_BYTES + 4 * STG_BYTES;       // [4 waves][2][128] f32 BN partial sums + [256] -centre / [N] f32 bias
constexpr int LDS_BYTES = 160 * 1024;
constexpr int MAX_BIAS_N = (LDS_BYTES - ACC_OFF) / 4;               // 4096
constexpr int NI = 8;                             // 16 x 16 accumulator tiles per wave along N (MI of them along M: template)
constexpr int LOADS = 8;                          // global_load_lds instructions per wave per stage (4 A + 4 W row blocks)

// EPI 0: C = round(acc) (+ BN partial sums when stats != nullptr; C may be nullptr: statistics only); acc starts at -centre[n]
// EPI 1: C = round(act(acc + bias))                 (nn.Linear: bias / ReLU / GELU)
// EPI 2: C = round(round(acc + bias) + R)            (nn.Linear + residual)
struct Dev {
    const bf16_t* A; const bf16_t* W; bf16_t* C; const bf16_t* R;
    const float* bias; float* stats;
    const float* centre;
    int M, N, K, lda, ldw, ldc, ldr, act;
    int tiles_m, grid_m, ncol;
    int gs, g_hw, g_wo, g_hi, g_wi;     // row gather of a strided 1x1 convolution (gs <= 1 = off)
    int a_rows;
};

__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swz(int g) { return (0x78 >> (2 * g)) & 3; }          // {0, 2, 3, 1}: see gemm8w_kernel.h

// The accumulators are pinned to the AGPR half of the register file and the fragments to the VGPR half by the operand
// constraints: left to itself the allocator mixes the two classes (fragments in AGPRs, accumulator tiles spilled to scratch
// inside the K loop).  The instruction is opaque to the compiler's hazard recogniser: the only consumers of its result are the
// epilogue's v_accvgpr_reads, which sit behind an explicit s_nop pad (mfma_fence).
// (volatile + "memory": the MFMAs, the LDS fragment reads and the global->LDS loads stay in SOURCE order -- the interleave of
// a stage's second half is written out by hand, the compiler's scheduler cannot see through the asm to do it)
__device__ __forceinline__ void mfma(f32x4& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a) : "memory");
}
// first K stage of an output tile: the accumulator is DEFINED here (C operand = the inline constant 0; C and D of an MFMA
// share one register file, so a VGPR-resident initial value is not encodable) -- no VALU instruction ever writes an
// accumulator, so every definition the allocator sees is an AGPR one.  The storage centre of the convolution epilogue is
// therefore subtracted when the accumulators are read out (round(acc + (-centre))).
__device__ __forceinline__ void mfma_first0(f32x4& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(w), "v"(a) : "memory");
}
__device__ __forceinline__ void mfma_fence() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 44) asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
    else if constexpr (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else static_assert(N < 0, "add the literal");
}

// VAR (lab switches): bit 1 = interleave the post-barrier fragment reads / stage loads with the second half's MFMAs.
// Ablations (WRONG results, timing only): bit 2 = no barrier, bit 3 = no stage loads, bit 4 = no fragment reads, bit 5 = no epilogue.
// MI = 8: 256-row tiles, all 256 AGPRs are accumulators; MI = 7: 224-row tiles (M of the ResNeXt activations is 49 x 2^k).
template <int MI, int EPI, int VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4w_kernel(Dev p) {
    constexpr int BM = MI * 32;
    constexpr int ESTORES = MI * 4;                        // global stores per lane per full tile epilogue
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    constexpr bool LIN = EPI >= 1, RES = EPI == 2;
    constexpr bool FLAT = LIN;
    const int b = blockIdx.x;
    const int G = gridDim.x;
    int ti, tj, nt;
    if constexpr (FLAT) {
        const int q = (b & 7) * (G >> 3) + (b >> 3);
        const int total = p.tiles_m * p.ncol;
        nt = q < total ? (total - q + G - 1) / G : 0;
        ti = q / p.ncol;
        tj = q - ti * p.ncol;
    } else {
        const int xcd = b & 7, s = b >> 3;
        tj = s % p.ncol;
        ti = (s / p.ncol) * 8 + xcd;
        nt = ti < p.tiles_m ? (p.tiles_m - ti + p.grid_m - 1) / p.grid_m : 0;
    }
    const int KS = p.K / BK;
    const int S = nt * KS;
    if (S == 0) {
        if (EPI == 0 && p.stats && ti < p.grid_m) {
            p.stats[((long)ti * 2 + 0) * p.N + tj * BN + tid] = 0.f;
            p.stats[((long)ti * 2 + 1) * p.N + tj * BN + tid] = 0.f;
        }
        return;
    }
    const int step_i = FLAT ? G / p.ncol : p.grid_m;
    const int step_j = FLAT ? G - step_i * p.ncol : 0;

    const bf16_t* __restrict__ A = p.A;
    const bf16_t* __restrict__ W = p.W;

    // ---- staging: wave w lands row blocks 4w .. 4w+3 (16 rows x 64 B each) of both operands per stage ----
    const int srow = lane >> 2;
    const int slog = (lane & 3) ^ swz((lane >> 4) & 3);
    unsigned w_off[4], a_raw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w_off[j] = (unsigned)(tj * BN + (wave * 4 + j) * 16 + srow) * (unsigned)p.ldw + slog * 8;
        int r = (wave * 4 + j) * 16 + srow;
        if (r >= BM) r = BM - 1;                            // BM < 256: rows of the unused part of the A region
        a_raw[j] = (unsigned)(ti * BM + r) * (unsigned)p.lda + slog * 8;
    }
    auto gathered = [&](int i_tile, int j) __attribute__((always_inline)) -> unsigned {
        int r = (wave * 4 + j) * 16 + srow;
        if (r >= BM) r = BM - 1;
        const unsigned m = (unsigned)min(i_tile * BM + r, p.M - 1);
        const unsigned bi = m / (unsigned)p.g_hw, rem = m - bi * (unsigned)p.g_hw;
        const unsigned oy = rem / (unsigned)p.g_wo, ox = rem - oy * (unsigned)p.g_wo;
        return ((bi * p.g_hi + oy * p.gs) * p.g_wi + ox * p.gs) * (unsigned)p.lda + slog * 8;
    };
    if (p.gs > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a_raw[j] = gathered(ti, j);
    }
    const unsigned a_unit = (unsigned)BM * (unsigned)p.lda, w_unit = (unsigned)BN * (unsigned)p.ldw;
    const unsigned a_lim = (unsigned)(p.a_rows - 1) * (unsigned)p.lda + 24;
    int l_t = 0, l_ks = 0, l_j = tj, l_i = ti;
    auto issue = [&](int buf) __attribute__((always_inline)) {
        if constexpr (VAR & 8) return;
        char* base = smem + buf * STAGE_BYTES + wave * 4096;
        const int k0 = l_ks * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(A + min(a_raw[j], a_lim) + k0, base + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(W + w_off[j] + k0, base + A_BYTES + j * 1024);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if (++l_ks == KS) {
            l_ks = 0;
            if (l_t + 1 < nt) {
                ++l_t;
                int di = step_i, dj = step_j;
                if constexpr (FLAT) {
                    if (l_j + dj >= p.ncol) { dj -= p.ncol; ++di; }
                    l_j += dj;
                }
                const unsigned da = (unsigned)di * a_unit, dw = (unsigned)dj * w_unit;
                l_i += di;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (p.gs > 1) a_raw[j] = gathered(l_i, j);
                    else a_raw[j] += da;
                    w_off[j] += dw;
                }
            }
        }
    };

    // ---- fragment addressing: lane -> row lane & 15 of a 16-row block, logical chunk lane >> 4 ----
    const int f_off = (lane & 15) * 64 + (((lane >> 4) ^ swz((lane >> 2) & 3)) << 4);
    const int a_base = wm * (BM / 2) * 64 + f_off;
    const int w_base = A_BYTES + wn * (BN / 2) * 64 + f_off;

    bf16x8 fa[2][MI], fw[2][NI];
    f32x4 acc[NI][MI];
    auto read_frags = [&](int buf, auto P) __attribute__((always_inline)) {
        constexpr int q = decltype(P)::value;
        if constexpr (VAR & 16) return;
        const char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fw[q][ni] = *reinterpret_cast<const bf16x8*>(sb + w_base + ni * 1024);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[q][mi] = *reinterpret_cast<const bf16x8*>(sb + a_base + mi * 1024);
    };
    float* lds_acc = reinterpret_cast<float*>(smem + ACC_OFF);
    // this lane's four columns of accumulator tile ni: + ni * 16 (floats)
    const float* cen = lds_acc + 1024 + wn * (BN / 2) + (lane >> 4) * 4;
    // first = this K stage opens an output tile: its MFMAs define the accumulators (from 0) instead of adding to them
    auto mma_half = [&](auto P, auto HALF, bool first) __attribute__((always_inline)) {
        constexpr int q = decltype(P)::value, h = decltype(HALF)::value;
        (void)cen;
        if (first) {
#pragma unroll
            for (int ni = (NI / 2) * h; ni < (NI / 2) * (h + 1); ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) mfma_first0(acc[ni][mi], fw[q][ni], fa[q][mi]);
        } else {
#pragma unroll
            for (int ni = (NI / 2) * h; ni < (NI / 2) * (h + 1); ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    mfma(acc[ni][mi], fw[q][ni], fa[q][mi]);
        }
    };

    if constexpr (LIN) {
        for (int i = tid; i < p.N; i += 256) lds_acc[i] = p.bias ? p.bias[i] : 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_acc[tid + 256 * i] = 0.f;                   // [4 waves][2][128]
        lds_acc[1024 + tid] = p.centre ? -p.centre[tj * BN + tid] : 0.f;            // -centre of the column tile's 256 columns
    }
    char* stg = smem + NSTAGE * STAGE_BYTES + wave * STG_BYTES;
    const int e_row = lane & 15;                             // accumulator layout: m = mi*16 + (lane & 15), n = ni*16 + (lane >> 4)*4 + e
    const int e_wchunk = lane >> 5, e_wsub = ((lane >> 4) & 1) * 8;
    const int r_chunk = lane & 15, r_row0 = lane >> 4;       // read-back: row 4j + (lane >> 4), 16-byte chunk lane & 15

    auto epilogue = [&](int m0, int n0) __attribute__((always_inline)) -> int {
        const bool full = m0 + BM <= p.M;
        if constexpr (VAR & 32) { if (m0 != 0) return 0; }
        mfma_fence();
        float st_sum[8], st_sq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { st_sum[e] = 0.f; st_sq[e] = 0.f; }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            bf16x8 rr[4];
            if constexpr (RES) {                             // residual rows of this block (the stores of the previous block cover the latency)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int m = m0 + wm * (BM / 2) + mi * 16 + j * 4 + r_row0;
                    if (m >= p.M) m = p.M - 1;
                    rr[j] = *reinterpret_cast<const bf16x8*>(p.R + (long)m * p.ldr + n0 + wn * (BN / 2) + r_chunk * 8);
                }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                bf16x4 q;
                if constexpr (LIN) {
                    const f32x4 bias_r = *reinterpret_cast<const f32x4*>(lds_acc + n0 + wn * (BN / 2) + ni * 16 + (lane >> 4) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[ni][mi][e] + bias_r[e];
                        if constexpr (!RES) {
                            if (p.act == CVCL_ACT_RELU) v = fmaxf(v, 0.f);
                            else if (p.act == CVCL_ACT_GELU) v = gelu_erf_fast(v);
                        }
                        q[e] = (bf16_t)v;
                    }
                } else {
                    const f32x4 cn = *reinterpret_cast<const f32x4*>(cen + ni * 16);     // -centre of these four columns
                    q = bf16x4{(bf16_t)(acc[ni][mi][0] + cn[0]), (bf16_t)(acc[ni][mi][1] + cn[1]), (bf16_t)(acc[ni][mi][2] + cn[2]),
                               (bf16_t)(acc[ni][mi][3] + cn[3])};
                }
                const int chunk = ni * 2 + e_wchunk;
                *reinterpret_cast<bf16x4*>(stg + e_row * 256 + ((chunk ^ e_row) << 4) + e_wsub) = q;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = j * 4 + r_row0;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * 256 + ((r_chunk ^ row) << 4));
                const int m = m0 + wm * (BM / 2) + mi * 16 + row, n = n0 + wn * (BN / 2) + r_chunk * 8;
                if (full || m < p.M) {
                    if constexpr (LIN) {
                        if constexpr (RES) {
                            const bf16x8 r = rr[j];
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)r[e]);
                        }
                        stream_store(v, reinterpret_cast<bf16x8*>(p.C + (long)m * p.ldc + n));
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = (float)v[e];
                            st_sum[e] += f;
                            st_sq[e] = fmaf(f, f, st_sq[e]);
                        }
                        if (p.C) stream_store(v, reinterpret_cast<bf16x8*>(p.C + (long)m * p.ldc + n));
                    }
                }
            }
        }
        if (EPI == 0 && p.stats) {                           // this tile's column sums into the wave's slot, fixed order
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    st_sum[e] += __shfl_xor(st_sum[e], o, 64);
                    st_sq[e] += __shfl_xor(st_sq[e], o, 64);
                }
            }
            if (lane < 16) {
                float* s0 = lds_acc + (wave * 2 + 0) * 128 + lane * 8;
                float* s1 = lds_acc + (wave * 2 + 1) * 128 + lane * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) { s0[e] += st_sum[e]; s1[e] += st_sq[e]; }
            }
        }
        return (full && (LIN || p.C != nullptr)) ? ESTORES : 0;
    };

    if constexpr (VAR & 16) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < NI; ++i) fw[q][i] = bf16x8{};
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[q][i] = bf16x8{};
        }
    }
    // ---- prologue: stages 0..3 in flight, stage 0 landed and in registers ----
    issue(0); advance(); issue(1); advance(); issue(2); advance(); issue(3); advance();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this thread's LDS writes above (partial sums, -centre / bias)
    wait_vm<3 * LOADS>();
    __builtin_amdgcn_s_barrier();
    read_frags(0, std::integral_constant<int, 0>{});

    // second half of a stage with the memory operations of the pipeline written between its MFMAs, one per MFMA: the NI + MI
    // fragment reads of stage g+1 (into the other register set), then the 8 global->LDS loads of stage g+4
    auto mma_half1_interleaved = [&](int g, auto P, auto FIRST) __attribute__((always_inline)) {
        constexpr int q = decltype(P)::value;
        constexpr bool first = decltype(FIRST)::value;
        const char* sb = smem + ((g + 1) & 3) * STAGE_BYTES;
        char* lbase = smem + (g & 3) * STAGE_BYTES + wave * 4096;
        const int k0 = l_ks * BK;
        int slot = 0;
#pragma unroll
        for (int ni = NI / 2; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if constexpr (first) mfma_first0(acc[ni][mi], fw[q][ni], fa[q][mi]);
                else mfma(acc[ni][mi], fw[q][ni], fa[q][mi]);
                const int sl = slot++;
                if (sl < NI) {
                    if constexpr (!(VAR & 16)) fw[1 - q][sl] = *reinterpret_cast<const bf16x8*>(sb + w_base + sl * 1024);
                } else if (sl < NI + MI) {
                    if constexpr (!(VAR & 16)) fa[1 - q][sl - NI] = *reinterpret_cast<const bf16x8*>(sb + a_base + (sl - NI) * 1024);
                } else if (sl < NI + MI + 4) {
                    const int j = sl - NI - MI;
                    if constexpr (!(VAR & 8)) glds16(A + min(a_raw[j], a_lim) + k0, lbase + j * 1024);
                } else if (sl < NI + MI + 8) {
                    const int j = sl - NI - MI - 4;
                    if constexpr (!(VAR & 8)) glds16(W + w_off[j] + k0, lbase + A_BYTES + j * 1024);
                }
            }
        static_assert(NI + MI + 8 <= (NI / 2) * MI, "the second half must have an MFMA for every read / load");
    };

    // Loop nest: output tiles outside, K stages inside, the FIRST stage of every tile peeled (its MFMAs define the accumulators)
    // and the epilogue behind the K loop -- the accumulators live from one definition to one read inside a tile iteration and
    // cross no conditional (with the epilogue as a branch inside a flat stage loop the allocator spilled accumulator tiles in
    // the K loop).  The LOAD side is not tiled: it runs 3.5 stages ahead across tile boundaries (issue / advance).
    int after_epi = 0, epi_ops = 0;
    int c_i = ti, c_j = tj;
    auto step = [&](int g, auto P, auto FIRST) __attribute__((always_inline)) {
        constexpr int q = decltype(P)::value;
        constexpr bool first = decltype(FIRST)::value;
        mma_half(P, std::integral_constant<int, 0>{}, first);
        // stage g+1 has landed (this wave's part): exactly the 2 x LOADS younger loads may be outstanding (+ an epilogue's stores)
        if (after_epi > 0 && epi_ops == ESTORES) wait_vm<2 * LOADS + ESTORES>();
        else wait_vm<2 * LOADS>();
        if (after_epi > 0) --after_epi;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!(VAR & 4)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (VAR & 2) {
            mma_half1_interleaved(g, P, FIRST);
        } else {
            read_frags((g + 1) & 3, std::integral_constant<int, 1 - q>{});
            issue(g & 3);                                   // stage g+4 into the buffer stage g occupied
            mma_half(P, std::integral_constant<int, 1>{}, first);
        }
        advance();
    };
    for (int t = 0; t < nt; ++t) {                           // (KS % 4 == 0: a tile starts on ring buffer 0 and fragment set 0)
        step(0, std::integral_constant<int, 0>{}, std::true_type{});
        step(1, std::integral_constant<int, 1>{}, std::false_type{});
        for (int ks = 2; ks < KS; ks += 2) {
            step(ks, std::integral_constant<int, 0>{}, std::false_type{});
            step(ks + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }
        epi_ops = epilogue(c_i * BM, c_j * BN);
        after_epi = 3;
        c_i += step_i;
        if constexpr (FLAT) {
            c_j += step_j;
            if (c_j >= p.ncol) { c_j -= p.ncol; ++c_i; }
        }
    }
    wait_vm<0>();

    if (EPI == 0 && p.stats) {
        __syncthreads();
        // column strip wn (128 columns): waves (0, wn) and (1, wn)
        const int wn_ = tid >> 7, c = tid & 127, n = tj * BN + tid;
        const float sv = lds_acc[((0 * 2 + wn_) * 2 + 0) * 128 + c] + lds_acc[((1 * 2 + wn_) * 2 + 0) * 128 + c];
        const float qv = lds_acc[((0 * 2 + wn_) * 2 + 1) * 128 + c] + lds_acc[((1 * 2 + wn_) * 2 + 1) * 128 + c];
        p.stats[((long)ti * 2 + 0) * p.N + n] = sv;
        p.stats[((long)ti * 2 + 1) * p.N + n] = qv;
    }
}

}  // namespace g4w
