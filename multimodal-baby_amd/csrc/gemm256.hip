// 256 x 256 bf16 GEMM with a phase-interleaved schedule for the MFMA-bound shapes (ResNeXt layers 3-4 1x1 convolutions,
// ViT-B linears):   C[M,N] = A[M,K] . W[N,K]^T   (+ epilogues of gemm.hip: EPI 0 conv + BN statistics, EPI 1 bias /
// activation / residual).  Same interface as cvcl_gemm; selected by the dispatcher in gemm.hip.
//
// Why a second kernel: the 128 x 128 kernel moves 1 B of operand per 64 FLOP and keeps 64 KiB in flight per CU; on
// K >= 512 shapes it sits at 700-770 TFLOP/s with the LDS read port as busy as the MFMA pipe.  A 256 x 256 tile halves
// the operand bytes per FLOP and (128 x 64 per wave) cuts LDS fragment reads per MFMA from 1 to 0.75.
//
// Structure (one workgroup of 8 waves per CU, persistent over output tiles):
//   * LDS: 2 K-tile buffers x 4 "halves" (A0, A1, B0, B1; 128 rows x 128 B each) = 128 KiB, filled by
//     global_load_lds_dwordx4 with the XOR chunk swizzle on the source side (as gemm_glds_kernel), + 8 x 4 KiB
//     wave-private strips for the epilogue.  Half h of A holds the rows every wave needs for its sub-tile h
//     (rows wm*128 + h*64 ..), half h of B the columns wn*64 + h*32 ...
//   * a K tile is consumed in 4 phases, one 64 x 32 quadrant of the wave's 128 x 64 output each, ordered
//     (A0,B0) (A0,B1) (A1,B1) (A1,B0) so that a phase reads at most one new A sub-tile (8 ds_read_b128) and one new B
//     sub-tile (4); B0 stays in registers for the 4th phase.  Every phase = [fragment reads + 2 global_load_lds +
//     counted vmcnt | s_barrier | 8 MFMA 32x32x16 | s_barrier].
//   * the two wave groups (wm = 0 / 1; one wave of each per SIMD) run half a phase apart: while one group issues its
//     MFMAs the other issues LDS reads and global loads.
//   * staging runs 4 phases (one K tile) ahead: phase p of K tile G issues B1(G+1), A1(G+1), A0(G+2), B0(G+2) for
//     p = 0..3, and waits with vmcnt(6) for the half issued 4 phases earlier, which is first read one phase later
//     (RAW: own vmcnt + a barrier before any reader; WAR: a half is re-staged >= 3 half-phases after its last ds_read of
//     either group -- derivation in DESIGN.md).
#include <cstdlib>

#include "cvcl_common.h"

namespace {

constexpr int T2 = 256;                 // tile edge
constexpr int HALF_BYTES = 128 * 128;   // 128 rows x 64 bf16
constexpr int BUF_BYTES = 4 * HALF_BYTES;
constexpr int STRIP_BYTES = 4096;       // epilogue strip per wave: 32 rows x 128 B
constexpr int G2_LDS = 2 * BUF_BYTES + 8 * STRIP_BYTES;   // 160 KiB

struct G2Dev {
    const bf16_t* A; const bf16_t* W; bf16_t* C; const bf16_t* R;
    const float* bias; float* stats;
    int M, N, K, lda, ldw, ldc, ldr, act;
    int tiles_m, tiles_m8, tiles_n, ktiles;
};

__device__ __forceinline__ void g2_glds16(const bf16_t* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

__device__ inline float g2_act(float v, int act) {
    if (act == CVCL_ACT_RELU) return fmaxf(v, 0.f);
    if (act == CVCL_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    return v;
}

// linear tile index -> (tm, tn): tiles of one XCD (lt % 8) walk tn fastest over their own tm's (tm % 8 == xcd), so the
// A rows of a tm are fetched into one XCD's L2 once
__device__ __forceinline__ bool g2_tile(const G2Dev& p, int lt, int& tm, int& tn) {
    const int xcd = lt & 7, idx = lt >> 3;
    tn = idx % p.tiles_n;
    tm = (idx / p.tiles_n) * 8 + xcd;
    return tm < p.tiles_m;
}

template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(G2Dev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;

    // ---- this workgroup's output tiles (valid ones only), compacted into a sequence ----
    const int total_lt = p.tiles_m8 * p.tiles_n;
    int nseq = 0;
    for (int lt = blockIdx.x; lt < total_lt; lt += gridDim.x) {
        int tm, tn;
        if (g2_tile(p, lt, tm, tn)) ++nseq;
    }
    const int totalG = nseq * p.ktiles;
    if (totalG == 0) return;

    // sequence position -> tile coordinates (walks the same enumeration; stagings run ahead of the compute position)
    auto seq_tile = [&](int seq, int& m0, int& n0) {
        int k = 0;
        for (int lt = blockIdx.x; lt < total_lt; lt += gridDim.x) {
            int tm, tn;
            if (g2_tile(p, lt, tm, tn)) {
                if (k == seq) { m0 = tm * T2; n0 = tn * T2; return; }
                ++k;
            }
        }
        m0 = 0; n0 = 0;
    };

    // ---- staging: wave w issues row blocks rb = 2w, 2w+1 (8 rows each) of a half; lane -> row rb*8 + lane/8 ----
    int a_trow[2], b_trow[2], s_sw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = (wave * 2 + j) * 8 + (lane >> 3);              // local row in the half, 0..127
        a_trow[j] = (lr >> 6) * 128 + (lr & 63);                      // + h*64: tile row
        b_trow[j] = (lr >> 5) * 64 + (lr & 31);                       // + h*32: tile column (row of W)
        s_sw[j] = ((lane & 7) ^ ((lr >> 1) & 7)) * 8;                 // logical chunk that belongs at this lane's position
    }
    int st_seq = -1, st_m0 = 0, st_n0 = 0;
    // which: 0 = A0, 1 = A1, 2 = B0, 3 = B1
    auto stage = [&](int Gt, int which) -> bool {
        if (Gt >= totalG) return false;
        const int seq = Gt / p.ktiles, kt = Gt - seq * p.ktiles;
        if (seq != st_seq) { st_seq = seq; seq_tile(seq, st_m0, st_n0); }
        char* base = smem + (Gt & 1) * BUF_BYTES + which * HALF_BYTES + wave * 2048;
        if (which < 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int m = st_m0 + a_trow[j] + which * 64;
                if (m >= p.M) m = p.M - 1;                            // tail rows: any valid row, masked at the store
                g2_glds16(p.A + (long)m * p.lda + kt * 64 + s_sw[j], base + j * 1024);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = st_n0 + b_trow[j] + (which - 2) * 32;
                g2_glds16(p.W + (long)n * p.ldw + kt * 64 + s_sw[j], base + j * 1024);
            }
        }
        return true;
    };

    // ---- fragment addressing (local row fixed per lane; chunk = 2*ks + h, swizzled by the row) ----
    int fa_off[2], fa_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = wm * 64 + i * 32 + l31;
        fa_off[i] = lr * 128; fa_sw[i] = (lr >> 1) & 7;
    }
    const int lrb = wn * 32 + l31;
    const int fb_off = lrb * 128, fb_sw = (lrb >> 1) & 7;

    f32x16 acc[2][2][2];               // [qm][qn][m-tile]
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][b][i][e] = 0.f;
    };
    zero_acc();

    // ---- prologue: K tile 0 completely, A0/B0 of K tile 1 ----
    stage(0, 0); stage(0, 2); stage(0, 3); stage(0, 1);
    const bool pa = stage(1, 0), pb = stage(1, 2);
    if (pa && pb) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                       // group 1 runs half a phase behind group 0

    bf16x8 fa[2][4], fb0[4], fb1[4];
    auto read_a = [&](const char* buf, int half) {
        const char* hb = buf + half * HALF_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fa[i][ks] = *reinterpret_cast<const bf16x8*>(hb + fa_off[i] + (((ks * 2 + h) ^ fa_sw[i]) << 4));
    };
    auto read_b = [&](const char* buf, int half, bf16x8 (&fb)[4]) {
        const char* hb = buf + (2 + half) * HALF_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[ks] = *reinterpret_cast<const bf16x8*>(hb + fb_off + (((ks * 2 + h) ^ fb_sw) << 4));
    };
    auto mma = [&](int qm, int qn, bf16x8 (&fb)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[qm][qn][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks], fa[i][ks], acc[qm][qn][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto end_load_section = [&](bool staged) {
        if (staged) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto end_mfma_section = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    float st_sum[8], st_sq[8];
    char* strip = smem + 2 * BUF_BYTES + wave * STRIP_BYTES;
    int c_seq = 0, c_kt = 0, c_m0, c_n0;
    seq_tile(0, c_m0, c_n0);

    for (int G = 0; G < totalG; ++G) {
        const char* buf = smem + (G & 1) * BUF_BYTES;
        bool st;
        // phase 0: quadrant (A0, B0)
        read_a(buf, 0);
        read_b(buf, 0, fb0);
        st = stage(G + 1, 3);
        end_load_section(st);
        mma(0, 0, fb0);
        end_mfma_section();
        // phase 1: quadrant (A0, B1)
        read_b(buf, 1, fb1);
        st = stage(G + 1, 1);
        end_load_section(st);
        mma(0, 1, fb1);
        end_mfma_section();
        // phase 2: quadrant (A1, B1)
        read_a(buf, 1);
        st = stage(G + 2, 0);
        end_load_section(st);
        mma(1, 1, fb1);
        end_mfma_section();
        // phase 3: quadrant (A1, B0) -- B0 still in registers
        st = stage(G + 2, 2);
        end_load_section(st);
        mma(1, 0, fb0);
        end_mfma_section();

        if (++c_kt == p.ktiles) {
            // ---- epilogue of output tile (c_m0, c_n0): 4 x 32-row pieces through the wave's strip ----
            if (EPI == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { st_sum[e] = 0.f; st_sq[e] = 0.f; }
            }
#pragma unroll
            for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // rows wm*128 + qm*64 + i*32 + l31; this lane holds for that row: n = qn*32 + 8g + 4h + e
#pragma unroll
                    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            bf16x4 q;
                            if constexpr (EPI == 1) {
                                const int n_glob = c_n0 + wn * 64 + qn * 32 + 8 * g + 4 * h;
                                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                                if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n_glob);
#pragma unroll
                                for (int e = 0; e < 4; ++e) q[e] = (bf16_t)g2_act(acc[qm][qn][i][4 * g + e] + bv[e], p.act);
                            } else {
                                q = bf16x4{(bf16_t)acc[qm][qn][i][4 * g + 0], (bf16_t)acc[qm][qn][i][4 * g + 1],
                                           (bf16_t)acc[qm][qn][i][4 * g + 2], (bf16_t)acc[qm][qn][i][4 * g + 3]};
                            }
                            const int chunk = qn * 4 + g;
                            *reinterpret_cast<bf16x4*>(strip + l31 * 128 + ((chunk ^ (l31 & 7)) << 4) + h * 8) = q;
                        }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = j * 8 + (lane >> 3), chunk = lane & 7;
                        const int m = c_m0 + wm * 128 + qm * 64 + i * 32 + row, n = c_n0 + wn * 64 + chunk * 8;
                        bf16x8 v = *reinterpret_cast<const bf16x8*>(strip + row * 128 + ((chunk ^ (row & 7)) << 4));
                        if (m < p.M) {
                            if constexpr (EPI == 1) {
                                if (p.R) {
                                    const bf16x8 r = *reinterpret_cast<const bf16x8*>(p.R + (long)m * p.ldr + n);
#pragma unroll
                                    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)r[e]);
                                }
                            } else {
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const float f = (float)v[e];
                                    st_sum[e] += f;
                                    st_sq[e] = fmaf(f, f, st_sq[e]);
                                }
                            }
                            if (EPI != 0 || p.C) *reinterpret_cast<bf16x8*>(p.C + (long)m * p.ldc + n) = v;
                        }
                    }
                }
            if (EPI == 0 && p.stats) {
                // rows of this wave's 128 x 64 piece: reduce the 8 row lanes (fixed order); statistics row = 2*tm + wm
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#pragma unroll
                    for (int o = 8; o <= 32; o <<= 1) {
                        st_sum[e] += __shfl_xor(st_sum[e], o, 64);
                        st_sq[e] += __shfl_xor(st_sq[e], o, 64);
                    }
                }
                if (lane < 8) {
                    const long srow = (long)(c_m0 / T2) * 2 + wm;
                    float* d0 = p.stats + (srow * 2 + 0) * p.N + c_n0 + wn * 64 + lane * 8;
                    float* d1 = p.stats + (srow * 2 + 1) * p.N + c_n0 + wn * 64 + lane * 8;
                    *reinterpret_cast<f32x4*>(d0) = f32x4{st_sum[0], st_sum[1], st_sum[2], st_sum[3]};
                    *reinterpret_cast<f32x4*>(d0 + 4) = f32x4{st_sum[4], st_sum[5], st_sum[6], st_sum[7]};
                    *reinterpret_cast<f32x4*>(d1) = f32x4{st_sq[0], st_sq[1], st_sq[2], st_sq[3]};
                    *reinterpret_cast<f32x4*>(d1 + 4) = f32x4{st_sq[4], st_sq[5], st_sq[6], st_sq[7]};
                }
            }
            zero_acc();
            c_kt = 0;
            ++c_seq;
            if (c_seq < nseq) seq_tile(c_seq, c_m0, c_n0);
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                       // balance group 1's extra barrier
}

int g2_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace

// statistics rows the 256-tile kernel writes for an [M, N] output: 2 per 256-row tile (one per wave row group)
extern "C" int cvcl_gemm256_stats_rows(int M) { return 2 * cvcl_div_up(M, T2); }

// shapes the 256-tile kernel accepts (the dispatcher adds its own policy on top)
extern "C" int cvcl_gemm256_supported(int M, int N, int K, int lda, int ldw, int ldc) {
    return M >= T2 && N % T2 == 0 && K % 64 == 0 && K >= 128 && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0;
}

extern "C" int cvcl_gemm256(int epi, const cvcl_gemm_args* a, void* stream) {
    CVCL_CHECK_ARG(a && a->A && a->W && (a->C || a->stats), "cvcl_gemm256: null operand");
    CVCL_CHECK_ARG(cvcl_gemm256_supported(a->M, a->N, a->K, a->lda, a->ldw, a->ldc) && !a->a_scale && !(a->gather_stride > 1) &&
                       !a->exp_scale && !a->c_scale,
                   "cvcl_gemm256: unsupported shape / options (M %d N %d K %d)", a->M, a->N, a->K);
    CVCL_CHECK_ARG(epi == 0 || epi == 1, "cvcl_gemm256: epilogue %d", epi);
    CVCL_CHECK_ARG(epi == 1 || (!a->bias && !a->R && a->act == CVCL_ACT_NONE), "cvcl_gemm256: EPI 0 takes no bias/act/residual");
    CVCL_CHECK_ARG(epi == 0 || (a->C && !a->stats), "cvcl_gemm256: EPI 1 writes C and takes no statistics");
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    CVCL_CHECK_ARG(al16(a->A) && al16(a->W) && al16(a->C) && al16(a->R) && al16(a->bias) && (!a->R || a->ldr % 8 == 0),
                   "cvcl_gemm256: operands must be 16-byte aligned");
    G2Dev d;
    d.A = (const bf16_t*)a->A; d.W = (const bf16_t*)a->W; d.C = (bf16_t*)a->C; d.R = (const bf16_t*)a->R;
    d.bias = a->bias; d.stats = a->stats;
    d.M = a->M; d.N = a->N; d.K = a->K; d.lda = a->lda; d.ldw = a->ldw; d.ldc = a->ldc; d.ldr = a->ldr; d.act = a->act;
    d.tiles_m = cvcl_div_up(a->M, T2);
    d.tiles_m8 = (d.tiles_m + 7) / 8 * 8;
    d.tiles_n = a->N / T2;
    d.ktiles = a->K / 64;
    if (a->stats) CVCL_CHECK_ARG(a->stats_rows >= 2 * d.tiles_m, "cvcl_gemm256: stats_rows %d < %d", a->stats_rows, 2 * d.tiles_m);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)gemm256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) != hipSuccess) {
            cvcl_set_error("cvcl_gemm256: cannot raise the dynamic LDS limit to %d", G2_LDS);
            return CVCL_ELAUNCH;
        }
        attr_set = true;
    }
    const int total = d.tiles_m8 * d.tiles_n;
    int grid = g2_num_cus();
    if (grid > total) grid = total;
    CvclProfScope prof(stream, CVCL_K_GEMM);
    if (epi == 0) hipLaunchKernelGGL(gemm256_kernel<0>, dim3(grid), dim3(512), G2_LDS, (hipStream_t)stream, d);
    else hipLaunchKernelGGL(gemm256_kernel<1>, dim3(grid), dim3(512), G2_LDS, (hipStream_t)stream, d);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}
