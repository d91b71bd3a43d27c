// FP8 (OCP e4m3) linear layers for the ViT encoder: BASELINE configs[4] ("fp8 MFMA weights/activations").
//
//   C[M,N] (bf16) = act( (A8[M,K] . W8[N,K]^T) * sa[m] * sw[n] + bias[n] ) (+ R[M,N])
//
// A8 / W8 are e4m3 bytes with one fp32 scale per activation row (per token, dynamic: cvcl_quant_rows_fp8 /
// cvcl_layernorm_fp8) and per weight row (per output channel, static: cvcl_quant_rows_fp8 on the weight).  The products
// run on v_mfma_scale_f32_32x32x64_f8f6f4 with all block scales = 2^0: gfx950's non-scaled fp8 MFMA runs at the bf16 rate,
// only the block-scaled K = 64 form doubles it (MI355X_MICROARCH.md), and with unit block scales it is a plain e4m3 x e4m3
// -> fp32 MFMA; the row scales are applied to the fp32 accumulators in the epilogue.
//
// Same skeleton as gemm_glds_kernel (gemm.hip): 128 x 128 tile, 4 waves x (64 x 64), persistent over m tiles, operand tiles
// of 128 rows x 128 BYTES (= 128 k values: twice the K depth of the bf16 tile in the same LDS image) fetched with
// global_load_lds_dwordx4 under the source-side XOR chunk swizzle, two buffers, one barrier per K tile.  A lane's fragment
// for one MFMA is 32 consecutive k bytes of its row (two swizzled 16-byte chunks); A and B use the same (lane, byte) -> k
// assignment, which is all a contraction needs.
#include <algorithm>

#include "cvcl_common.h"
#include "gemm8f_kernel.h"

namespace {

constexpr int F8_BM = 128, F8_BN = 128;
constexpr int F8_OPER = 128 * 128;
constexpr int F8_BUF = 2 * F8_OPER;
constexpr int F8_LDS = 2 * F8_BUF;

typedef int v8i __attribute__((ext_vector_type(8)));

struct F8Dev {
    const unsigned char* A; const unsigned char* W; bf16_t* C; const bf16_t* R;
    const float* sa; const float* sw; const float* bias;
    const unsigned char* a_bs;        // MXA: e8m0 scale per 32-element block of A, tiled [K/128][M][4] (sa unused)
    unsigned char* C8; unsigned char* c_bs;   // MXOUT: e4m3 output [M][ldc8] + e8m0 block scales [M][N/32] instead of (1) / beside (2) bf16 C
    float* row_part;                  // MXOUT == 2: (sum, sum of squares) of the stored row per 64-column strip, [M][N / 64][2]
    int M, N, K, lda, ldw, ldc, ldr, act, num_m_tiles, ldc8;
    int xcd_split;                    // 1: XCD x = blockIdx.x % 8 owns row tiles x, x + 8, ...; 0: one list over all tiles
};

__device__ __forceinline__ void f8_glds16(const unsigned char* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

__device__ inline float f8_gelu(float v) { return gelu_bf16out(v); }      // the bf16 linear epilogue's GELU (cvcl_common.h)

// ---- quantisation: one wave per row; q = e4m3(x / s), s = amax / 448 (s = 1 for an all-zero row) ------------------------
// src: bf16 rows (SRC_F32 = false) or fp32 rows (weights); optional LayerNorm (gamma/beta != NULL) before quantising.
// K % 8 == 0, K <= 4096 (8 chunks of 8 per lane)
// MXOUT: 0 = bf16 C; 1 = MX output only (e4m3 + e8m0 per 32 columns); 2 (round 5, the PRODUCER of a LayerNorm-folded e4m3 linear:
// proj / fc2 of a ViT block, reference vision_transformer_dino_mugs.py:146-147) = the bf16 residual row AND its MX-quantised copy --
// the raw operand of the next qkv / fc1 -- AND the row's (sum, sum of squares) per 64-column strip for cvcl_row_stats_finalize.
template <int ACT, bool MXA, int MXOUT>
__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(F8Dev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int ktiles = p.K / 128, ntn = p.N / F8_BN;
    // tile walk: the workgroups of one XCD (blockIdx.x % 8: its own L2) share a list of output tiles ordered column-fastest
    // over that XCD's row tiles (x, x + 8, ...), so the tiles in flight on an XCD at any time cover ~S / ntn row tiles x all
    // columns -- every A tile is fetched from HBM once per XCD and W stays L2-resident -- while every workgroup slot stays
    // busy to the end whatever ntn is (a (row groups) x (columns) grid left 25% of the slots empty at ntn = 24)
    const int xcd = p.xcd_split ? (blockIdx.x & 7) : 0;
    const int slot = p.xcd_split ? (blockIdx.x >> 3) : blockIdx.x, nslots = p.xcd_split ? (gridDim.x >> 3) : gridDim.x;
    const int my_m_tiles = p.xcd_split ? ((p.num_m_tiles - xcd + 7) >> 3) : p.num_m_tiles;
    const int ntiles = my_m_tiles * ntn;
    auto tile_of = [&](int t, int& mt, int& n0) {
        const int ml = t / ntn;
        n0 = (t - ml * ntn) * F8_BN;
        mt = p.xcd_split ? ml * 8 + xcd : ml;
    };

    // staging: wave w, instruction j covers tile rows (4w + j) * 8 .. + 7; lane -> row + lane / 8, chunk position lane % 8
    int s_row[4];
    long w_off[4], a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        s_row[j] = r;
        (void)c;
    }
    auto set_rows = [&](int t) {
        int mt, n0;
        tile_of(t, mt, n0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = mt * F8_BM + s_row[j];
            if (m >= p.M) m = p.M - 1;
            const int c = (lane & 7) ^ ((s_row[j] >> 1) & 7);
            a_off[j] = (long)m * p.lda + c * 16;
            w_off[j] = (long)(n0 + s_row[j]) * p.ldw + c * 16;
        }
    };
    auto issue = [&](int buf, int kt) {
        char* base = smem + buf * F8_BUF + wave * 4 * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) f8_glds16(p.A + a_off[j] + kt * 128, base + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) f8_glds16(p.W + w_off[j] + kt * 128, base + F8_OPER + j * 1024);
    };

    int fw_off[2], fa_off[2], fw_sw[2], fa_sw[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rw = wn * 64 + q * 32 + l31, ra = wm * 64 + q * 32 + l31;
        fw_off[q] = F8_OPER + rw * 128; fw_sw[q] = (rw >> 1) & 7;
        fa_off[q] = ra * 128;           fa_sw[q] = (ra >> 1) & 7;
    }

    // MXA: the 4 block-scale bytes of a K tile (128 k = 4 blocks of 32) for this lane's rows of the two m tiles, fetched one
    // K tile ahead together with the operand tiles (same vmcnt wait)
    unsigned aw_next[2] = {0x7f7f7f7fu, 0x7f7f7f7fu}, aw[2] = {0x7f7f7f7fu, 0x7f7f7f7fu};
    auto load_scales = [&](int t, int kt) {
        if constexpr (MXA) {
            int mt_tile, n_unused;
            tile_of(t, mt_tile, n_unused);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                int m = mt_tile * F8_BM + wm * 64 + mt * 32 + l31;
                if (m >= p.M) m = p.M - 1;
                aw_next[mt] = *reinterpret_cast<const unsigned*>(p.a_bs + ((long)kt * p.M + m) * 4);      // coalesced over rows
            }
        }
    };

    int l_mt = slot, l_kt = 0, buf = 0;                 // l_mt: the tile (list index) being staged
    bool l_live = l_mt < ntiles;
    if (l_live) { set_rows(l_mt); issue(0, 0); load_scales(l_mt, 0); }

    for (int ct = slot; ct < ntiles; ct += nslots) {
        int cm, n0;
        tile_of(ct, cm, n0);
        const int m0 = cm * F8_BM;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // per-lane epilogue operands of this tile: weight-row scales and bias of output n = n0 + wn*64 + nt*32 + 8g + 4h + e,
        // activation-row scales (column of the MFMA output = this lane's row l31) and the residual rows.  They are fetched in
        // the first K iteration right behind the staging of the next K tile: the K loop hides them, whereas loads issued
        // ahead of the loop hold up its first wait (the operands of K tile 0 were staged one epilogue ago and have long
        // landed) and loads issued in the epilogue queue behind in-flight staging.
        f32x4 sw_r[2][4], bias_r[2][4];
        float sa_r[2] = {1.f, 1.f};
        bf16x8 rpre[8];
        // (the producer mode -- MXOUT == 2 -- sits at the 256-register cap: its residual rows are requested at the head of the
        // epilogue instead of riding through the K loop, which kept a spilled value's reload inside the loop)
        auto fetch_residual = [&]() {
            if (p.R) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int m = m0 + wm * 64 + j * 8 + (lane >> 3);
                    if (m >= p.M) m = p.M - 1;
                    rpre[j] = *reinterpret_cast<const bf16x8*>(p.R + (long)m * p.ldr + n0 + wn * 64 + (lane & 7) * 8);
                }
            }
        };
        auto fetch_epilogue_operands = [&]() {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn * 64 + nt * 32 + 8 * g + 4 * h;
                    sw_r[nt][g] = *reinterpret_cast<const f32x4*>(p.sw + n);
                    bias_r[nt][g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            if constexpr (!MXA) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    int m = m0 + wm * 64 + mt * 32 + l31;
                    if (m >= p.M) m = p.M - 1;
                    sa_r[mt] = p.sa[m];
                }
            }
            if constexpr (MXOUT != 2) fetch_residual();
        };

        for (int kt = 0; kt < ktiles; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (++l_kt == ktiles) {
                l_kt = 0;
                l_mt += nslots;
                l_live = l_mt < ntiles;
                if (l_live) set_rows(l_mt);
            }
            if constexpr (MXA) { aw[0] = aw_next[0]; aw[1] = aw_next[1]; }
            if (l_live) { issue(buf ^ 1, l_kt); load_scales(l_mt, l_kt); }
            if (kt == 0) fetch_epilogue_operands();
            const char* cur = smem + buf * F8_BUF;
#pragma unroll
            for (int s = 0; s < 2; ++s) {                         // two K = 64 steps per 128-byte row
                v8i fw[2], fa[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // the MFMA's k order per lane is interleaved: lane half h holds k 16h..16h+15 and 32+16h..32+16h+15 of the 64,
                    // and scale lane half h covers k 32h..32h+31 (tools/probes/mx_scale_probe_b.hip) -> chunks c0 and c0 + 2
                    const int c0 = 4 * s + h;
                    const u32x4 w0 = *reinterpret_cast<const u32x4*>(cur + fw_off[q] + ((c0 ^ fw_sw[q]) << 4));
                    const u32x4 w1 = *reinterpret_cast<const u32x4*>(cur + fw_off[q] + (((c0 + 2) ^ fw_sw[q]) << 4));
                    const u32x4 a0 = *reinterpret_cast<const u32x4*>(cur + fa_off[q] + ((c0 ^ fa_sw[q]) << 4));
                    const u32x4 a1 = *reinterpret_cast<const u32x4*>(cur + fa_off[q] + (((c0 + 2) ^ fa_sw[q]) << 4));
                    fw[q] = v8i{(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
                    fa[q] = v8i{(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                            fw[nt], fa[mt], acc[nt][mt], 0, 0, 0, 0x7f7f7f7f, 0,
                            MXA ? (int)(aw[mt] >> (8 * (2 * s + h))) : 0x7f7f7f7f);      // byte 0 = scale of k block 2s + h of this lane's row
            }
            buf ^= 1;
        }

        // ---- epilogue: scales, bias, activation -> bf16 through the consumed buffer -> full 128-byte rows (+ residual)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (MXOUT == 2) fetch_residual();
        char* stg = smem + (buf ^ 1) * F8_BUF + wave * 8192;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = mt * 32 + l31;
                    const int chunk = nt * 4 + g;
                    bf16x4 q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = fmaf(acc[nt][mt][4 * g + e] * sa_r[mt], sw_r[nt][g][e], bias_r[nt][g][e]);
                        if (ACT == CVCL_ACT_RELU) v = fmaxf(v, 0.f);
                        if (ACT == CVCL_ACT_GELU) v = f8_gelu(v);
                        q[e] = (bf16_t)v;
                    }
                    *reinterpret_cast<bf16x4*>(stg + row * 128 + ((chunk ^ (row & 7)) << 4) + h * 8) = q;
                }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = j * 8 + (lane >> 3), chunk = lane & 7;
            const int m = m0 + wm * 64 + row, n = n0 + wn * 64 + chunk * 8;
            bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((chunk ^ (row & 7)) << 4));
            if (m < p.M) {
                if (p.R) {
                    const bf16x8 r = rpre[j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)r[e]);
                }
                if constexpr (MXOUT != 1) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p.C + (long)m * p.ldc + n));
                if constexpr (MXOUT == 2) {
                    // (sum, sum of squares) of the STORED values over this wave's 64-column strip of the row (as gemm8w's EPI 2 + LNF):
                    // v_dot2 on the packed pairs (products of bf16 are exact in fp32), then the row's eight lanes by DPP
                    const u32x4 w4 = __builtin_bit_cast(u32x4, v);
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1 = dot2c_bf16(s1, w4[e], 0x3f803f80u);             // (1.0, 1.0)
                        s2 = dot2c_bf16(s2, w4[e], w4[e]);
                    }
                    s1 += dpp_quad_f32<0xB1>(s1); s2 += dpp_quad_f32<0xB1>(s2);          // lane ^ 1
                    s1 += dpp_quad_f32<0x4E>(s1); s2 += dpp_quad_f32<0x4E>(s2);          // lane ^ 2
                    s1 += dpp_quad_f32<0x141>(s1); s2 += dpp_quad_f32<0x141>(s2);        // row_half_mirror: the other quad
                    if (chunk == 0)
                        *reinterpret_cast<f32x2*>(p.row_part + ((long)m * (p.N >> 6) + ((n0 >> 6) + wn)) * 2) = f32x2{s1, s2};
                }
            }
            if constexpr (MXOUT != 0) {
                // e4m3 + one e8m0 scale per 32 columns of the row: the block = 4 adjacent lanes (chunks 4b .. 4b+3).  On the packed
                // bf16 words (cvcl_common.h): |x| maximum as 15-bit integers, then v_cvt_scalef32_pk_fp8_bf16 by the block scale
                const u32x4 vw = __builtin_bit_cast(u32x4, v);
                unsigned mb = bf16x8_absmax_bits(vw);
                mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xB1, 0xf, 0xf, true));
                mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x4E, 0xf, 0xf, true));
                const unsigned sb = mx_scale_byte(__uint_as_float(mb << 16));
                if (m < p.M) {
#ifdef CVCL_CVT_SCALE_MUL                                                    // (probe build: the instruction multiplies instead of dividing)
                    const u32x2 w = bf16x8_to_fp8_scaled(vw, mx_inv_scale(sb));
#else
                    const u32x2 w = bf16x8_to_fp8_scaled(vw, __uint_as_float(sb << 23));
#endif
                    __builtin_nontemporal_store(w, reinterpret_cast<u32x2*>(p.C8 + (long)m * p.ldc8 + n));
                    if ((chunk & 3) == 0) p.c_bs[((long)(n >> 7) * p.M + m) * 4 + ((n >> 5) & 3)] = (unsigned char)sb;
                }
            }
        }
    }
}

// LPR lanes per row (32 for K <= 1024: a 768-wide row is 96 chunks = 3 per lane with no idle lanes; 64 otherwise)
// reductions over LPR (32 | 64) consecutive lanes, result in every lane: the four steps inside a 16-lane row are DPP moves
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: VALU-rate, no LDS round trip); only the row-crossing steps are
// ds_bpermute shuffles -- 1 (2) instead of 5 (6) per reduction, three reductions per row on the LayerNorm + quantise path
__device__ inline float dpp_f(float v, int ctrl_sel) {
    const int x = __builtin_bit_cast(int, v);
    int r;
    if (ctrl_sel == 0) r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);          // quad_perm [1,0,3,2]
    else if (ctrl_sel == 1) r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    else if (ctrl_sel == 2) r = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);    // row_half_mirror
    else r = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true);                       // row_mirror
    return __builtin_bit_cast(float, r);
}
template <int LPR> __device__ inline float lanes_sum(float v) {
    v += dpp_f(v, 0); v += dpp_f(v, 1); v += dpp_f(v, 2); v += dpp_f(v, 3);
#pragma unroll
    for (int o = 16; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int LPR> __device__ inline float lanes_max(float v) {
    v = fmaxf(v, dpp_f(v, 0)); v = fmaxf(v, dpp_f(v, 1)); v = fmaxf(v, dpp_f(v, 2)); v = fmaxf(v, dpp_f(v, 3));
#pragma unroll
    for (int o = 16; o < LPR; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// EXACT: K == 8 LPR NCH, every chunk exists -- no branch around a load (each branch costs a serializing s_waitcnt vmcnt(0))
template <bool SRC_F32, int LPR, int NCH, bool EXACT>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const void* __restrict__ xin, long x_row_stride, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, unsigned char* __restrict__ q,
                                                             float* __restrict__ scale, long rows, int K) {
    const long row = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    const int lane = threadIdx.x % LPR;
    if (row >= rows) return;
    const int nch = K >> 3;
    // NCH chunks (of 8 elements) per lane held in registers: 3 x 32 lanes = a 768-wide row (ViT-B) with no idle slot and few
    // enough registers for 8 waves per SIMD -- the kernel is latency-bound, occupancy is what hides the row load
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + LPR * i;
        if (EXACT || c < nch) {
            if (SRC_F32) {
                const float* xr = (const float*)xin + row * x_row_stride + c * 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(xr), b = *reinterpret_cast<const f32x4*>(xr + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = b[e]; }
            } else {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>((const bf16_t*)xin + row * x_row_stride + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = (float)a[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        }
    }
    if (gamma) {                                         // nn.LayerNorm first (vit blocks: norm1 / norm2 feed the fp8 GEMMs)
        const float mean = lanes_sum<LPR>(s) / (float)K;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (EXACT || lane + LPR * i < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float c = v[i][e] - mean; v[i][e] = c; qq = fmaf(c, c, qq); }   // keep the centred value
            }
        const float rstd = 1.f / sqrtf(lanes_sum<LPR>(qq) / (float)K + eps);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + LPR * i;
            if (EXACT || c < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = fmaf(v[i][e], rstd * gamma[c * 8 + e], beta[c * 8 + e]);
            }
        }
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (EXACT || lane + LPR * i < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
        }
    amax = lanes_max<LPR>(amax);
    const float sc = amax > 0.f ? amax / 448.f : 1.f;
    if (lane == 0) scale[row] = sc;
    // y / sc for every element of the row, correctly rounded (the oracle's y / s), without a division per element: with
    // r = RN(1 / sc) computed once, q0 = RN(y r), rem = y - sc q0 (exact in an fma), q = RN(q0 + rem r) is RN(y / sc)
    // (Markstein's quotient refinement; no overflow / underflow here: |y / sc| <= 448, sc is a normal number)
    const float rinv = 1.0f / sc;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + LPR * i;
        if (EXACT || c < nch) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float q0 = v[i][e] * rinv;
                const float qd = fmaf(fmaf(-sc, q0, v[i][e]), rinv, q0);
                t[e] = fminf(fmaxf(qd, -448.f), 448.f);
            }
            u32x2 w = {pack4_fp8(t[0], t[1], t[2], t[3]), pack4_fp8(t[4], t[5], t[6], t[7])};
            *reinterpret_cast<u32x2*>(q + row * K + c * 8) = w;
        }
    }
}

// MX quantisation of bf16 rows (round 5: the raw residual rows that enter the first LayerNorm-folded qkv of a ViT -- every later
// block gets them from the proj / fc2 epilogue): e4m3 with one e8m0 scale per 32 elements, the block-scale bytes tiled
// [K / 128][rows][4] as the MX-input GEMMs read them.  Same helpers as the GEMM epilogues: bit-identical to their quantiser.
// One 16-byte chunk per thread, four adjacent lanes = one block.
__global__ __launch_bounds__(256) void quant_rows_mx_kernel(const bf16_t* __restrict__ x, long x_row_stride, unsigned char* __restrict__ q,
                                                            unsigned char* __restrict__ bs, long rows, int K) {
    const int cpr = K >> 3;                                  // chunks per row (a multiple of 16)
    const long total = rows * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / cpr;
        const int c = (int)(i - row * cpr);
        const u32x4 vw = *reinterpret_cast<const u32x4*>(x + row * x_row_stride + c * 8);
        unsigned mb = bf16x8_absmax_bits(vw);
        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xB1, 0xf, 0xf, true));
        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x4E, 0xf, 0xf, true));
        const unsigned sb = mx_scale_byte(__uint_as_float(mb << 16));
        const u32x2 w = bf16x8_to_fp8_scaled(vw, __uint_as_float(sb << 23));
        *reinterpret_cast<u32x2*>(q + row * K + c * 8) = w;
        if ((c & 3) == 0) bs[((long)(c >> 4) * rows + row) * 4 + ((c >> 2) & 3)] = (unsigned char)sb;
    }
}

int f8_num_cus() {
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

extern "C" int cvcl_quant_rows_fp8(int src_dtype, const void* x, long x_row_stride, const float* ln_gamma, const float* ln_beta,
                                   float ln_eps, void* q, float* scale, long rows, int K, void* stream) {
    CVCL_CHECK_ARG(x && q && scale && rows > 0 && K > 0 && K % 8 == 0 && K <= 4096 && x_row_stride % 8 == 0 &&
                       ((uintptr_t)x & 15) == 0 && ((uintptr_t)q & 7) == 0 && (!ln_gamma == !ln_beta),
                   "cvcl_quant_rows_fp8: bad args (K %d)", K);
    CvclProfScope prof(stream, CVCL_K_LAYERNORM);
    const bool narrow = K <= 1024;                      // 32 lanes per row
    dim3 grid(cvcl_div_up(rows, narrow ? 8 : 4));
    hipStream_t st = (hipStream_t)stream;
    unsigned char* qq = (unsigned char*)q;
    if (src_dtype == CVCL_F32) {
        if (narrow) hipLaunchKernelGGL((quant_rows_fp8_kernel<true, 32, 4, false>), grid, dim3(256), 0, st, x, x_row_stride, ln_gamma, ln_beta, ln_eps, qq, scale, rows, K);
        else hipLaunchKernelGGL((quant_rows_fp8_kernel<true, 64, 8, false>), grid, dim3(256), 0, st, x, x_row_stride, ln_gamma, ln_beta, ln_eps, qq, scale, rows, K);
    } else {
        if (K == 768) hipLaunchKernelGGL((quant_rows_fp8_kernel<false, 32, 3, true>), grid, dim3(256), 0, st, x, x_row_stride, ln_gamma, ln_beta, ln_eps, qq, scale, rows, K);
        else if (K < 768) hipLaunchKernelGGL((quant_rows_fp8_kernel<false, 32, 3, false>), grid, dim3(256), 0, st, x, x_row_stride, ln_gamma, ln_beta, ln_eps, qq, scale, rows, K);
        else if (narrow) hipLaunchKernelGGL((quant_rows_fp8_kernel<false, 32, 4, false>), grid, dim3(256), 0, st, x, x_row_stride, ln_gamma, ln_beta, ln_eps, qq, scale, rows, K);
        else hipLaunchKernelGGL((quant_rows_fp8_kernel<false, 64, 8, false>), grid, dim3(256), 0, st, x, x_row_stride, ln_gamma, ln_beta, ln_eps, qq, scale, rows, K);
    }
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

namespace {
// ---- the 8-wave 256 (192) x 256 kernel (gemm8f_kernel.h) for the large ViT shapes ----
template <int MT, int KIND, int ACT>
int launch_8f(const g8f::Dev& d, int grid, hipStream_t st) {
    static CvclLdsAttr attr;
    if (!attr.ready()) {
        if (hipFuncSetAttribute((const void*)g8f::gemm8f_kernel<MT, KIND, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, g8f::LDS_BYTES) != hipSuccess) {
            cvcl_set_error("cvcl_gemm_fp8: cannot raise the dynamic LDS limit to %d", g8f::LDS_BYTES);
            return CVCL_ELAUNCH;
        }
        attr.mark();
    }
    hipLaunchKernelGGL((g8f::gemm8f_kernel<MT, KIND, ACT>), dim3(grid), dim3(512), g8f::LDS_BYTES, st, d);
    return CVCL_OK;
}
// -1: not for this kernel; else the kind (0 per-row scales -> bf16, 1 MX input + residual -> bf16, 2 per-row scales -> MX output)
int pick_8f(const float* a_scale, const void* a_bs, const void* C, const void* c8, int act, const void* R, int M, int N, int K, int lda,
            int ldw) {
    if (N % 256 || K % 128 || K < 256 || lda % 16 || ldw % 16) return -1;
    if ((long)cvcl_div_up(M, 256) * (N / 256) < 96 || (long)M * lda >= (1L << 31) || (long)N * ldw >= (1L << 31)) return -1;
    if (a_bs) return (C && R && act == CVCL_ACT_NONE) ? 1 : -1;
    if (c8) return (!R && (act == CVCL_ACT_NONE || act == CVCL_ACT_GELU)) ? 2 : -1;
    return (C && !R && (act == CVCL_ACT_NONE || act == CVCL_ACT_GELU)) ? 0 : -1;
}

template <int ACT, bool MXA, int MXOUT>
int launch_fp8(const F8Dev& d, dim3 grid, hipStream_t st) {
    static CvclLdsAttr attr_set;
    if (!attr_set.ready()) {
        if (hipFuncSetAttribute((const void*)gemm_fp8_kernel<ACT, MXA, MXOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS) != hipSuccess) {
            cvcl_set_error("cvcl_gemm_fp8: cannot raise the dynamic LDS limit");
            return CVCL_ELAUNCH;
        }
        attr_set.mark();
    }
    hipLaunchKernelGGL((gemm_fp8_kernel<ACT, MXA, MXOUT>), grid, dim3(256), F8_LDS, st, d);
    return CVCL_OK;
}
}  // namespace

// a_scale: per-row fp32 scales [M], or NULL with a_block_scales (e8m0, MX) tiled [K/128][M][4]: the four block scales of one
// row's 128-wide K tile are one word, rows adjacent, so a wave's 32 rows read them coalesced.  C: bf16 [M][ldc], or NULL with
// c8 [M][ldc8] e4m3 + c_block_scales [N/128][M][4] (MX output for the next fp8 GEMM; no residual in that mode).
// Round 5: ln_stats / ln_colsum = the CONSUMER of a folded LayerNorm (MX input = the raw residual rows; the 8-wave kernel only:
// cvcl_gemm_fp8_ln_supported); row_part = the PRODUCER (bf16 C + residual AND the MX copy c8 / c_block_scales of the stored rows
// AND their strip sums; the 128 x 128 kernel).
extern "C" int cvcl_gemm_fp8_ln_supported(int M, int N, int K) {
    if (N % 256 || K % 128 || K < 256 || (long)cvcl_div_up(M, 256) * (N / 256) < 96 || (long)M * K >= (1L << 31) || (long)N * K >= (1L << 31)) return 0;
    const int share = cvcl_gemm_cu_share();
    const int cus = share > 0 && share < f8_num_cus() ? share : f8_num_cus();
    const long total = (long)cvcl_div_up(M, 256) * (N / 256);
    const long g = total < cus ? ((total + 7) & ~7L) : (cus & ~7);
    return g > 0;
}

extern "C" int cvcl_gemm_fp8_ex(const cvcl_gemm_fp8_args* x, void* stream) {
    CVCL_CHECK_ARG(x, "cvcl_gemm_fp8: null arguments");
    const void* A8 = x->A8; const float* a_scale = x->a_scale; const void* a_block_scales = x->a_block_scales; const int lda = x->lda;
    const void* W8 = x->W8; const float* w_scale = x->w_scale; const int ldw = x->ldw;
    void* C = x->C; const int ldc = x->ldc; void* c8 = x->c8; void* c_block_scales = x->c_block_scales; const int ldc8 = x->ldc8;
    const float* bias = x->bias; const int act = x->act; const void* R = x->R; const int ldr = x->ldr;
    const int M = x->M, N = x->N, K = x->K;
    const bool consumer = x->ln_stats != nullptr, producer = x->row_part != nullptr;
    CVCL_CHECK_ARG(A8 && W8 && w_scale && M > 0 && N > 0 && K > 0 && (!a_scale != !a_block_scales) && (producer || (!C != !c8)) &&
                       (!c8 == !c_block_scales), "cvcl_gemm_fp8: null / inconsistent operands");
    CVCL_CHECK_ARG(!consumer || (a_block_scales && x->ln_colsum && bias && !R && !producer && ((uintptr_t)x->ln_stats & 15) == 0 &&
                                 ((uintptr_t)x->ln_colsum & 15) == 0 && cvcl_gemm_fp8_ln_supported(M, N, K)),
                   "cvcl_gemm_fp8: ln_stats needs MX input, ln_colsum, the folded bias, no residual and a shape the 8-wave kernel takes");
    CVCL_CHECK_ARG(!x->ln_colsum || consumer, "cvcl_gemm_fp8: ln_colsum without ln_stats");
    CVCL_CHECK_ARG(!producer || (C && c8 && c_block_scales && R && a_block_scales && act == CVCL_ACT_NONE && N % 64 == 0 &&
                                 ((uintptr_t)x->row_part & 7) == 0),
                   "cvcl_gemm_fp8: row_part goes with MX input, bias + residual, and both outputs (C and c8 / c_block_scales)");
    CVCL_CHECK_ARG(K % 128 == 0 && N % 128 == 0 && lda % 16 == 0 && ldw % 16 == 0 && (!C || ldc % 8 == 0) && (!c8 || ldc8 % 8 == 0) &&
                       (!R || ldr % 8 == 0),
                   "cvcl_gemm_fp8: needs K %% 128 == 0, N %% 128 == 0 and 16-byte aligned rows (M %d N %d K %d)", M, N, K);
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    CVCL_CHECK_ARG(al16(A8) && al16(W8) && al16(C) && al16(R) && al16(bias) && al16(w_scale) && (((uintptr_t)c8 & 7) == 0) &&
                       (((uintptr_t)a_block_scales & 3) == 0), "cvcl_gemm_fp8: operands must be 16-byte aligned");
    CVCL_CHECK_ARG(act == CVCL_ACT_NONE || act == CVCL_ACT_RELU || act == CVCL_ACT_GELU, "cvcl_gemm_fp8: activation %d", act);
    CVCL_CHECK_ARG(!c8 || !R || producer, "cvcl_gemm_fp8: the MX output mode takes no residual");
    hipStream_t st = (hipStream_t)stream;
    // Large shapes: the 8-wave kernel (gemm8f_kernel.h)
    int kind = (M >= 4 && !producer) ? pick_8f(a_scale, a_block_scales, C, c8, act, R, M, N, K, lda, ldw) : -1;
    if (consumer) kind = c8 ? 4 : 3;                         // (shape vetted by cvcl_gemm_fp8_ln_supported above)
    if (kind >= 0) {
        bool use8f = true;
        g8f::Dev g;
        g.A = (const unsigned char*)A8; g.W = (const unsigned char*)W8; g.C = (bf16_t*)C; g.R = (const bf16_t*)R;
        g.sa = a_scale; g.sw = w_scale; g.bias = bias; g.a_bs = (const unsigned char*)a_block_scales;
        g.C8 = (unsigned char*)c8; g.c_bs = (unsigned char*)c_block_scales;
        g.ln_stats = x->ln_stats; g.ln_colsum = x->ln_colsum;
        g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr; g.ldc8 = ldc8; g.act = act;
        g.ncol = N / 256;
        const int share = cvcl_gemm_cu_share();
        const int cus = share > 0 && share < f8_num_cus() ? share : f8_num_cus();
        int bm = 256;
        long best = -1;
        // rounds x tile height decides, with 7 % in favour of 256 rows (8 MFMAs per 12 fragment reads against 6 per 10: measured on
        // fc1 of ViT-B, 10 rounds of 256 rows beat 13 of 192 although 2560 > 2496 row units -- C5 step 8.52 -> 8.41 ms)
        for (int hgt : {256, 192}) {
            const long total = (long)cvcl_div_up(M, hgt) * g.ncol;
            const long gg = total < cus ? ((total + 7) & ~7L) : (cus & ~7);
            const long cost = ((total + gg - 1) / gg) * hgt * (hgt == 192 ? 107 : 100);
            if (best < 0 || cost < best) { best = cost; bm = hgt; }
        }
        { static const int force_bm = cvcl_lab_int("CVCL_F8_BM", 0); if (force_bm == 256 || force_bm == 192) bm = force_bm; }
        g.tiles_m = cvcl_div_up(M, bm);
        const long total = (long)g.tiles_m * g.ncol;
        const int grid8 = total < cus ? (int)((total + 7) & ~7L) : (cus & ~7);
        // the last round of 256-wide tiles must be reasonably full: at N = 768 (proj / fc2 of ViT-B, 2.3 rounds of 256-row tiles
        // = 77 % of the slots) the 128 x 128 kernel below (two workgroups per CU, 4.6 rounds = 92 %) is the faster one
        use8f = consumer || (double)total / ((double)((total + grid8 - 1) / grid8) * grid8) >= 0.85;
      if (use8f) {
        CvclProfScope prof(stream, CVCL_K_GEMM);
        int rc;
        if (kind >= 3) {
            const bool gelu = act == CVCL_ACT_GELU;
            CVCL_CHECK_ARG(act == CVCL_ACT_NONE || gelu, "cvcl_gemm_fp8: the folded kinds take no ReLU");
            if (bm == 256)
                rc = kind == 3 ? (gelu ? launch_8f<4, 3, CVCL_ACT_GELU>(g, grid8, st) : launch_8f<4, 3, CVCL_ACT_NONE>(g, grid8, st))
                               : (gelu ? launch_8f<4, 4, CVCL_ACT_GELU>(g, grid8, st) : launch_8f<4, 4, CVCL_ACT_NONE>(g, grid8, st));
            else
                rc = kind == 3 ? (gelu ? launch_8f<3, 3, CVCL_ACT_GELU>(g, grid8, st) : launch_8f<3, 3, CVCL_ACT_NONE>(g, grid8, st))
                               : (gelu ? launch_8f<3, 4, CVCL_ACT_GELU>(g, grid8, st) : launch_8f<3, 4, CVCL_ACT_NONE>(g, grid8, st));
        } else if (bm == 256)
            rc = kind == 1 ? launch_8f<4, 1, CVCL_ACT_NONE>(g, grid8, st)
               : kind == 2 ? (act == CVCL_ACT_GELU ? launch_8f<4, 2, CVCL_ACT_GELU>(g, grid8, st) : launch_8f<4, 2, CVCL_ACT_NONE>(g, grid8, st))
                           : (act == CVCL_ACT_GELU ? launch_8f<4, 0, CVCL_ACT_GELU>(g, grid8, st) : launch_8f<4, 0, CVCL_ACT_NONE>(g, grid8, st));
        else
            rc = kind == 1 ? launch_8f<3, 1, CVCL_ACT_NONE>(g, grid8, st)
               : kind == 2 ? (act == CVCL_ACT_GELU ? launch_8f<3, 2, CVCL_ACT_GELU>(g, grid8, st) : launch_8f<3, 2, CVCL_ACT_NONE>(g, grid8, st))
                           : (act == CVCL_ACT_GELU ? launch_8f<3, 0, CVCL_ACT_GELU>(g, grid8, st) : launch_8f<3, 0, CVCL_ACT_NONE>(g, grid8, st));
        if (rc) return rc;
        CVCL_LAUNCH_CHECK();
        return CVCL_OK;
      }
    }
    F8Dev d;
    d.A = (const unsigned char*)A8; d.W = (const unsigned char*)W8; d.C = (bf16_t*)C; d.R = (const bf16_t*)R;
    d.sa = a_scale; d.sw = w_scale; d.bias = bias;
    d.a_bs = (const unsigned char*)a_block_scales; d.C8 = (unsigned char*)c8; d.c_bs = (unsigned char*)c_block_scales;
    d.row_part = x->row_part;
    d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldw = ldw; d.ldc = ldc; d.ldr = ldr; d.act = act; d.ldc8 = ldc8;
    d.num_m_tiles = cvcl_div_up(M, F8_BM);
    const int ntn = N / F8_BN;
    const long total_tiles = (long)d.num_m_tiles * ntn;
    const int share2 = cvcl_gemm_cu_share();
    int g = 2 * (share2 > 0 && share2 < f8_num_cus() ? share2 : f8_num_cus());      // persistent: two workgroups per CU (of the caller's share)
    d.xcd_split = d.num_m_tiles >= 16 ? 1 : 0;
    if (d.xcd_split) {
        const long per_xcd = (long)cvcl_div_up(d.num_m_tiles, 8) * ntn;      // the longest per-XCD list
        if (g / 8 > per_xcd) g = (int)per_xcd * 8;
        g &= ~7;
    } else if (g > total_tiles) {
        g = (int)total_tiles;
    }
    dim3 grid(g);
    CvclProfScope prof(stream, CVCL_K_GEMM);
    const bool mxa = a_block_scales != nullptr, mxo = c8 != nullptr;
    int rc;
    if (producer) {
        rc = launch_fp8<CVCL_ACT_NONE, true, 2>(d, grid, st);
    } else if (mxo) {
        CVCL_CHECK_ARG(!mxa, "cvcl_gemm_fp8: MX input together with MX output is not instantiated");
        rc = act == CVCL_ACT_GELU ? launch_fp8<CVCL_ACT_GELU, false, 1>(d, grid, st)
           : act == CVCL_ACT_RELU ? launch_fp8<CVCL_ACT_RELU, false, 1>(d, grid, st) : launch_fp8<CVCL_ACT_NONE, false, 1>(d, grid, st);
    } else if (mxa) {
        CVCL_CHECK_ARG(act == CVCL_ACT_NONE, "cvcl_gemm_fp8: MX input is instantiated without activation only");
        rc = launch_fp8<CVCL_ACT_NONE, true, 0>(d, grid, st);
    } else {
        rc = act == CVCL_ACT_GELU ? launch_fp8<CVCL_ACT_GELU, false, 0>(d, grid, st)
           : act == CVCL_ACT_RELU ? launch_fp8<CVCL_ACT_RELU, false, 0>(d, grid, st) : launch_fp8<CVCL_ACT_NONE, false, 0>(d, grid, st);
    }
    if (rc) return rc;
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_gemm_fp8_mx(const void* A8, const float* a_scale, const void* a_block_scales, int lda, const void* W8,
                                const float* w_scale, int ldw, void* C, int ldc, void* c8, void* c_block_scales, int ldc8,
                                const float* bias, int act, const void* R, int ldr, int M, int N, int K, void* stream) {
    cvcl_gemm_fp8_args x = {};
    x.A8 = A8; x.a_scale = a_scale; x.a_block_scales = a_block_scales; x.lda = lda; x.W8 = W8; x.w_scale = w_scale; x.ldw = ldw;
    x.C = C; x.ldc = ldc; x.c8 = c8; x.c_block_scales = c_block_scales; x.ldc8 = ldc8; x.bias = bias; x.act = act; x.R = R; x.ldr = ldr;
    x.M = M; x.N = N; x.K = K;
    return cvcl_gemm_fp8_ex(&x, stream);
}

extern "C" int cvcl_quant_rows_mx(const void* x, long x_row_stride, void* q, void* block_scales, long rows, int K, void* stream) {
    CVCL_CHECK_ARG(x && q && block_scales && rows > 0 && K > 0 && K % 128 == 0 && x_row_stride % 8 == 0 && ((uintptr_t)x & 15) == 0 &&
                       ((uintptr_t)q & 7) == 0, "cvcl_quant_rows_mx: bad args (K %d)", K);
    CvclProfScope prof(stream, CVCL_K_LAYERNORM);
    const long total = rows * (K >> 3);
    const int grid = (int)std::min<long>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(quant_rows_mx_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_row_stride, (unsigned char*)q,
                       (unsigned char*)block_scales, rows, K);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_gemm_fp8(const void* A8, const float* a_scale, int lda, const void* W8, const float* w_scale, int ldw, void* C, int ldc,
                             const float* bias, int act, const void* R, int ldr, int M, int N, int K, void* stream) {
    return cvcl_gemm_fp8_mx(A8, a_scale, nullptr, lda, W8, w_scale, ldw, C, ldc, nullptr, nullptr, 0, bias, act, R, ldr, M, N, K, stream);
}
