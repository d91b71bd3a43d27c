// Backward of the ViT self-attention (reference: vision_transformer_dino_mugs.py:106-130 -- softmax(q k^T * scale) v per
// head; autograd's backward of it when the ViT is fine-tuned), bf16 operands, fp32 accumulation, head_dim 64.
//
//   P = softmax(S), S = scale * Q K^T          dV = P^T dO          dP = dO V^T
//   dS = P o (dP - D),  D_i = sum_d dO_id O_id          dQ = scale * dS K          dK = scale * dS^T Q
//
// Two kernels, both shaped like the forward kernel (vit.hip: one workgroup per (image, head), 32 x 32 MFMA tiles, the
// probabilities rebuilt from the saved log-sum-exp and kept in registers as the MFMA B operand):
//   attention_bwd_dq_kernel    a wave OWNS 32 queries (Q, dO fragments in registers, LSE and D as per-lane scalars) and
//                              streams over the key tiles in LDS:  S^T = K Q^T,  dP^T = V dO^T,  dQ^T += K^T dS^T
//   attention_bwd_dkv_kernel   a wave OWNS 32 keys (K, V fragments in registers) and streams over the query tiles in LDS:
//                              S = Q K^T,  dP = dO V^T,  dV^T += dO^T P,  dK^T += Q^T dS   (LSE / D per query from LDS)
// S and dP are computed in both (7 matrix products instead of 5): no atomics, no cross-wave reduction, deterministic.
// Each streamed operand sits in LDS once, row-major at pitch 144 B: the ds_read_b128 fragments are conflict-free, the transposed
// fragments (ds_read_b64_tr_b16 on the same rows) take 2-way bank conflicts -- a second copy at the conflict-free pitch of 192 B
// (as the forward kernel keeps for V) doubled the LDS to one workgroup per CU and measured slower (580 vs 335 us per ViT-B/16 layer at B = 256).
#include "cvcl_common.h"

namespace {

constexpr int AB_RP = 144;             // row-major pitch (128 + 16)
constexpr int AB_TP = 144;             // transposed reads use the same copy (2-way bank conflicts on them, half the LDS: two workgroups per CU)
constexpr int AB_TPAD_MAX_DKV = 288;   // two staged operands per kernel: 288 * 288 B = 81 KB (+ LSE, D)

typedef __bf16 ab_tr4 __attribute__((__vector_size__(4 * sizeof(__bf16))));
__device__ inline bf16x4 ab_tr_read(const char* p) {
    auto lp = reinterpret_cast<__attribute__((address_space(3))) ab_tr4*>((__attribute__((address_space(3))) char*)(p));
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16(lp));
}

// qkv: [B][T][3][heads][64] bf16;  o, d_o: [B][T][heads * 64] bf16;  lse: [B][heads][T] fp32 (log2 units);
// d_qkv: [B][T][3][heads][64] bf16 -- this kernel writes the q third.
__global__ __launch_bounds__(256, 2) void attention_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                               const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                               bf16_t* __restrict__ d_qkv, int B, int Tn, int heads, float scale,
                                                               int NT) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Tpad = 32 * NT;
    char* sK = smem;                       // [Tpad][AB_RP]  K rows (A operand of S^T)
    char* sV = sK + Tpad * AB_RP;          // [Tpad][AB_RP]  V rows (A operand of dP^T)
    char* sKt = sK;                        // the same rows, read transposed (A operand of dQ^T)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int hh = blockIdx.x % heads, b = blockIdx.x / heads;
    const int D = heads * 64;
    const bf16_t* base = qkv + (long)b * Tn * 3 * D;

    {   // all of a thread's staging loads are issued before the first LDS write (the plain loop paid a round trip per iteration)
        constexpr int IT = (AB_TPAD_MAX_DKV * 8 + 255) / 256;
        u32x4 kv[IT], vv[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = tid + it * 256;
            const int jc = min(i >> 3, Tn - 1), c = i & 7;
            kv[it] = *reinterpret_cast<const u32x4*>(base + (long)jc * 3 * D + D + hh * 64 + c * 8);
            vv[it] = *reinterpret_cast<const u32x4*>(base + (long)jc * 3 * D + 2 * D + hh * 64 + c * 8);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = tid + it * 256;
            const int j = i >> 3, c = i & 7;
            if (i < Tpad * 8) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4*>(sK + j * AB_RP + c * 16) = j < Tn ? kv[it] : z;
                *reinterpret_cast<u32x4*>(sV + j * AB_RP + c * 16) = j < Tn ? vv[it] : z;
            }
        }
    }
    __syncthreads();

    const float scale2 = scale * 1.4426950408889634f;
    const int nqt = (Tn + 31) / 32;
    const int l15 = lane & 15;
    const int t_lane_off = (4 * h + (l15 >> 2)) * AB_TP + (((lane >> 4) & 1) * 16 + (l15 & 3) * 4) * 2;

    for (int qt = wave; qt < nqt; qt += 4) {
        const int q0 = qt * 32;
        const int qrow = min(q0 + l31, Tn - 1);
        const bool q_ok = q0 + l31 < Tn;
        bf16x8 qf[4], dof[4];
        float dpart = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(base + (long)qrow * 3 * D + hh * 64 + ks * 16 + h * 8);
            const long orow = ((long)b * Tn + qrow) * D + hh * 64 + ks * 16 + h * 8;
            dof[ks] = *reinterpret_cast<const bf16x8*>(d_o + orow);
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(o + orow);
#pragma unroll
            for (int e = 0; e < 8; ++e) dpart = fmaf((float)dof[ks][e], (float)of[e], dpart);
        }
        const float Dq = dpart + __shfl_xor(dpart, 32, 64);                    // D of this lane's query
        const float lse_q = lse[((long)b * heads + hh) * Tn + qrow];

        f32x16 dq[2];                                                          // dQ^T: rows d = 32 dt + 8b + 4h + c, column = query
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;

        for (int t = 0; t < NT; ++t) {
            f32x16 acc, accp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (t * 32 + l31) * AB_RP + ks * 32 + h * 16);
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + (t * 32 + l31) * AB_RP + ks * 32 + h * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], acc, 0, 0, 0);       // S^T  [key][query]
                accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], accp, 0, 0, 0);    // dP^T [key][query]
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                const float p = key < Tn ? __builtin_amdgcn_exp2f(acc[r] * scale2 - lse_q) : 0.f;
                acc[r] = p * (accp[r] - Dq);                                    // dS^T (the factor `scale` is applied once, at the end)
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 sf;
#pragma unroll
                for (int e = 0; e < 8; ++e) sf[e] = (bf16_t)acc[8 * u + e];
                const char* kb = sKt + (t * 32 + 16 * u) * AB_TP + t_lane_off;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x4 k0 = ab_tr_read(kb + dt * 64);
                    const bf16x4 k1 = ab_tr_read(kb + 8 * AB_TP + dt * 64);
                    const bf16x8 kt = __builtin_shufflevector(k0, k1, 0, 1, 2, 3, 4, 5, 6, 7);
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, sf, dq[dt], 0, 0, 0);
                }
            }
        }
        if (q_ok) {
            bf16_t* drow = d_qkv + ((long)b * Tn + q0 + l31) * 3 * D + hh * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const bf16x4 v = {(bf16_t)(dq[dt][4 * bb + 0] * scale), (bf16_t)(dq[dt][4 * bb + 1] * scale),
                                      (bf16_t)(dq[dt][4 * bb + 2] * scale), (bf16_t)(dq[dt][4 * bb + 3] * scale)};
                    *reinterpret_cast<bf16x4*>(drow + dt * 32 + 8 * bb + 4 * h) = v;
                }
        }
    }
}

// writes the k and v thirds of d_qkv
__global__ __launch_bounds__(256, 2) void attention_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                                const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                                bf16_t* __restrict__ d_qkv, int B, int Tn, int heads, float scale,
                                                                int NT) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Tpad = 32 * NT;
    char* sQ = smem;                       // [Tpad][AB_RP]  Q rows (A operand of S)
    char* sO = sQ + Tpad * AB_RP;          // [Tpad][AB_RP]  dO rows (A operand of dP)
    char* sQt = sQ;                        // the same rows, read transposed (A operands of dK^T / dV^T)
    char* sOt = sO;
    float* sL = (float*)(sO + Tpad * AB_RP);    // [Tpad] log-sum-exp per query (+inf on padding rows: P = 0)
    float* sD = sL + Tpad;                      // [Tpad] D per query
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int hh = blockIdx.x % heads, b = blockIdx.x / heads;
    const int D = heads * 64;
    const bf16_t* base = qkv + (long)b * Tn * 3 * D;

    {
        constexpr int IT = (AB_TPAD_MAX_DKV * 8 + 255) / 256;
        u32x4 qv[IT], dv[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = tid + it * 256;
            const int jc = min(i >> 3, Tn - 1), c = i & 7;
            qv[it] = *reinterpret_cast<const u32x4*>(base + (long)jc * 3 * D + hh * 64 + c * 8);
            dv[it] = *reinterpret_cast<const u32x4*>(d_o + ((long)b * Tn + jc) * D + hh * 64 + c * 8);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = tid + it * 256;
            const int j = i >> 3, c = i & 7;
            if (i < Tpad * 8) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4*>(sQ + j * AB_RP + c * 16) = j < Tn ? qv[it] : z;
                *reinterpret_cast<u32x4*>(sO + j * AB_RP + c * 16) = j < Tn ? dv[it] : z;
            }
        }
    }
    // D_i = sum_d dO_id O_id and the saved log-sum-exp, one query per 8 lanes (16 B of each row per lane)
    for (int i = tid; i < Tpad * 8; i += 256) {
        const int j = i >> 3, c = i & 7;
        float part = 0.f;
        {
            const long orow = ((long)b * Tn + min(j, Tn - 1)) * D + hh * 64 + c * 8;
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(d_o + orow), bb = *reinterpret_cast<const bf16x8*>(o + orow);
#pragma unroll
            for (int e = 0; e < 8; ++e) part = fmaf((float)a[e], (float)bb[e], part);
            if (j >= Tn) part = 0.f;
        }
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if (c == 0) {
            sD[j] = part;
            sL[j] = j < Tn ? lse[((long)b * heads + hh) * Tn + j] : INFINITY;
        }
    }
    __syncthreads();

    const float scale2 = scale * 1.4426950408889634f;
    const int nkt = (Tn + 31) / 32;
    const int l15 = lane & 15;
    const int t_lane_off = (4 * h + (l15 >> 2)) * AB_TP + (((lane >> 4) & 1) * 16 + (l15 & 3) * 4) * 2;

    for (int kt = wave; kt < nkt; kt += 4) {
        const int k0 = kt * 32;
        const int krow = min(k0 + l31, Tn - 1);
        const bool k_ok = k0 + l31 < Tn;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(base + (long)krow * 3 * D + D + hh * 64 + ks * 16 + h * 8);
            vf[ks] = *reinterpret_cast<const bf16x8*>(base + (long)krow * 3 * D + 2 * D + hh * 64 + ks * 16 + h * 8);
        }
        f32x16 dk[2], dv[2];                                                   // dK^T, dV^T: rows d, column = this lane's key
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }

        for (int t = 0; t < NT; ++t) {                                         // query tiles
            f32x16 acc, accp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 qa = *reinterpret_cast<const bf16x8*>(sQ + (t * 32 + l31) * AB_RP + ks * 32 + h * 16);
                const bf16x8 oa = *reinterpret_cast<const bf16x8*>(sO + (t * 32 + l31) * AB_RP + ks * 32 + h * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], acc, 0, 0, 0);       // S  [query][key]
                accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[ks], accp, 0, 0, 0);     // dP [query][key]
            }
            float p[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {                                       // register rows 4g .. 4g+3 = queries t*32 + 8g + 4h + (0..3)
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + t * 32 + 8 * g + 4 * h);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + t * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int r = 4 * g + c;
                    const float pr = k_ok ? __builtin_amdgcn_exp2f(acc[r] * scale2 - l4[c]) : 0.f;   // padding queries: lse = +inf -> 0
                    p[r] = pr;
                    acc[r] = pr * (accp[r] - d4[c]);                            // dS
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 pf, sf;
#pragma unroll
                for (int e = 0; e < 8; ++e) { pf[e] = (bf16_t)p[8 * u + e]; sf[e] = (bf16_t)acc[8 * u + e]; }
                const char* ob = sOt + (t * 32 + 16 * u) * AB_TP + t_lane_off;
                const char* qb = sQt + (t * 32 + 16 * u) * AB_TP + t_lane_off;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x4 o0 = ab_tr_read(ob + dt * 64), o1 = ab_tr_read(ob + 8 * AB_TP + dt * 64);
                    const bf16x4 q0 = ab_tr_read(qb + dt * 64), q1 = ab_tr_read(qb + 8 * AB_TP + dt * 64);
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(o0, o1, 0, 1, 2, 3, 4, 5, 6, 7), pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), sf, dk[dt], 0, 0, 0);
                }
            }
        }
        if (k_ok) {
            bf16_t* krow_out = d_qkv + ((long)b * Tn + k0 + l31) * 3 * D + D + hh * 64;
            bf16_t* vrow_out = krow_out + D;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const bf16x4 a = {(bf16_t)(dk[dt][4 * bb + 0] * scale), (bf16_t)(dk[dt][4 * bb + 1] * scale),
                                      (bf16_t)(dk[dt][4 * bb + 2] * scale), (bf16_t)(dk[dt][4 * bb + 3] * scale)};
                    const bf16x4 c = {(bf16_t)dv[dt][4 * bb + 0], (bf16_t)dv[dt][4 * bb + 1], (bf16_t)dv[dt][4 * bb + 2],
                                      (bf16_t)dv[dt][4 * bb + 3]};
                    *reinterpret_cast<bf16x4*>(krow_out + dt * 32 + 8 * bb + 4 * h) = a;
                    *reinterpret_cast<bf16x4*>(vrow_out + dt * 32 + 8 * bb + 4 * h) = c;
                }
        }
    }
}

}  // namespace

// qkv [B][T][3][heads][64], o / d_o [B][T][heads*64] (bf16); lse [B][heads][T] fp32 in log2 units (cvcl_attention_train);
// d_qkv [B][T][3][heads][64] bf16, fully written.  32 < T <= 224 (the dK/dV kernel's LDS plan), head_dim 64.
extern "C" int cvcl_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* d_qkv, int B, int T,
                                  int heads, int head_dim, float scale, void* stream) {
    CVCL_CHECK_ARG(qkv && o && d_o && lse && d_qkv && B > 0 && heads > 0, "cvcl_attention_bwd: bad args");
    CVCL_CHECK_ARG(head_dim == 64 && T > 32 && T <= AB_TPAD_MAX_DKV,
                   "cvcl_attention_bwd: needs head_dim 64 and 32 < T <= %d (got hd %d, T %d)", AB_TPAD_MAX_DKV, head_dim, T);
    static CvclLdsAttr attr_set;
    if (!attr_set.ready()) {
        if (hipFuncSetAttribute((const void*)attention_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)attention_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            cvcl_set_error("cvcl_attention_bwd: cannot raise the dynamic LDS limit");
            return CVCL_ELAUNCH;
        }
        attr_set.mark();
    }
    const int nt = (T + 31) / 32, Tpad = nt * 32;
    hipStream_t s = (hipStream_t)stream;
    CvclProfScope prof(stream, CVCL_K_ATTENTION);
    hipLaunchKernelGGL(attention_bwd_dq_kernel, dim3(B * heads), dim3(256), (size_t)Tpad * (2 * AB_RP), s, (const bf16_t*)qkv,
                       (const bf16_t*)o, (const bf16_t*)d_o, lse, (bf16_t*)d_qkv, B, T, heads, scale, nt);
    hipLaunchKernelGGL(attention_bwd_dkv_kernel, dim3(B * heads), dim3(256), (size_t)Tpad * (2 * AB_RP) + (size_t)Tpad * 8, s,
                       (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)d_o, lse, (bf16_t*)d_qkv, B, T, heads, scale, nt);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}
