// Transformer-side kernels: DINO ViT image encoder (reference multimodal/vision_transformer_dino_mugs.py:87-250),
// the one-layer text transformer (multimodal/multimodal.py:553-573, nn.TransformerEncoderLayer) and the LSTM text
// encoder (multimodal/multimodal.py:513-552).  Every linear layer runs on the MFMA GEMM of gemm.hip; this file
// holds what is not a GEMM: patch gather, token assembly, LayerNorm, attention, embedding(+pos) gather, sequence
// pooling and the LSTM cell.
#include "cvcl_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// PatchEmbed conv (k = s = p) as unfold + GEMM (vit:162,166): cols[b*np + i][c*p*p + ky*p + kx], zero padded to Kpad
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col_patches_kernel(const float* __restrict__ x, T* __restrict__ cols, int B, int H,
                                                             int W, int p, int Kpad) {
    const int gh = H / p, gw = W / p, K = 3 * p * p;
    const long total = (long)B * gh * gw * Kpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad);
        const long r = i / Kpad;
        float v = 0.f;
        if (k < K) {
            const int c = k / (p * p), rem = k - c * p * p, ky = rem / p, kx = rem - ky * p;
            const int px = (int)(r % gw), py = (int)((r / gw) % gh), b = (int)(r / ((long)gw * gh));
            v = x[(((long)b * 3 + c) * H + py * p + ky) * W + px * p + kx];
        }
        cols[i] = ElemTraits<T>::from_f(v);
    }
}

// h[b][0] = cls + pos[0];  h[b][1+i] = tok[b][i] + pos[1+i]      (prepare_tokens, vit:232-243)
template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_kernel(const T* __restrict__ tok, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, T* __restrict__ h, int B, int Tn,
                                                           int D) {
    const long total = (long)B * Tn * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const long r = i / D;
        const int t = (int)(r % Tn);
        const long b = r / Tn;
        const float base = t == 0 ? cls[d] : ElemTraits<T>::to_f(tok[(b * (Tn - 1) + (t - 1)) * D + d]);
        h[i] = ElemTraits<T>::from_f(base + pos[(long)t * D + d]);
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dimension, one wave per row; rows may be strided (cls-token rows); output T or fp32
// ------------------------------------------------------------------------------------------------
template <typename T, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, long x_row_stride, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, TO* __restrict__ y,
                                                        long rows, int D) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const T* xr = x + row * x_row_stride;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += ElemTraits<T>::to_f(xr[d]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float c = ElemTraits<T>::to_f(xr[d]) - mean;
        q = fmaf(c, c, q);
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
    for (int d = lane; d < D; d += 64) {
        const float v = (ElemTraits<T>::to_f(xr[d]) - mean) * rstd * gamma[d] + beta[d];
        y[row * D + d] = ElemTraits<TO>::from_f(v);
    }
}

// ------------------------------------------------------------------------------------------------
// Attention, generic (any head_dim <= 128, any T, optional key padding mask): one wave per (b, head, query).
// qkv [B, T, 3, heads, hd] -> out [B, T, heads*hd].  P is rounded to the storage type before P.V.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attention_valu_kernel(const T* __restrict__ qkv, const int64_t* __restrict__ key_tok,
                                                             T* __restrict__ out, int B, int Tn, int heads, int hd,
                                                             float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* sq = (float*)smem + (size_t)wave * (hd + Tn);     // q vector then probabilities, per wave
    float* sp = sq + hd;
    const long widx = (long)blockIdx.x * 4 + wave;
    const long nwork = (long)B * heads * Tn;
    if (widx >= nwork) return;
    const int qi = (int)(widx % Tn), hh = (int)((widx / Tn) % heads), b = (int)(widx / ((long)Tn * heads));
    const int D = heads * hd;
    const T* base = qkv + (long)b * Tn * 3 * D;
    const T* qp = base + (long)qi * 3 * D + hh * hd;
    for (int d = lane; d < hd; d += 64) sq[d] = ElemTraits<T>::to_f(qp[d]);
    float mx = -INFINITY;
    for (int j = lane; j < Tn; j += 64) {
        const T* kp = base + (long)j * 3 * D + D + hh * hd;
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s = fmaf(sq[d], ElemTraits<T>::to_f(kp[d]), s);
        s *= scale;
        if (key_tok && key_tok[(long)b * Tn + j] == 0) s = -INFINITY;     // src_key_padding_mask (x == 0)
        sp[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < Tn; j += 64) {
        const float e = expf(sp[j] - mx);
        sp[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    for (int j = lane; j < Tn; j += 64) sp[j] = ElemTraits<T>::to_f(ElemTraits<T>::from_f(sp[j] / sum));
    T* op = out + ((long)b * Tn + qi) * D + hh * hd;
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < Tn; ++j) acc = fmaf(sp[j], ElemTraits<T>::to_f(base[(long)j * 3 * D + 2 * D + hh * hd + d]), acc);
        op[d] = ElemTraits<T>::from_f(acc);
    }
}

// ------------------------------------------------------------------------------------------------
// Attention, bf16 MFMA, head_dim 64 (ViT-B: 12 x 64): one workgroup per (b, head, 64 queries); each wave owns
// 16 queries.  S^T = K.Q^T on v_mfma_f32_16x16x32_bf16 with K fragments read straight from global (L2 resident),
// softmax in registers, P staged through LDS into the A-operand layout, V staged transposed in LDS for P.V.
// ------------------------------------------------------------------------------------------------
constexpr int ATT_TPAD_MAX = 288;      // keys padded to a multiple of 32 (T <= 288 covers ViT-B/14 at 224: 257)

__global__ __launch_bounds__(256) void attention_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int B,
                                                             int Tn, int heads, float scale, int Tpad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VT_PITCH = Tpad + 8;                                  // bf16 elements per transposed-V row
    bf16_t* sVt = (bf16_t*)smem;                                    // [64 d][VT_PITCH]
    bf16_t* sP = sVt + 64 * VT_PITCH;                               // [4 waves][16 q][Tpad + 8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int qblocks = (Tn + 63) / 64;
    const int qb = blockIdx.x % qblocks, hh = (blockIdx.x / qblocks) % heads, b = blockIdx.x / (qblocks * heads);
    const int D = heads * 64;
    const bf16_t* base = qkv + (long)b * Tn * 3 * D;

    // stage V^T (zeros beyond Tn) -- coalesced 16-B reads along d, scattered 2-B LDS writes
    for (int i = tid; i < Tpad * 8; i += 256) {
        const int j = i >> 3, dc = (i & 7) * 8;
        bf16x8 v;
        if (j < Tn) v = *reinterpret_cast<const bf16x8*>(base + (long)j * 3 * D + 2 * D + hh * 64 + dc);
        else { const u32x4 z = {0u, 0u, 0u, 0u}; v = __builtin_bit_cast(bf16x8, z); }
#pragma unroll
        for (int e = 0; e < 8; ++e) sVt[(dc + e) * VT_PITCH + j] = v[e];
    }
    // this wave's 16 queries as the MFMA B operand (cols = query, k = d): lane (query l15, d block g)
    const int q0 = qb * 64 + wave * 16;
    const int qrow = min(q0 + l15, Tn - 1);
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(base + (long)qrow * 3 * D + hh * 64 + ks * 32 + g * 8);

    // S^T tiles: rows = keys (16 per tile), cols = queries; lane holds keys 4g..4g+3 of each tile for query l15
    const int ntile = Tpad / 16;
    f32x4 s[ATT_TPAD_MAX / 16];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < ATT_TPAD_MAX / 16; ++t) {
        if (t < ntile) {
            const int krow = min(t * 16 + l15, Tn - 1);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(base + (long)krow * 3 * D + D + hh * 64 + ks * 32 + g * 8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = t * 16 + g * 4 + e;
                acc[e] = key < Tn ? acc[e] * scale : -INFINITY;
                mx = fmaxf(mx, acc[e]);
            }
            s[t] = acc;
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < ATT_TPAD_MAX / 16; ++t)
        if (t < ntile) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[t][e] = expf(s[t][e] - mx); sum += s[t][e]; }
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    // P (bf16) -> LDS [q][key] so it can be re-read as the A operand (8 consecutive keys per lane)
    bf16_t* myP = sP + wave * 16 * (Tpad + 8);
#pragma unroll
    for (int t = 0; t < ATT_TPAD_MAX / 16; ++t)
        if (t < ntile) {
            bf16x4 pv = {(bf16_t)(s[t][0] * inv), (bf16_t)(s[t][1] * inv), (bf16_t)(s[t][2] * inv), (bf16_t)(s[t][3] * inv)};
            *reinterpret_cast<bf16x4*>(myP + l15 * (Tpad + 8) + t * 16 + g * 4) = pv;
        }
    __syncthreads();                                                 // V^T complete (and P visible to its own wave)
    // O[16 q][64 d] = P[16 x Tpad] . V[Tpad x 64]: A = P rows (query l15, keys g*8..), B = V^T rows (d l15, keys g*8..)
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < Tpad / 32; ++ks) {
        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(myP + l15 * (Tpad + 8) + ks * 32 + g * 8);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sVt + (dt * 16 + l15) * VT_PITCH + ks * 32 + g * 8);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[dt], 0, 0, 0);
        }
    }
    // D[row = query 4g+e][col = d l15]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int q = q0 + g * 4 + e;
        if (q < Tn) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) out[((long)b * Tn + q) * D + hh * 64 + dt * 16 + l15] = (bf16_t)o[dt][e];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// text side
// ------------------------------------------------------------------------------------------------
// x[b][l][:] = table[tok[b][l]] (+ pos[l])                       (multimodal.py:496, 561-563)
__global__ __launch_bounds__(256) void embed_gather_pos_kernel(const float* __restrict__ table, const int64_t* __restrict__ tok,
                                                               const float* __restrict__ pos, float* __restrict__ x, int B,
                                                               int L, int E, int V) {
    const long total = (long)B * L * E;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const long r = i / E;
        const int l = (int)(r % L);
        const int64_t t = tok[r];
        float v = (t >= 0 && t < V) ? table[t * E + e] : NAN;
        if (pos) v += pos[(long)l * E + e];
        x[i] = v;
    }
}

// ret[b][:] = sum_l x[b][l][:] / len[b]   (all L positions, pads included: multimodal.py:573, Appendix C.1)
__global__ __launch_bounds__(256) void seq_sum_div_kernel(const float* __restrict__ x, const int64_t* __restrict__ len,
                                                          float* __restrict__ ret, int B, int L, int E) {
    const long total = (long)B * E;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const long b = i / E;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc += x[(b * L + l) * E + e];
        ret[i] = acc / (float)len[b];
    }
}

// LSTM cell for step t (gate order i,f,g,o; nn.LSTM): gates [B,4H] already = x_t W_ih^T + b_ih + b_hh + h W_hh^T.
// Sequences shorter than t+1 keep their state (packed-sequence semantics) and emit zeros (pad_packed_sequence).
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ gates, const int64_t* __restrict__ len, int t,
                                                        float* __restrict__ h, float* __restrict__ c, float* __restrict__ out,
                                                        int B, int L, int Hd) {
    const long total = (long)B * Hd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % Hd);
        const long b = i / Hd;
        const bool live = len[b] > t;
        float ho = 0.f;
        if (live) {
            const float* gp = gates + b * 4 * Hd;
            const float ig = 1.f / (1.f + expf(-gp[j]));
            const float fg = 1.f / (1.f + expf(-gp[Hd + j]));
            const float gg = tanhf(gp[2 * Hd + j]);
            const float og = 1.f / (1.f + expf(-gp[3 * Hd + j]));
            const float cn = fg * c[i] + ig * gg;
            ho = og * tanhf(cn);
            c[i] = cn;
            h[i] = ho;
        }
        if (out) out[(b * L + t) * Hd + j] = ho;
    }
}

int grid_for(long total, int per_block = 256, int cap = 8192) {
    long g = (total + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

}  // namespace

// ================================================================================================
extern "C" int cvcl_im2col_patches(int dtype, const float* x_nchw, void* cols, int B, int H, int W, int patch, int Kpad,
                                   void* stream) {
    CVCL_CHECK_ARG(x_nchw && cols && B > 0 && patch > 0 && H % patch == 0 && W % patch == 0 && Kpad >= 3 * patch * patch,
                   "cvcl_im2col_patches: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    const long total = (long)B * (H / patch) * (W / patch) * Kpad;
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(im2col_patches_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                           (float*)cols, B, H, W, patch, Kpad);
    else
        hipLaunchKernelGGL(im2col_patches_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                           (bf16_t*)cols, B, H, W, patch, Kpad);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_vit_assemble_tokens(int dtype, const void* tok, const float* cls, const float* pos, void* h, int B, int T,
                                        int D, void* stream) {
    CVCL_CHECK_ARG(tok && cls && pos && h && B > 0 && T > 1 && D > 0, "cvcl_vit_assemble_tokens: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    const long total = (long)B * T * D;
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(vit_assemble_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)tok,
                           cls, pos, (float*)h, B, T, D);
    else
        hipLaunchKernelGGL(vit_assemble_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)tok, cls, pos, (bf16_t*)h, B, T, D);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_layernorm(int dtype, const void* x, long x_row_stride, const float* gamma, const float* beta, float eps,
                              void* y, int y_is_f32, long rows, int D, void* stream) {
    CVCL_CHECK_ARG(x && gamma && beta && y && rows > 0 && D > 0 && x_row_stride >= D, "cvcl_layernorm: bad args");
    CvclProfScope prof(stream, CVCL_K_LAYERNORM);
    dim3 grid(cvcl_div_up(rows, 4));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL((layernorm_kernel<float, float>), grid, dim3(256), 0, s, (const float*)x, x_row_stride, gamma, beta,
                           eps, (float*)y, rows, D);
    else if (y_is_f32)
        hipLaunchKernelGGL((layernorm_kernel<bf16_t, float>), grid, dim3(256), 0, s, (const bf16_t*)x, x_row_stride, gamma, beta,
                           eps, (float*)y, rows, D);
    else
        hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, x_row_stride, gamma, beta,
                           eps, (bf16_t*)y, rows, D);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_attention(int dtype, const void* qkv, const int64_t* key_tok, void* out, int B, int T, int heads,
                              int head_dim, float scale, void* stream) {
    CVCL_CHECK_ARG(qkv && out && B > 0 && T > 0 && heads > 0 && head_dim > 0 && head_dim <= 128, "cvcl_attention: bad args");
    CvclProfScope prof(stream, CVCL_K_ATTENTION);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CVCL_BF16 && head_dim == 64 && !key_tok && T <= ATT_TPAD_MAX) {
        const int Tpad = (T + 31) / 32 * 32;
        const size_t lds = (size_t)(64 * (Tpad + 8) + 4 * 16 * (Tpad + 8)) * 2;
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)attention_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                cvcl_set_error("cvcl_attention: cannot raise the dynamic LDS limit");
                return CVCL_ELAUNCH;
            }
            attr_set = true;
        }
        const int grid = B * heads * ((T + 63) / 64);
        hipLaunchKernelGGL(attention_mfma_kernel, dim3(grid), dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)out, B, T, heads,
                           scale, Tpad);
        CVCL_LAUNCH_CHECK();
        return CVCL_OK;
    }
    const long nwork = (long)B * heads * T;
    const size_t lds = (size_t)4 * (head_dim + T) * sizeof(float);
    CVCL_CHECK_ARG(lds <= 64 * 1024, "cvcl_attention: sequence too long for the generic kernel (%d)", T);
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(attention_valu_kernel<float>, dim3(cvcl_div_up(nwork, 4)), dim3(256), lds, s, (const float*)qkv, key_tok,
                           (float*)out, B, T, heads, head_dim, scale);
    else
        hipLaunchKernelGGL(attention_valu_kernel<bf16_t>, dim3(cvcl_div_up(nwork, 4)), dim3(256), lds, s, (const bf16_t*)qkv,
                           key_tok, (bf16_t*)out, B, T, heads, head_dim, scale);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_embed_gather_pos(const float* table, const int64_t* tok, const float* pos, float* x, int B, int L, int E,
                                     int V, void* stream) {
    CVCL_CHECK_ARG(table && tok && x && B > 0 && L > 0 && E > 0 && V > 0, "cvcl_embed_gather_pos: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    hipLaunchKernelGGL(embed_gather_pos_kernel, dim3(grid_for((long)B * L * E)), dim3(256), 0, (hipStream_t)stream, table, tok,
                       pos, x, B, L, E, V);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_seq_sum_div(const float* x, const int64_t* len, float* ret, int B, int L, int E, void* stream) {
    CVCL_CHECK_ARG(x && len && ret && B > 0 && L > 0 && E > 0, "cvcl_seq_sum_div: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    hipLaunchKernelGGL(seq_sum_div_kernel, dim3(grid_for((long)B * E)), dim3(256), 0, (hipStream_t)stream, x, len, ret, B, L, E);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_lstm_cell(const float* gates, const int64_t* len, int t, float* h, float* c, float* out, int B, int L,
                              int Hd, void* stream) {
    CVCL_CHECK_ARG(gates && len && h && c && B > 0 && L > 0 && Hd > 0 && t >= 0 && t < L, "cvcl_lstm_cell: bad args");
    CvclProfScope prof(stream, CVCL_K_LSTM);
    hipLaunchKernelGGL(lstm_cell_kernel, dim3(grid_for((long)B * Hd)), dim3(256), 0, (hipStream_t)stream, gates, len, t, h, c, out,
                       B, L, Hd);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}
