// Backward pieces of the fine-tuned DINO ViT (reference: vision_transformer_dino_mugs.py:87-149 Mlp / Attention / Block under
// autograd) that are not GEMMs or the attention itself: LayerNorm backward on bf16 rows, GELU forward / backward on the saved
// pre-activation, token assembly backward.  The linears' data gradients are cvcl_gemm on transposed weight copies, their
// weight gradients cvcl_gemm_tn (wgrad.hip), the attention backward is attention_bwd.hip.
#include "cvcl_common.h"

namespace {

__device__ inline float row32_sum(float v) {            // sum over the 32 lanes that share a row (DPP inside 16, one shuffle across)
    auto dpp = [](float x, int sel) {
        const int xi = __builtin_bit_cast(int, x);
        int r;
        if (sel == 0) r = __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xf, 0xf, true);
        else if (sel == 1) r = __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xf, 0xf, true);
        else if (sel == 2) r = __builtin_amdgcn_update_dpp(0, xi, 0x141, 0xf, 0xf, true);
        else r = __builtin_amdgcn_update_dpp(0, xi, 0x140, 0xf, 0xf, true);
        return __builtin_bit_cast(float, r);
    };
    v += dpp(v, 0); v += dpp(v, 1); v += dpp(v, 2); v += dpp(v, 3);
    return v + __shfl_xor(v, 16, 64);
}

// LayerNorm backward over rows of D <= 1024 elements, 32 lanes per row, a lane owns chunks lane + 32 i (8 elements each):
//   xhat = (x - mean) rstd,  g = dy gamma,  dx = rstd (g - mean(g) - xhat mean(g xhat))  [+ add],  dgamma += dy xhat,  dbeta += dy
// Row groups walk the rows with a fixed stride, so a lane always owns the same columns and accumulates their dgamma / dbeta
// in registers; each group writes one partial row [2][D] at the end (summed in a fixed order by cvcl_colsum_f32).
template <typename TDY, int NCH, bool EXACT>
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const bf16_t* __restrict__ x, long x_row_stride, const float* __restrict__ gamma,
                                                                 const TDY* __restrict__ dy, long dy_row_stride, float eps,
                                                                 const bf16_t* __restrict__ add, bf16_t* __restrict__ dx, long dx_row_stride,
                                                                 float* __restrict__ partial, long rows, int D) {
    const int lane = threadIdx.x & 31;
    const long group = (long)blockIdx.x * 8 + (threadIdx.x >> 5), ngroups = (long)gridDim.x * 8;
    const int nch = D >> 3;
    float dg[NCH][8], db[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; }
    const float invD = 1.f / (float)D;
    for (long row = group; row < rows; row += ngroups) {
        float xv[NCH][8], gv[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 32 * i;
            if (EXACT || c < nch) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(x + row * x_row_stride + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) { xv[i][e] = (float)a[e]; s += xv[i][e]; }
                if constexpr (sizeof(TDY) == 2) {
                    const bf16x8 d = *reinterpret_cast<const bf16x8*>((const bf16_t*)dy + row * dy_row_stride + c * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) gv[i][e] = (float)d[e];
                } else {
                    const float* dp = (const float*)dy + row * dy_row_stride + c * 8;
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(dp), d1 = *reinterpret_cast<const f32x4*>(dp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gv[i][e] = d0[e]; gv[i][4 + e] = d1[e]; }
                }
            }
        }
        const float mean = row32_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (EXACT || lane + 32 * i < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float c = xv[i][e] - mean; xv[i][e] = c; q = fmaf(c, c, q); }
            }
        const float rstd = 1.f / sqrtf(row32_sum(q) * invD + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 32 * i;
            if (EXACT || c < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = xv[i][e] * rstd, d = gv[i][e];
                    xv[i][e] = xh;
                    dg[i][e] = fmaf(d, xh, dg[i][e]);
                    db[i][e] += d;
                    const float g = d * gamma[c * 8 + e];
                    gv[i][e] = g;
                    sg += g;
                    sgx = fmaf(g, xh, sgx);
                }
            }
        }
        sg = row32_sum(sg) * invD;
        sgx = row32_sum(sgx) * invD;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 32 * i;
            if (EXACT || c < nch) {
                bf16x8 o;
                bf16x8 ad;
                if (add) ad = *reinterpret_cast<const bf16x8*>(add + row * dx_row_stride + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = rstd * (gv[i][e] - sg - xv[i][e] * sgx);
                    if (add) v += (float)ad[e];
                    o[e] = (bf16_t)v;
                }
                *reinterpret_cast<bf16x8*>(dx + row * dx_row_stride + c * 8) = o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 32 * i;
        if (EXACT || c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                partial[(group * 2 + 0) * D + c * 8 + e] = dg[i][e];
                partial[(group * 2 + 1) * D + c * 8 + e] = db[i][e];
            }
        }
    }
}

// d_u == NULL: y = gelu(u);  else: y = d_u * gelu'(u), gelu'(u) = 0.5 (1 + erf(u / sqrt 2)) + u exp(-u^2 / 2) / sqrt(2 pi)
__global__ __launch_bounds__(256) void gelu_kernel(const bf16_t* __restrict__ u, const bf16_t* __restrict__ d_y, bf16_t* __restrict__ y, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(u + i * 8);
        bf16x8 o;
        if (d_y) {
            const bf16x8 d = *reinterpret_cast<const bf16x8*>(d_y + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (bf16_t)((float)d[e] * gelu_grad_fast((float)a[e]));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (bf16_t)gelu_bf16out((float)a[e]);
            }
        }
        *reinterpret_cast<bf16x8*>(y + i * 8) = o;
    }
}

// backward of cvcl_vit_assemble_tokens (h[b][0] = cls + pos[0], h[b][1 + p] = tok[b][p] + pos[1 + p]):
// d_tok[b][p] = dh[b][1 + p] (bf16 copy);  the sums over b for d_cls / d_pos are column sums of dh viewed as [B][T*D]
__global__ __launch_bounds__(256) void vit_tokens_bwd_kernel(const bf16_t* __restrict__ dh, bf16_t* __restrict__ d_tok, int B, int T, int D) {
    const long n8 = (long)B * (T - 1) * D / 8, per_img = (long)(T - 1) * D / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long b = i / per_img, r = i - b * per_img;
        *reinterpret_cast<bf16x8*>(d_tok + i * 8) = *reinterpret_cast<const bf16x8*>(dh + ((long)b * T * D + D) + r * 8);
    }
}

// out[j] = sum_b x[b][j] over B rows of n bf16 columns (d_pos / d_cls: the batch sum of the token gradients), fp32
__global__ __launch_bounds__(256) void batch_sum_bf16_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int B, long n) {
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n / 8; j += (long)gridDim.x * blockDim.x) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b0 = 0; b0 < B; b0 += 8) {                  // eight rows' loads in flight per wait, added in batch order
            bf16x8 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const bf16x8*>(x + (long)min(b0 + u, B - 1) * n + j * 8);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b0 + u < B) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += (float)a[u][e];
                }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) out[j * 8 + e] = acc[e];
    }
}

int grid_1d(long n, int cap = 4096) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// number of partial rows ([2][D] each) cvcl_layernorm_bwd_rows writes for `rows` rows
extern "C" int cvcl_layernorm_bwd_rows_partials(long rows) {
    long wgs = (rows + 7) / 8;
    if (wgs > 768) wgs = 768;                              // 3 workgroups per CU (register-limited residency); more only adds partial rows
    return (int)(wgs < 1 ? 8 : wgs * 8);
}

// x bf16 rows (stride x_row_stride elements), dy bf16 (dy_is_f32 = 0) or fp32 rows, dx bf16 rows = LN backward (+ add, nullable,
// same layout as dx); partial [cvcl_layernorm_bwd_rows_partials(rows)][2][D] fp32: dgamma / dbeta partial sums.  D % 8 == 0, D <= 1024.
extern "C" int cvcl_layernorm_bwd_rows(const void* x, long x_row_stride, const float* gamma, const void* dy, int dy_is_f32,
                                       long dy_row_stride, float eps, const void* add, void* dx, long dx_row_stride, float* partial,
                                       long rows, int D, void* stream) {
    CVCL_CHECK_ARG(x && gamma && dy && dx && partial && rows > 0, "cvcl_layernorm_bwd_rows: bad args");
    CVCL_CHECK_ARG(D % 8 == 0 && D <= 1024 && x_row_stride % 8 == 0 && dx_row_stride % 8 == 0 && dy_row_stride % (dy_is_f32 ? 4 : 8) == 0,
                   "cvcl_layernorm_bwd_rows: needs D %% 8 == 0, D <= 1024 and 16-byte aligned rows (D %d)", D);
    const int wgs = cvcl_layernorm_bwd_rows_partials(rows) / 8;
    CvclProfScope prof(stream, CVCL_K_LAYERNORM);
    if (!dy_is_f32 && D == 768)                               // ViT-B rows: exactly 3 chunks per lane, no branches around the loads
        hipLaunchKernelGGL((layernorm_bwd_rows_kernel<bf16_t, 3, true>), dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_row_stride, gamma,
                           (const bf16_t*)dy, dy_row_stride, eps, (const bf16_t*)add, (bf16_t*)dx, dx_row_stride, partial, rows, D);
    else if (!dy_is_f32 && D < 768)
        hipLaunchKernelGGL((layernorm_bwd_rows_kernel<bf16_t, 3, false>), dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_row_stride, gamma,
                           (const bf16_t*)dy, dy_row_stride, eps, (const bf16_t*)add, (bf16_t*)dx, dx_row_stride, partial, rows, D);
    else if (dy_is_f32)
        hipLaunchKernelGGL((layernorm_bwd_rows_kernel<float, 4, false>), dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_row_stride, gamma,
                           (const float*)dy, dy_row_stride, eps, (const bf16_t*)add, (bf16_t*)dx, dx_row_stride, partial, rows, D);
    else
        hipLaunchKernelGGL((layernorm_bwd_rows_kernel<bf16_t, 4, false>), dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_row_stride, gamma,
                           (const bf16_t*)dy, dy_row_stride, eps, (const bf16_t*)add, (bf16_t*)dx, dx_row_stride, partial, rows, D);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

// d_y == NULL: y = gelu(u) (erf form, the forward epilogue's approximation); else y = d_y * gelu'(u).  bf16, n % 8 == 0.
extern "C" int cvcl_gelu_bf16(const void* u, const void* d_y, void* y, long n, void* stream) {
    CVCL_CHECK_ARG(u && y && n > 0 && n % 8 == 0, "cvcl_gelu_bf16: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    hipLaunchKernelGGL(gelu_kernel, dim3(grid_1d(n / 8, 8192)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)u, (const bf16_t*)d_y,
                       (bf16_t*)y, n / 8);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

// dh [B][T][D] bf16 -> d_tok [B][T-1][D] bf16 (patch rows), d_pos [T][D] fp32 (batch sum; d_cls = its row 0)
extern "C" int cvcl_vit_tokens_bwd(const void* dh, void* d_tok, float* d_pos, int B, int T, int D, void* stream) {
    CVCL_CHECK_ARG(dh && d_tok && d_pos && B > 0 && T > 1 && D % 8 == 0, "cvcl_vit_tokens_bwd: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    hipLaunchKernelGGL(vit_tokens_bwd_kernel, dim3(grid_1d((long)B * (T - 1) * D / 8, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dh, (bf16_t*)d_tok, B, T, D);
    hipLaunchKernelGGL(batch_sum_bf16_kernel, dim3(grid_1d((long)T * D / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dh, d_pos, B,
                       (long)T * D);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}
