// Backward-side kernels of the ResNeXt trunk for --finetune_cnn (reference: VisionEncoder with finetune_cnn=True,
// multimodal/multimodal.py:175-179; the arithmetic is autograd through torchvision's Bottleneck).  These are used by the
// autograd-composed fine-tuning path (multimodal/trunk_train.py); the frozen-CNN fast path never runs them.
// NHWC activations viewed as [rows, C]; dtype T = storage type (fp32 parity mode / bf16), reductions in fp32.
#include <cstdlib>

#include "cvcl_common.h"

namespace {

int grid_for(long total, int per_block = 256, int cap = 8192) {
    long g = (total + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

// ---- BatchNorm (train mode) forward apply without ReLU:  y = x * scale + shift  (relu optional) -------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, T* __restrict__ y, long rows, int C,
                                                       int relu) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float v = fmaf(ElemTraits<T>::to_f(x[i]), scale[c], shift[c]);
        if (relu) v = fmaxf(v, 0.f);
        y[i] = ElemTraits<T>::from_f(v);
    }
}

// ---- BatchNorm backward, pass 1: per-channel partial sums of g and g * xhat, g = dy * (y > 0 if relu) ---------------
// xhat = (x - mean) * rstd.  partial rows [gridDim.x][2][C] (same layout as the forward statistics).
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                            const T* __restrict__ dy, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, long rows, int C, int relu,
                                                            float* __restrict__ partial) {
    __shared__ float ps[4][64], pq[4][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6, ch = blockIdx.y * 64 + c;
    float s = 0.f, q = 0.f;
    if (ch < C) {
        const float mu = mean[ch], rs = rstd[ch];
        for (long r = (long)blockIdx.x * 4 + sl; r < rows; r += (long)gridDim.x * 4) {
            float g = ElemTraits<T>::to_f(dy[r * C + ch]);
            if (relu && !(ElemTraits<T>::to_f(y[r * C + ch]) > 0.f)) g = 0.f;
            const float xh = (ElemTraits<T>::to_f(x[r * C + ch]) - mu) * rs;
            s += g;
            q = fmaf(g, xh, q);
        }
    }
    ps[sl][c] = s;
    pq[sl][c] = q;
    __syncthreads();
    if (sl == 0 && ch < C) {
        partial[((long)blockIdx.x * 2 + 0) * C + ch] = (ps[0][c] + ps[1][c]) + (ps[2][c] + ps[3][c]);
        partial[((long)blockIdx.x * 2 + 1) * C + ch] = (pq[0][c] + pq[1][c]) + (pq[2][c] + pq[3][c]);
    }
}

// reduce partial rows -> dbeta = sum g, dgamma = sum g*xhat (fp64 accumulation, fixed order)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int prow, int C,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    double s = 0.0, q = 0.0;
    for (int r = 0; r < prow; ++r) {
        s += (double)partial[((long)r * 2 + 0) * C + ch];
        q += (double)partial[((long)r * 2 + 1) * C + ch];
    }
    dbeta[ch] = (float)s;
    dgamma[ch] = (float)q;
}

// ---- BatchNorm backward, pass 2: dx = gamma * rstd * (g - dbeta / n - xhat * dgamma / n) ------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                           const T* __restrict__ dy, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           T* __restrict__ dx, long rows, int C, int relu) {
    const long total = rows * C;
    const float inv_n = 1.f / (float)rows;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float g = ElemTraits<T>::to_f(dy[i]);
        if (relu && !(ElemTraits<T>::to_f(y[i]) > 0.f)) g = 0.f;
        const float xh = (ElemTraits<T>::to_f(x[i]) - mean[c]) * rstd[c];
        dx[i] = ElemTraits<T>::from_f(gamma[c] * rstd[c] * (g - dbeta[c] * inv_n - xh * dgamma[c] * inv_n));
    }
}

// ---- 2-D transpose out[c][r] = in[r][c] (operand re-layout for the weight-gradient GEMMs: contraction over rows) -------
template <typename T>
__global__ void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, long rows, int cols) {
    __shared__ float tile[32][33];
    const long by = (long)blockIdx.y * 32;
    const int bx = blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const long r = by + i;
        const int c = bx + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? ElemTraits<T>::to_f(in[r * cols + c]) : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int c = bx + i;
        const long r = by + threadIdx.x;
        if (c < cols && r < rows) out[(long)c * rows + r] = ElemTraits<T>::from_f(tile[threadIdx.x][i]);
    }
}

// ---- y = relu(a + b) and its backward mask are served by bn_add_relu / relu_bwd; elementwise add for gradients --------
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = ElemTraits<T>::from_f(ElemTraits<T>::to_f(a[i]) + ElemTraits<T>::to_f(b[i]));
}

// dx = dy where y > 0 (typed)
template <typename T>
__global__ __launch_bounds__(256) void relu_mask_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dx[i] = ElemTraits<T>::to_f(y[i]) > 0.f ? dy[i] : ElemTraits<T>::from_f(0.f);
}

// ---- max pool 3x3 / 2 pad 1 (NHWC) forward on an already-activated tensor, and backward (first arg-max wins) -----------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), b = (int)(p / ((long)Wo * Ho));
        float m = -INFINITY;
        for (int ky = 0; ky < 3; ++ky) {
            const int yin = 2 * oy - 1 + ky;
            if (yin < 0 || yin >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xin = 2 * ox - 1 + kx;
                if (xin < 0 || xin >= W) continue;
                m = fmaxf(m, ElemTraits<T>::to_f(x[(((long)b * H + yin) * W + xin) * C + c]));
            }
        }
        y[i] = ElemTraits<T>::from_f(m);
    }
}

// gather form of the backward (deterministic, no atomics): dx[in] = sum over the <= 4 windows containing `in` of
// dy[window] where `in` is that window's first arg-max
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                          int B, int H, int W, int C) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int xi = (int)(p % W), yi = (int)((p / W) % H), b = (int)(p / ((long)W * H));
        const float v = ElemTraits<T>::to_f(x[i]);
        float acc = 0.f;
        for (int oy = (yi + 1) / 2 - 1; oy <= (yi + 1) / 2; ++oy) {         // windows with 2*oy - 1 <= yi <= 2*oy + 1
            if (oy < 0 || oy >= Ho || 2 * oy - 1 > yi || 2 * oy + 1 < yi) continue;
            for (int ox = (xi + 1) / 2 - 1; ox <= (xi + 1) / 2; ++ox) {
                if (ox < 0 || ox >= Wo || 2 * ox - 1 > xi || 2 * ox + 1 < xi) continue;
                // is (yi, xi) the first arg-max of window (oy, ox)?  (row-major scan order, like torch)
                bool first = true;
                for (int ky = 0; ky < 3 && first; ++ky) {
                    const int yy = 2 * oy - 1 + ky;
                    if (yy < 0 || yy >= H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = 2 * ox - 1 + kx;
                        if (xx < 0 || xx >= W) continue;
                        const float u = ElemTraits<T>::to_f(x[(((long)b * H + yy) * W + xx) * C + c]);
                        const bool before = (yy < yi) || (yy == yi && xx < xi);
                        if (u > v || (before && u == v)) { first = false; break; }
                    }
                }
                if (first) acc += ElemTraits<T>::to_f(dy[(((long)b * Ho + oy) * Wo + ox) * C + c]);
            }
        }
        dx[i] = ElemTraits<T>::from_f(acc);
    }
}

// ---- avg pool backward: dx[b,p,c] = d_pooled[b,c] / HW -------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dp, T* __restrict__ dx, int B, int HW, int C) {
    const long total = (long)B * HW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long b = i / ((long)HW * C);
        dx[i] = ElemTraits<T>::from_f(dp[b * C + c] / (float)HW);
    }
}

// ---- zero-stuffing for the data gradient of a stride-2 convolution: z[b, 2*oy, 2*ox, :] = dy[b, oy, ox, :], 0 elsewhere ---
template <typename T>
__global__ __launch_bounds__(256) void zero_stuff_kernel(const T* __restrict__ dy, T* __restrict__ z, int B, int Ho, int Wo, int C) {
    const int H = 2 * Ho, W = 2 * Wo;
    const long total = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int xi = (int)(p % W), yi = (int)((p / W) % H), b = (int)(p / ((long)W * H));
        z[i] = (!(yi & 1) && !(xi & 1)) ? dy[(((long)b * Ho + (yi >> 1)) * Wo + (xi >> 1)) * C + c] : ElemTraits<T>::from_f(0.f);
    }
}

// ---- weight gradient of a KxK convolution (grouped or not), direct form ----------------------------------------------------
// dW[co][ci][ky][kx] = sum_{b,oy,ox} dY[b,oy,ox,co] * X[b, oy*s - pad + ky, ox*s - pad + kx, g0 + ci]
// x_nchw != 0: X is the NCHW fp32 image (stem).  One workgroup per (co, ci-slice): threads stride over pixels, block reduce.
// O(#weights * pixels) loads: fine for the parity-sized fine-tuning path, not a bench kernel.
template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_direct_kernel(const void* __restrict__ xin, const T* __restrict__ dy,
                                                                float* __restrict__ dw, int B, int H, int W, int Cin, int Cout,
                                                                int cg, int k, int stride, int pad, int Ho, int Wo,
                                                                int x_nchw) {
    __shared__ float scratch[8];
    const int widx = blockIdx.x;                           // (co, ci, ky, kx)
    const int kx = widx % k, ky = (widx / k) % k, ci = (widx / (k * k)) % cg, co = widx / (k * k * cg);
    const int cin_abs = (co / (Cout / (Cin / cg))) * cg + ci;       // groups = Cin / cg; group of co = co / (Cout / groups)
    const long npix = (long)B * Ho * Wo;
    float acc = 0.f;
    for (long p = threadIdx.x; p < npix; p += blockDim.x) {
        const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), b = (int)(p / ((long)Wo * Ho));
        const int yi = oy * stride - pad + ky, xi = ox * stride - pad + kx;
        if (yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
        float xv;
        if (x_nchw) xv = ((const float*)xin)[(((long)b * Cin + cin_abs) * H + yi) * W + xi];
        else xv = ElemTraits<T>::to_f(((const T*)xin)[(((long)b * H + yi) * W + xi) * Cin + cin_abs]);
        acc = fmaf(ElemTraits<T>::to_f(dy[p * Cout + co]), xv, acc);
    }
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) dw[widx] = acc;
}

// batch mean / rstd from the forward statistics rows [prow][2][C] (same fp64 reduction order as bn_finalize, biased variance)
__global__ __launch_bounds__(256) void bn_moments_kernel(const float* __restrict__ stats, int prow, long count, float eps, int C,
                                                         float* __restrict__ mean, float* __restrict__ rstd) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    double s = 0.0, q = 0.0;
    for (int r = 0; r < prow; ++r) {
        s += (double)stats[((long)r * 2 + 0) * C + ch];
        q += (double)stats[((long)r * 2 + 1) * C + ch];
    }
    const double m = s / (double)count;
    double var = q / (double)count - m * m;
    if (var < 0.0) var = 0.0;
    mean[ch] = (float)m;
    rstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
}

// weight of the data-gradient convolution of a grouped 3x3 conv with as many outputs as inputs per group:
// out[g*cg + ci][co_l][ky][kx] = w[g*cg + co_l][ci][2-ky][2-kx]   (OIHW f32 -> OIHW f32)
__global__ __launch_bounds__(256) void gconv_wflip_kernel(const float* __restrict__ w, float* __restrict__ out, int C, int cg) {
    const long total = (long)C * cg * 9;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % 9), col = (int)((i / 9) % cg), o = (int)(i / (9 * cg));
        const int g = o / cg, ci = o % cg;
        out[i] = w[((long)(g * cg + col) * cg + ci) * 9 + (8 - t)];
    }
}

}  // namespace

#define CVCL_DISPATCH_T(dtype, KERNEL, GRID, BLOCK, STREAM, ...)                                            \
    do {                                                                                                    \
        if ((dtype) == CVCL_F32) hipLaunchKernelGGL(KERNEL<float>, GRID, BLOCK, 0, (hipStream_t)(STREAM), __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<bf16_t>, GRID, BLOCK, 0, (hipStream_t)(STREAM), __VA_ARGS__);        \
    } while (0)

extern "C" int cvcl_bn_apply(int dtype, const void* x, const float* scale, const float* shift, void* y, long rows, int C, int relu,
                             void* stream) {
    CVCL_CHECK_ARG(x && scale && shift && y && rows > 0 && C > 0, "cvcl_bn_apply: bad args");
    CvclProfScope prof(stream, CVCL_K_BN_APPLY);
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, (const float*)x, scale,
                           shift, (float*)y, rows, C, relu);
    else
        hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           scale, shift, (bf16_t*)y, rows, C, relu);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_bn_bwd(int dtype, const void* x, const void* y, const void* dy, const float* mean, const float* rstd,
                           const float* gamma, float* dgamma, float* dbeta, void* dx, long rows, int C, int relu, float* partial,
                           int partial_rows, void* stream) {
    CVCL_CHECK_ARG(x && dy && mean && rstd && gamma && dgamma && dbeta && dx && partial && rows > 0 && C > 0 && (!relu || y),
                   "cvcl_bn_bwd: bad args");
    int g = (int)((rows + 255) / 256);
    if (g < 1) g = 1;
    if (g > 256) g = 256;
    CVCL_CHECK_ARG(partial_rows >= g, "cvcl_bn_bwd: partial_rows %d < %d", partial_rows, g);
    CvclProfScope prof(stream, CVCL_K_OTHER);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(g, cvcl_div_up(C, 64));
    if (dtype == CVCL_F32) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)y, (const float*)dy, mean,
                           rstd, rows, C, relu, partial);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cvcl_div_up(C, 256)), dim3(256), 0, s, partial, g, C, dgamma, dbeta);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid_for(rows * C)), dim3(256), 0, s, (const float*)x, (const float*)y,
                           (const float*)dy, mean, rstd, gamma, dgamma, dbeta, (float*)dx, rows, C, relu);
    } else {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy,
                           mean, rstd, rows, C, relu, partial);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cvcl_div_up(C, 256)), dim3(256), 0, s, partial, g, C, dgamma, dbeta);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid_for(rows * C)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)y,
                           (const bf16_t*)dy, mean, rstd, gamma, dgamma, dbeta, (bf16_t*)dx, rows, C, relu);
    }
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_bn_batch_moments(const float* stats, int stats_rows, long count, float eps, float* mean, float* rstd, int C,
                                     void* stream) {
    CVCL_CHECK_ARG(stats && mean && rstd && stats_rows > 0 && count > 0 && C > 0, "cvcl_bn_batch_moments: bad args");
    CvclProfScope prof(stream, CVCL_K_BN_FINALIZE);
    hipLaunchKernelGGL(bn_moments_kernel, dim3(cvcl_div_up(C, 256)), dim3(256), 0, (hipStream_t)stream, stats, stats_rows, count, eps, C,
                       mean, rstd);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_gconv_weight_dgrad(const float* w, float* out, int C, int cin_per_group, void* stream) {
    CVCL_CHECK_ARG(w && out && C > 0 && cin_per_group > 0 && C % cin_per_group == 0, "cvcl_gconv_weight_dgrad: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    hipLaunchKernelGGL(gconv_wflip_kernel, dim3(grid_for((long)C * cin_per_group * 9)), dim3(256), 0, (hipStream_t)stream, w, out, C,
                       cin_per_group);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_transpose(int dtype, const void* in, void* out, long rows, int cols, void* stream) {
    CVCL_CHECK_ARG(in && out && rows > 0 && cols > 0, "cvcl_transpose: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    dim3 grid(cvcl_div_up(cols, 32), cvcl_div_up(rows, 32));
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(32, 8), 0, (hipStream_t)stream, (const float*)in, (float*)out, rows, cols);
    else
        hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(32, 8), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, rows, cols);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_add(int dtype, const void* a, const void* b, void* y, long n, void* stream) {
    CVCL_CHECK_ARG(a && b && y && n > 0, "cvcl_add: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(add_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)y, n);
    else
        hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_relu_mask(int dtype, const void* y, const void* dy, void* dx, long n, void* stream) {
    CVCL_CHECK_ARG(y && dy && dx && n > 0, "cvcl_relu_mask: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(relu_mask_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)y, (const float*)dy, (float*)dx, n);
    else
        hipLaunchKernelGGL(relu_mask_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (const bf16_t*)dy, (bf16_t*)dx, n);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_maxpool3x3s2(int dtype, const void* x, const void* dy, void* out, int B, int H, int W, int C, void* stream) {
    CVCL_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C > 0, "cvcl_maxpool3x3s2: bad args");
    CvclProfScope prof(stream, CVCL_K_MAXPOOL);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    if (!dy) {       // forward: out = pooled [B,Ho,Wo,C]
        if (dtype == CVCL_F32) hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for((long)B * Ho * Wo * C)), dim3(256), 0, s, (const float*)x, (float*)out, B, H, W, C);
        else hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid_for((long)B * Ho * Wo * C)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, B, H, W, C);
    } else {         // backward: out = dx [B,H,W,C]
        if (dtype == CVCL_F32) hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, s, (const float*)x, (const float*)dy, (float*)out, B, H, W, C);
        else hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, B, H, W, C);
    }
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_avgpool_bwd(int dtype, const float* d_pooled, void* dx, int B, int HW, int C, void* stream) {
    CVCL_CHECK_ARG(d_pooled && dx && B > 0 && HW > 0 && C > 0, "cvcl_avgpool_bwd: bad args");
    CvclProfScope prof(stream, CVCL_K_AVGPOOL);
    if (dtype == CVCL_F32) hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(grid_for((long)B * HW * C)), dim3(256), 0, (hipStream_t)stream, d_pooled, (float*)dx, B, HW, C);
    else hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(grid_for((long)B * HW * C)), dim3(256), 0, (hipStream_t)stream, d_pooled, (bf16_t*)dx, B, HW, C);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_zero_stuff2(int dtype, const void* dy, void* z, int B, int Ho, int Wo, int C, void* stream) {
    CVCL_CHECK_ARG(dy && z && B > 0 && Ho > 0 && Wo > 0 && C > 0, "cvcl_zero_stuff2: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    const long total = (long)B * 4 * Ho * Wo * C;
    if (dtype == CVCL_F32) hipLaunchKernelGGL(zero_stuff_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (float*)z, B, Ho, Wo, C);
    else hipLaunchKernelGGL(zero_stuff_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (bf16_t*)z, B, Ho, Wo, C);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_conv_wgrad_direct(int dtype, const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                                      int cin_per_group, int k, int stride, int pad, int x_is_nchw_f32, void* stream) {
    CVCL_CHECK_ARG(x && dy && dw && B > 0 && k > 0 && stride > 0 && cin_per_group > 0 && Cin % cin_per_group == 0 &&
                       Cout % (Cin / cin_per_group) == 0, "cvcl_conv_wgrad_direct: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const int nw = Cout * cin_per_group * k * k;
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(conv_wgrad_direct_kernel<float>, dim3(nw), dim3(256), 0, (hipStream_t)stream, x, (const float*)dy, dw, B, H, W,
                           Cin, Cout, cin_per_group, k, stride, pad, Ho, Wo, x_is_nchw_f32);
    else
        hipLaunchKernelGGL(conv_wgrad_direct_kernel<bf16_t>, dim3(nw), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)dy, dw, B, H, W,
                           Cin, Cout, cin_per_group, k, stride, pad, Ho, Wo, x_is_nchw_f32);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}
