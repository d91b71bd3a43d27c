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

// All elementwise / reduction kernels below work on 16-byte chunks (8 bf16 / 4 fp32 channels) with the channel chunk
// fixed per thread (grid * 256 is a multiple of the chunks per row), so per-channel vectors are loaded once per thread.

// ---- BatchNorm forward apply:  y = x * scale + shift (+ReLU) ---------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, T* __restrict__ y, long rows, int C,
                                                       int relu) {
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    const int CC = C / EPC;
    const long total = rows * CC;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(i0 % CC) * EPC;
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = scale[c + e]; sh[e] = shift[c + e]; }
    const float floor_ = relu ? 0.f : -INFINITY;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = i0; i < total; i += stride) {
        Chunk<T> a, o;
        a.load(x + i * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.set(e, fmaxf(fmaf(a.get(e), sc[e], sh[e]), floor_));
        o.store(y + i * EPC);
    }
}

// ---- BatchNorm backward ---------------------------------------------------------------------------------------------
// g = dy * mask, mask: mode 0 none | mode 1 [x*scale+shift > 0] (ReLU right after the BN, recomputed -- no read of y) |
// mode 2 [out > 0] with `out` the block output relu(bn(x) + identity).
// pass 1: per-channel partial sums of g and g * xhat, xhat = (x - mean) * rstd;  partial rows [gridDim.x][2][C]
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ out,
                                                            const T* __restrict__ dy, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, long rows, int C, int mode,
                                                            float* __restrict__ partial) {
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    __shared__ float red[256 * EPC * 2];
    const int CC = C / EPC;
    const int ccb = CC < 256 ? CC : 256;               // chunk columns handled by this block
    const int RL = 256 / ccb;                          // row lanes
    const int cc = blockIdx.y * ccb + threadIdx.x % ccb, rl = threadIdx.x / ccb;
    const int c = cc * EPC;
    float sc[EPC], sh[EPC], mu[EPC], rs[EPC], s[EPC], q[EPC];
    const float* scp = mode == 1 ? scale : mean;          // (unconditional 16-byte loads: see bn_bwd_apply_kernel)
    const float* shp = mode == 1 ? shift : mean;
#pragma unroll
    for (int v4 = 0; v4 < EPC / 4; ++v4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(scp + c + 4 * v4), b = *reinterpret_cast<const f32x4*>(shp + c + 4 * v4);
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(mean + c + 4 * v4), r4 = *reinterpret_cast<const f32x4*>(rstd + c + 4 * v4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[4 * v4 + e] = mode == 1 ? a[e] : 0.f; sh[4 * v4 + e] = mode == 1 ? b[e] : 0.f;
            mu[4 * v4 + e] = m4[e]; rs[4 * v4 + e] = r4[e]; s[4 * v4 + e] = 0.f; q[4 * v4 + e] = 0.f;
        }
    }
    // four rows' loads (clamped addresses: always valid, no branch around a load) are in flight before the first is used; the rows
    // are then accumulated in the order of the one-row-at-a-time loop (same sums, bit for bit).  One row per iteration kept ~6 MB in
    // flight on the whole chip: the pass ran at half the streaming rate
    constexpr int UN = 4;
    const long rstep = (long)gridDim.x * RL;
    for (long r = (long)blockIdx.x * RL + rl; r < rows; r += UN * rstep) {
        Chunk<T> xv[UN], gv[UN], ov[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long rr = r + u * rstep;
            const long off = (rr < rows ? rr : rows - 1) * C + c;
            xv[u].load(x + off);
            gv[u].load(dy + off);
            if (mode == 2) ov[u].load(out + off);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (r + u * rstep < rows) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float xe = xv[u].get(e);
                    float g = gv[u].get(e);
                    if (mode == 1 && !(fmaf(xe, sc[e], sh[e]) > 0.f)) g = 0.f;
                    if (mode == 2 && !(ov[u].get(e) > 0.f)) g = 0.f;
                    s[e] += g;
                    q[e] = fmaf(g, (xe - mu[e]) * rs[e], q[e]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        red[(threadIdx.x * EPC + e) * 2 + 0] = s[e];
        red[(threadIdx.x * EPC + e) * 2 + 1] = q[e];
    }
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float ts = 0.f, tq = 0.f;
            for (int l = 0; l < RL; ++l) {
                ts += red[((l * ccb + threadIdx.x) * EPC + e) * 2 + 0];
                tq += red[((l * ccb + threadIdx.x) * EPC + e) * 2 + 1];
            }
            partial[((long)blockIdx.x * 2 + 0) * C + c + e] = ts;
            partial[((long)blockIdx.x * 2 + 1) * C + c + e] = tq;
        }
    }
}

// reduce the partial rows (fp64, fixed order): dbeta = sum g, dgamma = sum g*xhat, and the apply-pass coefficients
//   dx = gamma*rstd*(g - dbeta/n - xhat*dgamma/n) = k1*g + k2*x + k3
// 16 channels x 64 row slices per workgroup (as bn_finalize_kernel: the kernel is nothing but dependent-load latency -- with 32
// channels x 8 slices a 512-row partial image took eight round trips per thread and a 64-channel layer ran on two workgroups: 15 us)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int prow, int C, long n,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ coef) {
    constexpr int NS = 64, NC = 16;
    __shared__ double ss[NS][NC], sq[NS][NC];
    const int cl = threadIdx.x & (NC - 1), sl = threadIdx.x / NC, ch = blockIdx.x * NC + cl;
    double s = 0.0, q = 0.0;
    if (ch < C && sl < prow) {
        const int cnt = (prow - sl + NS - 1) / NS;
        s = ordered_sum<8, double>(cnt, [&](int j) { return partial[((long)(sl + NS * j) * 2 + 0) * C + ch]; });
        q = ordered_sum<8, double>(cnt, [&](int j) { return partial[((long)(sl + NS * j) * 2 + 1) * C + ch]; });
    }
    ss[sl][cl] = s;
    sq[sl][cl] = q;
    __syncthreads();
    if (sl == 0 && ch < C) {
        for (int l = 1; l < NS; ++l) { s += ss[l][cl]; q += sq[l][cl]; }
        const float db = (float)s, dg = (float)q;
        dbeta[ch] = db;
        dgamma[ch] = dg;
        const float inv_n = 1.f / (float)n, gr = gamma[ch] * rstd[ch];
        const float k2 = -gr * rstd[ch] * dg * inv_n;
        coef[ch] = gr;
        coef[C + ch] = k2;
        coef[2 * C + ch] = -gr * db * inv_n - k2 * mean[ch];
    }
}

// pass 2: dx = k1*g + k2*x + k3; optionally also stores g (the gradient of the residual identity, mode 2)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ out,
                                                           const T* __restrict__ dy, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ coef,
                                                           T* __restrict__ dx, T* __restrict__ g_out, long rows, int C, int mode) {
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    const int CC = C / EPC;
    const long total = rows * CC;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(i0 % CC) * EPC;
    // per-channel coefficients as 16-byte loads from unconditional addresses (a `mode == 1 ? scale[c + e] : 0.f` per element made the
    // compiler wait for every channel's load in turn: 40 dependent L2 round trips in front of four chunks of work -- the pass ran at
    // 2.7 TB/s on the 128-channel layers)
    float sc[EPC], sh[EPC], k1[EPC], k2[EPC], k3[EPC];
    const float* scp = mode == 1 ? scale : coef;
    const float* shp = mode == 1 ? shift : coef;
#pragma unroll
    for (int v4 = 0; v4 < EPC / 4; ++v4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(scp + c + 4 * v4), b = *reinterpret_cast<const f32x4*>(shp + c + 4 * v4);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(coef + c + 4 * v4), c2 = *reinterpret_cast<const f32x4*>(coef + C + c + 4 * v4);
        const f32x4 c3 = *reinterpret_cast<const f32x4*>(coef + 2 * C + c + 4 * v4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[4 * v4 + e] = mode == 1 ? a[e] : 0.f; sh[4 * v4 + e] = mode == 1 ? b[e] : 0.f;
            k1[4 * v4 + e] = c1[e]; k2[4 * v4 + e] = c2[e]; k3[4 * v4 + e] = c3[e];
        }
    }
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = i0; i < total; i += 2 * stride) {          // two independent chunk sets in flight per thread (as in bn_add_relu)
        const long j = i + stride;
        const bool has_j = j < total;
        const long jj = has_j ? j : i;                        // (clamped: the loads are unconditional)
        Chunk<T> xv[2], gv[2], ov[2], o, go;
        xv[0].load(x + i * EPC);
        gv[0].load(dy + i * EPC);
        if (mode == 2) ov[0].load(out + i * EPC);
        xv[1].load(x + jj * EPC);
        gv[1].load(dy + jj * EPC);
        if (mode == 2) ov[1].load(out + jj * EPC);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !has_j) break;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float xe = xv[u].get(e);
                float g = gv[u].get(e);
                if (mode == 1 && !(fmaf(xe, sc[e], sh[e]) > 0.f)) g = 0.f;
                if (mode == 2 && !(ov[u].get(e) > 0.f)) g = 0.f;
                go.set(e, g);
                o.set(e, fmaf(k1[e], g, fmaf(k2[e], xe, k3[e])));
            }
            const long k = u ? j : i;
            o.store(dx + k * EPC);
            if (g_out) go.store(g_out + k * EPC);
        }
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

// ---- y = a + b (optionally ReLU), dx = dy * [y > 0]: 16-byte chunks; n is a multiple of the chunk size -----------------
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long n, int relu) {
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    const float floor_ = relu ? 0.f : -INFINITY;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n / EPC; i += (long)gridDim.x * blockDim.x) {
        Chunk<T> av, bv, o;
        av.load(a + i * EPC);
        bv.load(b + i * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.set(e, fmaxf(av.get(e) + bv.get(e), floor_));
        o.store(y + i * EPC);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void relu_mask_kernel(const T* y, const T* dy, T* dx, long n) {
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n / EPC; i += (long)gridDim.x * blockDim.x) {
        Chunk<T> yv, gv, o;
        yv.load(y + i * EPC);
        gv.load(dy + i * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.set(e, yv.get(e) > 0.f ? gv.get(e) : 0.f);
        o.store(dx + i * EPC);
    }
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
// dy[window] where `in` is that window's first arg-max (row-major scan order, like torch).  One 16-byte channel chunk
// per thread; the 3x3 neighbourhoods of the candidate windows overlap in a 5x5 patch that is loaded once.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                          int B, int H, int W, int C) {
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, CC = C / EPC;
    const long total = (long)B * H * W * CC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CC) * EPC;
        const long p = i / CC;
        const int xi = (int)(p % W), yi = (int)((p / W) % H), b = (int)(p / ((long)W * H));
        Chunk<T> v, o;
        v.load(x + p * C + c);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        for (int oy = (yi + 1) / 2 - 1; oy <= (yi + 1) / 2; ++oy) {         // windows with 2*oy - 1 <= yi <= 2*oy + 1
            if (oy < 0 || oy >= Ho || 2 * oy - 1 > yi || 2 * oy + 1 < yi) continue;
            for (int ox = (xi + 1) / 2 - 1; ox <= (xi + 1) / 2; ++ox) {
                if (ox < 0 || ox >= Wo || 2 * ox - 1 > xi || 2 * ox + 1 < xi) continue;
                bool first[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) first[e] = true;
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = 2 * oy - 1 + ky;
                    if (yy < 0 || yy >= H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = 2 * ox - 1 + kx;
                        if (xx < 0 || xx >= W || (yy == yi && xx == xi)) continue;
                        Chunk<T> u;
                        u.load(x + (((long)b * H + yy) * W + xx) * C + c);
                        const bool before = (yy < yi) || (yy == yi && xx < xi);
#pragma unroll
                        for (int e = 0; e < EPC; ++e)
                            if (u.get(e) > v.get(e) || (before && u.get(e) == v.get(e))) first[e] = false;
                    }
                }
                Chunk<T> g;
                g.load(dy + (((long)b * Ho + oy) * Wo + ox) * C + c);
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (first[e]) acc[e] += g.get(e);
            }
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.set(e, acc[e]);
        o.store(dx + p * C + c);
    }
}

// max pool with recorded arg-max (first maximum in row-major window order, like torch): the backward then needs, per input
// pixel, the <= 4 covering windows' index bytes and gradients instead of re-scanning 4 x 9 inputs
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_idx_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                              int B, int H, int W, int C) {
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, CC = C / EPC;
    const long total = (long)B * Ho * Wo * CC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CC) * EPC;
        const long p = i / CC;
        const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), b = (int)(p / ((long)Wo * Ho));
        float m[EPC];
        uint8_t k[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) { m[e] = -INFINITY; k[e] = 0; }
        for (int ky = 0; ky < 3; ++ky) {
            const int yin = 2 * oy - 1 + ky;
            if (yin < 0 || yin >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xin = 2 * ox - 1 + kx;
                if (xin < 0 || xin >= W) continue;
                Chunk<T> v;
                v.load(x + (((long)b * H + yin) * W + xin) * C + c);
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (v.get(e) > m[e]) { m[e] = v.get(e); k[e] = (uint8_t)(ky * 3 + kx); }
            }
        }
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) { o.set(e, m[e]); idx[p * C + c + e] = k[e]; }
        o.store(y + p * C + c);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const uint8_t* __restrict__ idx, const T* __restrict__ dy,
                                                              T* __restrict__ dx, int B, int H, int W, int C) {
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, CC = C / EPC;
    const long total = (long)B * H * W * CC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CC) * EPC;
        const long p = i / CC;
        const int xi = (int)(p % W), yi = (int)((p / W) % H), b = (int)(p / ((long)W * H));
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        for (int oy = (yi + 1) / 2 - 1; oy <= (yi + 1) / 2; ++oy) {
            if (oy < 0 || oy >= Ho || 2 * oy - 1 > yi || 2 * oy + 1 < yi) continue;
            for (int ox = (xi + 1) / 2 - 1; ox <= (xi + 1) / 2; ++ox) {
                if (ox < 0 || ox >= Wo || 2 * ox - 1 > xi || 2 * ox + 1 < xi) continue;
                const int code = (yi - (2 * oy - 1)) * 3 + (xi - (2 * ox - 1));
                const long q = (((long)b * Ho + oy) * Wo + ox) * C + c;
                Chunk<T> g;
                g.load(dy + q);
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (idx[q + e] == code) acc[e] += g.get(e);
            }
        }
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.set(e, acc[e]);
        o.store(dx + p * C + c);
    }
}

// ---- bf16 -> f32 copy (spatial head: the per-location features leave the bf16 trunk for the fp32 projection) -----------
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 8; i += (long)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
        *reinterpret_cast<f32x4*>(y + i * 8) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        *reinterpret_cast<f32x4*>(y + i * 8 + 4) = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 8; i += (long)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8), b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
        *reinterpret_cast<bf16x8*>(y + i * 8) = bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3],
                                                       (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
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
    constexpr int EPC = ElemTraits<T>::kPerChunk;
    const int H = 2 * Ho, W = 2 * Wo, CC = C / EPC;
    const long total = (long)B * H * W * CC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CC) * EPC;
        const long p = i / CC;
        const int xi = (int)(p % W), yi = (int)((p / W) % H), b = (int)(p / ((long)W * H));
        Chunk<T> v;
        if (!(yi & 1) && !(xi & 1)) v.load(dy + (((long)b * Ho + (yi >> 1)) * Wo + (xi >> 1)) * C + c);
        else v.zero();
        v.store(z + p * C + c);
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

// batch mean / rstd from the forward statistics rows [prow][2][C] (fp64, biased variance: what bn_finalize normalised with)
__global__ __launch_bounds__(256) void bn_moments_kernel(const float* __restrict__ stats, int prow, long count, float eps, int C,
                                                         float* __restrict__ mean, float* __restrict__ rstd,
                                                         float* __restrict__ centre_track) {
    __shared__ double ss[8][32], sq[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5, ch = blockIdx.x * 32 + cl;
    double s = 0.0, q = 0.0;
    if (ch < C && sl < prow) {
        const int cnt = (prow - sl + 7) / 8;
        s = ordered_sum<8, double>(cnt, [&](int j) { return stats[((long)(sl + 8 * j) * 2 + 0) * C + ch]; });
        q = ordered_sum<8, double>(cnt, [&](int j) { return stats[((long)(sl + 8 * j) * 2 + 1) * C + ch]; });
    }
    ss[sl][cl] = s;
    sq[sl][cl] = q;
    __syncthreads();
    if (sl == 0 && ch < C) {
        for (int l = 1; l < 8; ++l) { s += ss[l][cl]; q += sq[l][cl]; }
        const double m = s / (double)count;
        double var = q / (double)count - m * m;
        if (var < 0.0) var = 0.0;
        mean[ch] = (float)m;
        rstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
        // the statistics are those of the tensor stored as y - c: c + m is the batch mean of y, the next step's storage centre
        if (centre_track) centre_track[ch] = (float)((double)centre_track[ch] + m);
    }
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

namespace {
// grid for the chunk kernels whose threads keep a fixed channel chunk: (grid * 256) % chunks_per_row == 0
int chunk_grid(long total_chunks, int cc, int per_thread) {
    int mult = 1;
    while ((mult * 256) % cc) ++mult;
    long g = (total_chunks + 256L * per_thread - 1) / (256L * per_thread);
    if (g < 1) g = 1;
    if (g > 16384) g = 16384;
    return (int)((g + mult - 1) / mult * mult);
}
int epc_of(int dtype) { return dtype == CVCL_BF16 ? 8 : 4; }
int bn_bwd_rows(int dtype, long rows, int C) {
    const int cc = C / epc_of(dtype), ccb = cc < 256 ? cc : 256, rl = 256 / ccb, ny = cc / ccb;
    long g = (rows + (long)rl * 8 - 1) / ((long)rl * 8);           // >= 8 rows per row lane
    const long cap = 512 / ny > 0 ? 512 / ny : 1;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}
}  // namespace

extern "C" int cvcl_bn_apply(int dtype, const void* x, const float* scale, const float* shift, void* y, long rows, int C, int relu,
                             void* stream) {
    CVCL_CHECK_ARG(x && scale && shift && y && rows > 0 && C > 0 && C % epc_of(dtype) == 0, "cvcl_bn_apply: bad args");
    const int cc = C / epc_of(dtype);
    CVCL_CHECK_ARG((cc & (cc - 1)) == 0 || 256 % cc == 0 || cc % 256 == 0, "cvcl_bn_apply: unsupported channel count %d", C);
    CvclProfScope prof(stream, CVCL_K_BN_APPLY);
    const int grid = chunk_grid(rows * cc, cc, 4);
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, scale, shift,
                           (float*)y, rows, C, relu);
    else
        hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scale, shift,
                           (bf16_t*)y, rows, C, relu);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_bn_bwd_partial_rows(int dtype, long rows, int C) { return bn_bwd_rows(dtype, rows, C); }

extern "C" int cvcl_bn_bwd(int dtype, int mode, const void* x, const void* out, const void* dy, const float* scale,
                           const float* shift, const float* mean, const float* rstd, const float* gamma, float* dgamma,
                           float* dbeta, void* dx, void* g_out, long rows, int C, float* partial, int partial_rows, float* coef,
                           void* stream) {
    CVCL_CHECK_ARG(x && dy && mean && rstd && gamma && dgamma && dbeta && dx && partial && coef && rows > 0 && C > 0,
                   "cvcl_bn_bwd: null pointer");
    CVCL_CHECK_ARG(mode >= 0 && mode <= 2 && (mode != 1 || (scale && shift)) && (mode != 2 || out), "cvcl_bn_bwd: bad mode %d", mode);
    const int epc = epc_of(dtype), cc = C / epc;
    CVCL_CHECK_ARG(C % epc == 0 && (cc & (cc - 1)) == 0, "cvcl_bn_bwd: channel count %d must be a power of two >= %d", C, epc);
    const int g = bn_bwd_rows(dtype, rows, C);
    CVCL_CHECK_ARG(partial_rows >= g, "cvcl_bn_bwd: partial_rows %d < %d", partial_rows, g);
    CVCL_CHECK_ARG((((uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)coef) & 15) == 0,
                   "cvcl_bn_bwd: scale / shift / mean / rstd / coef must be 16-byte aligned");
    CvclProfScope prof(stream, CVCL_K_BN_BWD);
    hipStream_t s = (hipStream_t)stream;
    const int ccb = cc < 256 ? cc : 256;
    dim3 rgrid(g, cc / ccb);
    const int agrid = chunk_grid(rows * cc, cc, 4);
    if (dtype == CVCL_F32) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, rgrid, dim3(256), 0, s, (const float*)x, (const float*)out, (const float*)dy,
                           scale, shift, mean, rstd, rows, C, mode, partial);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cvcl_div_up(C, 16)), dim3(1024), 0, s, partial, g, C, rows, mean, rstd, gamma,
                           dgamma, dbeta, coef);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(agrid), dim3(256), 0, s, (const float*)x, (const float*)out,
                           (const float*)dy, scale, shift, coef, (float*)dx, (float*)g_out, rows, C, mode);
    } else {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, rgrid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)out,
                           (const bf16_t*)dy, scale, shift, mean, rstd, rows, C, mode, partial);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cvcl_div_up(C, 16)), dim3(1024), 0, s, partial, g, C, rows, mean, rstd, gamma,
                           dgamma, dbeta, coef);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(agrid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)out,
                           (const bf16_t*)dy, scale, shift, coef, (bf16_t*)dx, (bf16_t*)g_out, rows, C, mode);
    }
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_bn_batch_moments(const float* stats, int stats_rows, long count, float eps, float* mean, float* rstd,
                                     float* centre_track, int C, void* stream) {
    CVCL_CHECK_ARG(stats && mean && rstd && stats_rows > 0 && count > 0 && C > 0, "cvcl_bn_batch_moments: bad args");
    CvclProfScope prof(stream, CVCL_K_BN_FINALIZE);
    hipLaunchKernelGGL(bn_moments_kernel, dim3(cvcl_div_up(C, 32)), dim3(256), 0, (hipStream_t)stream, stats, stats_rows, count, eps, C,
                       mean, rstd, centre_track);
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

extern "C" int cvcl_add(int dtype, const void* a, const void* b, void* y, long n, int relu, void* stream) {
    CVCL_CHECK_ARG(a && b && y && n > 0 && n % epc_of(dtype) == 0, "cvcl_add: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(add_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)y, n, relu);
    else
        hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n, relu);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_relu_mask(int dtype, const void* y, const void* dy, void* dx, long n, void* stream) {
    CVCL_CHECK_ARG(y && dy && dx && n > 0 && n % epc_of(dtype) == 0, "cvcl_relu_mask: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    if (dtype == CVCL_F32)
        hipLaunchKernelGGL(relu_mask_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)y, (const float*)dy, (float*)dx, n);
    else
        hipLaunchKernelGGL(relu_mask_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (const bf16_t*)dy, (bf16_t*)dx, n);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_maxpool3x3s2(int dtype, const void* x, const void* dy, void* out, int B, int H, int W, int C, void* stream) {
    CVCL_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C > 0 && C % epc_of(dtype) == 0, "cvcl_maxpool3x3s2: bad args");
    CvclProfScope prof(stream, CVCL_K_MAXPOOL);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    if (!dy) {       // forward: out = pooled [B,Ho,Wo,C]
        if (dtype == CVCL_F32) hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for((long)B * Ho * Wo * C)), dim3(256), 0, s, (const float*)x, (float*)out, B, H, W, C);
        else hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid_for((long)B * Ho * Wo * C)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, B, H, W, C);
    } else {         // backward: out = dx [B,H,W,C]
        if (dtype == CVCL_F32) hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for((long)B * H * W * C / 4)), dim3(256), 0, s, (const float*)x, (const float*)dy, (float*)out, B, H, W, C);
        else hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid_for((long)B * H * W * C / 8)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, B, H, W, C);
    }
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_maxpool3x3s2_idx(int dtype, const void* x, const void* dy, void* out, uint8_t* idx, int B, int H, int W, int C,
                                     void* stream) {
    CVCL_CHECK_ARG(out && idx && (x || dy) && B > 0 && H > 0 && W > 0 && C > 0 && C % epc_of(dtype) == 0, "cvcl_maxpool3x3s2_idx: bad args");
    CvclProfScope prof(stream, CVCL_K_MAXPOOL);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    const int epc = epc_of(dtype);
    if (!dy) {       // forward: out = pooled [B,Ho,Wo,C], idx = arg-max code 0..8 per output element
        const int g = grid_for((long)B * Ho * Wo * C / epc);
        if (dtype == CVCL_F32) hipLaunchKernelGGL(maxpool_fwd_idx_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, (float*)out, idx, B, H, W, C);
        else hipLaunchKernelGGL(maxpool_fwd_idx_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, idx, B, H, W, C);
    } else {         // backward: out = dx [B,H,W,C]
        const int g = grid_for((long)B * H * W * C / epc);
        if (dtype == CVCL_F32) hipLaunchKernelGGL(maxpool_bwd_idx_kernel<float>, dim3(g), dim3(256), 0, s, idx, (const float*)dy, (float*)out, B, H, W, C);
        else hipLaunchKernelGGL(maxpool_bwd_idx_kernel<bf16_t>, dim3(g), dim3(256), 0, s, idx, (const bf16_t*)dy, (bf16_t*)out, B, H, W, C);
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
    CVCL_CHECK_ARG(dy && z && B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % epc_of(dtype) == 0, "cvcl_zero_stuff2: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    const long total = (long)B * 4 * Ho * Wo * C / epc_of(dtype);
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

extern "C" int cvcl_bf16_to_f32(const void* x, float* y, long n, void* stream) {
    CVCL_CHECK_ARG(x && y && n > 0 && n % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "cvcl_bf16_to_f32: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, n);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_f32_to_bf16(const float* x, void* y, long n, void* stream) {
    CVCL_CHECK_ARG(x && y && n > 0 && n % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "cvcl_f32_to_bf16: bad args");
    CvclProfScope prof(stream, CVCL_K_OTHER);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}
