// Contrastive head kernels: embedding mean-pool, L2 normalise, similarity logits, symmetric InfoNCE.
// All fp32 (the reference computes these in fp32; they are latency/HBM-bound scans, not GEMM work,
// except the similarity contraction which runs on the exact-fp32 MFMA GEMM of gemm.hip).
#include "cvcl_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// embedding gather + mean-pool                                   multimodal/multimodal.py:496-503
// one workgroup per utterance; lanes stride over E so every table row is read as one coalesced burst
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void embed_meanpool_fwd_kernel(const float* __restrict__ table,
                                                                 const int64_t* __restrict__ tok,
                                                                 const int64_t* __restrict__ len,
                                                                 float* __restrict__ ret, float* __restrict__ out,
                                                                 int L, int E, int V) {
    const int b = blockIdx.x;
    const float inv_den = (float)len[b];
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float acc = 0.f;
        for (int l = 0; l < L; ++l) {
            const int64_t t = tok[(long)b * L + l];
            const float v = (t >= 0 && t < V) ? table[t * E + e] : NAN;
            if (out) out[((long)b * L + l) * E + e] = v;
            acc += v;
        }
        ret[(long)b * E + e] = acc / inv_den;
    }
}

// one workgroup per vocabulary row.  The B*L token ids are staged into LDS (independent coalesced loads) and scanned 512 at a
// time into an ORDERED match list ((b,l) order: per-wave ballots + a prefix over the 8 waves); the matches are then dealt
// round-robin to 16 groups of 32 threads, each thread owning float4 columns of E, four matches' operands in flight per group --
// <sos> / <eos> occur in every utterance (256 matches at B = 256): 16 sequential accumulations per group instead of the 80 of the
// round-4 form (4 groups, batches of 8 behind two dependent loads each: 61 us) -- and the 16 partial sums are combined in a fixed
// order.  Deterministic, no atomics; writes every row (zeros where the word does not occur; row 0 = padding_idx gets no gradient).
// d_ret[b] / len[b] is accumulated as d_ret[b] * (1 / len[b]) (one division per match instead of one per element: <= 1 ulp).
constexpr int EMB_CHUNK = 4096;
static_assert(EMB_CHUNK <= 65536, "smatch holds chunk-relative positions as 16-bit values");
constexpr int EMB_GROUPS = 16;
constexpr int EMB_PASS = 512;                      // floats of E per pass (the groups' partial sums: 16 x 512 floats of LDS)
// MEANPOOL: the source row of position p is utterance p / L, scaled by 1 / len (embedding mean-pool, multimodal.py:496-503);
// otherwise position p itself, unscaled (cvcl_embed_rows_bwd: the per-word embedding gather of the LSTM / transformer text encoders).
template <bool MEANPOOL>
__global__ __launch_bounds__(512) void embed_meanpool_bwd_kernel(const float* __restrict__ d_ret,
                                                                 const int64_t* __restrict__ tok,
                                                                 const int64_t* __restrict__ len,
                                                                 float* __restrict__ d_table, int B, int L, int E) {
    __shared__ int stok[EMB_CHUNK];
    __shared__ unsigned short smatch[EMB_CHUNK];            // chunk-relative positions (0..4095): 16 + 8 + 32 KiB of LDS in all
    __shared__ int wcount[8];
    __shared__ __attribute__((aligned(16))) float part[EMB_GROUPS][EMB_PASS];
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = tid >> 5, t32 = tid & 31;
    const int total = B * L;
    const bool vec = (E & 3) == 0 && (((uintptr_t)d_ret) & 15) == 0;
    for (int ebase = 0; ebase < E; ebase += EMB_PASS) {
        f32x4 acc[EMB_PASS / 128];
#pragma unroll
        for (int i = 0; i < EMB_PASS / 128; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (v != 0) {
            for (int c0 = 0; c0 < total; c0 += EMB_CHUNK) {
                const int n = min(EMB_CHUNK, total - c0);
                __syncthreads();
                for (int i = tid; i < n; i += 512) stok[i] = (int)tok[c0 + i];
                __syncthreads();
                int nm = 0;                                             // matches so far (uniform)
                for (int base = 0; base < n; base += 512) {
                    const bool hit = (base + tid) < n && stok[base + tid] == v;
                    const unsigned long long mm = __ballot(hit);
                    if (lane == 0) wcount[wave] = __popcll(mm);
                    __syncthreads();
                    int before = nm, all = nm;
#pragma unroll
                    for (int w = 0; w < 8; ++w) { const int c = wcount[w]; if (w < wave) before += c; all += c; }
                    if (hit) smatch[before + __popcll(mm & ((1ull << lane) - 1ull))] = (unsigned short)(base + tid);
                    nm = all;
                    __syncthreads();
                }
                // group grp: matches grp, grp + 16, ... in order, four at a time
                for (int j0 = grp; j0 < nm; j0 += 4 * EMB_GROUPS) {
                    int bb[4];
                    float rden[4];
                    f32x4 val[4][EMB_PASS / 128];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int j = j0 + k * EMB_GROUPS;
                        bb[k] = j < nm ? (MEANPOOL ? (c0 + smatch[j]) / L : c0 + smatch[j]) : -1;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        rden[k] = (MEANPOOL && bb[k] >= 0) ? (float)len[bb[k]] : 1.f;
#pragma unroll
                        for (int i = 0; i < EMB_PASS / 128; ++i) {
                            const int e = ebase + (t32 + 32 * i) * 4;
                            val[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (bb[k] >= 0 && e < E) {
                                const float* src = d_ret + (long)bb[k] * E + e;
                                if (vec) val[k][i] = *reinterpret_cast<const f32x4*>(src);
                                else {
#pragma unroll
                                    for (int q = 0; q < 4; ++q) val[k][i][q] = e + q < E ? src[q] : 0.f;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if constexpr (MEANPOOL) {
                            const float r = 1.f / rden[k];
#pragma unroll
                            for (int i = 0; i < EMB_PASS / 128; ++i) acc[i] += val[k][i] * r;   // (padding slots add 0 * 1)
                        } else {
#pragma unroll
                            for (int i = 0; i < EMB_PASS / 128; ++i) acc[i] += val[k][i];
                        }
                    }
                }
            }
        }
        // combine the groups' partial sums in a fixed order (deterministic)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EMB_PASS / 128; ++i) *reinterpret_cast<f32x4*>(&part[grp][(t32 + 32 * i) * 4]) = acc[i];
        __syncthreads();
        {
            const int e = ebase + tid;
            if (e < E) {
                float t = 0.f;
#pragma unroll
                for (int g = 0; g < EMB_GROUPS; ++g) t += part[g][tid];
                d_table[(long)v * E + e] = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// F.normalize rows                                               multimodal/multimodal.py:736,743
// one wave per row, 4 rows per workgroup
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ norm, int N, int E, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* xr = x + (long)row * E;
    float ss = 0.f;
    for (int e = lane; e < E; e += 64) ss = fmaf(xr[e], xr[e], ss);
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float den = fmaxf(nrm, eps);
    for (int e = lane; e < E; e += 64) y[(long)row * E + e] = xr[e] / den;
    if (lane == 0) norm[row] = nrm;
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ norm,
                                                         const float* __restrict__ dy, float* __restrict__ dx,
                                                         int N, int E, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* yr = y + (long)row * E;
    const float* gr = dy + (long)row * E;
    float dot = 0.f;
    for (int e = lane; e < E; e += 64) dot = fmaf(yr[e], gr[e], dot);
    dot = wave_sum(dot);
    const float nrm = norm[row];
    if (nrm < eps) {                 // clamp_min active: y = x / eps, no projection term
        for (int e = lane; e < E; e += 64) dx[(long)row * E + e] = gr[e] / eps;
    } else {
        for (int e = lane; e < E; e += 64) dx[(long)row * E + e] = (gr[e] - yr[e] * dot) / nrm;
    }
}

// ---------------------------------------------------------------------------------------------
// symmetric InfoNCE                        multimodal/multimodal.py:801-818, multimodal/utils.py:106-108
// rows: one wave per row.  columns: 64 columns x 16 row-slices per workgroup, online softmax per slice.
// per-row/column results: ce = lse - diag, hit = (argmax == index), entropy = lse - sum p x.
// ---------------------------------------------------------------------------------------------
struct OnlineSm {                    // running max m, sum exp(x-m), sum x exp(x-m), argmax
    float m, s, t;
    int arg;
    __device__ inline void init() { m = -INFINITY; s = 0.f; t = 0.f; arg = 0x7fffffff; }
    __device__ inline void push(float x, int idx) {
        if (x > m) {
            const float r = expf(m - x);          // exp(-inf) = 0 on the first element
            s = s * r + 1.f;
            t = t * r + x;
            m = x;
            arg = idx;
        } else {
            const float e = expf(x - m);
            s += e;
            t = fmaf(x, e, t);
        }
    }
    __device__ inline void merge(float m2, float s2, float t2, int arg2) {
        if (m2 > m || (m2 == m && arg2 < arg)) arg = arg2;
        const float mm = fmaxf(m, m2);
        if (mm == -INFINITY) return;
        const float r1 = expf(m - mm), r2 = expf(m2 - mm);
        s = s * r1 + s2 * r2;
        t = t * r1 + t2 * r2;
        m = mm;
    }
};

__global__ __launch_bounds__(256) void infonce_rows_kernel(const float* __restrict__ logits, int N,
                                                           float* __restrict__ row_lse, float* __restrict__ ws) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* xr = logits + (long)row * N;
    OnlineSm st;
    st.init();
    for (int j = lane; j < N; j += 64) st.push(xr[j], j);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(st.m, o, 64), s2 = __shfl_xor(st.s, o, 64), t2 = __shfl_xor(st.t, o, 64);
        const int a2 = __shfl_xor(st.arg, o, 64);
        st.merge(m2, s2, t2, a2);
    }
    if (lane == 0) {
        const float lse = st.m + logf(st.s);
        row_lse[row] = lse;
        ws[0 * N + row] = lse - xr[row];                 // cross entropy of this row
        ws[1 * N + row] = (st.arg == row) ? 1.f : 0.f;   // argmax hit
        ws[2 * N + row] = lse - st.t / st.s;             // entropy
    }
}

__global__ __launch_bounds__(1024) void infonce_cols_kernel(const float* __restrict__ logits, int N,
                                                            float* __restrict__ col_lse, float* __restrict__ ws) {
    __shared__ float sm_m[16][64], sm_s[16][64], sm_t[16][64];
    __shared__ int sm_a[16][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6, col = blockIdx.x * 64 + c;
    OnlineSm st;
    st.init();
    if (col < N)
        for (int r = sl; r < N; r += 16) st.push(logits[(long)r * N + col], r);
    sm_m[sl][c] = st.m; sm_s[sl][c] = st.s; sm_t[sl][c] = st.t; sm_a[sl][c] = st.arg;
    __syncthreads();
    if (sl == 0 && col < N) {
        for (int i = 1; i < 16; ++i) st.merge(sm_m[i][c], sm_s[i][c], sm_t[i][c], sm_a[i][c]);
        const float lse = st.m + logf(st.s);
        col_lse[col] = lse;
        ws[3 * N + col] = lse - logits[(long)col * N + col];
        ws[4 * N + col] = (st.arg == col) ? 1.f : 0.f;
        ws[5 * N + col] = lse - st.t / st.s;
    }
}

// out[r] = entropy of softmax(x[r, :])  (get_entropy, reference multimodal/utils.py:106-108); one wave per row
__global__ __launch_bounds__(256) void row_entropy_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int N) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    OnlineSm st;
    st.init();
    for (int j = lane; j < N; j += 64) st.push(x[(long)row * N + j], j);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(st.m, o, 64), s2 = __shfl_xor(st.s, o, 64), t2 = __shfl_xor(st.t, o, 64);
        const int a2 = __shfl_xor(st.arg, o, 64);
        st.merge(m2, s2, t2, a2);
    }
    if (lane == 0) out[row] = st.m + logf(st.s) - st.t / st.s;
}

// scalars = {infonce, image_accuracy, text_accuracy, image_entropy, text_entropy}; fixed summation order
__global__ __launch_bounds__(256) void infonce_finalize_kernel(const float* __restrict__ ws, int N,
                                                               float* __restrict__ scalars) {
    __shared__ float scratch[8];
    float a[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float v = 0.f;
        for (int i = threadIdx.x; i < N; i += blockDim.x) v += ws[k * N + i];
        a[k] = block_sum(v, scratch);
    }
    if (threadIdx.x == 0) {
        const float n = (float)N;
        scalars[0] = (a[0] / n + a[3] / n) / 2.f;
        scalars[1] = a[1] / n;
        scalars[2] = a[4] / n;
        scalars[3] = a[2] / n;
        scalars[4] = a[5] / n;
    }
}

__global__ __launch_bounds__(256) void infonce_bwd_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ row_lse,
                                                          const float* __restrict__ col_lse,
                                                          const float* __restrict__ d_loss,
                                                          float* __restrict__ d_logits, int N) {
    const float g = *d_loss / (2.f * (float)N);
    const long total = (long)N * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i - (long)r * N);
        const float x = logits[i];
        float v = expf(x - row_lse[r]) + expf(x - col_lse[c]);
        if (r == c) v -= 2.f;
        d_logits[i] = g * v;
    }
}

// partial[b] = sum over a grid-strided slice of a[i]*b[i]; final = fixed-order sum of partials
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          long n, float* __restrict__ partial) {
    __shared__ float scratch[8];
    float v = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        v = fmaf(a[i], b[i], v);
    v = block_sum(v, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = v;
}
__global__ __launch_bounds__(256) void sum_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ float scratch[8];
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += partial[i];
    v = block_sum(v, scratch);
    if (threadIdx.x == 0) *out = v;
}

}  // namespace

// ================================================================================================
extern "C" int cvcl_embed_meanpool_fwd(const float* table, const int64_t* tok, const int64_t* len, float* ret,
                                       float* out_ble, int B, int L, int E, int V, void* stream) {
    CVCL_CHECK_ARG(table && tok && len && ret, "cvcl_embed_meanpool_fwd: null pointer");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    CVCL_CHECK_ARG(B > 0 && L > 0 && E > 0 && V > 0, "cvcl_embed_meanpool_fwd: bad shape");
    hipLaunchKernelGGL(embed_meanpool_fwd_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, table, tok, len, ret,
                       out_ble, L, E, V);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

// ---------------------------------------------------------------------------------------------
// embedding_type == "spatial", sim == "max" (reference multimodal/multimodal.py:770-780): the match map
// mm[(i,p)][(t,l)] = <image location p of image i, word l of utterance t> comes from one fp32 GEMM; this kernel takes the
// best location per word, sums over ALL L positions and divides by the utterance length:
//   logits[i][t] = exp(neg_log_temp) * sum_l max_p mm[(i,p)][(t,l)] / len[t];   arg[i][(t,l)] = that location (first max)
// one workgroup per image i; thread per (t,l) column (coalesced over columns), then thread per t.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spatial_max_fwd_kernel(const float* __restrict__ mm, const int64_t* __restrict__ len,
                                                              const float* __restrict__ neg_log_temp, float* __restrict__ logits,
                                                              uint8_t* __restrict__ arg, int HW, int Bt, int L, int chunk_t) {
    extern __shared__ float colmax[];                           // [chunk_t * L]: the utterances are walked chunk_t at a time
    const int i = blockIdx.x, ncol = Bt * L;
    const float* base = mm + (long)i * HW * ncol;
    const float scale = expf(*neg_log_temp);
    for (int t0 = 0; t0 < Bt; t0 += chunk_t) {
        const int nt = min(chunk_t, Bt - t0), c0 = t0 * L;
        for (int cc = threadIdx.x; cc < nt * L; cc += blockDim.x) {
            const int c = c0 + cc;
            float best = base[c];
            int bp = 0;
            for (int p = 1; p < HW; ++p) {
                const float v = base[(long)p * ncol + c];
                if (v > best) { best = v; bp = p; }
            }
            colmax[cc] = best;
            arg[(long)i * ncol + c] = (uint8_t)bp;
        }
        __syncthreads();
        for (int t = threadIdx.x; t < nt; t += blockDim.x) {
            float s = 0.f;
            for (int l = 0; l < L; ++l) s += colmax[t * L + l];
            logits[(long)i * Bt + t0 + t] = s / (float)len[t0 + t] * scale;
        }
        __syncthreads();
    }
}

// d_mm[(i,p)][(t,l)] = (p == arg[i][(t,l)]) * d_logits[i][t] * exp(neg_log_temp) / len[t]   (dense, fully overwritten)
__global__ __launch_bounds__(256) void spatial_max_bwd_kernel(const float* __restrict__ d_logits, const uint8_t* __restrict__ arg,
                                                              const int64_t* __restrict__ len, const float* __restrict__ neg_log_temp,
                                                              float* __restrict__ d_mm, int Bi, int HW, int Bt, int L) {
    const int ncol = Bt * L;
    const long total = (long)Bi * HW * ncol;
    const float scale = expf(*neg_log_temp);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % ncol);
        const long r = e / ncol;
        const int p = (int)(r % HW), i = (int)(r / HW);
        const int t = c / L;
        d_mm[e] = arg[(long)i * ncol + c] == p ? d_logits[(long)i * Bt + t] * scale / (float)len[t] : 0.f;
    }
}

extern "C" int cvcl_embed_meanpool_bwd(const float* d_ret, const int64_t* tok, const int64_t* len, float* d_table,
                                       int B, int L, int E, int V, void* stream) {
    CVCL_CHECK_ARG(d_ret && tok && len && d_table, "cvcl_embed_meanpool_bwd: null pointer");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    CVCL_CHECK_ARG(B > 0 && L > 0 && E > 0 && V > 0, "cvcl_embed_meanpool_bwd: bad shape");
    hipLaunchKernelGGL(embed_meanpool_bwd_kernel<true>, dim3(V), dim3(512), 0, (hipStream_t)stream, d_ret, tok, len, d_table,
                       B, L, E);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

// d_table[v] = sum over positions p with tok[p] == v of dx[p, :] (position order within fixed match groups); padding row 0 -> 0
extern "C" int cvcl_embed_rows_bwd(const float* dx, const int64_t* tok, float* d_table, int n_pos, int E, int V, void* stream) {
    CVCL_CHECK_ARG(dx && tok && d_table && n_pos > 0 && E > 0 && V > 0, "cvcl_embed_rows_bwd: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    hipLaunchKernelGGL(embed_meanpool_bwd_kernel<false>, dim3(V), dim3(512), 0, (hipStream_t)stream, dx, tok, (const int64_t*)nullptr,
                       d_table, n_pos, 1, E);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_l2norm_fwd(const float* x, float* y, float* norm, int N, int E, float eps, void* stream) {
    CVCL_CHECK_ARG(x && y && norm && N > 0 && E > 0, "cvcl_l2norm_fwd: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cvcl_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, x, y, norm, N, E, eps);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_l2norm_bwd(const float* y, const float* norm, const float* dy, float* dx, int N, int E, float eps,
                               void* stream) {
    CVCL_CHECK_ARG(y && norm && dy && dx && N > 0 && E > 0, "cvcl_l2norm_bwd: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cvcl_div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, y, norm, dy, dx, N, E, eps);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_sim_logits_fwd(const float* img, const float* txt, const float* neg_log_temp, float* logits,
                                   int Ni, int Nt, int E, void* stream) {
    CVCL_CHECK_ARG(img && txt && neg_log_temp && logits, "cvcl_sim_logits_fwd: null pointer");
    cvcl_gemm_args a = {};
    a.A = img; a.W = txt; a.C = logits;
    a.M = Ni; a.N = Nt; a.K = E; a.lda = E; a.ldw = E; a.ldc = Nt;
    a.exp_scale = neg_log_temp;
    return cvcl_gemm(CVCL_F32, &a, stream);
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// (round 5: the gradient GEMMs read their operands K-major in place -- the workspace only holds the 256 partial sums of the
// temperature gradient; the arguments stay for ABI stability)
extern "C" size_t cvcl_sim_logits_bwd_workspace_bytes(int, int, int) { return al256(256 * 4); }

extern "C" int cvcl_sim_logits_bwd_rows(const float* img, const float* txt, const float* neg_log_temp, const float* logits,
                                        const float* d_logits, float* d_img_rows, float* d_txt_rows, float* d_neg_log_temp,
                                        int Ni, int Nt, int E, int i0, int ni, int t0, int nt, void* workspace, size_t workspace_bytes,
                                        void* stream) {
    CVCL_CHECK_ARG(img && txt && neg_log_temp && d_logits, "cvcl_sim_logits_bwd: null pointer");
    CVCL_CHECK_ARG(Ni > 0 && Nt > 0 && E > 0 && i0 >= 0 && ni >= 0 && i0 + ni <= Ni && t0 >= 0 && nt >= 0 && t0 + nt <= Nt,
                   "cvcl_sim_logits_bwd: row ranges [%d, +%d) of %d / [%d, +%d) of %d", i0, ni, Ni, t0, nt, Nt);
    int rc;
    if (d_img_rows && ni > 0) {                    // d_img[i0 + r] = s * sum_t dS[i0 + r][t] txt[t]: W = txt as it lies ([K = Nt][N = E])
        cvcl_gemm_args a = {};
        a.A = d_logits + (long)i0 * Nt; a.W = txt; a.C = d_img_rows;
        a.M = ni; a.N = E; a.K = Nt; a.lda = Nt; a.ldw = E; a.ldc = E;
        a.w_trans = 1;
        a.exp_scale = neg_log_temp;
        if ((rc = cvcl_gemm(CVCL_F32, &a, stream))) return rc;
    }
    if (d_txt_rows && nt > 0) {                    // d_txt[t0 + r] = s * sum_i dS[i][t0 + r] img[i]: A = the column block of dS, K-major
        cvcl_gemm_args a = {};
        a.A = d_logits + t0; a.W = img; a.C = d_txt_rows;
        a.M = nt; a.N = E; a.K = Ni; a.lda = Nt; a.ldw = E; a.ldc = E;
        a.a_trans = 1; a.w_trans = 1;
        a.exp_scale = neg_log_temp;
        if ((rc = cvcl_gemm(CVCL_F32, &a, stream))) return rc;
    }
    if (d_neg_log_temp) {                          // d/d(log s) of s*M  =  sum(dS * logits)
        CvclProfScope prof(stream, CVCL_K_HEAD);
        CVCL_CHECK_ARG(logits && workspace, "cvcl_sim_logits_bwd: logits and a workspace are needed for the temperature gradient");
        if (workspace_bytes < cvcl_sim_logits_bwd_workspace_bytes(Ni, Nt, E)) {
            cvcl_set_error("cvcl_sim_logits_bwd: workspace too small");
            return CVCL_EWORKSPACE;
        }
        float* part = (float*)workspace;
        hipLaunchKernelGGL(dot_partial_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, d_logits, logits,
                           (long)Ni * Nt, part);
        hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, 256, d_neg_log_temp);
        CVCL_LAUNCH_CHECK();
    }
    return CVCL_OK;
}

extern "C" int cvcl_sim_logits_bwd(const float* img, const float* txt, const float* neg_log_temp, const float* logits,
                                   const float* d_logits, float* d_img, float* d_txt, float* d_neg_log_temp,
                                   int Ni, int Nt, int E, void* workspace, size_t workspace_bytes, void* stream) {
    return cvcl_sim_logits_bwd_rows(img, txt, neg_log_temp, logits, d_logits, d_img, d_txt, d_neg_log_temp, Ni, Nt, E, 0, Ni, 0, Nt,
                                    workspace, workspace_bytes, stream);
}

extern "C" size_t cvcl_infonce_workspace_bytes(int N) { return (size_t)6 * N * sizeof(float); }

extern "C" int cvcl_infonce_fwd(const float* logits, int N, float* scalars5, float* row_lse, float* col_lse,
                                void* workspace, size_t workspace_bytes, void* stream) {
    CVCL_CHECK_ARG(logits && scalars5 && row_lse && col_lse && workspace && N > 0, "cvcl_infonce_fwd: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    if (workspace_bytes < cvcl_infonce_workspace_bytes(N)) {
        cvcl_set_error("cvcl_infonce_fwd: workspace too small");
        return CVCL_EWORKSPACE;
    }
    float* ws = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(infonce_rows_kernel, dim3(cvcl_div_up(N, 4)), dim3(256), 0, s, logits, N, row_lse, ws);
    hipLaunchKernelGGL(infonce_cols_kernel, dim3(cvcl_div_up(N, 64)), dim3(1024), 0, s, logits, N, col_lse, ws);
    hipLaunchKernelGGL(infonce_finalize_kernel, dim3(1), dim3(256), 0, s, ws, N, scalars5);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_infonce_bwd(const float* logits, const float* row_lse, const float* col_lse, const float* d_loss,
                                float* d_logits, int N, void* stream) {
    CVCL_CHECK_ARG(logits && row_lse && col_lse && d_loss && d_logits && N > 0, "cvcl_infonce_bwd: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    const long total = (long)N * N;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(infonce_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, row_lse, col_lse,
                       d_loss, d_logits, N);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_row_entropy(const float* x, float* out, int R, int N, void* stream) {
    CVCL_CHECK_ARG(x && out && R > 0 && N > 0, "cvcl_row_entropy: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    hipLaunchKernelGGL(row_entropy_kernel, dim3(cvcl_div_up(R, 4)), dim3(256), 0, (hipStream_t)stream, x, out, R, N);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_spatial_max_fwd(const float* mm, const int64_t* len, const float* neg_log_temp, float* logits, uint8_t* arg,
                                    int Bi, int HW, int Bt, int L, void* stream) {
    CVCL_CHECK_ARG(mm && len && neg_log_temp && logits && arg && Bi > 0 && HW > 0 && HW <= 256 && Bt > 0 && L > 0,
                   "cvcl_spatial_max_fwd: bad args");
    CVCL_CHECK_ARG(L <= 8192, "cvcl_spatial_max_fwd: L = %d exceeds the LDS row buffer", L);
    // the per-utterance maxima live in LDS, at most 8192 (utterance, word) columns (32 KiB) at a time: the global batch of a
    // data-parallel run (2048 utterances padded to 25 words) is walked in chunks, same values as one pass
    const int chunk_t = Bt < 8192 / L ? Bt : 8192 / L;
    const size_t lds = (size_t)chunk_t * L * sizeof(float);
    CvclProfScope prof(stream, CVCL_K_HEAD);
    hipLaunchKernelGGL(spatial_max_fwd_kernel, dim3(Bi), dim3(256), lds, (hipStream_t)stream, mm, len, neg_log_temp, logits, arg, HW, Bt, L,
                       chunk_t);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

// ---------------------------------------------------------------------------------------------
// language-model cross entropy (reference multimodal/multimodal.py:884-889: F.cross_entropy(..., ignore_index = PAD,
// reduction "none")): one workgroup per token row of logits [R, V]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void token_ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int V,
                                                           int ignore, float* __restrict__ loss, float* __restrict__ lse) {
    __shared__ float scratch[8];
    const long r = blockIdx.x;
    const float* row = logits + r * V;
    float mx = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, row[v]);
    mx = block_max(mx, scratch);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) s += expf(row[v] - mx);
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) {
        const float l = mx + logf(s);
        lse[r] = l;
        const int64_t lab = labels[r];
        loss[r] = (lab == ignore || lab < 0 || lab >= V) ? 0.f : l - row[lab];
    }
}

__global__ __launch_bounds__(256) void token_ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                           const float* __restrict__ lse, const float* __restrict__ d_loss,
                                                           float* __restrict__ d_logits, int V, int ignore) {
    const long r = blockIdx.x;
    const int64_t lab = labels[r];
    const bool live = !(lab == ignore || lab < 0 || lab >= V);
    const float g = live ? d_loss[r] : 0.f, l = lse[r];
    for (int v = threadIdx.x; v < V; v += 256) {
        const float p = expf(logits[r * V + v] - l);
        d_logits[r * V + v] = (p - (v == lab ? 1.f : 0.f)) * g;
    }
}

// the three masked means of multimodal_lit.py:284-300 (all non-pad tokens / without <sos> / without <sos>, <eos>) and their
// token counts; single workgroup, fixed order.  bwd: d_loss[r] = sum_k d_means[k] * mask_k[r] / count_k
__global__ __launch_bounds__(256) void lm_summaries_kernel(const float* __restrict__ loss, const int64_t* __restrict__ labels, int R,
                                                           int pad, int sos, int eos, float* __restrict__ means,
                                                           float* __restrict__ counts) {
    __shared__ float scratch[8];
    float s[3] = {0.f, 0.f, 0.f}, n[3] = {0.f, 0.f, 0.f};
    for (int r = threadIdx.x; r < R; r += 256) {
        const int64_t lab = labels[r];
        const bool m0 = lab != pad, m1 = m0 && lab != sos, m2 = m1 && lab != eos;
        const float v = loss[r];
        s[0] += v; n[0] += m0;                       // reference :287 sums the unmasked loss (pads are already 0)
        if (m1) { s[1] += v; n[1] += 1.f; }
        if (m2) { s[2] += v; n[2] += 1.f; }
    }
    for (int k = 0; k < 3; ++k) {
        const float ts = block_sum(s[k], scratch), tn = block_sum(n[k], scratch);
        if (threadIdx.x == 0) { means[k] = ts / tn; counts[k] = tn; }
    }
}

__global__ __launch_bounds__(256) void lm_summaries_bwd_kernel(const float* __restrict__ d_means, const int64_t* __restrict__ labels,
                                                               const float* __restrict__ counts, int R, int pad, int sos, int eos,
                                                               float* __restrict__ d_loss) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
        const int64_t lab = labels[r];
        const bool m0 = lab != pad, m1 = m0 && lab != sos, m2 = m1 && lab != eos;
        float g = d_means[0] / counts[0];
        if (m1) g += d_means[1] / counts[1];
        if (m2) g += d_means[2] / counts[2];
        d_loss[r] = g;
    }
}

// d_neg_log_temp = sum d_logits * logits (logits = match * exp(nlt)); one workgroup, fixed order
__global__ __launch_bounds__(1024) void dot_all_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                       float* __restrict__ out) {
    __shared__ float scratch[16];
    float acc = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) acc = fmaf(a[i], b[i], acc);
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) *out = acc;
}

extern "C" int cvcl_spatial_max_bwd(const float* d_logits, const uint8_t* arg, const int64_t* len, const float* neg_log_temp,
                                    const float* logits, float* d_mm, float* d_neg_log_temp, int Bi, int HW, int Bt, int L,
                                    void* stream) {
    CVCL_CHECK_ARG(d_logits && arg && len && neg_log_temp && d_mm && Bi > 0 && HW > 0 && Bt > 0 && L > 0 &&
                       (!d_neg_log_temp || logits), "cvcl_spatial_max_bwd: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    if (d_neg_log_temp)
        hipLaunchKernelGGL(dot_all_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_logits, logits, (long)Bi * Bt, d_neg_log_temp);
    const long total = (long)Bi * HW * Bt * L;
    long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(spatial_max_bwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, d_logits, arg, len, neg_log_temp, d_mm, Bi,
                       HW, Bt, L);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_token_ce_fwd(const float* logits, const int64_t* labels, float* loss, float* lse, long R, int V, int ignore_index,
                                 void* stream) {
    CVCL_CHECK_ARG(logits && labels && loss && lse && R > 0 && V > 0, "cvcl_token_ce_fwd: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    hipLaunchKernelGGL(token_ce_fwd_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, labels, V, ignore_index, loss, lse);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_token_ce_bwd(const float* logits, const int64_t* labels, const float* lse, const float* d_loss, float* d_logits,
                                 long R, int V, int ignore_index, void* stream) {
    CVCL_CHECK_ARG(logits && labels && lse && d_loss && d_logits && R > 0 && V > 0, "cvcl_token_ce_bwd: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    hipLaunchKernelGGL(token_ce_bwd_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, labels, lse, d_loss, d_logits, V,
                       ignore_index);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_lm_loss_summaries(const float* loss, const int64_t* labels, const float* d_means, float* means, float* counts,
                                      float* d_loss, int R, int pad, int sos, int eos, void* stream) {
    CVCL_CHECK_ARG(labels && counts && R > 0 && ((loss && means) || (d_means && d_loss)), "cvcl_lm_loss_summaries: bad args");
    CvclProfScope prof(stream, CVCL_K_HEAD);
    if (!d_loss)
        hipLaunchKernelGGL(lm_summaries_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss, labels, R, pad, sos, eos, means, counts);
    else
        hipLaunchKernelGGL(lm_summaries_bwd_kernel, dim3(cvcl_div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, d_means, labels, counts,
                           R, pad, sos, eos, d_loss);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}
