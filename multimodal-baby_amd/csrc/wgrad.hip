// Weight-gradient GEMMs of the fine-tuning path (--finetune_cnn; reference: autograd through torchvision's Conv2d,
// call sites multimodal/multimodal.py:155-158,175-179):
//
//     C[n][k] = sum_m A[m][n] * B[m][k]            ("TN" GEMM: the contraction runs over the ROWS of both operands)
//
// A = dY [pixels, Cout], B = X [pixels, Cin] as they sit in HBM (NHWC, channel-contiguous) -- no transposed copies.
// bf16 kernel: 128 x 128 output tile per workgroup (4 waves, 64 x 64 each = 2 x 2 MFMA 32x32x16 tiles), 64 rows of both
// operands staged per step into a row-major LDS image (pitch 320 B); the MFMA fragments need 8 consecutive m for a
// fixed column, i.e. the transpose of what is stored, and are read with gfx950's ds_read_b64_tr_b16 (two per
// fragment).  Pitch 320 B = 80 dwords puts the 4 rows x 2 column groups that the first 32 lanes touch on disjoint
// banks.  The pixel dimension is split S ways (deterministic: fp32 partials [S][N][K], then a fixed-order reduction);
// workgroups of one split land on one XCD (linear id % 8 == s % 8) so that the operand rows they share are fetched into
// that XCD's L2 once.
//
// conv-tap mode (grouped 3x3 weight gradient): grid.z = tap (ky, kx); row m = (b, oy, ox) of B is taken from pixel
// (b, oy*stride + ky - 1, ox*stride + kx - 1) (zeros outside the image) and only the diagonal tiles (same 128-channel
// slab of A and B) are computed: every group of <= 32 channels lies inside one slab; the reduction kernel picks the
// block-diagonal out.  The dense 128 x 128 products waste MFMA work (x4 .. x32), which is cheap; HBM traffic is what counts.
#include <cstdlib>

#include "cvcl_common.h"

namespace {

constexpr int TN_T = 128;            // output tile edge
constexpr int TN_BM = 64;            // rows (pixels) per step
constexpr int TN_PITCH = 320;        // LDS bytes per staged row (256 B of data + 64 B pad)
constexpr int TN_TILE_BYTES = TN_BM * TN_PITCH;

struct TnDev {
    const bf16_t* A; const bf16_t* B; float* P;
    long M;                   // rows of A (contraction length)
    int N, K, lda, ldb;
    int S; long chunk;        // split count, rows per split (multiple of TN_BM)
    int tiles_n, tiles_k, diag;
    int taps, Ho, Wo, Hi, Wi, stride;      // conv-tap gather on B (taps == 9) -- else taps == 1
    float* colsum;            // optional: per-split column sums of A, [S][tiles_n][128] (written by the tk == 0 workgroups)
};

typedef __bf16 tr_bf16x4 __attribute__((__vector_size__(4 * sizeof(__bf16))));

__device__ inline bf16x4 lds_tr_read(const char* p) {
    auto lp = reinterpret_cast<__attribute__((address_space(3))) tr_bf16x4*>(
        (__attribute__((address_space(3))) char*)(p));
    tr_bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(lp);
    return __builtin_bit_cast(bf16x4, v);
}

// CONV: the B operand is gathered at a convolution tap (grouped 3x3 weight gradient); otherwise both operands are plain row-major
// matrices and every load is unconditional from a clamped row / column (a branch around a load costs a serializing vmcnt(0), and the
// tap path's index arithmetic -- 64-bit divisions per load -- otherwise sits in the plain GEMM's loop too)
template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_kernel(TnDev p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * TN_TILE_BYTES];
    char* sA = smem;
    char* sB = smem + TN_TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;

    // linear id -> (split s, tile): id % 8 = s % 8 (XCD), then tiles of one split adjacent
    const int ntile = p.diag ? p.tiles_n : p.tiles_n * p.tiles_k;
    int id = blockIdx.x, s, tile;
    if (p.S % 8 == 0) { s = (id & 7) + 8 * ((id >> 3) / ntile); tile = (id >> 3) % ntile; }
    else { s = id / ntile; tile = id % ntile; }
    const int tn = p.diag ? tile : tile / p.tiles_k, tk = p.diag ? tile : tile % p.tiles_k;
    const int tap = blockIdx.y;
    const int n0 = tn * TN_T, k0 = tk * TN_T;
    const long m_begin = (long)s * p.chunk;
    long m_end = m_begin + p.chunk;
    if (m_end > p.M) m_end = p.M;

    // staging role: 16 chunks (16 B = 8 channels) per row, 16 rows per pass, 4 passes per operand
    const int s_chunk = tid & 15, s_row0 = tid >> 4;
    const bool a_col_ok = n0 + s_chunk * 8 < p.N, b_col_ok = k0 + s_chunk * 8 < p.K;
    const int ky = tap / 3 - 1, kx = tap % 3 - 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    u32x4 ra[4], rb[4];
    const int a_col = min(n0 + s_chunk * 8, p.N - 8), b_col = min(k0 + s_chunk * 8, p.K - 8);     // clamped: always a valid 16-byte chunk
    // full 64-row steps of interior tiles (all but a split's last step and the ragged edge tiles): one address product per operand
    // and no value-or-zero selects -- the general form below spent 7.4 VALU instructions per MFMA on them (profiles/r03_pmc_sq_finetune.txt)
    const bool cols_full = n0 + TN_T <= p.N && k0 + TN_T <= p.K;
    const long a16 = 16L * p.lda, b16 = 16L * p.ldb;
    auto load_step = [&](long m0) {
        if constexpr (!CONV) {
            if (cols_full && m0 + TN_BM <= m_end) {
                const bf16_t* a = p.A + (m0 + s_row0) * (long)p.lda + a_col;
                const bf16_t* b = p.B + (m0 + s_row0) * (long)p.ldb + b_col;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ra[i] = *reinterpret_cast<const u32x4*>(a + i * a16);
                    rb[i] = *reinterpret_cast<const u32x4*>(b + i * b16);
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = m0 + s_row0 + 16 * i;
            const u32x4 z = {0u, 0u, 0u, 0u};
            if constexpr (!CONV) {
                const long mc = m < m_end ? m : m_end - 1;
                const u32x4 va = *reinterpret_cast<const u32x4*>(p.A + mc * p.lda + a_col);
                const u32x4 vb = *reinterpret_cast<const u32x4*>(p.B + mc * p.ldb + b_col);
                ra[i] = (m < m_end && a_col_ok) ? va : z;
                rb[i] = (m < m_end && b_col_ok) ? vb : z;
            } else {
                // same rule for the gathered operand: both loads are issued unconditionally from clamped coordinates and the
                // predicates (row inside the split, tap inside the image) select value or zero afterwards; M < 2^31 (checked by
                // the host), so the pixel decomposition is 32-bit
                const unsigned mc = (unsigned)(m < m_end ? m : m_end - 1);
                const u32x4 va = *reinterpret_cast<const u32x4*>(p.A + (long)mc * p.lda + a_col);
                const unsigned t = mc / (unsigned)p.Wo, ox = mc - t * (unsigned)p.Wo;
                const unsigned b = t / (unsigned)p.Ho, oy = t - b * (unsigned)p.Ho;
                const int yi = (int)oy * p.stride + ky, xi = (int)ox * p.stride + kx;
                const bool inside = yi >= 0 && yi < p.Hi && xi >= 0 && xi < p.Wi;
                const int yc = min(max(yi, 0), p.Hi - 1), xc = min(max(xi, 0), p.Wi - 1);
                const u32x4 vb = *reinterpret_cast<const u32x4*>(p.B + (((long)b * p.Hi + yc) * p.Wi + xc) * p.ldb + b_col);
                ra[i] = (m < m_end && a_col_ok) ? va : z;
                rb[i] = (m < m_end && b_col_ok && inside) ? vb : z;
            }
        }
    };
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool want_cs = p.colsum != nullptr && tk == 0;
    auto write_step = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = (s_row0 + 16 * i) * TN_PITCH + s_chunk * 16;
            *reinterpret_cast<u32x4*>(sA + off) = ra[i];
            *reinterpret_cast<u32x4*>(sB + off) = rb[i];
            if (want_cs) {
                const bf16x8 v = __builtin_bit_cast(bf16x8, ra[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[e] += (float)v[e];
            }
        }
    };

    // fragment addressing: 16-lane group g4 reads a [4 m][16 col] block; lane q supplies row q/4, cols 4*(q%4)..+3
    const int g4 = lane >> 4, q = lane & 15;
    const int frag_row = (g4 >> 1) * 8 + (q >> 2);                 // + kk*16 + h*4
    const int frag_col = (g4 & 1) * 16 + (q & 3) * 4;              // + wave/tile column base
    const char* fa = sA + frag_row * TN_PITCH + (wn * 64 + frag_col) * 2;
    const char* fb = sB + frag_row * TN_PITCH + (wk * 64 + frag_col) * 2;

    if (m_begin < m_end) {
        load_step(m_begin);
        write_step();
        __syncthreads();
        for (long m0 = m_begin; m0 < m_end; m0 += TN_BM) {
            const bool more = m0 + TN_BM < m_end;
            if (more) load_step(m0 + TN_BM);
#pragma unroll
            for (int kk = 0; kk < TN_BM / 16; ++kk) {
                bf16x8 af[2], bfr[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x4 a0 = lds_tr_read(fa + (kk * 16) * TN_PITCH + t * 64);
                    const bf16x4 a1 = lds_tr_read(fa + (kk * 16 + 4) * TN_PITCH + t * 64);
                    const bf16x4 b0 = lds_tr_read(fb + (kk * 16) * TN_PITCH + t * 64);
                    const bf16x4 b1 = lds_tr_read(fb + (kk * 16 + 4) * TN_PITCH + t * 64);
                    af[t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    bfr[t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) {
                write_step();
                __syncthreads();
            }
        }
    }

    if (want_cs) {                 // column sums of this split's rows of A: reduce the 16 staging row lanes (fixed order)
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[s_row0 * 128 + s_chunk * 8 + e] = csum[e];
        __syncthreads();
        if (tid < 128) {
            float a = 0.f;
            for (int l = 0; l < 16; ++l) a += red[l * 128 + tid];
            p.colsum[((long)s * p.tiles_n + tn) * TN_T + tid] = a;
        }
    }

    // partial tile -> P[tap][s][...]: full [N][K] matrix per (tap, s), or the stack of diagonal tiles
    float* out;
    long ldo;
    if (p.diag) { out = p.P + (((long)tap * p.S + s) * p.tiles_n + tn) * TN_T * TN_T; ldo = TN_T; }
    else { out = p.P + ((long)tap * p.S + s) * p.N * p.K + (long)n0 * p.K + k0; ldo = p.K; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wk * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wn * 64 + i * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                if (p.diag || (n0 + row < p.N && k0 + col < p.K)) out[(long)row * ldo + col] = acc[i][j][r];
            }
        }
}

// fp32 parity mode: plain LDS-tiled VALU kernel, 64 x 64 tile, 4 x 4 outputs per thread, same split / partial layout
struct TnF32Dev { const float* A; const float* B; float* P; long M; int N, K, lda, ldb, S; long chunk; int tiles_k; };

__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(TnF32Dev p) {
    __shared__ float sA[16][64 + 4], sB[16][64 + 4];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, s = blockIdx.y;
    const int n0 = (tile / p.tiles_k) * 64, k0 = (tile % p.tiles_k) * 64;
    const long m_begin = (long)s * p.chunk;
    long m_end = m_begin + p.chunk;
    if (m_end > p.M) m_end = p.M;
    const int tr = tid >> 4, tc = tid & 15;          // thread's 4 x 4 block: rows tr*4.., cols tc*4..
    float acc[4][4] = {};
    for (long m0 = m_begin; m0 < m_end; m0 += 16) {
        for (int e = tid; e < 16 * 64; e += 256) {
            const int r = e >> 6, c = e & 63;
            const long m = m0 + r;
            sA[r][c] = (m < m_end && n0 + c < p.N) ? p.A[m * p.lda + n0 + c] : 0.f;
            sB[r][c] = (m < m_end && k0 + c < p.K) ? p.B[m * p.ldb + k0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = sA[r][tr * 4 + i]; b[i] = sB[r][tc * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* out = p.P + (long)s * p.N * p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tr * 4 + i, k = k0 + tc * 4 + j;
            if (n < p.N && k < p.K) out[(long)n * p.K + k] = acc[i][j];
        }
}

// C[n][k] = sum_s P[s][n][k]   (fixed order -> deterministic); ldc/col_limit let the stem drop its padded columns
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ P, float* __restrict__ C, int S, int N, int K,
                                                        int k_keep) {
    const long total = (long)N * k_keep;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / k_keep), k = (int)(i % k_keep);
        C[i] = ordered_sum<8, float>(S, [&](int s) { return P[((long)s * N + n) * K + k]; });
    }
}

// grouped 3x3: dW[co][ci][tap] = sum_s P[tap][s][co / 128][co % 128][(co / cg * cg + ci) % 128]
__global__ __launch_bounds__(256) void gconv_wgrad_reduce_kernel(const float* __restrict__ P, float* __restrict__ dw, int S, int C,
                                                                 int cg) {
    const long total = (long)C * cg * 9;
    const int tiles = C / TN_T;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 9), ci = (int)((i / 9) % cg), co = (int)(i / (9 * cg));
        const int cin = (co / cg) * cg + ci;
        const float* src = P + (((long)tap * S * tiles) + co / TN_T) * TN_T * TN_T + (long)(co % TN_T) * TN_T + (cin % TN_T);
        dw[i] = ordered_sum<8, float>(S, [&](int s) { return src[(long)s * tiles * TN_T * TN_T]; });
    }
}

// stem 7x7/2 pad 3: bf16 patch matrix col[p][(c*7 + ky)*7 + kx] (147 columns padded to 160 with zeros), p = (b, oy, ox)
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, bf16_t* __restrict__ col, int B, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo * 20;                 // 20 chunks of 8 columns per pixel
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 20);
        const long pix = i / 20;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
        const long b = pix / ((long)Wo * Ho);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = ch * 8 + e;
            float f = 0.f;
            if (j < 147) {
                const int c = j / 49, kyy = (j / 7) % 7, kxx = j % 7;
                const int yi = 2 * oy - 3 + kyy, xi = 2 * ox - 3 + kxx;
                if (yi >= 0 && yi < H && xi >= 0 && xi < W) f = x[((b * 3 + c) * H + yi) * W + xi];
            }
            v[e] = (bf16_t)f;
        }
        *reinterpret_cast<bf16x8*>(col + pix * 160 + ch * 8) = v;
    }
}

struct TnPlan { int S; long chunk; int tiles_n, tiles_k, ntile; };

TnPlan tn_plan(long M, int N, int K, int taps, bool diag, int tile, int target_wgs = 512) {
    TnPlan pl;
    pl.tiles_n = cvcl_div_up(N, tile);
    pl.tiles_k = cvcl_div_up(K, tile);
    pl.ntile = diag ? pl.tiles_n : pl.tiles_n * pl.tiles_k;
    long want = cvcl_div_up(target_wgs, (long)pl.ntile * taps);      // default ~2 workgroups per CU in total (1024: partial-tile traffic dominates, 256: too few)
    const long max_s = cvcl_div_up(M, 1024);                         // >= 1024 rows per split
    if (want > max_s) want = max_s;
    if (want < 1) want = 1;
    if (want >= 8) want = (want + 7) / 8 * 8;
    pl.chunk = (cvcl_div_up(M, want) + TN_BM - 1) / TN_BM * TN_BM;
    pl.S = cvcl_div_up(M, pl.chunk);
    if (pl.S >= 8 && pl.S % 8) {                                     // keep the XCD mapping exact: pad with empty splits
        pl.S = (pl.S + 7) / 8 * 8;
    }
    return pl;
}

size_t tn_ws_bytes(const TnPlan& pl, int N, int K, int taps, bool diag) {
    return diag ? (size_t)taps * pl.S * pl.tiles_n * TN_T * TN_T * 4 : (size_t)taps * pl.S * N * K * 4;
}

int reduce_grid(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" size_t cvcl_gemm_tn_workspace_bytes(int dtype, long M, int N, int K) {
    const TnPlan pl = tn_plan(M, N, K, 1, false, dtype == CVCL_BF16 ? TN_T : 64);
    return tn_ws_bytes(pl, N, K, 1, false);
}

extern "C" int cvcl_gemm_tn(int dtype, const void* A, int lda, const void* B, int ldb, long M, int N, int K, float* C,
                            int k_keep, void* workspace, size_t workspace_bytes, void* stream) {
    CVCL_CHECK_ARG(A && B && C && workspace && M > 0 && N > 0 && K > 0 && lda >= N && ldb >= K && k_keep > 0 && k_keep <= K,
                   "cvcl_gemm_tn: bad args");
    hipStream_t st = (hipStream_t)stream;
    const bool bf = dtype == CVCL_BF16;
    const TnPlan pl = tn_plan(M, N, K, 1, false, bf ? TN_T : 64);
    if (workspace_bytes < tn_ws_bytes(pl, N, K, 1, false)) {
        cvcl_set_error("cvcl_gemm_tn: workspace too small");
        return CVCL_EWORKSPACE;
    }
    CvclProfScope prof(stream, CVCL_K_WGRAD);
    if (bf) {
        CVCL_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
                       "cvcl_gemm_tn: bf16 operands need 16-byte aligned rows (N, K, lda, ldb multiples of 8)");
        TnDev d = {};
        d.A = (const bf16_t*)A; d.B = (const bf16_t*)B; d.P = (float*)workspace;
        d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.S = pl.S; d.chunk = pl.chunk;
        d.tiles_n = pl.tiles_n; d.tiles_k = pl.tiles_k; d.diag = 0; d.taps = 1;
        hipLaunchKernelGGL(gemm_tn_bf16_kernel<false>, dim3(pl.ntile * pl.S, 1), dim3(256), 0, st, d);
    } else {
        TnF32Dev d = {(const float*)A, (const float*)B, (float*)workspace, M, N, K, lda, ldb, pl.S, pl.chunk, pl.tiles_k};
        hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3(pl.ntile, pl.S), dim3(256), 0, st, d);
    }
    CVCL_LAUNCH_CHECK();
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(reduce_grid((long)N * k_keep)), dim3(256), 0, st, (const float*)workspace, C, pl.S, N, K,
                       k_keep);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

// ---- linear layer: weight gradient C = A^T B and bias gradient colsum[n] = sum_m A[m][n] from one pass over A (= dY) ------------
namespace {
__global__ __launch_bounds__(256) void tn_colsum_reduce_kernel(const float* __restrict__ CS, int S, int tiles, int N, float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    out[n] = (float)ordered_sum<8, double>(S, [&](int s) { return CS[((long)s * tiles + n / TN_T) * TN_T + n % TN_T]; });   // fixed order
}
size_t tn_colsum_off(const TnPlan& pl, int N, int K) { return (tn_ws_bytes(pl, N, K, 1, false) + 255) & ~(size_t)255; }
}  // namespace

extern "C" size_t cvcl_gemm_tn_colsum_workspace_bytes(long M, int N, int K) {
    const TnPlan pl = tn_plan(M, N, K, 1, false, TN_T);
    return tn_colsum_off(pl, N, K) + (size_t)pl.S * pl.tiles_n * TN_T * 4;
}

// bf16 A [M, lda >= N], B [M, ldb >= K] -> C [N][k_keep] fp32 = A^T B and colsum [N] fp32 = column sums of A
// (nn.Linear backward: A = dY, B = X -> dW and db; the staging pass of the TN kernel adds up the dY tile it holds anyway)
extern "C" int cvcl_gemm_tn_colsum(const void* A, int lda, const void* B, int ldb, long M, int N, int K, float* C, int k_keep,
                                   float* colsum, void* workspace, size_t workspace_bytes, void* stream) {
    CVCL_CHECK_ARG(A && B && C && colsum && workspace && M > 0 && N > 0 && K > 0 && lda >= N && ldb >= K && k_keep > 0 && k_keep <= K,
                   "cvcl_gemm_tn_colsum: bad args");
    CVCL_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
                   "cvcl_gemm_tn_colsum: bf16 operands need 16-byte aligned rows (N, K, lda, ldb multiples of 8)");
    const TnPlan pl = tn_plan(M, N, K, 1, false, TN_T);
    if (workspace_bytes < cvcl_gemm_tn_colsum_workspace_bytes(M, N, K)) {
        cvcl_set_error("cvcl_gemm_tn_colsum: workspace too small");
        return CVCL_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    CvclProfScope prof(stream, CVCL_K_WGRAD);
    TnDev d = {};
    d.A = (const bf16_t*)A; d.B = (const bf16_t*)B; d.P = (float*)workspace;
    d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.S = pl.S; d.chunk = pl.chunk;
    d.tiles_n = pl.tiles_n; d.tiles_k = pl.tiles_k; d.diag = 0; d.taps = 1;
    d.colsum = (float*)((char*)workspace + tn_colsum_off(pl, N, K));
    hipLaunchKernelGGL(gemm_tn_bf16_kernel<false>, dim3(pl.ntile * pl.S, 1), dim3(256), 0, st, d);
    CVCL_LAUNCH_CHECK();
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(reduce_grid((long)N * k_keep)), dim3(256), 0, st, (const float*)workspace, C, pl.S, N, K, k_keep);
    hipLaunchKernelGGL(tn_colsum_reduce_kernel, dim3(cvcl_div_up(N, 256)), dim3(256), 0, st, (const float*)d.colsum, pl.S, pl.tiles_n, N, colsum);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

namespace {
// ---- grouped 3x3 weight gradient, all nine taps in ONE pass over the operands (round 3) ------------------------------------------
// The tap-at-a-time form above reads dY and X nine times (layer 1 at B = 256: 3.7 GB per launch, 562 us).  Here a workgroup owns a
// 128-channel slab and walks bands of TH output rows of an image: the band of X (with its halo rows / columns, zeros outside the
// image) and the band of dY (rows padded to a multiple of 16 pixels with zeros) are staged once into LDS as pixel-major rows
// (pitch 320 B, the TN kernel's layout); wave w owns the diagonal 32 x 32 block w of the slab (every group of <= 32 channels lies
// inside one) for ALL nine taps -- 9 accumulator tiles = 144 registers, kept across the workgroup's bands.  Per 16 pixels of a row:
// one dY fragment (two ds_read_b64_tr_b16) and nine X fragments whose rows are the same pixels shifted by the tap: in the staged
// band a tap is a constant row offset (ky * Wp + kx) and the stride a row multiplier, so every tap is two transposing reads at
// another address.  Partial blocks [workgroup][tap][wave][32][32] fp32, reduced in a fixed order (deterministic).
constexpr int GW_PITCH = TN_PITCH;
constexpr int GW_UN = 6;                // staging loads in flight per thread (in-place staging of an image's first band)
constexpr int GW_XN = 4, GW_DN = 4;     // register image of an incremental band: <= 64 pixels of X and of dY (16 per thread row lane and slot)

struct GwDev {
    const bf16_t* x; const bf16_t* dy; float* P;
    int B, H, W, C, stride, Ho, Wo, Wo_pad, TH, bands, rows_in, x_rows, dy_rows;
    int pipe;        // the incremental band fits the register image: the next band is fetched behind the current one's MFMAs
};

__global__ __launch_bounds__(256, 2) void gconv_wgrad_band_kernel(GwDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Wp = p.W + 2, R = p.rows_in;
    char* sX = smem;                                         // [x_rows][GW_PITCH]: a RING of R input rows x Wp pixels + zero slack rows
    char* sD = smem + (size_t)p.x_rows * GW_PITCH;           // [TH * Wo_pad][GW_PITCH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = blockIdx.y * 128;
    const int s_chunk = tid & 15, s_row0 = tid >> 4;
    const int n_d = p.TH * p.Wo_pad;
    const u32x4 z = {0u, 0u, 0u, 0u};
    // slack rows behind the ring (read by the padded pixels of the last slot's row, multiplied by dY = 0: must be finite) -- zeroed once
    for (int r = R * Wp + s_row0; r < p.x_rows; r += 16) *reinterpret_cast<u32x4*>(sX + r * GW_PITCH + s_chunk * 16) = z;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int g4 = lane >> 4, q = lane & 15;
    const int frag_row = (g4 >> 1) * 8 + (q >> 2);                 // + kk*16 + {0, 4}
    const int frag_col = (g4 & 1) * 16 + (q & 3) * 4;
    const char* fa = sD + frag_row * GW_PITCH + (wave * 32 + frag_col) * 2;
    const char* fb = sX + (frag_row * p.stride) * GW_PITCH + (wave * 32 + frag_col) * 2;
    const int KK = p.Wo_pad / 16;
    const int step4 = 4 * p.stride * GW_PITCH;

    // A workgroup takes a CONTIGUOUS range of bands.  Input row iy of the image lives in ring slot (iy + 1) % R: consecutive bands of an
    // image share 3 - stride input rows, which stay where they are -- only the TH * stride new rows of a band are staged (every byte
    // of X is read once; the tap-at-a-time form read it nine times).  The new rows and the dY band of the NEXT band are fetched into
    // registers before the current band is multiplied and written to LDS behind it (the global round trip hides behind the MFMAs);
    // the first band of an image (all R rows) is staged in place.
    const int n_items = p.B * p.bands, per_wg = (n_items + (int)gridDim.x - 1) / (int)gridDim.x;
    const int item_begin = blockIdx.x * per_wg, item_end = min(n_items, item_begin + per_wg);
    const int inc_rows = p.TH * p.stride, n_xi = inc_rows * Wp;            // incremental band: new input rows / their pixels

    // X rows iy_first .. iy_first + rows - 1 of image b -> ring, in place (batches of GW_UN loads per thread)
    auto stage_x_now = [&](int b, int iy_first, int rows) {
        const int n_x = rows * Wp;
        for (int r0 = s_row0; r0 < n_x; r0 += 16 * GW_UN) {
            u32x4 v[GW_UN];
            bool ok[GW_UN];
            int dst[GW_UN];
#pragma unroll
            for (int i = 0; i < GW_UN; ++i) {
                const int r = min(r0 + 16 * i, n_x - 1);
                const int j = r / Wp, rx = r - j * Wp;
                const int yi = iy_first + j, xi = rx - 1;
                ok[i] = yi >= 0 && yi < p.H && xi >= 0 && xi < p.W;
                const int yc = min(max(yi, 0), p.H - 1), xc = min(max(xi, 0), p.W - 1);
                v[i] = *reinterpret_cast<const u32x4*>(p.x + (((long)b * p.H + yc) * p.W + xc) * p.C + c0 + s_chunk * 8);
                dst[i] = (((yi + 1) % R) * Wp + rx) * GW_PITCH + s_chunk * 16;
            }
#pragma unroll
            for (int i = 0; i < GW_UN; ++i)
                if (r0 + 16 * i < n_x) *reinterpret_cast<u32x4*>(sX + dst[i]) = ok[i] ? v[i] : z;
        }
    };
    auto stage_d_now = [&](int b, int oy0) {
        for (int r0 = s_row0; r0 < n_d; r0 += 16 * GW_UN) {
            u32x4 v[GW_UN];
            bool ok[GW_UN];
#pragma unroll
            for (int i = 0; i < GW_UN; ++i) {
                const int r = min(r0 + 16 * i, n_d - 1);
                const int ty = r / p.Wo_pad, ox = r - ty * p.Wo_pad;
                const int oy = oy0 + ty;
                ok[i] = oy < p.Ho && ox < p.Wo;
                const int yc = min(oy, p.Ho - 1), xc = min(ox, p.Wo - 1);
                v[i] = *reinterpret_cast<const u32x4*>(p.dy + (((long)b * p.Ho + yc) * p.Wo + xc) * p.C + c0 + s_chunk * 8);
            }
#pragma unroll
            for (int i = 0; i < GW_UN; ++i) {
                const int r = r0 + 16 * i;
                if (r < n_d) *reinterpret_cast<u32x4*>(sD + r * GW_PITCH + s_chunk * 16) = ok[i] ? v[i] : z;
            }
        }
    };
    // register image of one incremental band: GW_XN x 16 pixels of X, GW_DN x 16 pixels of dY per thread row lane
    u32x4 px[GW_XN], pd[GW_DN];
    unsigned px_ok = 0, pd_ok = 0;
    auto fetch = [&](int b, int oy0, int iy_first) {
        px_ok = 0; pd_ok = 0;
#pragma unroll
        for (int i = 0; i < GW_XN; ++i) {
            const int r = min(s_row0 + 16 * i, n_xi - 1);
            const int j = r / Wp, rx = r - j * Wp;
            const int yi = iy_first + j, xi = rx - 1;
            if (yi >= 0 && yi < p.H && xi >= 0 && xi < p.W) px_ok |= 1u << i;
            const int yc = min(max(yi, 0), p.H - 1), xc = min(max(xi, 0), p.W - 1);
            px[i] = *reinterpret_cast<const u32x4*>(p.x + (((long)b * p.H + yc) * p.W + xc) * p.C + c0 + s_chunk * 8);
        }
#pragma unroll
        for (int i = 0; i < GW_DN; ++i) {
            const int r = min(s_row0 + 16 * i, n_d - 1);
            const int ty = r / p.Wo_pad, ox = r - ty * p.Wo_pad;
            const int oy = oy0 + ty;
            if (oy < p.Ho && ox < p.Wo) pd_ok |= 1u << i;
            const int yc = min(oy, p.Ho - 1), xc = min(ox, p.Wo - 1);
            pd[i] = *reinterpret_cast<const u32x4*>(p.dy + (((long)b * p.Ho + yc) * p.Wo + xc) * p.C + c0 + s_chunk * 8);
        }
    };
    auto commit = [&](int iy_first) {
#pragma unroll
        for (int i = 0; i < GW_XN; ++i) {
            const int r = s_row0 + 16 * i;
            if (r < n_xi) {
                const int j = r / Wp, rx = r - j * Wp;
                *reinterpret_cast<u32x4*>(sX + ((((iy_first + j + 1) % R) * Wp + rx) * GW_PITCH + s_chunk * 16)) = ((px_ok >> i) & 1) ? px[i] : z;
            }
        }
#pragma unroll
        for (int i = 0; i < GW_DN; ++i) {
            const int r = s_row0 + 16 * i;
            if (r < n_d) *reinterpret_cast<u32x4*>(sD + r * GW_PITCH + s_chunk * 16) = ((pd_ok >> i) & 1) ? pd[i] : z;
        }
    };

    bool staged = false;                                     // this band's operands are already in LDS (committed behind the previous band)
    for (int item = item_begin; item < item_end; ++item) {
        const int b = item / p.bands, band = item - b * p.bands;
        const int oy0 = band * p.TH, iy0 = oy0 * p.stride - 1;
        if (!staged) {
            __syncthreads();                                 // the previous band's readers are done
            const bool full = item == item_begin || band == 0;
            const int n_new = full ? R : inc_rows;
            stage_x_now(b, iy0 + R - n_new, n_new);
            stage_d_now(b, oy0);
        }
        __syncthreads();
        const bool pipe = p.pipe && item + 1 < item_end && band + 1 < p.bands;          // the next band continues this image
        if (pipe) fetch(b, oy0 + p.TH, iy0 + inc_rows + R - inc_rows);
        const int th = min(p.TH, p.Ho - oy0);
        for (int ty = 0; ty < th; ++ty) {
            const char* fa_r = fa + (ty * p.Wo_pad) * GW_PITCH;
            int slot_off[3];                                 // byte offset of the ring slot of input row iy0 + ty * stride + ky
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) slot_off[ky] = (((iy0 + ty * p.stride + ky + 1) % R) * Wp) * GW_PITCH;
            for (int kk = 0; kk < KK; ++kk) {
                const bf16x4 a0 = lds_tr_read(fa_r + (kk * 16) * GW_PITCH);
                const bf16x4 a1 = lds_tr_read(fa_r + (kk * 16 + 4) * GW_PITCH);
                const bf16x8 af = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                const char* fb_k = fb + (kk * 16 * p.stride) * GW_PITCH;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {             // one input row's three taps at a time: six reads, then three MFMAs
                    bf16x4 b0[3], b1[3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const char* pt = fb_k + slot_off[ky] + kx * GW_PITCH;
                        b0[kx] = lds_tr_read(pt);
                        b1[kx] = lds_tr_read(pt + step4);
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const bf16x8 bfr = __builtin_shufflevector(b0[kx], b1[kx], 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[ky * 3 + kx], 0, 0, 0);
                    }
                }
            }
        }
        staged = false;
        if (pipe) {
            __syncthreads();                                 // this band's readers are done: the next band's rows may land
            commit(iy0 + inc_rows + R - inc_rows);
            staged = true;
        }
    }
    // partial blocks: P[((slab * G + wg) * 9 + tap) * 4 + wave][co 32][ci 32]
    float* out = p.P + ((((long)blockIdx.y * gridDim.x + blockIdx.x) * 9) * 4 + wave) * 1024;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* o = out + (long)t * 4 * 1024;
        const int col = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
            o[row * 32 + col] = acc[t][r];
        }
    }
}

// dW[co][ci][tap] = sum over the slab's workgroups (fixed order) of the (co, group-local ci) element of the diagonal block.
// One thread per element of the [tap][wave][32][32] partial blocks (consecutive threads = consecutive columns: the G partial
// images are read as full lines; a thread per OUTPUT element gathered one float per 4 KB and took longer than the band kernel);
// the elements outside the block diagonal are dropped.
__global__ __launch_bounds__(256) void gconv_wgrad_band_reduce_kernel(const float* __restrict__ P, float* __restrict__ dw, int G, int C,
                                                                      int cg) {
    const long per_slab = 9L * 4 * 1024;
    const long total = (long)(C / 128) * per_slab;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int slab = (int)(e / per_slab);
        const int rem = (int)(e - (long)slab * per_slab);
        const int tap = rem >> 12, w = (rem >> 10) & 3, row = (rem >> 5) & 31, col = rem & 31;
        if (row / cg != col / cg) continue;                  // (wave-divergent skip of whole 16-byte groups for cg >= 4)
        const float* src = P + (long)slab * G * per_slab + rem;
        const float v = ordered_sum<8, float>(G, [&](int g) { return src[(long)g * per_slab]; });
        const int co = slab * 128 + w * 32 + row;
        dw[((long)co * cg + col % cg) * 9 + tap] = v;
    }
}

struct GwPlan { int TH, bands, rows_in, Wo_pad, x_rows, dy_rows, G, slabs, pipe; size_t lds, ws; bool ok; };
GwPlan gw_plan(int B, int H, int W, int C, int stride) {
    GwPlan g = {};
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1, Wp = W + 2;
    g.Wo_pad = (Wo + 15) & ~15;
    g.slabs = C / 128;
    auto x_rows_of = [&](int th) {
        const int rows_in = (th - 1) * stride + 3;
        // ring of rows_in input rows + the slack the padded pixels of the LAST slot's row read (finite zeros)
        return rows_in * Wp + g.Wo_pad * stride + 2;
    };
    auto lds_of = [&](int th) { return (size_t)(x_rows_of(th) + th * g.Wo_pad) * GW_PITCH; };
    // two workgroups per CU (one stages while the other multiplies): bands of <= 80 KB when one output row fits that
    auto fits_regs = [&](int th) { return th * stride * Wp <= 16 * GW_XN && th * g.Wo_pad <= 16 * GW_DN; };
    int TH = Ho;
    while (TH > 1 && (lds_of(TH) > 80 * 1024 || !fits_regs(TH))) --TH;
    if (!fits_regs(TH)) {                                    // (stride-2 layers at 56 x 56: a band's new rows exceed the register image:
        TH = Ho;                                             //  staged in place, one workgroup per CU with as tall a band as fits)
        while (TH > 1 && lds_of(TH) > 160 * 1024) --TH;
    }
    g.pipe = fits_regs(TH);
    g.ok = lds_of(TH) <= 160 * 1024 && C % 128 == 0;
    g.TH = TH; g.bands = cvcl_div_up(Ho, TH); g.rows_in = (TH - 1) * stride + 3;
    g.x_rows = x_rows_of(TH); g.dy_rows = TH * g.Wo_pad; g.lds = lds_of(TH);
    const long items = (long)B * g.bands;
    long G = 512 / (g.slabs > 0 ? g.slabs : 1);
    if (G < 1) G = 1;
    if (G > items) G = items;
    g.G = (int)G;
    g.ws = (size_t)g.slabs * g.G * 9 * 4 * 1024 * sizeof(float);
    return g;
}
}  // namespace

extern "C" size_t cvcl_gconv3x3_wgrad_workspace_bytes(int B, int H, int W, int C, int stride) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const TnPlan pl = tn_plan((long)B * Ho * Wo, C, C, 9, true, TN_T);
    const size_t tap_form = tn_ws_bytes(pl, C, C, 9, true);
    const GwPlan g = gw_plan(B, H, W, C, stride);
    return g.ok && g.ws > tap_form ? g.ws : tap_form;       // (either form may run: see cvcl_gconv3x3_wgrad)
}

// bf16 only (the fp32 parity mode uses cvcl_conv_wgrad_direct)
extern "C" int cvcl_gconv3x3_wgrad(const void* x, const void* dy, float* dw, int B, int H, int W, int C, int groups, int stride,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    CVCL_CHECK_ARG(x && dy && dw && workspace && B > 0 && (stride == 1 || stride == 2) && groups > 0 && C % groups == 0,
                   "cvcl_gconv3x3_wgrad: bad args");
    const int cg = C / groups;
    CVCL_CHECK_ARG(C % TN_T == 0 && cg <= TN_T && TN_T % cg == 0, "cvcl_gconv3x3_wgrad: unsupported C=%d groups=%d", C, groups);
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long M = (long)B * Ho * Wo;
    CVCL_CHECK_ARG(M < (1L << 31), "cvcl_gconv3x3_wgrad: B * Ho * Wo must be below 2^31");
    hipStream_t st = (hipStream_t)stream;
    // one pass over the operands with all nine taps (band kernel) when a band fits the LDS; $CVCL_GCONV_WGRAD_BAND=0: the tap-at-a-time form
    static const bool band_on = cvcl_lab_int("CVCL_GCONV_WGRAD_BAND", 1) != 0;
    const GwPlan gw = gw_plan(B, H, W, C, stride);
    if (band_on && gw.ok && cg <= 32 && 32 % cg == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0) {
        if (workspace_bytes < gw.ws) {
            cvcl_set_error("cvcl_gconv3x3_wgrad: workspace too small");
            return CVCL_EWORKSPACE;
        }
        static CvclLdsAttr attr;
        if (!attr.ready()) {
            if (hipFuncSetAttribute((const void*)gconv_wgrad_band_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                cvcl_set_error("cvcl_gconv3x3_wgrad: cannot raise the dynamic LDS limit");
                return CVCL_ELAUNCH;
            }
            attr.mark();
        }
        CvclProfScope prof(stream, CVCL_K_WGRAD);
        GwDev d = {(const bf16_t*)x, (const bf16_t*)dy, (float*)workspace, B, H, W, C, stride, Ho, Wo, gw.Wo_pad, gw.TH, gw.bands,
                   gw.rows_in, gw.x_rows, gw.dy_rows, gw.pipe};
        hipLaunchKernelGGL(gconv_wgrad_band_kernel, dim3(gw.G, gw.slabs), dim3(256), gw.lds, st, d);
        CVCL_LAUNCH_CHECK();
        hipLaunchKernelGGL(gconv_wgrad_band_reduce_kernel, dim3(reduce_grid((long)gw.slabs * 9 * 4 * 1024)), dim3(256), 0, st,
                           (const float*)workspace, dw, gw.G, C, cg);
        CVCL_LAUNCH_CHECK();
        return CVCL_OK;
    }
    const TnPlan pl = tn_plan(M, C, C, 9, true, TN_T);
    if (workspace_bytes < tn_ws_bytes(pl, C, C, 9, true)) {
        cvcl_set_error("cvcl_gconv3x3_wgrad: workspace too small");
        return CVCL_EWORKSPACE;
    }
    CvclProfScope prof(stream, CVCL_K_WGRAD);
    TnDev d = {};
    d.A = (const bf16_t*)dy; d.B = (const bf16_t*)x; d.P = (float*)workspace;
    d.M = M; d.N = C; d.K = C; d.lda = C; d.ldb = C; d.S = pl.S; d.chunk = pl.chunk;
    d.tiles_n = pl.tiles_n; d.tiles_k = pl.tiles_k; d.diag = 1;
    d.taps = 9; d.Ho = Ho; d.Wo = Wo; d.Hi = H; d.Wi = W; d.stride = stride;
    hipLaunchKernelGGL(gemm_tn_bf16_kernel<true>, dim3(pl.ntile * pl.S, 9), dim3(256), 0, st, d);
    CVCL_LAUNCH_CHECK();
    hipLaunchKernelGGL(gconv_wgrad_reduce_kernel, dim3(reduce_grid((long)C * cg * 9)), dim3(256), 0, st, (const float*)workspace, dw,
                       pl.S, C, cg);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

extern "C" int cvcl_stem_im2col(const float* x_nchw, void* col_bf16, int B, int H, int W, void* stream) {
    CVCL_CHECK_ARG(x_nchw && col_bf16 && B > 0 && H % 2 == 0 && W % 2 == 0, "cvcl_stem_im2col: bad args");
    CvclProfScope prof(stream, CVCL_K_STEM);
    const long total = (long)B * (H / 2) * (W / 2) * 20;
    hipLaunchKernelGGL(stem_im2col_kernel, dim3(reduce_grid(total) * 2), dim3(256), 0, (hipStream_t)stream, x_nchw, (bf16_t*)col_bf16, B, H, W);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

