// Host side of the 8-wave 256 (224) x 256 bf16 GEMM (gemm8w_kernel.h): shape support, tile-height choice, launch.
// Same interface as cvcl_gemm (include/cvcl_hip.h); selected by the dispatcher in gemm.hip for the MFMA-bound shapes --
// the 1x1 convolutions of ResNeXt layers 2-4 (reference call site multimodal/multimodal.py:101) and the ViT linears
// (multimodal/vision_transformer_dino_mugs.py:92-94,113-115).
#include <cstdlib>

#include "gemm8w_kernel.h"

namespace {

int g8_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        const int cap = cvcl_lab_int("CVCL_G8_CUS", 0);      // (lab: run the 8-wave kernels on part of the chip)
        if (cap > 0 && cap < n) n = cap;
    }
    const int share = cvcl_gemm_cu_share();                  // cvcl_set_gemm_cu_share: the host's co-scheduling hint
    return share > 0 && share < n ? share : n;
}

// grid rows (workgroups per column tile): a multiple of 8 so that the column tiles of one m-tile share an XCD, at most one
// workgroup per CU in total, no more than there are m-tiles (rounded up to 8)
// (round 4) ... and no more workgroups than the number of rounds needs: 224 m-tiles on 64 grid rows are 3.5 -> 4 rounds, which 56
// grid rows also do in 4 -- the same launch time, but the 32 CUs left out are free for the OTHER trunk stream's kernels for the
// whole launch instead of idling through the tail of the last round.
int g8_grid_m(int tiles_m, int ncol) {
    int gm = (g8_num_cus() / ncol) & ~7;
    if (gm < 8) gm = 8;
    const int need = (tiles_m + 7) & ~7;
    if (gm > need) gm = need;
    static const bool lean_on = cvcl_lab_int("CVCL_LEAN_GRID", 1) != 0;
    const int rounds = cvcl_div_up(tiles_m, gm);
    const int lean = (cvcl_div_up(tiles_m, rounds) + 7) & ~7;      // the smallest multiple of 8 with the same number of rounds
    return lean_on && lean < gm ? lean : gm;
}

template <int MI, int EPI, bool LNF = false>
int g8_launch(const g8w::Dev& d, int grid, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)g8w::gemm8w_kernel<MI, EPI, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, g8w::LDS_BYTES) != hipSuccess) {
            cvcl_set_error("cvcl_gemm8w: cannot raise the dynamic LDS limit to %d", g8w::LDS_BYTES);
            return CVCL_ELAUNCH;
        }
        attr = true;
    }
    hipLaunchKernelGGL((g8w::gemm8w_kernel<MI, EPI, LNF>), dim3(grid), dim3(512), g8w::LDS_BYTES, stream, d);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

}  // namespace

// shapes the kernel accepts (the dispatcher adds its own policy on top)
extern "C" int cvcl_gemm8w_supported(int M, int N, int K, int lda, int ldw, int ldc) {
    return M >= 1 && N % 256 == 0 && N / 256 <= 32 && N <= g8w::MAX_BIAS_N && K % 128 == 0 && K >= 128 && lda % 8 == 0 && ldw % 8 == 0 &&
           ldc % 8 == 0;
}

// Tile height for an [M, N] output: 256 rows (MI 8) or 224 (MI 7).  M of the ResNeXt activations is 49 * 2^k: 224 = 7 * 32
// divides it and leaves 7/8 of the workgroup slots busy in the last round where 256-row tiles leave ~5/8.  Picks the height
// with the smaller (rounds x height); ties -> 256.
extern "C" int cvcl_gemm8w_tile_rows(int M, int N) {
    const int ncol = N / 256;
    int best = 256;
    long best_cost = -1;
    for (int bm : {256, 224}) {
        const int tiles = cvcl_div_up(M, bm), gm = g8_grid_m(tiles, ncol);
        const long cost = (long)cvcl_div_up(tiles, gm) * bm;
        if (best_cost < 0 || cost < best_cost) { best = bm; best_cost = cost; }
    }
    return best;
}

// BN-statistics rows cvcl_gemm8w writes for an [M, N] output (one per grid row)
extern "C" int cvcl_gemm8w_stats_rows(int M, int N) {
    const int bm = cvcl_gemm8w_tile_rows(M, N);
    return g8_grid_m(cvcl_div_up(M, bm), N / 256);
}

// epi 0: convolution epilogue (round + BN partial sums; C may be NULL = statistics only); epi 1: bias / activation / residual
extern "C" int cvcl_gemm8w(int epi, const cvcl_gemm_args* a, void* stream) {
    CVCL_CHECK_ARG(a && a->A && a->W && (a->C || a->stats), "cvcl_gemm8w: null operand");
    CVCL_CHECK_ARG(cvcl_gemm8w_supported(a->M, a->N, a->K, a->lda, a->ldw, a->ldc) && !a->a_scale &&
                       !a->exp_scale && !a->c_scale && !a->C_pre && !a->G,
                   "cvcl_gemm8w: unsupported shape / options (M %d N %d K %d)", a->M, a->N, a->K);
    CVCL_CHECK_ARG(epi == 0 || epi == 1, "cvcl_gemm8w: epilogue %d", epi);
    CVCL_CHECK_ARG(epi == 1 || (!a->bias && !a->R && a->act == CVCL_ACT_NONE), "cvcl_gemm8w: epilogue 0 takes no bias / activation / residual");
    CVCL_CHECK_ARG(epi == 0 || (a->C && !a->stats), "cvcl_gemm8w: epilogue 1 writes C and takes no statistics");
    CVCL_CHECK_ARG(epi == 0 || !a->centre, "cvcl_gemm8w: centre goes with the convolution epilogue only");
    CVCL_CHECK_ARG(!a->ln_stats || (epi == 1 && !a->R && a->ln_colsum && a->bias && !a->row_part),
                   "cvcl_gemm8w: ln_stats goes with the bias / activation epilogue and needs ln_colsum and the folded bias");
    CVCL_CHECK_ARG(!a->row_part || (epi == 1 && a->R), "cvcl_gemm8w: row_part goes with the bias + residual epilogue");
    CVCL_CHECK_ARG(!a->ln_colsum || a->ln_stats, "cvcl_gemm8w: ln_colsum without ln_stats");
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    CVCL_CHECK_ARG(al16(a->A) && al16(a->W) && al16(a->C) && al16(a->R) && al16(a->bias) && (!a->R || a->ldr % 8 == 0) &&
                       al16(a->ln_stats) && al16(a->ln_colsum) && (((uintptr_t)a->row_part & 7) == 0),
                   "cvcl_gemm8w: operands must be 16-byte aligned");
    const bool gather = a->gather_stride > 1;
    if (gather)
        CVCL_CHECK_ARG(a->gather_ho > 0 && a->gather_wo > 0 && a->M % (a->gather_ho * a->gather_wo) == 0 &&
                           (a->gather_ho - 1) * a->gather_stride < a->gather_hi && (a->gather_wo - 1) * a->gather_stride < a->gather_wi,
                       "cvcl_gemm8w: gather geometry");
    const long a_rows = gather ? (long)(a->M / (a->gather_ho * a->gather_wo)) * a->gather_hi * a->gather_wi : a->M;
    CVCL_CHECK_ARG(a_rows * a->lda < (1L << 31) && (long)a->N * a->ldw < (1L << 31), "cvcl_gemm8w: operand offsets must fit 31 bits");
    g8w::Dev d;
    d.A = (const bf16_t*)a->A; d.W = (const bf16_t*)a->W; d.C = (bf16_t*)a->C; d.R = (const bf16_t*)a->R;
    d.bias = a->bias; d.stats = a->stats; d.centre = a->centre;
    d.ln_stats = a->ln_stats; d.ln_colsum = a->ln_colsum; d.row_part = a->row_part;
    d.M = a->M; d.N = a->N; d.K = a->K; d.lda = a->lda; d.ldw = a->ldw; d.ldc = a->ldc; d.ldr = a->ldr; d.act = a->act;
    d.ncol = a->N / 256;
    d.gs = gather ? a->gather_stride : 1; d.g_hw = gather ? a->gather_ho * a->gather_wo : 1; d.g_wo = gather ? a->gather_wo : 1;
    d.g_hi = a->gather_hi; d.g_wi = a->gather_wi; d.a_rows = (int)a_rows;
    int bm, grid;
    if (epi == 0) {                                          // column-fixed mapping: grid_m workgroups per column tile
        bm = cvcl_gemm8w_tile_rows(a->M, a->N);
        d.tiles_m = cvcl_div_up(a->M, bm);
        d.grid_m = g8_grid_m(d.tiles_m, d.ncol);
        if (a->stats) CVCL_CHECK_ARG(a->stats_rows >= d.grid_m, "cvcl_gemm8w: stats_rows %d < %d", a->stats_rows, d.grid_m);
        grid = d.grid_m * d.ncol;
    } else {                                                 // flat mapping: every CU takes tiles q, q + grid, ... of the column-fastest list
        bm = 256;
        long best = -1;
        for (int h : {256, 224}) {
            const long total = (long)cvcl_div_up(a->M, h) * d.ncol;
            const long g = total < g8_num_cus() ? ((total + 7) & ~7L) : (g8_num_cus() & ~7);
            const long cost = ((total + g - 1) / g) * h;
            if (best < 0 || cost < best) { best = cost; bm = h; }
        }
        d.tiles_m = cvcl_div_up(a->M, bm);
        const long total = (long)d.tiles_m * d.ncol;
        grid = total < g8_num_cus() ? (int)((total + 7) & ~7L) : (g8_num_cus() & ~7);
        {                                                    // as g8_grid_m: the smallest grid with the same number of rounds
            static const bool lean_on = cvcl_lab_int("CVCL_LEAN_GRID", 1) != 0;
            const long rounds = (total + grid - 1) / grid;
            const int lean = (int)(((total + rounds - 1) / rounds + 7) & ~7L);
            if (lean_on && lean < grid) grid = lean;
        }
        d.grid_m = 0;
    }
    CvclProfScope prof(stream, CVCL_K_GEMM8W);
    // the linear epilogue comes in two instantiations: activation (no residual) and residual (no activation) -- the only
    // combinations nn.Linear call sites on the path use (vit:92-94 fc1 + GELU, :113-115 / :146-147 proj, fc2 + residual)
    CVCL_CHECK_ARG(epi == 0 || !(a->R && a->act != CVCL_ACT_NONE), "cvcl_gemm8w: activation and residual together are not implemented");
    const int k = epi == 0 ? 0 : (a->R ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
    if (a->ln_stats) return bm == 256 ? g8_launch<8, 1, true>(d, grid, st) : g8_launch<7, 1, true>(d, grid, st);
    if (a->row_part) return bm == 256 ? g8_launch<8, 2, true>(d, grid, st) : g8_launch<7, 2, true>(d, grid, st);
    if (bm == 256) return k == 0 ? g8_launch<8, 0>(d, grid, st) : k == 1 ? g8_launch<8, 1>(d, grid, st) : g8_launch<8, 2>(d, grid, st);
    return k == 0 ? g8_launch<7, 0>(d, grid, st) : k == 1 ? g8_launch<7, 1>(d, grid, st) : g8_launch<7, 2>(d, grid, st);
}
