// Host side of the 8-wave 256 (224) x 256 bf16 GEMM (gemm8w_kernel.h): shape support, tile-height choice, launch.
// Same interface as cvcl_gemm (include/cvcl_hip.h); selected by the dispatcher in gemm.hip for the MFMA-bound shapes --
// the 1x1 convolutions of ResNeXt layers 2-4 (reference call site multimodal/multimodal.py:101) and the ViT linears
// (multimodal/vision_transformer_dino_mugs.py:92-94,113-115).
#include <algorithm>
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

template <int MI, int EPI, bool LNF = false, int ACT = CVCL_ACT_NONE>
int g8_launch(const g8w::Dev& d, int grid, hipStream_t stream) {
    static CvclLdsAttr attr;
    if (!attr.ready()) {
        if (hipFuncSetAttribute((const void*)g8w::gemm8w_kernel<MI, EPI, LNF, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, g8w::LDS_BYTES) != hipSuccess) {
            cvcl_set_error("cvcl_gemm8w: cannot raise the dynamic LDS limit to %d", g8w::LDS_BYTES);
            return CVCL_ELAUNCH;
        }
        attr.mark();
    }
    hipLaunchKernelGGL((g8w::gemm8w_kernel<MI, EPI, LNF, ACT>), dim3(grid), dim3(512), g8w::LDS_BYTES, stream, d);
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}

// ---- linear epilogue (EPI 1 / 2): tile height and grid ------------------------------------------------------------------------
// (Round 5, measured and NOT kept -- profiles/r05_ab_walk.txt: a REMAINDER launch.  A launch of T tiles on G workgroups costs
// ceil(T / G) rounds whatever the last round holds (ViT-B/14 proj at B = 256: 771 tiles on 256 CUs = 3.01 -> 4 rounds), so the rows
// whole rounds hold were run first and the rest as one more round of 64..192-row tiles (MI 2..6 instantiations).  Same box, two
// trunk passes in flight: C4 11.60 -> 11.66 ms, C4 at patch 14 15.30 -> 15.55 ms -- the short tiles re-read the 256-column W stage
// for few MFMAs, and with two passes in flight the CUs a lean grid leaves idle in its last round are not idle: the other pass's
// kernels run there.)
struct G8Plan { int bm, grid; };

int g8_lean(long total, int grid) {                          // the smallest grid (multiple of 8) with the same number of rounds
    static const bool lean_on = cvcl_lab_int("CVCL_LEAN_GRID", 1) != 0;
    const long rounds = (total + grid - 1) / grid;
    const int lean = (int)(((total + rounds - 1) / rounds + 7) & ~7L);
    return lean_on && lean < grid ? lean : grid;
}

G8Plan g8_plan(int M, int ncol) {
    const int cus = g8_num_cus() & ~7;
    G8Plan best = {256, 8};
    long best_cost = -1;
    for (int h : {256, 224}) {
        const long total = (long)cvcl_div_up(M, h) * ncol;
        const int g = total < cus ? (int)((total + 7) & ~7L) : cus;
        const long cost = ((total + g - 1) / g) * h;
        if (best_cost < 0 || cost < best_cost) { best = {h, g8_lean(total, g)}; best_cost = cost; }
    }
    return best;
}

// super-row height of the supertile walk (gemm8w_kernel.h): the G / 8 workgroups of an XCD multiply sr m-tiles x (G / 8) / sr
// column tiles at a time -- 8 x 4 on a full chip; with fewer column tiles than that, all of them
int g8_superrow(int grid, int ncol) {
    const int cpx = grid >> 3;
    int sr = 8;
    if (ncol * sr < cpx) sr = cvcl_div_up(cpx, ncol);
    static const int sr_lab = cvcl_lab_int("CVCL_G8_SUPERROW", 0);   // (lab: 0 = the rule above; 1 = the column-fastest list of rounds 2-4)
    return sr_lab > 0 ? sr_lab : sr;
}

int g8_linear(g8w::Dev d, const cvcl_gemm_args* a, hipStream_t st) {
    // the linear epilogue comes in two instantiations: activation (no residual) and residual (no activation) -- the only
    // combinations nn.Linear call sites on the path use (vit:92-94 fc1 + GELU, :113-115 / :146-147 proj, fc2 + residual)
    CVCL_CHECK_ARG(!(a->R && a->act != CVCL_ACT_NONE), "cvcl_gemm8w: activation and residual together are not implemented");
    const G8Plan pl = g8_plan(a->M, d.ncol);
    CvclProfScope prof(st, CVCL_K_GEMM8W);
    d.tiles_m = cvcl_div_up(d.M, pl.bm);
    d.grid_m = g8_superrow(pl.grid, d.ncol);
    const bool tall = pl.bm == 256;
    if (a->row_part) return tall ? g8_launch<8, 2, true>(d, pl.grid, st) : g8_launch<7, 2, true>(d, pl.grid, st);
    if (a->R) return tall ? g8_launch<8, 2>(d, pl.grid, st) : g8_launch<7, 2>(d, pl.grid, st);
    // the bias / activation epilogue: one instantiation per activation (gemm8w_kernel.h "ACT")
#define G8_ACT(LNF_)                                                                                                                       \
    switch (a->act) {                                                                                                                      \
    case CVCL_ACT_GELU: return tall ? g8_launch<8, 1, LNF_, CVCL_ACT_GELU>(d, pl.grid, st) : g8_launch<7, 1, LNF_, CVCL_ACT_GELU>(d, pl.grid, st); \
    case CVCL_ACT_RELU: return tall ? g8_launch<8, 1, LNF_, CVCL_ACT_RELU>(d, pl.grid, st) : g8_launch<7, 1, LNF_, CVCL_ACT_RELU>(d, pl.grid, st); \
    default: return tall ? g8_launch<8, 1, LNF_>(d, pl.grid, st) : g8_launch<7, 1, LNF_>(d, pl.grid, st);                                 \
    }
    if (a->ln_stats) { G8_ACT(true) }
    G8_ACT(false)
#undef G8_ACT
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
    d.stats_acc = a->stats && a->stats_rows == CVCL_STATS_ACCUMULATE;
    d.ln_stats = a->ln_stats; d.ln_colsum = a->ln_colsum; d.row_part = a->row_part;
    d.M = a->M; d.N = a->N; d.K = a->K; d.lda = a->lda; d.ldw = a->ldw; d.ldc = a->ldc; d.ldr = a->ldr; d.act = a->act;
    d.ncol = a->N / 256;
    d.gs = gather ? a->gather_stride : 1; d.g_hw = gather ? a->gather_ho * a->gather_wo : 1; d.g_wo = gather ? a->gather_wo : 1;
    d.g_hi = a->gather_hi; d.g_wi = a->gather_wi; d.a_rows = (int)a_rows;
    int bm, grid;
    // a plain product without BN statistics has no reason to keep a column tile per workgroup: it takes the linear epilogue's
    // supertile walk (bias NULL = 0) -- the large square products of tools/blaslt_compare.py; every convolution of the trunk asks
    // for statistics and keeps the column-fixed mapping
    if (epi == 0 && !a->stats && !a->centre && !gather && a->C) epi = 1;
    if (epi == 0) {                                          // column-fixed mapping: grid_m workgroups per column tile
        bm = cvcl_gemm8w_tile_rows(a->M, a->N);
        d.tiles_m = cvcl_div_up(a->M, bm);
        d.grid_m = g8_grid_m(d.tiles_m, d.ncol);
        if (a->stats) CVCL_CHECK_ARG(d.stats_acc || a->stats_rows >= d.grid_m, "cvcl_gemm8w: stats_rows %d < %d", a->stats_rows, d.grid_m);
        grid = d.grid_m * d.ncol;
    } else {
        return g8_linear(d, a, (hipStream_t)stream);
    }
    CvclProfScope prof(stream, CVCL_K_GEMM8W);
    hipStream_t st = (hipStream_t)stream;
    return bm == 256 ? g8_launch<8, 0>(d, grid, st) : g8_launch<7, 0>(d, grid, st);
}
