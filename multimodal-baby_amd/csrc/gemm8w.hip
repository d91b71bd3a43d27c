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

// ---- linear epilogue (EPI 1 / 2): tile height, grid and -- round 5 -- the REMAINDER launch ----------------------------------
// A launch of T tiles on G workgroups costs ceil(T / G) rounds whatever the last round holds (ViT-B/14 proj at B = 256: 771 tiles on
// 256 CUs = 3.01 -> 4 rounds).  When the last round is poorly filled, the rows are cut in two: `rows_main` = the m-tiles that
// whole rounds hold, at 256 | 224 rows per tile, and the remaining rows as ONE more round of shorter tiles (64 .. 256 rows, a
// second launch of the same kernel at a smaller MI on the same stream).  Costs are in tile ROWS per workgroup (a round of MI-row
// tiles ~ MI), the remainder's from the measured per-MI times (tools/g8_rem_bench.py): shorter tiles re-read the 256-column W
// stage for fewer MFMAs, so a 64-row round costs about what a 130-row one does, plus the second launch's fill and drain.
struct G8Plan { int bm, grid, rows_main, bm_rem, grid_rem; long cost; };

long g8_rem_cost(int bm_rem) { return (bm_rem < 128 ? 128 : bm_rem) + 40; }

int g8_lean(long total, int grid) {                          // the smallest grid (multiple of 8) with the same number of rounds
    static const bool lean_on = cvcl_lab_int("CVCL_LEAN_GRID", 1) != 0;
    const long rounds = (total + grid - 1) / grid;
    const int lean = (int)(((total + rounds - 1) / rounds + 7) & ~7L);
    return lean_on && lean < grid ? lean : grid;
}

G8Plan g8_plan(int M, int ncol, bool allow_remainder) {
    const int cus = g8_num_cus() & ~7;
    G8Plan best = {256, 8, M, 0, 0, -1};
    for (int h : {256, 224}) {                               // one launch (rounds 2-4)
        const long total = (long)cvcl_div_up(M, h) * ncol;
        const int g = total < cus ? (int)((total + 7) & ~7L) : cus;
        const long cost = ((total + g - 1) / g) * h;
        if (best.cost < 0 || cost < best.cost) best = {h, g8_lean(total, g), M, 0, 0, cost};
    }
    static const bool rem_on = cvcl_lab_int("CVCL_G8_REMAINDER", 1) != 0;
    if (!allow_remainder || !rem_on) return best;
    for (int h : {256, 224}) {                               // whole rounds + one round of shorter tiles
        const long total = (long)(M / h) * ncol;             // full tiles only
        const long r = total / cus;
        if (r < 1) continue;
        const int mt = (int)std::min<long>(r * cus / ncol, M / h);
        const int rows_main = mt * h, mr = M - rows_main;
        if (mr <= 0) continue;
        const int per_round = cus / ncol;                    // m-tiles one round holds
        if (per_round < 1) continue;
        int bm_rem = ((cvcl_div_up(mr, per_round) + 31) / 32) * 32;
        if (bm_rem < 64) bm_rem = 64;
        if (bm_rem > 256) continue;
        const long main_tiles = (long)mt * ncol, rem_tiles = (long)cvcl_div_up(mr, bm_rem) * ncol;
        const long cost = cvcl_div_up(main_tiles, cus) * h + g8_rem_cost(bm_rem);
        if (cost < best.cost)
            best = {h, g8_lean(main_tiles, cus), rows_main, bm_rem, (int)((rem_tiles + 7) & ~7L), cost};
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

template <int EPI, bool LNF>
int g8_launch_mi(int bm, const g8w::Dev& d, int grid, hipStream_t st) {
    switch (bm / 32) {
        case 8: return g8_launch<8, EPI, LNF>(d, grid, st);
        case 7: return g8_launch<7, EPI, LNF>(d, grid, st);
        default: break;
    }
    if constexpr (LNF) {                                     // the shorter remainder tiles: the ViT's folded linears only
        switch (bm / 32) {
            case 6: return g8_launch<6, EPI, LNF>(d, grid, st);
            case 5: return g8_launch<5, EPI, LNF>(d, grid, st);
            case 4: return g8_launch<4, EPI, LNF>(d, grid, st);
            case 3: return g8_launch<3, EPI, LNF>(d, grid, st);
            case 2: return g8_launch<2, EPI, LNF>(d, grid, st);
            default: break;
        }
    }
    cvcl_set_error("cvcl_gemm8w: no %d-row tile for this epilogue", bm);
    return CVCL_EINVAL;
}

int g8_linear(g8w::Dev d, const cvcl_gemm_args* a, hipStream_t st) {
    // the linear epilogue comes in two instantiations: activation (no residual) and residual (no activation) -- the only
    // combinations nn.Linear call sites on the path use (vit:92-94 fc1 + GELU, :113-115 / :146-147 proj, fc2 + residual)
    CVCL_CHECK_ARG(!(a->R && a->act != CVCL_ACT_NONE), "cvcl_gemm8w: activation and residual together are not implemented");
    const bool lnf = a->ln_stats || a->row_part;
    const G8Plan pl = g8_plan(a->M, d.ncol, lnf);
    CvclProfScope prof(st, CVCL_K_GEMM8W);
    auto launch = [&](const g8w::Dev& dd, int bm, int grid) -> int {
        if (a->ln_stats) return g8_launch_mi<1, true>(bm, dd, grid, st);
        if (a->row_part) return g8_launch_mi<2, true>(bm, dd, grid, st);
        return a->R ? g8_launch_mi<2, false>(bm, dd, grid, st) : g8_launch_mi<1, false>(bm, dd, grid, st);
    };
    g8w::Dev m = d;
    m.M = pl.rows_main; m.a_rows = pl.rows_main;
    m.tiles_m = cvcl_div_up(m.M, pl.bm);
    m.grid_m = g8_superrow(pl.grid, d.ncol);
    int rc = launch(m, pl.bm, pl.grid);
    if (rc != CVCL_OK || !pl.bm_rem) return rc;
    g8w::Dev r = d;                                          // the remaining rows: every row-indexed operand moves down by rows_main
    const long o = pl.rows_main;
    r.A = d.A + o * d.lda; r.C = d.C + o * d.ldc;
    if (d.R) r.R = d.R + o * d.ldr;
    if (d.ln_stats) r.ln_stats = d.ln_stats + o * 2;
    if (d.row_part) r.row_part = d.row_part + o * (d.N >> 6) * 2;
    r.M = d.M - pl.rows_main; r.a_rows = r.M;
    r.tiles_m = cvcl_div_up(r.M, pl.bm_rem);
    r.grid_m = g8_superrow(pl.grid_rem, d.ncol);
    return launch(r, pl.bm_rem, pl.grid_rem);
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

// the launch plan of the linear epilogue for an [M, N] output under the current CU share: plan5 = {tile rows, workgroups, rows of the
// main launch, tile rows of the remainder launch (0 = none), its workgroups}; folded = the ln_stats / row_part epilogues
extern "C" int cvcl_gemm8w_linear_plan(int M, int N, int folded, int* plan5) {
    CVCL_CHECK_ARG(plan5 && M >= 1 && N >= 256 && N % 256 == 0, "cvcl_gemm8w_linear_plan: bad args");
    const G8Plan pl = g8_plan(M, N / 256, folded != 0);
    plan5[0] = pl.bm; plan5[1] = pl.grid; plan5[2] = pl.rows_main; plan5[3] = pl.bm_rem; plan5[4] = pl.grid_rem;
    return CVCL_OK;
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
    // a plain product without BN statistics has no reason to keep a column tile per workgroup: it takes the linear epilogue's
    // supertile walk (bias NULL = 0) -- the large square products of tools/blaslt_compare.py; every convolution of the trunk asks
    // for statistics and keeps the column-fixed mapping
    if (epi == 0 && !a->stats && !a->centre && !gather && a->C) epi = 1;
    if (epi == 0) {                                          // column-fixed mapping: grid_m workgroups per column tile
        bm = cvcl_gemm8w_tile_rows(a->M, a->N);
        d.tiles_m = cvcl_div_up(a->M, bm);
        d.grid_m = g8_grid_m(d.tiles_m, d.ncol);
        if (a->stats) CVCL_CHECK_ARG(a->stats_rows >= d.grid_m, "cvcl_gemm8w: stats_rows %d < %d", a->stats_rows, d.grid_m);
        grid = d.grid_m * d.ncol;
    } else {
        return g8_linear(d, a, (hipStream_t)stream);
    }
    CvclProfScope prof(stream, CVCL_K_GEMM8W);
    hipStream_t st = (hipStream_t)stream;
    return bm == 256 ? g8_launch<8, 0>(d, grid, st) : g8_launch<7, 0>(d, grid, st);
}
