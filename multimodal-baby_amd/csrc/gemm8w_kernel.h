// 256 (224 / 192) x 256 bf16 GEMM tile for the MFMA-bound shapes: one 8-wave workgroup per CU, (16 MI) x 64 per wave.
//
//      C[M,N] = A[M,K] . W[N,K]^T          (A, W row-major, K contiguous; N % 256 == 0, K % 128 == 0)
//
// Why: the 128 x 128 x 64 two-buffer kernel of gemm.hip reads 1 LDS fragment per MFMA, pulls 64 B/clk/CU through the L1 fill
// path and meets a workgroup barrier every 16 MFMAs (512 cycles) with one K tile of prefetch -- it sits at ~0.3 of the bf16
// peak on K >= 512 shapes (profiles/r01_pmc_gemm_sq.json).  Here a wave owns a (16 MI) x 64 block of the output on
// v_mfma_f32_16x16x32_bf16 (MI x 4 accumulator tiles = 128 registers at MI = 8, two waves per SIMD): 0.375 fragment reads per
// MFMA, 32 B/clk/CU of operand traffic, one barrier per 32 MFMAs of a wave (1024 cycles of a SIMD's matrix pipe).
//
// Pipeline (per workgroup, persistent over its output tiles; "stage" = 32 k = 64 bytes per operand row):
//   * 4-stage LDS ring, 32 KiB per stage (A: 256 rows x 64 B, W: 256 rows x 64 B), filled by global_load_lds_dwordx4.
//     A wave instruction lands 16 rows x 64 B; the 16-byte chunk c of row r sits at chunk position c ^ swz((r >> 2) & 3)
//     (swizzle applied on the SOURCE address, the LDS image stays lane-linear), which makes the ds_read_b128 fragment reads
//     of a 16-row block (lane -> row lane & 15, chunk lane >> 4) conflict-free.
//   * fragments are double-buffered in registers.  Iteration g multiplies stage g out of registers; in its middle it waits
//     (counted vmcnt) for stage g+1, passes the one barrier of the iteration, issues the fragment reads of stage g+1 and the
//     loads of stage g+4 (into the buffer stage g occupied: every wave finished reading it before the barrier), and the second
//     half of the MFMAs covers those latencies.  3.5 stages of loads are in flight.
//   * vmcnt retires in issue order for loads and stores alike: the wait for stage g+1 allows exactly the 8 younger load
//     instructions (+ the epilogue's stores when one was issued in between).  Stages past the end of the workgroup's work are
//     issued as harmless re-loads so that the count stays constant.
//   * epilogue per 16-row block through 2 KiB of wave-private LDS: rounded to bf16, written as full 128-byte row segments,
//     BN partial sums of the stored values accumulated per lane (8 fixed columns) across all tiles of the workgroup.
#pragma once
#include <type_traits>

#include "cvcl_common.h"

namespace g8w {

constexpr int BN = 256;
constexpr int BK = 32;
constexpr int NSTAGE = 4;
constexpr int A_BYTES = 16384;                    // 256 rows x 64 B
constexpr int STAGE_BYTES = 2 * A_BYTES;          // + W: 256 rows x 64 B
constexpr int STG_BYTES = 2048;                   // per-wave epilogue staging: 16 rows x 128 B
constexpr int ACC_OFF = NSTAGE * STAGE_BYTES + 8 * STG_BYTES;       // [8 waves][2][64] f32 BN partial sums / [N] f32 bias
constexpr int LDS_BYTES = 160 * 1024;                               // everything: 128 KiB ring + 16 KiB staging + 16 KiB
constexpr int MAX_BIAS_N = (LDS_BYTES - ACC_OFF) / 4;               // 4096

// EPI 0: C = round(acc) (+ BN partial sums when stats != nullptr; C may be nullptr: statistics only)
// EPI 1: C = round(act(acc + bias))                 (nn.Linear: bias / ReLU / GELU)
// EPI 2: C = round(round(acc + bias) + R)            (nn.Linear + residual: the residual rows are fetched two 16-row blocks ahead)
// LNF (round 4; "LayerNorm folded", reference vision_transformer_dino_mugs.py:136-149 Block: x + attn(norm1(x)), x + mlp(norm2(x))):
//   EPI 1 + LNF: the A rows are the RAW residual stream x and W carries gamma (W' = W diag(gamma)): with the row's
//                (rstd, -mean rstd) = ln_stats[m][0..1], the column sums s[n] = sum_k W'[n][k] = ln_colsum and the folded bias
//                b'[n] = bias[n] + sum_k W[n][k] beta[k] (passed as bias):  C = round(act(rstd acc - mean rstd s[n] + b'[n]))
//                = act(LayerNorm(x) W^T + b) with no normalised copy of x ever stored (the 24 LayerNorm passes of a ViT-B go).
//   EPI 2 + LNF: besides C the kernel leaves, per output row and 64-column strip, (sum, sum of squares) of the stored (rounded,
//                residual added) values: row_part[m][N / 64][2]; cvcl_row_stats_finalize turns them into the next LNF
//                launch's ln_stats.
struct Dev {
    const bf16_t* A; const bf16_t* W; bf16_t* C; const bf16_t* R;
    const float* bias; float* stats;
    int stats_acc;                  // stats is an int64 accumulator [8][2][N] (CVCL_STATS_ACCUMULATE), not partial rows
    const float* ln_stats; const float* ln_colsum; float* row_part;     // LNF (see above)
    const float* centre;            // EPI 0: storage centre of the output (NULL = 0): accumulators start at -centre[n]
    int M, N, K, lda, ldw, ldc, ldr, act;
    int tiles_m, grid_m, ncol;
    // row gather of a strided 1x1 convolution: output row m = (b, oy, ox) reads A row (b, oy * gs, ox * gs); gs <= 1 = off
    int gs, g_hw, g_wo, g_hi, g_wi;
    int a_rows;                     // rows of A (= M without the gather)
};

__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// LDS swizzle: the 16-byte chunk c of row r sits at chunk position c ^ swz((r >> 2) & 3), swz = {0, 2, 3, 1}.  A fragment read
// (ds_read_b128: lane -> row lane & 15, chunk lane >> 4) is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
// and the same + 32: each group holds the four 4-row sets g = 0..3 with chunks (c, c, c+1, c+1) in the order (0, 3, 1, 2) or
// (c+1 for 0, 3; c for 1, 2); with this permutation the four sets land on four different chunk positions, i.e. the 16 lanes hit
// the 16 different 16-byte slots of a 256-byte bank row.  (The plain c ^ g -- right for the 32-row fragments of the 32x32 MFMA --
// is 2-way conflicted here: SQ_LDS_BANK_CONFLICT was 50 % of SQ_LDS_IDX_ACTIVE.)
__device__ __forceinline__ int swz(int g) { return (0x78 >> (2 * g)) & 3; }

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// s_waitcnt vmcnt(n) for a count that is a constant only after unrolling (the switch folds), tied to the value it makes valid
__device__ __forceinline__ void wait_vm_value(int n, u32x4& v) {
#define CVCL_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(v) : : "memory"); break;
    switch (n) {
        CVCL_W(0) CVCL_W(1) CVCL_W(2) CVCL_W(3) CVCL_W(4) CVCL_W(5) CVCL_W(6) CVCL_W(7) CVCL_W(8) CVCL_W(9) CVCL_W(10) CVCL_W(11)
        CVCL_W(12) CVCL_W(13) CVCL_W(14) CVCL_W(15) CVCL_W(16) CVCL_W(17) CVCL_W(18) CVCL_W(19) CVCL_W(20) CVCL_W(21) CVCL_W(22)
        CVCL_W(23) CVCL_W(24) CVCL_W(25) CVCL_W(26) CVCL_W(27) CVCL_W(28) CVCL_W(29) CVCL_W(30) CVCL_W(31) CVCL_W(32) CVCL_W(33)
        CVCL_W(34) CVCL_W(35) CVCL_W(36) CVCL_W(37) CVCL_W(38) CVCL_W(39) CVCL_W(40)
        default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); break;
    }
#undef CVCL_W
}

// What bounds it, what was tried (pipeline ablations, 4-wave / 8-phase / staggered variants, store flavours): DESIGN.md section 5;
// the experiment switches live in tools/gemm_lab/gemm8w_lab_kernel.h, not here.
// ACT (round 5: a template parameter, EPI 1 only): the activation used to be tested per PAIR of outputs on the run-time p.act -- two
// scalar compares and branches around every pair, which cut the epilogue into basic blocks of one pair each: the GELU of a pair ran
// as one dependent chain (s_nop between its packed operations in the ISA) with no other pair to interleave.
template <int MI, int EPI, bool LNF, int ACT = CVCL_ACT_NONE>
__global__ __launch_bounds__(512, 2) void gemm8w_kernel(Dev p) {
    constexpr int BM = MI * 32;
    static_assert(!LNF || EPI >= 1, "LNF goes with the linear epilogues");
    static_assert(ACT == CVCL_ACT_NONE || EPI == 1, "activations go with the bias epilogue");
    constexpr int ESTORES = MI * 2 + (LNF && EPI == 2 ? MI * 2 : 0);     // global stores per lane per full tile epilogue
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                 // waves w and w + 4 (one SIMD) own the two row halves of a column strip

    // ---- this workgroup's output tiles ----
    // EPI 0 (BN statistics per column): fixed column tile j, m-tiles i, i + grid_m, ...; blocks of one XCD (blockIdx % 8) hold
    //   the column tiles of the same m-tiles, so an m-tile's A rows are fetched into one L2 once.
    // EPI 1 / 2 (linear epilogue), round 5 -- the SUPERTILE walk.  The tiles are put in one list: super-rows of `sr` m-tiles
    //   (p.grid_m), inside a super-row column by column (serpentine: odd super-rows right to left), inside a column the sr m-tiles.
    //   XCD x (= blockIdx % 8) owns the x-th eighth of that list and its G / 8 workgroups take consecutive entries, G / 8 at a time:
    //   the tiles an XCD multiplies AT ONE TIME form a block of sr m-tiles x (G / 8) / sr columns -- each of its A tiles is shared
    //   by (G / 8) / sr CUs and each W tile by sr CUs of the same L2 (sr = 8, 32 CUs: 8 + 4 operand tiles feed 32 output tiles;
    //   the column-fastest list of rounds 2-4 gave an XCD 3.5 m-tiles x all N / 256 columns: 12.5 operand tiles at N = 2304, 33 at
    //   N = 8192 -- profiles/r04_pmc_c4_summary.json: qkv pulled 3.7 x, fc1 6.5 x its operand bytes across the L2) -- and from one
    //   round to the next the A group stays while the W slab moves on.  Every CU busy whatever N / 256 is, as before.
    constexpr bool LIN = EPI >= 1, RES = EPI == 2;
    constexpr bool FLAT = LIN;
    const int b = blockIdx.x;
    const int G = gridDim.x;
    int ti, tj, nt;
    int L0 = 0;                                              // FLAT: this workgroup's first list entry; the next is cpx further
    const int cpx = G >> 3;
    // list entry -> (m-tile, column tile)
    auto decode = [&](int L, int& i_out, int& j_out) __attribute__((always_inline)) {
        const int per = p.grid_m * p.ncol;
        const int sr = L / per, rem = L - sr * per;
        const int ah = min(p.grid_m, p.tiles_m - sr * p.grid_m);
        const int col = rem / ah;
        i_out = sr * p.grid_m + (rem - col * ah);
        j_out = (sr & 1) ? p.ncol - 1 - col : col;
    };
    if constexpr (FLAT) {
        const int xcd = b & 7;
        const int total = p.tiles_m * p.ncol;
        const int S0 = (int)(((long)xcd * total) >> 3), S1 = (int)(((long)(xcd + 1) * total) >> 3);
        L0 = S0 + (b >> 3);
        nt = L0 < S1 ? (S1 - L0 + cpx - 1) / cpx : 0;
        ti = 0; tj = 0;
        if (nt > 0) decode(L0, ti, tj);
    } else {
        const int xcd = b & 7, s = b >> 3;
        tj = s % p.ncol;
        ti = (s / p.ncol) * 8 + xcd;
        nt = ti < p.tiles_m ? (p.tiles_m - ti + p.grid_m - 1) / p.grid_m : 0;
    }
    const int KS = p.K / BK;
    const int S = nt * KS;
    if (S == 0) {                                            // more workgroup rows than m-tiles: an all-zero statistics row
        if (EPI == 0 && p.stats && !p.stats_acc && ti < p.grid_m && tid < BN) {
            p.stats[((long)ti * 2 + 0) * p.N + tj * BN + tid] = 0.f;
            p.stats[((long)ti * 2 + 1) * p.N + tj * BN + tid] = 0.f;
        }
        return;
    }
    // EPI 0: tile k of this workgroup -> tile k + 1 is grid_m m-tiles further in the same column
    const int step_i = p.grid_m;

    const bf16_t* __restrict__ A = p.A;
    const bf16_t* __restrict__ W = p.W;

    // ---- staging: wave w lands row blocks 2w, 2w+1 (16 rows x 64 B each) of both operands per stage ----
    const int srow = lane >> 2;
    const int slog = (lane & 3) ^ swz((lane >> 4) & 3);     // logical chunk fetched by this lane (it lands at chunk lane & 3)
    unsigned w_off[2], a_raw[2];                            // element offsets of this lane's two rows of W / of A (current load tile)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        w_off[j] = (unsigned)(tj * BN + (wave * 2 + j) * 16 + srow) * (unsigned)p.ldw + slog * 8;
        int r = (wave * 2 + j) * 16 + srow;
        if (r >= BM) r = BM - 1;                            // BM < 256: rows of the unused part of the A region
        a_raw[j] = (unsigned)(ti * BM + r) * (unsigned)p.lda + slog * 8;
    }
    // gathered rows: no uniform step from tile to tile -- the two offsets are recomputed per tile (two integer divisions per
    // row, once per K / 32 stages)
    auto gathered = [&](int i_tile, int j) __attribute__((always_inline)) -> unsigned {
        int r = (wave * 2 + j) * 16 + srow;
        if (r >= BM) r = BM - 1;
        const unsigned m = (unsigned)min(i_tile * BM + r, p.M - 1);
        const unsigned bi = m / (unsigned)p.g_hw, rem = m - bi * (unsigned)p.g_hw;
        const unsigned oy = rem / (unsigned)p.g_wo, ox = rem - oy * (unsigned)p.g_wo;
        return ((bi * p.g_hi + oy * p.gs) * p.g_wi + ox * p.gs) * (unsigned)p.lda + slog * 8;
    };
    if (p.gs > 1) { a_raw[0] = gathered(ti, 0); a_raw[1] = gathered(ti, 1); }
    // the next tile is a uniform step further; rows past M (ragged last tile) are clamped to an address inside the last row
    // (any valid address will do: those rows are masked at the store) -- no per-lane state beyond the offsets
    const unsigned a_unit = (unsigned)BM * (unsigned)p.lda;
    const unsigned a_lim = (unsigned)(p.a_rows - 1) * (unsigned)p.lda + 24;
    int l_t = 0, l_ks = 0, l_j = tj, l_i = ti;                         // (tile, k stage) of the next stage to load; its column tile
    auto issue = [&](int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE_BYTES + wave * 2048;
        const int k0 = l_ks * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(A + min(a_raw[j], a_lim) + k0, base + j * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(W + w_off[j] + k0, base + A_BYTES + j * 1024);
    };
    // next stage to load; kept apart from issue() so that its branch does not split the loads from the MFMAs around them.
    // Past the end of the workgroup's work the last tile is re-loaded (never read)
    auto advance = [&]() __attribute__((always_inline)) {
        if constexpr (LNF && EPI == 1) {             // (here, behind the MFMAs, so that the branch does not split issue()'s region)
            // with the first stage of a tile, its epilogue operands: waves 0 / 1 land the 256 folded-bias / column-sum values of
            // the column tile, waves 2 / 3 the (rstd, -mean rstd) pairs of the upper / lower 128 rows, into the slot of the
            // tile's parity (1 KiB each; read K / 32 >= 4 stages later, behind several counted waits and barriers).  One load
            // more in these waves' queues only makes the next three counted waits conservative.
            if (l_ks == 0 && wave < 4) {
                char* dst = smem + ACC_OFF + (l_t & 1) * 4096 + wave * 1024;
                // (ragged last tile: whole 2-row granules past M are clamped to the granule that holds row M - 1 -- masked at the
                // store; that granule may reach one row past ln_stats[M - 1]: cvcl_hip.h asks for an even number of rows)
                const float* src = wave == 0 ? p.bias + l_j * BN + lane * 4 : wave == 1 ? p.ln_colsum + l_j * BN + lane * 4
                                 : p.ln_stats + (long)min(l_i * BM + (wave - 2) * 128 + lane * 2, (p.M - 1) & ~1) * 2;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                 (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
            }
        }
        if (++l_ks == KS) {
            l_ks = 0;
            if (l_t + 1 < nt) {
                ++l_t;
                if constexpr (FLAT) {
                    decode(L0 + l_t * cpx, l_i, l_j);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int rw = (wave * 2 + j) * 16 + srow;
                        a_raw[j] = (unsigned)(l_i * BM + min(rw, BM - 1)) * (unsigned)p.lda + slog * 8;
                        w_off[j] = (unsigned)(l_j * BN + rw) * (unsigned)p.ldw + slog * 8;
                    }
                } else {
                    l_i += step_i;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (p.gs > 1) a_raw[j] = gathered(l_i, j);
                        else a_raw[j] += (unsigned)step_i * a_unit;
                    }
                }
            }
        }
    };

    // ---- fragment addressing: lane -> row lane & 15 of a 16-row block, logical chunk lane >> 4 ----
    const int f_off = (lane & 15) * 64 + (((lane >> 4) ^ swz((lane >> 2) & 3)) << 4);
    const int a_base = wm * (BM / 2) * 64 + f_off;
    const int w_base = A_BYTES + wn * 64 * 64 + f_off;

    bf16x8 fa[2][MI], fw[2][4];
    f32x4 acc[4][MI];
    auto read_frags = [&](int buf, auto P) __attribute__((always_inline)) {
        constexpr int q = decltype(P)::value;
        const char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) fw[q][ni] = *reinterpret_cast<const bf16x8*>(sb + w_base + ni * 1024);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[q][mi] = *reinterpret_cast<const bf16x8*>(sb + a_base + mi * 1024);
    };
    auto mma_half = [&](auto P, auto HALF) __attribute__((always_inline)) {
        constexpr int q = decltype(P)::value, h = decltype(HALF)::value;
#pragma unroll
        for (int ni = 2 * h; ni < 2 * h + 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[q][ni], fa[q][mi], acc[ni][mi], 0, 0, 0);
    };
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Per-wave LDS slots next to the ring (registers are all spoken for: 128 accumulators + 96 fragment registers per lane):
    // EPI 0: BN partial sums [wave][2][64], added to once per tile in a fixed order (deterministic);
    // EPI 1: the column tile's 256 bias values, read back in the accumulator layout by the epilogue.
    float* lds_acc = reinterpret_cast<float*>(smem + ACC_OFF);
    if constexpr (LIN && LNF && EPI == 1) {
        // (per-tile epilogue operands arrive by DMA with the tile's first stage: issue())
    } else if constexpr (LIN) {                              // the whole bias vector (N <= 4096 floats fit beside the ring)
        for (int i = tid; i < p.N; i += 512) lds_acc[i] = p.bias ? p.bias[i] : 0.f;
    } else {
        lds_acc[tid] = 0.f;
        lds_acc[tid + 512] = 0.f;
        // centred storage: -centre of the column tile's 256 columns behind the partial sums; every accumulator tile starts from it
        if (tid < BN) lds_acc[1024 + tid] = p.centre ? -p.centre[tj * BN + tid] : 0.f;
    }
    // (ordered before the first epilogue by the prologue's barrier)
    // this lane's four columns of accumulator tile ni are cen_of() + ni * 16.  The address is re-derived from the thread id behind
    // an opaque copy wherever it is used: hoisted out of the K loop it is one more loop-invariant VGPR in a kernel that sits at
    // exactly 256, and the compiler then spills to scratch -- whose loads would also sit in the counted vmcnt queue
    auto cen_of = [&]() __attribute__((always_inline)) -> const float* {
        int t = tid;
        asm volatile("" : "+v"(t));
        return lds_acc + 1024 + wn * 64 + ((t >> 4) & 3) * 4;
    };

    char* stg = smem + NSTAGE * STAGE_BYTES + wave * STG_BYTES;
    const int e_row = lane & 15;                             // accumulator layout: m = mi*16 + (lane & 15), n = ni*16 + (lane >> 4)*4 + e
    const int e_wchunk = lane >> 5, e_wsub = ((lane >> 4) & 1) * 8;
    const int e_wsw = (e_row >> 1) & 7;
    const int r_chunk = lane & 7, r_row0 = lane >> 3;        // read-back: row 8j + (lane >> 3), 16-byte chunk lane & 7

    // -> a lower bound on the VMEM instructions this call issued (exact for a full tile with stores and no residual)
    // FULL (round 5): a tile with all BM rows inside M and an output to store runs a copy of the epilogue WITHOUT the per-row-group
    // tests (m < M, C != NULL): every such test is a scalar branch that ends a basic block, and the blocks -- one row group each --
    // were scheduled one by one (the staging read-back of a group could not move above the stores of the previous one)
    auto epilogue_body = [&](int m0, int n0, int parity, auto FULL, const bool full_rt) __attribute__((always_inline)) -> int {
        constexpr int fmode = decltype(FULL)::value;           // 1 / 0: known at compile time; -1: the run-time value (RES)
        const bool full = fmode < 0 ? full_rt : fmode == 1;
        float st_sum[8], st_sq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { st_sum[e] = 0.f; st_sq[e] = 0.f; }
        // EPI 2: the residual rows of block mi are in flight from block mi - RAHEAD on (an exposed load per block cost ~0.7 us x MI
        // per tile: the proj linear of ViT-B ran 104 us against 72 us of K loop).
        // Round 5 -- the loads are inline asm with HAND-COUNTED waits.  As plain C++ loads they were waited for by the compiler,
        // and hipcc answers "loads and stores both outstanding" with s_waitcnt vmcnt(0) (it models their completion as unordered):
        // every use of a residual row -- 2 MI per tile, each behind the previous row group's stores -- DRAINED the queue: the
        // acknowledgement of the stores just issued plus the three K stages of prefetch in flight, ~0.7 us x 2 MI per tile (proj
        // 102 us in the network against 64 us with the plain epilogue).  Loads and stores do retire in issue order (the K loop's
        // counted waits rest on the same fact), so the wait for rows (mi, j) allows exactly the operations issued after their
        // load: the later residual loads and the stores of the row groups before (res_younger below).  RAHEAD 2 -> as many
        // blocks as the registers the finished accumulators leave hold.
        constexpr int RAHEAD = MI <= 7 ? (LNF ? 5 : MI) : (LNF ? 4 : 2);      // (MI = 8: what compiles without scratch)
        constexpr int RSTORES = LNF ? 2 : 1;                     // stores per row group: C (+ row_part)
        u32x4 rr[MI][2];
        auto load_res = [&](int mi) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int m = m0 + wm * (BM / 2) + mi * 16 + j * 8 + r_row0;
                if (m >= p.M) m = p.M - 1;
                const bf16_t* src = p.R + (long)m * p.ldr + n0 + wn * 64 + r_chunk * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rr[mi][j]) : "v"(src) : "memory");
            }
        };
        // operations issued between the load of rows (mi, j) and their use, in a FULL tile (every store is issued): the loads of
        // later row groups that are already out + the stores of the row groups before
        auto res_younger = [](int mi, int j) {
            const int i = 2 * mi + j;                                              // row group: loads are issued in this order
            const int loaded = 2 * (mi + RAHEAD < MI ? mi + RAHEAD + 1 : MI);      // row-group loads issued before this use
            const int older_blocks = mi > RAHEAD ? mi - RAHEAD : 0;                // blocks whose stores precede the load of (mi, j)
            return (loaded - 1 - i) + RSTORES * (i - 2 * older_blocks);
        };
        if constexpr (RES) {
#pragma unroll
            for (int mi = 0; mi < RAHEAD; ++mi) load_res(mi);
        }
        const float* lnf = lds_acc + (LNF && EPI == 1 ? parity * 1024 : 0);     // LNF consumer: [256 b'][256 s][256 rows x (rstd, -mean rstd)]
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if constexpr (RES) { if (mi + RAHEAD < MI) load_res(mi + RAHEAD); }
            f32x2 rs = {1.f, 0.f};
            if constexpr (LNF && EPI == 1) rs = *reinterpret_cast<const f32x2*>(lnf + 512 + (wm * (BM / 2) + mi * 16 + e_row) * 2);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                bf16x4 q;
                if constexpr (LIN) {
                    const int cb = (LNF && EPI == 1 ? 0 : n0) + wn * 64 + ni * 16 + (lane >> 4) * 4;
                    const f32x4 bias_r = *reinterpret_cast<const f32x4*>((LNF && EPI == 1 ? lnf : lds_acc) + cb);
                    f32x4 cs_r = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (LNF && EPI == 1) cs_r = *reinterpret_cast<const f32x4*>(lnf + 256 + cb);
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {                       // pairs: packed fp32 arithmetic (bit-identical per element)
                        f32x2 v = f32x2{acc[ni][mi][e], acc[ni][mi][e + 1]};
                        const f32x2 b2 = f32x2{bias_r[e], bias_r[e + 1]};
                        if constexpr (LNF && EPI == 1)
                            v = __builtin_elementwise_fma(v, f32x2{rs[0], rs[0]}, __builtin_elementwise_fma(f32x2{rs[1], rs[1]}, f32x2{cs_r[e], cs_r[e + 1]}, b2));
                        else v = v + b2;
                        if constexpr (ACT == CVCL_ACT_RELU) v = f32x2{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
                        else if constexpr (ACT == CVCL_ACT_GELU) v = gelu_bf16out2(v);
                        q[e] = (bf16_t)v[0];
                        q[e + 1] = (bf16_t)v[1];
                    }
                } else {
                    q = bf16x4{(bf16_t)acc[ni][mi][0], (bf16_t)acc[ni][mi][1], (bf16_t)acc[ni][mi][2], (bf16_t)acc[ni][mi][3]};
                }
                if constexpr (EPI != 0) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};     // ready for the next output tile
                const int chunk = ni * 2 + e_wchunk;
                *reinterpret_cast<bf16x4*>(stg + e_row * 128 + ((chunk ^ e_wsw) << 4) + e_wsub) = q;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = j * 8 + r_row0;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((r_chunk ^ ((row >> 1) & 7)) << 4));
                const int m = m0 + wm * (BM / 2) + mi * 16 + row, n = n0 + wn * 64 + r_chunk * 8;
                if constexpr (RES) {
                    if (!full) wait_vm<0>();                         // ragged last tile: lanes skip stores -- drain instead of counting
                    wait_vm_value(res_younger(mi, j), rr[mi][j]);
                }
                if (full || m < p.M) {
                    if constexpr (LIN) {
                        if constexpr (RES) {
                            const bf16x8 r = __builtin_bit_cast(bf16x8, rr[mi][j]);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)r[e]);
                        }
                        stream_store(v, reinterpret_cast<bf16x8*>(p.C + (long)m * p.ldc + n));
                        if constexpr (LNF && EPI == 2) {
                            // (sum, sum of squares) of the STORED values over this wave's 64-column strip of the row: v_dot2 on
                            // the packed pairs (products of bf16 are exact in fp32), then the row's eight lanes by DPP
                            const u32x4 w4 = __builtin_bit_cast(u32x4, v);
                            float s1 = 0.f, s2 = 0.f;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                s1 = dot2c_bf16(s1, w4[e], 0x3f803f80u);             // (1.0, 1.0)
                                s2 = dot2c_bf16(s2, w4[e], w4[e]);
                            }
                            s1 += dpp_quad_f32<0xB1>(s1); s2 += dpp_quad_f32<0xB1>(s2);          // lane ^ 1
                            s1 += dpp_quad_f32<0x4E>(s1); s2 += dpp_quad_f32<0x4E>(s2);          // lane ^ 2
                            s1 += dpp_quad_f32<0x141>(s1); s2 += dpp_quad_f32<0x141>(s2);        // row_half_mirror: the other quad
                            if (r_chunk == 0)
                                *reinterpret_cast<f32x2*>(p.row_part + ((long)m * (p.N >> 6) + ((n0 >> 6) + wn)) * 2) = f32x2{s1, s2};
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = (float)v[e];
                            st_sum[e] += f;
                            st_sq[e] = fmaf(f, f, st_sq[e]);
                        }
                        if (full || p.C) stream_store(v, reinterpret_cast<bf16x8*>(p.C + (long)m * p.ldc + n));
                    }
                }
            }
        }
        if (EPI == 0 && p.stats) {                           // this tile's column sums into the wave's slot, fixed order
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int o = 8; o <= 32; o <<= 1) {
                    st_sum[e] += __shfl_xor(st_sum[e], o, 64);
                    st_sq[e] += __shfl_xor(st_sq[e], o, 64);
                }
            }
            if (lane < 8) {
                float* s0 = lds_acc + (wave * 2 + 0) * 64 + lane * 8;
                float* s1 = lds_acc + (wave * 2 + 1) * 64 + lane * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) { s0[e] += st_sum[e]; s1[e] += st_sq[e]; }
            }
        }
        if constexpr (EPI == 0) {                            // the next output tile's accumulators start at -centre (one column
            const float* cen = cen_of();                     // group at a time: four live registers, not sixteen)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(cen + ni * 16);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = c4;
            }
        }
        return full ? ESTORES : 0;
    };
    auto epilogue = [&](int m0, int n0, int parity) __attribute__((always_inline)) -> int {
        // (the residual epilogue keeps ONE copy with the run-time test: without the branches the compiler holds more of the tile's
        // residual rows and spills -- and a scratch reload waits vmcnt(0) in the middle of the counted waits; C is never NULL there)
        if constexpr (RES) return epilogue_body(m0, n0, parity, std::integral_constant<int, -1>{}, m0 + BM <= p.M);
        else {
            if (m0 + BM <= p.M && p.C != nullptr) return epilogue_body(m0, n0, parity, std::integral_constant<int, 1>{}, true);
            return epilogue_body(m0, n0, parity, std::integral_constant<int, 0>{}, false);
        }
    };

    // ---- prologue: stages 0..3 in flight, stage 0 landed and in registers ----
    issue(0); advance(); issue(1); advance(); issue(2); advance(); issue(3); advance();
    if constexpr (EPI == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this thread's LDS writes above (-centre)
    wait_vm<12>();
    __builtin_amdgcn_s_barrier();
    read_frags(0, std::integral_constant<int, 0>{});
    if constexpr (EPI == 0) {
        const float* cen = cen_of();
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(cen + ni * 16);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = c4;
        }
    }

    // stores issued by an epilogue sit between stage loads in the in-order vmcnt queue for the next three waits
    // (the count passed to s_waitcnt must not exceed the number of younger instructions; a smaller one only waits longer)
    int after_epi = 0;          // iterations left in which the last epilogue's stores are younger than the awaited stage
    int epi_ops = 0;            // lower bound on what that epilogue issued: ESTORES or 0
    int c_ks = 0, c_i = ti, c_j = tj;                        // k stage / (m-tile, column tile) of the tile being multiplied
    int c_t = 0;                                             // index of the tile being multiplied (LNF: parity of its operand slot)
    auto step = [&](int g, auto P) __attribute__((always_inline)) {
        constexpr int q = decltype(P)::value;
        mma_half(P, std::integral_constant<int, 0>{});
        // stage g+1 has landed (this wave's part); the fragment reads of stage g are complete
        if (after_epi > 0 && epi_ops == ESTORES) wait_vm<8 + ESTORES>();
        else wait_vm<8>();
        if (after_epi > 0) --after_epi;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_frags((g + 1) & 3, std::integral_constant<int, 1 - q>{});
        issue(g & 3);                                       // stage g+4 into the buffer stage g occupied
        mma_half(P, std::integral_constant<int, 1>{});
        // one MFMA between any two of the 4 + MI reads / 4 loads
#pragma unroll
        for (int i = 0; i < 4 + MI; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        advance();
        if (++c_ks == KS) {
            c_ks = 0;
            if constexpr (LNF && EPI == 1) { epi_ops = epilogue(c_i * BM, c_j * BN, c_t & 1); ++c_t; }
            else epi_ops = epilogue(c_i * BM, c_j * BN, 0);
            after_epi = 3;
            if constexpr (FLAT) {
                if constexpr (!(LNF && EPI == 1)) ++c_t;
                if (c_t < nt) decode(L0 + c_t * cpx, c_i, c_j);
            } else c_i += step_i;
        }
    };
    for (int g = 0; g < S; g += 2) {
        step(g, std::integral_constant<int, 0>{});
        step(g + 1, std::integral_constant<int, 1>{});
    }
    wait_vm<0>();                                            // the re-loads past the end must not outlive the workgroup

    if (EPI == 0 && p.stats) {
        __syncthreads();
        if (tid < BN) {
            const int wn_ = tid >> 6, c = tid & 63, n = tj * BN + tid;   // column strip wn_: waves wn_ (upper rows) and wn_ + 4
            const float sv = lds_acc[((wn_) * 2 + 0) * 64 + c] + lds_acc[((wn_ + 4) * 2 + 0) * 64 + c];
            const float qv = lds_acc[((wn_) * 2 + 1) * 64 + c] + lds_acc[((wn_ + 4) * 2 + 1) * 64 + c];
            cvcl_bn_stats_out(p.stats, p.stats_acc, ti, p.N, n, sv, qv);
        }
    }
}

}  // namespace g8w
