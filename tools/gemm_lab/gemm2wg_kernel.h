// 256 (224) x 128 bf16 GEMM tile, FOUR waves per workgroup and TWO workgroups per CU (round 5 experiment).
//
//      C[M,N] = A[M,K] . W[N,K]^T          (A, W row-major, K contiguous; N % 128 == 0, K % 64 == 0)
//
// Why (next to gemm8w_kernel.h): the 8-wave kernel's two waves per SIMD belong to ONE workgroup and meet at every stage barrier, so
// they run the K loop, the epilogue and the prologue of a tile in lock step: an epilogue (VALU + stores) never overlaps the other
// wave's MFMAs, and every wait is a wait of the whole CU (SQ counters: MFMA pipe 31-37 % busy, MFMA / VALU co-execution 6 %).  Here
// the same wave tile -- (16 MI) x 64, 128 accumulators, 0.375 fragment reads per MFMA -- sits in a 4-wave workgroup (2 x 2 waves,
// one per SIMD) with its own 3-stage ring; two such workgroups share a CU and nothing synchronises them: one's epilogue, prologue
// and barrier waits are covered by the other's K loop.  Price: (256 + 128) operand rows per 256 x 128 outputs instead of
// (256 + 256) per 256 x 256 -- 1.5 x the L2 -> LDS bytes per FLOP.
//
// Pipeline per workgroup: "stage" = 32 k = 64 bytes per operand row, 24 KiB (A 16 KiB + W 8 KiB); 3-stage ring (72 KiB) + 4 x 2 KiB
// of epilogue staging = 80 KiB: exactly half a CU's LDS.  Iteration g multiplies stage g out of registers; in its middle it waits
// (counted vmcnt: 6 younger loads) for stage g+1, passes the one barrier, reads the fragments of g+1 and issues the loads of g+3
// into the buffer stage g occupied.
#pragma once
#include <type_traits>

#include "../../multimodal-baby_amd/csrc/cvcl_common.h"

namespace g2wg {

constexpr int BN = 128;
constexpr int BK = 32;
constexpr int NSTAGE = 3;
constexpr int W_BYTES = 8192;                     // 128 rows x 64 B
constexpr int STG_BYTES = 2048;                   // per-wave epilogue staging: 16 rows x 128 B
constexpr int LDS_BYTES = 80 * 1024;              // MI = 8: 3 x 24 KiB + 8 KiB; MI = 7: 3 x 22 KiB + 8 KiB + 2 KiB of bias slots

struct Dev {
    const bf16_t* A; const bf16_t* W; bf16_t* C;
    const float* bias;
    int M, N, K, lda, ldw, ldc, act;
    int tiles_m, ncol, sr;          // m-tiles, column tiles (N / 128), super-row height of the tile walk
    int skew;                       // VAR bit 1: start delay of the second dispatch wave, 10 ns ticks
};

__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swz(int g) { return (0x78 >> (2 * g)) & 3; }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// EPI 0: C = round(acc); EPI 1: C = round(act(acc + bias))
// VAR bit 0: s_setprio(1) around the MFMA halves
template <int MI, int EPI, int VAR>
__global__ __launch_bounds__(256, 2) void gemm2wg_kernel(Dev p) {
    constexpr int BM = MI * 32;
    constexpr int ESTORES = MI * 2;
    constexpr int A_BYTES = BM * 64, STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int BIAS_OFF = NSTAGE * STAGE_BYTES + 4 * STG_BYTES;           // EPI 1: [2][256] floats (parity slots; 128 used)
    static_assert(BIAS_OFF + (EPI == 1 ? 2048 : 0) <= LDS_BYTES, "half a CU's LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // supertile walk (as gemm8w_kernel.h): list of super-rows of sr m-tiles, column by column (serpentine); XCD x owns the x-th
    // eighth, its G / 8 workgroups take consecutive entries
    const int b = blockIdx.x, G = gridDim.x, cpx = G >> 3;
    auto decode = [&](int L, int& i_out, int& j_out) __attribute__((always_inline)) {
        const int per = p.sr * p.ncol;
        const int s = L / per, rem = L - s * per;
        const int ah = min(p.sr, p.tiles_m - s * p.sr);
        const int col = rem / ah;
        i_out = s * p.sr + (rem - col * ah);
        j_out = (s & 1) ? p.ncol - 1 - col : col;
    };
    const int xcd = b & 7;
    const int total = p.tiles_m * p.ncol;
    const int S0 = (int)(((long)xcd * total) >> 3), S1 = (int)(((long)(xcd + 1) * total) >> 3);
    const int L0 = S0 + (b >> 3);
    const int nt = L0 < S1 ? (S1 - L0 + cpx - 1) / cpx : 0;
    if (nt == 0) return;
    int ti, tj;
    decode(L0, ti, tj);
    const int KS = p.K / BK;
    const int S = nt * KS;

    const bf16_t* __restrict__ A = p.A;
    const bf16_t* __restrict__ W = p.W;

    // ---- staging: wave w lands A row blocks 4w .. 4w+3 and W row blocks 2w, 2w+1 (16 rows x 64 B each) per stage ----
    const int srow = lane >> 2;
    const int slog = (lane & 3) ^ swz((lane >> 4) & 3);
    unsigned a_off[4], w_off[2];
    const unsigned a_lim = (unsigned)(p.M - 1) * (unsigned)p.lda + 24;
    auto set_tile = [&](int i, int j) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = min(wave * 4 + q, BM / 16 - 1) * 16 + srow;      // MI = 7: wave 3 lands block 13 three times (same bytes)
            a_off[q] = min((unsigned)(i * BM + r) * (unsigned)p.lda + slog * 8, a_lim);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) w_off[q] = (unsigned)(j * BN + (wave * 2 + q) * 16 + srow) * (unsigned)p.ldw + slog * 8;
    };
    set_tile(ti, tj);
    int l_t = 0, l_ks = 0, l_j = tj;
    auto issue = [&](int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE_BYTES;
        const int k0 = l_ks * BK;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(A + a_off[q] + k0, base + min(wave * 4 + q, BM / 16 - 1) * 1024);
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(W + w_off[q] + k0, base + A_BYTES + (wave * 2 + q) * 1024);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if constexpr (EPI == 1) {
            // with the first stage of a tile, its 128 bias values into the slot of the tile's parity (wave 0; lanes 32-63 fetch the
            // next 128 floats or, at the right edge, the same ones again).  One load more in wave 0's queue only makes the next
            // counted wait conservative
            if (l_ks == 0 && wave == 0) {
                const float* src = p.bias + min(l_j * BN + lane * 4, p.N - 4);
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                 (void __attribute__((address_space(3)))*)(smem + BIAS_OFF + (l_t & 1) * 1024), 16, 0, 0);
            }
        }
        if (++l_ks == KS) {
            l_ks = 0;
            if (l_t + 1 < nt) {
                ++l_t;
                int i, j;
                decode(L0 + l_t * cpx, i, j);
                l_j = j;
                set_tile(i, j);
            }
        }
    };

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
        if constexpr (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 2 * h; ni < 2 * h + 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[q][ni], fa[q][mi], acc[ni][mi], 0, 0, 0);
        if constexpr (VAR & 1) __builtin_amdgcn_s_setprio(0);
    };
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    char* stg = smem + NSTAGE * STAGE_BYTES + wave * STG_BYTES;
    const int e_row = lane & 15;
    const int e_wchunk = lane >> 5, e_wsub = ((lane >> 4) & 1) * 8;
    const int e_wsw = (e_row >> 1) & 7;
    const int r_chunk = lane & 7, r_row0 = lane >> 3;

    auto epilogue = [&](int m0, int n0, int parity) __attribute__((always_inline)) -> int {
        const bool full = m0 + BM <= p.M;
        const float* lb = reinterpret_cast<const float*>(smem + BIAS_OFF + parity * 1024);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                bf16x4 q;
                if constexpr (EPI == 1) {
                    const f32x4 bias_r = *reinterpret_cast<const f32x4*>(lb + wn * 64 + ni * 16 + (lane >> 4) * 4);
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        f32x2 v = f32x2{acc[ni][mi][e], acc[ni][mi][e + 1]} + f32x2{bias_r[e], bias_r[e + 1]};
                        if (p.act == CVCL_ACT_RELU) v = f32x2{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
                        else if (p.act == CVCL_ACT_GELU) v = gelu_bf16out2(v);
                        q[e] = (bf16_t)v[0];
                        q[e + 1] = (bf16_t)v[1];
                    }
                } else {
                    q = bf16x4{(bf16_t)acc[ni][mi][0], (bf16_t)acc[ni][mi][1], (bf16_t)acc[ni][mi][2], (bf16_t)acc[ni][mi][3]};
                }
                acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int chunk = ni * 2 + e_wchunk;
                *reinterpret_cast<bf16x4*>(stg + e_row * 128 + ((chunk ^ e_wsw) << 4) + e_wsub) = q;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = j * 8 + r_row0;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((r_chunk ^ ((row >> 1) & 7)) << 4));
                const int m = m0 + wm * (BM / 2) + mi * 16 + row, n = n0 + wn * 64 + r_chunk * 8;
                if (full || m < p.M) stream_store(v, reinterpret_cast<bf16x8*>(p.C + (long)m * p.ldc + n));
            }
        }
        return full ? ESTORES : 0;
    };

    // VAR bit 1: the workgroups of the SECOND dispatch wave (blockIdx >= G / 2: the second slot of every CU) start half a tile period
    // late (p.skew, 10 ns ticks) -- equal tiles keep two co-resident workgroups in phase as surely as a barrier would
    if constexpr (VAR & 2) {
        if (b >= (G >> 1)) {
            const long t0 = wall_clock64();
            while ((long)wall_clock64() - t0 < (long)p.skew) __builtin_amdgcn_s_sleep(16);
        }
    }
    // ---- prologue: stages 0..2 in flight, stage 0 landed and in registers ----
    issue(0); advance(); issue(1); advance(); issue(2); advance();
    wait_vm<12>();
    __builtin_amdgcn_s_barrier();
    read_frags(0, std::integral_constant<int, 0>{});

    int after_epi = 0, epi_ops = 0;
    int c_ks = 0, c_i = ti, c_j = tj, c_t = 0;
    int rbuf = 1;                                            // ring slot of stage g+1
    auto step = [&](auto P) __attribute__((always_inline)) {
        constexpr int q = decltype(P)::value;
        mma_half(P, std::integral_constant<int, 0>{});
        // stage g+1 has landed (this wave's part): only stage g+2's six loads (+ an epilogue's stores) are younger
        if (after_epi > 0 && epi_ops == ESTORES) wait_vm<6 + ESTORES>();
        else wait_vm<6>();
        if (after_epi > 0) --after_epi;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_frags(rbuf, std::integral_constant<int, 1 - q>{});
        issue(rbuf == 0 ? 2 : rbuf - 1);                     // stage g+3 into the buffer stage g occupied
        mma_half(P, std::integral_constant<int, 1>{});
#pragma unroll
        for (int i = 0; i < 4 + MI; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
        rbuf = rbuf == 2 ? 0 : rbuf + 1;
        advance();
        if (++c_ks == KS) {
            c_ks = 0;
            epi_ops = epilogue(c_i * BM, c_j * BN, c_t & 1);
            after_epi = 2;
            ++c_t;
            if (c_t < nt) decode(L0 + c_t * cpx, c_i, c_j);
        }
    };
    for (int g = 0; g < S; g += 2) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
    }
    wait_vm<0>();
}

}  // namespace g2wg
