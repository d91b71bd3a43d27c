// Shared device helpers for libcvcl_hip (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cvcl_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define CVCL_WAVE 64

// thread-local last error text (cvcl_last_error)
void cvcl_set_error(const char* fmt, ...);

// Run-time switches (api.cpp holds the only getenv calls of the library).  The PRODUCT reads four environment variables -- the
// table "Run-time switches" in include/cvcl_hip.h: cvcl_env_on(name) is false when $name starts with '0'.  Everything else that
// earlier rounds could toggle is a LAB switch: cvcl_lab_int(name, default) reads $name only in a library built with -DCVCL_LAB
// (tools/README.md) and is the constant `default` in the product build.
int cvcl_gemm_cu_share();                                   // the value set by cvcl_set_gemm_cu_share (0 = all CUs)
bool cvcl_env_on(const char* name);
int cvcl_lab_int(const char* name, int dflt);

// optional HIP-event timing of a launch (see cvcl_prof_enable in include/cvcl_hip.h)
bool cvcl_prof_on();
void* cvcl_prof_begin(void* stream, int cls);
void cvcl_prof_end(void* handle, void* stream);
struct CvclProfScope {
    void* h; void* s;
    CvclProfScope(void* stream, int cls) : h(cvcl_prof_begin(stream, cls)), s(stream) {}
    ~CvclProfScope() { cvcl_prof_end(h, s); }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is PER DEVICE: a per-process "already done" flag makes the first launch on a
// second GPU of the same process fail.  One bit per device ordinal, set after the attribute call succeeded
// (two threads racing both make the call: harmless).
#ifdef __cplusplus
#include <atomic>
struct CvclLdsAttr {
    std::atomic<unsigned long long> devs{0};
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
    bool ready() const { return (devs.load(std::memory_order_acquire) >> dev()) & 1ull; }
    void mark() { devs.fetch_or(1ull << dev(), std::memory_order_release); }
};
#endif

#define CVCL_CHECK_ARG(cond, ...)                \
    do {                                         \
        if (!(cond)) {                           \
            cvcl_set_error(__VA_ARGS__);         \
            return CVCL_EINVAL;                  \
        }                                        \
    } while (0)

#define CVCL_LAUNCH_CHECK()                                                    \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            cvcl_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,      \
                           hipGetErrorString(e_));                             \
            return CVCL_ELAUNCH;                                               \
        }                                                                      \
    } while (0)

__host__ __device__ static inline int cvcl_div_up(long a, long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// ---- element type traits -------------------------------------------------------------------
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int kPerChunk = 4;   // elements per 16-byte chunk
    __device__ static inline float to_f(float v) { return v; }
    __device__ static inline float from_f(float v) { return v; }
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int kPerChunk = 8;
    __device__ static inline float to_f(bf16_t v) { return (float)v; }
    __device__ static inline bf16_t from_f(float v) { return (bf16_t)v; }   // v_cvt_pk_bf16_f32, RNE
};

// 16-byte chunk <-> floats
template <typename T> struct Chunk;
template <> struct Chunk<float> {
    f32x4 v;
    __device__ inline void load(const float* p) { v = *reinterpret_cast<const f32x4*>(p); }
    __device__ inline void store(float* p) const { *reinterpret_cast<f32x4*>(p) = v; }
    __device__ inline void zero() { v = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ inline float get(int i) const { return v[i]; }
    __device__ inline void set(int i, float f) { v[i] = f; }
};
template <> struct Chunk<bf16_t> {
    bf16x8 v;
    __device__ inline void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
    __device__ inline void store(bf16_t* p) const { *reinterpret_cast<bf16x8*>(p) = v; }
    __device__ inline void zero() {
        u32x4 z = {0u, 0u, 0u, 0u};
        v = __builtin_bit_cast(bf16x8, z);
    }
    __device__ inline float get(int i) const { return (float)v[i]; }
    __device__ inline void set(int i, float f) { v[i] = (bf16_t)f; }
};

// two bf16 in one register <-> two floats (v_lshlrev / v_and; v_cvt_pk_bf16_f32): bandwidth-bound kernels write their VALU work on
// register pairs so that it compiles to the packed fp32 instructions (v_pk_fma_f32, v_pk_add_f32) -- half the issue slots
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 widen2(unsigned u) {
    return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
// acc + a.lo b.lo + a.hi b.hi on packed bf16 pairs (v_dot2c_f32_bf16; products of bf16 are exact in fp32).  Inline asm on the raw
// dwords: with __builtin_amdgcn_fdot2_f32_bf16 on __builtin_bit_cast(bf16x2, w[e]) of a bit-cast bf16x8 this hipcc (ROCm 7.2) feeds
// dword 0 to all four calls (seen in the ISA; the same happened with the cvt_scalef32 builtin below).
__device__ __forceinline__ float dot2c_bf16(float acc, unsigned a, unsigned b) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}
template <int CTRL> __device__ __forceinline__ float dpp_quad_f32(float v) {     // v of the lane the DPP control selects (0 if none)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ unsigned round2(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
// relu on a rounded bf16 pair: rounding keeps the sign, and a bf16 is negative (or -0) exactly when its bits are a negative
// int16 -> one packed integer max
__device__ __forceinline__ unsigned relu2(unsigned pair) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pair), s16x2{0, 0}));
}

// streamed-out results: nontemporal by default (the L2 keeps the operands); -DCVCL_PLAIN_STORES builds the same kernels with
// ordinary stores (experiment: does the consumer find the tensor in the Infinity Cache?)
template <typename V> __device__ __forceinline__ void stream_store(V v, V* dst) {
#ifdef CVCL_PLAIN_STORES
    *dst = v;
#else
    __builtin_nontemporal_store(v, dst);
#endif
}

// ---- wave / block reductions ---------------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum through LDS scratch of >= (blockDim/64) floats; result broadcast to all threads
__device__ inline float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
__device__ inline float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// ---- BatchNorm statistics ACCUMULATORS (round 6; cvcl_hip.h "CVCL_STATS_ACCUMULATE") ---------------------------------------------
// The convolution kernels hand their per-channel (sum, sum of squares) to BatchNorm either as one partial ROW per workgroup
// (stats[rows][2][N] floats, reduced by cvcl_bn_finalize) or, in accumulate mode, by ADDING them to an int64 accumulator
// acc[8][2][N]: fixed point with 24 fractional bits, one row per XCD (the adds of an XCD's workgroups meet in that XCD's L2; measured
// free next to the row stores, and exact: tools/probes/atomic_probe.hip).  Integer addition commutes, so the totals are bit-
// deterministic whatever the arrival order; the rounding of a partial sum to 2^-24 moves a mean by < 1e-8 / a variance by < 1e-8
// absolute at the trunk's row counts (eps is 1e-5).  A consumer sums the 8 rows of its channels and forms (scale, shift) itself.
constexpr int kBnAccRows = 8;
constexpr double kBnAccScale = 16777216.0;                  // 2^24
__device__ __forceinline__ int cvcl_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }     // HW_REG_XCC_ID[3:0]
// A partial that is not finite, or beyond 2^28 in magnitude (a workgroup's sum over ~3000 stored bf16 values: activations of ~300 rms),
// cannot be represented: it POISONS the channel instead -- the sum-of-squares accumulator is raised to 2^62 (atomic max; the legitimate
// adds of <= 1024 workgroups stay below that), which the consumer turns into a NaN affine, exactly what cvcl_bn_finalize produces from
// a non-finite partial row: a diverged run fails as loudly in either form.
constexpr long long kBnAccPoison = 1LL << 62;
constexpr float kBnAccMaxPartial = 268435456.f;             // 2^28
__device__ __forceinline__ void cvcl_bn_acc_add(float* stats_as_acc, int N, int n, float s, float q) {
    long long* row = reinterpret_cast<long long*>(stats_as_acc) + (long)cvcl_xcc_id() * 2 * N;
    if (fabsf(s) < kBnAccMaxPartial && fabsf(q) < kBnAccMaxPartial) {          // (false for NaN / inf)
        __hip_atomic_fetch_add(row + n, __double2ll_rn((double)s * kBnAccScale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(row + N + n, __double2ll_rn((double)q * kBnAccScale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_fetch_max(row + N + n, kBnAccPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// one partial (s, q) of channel n from workgroup row `row`: a row store or an accumulator add
__device__ __forceinline__ void cvcl_bn_stats_out(float* stats, int acc_mode, long row, int N, int n, float s, float q) {
    if (acc_mode) cvcl_bn_acc_add(stats, N, n, s, q);
    else {
        stats[(row * 2 + 0) * N + n] = s;
        stats[(row * 2 + 1) * N + n] = q;
    }
}

// the running-statistics update r <- (1 - m) r + m x: one spelling, shared by the in-place, the deferred and the on-load forms
__device__ inline float bn_ema(float r, float x, float m) { return fmaf(m, x, (1.f - m) * r); }

// finalize-on-load source: the accumulators of the layer whose output a kernel is about to normalise (csrc/resnext.hip)
struct BnSrc {
    const long long* acc;           // [8][2][C] accumulators of the STORED tensor (NULL: no finalize-on-load, read scale / shift)
    int C;
    double count;
    const float* gamma; const float* beta; const float* centre;
    float* running_mean; float* running_var; int64_t* nbt;
    float momentum, eps;
    float* moments; int moments_ld;
    float* scale_out; float* shift_out;
};

// channels [c0, c0 + NCH) by the first NCH threads of the workgroup; on return (behind a barrier) sc_out / sh_out [NCH] in LDS hold
// the slice's affine.  The expressions after the sums are bn_finalize_kernel's.
template <int NCH>
__device__ __forceinline__ void bn_slice_affine(const BnSrc& b, int c0, bool publish, float* sc_out, float* sh_out) {
    const int c = threadIdx.x;
    if (c < NCH) {
        const int ch = c0 + c;
        long long v[2 * kBnAccRows];
#pragma unroll
        for (int r = 0; r < kBnAccRows; ++r) {
            v[2 * r] = b.acc[((long)r * 2 + 0) * b.C + ch];
            v[2 * r + 1] = b.acc[((long)r * 2 + 1) * b.C + ch];
        }
        long long S = 0, Q = 0;
        bool poisoned = false;
#pragma unroll
        for (int r = 0; r < kBnAccRows; ++r) { S += v[2 * r]; Q += v[2 * r + 1]; poisoned |= v[2 * r + 1] >= kBnAccPoison || v[2 * r + 1] < 0; }
        const double s = (double)S * (1.0 / kBnAccScale), q = (double)Q * (1.0 / kBnAccScale);
        const double mean = s / b.count;
        double var = q / b.count - mean * mean;
        if (var < 0.0) var = 0.0;
        if (poisoned) var = __builtin_nan("");                    // a non-finite / unrepresentable partial sum: NaN affine (cvcl_bn_acc_add)
        const float sc = b.gamma[ch] / sqrtf((float)var + b.eps);
        const float sh = b.beta[ch] - (float)mean * sc;
        sc_out[c] = sc;
        sh_out[c] = sh;
        if (publish) {
            if (b.scale_out) { b.scale_out[ch] = sc; b.shift_out[ch] = sh; }
            const double unbiased = b.count > 1.0 ? var * b.count / (b.count - 1.0) : var;
            const float true_mean = b.centre ? (float)(mean + (double)b.centre[ch]) : (float)mean;
            if (b.moments) {
                b.moments[ch] = true_mean;
                b.moments[b.moments_ld + ch] = (float)unbiased;
            } else if (b.running_mean) {
                b.running_mean[ch] = bn_ema(b.running_mean[ch], true_mean, b.momentum);
                b.running_var[ch] = bn_ema(b.running_var[ch], (float)unbiased, b.momentum);
            }
        }
    }
    if (publish && !b.moments && b.nbt && c0 == 0 && threadIdx.x == 0) *b.nbt += 1;
    __syncthreads();
}

// cvcl_conv1x1_gram with the operand's BatchNorm affine formed inside the kernel from `src` (internal, C++ linkage: bn_gram.hip)
int cvcl_conv1x1_gram_src(const void* A, int lda, long M, int K, const float* a_scale, const float* a_shift, const BnSrc* src, int a_relu,
                          void* workspace, size_t workspace_bytes, const double** gram_out, void* stream);

// sum_{i < n} load(i), UN loads in flight per wait, additions in index order (same result as the plain loop, deterministic).
// A `for (...) acc += p[i]` loop waits for every load before issuing the next (one L2 / HBM round trip per element): these
// reductions are pure latency, so batching the loads is the whole optimisation.  load(i) must be valid for every i < n.
template <int UN, typename ACC, typename F>
__device__ inline ACC ordered_sum(int n, F load) {
    ACC acc = (ACC)0;
    for (int i0 = 0; i0 < n; i0 += UN) {
        ACC v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = (ACC)load(min(i0 + u, n - 1));
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (i0 + u < n) acc += v[u];
    }
    return acc;
}

// ---- GELU (erf form) with Abramowitz-Stegun 7.1.26 for erf (|error| < 1.5e-7, far below bf16 resolution) on v_rcp_f32 / v_exp_f32;
// shared by the GEMM epilogues and the standalone passes so that fused and unfused paths round identically
__device__ inline float gelu_erf_fast(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-x * x * 1.4426950408889634f);
    const float erf_abs = fmaf(-poly, e, 1.f);
    return 0.5f * v * (1.f + copysignf(erf_abs, v));
}
// ---- GELU for bf16 / e4m3 OUTPUTS (round 4): gelu(v) = v Phi(v) = v sigmoid(z(v)) with z = logit(Phi(v)), an odd function fitted
// by a quintic v (a0 + a1 v^2 + a2 v^4) (minimax on [-6, 6], v^2 clamped at 36: beyond it the sigmoid is saturated):
// |error| <= 2.6e-5 absolute against the erf form in float64 (tools/fit_gelu.py) -- below half a bf16 ulp for |gelu| >= 0.013 and
// at most ~1 ulp of the values it is added to downstream -- in 9 VALU instructions (v_exp_f32 + v_rcp_f32 among them) instead of
// the 15 of the Abramowitz-Stegun form above: the GELU epilogues of the ViT MLP are VALU-issue-bound (profiles/r04_pmc_c5_summary.txt:
// fc1 in e4m3 spent 41 VALU instructions per MFMA).  Every bf16 path (fused epilogues and the standalone pass) uses THIS function, so
// fused and unfused forms still round identically; the exact-fp32 parity mode keeps erff.
__device__ inline float gelu_bf16out(float v) {
    const float u = fminf(v * v, 36.f);
    float p = fmaf(u, -0.0007030378797f * -1.4426950408889634f, 0.07401131995f * -1.4426950408889634f);
    p = fmaf(p, u, 1.595015736f * -1.4426950408889634f);               // -log2(e) (a0 + a1 u + a2 u^2)
    const float e = __builtin_amdgcn_exp2f(v * p);                     // exp(-z)
    return v * __builtin_amdgcn_rcpf(1.f + e);
}
// the same function on a pair (bit-identical per element: the same IEEE operations): v_pk_mul / v_pk_fma / v_pk_add carry two
// elements per instruction -- 12 instructions per pair instead of 18 (the two transcendentals and the clamp stay per element)
__device__ inline f32x2 gelu_bf16out2(f32x2 v) {
    f32x2 u = v * v;
    u = f32x2{fminf(u[0], 36.f), fminf(u[1], 36.f)};
    constexpr float L = -1.4426950408889634f;
    f32x2 p = __builtin_elementwise_fma(u, f32x2{-0.0007030378797f * L, -0.0007030378797f * L}, f32x2{0.07401131995f * L, 0.07401131995f * L});
    p = __builtin_elementwise_fma(p, u, f32x2{1.595015736f * L, 1.595015736f * L});
    const f32x2 t = v * p;
    const f32x2 d = f32x2{1.f, 1.f} + f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    return v * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
// d gelu(v) / dv in the erf form (backward of the fused GELU epilogue / cvcl_gelu_bf16).  NOTE: the bf16 FORWARD epilogues use the
// fitted gelu_bf16out above (|gelu_bf16out - gelu_erf| <= 2.6e-5), the backward differentiates the erf form: the two are the same
// function to ~1e-4 of the derivative -- far below the bf16 rounding of the gradients it multiplies, and it keeps the fine-tuning
// path's gradients those of the function the fp32 parity mode evaluates.
__device__ inline float gelu_grad_fast(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-x * x * 1.4426950408889634f);      // exp(-v^2 / 2)
    const float cdf = 0.5f * (1.f + copysignf(fmaf(-poly, e, 1.f), v));
    return fmaf(v, 0.3989422804014327f * e, cdf);
}

// ---- fp8 e4m3 / MX (e8m0 block scale) helpers shared by the fp8 GEMM epilogue and the attention epilogue
// e8m0 block scale (power of two) that maps a block's amax into the e4m3 range: 2^ceil(log2(amax / 448)), byte = exponent + 127
__device__ inline unsigned mx_scale_byte(float amax) {
    const unsigned bits = __float_as_uint(amax * (1.f / 448.f));
    unsigned e = (bits >> 23) & 0xffu;
    if (bits & 0x7fffffu) e += 1;
    return e < 1u ? 1u : (e > 254u ? 254u : e);
}
__device__ inline float mx_inv_scale(unsigned byte) { return __uint_as_float((254u - byte) << 23); }   // 2^(127 - byte)
// MX block quantisation of eight bf16 values held as four packed dwords (round 4; replaces unpack + fmax + multiply + cvt per
// element): |x| as 15-bit integers (bf16 magnitudes order like their bit patterns), v_pk_max_u16 tree -> this lane's amax in the
// low half; the caller combines lanes, then v_cvt_scalef32_pk_fp8_bf16 divides by the power-of-two block scale and rounds to e4m3.
__device__ inline unsigned bf16x8_absmax_bits(u32x4 w) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 a = __builtin_bit_cast(u16x2, w[0] & 0x7fff7fffu), b = __builtin_bit_cast(u16x2, w[1] & 0x7fff7fffu);
    const u16x2 c = __builtin_bit_cast(u16x2, w[2] & 0x7fff7fffu), d = __builtin_bit_cast(u16x2, w[3] & 0x7fff7fffu);
    const u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
    const unsigned mm = __builtin_bit_cast(unsigned, m);
    return max(mm & 0xffffu, mm >> 16);
}
__device__ inline u32x2 bf16x8_to_fp8_scaled(u32x4 w, float scale) {       // e4m3(x / scale), scale = 2^k
    // (inline asm: through __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16 this hipcc converts the FIRST dword four times -- seen in the
    // ISA, ROCm 7.2; the trailing s_nop covers the op_sel destination write before the next VALU reads the register)
    unsigned lo = 0, hi = 0;
    asm("v_cvt_scalef32_pk_fp8_bf16 %0, %2, %6\n\t"
        "v_cvt_scalef32_pk_fp8_bf16 %1, %4, %6\n\t"
        "v_cvt_scalef32_pk_fp8_bf16 %0, %3, %6 op_sel:[0,0,1]\n\t"
        "v_cvt_scalef32_pk_fp8_bf16 %1, %5, %6 op_sel:[0,0,1]\n\t"
        "s_nop 0"
        : "+v"(lo), "+v"(hi) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(scale));
    return u32x2{lo, hi};
}
__device__ inline unsigned pack4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
#endif  // __HIPCC__
