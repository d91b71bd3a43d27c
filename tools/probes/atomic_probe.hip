// round 6 probe: can BatchNorm partial sums be ACCUMULATED (order-independent int64 fixed point) instead of written as partial rows?
//   mode 0: every workgroup writes its partial row (what the convolution kernels do today)
//   mode 1: agent-scope int64 atomic adds into ONE row
//   mode 2: workgroup-scope (L2-executed, no sc1) int64 atomic adds into the row of the workgroup's XCD (HW_REG_XCC_ID); a second
//           kernel sums the 8 rows
// prints the time of the producer kernel and whether the sums are exact.   hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }       // HW_REG_XCC_ID[3:0]

template <int MODE>
__global__ __launch_bounds__(512) void producer(float* rows, long long* acc, int C, int cols_per_wg, int spin) {
    // some work first so that the workgroups do not arrive in lock step
    float x = threadIdx.x * 0.001f;
    for (int i = 0; i < spin + (blockIdx.x & 15) * 8; ++i) x = x * 1.0001f + 0.5f;
    const int ncol_tiles = C / cols_per_wg, ct = blockIdx.x % ncol_tiles;
    for (int c = threadIdx.x; c < 2 * cols_per_wg; c += blockDim.x) {
        const int which = c / cols_per_wg, col = ct * cols_per_wg + c % cols_per_wg;
        const float v = (float)((blockIdx.x * 7 + col * 3 + which) % 1000) * 0.125f + (x > 1e30f ? 1.f : 0.f);
        if (MODE == 0) rows[((long)blockIdx.x * 2 + which) * C + col] = v;
        else {
            const long long f = (long long)__double2ll_rn((double)v * 1048576.0);
            if (MODE == 1) __hip_atomic_fetch_add(acc + (long)which * C + col, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(acc + ((long)xcc_id() * 2 + which) * C + col, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

__global__ void consumer(const long long* acc, int nrows, int C, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * C) return;
    long long s = 0;
    for (int r = 0; r < nrows; ++r) s += acc[(long)r * 2 * C + i];
    out[i] = (double)s / 1048576.0;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 224, C = argc > 2 ? atoi(argv[2]) : 512, cpw = argc > 3 ? atoi(argv[3]) : 256;
    const int spin = argc > 4 ? atoi(argv[4]) : 2000;
    float* rows; long long* acc; double* out;
    CK(hipMalloc(&rows, (size_t)grid * 2 * C * 4));
    CK(hipMalloc(&acc, (size_t)8 * 2 * C * 8));
    CK(hipMalloc(&out, (size_t)2 * C * 8));
    std::vector<double> want(2 * C, 0.0), got(2 * C);
    for (int b = 0; b < grid; ++b)
        for (int c = 0; c < 2 * cpw; ++c) {
            const int which = c / cpw, col = (b % (C / cpw)) * cpw + c % cpw;
            want[which * C + col] += (double)(float)((b * 7 + col * 3 + which) % 1000) * 0.125;
        }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipMemsetAsync(acc, 0, (size_t)8 * 2 * C * 8, 0));
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(producer<0>, dim3(grid), dim3(512), 0, 0, rows, acc, C, cpw, spin);
            else if (mode == 1) hipLaunchKernelGGL(producer<1>, dim3(grid), dim3(512), 0, 0, rows, acc, C, cpw, spin);
            else hipLaunchKernelGGL(producer<2>, dim3(grid), dim3(512), 0, 0, rows, acc, C, cpw, spin);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        int bad = -1;
        if (mode) {
            hipLaunchKernelGGL(consumer, dim3((2 * C + 255) / 256), dim3(256), 0, 0, acc, mode == 1 ? 1 : 8, C, out);
            CK(hipMemcpy(got.data(), out, (size_t)2 * C * 8, hipMemcpyDeviceToHost));
            bad = 0;
            for (int i = 0; i < 2 * C; ++i) bad += got[i] != want[i];
        }
        printf("grid %d C %d cols/wg %d mode %d: %.2f us  mismatches %d\n", grid, C, cpw, mode, best * 1e3f, bad);
    }
    return 0;
}
