// Does a consumer that walks a tensor in the OPPOSITE order of its producer hit the Infinity Cache / L2 for the part written last?
// producer: streams `bytes` of bf16-like data front to back (block b writes chunk b); consumer: reads it front to back or back to
// front and reduces.  Times the consumer alone (HIP events).  hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CHUNK = 64 * 1024;   // bytes per block
__global__ __launch_bounds__(256) void wr(f32x4* p, float v, int nt) {
    f32x4* q = p + (size_t)blockIdx.x * (CHUNK / 16);
    for (int i = threadIdx.x; i < CHUNK / 16; i += 256) {
        if (nt) __builtin_nontemporal_store(f32x4{v, v, v, v}, q + i);
        else q[i] = f32x4{v, v, v, v};
    }
}
// producer that also reads: dst = 2 * src (what an elementwise pass or a GEMM with a wide operand does to the caches)
__global__ __launch_bounds__(256) void cp(const f32x4* s, f32x4* p, int nt) {
    const f32x4* a = s + (size_t)blockIdx.x * (CHUNK / 16);
    f32x4* q = p + (size_t)blockIdx.x * (CHUNK / 16);
    for (int i = threadIdx.x; i < CHUNK / 16; i += 256) {
        const f32x4 v = a[i] * 2.f;
        if (nt) __builtin_nontemporal_store(v, q + i);
        else q[i] = v;
    }
}
__global__ __launch_bounds__(256) void rd(const f32x4* p, float* out, int reverse, int nt) {
    const size_t b = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const f32x4* q = p + b * (CHUNK / 16);
    f32x4 a = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < CHUNK / 16; i += 256) { f32x4 t = nt ? __builtin_nontemporal_load(q + i) : q[i]; a += t; }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = 1.f;
}
int main() {
    const size_t sizes_mb[] = {51, 103, 205, 308, 411, 616, 822};
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20; const int nblk = (int)(bytes / CHUNK);
        f32x4* buf; hipMalloc(&buf, bytes);
        f32x4* src; hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
        for (int copy = 0; copy < 2; ++copy)
        for (int nt = 0; nt < 2; ++nt)
        for (int reverse = 0; reverse < 2; ++reverse) {
            double tot = 0; const int reps = 10;
            for (int r = 0; r < reps + 2; ++r) {
                if (copy) hipLaunchKernelGGL(cp, dim3(nblk), dim3(256), 0, 0, src, buf, nt);
                else hipLaunchKernelGGL(wr, dim3(nblk), dim3(256), 0, 0, buf, (float)r, nt);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(rd, dim3(nblk), dim3(256), 0, 0, buf, out, reverse, 0);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (r >= 2) tot += ms;
            }
            printf("%5zu MB  producer %s, %s stores, consumer %s: %8.1f us  %6.2f TB/s\n", mb, copy ? "copy " : "write", nt ? "nontemporal" : "plain      ", reverse ? "reverse" : "forward", tot / 10 * 1e3,
                   bytes / (tot / 10 * 1e-3) / 1e12);
        }
        hipFree(buf); hipFree(src);
    }
    return 0;
}
