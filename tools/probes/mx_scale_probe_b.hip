// probe: which lane's scale_b byte covers which (lane, byte) position of the B operand of v_mfma_scale_f32_32x32x64_f8f6f4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(int L, int J, int X, int useA, float* c) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0; }
    if (l == L) b[J >> 2] = 0x38 << (8 * (J & 3));
    int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;
    if (l == X) { if (useA) sa = 0x7f7f7f80; else sb = 0x7f7f7f80; }
    f32x16 acc = {};
    if (useA) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, acc, 0, 0, 0, sa, 0, sb);   // sparse operand first
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s = fmaxf(s, acc[i]);
    c[l] = s;
}
int main() {
    float* c; hipMalloc(&c, 256); float hc[64];
    for (int useA = 0; useA < 2; ++useA) {
        printf("sparse operand is %s\n", useA ? "A (first)" : "B (second)");
        for (int L : {5, 37}) for (int X : {5, 37}) {
            printf(" data lane %2d scale lane %2d: ", L, X);
            for (int J = 0; J < 32; ++J) {
                hipLaunchKernelGGL(k, 1, 64, 0, 0, L, J, X, useA, c);
                hipMemcpy(hc, c, 256, hipMemcpyDeviceToHost);
                float m = 0; for (int i = 0; i < 64; ++i) m = fmaxf(m, hc[i]);
                printf("%g", m);
            }
            printf("\n");
        }
    }
    return 0;
}
