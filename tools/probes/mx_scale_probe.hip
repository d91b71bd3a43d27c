// probe of v_mfma_scale_f32_32x32x64_f8f6f4 scale-operand semantics (which lanes' scale bytes apply to which row / k block)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int OPA, int OPB>
__global__ void k(const int* sa, const int* sb, float* c) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }      // e4m3 1.0 everywhere
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, OPA, sa[l], OPB, sb[l]);
    for (int i = 0; i < 16; ++i) c[l * 16 + i] = acc[i];
}
int main() {
    int *sa, *sb; float* c;
    hipMalloc(&sa, 256); hipMalloc(&sb, 256); hipMalloc(&c, 64 * 16 * 4);
    int ha[64], hb[64]; float hc[1024];
    auto run = [&](const char* name, int opa, int opb) {
        hipMemcpy(sa, ha, 256, hipMemcpyHostToDevice); hipMemcpy(sb, hb, 256, hipMemcpyHostToDevice);
        if (opa == 0 && opb == 0) hipLaunchKernelGGL((k<0, 0>), 1, 64, 0, 0, sa, sb, c);
        if (opa == 1 && opb == 0) hipLaunchKernelGGL((k<1, 0>), 1, 64, 0, 0, sa, sb, c);
        if (opa == 2 && opb == 0) hipLaunchKernelGGL((k<2, 0>), 1, 64, 0, 0, sa, sb, c);
        if (opa == 3 && opb == 0) hipLaunchKernelGGL((k<3, 0>), 1, 64, 0, 0, sa, sb, c);
        hipMemcpy(hc, c, 4096, hipMemcpyDeviceToHost);
        printf("%s: lane0 regs:", name);
        for (int i = 0; i < 16; ++i) printf(" %g", hc[i]);
        printf(" | lane5 r0 %g lane33 r0 %g lane40 r3 %g\n", hc[5 * 16], hc[33 * 16], hc[40 * 16 + 3]);
    };
    for (int l = 0; l < 64; ++l) { ha[l] = 0x7f7f7f7f; hb[l] = 0x7f7f7f7f; }
    run("unit scales (expect 64)", 0, 0);
    // A scale byte0 = 2^1 for lanes 0..31 only (k block 0?), unit elsewhere
    for (int l = 0; l < 64; ++l) ha[l] = l < 32 ? 0x7f7f7f80 : 0x7f7f7f7f;
    run("A byte0=2 lanes<32, opsel 0 (expect 96 if lanes<32 scale k-block 0 of their row)", 0, 0);
    for (int l = 0; l < 64; ++l) ha[l] = l >= 32 ? 0x7f7f7f80 : 0x7f7f7f7f;
    run("A byte0=2 lanes>=32, opsel 0", 0, 0);
    // only lane 3 has scale 2 in byte 0: which output rows change? (D row index = A row i)
    for (int l = 0; l < 64; ++l) ha[l] = l == 3 ? 0x7f7f7f80 : 0x7f7f7f7f;
    run("A byte0=2 lane 3 only", 0, 0);
    printf("   rows touched: ");
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) if (hc[l * 16 + i] != 64.f) { printf("[lane %d reg %d = %g] ", l, i, hc[l * 16 + i]); if (l > 1) goto done; }
done:
    printf("\n");
    // byte select via opsel: put 2 in byte 1
    for (int l = 0; l < 64; ++l) ha[l] = 0x7f7f807f;
    run("A byte1=2 all lanes, opsel 0", 0, 0);
    run("A byte1=2 all lanes, opsel 1", 1, 0);
    for (int l = 0; l < 64; ++l) ha[l] = 0x7f807f7f;
    run("A byte2=2 all lanes, opsel 2", 2, 0);
    for (int l = 0; l < 64; ++l) ha[l] = 0x807f7f7f;
    run("A byte3=2 all lanes, opsel 3", 3, 0);
    return 0;
}
