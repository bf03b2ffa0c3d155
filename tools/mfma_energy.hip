// mfma_energy.hip <variant> -- what does a matrix instruction cost at the package power cap?  2048 single-wave workgroups
// (two per SIMD, like k_ssb_split16) issue back-to-back MFMAs on random register operands, four independent accumulators.
// variant 0: v_mfma_f32_16x16x32_f16   1: v_mfma_i32_16x16x64_i8   2: v_mfma_f32_16x16x32_bf16   3: v_mfma_f32_32x32x16_f16
// Runs ~3 s; prints time per launch and instructions per second.  Sample rocm-smi beside it (tools/mfma_energy.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 4096;
template <int V>
__global__ __launch_bounds__(64, 2) void k(const unsigned *__restrict__ seed, float *__restrict__ out)
{
    const int l = threadIdx.x;
    unsigned s = seed[l] ^ (blockIdx.x * 2654435761u);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    if constexpr (V == 0) {
        h8 a[4], b[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) { a[j][e] = (_Float16)(((rnd() >> 8) & 1 ? 1.0f : -1.0f) * (1.0f + (float)(rnd() >> 22) / 1024.0f)); b[j][e] = (_Float16)(((rnd() >> 8) & 1 ? 1.0f : -1.0f) * (1.0f + (float)(rnd() >> 22) / 1024.0f)); }
        f4 c[4] = {};
        for (int it = 0; it < ITER; it += 4)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[(j + u) & 3], c[j], 0, 0, 0);
        out[blockIdx.x * 64 + l] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else if constexpr (V == 1) {
        i4 a[4], b[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) { a[j][e] = (int)rnd(); b[j][e] = (int)rnd(); }
        i4 c[4] = {};
        for (int it = 0; it < ITER; it += 4)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j], b[(j + u) & 3], c[j], 0, 0, 0);
        out[blockIdx.x * 64 + l] = (float)(c[0][0] + c[1][1] + c[2][2] + c[3][3]);
    } else if constexpr (V == 3) {
        typedef float f16v __attribute__((ext_vector_type(16)));
        h8 a[4], b[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) { a[j][e] = (_Float16)(((rnd() >> 8) & 1 ? 1.0f : -1.0f) * (1.0f + (float)(rnd() >> 22) / 1024.0f)); b[j][e] = (_Float16)(((rnd() >> 8) & 1 ? 1.0f : -1.0f) * (1.0f + (float)(rnd() >> 22) / 1024.0f)); }
        f16v c[4] = {};
        for (int it = 0; it < ITER; it += 4)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(j + u) & 3], c[j], 0, 0, 0);
        out[blockIdx.x * 64 + l] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else {
        b8 a[4], b[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) { a[j][e] = (__bf16)(((rnd() >> 8) & 1 ? 1.0f : -1.0f) * (1.0f + (float)(rnd() >> 25) / 128.0f)); b[j][e] = (__bf16)(((rnd() >> 8) & 1 ? 1.0f : -1.0f) * (1.0f + (float)(rnd() >> 25) / 128.0f)); }
        f4 c[4] = {};
        for (int it = 0; it < ITER; it += 4)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[(j + u) & 3], c[j], 0, 0, 0);
        out[blockIdx.x * 64 + l] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    }
}
int main(int argc, char **argv)
{
    const int v = argc > 1 ? atoi(argv[1]) : 0;
    unsigned hs[64]; for (int i = 0; i < 64; ++i) hs[i] = 12345u * (i + 1);
    unsigned *ds; float *dout;
    hipMalloc(&ds, sizeof hs); hipMalloc(&dout, 2048 * 64 * 4);
    hipMemcpy(ds, hs, sizeof hs, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { if (v == 0) k<0><<<2048, 64, 40960>>>(ds, dout); else if (v == 1) k<1><<<2048, 64, 40960>>>(ds, dout); else if (v == 3) k<3><<<2048, 64, 40960>>>(ds, dout); else k<2><<<2048, 64, 40960>>>(ds, dout); };
    for (int w = 0; w < 200; ++w) launch();
    hipDeviceSynchronize();
    int n = 0; float total = 0;
    while (total < 3000.0f) {
        hipEventRecord(e0);
        for (int w = 0; w < 200; ++w) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); total += ms; n += 200;
    }
    const double inst = 2048.0 * ITER * 4;
    printf("variant %d: %.4f ms per launch, %.3e MFMA/s, %.2f cycles per MFMA per SIMD at 2.4 GHz\n", v, total / n, inst / (total / n * 1e-3),
           (total / n * 1e-3) * 2.4e9 / (inst / 1024.0));
    return 0;
}
