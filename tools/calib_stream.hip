// calib_stream.hip -- what does the cfg3 access pattern cost with no arithmetic at all?
// One 64-lane workgroup per channel (like the fused kernels): reads the channel's 4096 complex
// samples (32 KB) in 1 KB wave-loads, four passes of eight loads with the next pass prefetched,
// reads `state_bytes` of per-channel state, writes 4 KB of audio and the state back.  Variants:
//   v0  input + audio only            v1  + 2.5 KB state in and out (the chain's S_in / S_out)
//   v2  v1 + 32 KB of shared table reads per channel (the LO of the call, L2-resident)
// Prints the achieved HBM GB/s of each, i.e. the practical roof for this traffic shape on this box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int V>
__global__ __launch_bounds__(64, 2) void k_stream(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                  float *__restrict__ state, const float4 *__restrict__ table)
{
    const int lane = threadIdx.x;
    const size_t c = blockIdx.x;
    const float4 *src = in + c * 2048;          // 4096 complex = 2048 float4
    float4 acc = make_float4(0, 0, 0, 0);
    float st[10];
    if (V >= 1)
        for (int j = 0; j < 10; ++j) st[j] = state[c * 640 + j * 64 + lane];
    float4 raw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = src[i * 64 + lane];
    for (int pass = 0; pass < 4; ++pass) {
        float4 lo[8];
        if (V >= 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) lo[i] = table[pass * 512 + i * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc.x += raw[i].x; acc.y += raw[i].y; acc.z += raw[i].z; acc.w += raw[i].w;
            if (V >= 2) { acc.x += lo[i].x; acc.y += lo[i].w; }
        }
        if (pass < 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) raw[i] = src[(pass + 1) * 512 + i * 64 + lane];
        }
        out[c * 256 + pass * 64 + lane] = acc;  // 1 KB of audio per pass
    }
    if (V >= 1)
        for (int j = 0; j < 10; ++j) state[c * 640 + j * 64 + lane] = st[j] + acc.x;
}

int main(int argc, char **argv)
{
    const size_t C = 65536;
    const size_t lds = argc > 1 ? (size_t)atoi(argv[1]) : 0;   // dynamic LDS per workgroup: caps the workgroups per CU
    float4 *in, *out, *table; float *state;
    if (hipMalloc(&in, C * 32768) != hipSuccess || hipMalloc(&out, C * 4096) != hipSuccess ||
        hipMalloc(&state, C * 2560) != hipSuccess || hipMalloc(&table, 32768) != hipSuccess) return 1;
    (void)hipMemset(in, 0, C * 32768); (void)hipMemset(state, 0, C * 2560); (void)hipMemset(table, 0, 32768);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)k_stream<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)k_stream<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)k_stream<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    printf("dynamic LDS %zu B per workgroup\n", lds);
    for (int v = 0; v < 3; ++v) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipEventRecord(e0);
            if (v == 0) k_stream<0><<<C, 64, lds>>>(in, out, state, table);
            if (v == 1) k_stream<1><<<C, 64, lds>>>(in, out, state, table);
            if (v == 2) k_stream<2><<<C, 64, lds>>>(in, out, state, table);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double bytes = (double)C * (32768 + 4096 + (v >= 1 ? 2 * 2560 : 0));
        printf("v%d  %.4f ms  %.0f GB/s of HBM traffic (%.3f GB)%s\n", v, best, bytes / best / 1e6, bytes / 1e9,
               v == 2 ? "  + 2.1 GB of L2-resident table reads" : "");
    }
    return 0;
}
