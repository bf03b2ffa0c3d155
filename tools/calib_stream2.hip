// calib_stream2.hip -- the streaming roof of the cfg3 traffic shape for the PERSISTENT launch shape of k_ssb_split16:
// G single-wave workgroups, workgroup b streams channels b, b + G, ...: 32 KB of I/Q per channel in 1 KB buffer loads
// (nt, like the kernel), the next pass always prefetched (across the channel boundary too), 2.5 KB of state in and
// out, 4 KB of audio out.  No arithmetic beyond one add per loaded register.  Steady state: 300 launches, HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
template <int AUX>
__global__ __launch_bounds__(64, 2) void k_stream(const float *__restrict__ in, float *__restrict__ out, float *__restrict__ state,
                                                  unsigned channels)
{
    const int lane = threadIdx.x;
    u4v raw[8];
    unsigned c = blockIdx.x;
    __amdgpu_buffer_rsrc_t rs = rsrc(in + (size_t)c * 8192, 32768);
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + i * 1024, 0, AUX);
    for (; c < channels; c += gridDim.x) {
        float st[10];
        for (int j = 0; j < 10; ++j) st[j] = state[(size_t)c * 640 + j * 64 + lane];
        const unsigned cn = c + gridDim.x;
        const __amdgpu_buffer_rsrc_t rn = rsrc(in + (size_t)cn * 8192, cn < channels ? 32768u : 0u);
        __amdgpu_buffer_rsrc_t ro = rsrc(out + (size_t)c * 1024, 4096);
        u4v acc = { 0, 0, 0, 0 };
        for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += raw[i];
            const __amdgpu_buffer_rsrc_t r = pass < 3 ? rs : rn;
            const int so = pass < 3 ? (pass + 1) * 8192 : 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16 + i * 1024, so, AUX);
            __builtin_amdgcn_raw_buffer_store_b128(acc, ro, lane * 16, pass * 1024, 0);
        }
        for (int j = 0; j < 10; ++j) state[(size_t)c * 640 + j * 64 + lane] = st[j] + __uint_as_float(acc.x);
        rs = rn;
    }
}
int main(int argc, char **argv)
{
    const unsigned C = 65536;
    float *in, *out, *state;
    if (hipMalloc(&in, (size_t)C * 32768) != hipSuccess || hipMalloc(&out, (size_t)C * 4096) != hipSuccess ||
        hipMalloc(&state, (size_t)C * 2560) != hipSuccess) return 1;
    (void)hipMemset(in, 0, (size_t)C * 32768); (void)hipMemset(state, 0, (size_t)C * 2560);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double bytes = (double)C * (32768 + 4096 + 2 * 2560);
    const int grids[] = { 1024, 2048, 4096, 8192, 65536 };
    for (int aux = 0; aux < 2; ++aux)
        for (int g : grids) {
            const size_t lds = g == 1024 ? 40960 : 16384;         // 1024: one wave per SIMD; others two or more
            for (int w = 0; w < 300; ++w) { if (aux) k_stream<2><<<g, 64, lds>>>(in, out, state, C); else k_stream<0><<<g, 64, lds>>>(in, out, state, C); }
            (void)hipEventRecord(e0);
            for (int w = 0; w < 300; ++w) { if (aux) k_stream<2><<<g, 64, lds>>>(in, out, state, C); else k_stream<0><<<g, 64, lds>>>(in, out, state, C); }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 300;
            printf("%s loads, grid %5d: %.4f ms  %.0f GB/s (%.3f GB per launch)\n", aux ? "nt     " : "default", g, ms, bytes / ms / 1e6, bytes / 1e9);
        }
    return 0;
}
