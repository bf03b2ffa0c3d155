// calib_stream_tx.hip -- the streaming roof of the TX traffic shape (write-dominated): per channel 4 KB of audio in, 32 KB of
// I/Q out, 1 KB of state in and out; one single-wave workgroup per channel (the launch shape of k_tx_split16) and a
// persistent grid; 1 KB buffer stores, no arithmetic.  Steady state (300 + 300 launches).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
template <int AUX>
__global__ __launch_bounds__(64, 2) void k_stream(const float *__restrict__ in, float *__restrict__ out, float *__restrict__ state, unsigned channels)
{
    const int lane = threadIdx.x;
    for (unsigned c = blockIdx.x; c < channels; c += gridDim.x) {
        const __amdgpu_buffer_rsrc_t ri = rsrc(in + (size_t)c * 1024, 4096), ro = rsrc(out + (size_t)c * 8192, 32768);
        float st[4];
        for (int j = 0; j < 4; ++j) st[j] = state[(size_t)c * 256 + j * 64 + lane];
        u4v a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = __builtin_amdgcn_raw_buffer_load_b128(ri, lane * 16 + i * 1024, 0, 0);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u4v v = a[pass]; v.x += i;
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, lane * 16 + i * 1024, pass * 8192, AUX);
            }
        }
        for (int j = 0; j < 4; ++j) state[(size_t)c * 256 + j * 64 + lane] = st[j] + __uint_as_float(a[0].x);
    }
}
int main()
{
    const unsigned C = 65536;
    float *in, *out, *state;
    if (hipMalloc(&in, (size_t)C * 4096) != hipSuccess || hipMalloc(&out, (size_t)C * 32768) != hipSuccess || hipMalloc(&state, (size_t)C * 1024) != hipSuccess) return 1;
    (void)hipMemset(in, 0, (size_t)C * 4096); (void)hipMemset(state, 0, (size_t)C * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double bytes = (double)C * (4096 + 32768 + 2 * 1024);
    const int grids[] = { 2048, 4096, 65536 };
    for (int aux = 0; aux < 2; ++aux)
        for (int g : grids) {
            for (int w = 0; w < 300; ++w) { if (aux) k_stream<2><<<g, 64, 16384>>>(in, out, state, C); else k_stream<0><<<g, 64, 16384>>>(in, out, state, C); }
            (void)hipEventRecord(e0);
            for (int w = 0; w < 300; ++w) { if (aux) k_stream<2><<<g, 64, 16384>>>(in, out, state, C); else k_stream<0><<<g, 64, 16384>>>(in, out, state, C); }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 300;
            printf("%s stores, grid %5d: %.4f ms  %.0f GB/s (%.3f GB per launch)\n", aux ? "nt     " : "default", g, ms, bytes / ms / 1e6, bytes / 1e9);
        }
    return 0;
}
