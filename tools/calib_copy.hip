// calib_copy.hip -- known-byte-count kernels to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on
// gfx950 for the access widths the RX kernels use (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads
// half of a wide coalesced stream; other widths and WRITE_SIZE must be calibrated).
//   k_copy16: 16 B/lane loads, 16 B/lane stores      (the fused kernel's I/Q load and audio store)
//   k_copy8 :  8 B/lane loads,  8 B/lane stores      (q15 slot loads)
//   k_copy4 :  4 B/lane loads,  4 B/lane stores      (state read / write-back)
// Each moves exactly N bytes in and N bytes out (N = 1 GiB, larger than the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ __launch_bounds__(256) void k_copy(const T *__restrict__ a, T *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main()
{
    const size_t N = 1ull << 30;
    void *a, *b;
    if (hipMalloc(&a, N) != hipSuccess || hipMalloc(&b, N) != hipSuccess) return 1;
    (void)hipMemset(a, 1, N); (void)hipMemset(b, 0, N);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        k_copy<float4><<<2048, 256>>>((const float4 *)a, (float4 *)b, N / 16);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("copy16 %.3f ms  %.1f GB/s (read+write)\n", ms, 2.0 * N / ms / 1e6);
        k_copy<float2><<<2048, 256>>>((const float2 *)a, (float2 *)b, N / 8);
        k_copy<float><<<2048, 256>>>((const float *)a, (float *)b, N / 4);
        (void)hipDeviceSynchronize();
    }
    return 0;
}
