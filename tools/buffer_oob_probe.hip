// buffer_oob_probe.hip -- does the range check of a raw buffer store / load on gfx950 include the scalar offset?
// A 4 KB window in the middle of a 12 KB allocation; stores with soffset = -1024, +4096 (just past the end) and 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__global__ void k(float *base, int soff, unsigned *ld)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base + 1024, 0, 4096, 0x00020000);
    const u4v v = { 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u };
    __builtin_amdgcn_raw_buffer_store_b128(v, r, threadIdx.x * 16, soff, 0);
    const u4v g = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, soff, 0);
    ld[threadIdx.x] = g.x;
}
int main()
{
    float *d; unsigned *ld;
    hipMalloc(&d, 12288); hipMalloc(&ld, 256);
    static unsigned h[3072], hl[64];
    const int offs[3] = { -1024, 4096, 0 };
    for (int t = 0; t < 3; ++t) {
        hipMemset(d, 0x22, 12288);
        k<<<1, 64>>>(d, offs[t], ld);
        hipMemcpy(h, d, 12288, hipMemcpyDeviceToHost); hipMemcpy(hl, ld, 256, hipMemcpyDeviceToHost);
        int before = 0, inside = 0, after = 0;
        for (int i = 0; i < 1024; ++i) before += h[i] == 0x11111111u;
        for (int i = 1024; i < 2048; ++i) inside += h[i] == 0x11111111u;
        for (int i = 2048; i < 3072; ++i) after += h[i] == 0x11111111u;
        printf("soffset %5d: words written before the window %d, inside %d, after %d; load of lane 0 returned %#x\n", offs[t], before, inside, after, hl[0]);
    }
    return 0;
}
