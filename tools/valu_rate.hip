// valu_rate.hip -- micro-benchmark: f32 VALU issue rates on gfx950 that bound the exact
// (mul,add) and fused (fma) FIR tap loops.  Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITERS 4096
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float c0, float c1)
{
    float a[8];
    float2_ pa[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; pa[i] = float2_{a[i], a[i] + 0.5f}; }
    float x = c0, y = c1;
    float2_ px = {c0, c1}, py = {c1, c0};
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {            // v_mul_f32 + v_add_f32 (separately rounded)
                float p;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(a[i]));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(p), "v"(y));
            } else if (MODE == 1) {     // v_fma_f32
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(x), "v"(a[i]), "v"(y));
            } else if (MODE == 2) {     // v_pk_mul_f32 + v_pk_add_f32
                float2_ p;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(px), "v"(pa[i]));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pa[i]) : "v"(p), "v"(py));
            } else if (MODE == 3) {     // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pa[i]) : "v"(px), "v"(pa[i]), "v"(py));
            } else if (MODE == 4) {     // v_fmac_f32 (VOP2)
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            } else if (MODE == 5) {     // v_mul_f32 with SGPR coefficient + v_add_f32
                float p;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "s"(c0), "v"(a[i]));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(p), "v"(y));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + pa[i].x + pa[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, double lane_ops_per_inst_pair, int waves_per_simd)
{
    float *d; hipMalloc(&d, 256 * 4 * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * waves_per_simd), blk(256);
    k<MODE><<<grid, blk>>>(d, 1.0001f, 1e-9f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, blk>>>(d, 1.0001f, 1e-9f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double macs = (double)grid.x * blk.x * N_ITERS * 8 * lane_ops_per_inst_pair;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f TMAC/s\n", name, waves_per_simd, ms, macs / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_mul+v_add", 1, w);
        run<5>("v_mul(sgpr)+v_add", 1, w);
        run<1>("v_fma", 1, w);
        run<4>("v_fmac", 1, w);
        run<2>("v_pk_mul+v_pk_add", 2, w);
        run<3>("v_pk_fma", 2, w);
    }
    return 0;
}
