// cw_step_rate.hip -- what one systolic DF1 step of k_cw_fused (csrc/rx_cw.hip) costs a wavefront, alone and with partners on its SIMD:
// single-wave workgroups, W per SIMD (dynamic LDS caps the residency), no memory traffic, the step in several forms.
//
//   0  eight dependent v_add_f32                       1  eight independent v_add_f32
//   2  eight dependent v_pk_mul_f32                    3  eight independent v_pk_mul_f32
//   4  dpp row_shr:1 -> add, dependent, x4
//   5  the product's step x4: dpp move, select, b0 * x, two packed multiplies, four dependent adds
//   6  the same with the delay-line products as four plain multiplies
//   7  "merged" input select: lanes stage-major inside a row (row_shr:4, bank_mask keeps the stage-0 lanes' own input): no select
//   8  form 7 with b0 * x as a DPP multiply (stage-0 lanes multiplied beforehand): the neighbour's output is one instruction from the sum
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/cw_step_rate tools/cw_step_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v2f __attribute__((ext_vector_type(2)));
#pragma clang fp contract(off)

__device__ __forceinline__ float dpp_shr(float v, int)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, const float *__restrict__ xin, int iters, float b0, float b1, float b2, float a1, float a2)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    float r[8];
    v2f pr[8];
    for (int i = 0; i < 8; ++i) { r[i] = lane * 1e-3f + i; pr[i] = v2f{ r[i], r[i] + 0.5f }; }
    const v2f c1 = { b1, a1 }, c2 = { b2, a2 };
    v2f PA = { 0.1f * lane, 0.2f }, PB = { 0.3f, 0.01f * lane };
    const int s = lane & 3;
    const bool first = MODE >= 7 ? ((lane >> 2) & 3) == 0 : s == 0;
    float y = 0.0f, m = 0.0f;
    float4 xq = reinterpret_cast<const float4 *>(xin)[lane & 15];
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[0]) : "v"(b0));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(b0));
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pr[0]) : "v"(c1));
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pr[i]) : "v"(c1));
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { y = dpp_shr(y, 0) + b0; }
        } else {
            auto step = [&](float xs, v2f &A, v2f &B) -> float {
                float p0, xi;
                if constexpr (MODE == 5 || MODE == 6) {
                    const float prev = dpp_shr(A.y, 0);
                    xi = first ? xs : prev;
                    p0 = b0 * xi;
                } else if constexpr (MODE == 7) {
                    // stage-major lanes: lane l of a row takes the output of lane l - 4; bank 0 (the stage-0 lanes) keeps `old` = its own input
                    xi = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(xs), __float_as_int(A.y), 0x114, 0xf, 0xe, false));
                    p0 = b0 * xi;
                } else {
                    xi = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(xs), __float_as_int(A.y), 0x114, 0xf, 0xe, false));
                    float q = b0 * xs;                                  // stage-0 lanes: ready before the neighbour's output exists
                    asm("s_nop 1\n v_mul_f32_dpp %0, %1, %2 row_shr:4 row_mask:0xf bank_mask:0xe" : "+v"(q) : "v"(A.y), "v"(b0));
                    p0 = q;
                }
                float t1x, t1y, t2x, t2y;
                if constexpr (MODE == 6) {
                    t1x = A.x * b1; t1y = A.y * a1; t2x = B.x * b2; t2y = B.y * a2;
                } else {
                    const v2f t1 = A * c1, t2 = B * c2;
                    t1x = t1.x; t1y = t1.y; t2x = t2.x; t2y = t2.y;
                }
                float yy = p0 + t1x;
                yy = yy + t2x;
                yy = yy + t1y;
                yy = yy + t2y;
                B = v2f{ xi, yy };
                return yy;
            };
            const float o0 = step(xq.x, PA, PB);
            const float o1 = step(xq.y, PB, PA);
            const float o2 = step(xq.z, PA, PB);
            const float o3 = step(xq.w, PB, PA);
            m = fmaxf(fmaxf(m, fabsf(o0)), fmaxf(fabsf(o1), fmaxf(fabsf(o2), fabsf(o3))));
            xq.x += 1e-9f;                                               // (keeps the loop from being hoisted; one more instruction per four steps)
        }
    }
    float acc = y + m + PA.x + PA.y + PB.x + PB.y;
    for (int i = 0; i < 8; ++i) acc += r[i] + pr[i].x + pr[i].y;
    out[blockIdx.x * 64 + lane] = acc;
    if (acc == 1.2345f) lds[lane] = acc;
}

template <int MODE>
static void run(const char *name, int per_iter)
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out, *xin;
    (void)hipMalloc(&out, (size_t)cus * 16 * 64 * 4); (void)hipMalloc(&xin, 1024);
    (void)hipMemset(xin, 0, 1024);
    const int iters = 100000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w : { 1, 2, 3, 4 }) {
        const size_t ldsb = w == 1 ? 40000 : w == 2 ? 20000 : w == 3 ? 13000 : 10000;
        const int grid = cus * 4 * w;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), ldsb, 0, out, xin, iters / 10, 0.5f, 0.25f, 0.125f, 0.3f, -0.2f);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), ldsb, 0, out, xin, iters, 0.5f, 0.25f, 0.125f, 0.3f, -0.2f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        const double ns_unit = best * 1e6 / ((double)iters * per_iter);           // per wave
        printf("%-44s %d waves/SIMD: %.3f ms  %.2f ns per unit per wave (%.1f cycles at 2.4 GHz), %.2f ns per unit per SIMD (%.1f cycles)\n", name, w, best, ns_unit,
               ns_unit * 2.4, ns_unit / w, ns_unit / w * 2.4);
    }
    (void)hipFree(out); (void)hipFree(xin);
}

int main()
{
    run<0>("0 dependent v_add_f32 (unit: instruction)", 8);
    run<1>("1 independent v_add_f32", 8);
    run<2>("2 dependent v_pk_mul_f32", 8);
    run<3>("3 independent v_pk_mul_f32", 8);
    run<4>("4 dpp -> add chain (unit: pair)", 4);
    run<5>("5 product step (unit: step)", 4);
    run<6>("6 step, four plain multiplies", 4);
    run<7>("7 step, merged select (stage-major lanes)", 4);
    run<8>("8 step, merged select + DPP multiply", 4);
    return 0;
}
