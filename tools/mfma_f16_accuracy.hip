// mfma_f16_accuracy.hip -- how v_mfma_f32_16x16x32_f16 adds its 32 products and the accumulator (gfx950).
// Row i of A x column 0 of B is a chosen list of 32 products; C is chosen too.  Prints the result next to the exact
// sum so that the internal alignment width and the rounding of the adder can be read off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, const float *C, float *D)   // A[16][32], B[32][16], C[16][16]
{
    const int l = threadIdx.x;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)A[(l & 15) * 32 + 8 * (l >> 4) + j];
        b[j] = (_Float16)B[(8 * (l >> 4) + j) * 16 + (l & 15)];
    }
    f4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * (l >> 4) + r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
int main()
{
    static float hA[16 * 32], hB[32 * 16], hC[256], hD[256];
    double exact[16];
    for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) hB[k * 16 + n] = 1.0f;     // B = ones: D[i][*] = sum_k A[i][k] + C
    // row 0..11: one product 2^12 * ... we need big products: use B column scale instead: row i: A[i][0] = 2^12 (times B=1)
    // products are A values themselves (B = 1), so the spread is limited to f16 range: big = 2^15, small = 2^-g
    for (int i = 0; i < 16; ++i) {
        for (int k = 0; k < 32; ++k) hA[i * 32 + k] = 0.0f;
        for (int n = 0; n < 16; ++n) hC[i * 16 + n] = 0.0f;
    }
    // rows 0..7: big = 2^15 at k=0, 31 addends of 2^(15-24+i-3): relative 2^-27 .. 2^-20 each
    for (int i = 0; i < 8; ++i) {
        hA[i * 32] = 32768.0f;
        for (int k = 1; k < 32; ++k) hA[i * 32 + k] = ldexpf(1.0f, 15 - 27 + i);
    }
    // rows 8..11: the same small addends with the big value in C instead of in a product
    for (int i = 8; i < 12; ++i) {
        for (int n = 0; n < 16; ++n) hC[i * 16 + n] = 32768.0f;
        for (int k = 0; k < 32; ++k) hA[i * 32 + k] = ldexpf(1.0f, 15 - 27 + 2 * (i - 8));
    }
    // row 12: rounding of the final result: 2^15 + 31 * 2^-9 * ... exact sum needs 25+ bits
    hA[12 * 32] = 32768.0f; for (int k = 1; k < 32; ++k) hA[12 * 32 + k] = (k & 1) ? ldexpf(1.0f, -9) : ldexpf(1.5f, -9);
    // row 13: cancellation: +2^15, -2^15, then small ones
    hA[13 * 32] = 32768.0f; hA[13 * 32 + 1] = -32768.0f; for (int k = 2; k < 32; ++k) hA[13 * 32 + k] = ldexpf(1.0f, -14);
    // row 14: negative big + positive small (sign of the truncation)
    hA[14 * 32] = -32768.0f; for (int k = 1; k < 32; ++k) hA[14 * 32 + k] = ldexpf(1.0f, 15 - 25);
    // row 15: pseudo-random same-sign values
    for (int k = 0; k < 32; ++k) hA[15 * 32 + k] = (float)(_Float16)(1000.0f + 37.77f * k);
    for (int i = 0; i < 16; ++i) { double s = hC[i * 16]; for (int k = 0; k < 32; ++k) s += (double)(float)(_Float16)hA[i * 32 + k]; exact[i] = s; }
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i)
        printf("row %2d: mfma %.10g  exact %.10g  rn(exact) %.10g  diff/ulp %.3f\n", i, hD[i * 16], exact[i], (double)(float)exact[i],
               (hD[i * 16] - exact[i]) / ldexp(1.0, ilogb(exact[i] == 0 ? 1 : fabs(exact[i])) - 23));
    return 0;
}
