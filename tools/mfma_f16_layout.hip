// mfma_f16_layout.hip -- empirical check of the v_mfma_f32_16x16x32_f16 operand layout assumed by
// the split-precision decimator: A[i][k]: lane l holds i = l&15, k = 8*(l>>4)+j (j = 0..7);
// B[k][n]: lane l holds n = l&15, k = 8*(l>>4)+j; D[i][n]: lane l holds i = 4*(l>>4)+r, n = l&15.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, float *D)   // A[16][32], B[32][16] row-major f32 (exactly f16-representable)
{
    const int l = threadIdx.x;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)A[(l & 15) * 32 + 8 * (l >> 4) + j];
        b[j] = (_Float16)B[(8 * (l >> 4) + j) * 16 + (l & 15)];
    }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
int main()
{
    float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) hA[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) hB[k * 16 + n] = (float)((k * 5 + n * 13) % 9 - 4) * 0.5f;
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += hA[i * 32 + k] * hB[k * 16 + n]; ref[i * 16 + n] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (hD[i] != ref[i]) ++bad;
    printf("mfma_f32_16x16x32_f16 layout check: %d mismatches of 256\n", bad);
    return bad != 0;
}
