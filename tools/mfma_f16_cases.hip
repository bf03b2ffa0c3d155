// mfma_f16_cases.hip -- random-case dump for modelling the v_mfma_f32_16x16x32_f16 adder: NCASE x (A[16][32], B[32][16],
// C[16][16], D[16][16]) as f32 to the file given on the command line (analysed by tools/mfma_f16_model.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, const float *C, float *D)
{
    const int l = threadIdx.x, cs = blockIdx.x;
    A += cs * 512; B += cs * 512; C += cs * 256; D += cs * 256;
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
int main(int argc, char **argv)
{
    const int NC = 256;
    std::vector<float> hA(NC * 512), hB(NC * 512), hC(NC * 256), hD(NC * 256);
    unsigned long long s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
    for (int cs = 0; cs < NC; ++cs) {
        const double spreadA = (cs % 4) * 4.0, spreadB = ((cs / 4) % 4) * 3.0;     // exponent spread in bits
        for (int i = 0; i < 512; ++i) {
            hA[cs * 512 + i] = (float)(_Float16)((rnd() * 2 - 1) * 32768.0 * exp2(-spreadA * rnd()));
            hB[cs * 512 + i] = (float)(_Float16)((rnd() * 2 - 1) * 32768.0 * exp2(-spreadB * rnd()));
        }
        const double cscale = (cs / 16) % 4 == 0 ? 0.0 : exp2(30.0 + 3 * ((cs / 16) % 4));
        for (int i = 0; i < 256; ++i) hC[cs * 256 + i] = (float)((rnd() * 2 - 1) * cscale);
    }
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, hC.size() * 4); hipMalloc(&dD, hD.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC.data(), hC.size() * 4, hipMemcpyHostToDevice);
    k<<<NC, 64>>>(dA, dB, dC, dD);
    if (hipMemcpy(hD.data(), dD, hD.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    FILE *f = fopen(argc > 1 ? argv[1] : "mfma_cases.bin", "wb");
    if (!f) return 3;
    fwrite(hA.data(), 4, hA.size(), f); fwrite(hB.data(), 4, hB.size(), f); fwrite(hC.data(), 4, hC.size(), f); fwrite(hD.data(), 4, hD.size(), f);
    fclose(f);
    printf("wrote %d cases\n", NC);
    return 0;
}
