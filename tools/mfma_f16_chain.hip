// mfma_f16_chain.hip <in.bin> <out.bin> -- runs chains of KS v_mfma_f32_16x16x32_f16 (one accumulator) over operand tiles
// prepared by tools/mfma_f16_chain.py: header int32 {ncase, ks}, then per case ks x (A[16][32], B[32][16]) f32 values
// that are exactly f16-representable.  Output: D[16][16] f32 per case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *T, float *D, int ks)
{
    const int l = threadIdx.x, cs = blockIdx.x;
    T += (size_t)cs * ks * 1024; D += cs * 256;
    f4 c = {0, 0, 0, 0};
    for (int s = 0; s < ks; ++s) {
        const float *A = T + s * 1024, *B = A + 512;
        h8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = (_Float16)A[(l & 15) * 32 + 8 * (l >> 4) + j];
            b[j] = (_Float16)B[(8 * (l >> 4) + j) * 16 + (l & 15)];
        }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
int main(int argc, char **argv)
{
    if (argc < 3) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[2];
    if (fread(hdr, 4, 2, f) != 2) return 3;
    const int nc = hdr[0], ks = hdr[1];
    std::vector<float> T((size_t)nc * ks * 1024), D((size_t)nc * 256);
    if (fread(T.data(), 4, T.size(), f) != T.size()) return 4;
    fclose(f);
    float *dT, *dD;
    hipMalloc(&dT, T.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dT, T.data(), T.size() * 4, hipMemcpyHostToDevice);
    k<<<nc, 64>>>(dT, dD, ks);
    if (hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 5;
    f = fopen(argv[2], "wb");
    fwrite(D.data(), 4, D.size(), f);
    fclose(f);
    printf("ran %d chains of %d\n", nc, ks);
    return 0;
}
