// tools/mix_split_check.hip -- the f16 hi/lo split by v_fma_mixlo/hi_f16 (rx_split16.hip) against the
// convert / subtract / convert sequence it replaces, bit for bit, on the GPU.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
__global__ void k(const float* src, unsigned* dst, float s) {
  float a = src[threadIdx.x], b = src[threadIdx.x + 64];
  unsigned h, l;
  asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\t"
      "v_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(h) : "v"(a), "v"(b), "s"(s));
  asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=&v"(l) : "v"(a), "v"(b), "s"(s), "v"(h));
  dst[threadIdx.x] = h;
  dst[threadIdx.x + 64] = l;
}
int main() {
  float *s; unsigned *d; hipMalloc(&s, 128*4); hipMalloc(&d, 128*4);
  float hs[128]; for (int i = 0; i < 128; ++i) hs[i] = (i - 50) * 0.013771f * (i % 7 == 0 ? 1e-3f : 1.0f);
  hipMemcpy(s, hs, sizeof hs, hipMemcpyHostToDevice);
  k<<<1,64>>>(s, d, 1024.0f);
  unsigned hd[128]; hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    float a = hs[i] * 1024.0f, b = hs[i + 64] * 1024.0f;
    _Float16 ha = (_Float16)a, hb = (_Float16)b;
    _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    unsigned short uha, uhb, ula, ulb; std::memcpy(&uha, &ha, 2); std::memcpy(&uhb, &hb, 2); std::memcpy(&ula, &la, 2); std::memcpy(&ulb, &lb, 2);
    unsigned eh = uha | (uhb << 16), el = ula | (ulb << 16);
    if (eh != hd[i] || el != hd[i + 64]) { ++bad; if (bad < 5) printf("lane %d: hi %08x want %08x lo %08x want %08x\n", i, hd[i], eh, hd[i+64], el); }
  }
  printf("bad=%d\n", bad);
  return bad != 0;
}
