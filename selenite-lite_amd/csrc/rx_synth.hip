// rx_synth.hip -- synthetic I/Q generator of SURVEY.md 8d (build-defined, measurement support).
//
// Per channel: three complex tones (integer phase accumulators, table-lerp sin/cos in CMSIS
// arithmetic) + uniform noise from a counter-based 64-bit mixer.  One source, compiled for host
// and device, fixed f32 operation order, no contraction: identical bits on both.
#include "rx_internal.h"

#pragma clang fp contract(off)

namespace srx {

__host__ __device__ static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ static inline uint64_t splitmix64(uint64_t &s)
{
    s += 0x9E3779B97F4A7C15ull;
    return mix64(s);
}

struct SynthChan {
    uint32_t step[3], ph0[3];
    uint64_t nseed;
};

__host__ __device__ static inline SynthChan synth_chan(uint32_t c, uint64_t seed)
{
    SynthChan k;
    uint64_t s = seed ^ ((uint64_t)c * 0xD1B54A32D192ED03ull);
    const uint64_t r0 = splitmix64(s), r1 = splitmix64(s), r2 = splitmix64(s), r3 = splitmix64(s);
    k.step[0] = 0x02000000u + ((uint32_t)r0 & 0x00FFFFFFu);   // fs/128 .. 1.5 fs/128
    k.step[1] = (uint32_t)(r0 >> 32);
    k.step[2] = (uint32_t)r1;
    k.ph0[0] = (uint32_t)(r1 >> 32);
    k.ph0[1] = (uint32_t)r2;
    k.ph0[2] = (uint32_t)(r2 >> 32);
    k.nseed = r3;
    return k;
}

__host__ __device__ static inline float2 synth_sample(const SynthChan &k, const float *T, uint64_t n)
{
    const float amp[3] = { 0.4f, 0.2f, 0.1f };
    float vi = 0.0f, vq = 0.0f;
    for (int t = 0; t < 3; ++t) {
        const uint32_t ph = k.ph0[t] + (uint32_t)n * k.step[t];
        const float x = (float)(ph >> 8) * kNcoK;
        const float cs = cos_f32<0>(T, x), sn = sin_f32<0>(T, x);
        const float pc = amp[t] * cs, ps = amp[t] * sn;
        vi = vi + pc;
        vq = vq + ps;
    }
    const uint64_t h = mix64(k.nseed + n * 0x9E3779B97F4A7C15ull);
    const float ui = (float)(uint32_t)(h >> 40) * 0x1p-24f;
    const float uq = (float)(uint32_t)((h >> 16) & 0xFFFFFFu) * 0x1p-24f;
    const float ni = (ui - 0.5f) * 0.1f, nq = (uq - 0.5f) * 0.1f;
    return make_float2(vi + ni, vq + nq);
}

__global__ __launch_bounds__(256) void k_synth(float2 *iq, const float *sintab, uint32_t first_channel,
                                               uint32_t nch, uint64_t first_sample, uint32_t nsamp,
                                               uint64_t seed)
{
    __shared__ float tab[516];
    for (uint32_t i = threadIdx.x; i < 513; i += blockDim.x) tab[i] = sintab[i];
    __syncthreads();
    const uint32_t ci = blockIdx.y;
    const SynthChan k = synth_chan(first_channel + ci, seed);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nsamp; i += gridDim.x * blockDim.x)
        iq[(size_t)ci * nsamp + i] = synth_sample(k, tab, first_sample + i);
}

hipError_t launch_synth(float *dIQ, const float *sintab, uint32_t first_channel, uint32_t nch,
                        uint64_t first_sample, uint32_t nsamp, uint64_t seed, hipStream_t st)
{
    if (nch == 0 || nsamp == 0) return hipSuccess;
    uint32_t gx = (nsamp + 255) / 256;
    if (gx > 64) gx = 64;
    // gridDim.y is limited to 65535: loop over channel slabs
    for (uint32_t c0 = 0; c0 < nch; c0 += 32768) {
        const uint32_t n = (nch - c0 < 32768) ? (nch - c0) : 32768;
        hipLaunchKernelGGL(k_synth, dim3(gx, n), dim3(256), 0, st,
                           reinterpret_cast<float2 *>(dIQ) + (size_t)c0 * nsamp, sintab,
                           first_channel + c0, n, first_sample, nsamp, seed);
    }
    return hipGetLastError();
}

void synth_host(float *iq, const float *sintab, uint32_t first_channel, uint32_t nch,
                uint64_t first_sample, uint32_t nsamp, uint64_t seed)
{
    for (uint32_t ci = 0; ci < nch; ++ci) {
        const SynthChan k = synth_chan(first_channel + ci, seed);
        float *out = iq + (size_t)ci * nsamp * 2;
        for (uint32_t i = 0; i < nsamp; ++i) {
            const float2 v = synth_sample(k, sintab, first_sample + i);
            out[2 * i] = v.x;
            out[2 * i + 1] = v.y;
        }
    }
}

}  // namespace srx
