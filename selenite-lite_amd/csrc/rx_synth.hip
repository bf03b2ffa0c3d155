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



// ------------------------------------------------------------------------------------------
// k_stream_roof -- measurement support (bench.py `streaming_roof`, SURVEY.md 8d "practical HBM ceiling"): the traffic of one
// process call with NO arithmetic.  Launched the way the fused kernels are: a persistent grid of single-wave workgroups, workgroup
// b streams channels b, b + G, ...; per channel `in_bytes` of input in 1 KB non-temporal buffer loads (eight in flight, the next
// eight always prefetched, across the channel boundary too), `state_bytes` of state read and written back, `out_bytes` of
// audio written as 1 KB non-temporal stores spread evenly over the passes.  One integer add per loaded register keeps the loads
// alive.  What this kernel takes is the floor for ANY kernel that moves the algorithmic bytes of the call with this access
// pattern; the DSP kernels are reported as a fraction of it next to the fraction of the nominal 8 TB/s.
// ------------------------------------------------------------------------------------------
typedef unsigned int u4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t roof_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

__global__ __launch_bounds__(64, 2) void k_stream_roof(const char *__restrict__ in, char *__restrict__ out, float *__restrict__ state,
                                                       uint32_t channels, uint32_t in_bytes, uint32_t out_bytes, uint32_t state_words,
                                                       size_t in_stride, size_t out_stride)
{
    const int lane = threadIdx.x;
    constexpr int NL = 8;                                         // 1 KB wave loads per pass
    const uint32_t npass = (in_bytes + NL * 1024u - 1u) / (NL * 1024u);
    const uint32_t out_per_pass = ((out_bytes + npass - 1u) / npass + 1023u) & ~1023u;    // whole 1 KB stores; the range check drops the surplus
    u4s raw[NL];
    uint32_t c = blockIdx.x;
    if (c >= channels) return;
    __amdgpu_buffer_rsrc_t rs = roof_rsrc(in + (size_t)c * in_stride, in_bytes);
#pragma unroll
    for (int i = 0; i < NL; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + i * 1024, 0, 2);
    for (; c < channels; c += gridDim.x) {
        float *st = state + (size_t)c * state_words;
        float sv[16];
        const uint32_t nsv = (state_words + 63u) / 64u;           // <= 16 (asserted on the host)
        for (uint32_t j = 0; j < nsv; ++j) sv[j] = (j * 64u + lane < state_words) ? st[j * 64u + lane] : 0.0f;
        const uint32_t cn = c + gridDim.x;
        const __amdgpu_buffer_rsrc_t rn = roof_rsrc(in + (size_t)cn * in_stride, cn < channels ? in_bytes : 0u);
        const __amdgpu_buffer_rsrc_t ro = roof_rsrc(out + (size_t)c * out_stride, out_bytes);
        u4s acc = { 0u, 0u, 0u, 0u };
        for (uint32_t pass = 0; pass < npass; ++pass) {
#pragma unroll
            for (int i = 0; i < NL; ++i) acc += raw[i];
            const bool last = pass + 1 == npass;
            const __amdgpu_buffer_rsrc_t r = last ? rn : rs;
            const int so = last ? 0 : (int)((pass + 1) * NL * 1024u);
#pragma unroll
            for (int i = 0; i < NL; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16 + i * 1024, so, 2);
            for (uint32_t o = 0; o < out_per_pass; o += 1024u)
                __builtin_amdgcn_raw_buffer_store_b128(acc, ro, lane * 16, (int)(pass * out_per_pass + o), 2);
        }
        for (uint32_t j = 0; j < nsv; ++j)
            if (j * 64u + lane < state_words) st[j * 64u + lane] = sv[j] + __uint_as_float(acc.x & 1u);
        rs = rn;
    }
}

hipError_t launch_stream_roof(const void *in, void *out, float *state, uint32_t channels, uint32_t in_bytes, uint32_t out_bytes,
                              uint32_t state_words, hipStream_t st)
{
    if (state_words > 16u * 64u) return hipErrorInvalidValue;
    static int resident = 0;
    if (resident == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_stream_roof, 64, 0) == hipSuccess && per_cu > 0 &&
            hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            resident = (per_cu < 8 ? per_cu : 8) * prop.multiProcessorCount;     // 8 single-wave workgroups per CU: the launch shape of k_ssb_split16
        else
            resident = 2048;
    }
    const uint32_t grid = (uint32_t)resident < channels ? (uint32_t)resident : channels;
    hipLaunchKernelGGL(k_stream_roof, dim3(grid), dim3(64), 0, st, static_cast<const char *>(in), static_cast<char *>(out), state,
                       channels, in_bytes, out_bytes, state_words, (size_t)in_bytes, (size_t)out_bytes);
    return hipGetLastError();
}

}  // namespace srx
