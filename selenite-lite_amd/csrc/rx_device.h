// rx_device.h -- device-side primitives of the Selenite RX block path (gfx950).
//
// Each helper reproduces, operation for operation, one CMSIS-DSP 1.5.3 f32 primitive of the
// reference (citations relative to /root/reference/Drivers/CMSIS/DSP/Source).  ARITH selects the
// rounding contract (include/selenite_rx.h):
//   0  SELENITE_ARITH_CMSIS  product rounded, then sum rounded (what the reference C code does)
//   1  SELENITE_ARITH_FMA    same order, the multiply-add of the FIR tap loops (mac<>) fused;
//                            NCO, biquad recurrence, AGC law and magnitude keep the reference
//                            rounding in both modes (fusing the IIR recurrence moves a high-Q
//                            cascade by more than the 1e-5 the north star allows)
// The translation unit is compiled with -ffp-contract=off; fusion happens only through
// __builtin_fmaf below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace srx {

constexpr int kWave = 64;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// 16-byte LDS access as ONE ds_read_b128 / ds_write_b128.  HIP's float4 is a struct whose loads the backend may only assume to
// be 8-byte aligned: it then emits ds_read2_b64 -- two 8-byte accesses banked modulo 32 dwords, i.e. 2-way conflicts at a
// 16-byte lane stride and four times the LDS cycles of a ds_read_b128 (measured: the Hilbert reads of k_ssb_split16 owned
// 69 % of the kernel's bank-conflict cycles, profiles/r3/lds_conflicts.txt).  A native 4-vector carries the alignment.
__device__ __forceinline__ v4f lds_ld4(const float *p) { return *reinterpret_cast<const v4f *>(__builtin_assume_aligned(p, 16)); }
__device__ __forceinline__ void lds_st4(float *p, v4f v) { *reinterpret_cast<v4f *>(__builtin_assume_aligned(p, 16)) = v; }
__device__ __forceinline__ float4 lds_ld4f(const float *p) { const v4f v = lds_ld4(p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void lds_st4f(float *p, float4 v) { lds_st4(p, v4f{ v.x, v.y, v.z, v.w }); }

constexpr float kNcoK = 0x1.921fb6p-22f;        // 2*pi / 2^24: radians per (phase >> 8) unit
constexpr float kInv2Pi = 0.159154943092f;      // arm_sin_f32.c:88, arm_cos_f32.c:81

template <int ARITH>
__host__ __device__ __forceinline__ float mac(float acc, float x, float c)
{
    if constexpr (ARITH == 1) {
        return __builtin_fmaf(x, c, acc);
    } else {
        float p = x * c;
        return acc + p;
    }
}

// FastMathFunctions/arm_sin_f32.c:72-119; T = sinTable_f32[513] (LDS or global)
template <int ARITH>
__host__ __device__ __forceinline__ float sin_f32(const float *T, float x)
{
    if ((x < 0.0f) && (x >= -1.9e-7f)) return x;
    float in = x * kInv2Pi;
    int n = (int)in;
    if (x < 0.0f) n--;
    in = in - (float)n;
    float findex = 512.0f * in;
    unsigned index = ((unsigned)(unsigned short)findex) & 0x1ffu;
    float fract = findex - (float)index;
    float a = T[index], b = T[index + 1];
    float w = 1.0f - fract;
    float p0 = w * a, p1 = fract * b;
    return p0 + p1;
}

// FastMathFunctions/arm_cos_f32.c:70-111
template <int ARITH>
__host__ __device__ __forceinline__ float cos_f32(const float *T, float x)
{
    float p = x * kInv2Pi;
    float in = p + 0.25f;
    int n = (int)in;
    if (in < 0.0f) n--;
    in = in - (float)n;
    float findex = 512.0f * in;
    unsigned index = ((unsigned)(unsigned short)findex) & 0x1ffu;
    float fract = findex - (float)index;
    float a = T[index], b = T[index + 1];
    float w = 1.0f - fract;
    float p0 = w * a, p1 = fract * b;
    return p0 + p1;
}

// NCO local oscillator sample for an integer phase: LO = cos(x) - j sin(x),
// x = (float)(phase >> 8) * 2pi/2^24  (build-defined, DESIGN.md "NCO")
template <int ARITH>
__host__ __device__ __forceinline__ float2 nco_lo(const float *T, uint32_t phase)
{
    float x = (float)(phase >> 8) * kNcoK;
    float c = cos_f32<ARITH>(T, x);
    float s = sin_f32<ARITH>(T, x);
    return make_float2(c, -s);
}

// Two LO samples (integer phases pe, po) at once, the same arithmetic as nco_lo<0> restated for the vector ALU:
//   * x >= 0 here (x comes from an unsigned phase), so the negative-argument branches of arm_sin_f32.c:77-95 and
//     arm_cos_f32.c:86-89 never run;
//   * for in >= 0, `in - (float)(int)in` is in - floor(in), which v_fract_f32 returns exactly (the difference is
//     always representable and < 1); likewise findex - (float)index with index = (uint16)findex, findex in [0, 512);
//   * every multiply / add is the reference's, in the reference's order, two at a time in v_pk_* (no contraction).
// tests/test_nco_fast_path.py checks the restatement against the oracle for all 2^24 phase values (numpy model, CPU)
// and the kernels that use it against the oracle's mixed samples (GPU).  T: sinTable_f32[513] in LDS.
typedef float lo_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nco_lo_pair(const float *T, uint32_t pe, uint32_t po, lo_v2f &le, lo_v2f &lo)
{
    const lo_v2f u = { (float)(pe >> 8), (float)(po >> 8) };
    const lo_v2f x = u * kNcoK;
    const lo_v2f ps = x * kInv2Pi;                              // arm_sin_f32.c:88 / arm_cos_f32.c:81
    const lo_v2f pc = ps + 0.25f;                               // arm_cos_f32.c:81
    const lo_v2f is = { __builtin_amdgcn_fractf(ps.x), __builtin_amdgcn_fractf(ps.y) };
    const lo_v2f ic = { __builtin_amdgcn_fractf(pc.x), __builtin_amdgcn_fractf(pc.y) };
    const lo_v2f fs = is * 512.0f, fc = ic * 512.0f;            // findex
    const uint32_t s0 = (uint32_t)fs.x, s1 = (uint32_t)fs.y, c0 = (uint32_t)fc.x, c1 = (uint32_t)fc.y;
    const lo_v2f rs = { __builtin_amdgcn_fractf(fs.x), __builtin_amdgcn_fractf(fs.y) };      // fract
    const lo_v2f rc = { __builtin_amdgcn_fractf(fc.x), __builtin_amdgcn_fractf(fc.y) };
    const lo_v2f as = { T[s0], T[s1] }, bs = { T[s0 + 1], T[s1 + 1] };
    const lo_v2f ac = { T[c0], T[c1] }, bc = { T[c0 + 1], T[c1 + 1] };
    const lo_v2f ws = 1.0f - rs, wc = 1.0f - rc;
    const lo_v2f s = ws * as + rs * bs;                         // (1 - fract) * a + fract * b
    const lo_v2f c = wc * ac + rc * bc;
    le = lo_v2f{ c.x, -s.x };
    lo = lo_v2f{ c.y, -s.y };
}

// ComplexMathFunctions/arm_cmplx_mult_cmplx_f32.c:186-187
template <int ARITH>
__device__ __forceinline__ float2 cmul(float2 A, float2 B)
{
    float a = A.x, b = A.y, c = B.x, d = B.y;
    float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    return make_float2(ac - bd, ad + bc);
}

// ComplexMathFunctions/arm_cmplx_mag_f32.c:72-149 (arm_sqrt_f32: arm_math.h:5726-5752)
template <int ARITH>
__device__ __forceinline__ float cmag(float re, float im)
{
    float rr = re * re, ii = im * im;
    float s = rr + ii;
    return (s >= 0.0f) ? sqrtf(s) : 0.0f;   // correctly rounded (hipcc default); __fsqrt_rn is not
}

// FM discriminator (build-defined, DESIGN.md section 2 "FM"; oracle/fm_atan.h states the same operations): the arctangent
// CMSIS-DSP 1.5.3 does not have -- t = min / max (correctly rounded), atan(t) / t as a degree-8 polynomial in t^2 by Horner
// (product rounded, then sum rounded: no fused multiply-add in any arithmetic mode), octant fix-ups
__device__ __forceinline__ float fm_atan2(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    const float t = __fdiv_rn(mn, mx);
    const float s = t * t;
    constexpr float c[9] = { 0x1.000000p+0f, -0x1.5554a2p-2f, 0x1.997232p-3f, -0x1.22de60p-3f, 0x1.b3ae74p-4f,
                             -0x1.330372p-4f, 0x1.5ce0b0p-5f, -0x1.0639f6p-6f, 0x1.73776ap-9f };
    float p = c[8];
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        p = p * s;
        p = p + c[k];
    }
    float r = p * t;
    if (ay > ax) r = 0x1.921fb6p+0f - r;
    if (x < 0.0f) r = 0x1.921fb6p+1f - r;
    if (y < 0.0f) r = -r;
    return r;
}
// z[n] * conj(z[n-1]) (arm_cmplx_conj_f32.c:161-162, arm_cmplx_mult_cmplx_f32.c:186-187), its angle in half turns (arm_scale_f32)
__device__ __forceinline__ float fm_disc(float zr, float zi, float pr, float pi_)
{
    const float c = pr, d = -pi_;
    const float ac = zr * c, bd = zi * d, ad = zr * d, bc = zi * c;
    const float re = ac - bd, im = ad + bc;
    return fm_atan2(im, re) * 0x1.45f306p-2f;
}

// FilteringFunctions/arm_biquad_cascade_df1_f32.c:220 -- one DF1 section, left-to-right sum,
// feedback added
template <int ARITH>
__device__ __forceinline__ float biquad_step(float b0, float b1, float b2, float a1, float a2,
                                             float x, float &x1, float &x2, float &y1, float &y2)
{
    float p0 = b0 * x, p1 = b1 * x1, p2 = b2 * x2, p3 = a1 * y1, p4 = a2 * y2;
    float y = p0 + p1;
    y = y + p2;
    y = y + p3;
    y = y + p4;
    x2 = x1; x1 = x; y2 = y1; y1 = y;
    return y;
}

struct AgcParams {
    float target, attack, decay, gain_min, gain_max, env_floor;
};

// AGC gain law (build-defined, DESIGN.md "AGC"); same statement order as oracle/rx_oracle.c
template <int ARITH>
__device__ __forceinline__ float agc_update(const AgcParams &p, float gain, float env)
{
    float e = (env < p.env_floor) ? p.env_floor : env;
    float d = __fdiv_rn(p.target, e);
    if (d > p.gain_max) d = p.gain_max;
    if (d < p.gain_min) d = p.gain_min;
    float diff = d - gain;
    float rate = (diff < 0.0f) ? p.attack : p.decay;
    float q = rate * diff;
    return gain + q;
}

// the same law in two pieces (identical operations, identical order per block): the desired gain of a
// block depends only on its envelope, the recurrence only on the previous gain
__device__ __forceinline__ float agc_desired(const AgcParams &p, float env)
{
    float e = (env < p.env_floor) ? p.env_floor : env;
    float d = __fdiv_rn(p.target, e);
    if (d > p.gain_max) d = p.gain_max;
    if (d < p.gain_min) d = p.gain_min;
    return d;
}
__device__ __forceinline__ float agc_step(const AgcParams &p, float gain, float d)
{
    float diff = d - gain;
    float rate = (diff < 0.0f) ? p.attack : p.decay;
    float q = rate * diff;
    return gain + q;
}

// SupportFunctions/arm_q15_to_float.c:87 and arm_float_to_q15.c:117 (ARM_MATH_ROUNDING off: what the firmware builds) / :90-101 (`round`:
// the ARM_MATH_ROUNDING variant -- in = in * 32768; in += in > 0 ? 0.5 : -0.5; truncate, saturate -- selenite_rx_config::q15_rounding)
__device__ __forceinline__ float q15_to_float(int16_t v) { return (float)v / 32768.0f; }
__device__ __forceinline__ int16_t float_to_q15(float f, uint32_t round = 0u)
{
    float v = f * 32768.0f;
    if (round) v = v + (v > 0.0f ? 0.5f : -0.5f);
    int q = (int)v;                      // v_cvt_i32_f32: truncates, saturates at int32 range
    q = q > 32767 ? 32767 : q;
    q = q < -32768 ? -32768 : q;
    return (int16_t)q;
}

// Four audio samples as two dwords of int16, the form the fused kernels store: the same conversion (v_cvt_i32_f32 truncates and saturates at the
// int32 range, v_cvt_pk_i16_i32 saturates to 16 bit and packs).  The ARM_MATH_ROUNDING build is a wave-uniform BRANCH the compiler cannot turn
// into selects (the empty volatile asm cannot be speculated): the truncating build, which is the firmware's, pays one scalar test per store.
typedef short q15x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void float4_to_q15(float a, float b, float c, float d, uint32_t round, uint32_t &w0, uint32_t &w1)
{
    a *= 32768.0f; b *= 32768.0f; c *= 32768.0f; d *= 32768.0f;
    if (round) {
        a = a + (a > 0.0f ? 0.5f : -0.5f); b = b + (b > 0.0f ? 0.5f : -0.5f);
        c = c + (c > 0.0f ? 0.5f : -0.5f); d = d + (d > 0.0f ? 0.5f : -0.5f);
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    w0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)a, (int)b));
    w1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)c, (int)d));
}

// max over the 64 lanes of a wavefront (exact: max is associative/commutative for non-NaN)
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// DPP row rotate inside each row of 16 lanes: lane l receives lane (l & ~15) | ((l + n) & 15)
template <int N>
__device__ __forceinline__ uint32_t row_ror(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xf, 0xf, false);
}
// maximum over each row of 16 lanes, left in every lane of the row (4 v_max_*_dpp, no LDS round trip)
__device__ __forceinline__ uint32_t row16_umax(uint32_t v)
{
    v = max(v, row_ror<8>(v));
    v = max(v, row_ror<4>(v));
    v = max(v, row_ror<2>(v));
    v = max(v, row_ror<1>(v));
    return v;
}
__device__ __forceinline__ float row16_fmax(float v)
{
    v = fmaxf(v, __uint_as_float(row_ror<8>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(row_ror<4>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(row_ror<2>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(row_ror<1>(__float_as_uint(v))));
    return v;
}
// wave-uniform maximum of non-negative floats through their bit patterns (monotone for x >= 0): the
// cross-row step runs on the scalar ALU
__device__ __forceinline__ uint32_t wave_umax_bits(float nonneg)
{
    const uint32_t v = row16_umax(__float_as_uint(nonneg));
    const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

// the same maximum with the cross-row step on the DPP path too (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 holds
// the maximum of the wave): 6 v_max_u32_dpp + 1 v_readlane instead of 4 + 4 v_readlane + the moves back into vector registers -- for the
// kernels that are bound by their vector-instruction count (k_hilb_split16)
__device__ __forceinline__ uint32_t wave_umax_bits_dpp(float nonneg)
{
    uint32_t v = row16_umax(__float_as_uint(nonneg));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// sample loaders: f32 slot or q15 slot (int16 interleaved I/Q, dsp_if.c:286-289)
__device__ __forceinline__ float2 load_iq(const float *p, size_t i)
{
    return reinterpret_cast<const float2 *>(p)[i];
}
__device__ __forceinline__ float2 load_iq(const int16_t *p, size_t i)
{
    short2 v = reinterpret_cast<const short2 *>(p)[i];
    return make_float2(q15_to_float(v.x), q15_to_float(v.y));
}
__device__ __forceinline__ void store_audio(float *p, size_t i, float v, uint32_t = 0u) { p[i] = v; }
__device__ __forceinline__ void store_audio(int16_t *p, size_t i, float v, uint32_t round = 0u) { p[i] = float_to_q15(v, round); }

}  // namespace srx
