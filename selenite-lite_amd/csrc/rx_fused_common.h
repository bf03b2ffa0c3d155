// rx_fused_common.h -- geometry, argument block and device helpers shared by the fused SSB kernels
// (rx_fused.hip: k_ssb_fused / k_ssb_mfma; rx_split16.hip: the SELENITE_ARITH_SPLIT16 kernels).
#pragma once
#include "rx_internal.h"

#pragma clang fp contract(off)

namespace srx {

typedef float v2f __attribute__((ext_vector_type(2)));

template <int ND, int M, int NH>
struct Geo {
    static constexpr int P = 256;                                   // decimated outputs per pass
    static constexpr int T = P * M;                                 // complex inputs per pass
    static constexpr int HQ = ND ? (ND - 1 + M - 1) / M : 0;        // decimator history, phase-samples
    static constexpr int HQ4 = (HQ + 3) & ~3;
    static constexpr int F = ND ? (HQ4 * M + 1 - ND) : 0;           // leading zero-pad taps
    static constexpr int NCQ = HQ4 * M + 1;                         // padded taps cq[0 .. HQ4*M]
    static constexpr int NCR = (NCQ + 63) / 64;                     // coefficient VGPRs per lane
    static constexpr int PLEN = HQ4 + P;                            // complex elements per phase
    // Polyphase image: per phase p an array of (I,Q) float2 elements; each group of 4 elements
    // (the 4 outputs one lane owns) occupies THREE 16-byte slots (48 B, last slot unused), so the
    // ds_read_b128 of lane l at compile-time offset o is at 48*l + imm: lane stride 3 slots is
    // conflict-free for every b128 lane group and needs no per-read address arithmetic.
    static constexpr int PSF = 12 * (PLEN / 4);                     // floats per phase array
    static constexpr int HH = NH ? NH - 1 : 0;                      // Hilbert history
    static constexpr int HH4 = (HH + 3) & ~3;
    static constexpr int FH = HH4 - HH;                             // leading pad of the FIR window
    static constexpr int DLEN = HH4 + P + 4;
    static constexpr int oTab = 0;
    static constexpr int oS = 516;                                  // [M][PSF]          (ND > 0)
    static constexpr int oD = oS + (ND ? M * PSF : 0);              // [2 rails][DLEN]
    static constexpr int total = oD + 2 * DLEN;
    __host__ __device__ static constexpr int elem(int idx) { return 12 * (idx >> 2) + 2 * (idx & 3); }
};

struct FusedArgs {
    const float *cq;        // padded decimator taps: cq[k'] = dec[k' - F] (k' >= F), else 0; 64*NCR floats
    uint32_t delay_idx;     // index of the unit tap in delay_coeffs
    uint32_t upper;         // 1: audio = I' - Q'   0: audio = I' + Q'
    uint32_t am;            // 1: audio = |I + jQ| (arm_cmplx_mag_f32); the Hilbert pair and its state are untouched
                            // 2: FM -- angle of z[n] conj(z[n-1]) in half turns; the pair's delay lines keep running (k_ssb_fused only)
    uint32_t nco;           // k_ssb_fused: NCO flavour of the launch (0 off, 1 per channel per sample, 2 shared table, 4 per-channel periodic LO
                            // in registers) -- a run-time switch since round 4 (launch_one sets it)
    const float *ptab;      // DENSE instantiations of k_ssb_fused: the FIR pair's taps, [2][DenseTab::LEN] (delay rail, Hilbert rail), padded tap k at index k + FH + 3
    float *ptab_lds;        // ... their copy in LDS (set by the kernel)
    uint32_t dense_t0;      // ... first step with a tap that is not padding
    uint32_t group;         // lanes per DSP block = (block / M) / 4
    uint32_t pass_out;      // k_ssb_fused: audio samples a full pass produces = the largest whole number of DSP blocks in 256 (256 when
                            // block / M divides 256; 240 for the firmware's 96-frame blocks by 4, 192 for 96 frames without decimator)
    uint32_t grp_shift;     // k_ssb_mfma: phase group of a wave = (wave >> grp_shift) & 1
    const void *btab16;     // k_ssb_split16: Toeplitz operand, f16 hi/lo fragments
    float split_post;       // k_hilb_split16: exact power-of-two rescale of the MFMA result
    int split_sc;           // k_ssb_split16: the taps were scaled by 2^split_sc before their f16 hi/lo split
    uint32_t inl;           // k_ssb_split16 / k_hilb_split16, SELENITE_ARITH_AUTO: 1 = the workgroup recomputes its flagged and held channels itself, behind
                            // its last channel (one launch per call); 0 = it only raises their flags (k_hist_exact and the rerun pass follow)
    uint32_t dec2;          // k_ssb_split16 (run-time-geometry instantiations): decimation by 2 M on the by-M Toeplitz product -- a pass is
                            // pass_out * 2 M input samples and only every second output of the tile is an output of the chain (1: the even ones,
                            // 2: the odd ones); 0: the plain by-M kernel
};

// Workgroups of these kernels are ONE wavefront: LDS instructions of a wave execute in issue order,
// so a store is visible to any lane's later load without s_barrier.  What is needed is only that
// the compiler keeps the program order of LDS accesses: a wavefront-scope fence (emits nothing)
// plus the wave_barrier scheduling fence.  __syncthreads() would add "s_waitcnt vmcnt(0)", which
// drains the next pass's HBM prefetch and stalls the wave for a full memory latency per pass.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Prologue fill of N LDS words from HBM state.  `index(i)` is the element of `base` that slot i takes,
// negative for slots in front of the state (they are zero).  Loads are unconditional from a clamped
// index and masked afterwards: a conditional load becomes an exec-masked branch with its own
// s_waitcnt vmcnt(0), and the N/64 round trips of a wavefront then queue behind one another
// (measured: 8 serialized trips = 25 % of a workgroup's lifetime).  All loads are issued before the
// first store.
template <int N, typename IndexFn, typename StoreFn>
__device__ __forceinline__ void batched_fill(int lane, const float *__restrict__ base, IndexFn index, StoreFn store)
{
    constexpr int NI = (N + 63) / 64;
    float v[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int i = (N % 64 == 0) ? j * 64 + lane : min(j * 64 + lane, N - 1);
        const int e = index(i);
        const float x = base[e < 0 ? 0 : e];
        v[j] = e < 0 ? 0.0f : x;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int i = j * 64 + lane;
        if (N % 64 == 0 || i < N) store(i, v[j]);
    }
}

// the two halves of batched_fill on their own: the loads of a channel's state can then be issued long before its LDS slots are free
template <int N, typename IndexFn>
__device__ __forceinline__ void batched_load(int lane, const float *__restrict__ base, IndexFn index, float (&v)[(N + 63) / 64])
{
#pragma unroll
    for (int j = 0; j < (N + 63) / 64; ++j) {
        const int i = (N % 64 == 0) ? j * 64 + lane : min(j * 64 + lane, N - 1);
        const int e = index(i);
        const float x = base[e < 0 ? 0 : e];
        v[j] = e < 0 ? 0.0f : x;
    }
}
template <int N, typename StoreFn>
__device__ __forceinline__ void batched_store(int lane, const float (&v)[(N + 63) / 64], StoreFn store)
{
#pragma unroll
    for (int j = 0; j < (N + 63) / 64; ++j) {
        const int i = j * 64 + lane;
        if (N % 64 == 0 || i < N) store(i, v[j]);
    }
}

__device__ __forceinline__ float f4get(const float4 &v, int e)
{
    return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
}

// (I,Q) += (I,Q) * c  -- one v_pk_fma_f32, or v_pk_mul_f32 + v_pk_add_f32 in the CMSIS arithmetic
template <int ARITH>
__device__ __forceinline__ v2f mac2(v2f acc, v2f w, float c)
{
    const v2f c2 = { c, c };
    if constexpr (ARITH == 1) {
        return __builtin_elementwise_fma(w, c2, acc);
    } else {
        const v2f p = w * c2;
        return acc + p;
    }
}

// arm_cmplx_mult_cmplx_f32 on one (re, im) register pair: (a*c - b*d, a*d + b*c) with the four
// products and the two sums rounded separately (ComplexMathFunctions/arm_cmplx_mult_cmplx_f32.c:186-187).
// Three packed instructions: v_pk_mul_f32 x2 (operand halves picked by op_sel) + v_pk_add_f32.
__device__ __forceinline__ v2f cmul_pk(v2f A, v2f L)
{
    // The compiler does not fold the half swaps into op_sel (it emits v_mov/v_xor pairs), hence asm:
    //   t1 = (a*c, a*d)   t2 = (b*d, b*c)   r = (t1.lo - t2.lo, t1.hi + t2.hi)
    // s_nop: packed-f32 results need one wait state before a non-packed consumer on gfx950 (the
    // compiler inserts the same s_nop in its own code; it cannot see into the asm block).
    v2f t1, t2, r;
    asm("v_pk_mul_f32 %0, %3, %4 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %3, %4 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "s_nop 0\n\t"
        "v_pk_add_f32 %2, %0, %1 neg_lo:[0,1]\n\t"
        "s_nop 0"
        : "=&v"(t1), "=&v"(t2), "=v"(r)
        : "v"(A), "v"(L));
    return r;
}

// two complex multiplies in one block: the dependent v_pk_add_f32 of each sits three instructions behind
// its v_pk_mul_f32 pair, so only the block's last result needs the wait state before a non-packed consumer
__device__ __forceinline__ void cmul_pk2(v2f A, v2f B, v2f LA, v2f LB, v2f &ra, v2f &rb)
{
    v2f t1, t2, t3, t4;
    asm("v_pk_mul_f32 %0, %6, %8 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %6, %8 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %2, %7, %9 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %7, %9 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %4, %0, %1 neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %5, %2, %3 neg_lo:[0,1]\n\t"
        "s_nop 0"
        : "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(ra), "=&v"(rb)
        : "v"(A), "v"(B), "v"(LA), "v"(LB));
}

constexpr int SRX_IN_AUX = 2;    // cache policy of the streamed input loads (2 = nt)
constexpr int SRX_OUT_AUX = 2;   // cache policy of the audio stores: 2 = nt.  Written once, never read by the chain: streaming them past the caches
                                 // leaves the 256 MB Infinity Cache to the per-channel state, which IS read back by the next call (measured: -1.6 % time;
                                 // the same policy on the state loads / stores costs +1.6 %: profiles/r3/README.md)
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));

// Raw buffer resource over a wave-uniform byte range (cdna_hip_programming.md T8): loads beyond the
// range return 0 WITHOUT memory traffic and stores beyond it are dropped, so "prefetch the next pass"
// needs no branch in the last pass -- the scalar offset is simply pushed out of range.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// two complex samples per lane per load (sample n = 128 i + 2 lane of a pass)
template <typename T> struct BRaw;
template <> struct BRaw<float> {
    typedef u4v type;
    static constexpr int kBytes = 16;
    static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, SRX_IN_AUX);      // nt: streamed once
    }
    static __device__ __forceinline__ void unpack(const type &r, v2f &a, v2f &b)
    {
        a = v2f{ __uint_as_float(r.x), __uint_as_float(r.y) };
        b = v2f{ __uint_as_float(r.z), __uint_as_float(r.w) };
    }
};
template <> struct BRaw<int16_t> {
    typedef u2v type;
    static constexpr int kBytes = 8;
    static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, SRX_IN_AUX);
    }
    static __device__ __forceinline__ void unpack(const type &r, v2f &a, v2f &b)
    {
        a = v2f{ q15_to_float((int16_t)(r.x & 0xffffu)), q15_to_float((int16_t)(r.x >> 16)) };
        b = v2f{ q15_to_float((int16_t)(r.y & 0xffffu)), q15_to_float((int16_t)(r.y >> 16)) };
    }
};

// four audio samples per lane per store
template <typename T> struct BOut;
template <> struct BOut<float> {
    static constexpr int kBytes = 16;
    // cached: the audio will be read back (phase 1 of the global-gain call: the gain pass reads it) -- default policy instead of nt
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, int voff, int soff, const float (&au)[4], bool cached = false, uint32_t = 0u)
    {
        const u4v v = { __float_as_uint(au[0]), __float_as_uint(au[1]), __float_as_uint(au[2]), __float_as_uint(au[3]) };
        if (cached) __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, SRX_OUT_AUX);
    }
};
template <> struct BOut<int16_t> {
    static constexpr int kBytes = 8;
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, int voff, int soff, const float (&au)[4], bool = false, uint32_t round = 0u)
    {
        uint32_t w0, w1;
        float4_to_q15(au[0], au[1], au[2], au[3], round, w0, w1);
        const u2v v = { w0, w1 };
        __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, SRX_OUT_AUX);
    }
};

// raw global loads: two complex samples per lane per instruction
template <typename TIn> struct Raw;
template <> struct Raw<float> {
    typedef float4 type;
    static __device__ __forceinline__ type load(const float *src, size_t cplx_index)
    {
        // streamed once: non-temporal, so the shared LO / coefficient tables keep their cache lines
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(src + 2 * cplx_index));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    static __device__ __forceinline__ void unpack(const type &r, float2 &a, float2 &b)
    {
        a = make_float2(r.x, r.y); b = make_float2(r.z, r.w);
    }
};
template <> struct Raw<int16_t> {
    typedef short4 type;
    static __device__ __forceinline__ type load(const int16_t *src, size_t cplx_index)
    {
        return *reinterpret_cast<const short4 *>(src + 2 * cplx_index);
    }
    static __device__ __forceinline__ void unpack(const type &r, float2 &a, float2 &b)
    {
        a = make_float2(q15_to_float(r.x), q15_to_float(r.y));
        b = make_float2(q15_to_float(r.z), q15_to_float(r.w));
    }
};

// arm_fir_f32 with type-III Hilbert taps for 4 adjacent outputs n = 4*lane + r.
// dq: decimated Q rail, new samples start at HH4.  y[n] = sum_k h[k] * dq[n + k + FH].
// The taps live lane-distributed in hreg (lane k of hreg[k>>6] = h[k]), loaded ONCE per kernel and
// fetched by v_readlane: reading them from memory inside the pass loop costs an L2 round trip per
// pass (the compiler cannot hoist the loads above the audio stores it must assume may alias).
// one read-and-accumulate step t of the Hilbert FIR; tap(k) yields h[k] as a wave-uniform value
template <int ARITH, int ND, int M, int NH, typename TapFn>
__device__ __forceinline__ void hilbert_tstep(int t, const float *dq, int lane, TapFn tap, float (&acc)[4])
{
    using G = Geo<ND, M, NH>;
    constexpr int C = (NH - 1) / 2;
    const v4f W = lds_ld4(dq + 4 * lane + 4 * t);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float w = W[e];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * t + e - r - G::FH;
            if (k < 0 || k >= NH || (((k - C) & 1) == 0)) continue;   // structural zeros
            acc[r] = mac<ARITH>(acc[r], w, tap(k));
        }
    }
}
template <int ND, int M, int NH>
struct HilbertSteps { static constexpr int N = (Geo<ND, M, NH>::HH4 + 3) / 4 + 1; };   // t = 0 .. N-1

template <int ARITH, int ND, int M, int NH>
__device__ __forceinline__ void hilbert_quad(const float *dq, int lane, const float (&hreg)[(NH + 63) / 64 ? (NH + 63) / 64 : 1],
                                             float (&acc)[4])
{
#pragma unroll
    for (int t = 0; t < HilbertSteps<ND, M, NH>::N; ++t)
        hilbert_tstep<ARITH, ND, M, NH>(t, dq, lane, [&](int k) {
            return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hreg[k >> 6]), k & 63)); }, acc);
}


constexpr int SRX_IMG_ALIGN = 128;     // halfs
// ---- SELENITE_ARITH_SPLIT16 geometry (rx_split16.hip; the host-side table builder in rx_fused.hip
// needs KS and the fragment order) ----
template <int NCO, int ND, int M, int NH>
struct GeoS {
    using G = Geo<ND, M, NH>;
    static constexpr int HS = G::HQ4 * M;                 // history samples in front
    static constexpr int XN = HS + G::T;
    // physical image rows: the 16 M samples one row of the 16 x 16 output tile advances by (64 for /4, 32 for /2), padded by a
    // quarter (80 / 40 halfs): the A-fragment ds_read_b128 of lane l sits at RSTR (l&15) + 8 (l>>4) + phys(32 kk), conflict free,
    // and a fragment never straddles a row (8 (l>>4) + 32 kk mod RL + 7 < RL)
    static constexpr int RL = 16 * M;
    static constexpr int RSTR = RL + RL / 4;
    static constexpr int XROWS = XN / RL;
    static constexpr int IMG = (RSTR * XROWS + SRX_IMG_ALIGN - 1) / SRX_IMG_ALIGN * SRX_IMG_ALIGN;   // halfs per image (stride a multiple of 256 B: two images' words leave in one ds_write2st64_b32)
    static constexpr int KTOT = G::HQ4 * M + M * 15 + 1;  // padded taps 0 .. HQ4*M  +  shift of 15 outputs
    static constexpr int KS = (KTOT + 31) / 32;           // MFMA k-steps of 32
    static constexpr int oTab = 0;                        // floats; the sine table only when the LO is computed in the kernel
    static constexpr int oX = (NCO == 1 || NCO == 4) ? 516 : 0;   // 4 images of IMG halfs = 2*IMG floats
    static constexpr int oHf = oX + 2 * IMG;              // f32 copy of the HS history samples, (I, Q) pairs
    static constexpr int oD = oHf + 2 * HS;
    static constexpr int total = oD + 2 * G::DLEN;
    __host__ __device__ static constexpr int phys(int f) { return RSTR * (f / RL) + (f % RL); }
};
template <int NH>
struct GeoH {
    static constexpr int HH = NH - 1;                              // history samples (even)
    static constexpr int KS = (NH + 15 + 31) / 32;                 // MFMA k-steps of 32
    static constexpr int XN = 240 + 32 * KS;                       // highest image index read + 1
    // image layout: LINEAR.  The A fragment of lane l is the 16 bytes at slot 2 (l&15) + (l>>4) + 4 kk (16-byte slots), and ds_read_b128 is served in
    // the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): within each the sixteen slots are distinct mod 16
    // -- conflict free as it stands.  (Until round 5 every 128 samples were padded by 16 bytes, which is what a grouping by contiguous sixteen
    // lanes would ask for: with the real groups it made every fragment read a 2-way conflict, 42 % of the kernel's LDS cycles.)
    __host__ __device__ static constexpr int phys(int u) { return u; }
    static constexpr int IMG = (XN + 8 + 7) & ~7;                  // halfs per image
    static constexpr int DIL = HH + 256;                           // f32 I rail: [history | new]
    static constexpr int oTab = 0, oX = 516, oDI = oX + IMG /* 2 images of IMG halfs */, oDQ = oDI + DIL + 2, oO = oDQ + DIL + 2, total = oO + 256;
    static_assert(HH % 2 == 0 && HH <= 256, "Hilbert history");
};

// instantiated shapes of the two kernel families (ND, M, NH) / (NH)
#define SRX_SPLIT16_SHAPES(X) X(256, 4, 63) X(128, 4, 63) X(256, 4, 127) X(128, 4, 127) X(256, 4, 31) X(128, 4, 31) \
                              X(256, 2, 63) X(128, 2, 63) X(256, 2, 127) X(128, 2, 127) X(256, 2, 31) X(128, 2, 31)
#define SRX_HILB16_SHAPES(X) X(63) X(127) X(31)
hipError_t launch_ssb_split16(int nd, int m, int nh, const RxParams &p, const FusedArgs &fa, const void *src, bool q15,
                              void *dst, hipStream_t st);
hipError_t launch_hilb_split16(int nh, const RxParams &p, const FusedArgs &fa, const void *src, bool q15, void *dst,
                               hipStream_t st);

}  // namespace srx
