// rx_cw.hip -- fused CW narrow-filter kernel (BASELINE cfg4): NCO/BFO mix -> real part ->
// arm_biquad_cascade_df1_f32 (NS stages) -> AGC, one launch, gfx950.
//
// The DF1 recurrence (arm_biquad_cascade_df1_f32.c:220) has no time parallelism without changing
// the rounding, so parallelism is channels x stages.  A wavefront is a SYSTOLIC array:
//
//      lane = 4*channel + stage        (16 channels x 4 stages; NS = 4)
//
// at step k stage s works on sample k-s; a stage's output reaches the next stage's lane with one
// DPP row_shr:1 (lanes of a channel are adjacent and never straddle a 16-lane row).  Value for
// value this is the reference's stage-outer loop: each stage consumes exactly the previous stage's
// output sequence, left-to-right sums, feedback added, no fusion (both arithmetic modes -- the
// recurrence keeps the reference rounding).
//
// Per DSP block (BLK samples) and workgroup (one wavefront, 16 channels):
//   1. coalesced dwordx4 loads (2 complex samples per lane), NCO mix real part, into ONE LDS tile
//      x[16][BLK+4] (f32).  Shared LO: all 8 loads of a chunk use the same LO float4.
//   2. BLK+3 systolic steps (3 masked fill + 3 masked drain steps so a block's envelope is complete
//      before it is scaled); stage 3 writes y[n] over x[n] in place (x[n] was consumed 3 steps ago)
//      and tracks max|y|.
//   3. AGC gain law per channel; the tile goes out as 16 rows of 1 KiB: ds_read_b128, scale by the
//      channel's gain (v_readlane), one global dwordx4 store per lane per row.
// State (4 floats per channel-stage) is one coalesced dwordx4 load/store per lane per call.
#include "rx_internal.h"

#pragma clang fp contract(off)

namespace srx {

__device__ __forceinline__ void cw_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float dpp_row_shr1(float v)
{
    // lane l receives lane l-1 (within its row of 16); lanes 0,16,32,48 keep `v` (they are stage 0
    // lanes and never use the shifted value)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x111, 0xf, 0xf, false));
}

template <typename T> struct CwRaw;
template <> struct CwRaw<float> {
    typedef float4 type;
    static __device__ __forceinline__ type load(const float *src, size_t cplx) { return *reinterpret_cast<const float4 *>(src + 2 * cplx); }
    static __device__ __forceinline__ void unpack(const type &r, float2 &a, float2 &b) { a = make_float2(r.x, r.y); b = make_float2(r.z, r.w); }
};
template <> struct CwRaw<int16_t> {
    typedef short4 type;
    static __device__ __forceinline__ type load(const int16_t *src, size_t cplx) { return *reinterpret_cast<const short4 *>(src + 2 * cplx); }
    static __device__ __forceinline__ void unpack(const type &r, float2 &a, float2 &b)
    {
        a = make_float2(q15_to_float(r.x), q15_to_float(r.y));
        b = make_float2(q15_to_float(r.z), q15_to_float(r.w));
    }
};

constexpr int kCwCh = 16;      // channels per wavefront
constexpr int kCwNs = 4;       // biquad stages (lanes per channel)

template <int NCO, int BLK, typename TIn, typename TOut>
__global__ __launch_bounds__(64) void k_cw_fused(RxParams p, const TIn *__restrict__ src, TOut *__restrict__ dst)
{
    constexpr int RS = BLK + 4;                         // tile row stride (floats); rows stay 16 B aligned
    constexpr int NCHUNK = BLK / 64;                    // input chunks of 64 samples x 16 channels
    __shared__ __attribute__((aligned(16))) float tile[kCwCh * RS + 4 * 64];
    __shared__ float tab[NCO == 1 ? 516 : 4];
    const int lane = threadIdx.x;
    const int s = lane & 3, ch = lane >> 2;
    const uint32_t c0 = blockIdx.x * kCwCh;
    // the last workgroup of a channel count that is not a multiple of 16: lanes past the end work on a
    // copy of the last channel (every load index is clamped) and none of their stores is issued
    const bool live = c0 + ch < p.channels;
    const uint32_t c = live ? c0 + ch : p.channels - 1;

    if constexpr (NCO == 1)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    // per-lane stage constants and state
    const float b0 = p.biq_c[5 * s], b1 = p.biq_c[5 * s + 1], b2 = p.biq_c[5 * s + 2];
    const float a1 = p.biq_c[5 * s + 3], a2 = p.biq_c[5 * s + 4];
    float4 st = *reinterpret_cast<const float4 *>(p.biq_state + ((size_t)c * kCwNs + s) * 4);
    float x1 = st.x, x2 = st.y, y1 = st.z, y2 = st.w;
    float gain = p.agc ? p.gain[c] : 1.0f;
    // load-phase geometry: load j of a chunk covers channel 2j + (lane>>5), samples 2*(lane&31), +1
    const int lch = lane >> 5, lsm = 2 * (lane & 31);
    const uint32_t ph_own = NCO ? p.phase[c] : 0u, st_own = NCO ? p.step[c] : 0u;
    // stage-3 lanes write y[4i..4i+3] at tile[ch][4i]; other lanes write a private dummy float4
    float *wbase = (s == 3) ? (tile + ch * RS) : (tile + kCwCh * RS + 4 * lane);
    const int wstride = (s == 3) ? 1 : 0;
    const float *rbase = tile + ch * RS;

    typedef typename CwRaw<TIn>::type raw_t;
    raw_t raw[8];
    float4 lo4 = make_float4(1.0f, 0.0f, 1.0f, 0.0f);
    auto issue_loads = [&](uint32_t n_first) {          // chunk of 64 samples x 16 channels starting at n_first
#pragma unroll
        for (int j = 0; j < 8; ++j)
            raw[j] = CwRaw<TIn>::load(src, (size_t)min(c0 + 2 * j + lch, p.channels - 1) * p.in_stride + n_first + lsm);
        if constexpr (NCO == 2) lo4 = *reinterpret_cast<const float4 *>(p.lo + n_first + lsm);
    };
    auto mix_write = [&](uint32_t n_first, int q) {      // NCO mix (real part) of the loaded chunk into the tile
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float2 a, b;
            CwRaw<TIn>::unpack(raw[j], a, b);
            float xa, xb;
            if constexpr (NCO == 0) {
                xa = a.x; xb = b.x;
            } else {
                float2 la, lb;
                if constexpr (NCO == 2) {
                    la = make_float2(lo4.x, lo4.y); lb = make_float2(lo4.z, lo4.w);
                } else {
                    const uint32_t cj = min(c0 + 2 * j + lch, p.channels - 1);
                    const uint32_t phj = p.phase[cj], stj = p.step[cj];
                    la = nco_lo<0>(tab, phj + (n_first + lsm) * stj);
                    lb = nco_lo<0>(tab, phj + (n_first + lsm + 1) * stj);
                }
                xa = cmul<0>(a, la).x;                        // arm_cmplx_mult_cmplx_f32 real part: ac - bd
                xb = cmul<0>(b, lb).x;
            }
            *reinterpret_cast<float2 *>(tile + (2 * j + lch) * RS + 64 * q + lsm) = make_float2(xa, xb);
        }
    };
    // one DF1 step of this lane's stage; xs = stage-0 input of the step
    auto step = [&](float xs) -> float {
        const float prev = dpp_row_shr1(y1);
        const float xin = (s == 0) ? xs : prev;
        const float p0 = b0 * xin, p1 = b1 * x1, p2 = b2 * x2, p3 = a1 * y1, p4 = a2 * y2;
        float y = p0 + p1;
        y = y + p2;
        y = y + p3;
        y = y + p4;
        x2 = x1; x1 = xin; y2 = y1; y1 = y;
        return y;
    };
    // the same with the state update suppressed where stage s has no sample at step k (fill / drain)
    auto step_masked = [&](float xs, int k) -> float {
        const float ox1 = x1, ox2 = x2, oy1 = y1, oy2 = y2;
        const float y = step(xs);
        const bool valid = (k - s >= 0) && (k - s < BLK);
        x1 = valid ? x1 : ox1; x2 = valid ? x2 : ox2; y1 = valid ? y1 : oy1; y2 = valid ? y2 : oy2;
        return y;
    };

    const uint32_t nblk = p.block_size / BLK;
    issue_loads(0);
    cw_lds_sync();
    for (uint32_t blk = 0; blk < nblk; ++blk) {
        const uint32_t n0 = blk * BLK;
        float m = 0.0f, ycarry = 0.0f;
#pragma unroll 1
        for (int q = 0; q < NCHUNK; ++q) {
            // ---- 1. this chunk's input into the tile; next chunk's HBM loads in flight meanwhile ----
            mix_write(n0 + 64 * q, q);
            {
                const uint32_t nxt = n0 + 64 * (q + 1);               // next chunk (may be the next block's first)
                if (nxt < p.block_size) issue_loads(nxt);
            }
            cw_lds_sync();
            // ---- 2. 16 trips of 4 systolic steps; stage-0 input is one aligned float4 per trip ----
            int i = 16 * q;
            float4 xq = *reinterpret_cast<const float4 *>(rbase + 4 * i);
            if (q == 0) {                                             // block prologue: 3 fill steps + step 3
                float4 xn = *reinterpret_cast<const float4 *>(rbase + 4);
                (void)step_masked(xq.x, 0);
                (void)step_masked(xq.y, 1);
                (void)step_masked(xq.z, 2);
                ycarry = step(xq.w);                                  // y[0] in stage-3 lanes
                m = fmaxf(m, fabsf(ycarry));
                xq = xn;
                i = 1;
            }
#pragma unroll 1
            for (; i < 16 * q + 16; ++i) {
                // prefetch the next trip's input (stays inside this chunk; harmless re-read at the end)
                const int inext = (i + 1 < 16 * q + 16) ? i + 1 : i;
                const float4 xn = *reinterpret_cast<const float4 *>(rbase + 4 * inext);
                const float ya = step(xq.x), yb = step(xq.y), yc = step(xq.z), yd = step(xq.w);
                // outputs y[4i-3 .. 4i]: the aligned group y[4i-4 .. 4i-1] is complete after yc
                *reinterpret_cast<float4 *>(wbase + (4 * i - 4) * wstride) = make_float4(ycarry, ya, yb, yc);
                m = fmaxf(fmaxf(m, fabsf(ya)), fmaxf(fabsf(yb), fmaxf(fabsf(yc), fabsf(yd))));
                ycarry = yd;
                xq = xn;
            }
        }
        // ---- drain: stages 1..3 finish samples BLK-3 .. BLK-1 ----
        {
            const float ya = step_masked(0.0f, BLK), yb = step_masked(0.0f, BLK + 1), yc = step_masked(0.0f, BLK + 2);
            *reinterpret_cast<float4 *>(wbase + (BLK - 4) * wstride) = make_float4(ycarry, ya, yb, yc);
            m = fmaxf(fmaxf(m, fabsf(ya)), fmaxf(fabsf(yb), fabsf(yc)));
        }
        cw_lds_sync();
        // ---- 3. AGC gain law (stage-3 lanes hold max|y| of their channel) and scaled store ----
        if (p.agc) gain = agc_update<0>(p.agcp, gain, m);
#pragma unroll 4
        for (int r = 0; r < kCwCh; ++r) {
            const float g = __shfl(gain, 4 * r + 3, 64);
#pragma unroll
            for (int h = 0; h < (BLK / 4 + 63) / 64; ++h) {
                const int t = 4 * (lane + 64 * h);
                if (t < BLK && c0 + r < p.channels) {
                    float4 v = *reinterpret_cast<const float4 *>(tile + r * RS + t);
                    v.x = v.x * g; v.y = v.y * g; v.z = v.z * g; v.w = v.w * g;
                    const size_t o = (size_t)(c0 + r) * p.out_stride + n0 + t;
                    if constexpr (sizeof(TOut) == 4) {
                        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(dst) + o) = v;
                    } else {
                        short4 q4;
                        q4.x = float_to_q15(v.x); q4.y = float_to_q15(v.y);
                        q4.z = float_to_q15(v.z); q4.w = float_to_q15(v.w);
                        *reinterpret_cast<short4 *>(reinterpret_cast<int16_t *>(dst) + o) = q4;
                    }
                }
            }
        }
        cw_lds_sync();
    }
    if (!live) return;
    *reinterpret_cast<float4 *>(p.biq_state + ((size_t)c * kCwNs + s) * 4) = make_float4(x1, x2, y1, y2);
    if (s == 3) {
        if (p.agc) p.gain[c] = gain;
        if constexpr (NCO != 0) p.phase[c] = ph_own + p.block_size * st_own;
    }
}

bool cw_fused_ok(const selenite_rx_config &g, uint32_t block_size)
{
    return mode_is_cw(g.mode) && g.nd_taps == 0 && g.decim == 1 && g.nh_taps == 0 && g.n_biquad == kCwNs &&
           g.block == 256 && block_size % g.block == 0;
}

template <int NCO, typename TIn, typename TOut>
static hipError_t cw_launch(const RxParams &p, const void *src, void *dst, hipStream_t st)
{
    hipLaunchKernelGGL((k_cw_fused<NCO, 256, TIn, TOut>), dim3((p.channels + kCwCh - 1) / kCwCh), dim3(64), 0, st, p,
                       static_cast<const TIn *>(src), static_cast<TOut *>(dst));
    return hipGetLastError();
}

hipError_t launch_cw_fused(const RxParams &p, const void *src, bool src_q15, void *dst, bool dst_q15, hipStream_t st)
{
    if (src_q15 != dst_q15) return hipErrorNotSupported;
    if (src_q15) {
        if (p.nco == 2) return cw_launch<2, int16_t, int16_t>(p, src, dst, st);
        if (p.nco == 1) return cw_launch<1, int16_t, int16_t>(p, src, dst, st);
        return cw_launch<0, int16_t, int16_t>(p, src, dst, st);
    }
    if (p.nco == 2) return cw_launch<2, float, float>(p, src, dst, st);
    if (p.nco == 1) return cw_launch<1, float, float>(p, src, dst, st);
    return cw_launch<0, float, float>(p, src, dst, st);
}

}  // namespace srx
