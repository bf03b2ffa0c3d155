// rx_cw.hip -- fused CW narrow-filter kernel (BASELINE cfg4): NCO/BFO mix -> real part ->
// arm_biquad_cascade_df1_f32 (NS stages) -> AGC, one launch, gfx950.
//
// The DF1 recurrence (arm_biquad_cascade_df1_f32.c:220) has no time parallelism without changing
// the rounding, so parallelism is channels x stages.  A wavefront is a SYSTOLIC array:
//
//      lane = NS*channel + stage       (NS = 2, 4 or 8 stages: 32, 16 or 8 channels per wavefront)
//
// at step k stage s works on sample k-s; a stage's output reaches the next stage's lane with one
// DPP row_shr:1 (lanes of a channel are adjacent and never straddle a 16-lane row: NS divides 16).  Value for
// value this is the reference's stage-outer loop: each stage consumes exactly the previous stage's
// output sequence, left-to-right sums, feedback added, no fusion (both arithmetic modes -- the
// recurrence keeps the reference rounding).
//
// Per DSP block (BLK samples) and workgroup (one wavefront, 16 channels):
//   1. coalesced dwordx4 loads (2 complex samples per lane), NCO mix real part, into ONE LDS tile
//      x[16][BLK+4] (f32).  Shared LO: all 8 loads of a chunk use the same LO float4.
//   2. BLK+NS-1 systolic steps (NS-1 masked fill + NS-1 masked drain steps so a block's envelope is complete
//      before it is scaled); the last stage writes y[n] over x[n] in place (x[n] was consumed NS-1 steps ago)
//      and tracks max|y|.
//   3. AGC gain law per channel; the tile goes out as 16 rows of 1 KiB: ds_read_b128, scale by the
//      channel's gain (v_readlane), one global dwordx4 store per lane per row.
// State (4 floats per channel-stage) is one coalesced dwordx4 load/store per lane per call.
#include "rx_internal.h"
#include <cstdlib>

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
    // lane l receives lane l-1 (within its row of 16); lanes 0,16,32,48 read 0 (bound_ctrl: they are stage 0
    // lanes and never use the shifted value) -- no `old` operand, so no register copy in front of the DPP move
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
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

// Geometry of the systolic array for NS stages per channel (NS = 2, 4, 8: lanes of a channel are adjacent and never
// straddle a 16-lane DPP row): CH = 64 / NS channels per wavefront, input chunks of CS = 1024 / CH samples (eight
// dwordx4 loads of two complex samples per lane and chunk), D = NS - 1 fill / drain steps per DSP block.  Stage NS-1
// emits y[k - D] at step k; trips are four steps, so an aligned float4 of outputs is NCAR values carried from the
// trip before plus the first 4 - NCAR of this one (NCAR = 4 - D mod 4), PRO = D / 4 + 1 trips behind the input.
template <int NS>
struct CwGeo {
    static_assert(NS == 2 || NS == 4 || NS == 8, "systolic CW kernel: 2, 4 or 8 biquad stages");
    static constexpr int CH = 64 / NS;
    static constexpr int CS = 1024 / CH;
    static constexpr int LPC = CS / 2;              // lanes per channel row of a load
    static constexpr int CPL = 64 / LPC;            // channels per load
    static constexpr int D = NS - 1;
    static constexpr int R = D % 4;                 // outputs of a trip that complete the pending group
    static constexpr int NCAR = 4 - R;
    static constexpr int PRO = D / 4 + 1;
};

template <int NS, int NCO, int BLK, typename TIn, typename TOut>
__global__ __launch_bounds__(64) void k_cw_fused(RxParams p, const TIn *__restrict__ src, TOut *__restrict__ dst)
{
    using CG = CwGeo<NS>;
    constexpr int CH = CG::CH, CS = CG::CS, D = CG::D, R = CG::R, NCAR = CG::NCAR, PRO = CG::PRO;
    constexpr int RS = BLK + 4;                         // tile row stride (floats); rows stay 16 B aligned
    constexpr int NCHUNK = BLK / CS;                    // input chunks of CS samples x CH channels
    constexpr int TPC = CS / 4;                         // trips per chunk
    static_assert(BLK % CS == 0 && PRO <= TPC && 4 * PRO <= BLK, "block holds whole chunks; prologue inside the first chunk");
    __shared__ __attribute__((aligned(16))) float tile[CH * RS + 4 * 64];
    __shared__ float tab[NCO == 1 ? 516 : 4];
    const int lane = threadIdx.x;
    const int s = lane & (NS - 1), ch = lane / NS;
    const uint32_t c0 = blockIdx.x * CH;
    // the last workgroup of a channel count that is not a multiple of CH: lanes past the end work on a
    // copy of the last channel (every load index is clamped) and none of their stores is issued
    const bool live = c0 + ch < p.channels;
    const uint32_t c = live ? c0 + ch : p.channels - 1;

    if constexpr (NCO == 1)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    // per-lane stage constants and state
    const float b0 = p.biq_c[5 * s], b1 = p.biq_c[5 * s + 1], b2 = p.biq_c[5 * s + 2];
    const float a1 = p.biq_c[5 * s + 3], a2 = p.biq_c[5 * s + 4];
    float4 st = *reinterpret_cast<const float4 *>(p.biq_state + ((size_t)c * NS + s) * 4);
    float x1 = st.x, x2 = st.y, y1 = st.z, y2 = st.w;
    float gain = p.agc ? p.gain[c] : 1.0f;
    bool nonfinite = false;                                           // any audio sample of this wavefront NaN / Inf
    // load-phase geometry: load j of a chunk covers channel CPL*j + lane/LPC, samples 2*(lane%LPC), +1
    const int lch = lane / CG::LPC, lsm = 2 * (lane % CG::LPC);
    const uint32_t ph_own = NCO ? p.phase[c] : 0u, st_own = NCO ? p.step[c] : 0u;
    // last-stage lanes write y[4i..4i+3] at tile[ch][4i]; other lanes write a private dummy float4
    float *wbase = (s == NS - 1) ? (tile + ch * RS) : (tile + CH * RS + 4 * lane);
    const int wstride = (s == NS - 1) ? 1 : 0;
    const float *rbase = tile + ch * RS;

    typedef typename CwRaw<TIn>::type raw_t;
    raw_t raw[8];
    float4 lo4 = make_float4(1.0f, 0.0f, 1.0f, 0.0f);
    auto issue_loads = [&](uint32_t n_first) {          // chunk of CS samples x CH channels starting at n_first
#pragma unroll
        for (int j = 0; j < 8; ++j)
            raw[j] = CwRaw<TIn>::load(src, (size_t)min(c0 + CG::CPL * j + lch, p.channels - 1) * p.in_stride + n_first + lsm);
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
                    const uint32_t cj = min(c0 + CG::CPL * j + lch, p.channels - 1);
                    const uint32_t phj = p.phase[cj], stj = p.step[cj];
                    lo_v2f va, vb;                                    // arm_sin/cos_f32 restated for the vector ALU: same bits (rx_device.h)
                    const uint32_t pe = phj + (n_first + lsm) * stj;
                    nco_lo_pair(tab, pe, pe + stj, va, vb);
                    la = make_float2(va.x, va.y); lb = make_float2(vb.x, vb.y);
                }
                xa = cmul<0>(a, la).x;                        // arm_cmplx_mult_cmplx_f32 real part: ac - bd
                xb = cmul<0>(b, lb).x;
            }
            *reinterpret_cast<float2 *>(tile + (CG::CPL * j + lch) * RS + CS * q + lsm) = make_float2(xa, xb);
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
    // four steps of the steady state.  Each delay line is ONE register pair whose halves swap roles every step: on
    // an even step (x1, x2) = (lo, hi) and the new input overwrites hi (x2 is dead by then), on an odd step
    // (x1, x2) = (hi, lo) and the packed multiply reads its first operand's halves swapped (op_sel).  So the packed
    // products (b1 x1, b2 x2) and (a1 y1, a2 y2) never need a register copy: 9 vector instructions per step (DPP
    // move, select, 1 + 2 multiplies, 4 adds).  Same products, same left-to-right sum as step().
    typedef float cw_v2f __attribute__((ext_vector_type(2)));
    const cw_v2f b12 = { b1, b2 }, a12 = { a1, a2 };
    auto trip4 = [&](const float4 &xq, float (&o)[4]) {
        cw_v2f X = { x1, x2 }, Y = { y1, y2 };
        const float xs[4] = { xq.x, xq.y, xq.z, xq.w };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float prev = dpp_row_shr1((j & 1) ? Y.y : Y.x);
            const float xin = (s == 0) ? xs[j] : prev;
            cw_v2f px, py;
            if (j & 1) {
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(px) : "v"(X), "v"(b12));
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(py) : "v"(Y), "v"(a12));
            } else {
                px = X * b12;
                py = Y * a12;
            }
            const float p0 = b0 * xin;
            float y = p0 + px.x;
            y = y + px.y;
            y = y + py.x;
            y = y + py.y;
            if (j & 1) { X.x = xin; Y.x = y; } else { X.y = xin; Y.y = y; }
            o[j] = y;
        }
        x1 = X.x; x2 = X.y; y1 = Y.x; y2 = Y.y;
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
        float m = 0.0f;
        float car[NCAR];                                              // outputs waiting for the rest of their float4
#pragma unroll
        for (int v = 0; v < NCAR; ++v) car[v] = 0.0f;
#pragma unroll 1
        for (int q = 0; q < NCHUNK; ++q) {
            // ---- 1. this chunk's input into the tile; next chunk's HBM loads in flight meanwhile ----
            mix_write(n0 + CS * q, q);
            {
                const uint32_t nxt = n0 + CS * (q + 1);               // next chunk (may be the next block's first)
                if (nxt < p.block_size) issue_loads(nxt);
            }
            cw_lds_sync();
            // ---- 2. TPC trips of 4 systolic steps; stage-0 input is one aligned float4 per trip ----
            int i = TPC * q;
            float4 xq = lds_ld4f(rbase + 4 * i);
            if (q == 0) {                                             // block prologue: the D fill steps, first outputs into `car`
#pragma unroll
                for (int tpro = 0; tpro < PRO; ++tpro) {
                    const float4 xn = lds_ld4f(rbase + 4 * (tpro + 1));
                    const float o[4] = { step_masked(xq.x, 4 * tpro), step_masked(xq.y, 4 * tpro + 1),
                                         step_masked(xq.z, 4 * tpro + 2), step_masked(xq.w, 4 * tpro + 3) };
                    if (tpro == PRO - 1) {
#pragma unroll
                        for (int j = R; j < 4; ++j) {                 // y[0 .. NCAR-1] (last-stage lanes)
                            car[j - R] = o[j];
                            m = fmaxf(m, fabsf(o[j]));
                        }
                    }
                    xq = xn;
                }
                i = PRO;
            }
#pragma unroll 1
            for (; i < TPC * q + TPC; ++i) {
                // prefetch the next trip's input (stays inside this chunk; harmless re-read at the end)
                const int inext = (i + 1 < TPC * q + TPC) ? i + 1 : i;
                const float4 xn = lds_ld4f(rbase + 4 * inext);
                float o[4];
                trip4(xq, o);
                // the aligned group y[4(i-PRO) .. +3]: NCAR carried outputs, then the first R of this trip
                float gq[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) gq[v] = v < NCAR ? car[v] : o[v - NCAR];
                *reinterpret_cast<float4 *>(wbase + 4 * (i - PRO) * wstride) = make_float4(gq[0], gq[1], gq[2], gq[3]);
                m = fmaxf(fmaxf(m, fabsf(o[0])), fmaxf(fabsf(o[1]), fmaxf(fabsf(o[2]), fabsf(o[3]))));
#pragma unroll
                for (int v = 0; v < NCAR; ++v) car[v] = o[R + v];
                xq = xn;
            }
        }
        // ---- drain: stages 1..NS-1 finish samples BLK-D .. BLK-1 ----
        {
            float seq[4 * PRO];
#pragma unroll
            for (int v = 0; v < NCAR; ++v) seq[v] = car[v];
#pragma unroll
            for (int w = 0; w < D; ++w) {
                seq[NCAR + w] = step_masked(0.0f, BLK + w);
                m = fmaxf(m, fabsf(seq[NCAR + w]));
            }
#pragma unroll
            for (int tpro = 0; tpro < PRO; ++tpro)
                *reinterpret_cast<float4 *>(wbase + (BLK - 4 * PRO + 4 * tpro) * wstride) =
                    make_float4(seq[4 * tpro], seq[4 * tpro + 1], seq[4 * tpro + 2], seq[4 * tpro + 3]);
        }
        cw_lds_sync();
        // ---- 3. AGC gain law (last-stage lanes hold max|y| of their channel) and scaled store ----
        if (p.agc) gain = agc_update<0>(p.agcp, gain, m);
#pragma unroll 4
        for (int r = 0; r < CH; ++r) {
            const float g = __shfl(gain, NS * r + NS - 1, 64);
#pragma unroll
            for (int h = 0; h < (BLK / 4 + 63) / 64; ++h) {
                const int t = 4 * (lane + 64 * h);
                if (t < BLK && c0 + r < p.channels) {
                    float4 v = lds_ld4f(tile + r * RS + t);
                    v.x = v.x * g; v.y = v.y * g; v.z = v.z * g; v.w = v.w * g;
                    {   // ARM_MATH_NANINF (arm_math.h:405): x * 0 is NaN iff x is not finite
                        const float z = __builtin_fmaf(v.w, 0.0f, __builtin_fmaf(v.z, 0.0f, __builtin_fmaf(v.y, 0.0f, v.x * 0.0f)));
                        nonfinite = nonfinite || (z != z);
                    }
                    const size_t o = (size_t)(c0 + r) * p.out_stride + n0 + t;
                    // (non-temporal: written once, never read by the chain)
                    if constexpr (sizeof(TOut) == 4) {
                        v4f *at = reinterpret_cast<v4f *>(reinterpret_cast<float *>(dst) + o);
                        if (p.out_cached) *at = v4f{ v.x, v.y, v.z, v.w };       // global gain, phase 1: the gain pass reads it back
                        else __builtin_nontemporal_store(v4f{ v.x, v.y, v.z, v.w }, at);
                    } else {
                        typedef short s4v __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(s4v{ float_to_q15(v.x), float_to_q15(v.y), float_to_q15(v.z), float_to_q15(v.w) },
                                                    reinterpret_cast<s4v *>(reinterpret_cast<int16_t *>(dst) + o));
                    }
                }
            }
        }
        cw_lds_sync();
    }
    if (nonfinite) p.flags[kFlagNanInf] = 1u;                         // read by selenite_rx_sync / the host-pointer calls
    if (!live) return;
    *reinterpret_cast<float4 *>(p.biq_state + ((size_t)c * NS + s) * 4) = make_float4(x1, x2, y1, y2);
    if (s == NS - 1) {
        if (p.agc) p.gain[c] = gain;
        if constexpr (NCO != 0) p.phase[c] = ph_own + p.block_size * st_own;
    }
}

// DSP blocks the systolic kernel is instantiated for: whole input chunks of CS = 1024 / (64 / NS) samples (32 / 64 / 128 for
// 2 / 4 / 8 stages) and a tile of (64 / NS) x (BLK + 4) floats inside the 64 KB of static LDS -- 128, 192, 256, 512 where that
// holds, and the firmware's 96-frame slot (Core/Inc/dsp_if.h:69-73) for 2 stages
static bool cw_block_ok(uint32_t ns, uint32_t blk)
{
    if (ns == 2) return blk == 96 || blk == 128 || blk == 192 || blk == 256;
    if (ns == 4) return blk == 128 || blk == 192 || blk == 256 || blk == 512;
    if (ns == 8) return blk == 128 || blk == 256 || blk == 512;
    return false;
}

bool cw_fused_ok(const selenite_rx_config &g, uint32_t block_size)
{
    return mode_is_cw(g.mode) && g.nd_taps == 0 && g.decim == 1 && g.nh_taps == 0 && cw_block_ok(g.n_biquad, g.block) &&
           block_size % g.block == 0;
}

template <int NS, int NCO, typename TIn, typename TOut>
static hipError_t cw_launch(const RxParams &p, const void *src, void *dst, hipStream_t st)
{
    constexpr int CH = CwGeo<NS>::CH;
    static const size_t pad = std::getenv("SELENITE_RX_CW_LDS_PAD") ? (size_t)std::atoi(std::getenv("SELENITE_RX_CW_LDS_PAD")) : 0;   // occupancy experiments
    const dim3 grid((p.channels + CH - 1) / CH);
#define CW_BLK(B_)                                                                                                          \
    if (p.block == B_) {                                                                                                    \
        hipLaunchKernelGGL((k_cw_fused<NS, NCO, B_, TIn, TOut>), grid, dim3(64), pad, st, p, static_cast<const TIn *>(src), \
                           static_cast<TOut *>(dst));                                                                      \
        return hipGetLastError();                                                                                           \
    }
    CW_BLK(256)
    CW_BLK(128)
    if constexpr (NS != 8) { CW_BLK(192) }
    if constexpr (NS != 2) { CW_BLK(512) }
    if constexpr (NS == 2) { CW_BLK(96) }
#undef CW_BLK
    return hipErrorNotSupported;
}

template <int NS>
static hipError_t cw_launch_ns(const RxParams &p, const void *src, bool q15, void *dst, hipStream_t st)
{
    if (q15) {
        if (p.nco == 2) return cw_launch<NS, 2, int16_t, int16_t>(p, src, dst, st);
        if (p.nco == 1) return cw_launch<NS, 1, int16_t, int16_t>(p, src, dst, st);
        return cw_launch<NS, 0, int16_t, int16_t>(p, src, dst, st);
    }
    if (p.nco == 2) return cw_launch<NS, 2, float, float>(p, src, dst, st);
    if (p.nco == 1) return cw_launch<NS, 1, float, float>(p, src, dst, st);
    return cw_launch<NS, 0, float, float>(p, src, dst, st);
}

hipError_t launch_cw_fused(const RxParams &p, const void *src, bool src_q15, void *dst, bool dst_q15, hipStream_t st)
{
    if (src_q15 != dst_q15) return hipErrorNotSupported;
    if (p.nbiq == 2) return cw_launch_ns<2>(p, src, src_q15, dst, st);
    if (p.nbiq == 4) return cw_launch_ns<4>(p, src, src_q15, dst, st);
    if (p.nbiq == 8) return cw_launch_ns<8>(p, src, src_q15, dst, st);
    return hipErrorNotSupported;
}

}  // namespace srx
