// rx_hist_exact.h -- hist_exact_channel: the FIR-pair history of one channel recomputed in the reference's arithmetic
// (SELENITE_ARITH_AUTO).  Shared by k_hist_exact (rx_generic.hip) and the kernels that recompute a channel themselves.
#pragma once
#include "rx_internal.h"

#pragma clang fp contract(off)

namespace srx {

// The Hilbert-pair history of ONE channel recomputed in the reference's arithmetic (SELENITE_ARITH_AUTO: what makes the exact rerun of a
// channel exact across a call boundary).  The call before left the channel on the matrix kernel: its FIR-pair history (the last nh - 1
// decimated samples of both rails) has split16 precision, but the mixed samples in front of the decimator state are in hist_ext (word:
// which buffer, which format).  arm_fir_decimate_f32.c:193-284 -- one accumulator from 0, taps ascending, product rounded, then sum rounded -- on
//   T = (one unused slot) ++ hist_ext[0 .. L - 1) (positions [E - H - L + 1, E - H)) ++ decimator state ([E - H, E)),  H = nd - 1, L = ext_len = M * HH4:
// history entry r (r = nh - 2 the newest) is the decimator output whose newest sample sits at E - M (nh - 1 - r): T[t0 .. t0 + nd), t0 = L - M (nh - 1 - r).
// One wavefront; `lds`: 2 (L + H) floats of scratch.  Used by k_hist_exact (rx_generic.hip) and by the matrix kernels when they recompute a
// channel themselves (rx_split16_kernels.h).
__device__ __forceinline__ void hist_exact_channel(const RxParams &p, uint32_t c, uint32_t word, float *lds, int lane)
{
    const uint32_t nd = p.nd, M = p.decim, HH = p.nh - 1u, L = p.ext_len, H = nd - 1u;
    float *TI = lds, *TQ = lds + (L + H);
    const uint32_t buf = (word >> kExtBufShift) & 1u;
    // (the row starts one sample late -- rx_split16_kernels.h: T[0] meets no tap; its last entry repeats the state's first one)
    const float2 *ext = p.hist_ext + (size_t)buf * p.ext_buf_stride + (size_t)c * L;
    if (word & kExtQ15) {
        // an int16-slot call left its RAW samples: arm_q15_to_float and the NCO mix again, sample by sample, with the arithmetic
        // of the chain (same phases: T[i] sits H + L - i samples in front of the channel's current phase)
        const short2 *raw = reinterpret_cast<const short2 *>(ext);
        const uint32_t ph_e = p.nco ? p.phase[c] : 0u, step = p.nco ? p.step[c] : 0u;
        for (uint32_t i = lane; i < L; i += kWave) {
            float2 v = make_float2(0.0f, 0.0f);
            if (i) {
                const short2 q = raw[i - 1];
                v = make_float2(q15_to_float(q.x), q15_to_float(q.y));
                if (p.nco) v = cmul<0>(v, nco_lo<0>(p.sintab, ph_e - (H + L - i) * step));
            }
            TI[i] = v.x; TQ[i] = v.y;
        }
    } else {
        for (uint32_t i = lane; i < L; i += kWave) {
            const float2 v = i ? ext[i - 1] : make_float2(0.0f, 0.0f);
            TI[i] = v.x; TQ[i] = v.y;
        }
    }
    for (uint32_t i = lane; i < H; i += kWave) {
        TI[L + i] = p.dec_state[((size_t)c * 2 + 0) * H + i];
        TQ[L + i] = p.dec_state[((size_t)c * 2 + 1) * H + i];
    }
    __syncthreads();
    for (uint32_t r = lane; r < HH; r += kWave) {
        const uint32_t t0 = L - M * (HH - r);
        float ai = 0.0f, aq = 0.0f;
        for (uint32_t k = 0; k < nd; ++k) {
            const float ck = p.dec_c[k];
            const float pi_ = TI[t0 + k] * ck, pq_ = TQ[t0 + k] * ck;
            ai = ai + pi_;
            aq = aq + pq_;
        }
        p.fir_state[((size_t)c * 2 + 0) * HH + r] = ai;
        p.fir_state[((size_t)c * 2 + 1) * HH + r] = aq;
    }
}

}  // namespace srx
