// rx_design.cpp -- coefficient design helpers of include/selenite_rx.h (host only).
//
// Conveniences for callers: the chain itself only ever sees coefficient arrays.  Everything is
// computed in double and rounded once to float; arrays are written in CMSIS order
// ({b[N-1] .. b[0]}, arm_fir_decimate_f32.c:64-69) and with CMSIS's biquad sign convention
// (feedback added, arm_biquad_cascade_df1_f32.c:52-63).
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/selenite_rx.h"

static const double kPi = 3.14159265358979323846;

static double hamming(uint32_t n, uint32_t num_taps)
{
    return 0.54 - 0.46 * std::cos(2.0 * kPi * (double)n / (double)(num_taps - 1));
}

extern "C" int selenite_rx_design_lowpass(float *coeffs, uint32_t num_taps, double cutoff)
{
    if (!coeffs || num_taps < 2 || !(cutoff > 0.0 && cutoff < 0.5)) return SELENITE_RX_ARGUMENT_ERROR;
    std::vector<double> h(num_taps);
    double sum = 0.0;
    for (uint32_t n = 0; n < num_taps; ++n) {
        const double m = (double)n - (double)(num_taps - 1) / 2.0;
        const double x = 2.0 * cutoff * m;
        const double sinc = (x == 0.0) ? 1.0 : std::sin(kPi * x) / (kPi * x);
        h[n] = 2.0 * cutoff * sinc * hamming(n, num_taps);
        sum += h[n];
    }
    for (uint32_t n = 0; n < num_taps; ++n) coeffs[num_taps - 1 - n] = (float)(h[n] / sum);
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_design_hilbert(float *hilb, float *delay, uint32_t num_taps)
{
    if (!hilb || !delay || num_taps < 3 || (num_taps % 2) == 0) return SELENITE_RX_ARGUMENT_ERROR;
    const int c = (int)(num_taps - 1) / 2;
    for (uint32_t n = 0; n < num_taps; ++n) {
        const int m = (int)n - c;
        double v = 0.0;
        if (m & 1) v = 2.0 / (kPi * (double)m) * hamming(n, num_taps);
        hilb[num_taps - 1 - n] = (float)v;
        delay[num_taps - 1 - n] = (m == 0) ? 1.0f : 0.0f;
    }
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_design_bandpass(float *coeffs, uint32_t n_stages, double f0, double q)
{
    if (!coeffs || n_stages == 0 || !(f0 > 0.0 && f0 < 0.5) || !(q > 0.0)) return SELENITE_RX_ARGUMENT_ERROR;
    const double w0 = 2.0 * kPi * f0, alpha = std::sin(w0) / (2.0 * q), a0 = 1.0 + alpha;
    const double b0 = alpha / a0, b1 = 0.0, b2 = -alpha / a0;
    const double a1 = -2.0 * std::cos(w0) / a0, a2 = (1.0 - alpha) / a0;
    for (uint32_t s = 0; s < n_stages; ++s) {
        coeffs[5 * s + 0] = (float)b0;
        coeffs[5 * s + 1] = (float)b1;
        coeffs[5 * s + 2] = (float)b2;
        coeffs[5 * s + 3] = (float)(-a1);
        coeffs[5 * s + 4] = (float)(-a2);
    }
    return SELENITE_RX_SUCCESS;
}
