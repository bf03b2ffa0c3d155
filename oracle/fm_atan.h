/*
 * fm_atan.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The arctangent of the FM discriminator.  CMSIS-DSP 1.5.3 has none (no arm_atan2_f32 before 1.10), so -- like the AGC
 * gain law -- it is BUILD-DEFINED ("FM"): a stated sequence of f32 operations that the oracle, the
 * real-CMSIS harness (ref_chain.c) and the HIP kernels (csrc/rx_device.h, fm_atan2) all perform, each operation rounded
 * once (no fused multiply-add in any arithmetic mode, correctly rounded division):
 *
 *   ax = |x|, ay = |y|, mx = max, mn = min;  mx == 0 -> 0
 *   t = mn / mx                   in [0, 1]
 *   s = t * t
 *   p = C8;  p = p * s;  p = p + Ck   for k = 7 .. 0      (atan(t) / t as a polynomial in t^2: Chebyshev fit on [0, 1])
 *   r = p * t
 *   ay > ax: r = pi/2 - r;   x < 0: r = pi - r;   y < 0: r = -r
 *
 * |r - atan2(y, x)| <= 4e-7 rad over the plane (tests/test_oracle_golden.py::test_fm_arctangent_accuracy).
 */
#ifndef FM_ATAN_H
#define FM_ATAN_H

#define FM_ATAN_PI      0x1.921fb6p+1f
#define FM_ATAN_HALF_PI 0x1.921fb6p+0f
#define FM_AUDIO_SCALE  0x1.45f306p-2f     /* 1 / pi: the discriminator's audio is the phase step per output sample in half turns */

static const float fm_atan_c[9] = {
    0x1.000000p+0f, -0x1.5554a2p-2f, 0x1.997232p-3f, -0x1.22de60p-3f, 0x1.b3ae74p-4f,
    -0x1.330372p-4f, 0x1.5ce0b0p-5f, -0x1.0639f6p-6f, 0x1.73776ap-9f
};

static inline float fm_atan2_f32(float y, float x)
{
    const float ax = x < 0.0f ? -x : x, ay = y < 0.0f ? -y : y;
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    const float t = mn / mx;
    const float s = t * t;
    float p = fm_atan_c[8];
    for (int k = 7; k >= 0; --k) {
        p = p * s;
        p = p + fm_atan_c[k];
    }
    float r = p * t;
    if (ay > ax) r = FM_ATAN_HALF_PI - r;
    if (x < 0.0f) r = FM_ATAN_PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

#endif
