/*
 * dsp_if_slot.c -- host-side C mirror of the firmware's per-block RX callback slot, calling the
 * MI355X library through the C-ABI only (include/selenite_rx.h).
 *
 * In the firmware the slot is HAL_I2SEx_TxRx{Half,}CpltCallback (Core/Src/dsp_if.c:50-54,63-67):
 * DSP_In_Buff_Write() receives 192 uint16 = 96 interleaved int16 I/Q frames per millisecond
 * (Core/Inc/dsp_if.h:69-73, interleave dsp_if.c:286-289) and DSP_Out_Buff_Read() hands the same
 * amount back.  Here the same two functions exist for a BATCH of channels: the int16 frames of
 * every channel go in, demodulated int16 audio comes out, and DSP_Set_Mode() is no longer empty.
 *
 * Build (GPU box):  gcc -O2 -I../../include dsp_if_slot.c -L.. -lselenite_rx -Wl,-rpath,'$ORIGIN/..' -lm -o dsp_if_slot
 * This file is an integration example and a smoke test of the pure-C linkage; it contains no DSP.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "selenite_rx.h"

#define DSP_CHANNELS   64u      /* independent receivers handled per callback */
#define DSP_BLOCK      96u      /* complex samples per DSP block (AGC update) = one I2S half-buffer: 192 uint16 words, 96 I/Q
                                 * frames per millisecond at 96 kHz (Core/Inc/dsp_if.h:69-73); 24 audio samples behind the /4
                                 * decimator.  Round 3: this geometry runs on the fused kernel (passes of ten blocks) */
#define DSP_DECIM      4u
#define DSP_ND_TAPS    256u
#define DSP_NH_TAPS    63u

static selenite_rx_instance *rx;

/* mirrors void DSP_Init(void) (dsp_if.c:377-383) */
int DSP_Init(void)
{
    static float dec[DSP_ND_TAPS], hilb[DSP_NH_TAPS], dly[DSP_NH_TAPS];
    selenite_rx_config cfg;
    memset(&cfg, 0, sizeof cfg);
    if (selenite_rx_design_lowpass(dec, DSP_ND_TAPS, 0.4 / DSP_DECIM)) return -1;
    if (selenite_rx_design_hilbert(hilb, dly, DSP_NH_TAPS)) return -1;
    cfg.struct_size = sizeof cfg;
    cfg.abi_version = SELENITE_RX_ABI_VERSION;
    cfg.channels = DSP_CHANNELS; cfg.block = DSP_BLOCK; cfg.decim = DSP_DECIM;
    cfg.nd_taps = DSP_ND_TAPS; cfg.nh_taps = DSP_NH_TAPS;
    cfg.arith = SELENITE_ARITH_CMSIS;
    cfg.mode = SELENITE_MODE_LSB;            /* RXTX_Init() boots in LSB (rxtx_if.c:686-699) */
    cfg.nco_enable = 1; cfg.nco_step_all = 0x01000000u;
    cfg.agc_enable = 1;
    cfg.dec_coeffs = dec; cfg.hilb_coeffs = hilb; cfg.delay_coeffs = dly;
    cfg.agc_target = 0.5f; cfg.agc_attack = 0.5f; cfg.agc_decay = 0.05f;
    cfg.agc_gain_min = 1e-3f; cfg.agc_gain_max = 1e4f; cfg.agc_env_floor = 1e-6f; cfg.agc_gain_init = 1.0f;
    return selenite_rx_init(&rx, &cfg);
}

/* mirrors void DSP_Set_Mode(uint8_t mode) (dsp_if.c:367-370; called from PTT_Set_Mode, rxtx_if.c:640-648) */
void DSP_Set_Mode(uint8_t mode) { (void)selenite_rx_set_mode(rx, mode); }

/* the slot: what sits between DSP_In_Buff_Write and DSP_Out_Buff_Read for a batch of channels.
 * pbuf_in : int16 [DSP_CHANNELS][frames][2]   (I,Q interleaved, dsp_if.c:286-289)
 * pbuf_out: int16 [DSP_CHANNELS][frames/DSP_DECIM] */
void DSP_Process_Block(const int16_t *pbuf_in, int16_t *pbuf_out, uint16_t frames)
{
    selenite_rx_process_q15(rx, pbuf_in, pbuf_out, frames);
}

int main(void)
{
    const uint32_t frames = DSP_BLOCK;                       /* one slot per call, as the firmware's callback gets it */
    int rc = DSP_Init();
    if (rc != SELENITE_RX_SUCCESS) {
        fprintf(stderr, "DSP_Init failed: %d (%s)\n", rc, selenite_rx_error_string(NULL));
        return rc == SELENITE_RX_DEVICE_ERROR ? 77 : 1;      /* 77: no GPU here */
    }
    float *f = malloc(sizeof(float) * DSP_CHANNELS * frames * 2);
    int16_t *in = malloc(sizeof(int16_t) * DSP_CHANNELS * frames * 2);
    int16_t *out = malloc(sizeof(int16_t) * DSP_CHANNELS * frames / DSP_DECIM);
    for (int call = 0; call < 24; ++call) {                  /* 24 ms of signal */
        selenite_rx_synth_iq_host(f, 0, DSP_CHANNELS, (uint64_t)call * frames, frames, 0x5E1E917Eull);
        for (size_t i = 0; i < (size_t)DSP_CHANNELS * frames * 2; ++i) in[i] = (int16_t)(f[i] * 32768.0f);
        if (call == 16) DSP_Set_Mode(SELENITE_MODE_USB);
        DSP_Process_Block(in, out, (uint16_t)frames);
        if (selenite_rx_status(rx) != SELENITE_RX_SUCCESS) {
            fprintf(stderr, "process failed: %s\n", selenite_rx_error_string(rx));
            return 1;
        }
        long peak = 0;
        for (size_t i = 0; i < (size_t)DSP_CHANNELS * frames / DSP_DECIM; ++i)
            if (labs(out[i]) > peak) peak = labs(out[i]);
        if (call % 8 == 7) printf("call %d: kernel %s, audio peak %ld / 32768\n", call, selenite_rx_kernel_name(rx), peak);
    }
    selenite_rx_free(rx);
    free(f); free(in); free(out);
    return 0;
}
