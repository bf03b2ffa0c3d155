// ring.hip -- batched DSP ring buffers in HBM (include/selenite_ring.h; reference: Core/Src/dsp_if.c
// :83-340).  One wavefront moves one ring's packet: the pointer logic of the reference call is
// evaluated once per ring in scalar form, then the lanes copy frames.  A write of nf frames
// touches slots (w0 + k) mod N for k = 0..nf (k = nf is the repeated last frame, dsp_if.c:171-173
// / :292-294) and leaves wr = (w0 + nf) mod N; when nf + 1 > N the sequential reference keeps the
// LAST frame that lands on a slot, so lane k writes only if k + N > nf.
//
// HBM-bound integer work: 4 bytes read + 4 bytes written per frame; nothing here wants LDS or MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/selenite_ring.h"
#include "../../include/selenite_rx.h"

struct selenite_ring {
    uint32_t channels = 0, frames = 0;
    int device = 0;
    int16_t *d_i = nullptr, *d_q = nullptr;
    uint8_t *d_en = nullptr;
    uint16_t *d_rd = nullptr, *d_wr = nullptr;
    int16_t *d_io = nullptr;
    size_t io_bytes = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    int status = 0;
    std::string err;
};

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerGroup = 4;

struct RingArgs {
    uint32_t channels, n;        // n = frames per ring (DSP_BUFF_SIZE)
    int16_t *i, *q;
    uint8_t *en;
    uint16_t *rd, *wr;
};

// OUT = 0: DSP_In_Buff_Write (gap counts only once a reader primed the ring, dsp_if.c:252-265)
// OUT = 1: DSP_Out_Buff_Write (first call parks wr half a ring ahead of rd, dsp_if.c:124-134)
template <int OUT>
__global__ __launch_bounds__(kWave * kWavesPerGroup) void k_ring_write(RingArgs a, const int16_t *__restrict__ src,
                                                                       uint32_t size_words)
{
    const uint32_t c = blockIdx.x * kWavesPerGroup + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= a.channels) return;
    const uint16_t n = (uint16_t)a.n;
    uint16_t wr = a.wr[c];
    const uint16_t rd = a.rd[c];
    uint8_t en = a.en[c];
    uint16_t gap = 0;
    if (OUT) {
        if (en == 0) {
            wr = (uint16_t)(rd + n / 2U);
            if (wr >= n) wr = (uint16_t)(wr - n);
            en = 1;
        }
    }
    if (OUT || en) {
        gap = wr;
        if (rd > wr) gap = (uint16_t)(gap + n);
        gap = (uint16_t)(gap - rd);
    }
    if (gap > (3U * n / 4U)) {                       // writer ahead: step back one slot
        if (wr < 1U) wr = (uint16_t)(wr + n);
        wr = (uint16_t)(wr - 1U);
    }
    if (gap < (n / 4U)) {                            // reader ahead: step forward one slot
        wr = (uint16_t)(wr + 1U);
        if (wr >= n) wr = (uint16_t)(wr - n);
    }
    const uint32_t nf = size_words / 2U;
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src + (size_t)c * size_words);
    int16_t *ri = a.i + (size_t)c * a.n, *rq = a.q + (size_t)c * a.n;
    for (uint32_t k = lane; k <= nf; k += kWave) {
        if (k + a.n <= nf) continue;                 // a later frame lands on this slot
        const uint32_t fr = s32[k < nf ? k : nf - 1];
        const uint32_t slot = (wr + k) % a.n;
        ri[slot] = (int16_t)(fr & 0xFFFFu);
        rq[slot] = (int16_t)(fr >> 16);
    }
    if (lane == 0) {
        a.wr[c] = (uint16_t)((wr + nf) % a.n);
        if (OUT) a.en[c] = en;
    }
}

// IN = 1: DSP_In_Buff_Read (first call parks rd half a ring behind wr; >= N resets to 0, dsp_if.c:316-326)
// IN = 0: DSP_Out_Buff_Read (dsp_if.c:204-219)
template <int IN>
__global__ __launch_bounds__(kWave * kWavesPerGroup) void k_ring_read(RingArgs a, int16_t *__restrict__ dst,
                                                                      uint32_t size_words)
{
    const uint32_t c = blockIdx.x * kWavesPerGroup + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= a.channels) return;
    const uint16_t n = (uint16_t)a.n;
    uint16_t rd = a.rd[c];
    if (IN) {
        if (a.en[c] == 0) {
            rd = (uint16_t)(a.wr[c] + n / 2U);
            if (rd >= n) rd = 0;
            if (lane == 0) a.en[c] = 1;
        }
    }
    const uint32_t nf = size_words / 2U;
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + (size_t)c * size_words);
    const int16_t *ri = a.i + (size_t)c * a.n, *rq = a.q + (size_t)c * a.n;
    for (uint32_t k = lane; k < nf; k += kWave) {
        const uint32_t slot = (rd + k) % a.n;
        d32[k] = (uint32_t)(uint16_t)ri[slot] | ((uint32_t)(uint16_t)rq[slot] << 16);
    }
    if (lane == 0) a.rd[c] = (uint16_t)((rd + nf) % a.n);
}

int fail(selenite_ring *R, int code, const std::string &msg)
{
    if (R && R->status == 0) { R->status = code; R->err = msg; }
    return code;
}

#define RCHK(R, call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((R), SELENITE_RX_DEVICE_ERROR, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)

RingArgs args_of(const selenite_ring *R)
{
    return RingArgs{ R->channels, R->frames, R->d_i, R->d_q, R->d_en, R->d_rd, R->d_wr };
}

dim3 grid_of(const selenite_ring *R) { return dim3((R->channels + kWavesPerGroup - 1) / kWavesPerGroup); }

bool size_ok(selenite_ring *R, uint32_t size_words, const char *who)
{
    if (!R) return false;
    if (size_words < 2U || (size_words & 1U)) {
        fail(R, SELENITE_RX_LENGTH_ERROR, std::string(who) + ": size is not a non-zero whole number of I/Q frames");
        return false;
    }
    return true;
}

int launch_write(selenite_ring *R, bool out, const int16_t *dSrc, uint32_t size_words)
{
    RCHK(R, hipSetDevice(R->device));
    if (out) hipLaunchKernelGGL(k_ring_write<1>, grid_of(R), dim3(kWave * kWavesPerGroup), 0, R->stream, args_of(R), dSrc, size_words);
    else hipLaunchKernelGGL(k_ring_write<0>, grid_of(R), dim3(kWave * kWavesPerGroup), 0, R->stream, args_of(R), dSrc, size_words);
    RCHK(R, hipGetLastError());
    return 0;
}

int launch_read(selenite_ring *R, bool in, int16_t *dDst, uint32_t size_words)
{
    RCHK(R, hipSetDevice(R->device));
    if (in) hipLaunchKernelGGL(k_ring_read<1>, grid_of(R), dim3(kWave * kWavesPerGroup), 0, R->stream, args_of(R), dDst, size_words);
    else hipLaunchKernelGGL(k_ring_read<0>, grid_of(R), dim3(kWave * kWavesPerGroup), 0, R->stream, args_of(R), dDst, size_words);
    RCHK(R, hipGetLastError());
    return 0;
}

int ensure_io(selenite_ring *R, size_t bytes)
{
    if (R->io_bytes >= bytes) return 0;
    RCHK(R, hipStreamSynchronize(R->stream));
    if (R->d_io) (void)hipFree(R->d_io);
    R->d_io = nullptr; R->io_bytes = 0;
    RCHK(R, hipMalloc((void **)&R->d_io, bytes));
    R->io_bytes = bytes;
    return 0;
}

int host_write(selenite_ring *R, bool out, const int16_t *src, uint32_t size_words)
{
    const size_t bytes = (size_t)R->channels * size_words * sizeof(int16_t);
    if (ensure_io(R, bytes)) return R->status;
    RCHK(R, hipMemcpyAsync(R->d_io, src, bytes, hipMemcpyHostToDevice, R->stream));
    if (launch_write(R, out, R->d_io, size_words)) return R->status;
    RCHK(R, hipStreamSynchronize(R->stream));
    return 0;
}

int host_read(selenite_ring *R, bool in, int16_t *dst, uint32_t size_words)
{
    const size_t bytes = (size_t)R->channels * size_words * sizeof(int16_t);
    if (ensure_io(R, bytes)) return R->status;
    if (launch_read(R, in, R->d_io, size_words)) return R->status;
    RCHK(R, hipMemcpyAsync(dst, R->d_io, bytes, hipMemcpyDeviceToHost, R->stream));
    RCHK(R, hipStreamSynchronize(R->stream));
    return 0;
}

int reset(selenite_ring *R)
{
    const size_t C = R->channels, N = R->frames;
    RCHK(R, hipMemsetAsync(R->d_i, 0, C * N * sizeof(int16_t), R->stream));
    RCHK(R, hipMemsetAsync(R->d_q, 0, C * N * sizeof(int16_t), R->stream));
    RCHK(R, hipMemsetAsync(R->d_en, 0, C, R->stream));
    RCHK(R, hipMemsetAsync(R->d_rd, 0, C * sizeof(uint16_t), R->stream));
    RCHK(R, hipMemsetAsync(R->d_wr, 0, C * sizeof(uint16_t), R->stream));
    RCHK(R, hipStreamSynchronize(R->stream));
    return 0;
}

}  // namespace

extern "C" int selenite_ring_init(selenite_ring **out, uint32_t channels, uint32_t frames)
{
    if (!out) return SELENITE_RX_ARGUMENT_ERROR;
    *out = nullptr;
    if (channels == 0) return SELENITE_RX_ARGUMENT_ERROR;
    if (frames < 4 || frames > 32767) return SELENITE_RX_LENGTH_ERROR;   // uint16 gap arithmetic must not wrap
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return SELENITE_RX_DEVICE_ERROR;   // no CPU fallback
    selenite_ring *R = new selenite_ring;
    R->channels = channels;
    R->frames = frames;
    auto bail = [&](int code) { selenite_ring_free(R); return code; };
    if (hipGetDevice(&R->device) != hipSuccess) return bail(SELENITE_RX_DEVICE_ERROR);
    if (hipStreamCreateWithFlags(&R->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(SELENITE_RX_DEVICE_ERROR);
    R->stream = R->own_stream;
    const size_t C = channels, N = frames;
    if (hipMalloc((void **)&R->d_i, C * N * sizeof(int16_t)) != hipSuccess ||
        hipMalloc((void **)&R->d_q, C * N * sizeof(int16_t)) != hipSuccess ||
        hipMalloc((void **)&R->d_en, C) != hipSuccess ||
        hipMalloc((void **)&R->d_rd, C * sizeof(uint16_t)) != hipSuccess ||
        hipMalloc((void **)&R->d_wr, C * sizeof(uint16_t)) != hipSuccess)
        return bail(SELENITE_RX_DEVICE_ERROR);
    if (reset(R)) return bail(SELENITE_RX_DEVICE_ERROR);
    *out = R;
    return SELENITE_RX_SUCCESS;
}

extern "C" void selenite_ring_free(selenite_ring *R)
{
    if (!R) return;
    void *ptrs[] = { R->d_i, R->d_q, R->d_en, R->d_rd, R->d_wr, R->d_io };
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (R->own_stream) (void)hipStreamDestroy(R->own_stream);
    delete R;
}

extern "C" int selenite_ring_status(const selenite_ring *R) { return R ? R->status : SELENITE_RX_ARGUMENT_ERROR; }
extern "C" const char *selenite_ring_error_string(const selenite_ring *R) { return R ? R->err.c_str() : "null ring"; }

extern "C" int selenite_ring_set_stream(selenite_ring *R, void *hip_stream)
{
    if (!R) return SELENITE_RX_ARGUMENT_ERROR;
    RCHK(R, hipStreamSynchronize(R->stream));
    R->stream = hip_stream ? (hipStream_t)hip_stream : R->own_stream;
    return 0;
}

extern "C" int selenite_ring_sync(selenite_ring *R)
{
    if (!R) return SELENITE_RX_ARGUMENT_ERROR;
    RCHK(R, hipStreamSynchronize(R->stream));
    return R->status;
}

extern "C" void selenite_ring_in_write_device(selenite_ring *R, const int16_t *dSrc, uint16_t size_words)
{
    if (size_ok(R, size_words, "selenite_ring_in_write")) launch_write(R, false, dSrc, size_words);
}
extern "C" void selenite_ring_out_write_device(selenite_ring *R, const int16_t *dSrc, uint32_t size_bytes)
{
    if (size_ok(R, size_bytes / 2U, "selenite_ring_out_write")) launch_write(R, true, dSrc, size_bytes / 2U);
}
extern "C" void selenite_ring_in_read_device(selenite_ring *R, int16_t *dDst, uint32_t size_bytes)
{
    if (size_ok(R, size_bytes / 2U, "selenite_ring_in_read")) launch_read(R, true, dDst, size_bytes / 2U);
}
extern "C" void selenite_ring_out_read_device(selenite_ring *R, int16_t *dDst, uint16_t size_words)
{
    if (size_ok(R, size_words, "selenite_ring_out_read")) launch_read(R, false, dDst, size_words);
}

extern "C" void selenite_ring_mute(selenite_ring *R)
{
    if (!R) return;
    const size_t bytes = (size_t)R->channels * R->frames * sizeof(int16_t);
    if (hipMemsetAsync(R->d_i, 0, bytes, R->stream) != hipSuccess || hipMemsetAsync(R->d_q, 0, bytes, R->stream) != hipSuccess)
        fail(R, SELENITE_RX_DEVICE_ERROR, "selenite_ring_mute: memset failed");
}

extern "C" void selenite_ring_in_write(selenite_ring *R, const int16_t *src, uint16_t size_words)
{
    if (size_ok(R, size_words, "selenite_ring_in_write")) host_write(R, false, src, size_words);
}
extern "C" void selenite_ring_out_write(selenite_ring *R, const int16_t *src, uint32_t size_bytes)
{
    if (size_ok(R, size_bytes / 2U, "selenite_ring_out_write")) host_write(R, true, src, size_bytes / 2U);
}
extern "C" void selenite_ring_in_read(selenite_ring *R, int16_t *dst, uint32_t size_bytes)
{
    if (size_ok(R, size_bytes / 2U, "selenite_ring_in_read")) host_read(R, true, dst, size_bytes / 2U);
}
extern "C" void selenite_ring_out_read(selenite_ring *R, int16_t *dst, uint16_t size_words)
{
    if (size_ok(R, size_words, "selenite_ring_out_read")) host_read(R, false, dst, size_words);
}

extern "C" int selenite_ring_get_state(selenite_ring *R, selenite_ring_state_view *v)
{
    if (!R || !v) return SELENITE_RX_ARGUMENT_ERROR;
    const size_t C = R->channels, N = R->frames;
    RCHK(R, hipStreamSynchronize(R->stream));
    if (v->i) RCHK(R, hipMemcpy(v->i, R->d_i, C * N * sizeof(int16_t), hipMemcpyDeviceToHost));
    if (v->q) RCHK(R, hipMemcpy(v->q, R->d_q, C * N * sizeof(int16_t), hipMemcpyDeviceToHost));
    if (v->buff_enable) RCHK(R, hipMemcpy(v->buff_enable, R->d_en, C, hipMemcpyDeviceToHost));
    if (v->rd_ptr) RCHK(R, hipMemcpy(v->rd_ptr, R->d_rd, C * sizeof(uint16_t), hipMemcpyDeviceToHost));
    if (v->wr_ptr) RCHK(R, hipMemcpy(v->wr_ptr, R->d_wr, C * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int selenite_ring_set_state(selenite_ring *R, const selenite_ring_state_view *v)
{
    if (!R || !v) return SELENITE_RX_ARGUMENT_ERROR;
    const size_t C = R->channels, N = R->frames;
    for (size_t c = 0; c < C; ++c) {
        if ((v->rd_ptr && v->rd_ptr[c] >= N) || (v->wr_ptr && v->wr_ptr[c] >= N))
            return fail(R, SELENITE_RX_ARGUMENT_ERROR, "selenite_ring_set_state: pointer outside the ring");
    }
    RCHK(R, hipStreamSynchronize(R->stream));
    if (v->i) RCHK(R, hipMemcpy(R->d_i, v->i, C * N * sizeof(int16_t), hipMemcpyHostToDevice));
    if (v->q) RCHK(R, hipMemcpy(R->d_q, v->q, C * N * sizeof(int16_t), hipMemcpyHostToDevice));
    if (v->buff_enable) RCHK(R, hipMemcpy(R->d_en, v->buff_enable, C, hipMemcpyHostToDevice));
    if (v->rd_ptr) RCHK(R, hipMemcpy(R->d_rd, v->rd_ptr, C * sizeof(uint16_t), hipMemcpyHostToDevice));
    if (v->wr_ptr) RCHK(R, hipMemcpy(R->d_wr, v->wr_ptr, C * sizeof(uint16_t), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int selenite_ring_time_device(selenite_ring *R, const int16_t *dSrc, int16_t *dDst, uint16_t size_words,
                                         uint32_t iters, float *ms_per_pair)
{
    if (!R || !ms_per_pair || iters == 0) return SELENITE_RX_ARGUMENT_ERROR;
    if (!size_ok(R, size_words, "selenite_ring_time_device")) return R->status;
    struct EventPair {                                      // destroyed on every exit path
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    hipEvent_t &e0 = ev.e0, &e1 = ev.e1;
    RCHK(R, hipEventCreate(&e0));
    RCHK(R, hipEventCreate(&e1));
    RCHK(R, hipEventRecord(e0, R->stream));
    for (uint32_t k = 0; k < iters; ++k) {
        if (launch_write(R, false, dSrc, size_words)) return R->status;
        if (launch_read(R, true, dDst, size_words)) return R->status;
    }
    RCHK(R, hipEventRecord(e1, R->stream));
    RCHK(R, hipEventSynchronize(e1));
    float ms = 0.0f;
    RCHK(R, hipEventElapsedTime(&ms, e0, e1));
    *ms_per_pair = ms / (float)iters;
    return 0;
}
