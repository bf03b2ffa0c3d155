// cw_pattern_roof.hip -- what the memory system gives the FETCH PATTERN of k_cw_fused<4,2,256> (csrc/rx_cw.hip), no arithmetic.
//
// k_cw_fused: one single-wave workgroup per 16 channels (4096 workgroups for cfg4, 9 resident per CU through its 16.6 KB of LDS); the
// wave fetches its 16 rows (32 KB apart: iq[C][B][2], B = 4096 f32 complex samples) as bursts of sixteen 1 KB buffer loads, ONE
// row each, one burst ahead of the systolic steps; per 256-sample DSP block it stores 16 rows x 1 KB of audio.
// This tool reproduces exactly that (PL = 1) and the alternatives a restaged kernel could have:
//
//   PL    = 1 KB pieces of ONE row per burst, fetched back to back: a burst covers 16 / PL rows x PL KB contiguous
//           (PL = 1: 16 rows x 1 KB, the product; 2: 8 x 2 KB; 4: 4 x 4 KB; 16: 1 row x 16 KB)
//   DEPTH = bursts in flight (1: the product's one chunk ahead; 2: two)
//   lds   = dynamic LDS per workgroup (sets the residency: 16640 B -> 9 per CU like the product)
//   persistent = 0: one workgroup per 16 rows (the product's launch); 1: 9 x CUs workgroups striding over the row groups
//
// The consumption order is the product's: wait for burst t (one integer add per loaded register), request burst t + DEPTH into the
// registers just freed, and after every second burst (one DSP block: 2 KB per row consumed) the 16 KB store burst, 16 rows x 1 KB.
// State: 1 KB per wave in (start) and out (end), like the kernel's float4 per lane.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/cw_pattern_roof tools/cw_pattern_roof.hip ; tools/cw_pattern_roof [channels] [samples]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

template <int PL, int DEPTH, int SPREAD = 0>
__global__ __launch_bounds__(64) void k_pat(const char *__restrict__ in, char *__restrict__ out, float *__restrict__ state,
                                            unsigned groups, unsigned in_row, unsigned out_row, unsigned work)
{
    extern __shared__ float dummy[];
    constexpr int RB = 16 / PL;                              // rows per burst
    const int lane = threadIdx.x;
    const unsigned T = in_row / 1024u;                       // bursts per row group (16 rows x in_row bytes / 16 KB)
    u4v acc = { 0u, 0u, 0u, 0u };
    for (unsigned g = blockIdx.x; g < groups; g += gridDim.x) {
        const __amdgpu_buffer_rsrc_t ri = rsrc(in + (size_t)g * 16u * in_row, 16u * in_row);
        const __amdgpu_buffer_rsrc_t ro = rsrc(out + (size_t)g * 16u * out_row, 16u * out_row);
        const float4 st = reinterpret_cast<const float4 *>(state)[(size_t)g * 64 + lane];
        u4v A[16], B[DEPTH == 2 ? 16 : 1];
        auto issue = [&](auto &R, unsigned t) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned past = t >= T ? 0x70000000u : 0u;      // past the end: outside the descriptor's range, no memory access
            const unsigned sc = t / PL, gq = t % PL;         // super-chunk (PL KB of every row), row quarter of it
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned row = gq * RB + j / PL, piece = j % PL;
                R[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, lane * 16, (int)(row * in_row + (sc * PL + piece) * 1024u + past), 2);
            }
        };
        auto consume = [&](auto &R) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += R[j];
            __builtin_amdgcn_sched_barrier(0);
        };
        // `work` dependent full-rate vector instructions per burst, BEHIND the request for the next burst like the systolic steps of the
        // product (~1480 vector instructions per chunk there: SQ_INSTS_VALU / (workgroups x chunks)): how much arithmetic between two
        // bursts the pattern tolerates before the waves stop covering the memory latency
        auto busy = [&]() {
            float w0 = __uint_as_float(acc.y);
            for (unsigned i = 0; i < work / 8u; ++i)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n"
                             "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0" : "+v"(w0));
            acc.y = __float_as_uint(w0);
            __builtin_amdgcn_sched_barrier(0);
        };
        // SPREAD: the burst's sixteen loads not back to back but one per sixteenth of the arithmetic (does a wave that dumps 16 KB of requests
        // at once stand in its own way?)
        auto issue_spread = [&](auto &R, unsigned t) {
            const unsigned past = t >= T ? 0x70000000u : 0u;
            const unsigned sc = t / PL, gq = t % PL;
            float w0 = __uint_as_float(acc.y);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned row = gq * RB + j / PL, piece = j % PL;
                R[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, lane * 16, (int)(row * in_row + (sc * PL + piece) * 1024u + past), 2);
                for (unsigned i = 0; i < work / 128u; ++i)
                    asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n"
                                 "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0" : "+v"(w0));
                __builtin_amdgcn_sched_barrier(0);
            }
            acc.y = __float_as_uint(w0);
        };
        auto stores = [&](unsigned blk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b128(acc, ro, lane * 16, (int)(r * out_row + blk * 1024u), 2);
        };
        issue(A, 0);
        if constexpr (DEPTH == 2) issue(B, 1);
        for (unsigned t = 0; t < T; t += 2) {
            if constexpr (SPREAD) {
                static_assert(DEPTH == 1, "spread: depth 1");
                consume(A); issue_spread(A, t + 1);
                consume(A); stores(t / 2); issue_spread(A, t + 2);
                continue;
            } else if constexpr (DEPTH == 1) {
                consume(A); issue(A, t + 1); busy();
                consume(A); issue(A, t + 2);
            } else {
                consume(A); issue(A, t + 2); busy();
                consume(B); issue(B, t + 3);
            }
            stores(t / 2);
            busy();
        }
        float4 so = st;
        so.x += __uint_as_float(acc.x & 1u);
        reinterpret_cast<float4 *>(state)[(size_t)g * 64 + lane] = so;
    }
    if (acc.x == 0x12345678u) dummy[lane] = 1.0f;
}

typedef void (*kern_t)(const char *, char *, float *, unsigned, unsigned, unsigned, unsigned);
struct Variant { const char *name; kern_t k; };

int main(int argc, char **argv)
{
    const unsigned C = argc > 1 ? (unsigned)atoi(argv[1]) : 65536u, BS = argc > 2 ? (unsigned)atoi(argv[2]) : 4096u;
    const bool sweep_work = argc > 3 && atoi(argv[3]) != 0;   // third argument: the arithmetic sweep instead of the pattern table
    const unsigned in_row = BS * 8u, out_row = BS * 4u, groups = C / 16u;
    char *in, *out; float *state;
    if (hipMalloc(&in, (size_t)C * in_row) != hipSuccess || hipMalloc(&out, (size_t)C * out_row) != hipSuccess ||
        hipMalloc(&state, (size_t)groups * 1024) != hipSuccess) return 1;
    (void)hipMemset(in, 0, (size_t)C * in_row); (void)hipMemset(state, 0, (size_t)groups * 1024);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double bytes = (double)C * (in_row + out_row) + 2.0 * groups * 1024;
    const Variant vs[] = {
        { "16 rows x 1 KB (product)   depth 1", k_pat<1, 1> }, { "16 rows x 1 KB             depth 2", k_pat<1, 2> },
        { " 8 rows x 2 KB             depth 1", k_pat<2, 1> }, { " 8 rows x 2 KB             depth 2", k_pat<2, 2> },
        { " 4 rows x 4 KB             depth 1", k_pat<4, 1> }, { " 4 rows x 4 KB             depth 2", k_pat<4, 2> },
        { " 2 rows x 8 KB             depth 1", k_pat<8, 1> }, { " 1 row  x 16 KB            depth 1", k_pat<16, 1> },
        { "16 rows x 1 KB spread      depth 1", k_pat<1, 1, 1> },
    };
    const size_t ldss[] = { 16640, 20000, 32768, 13000, 10000, 6000 };   // 9, 8, 5 (4 with the allocation granule), 12, 16, 26 workgroups per CU
    const int NIT = 40;
    std::vector<hipEvent_t> ev(NIT + 1);
    for (auto &e : ev) (void)hipEventCreate(&e);
    printf("# cw_pattern_roof: %u channels x %u samples (rows of %u B in, %u B out), %.3f GB per launch, %d CUs; median / min ms of %d launches, TB/s of the median, frac of 8 TB/s\n",
           C, BS, in_row, out_row, bytes / 1e9, cus, NIT);
    const unsigned works[] = { 0, 256, 512, 768, 1024, 1280, 1536, 1792, 2048, 2560, 3072 };
    for (int rep = 0; rep < 2; ++rep)
        for (int persistent = 0; persistent < 2; ++persistent)
            for (size_t lds : ldss) {
                if ((rep == 1 || sweep_work) && lds != 16640 && lds != 20000) continue;
                for (const Variant &v : vs) for (unsigned work : works) {
                    if (!sweep_work && work != 0) continue;
                    if (sweep_work && (&v != &vs[0] && &v != &vs[1] && &v != &vs[8])) continue;
                    if (!sweep_work && &v == &vs[8]) continue;
                    int per_cu = 0;
                    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v.k, 64, lds);
                    const unsigned grid = persistent ? std::min(groups, (unsigned)(per_cu * cus)) : groups;
                    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(v.k, dim3(grid), dim3(64), lds, 0, in, out, state, groups, in_row, out_row, work);
                    (void)hipEventRecord(ev[0]);
                    for (int i = 0; i < NIT; ++i) {
                        hipLaunchKernelGGL(v.k, dim3(grid), dim3(64), lds, 0, in, out, state, groups, in_row, out_row, work);
                        (void)hipEventRecord(ev[i + 1]);
                    }
                    (void)hipEventSynchronize(ev[NIT]);
                    std::vector<float> ms(NIT);
                    for (int i = 0; i < NIT; ++i) (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
                    std::sort(ms.begin(), ms.end());
                    const float med = ms[NIT / 2];
                    printf("%s  lds %5zu (%d/CU) %s grid %5u work %4u: %.4f / %.4f ms  %.3f TB/s  %.3f\n", v.name, lds, per_cu,
                           persistent ? "persistent" : "one-shot  ", grid, work, med, ms[0], bytes / med / 1e9, bytes / med / 1e9 / 8.0);
                }
            }
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
