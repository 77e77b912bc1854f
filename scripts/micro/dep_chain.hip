// Latency of a DEPENDENT vector instruction for a lone wavefront on gfx950: the --dcfilter recurrence
// z = t + z * dc_b (convert.c:135-136) is two of them per sample and channel, and nothing can be run beside
// them that would shorten the chain.  hipcc --offload-arch=gfx950 -O2 dep_chain.hip -o dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../readsb-protobuf_amd/csrc/msd_dc_chain_asm.h"
#define REP8(S) S S S S S S S S
#define REP64(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
template <int KIND>
__global__ void __launch_bounds__(64) chain(float *out, int iters, float a, float b)
{
    float z = threadIdx.x * 1e-3f, y = z + 1.0f;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0)      // mul -> add, one chain: the recurrence itself
            asm volatile(REP64("v_mul_f32 %0, %0, %2\n v_add_f32 %0, %1, %0\n") : "+v"(z) : "v"(a), "v"(b));
        else if (KIND == 1) // two chains interleaved in one wavefront
            asm volatile(REP64("v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %3\n v_add_f32 %0, %2, %0\n v_add_f32 %1, %2, %1\n") : "+v"(z), "+v"(y) : "v"(a), "v"(b));
        else                // add -> add (dependent, 2-cycle-issue class)
            asm volatile(REP64("v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n") : "+v"(z) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 64 + threadIdx.x] = z + y;
}
// the chain loop of msd_dcfilter_kernel as it is: lanes 0 and 1 only, sixteen values from LDS per trip, sixteen back
template <int LANES>
__global__ void __launch_bounds__(64) chain_lds(float *out, int iters, float a, float b)
{
    __shared__ __attribute__((aligned(16))) float tv[2][2048], zv[2][2048];
    for (int i = threadIdx.x; i < 2048; i += 64) { tv[0][i] = a * i; tv[1][i] = a * (i + 1); }
    __syncthreads();
    float z = 0.f;
    if (threadIdx.x < LANES) {
        const int ch = threadIdx.x & 1;
        const float4 *t4 = reinterpret_cast<const float4 *>(tv[ch]);
        float4 *z4 = reinterpret_cast<float4 *>(zv[ch]);
        for (int it = 0; it < iters; ++it) {
            float4 x[4];
            for (int u = 0; u < 4; ++u) x[u] = t4[u];
            for (int i = 0; i < 2048; i += 16) {
                float4 nx[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) nx[u] = t4[((i + 16) >> 2) + u < 512 ? ((i + 16) >> 2) + u : 0];
#define ST(T, OUT) asm volatile("v_mul_f32 %0, %1, %3\n\tv_add_f32 %0, %2, %0" : "=&v"(OUT) : "v"(z), "v"(T), "v"(b)); z = OUT;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 o;
                    ST(x[u].x, o.x) ST(x[u].y, o.y) ST(x[u].z, o.z) ST(x[u].w, o.w)
                    z4[(i >> 2) + u] = o;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = nx[u];
            }
        }
    }
    out[threadIdx.x] = z + zv[0][5];
}
template <int LANES>
static void run_lds(const char *name)
{
    float *out; hipMalloc(&out, 64 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    chain_lds<LANES><<<1, 64>>>(out, 10, 1e-7f, 0.9999974f);
    hipEventRecord(e0); chain_lds<LANES><<<1, 64>>>(out, iters, 1e-7f, 0.9999974f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %6.2f ns per sample (%.3f ms)\n", name, ms * 1e6 / ((double)iters * 2048), ms);
    hipFree(out);
}
// the hand-scheduled loop (msd_dc_chain_asm.h) against the plain one: same bits, and how fast
__global__ void __launch_bounds__(64) chain_asm(float *zout, float *state, int iters, float a, float b, int use_asm)
{
    __shared__ __attribute__((aligned(16))) float tv[2][2048 + 32], zv[2][2048 + 32];
    for (int i = threadIdx.x; i < 2048 + 32; i += 64) { tv[0][i] = a * (float)((i * 7919) % 1000 - 400); tv[1][i] = a * (float)((i * 104729) % 977 - 300); }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int ch = threadIdx.x;
        float z = 0.001f * (ch + 1);
        for (int it = 0; it < iters; ++it) {
            if (use_asm) {
                uint32_t ta = (uint32_t)(uintptr_t)tv[ch], za = (uint32_t)(uintptr_t)zv[ch], n = 2048 / 32;
                MSD_DC_CHAIN_ASM(z, b, ta, za, n);
            } else {
                for (int i = 0; i < 2048; ++i) {
                    float o;
                    asm volatile("v_mul_f32 %0, %1, %3\n\tv_add_f32 %0, %2, %0" : "=&v"(o) : "v"(z), "v"(tv[ch][i]), "v"(b));
                    z = o;
                    zv[ch][i] = z;
                }
            }
        }
        state[ch] = z;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) { zout[i] = zv[0][i]; zout[2048 + i] = zv[1][i]; }
}
static void run_asm()
{
    float *zo[2], *st[2]; static float h[2][4096 + 2];
    for (int k = 0; k < 2; ++k) { hipMalloc(&zo[k], 4096 * sizeof(float)); hipMalloc(&st[k], 2 * sizeof(float)); }
    for (int k = 0; k < 2; ++k) {
        chain_asm<<<1, 64>>>(zo[k], st[k], 3, 2.6e-6f, 0.9999974f, k);
        hipMemcpy(h[k], zo[k], 4096 * sizeof(float), hipMemcpyDeviceToHost); hipMemcpy(h[k] + 4096, st[k], 2 * sizeof(float), hipMemcpyDeviceToHost);
    }
    printf("hand-scheduled loop == plain loop, 3 x 2048 samples x 2 channels, bit for bit: %s\n", memcmp(h[0], h[1], sizeof(h[0])) == 0 ? "yes" : "NO");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipEventRecord(e0); chain_asm<<<1, 64>>>(zo[1], st[1], iters, 2.6e-6f, 0.9999974f, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %6.2f ns per sample (%.3f ms)\n", "the hand-scheduled loop, lanes 0-1", ms * 1e6 / ((double)iters * 2048), ms);
}
template <int KIND>
static void run(const char *name, int blocks, int per_iter)
{
    float *out; hipMalloc(&out, blocks * 64 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    chain<KIND><<<blocks, 64>>>(out, 100, 1e-7f, 0.9999974f);
    hipEventRecord(e0); chain<KIND><<<blocks, 64>>>(out, iters, 1e-7f, 0.9999974f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int clk; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double n = (double)iters * 64 * per_iter;
    printf("%-46s %5d wavefront(s): %6.2f cycles per instruction (%.3f ms, clock %d MHz)\n", name, blocks, ms * 1e-3 * clk * 1e3 / n, ms, clk / 1000);
    hipFree(out);
}
int main()
{
    for (int blocks : {1, 256}) {
        run<0>("v_mul_f32 -> v_add_f32, one dependent chain", blocks, 2);
        run<1>("two such chains interleaved in one wavefront", blocks, 4);
        run<2>("v_add_f32 -> v_add_f32, one dependent chain", blocks, 2);
    }
    run_lds<2>("the kernel's loop, lanes 0-1, LDS in and out");
    run_lds<64>("the same with all 64 lanes");
    run_asm();
    return 0;
}
