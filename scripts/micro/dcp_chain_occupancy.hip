// What one step z = t + z * b of the parallel-in-time --dcfilter evaluation costs a wavefront, by how many wavefronts share a SIMD and
// by where t comes from (round 6).  hipcc --offload-arch=gfx950 -O3 dcp_chain_occupancy.hip -o dcp_chain_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#define STEP(Z, T, B) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %2, %0" : "+v"(Z) : "v"(B), "v"(T))
// KIND 0: t in a register; 1: t by ds_read_b128 broadcast, sixteen steps ahead (msd_dcp_eval_kernel's loop); 2: v_readlane per step
template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, int steps, float a, float b)
{
    __shared__ __attribute__((aligned(16))) float tb[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    tb[wv][lane] = a * lane; tb[wv][lane + 64] = a; tb[wv][lane + 128] = -a; tb[wv][lane + 192] = a * 3;
    __syncthreads();
    float z = lane * 1e-3f;
    const float4 *tq = reinterpret_cast<const float4 *>(tb[wv]);
    if (KIND == 0) {
        for (int s = 0; s < steps; s += 16) {
#pragma unroll
            for (int q = 0; q < 16; ++q) STEP(z, a, b);
        }
    } else if (KIND == 1) {
        float4 x[4] = {tq[0], tq[1], tq[2], tq[3]};
        for (int s = 0; s < steps; s += 16) {
            const int nq = ((s + 16) >> 2) & 63;
            const float4 n0 = tq[nq], n1 = tq[nq + 1], n2 = tq[nq + 2], n3 = tq[nq + 3];
#pragma unroll
            for (int q = 0; q < 4; ++q) { STEP(z, x[q].x, b); STEP(z, x[q].y, b); STEP(z, x[q].z, b); STEP(z, x[q].w, b); }
            x[0] = n0, x[1] = n1, x[2] = n2, x[3] = n3;
        }
    } else {
        float tv = tb[wv][lane];
        for (int s = 0; s < steps; s += 16) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float t = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(tv), q));
                asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %2, %0" : "+v"(z) : "v"(b), "s"(t));
            }
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = z;
}
template <int KIND>
static void run(const char *name, int grid, int steps)
{
    float *out; hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<grid, 256>>>(out, steps, 1e-7f, 0.9999974f);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0); k<KIND><<<grid, 256>>>(out, steps, 1e-7f, 0.9999974f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-28s grid %5d (%4.1f wavefronts per SIMD)  %7.3f ms  %6.2f ns per step of a wavefront, %6.2f ns per step and SIMD-slot\n", name, grid, grid * 4 / 1024.0,
           best, best * 1e6 / steps, best * 1e6 / steps / (grid * 4 / 1024.0 > 1 ? grid * 4 / 1024.0 : 1));
    hipFree(out);
}
int main()
{
    const int steps = 32768;
    for (int grid : {1, 256, 512, 1024, 2048, 4096}) {
        run<0>("t in a register", grid, steps);
        run<1>("t by LDS broadcast", grid, steps);
        run<2>("t by v_readlane", grid, steps);
    }
    return 0;
}
