// LDS read micro-benchmark: cost of 2-byte-misaligned ds_read_b32/b64/b128 vs aligned and vs u16 d16 pairs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) bench(uint32_t *out, int iters, int misalign, int stride)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    for (int i = threadIdx.x; i < 32768 / 4; i += 256)
        reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t addr = (uint32_t)(threadIdx.x & 63) * stride + misalign + (threadIdx.x >> 6) * 4096;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        uint32_t a0, a1, a2, a3;
        if (MODE == 0) { // 4 x b32
            asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:64\n ds_read_b32 %2, %4 offset:128\n ds_read_b32 %3, %4 offset:192\n s_waitcnt lgkmcnt(0)"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(addr));
            acc ^= a0 ^ a1 ^ a2 ^ a3;
        } else if (MODE == 1) { // 4 x b64
            uint64_t b0, b1, b2, b3;
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:64\n ds_read_b64 %2, %4 offset:128\n ds_read_b64 %3, %4 offset:192\n s_waitcnt lgkmcnt(0)"
                         : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(addr));
            acc ^= (uint32_t)(b0 ^ b1 ^ b2 ^ b3) ^ (uint32_t)((b0 ^ b1 ^ b2 ^ b3) >> 32);
        } else if (MODE == 2) { // 4 x b128
            uint4 b0, b1, b2, b3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:64\n ds_read_b128 %2, %4 offset:128\n ds_read_b128 %3, %4 offset:192\n s_waitcnt lgkmcnt(0)"
                         : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(addr));
            acc ^= b0.x ^ b1.y ^ b2.z ^ b3.w;
        } else if (MODE == 3) { // 4 x (u16_d16 + u16_d16_hi) = 4 packed pairs
            a0 = a1 = a2 = a3 = 0;
            asm volatile("ds_read_u16_d16 %0, %4\n ds_read_u16_d16_hi %0, %4 offset:2\n ds_read_u16_d16 %1, %4 offset:64\n ds_read_u16_d16_hi %1, %4 offset:66\n"
                         "ds_read_u16_d16 %2, %4 offset:128\n ds_read_u16_d16_hi %2, %4 offset:130\n ds_read_u16_d16 %3, %4 offset:192\n ds_read_u16_d16_hi %3, %4 offset:194\n s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(addr));
            acc ^= a0 ^ a1 ^ a2 ^ a3;
        } else if (MODE == 4) { // 4 x read2_b32 (two dwords each)
            uint64_t b0, b1, b2, b3;
            asm volatile("ds_read2_b32 %0, %4 offset0:0 offset1:1\n ds_read2_b32 %1, %4 offset0:16 offset1:17\n ds_read2_b32 %2, %4 offset0:32 offset1:33\n ds_read2_b32 %3, %4 offset0:48 offset1:49\n s_waitcnt lgkmcnt(0)"
                         : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(addr));
            acc ^= (uint32_t)(b0 ^ b1 ^ b2 ^ b3) ^ (uint32_t)((b0 ^ b1 ^ b2 ^ b3) >> 32);
        }
        addr = (addr + (acc & 0)) ; // keep a dependence
    }
    const unsigned long long c1 = clock64();
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        out[65536 + 2 * blockIdx.x] = (uint32_t)(c1 - c0);
        out[65536 + 2 * blockIdx.x + 1] = (uint32_t)(t1 - t0);
    }
}

template <int MODE>
int run(const char *name, uint32_t *d_out, int misalign, int stride)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    bench<MODE><<<256, 256>>>(d_out, 10, misalign, stride);
    CK(hipEventRecord(e0));
    bench<MODE><<<256, 256>>>(d_out, iters, misalign, stride);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    uint32_t h[2];
    CK(hipMemcpy(h, d_out + 65536, 8, hipMemcpyDeviceToHost));
    // one workgroup (4 waves) per CU: per iteration each wave issues 4 (or 8) LDS instructions
    printf("%-22s misalign=%d stride=%3d: %.3f ms, clock64 %.1f per iter (4 waves/CU, 4 reads each)\n", name, misalign, stride, ms,
           (double)h[0] / iters);
    return 0;
}

int main()
{
    uint32_t *d_out;
    CK(hipMalloc(&d_out, (65536 + 1024) * 4));
    const int strides[3] = {4, 24, 36};
    for (int s = 0; s < 3; ++s)
        for (int mis = 0; mis <= 2; mis += 2) {
            run<0>("4x ds_read_b32", d_out, mis, strides[s]);
            run<1>("4x ds_read_b64", d_out, mis, strides[s]);
            run<4>("4x ds_read2_b32", d_out, mis ? 4 : 0, strides[s]);
            run<3>("4x (u16_d16,d16_hi)", d_out, mis, strides[s]);
        }
    for (int mis = 0; mis <= 8; mis += 2)
        run<2>("4x ds_read_b128", d_out, mis, 16);
    for (int mis = 0; mis <= 8; mis += 2)
        run<1>("4x ds_read_b64 s8", d_out, mis, 8);
    for (int mis = 0; mis <= 4; mis += 2)
        run<0>("4x ds_read_b32 s4", d_out, mis, 4);
    return 0;
}
