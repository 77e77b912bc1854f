// Exhaustive check of msd_sqrt_cr (msd_mag_impl.h) against the compiler's correctly rounded sqrtf over every float in
// [2^-40, 2] and zero: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../readsb-protobuf_amd/csrc -I../../include sqrt_check.hip -o sqrt_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "msd_mag_impl.h"

__global__ void check(uint32_t first, uint32_t count, unsigned long long *bad, uint32_t *example)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = i; k < count; k += gridDim.x * blockDim.x) {
        const float x = __uint_as_float(first + k);
        const float a = msd_sqrt_cr(x), b = __builtin_sqrtf(x);
        if (__float_as_uint(a) != __float_as_uint(b)) {
            atomicAdd(bad, 1ull);
            *example = first + k;
        }
    }
}

int main()
{
    unsigned long long *bad, hbad = 0;
    uint32_t *ex, hex = 0;
    hipMalloc(&bad, 8);
    hipMalloc(&ex, 4);
    hipMemset(bad, 0, 8);
    hipMemset(ex, 0, 4);
    const uint32_t lo = (uint32_t)(127 - 40) << 23, hi = (uint32_t)(127 + 1) << 23; /* [2^-40, 2) */
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, lo, hi - lo, bad, ex);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, 0u, 1u, bad, ex); /* zero */
    hipDeviceSynchronize();
    hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hex, ex, 4, hipMemcpyDeviceToHost);
    printf("msd_sqrt_cr vs sqrtf over [2^-40, 2) and 0: %u values, %llu differ (example bits %08x)\n", hi - lo + 1, hbad, hex);
    return hbad != 0;
}
