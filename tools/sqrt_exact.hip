// Exhaustive check of hkd::sqrt_unit (hk_device.h: the compiler's correctly rounded sqrtf without its range scaling and class check)
// against sqrtf for EVERY binary32 in [2^-96, 2^32) and 0:   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -I hikari.jl_amd/csrc tools/sqrt_exact.hip -o /tmp/sqrt_exact && /tmp/sqrt_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include "hikari_mi355x.h"
#include "hk_device.h"
__global__ void k_check(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* first_bad) {
    const unsigned long long n = (unsigned long long)hi - lo;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo + (uint32_t)i);
        const float a = hkd::sqrt_unit(x), b = sqrtf(x);
        if (__float_as_uint(a) != __float_as_uint(b)) {
            atomicAdd(bad, 1ull);
            atomicMin(first_bad, lo + (uint32_t)i);
        }
    }
}
int main() {
    unsigned long long* bad;
    uint32_t* first;
    hipMalloc(&bad, 8);
    hipMalloc(&first, 4);
    hipMemset(bad, 0, 8);
    hipMemset(first, 0xff, 4);
    const uint32_t lo = (127u - 96u) << 23, hi = (127u + 32u) << 23;   // [2^-96, 2^32)
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, lo, hi, bad, first);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, 0u, 1u, bad, first);   // 0
    unsigned long long h = 0;
    uint32_t f = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
    printf("sqrt_unit vs sqrtf over [2^-96, 2^32) and 0: %llu of %llu values differ%s\n", h, (unsigned long long)hi - lo + 1, h ? "" : " (exact)");
    if (h) printf("first differing bit pattern: 0x%08x\n", f);
    return h ? 1 : 0;
}
