// VALU issue-rate microbenchmark for gfx950: how many shader cycles one wave64 instruction of each kind occupies a SIMD, at 1..8
// resident waves per SIMD.  The path-tracing kernels are VALU-issue bound; this measures the ceiling they are priced
// against and which instructions are expensive.   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// X(index, label, asm over %0 (in/out VGPR), %1 / %2 (input VGPRs), clobbers)
#define KINDS(X) \
    X(0, "v_fma_f32", "v_fma_f32 %0, %0, %2, %3", "s22") \
    X(1, "v_mul_f32", "v_mul_f32 %0, %0, %2", "s22") \
    X(2, "v_add_u32", "v_add_u32 %0, %0, %2", "s22") \
    X(3, "v_mov_b32", "v_mov_b32 %0, %2", "s22") \
    X(4, "v_and_b32", "v_and_b32 %0, %0, %2", "s22") \
    X(5, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %0", "s22") \
    X(6, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 5", "s22") \
    X(7, "v_max_f32", "v_max_f32 %0, %0, %2", "s22") \
    X(8, "v_med3_f32", "v_med3_f32 %0, %0, %2, %3", "s22") \
    X(9, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0", "s22") \
    X(10, "v_cmp_lt_f32 vcc (e32)", "v_cmp_lt_f32 vcc, %0, %2", "vcc") \
    X(11, "v_cmp_lt_f32 s[20:21] (e64)", "v_cmp_lt_f32 s[20:21], %0, %2", "s20", "s21") \
    X(12, "v_cndmask_b32 vcc (e32)", "v_cndmask_b32 %0, %0, %2, vcc", "s22") \
    X(13, "v_cndmask_b32 s[20:21] (e64)", "v_cndmask_b32 %0, %0, %2, s[20:21]", "s22") \
    X(14, "v_cmp + v_cndmask pair", "v_cmp_lt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %3, vcc", "vcc") \
    X(15, "v_rcp_f32", "v_rcp_f32 %0, %0", "s22") \
    X(16, "v_sqrt_f32", "v_sqrt_f32 %0, %0", "s22") \
    X(17, "v_rsq_f32", "v_rsq_f32 %0, %0", "s22") \
    X(18, "v_exp_f32", "v_exp_f32 %0, %0", "s22") \
    X(19, "v_log_f32", "v_log_f32 %0, %0", "s22") \
    X(20, "v_sin_f32", "v_sin_f32 %0, %0", "s22") \
    X(21, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %2", "s22") \
    X(22, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %2", "s22") \
    X(23, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %2, %3", "s22") \
    X(24, "v_add_co_u32 vcc", "v_add_co_u32 %0, vcc, %0, %2", "vcc") \
    X(25, "v_readfirstlane_b32", "v_readfirstlane_b32 s20, %0", "s20") \
    X(26, "v_readlane_b32", "v_readlane_b32 s20, %0, 5", "s20") \
    X(27, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf", "s22") \
    X(28, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, -1, %0", "s22") \
    X(29, "v_perm_b32", "v_perm_b32 %0, %0, %2, %3", "s22") \
    X(30, "v_xor3? (v_xor_b32)", "v_xor_b32 %0, %0, %2", "s22") \
    X(31, "v_fma_f64 (pair regs)", "v_fma_f64 %1, %1, %4, %4", "s22") \
    X(32, "v_pk_fma_f32 (pair regs)", "v_pk_fma_f32 %1, %1, %4, %4", "s22") \
    X(33, "v_cvt_u32_f32", "v_cvt_u32_f32 %0, %0", "s22") \
    X(34, "v_floor_f32", "v_floor_f32 %0, %0", "s22") \
    X(35, "v_fract_f32", "v_fract_f32 %0, %0", "s22") \
    X(36, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %2", "s22") \
    X(37, "v_frexp_mant_f32", "v_frexp_mant_f32 %0, %0", "s22") \
    X(38, "v_mad_u64_u32", "v_mad_u64_u32 %1, vcc, %2, %3, %1", "vcc") \
    X(39, "v_div_scale+fmas+fixup (IEEE a/b)", "v_div_scale_f32 %0, vcc, %0, %2, %0\n v_div_fmas_f32 %0, %0, %2, %3\n v_div_fixup_f32 %0, %0, %2, %3", "vcc") \
    X(40, "s_nop (SALU only loop)", "s_nop 0", "s22") \
    X(41, "ds_bpermute_b32 + wait", "ds_bpermute_b32 %0, %2, %0\n s_waitcnt lgkmcnt(0)", "s22") \
    X(42, "v_cmpx_lt_f32 (writes exec)", "v_cmpx_lt_f32 exec, %3, %2", "s22") \
    X(43, "v_fma_f32 + s_add_u32 (pair)", "v_fma_f32 %0, %0, %2, %3\n s_add_u32 s22, s22, 1", "s22", "scc") \
    X(44, "v_fma_f32 + 2 s_add_u32", "v_fma_f32 %0, %0, %2, %3\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1", "s22", "s23", "scc") \
    X(45, "s_add_u32 alone", "s_add_u32 s22, s22, 1", "s22", "scc") \
    X(46, "v_fma_f32 + s_nop 0", "v_fma_f32 %0, %0, %2, %3\n s_nop 0", "s22") \
    X(47, "v_fma_f32 + ds_read_b32 (no wait)", "v_fma_f32 %0, %0, %2, %3\n ds_read_b64 %1, %2", "s22") \
    X(48, "v_fma_f32 + s_and_b64 + s_cbranch (not taken)", "v_fma_f32 %0, %0, %2, %3\n s_and_b64 s[20:21], exec, exec\n s_cbranch_scc0 1f\n1:", "s20", "s21", "scc") \
    X(49, "2 v_fma_f32 (dependent pair)", "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3", "s22")


template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float a, float b) {
    float x[8];
    double y[8];
    for (int k = 0; k < 8; ++k) { x[k] = threadIdx.x * 0.001f + k + 1.0f; y[k] = x[k]; }
    const double a2 = a;
    asm volatile("s_mov_b64 vcc, exec\n s_mov_b64 s[20:21], exec" : : : "vcc", "s20", "s21");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) {      // 64 statements per loop trip (the loop's own s_add / s_cmp / s_cbranch are 5 % of the stream), eight independent chains
            const int k = kk & 7;
            // operands: %0 = x[k] (32-bit, in/out), %1 = y[k] (register pair, in/out), %2 = a, %3 = b (32-bit inputs), %4 = a2 (pair input)
#define X(I, LABEL, ASM, ...) if (KIND == I) asm volatile(ASM : "+v"(x[k]), "+v"(y[k]) : "v"(a), "v"(b), "v"(a2) : __VA_ARGS__);
            KINDS(X)
#undef X
        }
    }
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) s += x[k] + (float)y[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, int n_cu, int khz) {
    const int iters = 1 << 10;
    std::printf("%-36s", name);
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = n_cu * waves_per_simd;      // 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU
        float* out;
        CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        k_rate<KIND><<<blocks, 256>>>(out, 8, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e0));
        k_rate<KIND><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double steps_per_simd = (double)iters * 64 * waves_per_simd;    // asm statements one SIMD executed
        std::printf("  %6.2f", ms * 1e-3 * khz * 1e3 / steps_per_simd);
        CHECK(hipFree(out));
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
    std::printf("\n");
}

int main() {
    std::setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    int khz = 0;
    CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    std::printf("%s, %d CUs; cycles (at the %d MHz max clock) one SIMD spends per statement, at 1 / 2 / 4 / 8 waves per SIMD;\n"
                "each statement runs in 8 independent chains per wave, 64 statements per loop trip (launch overhead of ~10 us is included)\n", p.name, p.multiProcessorCount, khz / 1000);
    std::printf("%-36s  %6s  %6s  %6s  %6s\n", "statement", "1", "2", "4", "8");
#define X(I, LABEL, ASM, ...) run<I>(LABEL, p.multiProcessorCount, khz);
    KINDS(X)
#undef X
    return 0;
}
