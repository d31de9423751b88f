// valu_rate.hip -- issue rate of the VALU ops the Keccak round uses, on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Prints wave-instructions per cycle-equivalent: lanes/clk/SIMD assuming the
// clock reported by hipDeviceProp (and the measured wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t sb, uint32_t sc) {
    uint32_t a[16];
    const uint32_t t = threadIdx.x + blockIdx.x * 256;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = t * 2654435761u + i;
    uint32_t b = t ^ sb, c = t + sc;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define STEP(i)                                                                                  \
    if constexpr (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));            \
    if constexpr (OP == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c)); \
    if constexpr (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));    \
    if constexpr (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    if constexpr (OP == 4) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    if constexpr (OP == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    if constexpr (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));            \
    if constexpr (OP == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));                \
    if constexpr (OP == 8) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xd2" : "+v"(a[i]) : "v"(b), "s"(sc)); \
    if constexpr (OP == 9) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));           \
    if constexpr (OP == 10) asm volatile("v_alignbit_b32 %0, %0, %0, 9" : "+v"(a[i]));            \
    if constexpr (OP == 11) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    if constexpr (OP == 12) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));    \
    if constexpr (OP == 13) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));  \
    if constexpr (OP == 14) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    if constexpr (OP == 15) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    if constexpr (OP == 16) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));           \
    if constexpr (OP == 17) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));                 \
    if constexpr (OP == 18) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    if constexpr (OP == 19) asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b));        \
    if constexpr (OP == 20) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    if constexpr (OP == 21) asm volatile("v_alignbit_b32 %0, %1, %2, 7" : "=v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(b)); \
    if constexpr (OP == 22) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(c));        \
    if constexpr (OP == 23) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(a[i]) : "v"(c));    \
    if constexpr (OP == 24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));         \
    if constexpr (OP == 25) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    if constexpr (OP == 26) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));     \
    if constexpr (OP == 27) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(a[i]));                   \
    if constexpr (OP == 28) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));             \
    if constexpr (OP == 29) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));                   \
    if constexpr (OP == 30) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(b), "v"(c)); \
    if constexpr (OP == 31) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(STEP)
#undef STEP
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= a[i];
    out[t] = x;
}

// 64-bit ops on pairs
template <int OP>
__global__ void __launch_bounds__(256) k64(uint64_t* out, int iters, uint64_t sb) {
    uint64_t a[8];
    const uint32_t t = threadIdx.x + blockIdx.x * 256;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (uint64_t)t * 0x9E3779B97F4A7C15ull + i;
    uint64_t b = t ^ sb;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (OP == 0) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[i]));
                if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 2) asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            }
        }
    }
    uint64_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) x ^= a[i];
    out[t] = x;
}

template <class F>
double time_ms(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %.3f GHz\n", p.gcnArchName, cus, ghz);
    uint32_t* out;
    hipMalloc(&out, 256 * 8 * 256 * 8 * 8);
    const int iters = 2000;
    const char* names[] = {"v_xor_b32(e32)", "v_bitop3_b32 vvv", "v_alignbit_b32 vv,imm", "v_fma_f32", "v_and_or_b32",
                           "v_perm_b32", "v_add_u32", "v_mov_b32", "v_bitop3 v,v,s", "v_xor_b32 s,v",
                           "v_alignbit v,v(same),imm", "v_bfi_b32", "v_lshl_or_b32", "v_alignbyte_b32", "v_or3_b32",
                           "v_xad_u32", "v_and_b32", "v_lshlrev_b32", "v_add3_u32", "v_xor_b32_e64",
                           "v_alignbit v,v,v(shift)", "v_alignbit dst!=src0,imm", "v_lshlrev_b32 v,v", "v_alignbit v,v(same),v",
                           "v_mul_u32_u24", "v_mad_u32_u24", "v_lshl_add_u32", "v_bfe_u32", "v_mul_f32", "v_lshrrev_b32 imm",
                           "v_bitop3 dst!=src0", "v_mul_lo_u32"};
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
        printf("--- %d wave(s) per SIMD ---\n", wps);
#define RUN(OP)                                                                                   \
    {                                                                                             \
        double ms = time_ms([&] { hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u, 777u); }); \
        double winstr = (double)blocks * 4 * iters * 64.0;                                        \
        double per_simd_per_s = winstr / (cus * 4.0) / (ms * 1e-3);                               \
        printf("%-26s %8.3f ms  %6.2f cycles/wave-instr @%.2fGHz  (%.1f lanes/clk/SIMD)\n", names[OP], ms, \
               ghz * 1e9 / per_simd_per_s, ghz, 64.0 * per_simd_per_s / (ghz * 1e9));              \
    }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14)
        RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(28) RUN(29)
        RUN(30) RUN(31)
        const char* n64[] = {"v_lshlrev_b64", "v_pk_fma_f32", "v_pk_mov_b32"};
#define RUN64(OP)                                                                                 \
    {                                                                                             \
        double ms = time_ms([&] { hipLaunchKernelGGL(k64<OP>, dim3(blocks), dim3(256), 0, 0, (uint64_t*)out, iters, 99ull); }); \
        double winstr = (double)blocks * 4 * iters * 64.0;                                        \
        double per_simd_per_s = winstr / (cus * 4.0) / (ms * 1e-3);                               \
        printf("%-26s %8.3f ms  %6.2f cycles/wave-instr\n", n64[OP], ms, ghz * 1e9 / per_simd_per_s); \
    }
        RUN64(0) RUN64(1) RUN64(2)
    }
    return 0;
}
