// Issue-rate microbenchmark for the integer VALU operations the pseudoalignment kernels are made of.
// Every kernel runs CHAIN repetitions of one operation on 8 independent registers per lane with WAVES waves per
// SIMD resident; the host prints SIMD cycles per wave-instruction (nominal clock from the device properties).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o valu_rates valu_rates.hip && ./valu_rates
// (generated table of operations: see the OPS list; results in profiles/r1/valu_rates_*.txt)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHAIN 32768

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x[8];
    uint64_t y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 2654435761u + i + a; y[i] = x[i] * 0x9E3779B97F4A7C15ull; }
    for (int it = 0; it < CHAIN / 8; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 1) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 2) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 3) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 4) asm volatile("v_or_b32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 5) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[i]) );
                if (OP == 6) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 7) asm volatile("v_min_u32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 8) asm volatile("v_mov_b32 %0, %0" : "+v"(x[i]) );
                if (OP == 9) asm volatile("v_not_b32 %0, %0" : "+v"(x[i]) );
                if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b));
                if (OP == 11) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x[i]), "v"(b) : "vcc");
                if (OP == 12) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(x[i]), "v"(b) : "s20", "s21");
                if (OP == 13) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 14) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 15) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 16) asm volatile("v_mad_u32_u24 %0, %1, %0, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 17) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(y[i]) : "v"(b), "v"(x[i]) : "s20", "s21");
                if (OP == 18) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 19) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 20) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 21) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 22) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(x[i]) );
                if (OP == 23) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[i]) : "v"(b));
                if (OP == 24) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 25) asm volatile("v_bfrev_b32 %0, %0" : "+v"(x[i]) );
                if (OP == 26) asm volatile("v_ffbl_b32 %0, %0" : "+v"(x[i]) );
                if (OP == 27) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 28) asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 29) asm volatile("v_pk_mul_lo_u16 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 30) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 31) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(y[i]));
                if (OP == 32) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(y[i]));
                if (OP == 33) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(x[i]) : "s20", "s21");
                if (OP == 34) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(x[i]) : "s20", "s21");
                if (OP == 35) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) );
                if (OP == 36) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 37) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 38) asm volatile("s_add_u32 s20, s20, 3" : : "v"(x[i]) : "s20", "s21");
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= x[i] ^ (uint32_t)y[i] ^ (uint32_t)(y[i] >> 32);
    if (s == 0x12345) out[threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* d, int waves) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * waves;  // blocks of 4 waves: `waves` per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_rate<OP><<<blocks, 256>>>(d, 1, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<OP><<<blocks, 256>>>(d, 1, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * CHAIN * 8;
    const double simds = p.multiProcessorCount * 4.0;
    const double clk = p.clockRate * 1e3;
    printf("%-22s waves/SIMD %d  %8.3f ms  %6.2f SIMD cycles per wave-instruction\n", name, waves, ms,
           ms * 1e-3 * clk * simds / wave_instr);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 4096);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, nominal clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    for (int waves : {8, 2, 1}) {
        run<0>("v_add_u32", d, waves);
        run<1>("v_sub_u32", d, waves);
        run<2>("v_xor_b32", d, waves);
        run<3>("v_and_b32", d, waves);
        run<4>("v_or_b32", d, waves);
        run<5>("v_lshlrev_b32", d, waves);
        run<6>("v_lshrrev_b32", d, waves);
        run<7>("v_min_u32", d, waves);
        run<8>("v_mov_b32", d, waves);
        run<9>("v_not_b32", d, waves);
        run<10>("v_cndmask_b32", d, waves);
        run<11>("v_cmp_lt_u32 (vcc)", d, waves);
        run<12>("v_cmp_lt_u32 (sgpr)", d, waves);
        run<13>("v_mul_lo_u32", d, waves);
        run<14>("v_mul_hi_u32", d, waves);
        run<15>("v_mul_u32_u24", d, waves);
        run<16>("v_mad_u32_u24", d, waves);
        run<17>("v_mad_u64_u32", d, waves);
        run<18>("v_lshl_add_u32", d, waves);
        run<19>("v_add3_u32", d, waves);
        run<20>("v_and_or_b32", d, waves);
        run<21>("v_xad_u32", d, waves);
        run<22>("v_bfe_u32", d, waves);
        run<23>("v_alignbit_b32", d, waves);
        run<24>("v_perm_b32", d, waves);
        run<25>("v_bfrev_b32", d, waves);
        run<26>("v_ffbl_b32", d, waves);
        run<27>("v_bcnt_u32_b32", d, waves);
        run<28>("v_mbcnt_lo_u32_b32", d, waves);
        run<29>("v_pk_mul_lo_u16", d, waves);
        run<30>("v_pk_add_u16", d, waves);
        run<31>("v_lshlrev_b64", d, waves);
        run<32>("v_lshl_add_u64", d, waves);
        run<33>("v_readlane_b32", d, waves);
        run<34>("v_readfirstlane_b32", d, waves);
        run<35>("v_mov_b32 dpp row_shr", d, waves);
        run<36>("v_add_f32", d, waves);
        run<37>("v_fma_f32", d, waves);
        run<38>("s_add_u32 (scalar)", d, waves);
    }
    return 0;
}
