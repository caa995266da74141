// Issue-rate microbenchmark for the integer VALU operations the pseudoalignment kernels are made of.
// Every kernel runs CHAIN repetitions of one operation on 8 independent registers per lane with WAVES waves per
// SIMD resident; the host prints SIMD cycles per wave-instruction (nominal clock from the device properties).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o valu_rates valu_rates.hip && ./valu_rates
// (generated table of operations: see the OPS list; results in profiles/r1/valu_rates_*.txt)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHAIN 16384

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t* out, uint32_t a, uint32_t b) {
    asm volatile("s_mov_b64 vcc, -1\n s_mov_b64 s[24:25], -1\n s_mov_b32 s22, 5" ::: "vcc", "s24", "s25", "s22");
    uint32_t x[8];
    uint64_t y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 2654435761u + i + a; y[i] = x[i] * 0x9E3779B97F4A7C15ull; }
    for (int it = 0; it < CHAIN / 8; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[i]) : : "vcc", "s24", "s25");
                if (OP == 1) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 2) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x[i]) : : "vcc", "s24", "s25");
                if (OP == 3) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 4) asm volatile("v_max_u32 %0, %1, %0" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 5) asm volatile("v_min_i32 %0, %1, %0" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 6) asm volatile("v_and_b32 %0, 0x12345678, %0" : "+v"(x[i]) : : "vcc", "s24", "s25");
                if (OP == 7) asm volatile("v_add_u32 %0, 7, %0" : "+v"(x[i]) : : "vcc", "s24", "s25");
                if (OP == 8) asm volatile("v_add_u32 %0, s22, %0" : "+v"(x[i]) : : "vcc", "s24", "s25");
                if (OP == 9) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 10) asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 11) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 12) asm volatile("v_xnor_b32 %0, %1, %0" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, s[24:25]" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 15) asm volatile("v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 16) asm volatile("v_cmp_lt_u32 s[24:25], %1, %0\n v_cndmask_b32 %0, %0, %1, s[24:25]" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 17) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(x[i]), "v"(b) : "vcc");
                if (OP == 18) asm volatile("v_add_u32 %0, %1, %0\n v_mul_lo_u32 %0, %1, %0" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 19) asm volatile("v_add_u32 %0, %1, %0\n v_xor_b32 %0, %1, %0\n v_mul_lo_u32 %0, %1, %0" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 20) asm volatile("v_xor_b32 %0, %1, %0\n v_lshrrev_b32 %0, %1, %0" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 21) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : : "vcc", "s24", "s25");
                if (OP == 22) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : : "vcc", "s24", "s25");
                if (OP == 23) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 24) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(y[i]));
                if (OP == 25) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i]) : : "vcc", "s24", "s25");
                if (OP == 26) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x[i]) : : "vcc", "s24", "s25");
                if (OP == 27) asm volatile("v_sad_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 28) asm volatile("v_med3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 29) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 30) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 31) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 32) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x[i]) : "v"(b) : "vcc", "s24", "s25");
                if (OP == 33) { x[i] &= 0xffc; asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x[i]) : : "memory"); }
                if (OP == 34) { x[i] &= 0xffc; asm volatile("ds_bpermute_b32 %0, %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x[i]) : : "memory"); }
                if (OP == 35) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(y[i]));
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
    for (int waves : {8, 2}) {
        run<0>("v_lshlrev_b32 const", d, waves);
        run<1>("v_lshlrev_b32 vgpr", d, waves);
        run<2>("v_lshrrev_b32 const", d, waves);
        run<3>("v_ashrrev_i32 vgpr", d, waves);
        run<4>("v_max_u32", d, waves);
        run<5>("v_min_i32", d, waves);
        run<6>("v_and_b32 literal", d, waves);
        run<7>("v_add_u32 const", d, waves);
        run<8>("v_add_u32 sgpr", d, waves);
        run<9>("v_add_co_u32", d, waves);
        run<10>("v_addc_co_u32", d, waves);
        run<11>("v_subrev_u32", d, waves);
        run<12>("v_xnor_b32", d, waves);
        run<13>("v_cndmask vcc (vcc=-1 set)", d, waves);
        run<14>("v_cndmask sgpr pair", d, waves);
        run<15>("cmp(vcc)+cndmask pair", d, waves);
        run<16>("cmp(sgpr)+cndmask pair", d, waves);
        run<17>("v_cmp_eq_u32 (vcc)", d, waves);
        run<18>("add+mul_lo pair", d, waves);
        run<19>("add+add+mul_lo triple", d, waves);
        run<20>("xor+lshr pair", d, waves);
        run<21>("v_mov_b32 dpp quad_perm", d, waves);
        run<22>("v_add_u32 dpp row_shr", d, waves);
        run<23>("v_add_u32 sdwa", d, waves);
        run<24>("v_lshrrev_b64", d, waves);
        run<25>("v_cvt_f32_u32", d, waves);
        run<26>("v_cvt_f32_ubyte0", d, waves);
        run<27>("v_sad_u32", d, waves);
        run<28>("v_med3_u32", d, waves);
        run<29>("v_min3_u32", d, waves);
        run<30>("v_or3_b32", d, waves);
        run<31>("v_lshl_or_b32", d, waves);
        run<32>("v_bfi_b32", d, waves);
        run<33>("ds_read_b32", d, waves);
        run<34>("ds_bpermute_b32", d, waves);
        run<35>("v_pk_add_f32", d, waves);
    }
    return 0;
}
