// Microbenchmark 3: issue rate of single and packed FP32 VALU instructions on gfx950, and of the
// integer / conversion instructions the shadow-tap address arithmetic uses.  Every kernel runs the
// same number of instructions per lane in 8 independent dependency chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITER = 4096;
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float *out, float a, float b)
{
    float x[8];
    float2_t p[8];
    int n[8];
    unsigned long long m[8] = {};
    double dd[8];
    int sg[8] = {};
    const double da = a, db = b;
#pragma unroll
    for (int k = 0; k < 8; k++) { dd[k] = (double)(threadIdx.x + k) + 0.5; x[k] = (float)(threadIdx.x + k); p[k] = float2_t{x[k], x[k] + 1.0f}; n[k] = threadIdx.x + k; }
    const float2_t pa = {a, a}, pb = {b, b};
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pa), "v"(pb));
            if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
            if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pa));
            if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pa));
            if (MODE == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(n[k]) : "v"(n[0]));
            if (MODE == 6) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(n[k]) : "v"(x[k]));
            if (MODE == 7) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(n[k]) : "v"(x[k]));
            if (MODE == 8) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(n[k]) : "v"(n[0]));
            if (MODE == 9) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc" : : "v"(x[k]), "v"(a), "v"(n[k]) : "vcc");
            if (MODE == 10) asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(n[k]) : "v"(n[0]));
            if (MODE == 12) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[k]));
            if (MODE == 13) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[k]));
            if (MODE == 14) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[k]));
            if (MODE == 15) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
            if (MODE == 16) asm volatile("v_floor_f32 %0, %0" : "+v"(x[k]));
            if (MODE == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc");
            if (MODE == 18) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,1]" : "+v"(x[k]) : "v"(a), "v"(n[k]));
            if (MODE == 19) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(n[k]) : "v"(x[k]), "v"(a));
            if (MODE == 20) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
            if (MODE == 21) asm volatile("v_max3_f32 %0, |%0|, |%1|, |%2|" : "+v"(x[k]) : "v"(a), "v"(b));
            if (MODE == 22) asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m[k]) : "v"(x[k]), "v"(a));
            if (MODE == 23) asm volatile("v_lshl_add_u32 %0, %0, 5, %1" : "+v"(n[k]) : "v"(n[0]));
            if (MODE == 24) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dd[k]) : "v"(da), "v"(db));
            if (MODE == 25) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd[k]) : "v"(da));
            if (MODE == 26) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dd[k]) : "v"(da));
            if (MODE == 27) asm volatile("v_rcp_f64 %0, %0" : "+v"(dd[k]));
            if (MODE == 28) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dd[k]) : "v"(x[k]));
            if (MODE == 29) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x[k]) : "v"(dd[k]));
            if (MODE == 30) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sg[k]) : "v"(x[k]));
            if (MODE == 31) asm volatile("v_cmp_ge_f64 %0, %1, %2" : "=s"(m[k]) : "v"(dd[k]), "v"(da));
            if (MODE == 32) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x[k]));
            if (MODE == 33) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(x[k]));
            if (MODE == 34) asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(n[k]) : "v"(a));
            if (MODE == 35) asm volatile("v_log_f32 %0, %0" : "+v"(x[k]));
            if (MODE == 36) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k]));
            if (MODE == 37) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x[k]));
            if (MODE == 38) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(n[k]) : "v"(n[0]));
            if (MODE == 39) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x[k]) : "v"(n[k]), "s"(a));
            if (MODE == 40) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x[k]) : "v"(a), "v"(b));
            if (MODE == 11) asm volatile("v_cvt_f16_f32 %0, %0\n\tv_cvt_f32_f16 %0, %0" : "+v"(x[k]));
            if (MODE == 41) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[k]) : "v"(n[0]));
            if (MODE == 42) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(n[k]) : "v"(n[0]));
            if (MODE == 43) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[k]) : "v"(n[k]), "v"(n[0]) : "vcc");
            if (MODE == 44) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(m[k]));
            if (MODE == 45) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(dd[k]) : "v"(n[k]));
            if (MODE == 46) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[k]) : "v"(dd[k]));
            if (MODE == 47) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(n[k]) : "v"(n[0]));
            if (MODE == 48) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(dd[k]) : "v"(da) : "vcc");
            if (MODE == 49) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(dd[k]) : "v"(da), "v"(db));
            if (MODE == 50) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc" : "+v"(n[k]) : "v"(n[0]), "v"(sg[k]) : "vcc");
            if (MODE == 51) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(n[k]));
            if (MODE == 52) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(m[k]) : "v"(m[0]));
            if (MODE == 53) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[k]));
            if (MODE == 54) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[k]));
            if (MODE == 55) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[k]));
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) s += x[k] + p[k].x + p[k].y + (float)n[k] + (float)m[k] + (float)dd[k] + (float)sg[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, float *out, double per_instr_lane_ops, int instr_per_iter = 8)
{
    const int blocks = 256 * 16;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = (double)blocks * 4 * ITER * instr_per_iter;   // wave-level instructions
    const double per_simd_clk = wave_instr / 1024.0 / (ms * 1e-3) / 2.4e9;  // instructions per SIMD per 2.4 GHz clock
    printf("%-28s %8.3f ms  %6.3f wave-instr/SIMD/clk@2.4GHz  (%5.2f clk each)  %7.1f Gop/s lane ops\n", name, ms, per_simd_clk, 1.0 / per_simd_clk,
           wave_instr * 64 * per_instr_lane_ops / (ms * 1e-3) / 1e9);
}

int main()
{
    float *out;
    CHECK(hipMalloc(&out, 256 * 16 * 256 * 4));
    run<0>("v_fma_f32", out, 2);
    run<1>("v_pk_fma_f32", out, 4);
    run<2>("v_mul_f32", out, 1);
    run<3>("v_pk_mul_f32", out, 2);
    run<4>("v_pk_add_f32", out, 2);
    run<5>("v_mad_u32_u24", out, 1);
    run<6>("v_cvt_flr_i32_f32", out, 1);
    run<7>("v_alignbit_b32", out, 1);
    run<8>("v_add_lshl_u32", out, 1);
    run<9>("v_cmp_gt_f32 + v_addc", out, 1, 16);
    run<10>("v_med3_i32", out, 1);
    run<11>("cvt_f16_f32 + cvt_f32_f16", out, 1, 16);
    run<12>("v_rcp_f32", out, 1);
    run<13>("v_rsq_f32", out, 1);
    run<14>("v_sqrt_f32", out, 1);
    run<15>("v_max_f32", out, 1);
    run<16>("v_floor_f32", out, 1);
    run<17>("v_cndmask_b32", out, 1);
    run<18>("v_fma_mix_f32", out, 2);
    run<19>("v_cvt_pk_f16_f32", out, 2);
    run<20>("v_add_f32", out, 1);
    run<21>("v_max3_f32 |.|", out, 1);
    run<22>("v_cmp_gt_f32 -> sgpr", out, 1);
    run<23>("v_lshl_add_u32", out, 1);
    run<24>("v_fma_f64", out, 2);
    run<25>("v_mul_f64", out, 1);
    run<26>("v_add_f64", out, 1);
    run<27>("v_rcp_f64", out, 1);
    run<28>("v_cvt_f64_f32", out, 1);
    run<29>("v_cvt_f32_f64", out, 1);
    run<30>("v_readlane_b32", out, 1);
    run<31>("v_cmp_ge_f64 -> sgpr", out, 1);
    run<32>("v_cvt_f16_f32", out, 1);
    run<33>("v_cvt_f32_f16", out, 1);
    run<34>("v_pack_b32_f16", out, 1);
    run<35>("v_log_f32", out, 1);
    run<36>("v_exp_f32", out, 1);
    run<37>("v_cvt_u32_f32", out, 1);
    run<38>("v_lshl_or_b32", out, 1);
    run<39>("v_fma_mix_f32 f16hi,sgpr,f32", out, 2);
    run<40>("v_fma_f32 clamp", out, 2);
    run<41>("v_mul_lo_u32", out, 1);
    run<42>("v_mul_hi_u32", out, 1);
    run<43>("v_mad_u64_u32", out, 1);
    run<44>("v_lshlrev_b64", out, 1);
    run<45>("v_cvt_f64_i32", out, 1);
    run<46>("v_cvt_i32_f64", out, 1);
    run<47>("v_mul_u32_u24", out, 1);
    run<48>("v_div_scale_f64", out, 1);
    run<49>("v_div_fixup_f64", out, 1);
    run<50>("v_add_co_u32 + v_addc_co_u32", out, 1, 16);
    run<51>("v_mbcnt_lo_u32_b32", out, 1);
    run<52>("v_lshl_add_u64", out, 1);
    run<53>("v_cvt_f32_i32", out, 1);
    run<54>("v_cvt_i32_f32", out, 1);
    run<55>("v_mov_b32 dpp quad_perm", out, 1);
    return 0;
}
