// Issue rates of the vector instructions the check-node rule is made of, on MI355X (gfx950): SIMD cycles
// per wave64 instruction at 1 .. 8 waves per SIMD, every CU busy.  These set the VALU-issue ceiling of the
// frame-resident decoder kernel (bench.py).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/valu_rates tools/valu_rates.hip
// Each wave runs REPS x 16 independent instructions of one kind (16 registers, each instruction depends only
// on the one 16 earlier) between two s_memtime stamps; a workgroup is 4 waves (one per SIMD), k workgroups
// per CU give k waves per SIMD.  Reported: median over waves of (cycles between the stamps) / (instructions
// of one wave) = cycles of wave lifetime per instruction; divided by the k waves that share the SIMD that is
// the SIMD's cycles per instruction when the waves overlap perfectly.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int REPS = 16384;

enum Op { FMA, PK_FMA, PK_MUL, PK_ADD, EXP, RCP, LOG, BFI, MIX, XOR, AND_OR, MUL, ADD, FMAC, CNDMASK, LDEXP, ADD_ABS, FMA_K, CMP_E64, CND_E64, CND_NEG, CMP_E32, CND_VCC_SET, CND_E64_VCC, MED3, AND2, ASHR, CMP_F32_E32, PAIR_E64, PAIR_VCC, XOR_LIT };

template <int OP>
__global__ void __launch_bounds__(256) rate_k(unsigned long long* cycles, unsigned long long* ticks, float* sink, float seed) {
    float x[16];
    v2f y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = seed + 0.001f * (threadIdx.x + i); y[i] = v2f{x[i], x[i] + 0.5f}; }
    const float c1 = 0.999f, c2 = 0.0001f;
    const v2f p1 = v2f{0.999f, 0.998f}, p2 = v2f{0.0001f, 0.0002f};
    __syncthreads();
    asm volatile("s_mov_b64 s[20:21], 0x5555" : : : "s20", "s21");  // the lane mask of the select cases
    asm volatile("s_mov_b64 vcc, 0x5555" : : : "vcc");
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
            if constexpr (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(p1), "v"(p2));
            if constexpr (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(p1));
            if constexpr (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(p2));
            if constexpr (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            if constexpr (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            if constexpr (OP == LOG) asm volatile("v_log_f32 %0, %0" : "+v"(x[i]));
            if constexpr (OP == BFI) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(c1), "v"(c2));
            if constexpr (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
            if constexpr (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
            if constexpr (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
            if constexpr (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c2));
            if constexpr (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
            if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(c1));
            if constexpr (OP == LDEXP) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(x[i]));
            if constexpr (OP == ADD_ABS) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(x[i]) : "v"(c2));
            if constexpr (OP == FMA_K) asm volatile("v_fma_f32 %0, %0, 0.5, 0.5" : "+v"(x[i]));
            if constexpr (OP == CMP_E64) asm volatile("v_cmp_gt_i32_e64 s[20:21], 0, %0" : : "v"(x[i]) : "s20", "s21");
            if constexpr (OP == CND_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[i]) : "v"(c1));
            if constexpr (OP == CND_NEG) asm volatile("v_cndmask_b32_e64 %0, %0, -%0, s[20:21]" : "+v"(x[i]));
            if constexpr (OP == CMP_E32) asm volatile("v_cmp_gt_i32_e32 vcc, 0, %0" : : "v"(x[i]) : "vcc");
            if constexpr (OP == CMP_F32_E32) asm volatile("v_cmp_gt_f32_e32 vcc, 0, %0" : : "v"(x[i]) : "vcc");
            if constexpr (OP == CND_VCC_SET) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(c1));
            if constexpr (OP == CND_E64_VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(c1));
            if constexpr (OP == MED3) asm volatile("v_med3_f32 %0, %0, -%0, %1" : "+v"(x[i]) : "v"(c1));
            if constexpr (OP == AND2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
            if constexpr (OP == ASHR) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(x[i]));
            if constexpr (OP == XOR_LIT) asm volatile("v_xor_b32 %0, 0x80000000, %0" : "+v"(x[i]));
            // a sign the way the check-node rule handles it: compare into a mask, mask product on the scalar unit, select
            if constexpr (OP == PAIR_E64) asm volatile("v_cmp_gt_i32_e64 s[22:23], 0, %0\n\ts_xor_b64 s[24:25], s[22:23], s[20:21]\n\tv_cndmask_b32_e64 %0, %0, -%0, s[24:25]" : "+v"(x[i]) : : "s22", "s23", "s24", "s25", "scc");
            if constexpr (OP == PAIR_VCC) asm volatile("v_cmp_gt_i32_e32 vcc, 0, %0\n\ts_xor_b64 vcc, vcc, s[20:21]\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(c1) : "vcc", "scc");
            if constexpr (OP == MIX) {
                // the mix of the check-node rule per pair of edges: 8 transcendentals, 12 packed, 2 bit operations
                if (i < 8) {
                    if ((i & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                    else if ((i & 3) == 3) asm volatile("v_log_f32 %0, %0" : "+v"(x[i]));
                    else asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                } else if (i < 14) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(p1), "v"(p2));
                } else {
                    asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(c1), "v"(c2));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + y[i].x + y[i].y;
    if (s == 1234.5678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        cycles[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
        ticks[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = r1 - r0;
    }
}

template <int OP>
void run(const char* name, int insts_per_rep, unsigned long long* d_cycles, unsigned long long* d_ticks, float* d_sink, int cus) {
    printf("%-34s", name);
    for (int k : {1, 2, 4, 8}) {
        const int blocks = cus * k;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(rate_k<OP>, dim3(blocks), dim3(256), 0, 0, d_cycles, d_ticks, d_sink, 1.0f);  // warm
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(rate_k<OP>, dim3(blocks), dim3(256), 0, 0, d_cycles, d_ticks, d_sink, 1.0f);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h((size_t)blocks * 4), t((size_t)blocks * 4);
        CK(hipMemcpy(h.data(), d_cycles, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        CK(hipMemcpy(t.data(), d_ticks, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> clk(h.size());
        for (size_t i = 0; i < h.size(); ++i) clk[i] = (double)h[i] / (double)t[i] * 0.1;  // GHz: s_memrealtime ticks at 100 MHz
        std::sort(h.begin(), h.end());
        std::sort(clk.begin(), clk.end());
        const double ghz = clk[clk.size() / 2];
        const double insts = (double)REPS * insts_per_rep;                  // per wave
        const double per_wave = (double)h[h.size() / 2] / insts;            // cycles of wave lifetime per instruction
        // chip-wide: SIMD cycles that passed during the launch per instruction a SIMD retired
        const double per_simd = (ms * 1e6 * ghz) / (insts * k);
        printf("  k=%d: %5.2f/wave %4.2f/SIMD %4.2f GHz", k, per_wave, per_simd, ghz);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs; k workgroups of 4 waves per CU: cycles of wave lifetime per wave64 instruction (median wave), SIMD cycles per\n"
           "instruction over the whole launch (wall time x clock / instructions per SIMD), in-kernel clock (s_memtime / s_memrealtime)\n",
           prop.name, cus);
    unsigned long long *d_cycles, *d_ticks;
    float* d_sink;
    CK(hipMalloc(&d_cycles, (size_t)cus * 8 * 4 * sizeof(unsigned long long)));
    CK(hipMalloc(&d_ticks, (size_t)cus * 8 * 4 * sizeof(unsigned long long)));
    CK(hipMalloc(&d_sink, 64));
    run<FMA>("v_fma_f32", 16, d_cycles, d_ticks, d_sink, cus);
    run<PK_FMA>("v_pk_fma_f32", 16, d_cycles, d_ticks, d_sink, cus);
    run<PK_MUL>("v_pk_mul_f32", 16, d_cycles, d_ticks, d_sink, cus);
    run<PK_ADD>("v_pk_add_f32", 16, d_cycles, d_ticks, d_sink, cus);
    run<BFI>("v_bfi_b32", 16, d_cycles, d_ticks, d_sink, cus);
    run<EXP>("v_exp_f32", 16, d_cycles, d_ticks, d_sink, cus);
    run<RCP>("v_rcp_f32", 16, d_cycles, d_ticks, d_sink, cus);
    run<LOG>("v_log_f32", 16, d_cycles, d_ticks, d_sink, cus);
    run<MIX>("mix 8 trans + 12 packed + 2 bfi", 22, d_cycles, d_ticks, d_sink, cus);
    run<XOR>("v_xor_b32 (VOP2)", 16, d_cycles, d_ticks, d_sink, cus);
    run<AND_OR>("v_and_or_b32", 16, d_cycles, d_ticks, d_sink, cus);
    run<MUL>("v_mul_f32 (VOP2)", 16, d_cycles, d_ticks, d_sink, cus);
    run<ADD>("v_add_f32 (VOP2)", 16, d_cycles, d_ticks, d_sink, cus);
    run<FMAC>("v_fmac_f32 (VOP2)", 16, d_cycles, d_ticks, d_sink, cus);
    run<CNDMASK>("v_cndmask_b32", 16, d_cycles, d_ticks, d_sink, cus);
    run<LDEXP>("v_ldexp_f32", 16, d_cycles, d_ticks, d_sink, cus);
    run<ADD_ABS>("v_add_f32 |x| (VOP3)", 16, d_cycles, d_ticks, d_sink, cus);
    run<FMA_K>("v_fma_f32 inline constants", 16, d_cycles, d_ticks, d_sink, cus);
    run<CMP_E64>("v_cmp_gt_i32_e64 -> sgpr pair", 16, d_cycles, d_ticks, d_sink, cus);
    run<CND_E64>("v_cndmask_b32_e64, sgpr mask", 16, d_cycles, d_ticks, d_sink, cus);
    run<CND_NEG>("v_cndmask_b32_e64 x, -x, sgpr mask", 16, d_cycles, d_ticks, d_sink, cus);
    run<CMP_E32>("v_cmp_gt_i32_e32 -> vcc", 16, d_cycles, d_ticks, d_sink, cus);
    run<CMP_F32_E32>("v_cmp_gt_f32_e32 -> vcc", 16, d_cycles, d_ticks, d_sink, cus);
    run<CND_VCC_SET>("v_cndmask_b32_e32, vcc set before", 16, d_cycles, d_ticks, d_sink, cus);
    run<CND_E64_VCC>("v_cndmask_b32_e64, vcc", 16, d_cycles, d_ticks, d_sink, cus);
    run<MED3>("v_med3_f32 x, -x, c", 16, d_cycles, d_ticks, d_sink, cus);
    run<AND2>("v_and_b32 (VOP2)", 16, d_cycles, d_ticks, d_sink, cus);
    run<ASHR>("v_ashrrev_i32 (VOP2)", 16, d_cycles, d_ticks, d_sink, cus);
    run<XOR_LIT>("v_xor_b32 literal", 16, d_cycles, d_ticks, d_sink, cus);
    run<PAIR_E64>("cmp_e64 + s_xor + cndmask_e64 (2 VALU)", 32, d_cycles, d_ticks, d_sink, cus);
    run<PAIR_VCC>("cmp_e32 + s_xor vcc + cndmask_e32 (2 VALU)", 32, d_cycles, d_ticks, d_sink, cus);
    return 0;
}
