// What does ONE wave per SIMD pay for an instruction issued between two MFMAs? Loop body: four v_mfma_f32_32x32x16_f16 on four AccVGPR
// accumulators (32 cycles of the matrix pipe each), each followed by N copies of one filler instruction (inline asm, registers rotated so
// that the copies are independent unless the variant says otherwise). 256 workgroups of 256 threads: every SIMD of the chip holds one
// wave, as in k_mlp_ss3. Prints shader cycles per MFMA slot for N = 0, 2, 4, 6, 8, 12 and the slope over the last two points.
//   hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

#define AGPRS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23", \
    "a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47", \
    "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71"

enum { FMA, FMA_DEP, CVT_PKRTZ, FMA_MIX, PK_MAX, ACC_READ, SIN, EXP, DS_READ128, NOP0, NOP1, MAX, PK_FMA, MOV, FRACT, PAIR_DEP, ACC_READ_DEP, DS_READ64, MIX_CVT, CVT_PK_RNE, CVT_F16, SPLIT1, SPLIT2, SPLIT2_RNE, FMA_DIST2, PERM, PK_MUL, ACC_READ_SPACED, MIXLO, MIXLOHI, SPLIT3, SPLIT3_ONE, NKIND };
static const char* kNames[NKIND] = {"v_fma_f32 (independent)", "v_fma_f32 (one dependent chain)", "v_cvt_pkrtz_f16_f32", "v_fma_mix_f32", "v_pk_max_i16",
    "v_accvgpr_read_b32 (idle AccVGPRs)", "v_sin_f32", "v_exp_f32", "ds_read_b128 (wait once per 4 MFMAs)", "s_nop 0", "s_nop 1", "v_max_f32", "v_pk_fma_f32",
    "v_mov_b32", "v_fract_f32", "v_fma -> v_max dependent pairs (counted as 2)", "v_accvgpr_read -> v_fma dependent pairs (counted as 2)",
    "ds_read_b64 (wait once per 4 MFMAs)", "2 v_fma_mix -> v_cvt_pkrtz triples (counted as 3)", "v_cvt_pk_f16_f32 (gfx950, RNE)", "v_cvt_f16_f32",
    "hi/lo split, ONE chain: cvt_pkrtz, 2 mix, nop, cvt_pkrtz (N/4 splits, counted as 4)", "hi/lo split, TWO chains interleaved (N/8 double splits, counted as 8)",
    "the same with v_cvt_pk_f16_f32", "v_fma_f32, two chains alternating (dependent at distance 2)", "v_perm_b32", "v_pk_mul_f32", "v_accvgpr_read, v_fma alternating (independent)", "v_fma_mixlo_f16 (independent)", "v_fma_mixlo_f16 -> v_fma_mixhi_f16 pairs into one register (counted as 2)",
    "hi/lo split with mixlo/mixhi, TWO chains interleaved: 2 cvt_pkrtz, 2 mixlo, 2 mixhi (N/6 double splits, counted as 6)", "hi/lo split with mixlo/mixhi, ONE chain: cvt_pkrtz, mixlo, mixhi (counted as 3)"};

static float* g_out; static unsigned long long* g_ticks;
template <int KIND, int E>
__device__ __forceinline__ void filler(float (&x)[8], unsigned (&u)[8], u4v (&q)[4], unsigned long long (&pk)[4], unsigned lds_addr, float s) {
    constexpr int i = E % 8, j = (E + 3) % 8;
    if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(s));
    if constexpr (KIND == FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[0]) : "v"(s));
    if constexpr (KIND == CVT_PKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[j]));
    if constexpr (KIND == FMA_MIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(x[i]) : "v"(u[i]), "v"(s), "v"(x[j]));
    if constexpr (KIND == PK_MAX) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(u[i]) : "v"(u[j]));
    if constexpr (KIND == ACC_READ) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x[i]) : "n"(64 + i));
    if constexpr (KIND == SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(x[i]));
    if constexpr (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
    if constexpr (KIND == DS_READ128) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(q[E % 4]) : "v"(lds_addr), "n"((E % 16) * 1024));
    if constexpr (KIND == DS_READ64) asm volatile("ds_read_b64 %0, %1 offset:%c2" : "=v"(pk[E % 4]) : "v"(lds_addr), "n"((E % 16) * 1024));
    if constexpr (KIND == NOP0) asm volatile("s_nop 0");
    if constexpr (KIND == NOP1) asm volatile("s_nop 1");
    if constexpr (KIND == MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
    if constexpr (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[E % 4]) : "v"(pk[(E + 1) % 4]));
    if constexpr (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(s));
    if constexpr (KIND == FRACT) asm volatile("v_fract_f32 %0, %0" : "+v"(x[i]));
    if constexpr (KIND == PAIR_DEP) { if constexpr (E % 2 == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(s)); else asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[(E - 1) % 8]) : "v"(s)); }
    if constexpr (KIND == ACC_READ_DEP) { if constexpr (E % 2 == 0) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x[i]) : "n"(64 + i)); else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(E - 1) % 8]) : "v"(s)); }
    if constexpr (KIND == MIX_CVT) {
        if constexpr (E % 3 == 0) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(x[0]) : "v"(u[i]), "v"(s), "v"(x[2]));
        if constexpr (E % 3 == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(x[1]) : "v"(u[i]), "v"(s), "v"(x[3]));
        if constexpr (E % 3 == 2) asm volatile("s_nop 0\n\tv_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[4]) : "v"(x[0]), "v"(x[1]));
    }
    if constexpr (KIND == CVT_PK_RNE) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[j]));
    if constexpr (KIND == CVT_F16) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[i]) : "v"(x[i]));
    if constexpr (KIND == SPLIT1) {
        if constexpr (E % 4 == 0) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[0]) : "v"(x[4]), "v"(x[5]));
        if constexpr (E % 4 == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(x[0]) : "v"(u[0]), "v"(s), "v"(x[4]));
        if constexpr (E % 4 == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(x[1]) : "v"(u[0]), "v"(s), "v"(x[5]));
        if constexpr (E % 4 == 3) asm volatile("s_nop 0\n\tv_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[1]) : "v"(x[0]), "v"(x[1]));
    }
    if constexpr (KIND == SPLIT2 || KIND == SPLIT2_RNE) {
        if constexpr (E % 8 == 0) { if constexpr (KIND == SPLIT2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[0]) : "v"(x[4]), "v"(x[5])); else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[0]) : "v"(x[4]), "v"(x[5])); }
        if constexpr (E % 8 == 1) { if constexpr (KIND == SPLIT2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[2]) : "v"(x[6]), "v"(x[7])); else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[2]) : "v"(x[6]), "v"(x[7])); }
        if constexpr (E % 8 == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(x[0]) : "v"(u[0]), "v"(s), "v"(x[4]));
        if constexpr (E % 8 == 3) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(x[2]) : "v"(u[2]), "v"(s), "v"(x[6]));
        if constexpr (E % 8 == 4) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(x[1]) : "v"(u[0]), "v"(s), "v"(x[5]));
        if constexpr (E % 8 == 5) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(x[3]) : "v"(u[2]), "v"(s), "v"(x[7]));
        if constexpr (E % 8 == 6) { if constexpr (KIND == SPLIT2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[1]) : "v"(x[0]), "v"(x[1])); else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[1]) : "v"(x[0]), "v"(x[1])); }
        if constexpr (E % 8 == 7) { if constexpr (KIND == SPLIT2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[3]) : "v"(x[2]), "v"(x[3])); else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[3]) : "v"(x[2]), "v"(x[3])); }
    }
    if constexpr (KIND == MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[j]), "v"(x[i]));
    if constexpr (KIND == MIXLOHI) {
        if constexpr (E % 2 == 0) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[j]), "v"(x[i]));
        else asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[(E - 1) % 8]) : "v"(u[(E + 2) % 8]), "v"(x[i]));
    }
    if constexpr (KIND == SPLIT3) {
        if constexpr (E % 6 == 0) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[0]) : "v"(x[4]), "v"(x[5]));
        if constexpr (E % 6 == 1) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[2]) : "v"(x[6]), "v"(x[7]));
        if constexpr (E % 6 == 2) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[1]) : "v"(u[0]), "v"(x[4]));
        if constexpr (E % 6 == 3) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[3]) : "v"(u[2]), "v"(x[6]));
        if constexpr (E % 6 == 4) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[1]) : "v"(u[0]), "v"(x[5]));
        if constexpr (E % 6 == 5) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[3]) : "v"(u[2]), "v"(x[7]));
    }
    if constexpr (KIND == SPLIT3_ONE) {
        if constexpr (E % 3 == 0) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[0]) : "v"(x[4]), "v"(x[5]));
        if constexpr (E % 3 == 1) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[1]) : "v"(u[0]), "v"(x[4]));
        if constexpr (E % 3 == 2) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[1]) : "v"(u[0]), "v"(x[5]));
    }
    if constexpr (KIND == FMA_DIST2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[E % 2]) : "v"(s));
    if constexpr (KIND == PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[j]), "v"(u[(E + 1) % 8]), "v"(u[(E + 5) % 8]));
    if constexpr (KIND == PK_MUL) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pk[E % 4]) : "v"(pk[(E + 1) % 4]), "v"(pk[(E + 2) % 4]));
    if constexpr (KIND == ACC_READ_SPACED) { if constexpr (E % 2 == 0) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x[i]) : "n"(64 + i)); else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(E + 4) % 8]) : "v"(s)); }
}
template <int KIND, int N, int E = 0>
__device__ __forceinline__ void fillers(float (&x)[8], unsigned (&u)[8], u4v (&q)[4], unsigned long long (&pk)[4], unsigned lds_addr, float s, int base) {
    if constexpr (E < N) { filler<KIND, E>(x, u, q, pk, lds_addr, s); fillers<KIND, N, E + 1>(x, u, q, pk, lds_addr, s, base); }
}

template <int KIND, int N, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void k(int iters, float* out, float s, unsigned long long* ticks) {
    extern __shared__ u4v lds[];
    for (int i = threadIdx.x; i < 4096; i += THREADS) lds[i] = u4v{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    float x[8]; unsigned u[8]; u4v q[4]; unsigned long long pk[4] = {1, 2, 3, 4};
    for (int e = 0; e < 8; ++e) { x[e] = threadIdx.x * 0.01f + e; u[e] = threadIdx.x + e; }
    for (int e = 0; e < 4; ++e) q[e] = u4v{1u, 2u, 3u, 4u};
    const unsigned lds_addr = (threadIdx.x & 63) * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define SLOT(T)                                                                                                                          \
        asm volatile("v_mfma_f32_32x32x16_f16 a[%c0:%c1], %2, %3, a[%c0:%c1]" :: "n"(16 * T), "n"(16 * T + 15), "v"(a), "v"(b) : AGPRS); \
        fillers<KIND, N>(x, u, q, pk, lds_addr, s, 0);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);
        SLOT(0) SLOT(1) SLOT(2) SLOT(3)
        if constexpr (KIND == DS_READ128 || KIND == DS_READ64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int e = 0; e < 8; ++e) r += x[e] + (float)u[e];
    for (int e = 0; e < 4; ++e) r += (float)q[e][0] + (float)q[e][2] + (float)pk[e];
    if (r == 12345.f) out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int N>
double run(int iters) {
    hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 65536, 0, 10, g_out, 0.999f, g_ticks);
    hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 65536, 0, iters, g_out, 0.999f, g_ticks);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), g_ticks, 1024 * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    return sum / 1024 / ((double)iters * 4);
}

// the MFMA's own issue cost by form: MK 0 = a[] accumulators (as above), 1 = srcC 0 (no accumulator read), 2 = architectural accumulators,
// 3 = v_mfma_f32_16x16x32_f16 on a[] (4 passes: 16 cycles of pipe), 4 = two 32x32x16 back to back per slot; fillers: independent v_fma_f32
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MK, int N>
__global__ __launch_bounds__(256) void k2(int iters, float* out, float s, unsigned long long* ticks) {
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    float x[8]; unsigned u[8]; u4v q[4]; unsigned long long pk[4] = {1, 2, 3, 4};
    for (int e = 0; e < 8; ++e) { x[e] = threadIdx.x * 0.01f + e; u[e] = threadIdx.x + e; }
    f32x16 c[4] = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define SLOT2(T)                                                                                                                         \
        if constexpr (MK == 0) asm volatile("v_mfma_f32_32x32x16_f16 a[%c0:%c1], %2, %3, a[%c0:%c1]" :: "n"(16 * T), "n"(16 * T + 15), "v"(a), "v"(b) : AGPRS); \
        if constexpr (MK == 1) asm volatile("v_mfma_f32_32x32x16_f16 a[%c0:%c1], %2, %3, 0" :: "n"(16 * T), "n"(16 * T + 15), "v"(a), "v"(b) : AGPRS); \
        if constexpr (MK == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[T]) : "v"(a), "v"(b));                    \
        if constexpr (MK == 3) asm volatile("v_mfma_f32_16x16x32_f16 a[%c0:%c1], %2, %3, a[%c0:%c1]" :: "n"(16 * T), "n"(16 * T + 3), "v"(a), "v"(b) : AGPRS); \
        if constexpr (MK == 4) { asm volatile("v_mfma_f32_32x32x16_f16 a[%c0:%c1], %2, %3, a[%c0:%c1]" :: "n"(16 * T), "n"(16 * T + 15), "v"(a), "v"(b) : AGPRS); \
                                 asm volatile("v_mfma_f32_32x32x16_f16 a[%c0:%c1], %2, %3, a[%c0:%c1]" :: "n"(16 * ((T + 2) % 4)), "n"(16 * ((T + 2) % 4) + 15), "v"(a), "v"(b) : AGPRS); } \
        fillers<FMA, N>(x, u, q, pk, 0u, s, 0);                                                                                          \
        __builtin_amdgcn_sched_barrier(0);
        SLOT2(0) SLOT2(1) SLOT2(2) SLOT2(3)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int e = 0; e < 8; ++e) r += x[e];
    for (int t = 0; t < 4; ++t) r += c[t][0];
    if (r == 12345.f) out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MK, int N>
double run2(int iters) {
    hipLaunchKernelGGL((k2<MK, N>), dim3(256), dim3(256), 0, 0, 10, g_out, 0.999f, g_ticks);
    hipLaunchKernelGGL((k2<MK, N>), dim3(256), dim3(256), 0, 0, iters, g_out, 0.999f, g_ticks);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), g_ticks, 1024 * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    return sum / 1024 / ((double)iters * 4);
}
template <int MK>
void mk(int iters, const char* name) {
    printf("%-58s N=0 %6.1f  2 %6.1f  4 %6.1f  6 %6.1f  8 %6.1f  12 %6.1f  16 %6.1f   cycles per slot, v_fma fillers\n", name,
           run2<MK, 0>(iters), run2<MK, 2>(iters), run2<MK, 4>(iters), run2<MK, 6>(iters), run2<MK, 8>(iters), run2<MK, 12>(iters), run2<MK, 16>(iters));
    fflush(stdout);
}

// two waves per SIMD (512-thread workgroups): cycles per MFMA slot of ONE wave; the SIMD retires two slots in that time
template <int KIND, int N>
double run_two(int iters) {
    hipLaunchKernelGGL((k<KIND, N, 512>), dim3(256), dim3(512), 65536, 0, 10, g_out, 0.999f, g_ticks);
    hipLaunchKernelGGL((k<KIND, N, 512>), dim3(256), dim3(512), 65536, 0, iters, g_out, 0.999f, g_ticks);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), g_ticks, 1024 * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    return sum / 1024 / ((double)iters * 4);
}
template <int KIND, int N, int THREADS>
void wall(int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, N, THREADS>), dim3(256), dim3(THREADS), 65536, 0, 10, g_out, 0.999f, g_ticks);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, N, THREADS>), dim3(256), dim3(THREADS), 65536, 0, iters, g_out, 0.999f, g_ticks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), g_ticks, 1024 * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double ticks = sum / 1024, per_simd = (double)iters * 4 * (THREADS / 256);
    printf("wall clock: %d wave(s) per SIMD, %2d v_fma per MFMA: %.2f ns per MFMA per SIMD (event time %.3f ms), %.1f ticks per MFMA per SIMD -> %.3f GHz tick rate\n",
           THREADS / 256, N, ms * 1e6 / per_simd, ms, ticks / ((double)iters * 4) / (THREADS / 256), ticks / (ms * 1e6));
    fflush(stdout);
}
template <int KIND>
void kind_two(int iters) {
    printf("TWO waves per SIMD, %-32s N=0 %6.1f  2 %6.1f  4 %6.1f  6 %6.1f  8 %6.1f  12 %6.1f  16 %6.1f   cycles per MFMA slot of one wave (the SIMD: half of it per MFMA)\n",
           kNames[KIND], run_two<KIND, 0>(iters), run_two<KIND, 2>(iters), run_two<KIND, 4>(iters), run_two<KIND, 6>(iters), run_two<KIND, 8>(iters), run_two<KIND, 12>(iters), run_two<KIND, 16>(iters));
    fflush(stdout);
}
template <int KIND>
void kind(int iters) {
    const double c0 = run<KIND, 0>(iters), c2 = run<KIND, 2>(iters), c4 = run<KIND, 4>(iters), c6 = run<KIND, 6>(iters), c8 = run<KIND, 8>(iters), c12 = run<KIND, 12>(iters), c16 = run<KIND, 16>(iters);
    printf("%-58s N=0 %6.1f  2 %6.1f  4 %6.1f  6 %6.1f  8 %6.1f  12 %6.1f  16 %6.1f   cycles per MFMA slot; slope 8->16: %5.2f cycles per instruction\n",
           kNames[KIND], c0, c2, c4, c6, c8, c12, c16, (c16 - c8) / 8);
    fflush(stdout);
}
int main(int argc, char** argv) {
    (void)hipMalloc(&g_out, 4096); (void)hipMalloc(&g_ticks, 1024 * 8);
    const int it = 4000;
    if (argc > 1) {   // round 5: the lo halves straight from v_fma_mixlo_f16 / v_fma_mixhi_f16 (three instructions per pair instead of four)
        kind<CVT_PKRTZ>(it); kind<FMA_MIX>(it); kind<MIXLO>(it); kind<MIXLOHI>(it); kind<SPLIT1>(it); kind<SPLIT3_ONE>(it); kind<SPLIT2>(it); kind<SPLIT3>(it);
        return 0;
    }
    kind<FMA>(it); kind<FMA_DEP>(it); kind<PAIR_DEP>(it); kind<MAX>(it); kind<MOV>(it); kind<PK_FMA>(it); kind<CVT_PKRTZ>(it); kind<FMA_MIX>(it); kind<MIX_CVT>(it); kind<PK_MAX>(it);
    kind<ACC_READ>(it); kind<ACC_READ_DEP>(it); kind<SIN>(it); kind<EXP>(it); kind<FRACT>(it); kind<DS_READ128>(it); kind<DS_READ64>(it); kind<NOP0>(it); kind<NOP1>(it);
    kind<CVT_PK_RNE>(it); kind<CVT_F16>(it); kind<SPLIT1>(it); kind<SPLIT2>(it); kind<SPLIT2_RNE>(it); kind<FMA_DIST2>(it); kind<PERM>(it); kind<PK_MUL>(it); kind<ACC_READ_SPACED>(it);
    wall<FMA, 0, 256>(40000); wall<FMA, 8, 256>(40000); wall<FMA, 0, 512>(40000); wall<FMA, 8, 512>(40000); wall<FMA, 16, 512>(40000);
    kind_two<FMA>(it); kind_two<CVT_PKRTZ>(it); kind_two<ACC_READ>(it); kind_two<SPLIT1>(it); kind_two<SPLIT2>(it);
    mk<0>(it, "MFMA 32x32x16, AccVGPR accumulators"); mk<1>(it, "MFMA 32x32x16, srcC = 0"); mk<2>(it, "MFMA 32x32x16, architectural accumulators");
    mk<3>(it, "MFMA 16x16x32 (16 cycles of pipe)"); mk<4>(it, "two MFMA 32x32x16 per slot (64 cycles of pipe)");
    return 0;
}
