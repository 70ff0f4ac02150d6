// Measures what the vector ALU of a gfx950 SIMD issues per cycle, per instruction
// kind and for the instruction mix of the ray tracer's traversal loop, at a chosen
// number of resident waves per SIMD.  Built into raytracer_amd/libvalu_peak.so
// (a diagnostic beside the product library; bench.py prices the render kernel's
// measured VALU instruction count against the number this returns).
//
// Each wave runs `iters` trips of a loop whose body is kBody instructions of one
// kind (inline assembly, so the compiler can neither drop nor fuse them) on
// kChains independent register chains, and reports the cycles it took
// (s_memtime = shader clock).  instr per cycle per SIMD = waves per SIMD x
// iters x kBody / cycles of the slowest wave.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float F2 __attribute__((ext_vector_type(2)));

enum Kind {
    K_MUL = 0, K_ADD, K_FMA, K_PK_MUL, K_PK_ADD, K_PK_FMA, K_CNDMASK, K_MINMAX, K_MIN3, K_CMP_CNDMASK, K_DIV, K_SQRT, K_MOV, K_INT, K_MAX_I32, K_MAX3_I32, K_CMP_F32, K_CMP_I32, K_CNDMASK_S, K_MIX, K_COUNT
};

constexpr int kBody = 64;

#define REP4(X) X X X X
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)
#define REP64(X) REP16(X) REP16(X) REP16(X) REP16(X)

template <int KIND>
__global__ void __launch_bounds__(256) valuKernel(int iters, float seed, unsigned long long* cycles, float* sink)
{
    float a0 = seed + threadIdx.x, a1 = a0 * 1.25f, a2 = a0 + 3.f, a3 = a0 * 0.75f, b = 1.0000001f, c = 0.9999999f;
    F2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a1, a2 }, p3 = { a3, a0 }, q = { b, c };
    int i0 = threadIdx.x, i1 = threadIdx.x * 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (KIND == K_MUL) {
            REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if (KIND == K_ADD) {
            REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if (KIND == K_FMA) {
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (KIND == K_PK_MUL) {
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        } else if (KIND == K_PK_ADD) {
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        } else if (KIND == K_PK_FMA) {
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        } else if (KIND == K_CNDMASK) {
            asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_nop 1" : : "v"(a0), "v"(b) : "vcc");
            REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
        } else if (KIND == K_MINMAX) {
            REP16(asm volatile("v_max_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if (KIND == K_MIN3) {
            REP16(asm volatile("v_max3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (KIND == K_CMP_CNDMASK) {
            // the compare -> select pairs of the box test, with the wait states the compiler puts between them
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n s_nop 1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
        } else if (KIND == K_DIV) {
            // correctly rounded division as hipcc expands it (div_scale x2, rcp, 4 fma, mul, div_fmas, div_fixup: 11 instructions)
#pragma unroll
            for (int k = 0; k < 6; k++) {
                a0 = b / a0;
                asm volatile("" : "+v"(a0));
            }
        } else if (KIND == K_SQRT) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                a0 = __builtin_sqrtf(a0);
                asm volatile("" : "+v"(a0));
            }
        } else if (KIND == K_MOV) {
            REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == K_INT) {
            REP16(asm volatile("v_xor_b32 %0, %0, %1\n v_add_u32 %1, %1, %0\n v_lshlrev_b32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1" : "+v"(i0), "+v"(i1));)
        } else if (KIND == K_MAX_I32) {
            REP16(asm volatile("v_max_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_max_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if (KIND == K_MAX3_I32) {
            REP16(asm volatile("v_max3_i32 %0, %0, %4, %5\n v_min3_i32 %1, %1, %4, %5\n v_max3_u32 %2, %2, %4, %5\n v_min3_u32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (KIND == K_CMP_F32) {
            // compares alone, each into a scalar register pair of its own (nothing reads them)
            REP16(asm volatile("v_cmp_gt_f32 s[10:11], %0, %4\n v_cmp_lt_f32 s[12:13], %1, %4\n v_cmp_ngt_f32 s[14:15], %2, %4\n v_cmp_nlt_f32 vcc, %3, %4" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b) : "vcc", "s10", "s11", "s12", "s13", "s14", "s15");)
        } else if (KIND == K_CMP_I32) {
            REP16(asm volatile("v_cmp_gt_i32 s[10:11], %0, %4\n v_cmp_lt_i32 s[12:13], %1, %4\n v_cmp_gt_u32 s[14:15], %2, %4\n v_cmp_lt_u32 vcc, %3, %4" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b) : "vcc", "s10", "s11", "s12", "s13", "s14", "s15");)
        } else if (KIND == K_CNDMASK_S) {
            // selects on a mask in a scalar register pair that nothing rewrites, four independent chains
            asm volatile("v_cmp_gt_f32 s[10:11], %0, %1\n s_nop 4" : : "v"(a0), "v"(b) : "s10", "s11");
            REP16(asm volatile("v_cndmask_b32 %0, %0, %4, s[10:11]\n v_cndmask_b32 %1, %1, %4, s[10:11]\n v_cndmask_b32 %2, %2, %4, s[10:11]\n v_cndmask_b32 %3, %3, %4, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s10", "s11");)
        } else if (KIND == K_MIX) {
            // Proportions of the frame kernel's traversal loop as the headline frame runs it (the wave-shared
            // walk over origin frames; the loop's ISA weighted by the frame's counts of node visits, pretests,
            // whole triangle tests and pops, 147 M of the launch's 252 M vector instructions): per 64
            // instructions 11 multiplies, 7 adds / subs, 3 fused multiply-adds, 3 moves, 2 integer operations
            // (the fast kind), and 9 packed multiplies, 2 packed fused multiply-adds, 3 max, 6 min3 / max3,
            // 10 compares, 6 selects, a lane read and a part of the plane test's division (the slow kind).
            asm volatile(
                "v_pk_mul_f32 %4, %4, %10\n v_pk_mul_f32 %5, %5, %10\n v_pk_mul_f32 %6, %6, %10\n"
                "v_max_f32 %0, %0, %11\n v_max3_f32 %1, %1, %11, %0\n v_min3_f32 %2, %2, %11, %3\n"
                "v_cmp_ge_f32 s[10:11], %11, %1\n v_cmp_gt_f32 vcc, %1, %2\n"
                "v_mul_f32 %3, %3, %11\n v_sub_f32 %0, %0, %11\n"
                "v_cndmask_b32 %1, %1, %2, s[10:11]\n v_cmp_nlt_f32 s[12:13], %3, %1\n"
                "v_pk_mul_f32 %7, %7, %10\n v_pk_mul_f32 %4, %4, %10\n v_pk_mul_f32 %5, %5, %10\n"
                "v_max_f32 %0, %0, %11\n v_max3_f32 %1, %1, %11, %0\n v_min3_f32 %2, %2, %11, %3\n"
                "v_cmp_ge_f32 s[10:11], %11, %1\n v_cmp_gt_f32 vcc, %1, %2\n"
                "v_mul_f32 %3, %3, %11\n v_add_f32 %0, %0, %11\n"
                "v_cndmask_b32 %1, %1, %2, s[10:11]\n v_cmp_nlt_f32 s[14:15], %3, %1\n"
                "v_mov_b32 %0, %1\n v_lshl_add_u32 %8, %8, 2, %9\n v_cndmask_b32 %2, %2, %3, s[12:13]\n"
                "v_pk_mul_f32 %6, %6, %10\n v_pk_fma_f32 %7, %7, %10, %6\n v_pk_fma_f32 %4, %4, %10, %7\n"
                "v_mul_f32 %0, %0, %11\n v_fma_f32 %1, %1, %11, %0\n v_fma_f32 %2, %2, %11, %1\n"
                "v_pk_mul_f32 %5, %5, %10\n v_mul_f32 %0, %0, %11\n v_cmp_lt_f32 vcc, %3, %0\n"
                "v_mul_f32 %1, %1, %11\n v_mul_f32 %2, %2, %11\n v_mul_f32 %3, %3, %11\n v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n"
                "v_div_fixup_f32 %1, %1, %0, %2\n v_cmp_ge_f32 s[10:11], %11, %1\n v_cmp_lt_f32 s[12:13], %1, %3\n"
                "v_mul_f32 %0, %0, %11\n v_mul_f32 %2, %2, %11\n v_sub_f32 %3, %3, %11\n v_sub_f32 %1, %1, %11\n"
                "v_mul_f32 %0, %0, %11\n v_mul_f32 %2, %2, %11\n v_sub_f32 %3, %3, %0\n v_fma_f32 %1, %1, %11, %2\n"
                "v_max_f32 %0, %0, %11\n v_max3_f32 %1, %1, %11, %0\n v_min3_f32 %2, %2, %11, %3\n"
                "v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %3, %3, %2, vcc\n v_cndmask_b32 %0, %0, %1, s[14:15]\n"
                "v_readlane_b32 s16, %8, 3\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_add_u32 %9, %9, %8\n"
                "v_pk_mul_f32 %6, %6, %10\n v_cndmask_b32 %2, %2, %0, s[10:11]\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(i0), "+v"(i1)
                : "v"(q), "v"(b)
                : "vcc", "s10", "s11", "s12", "s13", "s14", "s15", "s16");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
        cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    // keep every chain alive
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + float(i0 + i1);
}

// VALU instructions one trip of the loop body issues, per kind
static int bodyInstructions(int kind)
{
    switch (kind) {
    case K_CMP_CNDMASK: return 64;     // 32 compares + 32 selects (the s_nop wait states are not VALU instructions)
    case K_DIV: return 6 * 11;
    case K_SQRT: return 66;            // four correctly rounded square roots as hipcc expands them (counted in the ISA)
    case K_MIX: return 64;             // vector instructions in the mixed body
    default: return kBody;
    }
}

template <int KIND>
static hipError_t launch(int grid, int iters, unsigned long long* cycles, float* sink, hipStream_t s)
{
    hipLaunchKernelGGL((valuKernel<KIND>), dim3(grid), dim3(256), 0, s, iters, 1.0f, cycles, sink);
    return hipGetLastError();
}

extern "C" {

const char* valu_peak_kind_name(int kind)
{
    static const char* names[K_COUNT] = { "v_mul_f32", "v_add_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_cndmask_b32", "v_min/max_f32", "v_min3/max3_f32",
        "v_cmp+v_cndmask", "IEEE divide (11 instr)", "IEEE sqrt", "v_mov_b32", "integer xor/add/shift", "v_min/max_i32/u32", "v_min3/max3_i32/u32", "v_cmp_f32 alone", "v_cmp_i32/u32 alone", "v_cndmask_b32 (sgpr mask)", "traversal-loop mix" };
    return kind >= 0 && kind < K_COUNT ? names[kind] : "?";
}
int valu_peak_kinds(void) { return K_COUNT; }
int valu_peak_mix_kind(void) { return K_MIX; }

// Runs kind `kind` with `wavesPerSimd` resident waves on every SIMD of the current device.
// out[0] = VALU instructions per cycle per SIMD (slowest wave), out[1] = the same from the
// median wave, out[2] = shader clock in MHz (cycles of the slowest wave / elapsed time),
// out[3] = chip-wide VALU wave-instructions per second.  Returns 0, or a HIP error code.
int valu_peak_run(int kind, int wavesPerSimd, int iters, double* out)
{
    if (kind < 0 || kind >= K_COUNT || wavesPerSimd < 1 || wavesPerSimd > 8 || iters < 1 || !out)
        return -1;
    int dev = 0;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess)
        e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess)
        return int(e);
    const int grid = prop.multiProcessorCount * wavesPerSimd; // 4-wave blocks: one wave per SIMD each
    unsigned long long* cycles = nullptr;
    float* sink = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    e = hipMalloc((void**)&cycles, size_t(grid) * 4 * sizeof(unsigned long long));
    if (e == hipSuccess)
        e = hipMalloc((void**)&sink, size_t(grid) * 256 * sizeof(float));
    if (e == hipSuccess)
        e = hipEventCreate(&ev0);
    if (e == hipSuccess)
        e = hipEventCreate(&ev1);
    float ms = 0.f;
    std::vector<unsigned long long> h(size_t(grid) * 4);
    for (int rep = 0; rep < 3 && e == hipSuccess; rep++) { // the last repetition counts (clocks ramped up)
        e = hipEventRecord(ev0, nullptr);
#define CASE(K) case K: if (e == hipSuccess) e = launch<K>(grid, iters, cycles, sink, nullptr); break;
        switch (kind) {
            CASE(K_MUL) CASE(K_ADD) CASE(K_FMA) CASE(K_PK_MUL) CASE(K_PK_ADD) CASE(K_PK_FMA) CASE(K_CNDMASK) CASE(K_MINMAX) CASE(K_MIN3) CASE(K_CMP_CNDMASK) CASE(K_DIV)
            CASE(K_SQRT) CASE(K_MOV) CASE(K_INT) CASE(K_MAX_I32) CASE(K_MAX3_I32) CASE(K_CMP_F32) CASE(K_CMP_I32) CASE(K_CNDMASK_S) CASE(K_MIX)
        }
#undef CASE
        if (e == hipSuccess)
            e = hipEventRecord(ev1, nullptr);
        if (e == hipSuccess)
            e = hipEventSynchronize(ev1);
        if (e == hipSuccess)
            e = hipEventElapsedTime(&ms, ev0, ev1);
    }
    if (e == hipSuccess)
        e = hipMemcpy(h.data(), cycles, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        std::sort(h.begin(), h.end());
        const double instr = double(iters) * bodyInstructions(kind);
        const double slow = double(h.back()), median = double(h[h.size() / 2]);
        out[0] = wavesPerSimd * instr / slow;
        out[1] = wavesPerSimd * instr / median;
        out[2] = slow / (double(ms) * 1e3);
        out[3] = double(grid) * 4.0 * instr / (double(ms) * 1e-3);
    }
    if (cycles)
        (void)hipFree(cycles);
    if (sink)
        (void)hipFree(sink);
    if (ev0)
        (void)hipEventDestroy(ev0);
    if (ev1)
        (void)hipEventDestroy(ev1);
    return int(e);
}

} // extern "C"
