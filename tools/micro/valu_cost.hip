// Microbenchmark: issue cost of the vector instructions the node step is made of, on gfx950.  Every SIMD runs `waves`
// wavefronts; each issues `iters` x 16 INDEPENDENT instructions of one kind back to back (throughput), or a dependent
// chain of them (latency).  Cycles per wave-instruction per SIMD = time x clock / instructions per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_cost.hip -o gpurun_out/valu_cost ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int OP, bool DEP>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b, unsigned w)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    unsigned u = w + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { // v_fma_f32
            if (DEP) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b));) }
            else { REP16(asm volatile("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %4, %1, %2, %3" : "=v"(r0), "+v"(r1) : "v"(a), "v"(b), "v"(r2));) }
        }
        if (OP == 1) { // v_cvt_f32_ubyte1
            if (DEP) { REP16(asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r0));) }
            else { REP16(asm volatile("v_cvt_f32_ubyte1 %0, %2\n v_cvt_f32_ubyte2 %1, %2" : "=v"(r0), "=v"(r1) : "v"(u));) }
        }
        if (OP == 2) { // v_fma_mix_f32
            if (DEP) { REP16(asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b));) }
            else { REP16(asm volatile("v_fma_mix_f32 %0, %4, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r0), "=v"(r1) : "v"(a), "v"(b), "v"(u));) }
        }
        if (OP == 3) { // v_perm_b32
            if (DEP) { REP16(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u) : "v"(w), "v"(0x00050004u));) }
            else { REP16(asm volatile("v_perm_b32 %0, %2, %3, %4\n v_perm_b32 %1, %2, %3, %4" : "=v"(r0), "=v"(r1) : "v"(u), "v"(w), "v"(0x00050004u));) }
        }
        if (OP == 4) { // v_max3_f32
            if (DEP) { REP16(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b));) }
            else { REP16(asm volatile("v_max3_f32 %0, %2, %3, %4\n v_max3_f32 %1, %2, %3, %4" : "=v"(r0), "=v"(r1) : "v"(a), "v"(b), "v"(r2));) }
        }
        if (OP == 5) { // v_min_u32 (VOP2)
            if (DEP) { REP16(asm volatile("v_min_u32 %0, %0, %1" : "+v"(u) : "v"(w));) }
            else { REP16(asm volatile("v_min_u32 %0, %2, %3\n v_max_u32 %1, %2, %3" : "=v"(r0), "=v"(r1) : "v"(u), "v"(w));) }
        }
        if (OP == 6) { // v_cndmask_b32
            const unsigned long long cond = __ballot(threadIdx.x & 1);
            if (DEP) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "s"(cond));) }
            else { REP16(asm volatile("v_cndmask_b32 %0, %2, %3, %4\n v_cndmask_b32 %1, %3, %2, %4" : "=v"(r0), "=v"(r1) : "v"(a), "v"(b), "s"(cond));) }
        }
        if (OP == 7) { // v_cmp_le_f32 (writes vcc)
            REP16(asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %0" : : "v"(a), "v"(r1) : "vcc");)
        }
        if (OP == 8) { // v_and_or_b32 (VOP3)
            if (DEP) { REP16(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u) : "v"(w), "v"(3u));) }
            else { REP16(asm volatile("v_and_or_b32 %0, %2, %3, %4\n v_and_or_b32 %1, %2, %3, %4" : "=v"(r0), "=v"(r1) : "v"(u), "v"(w), "v"(3u));) }
        }
        if (OP == 9) { // v_pk_fma_f32
            REP16(asm volatile("v_pk_fma_f32 %0, %2, %3, %4\n v_pk_fma_f32 %1, %2, %3, %4" : "=v"(*(double*)&r0), "=v"(*(double*)&r2) : "v"(*(double*)&r4), "v"(*(double*)&r6), "v"(*(double*)&r4));)
        }
        if (OP == 10) { // v_mad_u64_u32
            unsigned long long q;
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3\n v_mad_u64_u32 %0, vcc, %2, %1, %3" : "=v"(q) : "v"(u), "v"(w), "v"(*(unsigned long long*)&r4) : "vcc");)
        }
        if (OP == 11) { // v_lshl_add_u64
            unsigned long long q;
            REP16(asm volatile("v_lshl_add_u64 %0, %1, 4, %2\n v_lshl_add_u64 %0, %2, 4, %1" : "=v"(q) : "v"(*(unsigned long long*)&r4), "v"(*(unsigned long long*)&r6));)
        }
        if (OP == 12) { // ds_write_b32 + nothing else (LDS store issue)
            REP16(asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256" : : "v"(threadIdx.x * 4u), "v"(r0) : "memory");)
        }
        if (OP == 13) { // v_mov_b32
            REP16(asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %2" : "=v"(r0), "=v"(r1) : "v"(a));)
        }
        if (OP == 14) { // v_add_u32
            REP16(asm volatile("v_add_u32 %0, %2, %3\n v_add_u32 %1, %2, %3" : "=v"(r0), "=v"(r1) : "v"(u), "v"(w));)
        }
        if (OP == 15) { // v_bfe_u32 (VOP3)
            REP16(asm volatile("v_bfe_u32 %0, %2, 8, 8\n v_bfe_u32 %1, %2, 16, 8" : "=v"(r0), "=v"(r1) : "v"(u));)
        }
        if (OP == 16) { REP16(asm volatile("v_max_f32 %0, %2, %3\n v_min_f32 %1, %2, %3" : "=v"(r0), "=v"(r1) : "v"(a), "v"(b));) }
        if (OP == 17) { REP16(asm volatile("v_mul_f32 %0, %2, %3\n v_add_f32 %1, %2, %3" : "=v"(r0), "=v"(r1) : "v"(a), "v"(b));) }
        if (OP == 18) { REP16(asm volatile("v_and_b32 %0, %2, %3\n v_or_b32 %1, %2, %3" : "=v"(r0), "=v"(r1) : "v"(u), "v"(w));) }
        if (OP == 19) { REP16(asm volatile("v_lshlrev_b32 %0, 3, %2\n v_lshrrev_b32 %1, 5, %2" : "=v"(r0), "=v"(r1) : "v"(u));) }
        if (OP == 20) { REP16(asm volatile("v_cvt_f32_u32 %0, %2\n v_cvt_f32_u32 %1, %2" : "=v"(r0), "=v"(r1) : "v"(u));) }
        if (OP == 21) { REP16(asm volatile("v_min3_u32 %0, %2, %3, %4\n v_med3_f32 %1, %2, %3, %4" : "=v"(r0), "=v"(r1) : "v"(u), "v"(w), "v"(a));) }
        if (OP == 22) { unsigned long long m0, m1; REP16(asm volatile("v_cmp_le_f32_e64 %0, %2, %3\n v_cmp_gt_u32_e64 %1, %2, %3" : "=s"(m0), "=s"(m1) : "v"(a), "v"(r1));) }
        if (OP == 23) { REP16(asm volatile("v_or_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_or_b32_sdwa %1, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r0), "=v"(r1) : "v"(w), "v"(u));) }
        if (OP == 24) { REP16(asm volatile("v_sub_f32 %0, %2, %3\n v_subrev_f32 %1, %2, %3" : "=v"(r0), "=v"(r1) : "v"(a), "v"(b));) }
        if (OP == 25) { REP16(asm volatile("v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3" : "+v"(r0), "+v"(r1) : "v"(a), "v"(b));) }
        if (OP == 26) { REP16(asm volatile("v_mul_f32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_f32_sdwa %1, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r0), "=v"(r1) : "v"(a), "v"(u));) }
        if (OP == 27) { REP16(asm volatile("v_xor_b32 %0, %2, %3\n v_sub_u32 %1, %2, %3" : "=v"(r0), "=v"(r1) : "v"(u), "v"(w));) }
        if (OP == 28) { REP16(asm volatile("v_lshl_or_b32 %0, %2, 3, %3\n v_add3_u32 %1, %2, %3, %3" : "=v"(r0), "=v"(r1) : "v"(u), "v"(w));) }
        if (OP == 29) { REP16(asm volatile("v_mul_lo_u32 %0, %2, %3\n v_mul_u32_u24 %1, %2, %3" : "=v"(r0), "=v"(r1) : "v"(u), "v"(w));) }
        if (OP == 30) { REP16(asm volatile("v_rcp_f32 %0, %2\n v_sqrt_f32 %1, %2" : "=v"(r0), "=v"(r1) : "v"(a));) }
        if (OP == 31) { REP16(asm volatile("v_mad_u32_u24 %0, %2, %3, %3\n v_mad_i32_i24 %1, %2, %3, %3" : "=v"(r0), "=v"(r1) : "v"(u), "v"(w));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long*>(out)[1] = t1 - t0;
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + u == 12345.678f) out[0] = r0;
}

template <int OP, bool DEP>
static void run(const char* name, float* out)
{
    const int iters = 2000;
    for (int waves : { 1, 8 }) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP, DEP><<<1024 * waves, 64>>>(out, 10, 1.5f, 2.5f, 77u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<OP, DEP><<<1024 * waves, 64>>>(out, iters, 1.5f, 2.5f, 77u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cyc[2];
        hipMemcpy(cyc, out, 16, hipMemcpyDeviceToHost);
        const double per = DEP ? 16.0 : 32.0; // instructions per loop body
        printf("%-18s %s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (wall), wave 0 saw %.2f cycles per own instruction\n", name,
               DEP ? "dependent  " : "independent", waves, ms * 1e-3 * 2.4e9 / (double(waves) * iters * per), double(cyc[1]) / (iters * per));
    }
}

int main()
{
    float* out;
    hipMalloc(&out, 64);
    run<0, false>("v_fma_f32", out); run<0, true>("v_fma_f32", out);
    run<1, false>("v_cvt_f32_ubyte", out); run<1, true>("v_cvt_f32_ubyte", out);
    run<2, false>("v_fma_mix_f32", out); run<2, true>("v_fma_mix_f32", out);
    run<3, false>("v_perm_b32", out); run<3, true>("v_perm_b32", out);
    run<4, false>("v_max3_f32", out); run<4, true>("v_max3_f32", out);
    run<5, false>("v_min/max_u32", out); run<5, true>("v_min_u32", out);
    run<6, false>("v_cndmask_b32", out); run<6, true>("v_cndmask_b32", out);
    run<7, false>("v_cmp_le_f32", out);
    run<8, false>("v_and_or_b32", out); run<8, true>("v_and_or_b32", out);
    run<9, false>("v_pk_fma_f32", out);
    run<10, false>("v_mad_u64_u32", out);
    run<11, false>("v_lshl_add_u64", out);
    run<12, false>("ds_write_b32", out);
    run<13, false>("v_mov_b32", out);
    run<14, false>("v_add_u32", out);
    run<15, false>("v_bfe_u32", out);
    run<16, false>("v_max/min_f32", out);
    run<17, false>("v_mul/add_f32", out);
    run<18, false>("v_and/or_b32", out);
    run<19, false>("v_lshl/lshr_b32", out);
    run<20, false>("v_cvt_f32_u32", out);
    run<21, false>("v_min3_u32/med3", out);
    run<22, false>("v_cmp e64 -> sgpr", out);
    run<23, false>("v_or_b32_sdwa", out);
    run<24, false>("v_sub_f32", out);
    run<25, false>("v_fmac_f32", out);
    run<26, false>("v_mul/add_f32_sdwa", out);
    run<27, false>("v_xor/sub_u32", out);
    run<28, false>("v_lshl_or/add3", out);
    run<29, false>("v_mul_lo/u24", out);
    run<30, false>("v_rcp/sqrt", out);
    run<31, false>("v_mad_u32_u24", out);
    return 0;
}
