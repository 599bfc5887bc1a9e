// tools/ubench.hip -- instruction-throughput microbenchmarks for gfx950
// (measurement aid for DESIGN.md; not part of the product library).
// Each kernel runs ITER iterations of 16 independent instructions of one kind
// per wave; grid = 256 CUs x (waves per SIMD) x 4 SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITER 1024

template <int KIND>
__global__ void ub(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f;
    int s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {   // 16 x v_mul_f32 (8 independent chains x2)
            asm volatile(
                "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        }
        else if (KIND == 1) {  // 16 x v_pk_mul_f32 on 4 register pairs
            asm volatile(
                "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6)
                : "v"(*(double *)&b0));
        }
        else if (KIND == 2) {  // 16 x s_add_u32 (4 chains)
            asm volatile(
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        }
        else if (KIND == 3) {  // 16 x v_max_i32_dpp, dependent chain with required nops
            asm volatile(
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                : "+v"(*(int *)&a0));
        }
        else if (KIND == 4) {  // 16 dependent v_sub_f32 (latency chain)
            asm volatile(
                "v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n"
                "v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n"
                "v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n"
                "v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n"
                : "+v"(a0) : "v"(b0));
        }
        else if (KIND == 5) {  // 8 x v_mul_f32 + 8 x s_add_u32 interleaved
            asm volatile(
                "v_mul_f32 %0, %0, %12\n s_add_u32 %8, %8, 1\n v_mul_f32 %1, %1, %12\n s_add_u32 %9, %9, 1\n"
                "v_mul_f32 %2, %2, %12\n s_add_u32 %10, %10, 1\n v_mul_f32 %3, %3, %12\n s_add_u32 %11, %11, 1\n"
                "v_mul_f32 %4, %4, %12\n s_add_u32 %8, %8, 1\n v_mul_f32 %5, %5, %12\n s_add_u32 %9, %9, 1\n"
                "v_mul_f32 %6, %6, %12\n s_add_u32 %10, %10, 1\n v_mul_f32 %7, %7, %12\n s_add_u32 %11, %11, 1\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                  "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                : "v"(b0) : "scc");
        }
        else if (KIND == 6) {  // 16 x v_readlane_b32 to distinct SGPRs
            asm volatile(
                "v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 7\n v_readlane_b32 %2, %4, 9\n v_readlane_b32 %3, %4, 11\n"
                "v_readlane_b32 %0, %5, 3\n v_readlane_b32 %1, %5, 7\n v_readlane_b32 %2, %5, 9\n v_readlane_b32 %3, %5, 11\n"
                "v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 7\n v_readlane_b32 %2, %4, 9\n v_readlane_b32 %3, %4, 11\n"
                "v_readlane_b32 %0, %5, 3\n v_readlane_b32 %1, %5, 7\n v_readlane_b32 %2, %5, 9\n v_readlane_b32 %3, %5, 11\n"
                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a0), "v"(a1));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0 + s1 + s2 + s3;
}

template <int KIND>
static void run(const char *name, float *out)
{
    for (int wps : {1, 2, 4, 8}) {          // waves per SIMD
        dim3 grid(256 * wps), block(256);   // 256 CUs x wps blocks of 4 waves
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        ub<KIND><<<grid, block>>>(out, 64);
        hipDeviceSynchronize();
        hipEventRecord(a);
        ub<KIND><<<grid, block>>>(out, ITER);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        double inst_per_simd = (double)ITER * 16 * wps;          // wave-instructions issued per SIMD
        double ns_per_inst = ms * 1e6 / inst_per_simd;
        fflush(stdout);
        printf("%-28s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n",
               name, wps, ms, ns_per_inst, ns_per_inst * 2.4);
    }
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_mul_f32 x16 indep", out);
    run<1>("v_pk_mul_f32 x16 (4 chains)", out);
    run<2>("s_add_u32 x16", out);
    run<3>("v_max_i32_dpp dep chain", out);
    run<4>("v_sub_f32 dep chain", out);
    run<5>("8 v_mul + 8 s_add mix", out);
    run<6>("v_readlane x16", out);
    fflush(stdout);
    return 0;
}
