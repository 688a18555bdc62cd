// microbench.hip -- gfx950 measurements behind DESIGN.md's kernel decisions (build: tools/build_microbench.sh;
// run on the GPU box: ./tools/microbench [test] > gpurun_out/microbench.json).
//
//   gather   random per-lane loads (1 / 8 bytes) whose cache lines are host-known: every load of a launch touches a
//            DISTINCT 64-byte line (or a distinct 128-byte line), so bytes-per-request of rocprofv3's FETCH_SIZE can
//            be calibrated for this access shape, and lines per clock per CU measured for
//            buffers that sit in L2 / Infinity Cache / HBM
//   valu     cycles per wave-instruction of the f64 / int ops the march step is made of
//   lds      random ds_read_u8 / b64 rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                                 \
        }                                                                            \
    } while (0)

// line index permutation: i -> (i * MULT + add) mod n_lines (n_lines a power of two, MULT odd => bijection)
#define MULT 0x9E3779B1u

// one load per (lane, k): global load index g = (k * total_lanes + lane_global); line = perm(g); every g distinct
template <int BYTES>
__global__ void __launch_bounds__(256) gather_kernel(const uint8_t* buf, uint32_t line_mask, int line_shift, int per_lane,
                                                     uint32_t total_lanes, uint64_t* sink) {
    const uint32_t lane = blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0;
#pragma unroll 8
    for (int k = 0; k < per_lane; k++) {
        const uint32_t g = (uint32_t)k * total_lanes + lane;
        const uint32_t line = (g * MULT) & line_mask;
        const uint8_t* p = buf + ((uint64_t)line << line_shift) + ((g >> 7) & 56);  // some 8-aligned byte of the line
        if (BYTES == 1) acc += *p;
        else acc += *reinterpret_cast<const uint64_t*>(p);
    }
    if (acc == 0x123456789abcdefull) sink[0] = acc;  // never true in practice: keeps the loads alive
}

// ---- VALU throughput: N independent chains per lane, R repetitions ----
#define OPS_LOOP 512
template <int OP>
__global__ void __launch_bounds__(256) valu_kernel(double* out, double seed, int reps, unsigned long long* cycles) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1.5, a2 = a0 + 2.25, a3 = a0 + 3.125, a4 = a0 + 4.5, a5 = a0 + 5.75,
           a6 = a0 + 6.5, a7 = a0 + 7.25;
    const double c = 1.0000001, d = 0.25;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int k = 0; k < OPS_LOOP / 8; k++) {
            if (OP == 0) {  // v_add_f64
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(d));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a1) : "v"(d));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "v"(d));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a3) : "v"(d));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a4) : "v"(d));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a5) : "v"(d));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a6) : "v"(d));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a7) : "v"(d));
            } else if (OP == 1) {  // v_floor_f64
                asm volatile("v_floor_f64 %0, %0" : "+v"(a0));
                asm volatile("v_floor_f64 %0, %0" : "+v"(a1));
                asm volatile("v_floor_f64 %0, %0" : "+v"(a2));
                asm volatile("v_floor_f64 %0, %0" : "+v"(a3));
                asm volatile("v_floor_f64 %0, %0" : "+v"(a4));
                asm volatile("v_floor_f64 %0, %0" : "+v"(a5));
                asm volatile("v_floor_f64 %0, %0" : "+v"(a6));
                asm volatile("v_floor_f64 %0, %0" : "+v"(a7));
            } else if (OP == 2) {  // v_cvt_i32_f64
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i0) : "v"(a0));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i1) : "v"(a1));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i2) : "v"(a2));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i3) : "v"(a3));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i4) : "v"(a4));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i5) : "v"(a5));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i6) : "v"(a6));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i7) : "v"(a7));
            } else if (OP == 3) {  // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(c));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a1) : "v"(c));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(c));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a3) : "v"(c));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a4) : "v"(c));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a5) : "v"(c));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a6) : "v"(c));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a7) : "v"(c));
            } else if (OP == 4) {  // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(d));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(c), "v"(d));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(c), "v"(d));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(c), "v"(d));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a4) : "v"(c), "v"(d));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a5) : "v"(c), "v"(d));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a6) : "v"(c), "v"(d));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a7) : "v"(c), "v"(d));
            } else if (OP == 5) {  // v_cmp_lt_f64 (to an SGPR pair)
                unsigned long long m;
                asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(a0), "v"(a1));
                asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(a1), "v"(a2));
                asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(a2), "v"(a3));
                asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(a3), "v"(a4));
                asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(a4), "v"(a5));
                asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(a5), "v"(a6));
                asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(a6), "v"(a7));
                asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(a7), "v"(a0));
            } else if (OP == 6) {  // v_rcp_f64
                asm volatile("v_rcp_f64 %0, %0" : "+v"(a0));
                asm volatile("v_rcp_f64 %0, %0" : "+v"(a1));
                asm volatile("v_rcp_f64 %0, %0" : "+v"(a2));
                asm volatile("v_rcp_f64 %0, %0" : "+v"(a3));
                asm volatile("v_rcp_f64 %0, %0" : "+v"(a4));
                asm volatile("v_rcp_f64 %0, %0" : "+v"(a5));
                asm volatile("v_rcp_f64 %0, %0" : "+v"(a6));
                asm volatile("v_rcp_f64 %0, %0" : "+v"(a7));
            } else if (OP == 7) {  // v_add_u32
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(i4));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i1) : "v"(i4));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i2) : "v"(i4));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i3) : "v"(i4));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i5) : "v"(i4));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i6) : "v"(i4));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i7) : "v"(i4));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(i4));
            } else if (OP == 8) {  // v_lshrrev_b64
                unsigned long long u0 = i0, u1 = i1, u2 = i2, u3 = i3;
                asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(u0) : "v"(i4));
                asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(u1) : "v"(i4));
                asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(u2) : "v"(i4));
                asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(u3) : "v"(i4));
                asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(u0) : "v"(i4));
                asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(u1) : "v"(i4));
                asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(u2) : "v"(i4));
                asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(u3) : "v"(i4));
                i0 = (int)u0; i1 = (int)u1; i2 = (int)u2; i3 = (int)u3;
            } else if (OP == 9) {  // v_cndmask_b32 (vcc)
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i0) : "v"(i4));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i1) : "v"(i4));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i2) : "v"(i4));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i3) : "v"(i4));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i5) : "v"(i4));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i6) : "v"(i4));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i7) : "v"(i4));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i0) : "v"(i4));
            } else if (OP == 10) {  // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i0) : "v"(i4));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i1) : "v"(i4));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i2) : "v"(i4));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i3) : "v"(i4));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i5) : "v"(i4));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i6) : "v"(i4));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i7) : "v"(i4));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i0) : "v"(i4));
            } else if (OP == 11) {  // v_rndne_f64
                asm volatile("v_rndne_f64 %0, %0" : "+v"(a0));
                asm volatile("v_rndne_f64 %0, %0" : "+v"(a1));
                asm volatile("v_rndne_f64 %0, %0" : "+v"(a2));
                asm volatile("v_rndne_f64 %0, %0" : "+v"(a3));
                asm volatile("v_rndne_f64 %0, %0" : "+v"(a4));
                asm volatile("v_rndne_f64 %0, %0" : "+v"(a5));
                asm volatile("v_rndne_f64 %0, %0" : "+v"(a6));
                asm volatile("v_rndne_f64 %0, %0" : "+v"(a7));
            } else if (OP >= 13) {
#define EIGHT(TEMPLATE)                                       \
    asm volatile(TEMPLATE : "+v"(i0) : "v"(i4), "s"(mask));   \
    asm volatile(TEMPLATE : "+v"(i1) : "v"(i4), "s"(mask));   \
    asm volatile(TEMPLATE : "+v"(i2) : "v"(i4), "s"(mask));   \
    asm volatile(TEMPLATE : "+v"(i3) : "v"(i4), "s"(mask));   \
    asm volatile(TEMPLATE : "+v"(i5) : "v"(i4), "s"(mask));   \
    asm volatile(TEMPLATE : "+v"(i6) : "v"(i4), "s"(mask));   \
    asm volatile(TEMPLATE : "+v"(i7) : "v"(i4), "s"(mask));   \
    asm volatile(TEMPLATE : "+v"(i0) : "v"(i4), "s"(mask));
                const unsigned long long mask = 0x5555aaaa3333ccccull;
                if (OP == 13) { EIGHT("v_cndmask_b32_e64 %0, %0, %1, %2") }
                else if (OP == 14) { EIGHT("v_mov_b32 %0, %1") }
                else if (OP == 15) { EIGHT("v_and_b32 %0, %0, %1") }
                else if (OP == 16) { EIGHT("v_lshl_add_u32 %0, %0, 2, %1") }
                else if (OP == 17) { EIGHT("v_or3_b32 %0, %0, %1, %1") }
                else if (OP == 18) { EIGHT("v_max3_u32 %0, %0, %1, %1") }
                else if (OP == 19) { EIGHT("v_bfi_b32 %0, %1, %0, %1") }
                else if (OP == 20) { EIGHT("v_mul_u32_u24 %0, %0, %1") }
                else if (OP == 21) { EIGHT("v_add3_u32 %0, %0, %1, %1") }
                else if (OP == 22) { EIGHT("v_ashrrev_i32 %0, 2, %0") }
                else if (OP == 23) {  // v_cmp_lt_u32 to an SGPR pair
                    unsigned long long m;
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(i0), "v"(i4));
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(i1), "v"(i4));
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(i2), "v"(i4));
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(i3), "v"(i4));
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(i5), "v"(i4));
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(i6), "v"(i4));
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(i7), "v"(i4));
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(i0), "v"(i4));
                } else if (OP == 24) {  // v_mad_u64_u32
                    unsigned long long u0 = i0, u1 = i1, u2 = i2, u3 = i3, c0;
                    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(u0), "=s"(c0) : "v"(i4), "v"(i5));
                    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(u1), "=s"(c0) : "v"(i4), "v"(i5));
                    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(u2), "=s"(c0) : "v"(i4), "v"(i5));
                    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(u3), "=s"(c0) : "v"(i4), "v"(i5));
                    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(u0), "=s"(c0) : "v"(i4), "v"(i5));
                    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(u1), "=s"(c0) : "v"(i4), "v"(i5));
                    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(u2), "=s"(c0) : "v"(i4), "v"(i5));
                    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(u3), "=s"(c0) : "v"(i4), "v"(i5));
                    i0 = (int)u0; i1 = (int)u1; i2 = (int)u2; i3 = (int)u3;
                } else if (OP == 25) {  // v_cndmask_b32 (vcc), vcc set once before
                    asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(i0), "v"(i4) : "vcc");
                    EIGHT("v_cndmask_b32 %0, %0, %1, vcc")
                } else if (OP == 27 || OP == 28) {  // v_add_f64 x 8 with the rounding mode switched and restored once / four times
#define RM(m_) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), " #m_)
                    RM(2);
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(d));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a1) : "v"(d));
                    if (OP == 28) { RM(0); }
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "v"(d));
                    if (OP == 28) { RM(2); }
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a3) : "v"(d));
                    RM(0);
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a4) : "v"(d));
                    if (OP == 28) { RM(2); }
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a5) : "v"(d));
                    if (OP == 28) { RM(0); }
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a6) : "v"(d));
                    if (OP == 28) { RM(2); }
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a7) : "v"(d));
                    if (OP == 28) { RM(0); }
#undef RM
                } else if (OP == 26) {  // v_add_f64 with an SGPR pair operand (the magic-add floor)
                    const double magic = 6755399441055744.0;
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "s"(magic));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a1) : "s"(magic));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "s"(magic));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a3) : "s"(magic));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a4) : "s"(magic));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a5) : "s"(magic));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a6) : "s"(magic));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a7) : "s"(magic));
                }
            } else if (OP == 12) {  // v_cvt_f64_i32
                asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a0) : "v"(i0));
                asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a1) : "v"(i1));
                asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a2) : "v"(i2));
                asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a3) : "v"(i3));
                asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a4) : "v"(i4));
                asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a5) : "v"(i5));
                asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a6) : "v"(i6));
                asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a7) : "v"(i7));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---- LDS random byte / b64 reads ----
template <int BYTES>
__global__ void __launch_bounds__(256) lds_kernel(uint64_t* out, int reps, unsigned long long* cycles) {
    __shared__ uint64_t s[4096];  // 32 KiB
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = i * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x;
    uint64_t acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
#pragma unroll 16
        for (int k = 0; k < 64; k++) {
            x = x * 1664525u + 1013904223u;
            if (BYTES == 1) acc += reinterpret_cast<const uint8_t*>(s)[x >> 17];
            else acc += s[x >> 20];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}


// ---- ray pool exchange (march_pool_kernel's pool_swap): a lane's 13 doubles + 10 words <-> a slot of the wave's pool ----
// MODE 0: ds_wrxchg_rtn (shipped: swaps in place, no spare registers); 1: ds_read + ds_write of the same words (field-major
// layout); 2: ds_read_b128 + ds_write_b128, fields paired so that a lane's 16 bytes are contiguous
typedef __attribute__((address_space(3))) unsigned long long lds64;
typedef __attribute__((address_space(3))) unsigned int lds32;
template <int MODE>
__global__ void __launch_bounds__(256) xchg_kernel(uint64_t* out, int reps, int active, unsigned long long* cycles) {
    constexpr int SLOTS = 48;
    __shared__ __attribute__((aligned(16))) unsigned long long pool[4][18 * SLOTS];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 18 * SLOTS; i += 64) pool[w][i] = i * 0x9E3779B97F4A7C15ull + blockIdx.x;
    __syncthreads();
    unsigned long long d[13];
    unsigned int u[10];
    for (int i = 0; i < 13; i++) d[i] = threadIdx.x * 77ull + i;
    for (int i = 0; i < 10; i++) u[i] = threadIdx.x * 13u + i;
    lds64* base = (lds64*)&pool[w][0];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        const int s = (lane + r * 7) % SLOTS;
        if (lane < active) {
            if (MODE == 0) {
                lds64* q = base + s;
#pragma unroll
                for (int i = 0; i < 13; i++) d[i] = __hip_atomic_exchange(q + i * SLOTS, d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                lds32* v = (lds32*)(base + 13 * SLOTS) + s;
#pragma unroll
                for (int i = 0; i < 10; i++) u[i] = __hip_atomic_exchange(v + i * SLOTS, u[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            } else if (MODE == 1) {
                lds64* q = base + s;
#pragma unroll
                for (int i = 0; i < 13; i++) { unsigned long long t = q[i * SLOTS]; q[i * SLOTS] = d[i]; d[i] = t; }
                lds32* v = (lds32*)(base + 13 * SLOTS) + s;
#pragma unroll
                for (int i = 0; i < 10; i++) { unsigned int t = v[i * SLOTS]; v[i * SLOTS] = u[i]; u[i] = t; }
            } else {
                typedef unsigned int v4u __attribute__((ext_vector_type(4)));
                typedef __attribute__((address_space(3))) v4u lds128;
                lds128* q = (lds128*)base + s;   // pair p of slot s at (p * SLOTS + s) * 16
                unsigned int* dw = reinterpret_cast<unsigned int*>(d);
#pragma unroll
                for (int p = 0; p < 9; p++) {
                    v4u mine;
                    if (p < 6) mine = (v4u){dw[4 * p], dw[4 * p + 1], dw[4 * p + 2], dw[4 * p + 3]};
                    else if (p == 6) mine = (v4u){dw[24], dw[25], u[0], u[1]};
                    else mine = (v4u){u[4 * (p - 7) + 2], u[4 * (p - 7) + 3], u[4 * (p - 7) + 4], u[4 * (p - 7) + 5]};
                    const v4u t = q[p * SLOTS];
                    q[p * SLOTS] = mine;
                    if (p < 6) { dw[4 * p] = t.x; dw[4 * p + 1] = t.y; dw[4 * p + 2] = t.z; dw[4 * p + 3] = t.w; }
                    else if (p == 6) { dw[24] = t.x; dw[25] = t.y; u[0] = t.z; u[1] = t.w; }
                    else { u[4 * (p - 7) + 2] = t.x; u[4 * (p - 7) + 3] = t.y; u[4 * (p - 7) + 4] = t.z; u[4 * (p - 7) + 5] = t.w; }
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long acc = 0;
    for (int i = 0; i < 13; i++) acc += d[i];
    for (int i = 0; i < 10; i++) acc += u[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

static double run_gather(int bytes, const uint8_t* buf, uint64_t buf_bytes, int line_bytes, int per_lane, int blocks,
                         uint64_t* sink, int reps, const char* label) {
    int shift = line_bytes == 64 ? 6 : 7;
    uint64_t n_lines = buf_bytes >> shift;
    uint32_t mask = (uint32_t)(n_lines - 1);
    uint32_t lanes = (uint32_t)blocks * 256;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    // warm-up
    if (bytes == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, mask, shift, per_lane, lanes, sink);
    else hipLaunchKernelGGL(gather_kernel<8>, dim3(blocks), dim3(256), 0, 0, buf, mask, shift, per_lane, lanes, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; r++) {
        if (bytes == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, mask, shift, per_lane, lanes, sink);
        else hipLaunchKernelGGL(gather_kernel<8>, dim3(blocks), dim3(256), 0, 0, buf, mask, shift, per_lane, lanes, sink);
    }
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    double loads = (double)lanes * per_lane;
    bool distinct = loads <= (double)n_lines;
    printf("{\"test\": \"gather\", \"label\": \"%s\", \"load_bytes\": %d, \"buffer_MiB\": %.1f, \"line_bytes\": %d, "
           "\"loads_per_launch\": %.0f, \"all_lines_distinct\": %s, \"ms\": %.4f, \"Gloads_per_s\": %.2f, "
           "\"loads_per_clk_per_CU_at_2.4GHz\": %.3f, \"line_bytes_per_launch\": %.0f}\n",
           label, bytes, buf_bytes / 1048576.0, line_bytes, loads, distinct ? "true" : "false", ms, loads / ms / 1e6,
           loads / (ms * 1e-3) / 256.0 / 2.4e9, loads * line_bytes);
    fflush(stdout);
    return ms;
}

int main(int argc, char** argv) {
    std::string which = argc > 1 ? argv[1] : "all";
    CHECK(hipSetDevice(0));
    uint64_t* sink;
    CHECK(hipMalloc(&sink, 4096));
    if (which == "all" || which == "gather" || which == "calib") {
        const uint64_t GiB = 1ull << 30;
        uint8_t* buf;
        CHECK(hipMalloc(&buf, GiB));
        CHECK(hipMemset(buf, 1, GiB));
        CHECK(hipDeviceSynchronize());
        if (which == "calib") {
            // the launches rocprofv3 --pmc FETCH_SIZE is read on: exactly one launch of each shape, every load a distinct line
            //   A: 2^24 one-byte loads, each in a distinct 64-B line of 1 GiB (all 2^24 lines)   -> true bytes 1 GiB @64 B
            //   B: 2^23 one-byte loads, each in a distinct 128-B line of 1 GiB (all 2^23 lines)
            //   C: 2^24 eight-byte loads, distinct 64-B lines
            run_gather(1, buf, GiB, 64, 16, 4096, sink, 1, "calib_A_u8_64B_lines");
            run_gather(1, buf, GiB, 128, 8, 4096, sink, 1, "calib_B_u8_128B_lines");
            run_gather(8, buf, GiB, 64, 16, 4096, sink, 1, "calib_C_u64_64B_lines");
        } else {
            // rates: buffer in L2 (1 MiB), Infinity Cache (64 MiB), HBM (1 GiB); 1024 blocks = 4 per CU (16 waves/CU)
            for (int bytes : {1, 8}) {
                run_gather(bytes, buf, 1ull << 20, 64, 256, 1024, sink, 5, "L2_1MiB");
                run_gather(bytes, buf, 64ull << 20, 64, 256, 1024, sink, 5, "IC_64MiB");
                run_gather(bytes, buf, 128ull << 20, 64, 256, 1024, sink, 5, "IC_128MiB");
                run_gather(bytes, buf, GiB, 64, 256, 1024, sink, 5, "HBM_1GiB");
            }
            run_gather(1, buf, 1ull << 20, 64, 256, 2048, sink, 5, "L2_1MiB_8blocks_per_CU");
            run_gather(1, buf, 32768, 64, 256, 1024, sink, 5, "L1_32KiB");
        }
        CHECK(hipFree(buf));
    }
    if (which == "all" || which == "valu") {
        const char* names[] = {"v_add_f64", "v_floor_f64", "v_cvt_i32_f64", "v_mul_f64", "v_fma_f64", "v_cmp_lt_f64", "v_rcp_f64",
                               "v_add_u32", "v_lshrrev_b64", "v_cndmask_b32", "v_mul_lo_u32", "v_rndne_f64", "v_cvt_f64_i32",
                               "v_cndmask_b32_e64(sgpr)", "v_mov_b32", "v_and_b32", "v_lshl_add_u32", "v_or3_b32", "v_max3_u32",
                               "v_bfi_b32", "v_mul_u32_u24", "v_add3_u32", "v_ashrrev_i32", "v_cmp_lt_u32", "v_mad_u64_u32",
                               "v_cndmask_b32(vcc set)", "v_add_f64(sgpr)", "v_add_f64 x8 + 2 s_setreg(round mode)",
                               "v_add_f64 x8 + 8 s_setreg(round mode)"};
        double* out;
        unsigned long long* cyc;
        const int blocks = 1024;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
        CHECK(hipMalloc(&out, blocks * 256 * 8));
        CHECK(hipMalloc(&cyc, blocks * 8));
        std::vector<unsigned long long> h(blocks);
        for (int op = 0; op < 29; op++) {
            for (int wpb : {1024, 256}) {  // 4 waves per SIMD, 1 wave per SIMD
                const int reps = 64;
#define LAUNCH(OPN) hipLaunchKernelGGL(valu_kernel<OPN>, dim3(wpb), dim3(256), 0, 0, out, 1.25, reps, cyc)
                for (int rr = 0; rr < 2; rr++) {
                    switch (op) {
                        case 0: LAUNCH(0); break; case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break;
                        case 4: LAUNCH(4); break; case 5: LAUNCH(5); break; case 6: LAUNCH(6); break; case 7: LAUNCH(7); break;
                        case 8: LAUNCH(8); break; case 9: LAUNCH(9); break; case 10: LAUNCH(10); break; case 11: LAUNCH(11); break;
                        case 12: LAUNCH(12); break; case 13: LAUNCH(13); break; case 14: LAUNCH(14); break; case 15: LAUNCH(15); break;
                        case 16: LAUNCH(16); break; case 17: LAUNCH(17); break; case 18: LAUNCH(18); break; case 19: LAUNCH(19); break;
                        case 20: LAUNCH(20); break; case 21: LAUNCH(21); break; case 22: LAUNCH(22); break; case 23: LAUNCH(23); break;
                        case 24: LAUNCH(24); break; case 25: LAUNCH(25); break; case 26: LAUNCH(26); break; case 27: LAUNCH(27); break;
                        case 28: LAUNCH(28); break;
                    }
                }
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(h.data(), cyc, wpb * 8, hipMemcpyDeviceToHost));
                double s = 0;
                for (int i = 0; i < wpb; i++) s += (double)h[i];
                s /= wpb;
                const double per = s / ((double)reps * OPS_LOOP);
                const int waves_per_simd = wpb == 1024 ? 4 : 1;
                printf("{\"test\": \"valu\", \"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_instr_seen_by_one_wave\": %.2f, "
                       "\"simd_cycles_per_wave_instr\": %.2f}\n", names[op], waves_per_simd, per, per / waves_per_simd);
                fflush(stdout);
            }
        }
        CHECK(hipFree(out));
        CHECK(hipFree(cyc));
    }
    if (which == "all" || which == "lds") {
        uint64_t* out;
        unsigned long long* cyc;
        const int blocks = 1024;
        CHECK(hipMalloc(&out, blocks * 256 * 8));
        CHECK(hipMalloc(&cyc, blocks * 8));
        std::vector<unsigned long long> h(blocks);
        for (int bytes : {1, 8}) {
            const int reps = 64;
            for (int rr = 0; rr < 2; rr++) {
                if (bytes == 1) hipLaunchKernelGGL(lds_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, reps, cyc);
                else hipLaunchKernelGGL(lds_kernel<8>, dim3(blocks), dim3(256), 0, 0, out, reps, cyc);
            }
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
            double s = 0;
            for (int i = 0; i < blocks; i++) s += (double)h[i];
            s /= blocks;
            // 16 waves per CU each issuing reps*64 reads in s cycles
            printf("{\"test\": \"lds\", \"read_bytes\": %d, \"cycles_per_read_seen_by_one_wave\": %.2f, "
                   "\"CU_cycles_per_wave_read\": %.2f}\n", bytes, s / (reps * 64.0), s / (reps * 64.0) / 16.0);
            fflush(stdout);
        }
    }
    if (which == "all" || which == "xchg") {
        uint64_t* out;
        unsigned long long* cyc;
        CHECK(hipMalloc(&out, 1024 * 256 * 8));
        CHECK(hipMalloc(&cyc, 1024 * 8));
        std::vector<unsigned long long> h(1024);
        const char* names[] = {"ds_wrxchg_rtn b64 x13 + b32 x10 (shipped)", "ds_read + ds_write, same words", "ds_read_b128 + ds_write_b128 x9"};
        for (int mode = 0; mode < 3; mode++)
            for (int blocks : {1024, 256})
                for (int active : {28, 48}) {
                    const int reps = 512;
                    for (int rr = 0; rr < 2; rr++) {
                        if (mode == 0) hipLaunchKernelGGL(xchg_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, reps, active, cyc);
                        else if (mode == 1) hipLaunchKernelGGL(xchg_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, reps, active, cyc);
                        else hipLaunchKernelGGL(xchg_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, reps, active, cyc);
                    }
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
                    double sum = 0;
                    for (int i = 0; i < blocks; i++) sum += (double)h[i];
                    printf("{\"test\": \"xchg\", \"mode\": \"%s\", \"waves_per_CU\": %d, \"lanes\": %d, \"memtime_ticks_per_exchange_seen_by_one_wave\": %.1f}\n",
                           names[mode], blocks == 1024 ? 16 : 4, active, sum / blocks / reps);
                    fflush(stdout);
                }
    }
    return 0;
}
