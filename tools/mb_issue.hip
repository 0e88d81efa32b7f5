// mb_issue.hip -- VALU issue cost of the integer instructions the sketch kernels are made of, at 1 / 2 / 4 / 8 waves per SIMD (gfx950).
// What bench.py's `roofline.secondary` (VALU issue bound) is priced on: its peak assumes that a wave64 VALU instruction occupies
// its SIMD for 4 shader cycles however many waves share the SIMD (the microarch guide's constants table reads "2 cyc (SIMD-32); one
// wave alone: 4" for v_fma_f32 -- if 2 applied to these integer instructions with >= 2 waves resident, the bound would be half).
//
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mb_issue.hip -o /tmp/mb_issue && /tmp/mb_issue
// Method: every wave runs iters x 32 INDEPENDENT instructions of one kind (eight registers, no dependency between neighbours; each
// register's own chain is 8 instructions apart).  Every launch is timed twice, with ITERS_A and ITERS_B iterations, and the cost is
// taken from the DIFFERENCE of the two wall times (HIP events) -- launch, wave start-up (the waves of 256 workgroups of 1024 threads
// start tens of microseconds apart: a first version that divided one short launch's per-wave s_memtime by the waves per SIMD was
// fooled by waves that never ran together) and drain cancel:
//     cycles per wave-instruction per SIMD = (t_B - t_A) * f / ((ITERS_B - ITERS_A) * 8 * waves per SIMD),   f = the device's clock.
// Beside it: the same from the median wave's own s_memtime difference (= shader cycles, the guide's constants table) -- the two agree
// when the clock holds its nominal rate.  Occupancy is not assumed: every wave records HW_REG_HW_ID and the host counts waves per SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define CHECK(x)                                                   \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            std::printf("%s: %s\n", #x, hipGetErrorString(e));     \
            return 1;                                              \
        }                                                          \
    } while (0)

constexpr int ITERS_A = 1024, ITERS_B = 9216;

// (four rounds of the eight registers per loop iteration: the loop's own s_add / s_cmp / s_cbranch are 3 of 35 instructions)
#define BODY8(OP)                                                                                                         \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)          \
                 OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)          \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])         \
                 : "v"(c), "v"(d))

#define OP_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define OP_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define OP_NOT(i) "v_not_b32 %" #i ", %" #i "\n"
#define OP_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define OP_LSHL_ADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define OP_LSHL_OR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define OP_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 27\n"
#define OP_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define OP_MIN(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define OP_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_MAX3(i) "v_max3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define OP_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 30\n"
#define OP_BFREV(i) "v_bfrev_b32 %" #i ", %" #i "\n"
#define OP_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_CMP_CNDMASK(i) "v_cmp_lt_u32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define OP_ADD_XOR(i) "v_add_u32 %" #i ", %" #i ", %8\nv_lshl_add_u32 %" #i ", %" #i ", 3, %9\n"
#define OP_DPP_SHL(i) "v_mov_b32_dpp %" #i ", %" #i " wave_shl:1 row_mask:0xf bank_mask:0xf\n"
#define OP_MUL_LO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define OP_MAD_U24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define OP_SDWA(i) "v_lshlrev_b32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define OP_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
template <int WHICH> __global__ __launch_bounds__(1024) void probe(uint32_t* out, uint32_t seed, long long* cycles, uint32_t* hwid, int iters)
{
    extern __shared__ uint32_t lds_pad[]; // only sized: keeps the number of workgroups per CU at what the host asked for
    uint32_t r[8];
    for (int i = 0; i < 8; ++i) r[i] = ((threadIdx.x * 977u + i * 131u + seed) * 2654435761u);
    uint32_t c = (threadIdx.x * 4u) | 0x9E3779u, d = threadIdx.x * 0x01010101u + seed;
    asm volatile("v_cmp_gt_u32 vcc, %0, %1" ::"v"(c), "v"(d) : "vcc");
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (WHICH == 0) BODY8(OP_ADD);
        if (WHICH == 1) BODY8(OP_XOR);
        if (WHICH == 2) BODY8(OP_AND);
        if (WHICH == 3) BODY8(OP_NOT);
        if (WHICH == 4) BODY8(OP_LSHL);
        if (WHICH == 5) BODY8(OP_LSHL_ADD);
        if (WHICH == 6) BODY8(OP_LSHL_OR);
        if (WHICH == 7) BODY8(OP_BFE);
        if (WHICH == 8) BODY8(OP_BITOP3);
        if (WHICH == 9) BODY8(OP_MIN);
        if (WHICH == 10) BODY8(OP_MIN3);
        if (WHICH == 11) BODY8(OP_MAX3);
        if (WHICH == 12) BODY8(OP_DOT4);
        if (WHICH == 13) BODY8(OP_ALIGNBIT);
        if (WHICH == 14) BODY8(OP_BFREV);
        if (WHICH == 15) BODY8(OP_PERM);
        if (WHICH == 16) BODY8(OP_CNDMASK);
        if (WHICH == 17) BODY8(OP_DPP_SHL);
        if (WHICH == 18) BODY8(OP_MUL_LO);
        if (WHICH == 19) BODY8(OP_MAD_U24);
        if (WHICH == 20) BODY8(OP_SDWA);
        if (WHICH == 21) BODY8(OP_MBCNT);
        if (WHICH == 22) BODY8(OP_CMP_CNDMASK);
        if (WHICH == 23) BODY8(OP_ADD_XOR);
    }
    const long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= r[i];
    const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        cycles[gw] = t1 - t0;
        hwid[gw] = id;
    }
}

int main()
{
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const double f_khz = prop.clockRate;
    std::printf("%s, %d CUs, clock %d kHz; per cell: cycles per wave-instruction per SIMD from wall-time differences | from s_memtime differences (waves per SIMD seen)\n",
        prop.name, ncu, prop.clockRate);
    struct Op {
        const char* name;
        int per_iter; // wave-instructions per loop iteration
    };
    const Op ops[] = { { "v_add_u32", 32 }, { "v_xor_b32", 32 }, { "v_and_b32", 32 }, { "v_not_b32", 32 }, { "v_lshlrev_b32", 32 }, { "v_lshl_add_u32", 32 },
        { "v_lshl_or_b32", 32 }, { "v_bfe_u32", 32 }, { "v_bitop3_b32", 32 }, { "v_min_u32", 32 }, { "v_min3_u32", 32 }, { "v_max3_u32", 32 },
        { "v_dot4_u32_u8", 32 }, { "v_alignbit_b32", 32 }, { "v_bfrev_b32", 32 }, { "v_perm_b32", 32 }, { "v_cndmask_b32 (vcc)", 32 },
        { "v_mov_b32_dpp wave_shl:1", 32 }, { "v_mul_lo_u32", 32 }, { "v_mad_u32_u24", 32 }, { "v_lshlrev_b32_sdwa", 32 }, { "v_mbcnt_lo_u32_b32", 32 },
        { "v_cmp_lt_u32 + v_cndmask", 64 }, { "v_add_u32 + v_lshl_add_u32", 64 } };
    using K = void (*)(uint32_t*, uint32_t, long long*, uint32_t*, int);
    K kernels[] = { probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>, probe<9>, probe<10>, probe<11>,
        probe<12>, probe<13>, probe<14>, probe<15>, probe<16>, probe<17>, probe<18>, probe<19>, probe<20>, probe<21>, probe<22>, probe<23> };
    const int nk = sizeof kernels / sizeof kernels[0];
    const int max_threads = ncu * 2 * 1024;
    uint32_t *out, *hwid;
    long long* cyc;
    CHECK(hipMalloc(&out, (size_t)max_threads * 4));
    CHECK(hipMalloc(&cyc, (size_t)(max_threads / 64) * 8));
    CHECK(hipMalloc(&hwid, (size_t)(max_threads / 64) * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::printf("%-28s", "waves per SIMD ->");
    for (int w : { 1, 2, 4, 8 }) std::printf("  %-30d", w);
    std::printf("\n");
    for (int wh = 0; wh < nk; ++wh) {
        CHECK(hipFuncSetAttribute((const void*)kernels[wh], hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
        std::printf("%-28s", ops[wh].name);
        for (int w : { 1, 2, 4, 8 }) {
            // w <= 4: one workgroup of 256 w threads per CU (100 KB of LDS each: a second one does not fit); w = 8: two of 1024 (64 KB each)
            const int block = 256 * std::min(w, 4), per_cu = w > 4 ? w / 4 : 1;
            const size_t lds = per_cu == 1 ? 100 * 1024 : 64 * 1024;
            const int grid = ncu * per_cu, waves = grid * block / 64;
            float ms[2] = { 0, 0 };
            double med[2] = { 0, 0 };
            double occ_lo = 0, occ_hi = 0;
            hipLaunchKernelGGL(kernels[wh], dim3(grid), dim3(block), lds, 0, out, 1u, cyc, hwid, 64); // warm
            CHECK(hipDeviceSynchronize());
            for (int pass = 0; pass < 2; ++pass) {
                const int iters = pass ? ITERS_B : ITERS_A;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(a, 0));
                    hipLaunchKernelGGL(kernels[wh], dim3(grid), dim3(block), lds, 0, out, 2u + rep, cyc, hwid, iters);
                    CHECK(hipEventRecord(b, 0));
                    CHECK(hipDeviceSynchronize());
                    float t = 0;
                    CHECK(hipEventElapsedTime(&t, a, b));
                    best = std::min(best, t);
                }
                ms[pass] = best;
                std::vector<long long> h(waves);
                std::vector<uint32_t> id(waves);
                CHECK(hipMemcpy(h.data(), cyc, (size_t)waves * 8, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(id.data(), hwid, (size_t)waves * 4, hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                med[pass] = (double)h[h.size() / 2];
                // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]; the eight XCDs repeat
                // the same ids, so the count per id is divided by 8
                std::map<uint32_t, int> per_simd;
                for (uint32_t x : id) per_simd[(x >> 4 & 3) | (x >> 8 & 0xFF) << 2]++;
                std::vector<int> occ;
                for (auto& kv : per_simd) occ.push_back(kv.second);
                std::sort(occ.begin(), occ.end());
                occ_lo = occ.front() / 8.0;
                occ_hi = occ.back() / 8.0;
            }
            const double n_instr = (double)(ITERS_B - ITERS_A) * ops[wh].per_iter * w;
            const double wall_cyc = (double)(ms[1] - ms[0]) * f_khz / n_instr;
            const double mem_cyc = (med[1] - med[0]) / n_instr;
            std::printf("  %5.2f | %5.2f (%.0f-%.0f, %.2f ms)", wall_cyc, mem_cyc, occ_lo, occ_hi, ms[1]);
        }
        std::printf("\n");
    }
    return 0;
}
