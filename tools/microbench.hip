// microbench.hip -- issue-rate probes for the instructions the Bloom-prefilter kernel is made of (gfx950).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o /tmp/mb && /tmp/mb
// Prints cycles per wave-instruction per SIMD (4 = full rate for a wave64 on a 16-lane SIMD).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            std::printf("%s: %s\n", #x, hipGetErrorString(e));                     \
            return 1;                                                              \
        }                                                                          \
    } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;

#define BODY8(OP)                                                                                          \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                           \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                 : "v"(c), "s"(sc))

#define OP_MUL_LO(i) "v_mul_lo_u32 %" #i ", %" #i ", %9\n"
#define OP_MUL_U24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define OP_MUL_HI_U24(i) "v_mul_hi_u32_u24 %" #i ", %" #i ", %8\n"
#define OP_MAD_U24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define OP_LSHL(i) "v_lshlrev_b32 %" #i ", %8, %" #i "\n"
#define OP_LSHL_SDWA(i) "v_lshlrev_b32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define OP_AND_SDWA(i) "v_and_b32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define OP_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 30\n"
#define OP_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_AND3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %8 bitop3:0x80\n"
#define OP_LSHL_OR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define OP_DSREAD(i) "ds_read_b32 %" #i ", %" #i "\n"
#define OP_BPERM(i) "ds_bpermute_b32 %" #i ", %8, %" #i "\n"

template <int WHICH> __global__ __launch_bounds__(256) void probe(uint32_t* out, uint32_t seed, long long* cycles)
{
    __shared__ uint32_t lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (i * 2654435761u) & 0xFFFCu; // random dword-aligned addresses
    __syncthreads();
    uint32_t r[8];
    for (int i = 0; i < 8; ++i) r[i] = ((threadIdx.x * 977u + i * 131u + seed) * 2654435761u) & 0xFFFCu;
    uint32_t c = (threadIdx.x * 4u) | 0x9E3779u;
    if (WHICH == 13) c = ((threadIdx.x + 1) & 63) * 4;
    uint32_t sc = 0x9E3779B1u;
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        if (WHICH == 0) BODY8(OP_ADD);
        if (WHICH == 1) BODY8(OP_MUL_LO);
        if (WHICH == 2) BODY8(OP_MUL_U24);
        if (WHICH == 3) BODY8(OP_MUL_HI_U24);
        if (WHICH == 4) BODY8(OP_MAD_U24);
        if (WHICH == 5) BODY8(OP_LSHL);
        if (WHICH == 6) BODY8(OP_LSHL_SDWA);
        if (WHICH == 7) BODY8(OP_AND_SDWA);
        if (WHICH == 8) BODY8(OP_ALIGNBIT);
        if (WHICH == 9) BODY8(OP_PERM);
        if (WHICH == 10) BODY8(OP_AND3);
        if (WHICH == 11) BODY8(OP_LSHL_OR);
        if (WHICH == 12) { BODY8(OP_DSREAD); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (WHICH == 13) { BODY8(OP_BPERM); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    }
    const long long t1 = clock64();
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// semantics probes: DPP wave shift, SDWA shift amount, unaligned LDS read
__global__ void semantics(uint32_t* out)
{
    __shared__ uint32_t lds[64];
    const uint32_t lane = threadIdx.x;
    lds[lane] = 0x01010101u * lane;
    __syncthreads();
    uint32_t v = 1000 + lane, d = 0xDEAD;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(v));
    out[lane] = d; // which lane's value arrives with wave_shl:1
    uint32_t e = 0xDEAD;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(e) : "v"(v));
    out[64 + lane] = e;
    uint32_t h = 0x00050300u + (lane << 8), w = 1, s;
    asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(s) : "v"(h), "v"(w));
    out[128 + lane] = s; // expect 1 << ((3 + lane) & 31)
    uint32_t u = 0;
    if (lane < 60) {
        const uint32_t addr = (uint32_t)(uintptr_t)lds + lane + 1; // unaligned for most lanes
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(u) : "v"(addr) : "memory");
    }
    out[192 + lane] = u;
}

int main()
{
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    std::printf("%s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    uint32_t* out;
    long long* cyc;
    const int grid = prop.multiProcessorCount * 2; // 2 x 256 threads per CU = 2 waves per SIMD
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    CHECK(hipMalloc(&cyc, (size_t)grid * 8));
    const char* names[] = { "v_add_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_mad_u32_u24", "v_lshlrev_b32",
        "v_lshlrev_b32_sdwa", "v_and_b32_sdwa", "v_alignbit_b32", "v_perm_b32", "v_bitop3_b32", "v_lshl_or_b32",
        "ds_read_b32 random", "ds_bpermute_b32" };
    using K = void (*)(uint32_t*, uint32_t, long long*);
    K kernels[] = { probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>, probe<9>, probe<10>,
        probe<11>, probe<12>, probe<13> };
    for (int wh = 0; wh < 14; ++wh) {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL(kernels[wh], dim3(grid), dim3(256), 0, 0, out, 1u, cyc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kernels[wh], dim3(grid), dim3(256), 0, 0, out, 2u, cyc);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        std::vector<long long> h(grid);
        CHECK(hipMemcpy(h.data(), cyc, (size_t)grid * 8, hipMemcpyDeviceToHost));
        double mean = 0;
        for (long long c : h) mean += (double)c;
        mean /= grid;
        // per SIMD: 2 waves x ITERS x 8 instructions share the issue port
        const double n_wave_instr_per_simd = 2.0 * ITERS * UNROLL;
        std::printf("%-22s %8.3f ms  clock64 ticks/wave-instr/SIMD %.3f   (wall: %.2f ns per wave-instr per SIMD)\n", names[wh], ms,
            mean / n_wave_instr_per_simd, ms * 1e6 / n_wave_instr_per_simd);
    }
    uint32_t* sem;
    CHECK(hipMalloc(&sem, 256 * 4));
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, sem);
    CHECK(hipDeviceSynchronize());
    uint32_t hs[256];
    CHECK(hipMemcpy(hs, sem, sizeof hs, hipMemcpyDeviceToHost));
    std::printf("wave_shl:1 lanes 0,1,31,32,62,63 receive: %u %u %u %u %u %u\n", hs[0], hs[1], hs[31], hs[32], hs[62], hs[63]);
    std::printf("wave_shr:1 lanes 0,1,31,32,62,63 receive: %u %u %u %u %u %u\n", hs[64], hs[65], hs[95], hs[96], hs[126], hs[127]);
    std::printf("sdwa shift BYTE_1: lane0 %#x (expect 0x8) lane5 %#x (expect 0x100) lane40 %#x (expect 1<<((3+40)&31)=0x800)\n", hs[128], hs[133], hs[168]);
    std::printf("unaligned ds_read_b32 at +1,+2,+3,+4: %#x %#x %#x %#x\n", hs[192], hs[193], hs[194], hs[195]);
    return 0;
}
