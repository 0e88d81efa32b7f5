// mb_l2probe.hip -- how many random 4-byte probes of an L2-resident table does an MI355X serve, alone and next to a
// streaming read?  (Decides the middle tier of the filter: an exact 2^24-bit 12-mer bitmap lives in the L2, not in LDS.)
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mb_l2probe.hip -o /tmp/mbl2 && /tmp/mbl2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                   \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            std::printf("%s: %s\n", #x, hipGetErrorString(e));     \
            std::exit(1);                                          \
        }                                                          \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA6Bu;
    return x ^ (x >> 13);
}

// MODE 0: plain dword loads, 1: nontemporal, 2: byte loads, 3: 8-byte loads
// every lane: `rounds` rounds of 8 independent probes; a lane takes part in a probe with probability active/256
template <int MODE> __global__ __launch_bounds__(1024) void probe_kernel(const uint32_t* __restrict__ table, uint32_t mask_words, int rounds,
    uint32_t active, uint32_t* out)
{
    uint32_t s = mix(blockIdx.x * 1024u + threadIdx.x + 1u), acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            const uint32_t h = mix(s);
            v[j] = 0;
            if ((h >> 24) < active) {
                const uint32_t idx = h & mask_words;
                if (MODE == 0) v[j] = table[idx];
                if (MODE == 1) v[j] = __builtin_nontemporal_load(table + idx);
                if (MODE == 2) v[j] = reinterpret_cast<const uint8_t*>(table)[idx * 4u + (h >> 30)];
                if (MODE == 3) {
                    const uint2 t = reinterpret_cast<const uint2*>(table)[idx >> 1];
                    v[j] = t.x ^ t.y;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    if (acc == 0x12345u) out[0] = acc;
}

// streaming read (two 16-byte loads per lane and tile, as sketch_filter_kernel) with `p` probes per lane and tile
template <int P> __global__ __launch_bounds__(1024) void stream_probe_kernel(const uint4* __restrict__ bases, size_t n16, const uint32_t* __restrict__ table,
    uint32_t mask_words, uint32_t active, uint32_t* out)
{
    const size_t n_waves = (size_t)gridDim.x * 16, gw = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t tiles = n16 / 128, per = (tiles + n_waves - 1) / n_waves;
    size_t t = gw * per, t_end = t + per < tiles ? t + per : tiles;
    uint32_t acc = 0, s = mix((uint32_t)gw * 64u + (uint32_t)lane + 7u);
    for (; t < t_end; ++t) {
        const uint4 a = bases[t * 128 + lane * 2], b = bases[t * 128 + lane * 2 + 1];
        uint32_t key = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
        acc += key;
        uint32_t v[P ? P : 1];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            s = s * 1664525u + 1013904223u;
            const uint32_t h = mix(s);
            v[j] = 0;
            if ((h >> 24) < active) v[j] = table[h & mask_words];
        }
#pragma unroll
        for (int j = 0; j < P; ++j) acc += v[j];
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <typename F> static float time_ms(F f, int reps = 5)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    uint32_t *table, *out;
    const size_t max_table = 64u << 20;
    CHECK(hipMalloc(&table, max_table));
    CHECK(hipMemset(table, 0, max_table));
    CHECK(hipMalloc(&out, 64));
    const int grid = 256, rounds = 180; // 256 x 1024 lanes x 180 x 8 = 377 M probe slots
    const double slots = (double)grid * 1024 * rounds * 8;
    std::printf("probe slots per launch: %.0f M\n", slots / 1e6);
    std::printf("%-10s %-8s %-8s %10s %14s\n", "table", "mode", "active", "ms", "Gprobes/s");
    const char* names[4] = { "dword", "nt", "byte", "dwordx2" };
    for (size_t tb : { (size_t)32 << 10, (size_t)256 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20 })
        for (int mode = 0; mode < 4; ++mode)
            for (uint32_t active : { 256u, 128u, 38u }) {
                if (mode && (active != 256u || (tb != ((size_t)2 << 20) && tb != ((size_t)1 << 20)))) continue;
                const uint32_t mask = (uint32_t)(tb / 4 - 1);
                float ms = 0;
                if (mode == 0) ms = time_ms([&] { hipLaunchKernelGGL(probe_kernel<0>, dim3(grid), dim3(1024), 0, 0, table, mask, rounds, active, out); });
                if (mode == 1) ms = time_ms([&] { hipLaunchKernelGGL(probe_kernel<1>, dim3(grid), dim3(1024), 0, 0, table, mask, rounds, active, out); });
                if (mode == 2) ms = time_ms([&] { hipLaunchKernelGGL(probe_kernel<2>, dim3(grid), dim3(1024), 0, 0, table, mask, rounds, active, out); });
                if (mode == 3) ms = time_ms([&] { hipLaunchKernelGGL(probe_kernel<3>, dim3(grid), dim3(1024), 0, 0, table, mask, rounds, active, out); });
                std::printf("%-10zu %-8s %-8u %10.3f %14.1f\n", tb, names[mode], active, ms, slots * active / 256.0 / ms / 1e6);
            }
    // next to the streaming read of 1.5 GB
    const size_t n_bytes = (size_t)1500 << 20;
    uint4* bases;
    CHECK(hipMalloc(&bases, n_bytes));
    CHECK(hipMemset(bases, 1, n_bytes));
    std::printf("\nstreaming 1.5 GB (32 bytes per lane and tile) + P probes per lane and tile (47 M lane-tiles)\n");
    std::printf("%-10s %-4s %-8s %10s %12s %14s\n", "table", "P", "active", "ms", "stream GB/s", "Gprobes/s");
    for (size_t tb : { (size_t)2 << 20, (size_t)1 << 20, (size_t)256 << 10 })
        for (int P : { 0, 1, 2, 4, 8 })
            for (uint32_t active : { 256u, 96u, 38u }) {
                if (P == 0 && (active != 256u)) continue;
                if (P == 0 && tb != ((size_t)2 << 20)) continue;
                const uint32_t mask = (uint32_t)(tb / 4 - 1);
                const size_t n16 = n_bytes / 16;
                float ms = 0;
#define RUN(PP) ms = time_ms([&] { hipLaunchKernelGGL(stream_probe_kernel<PP>, dim3(256), dim3(1024), 0, 0, bases, n16, table, mask, active, out); })
                if (P == 0) RUN(0);
                if (P == 1) RUN(1);
                if (P == 2) RUN(2);
                if (P == 4) RUN(4);
                if (P == 8) RUN(8);
                const double probes = (double)(n16 / 2) * P * active / 256.0;
                std::printf("%-10zu %-4d %-8u %10.3f %12.0f %14.1f\n", tb, P, active, ms, n_bytes / ms / 1e6, probes / ms / 1e6);
            }
    return 0;
}
