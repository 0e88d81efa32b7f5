// device_common.h -- device-side helpers shared by the kernels of the hot path (hash, base encoding, read lookup).
#pragma once
#include "kernels.h"
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace drprg {
namespace dev {

#define HIP_TRY(x)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) return e_;                                                             \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (kernel, device): a process that drives several devices
// (drprg_hip_open_multi) has to set it once on each.  cache: one slot per device, owned by the call site.
constexpr int MAX_HIP_DEVICES = 64;
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes, size_t (&cache)[MAX_HIP_DEVICES])
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < MAX_HIP_DEVICES && bytes <= cache[dev]) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && dev < MAX_HIP_DEVICES) cache[dev] = bytes;
    return e;
}

// One launch, timed or not.  With a KernelTimer the two events ride on the kernel's own dispatch (hipExtLaunchKernelGGL: its start and
// end timestamps) instead of being recorded on the stream before and after it -- two barrier packets that cost a 0.56 ms step 5.7 us of
// idle GPU each (rocprofv3 --kernel-trace timeline, round 5).  hipEventElapsedTime(begin, end) is the kernel's duration either way.
template <typename... Args, typename... Given>
inline void launch_timed(const KernelTimer& timer, void (*kernel)(Args...), dim3 grid, dim3 block, size_t dyn_lds, hipStream_t stream, Given&&... args)
{
    if (timer.begin && timer.end) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)dyn_lds, stream, timer.begin, timer.end, 0, Args(args)...);
    else hipLaunchKernelGGL(kernel, grid, block, dyn_lds, stream, Args(args)...);
}

// ---------------------------------------------------------------------------------------------
// hash
// ---------------------------------------------------------------------------------------------
template <typename HT> struct HashTraits;
template <> struct HashTraits<uint32_t> {
    static constexpr uint32_t EMPTY = 0xFFFFFFFFu; // never a hash: this path serves k <= 15 (30 bits)
    // minimap hash64 restricted to <= 30 bits: every intermediate is taken mod 2^(2k), so 32-bit
    // arithmetic is exact; the final (key + key<<31) term vanishes below 31 bits.
    __device__ static inline uint32_t mix(uint32_t key, uint32_t mask)
    {
        key = (~key + (key << 21)) & mask;
        key = key ^ (key >> 24);
        key = (key + (key << 3) + (key << 8)) & mask;
        key = key ^ (key >> 14);
        key = (key + (key << 2) + (key << 4)) & mask;
        key = key ^ (key >> 28);
        return key;
    }
};
// 16 bytes that are read exactly once (the base stream): non-temporal, so that the stream does not push the probe tables
// out of the XCD's L2
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_once_16(const void* p)
{
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// (a << SH) + b in one full-rate instruction.  Spelled as inline assembly because the compiler otherwise folds
// key + (key << 3) + (key << 8) into v_mul_lo_u32, which issues at a quarter of the rate on gfx950.
template <int SH> __device__ __forceinline__ uint32_t lshl_add(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(SH), "v"(b));
    return d;
}

// bits [OFF + WIDTH - 1 : OFF] of x in one instruction (the compiler's own choice for __builtin_amdgcn_ubfe is shift + and)
template <int OFF, int WIDTH> __device__ __forceinline__ uint32_t bfe(uint32_t x)
{
    uint32_t d;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "n"(OFF), "n"(WIDTH));
    return d;
}

// hash64 restricted to 2K <= 30 bits for a compile-time K, without the intermediate masks: additions and left shifts only
// carry upwards, so bits >= 2K never reach bits < 2K; the three right shifts read exactly bits [2K-1 : s] through v_bfe_u32.
// Bits >= 2K of the argument are ignored; the result is exact in bits < 2K and ZERO above (the last step masks).
// 12 full-rate VALU instructions + the closing mask.
template <int K> __device__ __forceinline__ uint32_t mix_k(uint32_t key)
{
    static_assert(K >= 1 && K <= 15, "32-bit hash path");
    constexpr uint32_t MASK = (1u << (2 * K)) - 1u;
    uint32_t x = lshl_add<21>(key, ~key);
    if constexpr (2 * K > 24) x ^= bfe<24, (2 * K > 24 ? 2 * K - 24 : 1)>(x);
    x = lshl_add<8>(x, lshl_add<3>(x, x));
    if constexpr (2 * K > 14) x ^= bfe<14, (2 * K > 14 ? 2 * K - 14 : 1)>(x);
    x = lshl_add<4>(x, lshl_add<2>(x, x));
    if constexpr (2 * K > 28) return (x & MASK) ^ bfe<28, (2 * K > 28 ? 2 * K - 28 : 1)>(x);
    return x & MASK;
}

template <> struct HashTraits<uint64_t> {
    static constexpr uint64_t EMPTY = ~0ULL; // k <= 31: hashes stay below 2^62
    __device__ static inline uint64_t mix(uint64_t key, uint64_t mask)
    {
        key = (~key + (key << 21)) & mask;
        key = key ^ (key >> 24);
        key = (key + (key << 3) + (key << 8)) & mask;
        key = key ^ (key >> 14);
        key = (key + (key << 2) + (key << 4)) & mask;
        key = key ^ (key >> 28);
        key = (key + (key << 31)) & mask;
        return key;
    }
};

// index.h table_slot on the device: 32-bit keys (k <= 15) take one 32-bit multiply, 64-bit keys the 64-bit one
__device__ inline uint32_t table_slot_dev(uint32_t key, uint32_t bits) { return (key * 0x9E3779B1u) >> (32 - bits); }
__device__ inline uint32_t table_slot_dev(uint64_t key, uint32_t bits)
{
    return (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> (64 - bits));
}

// The exact table of 32-bit hashes, searched four slots per round trip: the aligned group of four that holds slot `sl`, then the groups
// after it.  Same answer as one slot after the other (the first slot in probe order that holds the hash or is empty; the table is a
// power of two >= 16 slots at <= 50 % load, index.cpp), in 1.1 dependent loads on average where linear probing takes 1.3 for a hit and
// 1.9 for a miss -- and what a wave waits for is its slowest lane.  true: found, sl = its slot; false: sl = the empty slot.
__device__ __forceinline__ bool table_find4(const uint32_t* __restrict__ slot_key, uint32_t tmask, uint32_t h, uint32_t& sl)
{
    constexpr uint32_t EMPTY = HashTraits<uint32_t>::EMPTY;
#ifdef DRPRG_PROBE_LINEAR // (measurement builds: one slot per round trip, rounds 1-4)
    while (true) {
        const uint32_t key = slot_key[sl];
        if (key == h) return true;
        if (key == EMPTY) return false;
        sl = (sl + 1) & tmask;
    }
#endif
    uint32_t base = sl & ~3u, from = 0xFu << (sl & 3u);
    while (true) {
        const uint4 q = *reinterpret_cast<const uint4*>(slot_key + base);
        const uint32_t hit = (q.x == h ? 1u : 0u) | (q.y == h ? 2u : 0u) | (q.z == h ? 4u : 0u) | (q.w == h ? 8u : 0u);
        const uint32_t end = hit | (q.x == EMPTY ? 1u : 0u) | (q.y == EMPTY ? 2u : 0u) | (q.z == EMPTY ? 4u : 0u) | (q.w == EMPTY ? 8u : 0u);
        const uint32_t m = end & from;
        if (m) {
            const uint32_t i = (uint32_t)__builtin_ctz(m);
            sl = base + i;
            return (hit >> i) & 1u;
        }
        base = (base + 4u) & tmask;
        from = 0xFu;
    }
}

// ASCII -> code byte: bits 0-1 base (A0 C1 G2 T3), bit 2 = not ACGT
__device__ inline uint32_t encode_base(uint32_t c)
{
    uint32_t u = c & 0xDFu; // upper case
    uint32_t x = (u >> 1) & 3u;
    x ^= x >> 1;
    bool ok = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
    return ok ? x : 4u;
}
// four bases at once (SWAR): same mapping on every byte of a dword
__device__ inline uint32_t encode4(uint32_t word)
{
    const uint32_t u = word & 0xDFDFDFDFu; // upper case
    const uint32_t sel = (u >> 1) & 0x03030303u; // A0 C1 T2 G3: a byte-wise index into two 4-entry tables
    // v_perm_b32 as a 4-entry byte LUT: selector bytes 0..3 pick bytes of the second operand
    const uint32_t code = __builtin_amdgcn_perm(0u, 0x02030100u, sel);   // -> A0 C1 G2 T3
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, sel); // the letter that index stands for
    const uint32_t diff = u ^ expect; // a byte is a valid base iff it equals the letter of its index
    const uint32_t bad = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u; // bit 7 set in every non-zero byte
    return code | (bad >> 5); // 0x80 >> 5 = 4: the "not ACGT" flag
}

// read holding global base position gp, searched upwards from read `lo` (offsets[lo] <= gp)
__device__ inline uint32_t find_read_from(const uint64_t* __restrict__ offsets, uint32_t n_reads, uint32_t lo, uint64_t gp)
{
    uint32_t step = 1, hi = lo + 1;
    while (hi < n_reads && offsets[hi] <= gp) { // gallop
        lo = hi;
        step <<= 1;
        hi = (n_reads - lo > step) ? lo + step : n_reads;
    }
    // invariant: offsets[lo] <= gp < offsets[hi]  (offsets[n_reads] = n_bases > gp)
    while (hi - lo > 1) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (offsets[mid] <= gp) lo = mid; else hi = mid;
    }
    return lo;
}

// read holding gp: the interpolated guess first (exact for fixed-length reads: two independent loads), else the gallop from
// `lo` (offsets[lo] <= gp; a read that starts shortly before the tile)
__device__ inline uint32_t find_read_guess(const uint64_t* __restrict__ offsets, uint32_t n_reads, uint32_t guess, uint32_t lo, uint64_t gp)
{
    if (guess >= n_reads) guess = n_reads - 1;
    if (offsets[guess] <= gp && gp < offsets[guess + 1]) return guess;
    return find_read_from(offsets, n_reads, lo, gp);
}

// read holding global base position gp, searched outwards from `guess` (any read index): gallops down or up, then
// bisects.  With near-uniform read lengths and guess = gp * n_reads / n_bases this costs two or three loads.
__device__ inline uint32_t find_read_near(const uint64_t* __restrict__ offsets, uint32_t n_reads, uint32_t guess, uint64_t gp)
{
    uint32_t lo = guess < n_reads ? guess : n_reads - 1, hi;
    if (offsets[lo] <= gp) {
        uint32_t step = 1;
        hi = lo + 1;
        while (hi < n_reads && offsets[hi] <= gp) {
            lo = hi;
            step <<= 1;
            hi = (n_reads - lo > step) ? lo + step : n_reads;
        }
    } else {
        uint32_t step = 1;
        hi = lo;
        while (true) {
            lo = hi > step ? hi - step : 0;
            if (offsets[lo] <= gp) break;
            hi = lo;
            step <<= 1;
        }
    }
    // invariant: offsets[lo] <= gp < offsets[hi]  (offsets[n_reads] = n_bases > gp)
    while (hi - lo > 1) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (offsets[mid] <= gp) lo = mid; else hi = mid;
    }
    return lo;
}

// 2 * len / (w + 1) of pandora's cluster threshold without a 64-bit division per call: for w < 200 and len < 2^23 (the longest
// read the hit key takes) floor(x / d) == (x * M) >> 32 with M = floor(2^32 / d) + 1, because x * (d * M - 2^32) < 2^32
__device__ inline uint32_t w1_reciprocal(int w) { return (uint32_t)(0x100000000ull / (uint32_t)(w + 1)) + 1u; }
__device__ inline uint64_t expected_minimizers(uint64_t len, int w, uint32_t magic)
{
    if (w < 200 && len < (1ull << 23)) return (uint64_t)__umulhi((uint32_t)len * 2u, magic);
    return len * 2 / (uint64_t)(w + 1);
}

// ---------------------------------------------------------------------------------------------
// wave / workgroup scans
// ---------------------------------------------------------------------------------------------
__device__ inline uint32_t wave_inclusive_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = __shfl_up(v, off);
        if (lane >= off) v += n;
    }
    return v;
}
__device__ inline uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t n = __shfl_xor(v, off);
        v = n > v ? n : v;
    }
    return v;
}

// exclusive scan over the workgroup (NW waves); s_w: NW + 1 words of LDS; returns the exclusive prefix, total in *total.
// Two barriers; s_w may be reused after the call returns on every thread.
template <int NW> __device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_w, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_inclusive_scan(v);
    __syncthreads(); // s_w free again (previous call)
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint32_t x = s_w[i];
        if (i < wave) base += x;
        sum += x;
    }
    *total = sum;
    return base + incl - v;
}

} // namespace dev
} // namespace drprg
