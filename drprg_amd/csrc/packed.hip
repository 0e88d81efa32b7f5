// packed.hip -- 2-bit packed reads (SURVEY.md section 8f NEXT-4 "optional 2-bit packing"; SketchArgs::packed in kernels.h): the
// conversions between the packed form and ASCII on the device.
//
// The packed form is what crosses PCIe and what stays in HBM (a quarter of the bytes); the kernels of the filtered launch sequence
// (sketch_filter.hip, candidates.hip verify_count_kernel) read it directly.  The direct sketch kernels (sketch_wave.hip,
// sketch_probe.hip: indexes beyond the filter's reach, 5 ms per 10 M reads and VALU-bound on their hashes) and the anchor scan of
// `discover` (anchor_scan.hip) keep their ASCII input: a packed batch is expanded once for them, 0.3 ms per 1.5 G bases at the
// bandwidth of the copy.  HBM-bound: n / 4 bytes read, n bytes written.
#include "device_common.h"

namespace drprg {
namespace dev {

constexpr int PK_THREADS = 256;

// one thread per word: 16 bases -> 16 bytes
__global__ __launch_bounds__(PK_THREADS) void unpack_kernel(const uint32_t* __restrict__ words, uint64_t n_bases, uint8_t* __restrict__ out)
{
    const uint64_t n_words = (n_bases + 15) >> 4, n_out = n_words + 4; // + the 64 bytes of 'N' behind the last base
    for (uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x; i < n_out; i += (uint64_t)gridDim.x * PK_THREADS) {
        const uint32_t w = i < n_words ? words[i] : 0u;
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // four letters -> their bytes: v_perm_b32 as a 4-entry table (selector byte j picks byte j of "GTCA" read as a dword)
            const uint32_t f = (w >> (8 * q)) & 0xFFu;
            const uint32_t sel = (f & 3u) | ((f >> 2) & 3u) << 8 | ((f >> 4) & 3u) << 16 | ((f >> 6) & 3u) << 24;
            o[q] = __builtin_amdgcn_perm(0u, 0x47544341u /* 'A' 'C' 'T' 'G' from the low byte up */, sel);
        }
        uint4 v = make_uint4(o[0], o[1], o[2], o[3]);
        const uint64_t b0 = i << 4;
        if (b0 + 16 > n_bases) { // the last word and the padding behind it: 'N' from n_bases on
            uint8_t* pb = reinterpret_cast<uint8_t*>(&v);
            for (int j = 0; j < 16; ++j)
                if (b0 + (uint64_t)j >= n_bases) pb[j] = 'N';
        }
        reinterpret_cast<uint4*>(out)[i] = v;
    }
}

__global__ __launch_bounds__(PK_THREADS) void restore_n_kernel(const uint64_t* __restrict__ npos, uint64_t n_npos, uint64_t n_bases, uint8_t* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x; i < n_npos; i += (uint64_t)gridDim.x * PK_THREADS)
        if (npos[i] < n_bases) out[npos[i]] = 'N';
}

// one thread per word: 16 bytes -> 16 letters (bits 2:1 of every byte, whatever it is) + the positions of the bytes that are not ACGTacgt.
// One 16-byte load per word (the batch pointer is 16-byte aligned by contract; byte loads otherwise and for the last word), the four letters
// of a dword gathered by one v_dot4_u32_u8 (weights 1, 4, 16, 64 on the fields at bits 2:1), the ACGT test on four bytes at a time (encode4).
__global__ __launch_bounds__(PK_THREADS) void pack_kernel(const uint8_t* __restrict__ bases, uint64_t n_bases, uint32_t* __restrict__ words,
    uint64_t* __restrict__ npos, uint64_t npos_cap, unsigned long long* __restrict__ n_npos)
{
    const uint64_t n_words = (n_bases + 15) >> 4;
    const bool aligned = (reinterpret_cast<uintptr_t>(bases) & 15u) == 0;
    for (uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * PK_THREADS) {
        const uint64_t b0 = i << 4;
        uint32_t in[4] = { 0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u }; // ('A': letter 0, a base)
        if (aligned && b0 + 16 <= n_bases) {
            const uint4 v = *reinterpret_cast<const uint4*>(bases + b0);
            in[0] = v.x; in[1] = v.y; in[2] = v.z; in[3] = v.w;
        } else {
            uint8_t* pb = reinterpret_cast<uint8_t*>(in);
            for (int j = 0; j < 16 && b0 + (uint64_t)j < n_bases; ++j) pb[j] = bases[b0 + j];
        }
        uint32_t w = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w |= (__builtin_amdgcn_udot4(in[q] & 0x06060606u, 0x40100401u, 0u, false) >> 1) << (8 * q);
            uint32_t bad = encode4(in[q]) & 0x04040404u;
            while (bad) { // rare
                const int j = (__ffs(bad) - 1) >> 3;
                bad &= bad - 1;
                const unsigned long long at = atomicAdd(n_npos, 1ull);
                if (at < npos_cap) npos[at] = b0 + (uint64_t)(4 * q + j);
            }
        }
        words[i] = w;
    }
}

__global__ __launch_bounds__(PK_THREADS) void mark_npos_kernel(const uint64_t* __restrict__ npos, uint64_t n_npos, uint64_t n_bases, uint32_t* __restrict__ bits32)
{
    for (uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x; i < n_npos; i += (uint64_t)gridDim.x * PK_THREADS)
        if (npos[i] < n_bases) atomicOr(&bits32[npos[i] >> 5], 1u << (npos[i] & 31)); // (u16 words i >> 4, little endian: the same bits)
}

hipError_t launch_mark_npos(const uint64_t* npos, uint64_t n_npos, uint64_t n_bases, uint16_t* bits, hipStream_t stream)
{
    if (!n_npos) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n_npos + PK_THREADS - 1) / PK_THREADS, 1u << 14);
    hipLaunchKernelGGL(mark_npos_kernel, dim3(grid), dim3(PK_THREADS), 0, stream, npos, n_npos, n_bases, reinterpret_cast<uint32_t*>(bits));
    return hipGetLastError();
}

hipError_t launch_unpack(const uint32_t* words, uint64_t n_bases, const uint64_t* npos, uint64_t n_npos, uint8_t* out, hipStream_t stream)
{
    const uint64_t n_out = ((n_bases + 15) >> 4) + 4;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n_out + PK_THREADS - 1) / PK_THREADS, 1u << 16);
    hipLaunchKernelGGL(unpack_kernel, dim3(grid), dim3(PK_THREADS), 0, stream, words, n_bases, out);
    HIP_TRY(hipGetLastError());
    if (n_npos) {
        const uint32_t g2 = (uint32_t)std::min<uint64_t>((n_npos + PK_THREADS - 1) / PK_THREADS, 1u << 14);
        hipLaunchKernelGGL(restore_n_kernel, dim3(g2), dim3(PK_THREADS), 0, stream, npos, n_npos, n_bases, out);
        HIP_TRY(hipGetLastError());
    }
    return hipSuccess;
}

hipError_t launch_pack(const uint8_t* bases, uint64_t n_bases, uint32_t* words, uint64_t* npos, uint64_t npos_cap, unsigned long long* n_npos,
    hipStream_t stream)
{
    const uint64_t n_words = (n_bases + 15) >> 4;
    if (!n_words) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n_words + PK_THREADS - 1) / PK_THREADS, 1u << 16);
    hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(PK_THREADS), 0, stream, bases, n_bases, words, npos, npos_cap, n_npos);
    return hipGetLastError();
}

} // namespace dev
} // namespace drprg
