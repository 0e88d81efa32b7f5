// rccl_dyn.h -- RCCL bound at run time (dlopen), for the one collective of the hot path: the sum of the per-GPU coverage
// vectors (the reference is single-process and has no counterpart: /root/reference/src/lib.rs:580-642).
//
// Why not link librccl: a host process that also holds PyTorch carries PyTorch's own copy of RCCL; two copies of the library in
// one process is asking for trouble, and a host without any multi-GPU use should not need the library at all.  dlopen by soname
// picks up the copy the process already has, if it has one.
#pragma once
#include <cstddef>
#include <cstdint>
#include <hip/hip_runtime_api.h>
#include <string>

namespace drprg {

struct Rccl {
    // (declarations restated from /opt/rocm/include/rccl/rccl.h, ROCm 7.2: the ABI of librccl.so.1)
    struct UniqueId {
        char internal[128];
    };
    typedef void* Comm;
    enum { Success = 0 };
    enum { Uint32 = 3 }; // ncclDataType_t
    enum { Sum = 0 };    // ncclRedOp_t
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommInitAll)(Comm*, int, const int*) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;

    // the process-wide binding; nullptr (and *why filled) when the library or one of its symbols is missing
    static const Rccl* get(std::string* why = nullptr);
};

} // namespace drprg
