// rccl_dyn.cpp -- see rccl_dyn.h
#include "rccl_dyn.h"
#include <cstdlib>
#include <dlfcn.h>
#include <mutex>

namespace drprg {

const Rccl* Rccl::get(std::string* why)
{
    static Rccl api;
    static std::string error;
    static bool ok = false;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        // DRPRG_HIP_RCCL_LIB names the one library to try (a host with its own build of RCCL; the test suite points it at a
        // file that does not exist to reach the no-library paths: -ENODEV from the communicator entries, the peer copy +
        // add kernel in drprg_hip_reduce)
        const char* forced = getenv("DRPRG_HIP_RCCL_LIB");
        std::string first_err;
        for (const char* name : { forced, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            if (!name || !*name) continue;
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
            const char* e = dlerror(); // one call: dlerror() clears the message it returns
            if (first_err.empty()) first_err = e ? e : "?";
            if (forced) break;
        }
        if (!h) {
            error = std::string(forced && *forced ? forced : "librccl.so.1") + " not found: " + (first_err.empty() ? "?" : first_err);
            return;
        }
        auto bind = [&](const char* sym, void** slot) {
            *slot = dlsym(h, sym);
            if (!*slot && error.empty()) error = std::string("librccl lacks ") + sym;
        };
        bind("ncclGetUniqueId", (void**)&api.GetUniqueId);
        bind("ncclCommInitRank", (void**)&api.CommInitRank);
        bind("ncclCommInitAll", (void**)&api.CommInitAll);
        bind("ncclCommDestroy", (void**)&api.CommDestroy);
        bind("ncclAllReduce", (void**)&api.AllReduce);
        bind("ncclReduce", (void**)&api.Reduce);
        bind("ncclGroupStart", (void**)&api.GroupStart);
        bind("ncclGroupEnd", (void**)&api.GroupEnd);
        bind("ncclGetErrorString", (void**)&api.GetErrorString);
        ok = error.empty();
    });
    if (!ok && why) *why = error;
    return ok ? &api : nullptr;
}

} // namespace drprg
