// rccl_dyn.cpp -- see rccl_dyn.h
#include "rccl_dyn.h"
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
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) {
            error = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "?");
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
