// sr_rccl.cpp -- dlopen binding of librccl (see sr_rccl.h).
#include "sr_rccl.h"

#include <dlfcn.h>

#include <mutex>

namespace sr {
namespace {

RcclApi g_api{};
bool g_ok = false;
std::string g_why;
std::once_flag g_once;

void bind() {
    // by soname first: an already loaded librccl (PyTorch's) is found that way; then the ROCm install
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { g_why = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "not found"); return; }
    struct { const char* name; void** slot; } syms[] = {
        {"ncclGetUniqueId", (void**)&g_api.GetUniqueId},   {"ncclCommInitRank", (void**)&g_api.CommInitRank},
        {"ncclCommInitAll", (void**)&g_api.CommInitAll},   {"ncclCommDestroy", (void**)&g_api.CommDestroy},
        {"ncclSend", (void**)&g_api.Send},                 {"ncclRecv", (void**)&g_api.Recv},
        {"ncclGroupStart", (void**)&g_api.GroupStart},     {"ncclGroupEnd", (void**)&g_api.GroupEnd},
        {"ncclGetErrorString", (void**)&g_api.GetErrorString},
    };
    for (auto& s : syms) {
        *s.slot = dlsym(h, s.name);
        if (!*s.slot) { g_why = std::string("librccl.so lacks ") + s.name; return; }
    }
    g_ok = true;
}

}  // namespace

const RcclApi* rccl_api(std::string* why) {
    std::call_once(g_once, bind);
    if (!g_ok && why) *why = g_why;
    return g_ok ? &g_api : nullptr;
}

}  // namespace sr
