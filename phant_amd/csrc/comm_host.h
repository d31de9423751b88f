// comm_host.h -- what comm.hip needs from the host side of a multi-GPU process: RCCL's types and enum values from its
// own header, its entry points resolved at run time, and one host thread per device for staging work.
//
// RCCL is looked up at run time (a symbol the process already has -- e.g. PyTorch's -- or librccl.so.1): the library
// links and loads without it, a single-device comm never needs it.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>  // ncclComm_t, ncclResult_t, ncclUint32, ncclSum ... (no symbol of it is linked)
#else
// A ROCm installation without RCCL's development headers still builds the library (a single-device comm never needs RCCL):
// the handful of names comm.hip uses, as rccl.h 2.x declares them.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include <string>
#include <thread>
#include <vector>

#include "host_threads.h"

namespace phant {

// the six RCCL entry points comm.hip uses, with the signatures rccl.h declares
struct Rccl {
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok() const { return comm_init_all && comm_destroy && all_reduce && group_start && group_end; }
};

inline bool load_rccl(Rccl& r, std::string& err) {
    void* h = nullptr;
    // a copy the process already carries (PyTorch ships its own) wins: two RCCLs in one process is asking for trouble
    void* probe = dlsym(RTLD_DEFAULT, "ncclCommInitAll");
    if (!probe) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) {
            err = "RCCL not found (librccl.so.1)";
            return false;
        }
    }
    auto sym = [&](const char* n) -> void* { return h ? dlsym(h, n) : dlsym(RTLD_DEFAULT, n); };
    r.comm_init_all = reinterpret_cast<decltype(r.comm_init_all)>(sym("ncclCommInitAll"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(sym("ncclCommDestroy"));
    r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(sym("ncclAllReduce"));
    r.group_start = reinterpret_cast<decltype(r.group_start)>(sym("ncclGroupStart"));
    r.group_end = reinterpret_cast<decltype(r.group_end)>(sym("ncclGroupEnd"));
    r.error_string = reinterpret_cast<decltype(r.error_string)>(sym("ncclGetErrorString"));
    if (!r.ok()) {
        err = "RCCL lacks an entry point";
        return false;
    }
    return true;
}

// work(d) for every device d < n: device 0 on the calling thread, the others on a host thread each (packing a shard and
// staging it is host work; the devices' streams run independently anyway)
// Nothing leaves a thread (host_threads.h): a shard that ran out of host memory comes back as std::bad_alloc on the calling thread,
// after every device's thread has been joined.
template <class F>
inline void for_each_device(uint32_t n, F&& work) {
    parallel_guarded(n, [&work](size_t d) { work((uint32_t)d); });
}

}  // namespace phant
