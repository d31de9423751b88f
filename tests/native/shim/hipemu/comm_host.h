// TEST INFRASTRUCTURE (tests/emu.py): what phant_amd/csrc/comm_host.h provides, for the host-emulated build -- the names
// comm.hip uses from rccl.h, an in-process sum in place of the collective (same call pattern: the all-reduce of every rank
// is noted between group_start and group_end and carried out at the end), devices one after the other (the host emulation
// of the HIP runtime is single-threaded).  Never part of libphant_gpu.so.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

typedef void* ncclComm_t;
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
enum ncclDataType_t { ncclUint32 = 3 };
enum ncclRedOp_t { ncclSum = 0 };

namespace phant {

struct Rccl {
    ncclResult_t (*comm_init_all)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*group_start)() = nullptr;
    ncclResult_t (*group_end)() = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    bool ok() const { return comm_init_all && comm_destroy && all_reduce && group_start && group_end; }
};

namespace emu_rccl {
struct Call {
    const uint32_t* send;
    uint32_t* recv;
    size_t count;
};
inline std::vector<Call> calls;
inline ncclResult_t init_all(ncclComm_t* c, int n, const int*) {
    for (int i = 0; i < n; ++i) c[i] = reinterpret_cast<ncclComm_t>((uintptr_t)(i + 1));
    return 0;
}
inline ncclResult_t destroy(ncclComm_t) { return 0; }
inline ncclResult_t all_reduce(const void* s, void* r, size_t count, ncclDataType_t dtype, ncclRedOp_t op, ncclComm_t, hipStream_t) {
    if (dtype != ncclUint32 || op != ncclSum) return 1;
    calls.push_back({static_cast<const uint32_t*>(s), static_cast<uint32_t*>(r), count});
    return 0;
}
inline ncclResult_t group_start() {
    calls.clear();
    return 0;
}
inline ncclResult_t group_end() {
    if (calls.empty()) return 0;
    const size_t count = calls[0].count;
    std::vector<uint32_t> sum(count, 0);
    for (const Call& c : calls) {
        if (c.count != count) return 1;
        for (size_t i = 0; i < count; ++i) sum[i] += c.send[i];
    }
    for (const Call& c : calls) std::memcpy(c.recv, sum.data(), count * 4);
    calls.clear();
    return 0;
}
inline const char* error_string(ncclResult_t) { return "emulated RCCL error"; }
}  // namespace emu_rccl

inline bool load_rccl(Rccl& r, std::string&) {
    r.comm_init_all = emu_rccl::init_all;
    r.comm_destroy = emu_rccl::destroy;
    r.all_reduce = emu_rccl::all_reduce;
    r.group_start = emu_rccl::group_start;
    r.group_end = emu_rccl::group_end;
    r.error_string = emu_rccl::error_string;
    return true;
}

template <class F>
inline void for_each_device(uint32_t n, F&& work) {
    for (uint32_t d = 0; d < n; ++d) work(d);
}

}  // namespace phant
