// The V2X link as an RCCL all-gather behind the C ABI (SURVEY.md §8(b) `qv2x_allgather_codes`): one fixed-size payload per agent
// (uint8 code planes + the sender's pose), every rank receives all of them.  The reference has no inference-time collective:
// its agents are rows of one batch (opencood/models/heter_model_baseline.py:216) and fusion_in_one.py:131-151 regroups them.
//
// RCCL is bound at run time (dlopen of the copy already in the process -- PyTorch's -- or of /opt/rocm's), so libqv2x.so has no
// link-time dependency on it and single-GPU users never touch it.  The communicator is the only state, it is opt-in and owned by
// the caller (qv2x_comm_init / qv2x_comm_destroy).
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace qv2x {
namespace {

struct UniqueId { char bytes[QV2X_COMM_ID_BYTES]; };                    // ncclUniqueId: 128 opaque bytes, passed by value
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllGatherFn all_gather = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        void* h = nullptr;
        for (int pass = 0; pass < 2 && !h; ++pass)                      // first the copy already loaded (PyTorch's), then the disk
            for (const char* n : names)
                if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0)))) break;
        if (!h) return;
        r.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
        r.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
        r.all_gather = (AllGatherFn)dlsym(h, "ncclAllGather");
        r.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
        r.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
        r.ok = r.get_unique_id && r.comm_init_rank && r.all_gather && r.comm_destroy;
    });
    return r;
}

int rccl_check(int rc, const char* what) {
    if (rc == 0) return QV2X_OK;
    const Rccl& r = rccl();
    return fail(-2000 - rc, "%s: RCCL error %d (%s)", what, rc, r.error_string ? r.error_string(rc) : "?");
}

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_comm_unique_id(void* id) {
    using namespace qv2x;
    if (!id) return fail(QV2X_EINVAL, "qv2x_comm_unique_id: null pointer");
    if (!rccl().ok) return fail(QV2X_EINVAL, "qv2x_comm_unique_id: librccl.so not found in the process or on the library path");
    return rccl_check(rccl().get_unique_id((UniqueId*)id), "ncclGetUniqueId");
}

extern "C" int qv2x_comm_init(const void* id, int world, int rank, void** comm) {
    using namespace qv2x;
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) return fail(QV2X_EINVAL, "qv2x_comm_init: bad arguments (world=%d rank=%d)", world, rank);
    if (!rccl().ok) return fail(QV2X_EINVAL, "qv2x_comm_init: librccl.so not found in the process or on the library path");
    UniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    *comm = nullptr;
    return rccl_check(rccl().comm_init_rank(comm, world, uid, rank), "ncclCommInitRank");
}

extern "C" int qv2x_allgather_codes(void* comm, const uint8_t* send, uint8_t* recv, int64_t bytes_per_rank, void* stream) {
    using namespace qv2x;
    if (!comm || !send || !recv || bytes_per_rank <= 0) return fail(QV2X_EINVAL, "qv2x_allgather_codes: bad arguments");
    if (!rccl().ok) return fail(QV2X_EINVAL, "qv2x_allgather_codes: librccl.so not loaded");
    return rccl_check(rccl().all_gather(send, recv, (size_t)bytes_per_rank, /* ncclUint8 */ 1, comm, (hipStream_t)stream), "ncclAllGather");
}

extern "C" int qv2x_comm_destroy(void* comm) {
    using namespace qv2x;
    if (!comm) return QV2X_OK;
    if (!rccl().ok) return fail(QV2X_EINVAL, "qv2x_comm_destroy: librccl.so not loaded");
    return rccl_check(rccl().comm_destroy(comm), "ncclCommDestroy");
}
