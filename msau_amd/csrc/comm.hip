// Data-parallel gradient exchange in the C ABI: RCCL sum all-reduce of one contiguous bucket of the flat fp32
// gradient, enqueued on a HIP stream like every other launch of the sequence (sequence.hip: MSAU_OP_ALLREDUCE) -- no Python,
// no torch.distributed call and no extra stream hop inside the backward sweep.  The reference has no distributed code
// (SURVEY 2.1); the exchange it would need sits between loss.backward() and optimizer.step()
// (train_chargrid_funsd_msau.py:57-59).
//
// RCCL is resolved at run time (dlopen): the copy the process already holds (PyTorch ships one) is re-used, and the library
// stays loadable -- and every kernel usable -- on a machine without RCCL.  One process per GPU; the communicator is created
// from a 128-byte unique id that rank 0 draws (msau_comm_unique_id) and the caller ships to the other ranks by whatever
// channel it has (msau_amd/model.py broadcasts it through the torch.distributed process group it was given).
#include "msau_common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {

typedef struct { char internal[128]; } nccl_uid;               // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclFloat32 = 7 };     // rccl.h: ncclResult_t / ncclRedOp_t / ncclDataType_t values

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(nccl_uid*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    char why[256] = "not tried";                               // why the library is unavailable (dlerror() may be NULL, and is per-thread)
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_resolve() {
    Rccl& r = g_rccl;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (int pass = 0; pass < 2 && !r.lib; ++pass)             // first: a copy that is already loaded (RTLD_NOLOAD)
        for (const char* n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
    if (!r.lib) {
        const char* e = dlerror();
        snprintf(r.why, sizeof(r.why), "%s", e ? e : "dlopen(librccl.so) failed without a message");
        return;
    }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
        snprintf(r.why, sizeof(r.why), "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
        r.lib = nullptr;
    }
}

// resolved once per process, whichever thread asks first (std::call_once: a second caller waits for the first to finish)
const Rccl* rccl() {
    std::call_once(g_rccl_once, rccl_resolve);
    return g_rccl.lib ? &g_rccl : nullptr;
}

int rccl_fail(const Rccl* r, const char* what, int rc) {
    return msau_set_error(MSAU_ERR_HIP, "%s: RCCL error %d (%s)", what, rc, r->GetErrorString ? r->GetErrorString(rc) : "?");
}

}  // namespace

extern "C" int msau_comm_available(void) { return rccl() != nullptr; }

extern "C" int msau_comm_unique_id(void* id_out, int bytes) {
    MSAU_CHECK_ARG(id_out && bytes == (int)sizeof(nccl_uid), "comm_unique_id: the id is %d bytes", (int)sizeof(nccl_uid));
    const Rccl* r = rccl();
    if (!r) return msau_set_error(MSAU_ERR_HIP, "comm_unique_id: librccl.so could not be loaded (%s)", g_rccl.why);
    nccl_uid id;
    int rc = r->GetUniqueId(&id);
    if (rc != kNcclSuccess) return rccl_fail(r, "comm_unique_id", rc);
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int msau_comm_init(void** comm_out, int world, int rank, const void* id, int bytes) {
    MSAU_CHECK_ARG(comm_out && id && bytes == (int)sizeof(nccl_uid) && world >= 1 && rank >= 0 && rank < world, "comm_init: bad args");
    const Rccl* r = rccl();
    if (!r) return msau_set_error(MSAU_ERR_HIP, "comm_init: librccl.so could not be loaded (%s)", g_rccl.why);
    nccl_uid uid;
    memcpy(&uid, id, sizeof(uid));
    nccl_comm c = nullptr;
    int rc = r->CommInitRank(&c, world, uid, rank);              // collective: every rank calls it, on its own current device
    if (rc != kNcclSuccess) return rccl_fail(r, "comm_init", rc);
    *comm_out = c;
    return 0;
}

extern "C" int msau_comm_destroy(void* comm) {
    if (!comm) return 0;
    const Rccl* r = rccl();
    if (!r) return 0;
    int rc = r->CommDestroy(static_cast<nccl_comm>(comm));
    return rc == kNcclSuccess ? 0 : rccl_fail(r, "comm_destroy", rc);
}

extern "C" int msau_allreduce_bucket(void* stream, void* comm, float* buf, int64_t count) {
    MSAU_CHECK_ARG(comm && buf && count > 0, "allreduce_bucket: bad args");
    const Rccl* r = rccl();
    if (!r) return msau_set_error(MSAU_ERR_HIP, "allreduce_bucket: RCCL is not loaded");
    int rc = r->AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, static_cast<nccl_comm>(comm), static_cast<hipStream_t>(stream));
    return rc == kNcclSuccess ? 0 : rccl_fail(r, "allreduce_bucket", rc);
}
