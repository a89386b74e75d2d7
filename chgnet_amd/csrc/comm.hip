// comm.hip -- the exchange steps of the multi-GPU path straight on RCCL: the all-gather of per-structure
// energies after a sharded sweep and the all-reduce of the 1.65 MB parameter gradient of a data-parallel train step
// (reference: none -- chgnet is single-device; trainer.py:399-411 is the step the all-reduce slots into).
//
// One communicator per process = per GPU.  librccl is opened at run time (dlopen): the engine library keeps loading on
// machines without RCCL; the copy next to the HIP runtime in use is preferred (see rccl()).  Rendezvous (handing rank 0's ncclUniqueId to the other ranks) is the caller's job -- the
// Python host side does it over a TCP socket on MASTER_ADDR (chgnet_amd/distributed.py), a launcher may use anything.
//
// The handful of RCCL types used here are declared locally (they are ABI-stable NCCL 2 types): the engine library builds on
// ROCm installs without the RCCL headers, as it loads without librccl.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;     // ncclFloat32
typedef enum { ncclSum = 0 } ncclRedOp_t;

#include <cstring>
#include <string>

#include "chgnet_hip.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    // RCCL must sit on the HIP runtime this library runs on.  A process that imported torch FIRST runs on torch's bundled
    // runtime (the loader resolved libamdhip64 to the copy already mapped) and must use torch's bundled librccl; one that
    // loaded this library first runs on the system runtime, and a torch imported later only ADDS its librccl to the process --
    // picking that copy up by name gave "ncclCommInitRank: unhandled cuda error" (two ROCm releases in one call chain).
    std::string beside;
    Dl_info hip_at{};
    if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &hip_at) && hip_at.dli_fname) {
      beside = hip_at.dli_fname;
      const size_t slash = beside.rfind('/');
      beside = slash == std::string::npos ? std::string() : beside.substr(0, slash);
    }
    const std::string candidates[] = {beside.empty() ? std::string() : beside + "/librccl.so.1", beside.empty() ? std::string() : beside + "/librccl.so",
                                      "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    for (const std::string& n : candidates) {
      if (n.empty()) continue;
      if ((x.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!x.handle) { x.error = std::string("librccl not found: ") + dlerror(); return x; }
    auto sym = [&](const char* s) { void* p = dlsym(x.handle, s); if (!p && x.error.empty()) x.error = std::string("librccl lacks ") + s; return p; };
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
    x.CommCount = reinterpret_cast<decltype(x.CommCount)>(dlsym(x.handle, "ncclCommCount"));   // optional: chg_comm_info falls back to the size it was created with
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    return x;
  }();
  return r;
}

thread_local std::string g_comm_error;

}  // namespace

struct chg_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  float* buf = nullptr;        // device staging: [send | recv], grow-only
  size_t buf_floats = 0;
  std::string err;
};

namespace {

int fail(chg_comm* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  g_comm_error = msg;
  return code;
}

int check_nccl(chg_comm* c, ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return CHG_OK;
  return fail(c, CHG_EHIP, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error"));
}

int check_hip(chg_comm* c, hipError_t e, const char* what) {
  if (e == hipSuccess) return CHG_OK;
  return fail(c, CHG_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

int reserve(chg_comm* c, size_t floats) {
  if (floats <= c->buf_floats) return CHG_OK;
  if (c->buf) hipFree(c->buf);
  c->buf = nullptr; c->buf_floats = 0;
  const size_t want = floats + floats / 4 + 1024;
  if (hipMalloc(reinterpret_cast<void**>(&c->buf), want * sizeof(float)) != hipSuccess) return fail(c, CHG_ENOMEM, "chg_comm: staging allocation failed");
  c->buf_floats = want;
  return CHG_OK;
}

}  // namespace

extern "C" {

int chg_comm_unique_id(uint8_t* id_out) {
  if (!id_out) return CHG_EINVAL;
  if (!rccl().error.empty()) return fail(nullptr, CHG_EUNSUPPORTED, rccl().error);
  ncclUniqueId id;
  const int s = check_nccl(nullptr, rccl().GetUniqueId(&id), "ncclGetUniqueId");
  if (s != CHG_OK) return s;
  static_assert(sizeof(ncclUniqueId) == CHG_COMM_ID_BYTES, "chgnet_hip.h: CHG_COMM_ID_BYTES");
  std::memcpy(id_out, &id, sizeof(id));
  return CHG_OK;
}

int chg_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device, chg_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return CHG_EINVAL;
  if (!rccl().error.empty()) return fail(nullptr, CHG_EUNSUPPORTED, rccl().error);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(nullptr, CHG_ENODEV, "chg_comm_create: no such device");
  chg_comm* c = new (std::nothrow) chg_comm();
  if (!c) return CHG_ENOMEM;
  c->rank = rank; c->world = world; c->device = device;
  int s = check_hip(c, hipSetDevice(device), "hipSetDevice");
  if (s == CHG_OK) s = check_hip(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate");
  if (s == CHG_OK) {
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    s = check_nccl(c, rccl().CommInitRank(&c->comm, world, uid, rank), "ncclCommInitRank");
  }
  if (s != CHG_OK) {
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return s;
  }
  *out = c;
  return CHG_OK;
}

int chg_comm_all_gather_f32(chg_comm* c, const float* send, int64_t count, float* recv) {
  if (!c || count < 0 || (count > 0 && (!send || !recv))) return CHG_EINVAL;
  if (count == 0) return CHG_OK;
  int s = check_hip(c, hipSetDevice(c->device), "hipSetDevice");
  if (s == CHG_OK) s = reserve(c, (size_t)count * ((size_t)c->world + 1));
  if (s != CHG_OK) return s;
  float* d_send = c->buf;
  float* d_recv = c->buf + count;
  s = check_hip(c, hipMemcpyAsync(d_send, send, sizeof(float) * count, hipMemcpyHostToDevice, c->stream), "upload");
  if (s == CHG_OK) s = check_nccl(c, rccl().AllGather(d_send, d_recv, (size_t)count, ncclFloat, c->comm, c->stream), "ncclAllGather");
  if (s == CHG_OK) s = check_hip(c, hipMemcpyAsync(recv, d_recv, sizeof(float) * count * c->world, hipMemcpyDeviceToHost, c->stream), "download");
  if (s == CHG_OK) s = check_hip(c, hipStreamSynchronize(c->stream), "synchronize");
  return s;
}

int chg_comm_all_reduce_sum_f32(chg_comm* c, float* data, int64_t count) {
  if (!c || count < 0 || (count > 0 && !data)) return CHG_EINVAL;
  if (count == 0) return CHG_OK;
  int s = check_hip(c, hipSetDevice(c->device), "hipSetDevice");
  if (s == CHG_OK) s = reserve(c, (size_t)count);
  if (s != CHG_OK) return s;
  s = check_hip(c, hipMemcpyAsync(c->buf, data, sizeof(float) * count, hipMemcpyHostToDevice, c->stream), "upload");
  if (s == CHG_OK) s = check_nccl(c, rccl().AllReduce(c->buf, c->buf, (size_t)count, ncclFloat, ncclSum, c->comm, c->stream), "ncclAllReduce");
  if (s == CHG_OK) s = check_hip(c, hipMemcpyAsync(data, c->buf, sizeof(float) * count, hipMemcpyDeviceToHost, c->stream), "download");
  if (s == CHG_OK) s = check_hip(c, hipStreamSynchronize(c->stream), "synchronize");
  return s;
}

// ---- device-pointer forms: enqueued on the caller's stream (the engine stream: energies and the gradient blob are already
// in HBM), no host bounce, no synchronisation -----------------------------------------------------------------------------
int chg_comm_all_gather_f32_device(chg_comm* c, const float* d_send, int64_t count, float* d_recv, void* stream) {
  if (!c || count < 0 || (count > 0 && (!d_send || !d_recv))) return CHG_EINVAL;
  if (count == 0) return CHG_OK;
  int s = check_hip(c, hipSetDevice(c->device), "hipSetDevice");
  if (s == CHG_OK) s = check_nccl(c, rccl().AllGather(d_send, d_recv, (size_t)count, ncclFloat, c->comm, static_cast<hipStream_t>(stream)), "ncclAllGather");
  return s;
}

int chg_comm_all_reduce_sum_f32_device(chg_comm* c, float* d_data, int64_t count, void* stream) {
  if (!c || count < 0 || (count > 0 && !d_data)) return CHG_EINVAL;
  if (count == 0) return CHG_OK;
  int s = check_hip(c, hipSetDevice(c->device), "hipSetDevice");
  if (s == CHG_OK) s = check_nccl(c, rccl().AllReduce(d_data, d_data, (size_t)count, ncclFloat, ncclSum, c->comm, static_cast<hipStream_t>(stream)), "ncclAllReduce");
  return s;
}

int chg_comm_reserve(chg_comm* c, int64_t floats, float** device_ptr) {
  if (!c || floats < 0 || !device_ptr) return CHG_EINVAL;
  int s = check_hip(c, hipSetDevice(c->device), "hipSetDevice");
  if (s == CHG_OK) s = reserve(c, (size_t)floats);
  if (s == CHG_OK) *device_ptr = c->buf;
  return s;
}

int chg_comm_info(chg_comm* c, int32_t* rank, int32_t* nranks, int32_t* device) {
  if (!c) return CHG_EINVAL;
  int n = c->world;
  if (rccl().CommCount) {
    const int s = check_nccl(c, rccl().CommCount(c->comm, &n), "ncclCommCount");
    if (s != CHG_OK) return s;
  }
  if (rank) *rank = c->rank;
  if (nranks) *nranks = n;           // what RCCL reports for this communicator
  if (device) *device = c->device;
  return CHG_OK;
}

int chg_comm_barrier(chg_comm* c) {
  if (!c) return CHG_EINVAL;
  float one = 1.0f;
  return chg_comm_all_reduce_sum_f32(c, &one, 1);
}

int chg_comm_destroy(chg_comm* c) {
  if (!c) return CHG_OK;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->comm && rccl().CommDestroy) rccl().CommDestroy(c->comm);
  if (c->buf) hipFree(c->buf);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return CHG_OK;
}

const char* chg_comm_last_error(const chg_comm* c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

}  // extern "C"
