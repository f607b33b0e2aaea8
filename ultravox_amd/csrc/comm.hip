// uvx_comm_*: the data-parallel exchange of the path behind the C ABI.
//
// Reference semantics: torch DDP under HF Trainer / accelerate (train.py:126-130, :282-288) - the gradients of the trainable
// parameters are all-reduced (sum) over the ranks and divided by the world size, ONE collective per optimizer step over
// the flat f32 bucket.  On MI355X the collective is RCCL over xGMI: one communicator per process (one process per GPU).
//
// RCCL is bound at RUN TIME (dlopen + dlsym of the six entry points used), not linked: libuvx.so keeps its dependency list
// (libamdhip64 + libc / libstdc++ / libdl), a host that never trains data-parallel needs no RCCL at all, and a process
// that already holds RCCL (PyTorch-ROCm loads its own librccl.so) shares that copy instead of getting a second one.
// The unique id is plain bytes: rank 0 creates it (uvx_comm_unique_id), the caller hands it to the other ranks by whatever
// channel it has (the Python trainer: torch.distributed's store; a native host: a file, MPI, a socket).
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include "common.h"
#include "kernels.h"
#include "../../include/uvx.h"

namespace {

// the slice of rccl.h this file uses (types restated so that no RCCL header is needed to build)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[UVX_COMM_ID_BYTES]; } ncclUniqueId;   // NCCL_UNIQUE_ID_BYTES = 128
enum { ncclSuccess = 0 };
enum { ncclFloat32 = 7 };
enum { ncclSumOp = 0 };

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;

const char* load_rccl() {   // nullptr on success, else what failed
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rccl.handle) return nullptr;
  void* h = nullptr;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {   // a copy already in the process (PyTorch's) first
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (int i = 0; !h && i < 4; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) return "librccl.so not found (set LD_LIBRARY_PATH to a ROCm lib directory or import torch first)";
  Rccl r;
  r.handle = h;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.GetVersion = (decltype(r.GetVersion))dlsym(h, "ncclGetVersion");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) return "librccl.so lacks the nccl* entry points";
  g_rccl = r;
  return nullptr;
}

const char* rccl_err(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"; }

__global__ void scale_f32_k(float* __restrict__ x, long long n, float s) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 v = *reinterpret_cast<float4*>(x + i);
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    *reinterpret_cast<float4*>(x + i) = v;
  } else {
    for (long long j = i; j < n; ++j) x[j] *= s;
  }
}

}  // namespace

struct uvx_comm {
  ncclComm_t comm;
  int rank, world;
};

extern "C" int32_t uvx_comm_unique_id(uint8_t* id) {
  UVX_CHECK(id != nullptr, UVX_ERR_INVALID, "uvx_comm_unique_id: null id buffer");
  const char* e = load_rccl();
  UVX_CHECK(e == nullptr, UVX_ERR_UNSUPPORTED, "uvx_comm: %s", e);
  ncclUniqueId u;
  const int rc = g_rccl.GetUniqueId(&u);
  UVX_CHECK(rc == ncclSuccess, UVX_ERR_RUNTIME, "ncclGetUniqueId: %s", rccl_err(rc));
  memcpy(id, u.internal, UVX_COMM_ID_BYTES);
  return UVX_OK;
}

extern "C" int32_t uvx_comm_init(uvx_comm_t** out, int32_t rank, int32_t world, const uint8_t* id) {
  UVX_CHECK(out != nullptr && id != nullptr, UVX_ERR_INVALID, "uvx_comm_init: null argument");
  UVX_CHECK(world >= 1 && rank >= 0 && rank < world, UVX_ERR_INVALID, "uvx_comm_init: rank %d of world %d", rank, world);
  const char* e = load_rccl();
  UVX_CHECK(e == nullptr, UVX_ERR_UNSUPPORTED, "uvx_comm: %s", e);
  ncclUniqueId u;
  memcpy(u.internal, id, UVX_COMM_ID_BYTES);
  ncclComm_t c = nullptr;
  const int rc = g_rccl.CommInitRank(&c, world, u, rank);     // binds the communicator to the CURRENT hip device
  UVX_CHECK(rc == ncclSuccess, UVX_ERR_RUNTIME, "ncclCommInitRank(rank %d / %d): %s", rank, world, rccl_err(rc));
  *out = new uvx_comm{c, rank, world};
  return UVX_OK;
}

extern "C" int32_t uvx_comm_world_size(const uvx_comm_t* c) { return c ? c->world : 1; }

extern "C" int32_t uvx_comm_version(void) {
  if (load_rccl() != nullptr || !g_rccl.GetVersion) return 0;
  int v = 0;
  return g_rccl.GetVersion(&v) == ncclSuccess ? v : 0;
}

// buf[i] = (sum over ranks of buf[i]) * scale, in place; scale = 1 / world gives torch DDP's gradient mean.  Asynchronous on
// `stream` (the caller's compute stream, or a side stream it orders with events to overlap the exchange with compute).
extern "C" int32_t uvx_comm_allreduce_f32(uvx_comm_t* c, void* stream, float* buf, int64_t n, float scale) {
  UVX_CHECK(c != nullptr && buf != nullptr && n >= 0, UVX_ERR_INVALID, "uvx_comm_allreduce_f32: bad argument");
  if (n == 0) return UVX_OK;
  hipStream_t st = (hipStream_t)stream;
  if (c->world > 1) {
    const int rc = g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSumOp, c->comm, st);
    UVX_CHECK(rc == ncclSuccess, UVX_ERR_RUNTIME, "ncclAllReduce: %s", rccl_err(rc));
  }
  if (scale != 1.0f) {
    const long long groups = (n + 3) / 4;
    hipLaunchKernelGGL(scale_f32_k, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, buf, (long long)n, scale);
    UVX_LAUNCH_CHECK();
  }
  return UVX_OK;
}

extern "C" int32_t uvx_comm_destroy(uvx_comm_t* c) {
  if (!c) return UVX_OK;
  const int rc = g_rccl.CommDestroy ? g_rccl.CommDestroy(c->comm) : 0;
  delete c;
  UVX_CHECK(rc == ncclSuccess, UVX_ERR_RUNTIME, "ncclCommDestroy: %s", rccl_err(rc));
  return UVX_OK;
}
