// Data-parallel gradient exchange of the C ABI (include/tfusion.h: tf_comm_*, tf_allreduce_bucket): one RCCL communicator per
// process (one process per GPU), in-place SUM all-reduce of a contiguous fp32 slice of the flat gradient buffer on the stream
// the caller names, so a layer's exchange runs behind the rest of the backward.  Replaces what the reference gets from
// Lightning's strategy="ddp" (runner/run_experiment.py:452: torch DDP's bucketed all-reduce).
//
// RCCL is bound at RUN time (dlopen of librccl.so.1, the copy already in the process when torch is loaded): the library keeps
// no link-time dependency on it, loads on a machine without RCCL, and every tf_comm_* entry reports a plain error there.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include "tf_kernels.h"

namespace {

// the five RCCL entry points used, with the ABI of rccl.h (ncclResult_t = int, ncclComm_t = opaque pointer, ncclUniqueId = 128 bytes
// passed BY VALUE to ncclCommInitRank, ncclFloat32 = 7, ncclSum = 0)
struct UniqueId { char internal[TF_COMM_ID_BYTES]; };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kFloat32 = 7, kSum = 0;

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
  char why[256] = "";
};

std::once_flag g_once;
Rccl g_rccl;

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle != nullptr) break;
  }
  if (g_rccl.handle == nullptr) {
    snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so.1 not loadable (%s)", dlerror());
    return;
  }
  g_rccl.get_unique_id = (GetUniqueIdFn)dlsym(g_rccl.handle, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (CommInitRankFn)dlsym(g_rccl.handle, "ncclCommInitRank");
  g_rccl.all_reduce = (AllReduceFn)dlsym(g_rccl.handle, "ncclAllReduce");
  g_rccl.comm_destroy = (CommDestroyFn)dlsym(g_rccl.handle, "ncclCommDestroy");
  g_rccl.error_string = (GetErrorStringFn)dlsym(g_rccl.handle, "ncclGetErrorString");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) {
    snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
    g_rccl.handle = nullptr;
  }
}

// 0 when RCCL is bound; otherwise TF_ERR_NO_RCCL with the reason in tf_last_error()
int need_rccl(const char* what) {
  std::call_once(g_once, load_rccl);
  if (g_rccl.handle != nullptr) return 0;
  char msg[400];
  snprintf(msg, sizeof(msg), "%s: %s", what, g_rccl.why);
  tf_set_error_msg(msg);
  return TF_ERR_NO_RCCL;
}

int rccl_fail(const char* what, int rc) {
  char msg[400];
  snprintf(msg, sizeof(msg), "%s: RCCL error %d (%s)", what, rc, g_rccl.error_string ? g_rccl.error_string(rc) : "?");
  tf_set_error_msg(msg);
  return TF_ERR_RCCL;
}

}  // namespace

struct TfComm {
  void* comm;
  int world, rank, device;
  long long calls, elems;      // collectives issued / fp32 elements reduced so far (every rank must report the same)
};

extern "C" {

int tf_comm_unique_id(void* id) {
  if (id == nullptr) { tf_set_error_msg("tf_comm_unique_id: null id"); return -1; }
  if (const int rc = need_rccl("tf_comm_unique_id")) return rc;
  UniqueId u;
  memset(&u, 0, sizeof(u));
  if (const int rc = g_rccl.get_unique_id(&u)) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id, &u, sizeof(u));
  return 0;
}

int tf_comm_create(TfComm** out, const void* id, int world, int rank) {
  if (out == nullptr || id == nullptr || world < 1 || rank < 0 || rank >= world) {
    tf_set_error_msg("tf_comm_create: invalid argument (need out, id, 0 <= rank < world)");
    return -1;
  }
  *out = nullptr;
  if (const int rc = need_rccl("tf_comm_create")) return rc;
  int dev = -1;
  if (const hipError_t e = hipGetDevice(&dev)) {
    char msg[200];
    snprintf(msg, sizeof(msg), "tf_comm_create: hipGetDevice: %s", hipGetErrorString(e));
    tf_set_error_msg(msg);
    return (int)e;
  }
  UniqueId u;
  memcpy(&u, id, sizeof(u));
  void* comm = nullptr;
  if (const int rc = g_rccl.comm_init_rank(&comm, world, u, rank)) return rccl_fail("ncclCommInitRank", rc);
  TfComm* c = new TfComm{comm, world, rank, dev, 0, 0};
  *out = c;
  return 0;
}

int tf_allreduce_bucket(TfComm* c, float* buf, long long n, tf_stream_t stream) {
  if (c == nullptr || c->comm == nullptr || (buf == nullptr && n > 0) || n < 0) {
    tf_set_error_msg("tf_allreduce_bucket: invalid argument");
    return -1;
  }
  if (n == 0) return 0;
  if (const int rc = g_rccl.all_reduce(buf, buf, (size_t)n, kFloat32, kSum, c->comm, (hipStream_t)stream)) return rccl_fail("ncclAllReduce", rc);
  c->calls += 1;
  c->elems += n;
  return 0;
}

int tf_comm_stats(const TfComm* c, int* world, int* rank, long long* calls, long long* elems) {
  if (c == nullptr) { tf_set_error_msg("tf_comm_stats: null communicator"); return -1; }
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  if (calls) *calls = c->calls;
  if (elems) *elems = c->elems;
  return 0;
}

int tf_comm_destroy(TfComm* c) {
  if (c == nullptr) { tf_set_error_msg("tf_comm_destroy: null communicator"); return -1; }
  int rc = 0;
  if (c->comm != nullptr && g_rccl.comm_destroy != nullptr) rc = g_rccl.comm_destroy(c->comm);
  delete c;
  return rc ? rccl_fail("ncclCommDestroy", rc) : 0;
}

}  // extern "C"
