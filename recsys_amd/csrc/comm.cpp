// Data-parallel collectives of the training step, issued on the step's OWN stream (include/rsx.h "collectives").
//
// tf.distribute.MirroredStrategy runs its cross-replica sums as ops INSIDE the training graph (fm/fm.py:184-194: the
// all-reduce of the dense gradients and the aggregation of the IndexedSlices are nodes of the same session run as forward and
// backward).  The equivalent here is a collective that is a node of the step's HIP graph: RCCL's ncclAllGather / ncclAllReduce
// enqueued on the stream the kernels are launched on -- capturable, no helper thread, nothing polling events of a capturing
// stream (rounds 4-5 went through torch.distributed's ProcessGroupNCCL, whose watchdog thread aborted captured xdeepfm.py runs
// one time in three and forced the default back to eager collectives between graph segments: +19 .. +110 % per step at world 1).
//
// RCCL is bound at RUN time (dlopen + dlsym), not at link time: librsx.so must load on hosts without RCCL (the CPU test tier
// resolves every symbol of include/rsx.h), and a process that has torch loaded already holds a copy of librccl -- two copies
// of one collective library in one process is asking for trouble, so the copy that is already mapped is the one used.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <new>

#include "rsx.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

// The binding table is process-wide and immutable once filled (std::call_once): not "mutable state" in the sense of the
// ABI's re-entrancy rule -- every communicator is a caller-owned handle.
Rccl g_rccl;
std::once_flag g_once;
thread_local char t_err[256] = "";

template <class F>
bool sym(void* lib, const char* name, F& out) {
  out = reinterpret_cast<F>(dlsym(lib, name));
  return out != nullptr;
}

void bind() {
  static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* lib = nullptr;
  if (const char* p = getenv("RSX_RCCL_LIB")) lib = dlopen(p, RTLD_NOW | RTLD_LOCAL);
  for (int pass = 0; pass < 2 && lib == nullptr; ++pass)      // pass 0: a copy this process has mapped already (torch's)
    for (const char* n : names)
      if (lib == nullptr) lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
  if (lib == nullptr) return;
  Rccl& r = g_rccl;
  r.lib = lib;
  r.ok = sym(lib, "ncclGetVersion", r.GetVersion) && sym(lib, "ncclGetUniqueId", r.GetUniqueId) &&
         sym(lib, "ncclCommInitRank", r.CommInitRank) && sym(lib, "ncclCommDestroy", r.CommDestroy) &&
         sym(lib, "ncclCommAbort", r.CommAbort) && sym(lib, "ncclAllGather", r.AllGather) &&
         sym(lib, "ncclAllReduce", r.AllReduce) && sym(lib, "ncclGroupStart", r.GroupStart) &&
         sym(lib, "ncclGroupEnd", r.GroupEnd) && sym(lib, "ncclGetErrorString", r.GetErrorString);
}

const Rccl* rccl() {
  std::call_once(g_once, bind);
  return g_rccl.ok ? &g_rccl : nullptr;
}

struct Comm {
  ncclComm_t c;
  int rank, world;
};

int fail(const Rccl* r, ncclResult_t rc) {
  snprintf(t_err, sizeof t_err, "%s", r->GetErrorString(rc));
  return RSX_ECOMM;
}

#define RSX_NCCL(call)                          \
  do {                                          \
    const ncclResult_t rc_ = (call);            \
    if (rc_ != ncclSuccess) return fail(r, rc_); \
  } while (0)

}  // namespace

extern "C" int rsx_comm_available_h(int* version_out) {
  const Rccl* r = rccl();
  if (r == nullptr) return RSX_EUNSUPPORTED;
  int v = 0;
  RSX_NCCL(r->GetVersion(&v));
  if (version_out != nullptr) *version_out = v;
  return RSX_OK;
}

extern "C" const char* rsx_comm_last_error_h(void) { return t_err; }

extern "C" int rsx_comm_unique_id_h(void* id_out) {
  if (id_out == nullptr) return RSX_EINVAL;
  const Rccl* r = rccl();
  if (r == nullptr) return RSX_EUNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == RSX_COMM_UNIQUE_ID_BYTES, "RSX_COMM_UNIQUE_ID_BYTES");
  RSX_NCCL(r->GetUniqueId(static_cast<ncclUniqueId*>(id_out)));
  return RSX_OK;
}

extern "C" int rsx_comm_init_h(const void* unique_id, int rank, int world, rsx_comm_t* comm_out) {
  if (unique_id == nullptr || comm_out == nullptr || world < 1 || rank < 0 || rank >= world) return RSX_EINVAL;
  const Rccl* r = rccl();
  if (r == nullptr) return RSX_EUNSUPPORTED;
  ncclUniqueId id;
  __builtin_memcpy(&id, unique_id, sizeof id);
  Comm* c = new (std::nothrow) Comm{nullptr, rank, world};
  if (c == nullptr) return RSX_ECOMM;
  const ncclResult_t rc = r->CommInitRank(&c->c, world, id, rank);      // (binds to the caller's current HIP device)
  if (rc != ncclSuccess) {
    delete c;
    return fail(r, rc);
  }
  *comm_out = c;
  return RSX_OK;
}

extern "C" int rsx_comm_destroy_h(rsx_comm_t comm) {
  if (comm == nullptr) return RSX_EINVAL;
  const Rccl* r = rccl();
  Comm* c = static_cast<Comm*>(comm);
  int out = RSX_OK;
  if (r != nullptr && c->c != nullptr) {
    const ncclResult_t rc = r->CommDestroy(c->c);
    if (rc != ncclSuccess) out = fail(r, rc);
  }
  delete c;
  return out;
}

extern "C" int rsx_comm_rank_world_h(rsx_comm_t comm, int* rank, int* world) {
  if (comm == nullptr) return RSX_EINVAL;
  const Comm* c = static_cast<const Comm*>(comm);
  if (rank != nullptr) *rank = c->rank;
  if (world != nullptr) *world = c->world;
  return RSX_OK;
}

extern "C" int rsx_all_gather(rsx_comm_t comm, const void* send, void* recv, size_t bytes_per_rank, rsx_stream_t stream) {
  if (comm == nullptr || send == nullptr || recv == nullptr) return RSX_EINVAL;
  if (bytes_per_rank == 0) return RSX_OK;
  const Rccl* r = rccl();
  if (r == nullptr) return RSX_EUNSUPPORTED;
  const Comm* c = static_cast<const Comm*>(comm);
  // whole words where the block allows it (every block of the step is a multiple of 16 bytes): fewer, wider elements
  if ((bytes_per_rank & 3) == 0)
    RSX_NCCL(r->AllGather(send, recv, bytes_per_rank / 4, ncclInt32, c->c, reinterpret_cast<hipStream_t>(stream)));
  else
    RSX_NCCL(r->AllGather(send, recv, bytes_per_rank, ncclInt8, c->c, reinterpret_cast<hipStream_t>(stream)));
  return RSX_OK;
}

extern "C" int rsx_all_reduce_sum_f32(rsx_comm_t comm, const float* send, float* recv, size_t n, rsx_stream_t stream) {
  if (comm == nullptr || send == nullptr || recv == nullptr) return RSX_EINVAL;
  if (n == 0) return RSX_OK;
  const Rccl* r = rccl();
  if (r == nullptr) return RSX_EUNSUPPORTED;
  const Comm* c = static_cast<const Comm*>(comm);
  RSX_NCCL(r->AllReduce(send, recv, n, ncclFloat32, ncclSum, c->c, reinterpret_cast<hipStream_t>(stream)));
  return RSX_OK;
}

extern "C" int rsx_all_reduce_all_gather(rsx_comm_t comm, float* grad, size_t n, const void* send, void* recv,
                                         size_t bytes_per_rank, rsx_stream_t stream) {
  if (comm == nullptr || grad == nullptr || send == nullptr || recv == nullptr || (bytes_per_rank & 3)) return RSX_EINVAL;
  const Rccl* r = rccl();
  if (r == nullptr) return RSX_EUNSUPPORTED;
  const Comm* c = static_cast<const Comm*>(comm);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  RSX_NCCL(r->GroupStart());                     // one fused launch: the two collectives share their proxy / kernel start-up
  ncclResult_t a = ncclSuccess, b = ncclSuccess;
  if (n != 0) a = r->AllReduce(grad, grad, n, ncclFloat32, ncclSum, c->c, st);
  if (bytes_per_rank != 0) b = r->AllGather(send, recv, bytes_per_rank / 4, ncclInt32, c->c, st);
  const ncclResult_t e = r->GroupEnd();
  if (a != ncclSuccess) return fail(r, a);
  if (b != ncclSuccess) return fail(r, b);
  if (e != ncclSuccess) return fail(r, e);
  return RSX_OK;
}
