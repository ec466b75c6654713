// Logging-only collective of the hot path: all-gather of the predicted maps
// [B_local, K, S, S] fp32 of every rank over RCCL / xGMI, as a C entry point a non-torch host can call.
//
// The reference has no collective on this path (one env per process, nav/collect.py:32-50, sharded by hand with
// --start_ep/--end_ep/--sem_gpu_id, nav/arguments.py:15-20); BASELINE.json's north_star adds this one to collate
// predictions for logging.  One process per GPU; the communicator is built from a 128-byte RCCL unique id that rank
// 0 creates and the host distributes however it likes (torch.distributed store, MPI, a file).
//
// RCCL is bound at run time, not at link time: a torch process already carries its own librccl.so and a second copy
// in the same process must be avoided, while a torch-free host gets ROCm's (/opt/rocm/lib/librccl.so.1).  The
// algorithm is RCCL's choice (on a fully connected 8-GPU xGMI mesh it selects a direct / tree schedule over the 7
// links for large messages); this wrapper adds no staging copy: `local` is sent from and `all`
// received into the caller's buffers, on the caller's stream.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "../../include/peanut_hip.h"
#include "common.h"

namespace peanut {
namespace {

struct Rccl {
  void* lib = nullptr;
  std::string origin;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_err;

void bind_rccl() {
  // already-loaded copies first (RTLD_NOLOAD), then a fresh load; PEANUT_RCCL_LIB overrides
  const char* env = getenv("PEANUT_RCCL_LIB");
  struct Try { const char* name; int flags; };
  const Try tries[] = {{env, RTLD_NOW | RTLD_GLOBAL},
                       {"librccl.so", RTLD_NOW | RTLD_NOLOAD},
                       {"librccl.so.1", RTLD_NOW | RTLD_NOLOAD},
                       {"librccl.so.1", RTLD_NOW | RTLD_GLOBAL},
                       {"librccl.so", RTLD_NOW | RTLD_GLOBAL},
                       {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL}};
  for (const Try& t : tries) {
    if (!t.name || !t.name[0]) continue;
    void* h = dlopen(t.name, t.flags);
    if (!h) continue;
    g_rccl.lib = h;
    g_rccl.origin = std::string(t.name) + ((t.flags & RTLD_NOLOAD) ? " (already loaded in this process)" : "");
    break;
  }
  if (!g_rccl.lib) {
    g_rccl_err = "RCCL not found (tried the loaded librccl.so, librccl.so.1, /opt/rocm/lib/librccl.so.1; set PEANUT_RCCL_LIB)";
    return;
  }
#define PEANUT_BIND(field, sym)                                                        \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, sym));     \
  if (!g_rccl.field) { g_rccl_err = std::string("RCCL symbol missing: ") + sym; return; }
  PEANUT_BIND(GetUniqueId, "ncclGetUniqueId");
  PEANUT_BIND(CommInitRank, "ncclCommInitRank");
  PEANUT_BIND(CommDestroy, "ncclCommDestroy");
  PEANUT_BIND(AllGather, "ncclAllGather");
  PEANUT_BIND(GetErrorString, "ncclGetErrorString");
#undef PEANUT_BIND
}

int need_rccl() {
  std::call_once(g_rccl_once, bind_rccl);
  if (!g_rccl_err.empty()) return fail(PEANUT_EHIP, g_rccl_err);
  return 0;
}

int nccl_fail(const char* what, ncclResult_t r) {
  return fail(PEANUT_EHIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}

}  // namespace
}  // namespace peanut

using namespace peanut;

struct peanut_comm {
  ncclComm_t comm = nullptr;
  int n_ranks = 1, rank = 0, device = 0;
};

extern "C" {

int peanut_comm_unique_id(unsigned char id[PEANUT_COMM_ID_BYTES]) {
  static_assert(PEANUT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id) return fail(PEANUT_EINVAL, "peanut_comm_unique_id: null argument");
  if (int rc = need_rccl()) return rc;
  ncclUniqueId u;
  ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int peanut_comm_create(peanut_comm_t** out, int n_ranks, int rank, const unsigned char id[PEANUT_COMM_ID_BYTES]) {
  if (!out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(PEANUT_EINVAL, "peanut_comm_create: bad arguments");
  auto c = new peanut_comm();
  c->n_ranks = n_ranks;
  c->rank = rank;
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; return fail(PEANUT_EHIP, "peanut_comm_create: no HIP device"); }
  if (n_ranks > 1) {
    if (!id) { delete c; return fail(PEANUT_EINVAL, "peanut_comm_create: null id"); }
    if (int rc = need_rccl()) { delete c; return rc; }
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, n_ranks, u, rank);     // collective: every rank must call it
    if (r != ncclSuccess) { delete c; return nccl_fail("ncclCommInitRank", r); }
  }
  *out = c;
  return 0;
}

void peanut_comm_destroy(peanut_comm_t* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}

int peanut_comm_info(peanut_comm_t* c, int* n_ranks, int* rank) {
  if (!c) return fail(PEANUT_EINVAL, "peanut_comm_info: null communicator");
  if (n_ranks) *n_ranks = c->n_ranks;
  if (rank) *rank = c->rank;
  return 0;
}

const char* peanut_comm_backend(void) {
  if (need_rccl()) return "";
  return g_rccl.origin.c_str();
}

int peanut_allgather_maps(peanut_comm_t* c, const float* local, float* all, size_t count, void* stream) {
  if (!c || !local || !all) return fail(PEANUT_EINVAL, "peanut_allgather_maps: null argument");
  if (c->n_ranks == 1) {   // one rank: the gathered tensor is the shard
    if (all != local) PEANUT_HIP_CHECK(hipMemcpyAsync(all, local, count * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
  }
  ncclResult_t r = g_rccl.AllGather(local, all, count, ncclFloat32, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return nccl_fail("ncclAllGather", r);
  return 0;
}

}  // extern "C"
