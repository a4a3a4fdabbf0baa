// K8: data-parallel gradient all-reduce over RCCL (xGMI).  librccl.so is opened lazily so that libmgr.so
// loads (and single-GPU paths work) on hosts without it.  One process per GPU; the unique id is created by
// rank 0 and distributed by the host (any out-of-band channel).
#include <dlfcn.h>
#include <cstdlib>

#include "common.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;
enum { ncclFloat = 7 };
enum { ncclSum = 0, ncclMax = 2 };

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.lib) return 0;
  // The ROCm installation's RCCL by ABSOLUTE path first: a process that has imported PyTorch already holds the wheel's own
  // librccl.so (built against the wheel's bundled HIP / HSA runtimes), and a bare dlopen("librccl.so") returns THAT one -
  // whose HSA wrapper is not the initialised runtime ("pfn_hsa_system_get_info failed with 4107 ... no ROCm-capable
  // device is detected", seen under torch.distributed.run).  RTLD_LOCAL: its symbols are only reached through dlsym here.
  const char* names[] = {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return mgr_fail(-3, "cannot dlopen librccl.so: %s", dlerror());
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
  g_rccl.CommUserRank = reinterpret_cast<decltype(g_rccl.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce)
    return mgr_fail(-3, "librccl.so lacks required symbols");
  g_rccl.lib = h;
  return 0;
}

const char* rccl_err(ncclResult_t r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"; }

}  // namespace

struct mgr_comm {
  mgr_ctx* ctx;
  ncclComm_t comm;
  int nranks, rank;
};

extern "C" {

int mgr_comm_unique_id(uint8_t id[MGR_UNIQUE_ID_BYTES]) {
  MGR_REQUIRE(id, "null argument");
  int r = rccl_load();
  if (r) return r;
  static_assert(sizeof(ncclUniqueId) == MGR_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId u;
  ncclResult_t e = g_rccl.GetUniqueId(&u);
  if (e) return mgr_fail(-3, "ncclGetUniqueId: %s", rccl_err(e));
  memcpy(id, &u, sizeof(u));
  return 0;
}

int mgr_comm_init_rank(mgr_ctx* c, int nranks, int rank, const uint8_t id[MGR_UNIQUE_ID_BYTES], mgr_comm** out) {
  MGR_REQUIRE(c && id && out, "null argument");
  MGR_REQUIRE(nranks > 0 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
  int r = rccl_load();
  if (r) return r;
  MGR_HIP(hipSetDevice(c->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t comm;
  ncclResult_t e = g_rccl.CommInitRank(&comm, nranks, u, rank);
  if (e) return mgr_fail(-3, "ncclCommInitRank: %s", rccl_err(e));
  mgr_comm* m = new mgr_comm{c, comm, nranks, rank};
  *out = m;
  return 0;
}

static int allreduce(mgr_comm* m, float* dbuf, size_t n, int op) {
  MGR_REQUIRE(m && dbuf, "null argument");
  if (n == 0) return 0;
  // (family MGR_K_ALLREDUCE times the SUM reductions - the gradient all-reduce of a step; the max reductions are bench.py's barriers)
  if (op == ncclSum) mgr_prof_begin(m->ctx, MGR_K_ALLREDUCE);
  ncclResult_t e = g_rccl.AllReduce(dbuf, dbuf, n, ncclFloat, op, m->comm, mgr_stream(m->ctx));
  if (op == ncclSum) mgr_prof_end(m->ctx, MGR_K_ALLREDUCE);
  if (e) return mgr_fail(-3, "ncclAllReduce: %s", rccl_err(e));
  return 0;
}

int mgr_allreduce_sum(mgr_comm* m, float* dbuf, size_t n) { return allreduce(m, dbuf, n, ncclSum); }
int mgr_allreduce_max(mgr_comm* m, float* dbuf, size_t n) { return allreduce(m, dbuf, n, ncclMax); }

int mgr_comm_count(mgr_comm* m, int* nranks_seen, int* rank_seen) {
  MGR_REQUIRE(m && nranks_seen && rank_seen, "null argument");
  MGR_REQUIRE(g_rccl.CommCount && g_rccl.CommUserRank, "librccl.so lacks ncclCommCount / ncclCommUserRank");
  ncclResult_t e = g_rccl.CommCount(m->comm, nranks_seen);
  if (e) return mgr_fail(-3, "ncclCommCount: %s", rccl_err(e));
  e = g_rccl.CommUserRank(m->comm, rank_seen);
  if (e) return mgr_fail(-3, "ncclCommUserRank: %s", rccl_err(e));
  return 0;
}

int mgr_comm_destroy(mgr_comm* m) {
  if (!m) return 0;
  if (g_rccl.CommDestroy) g_rccl.CommDestroy(m->comm);
  delete m;
  return 0;
}

}  // extern "C"
