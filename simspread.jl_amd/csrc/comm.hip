// The final score gather inside the library (north star: "RCCL over xGMI only for the final score gather"; SURVEY.md
// 8(b), 8(e)).  One process per GPU: rank 0 creates a unique id (ss_comm_unique_id), the host framework hands its 128
// bytes to the other processes (MPI / Distributed.jl / torch.distributed -- any channel), every process calls
// ss_comm_init(id, rank, nranks), and ss_gather_rows_* then moves the finished row blocks of the score matrix:
// a DIRECT exchange of exact row counts -- one ncclRecv per peer straight into its slice of the result, one ncclSend
// of the own block per peer, all inside one ncclGroupStart/End on the library's stream -- so that every one of the
// 7 point-to-point xGMI links of a GPU carries traffic at once (a ring all-gather is bound by one link).
// RCCL is loaded with dlopen when the communicator is created: the library has no link-time dependency on it and a
// single-GPU process never touches it.  The reference has no counterpart (no NCCL/MPI anywhere).
#include <dlfcn.h>

#include <mutex>

#include "common.hpp"

namespace ss {

// the slice of rccl.h this file needs (ABI of librccl.so.1, ROCm 7.x)
typedef struct { char internal[128]; } NcclUniqueId;
typedef void* NcclComm;
enum { NCCL_SUCCESS = 0 };
enum { NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8 };  // ncclDataType_t: ncclFloat = 7, ncclDouble = 8

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  NcclComm comm = nullptr;
  int rank = 0, nranks = 0;
  std::mutex mu;
};

static Rccl& rccl() {
  static Rccl r;
  return r;
}

static int rccl_load() {
  Rccl& r = rccl();
  if (r.handle) return SS_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.handle) break;
  }
  if (!r.handle) return fail(SS_EUNSUPPORTED, "librccl.so not found (%s): the in-library gather needs RCCL", dlerror());
#define SS_SYM(field, name)                                                             \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));                 \
  if (!r.field) return fail(SS_EUNSUPPORTED, "librccl.so has no symbol %s", name)
  SS_SYM(GetUniqueId, "ncclGetUniqueId");
  SS_SYM(CommInitRank, "ncclCommInitRank");
  SS_SYM(CommDestroy, "ncclCommDestroy");
  SS_SYM(GroupStart, "ncclGroupStart");
  SS_SYM(GroupEnd, "ncclGroupEnd");
  SS_SYM(Send, "ncclSend");
  SS_SYM(Recv, "ncclRecv");
  SS_SYM(GetErrorString, "ncclGetErrorString");
#undef SS_SYM
  return SS_OK;
}

#define SS_NCCL(call)                                                                              \
  do {                                                                                             \
    int _e = (call);                                                                               \
    if (_e != NCCL_SUCCESS)                                                                        \
      return fail(SS_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, rccl().GetErrorString(_e)); \
  } while (0)

int comm_unique_id(char* id128) {
  SS_TRY(require_init());
  if (!id128) return fail(SS_EINVAL, "id buffer is NULL");
  std::lock_guard<std::mutex> lk(rccl().mu);
  SS_TRY(rccl_load());
  NcclUniqueId id;
  SS_NCCL(rccl().GetUniqueId(&id));
  memcpy(id128, id.internal, 128);
  return SS_OK;
}

int comm_init(const char* id128, int rank, int nranks) {
  SS_TRY(require_init());
  if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(SS_EINVAL, "bad communicator arguments");
  std::lock_guard<std::mutex> lk(rccl().mu);
  SS_TRY(rccl_load());
  Rccl& r = rccl();
  if (r.comm) {
    SS_NCCL(r.CommDestroy(r.comm));
    r.comm = nullptr;
  }
  NcclUniqueId id;
  memcpy(id.internal, id128, 128);
  SS_NCCL(r.CommInitRank(&r.comm, nranks, id, rank));
  r.rank = rank;
  r.nranks = nranks;
  return SS_OK;
}

int comm_destroy() {
  std::lock_guard<std::mutex> lk(rccl().mu);
  Rccl& r = rccl();
  if (r.comm) {
    if (ctx().inited) (void)hipStreamSynchronize(ctx().stream);
    SS_NCCL(r.CommDestroy(r.comm));
    r.comm = nullptr;
  }
  r.nranks = 0;
  return SS_OK;
}

int comm_info(int* rank, int* nranks) {
  std::lock_guard<std::mutex> lk(rccl().mu);
  if (rank) *rank = rccl().rank;
  if (nranks) *nranks = rccl().comm ? rccl().nranks : 0;
  return SS_OK;
}

// elem: 4 (fp32) or 8 (fp64).  local: nrows_local x ncols row-major (device); full: sum(counts) x ncols row-major
// (device), needed on the receiving ranks only; root < 0: every rank receives.
int gather_rows(const void* local, int64_t ncols, const int64_t* counts, void* full, int root, int elem) {
  SS_TRY(require_init());
  std::lock_guard<std::mutex> lk(rccl().mu);
  Rccl& r = rccl();
  if (!r.comm) return fail(SS_EINVAL, "ss_comm_init has not been called");
  if (!counts || ncols < 0) return fail(SS_EINVAL, "bad gather arguments");
  if (root >= r.nranks) return fail(SS_EINVAL, "root outside the communicator");
  const bool receives = root < 0 || root == r.rank;
  if (receives && !full) return fail(SS_EINVAL, "the receiving rank needs the result buffer");
  int64_t total = 0;
  for (int p = 0; p < r.nranks; ++p) {
    if (counts[p] < 0) return fail(SS_EINVAL, "negative row count");
    total += counts[p];
  }
  if (counts[r.rank] > 0 && !local) return fail(SS_EINVAL, "local block is NULL");
  const int dt = elem == 4 ? NCCL_FLOAT32 : NCCL_FLOAT64;
  hipStream_t st = ctx().stream;
  int64_t start = 0, mine = 0;
  for (int p = 0; p < r.rank; ++p) mine += counts[p];
  if (receives && counts[r.rank] > 0)
    SS_HIP(hipMemcpyAsync(static_cast<char*>(full) + mine * ncols * elem, local, (size_t)counts[r.rank] * ncols * elem,
                          hipMemcpyDeviceToDevice, st));
  SS_NCCL(r.GroupStart());
  // inside the group no early return: a failed Send/Recv must still reach GroupEnd, or the group stays open and every
  // later RCCL call of this process nests in it and never launches
  int bad = NCCL_SUCCESS;
  const char* what = "";
  for (int p = 0; p < r.nranks && bad == NCCL_SUCCESS; ++p) {
    if (p != r.rank) {
      if (receives && counts[p] > 0) {
        bad = r.Recv(static_cast<char*>(full) + start * ncols * elem, (size_t)counts[p] * ncols, dt, p, r.comm, st);
        what = "ncclRecv";
      }
      if (bad == NCCL_SUCCESS && counts[r.rank] > 0 && (root < 0 || root == p)) {
        bad = r.Send(local, (size_t)counts[r.rank] * ncols, dt, p, r.comm, st);
        what = "ncclSend";
      }
    }
    start += counts[p];
  }
  const int end = r.GroupEnd();
  if (bad != NCCL_SUCCESS) return fail(SS_EHIP, "%s inside the gather group -> %s", what, r.GetErrorString(bad));
  if (end != NCCL_SUCCESS) return fail(SS_EHIP, "ncclGroupEnd -> %s", r.GetErrorString(end));
  (void)total;
  return SS_OK;
}

}  // namespace ss
