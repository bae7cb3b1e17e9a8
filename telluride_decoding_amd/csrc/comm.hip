// The one exchange step of the fit: an all-reduce(sum) of the packed
// sufficient statistics over an RCCL communicator, stream-ordered on the handle's stream.
//
// RCCL is bound at run time (dlopen): the library has no link-time dependency on it, a
// single-GPU user never loads it, and inside a PyTorch process the copy PyTorch has already
// mapped is the one that is used (two RCCL runtimes in one process do not share
// communicators).  The reference has no communication layer at all (process-level fan-out
// only, doc/DecodingCodelab.md:354-381); the packed layout is td_stats_pack's.
#include <dlfcn.h>
#include <link.h>

#include <mutex>

#include "td_common.h"

namespace {

// the slice of rccl.h this file needs (ABI-stable: NCCL 2.x)
typedef void* rcclComm;
struct RcclUniqueId { char internal[128]; };
constexpr int kRcclSum = 0;        // ncclSum
constexpr int kRcclFloat64 = 8;    // ncclFloat64

struct Rccl {
  void* lib = nullptr;
  std::string path, error;
  int (*get_unique_id)(RcclUniqueId*) = nullptr;
  int (*comm_init_rank)(rcclComm*, int, RcclUniqueId, int) = nullptr;
  int (*comm_destroy)(rcclComm) = nullptr;
  int (*comm_count)(rcclComm, int*) = nullptr;
  int (*all_reduce)(const void*, void*, size_t, int, int, rcclComm, hipStream_t) = nullptr;
  const char* (*error_string)(int) = nullptr;
};

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
    *static_cast<std::string*>(out) = info->dlpi_name;
    return 1;
  }
  return 0;
}

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::string> names;
    if (const char* env = getenv("TD_RCCL_LIB")) names.push_back(env);
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);      // the copy this process already runs
    if (!loaded.empty()) names.push_back(loaded);
    names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"});
    for (const auto& n : names) {
      r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (r.lib) { r.path = n; break; }
      r.error = dlerror();
    }
    if (!r.lib) return;
    auto sym = [&](const char* name) {
      void* p = dlsym(r.lib, name);
      if (!p) r.error = std::string("missing symbol ") + name + " in " + r.path;
      return p;
    };
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(sym("ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(sym("ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(sym("ncclCommDestroy"));
    r.comm_count = reinterpret_cast<decltype(r.comm_count)>(sym("ncclCommCount"));
    r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(sym("ncclAllReduce"));
    r.error_string = reinterpret_cast<decltype(r.error_string)>(sym("ncclGetErrorString"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.comm_count ||
        !r.all_reduce || !r.error_string) {
      dlclose(r.lib);
      r.lib = nullptr;
    }
  });
  return &r;
}

int need_rccl(td_handle* h, Rccl** out) {
  Rccl* r = rccl();
  if (!r->lib)
    return td_fail(h, TD_ERR_STATE, "RCCL is not available (%s); set TD_RCCL_LIB to librccl.so",
                   r->error.c_str());
  *out = r;
  return TD_OK;
}

#define TD_RCCL(h, r, expr)                                                          \
  do {                                                                               \
    const int _e = (expr);                                                           \
    if (_e != 0)                                                                     \
      return td_fail((h), TD_ERR_HIP, "%s failed: %s", #expr, (r)->error_string(_e)); \
  } while (0)

}  // namespace

extern "C" {

int td_rccl_available(td_handle* h) {
  Rccl* r = nullptr;
  return need_rccl(h, &r);       // dlopen + dlsym only: no RCCL call, no bootstrap thread
}

int td_rccl_unique_id(td_handle* h, void* id_out_128) {
  if (!id_out_128) return td_fail(h, TD_ERR_INVALID, "td_rccl_unique_id: NULL");
  Rccl* r = nullptr;
  TD_TRY(need_rccl(h, &r));
  RcclUniqueId id;
  TD_RCCL(h, r, r->get_unique_id(&id));
  memcpy(id_out_128, id.internal, sizeof(id.internal));
  return TD_OK;
}

int td_rccl_comm_create(td_handle* h, int num_ranks, int rank, const void* id_128, void** comm_out) {
  if (!h || !id_128 || !comm_out) return td_fail(h, TD_ERR_INVALID, "td_rccl_comm_create: NULL");
  TD_REQUIRE(h, num_ranks >= 1 && rank >= 0 && rank < num_ranks,
             "td_rccl_comm_create: rank %d of %d", rank, num_ranks);
  Rccl* r = nullptr;
  TD_TRY(need_rccl(h, &r));
  TD_HIP(h, hipSetDevice(h->device));
  RcclUniqueId id;
  memcpy(id.internal, id_128, sizeof(id.internal));
  rcclComm comm = nullptr;
  TD_RCCL(h, r, r->comm_init_rank(&comm, num_ranks, id, rank));
  *comm_out = comm;
  return TD_OK;
}

int td_rccl_comm_destroy(td_handle* h, void* comm) {
  if (!comm) return TD_OK;
  Rccl* r = nullptr;
  TD_TRY(need_rccl(h, &r));
  TD_RCCL(h, r, r->comm_destroy(static_cast<rcclComm>(comm)));
  return TD_OK;
}

int td_rccl_comm_count(td_handle* h, void* comm, int* num_ranks) {
  if (!comm || !num_ranks) return td_fail(h, TD_ERR_INVALID, "td_rccl_comm_count: NULL");
  Rccl* r = nullptr;
  TD_TRY(need_rccl(h, &r));
  TD_RCCL(h, r, r->comm_count(static_cast<rcclComm>(comm), num_ranks));
  return TD_OK;
}

int td_allreduce_f64(td_handle* h, double* buf_dev, int64_t count, void* rccl_comm) {
  if (!h || !buf_dev || !rccl_comm) return td_fail(h, TD_ERR_INVALID, "td_allreduce_f64: NULL");
  if (count <= 0) return TD_OK;
  Rccl* r = nullptr;
  TD_TRY(need_rccl(h, &r));
  TD_RCCL(h, r, r->all_reduce(buf_dev, buf_dev, (size_t)count, kRcclFloat64, kRcclSum,
                              static_cast<rcclComm>(rccl_comm), h->stream));
  return TD_OK;
}

int td_stats_allreduce(td_handle* h, td_stats* s, void* rccl_comm, int64_t total_file_slots,
                       int64_t file_slot, int64_t total_frames) {
  if (!h || !s || !rccl_comm) return td_fail(h, TD_ERR_INVALID, "td_stats_allreduce: NULL");
  int64_t len = 0;
  TD_TRY(td_stats_packed_len(h, s, total_file_slots, &len));
  // the packed buffer lives in the handle's solver arena: pack, collective, unpack and whatever
  // follows (the solve) are ordered on the handle's one stream
  void* buf = nullptr;
  TD_TRY(td_workspace(h, sizeof(double) * (size_t)len, &buf));
  double* packed = static_cast<double*>(buf);
  TD_TRY(td_stats_pack(h, s, packed, total_file_slots, file_slot));
  TD_TRY(td_allreduce_f64(h, packed, len, rccl_comm));
  return td_stats_unpack_known(h, s, packed, total_file_slots, total_frames);
}

}  // extern "C"
