// Data-parallel gradient exchange behind the C ABI: ia_comm_* = a thin layer over RCCL's
// ring / direct all-reduce across the xGMI links of one node, for hosts that bind only include/itemalign.h.  (The Python host of this
// repo goes through torch.distributed -- the same RCCL -- from item_alignment_amd/dist.py; reference finetune_multimodal.py:371-468 is
// the single-GPU loop being sharded.)  RCCL is bound at run time with dlopen: a copy already mapped into the process (torch ships
// its own) is reused, so the library has no link-time dependency on it and never brings a second instance in.
#include "common.h"
#include "../../include/itemalign.h"
#include <dlfcn.h>
#include <link.h>
#include <cstring>
#include <mutex>
#include <string>

namespace {
// the few RCCL declarations used (rccl.h 2.x: ncclUniqueId = 128 opaque bytes, ncclSum = 0, ncclFloat32 = 7, ncclBfloat16 = 9)
struct UniqueId { char internal[IA_COMM_ID_BYTES]; };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*GetVersionFn)(int*);
struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr; CommInitRankFn comm_init_rank = nullptr; AllReduceFn all_reduce = nullptr;
  CommDestroyFn comm_destroy = nullptr; GetErrorStringFn get_error_string = nullptr; GetVersionFn get_version = nullptr;
  bool usable = false;
  char why[256] = "";             // written once under g_once, read-only afterwards
};
Rccl g_rccl;
std::once_flag g_once;
// the text behind ia_comm_last_error(): per thread, so two host threads driving two communicators never write one buffer
thread_local char g_comm_error[256] = "";

// A copy of RCCL the process has already mapped under ANY file name (torch bundles its own librccl next to libtorch_hip.so, which a
// soname lookup with RTLD_NOLOAD can miss): walk the loaded objects and re-open that very file, so no second instance comes in.
int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
  if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl")) { *static_cast<std::string*>(out) = info->dlpi_name; return 1; }
  return 0;
}

bool bind_rccl() {
  std::call_once(g_once, [] {
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);
    if (!loaded.empty()) g_rccl.handle = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names) if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.handle) { snprintf(g_rccl.why, sizeof g_rccl.why, "librccl not found: %s", dlerror()); return; }
    const char* missing = nullptr;
    auto sym = [&](const char* name) { void* p = dlsym(g_rccl.handle, name); if (!p && !missing) missing = name; return p; };
    g_rccl.get_unique_id = (GetUniqueIdFn)sym("ncclGetUniqueId");
    g_rccl.comm_init_rank = (CommInitRankFn)sym("ncclCommInitRank");
    g_rccl.all_reduce = (AllReduceFn)sym("ncclAllReduce");
    g_rccl.comm_destroy = (CommDestroyFn)sym("ncclCommDestroy");
    g_rccl.get_version = (GetVersionFn)sym("ncclGetVersion");
    g_rccl.get_error_string = (GetErrorStringFn)dlsym(g_rccl.handle, "ncclGetErrorString");      // optional
    if (missing) { snprintf(g_rccl.why, sizeof g_rccl.why, "librccl lacks %s", missing); return; }
    // the enum values used below (ncclSum 0, ncclFloat32 7, ncclBfloat16 9) and the 128-byte id are those of the 2.x API
    int version = 0;
    if (g_rccl.get_version(&version) != 0 || version / 10000 != 2) {
      snprintf(g_rccl.why, sizeof g_rccl.why, "unsupported RCCL version code %d (the 2.x API is bound by value)", version);
      return;
    }
    g_rccl.usable = true;
  });
  if (!g_rccl.usable) snprintf(g_comm_error, sizeof g_comm_error, "%s", g_rccl.why);
  return g_rccl.usable;
}
int fail(int rc) {
  if (rc != 0) snprintf(g_comm_error, sizeof g_comm_error, "RCCL: %s", g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "error");
  return rc == 0 ? IA_OK : IA_ERR_LAUNCH;
}
}  // namespace

extern "C" const char* ia_comm_last_error(void) { return g_comm_error; }

extern "C" int ia_comm_unique_id(void* id_out) {
  if (!id_out) return IA_ERR_ARG;
  if (!bind_rccl()) return IA_ERR_UNSUPPORTED;
  UniqueId id;
  const int rc = g_rccl.get_unique_id(&id);
  if (rc == 0) std::memcpy(id_out, &id, sizeof id);
  return fail(rc);
}

extern "C" int ia_comm_init(const void* id, int rank, int world_size, void** comm_out) {
  if (!id || !comm_out || world_size < 1 || rank < 0 || rank >= world_size) return IA_ERR_ARG;
  if (!bind_rccl()) return IA_ERR_UNSUPPORTED;
  UniqueId u;
  std::memcpy(&u, id, sizeof u);
  *comm_out = nullptr;
  return fail(g_rccl.comm_init_rank(comm_out, world_size, u, rank));
}

extern "C" int ia_comm_allreduce_bucket(void* comm, void* buf, size_t count, int dtype, ia_stream_t stream) {
  if (!comm || !buf || count == 0 || (dtype != IA_COMM_F32 && dtype != IA_COMM_BF16)) return IA_ERR_ARG;
  if (!bind_rccl()) return IA_ERR_UNSUPPORTED;
  return fail(g_rccl.all_reduce(buf, buf, count, dtype == IA_COMM_F32 ? 7 : 9, 0 /* sum */, comm, (hipStream_t)stream));
}

extern "C" int ia_comm_finalize(void* comm) {
  if (!comm) return IA_ERR_ARG;
  if (!bind_rccl()) return IA_ERR_UNSUPPORTED;
  return fail(g_rccl.comm_destroy(comm));
}
