// rccl_api.hpp -- RCCL opened at run time (dlopen): libcilantro_hip.so itself does not depend on it.  The entry points the
// multi-device loop (multi.hip) and the rank communicator (c_api.hip: cilhip_rank_comm_*) call, with rccl.h's enum values.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

namespace cilhip {
typedef void* rccl_comm_t;
struct RcclApi {
  void* lib = nullptr;
  int (*CommInitAll)(rccl_comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  struct UniqueId { char internal[128]; };      // ncclUniqueId of rccl.h (NCCL_UNIQUE_ID_BYTES = 128), passed by value
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(rccl_comm_t*, int, UniqueId, int) = nullptr;
  bool load() {
    if (lib) return true;
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return false;
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    return CommInitAll && CommDestroy && AllReduce && GroupStart && GroupEnd && GetUniqueId && CommInitRank;
  }
};
constexpr int RCCL_DOUBLE = 8, RCCL_SUM = 0;      // ncclDouble / ncclSum of rccl.h (ncclDataType_t / ncclRedOp_t)
constexpr int RCCL_UINT64 = 5, RCCL_MIN = 3;      // ncclUint64 / ncclMin of rccl.h
}  // namespace cilhip
