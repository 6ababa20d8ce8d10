// rf_comm_*: the gradient exchange of the data-parallel train step as thin RCCL calls on an EXPLICIT communication
// stream gated by HIP events (SURVEY 8(b) "Comm: rf_comm_{init,allreduce_bucket,wait} over RCCL with an explicit comm
// stream"; replaces what Lightning's DDPStrategy(process_group_backend="nccl") does behind
// experiments/full_comparison.py:794 -- bucketed gradient all-reduce overlapped with backward).
//
//   producer stream (backward) --record--> [bucket_ready event] --wait--> comm stream: ncclAllReduce(bucket, in place)
//   comm stream --record--> [done event] --wait--> consumer stream (clip + AdamW)          (rf_comm_wait)
//
// Nothing here blocks the host.  RCCL is resolved at RUN time (dlsym on the process first -- a PyTorch process already
// carries an RCCL --, then librccl.so): librf_hip.so has no link-time dependency on it and loads on a box without RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace {

struct Api {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclBroadcast) broadcast = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  bool ok = false;
};

Api& api() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllReduce")) {
      h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) return;
    }
    a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    a.all_reduce = reinterpret_cast<decltype(a.all_reduce)>(dlsym(h, "ncclAllReduce"));
    a.broadcast = reinterpret_cast<decltype(a.broadcast)>(dlsym(h, "ncclBroadcast"));
    a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(h, "ncclGetErrorString"));
    a.ok = a.get_unique_id && a.comm_init_rank && a.comm_destroy && a.all_reduce && a.broadcast && a.error_string;
  });
  return a;
}

// ONE `ready` / `done` event pair per communicator is enough only because every collective of a communicator goes out
// on its ONE communication stream: collectives are totally ordered there, so `done` -- re-recorded after each of them --
// always marks the last one, and waiting on it covers all earlier ones; `ready` is consumed (hipStreamWaitEvent) before the
// next record overwrites it.  A second communication stream would need its own pair.
struct Comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;   // the communication stream
  hipEvent_t ready = nullptr;     // producer -> comm stream
  hipEvent_t done = nullptr;      // comm stream -> consumer
  int rank = 0, world = 1, device = 0;
  long launched = 0;
};

#define RF_HIP_TRY(expr)                         \
  do {                                           \
    hipError_t e__ = (expr);                     \
    if (e__ != hipSuccess) {                     \
      rf_g_last_error = hipGetErrorString(e__);  \
      return RF_ELAUNCH;                         \
    }                                            \
  } while (0)
#define RF_NCCL_TRY(expr)                             \
  do {                                                \
    ncclResult_t r__ = (expr);                        \
    if (r__ != ncclSuccess) {                         \
      rf_g_last_error = api().error_string(r__);      \
      return RF_ELAUNCH;                              \
    }                                                 \
  } while (0)

}  // namespace

extern "C" int rf_comm_available() { return api().ok ? 1 : 0; }

extern "C" int rf_comm_unique_id(void* id_out_128_bytes) {
  RF_REQUIRE(id_out_128_bytes);
  if (!api().ok) {
    rf_g_last_error = "rf_comm: RCCL (librccl.so) could not be resolved";
    return RF_EUNSUPPORTED;
  }
  static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id size");
  RF_NCCL_TRY(api().get_unique_id(static_cast<ncclUniqueId*>(id_out_128_bytes)));
  return RF_OK;
}

extern "C" int rf_comm_destroy(void* comm);

namespace {
// rf_comm_init's body: every failure path leaves through the caller, which frees whatever was already created
int comm_init_body(Comm* c, const void* id_128_bytes, int rank, int world) {
  RF_HIP_TRY(hipGetDevice(&c->device));
  ncclUniqueId id;
  std::memcpy(&id, id_128_bytes, sizeof(id));
  RF_NCCL_TRY(api().comm_init_rank(&c->comm, world, id, rank));  // collective: every rank calls it with the same id
  int lo = 0, hi = 0;
  RF_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
  RF_HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));  // highest priority: never queued behind compute
  RF_HIP_TRY(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  RF_HIP_TRY(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  return RF_OK;
}
}  // namespace

extern "C" int rf_comm_init(void** comm_out, const void* id_128_bytes, int rank, int world) {
  RF_REQUIRE(comm_out && id_128_bytes && world >= 1 && rank >= 0 && rank < world);
  if (!api().ok) {
    rf_g_last_error = "rf_comm: RCCL (librccl.so) could not be resolved";
    return RF_EUNSUPPORTED;
  }
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  int rc = comm_init_body(c, id_128_bytes, rank, world);
  if (rc != RF_OK) {  //  nothing of a failed init survives: communicator, stream, events, the struct
    const char* why = rf_g_last_error;
    (void)rf_comm_destroy(c);
    rf_g_last_error = why;
    *comm_out = nullptr;
    return rc;
  }
  *comm_out = c;
  return RF_OK;
}

// In-place all-reduce of `count` elements at `buf` (dtype 0 = fp32, 1 = bf16; SUM, or the mean over ranks when
// `average`) on the communication stream, ordered after everything enqueued so far on `producer_stream`.
extern "C" int rf_comm_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, int average, void* producer_stream) {
  RF_REQUIRE(comm && buf && count > 0 && (dtype == 0 || dtype == 1));
  Comm* c = static_cast<Comm*>(comm);
  RF_HIP_TRY(hipEventRecord(c->ready, static_cast<hipStream_t>(producer_stream)));
  RF_HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
  RF_NCCL_TRY(api().all_reduce(buf, buf, static_cast<size_t>(count), dtype == 0 ? ncclFloat32 : ncclBfloat16,
                               average ? ncclAvg : ncclSum, c->comm, c->stream));
  RF_HIP_TRY(hipEventRecord(c->done, c->stream));
  c->launched++;
  return RF_OK;
}

// One-time parameter broadcast from `root` (DDP construction, full_comparison.py:794,838), same stream discipline.
extern "C" int rf_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* producer_stream) {
  RF_REQUIRE(comm && buf && count > 0 && (dtype == 0 || dtype == 1));
  Comm* c = static_cast<Comm*>(comm);
  RF_REQUIRE(root >= 0 && root < c->world);
  RF_HIP_TRY(hipEventRecord(c->ready, static_cast<hipStream_t>(producer_stream)));
  RF_HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
  RF_NCCL_TRY(api().broadcast(buf, buf, static_cast<size_t>(count), dtype == 0 ? ncclFloat32 : ncclBfloat16, root, c->comm,
                              c->stream));
  RF_HIP_TRY(hipEventRecord(c->done, c->stream));
  c->launched++;
  return RF_OK;
}

// `consumer_stream` waits (on the device) for every collective launched so far; the host does not block.
extern "C" int rf_comm_wait(void* comm, void* consumer_stream) {
  RF_REQUIRE(comm);
  Comm* c = static_cast<Comm*>(comm);
  if (c->launched == 0) return RF_OK;
  RF_HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), c->done, 0));
  return RF_OK;
}

extern "C" int rf_comm_destroy(void* comm) {
  if (!comm) return RF_OK;
  Comm* c = static_cast<Comm*>(comm);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)api().comm_destroy(c->comm);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return RF_OK;
}
