// Gradient exchange of the data-parallel step over RCCL (xGMI), behind the C ABI: the replacement for the
// torch.nn.DataParallel the reference imports but never applies (attn_unet_data_parallel.py:32,1554).
//
// RCCL is bound at run time (dlopen): a process that already has librccl loaded -- PyTorch-ROCm's own copy, which
// torch.distributed's "nccl" backend uses -- shares that one instance instead of loading a second RCCL.  Every
// collective is enqueued on the CALLER's stream (a side HIP stream, so that it overlaps the rest of backward) and is
// capturable into a hipGraph like any other kernel launch.  No function here synchronises the host.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <mutex>
#include <string>

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  std::string why;        // dlerror() text of the failed load, read ONCE (dlerror() clears itself)
};

void rccl_load(Rccl& r);
Rccl g_rccl;
std::once_flag g_rccl_once;

// one load per process, safe against the RCCL watchdog / capture threads calling in concurrently
Rccl* rccl() {
  std::call_once(g_rccl_once, [] { rccl_load(g_rccl); });
  return g_rccl.ok ? &g_rccl : nullptr;
}
void rccl_load(Rccl& r) {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !r.h; ++pass)          // pass 0: an instance the process already holds (PyTorch's)
    for (const char* n : names) {
      r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (r.h) break;
    }
  if (!r.h) r.h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!r.h) { const char* e = dlerror(); r.why = e ? e : "dlopen failed"; return; }
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, name))
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce"); SYM(ReduceScatter, "ncclReduceScatter"); SYM(AllGather, "ncclAllGather");
  SYM(Broadcast, "ncclBroadcast"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter && r.AllGather && r.Broadcast;
  if (!r.ok) r.why = "a required ncclXxx symbol is missing";
}

int fail(Rccl* r, ncclResult_t e, const char* what) {
  coma_set_error("%s: %s", what, (r && r->GetErrorString) ? r->GetErrorString(e) : "RCCL error");
  return 3;
}
}  // namespace

#define RCCL_OR_FAIL()                                                              \
  Rccl* r = rccl();                                                                 \
  if (!r) { coma_set_error("librccl could not be loaded (dlopen): %s", g_rccl.why.c_str()); return 4; }

extern "C" int coma_comm_unique_id(void* id_out) {
  RCCL_OR_FAIL();
  COMA_CHECK(id_out, "comm_unique_id: null output");
  ncclUniqueId id;
  ncclResult_t e = r->GetUniqueId(&id);
  if (e != ncclSuccess) return fail(r, e, "ncclGetUniqueId");
  memcpy(id_out, &id, COMA_COMM_ID_BYTES);
  return 0;
}

extern "C" int coma_comm_init(const void* id, int32_t rank, int32_t nranks, void** comm_out) {
  RCCL_OR_FAIL();
  COMA_CHECK(id && comm_out && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad argument");
  static_assert(sizeof(ncclUniqueId) == COMA_COMM_ID_BYTES, "unique id size");
  ncclUniqueId uid;
  memcpy(&uid, id, COMA_COMM_ID_BYTES);
  ncclComm_t c = nullptr;
  ncclResult_t e = r->CommInitRank(&c, nranks, uid, rank);     // binds the communicator to the calling thread's current device
  if (e != ncclSuccess) return fail(r, e, "ncclCommInitRank");
  *comm_out = c;
  return 0;
}

extern "C" int coma_comm_destroy(void* comm) {
  RCCL_OR_FAIL();
  if (!comm) return 0;
  ncclResult_t e = r->CommDestroy((ncclComm_t)comm);
  return e == ncclSuccess ? 0 : fail(r, e, "ncclCommDestroy");
}

extern "C" int coma_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream) {
  RCCL_OR_FAIL();
  COMA_CHECK(comm && buf && n >= 0, "allreduce: bad argument");
  if (n == 0) return 0;
  ncclResult_t e = r->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
  return e == ncclSuccess ? 0 : fail(r, e, "ncclAllReduce");
}

extern "C" int coma_reduce_scatter_sum_f32(void* comm, const float* send, float* recv, int64_t n_per_rank, void* stream) {
  RCCL_OR_FAIL();
  COMA_CHECK(comm && send && recv && n_per_rank >= 0, "reduce_scatter: bad argument");
  if (n_per_rank == 0) return 0;
  ncclResult_t e = r->ReduceScatter(send, recv, (size_t)n_per_rank, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
  return e == ncclSuccess ? 0 : fail(r, e, "ncclReduceScatter");
}

extern "C" int coma_allgather_f32(void* comm, const float* send, float* recv, int64_t n_per_rank, void* stream) {
  RCCL_OR_FAIL();
  COMA_CHECK(comm && send && recv && n_per_rank >= 0, "allgather: bad argument");
  if (n_per_rank == 0) return 0;
  ncclResult_t e = r->AllGather(send, recv, (size_t)n_per_rank, ncclFloat32, (ncclComm_t)comm, (hipStream_t)stream);
  return e == ncclSuccess ? 0 : fail(r, e, "ncclAllGather");
}

extern "C" int coma_broadcast_f32(void* comm, float* buf, int64_t n, int32_t root, void* stream) {
  RCCL_OR_FAIL();
  COMA_CHECK(comm && buf && n >= 0, "broadcast: bad argument");
  if (n == 0) return 0;
  ncclResult_t e = r->Broadcast(buf, buf, (size_t)n, ncclFloat32, root, (ncclComm_t)comm, (hipStream_t)stream);
  return e == ncclSuccess ? 0 : fail(r, e, "ncclBroadcast");
}

/* ---- external events: a record node inside a captured step graph that streams OUTSIDE the graph can wait for ------------ */
extern "C" int coma_event_create(void** event_out) {
  COMA_CHECK(event_out, "event_create: bad argument");
  hipEvent_t e = nullptr;
  hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  if (rc != hipSuccess) { coma_set_error("hipEventCreateWithFlags: %s", hipGetErrorString(rc)); return 2; }
  *event_out = (void*)e;
  return 0;
}

extern "C" int coma_event_destroy(void* event) {
  if (!event) return 0;
  hipError_t rc = hipEventDestroy((hipEvent_t)event);
  if (rc != hipSuccess) { coma_set_error("hipEventDestroy: %s", hipGetErrorString(rc)); return 2; }
  return 0;
}

extern "C" int coma_event_record_external(void* event, void* stream) {
  COMA_CHECK(event, "event_record_external: bad argument");
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  hipError_t rc = hipStreamGetCaptureInfo_v2(s, &st, &id, &graph, &deps, &ndeps);
  if (rc != hipSuccess) { (void)hipGetLastError(); coma_set_error("hipStreamGetCaptureInfo_v2: %s", hipGetErrorString(rc)); return 2; }
  if (st != hipStreamCaptureStatusActive) {
    rc = hipEventRecord((hipEvent_t)event, s);
    if (rc != hipSuccess) { (void)hipGetLastError(); coma_set_error("hipEventRecord: %s", hipGetErrorString(rc)); return 2; }
    return 0;
  }
  // the record becomes a NODE of the graph being captured, behind everything the stream has captured so far, and the
  // stream's later work depends on that node (which orders nothing: the node only stamps the event)
  hipGraphNode_t node = nullptr;
  rc = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, (hipEvent_t)event);
  if (rc != hipSuccess) { (void)hipGetLastError(); coma_set_error("hipGraphAddEventRecordNode: %s", hipGetErrorString(rc)); return 2; }
  rc = hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
  if (rc != hipSuccess) { (void)hipGetLastError(); coma_set_error("hipStreamUpdateCaptureDependencies: %s", hipGetErrorString(rc)); return 2; }
  return 0;
}

extern "C" int coma_stream_wait_external(void* stream, void* event) {
  COMA_CHECK(event, "stream_wait_external: bad argument");
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing((hipStream_t)stream, &st);
  COMA_CHECK(st != hipStreamCaptureStatusActive, "stream_wait_external: the waiting stream must be outside the capture");
  hipError_t rc = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
  if (rc != hipSuccess) { (void)hipGetLastError(); coma_set_error("hipStreamWaitEvent(external): %s", hipGetErrorString(rc)); return 2; }
  return 0;
}
