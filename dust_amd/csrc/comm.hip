// comm.hip -- the multi-GPU path behind the C ABI: one process per GPU, RCCL over xGMI.
//
// north_star's partition: the pixels of a frame shard across the GPUs of a node as row bands (DustHipFrameParams::row_begin/row_end),
// and the finished bands are gathered onto one GPU. The reference has no multi-device path; what this file stands for on the
// reference's side is "the plugin selects devices" (crates/render/src/lib.rs:58-134). Three things live here:
//   * DustHipComm: an RCCL communicator bound to a context (dust_hip_comm_create), or a LOOPBACK group of `world` ranks on one
//     device (dust_hip_comm_create_local) whose collectives are device copies and small reduction kernels -- what a one-GPU box,
//     the C++ host mirror and the tests drive the same entry points with;
//   * dust_hip_gather_bands: grouped ncclSend / ncclRecv of every rank's rows of a G-buffer plane to the root, on the
//     communicator's OWN stream behind what the context's stream has enqueued so far -- so the band frame k+1 renders (into
//     another bound target) while frame k's rows travel;
//   * dust_hip_gi_exchange_run: steps 2-5 of the multi-GPU GI protocol of include/dust_hip.h (all-reduce MAX of the slot owners,
//     all-gather of the hash stamps, export, all-reduce SUM of the winning surfels, import) on the context's stream.
// librccl is opened on first use (dlopen): a single-GPU host never loads it, and the library has no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>) && !defined(DUST_NO_RCCL_HEADERS)
#include <rccl/rccl.h>
#define DUST_HAVE_RCCL 1
#else   // a build host without the RCCL headers: loopback groups only, RCCL communicators report DUST_ERR_UNSUPPORTED
#define DUST_HAVE_RCCL 0
#define DUST_HIP_NCCL_ID_BYTES 128
typedef struct ncclComm* ncclComm_t;
#endif

#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "capi_internal.hpp"

namespace {

using dust_internal::set_error;

#if DUST_HAVE_RCCL
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  std::string error;
};
Rccl* rccl() {  // opened once; null (with the reason in last_error) when the node has no RCCL
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) { r.error = std::string("librccl not found: ") + dlerror(); return; }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p) { ok = false; r.error = std::string("librccl lacks ") + n; } return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    if (!ok) { dlclose(r.lib); r.lib = nullptr; }
  });
  return r.lib ? &r : nullptr;
}
#endif
DustStatus no_rccl() { return set_error(DUST_ERR_UNSUPPORTED, "RCCL is not available on this node (librccl.so could not be opened)"); }
#if DUST_HAVE_RCCL
DustStatus nccl_fail(ncclResult_t e, const char* what) {
  Rccl* r = rccl();
  return set_error(DUST_ERR_HIP, std::string(what) + ": " + (r ? r->GetErrorString(e) : "rccl error"));
}
#endif
DustStatus hip_fail(hipError_t e, const char* what) { return set_error(DUST_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, #expr); } while (0)
#define NCCL_TRY(expr) do { ncclResult_t e_ = (expr); if (e_ != ncclSuccess) return nccl_fail(e_, #expr); } while (0)
#define DUST_TRY(expr) do { DustStatus s_ = (expr); if (s_ != DUST_OK) return s_; } while (0)

constexpr uint32_t kMaxWorld = 64;

// what one rank of a loopback group has asked for; the group runs the collective when its last rank has
struct LocalCall {
  int op = 0;  // 0 none, 1 gather_bands, 2 gi_exchange, 3 surfel exchange (the sharded trace's records)
  DustHipPipeline* pipe = nullptr;
  uint32_t planes = 0;  // bit i: plane i travels
  std::vector<uint32_t> cuts;
  uint32_t root = 0;
  void* dst = nullptr;
  size_t dst_bytes = 0;
  uint32_t row_begin = 0, row_end = 0, band_rows = 0, frame_index = 0;
};
struct LocalGroup {
  uint32_t world = 0;
  std::vector<LocalCall> calls;  // per rank
  uint32_t pending = 0;
  int op = 0;                    // the collective the pending calls belong to
};

}  // namespace

struct DustHipComm {
  DustHipContext* ctx = nullptr;  // retained
  uint32_t rank = 0, world = 1;
  ncclComm_t nccl = nullptr;                // RCCL communicator, or
  std::shared_ptr<LocalGroup> local;        // the loopback group this rank belongs to
  hipStream_t stream = nullptr;             // the gathers' own stream (RCCL communicators only)
  static constexpr uint32_t kTickets = 16;
  hipEvent_t ev_ready = nullptr, ev_done[kTickets] = {};  // ev_done[t % kTickets]: gather number t has completed
  uint64_t next_ticket = 1;                  // tickets count the gathers of this communicator from 1
};

namespace {

__global__ void k_reduce_u32_max(uint32_t* const* bufs, uint32_t world, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t m = 0;
    for (uint32_t r = 0; r < world; ++r) m = max(m, bufs[r][i]);
    for (uint32_t r = 0; r < world; ++r) bufs[r][i] = m;
  }
}
__global__ void k_reduce_i32_sum(int32_t* const* bufs, uint32_t world, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int32_t s = 0;
    for (uint32_t r = 0; r < world; ++r) s += bufs[r][i];
    for (uint32_t r = 0; r < world; ++r) bufs[r][i] = s;
  }
}

struct PlaneView { uint8_t* ptr; size_t row_bytes; uint32_t width, height; };
DustStatus plane_view(DustHipPipeline* p, DustHipPlane plane, PlaneView* v) {
  void* ptr = nullptr;
  size_t bytes = 0;
  DUST_TRY(dust_hip_pipeline_plane_device_ptr(p, plane, &ptr, &bytes));
  dust_internal::pipeline_size(p, &v->width, &v->height);
  v->ptr = static_cast<uint8_t*>(ptr);
  v->row_bytes = bytes / v->height;
  return DUST_OK;
}
DustStatus check_cuts(const uint32_t* cuts, uint32_t world, uint32_t height) {
  if (!cuts || cuts[0] != 0 || cuts[world] != height) return set_error(DUST_ERR_INVALID_ARGUMENT, "band cuts must run from 0 to the frame's height");
  for (uint32_t r = 0; r < world; ++r)
    if (cuts[r] > cuts[r + 1]) return set_error(DUST_ERR_INVALID_ARGUMENT, "band cuts must not decrease");
  return DUST_OK;
}

// ---- loopback: everything on the one context's stream, run by the call that completes the group
DustStatus run_local_gather(LocalGroup& g, hipStream_t st) {
  const LocalCall& c0 = g.calls[0];
  for (const LocalCall& c : g.calls)
    if (c.planes != c0.planes || c.root != c0.root || c.cuts != c0.cuts) return set_error(DUST_ERR_INVALID_ARGUMENT, "the ranks of a gather disagree about planes, root or cuts");
  const LocalCall& root = g.calls[c0.root];
  for (uint32_t pl = 0; pl < DUST_PLANE_COUNT; ++pl) {
    if (!((c0.planes >> pl) & 1u)) continue;
    PlaneView rv;
    DUST_TRY(plane_view(root.pipe, DustHipPlane(pl), &rv));
    uint8_t* dst = root.dst ? static_cast<uint8_t*>(root.dst) : rv.ptr;
    if (root.dst && root.dst_bytes < rv.row_bytes * rv.height) return set_error(DUST_ERR_INVALID_ARGUMENT, "gather destination smaller than the plane");
    for (uint32_t r = 0; r < g.world; ++r) {
      PlaneView v;
      DUST_TRY(plane_view(g.calls[r].pipe, DustHipPlane(pl), &v));
      if (v.row_bytes != rv.row_bytes || v.height != rv.height) return set_error(DUST_ERR_INVALID_ARGUMENT, "the ranks' frames differ in size");
      const size_t off = size_t(c0.cuts[r]) * v.row_bytes, n = size_t(c0.cuts[r + 1] - c0.cuts[r]) * v.row_bytes;
      if (n && v.ptr + off != dst + off) HIP_TRY(hipMemcpyAsync(dst + off, v.ptr + off, n, hipMemcpyDeviceToDevice, st));
    }
  }
  return DUST_OK;
}
DustStatus run_local_gi(LocalGroup& g, hipStream_t st) {
  const uint32_t W = g.world;
  const LocalCall& c0 = g.calls[0];
  std::vector<DustHipGiExchange> ex(W);
  for (uint32_t r = 0; r < W; ++r) {
    if (g.calls[r].band_rows != c0.band_rows || g.calls[r].frame_index != c0.frame_index) return set_error(DUST_ERR_INVALID_ARGUMENT, "the ranks of a GI exchange disagree about band rows or frame");
    ex[r].struct_size = sizeof(DustHipGiExchange);
    DUST_TRY(dust_internal::gi_exchange_view(g.calls[r].pipe, W * c0.band_rows, &ex[r]));
    if (ex[r].pool_size != ex[0].pool_size || ex[r].width != ex[0].width) return set_error(DUST_ERR_INVALID_ARGUMENT, "the ranks' GI buffers differ in size");
  }
  if (W == 1) {   // a group of one (an emulated rank, a world-1 job): nothing to reduce or gather, and no wait
    DUST_TRY(dust_hip_gi_export(g.calls[0].pipe, g.calls[0].row_begin, g.calls[0].row_end));
    return dust_hip_gi_import(g.calls[0].pipe, g.calls[0].row_begin, g.calls[0].row_end, c0.frame_index);
  }
  // the pointer tables the reduction kernels read (uploaded from `host`, which the wait below keeps alive until the copy is done)
  void** table = nullptr;
  HIP_TRY(hipMalloc(&table, sizeof(void*) * W * 2));
  std::vector<void*> host(W * 2);
  for (uint32_t r = 0; r < W; ++r) { host[r] = ex[r].slot_owner; host[W + r] = ex[r].merged; }
  hipError_t e = hipMemcpyAsync(table, host.data(), sizeof(void*) * W * 2, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // (the table is read from this frame's kernels only; the wait keeps `host` alive long enough)
  DustStatus status = e == hipSuccess ? DUST_OK : hip_fail(e, "pointer table upload");
  if (status == DUST_OK) {
    hipLaunchKernelGGL(k_reduce_u32_max, dim3(256), dim3(256), 0, st, reinterpret_cast<uint32_t* const*>(table), W, size_t(ex[0].pool_size));
    const size_t band_items = size_t(c0.band_rows) * ex[0].width;
    for (uint32_t r = 0; r < W && status == DUST_OK; ++r)      // all-gather: band r of rank r's `touched` into every other rank's
      for (uint32_t q = 0; q < W && status == DUST_OK; ++q)
        if (q != r) {
          e = hipMemcpyAsync(static_cast<uint32_t*>(ex[q].touched) + r * band_items, static_cast<uint32_t*>(ex[r].touched) + r * band_items, band_items * 4,
                             hipMemcpyDeviceToDevice, st);
          if (e != hipSuccess) status = hip_fail(e, "all-gather copy");
        }
    for (uint32_t r = 0; r < W && status == DUST_OK; ++r) status = dust_hip_gi_export(g.calls[r].pipe, g.calls[r].row_begin, g.calls[r].row_end);
    if (status == DUST_OK)
      hipLaunchKernelGGL(k_reduce_i32_sum, dim3(256), dim3(256), 0, st, reinterpret_cast<int32_t* const*>(table + W), W, size_t(ex[0].pool_size) * 4);
    for (uint32_t r = 0; r < W && status == DUST_OK; ++r)
      status = dust_hip_gi_import(g.calls[r].pipe, g.calls[r].row_begin, g.calls[r].row_end, c0.frame_index);
  }
  e = hipStreamSynchronize(st);
  (void)hipFree(table);
  if (status == DUST_OK && e != hipSuccess) status = hip_fail(e, "loopback GI exchange");
  return status;
}
// the sharded surfel trace's second half on a loopback group: rank r's run of each staging array into every other rank's copy, then every
// rank's own completion (records to their surfels, stamps, ordered apply) -- what three all-gathers and the same kernels do on N devices
DustStatus run_local_surfel(LocalGroup& g, hipStream_t st) {
  const uint32_t W = g.world;
  std::vector<dust_internal::SurfelStage> sv(W);
  for (uint32_t r = 0; r < W; ++r) {
    if (g.calls[r].frame_index != g.calls[0].frame_index) return set_error(DUST_ERR_INVALID_ARGUMENT, "the ranks of a surfel exchange disagree about the frame");
    DUST_TRY(dust_internal::surfel_stage_view(g.calls[r].pipe, W, &sv[r]));
    if (sv[r].rank != r) return set_error(DUST_ERR_INVALID_ARGUMENT, "a rank's pending surfel trace was made for another rank of the group");
    if (sv[r].slots_per_rank != sv[0].slots_per_rank || sv[r].pool_size != sv[0].pool_size) return set_error(DUST_ERR_INVALID_ARGUMENT, "the ranks' surfel pools differ in size");
  }
  const size_t S = sv[0].slots_per_rank;
  for (uint32_t r = 0; r < W; ++r)
    for (uint32_t q = 0; q < W; ++q)
      if (q != r) {
        const size_t widths[3] = {32, 16, 16};
        void* src[3] = {sv[r].req, sv[r].repl, sv[r].sun};
        void* dst[3] = {sv[q].req, sv[q].repl, sv[q].sun};
        for (int k = 0; k < 3; ++k)
          HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(dst[k]) + size_t(r) * S * widths[k], static_cast<uint8_t*>(src[k]) + size_t(r) * S * widths[k], S * widths[k],
                                 hipMemcpyDeviceToDevice, st));
      }
  for (uint32_t r = 0; r < W; ++r) DUST_TRY(dust_internal::surfel_finish(g.calls[r].pipe, g.calls[0].frame_index));
  return DUST_OK;
}
// rank `c`'s part of a collective on a loopback group: remember it; the call that completes the group runs it for everyone
DustStatus local_call(DustHipComm* c, LocalCall&& call) {
  LocalGroup& g = *c->local;
  // a call that cannot join the pending collective fails AND drops what was pending (the other ranks' records hold raw pipeline
  // pointers: nothing of a broken collective is kept around for a later call to complete)
  auto drop = [&g](const char* why) {
    for (LocalCall& lc : g.calls) lc = LocalCall();
    g.pending = 0;
    return set_error(DUST_ERR_INVALID_ARGUMENT, why);
  };
  if (g.calls[c->rank].op != 0) return drop("this rank already has a collective pending: every rank of the group must make the call before any makes the next (the pending collective was dropped)");
  if (g.pending && g.op != call.op) return drop("the ranks of a loopback group are in different collectives (the pending collective was dropped)");
  const int op = g.op = call.op;
  g.calls[c->rank] = std::move(call);
  if (++g.pending < g.world) return DUST_OK;
  const hipStream_t st = dust_internal::context_stream(c->ctx);
  const DustStatus s = op == 1 ? run_local_gather(g, st) : (op == 2 ? run_local_gi(g, st) : run_local_surfel(g, st));
  for (LocalCall& lc : g.calls) lc = LocalCall();
  g.pending = 0;
  return s;
}

}  // namespace

namespace dust { const char* comm_kernels_anchor() { return "k_reduce_u32_max"; } }

extern "C" {

DustStatus dust_hip_comm_unique_id(uint8_t id[DUST_HIP_COMM_ID_BYTES]) {
  if (!id) return set_error(DUST_ERR_INVALID_ARGUMENT, "null id");
#if DUST_HAVE_RCCL
  Rccl* r = rccl();
  if (!r) return no_rccl();
  static_assert(DUST_HIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  ncclUniqueId u;
  NCCL_TRY(r->GetUniqueId(&u));
  std::memcpy(id, u.internal, DUST_HIP_COMM_ID_BYTES);
  return DUST_OK;
#else
  return no_rccl();
#endif
}

DustStatus dust_hip_comm_create(DustHipContext* ctx, uint32_t rank, uint32_t world, const uint8_t id[DUST_HIP_COMM_ID_BYTES], DustHipComm** out) {
  if (!ctx || !id || !out || world == 0 || world > kMaxWorld || rank >= world) return set_error(DUST_ERR_INVALID_ARGUMENT, "bad communicator arguments");
#if DUST_HAVE_RCCL
  Rccl* r = rccl();
  if (!r) return no_rccl();
  HIP_TRY(hipSetDevice(dust_internal::context_device(ctx)));
  std::unique_ptr<DustHipComm> c(new (std::nothrow) DustHipComm);
  if (!c) return set_error(DUST_ERR_OUT_OF_MEMORY, "host allocation failed");
  ncclUniqueId u;
  std::memcpy(u.internal, id, DUST_HIP_COMM_ID_BYTES);
  NCCL_TRY(r->CommInitRank(&c->nccl, int(world), u, int(rank)));
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
  for (hipEvent_t& ev : c->ev_done)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    for (hipEvent_t ev : c->ev_done) if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    (void)r->CommDestroy(c->nccl);
    return hip_fail(e, "communicator stream");
  }
  c->ctx = ctx;
  dust_internal::context_retain(ctx);
  dust_internal::context_add_stream(ctx, c->stream);  // (gathers read pipelines' planes on it: every wait for the context covers it)
  c->rank = rank; c->world = world;
  *out = c.release();
  return DUST_OK;
#else
  (void)rank;
  return no_rccl();
#endif
}

DustStatus dust_hip_comm_create_local(DustHipContext* ctx, uint32_t world, DustHipComm** out) {
  if (!ctx || !out || world == 0 || world > kMaxWorld) return set_error(DUST_ERR_INVALID_ARGUMENT, "bad loopback group size");
  try {
    auto g = std::make_shared<LocalGroup>();
    g->world = world;
    g->calls.resize(world);
    std::vector<std::unique_ptr<DustHipComm>> made;
    for (uint32_t r = 0; r < world; ++r) {
      made.emplace_back(new DustHipComm);
      made.back()->rank = r; made.back()->world = world; made.back()->local = g;
    }
    for (uint32_t r = 0; r < world; ++r) {
      made[r]->ctx = ctx;
      dust_internal::context_retain(ctx);
      out[r] = made[r].release();
    }
    return DUST_OK;
  } catch (const std::bad_alloc&) {
    return set_error(DUST_ERR_OUT_OF_MEMORY, "host allocation failed");
  }
}

void dust_hip_comm_destroy(DustHipComm* c) {
  if (!c) return;
  (void)hipSetDevice(dust_internal::context_device(c->ctx));
  if (c->stream) (void)hipStreamSynchronize(c->stream);
#if DUST_HAVE_RCCL
  if (c->nccl) { if (Rccl* r = rccl()) (void)r->CommDestroy(c->nccl); }
#endif
  if (c->stream) dust_internal::context_remove_stream(c->ctx, c->stream);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  for (hipEvent_t ev : c->ev_done) if (ev) (void)hipEventDestroy(ev);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  dust_internal::context_release(c->ctx);
  delete c;
}

DustStatus dust_hip_comm_info(const DustHipComm* c, uint32_t* rank, uint32_t* world, uint32_t* is_local) {
  if (!c) return set_error(DUST_ERR_INVALID_ARGUMENT, "null communicator");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (is_local) *is_local = c->local ? 1u : 0u;
  return DUST_OK;
}

static DustStatus gather_planes(DustHipPipeline* p, DustHipComm* c, uint32_t planes, const uint32_t* cuts, uint32_t root, void* dst, size_t dst_bytes, uint64_t* ticket) {
  if (!p || !c || planes == 0 || (planes >> DUST_PLANE_COUNT) != 0 || root >= c->world) return set_error(DUST_ERR_INVALID_ARGUMENT, "bad gather arguments");
  if (ticket) *ticket = 0;
  if (dust_internal::pipeline_context(p) != c->ctx) return set_error(DUST_ERR_INVALID_ARGUMENT, "pipeline and communicator belong to different contexts");
  uint32_t height = 0, width = 0;
  dust_internal::pipeline_size(p, &width, &height);
  DUST_TRY(check_cuts(cuts, c->world, height));
  HIP_TRY(hipSetDevice(dust_internal::context_device(c->ctx)));
  if (c->world > 1 && !c->local) dust_internal::pipeline_note_collective(p);  // (RCCL's kernels will want workgroup slots beside this pipeline's launches)
  if (c->local) {
    LocalCall call;
    call.op = 1; call.pipe = p; call.planes = planes; call.cuts.assign(cuts, cuts + c->world + 1); call.root = root; call.dst = dst; call.dst_bytes = dst_bytes;
    return local_call(c, std::move(call));
  }
#if DUST_HAVE_RCCL
  Rccl* r = rccl();
  if (!r) return no_rccl();
  const bool is_root = c->rank == root;
  // behind the frame that has just been enqueued on the context's stream, on the communicator's own stream
  const hipStream_t main = dust_internal::context_stream(c->ctx);
  HIP_TRY(hipEventRecord(c->ev_ready, main));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
  PlaneView views[DUST_PLANE_COUNT];
  for (uint32_t pl = 0; pl < DUST_PLANE_COUNT; ++pl)
    if ((planes >> pl) & 1u) {
      DUST_TRY(plane_view(p, DustHipPlane(pl), &views[pl]));
      if (is_root && dst && dst_bytes < views[pl].row_bytes * views[pl].height) return set_error(DUST_ERR_INVALID_ARGUMENT, "gather destination smaller than the plane");
    }
  NCCL_TRY(r->GroupStart());  // ONE group for every plane and peer: all transfers of the frame are in flight together, each peer over its own link
  ncclResult_t ne = ncclSuccess;
  for (uint32_t pl = 0; pl < DUST_PLANE_COUNT && ne == ncclSuccess; ++pl) {
    if (!((planes >> pl) & 1u)) continue;
    const PlaneView& v = views[pl];
    uint8_t* out = is_root ? (dst ? static_cast<uint8_t*>(dst) : v.ptr) : nullptr;
    if (is_root) {
      for (uint32_t q = 0; q < c->world && ne == ncclSuccess; ++q) {
        const size_t off = size_t(cuts[q]) * v.row_bytes, n = size_t(cuts[q + 1] - cuts[q]) * v.row_bytes;
        if (q != root && n) ne = r->Recv(out + off, n, ncclChar, int(q), c->nccl, c->stream);
      }
    } else {
      const size_t off = size_t(cuts[c->rank]) * v.row_bytes, n = size_t(cuts[c->rank + 1] - cuts[c->rank]) * v.row_bytes;
      if (n) ne = r->Send(v.ptr + off, n, ncclChar, int(root), c->nccl, c->stream);
    }
  }
  const ncclResult_t ge = r->GroupEnd();
  if (ne != ncclSuccess) return nccl_fail(ne, "ncclSend / ncclRecv");
  if (ge != ncclSuccess) return nccl_fail(ge, "ncclGroupEnd");
  if (is_root && dst)   // the root's own rows (one plane: gather_bands with a destination)
    for (uint32_t pl = 0; pl < DUST_PLANE_COUNT; ++pl)
      if ((planes >> pl) & 1u) {
        const PlaneView& v = views[pl];
        const size_t off = size_t(cuts[root]) * v.row_bytes, n = size_t(cuts[root + 1] - cuts[root]) * v.row_bytes;
        if (n && dst != v.ptr) HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(dst) + off, v.ptr + off, n, hipMemcpyDeviceToDevice, c->stream));
      }
  const uint64_t t = c->next_ticket++;
  HIP_TRY(hipEventRecord(c->ev_done[t % DustHipComm::kTickets], c->stream));
  if (ticket) *ticket = t;
  return DUST_OK;
#else
  return no_rccl();
#endif
}
DustStatus dust_hip_gather_bands(DustHipPipeline* p, DustHipComm* c, DustHipPlane plane, const uint32_t* cuts, uint32_t root, void* dst, size_t dst_bytes,
                                 uint64_t* ticket) {
  if (plane >= DUST_PLANE_COUNT) return set_error(DUST_ERR_INVALID_ARGUMENT, "bad gather arguments");
  return gather_planes(p, c, 1u << plane, cuts, root, dst, dst_bytes, ticket);
}
DustStatus dust_hip_gather_planes(DustHipPipeline* p, DustHipComm* c, uint32_t plane_mask, const uint32_t* cuts, uint32_t root, uint64_t* ticket) {
  return gather_planes(p, c, plane_mask, cuts, root, nullptr, 0, ticket);
}

DustStatus dust_hip_comm_wait(DustHipComm* c, uint64_t ticket) {
  if (!c) return set_error(DUST_ERR_INVALID_ARGUMENT, "null communicator");
  if (c->local || c->next_ticket == 1) return DUST_OK;  // (a loopback group's collectives run on the context's stream itself)
  if (ticket >= c->next_ticket) return set_error(DUST_ERR_INVALID_ARGUMENT, "no such gather");
  // gathers complete in order on the communicator's stream: a ticket whose event has been reused since is covered by the latest one
  const uint64_t t = (ticket == 0 || ticket + DustHipComm::kTickets < c->next_ticket) ? c->next_ticket - 1 : ticket;
  HIP_TRY(hipSetDevice(dust_internal::context_device(c->ctx)));
  HIP_TRY(hipStreamWaitEvent(dust_internal::context_stream(c->ctx), c->ev_done[t % DustHipComm::kTickets], 0));
  return DUST_OK;
}

DustStatus dust_hip_comm_sync(DustHipComm* c) {
  if (!c) return set_error(DUST_ERR_INVALID_ARGUMENT, "null communicator");
  HIP_TRY(hipSetDevice(dust_internal::context_device(c->ctx)));
  if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
  return dust_hip_sync(c->ctx);
}

DustStatus dust_hip_gi_exchange_run(DustHipPipeline* p, DustHipComm* c, uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t frame_index) {
  if (!p || !c || band_rows == 0 || row_begin > row_end) return set_error(DUST_ERR_INVALID_ARGUMENT, "bad GI exchange arguments");
  if (dust_internal::pipeline_context(p) != c->ctx) return set_error(DUST_ERR_INVALID_ARGUMENT, "pipeline and communicator belong to different contexts");
  HIP_TRY(hipSetDevice(dust_internal::context_device(c->ctx)));
  if (c->world > 1 && !c->local) dust_internal::pipeline_note_collective(p);
  if (c->local) {
    LocalCall call;
    call.op = 2; call.pipe = p; call.row_begin = row_begin; call.row_end = row_end; call.band_rows = band_rows; call.frame_index = frame_index;
    return local_call(c, std::move(call));
  }
  DustHipGiExchange ex;
  ex.struct_size = sizeof ex;
  DUST_TRY(dust_internal::gi_exchange_view(p, c->world * band_rows, &ex));
  if (c->world == 1) {  // nothing to exchange: the band is the frame
    DUST_TRY(dust_hip_gi_export(p, row_begin, row_end));
    return dust_hip_gi_import(p, row_begin, row_end, frame_index);
  }
#if DUST_HAVE_RCCL
  Rccl* r = rccl();
  if (!r) return no_rccl();
  const hipStream_t st = dust_internal::context_stream(c->ctx);
  const size_t band_items = size_t(band_rows) * ex.width;
  NCCL_TRY(r->AllReduce(ex.slot_owner, ex.slot_owner, ex.pool_size, ncclUint32, ncclMax, c->nccl, st));
  NCCL_TRY(r->AllGather(static_cast<uint32_t*>(ex.touched) + c->rank * band_items, ex.touched, band_items, ncclUint32, c->nccl, st));  // in place: band r at r * band_items
  DUST_TRY(dust_hip_gi_export(p, row_begin, row_end));
  NCCL_TRY(r->AllReduce(ex.merged, ex.merged, size_t(ex.pool_size) * 4, ncclInt32, ncclSum, c->nccl, st));  // one contributor per slot: exact
  return dust_hip_gi_import(p, row_begin, row_end, frame_index);
#else
  return no_rccl();
#endif
}

DustStatus dust_hip_gi_surfel_exchange_run(DustHipPipeline* p, DustHipComm* c, uint32_t frame_index) {
  if (!p) return set_error(DUST_ERR_INVALID_ARGUMENT, "bad surfel exchange arguments");
  if (!c) {   // no communicator: nothing travels (a world of one; one emulated rank of N, whose peers' records are what the staging arrays hold)
    dust_internal::SurfelStage sv;
    DUST_TRY(dust_internal::surfel_stage_view(p, 0, &sv));
    return dust_internal::surfel_finish(p, frame_index);
  }
  if (dust_internal::pipeline_context(p) != c->ctx) return set_error(DUST_ERR_INVALID_ARGUMENT, "pipeline and communicator belong to different contexts");
  HIP_TRY(hipSetDevice(dust_internal::context_device(c->ctx)));
  if (c->world > 1 && !c->local) dust_internal::pipeline_note_collective(p);
  if (c->local) {
    LocalCall call;
    call.op = 3; call.pipe = p; call.frame_index = frame_index;
    return local_call(c, std::move(call));
  }
  dust_internal::SurfelStage sv;
  DUST_TRY(dust_internal::surfel_stage_view(p, c->world, &sv));
  if (sv.rank != c->rank) return set_error(DUST_ERR_INVALID_ARGUMENT, "the pending surfel trace was made for another rank than the communicator's");
  if (c->world > 1) {
#if DUST_HAVE_RCCL
    Rccl* r = rccl();
    if (!r) return no_rccl();
    const hipStream_t st = dust_internal::context_stream(c->ctx);
    const size_t S = sv.slots_per_rank;
    // in place: rank r's run stands at r * S of each array. One group: the three all-gathers share their rounds over the links.
    NCCL_TRY(r->GroupStart());
    ncclResult_t ne = r->AllGather(static_cast<uint8_t*>(sv.req) + size_t(c->rank) * S * 32, sv.req, S * 32, ncclChar, c->nccl, st);
    if (ne == ncclSuccess) ne = r->AllGather(static_cast<uint8_t*>(sv.repl) + size_t(c->rank) * S * 16, sv.repl, S * 16, ncclChar, c->nccl, st);
    if (ne == ncclSuccess) ne = r->AllGather(static_cast<uint8_t*>(sv.sun) + size_t(c->rank) * S * 16, sv.sun, S * 16, ncclChar, c->nccl, st);
    const ncclResult_t ge = r->GroupEnd();
    if (ne != ncclSuccess) return nccl_fail(ne, "ncclAllGather");
    if (ge != ncclSuccess) return nccl_fail(ge, "ncclGroupEnd");
#else
    return no_rccl();
#endif
  }
  return dust_internal::surfel_finish(p, frame_index);
}

}  // extern "C"
