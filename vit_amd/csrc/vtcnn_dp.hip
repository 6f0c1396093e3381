// Data-parallel gradient exchange for a host that drives several GPUs from ONE process (the C++ CLI).
// The reference trains on one device (utils/trainer.hpp:28-101); this is the multi-GPU form of the same
// loop: every rank runs forward + backward on its own mini-batch, the gradient arenas are summed over
// RCCL, and every rank applies the same SGD update with divisor ranks * batch_size.
//
// RCCL is opened at run time (dlopen), on the first vt_dp_init, so that the library itself carries no
// link-time dependency on it: a Python process that already mapped the RCCL bundled with its tensor
// library never sees a second copy unless it asks for these entry points.
#include "vtcnn_common.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {

// The handful of RCCL declarations this file needs (rccl.h: ncclResult_t, ncclDataType_t, ncclRedOp_t).
using nccl_comm = void *;
constexpr int NCCL_SUCCESS = 0;
constexpr int NCCL_FLOAT32 = 7;
constexpr int NCCL_SUM = 0;

struct Rccl {
  void *handle = nullptr;
  int (*CommInitAll)(nccl_comm *, int, const int *) = nullptr;
  int (*CommDestroy)(nccl_comm) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void open_rccl() {
  const char *override_path = getenv("VT_RCCL_LIBRARY");
  const char *names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    if (!n || !*n) continue;
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) return;
  auto sym = [&](const char *n) { return dlsym(g_rccl.handle, n); };
  g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(sym("ncclCommInitAll"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
  g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
  g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
  g_rccl.ok = g_rccl.CommInitAll && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.GroupStart && g_rccl.GroupEnd;
}

int rccl_fail(const char *what, int rc) {
  fprintf(stderr, "vtcnn_hip: %s failed: %s\n", what,
          g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
  return VT_EIO;
}

}  // namespace

struct vt_dp {
  std::vector<int> devices;
  std::vector<nccl_comm> comms;
  std::vector<hipEvent_t> done;  // one per rank: end of the last exchange issued on that rank's stream
  std::vector<char> issued;      // per rank, so that one host thread per rank may issue without sharing a flag
  // overlapped exchange (vt_dp_exchange_begin / _join): a stream of the exchange's own per rank, and the event that
  // orders it behind the compute stream
  std::vector<hipStream_t> side;
  std::vector<hipEvent_t> ready;
};

extern "C" {

int vt_dp_init(int ranks, const int *device_ids, vt_dp **out) {
  if (!out || !device_ids || ranks < 1 || ranks > 64) return VT_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) return VT_EUNSUPPORTED;
  for (int r = 0; r < ranks; ++r) {
    if (device_ids[r] < 0 || device_ids[r] >= ndev) return VT_EINVAL;
    for (int q = 0; q < r; ++q)
      if (device_ids[q] == device_ids[r]) return VT_EINVAL;  // one rank per GPU
  }
  std::call_once(g_rccl_once, open_rccl);
  if (!g_rccl.ok) return VT_EUNSUPPORTED;
  int keep = 0;
  (void)hipGetDevice(&keep);
  vt_dp *dp = nullptr;
  try {
    dp = new vt_dp;
    dp->devices.assign(device_ids, device_ids + ranks);
    dp->comms.assign(ranks, nullptr);
    dp->done.assign(ranks, nullptr);
    dp->issued.assign(ranks, 0);
    dp->side.assign(ranks, nullptr);
    dp->ready.assign(ranks, nullptr);
  } catch (...) {
    delete dp;
    return VT_ENOMEM;
  }
  int rc = g_rccl.CommInitAll(dp->comms.data(), ranks, dp->devices.data());
  if (rc != NCCL_SUCCESS) {
    delete dp;
    (void)hipSetDevice(keep);
    return rccl_fail("ncclCommInitAll", rc);
  }
  for (int r = 0; r < ranks; ++r) {
    if (hipSetDevice(dp->devices[r]) != hipSuccess ||
        hipEventCreateWithFlags(&dp->done[r], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&dp->ready[r], hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&dp->side[r], hipStreamNonBlocking) != hipSuccess) {
      (void)hipSetDevice(keep);
      vt_dp_destroy(dp);
      return VT_ENOMEM;
    }
  }
  (void)hipSetDevice(keep);
  *out = dp;
  return 0;
}

int vt_dp_destroy(vt_dp *dp) {
  if (!dp) return 0;
  int keep = 0;
  (void)hipGetDevice(&keep);
  for (size_t r = 0; r < dp->devices.size(); ++r) {
    (void)hipSetDevice(dp->devices[r]);
    (void)hipDeviceSynchronize();
    if (dp->done[r]) (void)hipEventDestroy(dp->done[r]);
    if (dp->ready[r]) (void)hipEventDestroy(dp->ready[r]);
    if (dp->side[r]) (void)hipStreamDestroy(dp->side[r]);
    if (dp->comms[r]) g_rccl.CommDestroy(dp->comms[r]);
  }
  (void)hipSetDevice(keep);
  delete dp;
  return 0;
}

int vt_dp_ranks(const vt_dp *dp) { return dp ? (int)dp->devices.size() : VT_EINVAL; }

int vt_dp_allreduce_async(vt_dp *dp, float *const *d_bufs, long first, long count, const vt_stream *streams) {
  if (!dp || !d_bufs || !streams || first < 0 || count < 0) return VT_EINVAL;
  const int ranks = (int)dp->devices.size();
  for (int r = 0; r < ranks; ++r)
    if (!d_bufs[r]) return VT_EINVAL;
  if (count == 0) return 0;
  int keep = 0;
  (void)hipGetDevice(&keep);
  int rc = g_rccl.GroupStart();
  if (rc != NCCL_SUCCESS) return rccl_fail("ncclGroupStart", rc);
  int bad = NCCL_SUCCESS;
  for (int r = 0; r < ranks && bad == NCCL_SUCCESS; ++r) {
    float *p = d_bufs[r] + first;
    bad = g_rccl.AllReduce(p, p, (size_t)count, NCCL_FLOAT32, NCCL_SUM, dp->comms[r], (hipStream_t)streams[r]);
  }
  rc = g_rccl.GroupEnd();
  if (bad != NCCL_SUCCESS) return rccl_fail("ncclAllReduce", bad);
  if (rc != NCCL_SUCCESS) return rccl_fail("ncclGroupEnd", rc);
  for (int r = 0; r < ranks; ++r) {
    (void)hipSetDevice(dp->devices[r]);
    if (hipEventRecord(dp->done[r], (hipStream_t)streams[r]) != hipSuccess) {
      (void)hipSetDevice(keep);
      return VT_EIO;
    }
  }
  (void)hipSetDevice(keep);
  for (int r = 0; r < ranks; ++r) dp->issued[r] = 1;
  return 0;
}

int vt_dp_allreduce_rank_async(vt_dp *dp, int rank, float *d_buf, long first, long count, vt_stream stream) {
  if (!dp || rank < 0 || rank >= (int)dp->devices.size() || !d_buf || first < 0 || count < 0) return VT_EINVAL;
  if (count == 0) return 0;
  int keep = 0;
  (void)hipGetDevice(&keep);
  if (keep != dp->devices[rank] && hipSetDevice(dp->devices[rank]) != hipSuccess) return VT_EINVAL;
  float *p = d_buf + first;
  const int rc = g_rccl.AllReduce(p, p, (size_t)count, NCCL_FLOAT32, NCCL_SUM, dp->comms[rank], (hipStream_t)stream);
  const hipError_t ev = rc == NCCL_SUCCESS ? hipEventRecord(dp->done[rank], (hipStream_t)stream) : hipSuccess;
  if (keep != dp->devices[rank]) (void)hipSetDevice(keep);
  if (rc != NCCL_SUCCESS) return rccl_fail("ncclAllReduce", rc);
  if (ev != hipSuccess) return VT_EIO;
  dp->issued[rank] = 1;
  return 0;
}

// One bucket of an overlapped exchange: the sum runs on the rank's exchange stream behind everything the compute
// stream holds at the time of the call, while later work on the compute stream runs beside it.
int vt_dp_exchange_begin(vt_dp *dp, int rank, float *d_buf, long first, long count, vt_stream compute) {
  if (!dp || rank < 0 || rank >= (int)dp->devices.size() || !d_buf || first < 0 || count < 0) return VT_EINVAL;
  if (count == 0) return 0;
  int keep = 0;
  (void)hipGetDevice(&keep);
  if (keep != dp->devices[rank] && hipSetDevice(dp->devices[rank]) != hipSuccess) return VT_EINVAL;
  int out = 0;
  float *p = d_buf + first;
  if (hipEventRecord(dp->ready[rank], (hipStream_t)compute) != hipSuccess ||
      hipStreamWaitEvent(dp->side[rank], dp->ready[rank], 0) != hipSuccess) {
    out = VT_EIO;
  } else {
    const int rc = g_rccl.AllReduce(p, p, (size_t)count, NCCL_FLOAT32, NCCL_SUM, dp->comms[rank], dp->side[rank]);
    if (rc != NCCL_SUCCESS) out = rccl_fail("ncclAllReduce", rc);
    else if (hipEventRecord(dp->done[rank], dp->side[rank]) != hipSuccess) out = VT_EIO;
    else dp->issued[rank] = 1;
  }
  if (keep != dp->devices[rank]) (void)hipSetDevice(keep);
  return out;
}

int vt_dp_exchange_join(vt_dp *dp, int rank, vt_stream compute) {
  if (!dp || rank < 0 || rank >= (int)dp->devices.size()) return VT_EINVAL;
  if (!dp->issued[rank]) return 0;
  return hipStreamWaitEvent((hipStream_t)compute, dp->done[rank], 0) == hipSuccess ? 0 : VT_EIO;
}

int vt_dp_wait(vt_dp *dp) {
  if (!dp) return VT_EINVAL;
  for (size_t r = 0; r < dp->devices.size(); ++r)
    if (dp->issued[r] && hipEventSynchronize(dp->done[r]) != hipSuccess) return VT_EIO;
  return 0;
}

}  // extern "C"
