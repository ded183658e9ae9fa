// qk_comm.hip -- the multi-GPU part of the C ABI: one process, k MI355X of one node, RCCL over xGMI.
//
// What it replaces (G = gpu_backend/kernel_state_ansatz.py of the reference):
//   G:149-199  rank / chunk bookkeeping on an mpi4py communicator (one process per GPU, device = rank % n_devices)
//   G:341-352, 415-419  the ring of pickled MPS between the ranks
//   G:428      comm.reduce(kernel_mat, SUM): every rank contributes a full-size matrix that is zero outside its tiles
// by: every device holds the whole (read-only) set -- each device's share travels ONCE as a packed image
// (qk_mps_set_allgather: one ncclAllGather of the planes) --, every device sweeps its share of the pair list in one
// persistent launch (qk_plan_create(world, rank): tiles dealt by cost) and the shares meet in ONE ncclAllGather of the
// packed values; a scatter kernel on every device then fills (and mirrors) the dense K.
//
// RCCL is resolved at run time (dlopen "librccl.so.1"): a process that already holds an RCCL -- PyTorch's -- keeps using
// that copy, a plain C caller gets /opt/rocm's; libqkgram.so has no link-time dependency on it and the single-GPU entry
// points work without it.
#include "qk_host.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok() const { return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString; }
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (x.handle) break;
    }
    if (!x.handle) return x;
    x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(x.handle, "ncclCommInitAll"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.handle, "ncclAllGather"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.handle, "ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.handle, "ncclGroupEnd"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.handle, "ncclGetErrorString"));
    return x;
  }();
  return r;
}

#define NCCL_TRY(expr)                                                                                          \
  do {                                                                                                          \
    const ncclResult_t r_ = (expr);                                                                             \
    if (r_ != ncclSuccess) return qk_fail(QK_EDEVICE, "%s failed: %s", #expr, rccl().GetErrorString(r_));       \
  } while (0)

struct DevMem {  // device memory of one rank, released with the communicator or when the job changes
  int device = 0;
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(const int dev, const size_t need) {
    if (need <= bytes) return QK_OK;
    release();
    device = dev;
    HIP_TRY(hipSetDevice(dev));
    HIP_TRY(hipMalloc(&p, need));
    bytes = need;
    return QK_OK;
  }
  void release() {
    if (p) {
      (void)hipSetDevice(device);
      (void)hipFree(p);
    }
    p = nullptr, bytes = 0;
  }
  template <typename T>
  T* as() const { return static_cast<T*>(p); }
};

}  // namespace

struct qk_comm {
  int n = 0;
  std::vector<int> devices;
  std::vector<qk_ctx*> ctx;
  std::vector<ncclComm_t> nccl;
  // the job of the last qk_gram_sharded call, kept while the same sets come back (a bench loop re-plans nothing)
  std::vector<const qk_mps_set*> job_x, job_y;
  std::vector<uint64_t> job_xid, job_yid;  // their uids: a destroyed set's address may be reused by another set
  std::vector<qk_plan*> plans;
  std::vector<DevMem> vals, all_vals, all_pairs, k;
  int64_t maxp = 0;
  qk_stats last_stats[16];
  double gather_ms = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;  // on rank 0's stream: around the all-gather
};

static void drop_job(qk_comm* c) {
  for (qk_plan* p : c->plans) qk_plan_destroy(p);
  c->plans.clear(), c->job_x.clear(), c->job_y.clear(), c->job_xid.clear(), c->job_yid.clear();
}

extern "C" int qk_comm_destroy(qk_comm* c) {
  if (!c) return QK_OK;
  for (int r = 0; r < (int)c->ctx.size(); ++r)
    if (c->ctx[(size_t)r]) (void)qk_ctx_synchronize(c->ctx[(size_t)r]);
  drop_job(c);
  for (auto* v : {&c->vals, &c->all_vals, &c->all_pairs, &c->k})
    for (DevMem& m : *v) m.release();
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (ncclComm_t nc : c->nccl)
    if (nc) (void)rccl().CommDestroy(nc);
  for (qk_ctx* x : c->ctx) qk_ctx_destroy(x);
  delete c;
  return QK_OK;
}

extern "C" int qk_comm_init_all(int32_t n_devices, const int32_t* device_ids, qk_comm** out) {
  if (!out || n_devices <= 0 || n_devices > 16) return qk_fail(QK_EINVAL, "qk_comm_init_all: bad argument (n_devices %d)", n_devices);
  const int have = qk_device_count();
  if (have <= 0) return qk_fail(QK_EDEVICE, "qk_comm_init_all: no HIP device available; this engine has no CPU fallback");
  if (!rccl().ok()) return qk_fail(QK_EDEVICE, "qk_comm_init_all: librccl.so.1 could not be loaded (%s)", rccl().handle ? "missing symbols" : dlerror());
  qk_comm* c = new (std::nothrow) qk_comm;
  if (!c) return qk_fail(QK_ENOMEM, "qk_comm_init_all: out of memory");
  c->n = n_devices;
  for (int r = 0; r < n_devices; ++r) {
    const int d = device_ids ? device_ids[r] : r;
    if (d < 0 || d >= have || std::find(c->devices.begin(), c->devices.end(), d) != c->devices.end()) {
      qk_comm_destroy(c);
      return qk_fail(QK_EINVAL, "qk_comm_init_all: device %d out of range [0, %d) or named twice (one rank per GPU)", d, have);
    }
    c->devices.push_back(d);
  }
  c->ctx.assign((size_t)n_devices, nullptr);
  c->nccl.assign((size_t)n_devices, nullptr);
  for (int r = 0; r < n_devices; ++r) {
    const int rc = qk_ctx_create(c->devices[(size_t)r], &c->ctx[(size_t)r]);
    if (rc != QK_OK) {
      qk_comm_destroy(c);
      return rc;
    }
  }
  const ncclResult_t nr = rccl().CommInitAll(c->nccl.data(), n_devices, c->devices.data());
  if (nr != ncclSuccess) {
    std::fill(c->nccl.begin(), c->nccl.end(), nullptr);
    qk_comm_destroy(c);
    return qk_fail(QK_EDEVICE, "qk_comm_init_all: ncclCommInitAll over %d device(s) failed: %s", n_devices, rccl().GetErrorString(nr));
  }
  c->vals.resize((size_t)n_devices), c->all_vals.resize((size_t)n_devices), c->all_pairs.resize((size_t)n_devices), c->k.resize((size_t)n_devices);
  (void)hipSetDevice(c->devices[0]);
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    qk_comm_destroy(c);
    return qk_fail(QK_EDEVICE, "qk_comm_init_all: hipEventCreate failed");
  }
  *out = c;
  return QK_OK;
}

extern "C" int32_t qk_comm_size(const qk_comm* c) { return c ? c->n : 0; }
extern "C" qk_ctx* qk_comm_ctx(qk_comm* c, int32_t rank) { return (c && rank >= 0 && rank < c->n) ? c->ctx[(size_t)rank] : nullptr; }

// ---- every device gets the whole set: ONE all-gather of the packed images --------------------------------------------
extern "C" int qk_mps_set_allgather(qk_comm* c, qk_mps_set* const* local, const int32_t* lo, int32_t total, qk_mps_set** full_out) {
  if (!c || !local || !lo || !full_out || total <= 0) return qk_fail(QK_EINVAL, "qk_mps_set_allgather: bad argument");
  const int n = c->n;
  int n_sites = 0;
  int64_t mx = 1;  // doubles per rank in the exchange: the largest image
  for (int r = 0; r < n; ++r) {
    full_out[r] = nullptr;
    const qk_mps_set* m = local[r];
    if (!m) continue;  // an empty share
    if (m->ctx != c->ctx[(size_t)r]) return qk_fail(QK_EINVAL, "qk_mps_set_allgather: share %d does not live on the communicator's context %d", r, r);
    if (m->precision != 64) return qk_fail(QK_EINVAL, "qk_mps_set_allgather: only fp64 sets are exchanged");
    if (n_sites && m->n_sites != n_sites) return qk_fail(QK_EINVAL, "qk_mps_set_allgather: shares differ in their number of sites");
    n_sites = m->n_sites;
    if (lo[r] < 0 || lo[r] + m->n_states > total) return qk_fail(QK_EINVAL, "qk_mps_set_allgather: share %d = states [%d, %d) outside [0, %d)", r, lo[r], lo[r] + m->n_states, total);
    mx = std::max<int64_t>(mx, m->bytes / (int64_t)sizeof(double));
  }
  if (!n_sites) return qk_fail(QK_EINVAL, "qk_mps_set_allgather: every share is empty");
  // tables of the whole set (host side: this is one process); a share's offsets move behind the images of the ranks before it
  const int stride = n_sites + 1;
  std::vector<int32_t> dims((size_t)total * stride, 0);
  std::vector<int64_t> offs((size_t)total * n_sites, 0);
  for (int r = 0; r < n; ++r) {
    const qk_mps_set* m = local[r];
    if (!m) continue;
    std::vector<int64_t> o((size_t)m->n_states * n_sites);
    const int rc = qk_mps_set_image(m, nullptr, nullptr, nullptr, o.data());
    if (rc != QK_OK) return rc;
    for (int s = 0; s < m->n_states; ++s) {
      if (dims[(size_t)(lo[r] + s) * stride] != 0) return qk_fail(QK_EINVAL, "qk_mps_set_allgather: state %d belongs to two shares", lo[r] + s);
      std::memcpy(&dims[(size_t)(lo[r] + s) * stride], &m->dims_true[(size_t)s * stride], sizeof(int32_t) * stride);
      for (int k = 0; k < n_sites; ++k) offs[(size_t)(lo[r] + s) * n_sites + k] = o[(size_t)s * n_sites + k] + (int64_t)r * mx;
    }
  }
  for (int s = 0; s < total; ++s)
    if (dims[(size_t)s * stride] != 1) return qk_fail(QK_EINVAL, "qk_mps_set_allgather: state %d belongs to no share", s);
  // send / receive buffers, the collective, the assembly
  std::vector<DevMem> send((size_t)n), recv((size_t)n);
  int rc = QK_OK;
  for (int r = 0; r < n && rc == QK_OK; ++r) {
    rc = send[(size_t)r].ensure(c->devices[(size_t)r], (size_t)mx * sizeof(double));
    if (rc == QK_OK) rc = recv[(size_t)r].ensure(c->devices[(size_t)r], (size_t)n * mx * sizeof(double));
    if (rc == QK_OK && local[r]) rc = qk_mps_set_copy_image(local[r], send[(size_t)r].as<double>(), mx);  // returns after the copy
  }
  auto cleanup = [&] {
    for (DevMem& m : send) m.release();
    for (DevMem& m : recv) m.release();
  };
  if (rc != QK_OK) {
    cleanup();
    return rc;
  }
  qk_range_push("qk:allgather_sets");
  ncclResult_t nr = rccl().GroupStart();
  for (int r = 0; r < n && nr == ncclSuccess; ++r) nr = rccl().AllGather(send[(size_t)r].p, recv[(size_t)r].p, (size_t)mx, ncclDouble, c->nccl[(size_t)r], c->ctx[(size_t)r]->stream);
  const ncclResult_t ne = rccl().GroupEnd();
  if (nr == ncclSuccess) nr = ne;
  for (int r = 0; r < n; ++r) (void)qk_ctx_synchronize(c->ctx[(size_t)r]);
  qk_range_pop();
  if (nr != ncclSuccess) {
    cleanup();
    return qk_fail(QK_EDEVICE, "qk_mps_set_allgather: ncclAllGather failed: %s", rccl().GetErrorString(nr));
  }
  for (int r = 0; r < n && rc == QK_OK; ++r) rc = qk_mps_set_from_packed(c->ctx[(size_t)r], total, n_sites, dims.data(), offs.data(), recv[(size_t)r].as<double>(), (int64_t)n * mx, &full_out[r]);
  cleanup();
  if (rc != QK_OK)
    for (int r = 0; r < n; ++r) {
      qk_mps_set_destroy(full_out[r]);
      full_out[r] = nullptr;
    }
  return rc;
}

// ---- the sharded Gram: one sweep launch per device, ONE all-gather, a scatter per device ----------------------------
static int prepare_job(qk_comm* c, qk_mps_set* const* xs, qk_mps_set* const* ys) {
  const int n = c->n;
  for (int r = 0; r < n; ++r)
    if (!xs[r] || (ys && !ys[r])) return qk_fail(QK_EINVAL, "qk_gram_sharded: set %d is NULL (every device needs the whole set: qk_mps_set_allgather)", r);
  bool same = (int)c->job_x.size() == n;
  for (int r = 0; r < n && same; ++r)
    same = c->job_x[(size_t)r] == xs[r] && c->job_xid[(size_t)r] == xs[r]->uid && c->job_y[(size_t)r] == (ys ? ys[r] : nullptr) && c->job_yid[(size_t)r] == (ys ? ys[r]->uid : 0);
  if (same) return QK_OK;
  drop_job(c);
  const qk_mps_set* x0 = xs[0];
  const qk_mps_set* y0 = ys ? ys[0] : nullptr;
  for (int r = 0; r < n; ++r) {
    const qk_mps_set* x = xs[r];
    const qk_mps_set* y = ys ? ys[r] : nullptr;
    if (x->ctx != c->ctx[(size_t)r] || (y && y->ctx != c->ctx[(size_t)r])) return qk_fail(QK_EINVAL, "qk_gram_sharded: set %d does not live on the communicator's context %d", r, r);
    if (x->dims_true != x0->dims_true || (y && y->dims_true != y0->dims_true))
      return qk_fail(QK_EINVAL, "qk_gram_sharded: device %d holds a different set than device 0 (every device needs the whole set: qk_mps_set_allgather)", r);
  }
  const bool sym = y0 == nullptr;
  // the plans of all ranks from ONE cost pass (this is one process: pass 1 and the deal of the tiles are the same for every rank)
  c->plans.assign((size_t)n, nullptr);
  c->maxp = 1;
  {
    const int rc = qk_plan_create_all(x0->n_sites, x0->n_states, x0->dims_true.data(), sym ? x0->n_states : y0->n_states, sym ? nullptr : y0->dims_true.data(),
                                      sym ? (QK_PLAN_SYMMETRIC | QK_PLAN_ORIENT) : 0u, n, c->plans.data());
    if (rc != QK_OK) {
      drop_job(c);
      return rc;
    }
    for (int r = 0; r < n; ++r) c->maxp = std::max<int64_t>(c->maxp, qk_plan_max_pairs_per_rank(c->plans[(size_t)r]));
  }
  // the pair table of ALL ranks, padded with -1 (this is one process: no collective needed for it)
  std::vector<int32_t> all((size_t)n * c->maxp * 2, -1);
  for (int r = 0; r < n; ++r) {
    const int64_t np = qk_plan_num_pairs(c->plans[(size_t)r]);
    if (np) std::memcpy(&all[(size_t)r * c->maxp * 2], qk_plan_pairs(c->plans[(size_t)r]), (size_t)np * 2 * sizeof(int32_t));
  }
  const int nx = x0->n_states, ny = sym ? nx : y0->n_states;
  for (int r = 0; r < n; ++r) {
    const int d = c->devices[(size_t)r];
    int rc = c->vals[(size_t)r].ensure(d, (size_t)c->maxp * sizeof(double));
    if (rc == QK_OK) rc = c->all_vals[(size_t)r].ensure(d, (size_t)n * c->maxp * sizeof(double));
    if (rc == QK_OK) rc = c->all_pairs[(size_t)r].ensure(d, all.size() * sizeof(int32_t));
    if (rc == QK_OK) rc = c->k[(size_t)r].ensure(d, (size_t)nx * ny * sizeof(double));
    if (rc != QK_OK) {
      drop_job(c);
      return rc;
    }
    hipError_t e = hipSetDevice(d);
    if (e == hipSuccess) e = hipMemcpy(c->all_pairs[(size_t)r].p, all.data(), all.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(c->vals[(size_t)r].p, 0, (size_t)c->maxp * sizeof(double));
    if (e != hipSuccess) {
      drop_job(c);
      return qk_fail(QK_EDEVICE, "qk_gram_sharded: pair table of device %d: %s", d, hipGetErrorString(e));
    }
  }
  c->job_x.assign(xs, xs + n);
  c->job_y.assign((size_t)n, nullptr), c->job_xid.assign((size_t)n, 0), c->job_yid.assign((size_t)n, 0);
  if (ys) c->job_y.assign(ys, ys + n);
  for (int r = 0; r < n; ++r) c->job_xid[(size_t)r] = xs[r]->uid, c->job_yid[(size_t)r] = ys ? ys[r]->uid : 0;
  return QK_OK;
}

// every context idle again before an error leaves qk_gram_sharded (launches of the other devices may be in flight)
static int sharded_fail(qk_comm* c, const int rc) {
  for (int r = 0; r < c->n; ++r) {
    (void)hipSetDevice(c->devices[(size_t)r]);
    (void)hipStreamSynchronize(c->ctx[(size_t)r]->stream);
  }
  return rc;
}

extern "C" int qk_gram_sharded(qk_comm* c, qk_mps_set* const* xsets, qk_mps_set* const* ysets, double* out_host, int64_t ld) {
  if (!c || !xsets || !xsets[0]) return qk_fail(QK_EINVAL, "qk_gram_sharded: null argument");
  const int n = c->n;
  int rc = prepare_job(c, xsets, ysets);
  if (rc != QK_OK) return rc;
  const bool sym = ysets == nullptr;
  const int nx = xsets[0]->n_states, ny = sym ? nx : ysets[0]->n_states;
  if (out_host && ld < nx) return qk_fail(QK_EINVAL, "qk_gram_sharded: ld %lld < %d columns", (long long)ld, nx);
  auto hip_ok = [](const hipError_t e, const char* what) { return e == hipSuccess ? QK_OK : qk_fail(QK_EDEVICE, "qk_gram_sharded: %s: %s", what, hipGetErrorString(e)); };
  // 1. every device sweeps its share.  Everything here is asynchronous -- on a set's first Gram that includes the kernels that make its
  //    derived images (interleaved, edge blocks, merged steps) --, so all devices are at work before the host waits for any of them
  {
    QkRangeGuard range_("qk:sharded_sweep");
    for (int r = 0; r < n && rc == QK_OK; ++r) {
      rc = hip_ok(hipSetDevice(c->devices[(size_t)r]), "hipSetDevice");
      if (rc == QK_OK) rc = hip_ok(hipMemsetAsync(c->k[(size_t)r].p, 0, (size_t)nx * ny * sizeof(double), c->ctx[(size_t)r]->stream), "hipMemsetAsync");
      if (rc == QK_OK) rc = qk_gram_values(c->ctx[(size_t)r], xsets[r], sym ? nullptr : ysets[r], c->plans[(size_t)r], c->vals[(size_t)r].as<double>(), nullptr);
    }
  }
  if (rc != QK_OK) return sharded_fail(c, rc);
  // 2. the ONE collective of the path: all-gather of the packed values over xGMI
  {
    QkRangeGuard range_("qk:allgather_values");
    rc = hip_ok(hipSetDevice(c->devices[0]), "hipSetDevice");
    if (rc == QK_OK) rc = hip_ok(hipEventRecord(c->ev0, c->ctx[0]->stream), "hipEventRecord");
    if (rc != QK_OK) return sharded_fail(c, rc);
    ncclResult_t nr = rccl().GroupStart();
    for (int r = 0; r < n && nr == ncclSuccess; ++r)
      nr = rccl().AllGather(c->vals[(size_t)r].p, c->all_vals[(size_t)r].p, (size_t)c->maxp, ncclDouble, c->nccl[(size_t)r], c->ctx[(size_t)r]->stream);
    const ncclResult_t ne = rccl().GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return sharded_fail(c, qk_fail(QK_EDEVICE, "qk_gram_sharded: ncclAllGather failed: %s", rccl().GetErrorString(nr)));
    rc = hip_ok(hipSetDevice(c->devices[0]), "hipSetDevice");
    if (rc == QK_OK) rc = hip_ok(hipEventRecord(c->ev1, c->ctx[0]->stream), "hipEventRecord");
    if (rc != QK_OK) return sharded_fail(c, rc);
  }
  // 3. every device fills (and mirrors) its own dense K; rank 0's goes to the caller
  for (int r = 0; r < n && rc == QK_OK; ++r)
    rc = qk_scatter(c->ctx[(size_t)r], c->all_pairs[(size_t)r].as<int32_t>(), c->all_vals[(size_t)r].as<double>(), (int64_t)n * c->maxp, c->k[(size_t)r].as<double>(), nx, sym ? 1 : 0);
  if (rc == QK_OK && out_host) {
    rc = hip_ok(hipSetDevice(c->devices[0]), "hipSetDevice");
    if (rc == QK_OK)
      rc = hip_ok(hipMemcpy2DAsync(out_host, (size_t)ld * sizeof(double), c->k[0].p, (size_t)nx * sizeof(double), (size_t)nx * sizeof(double), (size_t)ny, hipMemcpyDeviceToHost, c->ctx[0]->stream),
                  "hipMemcpy2DAsync");
  }
  if (rc != QK_OK) return sharded_fail(c, rc);
  for (int r = 0; r < n; ++r) {
    const int rs = qk_ctx_synchronize(c->ctx[(size_t)r]);
    if (rs != QK_OK && rc == QK_OK) rc = rs;  // (keep waiting for the others)
  }
  if (rc != QK_OK) return rc;
  float ms = 0;
  HIP_TRY(hipSetDevice(c->devices[0]));
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->gather_ms = ms;  // on rank 0's stream: includes waiting for the slowest rank's sweep
  for (int r = 0; r < n; ++r) (void)qk_get_stats(c->ctx[(size_t)r], &c->last_stats[r]);
  return QK_OK;
}

extern "C" int qk_comm_device_gram(qk_comm* c, int32_t rank, const double** k_dev) {
  if (!c || rank < 0 || rank >= c->n || !k_dev) return qk_fail(QK_EINVAL, "qk_comm_device_gram: bad argument");
  *k_dev = c->k[(size_t)rank].as<double>();
  return QK_OK;
}

extern "C" int qk_comm_stats(const qk_comm* c, int32_t rank, qk_stats* out, double* allgather_ms) {
  if (!c || rank < 0 || rank >= c->n) return qk_fail(QK_EINVAL, "qk_comm_stats: bad argument");
  if (out) *out = c->last_stats[rank];
  if (allgather_ms) *allgather_ms = c->gather_ms;
  return QK_OK;
}
