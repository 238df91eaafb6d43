// sn_plan.hip — launch plans: one host call enqueues a whole residual block.
//
// A step of the reference's training loops (src/as_rigid_as_possible/main.py:217-232, src/mesh_mnist/main.py:151-167,
// src/dense_correspondence/main.py:310-327) is a few hundred kernel launches of 5-400 us issued one by one from Python; an
// unmodified driver cannot capture a hipGraph, so the host cost of one launch (ctypes marshalling, output allocation,
// Function bodies: ~22 us measured) bounds every configuration with small batches.  A plan is the recorded launch list of one
// block (forward or backward): every entry is one entry point of include/sn_spmm.h with its arguments, pointer arguments
// expressed as (slot, byte offset).  Slots are base addresses the caller supplies per run: the block's workspace arenas (one
// allocation each, sized once per shape) and the tensors it reads or writes (features, weights, operator arrays).
// sn_plan_run walks the list and calls the SAME launchers a caller would call one by one, in the same order, on the caller's
// stream: same kernels, same grids, bit-identical results.  A plan holds no addresses and nothing on the device (its graph form
// at fixed addresses is a separate object, below): it is host memory owned by the caller (sn_plan_create / sn_plan_destroy),
// immutable while it runs, re-entrant across streams.
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "sn_spmm.h"

// fills and copies as kernels of this library (sn_kernels.hip says why they are not hipMemsetAsync / hipMemcpy2DAsync)
hipError_t sn_internal_fill2d(void *dst, int64_t pitch, int value, int64_t width, int64_t rows, hipStream_t s);
hipError_t sn_internal_copy2d(void *dst, int64_t dpitch, const void *src, int64_t spitch, int64_t width, int64_t rows, hipStream_t s);

namespace {

union PlanArg {
  int64_t i;
  double d;
  void *p;
};

template <class T>
inline T conv(const PlanArg &a) {
  if constexpr (std::is_pointer<T>::value) return (T)a.p;
  else if constexpr (std::is_floating_point<T>::value) return (T)a.d;
  else return (T)a.i;
}
template <class T>
constexpr char kind_of() {
  return std::is_pointer<T>::value ? 'p' : std::is_floating_point<T>::value ? 'd' : 'i';
}
template <class... A, size_t... I>
inline int call_(int (*fn)(A...), const PlanArg *a, std::index_sequence<I...>) {
  return fn(conv<A>(a[I])...);
}
template <class... A>
inline int call(int (*fn)(A...), const PlanArg *a) {
  return call_(fn, a, std::index_sequence_for<A...>{});
}
template <class... A>
struct Sig {
  static const char *str() {
    static const char s[] = {kind_of<A>()..., 0};
    return s;
  }
};
template <class... A>
const char *sig_of(int (*)(A...)) {
  return Sig<A...>::str();
}

struct Entry {
  const char *name;
  int (*tramp)(const PlanArg *);
  const char *sig;       // one char per parameter: p pointer, i integer, d floating point; the last one is the stream
};

#define SN_PLAN_FN(f) {#f, [](const PlanArg *a) -> int { return call(&f, a); }, sig_of(&f)},
const Entry kTable[] = {
#include "sn_plan_table.inc"
};
#undef SN_PLAN_FN
constexpr int kEntries = (int)(sizeof(kTable) / sizeof(kTable[0]));

constexpr int kMaxArgs = 32;
enum ArgKind : int32_t { kInt = 0, kDouble = 1, kPtr = 2, kNull = 3, kStream = 4 };
enum NodeKind : int32_t { kCall = 0, kMemset2D = 1, kCopy2D = 2 };

struct Node {
  int32_t kind, fn, nargs;
  int32_t akind[kMaxArgs];
  int32_t slot[kMaxArgs];
  PlanArg val[kMaxArgs];
};

}  // namespace

struct sn_plan {
  std::vector<Node> nodes;
  int32_t max_slot = -1;
};

struct sn_plan_exec {
  hipGraphExec_t exec = nullptr;
};

bool sn_internal_timing_on();

static int walk(const sn_plan *p, const uint64_t *slot_base, void *stream, int32_t *failed_node) {
  hipStream_t s = (hipStream_t)stream;
  const int nn = (int)p->nodes.size();
  for (int k = 0; k < nn; ++k) {
    const Node &n = p->nodes[k];
    int st = SN_OK;
    if (n.kind == kCall) {
      PlanArg a[kMaxArgs];
      for (int i = 0; i < n.nargs; ++i) {
        switch (n.akind[i]) {
          case kPtr: {
            const uint64_t base = slot_base[n.slot[i]];
            if (!base) st = SN_E_NULL;                   // a slot the plan dereferences was handed over empty
            a[i].p = (void *)(uintptr_t)(base + (uint64_t)n.val[i].i);
            break;
          }
          case kNull: a[i].p = nullptr; break;
          case kStream: a[i].p = stream; break;
          default: a[i] = n.val[i]; break;
        }
      }
      if (st == SN_OK) st = kTable[n.fn].tramp(a);
    } else if (n.kind == kMemset2D) {
      const uint64_t base = slot_base[n.slot[0]];
      if (!base) st = SN_E_NULL;
      else if (n.val[3].i > 0 && n.val[4].i > 0) {
        void *dst = (void *)(uintptr_t)(base + (uint64_t)n.val[0].i);
        st = (int)sn_internal_fill2d(dst, n.val[2].i, (int)n.val[1].i, n.val[3].i, n.val[4].i, s);
      }
    } else {
      const uint64_t db = slot_base[n.slot[0]], sb = slot_base[n.slot[2]];
      if (!db || !sb) st = SN_E_NULL;
      else if (n.val[4].i > 0 && n.val[5].i > 0) {
        void *dst = (void *)(uintptr_t)(db + (uint64_t)n.val[0].i);
        const void *src = (const void *)(uintptr_t)(sb + (uint64_t)n.val[2].i);
        st = (int)sn_internal_copy2d(dst, n.val[1].i, src, n.val[3].i, n.val[4].i, n.val[5].i, s);
      }
    }
    if (st != SN_OK) {
      if (failed_node) *failed_node = k;
      return st;
    }
  }
  return SN_OK;
}

extern "C" {

int sn_plan_create(sn_plan **out) {
  if (!out) return SN_E_NULL;
  *out = new (std::nothrow) sn_plan();
  return *out ? SN_OK : SN_E_WORKSPACE;
}

int sn_plan_destroy(sn_plan *p) {
  delete p;
  return SN_OK;
}

int32_t sn_plan_lookup(const char *name) {
  if (!name) return -1;
  for (int i = 0; i < kEntries; ++i)
    if (!strcmp(kTable[i].name, name)) return i;
  return -1;
}

int32_t sn_plan_entry_count(void) { return kEntries; }

const char *sn_plan_entry_name(int32_t fn) { return (fn >= 0 && fn < kEntries) ? kTable[fn].name : nullptr; }

const char *sn_plan_entry_signature(int32_t fn) { return (fn >= 0 && fn < kEntries) ? kTable[fn].sig : nullptr; }

int64_t sn_plan_length(const sn_plan *p) { return p ? (int64_t)p->nodes.size() : -1; }

int sn_plan_add_call(sn_plan *p, int32_t fn, int32_t nargs, const int32_t *kind, const int32_t *slot, const int64_t *ival,
                     const double *dval) {
  if (!p || !kind || !slot || !ival || !dval) return SN_E_NULL;
  if (fn < 0 || fn >= kEntries || nargs < 1 || nargs > kMaxArgs) return SN_E_SHAPE;
  const char *sig = kTable[fn].sig;
  if ((int)strlen(sig) != nargs) return SN_E_SHAPE;
  Node n;
  memset(&n, 0, sizeof(n));
  n.kind = kCall;
  n.fn = fn;
  n.nargs = nargs;
  for (int i = 0; i < nargs; ++i) {
    const int32_t k = kind[i];
    // the recorded kind must be what the entry point's parameter is: an integer where a pointer belongs (or a double where an
    // integer does) would be reinterpreted silently
    const bool ok = (sig[i] == 'p' && (k == kPtr || k == kNull || (k == kStream && i == nargs - 1))) || (sig[i] == 'i' && k == kInt) ||
                    (sig[i] == 'd' && k == kDouble);
    if (!ok) return SN_E_UNSUPPORTED;
    n.akind[i] = k;
    n.slot[i] = slot[i];
    if (k == kDouble) n.val[i].d = dval[i];
    else n.val[i].i = ival[i];
    if (k == kPtr) {
      if (slot[i] < 0 || ival[i] < 0) return SN_E_RANGE;
      if (slot[i] > p->max_slot) p->max_slot = slot[i];
    }
  }
  if (n.akind[nargs - 1] != kStream) return SN_E_UNSUPPORTED;
  p->nodes.push_back(n);
  return SN_OK;
}

// value-fill of a 2-D region (rows x width_bytes, row pitch in bytes; rows = 1: a flat range) of one slot
int sn_plan_add_memset(sn_plan *p, int32_t slot, int64_t offset, int32_t byte_value, int64_t pitch, int64_t width_bytes, int64_t rows) {
  if (!p) return SN_E_NULL;
  if (slot < 0 || offset < 0 || width_bytes < 0 || rows < 0 || (rows > 1 && pitch < width_bytes)) return SN_E_SHAPE;
  Node n;
  memset(&n, 0, sizeof(n));
  n.kind = kMemset2D;
  n.nargs = 5;
  n.slot[0] = slot;
  n.val[0].i = offset;
  n.val[1].i = byte_value;
  n.val[2].i = pitch;
  n.val[3].i = width_bytes;
  n.val[4].i = rows;
  if (slot > p->max_slot) p->max_slot = slot;
  p->nodes.push_back(n);
  return SN_OK;
}

// device-to-device copy of a 2-D region between two slots
int sn_plan_add_copy(sn_plan *p, int32_t dst_slot, int64_t dst_offset, int64_t dst_pitch, int32_t src_slot, int64_t src_offset,
                     int64_t src_pitch, int64_t width_bytes, int64_t rows) {
  if (!p) return SN_E_NULL;
  if (dst_slot < 0 || src_slot < 0 || dst_offset < 0 || src_offset < 0 || width_bytes < 0 || rows < 0 ||
      (rows > 1 && (dst_pitch < width_bytes || src_pitch < width_bytes)))
    return SN_E_SHAPE;
  Node n;
  memset(&n, 0, sizeof(n));
  n.kind = kCopy2D;
  n.nargs = 7;
  n.slot[0] = dst_slot;
  n.val[0].i = dst_offset;
  n.val[1].i = dst_pitch;
  n.slot[2] = src_slot;
  n.val[2].i = src_offset;
  n.val[3].i = src_pitch;
  n.val[4].i = width_bytes;
  n.val[5].i = rows;
  if (dst_slot > p->max_slot) p->max_slot = dst_slot;
  if (src_slot > p->max_slot) p->max_slot = src_slot;
  p->nodes.push_back(n);
  return SN_OK;
}

int sn_plan_run(const sn_plan *p, const uint64_t *slot_base, int32_t nslots, void *stream, int32_t *failed_node) {
  if (failed_node) *failed_node = -1;
  if (!p || (!slot_base && p->max_slot >= 0)) return SN_E_NULL;
  if (nslots <= p->max_slot) return SN_E_SHAPE;
  return walk(p, slot_base, stream, failed_node);
}

// ---- a plan at FIXED addresses as one graph launch ------------------------------------------------------------------------------
// The torch caching allocator hands the same blocks to the same sequence of requests: in a training loop the slot addresses of a
// plan run repeat from step to step (tools/scratch/plan_addr_probe.py: every run after the first step, on all three drivers).  For
// such a run the launch list can be captured ONCE into a hipGraph — kernel nodes only: fills and copies are kernels of this library —
// and enqueued by one hipGraphLaunch: 9-25 us of host time instead of 42-57 for the 6-9 launches of a block direction
// (tools/scratch/plan_graph_probe.py).  Same kernels, same arguments, same order: bit-identical.  The executable graph is the CALLER's
// object like the plan (create / launch / destroy), holds no device memory, and is only valid for the addresses it was made for:
// the caller keys it by them.
int sn_plan_instantiate(const sn_plan *p, const uint64_t *slot_base, int32_t nslots, sn_plan_exec **out, int32_t *failed_node) {
  if (failed_node) *failed_node = -1;
  if (!p || !out || (!slot_base && p->max_slot >= 0)) return SN_E_NULL;
  *out = nullptr;
  if (nslots <= p->max_slot) return SN_E_SHAPE;
  if (p->nodes.empty()) return SN_E_SHAPE;
  if (sn_internal_timing_on()) return SN_E_UNSUPPORTED;        // (event records of the per-launch timer do not belong in a graph)
  (void)hipGetLastError();
  hipStream_t cs = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  if (e != hipSuccess) return (int)e;
  e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    (void)hipStreamDestroy(cs);
    return (int)e;
  }
  const int st = walk(p, slot_base, (void *)cs, failed_node);
  hipGraph_t graph = nullptr;
  e = hipStreamEndCapture(cs, &graph);
  (void)hipStreamDestroy(cs);
  if (st != SN_OK || e != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    return st != SN_OK ? st : (e != hipSuccess ? (int)e : SN_E_UNSUPPORTED);
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess || !exec) return e != hipSuccess ? (int)e : SN_E_UNSUPPORTED;
  sn_plan_exec *x = new (std::nothrow) sn_plan_exec();
  if (!x) {
    (void)hipGraphExecDestroy(exec);
    return SN_E_WORKSPACE;
  }
  x->exec = exec;
  *out = x;
  return SN_OK;
}

// Enqueue: the graph when `stream` is an ordinary stream; the recorded launches one by one (sn_plan_run on the same addresses)
// when the stream is itself being captured by the caller or the per-launch timer is on.
int sn_plan_exec_launch(const sn_plan_exec *x, const sn_plan *p, const uint64_t *slot_base, int32_t nslots, void *stream,
                        int32_t *failed_node) {
  if (failed_node) *failed_node = -1;
  if (!x || !x->exec) return SN_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) {
    (void)hipGetLastError();
    cap = hipStreamCaptureStatusActive;
  }
  if (cap != hipStreamCaptureStatusNone || sn_internal_timing_on()) return sn_plan_run(p, slot_base, nslots, stream, failed_node);
  return (int)hipGraphLaunch(x->exec, s);
}

int sn_plan_exec_destroy(sn_plan_exec *x) {
  if (x) {
    if (x->exec) (void)hipGraphExecDestroy(x->exec);
    delete x;
  }
  return SN_OK;
}

}  // extern "C"
