// extern "C" surface of libsimspread_hip.so (see include/simspread_hip.h for the contract and the
// reference methods each entry point stands behind).  Host-side orchestration only: argument
// checks, staging of caller buffers, the stage-1 / stage-2 launch sequence, event timing.
#include <cstdlib>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <type_traits>

#include "graph.hpp"

namespace ss {

// ------------------------------------------------------------------ errors / context
std::string& last_error() {
  static thread_local std::string msg;
  return msg;
}

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

Ctx& ctx() {
  static Ctx c;
  return c;
}

int require_init() {
  if (!ctx().inited) return fail(SS_ENODEV, "ss_init(device) has not been called (the HIP path has no CPU fallback)");
  return SS_OK;
}

// registry of the per-thread Timing objects (their hipEvent pools): ss_shutdown destroys every thread's events, not
// only the calling thread's; a thread that exits destroys its own.  The mutex orders the two.
static std::mutex& timing_reg_mu() {
  static std::mutex m;
  return m;
}
static std::vector<Timing*>& timing_reg() {
  static std::vector<Timing*> v;
  return v;
}
Timing::Timing() {
  std::lock_guard<std::mutex> lk(timing_reg_mu());
  timing_reg().push_back(this);
}
Timing::~Timing() {
  std::lock_guard<std::mutex> lk(timing_reg_mu());
  auto& v = timing_reg();
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] == this) {
      v[i] = v.back();
      v.pop_back();
      break;
    }
  if (ctx().inited && generation == ctx().generation)
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
}

Timing& timing() {
  static thread_local Timing t;
  if (t.generation != ctx().generation) {  // the context these events belonged to is gone
    t.pool.clear();
    t.spans.clear();
    t.used = 0;
    t.hold = false;
    t.generation = ctx().generation;
  }
  return t;
}

std::string& path_note() {
  static thread_local std::string s;
  return s;
}
void path_add(const char* tag) {
  std::string& s = path_note();
  if (s.find(tag) != std::string::npos) return;
  if (!s.empty()) s += ",";
  s += tag;
}

void timing_begin_call() {
  Timing& t = timing();
  if (!t.hold) path_note().clear();
  t.dirty = true;
  if (t.hold) return;  // accumulate: the spans of this call join those already recorded
  t.used = 0;
  t.spans.clear();
  for (double& e : t.extra) e = 0;
  t.dirty = true;
}

int timing_mark(hipEvent_t* ev) {
  Timing& t = timing();
  if (t.used == t.pool.size()) {
    hipEvent_t e;
    SS_HIP(hipEventCreate(&e));
    t.pool.push_back(e);
  }
  *ev = t.pool[t.used++];
  SS_HIP(hipEventRecord(*ev, ctx().stream));
  return SS_OK;
}

void timing_span(int stage, hipEvent_t a, hipEvent_t b) { timing().spans.push_back({stage, a, b}); }
void timing_count(int stage, double inc) { timing().extra[stage] += inc; }

static int check_mem(int mem) {
  if (mem != SS_MEM_HOST && mem != SS_MEM_DEVICE) return fail(SS_EINVAL, "mem must be SS_MEM_HOST or SS_MEM_DEVICE");
  return SS_OK;
}
static int check_layout(int layout) {
  if (layout != SS_LAYOUT_ROWMAJOR && layout != SS_LAYOUT_COLMAJOR) return fail(SS_EINVAL, "unknown layout");
  return SS_OK;
}

// RAII marks: a span of one stage on the stream
struct StageTimer {
  int stage;
  hipEvent_t a = nullptr;
  bool ok = false;
  explicit StageTimer(int s) : stage(s) { ok = timing_mark(&a) == SS_OK; }
  void stop() {
    hipEvent_t b;
    if (ok && timing_mark(&b) == SS_OK) timing_span(stage, a, b);
    ok = false;
  }
  ~StageTimer() { stop(); }
};

// ------------------------------------------------------------------ element-wise entry points
template <class T>
static int cutoff_impl(const T* X, int64_t rows, int64_t cols, int64_t ld, T alpha, int weighted, T* out,
                       int64_t ldo, int mem) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  if (rows < 0 || cols < 0 || ld < rows || ldo < rows) return fail(SS_EINVAL, "cutoff: bad shape / leading dimension");
  if (rows * cols == 0) return SS_OK;
  if (!X || !out) return fail(SS_EINVAL, "cutoff: NULL buffer");
  hipStream_t st = ctx().stream;
  if (mem == SS_MEM_DEVICE) {
    SS_TRY(launch_cutoff<T>(X, rows, cols, ld, alpha, weighted != 0, out, ldo));
    return SS_OK;
  }
  DevBuf<T> din, dout;
  SS_TRY(din.alloc((size_t)rows * cols));
  SS_TRY(dout.alloc((size_t)rows * cols));
  SS_HIP(hipMemcpy2DAsync(din.p, rows * sizeof(T), X, ld * sizeof(T), rows * sizeof(T), cols, hipMemcpyHostToDevice, st));
  SS_TRY(launch_cutoff<T>(din.p, rows, cols, rows, alpha, weighted != 0, dout.p, rows));
  SS_HIP(hipMemcpy2DAsync(out, ldo * sizeof(T), dout.p, rows * sizeof(T), rows * sizeof(T), cols, hipMemcpyDeviceToHost, st));
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

template <class T>
static int row_degree_impl(const T* G, int64_t rows, int64_t cols, int64_t ld, int64_t* deg, int mem) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  if (rows < 0 || cols < 0 || ld < rows) return fail(SS_EINVAL, "k: bad shape / leading dimension");
  if (rows == 0) return SS_OK;
  if (!deg || (cols > 0 && !G)) return fail(SS_EINVAL, "k: NULL buffer");
  hipStream_t st = ctx().stream;
  DevBuf<T> din;
  const T* src = G;
  int64_t sld = ld;
  if (mem == SS_MEM_HOST && cols > 0) {
    SS_TRY(din.alloc((size_t)rows * cols));
    SS_HIP(hipMemcpy2DAsync(din.p, rows * sizeof(T), G, ld * sizeof(T), rows * sizeof(T), cols, hipMemcpyHostToDevice, st));
    src = din.p;
    sld = rows;
  }
  DevBuf<int> d;
  SS_TRY(d.alloc(rows));
  SS_TRY(launch_row_degree<T>(src, rows, cols, sld, d.p));
  std::vector<int> h(rows);
  SS_HIP(hipMemcpyAsync(h.data(), d.p, rows * sizeof(int), hipMemcpyDeviceToHost, st));
  SS_HIP(hipStreamSynchronize(st));
  if (mem == SS_MEM_HOST) {
    for (int64_t i = 0; i < rows; ++i) deg[i] = h[i];
  } else {
    std::vector<int64_t> h64(h.begin(), h.end());
    SS_HIP(hipMemcpy(deg, h64.data(), rows * sizeof(int64_t), hipMemcpyHostToDevice));
  }
  return SS_OK;
}

template <class T>
static int spread_impl(const T* G, int64_t rows, int64_t cols, int64_t ld, T* W, int64_t ldw, int mem) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  if (rows < 0 || cols < 0 || ld < rows || ldw < rows) return fail(SS_EINVAL, "spread: bad shape / leading dimension");
  if (rows * cols == 0) return SS_OK;
  if (!G || !W) return fail(SS_EINVAL, "spread: NULL buffer");
  hipStream_t st = ctx().stream;
  DevBuf<T> din, dout;
  DevBuf<int> deg;
  SS_TRY(deg.alloc(rows));
  const T* src = G;
  T* dst = W;
  int64_t sld = ld, dld = ldw;
  if (mem == SS_MEM_HOST) {
    SS_TRY(din.alloc((size_t)rows * cols));
    SS_TRY(dout.alloc((size_t)rows * cols));
    SS_HIP(hipMemcpy2DAsync(din.p, rows * sizeof(T), G, ld * sizeof(T), rows * sizeof(T), cols, hipMemcpyHostToDevice, st));
    src = din.p; dst = dout.p; sld = rows; dld = rows;
  }
  SS_TRY(launch_row_degree<T>(src, rows, cols, sld, deg.p));
  SS_TRY(launch_spread_dense<T>(src, rows, cols, sld, deg.p, dst, dld));
  if (mem == SS_MEM_HOST)
    SS_HIP(hipMemcpy2DAsync(W, ldw * sizeof(T), dout.p, rows * sizeof(T), rows * sizeof(T), cols, hipMemcpyDeviceToHost, st));
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

// ------------------------------------------------------------------ graph handles
// Every handle starts with its precision tag and its own lock: an entry point holds the handle's lock while it
// works on it (operands are cut lazily, workspaces live in the handle), so calls on DIFFERENT handles overlap
// and calls on the same handle queue up.
struct HandleHead {
  int dtype;  // 4 or 8 = sizeof(T), guards against mixing _f32/_f64 entry points
  std::mutex mu;
};
template <class T>
struct GraphBox : HandleHead {
  Graph<T> g;
};
template <class T>
struct SpMatBox : HandleHead {
  SpMat<T> m;
};

template <class T>
static int graph_check(const void* h, Graph<T>** out) {
  if (!h) return fail(SS_EINVAL, "graph handle is NULL");
  GraphBox<T>* b = const_cast<GraphBox<T>*>(reinterpret_cast<const GraphBox<T>*>(h));
  if (b->dtype != (int)sizeof(T)) return fail(SS_EINVAL, "graph handle was created with the other precision");
  *out = &b->g;
  return SS_OK;
}

template <class T>
static int graph_create_csr_impl(int64_t nq, int64_t ns, int64_t nf, int64_t nt, const int64_t* xq_ptr,
                                 const int32_t* xq_idx, const T* xq_val, const int64_t* xs_ptr, const int32_t* xs_idx,
                                 const T* xs_val, const int64_t* ys_ptr, const int32_t* ys_idx, const T* ys_val,
                                 int index_base, int mem, ss_graph** out) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  if (!out) return fail(SS_EINVAL, "out handle pointer is NULL");
  *out = nullptr;
  if (nq < 0 || ns < 0 || nf < 0 || nt < 0) return fail(SS_EINVAL, "negative node count");
  GraphBox<T>* box = new (std::nothrow) GraphBox<T>();
  if (!box) return fail(SS_ENOMEM, "host allocation failed");
  box->dtype = (int)sizeof(T);
  Graph<T>& g = box->g;
  g.nq = nq; g.ns = ns; g.nf = nf; g.nt = nt;
  int rc = csr_from_user<T>(nq, nf, xq_ptr, xq_idx, xq_val, index_base, mem, g.Xq);
  if (rc == SS_OK) rc = csr_from_user<T>(ns, nf, xs_ptr, xs_idx, xs_val, index_base, mem, g.Xs);
  if (rc == SS_OK) rc = csr_from_user<T>(ns, nt, ys_ptr, ys_idx, ys_val, index_base, mem, g.Ys);
  if (rc == SS_OK) rc = graph_finalize<T>(g);
  if (rc != SS_OK) { delete box; return rc; }
  *out = reinterpret_cast<ss_graph*>(box);
  return SS_OK;
}

template <class T>
static int graph_create_dense_impl(int64_t nq, int64_t ns, int64_t nf, int64_t nt, const T* Sq, int64_t ldq,
                                   const T* Ss, int64_t lds, const T* Y, int64_t ldy, int apply_cutoff, T alpha,
                                   int weighted, int mem, ss_graph** out) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  if (!out) return fail(SS_EINVAL, "out handle pointer is NULL");
  *out = nullptr;
  if (nq < 0 || ns < 0 || nf < 0 || nt < 0) return fail(SS_EINVAL, "negative node count");
  GraphBox<T>* box = new (std::nothrow) GraphBox<T>();
  if (!box) return fail(SS_ENOMEM, "host allocation failed");
  box->dtype = (int)sizeof(T);
  Graph<T>& g = box->g;
  g.nq = nq; g.ns = ns; g.nf = nf; g.nt = nt;
  int rc = csr_from_dense<T>(Sq, nq, nf, ldq, apply_cutoff != 0, alpha, weighted != 0, mem, g.Xq);
  if (rc == SS_OK) rc = csr_from_dense<T>(Ss, ns, nf, lds, apply_cutoff != 0, alpha, weighted != 0, mem, g.Xs);
  if (rc == SS_OK) rc = csr_from_dense<T>(Y, ns, nt, ldy, false, T(0), true, mem, g.Ys);
  if (rc == SS_OK) rc = graph_finalize<T>(g);
  if (rc != SS_OK) { delete box; return rc; }
  *out = reinterpret_cast<ss_graph*>(box);
  return SS_OK;
}

template <class T>
static int graph_create_general_impl(int64_t n, int64_t nr, int64_t nc, const int64_t* l_ptr, const int32_t* l_idx,
                                     const T* l_val, const int64_t* b_ptr, const int32_t* b_idx, const T* b_val,
                                     const int64_t* w_ptr, const int32_t* w_idx, const T* w_val, int index_base,
                                     int mem, ss_graph** out) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  if (!out) return fail(SS_EINVAL, "out handle pointer is NULL");
  *out = nullptr;
  if (n < 0 || nr < 0 || nc < 0) return fail(SS_EINVAL, "negative node count");
  GraphBox<T>* box = new (std::nothrow) GraphBox<T>();
  if (!box) return fail(SS_ENOMEM, "host allocation failed");
  box->dtype = (int)sizeof(T);
  Graph<T>& g = box->g;
  g.general = true;
  g.nq = nr; g.ns = n; g.nf = n; g.nt = nc;
  int rc = csr_from_user<T>(nr, n, l_ptr, l_idx, l_val, index_base, mem, g.Xq);
  if (rc == SS_OK) rc = csr_from_user<T>(n, n, b_ptr, b_idx, b_val, index_base, mem, g.XsT);
  if (rc == SS_OK) rc = csr_from_user<T>(nc, n, w_ptr, w_idx, w_val, index_base, mem, g.YsT);
  if (rc == SS_OK) rc = graph_finalize_general<T>(g);
  if (rc != SS_OK) { delete box; return rc; }
  *out = reinterpret_cast<ss_graph*>(box);
  return SS_OK;
}

// dense-similarity graph: raw similarities resident, labels sparse
template <class T>
static int graph_create_similarity_impl(int64_t nq, int64_t ns, int64_t nt, const T* Sq, int64_t ldq,
                                        const T* Ss, int64_t lds, const int64_t* y_ptr, const int32_t* y_idx,
                                        const T* y_val, int index_base, T alpha, int weighted, int mem,
                                        ss_graph** out) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  if (!out) return fail(SS_EINVAL, "out handle pointer is NULL");
  *out = nullptr;
  if (nq < 0 || ns < 0 || nt < 0) return fail(SS_EINVAL, "negative node count");
  if ((nq > 0 && (!Sq || ldq < nq)) || (ns > 0 && (!Ss || lds < ns)))
    return fail(SS_EINVAL, "similarity block: NULL pointer or ld < rows");
  GraphBox<T>* box = new (std::nothrow) GraphBox<T>();
  if (!box) return fail(SS_ENOMEM, "host allocation failed");
  box->dtype = (int)sizeof(T);
  Graph<T>& g = box->g;
  g.nq = nq; g.ns = ns; g.nf = ns; g.nt = nt;
  DenseSim<T>& d = g.dense;
  d.on = true; d.nq = nq; d.ns = ns; d.nf = ns; d.alpha = alpha; d.weighted = weighted != 0;
  hipStream_t st = ctx().stream;
  auto stage = [&](const T* src, int64_t rows, int64_t ld, DevBuf<T>& dst) -> int {
    SS_TRY(dst.alloc((size_t)rows * (size_t)ns));
    if (rows == 0 || ns == 0) return SS_OK;
    SS_HIP(hipMemcpy2DAsync(dst.p, rows * sizeof(T), src, ld * sizeof(T), rows * sizeof(T), ns,
                            mem == SS_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    return SS_OK;
  };
  int rc = stage(Sq, nq, ldq, d.Sq);
  if (rc == SS_OK) rc = stage(Ss, ns, lds, d.Ss);
  if (rc == SS_OK) rc = csr_from_user<T>(ns, nt, y_ptr, y_idx, y_val, index_base, mem, g.Ys);
  if (rc == SS_OK) rc = csr_transpose(g.Ys, g.YsT);
  if (rc == SS_OK) rc = graph_finalize_general_targets(g);
  if (rc == SS_OK) rc = dense_degrees(g);
  if (rc != SS_OK) { delete box; return rc; }
  *out = reinterpret_cast<ss_graph*>(box);
  return SS_OK;
}

// stage-2 operand of a graph: W = Ys' cut for the tile width of this precision
template <class T>
static int graph_sell(Graph<T>& g) {
  const int qt = sell_tile_width<T>();
  if (g.W_qt == qt) return SS_OK;
  int kcmax = sell_max_chunk<T>(qt);
  if (const char* e = getenv("SS_SELL_CHUNK")) {
    const int v = atoi(e);
    if (v >= 64 && v < kcmax) kcmax = v;
  }
  SS_TRY(sell_build<T>(g.YsT, kcmax, g.W));
  g.W_qt = qt;
  return SS_OK;
}

// stage-1 operands: X' (and Y' for source rows) cut into column chunks.  The chunk is sized so that an
// average sub-row fills one 64-lane load (SC ~ 64 * ns / mean row length), the chunk count is a multiple
// of 8 so that chunk c always meets the same XCD's L2.
template <class T>
static int graph_chunked(Graph<T>& g, bool need_y) {
  if (g.XsTc.SC == 0) {
    const int64_t ns = g.ns > 0 ? g.ns : 1;
    const double mean_len = g.XsT.rows > 0 ? (double)g.XsT.nnz / (double)g.XsT.rows : 0.0;
    int64_t sc = mean_len > 1.0 ? (int64_t)(64.0 * (double)ns / mean_len) : ns;
    // keep >= 8 single-wave workgroups per CU: (SC + 64) * sizeof(T) <= 20 KiB (measured at 100k x 100k, 1 %:
    // SC 6250 -> 7.7 ms, 4167 -> 4.5 ms, 3200 -> 5.9 ms for 2048 folds)
    const int64_t sc_max = (20 * 1024) / (int64_t)sizeof(T) - 64;
    if (sc > sc_max) sc = sc_max;
    if (sc < 256) sc = 256;
    if (const char* e = getenv("SS_TRANSFER_CHUNK")) {
      const long long v = atoll(e);
      if (v >= 16 && v <= 8192) sc = v;
    }
    int64_t nch = ceil_div(ns, sc);
    if (nch > 1) nch = ceil_div(nch, 8) * 8;
    sc = ceil_div(ceil_div(ns, nch), 4) * 4;
    SS_TRY(chunked_build<T>(g.XsT, (int)sc, 1, g.XsTc));
  }
  if (need_y && g.YsTc.SC == 0) SS_TRY(chunked_build<T>(g.YsT, g.XsTc.SC, 1, g.YsTc));
  return SS_OK;
}

// rows of T held at once (stage-1 output, stage-2 input)
static int64_t transfer_batch_rows(int64_t nrows, int64_t nj, size_t elem) {
  int64_t cap_bytes = 2LL << 30;
  if (const char* e = getenv("SS_TRANSFER_BYTES")) {
    const long long v = atoll(e);
    if (v >= (1 << 20)) cap_bytes = v;
  }
  int64_t rb = cap_bytes / ((nj > 0 ? nj : 1) * (int64_t)elem);
  rb &= ~7LL;
  if (rb < 8) rb = 8;
  return rb < nrows ? rb : nrows;
}

// run stage 1 + stage 2 over [row_begin, row_end) into dev_out (row-major nrows x nt, ld = ldo)
template <class T>
static int predict_rows_device(Graph<T>& g, int kind, int64_t row_begin, int64_t row_end, int clean, T* dev_out,
                               int64_t ldo) {
  const int64_t nrows = row_end - row_begin;
  const int64_t nj = g.ns;
  SS_TRY(graph_sell(g));
  if (!g.dense.on) SS_TRY(graph_chunked(g, kind == SS_ROWS_SOURCE));
  if (g.dense.on && kind == SS_ROWS_SOURCE && g.YsTc.SC == 0) {
    // the sparse target path needs Ys' cut into column chunks (same sizing rule as graph_chunked)
    const int64_t ns = g.ns > 0 ? g.ns : 1;
    const double mean_len = g.YsT.rows > 0 ? (double)g.YsT.nnz / (double)g.YsT.rows : 0.0;
    int64_t sc = mean_len > 1.0 ? (int64_t)(64.0 * (double)ns / mean_len) : ns;
    const int64_t sc_max = (20 * 1024) / (int64_t)sizeof(T) - 64;
    if (sc > sc_max) sc = sc_max;
    if (sc < 256) sc = 256;
    int64_t nch = ceil_div(ns, sc);
    if (nch > 1) nch = ceil_div(nch, 8) * 8;
    sc = ceil_div(ceil_div(ns, nch), 4) * 4;
    SS_TRY(chunked_build<T>(g.YsT, (int)sc, 1, g.YsTc));
  }
  const int64_t rb = transfer_batch_rows(nrows, nj, sizeof(T));
  // the transfer block lives in the handle so that repeated predictions do not re-allocate
  const size_t need = (size_t)rb * (size_t)(nj > 0 ? nj : 1);
  if (g.Tws.n < need) SS_TRY(g.Tws.alloc(need));
  DevBuf<T>& Tbuf = g.Tws;
  for (int64_t r0 = 0; r0 < nrows; r0 += rb) {
    const int64_t nb = (nrows - r0 < rb) ? (nrows - r0) : rb;
    {
      StageTimer t1(ST_TRANSFER);
      if (g.dense.on) {
        {
          const bool loo = (kind == 2);
          const bool srcrows = (kind == SS_ROWS_SOURCE);  // feature path here, target path added below
          if constexpr (std::is_same<T, float>::value) {
            // default: bf16 matrix cores on exact bf16 planes of the operands (dense_bf16.hip: 1.5x weighted, 3.2x
            // unweighted at 50k); SS_DENSE_BF16=0: the fp32-input MFMA kernel of dense.hip
            const bool use_bf16 = !(getenv("SS_DENSE_BF16") && atoi(getenv("SS_DENSE_BF16")) == 0);
            if (use_bf16)
              SS_TRY(launch_transfer_dense_bf16(g.dense, loo, loo ? g.dense.inv_kf_m1.p : g.inv_kf.p, g.inv_ks.p, g.ks.p,
                                                row_begin + r0, nb, Tbuf.p, nj, srcrows));
            else
              SS_TRY(launch_transfer_dense(g.dense, loo, loo ? g.dense.inv_kf_m1.p : g.inv_kf.p, g.inv_ks.p, g.ks.p,
                                           row_begin + r0, nb, Tbuf.p, nj, srcrows));
          } else {
            // fp64 (the reference's default precision): the fp64 matrix instruction, dense_f64.hip
            SS_TRY(launch_transfer_dense_f64(g.dense, loo, loo ? g.dense.inv_kf_m1.p : g.inv_kf.p, g.inv_ks.p, g.ks.p,
                                             row_begin + r0, nb, Tbuf.p, nj, srcrows));
          }
          if (srcrows) {
            // target path (Ys D_t^-1) Ys' D_s^-1 of the source rows: sparse, added to T
            const DevCsr<T>* L[2] = {&g.Ys, nullptr};
            const DevChunked<T>* M[2] = {&g.YsTc, nullptr};
            const T* inv1[2] = {g.inv_kt.p, nullptr};
            SS_TRY(launch_transfer<T>(1, L, inv1, M, g.inv_ks.p, row_begin + r0, nb, nj, Tbuf.p, nj, nullptr, true));
          }
        }
      } else if (kind == 2) {
        SS_TRY(launch_transfer_loo<T>(g.Xs, g.XsTc, g.kf.p, g.ks.p, row_begin + r0, nb, Tbuf.p, nj));
      } else if (kind == SS_ROWS_QUERY) {
        // query rows: SS_TRANSFER_V=2 selects the query-block kernels (measured variants of round 3)
        // when the chunk's sub-row offsets fit in LDS next to >= 8 waves of accumulators; default: the single-wave kernel
        const char* ev = getenv("SS_TRANSFER_V");
        const bool want_block = (ev && atoi(ev) >= 2) && transfer_block_fits<T>(g.XsT.rows, g.XsTc.SC);
        if (want_block) {
          if (g.XsTb.SC == 0) {
            int al = 32;
            if (const char* e = getenv("SS_TRANSFER_ALIGN")) al = atoi(e) == 1 ? 1 : 32;
            SS_TRY(chunked_build<T>(g.XsT, g.XsTc.SC, al, g.XsTb));
          }
          if (g.Cq.n < (size_t)(g.Xq.nnz > 0 ? g.Xq.nnz : 1)) SS_TRY(g.Cq.alloc((size_t)(g.Xq.nnz > 0 ? g.Xq.nnz : 1)));
          const bool fixed = !(getenv("SS_TRANSFER_FIX") && atoi(getenv("SS_TRANSFER_FIX")) == 0);
          SS_TRY(launch_transfer_block<T>(g.Xq, g.inv_kf.p, g.XsTb, g.inv_ks.p, row_begin + r0, nb, nj, Tbuf.p, nj, g.Cq.p,
                                          g.XsTb.vmax, fixed));
        } else {
          const DevCsr<T>* L[2] = {&g.Xq, nullptr};
          const DevChunked<T>* M[2] = {&g.XsTc, nullptr};
          const T* inv1[2] = {g.inv_kf.p, nullptr};
          SS_TRY(launch_transfer<T>(1, L, inv1, M, g.inv_ks.p, row_begin + r0, nb, nj, Tbuf.p, nj));
        }
      } else {
        // source rows: feature path + target path
        const DevCsr<T>* L[2] = {&g.Xs, &g.Ys};
        const DevChunked<T>* M[2] = {&g.XsTc, &g.YsTc};
        const T* inv1[2] = {g.inv_kf.p, g.inv_kt.p};
        SS_TRY(launch_transfer<T>(2, L, inv1, M, g.inv_ks.p, row_begin + r0, nb, nj, Tbuf.p, nj));
      }
      timing_count(ST_NTRANSFER, 1);
    }
    if (!g.W.sorted) {
      StageTimer t2(ST_SPMM);
      SS_TRY(launch_spmm_sell<T>(g.W, Tbuf.p, nj, nb, dev_out + r0 * ldo, ldo, clean ? g.kt.p : nullptr));
      timing_count(ST_NSPMM, 1);
    } else {
      // skew-sorted operand: scores come out in sorted target order, then go back through inv[] (+ clean!)
      const size_t need_s = (size_t)rb * (size_t)g.W.vrows;
      if (g.Sws.n < need_s) SS_TRY(g.Sws.alloc(need_s));
      {
        StageTimer t2(ST_SPMM);
        SS_TRY(launch_spmm_sell<T>(g.W, Tbuf.p, nj, nb, g.Sws.p, g.W.vrows, nullptr));
        timing_count(ST_NSPMM, 1);
      }
      StageTimer t3(ST_EPILOGUE);
      SS_TRY(launch_unpermute<T>(g.Sws.p, g.W.vrows, nb, g.nt, g.W.vfirst.p, g.W.inv.p, clean ? g.kt.p : nullptr,
                                 dev_out + r0 * ldo, ldo));
    }
  }
  if (kind == 2 && clean) {
    StageTimer t3(ST_EPILOGUE);
    SS_TRY(launch_loo_clean_fix<T>(g.YsT, g.kt.p, row_begin, nrows, dev_out, ldo));
  }
  return SS_OK;
}

// kind: SS_ROWS_QUERY, SS_ROWS_SOURCE, or 2 = leave-one-out
template <class T>
static int predict_impl(ss_graph* h, int kind, int64_t row_begin, int64_t row_end, int clean, T* out, int64_t ld,
                        int layout, int mem) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  SS_TRY(check_layout(layout));
  Graph<T>* gp = nullptr;
  SS_TRY(graph_check<T>(h, &gp));
  Graph<T>& g = *gp;
  const int64_t limit = (kind == SS_ROWS_QUERY) ? g.nq : g.ns;
  if (g.general && kind != SS_ROWS_QUERY)
    return fail(SS_EINVAL, "a general graph serves SS_ROWS_QUERY only");
  if (kind == 2) {
    if ((!g.dense.on && g.nq != 0) || g.ns != g.nf)
      return fail(SS_EINVAL, "leave-one-out needs a graph with nq == 0 and ns == nf (feature j named after source j)");
  } else if (kind != SS_ROWS_QUERY && kind != SS_ROWS_SOURCE) {
    return fail(SS_EINVAL, "rows_kind must be SS_ROWS_QUERY or SS_ROWS_SOURCE");
  }
  if (row_begin < 0 || row_end < row_begin || row_end > limit)
    return fail(SS_EINVAL, "row range [%lld,%lld) outside 0..%lld", (long long)row_begin, (long long)row_end,
                (long long)limit);
  const int64_t nrows = row_end - row_begin;
  const int64_t nt = g.nt;
  if (nrows == 0 || nt == 0) return SS_OK;
  if (!out) return fail(SS_EINVAL, "output buffer is NULL");
  const int64_t need_ld = (layout == SS_LAYOUT_ROWMAJOR) ? nt : nrows;
  if (ld < need_ld) return fail(SS_EINVAL, "leading dimension %lld < %lld", (long long)ld, (long long)need_ld);
  hipStream_t st = ctx().stream;
  timing_begin_call();
  hipEvent_t e_begin, e_end;
  SS_TRY(timing_mark(&e_begin));

  const bool direct = (mem == SS_MEM_DEVICE && layout == SS_LAYOUT_ROWMAJOR);
  DevBuf<T> scores;  // row-major nrows x nt
  T* dev_rm = out;
  int64_t ld_rm = ld;
  if (!direct) {
    SS_TRY(scores.alloc((size_t)nrows * nt));
    dev_rm = scores.p;
    ld_rm = nt;
  }
  SS_TRY(predict_rows_device<T>(g, kind, row_begin, row_end, clean, dev_rm, ld_rm));

  DevBuf<T> cm;  // column-major staging when the caller is on the host
  if (layout == SS_LAYOUT_COLMAJOR) {
    StageTimer t3(ST_EPILOGUE);
    T* dst = out;
    int64_t dld = ld;
    if (mem == SS_MEM_HOST) {
      SS_TRY(cm.alloc((size_t)nrows * nt));
      dst = cm.p;
      dld = nrows;
    }
    SS_TRY(launch_transpose<T>(dev_rm, nrows, nt, ld_rm, dst, dld));
  }
  SS_TRY(timing_mark(&e_end));
  timing_span(ST_TOTAL, e_begin, e_end);
  if (mem == SS_MEM_HOST) {
    StageTimer t5(ST_D2H);
    if (layout == SS_LAYOUT_ROWMAJOR) {
      SS_HIP(hipMemcpy2DAsync(out, ld * sizeof(T), dev_rm, ld_rm * sizeof(T), nt * sizeof(T), nrows,
                              hipMemcpyDeviceToHost, st));
    } else {
      SS_HIP(hipMemcpy2DAsync(out, ld * sizeof(T), cm.p, nrows * sizeof(T), nrows * sizeof(T), nt,
                              hipMemcpyDeviceToHost, st));
    }
    t5.stop();
  }
  // Staging buffers are released on return and host results must be complete: wait for the stream.  With the
  // scores written straight into the caller's device buffer there is nothing to release (the transfer block
  // lives in the handle), so the call returns as soon as the work is enqueued -- stream order, like a kernel
  // launch; ss_synchronize() or the caller's own stream synchronisation waits for it.
  if (!direct) SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

// k-fold: all folds of construct(y, X, members) + predict (+ clean!) from the resident graph
template <class T>
static int predict_kfold_impl(ss_graph* h, const int32_t* fold_of_source, int nfolds, int clean, T* out, int64_t ld,
                              int layout, int mem) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  SS_TRY(check_layout(layout));
  Graph<T>* gp = nullptr;
  SS_TRY(graph_check<T>(h, &gp));
  Graph<T>& g = *gp;
  if (g.general || (!g.dense.on && g.nq != 0) || g.ns != g.nf)
    return fail(SS_EINVAL, "k-fold needs a graph with nq == 0 and ns == nf (feature j named after source j)");
  if (nfolds < 1 || !fold_of_source) return fail(SS_EINVAL, "k-fold: bad fold assignment");
  const int64_t ns = g.ns, nt = g.nt;
  if (ns == 0 || nt == 0) return SS_OK;
  if (!out) return fail(SS_EINVAL, "output buffer is NULL");
  const int64_t need_ld = (layout == SS_LAYOUT_ROWMAJOR) ? nt : ns;
  if (ld < need_ld) return fail(SS_EINVAL, "leading dimension %lld < %lld", (long long)ld, (long long)need_ld);
  hipStream_t st = ctx().stream;
  // members of each fold, in source order (stable counting sort on the host)
  std::vector<int32_t> fold(ns);
  if (mem == SS_MEM_HOST) memcpy(fold.data(), fold_of_source, ns * sizeof(int32_t));
  else SS_HIP(hipMemcpy(fold.data(), fold_of_source, ns * sizeof(int32_t), hipMemcpyDeviceToHost));
  std::vector<int> start(nfolds + 1, 0), order(ns);
  for (int64_t i = 0; i < ns; ++i) {
    if (fold[i] < 0 || fold[i] >= nfolds) return fail(SS_EINVAL, "fold_of_source[%lld] = %d outside 0..%d", (long long)i, fold[i], nfolds - 1);
    start[fold[i] + 1]++;
  }
  for (int f = 0; f < nfolds; ++f) start[f + 1] += start[f];
  {
    std::vector<int> cur(start.begin(), start.end() - 1);
    for (int64_t i = 0; i < ns; ++i) order[cur[fold[i]]++] = (int)i;
  }
  DevBuf<int> d_fold, d_order, kf, ks, kt;
  DevBuf<T> inv_kf, inv_ks;
  SS_TRY(d_fold.alloc(ns)); SS_TRY(d_order.alloc(ns));
  SS_TRY(kf.alloc(g.nf)); SS_TRY(ks.alloc(ns)); SS_TRY(kt.alloc(nt));
  SS_TRY(inv_kf.alloc(g.nf)); SS_TRY(inv_ks.alloc(ns));
  SS_HIP(hipMemcpyAsync(d_fold.p, fold.data(), ns * sizeof(int), hipMemcpyHostToDevice, st));
  SS_HIP(hipMemcpyAsync(d_order.p, order.data(), ns * sizeof(int), hipMemcpyHostToDevice, st));
  SS_TRY(graph_sell(g));
  if (!g.dense.on) SS_TRY(graph_chunked(g, false));

  timing_begin_call();
  hipEvent_t e_begin, e_end;
  SS_TRY(timing_mark(&e_begin));
  const bool direct = (mem == SS_MEM_DEVICE && layout == SS_LAYOUT_ROWMAJOR);
  DevBuf<T> scores;
  T* dev_rm = out;
  int64_t ld_rm = ld;
  if (!direct) {
    SS_TRY(scores.alloc((size_t)ns * nt));
    dev_rm = scores.p;
    ld_rm = nt;
  }
  DevBuf<T> sorted_tmp;
  for (int phi = 0; phi < nfolds; ++phi) {
    const int64_t nm = start[phi + 1] - start[phi];
    if (nm == 0) continue;
    const int* members = d_order.p + start[phi];
    SS_HIP(hipMemcpyAsync(kf.p, g.kf.p, g.nf * sizeof(int), hipMemcpyDeviceToDevice, st));
    SS_HIP(hipMemcpyAsync(ks.p, g.ks.p, ns * sizeof(int), hipMemcpyDeviceToDevice, st));
    SS_HIP(hipMemcpyAsync(kt.p, g.kt.p, nt * sizeof(int), hipMemcpyDeviceToDevice, st));
    if (g.dense.on) {
      SS_TRY(dense_fold_degrees<T>(g, members, nm, kf.p, ks.p, kt.p));
    } else {
      SS_TRY(launch_fold_degrees<T>(g.Xs, g.XsT, g.Ys, members, nm, kf.p, ks.p, kt.p));
    }
    SS_TRY(launch_fold_inverse<T>(kf.p, ks.p, d_fold.p, phi, g.nf, ns, inv_kf.p, inv_ks.p));
    const int64_t rb = transfer_batch_rows(nm, ns, sizeof(T));
    const size_t need = (size_t)rb * (size_t)ns;
    if (g.Tws.n < need) SS_TRY(g.Tws.alloc(need));
    for (int64_t r0 = 0; r0 < nm; r0 += rb) {
      const int64_t nb = (nm - r0 < rb) ? (nm - r0) : rb;
      {
        StageTimer t1(ST_TRANSFER);
        if (g.dense.on) {
          // members' rows gathered into the query planes with this fold's 1/kf (0 on the members' own feature
          // columns); the epilogue multiplies by this fold's 1/ks (0 for the members as sources)
          if constexpr (std::is_same<T, float>::value)
            SS_TRY(launch_transfer_dense_bf16(g.dense, false, inv_kf.p, inv_ks.p, nullptr, r0, nb, g.Tws.p, ns, false,
                                              members));
          else
            SS_TRY(launch_transfer_dense_f64(g.dense, false, inv_kf.p, inv_ks.p, nullptr, r0, nb, g.Tws.p, ns, false,
                                             members));
        } else {
          const DevCsr<T>* L[2] = {&g.Xs, nullptr};
          const DevChunked<T>* M[2] = {&g.XsTc, nullptr};
          const T* inv1[2] = {inv_kf.p, nullptr};
          SS_TRY(launch_transfer<T>(1, L, inv1, M, inv_ks.p, r0, nb, ns, g.Tws.p, ns, members));
        }
        timing_count(ST_NTRANSFER, 1);
      }
      if (!g.W.sorted) {
        StageTimer t2(ST_SPMM);
        SS_TRY(launch_spmm_sell<T>(g.W, g.Tws.p, ns, nb, dev_rm, ld_rm, clean ? kt.p : nullptr, members + r0));
        timing_count(ST_NSPMM, 1);
      } else {
        const size_t need_s = (size_t)rb * (size_t)g.W.vrows + (size_t)rb * (size_t)nt;
        if (g.Sws.n < need_s) SS_TRY(g.Sws.alloc(need_s));
        T* packed = g.Sws.p + (size_t)rb * (size_t)g.W.vrows;  // member-ordered rows before the scatter
        {
          StageTimer t2(ST_SPMM);
          SS_TRY(launch_spmm_sell<T>(g.W, g.Tws.p, ns, nb, g.Sws.p, g.W.vrows, nullptr));
          timing_count(ST_NSPMM, 1);
        }
        StageTimer t3(ST_EPILOGUE);
        SS_TRY(launch_unpermute<T>(g.Sws.p, g.W.vrows, nb, nt, g.W.vfirst.p, g.W.inv.p, clean ? kt.p : nullptr, packed, nt));
        for (int64_t q = 0; q < nb; ++q)  // few folds x rows: row copies to the members' source rows
          SS_HIP(hipMemcpyAsync(dev_rm + (int64_t)order[start[phi] + r0 + q] * ld_rm, packed + q * nt, nt * sizeof(T),
                                hipMemcpyDeviceToDevice, st));
      }
    }
  }
  DevBuf<T> cm;
  if (layout == SS_LAYOUT_COLMAJOR) {
    StageTimer t3(ST_EPILOGUE);
    T* dst = out;
    int64_t dld = ld;
    if (mem == SS_MEM_HOST) {
      SS_TRY(cm.alloc((size_t)ns * nt));
      dst = cm.p;
      dld = ns;
    }
    SS_TRY(launch_transpose<T>(dev_rm, ns, nt, ld_rm, dst, dld));
  }
  SS_TRY(timing_mark(&e_end));
  timing_span(ST_TOTAL, e_begin, e_end);
  if (mem == SS_MEM_HOST) {
    StageTimer t5(ST_D2H);
    if (layout == SS_LAYOUT_ROWMAJOR)
      SS_HIP(hipMemcpy2DAsync(out, ld * sizeof(T), dev_rm, ld_rm * sizeof(T), nt * sizeof(T), ns, hipMemcpyDeviceToHost, st));
    else
      SS_HIP(hipMemcpy2DAsync(out, ld * sizeof(T), cm.p, ns * sizeof(T), ns * sizeof(T), nt, hipMemcpyDeviceToHost, st));
    t5.stop();
  }
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

// ------------------------------------------------------------------ raw SpMM
template <class T>
static int spmat_check(const void* h, SpMat<T>** out) {
  if (!h) return fail(SS_EINVAL, "matrix handle is NULL");
  SpMatBox<T>* b = const_cast<SpMatBox<T>*>(reinterpret_cast<const SpMatBox<T>*>(h));
  if (b->dtype != (int)sizeof(T)) return fail(SS_EINVAL, "matrix handle was created with the other precision");
  *out = &b->m;
  return SS_OK;
}

template <class T>
static int spmat_create_impl(int64_t rows, int64_t cols, const int64_t* ptr, const int32_t* idx, const T* val,
                             int index_base, int mem, ss_spmat** out) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  if (!out) return fail(SS_EINVAL, "out handle pointer is NULL");
  *out = nullptr;
  SpMatBox<T>* box = new (std::nothrow) SpMatBox<T>();
  if (!box) return fail(SS_ENOMEM, "host allocation failed");
  box->dtype = (int)sizeof(T);
  int rc = csr_from_user<T>(rows, cols, ptr, idx, val, index_base, mem, box->m.csr);
  if (rc != SS_OK) { delete box; return rc; }
  *out = reinterpret_cast<ss_spmat*>(box);
  return SS_OK;
}

template <class T>
static int spmm_impl(ss_spmat* h, const T* R, int64_t B, int64_t ldr, int r_layout, T* F, int64_t ldf, int f_layout,
                     int mem) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  SS_TRY(check_layout(r_layout));
  SS_TRY(check_layout(f_layout));
  SpMat<T>* mp = nullptr;
  SS_TRY(spmat_check<T>(h, &mp));
  SpMat<T>& m = *mp;
  const int64_t M = m.csr.rows, K = m.csr.cols;
  if (B < 0) return fail(SS_EINVAL, "B < 0");
  if (B == 0 || M == 0) return SS_OK;
  if (!F || (K > 0 && !R)) return fail(SS_EINVAL, "NULL operand");
  if (ldr < (r_layout == SS_LAYOUT_ROWMAJOR ? B : K)) return fail(SS_EINVAL, "ldr too small");
  if (ldf < (f_layout == SS_LAYOUT_ROWMAJOR ? B : M)) return fail(SS_EINVAL, "ldf too small");
  hipStream_t st = ctx().stream;
  timing_begin_call();

  // stage caller operands on the device in their own layout
  DevBuf<T> dR, dF;
  const T* Rd = R;
  T* Fd = F;
  int64_t ldr_d = ldr, ldf_d = ldf;
  if (mem == SS_MEM_HOST) {
    StageTimer t4(ST_H2D);
    const int64_t r_outer = (r_layout == SS_LAYOUT_ROWMAJOR) ? K : B, r_inner = (r_layout == SS_LAYOUT_ROWMAJOR) ? B : K;
    SS_TRY(dR.alloc((size_t)(r_outer > 0 ? r_outer : 1) * r_inner));
    if (r_outer > 0)
      SS_HIP(hipMemcpy2DAsync(dR.p, r_inner * sizeof(T), R, ldr * sizeof(T), r_inner * sizeof(T), r_outer,
                              hipMemcpyHostToDevice, st));
    Rd = dR.p;
    ldr_d = r_inner;
    const int64_t f_outer = (f_layout == SS_LAYOUT_ROWMAJOR) ? M : B, f_inner = (f_layout == SS_LAYOUT_ROWMAJOR) ? B : M;
    SS_TRY(dF.alloc((size_t)f_outer * f_inner));
    Fd = dF.p;
    ldf_d = f_inner;
  }
  hipEvent_t e_begin, e_end;
  SS_TRY(timing_mark(&e_begin));
  // Routing by width, row-major operands (measured at 100k x 100k / 1 %):
  //   B <= 4                          narrow kernel (fp32 B = 3, 4: spmm_csell.hip on 16-byte tile rows): W streamed once, the chunk of R in LDS, lanes of a row folded -- the
  //                                   HBM-bound regime (B = 1 0.13 ms = 4.6 TB/s of the 6 B/nnz operand)
  //   5 <= B, B*sizeof(T) <= 256 B    fp32: lane-per-row kernel on the compact sliced-ELL operand (spmm_csell.hip; B = 8 / 16 / 32 / 64
  //                                   0.19 / 0.18 / 0.35 / 0.70 ms); fp64 (and SS_CSELL=0): 2-D kernel (spmm_colgroup.hip), tile rows
  //                                   of 64 / 128 / 256 bytes; fp32 B = 5..16 0.22-0.24 ms, 32 0.44, 64 0.9
  //   wider, and pattern-only W (every value 1) above 128-byte tile rows: the SELL kernel of stage 2, which re-streams only
  //                                   the 2-byte indices (B = 64: 0.61 vs 0.82 ms); column-major operands always
  // SS_COL=0: no 2-D kernel (SELL instead); SS_COL_FROM: its first B, pattern-only wide cases included (comparisons, tests)
  const bool rowmajor = r_layout == SS_LAYOUT_ROWMAJOR && f_layout == SS_LAYOUT_ROWMAJOR;
  int col_from = 5;
  if (const char* e = getenv("SS_COL_FROM")) col_from = atoi(e);
  const bool col_wide_ok = !(m.csr.binary && B * (int64_t)sizeof(T) > 128) || getenv("SS_COL_FROM") != nullptr;
  const bool col = rowmajor && B >= col_from && B * (int64_t)sizeof(T) <= 256 && col_wide_ok &&
                   !(getenv("SS_COL") && atoi(getenv("SS_COL")) == 0);
  const bool narrow = rowmajor && B <= 4;
  DevBuf<T> Rt, Ft;
  if (col) {
    const int rowb = B * (int64_t)sizeof(T) <= 64 ? 64 : (B * (int64_t)sizeof(T) <= 128 ? 128 : 256);
    const int slot = rowb == 64 ? 0 : (rowb == 128 ? 1 : 2);
    const int bv = rowb / (int)sizeof(T);
    // the lane-per-row kernel on the compact sliced-ELL operand (spmm_csell.hip, round 3); SS_CSELL=0: the 2-D kernel
    bool done = false;
    if (!(getenv("SS_CSELL") && atoi(getenv("SS_CSELL")) == 0)) {
      // fp32 B <= 8: 32-byte tile rows (two pieces) instead of half-empty 64-byte ones
      const bool half = sizeof(T) == 4 && B <= 8 && !(getenv("SS_CSELL_ROW32") && atoi(getenv("SS_CSELL_ROW32")) == 0);
      const int crowb = half ? 32 : rowb, cslot = half ? 3 : slot, cbv = crowb / (int)sizeof(T);
      DevCsell<T>& cs = m.csell[cslot];
      if (!m.csell_tried[cslot]) {
        int kc = csell_chunk_cols(crowb);
        if (const char* e = getenv("SS_NARROW_CHUNK")) {
          const int v = atoi(e);
          if (v >= 16 && v < kc) kc = v;
        }
        SS_TRY(csell_build<T>(m.csr, kc, cbv, cs));
        m.csell_tried[cslot] = true;
      }
      if (cs.ok) {
        StageTimer t2(ST_SPMM);
        SS_TRY(launch_spmm_csell<T>(cs, Rd, ldr_d, (int)B, Fd, ldf_d, m.partial));
        timing_count(ST_NSPMM, 1);
        done = true;
      }
    }
    DevChunked<T>& op = m.col[slot];
    if (!done && op.SC == 0) {
      int kc = colgroup_chunk_cols<T>(bv);
      if (const char* e = getenv("SS_NARROW_CHUNK")) {
        const int v = atoi(e);
        if (v >= 16 && v < kc) kc = v;
      }
      SS_TRY(chunked_build<T>(m.csr, kc, 4, op));
    }
    if (!done) {
      StageTimer t2(ST_SPMM);
      SS_TRY(launch_spmm_colgroup<T>(op, bv, Rd, ldr_d, (int)B, Fd, ldf_d, m.partial));
      timing_count(ST_NSPMM, 1);
    }
  } else if (narrow && sizeof(T) == 8 && B == 2 && !(getenv("SS_CSELL_B12") && atoi(getenv("SS_CSELL_B12")) == 0) &&
             !(getenv("SS_CSELL") && atoi(getenv("SS_CSELL")) == 0) &&
             [&]() -> bool {   // fp64 B = 2 on 16-byte tile rows: 0.199 vs 0.234 ms for the narrow kernel (measured also: fp64
                               // B = 1 0.204 vs 0.200, fp32 B = 1 / 2 0.138 / 0.139 vs 0.112 / 0.116 -- those stay narrow)
               DevCsell<T>& cs = m.csell[5];
               if (!m.csell_tried[5]) {
                 if (csell_build<T>(m.csr, csell_chunk_cols(16), 16 / (int)sizeof(T), cs) != SS_OK) return false;
                 m.csell_tried[5] = true;
               }
               return cs.ok;
             }()) {
    StageTimer t2(ST_SPMM);
    SS_TRY(launch_spmm_csell<T>(m.csell[5], Rd, ldr_d, (int)B, Fd, ldf_d, m.partial));
    timing_count(ST_NSPMM, 1);
  } else if (narrow && B >= 3 && !(getenv("SS_CSELL_ROW16") && atoi(getenv("SS_CSELL_ROW16")) == 0) &&
             !(getenv("SS_CSELL") && atoi(getenv("SS_CSELL")) == 0) &&
             [&]() -> bool {   // B = 3, 4: the lane-per-row kernel with four columns per tile row (16 bytes in fp32: 0.136
                               // vs 0.150 ms at B = 4; 32 bytes in fp64)
               DevCsell<T>& cs = m.csell[4];
               if (!m.csell_tried[4]) {
                 if (csell_build<T>(m.csr, csell_chunk_cols(4 * (int)sizeof(T)), 4, cs) != SS_OK) return false;
                 m.csell_tried[4] = true;
               }
               return cs.ok;
             }()) {
    StageTimer t2(ST_SPMM);
    SS_TRY(launch_spmm_csell<T>(m.csell[4], Rd, ldr_d, (int)B, Fd, ldf_d, m.partial));
    timing_count(ST_NSPMM, 1);
  } else if (narrow) {
    int slot = 0, bv = 1;
    while (bv < B) { bv <<= 1; ++slot; }
    DevChunked<T>& op = m.narrow[slot];
    if (op.SC == 0) {
      int kc = narrow_chunk_cols<T>(bv);
      if (const char* e = getenv("SS_NARROW_CHUNK")) {
        const int v = atoi(e);
        if (v >= 16 && v < kc) kc = v;
      }
      SS_TRY(chunked_build<T>(m.csr, kc, 4, op));
    }
    StageTimer t2(ST_SPMM);
    SS_TRY(launch_spmm_chunked_narrow<T>(op, bv, Rd, ldr_d, (int)B, Fd, ldf_d, m.partial));
    timing_count(ST_NSPMM, 1);
  } else {
    const int qt = sell_tile_width<T>();
    if (m.sell_qt != qt) {
      int kcmax = sell_max_chunk<T>(qt);
      if (const char* e = getenv("SS_SELL_CHUNK")) {
        const int v = atoi(e);
        if (v >= 64 && v < kcmax) kcmax = v;
      }
      SS_TRY(sell_build<T>(m.csr, kcmax, m.sell));
      m.sell_qt = qt;
    }
    const T* Rc = Rd;   // column-major view: R(k,b) at Rc[b*ldrc + k]
    int64_t ldrc = ldr_d;
    if (r_layout == SS_LAYOUT_ROWMAJOR) {
      StageTimer t3(ST_EPILOGUE);
      SS_TRY(Rt.alloc((size_t)B * (K > 0 ? K : 1)));
      SS_TRY(launch_transpose<T>(Rd, K, B, ldr_d, Rt.p, K));
      Rc = Rt.p;
      ldrc = K;
    }
    T* Fc = Fd;
    int64_t ldfc = ldf_d;
    if (f_layout == SS_LAYOUT_ROWMAJOR) {
      SS_TRY(Ft.alloc((size_t)B * M));
      Fc = Ft.p;
      ldfc = M;
    }
    DevBuf<T> Fs;  // skew-sorted operand: sorted-order result, put back through inv[]
    if (!m.sell.sorted) {
      StageTimer t2(ST_SPMM);
      SS_TRY(launch_spmm_sell<T>(m.sell, Rc, ldrc, B, Fc, ldfc, nullptr));
      timing_count(ST_NSPMM, 1);
    } else {
      SS_TRY(Fs.alloc((size_t)B * m.sell.vrows));
      {
        StageTimer t2(ST_SPMM);
        SS_TRY(launch_spmm_sell<T>(m.sell, Rc, ldrc, B, Fs.p, m.sell.vrows, nullptr));
        timing_count(ST_NSPMM, 1);
      }
      StageTimer t3(ST_EPILOGUE);
      SS_TRY(launch_unpermute<T>(Fs.p, m.sell.vrows, B, M, m.sell.vfirst.p, m.sell.inv.p, nullptr, Fc, ldfc));
    }
    if (f_layout == SS_LAYOUT_ROWMAJOR) {
      StageTimer t3(ST_EPILOGUE);
      SS_TRY(launch_transpose<T>(Fc, B, M, ldfc, Fd, ldf_d));
    }
  }
  SS_TRY(timing_mark(&e_end));
  timing_span(ST_TOTAL, e_begin, e_end);
  if (mem == SS_MEM_HOST) {
    StageTimer t5(ST_D2H);
    const int64_t f_outer = (f_layout == SS_LAYOUT_ROWMAJOR) ? M : B, f_inner = (f_layout == SS_LAYOUT_ROWMAJOR) ? B : M;
    SS_HIP(hipMemcpy2DAsync(F, ldf * sizeof(T), dF.p, f_inner * sizeof(T), f_inner * sizeof(T), f_outer,
                            hipMemcpyDeviceToHost, st));
    t5.stop();
  }
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

}  // namespace ss

// ==================================================================== extern "C"
using namespace ss;

// Locking.  The process-wide context (device, stream) is read by every entry point and changed only by ss_init /
// ss_shutdown / ss_set_stream / ss_reset_stream: those take the context lock exclusively, everything else shares it.
// Work on a handle additionally holds that handle's own lock (HandleHead::mu).  Timings, the kernel-path note and
// the error message are per host thread.  So calls from several host threads / Julia tasks on different handles
// run concurrently (their kernels interleave on the library stream), calls on one handle are serialised.
static std::shared_mutex& ctx_mutex() {
  static std::shared_mutex m;
  return m;
}
// HIP's current device is per host thread: a thread that never called ss_init still has to launch on the library's
static inline void bind_device() {
  if (ctx().inited) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != ctx().device) (void)hipSetDevice(ctx().device);
  }
}
#define SS_API_EXCLUSIVE() std::unique_lock<std::shared_mutex> _ss_ctx_guard(ctx_mutex())
#define SS_API_LOCK()                                             \
  std::shared_lock<std::shared_mutex> _ss_ctx_guard(ctx_mutex()); \
  bind_device()
#define SS_HANDLE_LOCK(h) \
  std::unique_lock<std::mutex> _ss_handle_guard(reinterpret_cast<HandleHead*>(const_cast<void*>(static_cast<const void*>(h)))->mu)

template <class T>
static int jaccard_impl(const T* F, int64_t n, int64_t d, int64_t ld, T* S, int64_t lds_, int mem) {
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  if (n < 0 || d < 0 || ld < n || lds_ < n) return fail(SS_EINVAL, "similarity: bad shape / leading dimension");
  if (n == 0) return SS_OK;
  if ((d > 0 && !F) || !S) return fail(SS_EINVAL, "similarity: NULL buffer");
  hipStream_t st = ctx().stream;
  if (mem == SS_MEM_DEVICE) return launch_jaccard<T>(F, n, d, ld, S, lds_);
  DevBuf<T> dF, dS;
  SS_TRY(dF.alloc((size_t)n * (d > 0 ? d : 1)));
  SS_TRY(dS.alloc((size_t)n * n));
  if (d > 0)
    SS_HIP(hipMemcpy2DAsync(dF.p, n * sizeof(T), F, ld * sizeof(T), n * sizeof(T), d, hipMemcpyHostToDevice, st));
  SS_TRY(launch_jaccard<T>(dF.p, n, d, n, dS.p, n));
  SS_HIP(hipMemcpy2DAsync(S, lds_ * sizeof(T), dS.p, n * sizeof(T), n * sizeof(T), n, hipMemcpyDeviceToHost, st));
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

extern "C" {

static int shutdown_locked();

int ss_version(void) { return SS_VERSION; }

const char* ss_last_error(void) { return last_error().c_str(); }

int ss_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ss_init(int device) {
  SS_API_EXCLUSIVE();
  Ctx& c = ctx();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(SS_ENODEV, "no HIP device visible (%s)", hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(SS_EINVAL, "device %d outside 0..%d", device, n - 1);
  if (c.inited && c.device == device) return SS_OK;
  if (c.inited) shutdown_locked();
  SS_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  SS_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(SS_ENODEV, "device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
  c.num_cu = prop.multiProcessorCount;
  SS_HIP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
  c.stream = c.own_stream;
  c.device = device;
  c.inited = true;
  return SS_OK;
}

static int shutdown_locked() {
  Ctx& c = ctx();
  if (!c.inited) return SS_OK;
  (void)hipStreamSynchronize(c.stream);
  {
    // the exclusive context lock is held: no other thread is inside an entry point, so every thread's pool is quiescent
    std::lock_guard<std::mutex> lk(timing_reg_mu());
    for (Timing* t : timing_reg()) {
      if (t->generation == c.generation)
        for (hipEvent_t e : t->pool) (void)hipEventDestroy(e);
      t->pool.clear();
      t->spans.clear();
      t->used = 0;
    }
  }
  ++c.generation;  // other threads notice at their next call (their lists are already empty)
  (void)hipStreamDestroy(c.own_stream);
  c.own_stream = nullptr;
  c.stream = nullptr;
  c.inited = false;
  c.device = -1;
  return SS_OK;
}

int ss_shutdown(void) {
  SS_API_EXCLUSIVE();
  return shutdown_locked();
}

int ss_set_stream(void* hip_stream) {
  SS_API_EXCLUSIVE();
  SS_TRY(require_init());
  Ctx& c = ctx();
  SS_HIP(hipStreamSynchronize(c.stream));
  c.stream = reinterpret_cast<hipStream_t>(hip_stream);  // NULL = the null stream
  return SS_OK;
}

int ss_reset_stream(void) {
  SS_API_EXCLUSIVE();
  SS_TRY(require_init());
  Ctx& c = ctx();
  SS_HIP(hipStreamSynchronize(c.stream));
  c.stream = c.own_stream;
  return SS_OK;
}

int ss_synchronize(void) {
  SS_API_LOCK();
  SS_TRY(require_init());
  SS_HIP(hipStreamSynchronize(ctx().stream));
  return SS_OK;
}

int ss_comm_unique_id(char id[128]) {
  SS_API_LOCK();
  return comm_unique_id(id);
}
int ss_comm_init(const char id[128], int rank, int nranks) {
  SS_API_LOCK();
  return comm_init(id, rank, nranks);
}
int ss_comm_destroy(void) {
  SS_API_LOCK();
  return comm_destroy();
}
int ss_comm_info(int* rank, int* nranks) { return comm_info(rank, nranks); }
int ss_gather_rows_f32(const float* local, int64_t ncols, const int64_t* counts, float* full, int root) {
  SS_API_LOCK();
  return gather_rows(local, ncols, counts, full, root, 4);
}
int ss_gather_rows_f64(const double* local, int64_t ncols, const int64_t* counts, double* full, int root) {
  SS_API_LOCK();
  return gather_rows(local, ncols, counts, full, root, 8);
}

int ss_path_last(char* buf, int n) {
  if (!buf || n <= 0) return fail(SS_EINVAL, "ss_path_last: no buffer");
  const std::string& s = path_note();
  const size_t m = s.size() < (size_t)(n - 1) ? s.size() : (size_t)(n - 1);
  memcpy(buf, s.data(), m);
  buf[m] = 0;
  return SS_OK;
}

int ss_timing_hold(int enable) {
  SS_API_LOCK();
  SS_TRY(require_init());
  Timing& t = timing();
  t.hold = false;
  if (enable) {
    timing_begin_call();  // start from zero
    t.hold = true;
  }
  return SS_OK;
}

int ss_timing_last(double* ms, int n) {
  SS_API_LOCK();
  SS_TRY(require_init());
  if (!ms || n <= 0) return fail(SS_EINVAL, "ss_timing_last: bad buffer");
  Timing& t = timing();
  if (t.dirty) {
    SS_HIP(hipStreamSynchronize(ctx().stream));
    for (double& r : t.resolved) r = 0;
    for (const Timing::Span& s : t.spans) {
      float f = 0;
      SS_HIP(hipEventElapsedTime(&f, s.a, s.b));
      t.resolved[s.stage] += f;
    }
    t.resolved[ST_NSPMM] = t.extra[ST_NSPMM];
    t.resolved[ST_NTRANSFER] = t.extra[ST_NTRANSFER];
    t.dirty = false;
  }
  for (int i = 0; i < n && i < 8; ++i) ms[i] = t.resolved[i];
  return SS_OK;
}

int ss_similarity_jaccard_f32(const float* F, int64_t n, int64_t d, int64_t ld, float* S, int64_t lds_, int mem) {
  SS_API_LOCK();
  return jaccard_impl<float>(F, n, d, ld, S, lds_, mem);
}
int ss_similarity_jaccard_f64(const double* F, int64_t n, int64_t d, int64_t ld, double* S, int64_t lds_, int mem) {
  SS_API_LOCK();
  return jaccard_impl<double>(F, n, d, ld, S, lds_, mem);
}

int ss_cutoff_f32(const float* X, int64_t rows, int64_t cols, int64_t ld, float alpha, int weighted, float* out,
                  int64_t ldo, int mem) {
  SS_API_LOCK();
  return cutoff_impl<float>(X, rows, cols, ld, alpha, weighted, out, ldo, mem);
}
int ss_cutoff_f64(const double* X, int64_t rows, int64_t cols, int64_t ld, double alpha, int weighted, double* out,
                  int64_t ldo, int mem) {
  SS_API_LOCK();
  return cutoff_impl<double>(X, rows, cols, ld, alpha, weighted, out, ldo, mem);
}
int ss_row_degree_f32(const float* G, int64_t rows, int64_t cols, int64_t ld, int64_t* deg, int mem) {
  SS_API_LOCK();
  return row_degree_impl<float>(G, rows, cols, ld, deg, mem);
}
int ss_row_degree_f64(const double* G, int64_t rows, int64_t cols, int64_t ld, int64_t* deg, int mem) {
  SS_API_LOCK();
  return row_degree_impl<double>(G, rows, cols, ld, deg, mem);
}
int ss_spread_f32(const float* G, int64_t rows, int64_t cols, int64_t ld, float* W, int64_t ldw, int mem) {
  SS_API_LOCK();
  return spread_impl<float>(G, rows, cols, ld, W, ldw, mem);
}
int ss_spread_f64(const double* G, int64_t rows, int64_t cols, int64_t ld, double* W, int64_t ldw, int mem) {
  SS_API_LOCK();
  return spread_impl<double>(G, rows, cols, ld, W, ldw, mem);
}

int ss_graph_create_csr_f32(int64_t nq, int64_t ns, int64_t nf, int64_t nt, const int64_t* xq_ptr,
                            const int32_t* xq_idx, const float* xq_val, const int64_t* xs_ptr, const int32_t* xs_idx,
                            const float* xs_val, const int64_t* ys_ptr, const int32_t* ys_idx, const float* ys_val,
                            int index_base, int mem, ss_graph** out) {
  SS_API_LOCK();
  return graph_create_csr_impl<float>(nq, ns, nf, nt, xq_ptr, xq_idx, xq_val, xs_ptr, xs_idx, xs_val, ys_ptr, ys_idx,
                                      ys_val, index_base, mem, out);
}
int ss_graph_create_csr_f64(int64_t nq, int64_t ns, int64_t nf, int64_t nt, const int64_t* xq_ptr,
                            const int32_t* xq_idx, const double* xq_val, const int64_t* xs_ptr, const int32_t* xs_idx,
                            const double* xs_val, const int64_t* ys_ptr, const int32_t* ys_idx, const double* ys_val,
                            int index_base, int mem, ss_graph** out) {
  SS_API_LOCK();
  return graph_create_csr_impl<double>(nq, ns, nf, nt, xq_ptr, xq_idx, xq_val, xs_ptr, xs_idx, xs_val, ys_ptr, ys_idx,
                                       ys_val, index_base, mem, out);
}
int ss_graph_create_dense_f32(int64_t nq, int64_t ns, int64_t nf, int64_t nt, const float* Sq, int64_t ldq,
                              const float* Ss, int64_t lds, const float* Y, int64_t ldy, int apply_cutoff, float alpha,
                              int weighted, int mem, ss_graph** out) {
  SS_API_LOCK();
  return graph_create_dense_impl<float>(nq, ns, nf, nt, Sq, ldq, Ss, lds, Y, ldy, apply_cutoff, alpha, weighted, mem,
                                        out);
}
int ss_graph_create_dense_f64(int64_t nq, int64_t ns, int64_t nf, int64_t nt, const double* Sq, int64_t ldq,
                              const double* Ss, int64_t lds, const double* Y, int64_t ldy, int apply_cutoff,
                              double alpha, int weighted, int mem, ss_graph** out) {
  SS_API_LOCK();
  return graph_create_dense_impl<double>(nq, ns, nf, nt, Sq, ldq, Ss, lds, Y, ldy, apply_cutoff, alpha, weighted, mem,
                                         out);
}

int ss_graph_create_general_f32(int64_t n, int64_t nr, int64_t nc, const int64_t* l_ptr, const int32_t* l_idx,
                                const float* l_val, const int64_t* b_ptr, const int32_t* b_idx, const float* b_val,
                                const int64_t* w_ptr, const int32_t* w_idx, const float* w_val, int index_base, int mem,
                                ss_graph** out) {
  SS_API_LOCK();
  return graph_create_general_impl<float>(n, nr, nc, l_ptr, l_idx, l_val, b_ptr, b_idx, b_val, w_ptr, w_idx, w_val,
                                          index_base, mem, out);
}
int ss_graph_create_general_f64(int64_t n, int64_t nr, int64_t nc, const int64_t* l_ptr, const int32_t* l_idx,
                                const double* l_val, const int64_t* b_ptr, const int32_t* b_idx, const double* b_val,
                                const int64_t* w_ptr, const int32_t* w_idx, const double* w_val, int index_base,
                                int mem, ss_graph** out) {
  SS_API_LOCK();
  return graph_create_general_impl<double>(n, nr, nc, l_ptr, l_idx, l_val, b_ptr, b_idx, b_val, w_ptr, w_idx, w_val,
                                           index_base, mem, out);
}

int ss_graph_create_similarity_f32(int64_t nq, int64_t ns, int64_t nt, const float* Sq, int64_t ldq, const float* Ss,
                                   int64_t lds, const int64_t* y_ptr, const int32_t* y_idx, const float* y_val,
                                   int index_base, float alpha, int weighted, int mem, ss_graph** out) {
  SS_API_LOCK();
  return graph_create_similarity_impl<float>(nq, ns, nt, Sq, ldq, Ss, lds, y_ptr, y_idx, y_val, index_base, alpha,
                                             weighted, mem, out);
}
int ss_graph_create_similarity_f64(int64_t nq, int64_t ns, int64_t nt, const double* Sq, int64_t ldq, const double* Ss,
                                   int64_t lds, const int64_t* y_ptr, const int32_t* y_idx, const double* y_val,
                                   int index_base, double alpha, int weighted, int mem, ss_graph** out) {
  SS_API_LOCK();
  return graph_create_similarity_impl<double>(nq, ns, nt, Sq, ldq, Ss, lds, y_ptr, y_idx, y_val, index_base, alpha,
                                              weighted, mem, out);
}

int ss_graph_destroy(ss_graph* h) {
  SS_API_LOCK();
  if (!h) return SS_OK;
  const int dtype = *reinterpret_cast<int*>(h);
  if (dtype != 4 && dtype != 8) return fail(SS_EINVAL, "not a graph handle");
  {  // wait for a call that still works on the handle (the caller must not start new ones), then for the device
    SS_HANDLE_LOCK(h);
    if (ctx().inited) (void)hipStreamSynchronize(ctx().stream);
  }
  if (dtype == 4) delete reinterpret_cast<GraphBox<float>*>(h);
  else if (dtype == 8) delete reinterpret_cast<GraphBox<double>*>(h);
  else return fail(SS_EINVAL, "not a graph handle");
  return SS_OK;
}

int ss_graph_info(const ss_graph* h, int64_t sizes[7]) {
  SS_API_LOCK();
  if (!h) return fail(SS_EINVAL, "handle is NULL");
  SS_HANDLE_LOCK(h);
  if (!h || !sizes) return fail(SS_EINVAL, "NULL argument");
  const int dtype = *reinterpret_cast<const int*>(h);
  auto fill = [&](auto* b) {
    sizes[0] = b->g.nq; sizes[1] = b->g.ns; sizes[2] = b->g.nf; sizes[3] = b->g.nt;
    sizes[4] = b->g.Xq.nnz; sizes[5] = b->g.Xs.nnz; sizes[6] = b->g.Ys.nnz;
  };
  if (dtype == 4) fill(reinterpret_cast<const GraphBox<float>*>(h));
  else if (dtype == 8) fill(reinterpret_cast<const GraphBox<double>*>(h));
  else return fail(SS_EINVAL, "not a graph handle");
  return SS_OK;
}

int ss_graph_degrees(const ss_graph* h, int64_t* kf, int64_t* ks, int64_t* kt) {
  SS_API_LOCK();
  if (!h) return fail(SS_EINVAL, "handle is NULL");
  SS_HANDLE_LOCK(h);
  SS_TRY(require_init());
  if (!h) return fail(SS_EINVAL, "graph handle is NULL");
  const int dtype = *reinterpret_cast<const int*>(h);
  auto pull = [&](const DevBuf<int>& d, int64_t n, int64_t* out) -> int {
    if (!out || n == 0) return SS_OK;
    std::vector<int> tmp(n);
    SS_HIP(hipMemcpyAsync(tmp.data(), d.p, n * sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
    SS_HIP(hipStreamSynchronize(ctx().stream));
    for (int64_t i = 0; i < n; ++i) out[i] = tmp[i];
    return SS_OK;
  };
  auto run = [&](auto* b) -> int {
    SS_TRY(pull(b->g.kf, b->g.nf, kf));
    SS_TRY(pull(b->g.ks, b->g.ns, ks));
    SS_TRY(pull(b->g.kt, b->g.nt, kt));
    return SS_OK;
  };
  if (dtype == 4) return run(reinterpret_cast<const GraphBox<float>*>(h));
  if (dtype == 8) return run(reinterpret_cast<const GraphBox<double>*>(h));
  return fail(SS_EINVAL, "not a graph handle");
}

int ss_predict_f32(ss_graph* g, int rows_kind, int64_t row_begin, int64_t row_end, int clean, float* out, int64_t ld,
                   int layout, int mem) {
  SS_API_LOCK();
  if (!g) return fail(SS_EINVAL, "handle is NULL");
  SS_HANDLE_LOCK(g);
  if (rows_kind != SS_ROWS_QUERY && rows_kind != SS_ROWS_SOURCE) return fail(SS_EINVAL, "bad rows_kind");
  return predict_impl<float>(g, rows_kind, row_begin, row_end, clean, out, ld, layout, mem);
}
int ss_predict_f64(ss_graph* g, int rows_kind, int64_t row_begin, int64_t row_end, int clean, double* out, int64_t ld,
                   int layout, int mem) {
  SS_API_LOCK();
  if (!g) return fail(SS_EINVAL, "handle is NULL");
  SS_HANDLE_LOCK(g);
  if (rows_kind != SS_ROWS_QUERY && rows_kind != SS_ROWS_SOURCE) return fail(SS_EINVAL, "bad rows_kind");
  return predict_impl<double>(g, rows_kind, row_begin, row_end, clean, out, ld, layout, mem);
}
int ss_predict_loo_f32(ss_graph* g, int64_t i_begin, int64_t i_end, int clean, float* out, int64_t ld, int layout,
                       int mem) {
  SS_API_LOCK();
  if (!g) return fail(SS_EINVAL, "handle is NULL");
  SS_HANDLE_LOCK(g);
  return predict_impl<float>(g, 2, i_begin, i_end, clean, out, ld, layout, mem);
}
int ss_predict_loo_f64(ss_graph* g, int64_t i_begin, int64_t i_end, int clean, double* out, int64_t ld, int layout,
                       int mem) {
  SS_API_LOCK();
  if (!g) return fail(SS_EINVAL, "handle is NULL");
  SS_HANDLE_LOCK(g);
  return predict_impl<double>(g, 2, i_begin, i_end, clean, out, ld, layout, mem);
}

int ss_predict_kfold_f32(ss_graph* g, const int32_t* fold_of_source, int nfolds, int clean, float* out, int64_t ld,
                         int layout, int mem) {
  SS_API_LOCK();
  if (!g) return fail(SS_EINVAL, "handle is NULL");
  SS_HANDLE_LOCK(g);
  return predict_kfold_impl<float>(g, fold_of_source, nfolds, clean, out, ld, layout, mem);
}
int ss_predict_kfold_f64(ss_graph* g, const int32_t* fold_of_source, int nfolds, int clean, double* out, int64_t ld,
                         int layout, int mem) {
  SS_API_LOCK();
  if (!g) return fail(SS_EINVAL, "handle is NULL");
  SS_HANDLE_LOCK(g);
  return predict_kfold_impl<double>(g, fold_of_source, nfolds, clean, out, ld, layout, mem);
}

int ss_topl_f32(const float* scores, int64_t nrows, int64_t ncols, int64_t ld, int L, int32_t* idx, float* val,
                int mem) {
  SS_API_LOCK();
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  if (nrows < 0 || ncols < 0 || ld < ncols) return fail(SS_EINVAL, "top-L: bad shape / leading dimension");
  if (nrows == 0) return SS_OK;
  if (!scores || !idx || !val) return fail(SS_EINVAL, "top-L: NULL buffer");
  hipStream_t st = ctx().stream;
  if (mem == SS_MEM_DEVICE) return launch_topl(scores, nrows, ncols, ld, L, idx, val);
  DevBuf<float> ds, dv;
  DevBuf<int> di;
  SS_TRY(ds.alloc((size_t)nrows * ncols));
  SS_TRY(dv.alloc((size_t)nrows * L));
  SS_TRY(di.alloc((size_t)nrows * L));
  SS_HIP(hipMemcpy2DAsync(ds.p, ncols * sizeof(float), scores, ld * sizeof(float), ncols * sizeof(float), nrows,
                          hipMemcpyHostToDevice, st));
  SS_TRY(launch_topl(ds.p, nrows, ncols, ncols, L, di.p, dv.p));
  SS_HIP(hipMemcpyAsync(idx, di.p, (size_t)nrows * L * sizeof(int), hipMemcpyDeviceToHost, st));
  SS_HIP(hipMemcpyAsync(val, dv.p, (size_t)nrows * L * sizeof(float), hipMemcpyDeviceToHost, st));
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

int ss_rank_metrics_f32(const uint8_t* y, const float* yhat, int64_t n, double alpha, double out[4], int mem) {
  SS_API_LOCK();
  SS_TRY(require_init());
  SS_TRY(check_mem(mem));
  if (n <= 0) return fail(SS_EINVAL, "rank metrics: n must be positive");
  if (!y || !yhat || !out) return fail(SS_EINVAL, "rank metrics: NULL buffer");
  if (!(alpha > 0.0)) return fail(SS_EINVAL, "rank metrics: alpha must be positive");
  if (mem == SS_MEM_DEVICE) return launch_rank_metrics(y, yhat, n, alpha, out);
  hipStream_t st = ctx().stream;
  DevBuf<unsigned char> dy;
  DevBuf<float> ds;
  SS_TRY(dy.alloc((size_t)n));
  SS_TRY(ds.alloc((size_t)n));
  SS_HIP(hipMemcpyAsync(dy.p, y, (size_t)n, hipMemcpyHostToDevice, st));
  SS_HIP(hipMemcpyAsync(ds.p, yhat, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
  return launch_rank_metrics(dy.p, ds.p, n, alpha, out);
}

int ss_spmat_create_csr_f32(int64_t rows, int64_t cols, const int64_t* ptr, const int32_t* idx, const float* val,
                            int index_base, int mem, ss_spmat** out) {
  SS_API_LOCK();
  return spmat_create_impl<float>(rows, cols, ptr, idx, val, index_base, mem, out);
}
int ss_spmat_create_csr_f64(int64_t rows, int64_t cols, const int64_t* ptr, const int32_t* idx, const double* val,
                            int index_base, int mem, ss_spmat** out) {
  SS_API_LOCK();
  return spmat_create_impl<double>(rows, cols, ptr, idx, val, index_base, mem, out);
}
int ss_spmat_destroy(ss_spmat* h) {
  SS_API_LOCK();
  if (!h) return SS_OK;
  const int dtype = *reinterpret_cast<int*>(h);
  if (dtype != 4 && dtype != 8) return fail(SS_EINVAL, "not a matrix handle");
  {  // wait for a call that still works on the handle (the caller must not start new ones), then for the device
    SS_HANDLE_LOCK(h);
    if (ctx().inited) (void)hipStreamSynchronize(ctx().stream);
  }
  if (dtype == 4) delete reinterpret_cast<SpMatBox<float>*>(h);
  else if (dtype == 8) delete reinterpret_cast<SpMatBox<double>*>(h);
  else return fail(SS_EINVAL, "not a matrix handle");
  return SS_OK;
}
int ss_spmm_f32(ss_spmat* w, const float* R, int64_t B, int64_t ldr, int r_layout, float* F, int64_t ldf, int f_layout,
                int mem) {
  SS_API_LOCK();
  if (!w) return fail(SS_EINVAL, "handle is NULL");
  SS_HANDLE_LOCK(w);
  return spmm_impl<float>(w, R, B, ldr, r_layout, F, ldf, f_layout, mem);
}
int ss_spmm_f64(ss_spmat* w, const double* R, int64_t B, int64_t ldr, int r_layout, double* F, int64_t ldf,
                int f_layout, int mem) {
  SS_API_LOCK();
  if (!w) return fail(SS_EINVAL, "handle is NULL");
  SS_HANDLE_LOCK(w);
  return spmm_impl<double>(w, R, B, ldr, r_layout, F, ldf, f_layout, mem);
}
int ss_spmat_cost(const ss_spmat* h, int64_t B, double* bytes, double* flops) {
  SS_API_LOCK();
  if (!h) return fail(SS_EINVAL, "handle is NULL");
  SS_HANDLE_LOCK(h);
  if (!h) return fail(SS_EINVAL, "matrix handle is NULL");
  const int dtype = *reinterpret_cast<const int*>(h);
  int64_t rows, cols, nnz;
  if (dtype == 4) { auto* b = reinterpret_cast<const SpMatBox<float>*>(h); rows = b->m.csr.rows; cols = b->m.csr.cols; nnz = b->m.csr.nnz; }
  else if (dtype == 8) { auto* b = reinterpret_cast<const SpMatBox<double>*>(h); rows = b->m.csr.rows; cols = b->m.csr.cols; nnz = b->m.csr.nnz; }
  else return fail(SS_EINVAL, "not a matrix handle");
  const double vb = dtype;
  if (bytes) *bytes = (double)nnz * (vb + 4) + (double)(rows + 1) * 4 + (double)cols * B * vb + (double)rows * B * vb;
  if (flops) *flops = 2.0 * (double)nnz * (double)B;
  return SS_OK;
}

}  // extern "C"
