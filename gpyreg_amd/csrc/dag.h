// dag.h -- tile-level dataflow execution of the blocked factorization (round 5).
//
// The stream-ordered schedule of plan.h issues one launch per product of the recursion and one per 128 x 128 leaf:
// ~250 dependent launches at N_pad = 4096, during most of which a handful of the 256 CUs work (DESIGN.md section 9:
// 42 % of a two-sample step is the dependent chain of ONE sample).  Here the SAME launches are recorded instead of
// issued (plan.h: PlanRecorder), cut into their tiles -- each tile keeps the k-range, the operand order and therefore
// the bits it has in the launch -- and the tiles of ALL samples and ALL depths of the recursion become tasks of one
// dependency graph, executed by two co-resident persistent kernels:
//     dag_worker_kernel   2 workgroups per CU, 256 threads, the GEMM tile of gemm.h (128- or 64-tiles);
//     dag_leaf_kernel     a few workgroups (the pipelined leaf needs 122 KB of LDS: it cannot share a CU with two GEMM
//                         workgroups, so the workers stay off one CU per XCD -- cu_reserve_bail -- and the leaf
//                         servers find room there whatever the dispatcher does).
// Dependencies are counters, not kernel boundaries: every task has a `pending` count of unfinished predecessors
// (read-after-write, write-after-read and write-after-write on 64 x 64 cells of the three buffers A / W / T, derived
// on the host from the operand regions of the recorded launches); a finished task decrements its successors and
// pushes the ones that reach zero onto a ready ring; idle workgroups pop: the urgent ring (64-tile tasks: the
// latency-bound chain between two leaves) before the bulk rings (128-tile tasks, one ring per XCD by sample so
// that the workgroups of an XCD share operand panels in its L2), leaf servers the leaf ring.
// Hand-off forms (MI355X_MICROARCH.md, inter-workgroup visibility): results are stored write-through (sc1), every
// storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, then agent-scope atomics on the counters;
// a consumer pops (relaxed sc1 loads, CAS), runs ONE agent-scope acquire, waits for it, meets at a barrier and loads
// its operands with sc1 loads.  The leaf stores plainly and publishes with an agent-scope release fence.
// Every wait is bounded (wall clock): a hand-off that never comes sets the abort word, all workgroups leave, the host
// reports DAG_ABORT and re-runs the batch on the stream-ordered schedule -- never a hang, never a wrong number.
#pragma once
#include "gemm.h"
#include "leaf.h"
#include "plan.h"

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace gpc {

// Ready rings: one urgent and one bulk ring per TEAM (team of a sample = sample % nteams, nteams = min(samples, NQ)),
// and the leaf ring.  XCD x serves the urgent ring of team x (if there is one) -- the 62 workgroups of ONE XCD are
// plenty for the chain of one or two samples, share its tiles in one L2, and keep the number of workgroups that poll
// any one ring small -- and the bulk ring of team x % nteams, and steals from the other bulk rings when both are empty.
// (Round 5, first form: one urgent ring polled by all 496 workgroups -- 5 us per pop, 24 ms for a 1.4 ms batch.)
// Three rings per team: urgent (64-tile tasks with little slack: the chain between two leaves), crit (128-tile tasks
// with little slack: the panel products and the first tiles of a trailing update that the next subtree waits for),
// bulk (everything with slack: inverse products, far tiles of the updates, W^T W).  Popped in that order: FIFO rings
// alone let a task on the critical path queue behind hundreds of tiles nobody waits for (N = 8192, one sample: 32 ms
// against 12.8 ms stream-ordered).
enum { DAG_RING_URGENT = 0, DAG_RING_CRIT0 = NQ, DAG_RING_BULK0 = 2 * NQ, DAG_RING_LEAF = 3 * NQ, DAG_NRINGS = 3 * NQ + 1 };
enum { DAG_KIND_LEAF = 8 };  // 0..7: product, kind = (akm ? 4 : 0) | (bkm ? 2 : 0) | (64-tile ? 1 : 0)
constexpr int DAG_CELL = 64;
#ifndef GPC_DAG_HO
#define GPC_DAG_HO 2
#endif
constexpr int DAG_HO = GPC_DAG_HO;  // gemm.h: 2 = plain operand loads behind the acquire, 1 = sc1 operand loads

struct DagTask {  // 32 bytes, read-only on the device
  int launch;      // product: index into the launch table; leaf: diagonal offset (rows)
  int bx;          // product: the launch's linear tile index (gemm_tile decodes it)
  int kind;        // see DAG_KIND_*
  int ring;        // DAG_RING_URGENT / DAG_RING_BULK0 (the device adds the sample's team) / DAG_RING_LEAF
  int succ_begin, succ_count;
  int npred;
  int pad;
};

// control words of one run: every ring's (head, tail) on a 128-byte line of its own, then the global line
struct DagCtl {
  struct alignas(128) Line {
    unsigned long long ht;  // head in the low word, tail in the high word: one 8-byte load shows both
    int pad[30];
    __host__ __device__ int* head() { return reinterpret_cast<int*>(&ht); }
    __host__ __device__ int* tail() { return reinterpret_cast<int*>(&ht) + 1; }
  } ring[DAG_NRINGS];
  int abort;      // nonzero: a bounded wait ran out -- everybody leaves (1: nothing became ready, 2: a slot never arrived)
  int leaf_alive; // leaf servers that have started
  int remaining;  // tasks not yet finished; the kernels leave when it reaches zero
  int pad[29];
  int reserve_ctr[CTR_STRIDE];  // cu_reserve_bail's counters for the worker launch
  // statistics of the worker launch (GPC_DAG_LOG): ticks of the 100 MHz wall clock summed over the workgroups
  unsigned long long t_pop, t_acq, t_exec, t_done, n_tasks, n_kept, n_workers, t_life;
};

struct DagDev {  // kernel argument
  const DagTask* tasks;
  const int* succ;
  const GemmArgs* launches;      // = launches_w (absolute pointers, written by dag_init_kernel)
  const GemmArgs* launches_rel;  // the cached table: operand pointers as (buffer + 1) << 48 | byte offset
  GemmArgs* launches_w;
  int nlaunch;
  void* base[3];                 // sample 0's A, W, T of this run
  int* pending;  // [S][ntasks]
  int* slots;    // ring storage
  DagCtl* ctl;
  int ring_base[DAG_NRINGS];  // offset of each ring in `slots`
  int ntasks, S, nteams;
  int leaf_servers;     // leaf servers launched beside the workers (0: none -- the test hook)
  int urgent_cus;       // > 0: on the XCD that serves a team's urgent ring, the workgroups on CUs with CU_ID < urgent_cus (per
                        // shader engine) serve ONLY that ring, and no other workgroup serves it: chain tiles get CUs without a
                        // bulk neighbour (0: every workgroup of the XCD serves urgent, crit and bulk)
  int gate, gate_task;  // samples s >= gate start when sample s - gate has finished task gate_task (0: all start at once)
  // leaf arguments
  void* A;
  void* W;
  long long sA, sW;
  int npad, nvalid;
  double* logdet;
  int* info;
  const unsigned short* rsv;
  long long timeout_ticks;
  long long* trace;  // debug (GPC_DAG_TRACE): per (sample, task) [ready, started, popped, ended, completed] wall-clock ticks, worker
};

// ---------------------------------------------------------------------------------------------------- host: the graph
struct DagPlan {
  std::vector<DagTask> tasks;
  std::vector<int> succ;
  std::vector<GemmArgs> launches;
  std::vector<char> launch_akm, launch_bkm;
  std::vector<int> launch_bt;
  int ntasks = 0, nleaf = 0, n64 = 0, n128 = 0;
  int n_urgent = 0, n_crit = 0, n_bulk = 0;  // tasks per ring class (ring capacities)
  std::vector<int> leaf_task;               // task id of the k-th leaf
  long long nedges = 0;
  double crit_us = 0, work_us = 0;  // model figures (critical path, total work on one CU), for the log only
  // debug export (gpc_debug_dag): per task the regions it touches
  struct Region {
    int buf, r0, r1, c0, c1;
  };
  struct Info {
    Region c, a, b;
    int m0, n0, k0, k1, bt;
    double alpha;
    int beta;
  };
  std::vector<Info> info;
};

inline void host_tri_tile(int tile, int& ti, int& tj) {
  int i = (int)((std::sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while (i * (i + 1) / 2 > tile) --i;
  while ((i + 1) * (i + 2) / 2 <= tile) ++i;
  ti = i;
  tj = tile - i * (i + 1) / 2;
}

// Tile index -> (ti, tj) exactly as gemm_tile does it
inline void host_tile_of(const GemmArgs& g, int tiles_m, int tiles_n, int bx, int& ti, int& tj) {
  if (g.lower_only) {
    host_tri_tile(bx, ti, tj);
  } else if (g.khi == KHI_COL) {
    tj = tiles_n - 1 - bx / tiles_m;
    ti = bx % tiles_m;
  } else if (g.klo == KLO_COL) {
    tj = bx / tiles_m;
    ti = bx % tiles_m;
  } else if (g.khi == KHI_ROW) {
    ti = tiles_m - 1 - bx / tiles_n;
    tj = bx % tiles_n;
  } else {
    ti = bx / tiles_n;
    tj = bx % tiles_n;
  }
}

// `small_tiles`: a recorded launch with fewer 128-tiles per sample than this is cut into 64-tiles (urgent ring), the
// others into 128-tiles (bulk rings).  bases[3] = sample 0's A, W, T; esz = sizeof(T).
inline bool build_dag(const PlanRecorder& rec, const void* const bases[3], int npad, size_t esz, int small_tiles,
                      DagPlan& P, bool want_info = false, double crit_frac = 0.15) {
  if (rec.unsupported) return false;
  const int nc = npad / DAG_CELL;
  const size_t ncell = (size_t)3 * nc * nc;
  std::vector<int> last_writer(ncell, -1);
  std::vector<std::vector<int>> readers(ncell);
  std::vector<std::vector<int>> preds;
  auto locate = [&](const void* p, int& buf, int& r0, int& c0) -> bool {
    for (int b = 0; b < 3; ++b) {
      const char* lo = static_cast<const char*>(bases[b]);
      const char* q = static_cast<const char*>(p);
      if (q >= lo && q < lo + (size_t)npad * npad * esz) {
        const size_t off = (size_t)(q - lo) / esz;
        buf = b;
        r0 = (int)(off / npad);
        c0 = (int)(off % npad);
        return true;
      }
    }
    return false;
  };
  std::vector<int> deps;
  auto touch = [&](const DagPlan::Region& R, bool write, int t) {
    if (R.r1 <= R.r0 || R.c1 <= R.c0) return;
    for (int cr = R.r0 / DAG_CELL; cr <= (R.r1 - 1) / DAG_CELL; ++cr)
      for (int cc = R.c0 / DAG_CELL; cc <= (R.c1 - 1) / DAG_CELL; ++cc) {
        const size_t cell = ((size_t)R.buf * nc + cr) * nc + cc;
        if (last_writer[cell] >= 0) deps.push_back(last_writer[cell]);
        if (write) {
          for (int r : readers[cell]) deps.push_back(r);
          readers[cell].clear();
          last_writer[cell] = t;
        } else {
          readers[cell].push_back(t);
        }
      }
  };
  auto finish_task = [&](int t) {
    std::sort(deps.begin(), deps.end());
    deps.erase(std::unique(deps.begin(), deps.end()), deps.end());
    deps.erase(std::remove(deps.begin(), deps.end(), t), deps.end());
    preds.push_back(deps);
    deps.clear();
  };
  for (const PlanRecorder::Op& op : rec.ops) {
    if (op.kind == 1) {
      const int t = (int)P.tasks.size();
      DagTask d{};
      d.launch = op.off;
      d.bx = 0;
      d.kind = DAG_KIND_LEAF;
      d.ring = DAG_RING_LEAF;
      P.tasks.push_back(d);
      ++P.nleaf;
      // reads the lower cells of A's diagonal tile, writes them (L) and the whole tile of W
      const int o = op.off;
      DagPlan::Region lo0{0, o, o + 64, o, o + 64}, lo1{0, o + 64, o + 128, o, o + 128};
      touch(lo0, false, t);
      touch(lo1, false, t);
      touch(lo0, true, t);
      touch(lo1, true, t);
      touch(DagPlan::Region{1, o, o + 128, o, o + 128}, true, t);
      finish_task(t);
      if (want_info) {
        DagPlan::Info in{};
        in.c = DagPlan::Region{0, o, o + 128, o, o + 128};
        in.bt = 128;
        P.info.push_back(in);
      }
      continue;
    }
    GemmArgs g = op.g;
    const int tm128 = g.M / TILE, tn128 = g.N / TILE;
    const int tiles128 = g.lower_only ? tm128 * (tm128 + 1) / 2 : tm128 * tn128;
    const int bt = tiles128 < small_tiles ? 64 : 128;
    g.tiles_m = g.M / bt;
    g.tiles_n = g.N / bt;
    g.flags = 0;
    g.ntiles = g.lower_only ? g.tiles_m * (g.tiles_m + 1) / 2 : g.tiles_m * g.tiles_n;
    g.batch = 0;
    g.ctr = nullptr;
    g.rsv = nullptr;
    int bA, rA, cA, bB, rB, cB, bC, rC, cC;
    if (!locate(g.A, bA, rA, cA) || !locate(g.B, bB, rB, cB) || !locate(g.C, bC, rC, cC)) return false;
    const int li = (int)P.launches.size();
    P.launches.push_back(g);
    P.launch_akm.push_back(op.akm);
    P.launch_bkm.push_back(op.bkm);
    P.launch_bt.push_back(bt);
    for (int bx = 0; bx < g.ntiles; ++bx) {
      int ti, tj;
      host_tile_of(g, g.tiles_m, g.tiles_n, bx, ti, tj);
      const int m0 = ti * bt, n0 = tj * bt;
      const int m128 = (m0 / TILE) * TILE, n128 = (n0 / TILE) * TILE;
      int k0 = g.klo == KLO_ROW ? m128 : (g.klo == KLO_COL ? n128 : 0);
      int k1 = g.khi == KHI_ROW ? m128 + TILE : (g.khi == KHI_COL ? n128 + TILE : g.K);
      if (k1 > g.K) k1 = g.K;
      const int t = (int)P.tasks.size();
      DagTask d{};
      d.launch = li;
      d.bx = bx;
      d.kind = (op.akm ? 4 : 0) | (op.bkm ? 2 : 0) | (bt == 64 ? 1 : 0);
      d.ring = bt == 64 ? DAG_RING_URGENT : DAG_RING_BULK0;
      P.tasks.push_back(d);
      (bt == 64 ? P.n64 : P.n128)++;
      DagPlan::Region ra{bA, 0, 0, 0, 0}, rb{bB, 0, 0, 0, 0};
      if (k1 > k0) {
        ra = op.akm ? DagPlan::Region{bA, rA + k0, rA + k1, cA + m0, cA + m0 + bt}
                    : DagPlan::Region{bA, rA + m0, rA + m0 + bt, cA + k0, cA + k1};
        rb = op.bkm ? DagPlan::Region{bB, rB + k0, rB + k1, cB + n0, cB + n0 + bt}
                    : DagPlan::Region{bB, rB + n0, rB + n0 + bt, cB + k0, cB + k1};
      }
      const DagPlan::Region rc{bC, rC + m0, rC + m0 + bt, cC + n0, cC + n0 + bt};
      touch(ra, false, t);
      touch(rb, false, t);
      if (g.beta) touch(rc, false, t);
      touch(rc, true, t);
      finish_task(t);
      if (want_info) P.info.push_back(DagPlan::Info{rc, ra, rb, m0, n0, k0, k1, bt, g.alpha, g.beta});
    }
  }
  P.ntasks = (int)P.tasks.size();
  // successors = the reverse edges
  std::vector<int> cnt(P.ntasks, 0);
  for (int t = 0; t < P.ntasks; ++t) {
    P.tasks[t].npred = (int)preds[t].size();
    for (int p : preds[t]) ++cnt[p];
  }
  int run = 0;
  for (int t = 0; t < P.ntasks; ++t) {
    P.tasks[t].succ_begin = run;
    P.tasks[t].succ_count = cnt[t];
    run += cnt[t];
  }
  P.nedges = run;
  P.succ.assign(run, 0);
  std::vector<int> fill(P.ntasks, 0);
  for (int t = 0; t < P.ntasks; ++t)
    for (int p : preds[t]) P.succ[P.tasks[p].succ_begin + fill[p]++] = t;
  // Priority of a task = length of the longest path from it to the end (a model in microseconds: a 64-tile k-unit of
  // 128 costs 4.4 us, a 128-tile unit 16, a leaf 25, a hand-off 3).  The successors of a task are pushed in
  // descending priority, so that among the tasks one completion releases the one on the longest chain is popped first.
  std::vector<double> dur(P.ntasks), blevel(P.ntasks, 0.0);
  for (int t = 0; t < P.ntasks; ++t) {
    const DagTask& d = P.tasks[t];
    if (d.kind == DAG_KIND_LEAF) {
      dur[t] = 25.0 + 3.0;
    } else {
      const GemmArgs& g = P.launches[d.launch];
      int ti, tj;
      host_tile_of(g, g.tiles_m, g.tiles_n, d.bx, ti, tj);
      const int bt = P.launch_bt[d.launch];
      const int m128 = (ti * bt / TILE) * TILE, n128 = (tj * bt / TILE) * TILE;
      int k0 = g.klo == KLO_ROW ? m128 : (g.klo == KLO_COL ? n128 : 0);
      int k1 = g.khi == KHI_ROW ? m128 + TILE : (g.khi == KHI_COL ? n128 + TILE : g.K);
      if (k1 > g.K) k1 = g.K;
      const double units = std::max(0, k1 - k0) / 128.0;
      dur[t] = units * (bt == 64 ? 4.4 : 16.0) + 3.0;
    }
    P.work_us += dur[t];
  }
  for (int t = P.ntasks - 1; t >= 0; --t) {  // task ids are a topological order (launch order)
    double best = 0.0;
    for (int i = 0; i < P.tasks[t].succ_count; ++i) best = std::max(best, blevel[P.succ[P.tasks[t].succ_begin + i]]);
    blevel[t] = best + dur[t];
    P.crit_us = std::max(P.crit_us, blevel[t]);
  }
  for (int t = 0; t < P.ntasks; ++t) {
    int* b = P.succ.data() + P.tasks[t].succ_begin;
    std::stable_sort(b, b + P.tasks[t].succ_count, [&](int x, int y) { return blevel[x] > blevel[y]; });
  }
  // slack of a task = critical path - (longest path to its start + longest path from its start to the end)
  std::vector<double> tlevel(P.ntasks, 0.0);
  for (int t = 0; t < P.ntasks; ++t)
    for (int i = 0; i < P.tasks[t].succ_count; ++i) {
      const int t2 = P.succ[P.tasks[t].succ_begin + i];
      tlevel[t2] = std::max(tlevel[t2], tlevel[t] + dur[t]);
    }
  for (int t = 0; t < P.ntasks; ++t) {
    DagTask& d = P.tasks[t];
    if (d.kind == DAG_KIND_LEAF) {
      P.leaf_task.push_back(t);
      continue;
    }
    const double slack = P.crit_us - (tlevel[t] + blevel[t]);
    const bool critical = slack < crit_frac * P.crit_us;
    d.ring = critical ? ((d.kind & 1) ? DAG_RING_URGENT : DAG_RING_CRIT0) : DAG_RING_BULK0;
    (d.ring == DAG_RING_URGENT ? P.n_urgent : (d.ring == DAG_RING_CRIT0 ? P.n_crit : P.n_bulk))++;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------- device
__device__ __forceinline__ int dag_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Push task (s, t): entry = s * ntasks + t + 1 (0 = slot not yet written).
__device__ __forceinline__ void dag_push(const DagDev& d, int s, int t) {
  if (d.trace) d.trace[((size_t)s * d.ntasks + t) * 6 + 0] = wall_clock64();
  int ring = d.tasks[t].ring;
  if (ring != DAG_RING_LEAF) ring += s % d.nteams;
  const int pos = __hip_atomic_fetch_add(d.ctl->ring[ring].tail(), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(d.slots + d.ring_base[ring] + pos, s * d.ntasks + t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A finished task: the caller has drained its stores and met at the workgroup barrier.  Wave 0 decrements the
// successors (one lane each) and pushes the ones that became ready; lane 0 counts the task as done.
// `keep`: the calling workgroup is a GEMM worker and takes ONE of the tasks it has just made ready for itself -- the
// first in the successor list that is not a leaf (the list is sorted by priority) -- instead of sending it through a
// ring: the continuation of a chain costs no push, no poll and no pop.  Returns its entry (s * ntasks + t) or -1;
// valid in every lane of wave 0.
__device__ __forceinline__ int dag_complete(const DagDev& d, int s, int t, int keep) {
  const int lane = threadIdx.x & 63;
  const DagTask tk = d.tasks[t];
  int kept = -1;
  for (int i0 = 0; i0 < tk.succ_count; i0 += 64) {
    const int i = i0 + lane;
    int t2 = -1, ready = 0, mine = 0;
    if (i < tk.succ_count) {
      t2 = d.succ[tk.succ_begin + i];
      const int old = __hip_atomic_fetch_sub(d.pending + (size_t)s * d.ntasks + t2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ready = old == 1;
      mine = ready && keep && (keep == 2 ? d.tasks[t2].ring == DAG_RING_URGENT : d.tasks[t2].ring != DAG_RING_LEAF);
    }
    if (kept < 0) {
      const unsigned long long m = __ballot(mine);
      if (m) {
        const int src = __builtin_ctzll(m);
        kept = __shfl(t2, src, 64);
        if (lane == src) {
          ready = 0;  // (not pushed)
          if (d.trace) d.trace[((size_t)s * d.ntasks + t2) * 6 + 0] = -wall_clock64();  // negative: kept, not pushed
        }
      }
    }
    if (ready) dag_push(d, s, t2);
  }
  if (lane == 0) {
    if (d.gate > 0 && t == d.gate_task && s + d.gate < d.S) {  // the sample `gate` behind this one may start
      const int old = __hip_atomic_fetch_sub(d.pending + (size_t)(s + d.gate) * d.ntasks, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == 1) dag_push(d, s + d.gate, 0);
    }
    __hip_atomic_fetch_sub(&d.ctl->remaining, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return kept < 0 ? -1 : s * d.ntasks + kept;
}

// Pop for one workgroup (called by wave 0; the result is valid in every lane of it): an entry, or -1 when the graph
// is done or aborted.  `urgent`: the urgent ring this workgroup serves or -1; `home`: its bulk ring (the leaf ring for
// a leaf server); `steal`: it may take from the other bulk rings when its own rings are empty.
// An idle workgroup polls TWO lines (its rings' heads and tails ride with the status words in one gather) and backs
// off; the other bulk rings are looked at every eighth idle round only.
__device__ __forceinline__ int dag_pop(const DagDev& d, int urgent, int home, bool steal) {
  const int lane = threadIdx.x & 63;
  const long long t0 = wall_clock64();
  const int crit = steal ? home - DAG_RING_BULK0 + DAG_RING_CRIT0 : -1;  // (GEMM workers; a leaf server / a chain-only worker has its one ring)
  int idle = 0;
  for (;;) {
    const bool wide = steal && (idle & 7) == 7;
    // lane r < DAG_NRINGS: (head, tail) of ring r -- only the lanes of the rings looked at in this round load;
    // lanes DAG_NRINGS, DAG_NRINGS + 1: the two status words
    int h = 0, tl = 0;
    const bool look = (urgent >= 0 && lane == urgent) || (crit >= 0 && lane == crit) || (home >= 0 && lane == home) ||
                      (wide && lane >= DAG_RING_CRIT0 && lane < DAG_RING_LEAF && (lane - DAG_RING_CRIT0) % NQ < d.nteams);
    if (look) {
      const unsigned long long v = __hip_atomic_load(&d.ctl->ring[lane].ht, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      h = (int)(unsigned)v;
      tl = (int)(unsigned)(v >> 32);
    } else if (lane == DAG_NRINGS) {
      h = dag_ld(&d.ctl->remaining);
    } else if (lane == DAG_NRINGS + 1) {
      h = dag_ld(&d.ctl->abort);
    }
    const int remaining = __shfl(h, DAG_NRINGS, 64), aborted = __shfl(h, DAG_NRINGS + 1, 64);
    if (remaining <= 0 || aborted) return -1;
    const unsigned long long ne = __ballot(look && h < tl);
    if (ne) {
      int ring;
      if (urgent >= 0 && ((ne >> urgent) & 1ull))
        ring = urgent;
      else if (crit >= 0 && ((ne >> crit) & 1ull))
        ring = crit;
      else if (home >= 0 && ((ne >> home) & 1ull))
        ring = home;
      else {
        // another team's rings: its crit ring before any bulk ring; the search starts behind the own team, so that
        // thieves spread out
        const unsigned long long cm = (ne >> DAG_RING_CRIT0) & ((1ull << NQ) - 1), bm = (ne >> DAG_RING_BULK0) & ((1ull << NQ) - 1);
        const int me = home - DAG_RING_BULK0;
        const unsigned long long pick = cm ? cm : bm;
        const unsigned long long rot = ((pick >> me) | (pick << (NQ - me))) & ((1ull << NQ) - 1);
        ring = (cm ? DAG_RING_CRIT0 : DAG_RING_BULK0) + (me + __builtin_ctzll(rot)) % NQ;
      }
      const int hh = __shfl(h, ring, 64);
      int got = 0;
      if (lane == 0) {
        int expect = hh;
        got = __hip_atomic_compare_exchange_strong(d.ctl->ring[ring].head(), &expect, hh + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT)
                  ? 1
                  : 0;
      }
      got = __shfl(got, 0, 64);
      if (got) {
        int e = 0;
        if (lane == 0) {
          const int* slot = d.slots + d.ring_base[ring] + hh;
          int n = 0;
          while ((e = dag_ld(slot)) == 0) {  // the tail was advanced, the slot store is on its way
            __builtin_amdgcn_s_sleep(1);
            if (++n > (1 << 22)) {
              __hip_atomic_store(&d.ctl->abort, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              e = -1;
              break;
            }
          }
        }
        e = __shfl(e, 0, 64);
        return e > 0 ? e - 1 : -1;
      }
      continue;  // lost the race for that head: look again
    }
    // nothing: back off.  Every poll is a coherent load that goes to the memory fabric, and the rings' lines live in ONE
    // channel: 430 idle workgroups polling every microsecond slowed EVERY memory access of the working ones (a 64-tile
    // task of k = 128 ran 15-40 us instead of 5, a leaf 40 instead of 22).  Fast for a few rounds after a task (the
    // next one of a chain comes within microseconds), then about 1 us for the workgroups that serve an urgent or the
    // leaf ring and about 14 us for the others (bulk tasks run for hundreds of microseconds).
    ++idle;
    if (idle < 4) {
      __builtin_amdgcn_s_sleep(2);
    } else if (idle < 24) {
      __builtin_amdgcn_s_sleep(16);
    } else if (urgent >= 0 || !steal) {
      __builtin_amdgcn_s_sleep(40);
    } else {
      __builtin_amdgcn_s_sleep(127);
      __builtin_amdgcn_s_sleep(127);
      __builtin_amdgcn_s_sleep(127);
      __builtin_amdgcn_s_sleep(127);
    }
    if ((idle & 63) == 0) {
      const long long waited = wall_clock64() - t0;
      if (waited > d.timeout_ticks) {
        if (lane == 0) __hip_atomic_store(&d.ctl->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return -1;
      }
      // The leaf servers must be resident beside the workers.  If a millisecond into the launch not one of them has
      // started while nothing has been computed yet, the runtime has serialised the two launches (DESIGN.md section 3
      // step 19): give up at once -- the stream-ordered schedule answers -- instead of waiting out the long time-out.
      if (steal && d.leaf_servers > 0 && waited > 100000 && dag_ld(&d.ctl->leaf_alive) == 0 &&
          dag_ld(&d.ctl->remaining) == d.S * d.ntasks) {
        if (lane == 0) __hip_atomic_store(&d.ctl->abort, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return -1;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void dag_worker_kernel(DagDev d) {
  __shared__ __attribute__((aligned(16))) T smem[4 * opsz_of<T>(128)];
  __shared__ int cur;
  if (d.rsv && cu_reserve_bail(d.rsv, d.ctl->reserve_ctr)) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int x = (int)(xcc & (NQ - 1));
  int urgent = x < d.nteams ? DAG_RING_URGENT + x : -1;
  const int home = DAG_RING_BULK0 + x % d.nteams;
  bool bulk_too = true;
  if (d.urgent_cus > 0 && urgent >= 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const bool chain_cu = (int)((hw >> 8) & 0xf) < d.urgent_cus;
    if (chain_cu)
      bulk_too = false;  // this workgroup serves the urgent ring only
    else
      urgent = -1;       // and the others leave it alone
  }
  if (threadIdx.x == 0) atomicCAS(&d.ctl->pad[1], 0, (int)(wall_clock64() / 100));
  int kept = -1;  // (wave 0) the task this workgroup made ready and keeps for itself
  long long c_pop = 0, c_acq = 0, c_exec = 0, c_done = 0, n_task = 0, n_kept = 0;
  const long long c_start = wall_clock64();
  for (;;) {
    long long c0 = wall_clock64();
    if (threadIdx.x < 64) {
      n_kept += kept >= 0;
      const int e = kept >= 0 ? kept : dag_pop(d, urgent, bulk_too ? home : -1, bulk_too);
      c_pop += wall_clock64() - c0;
      c0 = wall_clock64();
      if (d.trace && threadIdx.x == 0 && e >= 0) d.trace[(size_t)e * 6 + 2] = c0;
      // ONE agent-scope acquire for everything the predecessors stored (invalidates this CU's L1), waited for before
      // the barrier that releases the other waves
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) cur = e;
      c_acq += wall_clock64() - c0;
    }
    __syncthreads();
    const int e = __builtin_amdgcn_readfirstlane(cur);
    __syncthreads();  // everyone holds e before wave 0 may overwrite it; also fences the LDS stages of the last tile
    if (e < 0) {
      if (threadIdx.x == 0) {
        atomicCAS(&d.ctl->pad[2], 0, (int)(wall_clock64() / 100));
        atomicAdd(&d.ctl->t_pop, (unsigned long long)c_pop);
        atomicAdd(&d.ctl->t_acq, (unsigned long long)c_acq);
        atomicAdd(&d.ctl->t_exec, (unsigned long long)c_exec);
        atomicAdd(&d.ctl->t_done, (unsigned long long)c_done);
        atomicAdd(&d.ctl->n_tasks, (unsigned long long)n_task);
        atomicAdd(&d.ctl->n_kept, (unsigned long long)n_kept);
        atomicAdd(&d.ctl->n_workers, 1ull);
        atomicAdd(&d.ctl->t_life, (unsigned long long)(wall_clock64() - c_start));
      }
      return;
    }
    c0 = wall_clock64();
    ++n_task;
    if (d.trace && threadIdx.x == 0) {
      long long* tr = d.trace + (size_t)e * 6;
      tr[1] = c0 - 0;  // popped + acquired + both barriers passed
      tr[5] = (long long)blockIdx.x | ((long long)x << 32);
    }
    const int s = e / d.ntasks, t = e - s * d.ntasks;
    const DagTask tk = d.tasks[t];
    const GemmArgs g = d.launches[__builtin_amdgcn_readfirstlane(tk.launch)];
    const int bx = __builtin_amdgcn_readfirstlane(tk.bx);
    switch (__builtin_amdgcn_readfirstlane(tk.kind)) {
      case 0: gemm_tile<T, false, false, 128, 4, DAG_HO>(g, bx, s, smem); break;
      case 1: gemm_tile<T, false, false, 64, 4, DAG_HO>(g, bx, s, smem); break;
      case 2: gemm_tile<T, false, true, 128, 4, DAG_HO>(g, bx, s, smem); break;
      case 3: gemm_tile<T, false, true, 64, 4, DAG_HO>(g, bx, s, smem); break;
      case 6: gemm_tile<T, true, true, 128, 4, DAG_HO>(g, bx, s, smem); break;
      case 7: gemm_tile<T, true, true, 64, 4, DAG_HO>(g, bx, s, smem); break;
      default: break;  // (no other kind is ever routed to a worker ring)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have left
    __syncthreads();
    c_exec += wall_clock64() - c0;
    c0 = wall_clock64();
    if (d.trace && threadIdx.x == 0) d.trace[(size_t)e * 6 + 3] = c0;
    if (threadIdx.x < 64) kept = dag_complete(d, s, t, bulk_too ? 1 : 2);
    c_done += wall_clock64() - c0;
    if (d.trace && threadIdx.x == 0) d.trace[(size_t)e * 6 + 4] = wall_clock64();
  }
}

template <typename T>
__global__ __launch_bounds__(256, 1) void dag_leaf_kernel(DagDev d, int fault) {
  __shared__ leaf5::Shared<T> sh;
  __shared__ int cur;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&d.ctl->leaf_alive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    d.ctl->pad[0] = (int)(wall_clock64() / 100);  // diagnostics (GPC_DAG_LOG): microseconds, when a leaf server started
    if (blockIdx.x < 4) {
      const unsigned long long v = __hip_atomic_load(&d.ctl->ring[DAG_RING_LEAF].ht, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      d.ctl->pad[4 + 4 * blockIdx.x] = (int)(unsigned)v;
      d.ctl->pad[5 + 4 * blockIdx.x] = (int)(unsigned)(v >> 32);
      d.ctl->pad[6 + 4 * blockIdx.x] = dag_ld(&d.ctl->remaining);
    }
  }
  for (;;) {
    if (threadIdx.x < 64) {
      const int e = dag_pop(d, -1, DAG_RING_LEAF, false);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) cur = e;
    }
    __syncthreads();
    const int e = __builtin_amdgcn_readfirstlane(cur);
    __syncthreads();
    if (e < 0) {
      if (threadIdx.x == 0 && blockIdx.x < 4) d.ctl->pad[7 + 4 * blockIdx.x] = (int)(wall_clock64() / 100);
      return;
    }
    const int s = e / d.ntasks, t = e - s * d.ntasks;
    if (d.trace && threadIdx.x == 0) {
      d.trace[(size_t)e * 6 + 1] = wall_clock64();
      d.trace[(size_t)e * 6 + 2] = d.trace[(size_t)e * 6 + 1];
      d.trace[(size_t)e * 6 + 5] = -1 - (long long)blockIdx.x;
    }
    const int off = __builtin_amdgcn_readfirstlane(d.tasks[t].launch);
    T* Ab = reinterpret_cast<T*>(d.A) + (size_t)s * d.sA + (size_t)off * d.npad + off;
    T* Wb = reinterpret_cast<T*>(d.W) + (size_t)s * d.sW + (size_t)off * d.npad + off;
    leaf5_body<T>(sh, Ab, d.npad, Wb, d.npad, off, d.logdet + s, d.info + s, max(0, min(TILE, d.nvalid - off)), fault);
    // plain stores: every storing wave drains, the workgroup meets, one agent-scope release writes the L2 back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (d.trace && threadIdx.x == 0) d.trace[(size_t)e * 6 + 3] = wall_clock64();
    if (threadIdx.x < 64) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      (void)dag_complete(d, s, t, 0);
      if (d.trace && threadIdx.x == 0) d.trace[(size_t)e * 6 + 4] = wall_clock64();
    }
  }
}

// pending[s][t] = npred[t]; tasks without predecessors are pushed (the first leaf of every sample)
__global__ __launch_bounds__(256) void dag_init_kernel(DagDev d) {
  const long long n = (long long)d.S * d.ntasks;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int s = (int)(i / d.ntasks), t = (int)(i - (long long)s * d.ntasks);
    const int np = d.tasks[t].npred;
    // (only task 0, the first leaf, is gated: its release by sample s - gate re-pushes TASK 0 -- any other root task of a
    // plan would never be released)
    const bool gated = t == 0 && np == 0 && d.gate > 0 && s >= d.gate;
    d.pending[i] = np + (gated ? 1 : 0);
    if (np == 0 && !gated) dag_push(d, s, t);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) d.ctl->remaining = (int)n;
  // the launch table with this run's buffer addresses
  for (int i = blockIdx.x * 256 + threadIdx.x; i < d.nlaunch; i += gridDim.x * 256) {
    GemmArgs g = d.launches_rel[i];
    auto abs = [&](const void* p) -> void* {
      const unsigned long long v = (unsigned long long)p;
      return static_cast<char*>(d.base[(v >> 48) - 1]) + (v & 0xffffffffffffull);
    };
    g.A = abs(g.A);
    g.B = abs(g.B);
    g.C = abs(g.C);
    d.launches_w[i] = g;
  }
}

inline const void* dag_fake_base(int buf) { return reinterpret_cast<const void*>((unsigned long long)(buf + 1) << 48); }

}  // namespace gpc
