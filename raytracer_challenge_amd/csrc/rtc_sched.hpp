// rtc_sched.hpp — wf_tq: the traversal kernel of the wavefront path as a per-wave SECTION SCHEDULER (round 3; included by
// rtc_feat.hip after rtc_device.hpp, whose device functions it is made of).
//
// wf_ts hands a wave 64 work items and lets every lane run its item's whole traversal: a wave lasts as long as its longest item and
// executes a BVH node step with a third of its lanes (most traversals end at the root or after one leaf, a few walk on).  Here a lane
// is a small state machine — IDLE -> PRO (load the item, everything in front of the walk: planes, quirk scans, gate, frame, the root's
// four boxes) -> WALK (at an inner node or at a leaf) -> EPI (write the answer) -> IDLE — and the wave runs ONE section per
// iteration for all lanes that wait for it: the section with the most waiting lanes (node steps are cheap and weigh four times; a
// section that has waited eight iterations goes first).  A lane that finishes is refilled from the wave's pool (blocks of 64 items
// from the per-XCD cursors wf_ts uses) without waiting for its neighbours, so no lane idles behind the longest walk of a chunk and a
// launch has no chunk-sized tail.  Same device functions, same operands, same acceptance rule per intersection: bit-identical
// hits (the parity tests run this kernel when RTC_WF_SCHED=1).
//
// Scope: kernel-argument programs (variants 0 and 1) with at most ONE walk op (OP_BVH / OP_MESH) whose root travels in the kernel
// arguments — every other op of such a program runs in PRO, in program order, the walk last (test order is free: ties are broken
// by key).  Roles as in wf_ts: trace (closest hit of the level's rays), then the shadow + Phong pass of the previous level's shade
// records.  A transparent hit's container pass does not run where it is found: the lane notes the item in a per-wave LDS list and
// goes on; the list is worked off as its own role (MODE_CONTAINERS for every lane) when it holds a wave's worth or the trace items are gone.
#pragma once

#ifndef RTC_SCHED_W_NODE
#define RTC_SCHED_W_NODE 4   // weight of a lane waiting at an inner node (cheap section)
#endif
#ifndef RTC_SCHED_W_EPI
#define RTC_SCHED_W_EPI 2
#endif
#ifndef RTC_SCHED_AGE
#define RTC_SCHED_AGE 8      // iterations a non-empty section may be passed over
#endif
#ifndef RTC_SCHED_MIN
#define RTC_SCHED_MIN 48     // while some lane walks, prologue / epilogue / Phong sections wait until this many lanes want them
#endif
#define RTC_CQ_ENTRIES 128   // per-wave list of items that wait for their container pass (flushed at 64)
#define RTC_CQ_BYTES (RTC_CQ_ENTRIES * 4)

#if defined(RTC_EMU)
#define RTC_WAVE_SYNC() do {} while (0)
#else
#define RTC_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

namespace {

enum { LS_IDLE = 0, LS_PRO = 1, LS_WALK = 2, LS_EPI = 3, LS_PHONG = 4 };
enum { ROLE_NONE = 0, ROLE_TRACE = 1, ROLE_SHADOW = 2, ROLE_CONT = 3 };

// A wave's supply of items: one block of up to 64 consecutive items of a role, handed to idle lanes rank by rank.
struct Pool {
  unsigned base, pos, cnt;  // items base + pos .. base + cnt - 1 are still to be handed out
  int role;                 // role of the block (ROLE_NONE: nothing fetched yet / stream exhausted)
  bool exhausted;           // the cursor ran past the last block
};

struct LaneCtx {
  int st, item;
  Ray ray, o;          // world-space ray of the pass; mesh walks: the object-space ray
  Trav T;
  Frame32 F;
  float lo, hi;        // the pass's t interval in f32 (t_interval32)
  int cur, sp;
  int light;           // shadow role: light of the current pass
  unsigned long long mask;
};

// true: the scene's kernel-argument program fits wf_tq (host and device agree through this one function)
static inline bool rtc_sched_fits(const DScene& S) {
  if (S.n_kops <= 0) return false;
  int walks = 0;
  for (int pc = 0; pc < S.n_kops; pc++)
    if (S.kops[pc].op == OP_BVH || S.kops[pc].op == OP_MESH) { walks++; if (S.kops[pc].pad[0] < 0 || S.kops[pc].a < 0) return false; }
  return walks <= 1;
}

// Next block of the launch's stream (trace blocks first, then shadow blocks) for this wave's XCD.
__device__ __forceinline__ void pool_fetch(Pool& P, unsigned* next, unsigned nx, unsigned x, unsigned ct, unsigned cs, unsigned nt, unsigned ns, int lane) {
  unsigned kx = 0;
  if (lane == 0) kx = atomicAdd(next, 1u);
  kx = __shfl(kx, 0);
  const unsigned long long blk = (unsigned long long)kx * nx + x;
  P.pos = 0;
  if (blk >= (unsigned long long)ct + cs) { P.exhausted = true; P.role = ROLE_NONE; P.cnt = 0; P.base = 0; return; }
  if (blk < ct) { P.role = ROLE_TRACE; P.base = (unsigned)blk * 64u; P.cnt = nt - P.base < 64u ? nt - P.base : 64u; }
  else { P.role = ROLE_SHADOW; P.base = (unsigned)(blk - ct) * 64u; P.cnt = ns - P.base < 64u ? ns - P.base : 64u; }
}

// Everything of a pass in front of its walk: the program's non-walk ops in order, then the walk op's gate, frame and root node.
// Leaves L.st = LS_WALK (cur, sp, F, o set) or LS_EPI (no walk / nothing left to walk).
template <int FEAT, int MODE, bool LDSC>
__device__ __forceinline__ void sched_begin_pass(const DScene& S, LaneCtx& L, Counters& C, int* __restrict__ stack, int stride, const LdsScene& LD) {
  Trav& T = L.T;
  const Ray& r = L.ray;
  T.mode = MODE;
  T.cubes_in_leaf = 0;
  if (S.quirk_reach2 > 0.0) {
    const double m = fmax(fmax(fabs(r.ox - S.abvh_frame[0]), fabs(r.oy - S.abvh_frame[1])), fabs(r.oz - S.abvh_frame[2])) + S.abvh_frame[3];
    const double len2 = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
    T.cubes_in_leaf = (len2 >= 0.0025 && m * m <= S.quirk_reach2 * len2) ? 1 : 0;
  }
  const LightCell lcell = light_grid_fetch<MODE>(S, r, T);
  int wpc = -1;
  for (int pc = 0; pc < S.n_kops; pc++) {
    const DOp op = S.kops[pc];
    if (op.op == OP_BVH || op.op == OP_MESH) { wpc = pc; continue; }
    if (op.op == OP_PRIM) {
      if (op.c >= 0) {
        const DPlaneK P = S.kplanes[op.c];
        C.analytic_tests++;
        C.kplanes++;
        const double oy = P.row[0] * r.ox + P.row[1] * r.oy + P.row[2] * r.oz + P.row[3] * 1.0;
        const double dy = P.row[0] * r.dx + P.row[1] * r.dy + P.row[2] * r.dz + P.row[3] * 0.0;
        if (!(fabs(dy - 0.0) < EPS) && !plane_behind(T, oy, dy)) {
          double t = -oy / dy;
          accept(T, C, P.prim, 1, &t);
        }
      } else {
        visit_prim<FEAT, LDSC>(S, op.a, r, T, C, 0, LD);
      }
    } else if ((op.op == OP_QUIRK || op.op == OP_QGRID) && op.c == 1 && T.cubes_in_leaf) {
      // the cubes' quirk scan: this ray's leaves test them
    } else if (op.op == OP_QUIRK) {
      for (int i = op.a; i < op.a + op.b; i++) visit_prim<FEAT, LDSC>(S, S.quirk_prim[i], r, T, C, 2, LD);
    } else if (op.op == OP_QGRID) {
      quirk_grid_scan<FEAT, LDSC>(S, op.pad[0] >= 0 ? S.kqgrid : S.qgrids[op.a], r, T, C, LD);
    }
  }
  L.st = LS_EPI;
  if (MODE == MODE_SHADOW_ANY && T.shadowed) return;
  if (wpc < 0) return;
  const DOp op = S.kops[wpc];
  const bool mesh = op.op == OP_MESH;
  if (mesh && FEAT != 0 && op.g >= 0 && !kops_gate(S, op.g, r, T, C)) return;
  int cur = op.a, sp = 0;
  L.o = r;
  if (!mesh && light_grid_candidates(S, lcell, T, C, cur, sp, stack, stride)) {
    if (cur == RTC_WALK_END) return;
  } else {
    const DKAux& A = S.kaux[op.pad[0]];
    if (mesh) L.o = to_object(A.xf, r);
    make_frame(A.frame, L.o, L.F);
    if (MODE == MODE_CONTAINERS) make_frame_point(A.frame, L.o, T.thi, L.F);
    walk_root_k<MODE == MODE_CONTAINERS>(A.root, L.F, T, C, cur, sp, stack, stride);
    if (cur == RTC_WALK_END) return;
  }
  t_interval32(T, L.lo, L.hi);
  L.cur = cur; L.sp = sp;
  L.st = LS_WALK;
}

// One role of one wave: refill, then one section per iteration, until the role's items are gone and every lane is idle.
// ROLE_TRACE returns early (all lanes idle) when the container list holds a wave's worth; the caller works the list off and comes back.
template <bool COUNT, int FEAT, int MODE, int ROLE, bool LDSC>
__device__ __forceinline__ void sched_run(const DScene& S, const DCamera& cam, const DPixelMap& pm, const DWave& W, const WorkMap& wm, int level, Pool& P, unsigned* next,
                                          unsigned nx, unsigned x, unsigned ct, unsigned cs, unsigned nt, unsigned ns, unsigned* __restrict__ cq, unsigned& cq_n, int fuel_left,
                                          double* __restrict__ hit_t, int* __restrict__ hit_prim, int* __restrict__ hit_k, int* __restrict__ stack, int stride,
                                          const LdsScene& LD, Counters& C, unsigned& n_rays, unsigned& n_container, unsigned& n_shadow) {
  const int lane = RTC_LANE_ID;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const size_t cap = W.cap;
  const bool mesh = S.has_mesh != 0;   // (at most one walk op: its kind is the scene's)
  const bool any_hit = MODE == MODE_SHADOW_ANY;
  LaneCtx L;
  L.st = LS_IDLE; L.item = -1; L.cur = RTC_WALK_END; L.sp = 0; L.light = 0; L.mask = 0ull; L.lo = 0.0f; L.hi = 0.0f;
  unsigned cq_pos = 0;   // ROLE_CONT: entries of the list handed out so far
  int age_p = 0, age_n = 0, age_l = 0, age_e = 0, age_f = 0;
  for (;;) {
    // ---- refill
    const unsigned long long m_idle = __ballot(L.st == LS_IDLE ? 1 : 0);
    bool more;
    if (ROLE == ROLE_CONT) more = cq_pos < cq_n;
    else more = P.role == ROLE && !(ROLE == ROLE_TRACE && cq_n >= 64u);
    if (more && m_idle) {
      if (ROLE == ROLE_CONT) {
        const unsigned rank = (unsigned)__popcll(m_idle & lt), avail = cq_n - cq_pos;
        if (L.st == LS_IDLE && rank < avail) { L.item = (int)cq[cq_pos + rank]; L.st = LS_PRO; }   // (item | key's last push index << 29)
        const unsigned take = (unsigned)__popcll(m_idle);
        cq_pos += take < avail ? take : avail;
      } else {
        if (P.pos == P.cnt) pool_fetch(P, next, nx, x, ct, cs, nt, ns, lane);
        if (P.role == ROLE && P.pos < P.cnt) {
          const unsigned rank = (unsigned)__popcll(m_idle & lt), avail = P.cnt - P.pos;
          if (L.st == LS_IDLE && rank < avail) { L.item = (int)(P.base + P.pos + rank); L.st = LS_PRO; L.light = 0; L.mask = 0ull; }
          const unsigned take = (unsigned)__popcll(m_idle);
          P.pos += take < avail ? take : avail;
        }
      }
    }
    // ---- which section
    const unsigned long long m_p = __ballot(L.st == LS_PRO ? 1 : 0);
    const unsigned long long m_n = __ballot(L.st == LS_WALK && L.cur >= 0 ? 1 : 0);
    const unsigned long long m_l = __ballot(L.st == LS_WALK && L.cur < 0 ? 1 : 0);
    const unsigned long long m_e = __ballot(L.st == LS_EPI ? 1 : 0);
    const unsigned long long m_f = ROLE == ROLE_SHADOW ? __ballot(L.st == LS_PHONG ? 1 : 0) : 0ull;
    if (!(m_p | m_n | m_l | m_e | m_f)) {
      // every lane idle: more items next iteration, or the role is over for this wave
      if (ROLE == ROLE_CONT) { if (cq_pos < cq_n) continue; break; }
      if (ROLE == ROLE_TRACE && cq_n >= 64u) break;
      if (P.role == ROLE && (P.pos < P.cnt || !P.exhausted)) continue;
      break;
    }
    // While lanes walk, the wide sections (prologue, epilogue, Phong: a wave's worth of the same code for every item) wait until
    // RTC_SCHED_MIN lanes want them: they then run nearly full, and the walkers of several refills walk together.  Epilogue and
    // prologue count as one section: a lane that is done is refilled and started in the same iteration.
    const bool walkers = (m_n | m_l) != 0ull;
    const int n_pe = __popcll(m_p | m_e), n_f = __popcll(m_f);
    int s_p = (m_p | m_e) && (!walkers || n_pe >= RTC_SCHED_MIN) ? n_pe + 64 : 0;
    int s_n = m_n ? __popcll(m_n) * RTC_SCHED_W_NODE + age_n * 64 / RTC_SCHED_AGE : 0;
    int s_l = m_l ? __popcll(m_l) + age_l * 64 / RTC_SCHED_AGE : 0;
    int s_e = 0;
    int s_f = m_f && (!walkers || n_f >= RTC_SCHED_MIN) ? n_f + 64 : 0;
    int sec = 0, best = s_n;                       // 0 node, 1 leaf, 2 prologue, 3 epilogue, 4 Phong
    if (s_l > best) { sec = 1; best = s_l; }
    if (s_p > best) { sec = 2; best = s_p; }
    if (s_e > best) { sec = 3; best = s_e; }
    if (s_f > best) { sec = 4; best = s_f; }
    age_n = (m_n && sec != 0) ? age_n + 1 : 0;
    age_l = (m_l && sec != 1) ? age_l + 1 : 0;
    age_p = (m_p && sec != 2) ? age_p + 1 : 0;
    age_e = (m_e && sec != 3) ? age_e + 1 : 0;
    age_f = (m_f && sec != 4) ? age_f + 1 : 0;

    if (sec == 0) {
      // ---- node step
      if (L.st == LS_WALK && L.cur >= 0) {
        DIAG_LOOP(24);
        C.accel_nodes++;
        float4 lox, loy, loz, hix, hiy, hiz;
        int4 cc;
        if (LDSC) {
          const float4* N = LD.nodes + L.cur;
          const int n = LD.n_nodes;
          lox = N[0]; loy = N[n]; loz = N[2 * n]; hix = N[3 * n]; hiy = N[4 * n]; hiz = N[5 * n];
          cc = as_int4(N[6 * n]);
        } else {
          const DBvhNode4* N = S.bvh + L.cur;
          lox = ld4(N->lox); loy = ld4(N->loy); loz = ld4(N->loz); hix = ld4(N->hix); hiy = ld4(N->hiy); hiz = ld4(N->hiz);
          cc = ld4(N->c);
        }
        node_step<MODE == MODE_CONTAINERS>(lox, loy, loz, hix, hiy, hiz, cc, L.F, L.lo, L.hi, L.cur, L.sp, stack, stride, any_hit);
        if (L.cur == RTC_WALK_END) L.st = LS_EPI;
      }
    } else if (sec == 1) {
      // ---- leaf
      if (L.st == LS_WALK && L.cur < 0) {
        DIAG_LOOP(25);
        const int first = (~L.cur) >> 3, cnt = ((~L.cur) & 7) + 1;
        if (mesh) {
          for (int i = first; i < first + cnt; i++) {
            double t, u, v;
            C.tri_tests++;
            if (LDSC) {
              double g[9];
#pragma unroll
              for (int c = 0; c < 9; c++) g[c] = LD.tris[c * LD.n_tris + i];
              if (tri_hit(g, L.o, t, u, v)) accept(L.T, C, LD.tri_prim[i], 1, &t);
            } else if (tri_hit(S.mtri + 9 * (size_t)i, L.o, t, u, v)) accept(L.T, C, S.mtri_prim[i], 1, &t);
          }
        } else {
          visit_prim<FEAT, LDSC>(S, first, L.ray, L.T, C, 1, LD);
        }
        if (MODE == MODE_SHADOW_ANY && L.T.shadowed) L.st = LS_EPI;
        else {
          if (MODE != MODE_SHADOW_ANY && MODE != MODE_CONTAINERS) t_interval32(L.T, L.lo, L.hi);
          if (L.sp == 0) L.st = LS_EPI;
          else { L.sp--; L.cur = stack[L.sp * stride]; }
        }
      }
    } else if (sec == 2) {
      // ---- epilogue
      bool push = false;
      if (L.st == LS_EPI) {
        DIAG_LOOP(27);
        if (ROLE == ROLE_TRACE) {
          const unsigned i = (unsigned)L.item;
          const Trav& T = L.T;
          int32_t* ch = W.child + (size_t)level * 2 * cap;
          const bool did_hit = T.best_prim != 0x7fffffff;
          if (level == 0 && hit_t) {
            uint64_t q = 0;
            (void)work_to_slot(wm, i, q);
            hit_t[q] = did_hit ? T.best_t : 0.0;
            hit_prim[q] = did_hit ? T.best_prim : -1;
            hit_k[q] = did_hit ? T.best_k : 0;
          }
          W.h_prim[i] = did_hit ? T.best_prim : -1;
          if (COUNT && W.dig != nullptr) {
            unsigned long long tb = 0ull;
            if (did_hit) __builtin_memcpy(&tb, &T.best_t, 8);
            W.dig[(size_t)level * cap + i] = rtc_hit_hash_base(tb, did_hit ? T.best_prim : -1, did_hit ? T.best_k : 0);
          }
          ch[i] = did_hit ? -1 : RTC_WF_MISS; ch[cap + i] = -1;
          if (did_hit) {
            W.h_t[i] = T.best_t;
            if (fuel_left > 0 && S.mat[8 * S.prims[T.best_prim].mat + 5] != 0.0) push = true;
          }
          L.st = LS_IDLE;
        } else if (ROLE == ROLE_CONT) {
          const unsigned i = (unsigned)L.item;
          double n1 = 1.0, n2 = 1.0;
          if (L.T.c1_prim >= 0) n1 = S.mat[8 * S.prims[L.T.c1_prim].mat + 6];
          if (L.T.c2_prim >= 0) n2 = S.mat[8 * S.prims[L.T.c2_prim].mat + 6];
          W.h_n12[i] = n1; W.h_n12[cap + i] = n2;
          L.st = LS_IDLE;
        } else {
          bool shadowed;
          if (MODE == MODE_SHADOW_ANY) shadowed = L.T.shadowed != 0;
          else shadowed = (L.T.best_prim != 0x7fffffff) && (S.prims[L.T.best_prim].flags & 1u) && (L.T.best_t < L.T.c1_t);
          if (shadowed) L.mask |= 1ull << L.light;
          L.light++;
          L.st = L.light < S.n_lights ? LS_PRO : LS_PHONG;
        }
      }
      if (ROLE == ROLE_TRACE) {
        // transparent hits: noted for the container role (wave-wide append; cq_n is the same number in every lane)
        const unsigned long long m_push = __ballot(push ? 1 : 0);
        // entry: item | the key's last push index << 29 (a primitive pushes at most four intersections; the host only picks this
        // kernel for launches of fewer than 2^29 work ids)
        if (push) cq[cq_n + (unsigned)__popcll(m_push & lt)] = (unsigned)L.item | ((unsigned)L.T.best_klast << 29);
        cq_n += (unsigned)__popcll(m_push);
        RTC_WAVE_SYNC();
      }
      // ---- refill the lanes the epilogue just freed (same rule as at the top of the loop)
      if (ROLE != ROLE_CONT) {
        const unsigned long long m_idle2 = __ballot(L.st == LS_IDLE ? 1 : 0);
        const bool more2 = P.role == ROLE && !(ROLE == ROLE_TRACE && cq_n >= 64u);
        if (more2 && m_idle2) {
          if (P.pos == P.cnt) pool_fetch(P, next, nx, x, ct, cs, nt, ns, lane);
          if (P.role == ROLE && P.pos < P.cnt) {
            const unsigned rank = (unsigned)__popcll(m_idle2 & lt), avail = P.cnt - P.pos;
            if (L.st == LS_IDLE && rank < avail) { L.item = (int)(P.base + P.pos + rank); L.st = LS_PRO; L.light = 0; L.mask = 0ull; }
            const unsigned take = (unsigned)__popcll(m_idle2);
            P.pos += take < avail ? take : avail;
          }
        }
      } else {
        const unsigned long long m_idle2 = __ballot(L.st == LS_IDLE ? 1 : 0);
        if (cq_pos < cq_n && m_idle2) {
          const unsigned rank = (unsigned)__popcll(m_idle2 & lt), avail = cq_n - cq_pos;
          if (L.st == LS_IDLE && rank < avail) { L.item = (int)cq[cq_pos + rank]; L.st = LS_PRO; }
          const unsigned take = (unsigned)__popcll(m_idle2);
          cq_pos += take < avail ? take : avail;
        }
      }
      // ---- prologue: the item's (next) ray, then everything in front of its walk
      if (L.st == LS_PRO) {
        DIAG_LOOP(26);
        bool go = true;
        if (ROLE == ROLE_TRACE || ROLE == ROLE_CONT) {
          const int klast = ROLE == ROLE_CONT ? (int)((unsigned)L.item >> 29) : 0;
          if (ROLE == ROLE_CONT) L.item &= 0x1fffffff;
          const unsigned i = (unsigned)L.item;
          if (level == 0) {
            uint64_t q = 0;
            if (!work_to_slot(wm, i, q)) { W.h_prim[i] = -1; L.st = LS_IDLE; go = false; }  // tile padding: no pixel gathers this id
            else L.ray = slot_ray(pm, cam, q);
          } else {
            double w_;
            L.ray = wf_load_ray(W, level, i, w_);
          }
          if (go) {
            reset_closest(L.T, MODE);
            if (ROLE == ROLE_TRACE) n_rays++;
            else {
              // the hit's key, as the closest pass left it (wf_trace_ray keeps it in registers; here it went through memory)
              n_container++;
              L.T.tlo = -DINF; L.T.thi = W.h_t[i]; L.T.best_t = L.T.thi;
              L.T.best_prim = W.h_prim[i]; L.T.best_klast = klast; L.T.best_k = klast;
            }
          }
        } else {
          // shadow role: the record's point and light L.light
          const unsigned s = (unsigned)L.item;
          const double* rr = W.sr;
          const double px = rr[s], py = rr[cap + s], pz = rr[2 * cap + s];
          const double* LG = S.lights + 6 * L.light;
          const double vx = LG[3] - px, vy = LG[4] - py, vz = LG[5] - pz;
          const double distance = sqrt(vx * vx + vy * vy + vz * vz);
          L.ray.ox = px; L.ray.oy = py; L.ray.oz = pz;
          L.ray.dx = vx / distance; L.ray.dy = vy / distance; L.ray.dz = vz / distance;
          n_shadow++;
          reset_closest(L.T, MODE);
          if (MODE == MODE_SHADOW_ANY) { L.T.thi = distance; L.T.unordered = 1; }
          L.T.light = L.light; L.T.c1_t = distance;
        }
        if (go) sched_begin_pass<FEAT, MODE, LDSC>(S, L, C, stack, stride, LD);
      }
    } else {
      // ---- Phong terms of a shade record whose shadow rays are all in (wf_shadow_rec's second half)
      if (ROLE == ROLE_SHADOW && L.st == LS_PHONG) {
        DIAG_LOOP(28);
        const unsigned s = (unsigned)L.item;
        double* cb = W.contrib + (size_t)level * 3 * cap;
        const double* r = W.sr;
        const double px = L.ray.ox, py = L.ray.oy, pz = L.ray.oz;   // the record's point: every shadow ray started there
        const double nx_ = r[3 * cap + s], ny_ = r[4 * cap + s], nz_ = r[5 * cap + s];
        const double cr = r[6 * cap + s], cg = r[7 * cap + s], cbl = r[8 * cap + s];
        const int node = W.sr_node[s];
        double ex, ey, ez, weight;
        if (level == 0) {
          uint64_t q = 0;
          (void)work_to_slot(wm, (unsigned)node, q);
          const Ray pr = slot_ray(pm, cam, q);
          ex = -pr.dx; ey = -pr.dy; ez = -pr.dz; weight = 1.0;
        } else {
          const double* rq = W.rq[level & 1];
          ex = -rq[3 * cap + node]; ey = -rq[4 * cap + node]; ez = -rq[5 * cap + node]; weight = rq[6 * cap + node];
        }
        const double* M = S.mat + 8 * W.sr_mat[s];
        const double ambient = M[0], diffuse = M[1], specular = M[2], shininess = M[3];
        double sr = 0.0, sg = 0.0, sb = 0.0;
        for (int l = 0; l < S.n_lights; l++) {
          const double* LG = S.lights + 6 * l;
          const double vx = LG[3] - px, vy = LG[4] - py, vz = LG[5] - pz;
          const double distance = sqrt(vx * vx + vy * vy + vz * vz);
          const double ldx = vx / distance, ldy = vy / distance, ldz = vz / distance;
          const bool shadowed = (L.mask >> l) & 1ull;
          const double er = cr * LG[0], eg = cg * LG[1], eb = cbl * LG[2];
          const double lr = er * ambient, lg = eg * ambient, lb = eb * ambient;
          const double ldn = ldx * nx_ + ldy * ny_ + ldz * nz_;
          double dr = 0.0, dg = 0.0, db = 0.0, pr = 0.0, pg = 0.0, pb = 0.0;
          if (!shadowed && ldn >= 0.0) {
            dr = er * diffuse * ldn; dg = eg * diffuse * ldn; db = eb * diffuse * ldn;
            const double mlx = -ldx, mly = -ldy, mlz = -ldz;
            const double d2 = 2.0 * (mlx * nx_ + mly * ny_ + mlz * nz_);
            const double rfx = mlx - nx_ * d2, rfy = mly - ny_ * d2, rfz = mlz - nz_ * d2;
            const double rde = rfx * ex + rfy * ey + rfz * ez;
            if (rde > 0.0) {
              const double f = rtc_pow(rde, shininess);
              pr = LG[0] * specular * f; pg = LG[1] * specular * f; pb = LG[2] * specular * f;
            }
          }
          sr += (lr + dr) + pr; sg += (lg + dg) + pg; sb += (lb + db) + pb;
        }
        cb[node] = weight * sr; cb[cap + node] = weight * sg; cb[2 * cap + node] = weight * sb;
        L.st = LS_IDLE;
      }
    }
  }
}

}  // namespace

// The kernel: same arguments, grid and cursors as wf_ts; dynamic LDS = [scene tables (LDSC)] [traversal stacks] [container lists].
template <bool COUNT, int FEAT, bool LDSC = false>
__global__ void __launch_bounds__(LDSC ? RTC_LDS_BLOCK : RTC_BLOCK, RTC_WF_TS_WAVES) wf_tq(DScene S, DCamera cam, DPixelMap pm, DWave W, int tl, int sl, unsigned n0, int slot, int fuel_left,
                                                                                      double* __restrict__ hit_t, int* __restrict__ hit_prim, int* __restrict__ hit_k, DStats* __restrict__ stats) {
  RTC_LDS_STACK(lds_stack);
  LdsScene LD = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
  int* stack = lds_stack + threadIdx.x;
  int stride = RTC_BLOCK;
#ifdef RTC_EMU
  static unsigned cq_store[RTC_CQ_ENTRIES * (RTC_BLOCK >= 64 ? RTC_BLOCK / 64 : 1)];
  unsigned* cq = cq_store;
#else
  unsigned* cq;
  if (LDSC) {
    int* base = (int*)lds_fill(S, (char*)lds_stack, LD);
    stack = base + threadIdx.x;
    stride = (int)blockDim.x;
    cq = (unsigned*)(base + (size_t)blockDim.x * (unsigned)S.bvh_stack) + RTC_CQ_ENTRIES * (threadIdx.x >> 6);
  } else {
    cq = (unsigned*)(lds_stack + (size_t)RTC_BLOCK * (unsigned)S.bvh_stack) + RTC_CQ_ENTRIES * (threadIdx.x >> 6);
  }
#endif
  Counters C = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned n_rays = 0, n_container = 0, n_shadow = 0;
#ifdef RTC_DIAG
  if (threadIdx.x < 64) s_diag[threadIdx.x] = 0ull;
  __syncthreads();
#endif
  const WorkMap wm = make_workmap(pm, cam);
  const unsigned nt = tl >= 0 ? wf_count(W, tl, n0) : 0u;
  unsigned ns = sl >= 0 ? W.counts[RTC_WF_SHADE_COUNT + sl] : 0u;
  if (ns > W.cap) ns = W.cap;
  const unsigned ct = (nt + 63u) / 64u, cs = (ns + 63u) / 64u;
  const unsigned nx = gridDim.x < 8u ? gridDim.x : 8u;
  const unsigned x = blockIdx.x % nx;
  unsigned* next = &W.counts[RTC_WF_CHUNK_NEXT + 32 * (8 * slot + (int)x)];
  const int lane = RTC_LANE_ID;
  Pool P = {0u, 0u, 0u, ROLE_NONE, false};
  unsigned cq_n = 0;
  pool_fetch(P, next, nx, x, ct, cs, nt, ns, lane);
  while (P.role == ROLE_TRACE) {
    sched_run<COUNT, FEAT, MODE_CLOSEST, ROLE_TRACE, LDSC>(S, cam, pm, W, wm, tl, P, next, nx, x, ct, cs, nt, ns, cq, cq_n, fuel_left, hit_t, hit_prim, hit_k, stack, stride, LD, C,
                                                           n_rays, n_container, n_shadow);
    if (cq_n) {
      sched_run<COUNT, FEAT, MODE_CONTAINERS, ROLE_CONT, LDSC>(S, cam, pm, W, wm, tl, P, next, nx, x, ct, cs, nt, ns, cq, cq_n, fuel_left, hit_t, hit_prim, hit_k, stack, stride, LD, C,
                                                              n_rays, n_container, n_shadow);
      cq_n = 0;
    }
  }
  if (P.role == ROLE_SHADOW) {
    if (S.all_cast_shadow)
      sched_run<COUNT, FEAT, MODE_SHADOW_ANY, ROLE_SHADOW, LDSC>(S, cam, pm, W, wm, sl, P, next, nx, x, ct, cs, nt, ns, cq, cq_n, fuel_left, hit_t, hit_prim, hit_k, stack, stride, LD, C,
                                                                 n_rays, n_container, n_shadow);
    else
      sched_run<COUNT, FEAT, MODE_SHADOW_CLOSEST, ROLE_SHADOW, LDSC>(S, cam, pm, W, wm, sl, P, next, nx, x, ct, cs, nt, ns, cq, cq_n, fuel_left, hit_t, hit_prim, hit_k, stack, stride, LD,
                                                                     C, n_rays, n_container, n_shadow);
  }
  if (C.nan_ts) atomicAdd(&stats->nan_ts, (unsigned long long)C.nan_ts);
#ifdef RTC_DIAG
  __syncthreads();
  if (threadIdx.x < 64 && s_diag[threadIdx.x]) atomicAdd(&stats->diag[threadIdx.x], s_diag[threadIdx.x]);
#endif
  if (COUNT) {
    if (tl == 0) atomicAdd(&stats->rays_primary, (unsigned long long)n_rays);
    atomicAdd(&stats->rays_container, (unsigned long long)n_container);
    atomicAdd(&stats->rays_shadow, (unsigned long long)n_shadow);
    atomicAdd(&stats->accel_nodes, (unsigned long long)C.accel_nodes);
    atomicAdd(&stats->group_tests, (unsigned long long)C.group_tests);
    atomicAdd(&stats->tri_tests, (unsigned long long)C.tri_tests);
    atomicAdd(&stats->analytic_tests, (unsigned long long)C.analytic_tests);
    atomicAdd(&stats->knodes, (unsigned long long)C.knodes);
    atomicAdd(&stats->kplanes, (unsigned long long)C.kplanes);
    atomicAdd(&stats->light_cells, (unsigned long long)C.light_cells);
    atomicAdd(&stats->kgroups, (unsigned long long)C.kgroups);
  }
}
