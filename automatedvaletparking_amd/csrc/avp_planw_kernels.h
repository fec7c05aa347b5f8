// avp_planw_kernels.h -- batched hybrid-A* planner, the GROUP forms: NW waves = one (start, goal) problem (NW = 1 the
// wave form, 2 the pair form, 4 the quad form), PW_WAVES / NW independent problems per workgroup = per CU, persistent
// groups of waves pull problems from a global counter -- and, when every problem has a workspace slot of its own, hand
// long searches to each other through a resume ring (time slicing, see plan_wave_kernel).
//
// plan_kernel (avp_plan_kernels.h) spends a whole 512-thread workgroup on one problem to shorten a pop's critical
// path: one problem per CU. A pop is a chain of dependent scalar fp64 code, so what a batch larger than the chip needs is
// problems in flight: here a search runs on one, two or four waves -- the same device functions, the same arithmetic in
// the same order, bit-identical results -- with no workgroup barrier after the map tables have been staged. Replaces the
// same reference code as plan_kernel: PathPlanner.a_star_plan (path_plan/path_planner.py:58-110), hybrid_a_star
// (path_plan/hybrid_a_star.py:72-389), Dijkstra (path_plan/compute_h.py), rs_curve.calc_optimal_path
// (path_plan/rs_curve.py:99-680), heapq order.
//
// Structure: every phase of a pop is a CALLED function (pw_ph_*) on two LDS addresses -- the group's state (PwSharedT)
// and the workgroup's constants (PwCommon, with copies of the kernel's arguments). Fully inlined, the pop loop held the
// union of all phases' registers: 256 VGPRs plus spills, two waves per SIMD. Called, each phase has the register demand
// of its own body; the kernel is compiled for 4 waves per SIMD (amdgpu_waves_per_eu propagates to the callees):
// 122 .. 128 VGPRs, no VGPR spill in any phase loop (the scratch the compiler reports is the callee-saved registers a
// phase saves on entry), 16 waves per CU.
//
// What differs from plan_kernel is only WHO does the work:
//   * the collision passes (pl_check_pass), the wave-parallel child resolution (pl_resolve_fast_wave) and the heuristic
//     sweep (with the group as the cooperating set) are the shared device functions;
//   * wave form: the shot is checked BEFORE the children are resolved, as in the reference; pair / quad form: wave 0
//     resolves the children WHILE the other waves sample and check the shot (pw_ph_shot_resolve; a collision-free shot ends
//     the search before expand_node in the reference: the counters the resolution moved are put back, as in plan_kernel);
//   * Reeds-Shepp: only the queries the pop can use are solved (the shot inside flag_radius; a child unless expand_node
//     drops it before calc_node_heuristic), solver group by solver group with the words of one set_path type group in
//     adjacent lanes: the duplicate test (rs_curve.py:137-156) is a few shuffles inside the group and the running
//     arg-min per query (:103-108, "<=": the later word wins a tie) a few LDS words -- no table of all 46 x 11 word
//     results (the 28 KB that keep plan_kernel at one problem per CU);
//   * NW > 1: the independent pieces of a pop are dealt to the group's waves behind a software barrier (an arrival
//     counter in LDS; LDS-only where only LDS is handed over): the collision passes, the solver rounds (drawn from one
//     counter, dearest first), the shot's sample rounds; the sampler's bookkeeping runs beside the segment origins. A pop
//     takes ~0.75 x (pair) / ~0.6 x (quad) the time of the wave form at a half / a quarter of the problems in flight --
//     what a batch of a few problems per CU needs: its long searches all run at once, so their latency IS the launch time.
// A shot with more than PW_RS_CAP samples or a configuration with more than PW_MAXCHILD children is handed back (status
// AVP_PLAN_RETRY, internal) and planned by plan_kernel in a later launch of the same call; the staged call
// (avp_plan_batch_staged) uses the same hand-back for searches that outlive its first stage. Results never depend on the
// kernel that produced them.
#pragma once
#include "avp_plan_kernels.h"

#ifndef PW_WAVES
#define PW_WAVES 16                   // waves per workgroup = per CU: 16 / NW problems in flight per CU (4 waves per SIMD: 128 VGPRs each)
#endif
#define PW_THREADS (64 * PW_WAVES)
#ifndef PW_RS_CAP
#define PW_RS_CAP 192                 // samples of one RS shot held per group (96 m of path; longer shots go to plan_kernel)
#endif
#ifndef PW_WQCAP
#define PW_WQCAP 512                  // (pose, point) candidates of one collision range; a range that overflows is halved
#endif
#define PW_MAXCHILD 16
#define PW_RSQ (PW_MAXCHILD + 1)      // RS queries per pop: the shot + the children
#define AVP_PLAN_RETRY 100            // internal: plan this problem with plan_kernel (never returned to the caller)
// phase timers of the instrumented instantiation (avp_plan_batch_ex, mode 2 | 0x100), written to phase_cycles[0 ..]:
// per-problem set-up, heap pop, children poses + hash look-ups + frames, sub-step collision passes, Reeds-Shepp
// evaluation, the shot (origins, samples, collision passes), the wave-parallel resolution, the serial resolution with
// its sweep extensions, the result record; [9] the number of collision passes, [10] of RS rounds
enum { PW_PH_INIT = 0, PW_PH_POP, PW_PH_CHILD, PW_PH_SUB, PW_PH_RS, PW_PH_SHOT, PW_PH_RESOLVE, PW_PH_SLOW, PW_PH_FINISH, PW_PH_NPASS, PW_PH_NROUND, PW_PH_COUNT = 12 };
#define PW_T(k) do { if constexpr (PROFILE) { if (gtid == 0) { const long long t_ = clock64(); s.phase[k] += (uint32_t)(t_ - t_ph); t_ph = t_; } } } while (0)

// Words of one solver group in lane order: the members of a set_path type group are adjacent and aligned to the group
// size (1, 2 or 4), the groups of a query are adjacent, queries follow each other: item = query * L + j.
//   group of solvers            words (type groups separated by |)                       lanes per query L
//   0 SLS                       0 | 1                                                     2
//   1 LSL                       2 3 | 4 5                                                 4
//   2 LSR                       6 7 | 8 9                                                 4
//   3 LRL                       10 11 14 15 | 12 13 16 17                                 8
//   4 LRLRn + LRLRp             18 19 22 23 | 20 21 24 25                                 8   (the only groups that span two solvers)
//   5 LRSL                      26 27 | 28 29 | 34 35 | 36 37                             8
//   6 LRSR                      30 31 | 32 33 | 38 39 | 40 41                             8
//   7 LRSLR                     42 43 | 44 45                                             4
static __device__ const int8_t PW_SG_L[8] = { 2, 4, 4, 8, 8, 8, 8, 4 };
static __device__ const int8_t PW_SG_SHIFT[8] = { 1, 2, 2, 3, 3, 3, 3, 2 };
static __device__ const int8_t PW_SG_OFF[8] = { 0, 2, 6, 10, 18, 26, 34, 42 };
static __device__ const int8_t PW_SG_GMAX[8] = { 1, 2, 2, 4, 4, 2, 2, 2 };
static __device__ const int8_t PW_ITEM_WORD[46] = { 0, 1,  2, 3, 4, 5,  6, 7, 8, 9,  10, 11, 14, 15, 12, 13, 16, 17,  18, 19, 22, 23, 20, 21, 24, 25,
                                                    26, 27, 28, 29, 34, 35, 36, 37,  30, 31, 32, 33, 38, 39, 40, 41,  42, 43, 44, 45 };
static __device__ const int8_t PW_ITEM_G[46] = { 1, 1,  2, 2, 2, 2,  2, 2, 2, 2,  4, 4, 4, 4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4, 4, 4,
                                                 2, 2, 2, 2, 2, 2, 2, 2,  2, 2, 2, 2, 2, 2, 2, 2,  2, 2, 2, 2 };
static __device__ const int8_t PW_ITEM_M[46] = { 0, 0,  0, 1, 0, 1,  0, 1, 0, 1,  0, 1, 2, 3, 0, 1, 2, 3,  0, 1, 2, 3, 0, 1, 2, 3,
                                                 0, 1, 0, 1, 0, 1, 0, 1,  0, 1, 0, 1, 0, 1, 0, 1,  0, 1, 0, 1 };

// Constants every wave of the workgroup reads (LDS): the lane-indexed members of avp_params (a dynamically indexed
// by-value kernel argument would be copied to every lane's scratch), the sub-step index tables, and COPIES of the
// kernel's arguments -- the phases of a pop are called functions (below) that take two LDS addresses, nothing else.
struct PwCommon {
    double k_steer[AVP_MAX_STEER], k_dth_dt[AVP_MAX_STEER], k_dth_ddt1[AVP_MAX_STEER], k_travel_ddt1;      // (sub-step j: x (j + 1))
    int8_t sub_child[PW_MAXCHILD * 4], sub_j[PW_MAXCHILD * 4], sub_steer[PW_MAXCHILD * 4];
    double sub_dth[PW_MAXCHILD * 4], sub_td[PW_MAXCHILD * 4];   // per sub-step pose t: the heading change ... / lw * ddt * (j + 1) and the signed travel speed * ddt * (j + 1) (hybrid_a_star.py:188-191)
    int8_t item_word[46], item_g[46], item_m[46], sg_l[8], sg_shift[8], sg_off[8], sg_gmax[8];
    PlChkEnv env;                     // what the called collision passes read of the map and the vehicle
    DevMap m;
    avp_params p;
    PlanDims dims;
    const double* starts; const double* goals;
    avp_plan_result_dev* results;
    double* paths; double* trace;
    int32_t max_path, max_trace, maxNodes, nchild, nsubs, retry_only;
    int64_t max_pops;                 // the caller's pop cap (status ITER_LIMIT)
    int64_t stage_pops;               // > 0: a search still running after this many pops is handed back (AVP_PLAN_RETRY)
    unsigned int* deferred;           // ... and counted here
    // time slicing (slice_pops > 0; see plan_wave_kernel): problem i lives in workspace slot i
    int32_t slice_pops, sl_pad;
    int64_t n;
    unsigned int* counter;            // unstarted problems are drawn from it
    unsigned int* sl;                 // {resume tickets served, resume tickets issued, searches finished}
    char* workspace;
};

// The waves that cooperate on one problem. NW = 1: one wave (wave-level syncs only). NW > 1: NW adjacent waves of the
// workgroup behind a software barrier -- a monotonic arrival counter per group and a generation count per wave in
// static LDS (s_barrier would stop the whole workgroup: the groups are independent searches).
__shared__ uint32_t PW_BAR_CNT[PW_WAVES];
__shared__ uint32_t PW_BAR_GEN[PW_WAVES];
__shared__ uint32_t PW_SUB_CNT[PW_WAVES];     // the same for the waves 1 .. NW-1 of a group alone (pw_sub_sync_lds)
__shared__ uint32_t PW_SUB_GEN[PW_WAVES];
#ifndef PW_SLICE_POPS
#define PW_SLICE_POPS 64       // default time slice of a search in pops (avp_plan_set_slice_pops; 32 ... 200, and a shorter first slice: no measurable difference)
#endif
#ifndef PW_ROTATE
#define PW_ROTATE 0             // (measured on the 4 096-problem batch, quad form: 92.6 ms with, 91.8 ms without -- no gain, off)
#endif
template <int NW>
struct PwGroup {
    static constexpr int N = 64 * NW;
    // Role of a wave inside its group (0 = the wave that runs the serial parts: pops, resolution, result record). The
    // hardware places wave k of a workgroup on SIMD k % 4: with PW_ROTATE the role index is rotated by the group number, so
    // that the groups of a CU have their wave 0 on different SIMDs (the serial parts wait on memory, not on the VALU: it
    // makes no measurable difference).
    static __device__ __forceinline__ int wv()
    {
        const int w = threadIdx.x >> 6;
        return (PW_ROTATE && NW > 1) ? ((w + w / NW) & (NW - 1)) : (w & (NW - 1));
    }
    static __device__ __forceinline__ int tid() { return wv() * 64 + (int)(threadIdx.x & 63); }
    // lanes of the group hand data to each other through LDS and through global memory (queues, arena): both must have landed
    static __device__ __forceinline__ void sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if constexpr (NW > 1) {
            if ((threadIdx.x & 63) == 0) {
                const int wave = threadIdx.x >> 6, grp = wave & ~(NW - 1);
                const uint32_t g = PW_BAR_GEN[wave] + 1u;
                PW_BAR_GEN[wave] = g;
                atomicAdd(&PW_BAR_CNT[grp], 1u);
                while (*(volatile uint32_t*)&PW_BAR_CNT[grp] < g * (uint32_t)NW) __builtin_amdgcn_s_sleep(1);
            }
        }
        wave_sync();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // The same for hand-overs through LDS only: the wave's outstanding GLOBAL stores (node records, heap entries, shot
    // samples: a few thousand cycles until acknowledged) are not waited for -- nobody reads them before a full sync().
    static __device__ __forceinline__ void sync_lds()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (NW > 1) {
            if ((threadIdx.x & 63) == 0) {
                const int wave = threadIdx.x >> 6, grp = wave & ~(NW - 1);
                const uint32_t g = PW_BAR_GEN[wave] + 1u;
                PW_BAR_GEN[wave] = g;
                atomicAdd(&PW_BAR_CNT[grp], 1u);
                while (*(volatile uint32_t*)&PW_BAR_CNT[grp] < g * (uint32_t)NW) __builtin_amdgcn_s_sleep(1);
            }
        }
        wave_sync();
    }
};

// LDS-only barrier among the waves 1 .. NW-1 of a group (the shot's checking waves while wave 0 resolves the children).
template <int NW>
AVP_D void pw_sub_sync_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (NW > 2) {
        if ((threadIdx.x & 63) == 0) {
            const int wave = threadIdx.x >> 6, grp = wave & ~(NW - 1);
            const uint32_t g = PW_SUB_GEN[wave] + 1u;
            PW_SUB_GEN[wave] = g;
            atomicAdd(&PW_SUB_CNT[grp], 1u);
            while (*(volatile uint32_t*)&PW_SUB_CNT[grp] < g * (uint32_t)(NW - 1)) __builtin_amdgcn_s_sleep(1);
        }
    }
    wave_sync();
}

// Reeds-Shepp evaluation scratch of one wave: running best per query (indexed by position in rsq), round scratch
struct PwRsBest {
    unsigned long long bestL[PW_RSQ], tmpL[PW_RSQ];
    int32_t bestW[PW_RSQ], tmpW[PW_RSQ];
    double best_l[PW_RSQ][AVP_RS_MAXSEG];
    uint8_t w_err[PW_RSQ];
};

// State of one problem = one group of NW waves (LDS). Field names follow PlShared where the shared device functions read them.
template <int NW>
struct PwSharedT {
    // lattice / id space and sweep (pl_sweep_init, pl_relax, pl_expand_bucket, pl_hquery_*)
    int32_t col0, row0, colMin, colMax, rowMin, rowMax, orow0, alias;
    int64_t goal_id;
    int32_t regular;
    int32_t E;
    uint32_t qcount[PL_NQ], qbase[PL_NQ];
    int32_t qover;
    static constexpr bool RELAX_PRECHECK = true;        // (pl_hquery_miss: sixteen searches per CU, the pre-check load saves atomics)
    uint32_t dF; int64_t idF;
    int32_t hasF;
    int64_t h_cells, h_misses;
    // A*
    int32_t nnodes, nheap, nclosed, closed_nonempty;
    int64_t global_index;
    int32_t cur;
    int32_t status, done;
    double goal[3];
    int32_t pid;
    int32_t rs_status, rs_npts, rs_first_coll, in_radius, collision;
    RsPath rs;
    int64_t n_checks, n_rs;
    MapTabs mt;
    uint32_t hq_d;
    int32_t next_child, need_sweep, have_d, fast;
    int64_t pending_id;
    int32_t next_cur, have_next;
    // per group, set once / per pop
    PlanWs w;                                  // this group's workspace slot
    int32_t slot, can_fast, n_passes, n_todo, book_status, shot_stop;
    int64_t n_pops;
    PlNode cn;                                 // the node being expanded (copy of its arena record)
    int32_t nq, rsq[PW_RSQ];                   // the Reeds-Shepp queries of this pop: 0 = the shot, 1 + i = child i
    uint32_t rs_next, pad3;                    // next unclaimed solver round of this pop (the group's waves draw from it)
    PlChild child[PW_MAXCHILD];
    // shot sampling
    int32_t smp_hi, smp_point_num;
    double smp_l[PW_RS_CAP];
    int8_t smp_seg[PW_RS_CAP];
    double seg_o[AVP_RS_MAXSEG][3];
    int8_t sub_t[PW_MAXCHILD * 4];             // the sub-step poses to check this pop (pose index = child * n_sub + step)
    // Scratch of the two kinds of phases that never overlap in a pop: the Reeds-Shepp evaluation (pw_ph_rs: its results
    // are copied to child[].L / rs before it returns) and the collision passes (their results go to first_coll / rs_first_coll).
    union {
        struct {
            RsFrame frame[PW_RSQ];
            PwRsBest rb[NW];
        };
        PlWaveChkT<PW_WQCAP> wchk[NW];
    };
    static constexpr int RS_CAP = PW_RS_CAP;
    int32_t fetch_go, fetch_nheap;             // (written by the shared pl_resolve_fast_wave; read by plan_kernel's lookahead only)
    int32_t wr_go, wr_done;
    static constexpr bool HEAP_POS = true;
    static constexpr int HEAP_LDS = 0;         // (no LDS heap top in these forms: the whole open list stays in the workspace)
    static constexpr bool LOOK_SECOND = false; // (no expansion lookahead in these forms)
    uint32_t phase[PW_PH_COUNT];               // instrumented instantiation only: shader cycles per phase (lane 0)
    int64_t snap[5];                           // counters saved before a resolution that runs beside the shot (pw_ph_shot_resolve)
    int32_t resume, fresh_done, park_now, sl_pad;   // time slicing: this problem continues a parked search / no unstarted problem is left / park decision
    int64_t slice_end;                          // ... the pop count at which this slice ends (kept here, not in a register across the phase calls)
    static constexpr bool POINT_FAST = NW == 1;       // (pl_check_narrow: the first look pays where the CU is issue bound)
    __device__ __forceinline__ PlWaveChkT<PW_WQCAP>& wave_chk() { return wchk[PwGroup<NW>::wv()]; }
};

static inline __host__ __device__ size_t pw_lds_waves_offset() { return (sizeof(PwCommon) + 15) & ~(size_t)15; }
template <int NW> static inline __host__ __device__ size_t pw_lds_group_stride() { return (sizeof(PwSharedT<NW>) + 15) & ~(size_t)15; }
template <int NW> static inline __host__ __device__ size_t pw_lds_tables_offset() { return pw_lds_waves_offset() + (PW_WAVES / NW) * pw_lds_group_stride<NW>(); }

// Every phase of a pop is a CALLED function on (this group's state, the workgroup's constants), both in LDS: the
// register demand of a phase is its own (fully inlined, the pop loop held the union of all of them: 256 VGPRs plus
// spills, two waves per SIMD), and nothing but two LDS addresses crosses a call. Every phase ends with the group in step.
#define PW_PHASE_ARGS AVP_LDS PwSharedT<NW>* sp, AVP_LDS const PwCommon* cp
#define PW_PHASE_REFS typedef PwGroup<NW> G; PwSharedT<NW>& s = *(PwSharedT<NW>*)sp; const PwCommon& c = *(const PwCommon*)cp; \
                      const int lane = threadIdx.x & 63, wv = G::wv(), gtid = G::tid(); (void)lane; (void)wv; (void)gtid

// Reeds-Shepp optimal paths of the queries s.rsq[0 .. s.nq) of this group (s.frame[k] set): calc_optimal_path's result
// per query in rb[0].bestW / best_l / w_err. The rounds (solver group x 64 items) alternate between the group's waves,
// each with a running best of its own; wave 0 merges them with the same rule.
template <int NW>
__device__ __noinline__ void pw_rs_eval(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    const int nq = s.nq;
    const double maxc = c.p.maxc;
    PwRsBest& rb = s.rb[wv];
    if (lane < nq) { rb.bestL[lane] = ~0ull; rb.bestW[lane] = -1; rb.w_err[lane] = 0; }
    wave_sync();
    // The rounds (solver group x 64 items) are drawn by the group's waves from one counter, dearest solver group first
    // (cycles per round, scripts/microbench/rs_words.hip: LRLRn + LRLRp 19 k, SLS 8 k, LRL / LRSL 6 k, LSR 6 k, LRSR 4.5 k,
    // LSL 4 k, LRSLR 3 k): list scheduling keeps the waves within one round of each other. Minimum and tie rule are
    // commutative, so who evaluates which round -- and in which order -- changes no result.
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = NW > 1 ? atomicAdd(&s.rs_next, 1u) : s.rs_next++;
        r = __shfl(r, 0, 64);
        int sg = -1, base = 0;
        {
            uint32_t left = r;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int g_ = (0x40352617 >> (4 * (7 - k))) & 7;               // order 4 0 3 5 2 6 1 7, dearest first
                const uint32_t nr = (uint32_t)(((nq << c.sg_shift[g_]) + 63) >> 6);
                if (sg < 0) { if (left < nr) { sg = g_; base = (int)left * 64; } else left -= nr; }
            }
        }
        if (sg < 0) break;
        {
            const int L = c.sg_l[sg], sh = c.sg_shift[sg], off = c.sg_off[sg], gmax = c.sg_gmax[sg];
            const int items = nq << sh;
            const int it = base + lane;
            const bool active = it < items;
            const int q = active ? it >> sh : 0, j = it & (L - 1);
            const int word = c.item_word[off + j], mm = c.item_m[off + j], g = c.item_g[off + j];
            double l[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
            bool ok = false;
            if (active) ok = rs_word(word, s.frame[q], l);
            double Lsum = 0;
#pragma unroll
            for (int i = 0; i < 5; i++) Lsum = Lsum + fabs(l[i]);            // path.L (rs_curve.py:145), unused tail is 0.0
            // set_path inside the type group, in word order: member e is final once members 0 .. e-1 are
            bool dup = false, acc = false, err = false;
            for (int e = 0; e < gmax; e++) {
                if (mm == e) {
                    const bool cand = ok && !dup && !(Lsum >= 1000.0);
                    acc = cand && (Lsum >= 0.01);
                    err = cand && !(Lsum >= 0.01);                           // the reference's assertion (rs_curve.py:153)
                }
                if (e + 1 < gmax) {
                    int src = lane - mm + e;                                 // member e of my group (groups are aligned to their size)
                    src = src < 0 ? 0 : src;
                    const int a_e = __shfl(acc ? 1 : 0, src, 64);
                    double sum = 0;
#pragma unroll
                    for (int i = 0; i < 5; i++) sum = sum + (__shfl(l[i], src, 64) - l[i]);
                    if (mm > e && e < g && a_e && sum <= 0.01) dup = true;
                }
            }
            // arg-min of the round per query, then merged into the running best: smaller length, or the same length and a
            // later word (calc_optimal_path's "<=" keeps the last of equal minima in word order)
            const double Lm = Lsum / maxc;
            const unsigned long long lb = (unsigned long long)__double_as_longlong(Lm);
            if (lane < nq) { rb.tmpL[lane] = ~0ull; rb.tmpW[lane] = -1; }
            wave_sync();
            if (acc) atomicMin(&rb.tmpL[q], lb);
            if (err) rb.w_err[q] = 1;
            wave_sync();
            if (acc && lb == rb.tmpL[q]) atomicMax(&rb.tmpW[q], word);
            wave_sync();
            if (acc && lb == rb.tmpL[q] && word == rb.tmpW[q]) {
                const unsigned long long bl = rb.bestL[q];
                if (lb < bl || (lb == bl && word > rb.bestW[q])) {
                    rb.bestL[q] = lb; rb.bestW[q] = word;
#pragma unroll
                    for (int i = 0; i < 5; i++) rb.best_l[q][i] = l[i];
                }
            }
            wave_sync();
        }
    }
    if constexpr (NW > 1) {
        G::sync_lds();
        if (wv == 0 && lane < nq) {
            PwRsBest& r0 = s.rb[0];
#pragma unroll
            for (int k = 1; k < NW; k++) {
                const PwRsBest& rk = s.rb[k];
                const unsigned long long lb = rk.bestL[lane];
                const int word = rk.bestW[lane];
                if (rk.w_err[lane]) r0.w_err[lane] = 1;
                if (word >= 0 && (lb < r0.bestL[lane] || (lb == r0.bestL[lane] && word > r0.bestW[lane]))) {
                    r0.bestL[lane] = lb; r0.bestW[lane] = word;
#pragma unroll
                    for (int i = 0; i < 5; i++) r0.best_l[lane][i] = rk.best_l[lane][i];
                }
            }
        }
        wave_sync();
    }
}

// Result of query slot q as pl_rs_fold_wave returns it: 0 path in `out` (normalised lengths), 1 no candidate, 2 assertion.
AVP_D int pw_rs_result(const PwRsBest& rb, int q, RsPath& out)
{
    out.n = 0; out.L = 0;
    if (rb.w_err[q]) return 2;
    const int wd = rb.bestW[q];
    if (wd < 0) return 1;
    const RsWord W = RS_WORDS[wd];
    out.n = W.n;
    out.t[0] = 0 < W.n ? W.a : (int8_t)-1; out.t[1] = 1 < W.n ? W.b : (int8_t)-1; out.t[2] = 2 < W.n ? W.c : (int8_t)-1;
    out.t[3] = 3 < W.n ? W.d : (int8_t)-1; out.t[4] = 4 < W.n ? W.e : (int8_t)-1;
    double Ln = 0;
#pragma unroll
    for (int i = 0; i < AVP_RS_MAXSEG; i++) { out.l[i] = rb.best_l[q][i]; Ln = Ln + fabs(out.l[i]); }
    out.L = Ln;
    return 0;
}

// ---- per-problem set-up: hybrid_a_star.__init__ (hybrid_a_star.py:72-124) ------------------------------------------
template <int NW>
__device__ __noinline__ void pw_ph_init(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    const DevMap& m = c.m;
    const PlanWs& w = s.w;
    const int64_t pid = s.pid;
    const double sx = c.starts[3 * pid], sy = c.starts[3 * pid + 1], sth = c.starts[3 * pid + 2];
    const double gx = c.goals[3 * pid], gy = c.goals[3 * pid + 1], gth = c.goals[3 * pid + 2];
    for (int64_t i = gtid; i < c.dims.hashCap; i += G::N) w.hash[i] = 0;
    if (gtid == 0) {
        const bool pose_ok = pl_pose_ok(sx, sy, sth) && pl_pose_ok(gx, gy, gth);
        s.status = !pose_ok ? 7 /* AVP_PLAN_BAD_POSE */ : (c.nchild > PW_MAXCHILD || c.nsubs > 4 * PW_MAXCHILD) ? AVP_PLAN_RETRY : 0; s.done = 0;      // (more children / sub-step poses than a group holds: plan_kernel plans it)
        s.nnodes = 0; s.nheap = 0; s.nclosed = 0; s.closed_nonempty = 0; s.have_next = 0; s.next_cur = -1;
        s.global_index = 0; s.cur = -1; s.n_checks = 0; s.n_rs = 0; s.n_pops = 0;
        s.goal[0] = gx; s.goal[1] = gy; s.goal[2] = pose_ok ? avp_pi_2_pi(gth) : 0.0;
        s.rs_status = 0; s.rs_npts = 0; s.in_radius = 0; s.collision = 0; s.rs.n = 0; s.rs.L = 0;
        s.E = 0; s.h_cells = 0; s.h_misses = 0;
    }
    G::sync();
    const bool go = s.status == 0;
    G::sync();                      // (pl_sweep_init's first lane may set the status: every wave has read it by now)
    if (go) pl_sweep_init<G>(m, w, s, c.dims, gx, gy);
    if (s.status == 0) {
        // hybrid_a_star.__init__: compute_path(x0, y0) (:89-91)
        const int64_t sid = avp_pos_to_index(m, sx, sy);
        pl_hquery_miss<false, G>(m, w, s, sid);
        if (gtid == 0) {
            if (s.hq_d == PL_UNSEEN) s.status = s.qover ? 5 : 2;
            else {
                PlNode& nd = w.nodes[0];
                nd.x = sx; nd.y = sy; nd.th = avp_pi_2_pi(sth); nd.g = 0; nd.h = 0; nd.f = 0;
                nd.index = 0; nd.parent_index = -1; nd.parent_pos = -1; nd.forward = 1; nd.steer_i = -1; nd.state = 1;
                s.nnodes = 1;
                pl_heap_push(w, s, 0, 0.0);
                pl_hash_put(w, c.dims.hashCap, 0);
            }
        }
        G::sync();
    }
}

// ---- the next node (path_planner.py:68-76): popped ahead by the previous resolution, or popped now ----------------------
template <int NW>
__device__ __noinline__ void pw_ph_pop(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    const PlanWs& w = s.w;
    G::sync();
    if (gtid == 0) {
        if (s.have_next) { s.have_next = 0; s.cur = s.next_cur; }
        else if (s.nheap == 0) { s.status = 1; }
        else if (s.n_pops >= c.max_pops) { s.status = 4; }
        else if (c.stage_pops > 0 && s.n_pops >= c.stage_pops) { s.status = AVP_PLAN_RETRY; }
        else {
            const uint32_t cc = pl_heap_pop(w, s);
            s.cur = (int32_t)cc;
            w.nodes[cc].state = 3;
        }
        if (s.status == 0) {
            const PlNode cn = w.nodes[s.cur];
            s.cn = cn;
            const int64_t n_pops = s.n_pops;
            if (c.trace && n_pops < c.max_trace) {
                double* t = c.trace + ((size_t)s.pid * c.max_trace + n_pops) * PL_TRACE_W;
                t[0] = (double)cn.index; t[1] = (double)cn.parent_index; t[2] = (double)avp_pos_to_index(c.m, cn.x, cn.y);
                t[3] = cn.x; t[4] = cn.y; t[5] = cn.th; t[6] = cn.g; t[7] = cn.h; t[8] = cn.f;
                t[9] = cn.forward; t[10] = cn.steer_i < 0 ? NAN : c.k_steer[cn.steer_i];
            }
            s.n_pops = n_pops + 1;
        }
    }
    G::sync_lds();
}

// ---- children poses (expand_node :134-151), their exact-equality look-ups, try_reach_goal's radius test (:308-312) ----
template <int NW>
__device__ __noinline__ void pw_ph_children(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    const DevMap& m = c.m;
    const avp_params& p = c.p;
    const PlanWs& w = s.w;
    const int nchild = c.nchild;
    if (wv == 0) {
        const double cnx = s.cn.x, cny = s.cn.y, cnth = s.cn.th;
        const double ddx = cnx - s.goal[0], ddy = cny - s.goal[1];
        const bool in_radius = avp_within_radius(ddx, ddy, p.flag_radius);   // ** is libm pow (hybrid_a_star.py:308)
        if (lane == 0) { s.in_radius = in_radius ? 1 : 0; s.collision = 0; s.rs_first_coll = 0x7fffffff; s.rs_npts = 0; s.rs_status = 0; s.rs.n = 0; }
        if (lane < nchild) {
            PlChild& ch = s.child[lane];
            const int si = lane % p.n_steer;
            const bool fwd = lane < p.n_steer;
            const double travel = fwd ? p.travel_dt : -p.travel_dt;
            const double th_ = avp_pi_2_pi(cnth + c.k_dth_dt[si]);
            ch.th = th_;
            double sth_, cth_;
            avp_sincos(th_, sth_, cth_);
            ch.x = cnx + travel * cth_;
            ch.y = cny + travel * sth_;
            ch.oob = (ch.x > m.b1 || ch.x < m.b0 || ch.y > m.b3 || ch.y < m.b2) ? 1 : 0;
            ch.found = pl_hash_find(w, c.dims.hashCap, ch.x, ch.y, ch.th);
            ch.found_state = ch.found >= 0 ? w.nodes[ch.found].state : 0;
            ch.id = avp_pos_to_index(m, ch.x, ch.y);
            ch.first_coll = 0x7fffffff;
            ch.rs_err = 0;
            ch.L = 0;
        }
        wave_sync();
        // compact list of the sub-step poses to check: sub_t[k] = pose index t = child * n_sub + step. Only the children
        // expand_node checks: a child equal to a closed node or out of bounds is dropped before (:155-165, once the closed
        // list is non-empty), one equal to an open node is re-costed without a check (:169-172, :219-230).
        bool need = false;
        if (lane < c.nsubs) {
            const PlChild& ch = s.child[c.sub_child[lane]];
            const bool found_closed = ch.found >= 0 && ch.found_state == 2, found_open = ch.found >= 0 && ch.found_state == 1;
            need = !(s.closed_nonempty && (found_closed || ch.oob)) && !found_open;
        }
        const unsigned long long mk = __ballot(need);
        if (need) s.sub_t[__popcll(mk & ((1ull << lane) - 1ull))] = (int8_t)lane;
        if (lane == 0) { s.n_todo = __popcll(mk); s.n_passes = (__popcll(mk) + PL_WPOSE - 1) / PL_WPOSE; }
    }
    G::sync_lds();
}

// ---- sub-step collision checks (:185-204), PL_WPOSE poses per pass, the passes dealt to the group's waves ------------
template <bool STAGE, int NW>
__device__ __noinline__ void pw_ph_substeps(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    const avp_params& p = c.p;
    const double cnx = s.cn.x, cny = s.cn.y, cnth = s.cn.th;
    const int ntodo = s.n_todo;
    for (int base = wv * PL_WPOSE; base < ntodo; base += NW * PL_WPOSE) {
        const int cnt = min(PL_WPOSE, ntodo - base);
        uint32_t* hits = &s.wave_chk().hit[0];
        pl_check_wave<STAGE>(c.env, s, cnt, [&](int k, double& x, double& y, double& th, double& cs, double& sn) {
            const int t = s.sub_t[base + k];
            const double td = c.sub_td[t];
            th = avp_pi_2_pi(cnth + c.sub_dth[t]);
            avp_sincos(th, sn, cs);
            x = cnx + td * cs;
            y = cny + td * sn;
        }, hits);
        if (lane < cnt && hits[lane]) { const int t = s.sub_t[base + lane]; atomicMin(&s.child[c.sub_child[t]].first_coll, (int)c.sub_j[t]); }
        wave_sync();
    }
    if (gtid == 0) s.can_fast = (s.closed_nonempty && (s.nnodes + c.nchild <= c.maxNodes)) ? 1 : 0;
    G::sync_lds();
}

// ---- Reeds-Shepp: the shot from the popped node (:326-332) and the children's lengths (:286-294) ----------------------
// Only the queries whose result the pop can use are evaluated: the shot inside flag_radius; a child unless expand_node
// drops it before it reaches calc_node_heuristic -- equal to a closed node or out of bounds (:155-165, once the closed
// list is non-empty), or new and colliding (:197-204). The reference does not solve those either.
template <int NW>
__device__ __noinline__ void pw_ph_rs(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    const avp_params& p = c.p;
    const int nchild = c.nchild;
    const bool in_radius = s.in_radius != 0;
    bool need = false;
    int k = 0;
    if (wv == 0) {
        if (lane == 0) need = in_radius;
        else if (lane <= nchild) {
            const PlChild& ch = s.child[lane - 1];
            const bool found_closed = ch.found >= 0 && ch.found_state == 2, found_open = ch.found >= 0 && ch.found_state == 1;
            need = !(s.closed_nonempty && (found_closed || ch.oob)) && !(!found_open && ch.first_coll != 0x7fffffff);
        }
        const unsigned long long mk = __ballot(need);
        k = __popcll(mk & ((1ull << lane) - 1ull));
        if (need) {
            s.rsq[k] = lane;
            if (lane == 0) s.frame[k] = rs_frame(s.cn.x, s.cn.y, s.cn.th, s.goal[0], s.goal[1], s.goal[2], p.maxc);
            else { const PlChild& ch = s.child[lane - 1]; s.frame[k] = rs_frame(ch.x, ch.y, ch.th, s.goal[0], s.goal[1], s.goal[2], p.maxc); }
        }
        if (lane == 0) { s.nq = __popcll(mk); s.rs_next = 0; }
    }
    G::sync_lds();
    pw_rs_eval<NW>(sp, cp);
    if (wv == 0) {
        if (need && lane >= 1) {
            RsPath rp;
            const int st = pw_rs_result(s.rb[0], k, rp);
            s.child[lane - 1].rs_err = (int8_t)st; s.child[lane - 1].L = st ? 0.0 : rp.L / p.maxc;
        }
        if (lane == 0 && in_radius) {
            RsPath rp;
            const int st = pw_rs_result(s.rb[0], 0, rp);
            s.rs_status = st;
            if (!st) { s.rs = rp; s.n_rs += 1; }
        }
    }
    G::sync_lds();
}

// ---- the shot: sample in path order, check, stop at the first colliding sample (:335-345) ------------------------------
template <bool STAGE, bool PROFILE, int NW>
__device__ __noinline__ void pw_ph_shot(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    const avp_params& p = c.p;
    const PlanWs& w = s.w;
    // (s.rs_status was set by the previous phase and is not written in this one: every wave of the group reads the same value)
    if (s.rs_status) { if (gtid == 0) s.status = 3; G::sync_lds(); return; }       // no Reeds-Shepp path / the reference's assertion
    // the sampler's index bookkeeping (one lane) beside the chain of segment origins (a wave)
    if (gtid == 0) s.book_status = pl_rs_sample_book(s, p);
    if (wv == (NW > 1 ? 1 : 0)) pl_rs_sample_origins(s, p);
    G::sync_lds();
    if (s.book_status) { if (gtid == 0) s.status = AVP_PLAN_RETRY; G::sync_lds(); return; }      // more samples than this form holds
    const PlNode cn = s.cn;
    const int total = s.smp_hi + 1;
    double cm, sm;
    avp_sincos(-cn.th, sm, cm);
    // Rounds in path order; a round deals `per` consecutive samples to every wave of the group and ends with the group
    // in step, so that the decision to stop sees every result of the round: the reference stops at the first colliding
    // sample, typically 5 .. 10 m down the path. A group of several waves starts with SHORT chunks (8 samples per round
    // in total, then 16, then 8 per wave): the samples get dearer along the path -- towards the obstacles around the
    // goal --, and a first round that spreads 32 samples over four waves was measured to cost more than the one or two
    // 8-sample passes a single wave needs before it may stop.
    int base0 = 0;
    for (int r = 0; base0 < total; r++) {
        const int per = NW == 1 ? PL_WPOSE : min(PL_WPOSE, (PL_WPOSE / NW) << r);
        const int base = base0 + wv * per;
        if (base < total) {
            const int cnt = min(per, total - base);
            if constexpr (PROFILE) { if (lane == 0) atomicAdd(&s.phase[PW_PH_NPASS], 1u); }
            double tx = 0.0, ty = 0.0, tth = 0.0;
            const int mine = base + lane;
            if (lane < cnt) pl_rs_sample_world(w, s, p, cn, cm, sm, mine, tx, ty, tth);
            uint32_t* hits = &s.wave_chk().hit[0];
            pl_check_wave<STAGE>(c.env, s, cnt, [&](int k, double& x, double& y, double& th, double& cs, double& sn) {
                x = tx; y = ty; th = avp_pi_2_pi(tth); /* :339 */
                avp_sincos(th, sn, cs);
            }, hits);
            if (lane < cnt && hits[lane]) atomicMin(&s.rs_first_coll, mine);
        }
        base0 += NW * per;
        G::sync_lds();
        // every sample before base0 has been checked: the smallest colliding index among them is final. Stop -- unless it may
        // lie in the trailing px == 0.0 tail the reference pops (rs_curve.py:588-592): that is only known once every sample
        // has been produced (rs_npts grows with them), so keep going then. ONE lane decides and the group meets again: a
        // wave that read the two words itself could see what a faster wave has already written in the next round.
        if constexpr (NW > 1) {
            if (gtid == 0) { const int fc = s.rs_first_coll, np = s.rs_npts; s.shot_stop = (fc != 0x7fffffff && fc < np) ? 1 : 0; }
            G::sync_lds();
            if (s.shot_stop) break;
        } else {
            const int fc = s.rs_first_coll, np = s.rs_npts;
            if (fc != 0x7fffffff && fc < np) break;
        }
    }
    G::sync_lds();
    if (gtid == 0) {
        // a hit at or past the trimmed length belongs to a popped entry (rs_curve.py:588-592)
        if (s.rs_first_coll != 0x7fffffff && s.rs_first_coll >= s.rs_npts) s.rs_first_coll = 0x7fffffff;
        if (s.rs_first_coll == 0x7fffffff) { s.n_checks += s.rs_npts; s.done = 1; }
        else { s.collision = 1; s.n_checks += s.rs_first_coll + 1; }
    }
    G::sync_lds();
}

// ---- child resolution in child order (:153-232): wave-parallel when every heuristic query hits the closed frontier ------
template <int NW>
__device__ __noinline__ void pw_ph_resolve_fast(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    if (gtid == 0) { s.next_child = 0; s.have_d = 0; s.need_sweep = 0; s.fast = s.can_fast; }
    if (wv == 0) {
        if (!s.can_fast && lane < c.nchild) s.child[lane].pre_d = pl_id_in_range(c.m, s.child[lane].id) ? s.w.dist[s.child[lane].id] : PL_UNSEEN;
        wave_sync();
        if (s.can_fast) {
            const PlNode cn = s.cn;
            pl_resolve_fast_wave<false>(c.m, c.p, s.w, s, c.dims, cn, c.nchild, s.n_pops < c.max_pops && !(c.stage_pops > 0 && s.n_pops >= c.stage_pops));
        }
    }
    G::sync();
}
// ... else lane 0 in child order, the group extending the heuristic sweep at every miss; then the node is closed (:235-239)
// ---- NW > 1, node inside flag_radius: the shot on the waves 1 .. NW-1 WHILE wave 0 resolves the children ------------------
// The shot's outcome is not an input of expand_node, so the resolution may run beside it (plan_kernel does the same): when
// the shot then turns out collision free the search ends at this pop, BEFORE expand_node in the reference -- the counters
// are put back (the arena / heap / hash side effects touch no node on the final path's parent chain). What this phase
// leaves behind equals pw_ph_shot followed by pw_ph_resolve_fast.
#ifndef PW_SPECULATE
#define PW_SPECULATE 1
#endif
template <bool STAGE, bool PROFILE, int NW>
__device__ __noinline__ void pw_ph_shot_resolve(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    static_assert(NW > 1, "needs a wave beside the resolving one");
    const avp_params& p = c.p;
    const PlanWs& w = s.w;
    if (s.rs_status) { if (gtid == 0) s.status = 3; G::sync_lds(); return; }       // no Reeds-Shepp path / the reference's assertion
    if (wv == 0) {
        if (lane == 0) {
            s.next_child = 0; s.have_d = 0; s.need_sweep = 0; s.fast = s.can_fast;
            s.snap[0] = s.nnodes; s.snap[1] = s.n_checks; s.snap[2] = s.n_rs; s.snap[3] = s.nclosed; s.snap[4] = s.nheap;
        }
        wave_sync();
        if (!s.can_fast) { if (lane < c.nchild) s.child[lane].pre_d = pl_id_in_range(c.m, s.child[lane].id) ? w.dist[s.child[lane].id] : PL_UNSEEN; }
        else {
            const PlNode cn0 = s.cn;
            pl_resolve_fast_wave<false>(c.m, p, w, s, c.dims, cn0, c.nchild, s.n_pops < c.max_pops && !(c.stage_pops > 0 && s.n_pops >= c.stage_pops));
        }
    } else {
        // the sampler's index bookkeeping (one lane; a whole-wave version of it measured the same) beside the chain of
        // segment origins (a wave; in the pair form behind it)
        if (wv == 1 && lane == 0) s.book_status = pl_rs_sample_book(s, p);
        if (wv == (NW > 2 ? 2 : 1)) { wave_sync(); pl_rs_sample_origins(s, p); }
        pw_sub_sync_lds<NW>();
        if (!s.book_status) {                           // (else: more samples than this form holds -- handed back below)
            constexpr int NS = NW - 1;                      // checking waves
            const int sw = wv - 1;
            const PlNode cn = s.cn;
            const int total = s.smp_hi + 1;
            double cm, sm;
            avp_sincos(-cn.th, sm, cm);
            int base0 = 0;
            for (int r = 0; base0 < total; r++) {
                const int per = NS == 1 ? PL_WPOSE : min(PL_WPOSE, max(1, PL_WPOSE / NS) << r);
                const int base = base0 + sw * per;
                if (base < total) {
                    const int cnt = min(per, total - base);
                    double tx = 0.0, ty = 0.0, tth = 0.0;
                    const int mine = base + lane;
                    if (lane < cnt) pl_rs_sample_world(w, s, p, cn, cm, sm, mine, tx, ty, tth);
                    uint32_t* hits = &s.wave_chk().hit[0];
                    pl_check_wave<STAGE>(c.env, s, cnt, [&](int k, double& x, double& y, double& th, double& cs, double& sn) {
                        x = tx; y = ty; th = avp_pi_2_pi(tth); /* :339 */
                        avp_sincos(th, sn, cs);
                    }, hits);
                    if (lane < cnt && hits[lane]) atomicMin(&s.rs_first_coll, mine);
                }
                base0 += NS * per;
                // (as in pw_ph_shot: every sample before base0 is checked; ONE lane decides whether the known hit ends the shot)
                if constexpr (NS > 1) {
                    pw_sub_sync_lds<NW>();
                    if (sw == 0 && lane == 0) { const int fc = s.rs_first_coll, np = s.rs_npts; s.shot_stop = (fc != 0x7fffffff && fc < np) ? 1 : 0; }
                    pw_sub_sync_lds<NW>();
                    if (s.shot_stop) break;
                } else {
                    wave_sync();
                    const int fc = s.rs_first_coll, np = s.rs_npts;
                    if (fc != 0x7fffffff && fc < np) break;
                }
            }
        }
    }
    G::sync();
    if (s.book_status) { if (gtid == 0) s.status = AVP_PLAN_RETRY; G::sync_lds(); return; }
    if (gtid == 0) {
        // a hit at or past the trimmed length belongs to a popped entry (rs_curve.py:588-592)
        if (s.rs_first_coll != 0x7fffffff && s.rs_first_coll >= s.rs_npts) s.rs_first_coll = 0x7fffffff;
        if (s.rs_first_coll == 0x7fffffff) {
            // success: the reference returns before expand_node -- undo the speculative bookkeeping
            s.nnodes = (int32_t)s.snap[0]; s.n_checks = s.snap[1]; s.n_rs = s.snap[2]; s.nclosed = (int32_t)s.snap[3]; s.nheap = (int32_t)s.snap[4];
            s.n_checks += s.rs_npts; s.done = 1;
        } else { s.collision = 1; s.n_checks += s.rs_first_coll + 1; }
    }
    G::sync_lds();
}

template <int NW>
__device__ __noinline__ void pw_ph_resolve_slow(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    const DevMap& m = c.m;
    const avp_params& p = c.p;
    const PlanWs& w = s.w;
    const int nchild = c.nchild, maxNodes = c.maxNodes;
    while (!s.fast) {
        if (gtid == 0) {
            const PlNode cn = s.cn;
            s.need_sweep = 0;
            int i = s.next_child;
            for (; i < nchild && s.status == 0; i++) {
                const PlChild ch = s.child[i];
                const int si = i % p.n_steer;
                const int is_forward = i < p.n_steer ? 1 : 0;
                const bool found_closed = ch.found >= 0 && ch.found_state == 2;
                if (s.closed_nonempty && (found_closed || ch.oob)) continue;          // :155-165
                const bool found_open = ch.found >= 0 && ch.found_state == 1;
                if (!found_open && ch.first_coll != 0x7fffffff) {
                    s.n_checks += ch.first_coll + 1;
                    if (s.nnodes >= maxNodes) { s.status = 5; break; }
                    const int32_t pos = s.nnodes++;
                    PlNode& nd = w.nodes[pos];
                    nd.x = ch.x; nd.y = ch.y; nd.th = ch.th; nd.g = 0; nd.h = 0; nd.f = 0;
                    nd.index = (int32_t)(s.global_index + i + 1); nd.parent_index = cn.index; nd.parent_pos = s.cur;
                    nd.forward = (int8_t)is_forward; nd.steer_i = (int8_t)si; nd.state = 2; nd.heap_pos = -1;
                    pl_hash_put(w, c.dims.hashCap, pos);
                    s.nclosed++; s.closed_nonempty = 1;
                    continue;
                }
                uint32_t hd;
                if (s.have_d) { hd = s.hq_d; s.have_d = 0; }
                else if (!pl_hquery_hit(m, s, ch.id, ch.pre_d, hd)) { s.pending_id = ch.id; s.need_sweep = 1; break; }
                if (hd == PL_UNSEEN) { if (!found_open) s.n_checks += p.n_sub; s.status = s.qover ? 5 : 2; break; }
                s.n_rs += 1;
                if (ch.rs_err) { s.status = ch.rs_err == 4 ? 5 : 3; break; }
                const double hv1 = (double)hd / 100, hv2 = ch.L;
                const double hval = hv2 > hv1 ? hv2 : hv1;
                if (!found_open) {
                    s.n_checks += p.n_sub;
                    if (s.nnodes >= maxNodes) { s.status = 5; break; }
                    const double g = pl_node_cost(p, is_forward, ch.th, cn.th, cn.forward);
                    const int32_t pos = s.nnodes++;
                    PlNode& nd = w.nodes[pos];
                    nd.x = ch.x; nd.y = ch.y; nd.th = ch.th; nd.g = g; nd.h = hval; nd.f = g + hval;
                    nd.index = (int32_t)(s.global_index + i + 1); nd.parent_index = cn.index; nd.parent_pos = s.cur;
                    nd.forward = (int8_t)is_forward; nd.steer_i = (int8_t)si; nd.state = 1;
                    pl_heap_push(w, s, (uint32_t)pos, g + hval);
                    pl_hash_put(w, c.dims.hashCap, pos);
                } else {
                    PlNode& chn = w.nodes[ch.found];
                    const double new_g = pl_node_cost(p, chn.forward, chn.th, cn.th, cn.forward);
                    const double new_f = hval + new_g;
                    if (new_f < chn.f) {
                        chn.f = new_f; chn.g = new_g; chn.h = hval;
                        pl_heap_set_key(w, s, chn.heap_pos, new_f);
                        chn.parent_index = cn.index; chn.parent_pos = s.cur;
                        chn.forward = (int8_t)is_forward; chn.steer_i = (int8_t)si;
                    }
                }
            }
            s.next_child = i;
        }
        G::sync();
        if (!s.need_sweep) break;
        pl_hquery_miss<false, G>(m, w, s, s.pending_id);
        if (gtid == 0) s.have_d = 1;
        if (gtid < nchild) s.child[gtid].pre_d = pl_id_in_range(m, s.child[gtid].id) ? w.dist[s.child[gtid].id] : PL_UNSEEN;
        G::sync();
    }
    if (gtid == 0 && s.status == 0) {
        w.nodes[s.cur].state = 2;
        s.nclosed++; s.closed_nonempty = 1;
        s.global_index += nchild;
    }
    G::sync();
}

// ---- finish_path (:351-389) + assembly (path_planner.py:100-108) ------------------------------------------------------------
// ---- time slicing: a search leaves its group (state -> its workspace slot, a ticket in the resume ring) / comes back ----
// What survives between two pops of a search is the head of PwSharedT up to the per-phase scratch union plus its workspace
// slot. The group that takes the search up again may sit on another XCD: agent-scope release / acquire around the hand-over
// (write-back / invalidate of the L2s; every wave fences its own stores).
#define PW_SAVE_WORDS(NW) ((int)((offsetof(PwSharedT<NW>, wchk) + 3) / 4))
AVP_D uint32_t* pw_ring_word(const PwCommon& c, uint32_t ticket)
{
    return (uint32_t*)(c.workspace + (size_t)(ticket % (uint32_t)c.n) * c.dims.bytes + c.dims.parkOff + (PL_PARK_BYTES - 4));
}
template <int NW>
__device__ __noinline__ void pw_ph_park(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    static_assert(offsetof(PwSharedT<NW>, wchk) + 8 <= PL_PARK_BYTES, "park area");
    uint32_t* dst = (uint32_t*)s.w.park;
    const int32_t pid = s.pid;
    const uint32_t* src = (const uint32_t*)&s;
    for (int i = gtid; i < PW_SAVE_WORDS(NW); i += G::N) dst[i] = src[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    G::sync();
    if (gtid == 0) {
        const uint32_t t = atomicAdd(c.sl + 1, 1u);
        __hip_atomic_store(pw_ring_word(c, t), (uint32_t)pid + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    G::sync();
}
template <int NW>
__device__ __noinline__ void pw_ph_restore(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    const uint32_t* src = (const uint32_t*)(c.workspace + (size_t)s.pid * c.dims.bytes + c.dims.parkOff);
    G::sync_lds();                                      // (every lane holds the address before the head of s is overwritten)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    uint32_t* dst = (uint32_t*)&s;
    for (int i = gtid; i < PW_SAVE_WORDS(NW); i += G::N) dst[i] = src[i];
    G::sync();
}
// The next problem of this group: an unstarted one while there are any, then a parked search from the ring (in the order
// they were parked), or 0x7fffffff once every search of the batch has finished. Lane 0 of the group.
AVP_D void pw_next_problem(const PwCommon& c, const int32_t* order, int32_t& pid, int32_t& resume, int32_t& fresh_done)
{
    pid = 0x7fffffff; resume = 0;
    if (!fresh_done) {
        const uint32_t t = atomicAdd(c.counter, 1u);
        if ((int64_t)t < c.n) { pid = order ? order[t] : (int32_t)t; return; }
        fresh_done = 1;
    }
    if (c.slice_pops <= 0) return;
    for (uint32_t spin = 0; spin < (1u << 26); spin++) {         // (bounded: a lost ticket must not hang the device)
        if ((int64_t)__hip_atomic_load(c.sl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= c.n) return;
        const uint32_t h = __hip_atomic_load(c.sl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t tl = __hip_atomic_load(c.sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)(tl - h) > 0) {
            if (atomicCAS(c.sl + 0, h, h + 1u) != h) continue;
            uint32_t* e = pw_ring_word(c, h);                    // ticket h is ours: its entry follows the ticket's issue at once
            uint32_t v = 0;
            for (uint32_t w2 = 0; w2 < (1u << 24); w2++) {
                v = __hip_atomic_load(e, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (v) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (!v) return;
            __hip_atomic_store(e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pid = (int32_t)(v - 1u); resume = 1;
            return;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

template <int NW>
__device__ __noinline__ void pw_ph_finish(PW_PHASE_ARGS)
{
    PW_PHASE_REFS;
    const avp_params& p = c.p;
    const PlanWs& w = s.w;
    G::sync();
    if (s.status == 1 && s.cur >= 0 && s.in_radius && s.rs.n > 0 && s.rs_status == 0 && s.collision) {
        // the reference hands back the last (colliding) shot when the open list runs empty (path_planner.py:100-108):
        // the early exit of the shot's checks may have left samples unproduced
        const PlNode cl = w.nodes[s.cur];
        double cm, sm;
        avp_sincos(-cl.th, sm, cm);
        for (int i = gtid; i <= s.smp_hi; i += G::N) { double a, b, cc; pl_rs_sample_world(w, s, p, cl, cm, sm, i, a, b, cc); }
        G::sync();
    }
    if (gtid == 0) {
        if (s.status == AVP_PLAN_RETRY) { c.results[s.pid].status = AVP_PLAN_RETRY; if (c.deferred) atomicAdd(c.deferred, 1u); }
        else pl_write_result<false>(p, w, s, c.k_travel_ddt1, c.k_dth_ddt1, c.results, c.paths, c.max_path, s.pid, s.n_pops, s.slot, 0ll);
    }
    G::sync();
}

template <bool STAGE, bool PROFILE = false, int NW = 1>
__global__ __launch_bounds__(PW_THREADS) PW_OCC void plan_wave_kernel(DevMap m, avp_params p, const double* __restrict__ starts,
                                                               const double* __restrict__ goals, int64_t n, int32_t maxNodes,
                                                               char* __restrict__ workspace, unsigned int* __restrict__ counter,
                                                               avp_plan_result_dev* __restrict__ results,
                                                               double* __restrict__ paths, int32_t max_path,
                                                               double* __restrict__ trace, int32_t max_trace,
                                                               const int32_t* __restrict__ order, int32_t stage_pops, int32_t retry_only,
                                                               unsigned int* __restrict__ deferred, int32_t gate_lo, int32_t gate_hi,
                                                               int32_t slice_pops, unsigned int* __restrict__ sl)
{
    typedef PwGroup<NW> G;
    // second stage of a staged call (retry_only with a gate): the host launches every form, the one whose range
    // (gate_lo, gate_hi] holds the number of searches the first stage handed back plans them, the others end here
    if (retry_only && gate_hi > 0) { const unsigned int d = *(volatile unsigned int*)deferred; if (d <= (unsigned int)gate_lo || d > (unsigned int)gate_hi) return; }
    static_assert(PW_WAVES % NW == 0, "groups of NW adjacent waves");
    avp_lds_tables_fill<true>();
    rs_lds_tables_fill();
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
    PwCommon& c = *reinterpret_cast<PwCommon*>(pw_smem);
    const int tid = threadIdx.x, grp = tid / G::N, lane = tid & 63, gtid = G::tid();
    PwSharedT<NW>& s = *reinterpret_cast<PwSharedT<NW>*>(pw_smem + pw_lds_waves_offset() + (size_t)grp * pw_lds_group_stride<NW>());
    AVP_LDS PwSharedT<NW>* const sp = (AVP_LDS PwSharedT<NW>*)&s;
    AVP_LDS const PwCommon* const cp = (AVP_LDS const PwCommon*)&c;
    {
        const PlanDims dims = plan_dims(m.S, m.Sy, maxNodes);
        const int32_t slot = (int32_t)blockIdx.x * (PW_WAVES / NW) + grp;
        if (gtid == 0) { s.w = plan_carve(workspace + (size_t)slot * dims.bytes, dims); s.slot = slot; }
        if (tid == 0) {
            c.m = m; c.p = p; c.dims = dims;
            c.starts = starts; c.goals = goals; c.results = results; c.paths = paths; c.trace = trace;
            c.max_path = max_path; c.max_trace = max_trace; c.maxNodes = maxNodes; c.nchild = 2 * p.n_steer; c.nsubs = 2 * p.n_steer * p.n_sub; c.retry_only = retry_only;
            c.max_pops = p.max_pops > 0 ? p.max_pops : (int64_t)1 << 40;
            c.stage_pops = stage_pops; c.deferred = retry_only ? nullptr : deferred;
            c.slice_pops = (PROFILE || retry_only) ? 0 : slice_pops; c.n = n; c.counter = counter; c.sl = sl; c.workspace = workspace;
        }
        if (gtid == 0) s.fresh_done = 0;
        if (tid < PW_WAVES) { PW_BAR_CNT[tid] = 0; PW_BAR_GEN[tid] = 0; PW_SUB_CNT[tid] = 0; PW_SUB_GEN[tid] = 0; }
    }
    // (a group holds PW_MAXCHILD children = PW_MAXCHILD / 2 steering angles: a configuration with more is handed to plan_kernel)
#pragma unroll
    for (int k = 0; k < PW_MAXCHILD / 2; k++) if (tid == k) {
        c.k_steer[k] = p.steer[k]; c.k_dth_dt[k] = p.dth_dt[k]; c.k_dth_ddt1[k] = p.dth_ddt1[k];
    }
    if (tid == 0) c.k_travel_ddt1 = p.travel_ddt1;
    if (tid < PW_MAXCHILD * 4 && p.n_sub > 0 && p.n_steer > 0) { const int ci = tid / p.n_sub; c.sub_child[tid] = (int8_t)ci; c.sub_j[tid] = (int8_t)(tid - ci * p.n_sub); c.sub_steer[tid] = (int8_t)(ci % p.n_steer); }
    if (tid < 46) { c.item_word[tid] = PW_ITEM_WORD[tid]; c.item_g[tid] = PW_ITEM_G[tid]; c.item_m[tid] = PW_ITEM_M[tid]; }
    if (tid < 8) { c.sg_l[tid] = PW_SG_L[tid]; c.sg_shift[tid] = PW_SG_SHIFT[tid]; c.sg_off[tid] = PW_SG_OFF[tid]; c.sg_gmax[tid] = PW_SG_GMAX[tid]; }
    MapTabs mt;
    if (STAGE) {
        uint64_t* lb = reinterpret_cast<uint64_t*>(pw_smem + pw_lds_tables_offset<NW>());
        double* lx = reinterpret_cast<double*>(lb + (size_t)m.nx * m.wpc);
        double* ly = lx + m.nx;
        for (int i = tid; i < m.nx * m.wpc; i += PW_THREADS) lb[i] = m.colBits[i];
        for (int i = tid; i < m.nx; i += PW_THREADS) lx[i] = m.X[i];
        for (int i = tid; i < m.ny; i += PW_THREADS) ly[i] = m.Y[i];
        mt.X = lx; mt.Y = ly; mt.bits = lb;
    } else { mt.X = m.X; mt.Y = m.Y; mt.bits = m.colBits; }
    if (gtid == 0) s.mt = mt;
    if (tid == 0) pl_chk_env_fill(c.env, m, p, STAGE ? mt.X : nullptr, STAGE ? mt.Y : nullptr, STAGE ? mt.bits : nullptr);
    __syncthreads();
    // per sub-step pose t = child * n_sub + j: heading change and signed travel, from the LDS copies above (the reference's
    // left-to-right products, hybrid_a_star.py:188-191: ... * ddt * (j + 1), speed * ddt * (j + 1))
    if (tid < PW_MAXCHILD * 4 && p.n_sub > 0 && p.n_steer > 0 && p.n_steer <= PW_MAXCHILD / 2) {
        const int ci = c.sub_child[tid], j = c.sub_j[tid], si = c.sub_steer[tid];
        const double tj = c.k_travel_ddt1 * (double)(j + 1);
        c.sub_td[tid] = ci < p.n_steer ? tj : -tj;
        c.sub_dth[tid] = c.k_dth_ddt1[si] * (double)(j + 1);
    }
    __syncthreads();                                   // the last workgroup barrier: from here on every group is on its own

    // Time slicing (slice_pops > 0; the host gives every problem its own workspace slot): a search that is still running
    // after slice_pops pops is parked when another problem is waiting for a group -- an unstarted one, or a parked one in
    // the resume ring -- and the group takes the next problem: unstarted ones first, then the parked ones in turn. The long
    // searches of a batch (a fifth of the problems, nine tenths of the pops) advance side by side instead of one after the
    // other, and the batch ends within a slice of the moment the work runs out, not within a whole long search of it.
    // A search is the same search wherever it runs: results do not depend on the slicing.
    const bool sliced = !PROFILE && !retry_only && slice_pops > 0;
    for (;;) {
        G::sync();
        if (gtid == 0) {
            int32_t pid, resume, fd = s.fresh_done;
            pw_next_problem(c, order, pid, resume, fd);
            s.pid = pid; s.resume = resume; s.fresh_done = fd;
            if (sliced && pid < n) { s.w = plan_carve(workspace + (size_t)pid * c.dims.bytes, c.dims); s.slot = pid; }
        }
        G::sync();
        if (s.pid >= n) break;
        if (retry_only && results[s.pid].status != AVP_PLAN_RETRY) continue;      // (second stage: only what the first handed back)
        long long t_ph = PROFILE ? clock64() : 0ll;
        if constexpr (PROFILE) { if (gtid == 0) for (int k = 0; k < PW_PH_COUNT; k++) s.phase[k] = 0; }
        if (sliced && s.resume) pw_ph_restore<NW>(sp, cp);
        else pw_ph_init<NW>(sp, cp);
        PW_T(PW_PH_INIT);
        if (gtid == 0) s.slice_end = sliced ? s.n_pops + slice_pops : (int64_t)1 << 62;
        G::sync_lds();
        bool parked = false;
        // ---- main loop: path_planner.py:68-98 ------------------------------------------------------
        while (s.status == 0 && !s.done) {
            if (s.n_pops >= s.slice_end) {
                if (gtid == 0) {
                    const uint32_t h = __hip_atomic_load(sl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), tl = __hip_atomic_load(sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s.park_now = ((int64_t)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n || (int32_t)(tl - h) > 0) ? 1 : 0;
                }
                G::sync_lds();
                if (s.park_now) { pw_ph_park<NW>(sp, cp); parked = true; break; }
                G::sync_lds();                          // (every wave has compared before the end of the slice moves)
                if (gtid == 0) s.slice_end += slice_pops;
                G::sync_lds();
            }
            pw_ph_pop<NW>(sp, cp);
            if (s.status != 0) break;
            PW_T(PW_PH_POP);
            pw_ph_children<NW>(sp, cp);
            PW_T(PW_PH_CHILD);
            pw_ph_substeps<STAGE, NW>(sp, cp);
            if constexpr (PROFILE) { if (gtid == 0) atomicAdd(&s.phase[PW_PH_NPASS], (uint32_t)s.n_passes); }
            PW_T(PW_PH_SUB);
            pw_ph_rs<NW>(sp, cp);
            if constexpr (PROFILE) { if (gtid == 0) s.phase[PW_PH_NROUND] += s.nq; }
            PW_T(PW_PH_RS);
            bool resolved = false;
            if (s.in_radius) {
                if constexpr (NW > 1 && PW_SPECULATE) { pw_ph_shot_resolve<STAGE, PROFILE, NW>(sp, cp); resolved = true; }
                else pw_ph_shot<STAGE, PROFILE, NW>(sp, cp);
            }
            PW_T(PW_PH_SHOT);
            if (s.status != 0 || s.done) break;
            if (!resolved) pw_ph_resolve_fast<NW>(sp, cp);
            PW_T(PW_PH_RESOLVE);
            pw_ph_resolve_slow<NW>(sp, cp);
            PW_T(PW_PH_SLOW);
        }
        if (parked) continue;
        if constexpr (PROFILE) t_ph = clock64();
        pw_ph_finish<NW>(sp, cp);
        if (sliced && gtid == 0) atomicAdd(sl + 2, 1u);
        if constexpr (PROFILE) { if (gtid == 0 && s.status != AVP_PLAN_RETRY) { s.phase[PW_PH_FINISH] += (uint32_t)(clock64() - t_ph); for (int k = 0; k < PW_PH_COUNT; k++) results[s.pid].phase_cycles[k] = s.phase[k]; } }
        G::sync();
    }
}
