// avp_plan_kernels.h -- batched hybrid-A* planner: one workgroup = one (start, goal) problem,
// persistent workgroups pull problems from a global counter (plan_kernel), and the device functions it shares with the
// group forms of avp_planw_kernels.h (one, two or four waves per problem).
//
// Replaces, per problem, PathPlanner.a_star_plan (path_plan/path_planner.py:58-110) with
// hybrid_a_star.{__init__, expand_node, calc_node_cost, calc_node_heuristic, try_reach_goal,
// try_rs_curve, finish_path} (path_plan/hybrid_a_star.py:72-389) and the Dijkstra heuristic
// (path_plan/compute_h.py). The pop order (heap layout, ties, in-place key updates) follows
// CPython's heapq exactly; closed/open membership is exact fp64 equality through a hash table.
//
// Heuristic field ("holonomic with obstacles"): the reference runs a resumable 8-connected integer
// Dijkstra (10/14) from the goal over the goal-anchored lattice, keyed by the aliasing grid id of
// map/costmap.py:319-329, stopping at every queried cell and never expanding the cells it stopped
// at. Here the same field is produced by a workgroup-parallel bucketed sweep (buckets of 10 cost
// units = the minimum edge weight, so a whole bucket is final at once), extended lazily by each
// query that misses, with the reference's two order-dependent effects reproduced exactly:
//   * terminator cells (queries that were not yet closed) are flagged and never relax neighbours;
//   * hit/miss of a query is decided against the key (distance, id) of the last miss, which is the
//     reference's closed set, so the set of terminators is the same;
//   * ids alias between the last map column and the first column of the next row; the owner of an
//     aliased id (the lattice cell whose position is stored in the reference's Grid) is the first
//     discoverer in (distance, id, neighbour) order, resolved with a 64-bit atomicMin key.
// See DESIGN.md for the equivalence argument and the measured agreement with the heapq-exact oracle.
#pragma once
#include "avp_device.h"
#include "avp_check_kernels.h"   // wave_sync
#include "avp_rs_kernels.h"

#ifndef PL_THREADS
#define PL_THREADS 512
#endif
#define PL_QCAP 32768                 // entries per rotating bucket queue
#define PL_NQ 4                       // rotating bucket queues
#define PL_MAXCHILD (2 * AVP_MAX_STEER)   // children of one expansion: a lane of the resolving wave each
#define PL_MAXSUBS AVP_MAX_SUBS          // sub-step poses of one expansion (children x sub-steps)
#define PL_RSQ 11                     // RS queries evaluated per pass (46 words each): the shot + 10 children (<= 16)
#define PL_CHK_MAX 1280               // poses per collision pass (shot samples + sub-steps)
#define PL_RS_CAP 1024                // samples of one RS shot
#define PL_UNSEEN 0x7fffffffu
#ifndef PL_SWEEP_U
#define PL_SWEEP_U 4          // (entry, neighbour) pairs a lane relaxes per trip of the bucket loop
#endif
#ifndef PL_RELAX_PRECHECK
#define PL_RELAX_PRECHECK 1              // (measured: the heuristic sweep 15 % shorter; 0 = every relaxation goes straight to the atomic)
#endif
#define PL_TRACE_W 11
#define PL_SCHED_ROUNDS (PL_THREADS >= 512 ? 4 : 8)   // rounds of the RS word schedule: 12 solver chunks of <= 64 lanes over PL_THREADS / 64 waves, with slack
#define PL_FLAG_T 1
// phase timers (thread 0, s_memtime): init, heap pop, (two unused slots), speculative resolution || shot sampling
// and checks, children stage || sub-step checks, RS words .. set_path / arg-min || sampler replay, the rest of the
// resolution (fast path when not speculated, slow path), of which sweep extensions, finish
enum { PH_INIT = 0, PH_POP, PH_RES_CLASSIFY, PH_RES_WRITE, PH_SHOT_CHECK, PH_CHILD, PH_CHILD_RS, PH_RESOLVE, PH_SWEEP, PH_FINISH,
       PH_RES_PUSH, PH_RS_WORDS, PH_CHILD_W0, PH_SHOT_ROUND0, PH_SHOT_REST, PH_SPARE,
       PH_WAVE0 = 16,          // [PH_WAVE0 + 5 * wave + k]: arrival of `wave` at barrier k of a pop, cycles since the pop started:
                               // k = 0 children / sub-steps done, 1 RS words done, 2 set_path / arg-min / replay done,
                               // 3 resolution / shot checks done, 4 end of the pop
       PH_X0 = PH_WAVE0 + 5 * 8,   // 8 fine-grained probes (see the PH_X uses)
       PH_COUNT = PH_X0 + 8 };
#define PH_X(k, t0) do { if constexpr (PROFILE) s.phase[PH_X0 + (k)] += clock64() - (t0); } while (0)
#define PH_MARK(k) do { if constexpr (PROFILE) { if (ph_on && (threadIdx.x & 63) == 0) s.phase[PH_WAVE0 + 5 * (threadIdx.x >> 6) + (k)] += clock64() - t_pop0; } } while (0)
// The timers are compiled into the PROFILE instantiation only (avp_plan_batch_profile): s_memtime instrumentation costs
// ~10 % of the wave cycles, so the production kernel carries none and reports phase_cycles = 0.
#define PH_NOW() (PROFILE ? clock64() : 0ll)
#define PH_ACC(k, t0) do { if constexpr (PROFILE) { if (threadIdx.x == 0) s.phase[k] += clock64() - (t0); } } while (0)

// Who cooperates on one problem: the whole workgroup (plan_kernel: one problem per workgroup) or one wave
// (plan_wave_kernel: one problem per wave, eight independent problems per workgroup).
struct CoopWG {
    static constexpr int N = PL_THREADS;
    static __device__ __forceinline__ int tid() { return threadIdx.x; }
    static __device__ __forceinline__ void sync() { __syncthreads(); }
};
struct CoopWave {
    static constexpr int N = 64;
    static __device__ __forceinline__ int tid() { return threadIdx.x & 63; }
    // lanes of one wave hand data to each other through LDS and through global memory (queues, arena): both must have landed
    static __device__ __forceinline__ void sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); wave_sync(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
};

struct PlNode {
    double x, y, th, g, h, f;
    int32_t index, parent_index, parent_pos, heap_pos;
    int8_t forward, steer_i, state, pad0;   // state: 1 open, 2 closed, 3 popped (being expanded)
    int32_t pad1;
};

struct avp_plan_result_dev {          // mirrors avp_plan_result in include/avp.h
    int32_t status, n_pops, n_astar, n_rs_pts, n_final, rs_n, in_radius_last, rs_collision;
    int64_t n_checks, n_rs, n_closed, n_open, h_cells, h_misses, global_index, n_nodes;
    int8_t rs_types[8];
    double rs_lengths[5];
    double rs_L;
    double rs_start[3];               // RS sample 0 (the popped node's pose)
    int32_t rs_dir0, slot;
    int64_t phase_cycles[64];         // diagnostics: shader cycles per phase (see PH_* below)
};

struct PlHeapEnt { double f; uint32_t node; uint32_t pad; };
// The first S::HEAP_LDS entries of the open list's binary heap live in LDS (members hl_f / hl_n of the state struct;
// 0 in the wave form), the rest in the slot's workspace. Same array, same order, two homes.
#ifndef PL_HEAP_LDS
#define PL_HEAP_LDS 0                  // (measured: 1024 or 2048 entries in LDS make config[1] 3-5 % SLOWER than the L1/L2-resident array)
#endif
#ifndef PL_HEAP_POS
#define PL_HEAP_POS 1
#endif

struct PlanWs {                       // per-slot workspace carve (device pointers)
    uint32_t* dist;                   // [idCap]
    uint8_t* flags;                   // [idCap]
    unsigned long long* aliasKey;     // [rowCap]
    unsigned long long* queue;        // [PL_NQ][PL_QCAP]  (dist << 32 | id)
    PlNode* nodes;                    // [maxNodes]
    PlHeapEnt* heap;                  // [maxNodes] binary heap, key copied next to the node position
    uint32_t* hash;                   // [hashCap] node position + 1, 0 = empty
    double* rsbuf;                    // [PL_RS_CAP * 3]
    int8_t* rsdir;                    // [PL_RS_CAP]
    char* park;                       // [PL_PARK_BYTES] a parked search's LDS state + its ring word (time-sliced group forms)
};
#define PL_PARK_BYTES 8192

struct PlanDims { int64_t idCap, rowCap, maxNodes, hashCap; size_t bytes, parkOff; };

static inline __host__ __device__ size_t pl_al(size_t v) { return (v + 255) & ~(size_t)255; }

static inline __host__ __device__ PlanDims plan_dims(int32_t S, int32_t Sy, int32_t maxNodes)
{
    PlanDims d;
    d.idCap = (int64_t)S * (Sy + 3) + 16;
    d.rowCap = Sy + 4;
    d.maxNodes = maxNodes;
    int64_t h = 1;
    while (h < 2 * (int64_t)maxNodes) h <<= 1;
    d.hashCap = h;
    size_t b = 0;
    b += pl_al((size_t)d.idCap * 4);
    b += pl_al((size_t)d.idCap);
    b += pl_al((size_t)d.rowCap * 8);
    b += pl_al((size_t)PL_NQ * PL_QCAP * 8);
    b += pl_al((size_t)maxNodes * sizeof(PlNode));
    b += pl_al((size_t)maxNodes * sizeof(PlHeapEnt));
    b += pl_al((size_t)d.hashCap * 4);
    b += pl_al((size_t)PL_RS_CAP * 3 * 8);
    b += pl_al((size_t)PL_RS_CAP);
    d.parkOff = b;
    b += pl_al((size_t)PL_PARK_BYTES);
    d.bytes = b;
    return d;
}

static inline __device__ PlanWs plan_carve(char* base, const PlanDims& d)
{
    PlanWs w;
    size_t o = 0;
    w.dist = (uint32_t*)(base + o); o += pl_al((size_t)d.idCap * 4);
    w.flags = (uint8_t*)(base + o); o += pl_al((size_t)d.idCap);
    w.aliasKey = (unsigned long long*)(base + o); o += pl_al((size_t)d.rowCap * 8);
    w.queue = (unsigned long long*)(base + o); o += pl_al((size_t)PL_NQ * PL_QCAP * 8);
    w.nodes = (PlNode*)(base + o); o += pl_al((size_t)d.maxNodes * sizeof(PlNode));
    w.heap = (PlHeapEnt*)(base + o); o += pl_al((size_t)d.maxNodes * sizeof(PlHeapEnt));
    w.hash = (uint32_t*)(base + o); o += pl_al((size_t)d.hashCap * 4);
    w.rsbuf = (double*)(base + o); o += pl_al((size_t)PL_RS_CAP * 3 * 8);
    w.rsdir = (int8_t*)(base + o);
    w.park = base + d.parkOff;
    return w;
}

// ---- expansion lookahead (plan_kernel<.., LOOK = true>) -------------------------------------------------------------
// Everything the expansion of a node costs before the sequential resolution -- the children's poses, their sub-step
// collision checks, the 11 Reeds-Shepp solves, the sampled shot and its collision checks -- is a pure function of
// (node pose, goal, map, params). Workgroups that have no problem of their own (the batch is smaller than the chip, or
// their problems are finished) serve as HELPERS: the owners post the nodes at the top of their open lists to a job
// ring, a helper computes the expansion record with the very same device code, and the owner, when it pops that node
// later, reads the record instead of recomputing (a record is used only if it is keyed with the node's exact pose
// bits; whether one exists changes the time, never a result).
#define PL_REC_WORDS 88               // u64 words of one record: 5 x 16 per-child words + 8 header words
#define PL_JOB_WORDS 8
#define PL_JCAP 65536                 // entries of the job ring
#ifndef PL_LOOK_TOP
#define PL_LOOK_TOP 16                // heap slots an owner posts per pop
#endif
#ifndef PL_LOOK_SLEEP
#define PL_LOOK_SLEEP 8                 // s_sleep argument of a helper waiting for its job (x 64 cycles). Rounds 2 - 5: 64 (16 .. 127 measured alike); with predicted
                                        // children in the rings a record's lead is ~2 pops and the half a sleep a job waits for its helper shows: 8 measures 0.5 % faster
#endif
#define PL_LOOK_HRS (pl_al((size_t)PL_RS_CAP * 3 * 8) + pl_al((size_t)PL_RS_CAP))   // sample scratch of a helper-only workgroup
#ifndef PL_LOOK_KSPAN
#define PL_LOOK_KSPAN 1                // the children posted ahead steer within this many steps of their parent
#endif
#define PL_LOOK_KIDS (2 * PL_LOOK_KSPAN + 1)   // child records per node: same gear, steering index -1 / 0 / +1 from the node's own
#ifndef PL_LOOK_KIDS_ON_HIT
#define PL_LOOK_KIDS_ON_HIT 0         // post the likely children on record pops too (they then need PL_LOOK_WAIT to be of use)
#endif
#ifndef PL_LOOK_BACKLOG
#define PL_LOOK_BACKLOG 16             // child jobs are posted only while at most this many jobs wait in a ring
#endif
#ifndef PL_LOOK_FAULT
#define PL_LOOK_FAULT 0               // test builds (scripts/look_soak.py): > 0 = a helper publishes the records of the nodes divisible by it with ONE
#endif                                // key word flipped behind the ready bit's back -- the owner must turn them down (same results, fewer records used)
#ifndef PL_LOOK_LATE
#define PL_LOOK_LATE 1                // a long pop adopts its node's record when that lands while the pop is under way (pl_look_late)
#endif
#ifndef PL_LOOK_WAIT
#define PL_LOOK_WAIT 15000            // cycles an owner waits for a record that is posted but not finished (round 3: 0 / 10 k / 20 k: 21.4 / 20.7 / 20.7 ms; round 6, with predicted children: 10 k / 15 k / 20 k within 0.5 %)
#endif
struct PlLook {
    unsigned long long* ctrl;         // ring r (0: children halves, 1: shot halves): [64 r] tail, [64 r + 16] head; [32] problems finished,
                                      // [48] helpers alive (one 128-B line each); [8], [24], [88], [72 ..] diagnostics
    unsigned long long* jobs;         // [2][PL_JCAP][PL_JOB_WORDS]: PL_JOB_W0 (tag, owner's workgroup, gear, chain depth), pose, threshold, problem, -, sequence number
    unsigned long long* state;        // [entries]: tag << 8 | generation << 3 | bit 0 posted, bit 1 children half ready, bit 2 shot half ready
    unsigned long long* recs;         // [entries][PL_REC_WORDS]
    char* hrs;                        // [helper-only workgroups][PL_LOOK_HRS]
    int32_t on, main_blocks;          // workgroups [0, main_blocks) own a workspace slot and take problems
    uint32_t emask, pad;              // entries - 1 (a power of two)
};
// Every expansion is posted as TWO jobs that two helpers serve at the same time -- the children half (children poses,
// sub-step checks, the children's Reeds-Shepp lengths) and the shot half (the node's own Reeds-Shepp path, sampled and
// checked) -- so a record is there after ~35 k cycles instead of ~55 k; a helper serves one kind only (even / odd
// workgroups), which keeps its RS word schedule fixed.
//
// THE RECORD STORE (round 6) is a direct-mapped table of fixed size -- PL_LOOK_ENTRIES records whatever the batch size and
// the node arena (until round 5: one slot per (problem, arena node, 4): n x max_nodes x 2 832 B, 26 GB at pop cap 3 000, which
// silently switched the lookahead off). A record is named by its TAG = 48 bits of the hash of the POSE it expands, mixed with the
// problem (pl_look_tag) -- so a node that does not exist yet has the same name as the arena node it will be, whoever posts it:
// searches dive (a quarter of all pops expand a child of the node popped just before, too soon after its creation for a job
// posted then), so children are posted ahead of their creation: the three that keep the gear and steer within one step at the
// start of a pop that takes the long way, every child that beats the rest of the open list as soon as the parent's own record is
// in the owner's hands (pl_look_predict), and the next levels of such a dive by the helpers themselves (pl_look_chain).
// The entry of a tag is hash(tag); its state word carries the tag, so two tags that share an entry never share a record:
//   * an owner CLAIMS the entry for a tag with a compare-and-swap before it posts the job (pl_look_claim): an empty entry or
//     one whose record is COMPLETE (both halves there: nobody will write it any more) may be taken over, an entry whose jobs
//     are still in flight may not (that post is skipped: the pop goes the long way, as it would without the lookahead);
//   * helpers write the payload, drain, then OR their ready bit into the state word -- the entry is theirs until both bits are set;
//   * a reader takes the state word, copies the record, and takes the state word AGAIN (seqlock): any take-over starts
//     with the claim's change of that word (tag, generation count), which precedes the new helper's first payload store by a
//     trip through the job ring, so a copy framed by two equal state words is a copy of ONE record. The key words inside the
//     record (exact pose bits, problem, goal) are still checked: they are what ties the record to the node's pose.
// Whether a record exists, is evicted or is refused changes the time of a pop, never a result (tests/test_gpu_lookahead.py).
// When helpers are scarce -- fewer than PL_LOOK_SCARCE_X4 / 4 per workgroup still planning: the tail of a batch larger than the chip, or a batch
// of mostly long searches -- the blanket posting of the first PL_LOOK_TOP heap slots only lengthens the queue the urgent jobs wait in: the records
// all come late (768 problems: 1.7 % record pops at 106 k jobs, 36 ms against 34 without the lookahead). Then an owner posts the first
// PL_LOOK_TOP_BUSY slots while more than PL_LOOK_BACKLOG jobs wait, and nothing while more than PL_LOOK_BACKLOG2 do (768 problems: 36 % record
// pops, 26.5 ms; scripts/look_scale.py, profiles/NOTEBOOK.md). With that no gate on the helpers : owners ratio is needed (PL_LOOK_RATIO_X4 = 0;
// 2 .. 12 measured before the narrowing existed: 4 was the best then).
#ifndef PL_LOOK_SCARCE_X4
#define PL_LOOK_SCARCE_X4 8          // "scarce": 4 x helpers < this x the workgroups that are no helpers yet
#endif
#ifndef PL_LOOK_B2_DIV
#define PL_LOOK_B2_DIV 0             // (> 0: PL_LOOK_BACKLOG2 = helpers / this, at least 8 -- measured, no better than the fixed 32)
#endif
#ifndef PL_LOOK_TOP_BUSY
#define PL_LOOK_TOP_BUSY 8
#endif
#ifndef PL_LOOK_BACKLOG2
#define PL_LOOK_BACKLOG2 32
#endif
#ifndef PL_LOOK_TOP_BUSY2
#define PL_LOOK_TOP_BUSY2 0
#endif
#ifndef PL_LOOK_RATIO_X4
#define PL_LOOK_RATIO_X4 0           // the owners use the lookahead once 4 x helpers >= this x the workgroups that are no helpers yet (0: from the first helper on)
#endif
#ifndef PL_LOOK_ENT_LOG2
#define PL_LOOK_ENT_LOG2 18            // 262 144 records x 704 B = 184 MB (config[1] keeps ~3 000 alive; a launch zeroes the 2 MB of state words)
#endif
#define PL_LOOK_ENTRIES (1u << PL_LOOK_ENT_LOG2)
// (the number of entries is a property of the launch -- PlLook::emask --: avp_plan_set_look_entries shrinks the store for the tests that
//  want tags to collide and entries to be taken over all the time; entries = a power of two)
static inline __host__ __device__ size_t pl_look_state_bytes(uint32_t entries) { return pl_al((size_t)entries * 8); }
static inline __host__ __device__ size_t pl_look_bytes(uint32_t entries, int32_t helper_blocks)
{
    return 1024 + 2 * (size_t)PL_JCAP * PL_JOB_WORDS * 8 + pl_look_state_bytes(entries) +
           (size_t)entries * PL_REC_WORDS * 8 + (size_t)helper_blocks * PL_LOOK_HRS;
}
#define PL_LOOK_NODE_MAX (1 << 30)     // (records are named by pose, not by arena index: no limit of their own)
#define PL_LOOK_PID_MAX (1 << 30)
__device__ __forceinline__ size_t pl_look_ent(const PlLook& look, unsigned long long tag)
{
    unsigned long long z = tag * 0x9E3779B97F4A7C15ULL;
    z ^= z >> 32; z *= 0xd6e8feb86659fd93ULL; z ^= z >> 29;
    return (size_t)(z & (unsigned long long)look.emask);
}
#define PL_JOB_TAG(w0) ((w0) & 0xffffffffffffull)
// agent-scope relaxed accesses (sc1): payload stores, s_waitcnt vmcnt(0), flag store on the producer side; flag load,
// then payload loads on the consumer side -- the "sc1 payload -> drained -> sc1 flag" hand-off of MI355X_MICROARCH.md
// (every access of both sides bypasses the non-coherent L1). PL_LOOK_ATOMICS = 1 builds the same protocol from
// release stores / acquire loads at agent scope instead (buffer_wbl2 / buffer_inv around every flag): measured
// (profiles/r03_lookahead_soak.json) it changes no result and costs time, so the default stays 0.
#ifndef PL_LOOK_ATOMICS
#define PL_LOOK_ATOMICS 0
#endif
#if PL_LOOK_ATOMICS
#define PL_LOOK_DRAIN() do { } while (0)
#define PL_FLAG_ST64(q, v) __hip_atomic_store((q), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
#define PL_FLAG_LD64(q) __hip_atomic_load((q), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
#define PL_FLAG_LD32(q) __hip_atomic_load((q), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
#define PL_FLAG_OR64(q, v) __hip_atomic_fetch_or((q), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
#else
#define PL_LOOK_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PL_FLAG_ST64(q, v) pl_st64((q), (v))
#define PL_FLAG_LD64(q) pl_ld64(q)
#define PL_FLAG_LD32(q) pl_ld32(q)
#define PL_FLAG_OR64(q, v) atomicOr((q), (v))
#endif
__device__ __forceinline__ unsigned long long pl_ld64(const unsigned long long* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pl_st64(unsigned long long* q, unsigned long long v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t pl_ld32(const uint32_t* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pl_st32(uint32_t* q, uint32_t v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long pl_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ double pl_unbits(unsigned long long v) { return __longlong_as_double((long long)v); }
// key word 83 of a record: the problem and the goal heading (the goal position has words of its own)
__device__ __forceinline__ unsigned long long pl_look_key3(int64_t pid, double goal_th) { return (unsigned long long)pid ^ (pl_bits(goal_th) * 0x9E3779B97F4A7C15ULL); }

// ---- what a collision pass reads of the map and the vehicle: one copy per workgroup in LDS ------------
// The passes are CALLED functions (pl_check_pass below): everything they need travels as three LDS addresses, so no
// kernel argument has to be handed through registers or the stack, and their register demand is their own.
// register budget of plan_wave_kernel and of the functions it calls (the attribute propagates): 512 / PW_WAVES_PER_EU per lane
#ifndef PW_WAVES_PER_EU
#define PW_WAVES_PER_EU 4
#endif
#define PW_OCC __attribute__((amdgpu_waves_per_eu(PW_WAVES_PER_EU, PW_WAVES_PER_EU)))
struct PlChkEnv {
    int32_t nx, ny, wpc, kind;                        // kind: avp_params.checker_kind
    double b0, dx, b2, dy;
    double fp_xr, fp_xf, fp_yr, fp_yl;                // inflated footprint (map/costmap.py:97-101)
    double circ_rd, circ_cf, circ_cr;                 // two-circle model (collision_check.py:92-98)
    uint32_t lX, lY, lBits, maybe_wide;               // LDS addresses of the staged tables (STAGE); maybe_wide: a footprint's AABB can span > 64 map columns or > 2 bitmap words of rows on this map
    uint32_t big, big_pad;                            // more than 8 191 nodes on an axis: cell indices do not fit the queue's 13 bits -- the passes take the lane-per-pose form
    const double* gX; const double* gY; const uint64_t* gBits;   // ... or the tables in HBM / L2
};
// the map tables as a collision pass reads them: LDS copies (STAGE) or through L1 / L2
template <bool STAGE> struct PlTabs;
template <> struct PlTabs<true> {
    AVP_LDS const double* X; AVP_LDS const double* Y; AVP_LDS const uint64_t* bits;
    __device__ __forceinline__ PlTabs(const PlChkEnv& e) : X((AVP_LDS const double*)(uintptr_t)e.lX), Y((AVP_LDS const double*)(uintptr_t)e.lY), bits((AVP_LDS const uint64_t*)(uintptr_t)e.lBits) {}
};
template <> struct PlTabs<false> {
    const double* X; const double* Y; const uint64_t* bits;
    __device__ __forceinline__ PlTabs(const PlChkEnv& e) : X(e.gX), Y(e.gY), bits(e.gBits) {}
};
__device__ __forceinline__ void pl_chk_env_fill(PlChkEnv& e, const DevMap& m, const avp_params& p, const void* lX, const void* lY, const void* lBits)
{
    e.nx = m.nx; e.ny = m.ny; e.wpc = m.wpc; e.kind = p.checker_kind;
    e.b0 = m.b0; e.dx = m.dx; e.b2 = m.b2; e.dy = m.dy;
    e.fp_xr = p.fp_xr; e.fp_xf = p.fp_xf; e.fp_yr = p.fp_yr; e.fp_yl = p.fp_yl;
    e.circ_rd = p.circ_rd; e.circ_cf = p.circ_cf; e.circ_cr = p.circ_cr;
    e.lX = lX ? (uint32_t)(uintptr_t)(AVP_LDS const void*)lX : 0u; e.lY = lY ? (uint32_t)(uintptr_t)(AVP_LDS const void*)lY : 0u;
    e.lBits = lBits ? (uint32_t)(uintptr_t)(AVP_LDS const void*)lBits : 0u;
    {
        // an AABB is at most the inflated rectangle's diagonal wide and tall (+ 1 node for the inclusive ends, + 1 for the grid's
        // phase): below 62 cells on both axes no pose of this map ever needs more than 64 columns or two 64-row bitmap words
        const double diag = sqrt((p.fp_xf - p.fp_xr) * (p.fp_xf - p.fp_xr) + (p.fp_yl - p.fp_yr) * (p.fp_yl - p.fp_yr));
        e.maybe_wide = (diag / m.dx + 3.0 < 64.0 && diag / m.dy + 3.0 < 64.0) ? 0u : 1u;
    }
    e.big = (m.nx > 8191 || m.ny > 8191) ? 1u : 0u; e.big_pad = 0;
    e.gX = m.X; e.gY = m.Y; e.gBits = m.colBits;
}

// ---- lane-per-pose collision test through the column bitmaps (the fallback of a pass; the two-circle model) -----
template <int KIND, class TX, class TB>
__device__ __forceinline__ bool pl_check_pose(const PlChkEnv& e, TX X, TX Y, TB colBits, double x, double y, double th, double cs, double sn)
{
    if constexpr (KIND == 1) {
        const double Rd = e.circ_rd;
        const AvpRd2 rd2 = avp_circle_rd2(Rd);
        const double fx = x + e.circ_cf * cs, fy = y + e.circ_cf * sn;
        const double rx = x + e.circ_cr * cs, ry = y + e.circ_cr * sn;
        double right, left, upper, down;
        if (fx >= rx) { right = fx + Rd; left = rx - Rd; } else { right = rx + Rd; left = fx - Rd; }
        if (fy >= ry) { upper = fy + Rd; down = ry - Rd; } else { upper = ry + Rd; down = fy - Rd; }
        const int ixlo = avp_first_gt(X, e.nx, e.b0, e.dx, left), ixhi = avp_last_lt(X, e.nx, e.b0, e.dx, right);
        const int iylo = avp_first_gt(Y, e.ny, e.b2, e.dy, down), iyhi = avp_last_lt(Y, e.ny, e.b2, e.dy, upper);
        if (iylo > iyhi) return false;
        for (int ix = ixlo; ix <= ixhi; ix++) {
            const double px = X[ix];
            for (int w = iylo >> 6; w <= (iyhi >> 6); w++) {
                uint64_t bits = colBits[(size_t)ix * e.wpc + w];
                if (w == (iylo >> 6)) bits &= ~0ull << (iylo & 63);
                if (w == (iyhi >> 6)) bits &= ~0ull >> (63 - (iyhi & 63));
                while (bits) {
                    const int bpos = __ffsll((unsigned long long)bits) - 1;
                    bits &= bits - 1;
                    const double py = Y[(w << 6) + bpos];
                    const double d0x = px - fx, d0y = py - fy, d1x = px - rx, d1y = py - ry;
                    if (avp_circle_hit2(d0x, d0y, Rd, rd2)) return true;
                    if (avp_circle_hit2(d1x, d1y, Rd, rd2)) return true;
                }
            }
        }
        return false;
    } else {
    Footprint f;
    avp_footprint_setup_cs(e, x, y, cs, sn, f);
    double xmin, xmax, ymin, ymax;
    avp_footprint_aabb(f, xmin, xmax, ymin, ymax);
    const int ixlo = avp_first_ge(X, e.nx, e.b0, e.dx, xmin), ixhi = avp_last_le(X, e.nx, e.b0, e.dx, xmax);
    const int iylo = avp_first_ge(Y, e.ny, e.b2, e.dy, ymin), iyhi = avp_last_le(Y, e.ny, e.b2, e.dy, ymax);
    if (iylo > iyhi) return false;
    for (int ix = ixlo; ix <= ixhi; ix++) {
        const double px = X[ix];
        for (int w = iylo >> 6; w <= (iyhi >> 6); w++) {
            uint64_t bits = colBits[(size_t)ix * e.wpc + w];
            if (w == (iylo >> 6)) bits &= ~0ull << (iylo & 63);
            if (w == (iyhi >> 6)) bits &= ~0ull >> (63 - (iyhi & 63));
            while (bits) {
                const int bpos = __ffsll((unsigned long long)bits) - 1;
                bits &= bits - 1;
                if (avp_footprint_point_hit(f, px, Y[(w << 6) + bpos])) return true;
            }
        }
    }
    return false;
    }
}

// ---- shared (LDS) state of one problem ----------------------------------------------------------
struct PlChild {
    double x, y, th;                  // child pose
    double L;                         // RS length to the goal [m]
    int64_t id;                       // grid id of (x, y)
    int32_t found;                    // node position of an equal open/closed node, -1 none
    int32_t first_coll;               // first colliding sub-step, -1 none
    int8_t found_state, oob, rs_err, pad;
    uint32_t pre_d;                   // dist[id] read ahead of the sequential resolution
    // fast-path resolution (all heuristic queries hit): class, arena slot, costs, prefetched open-node fields
    int32_t cls, pos, old_heap_pos;
    double g, h, f, old_f;
};
enum { CL_SKIP = 0, CL_NEW_CLOSED = 1, CL_NEW_OPEN = 2, CL_IMPROVE = 3, CL_KEEP = 4 };

// Map tables used by the collision passes: either the HBM/L2 copies or the LDS-staged copies.
struct MapTabs { const double* X; const double* Y; const uint64_t* bits; };

#define PL_WPOSE 8                    // poses per wave per collision pass
#ifndef PL_WPOSE0
#define PL_WPOSE0 3                   // ... in the first round over the shot's samples
#endif
#ifndef PL_SHOT_WAVES
#define PL_SHOT_WAVES 8               // waves that sample and check the shot (at most; wave 0 is busy with the resolution)
#endif
#ifndef PL_SUB_WAVES
#define PL_SUB_WAVES 6                // waves that check the sub-step poses (waves 1 ..)
#endif
#ifndef PL_WQCAP
#define PL_WQCAP (PL_THREADS >= 512 ? 1024 : 512)   // (pose, point) candidates per wave; more fall back to the lane-per-pose walk
#endif
template <int QCAP>
struct PlWaveChkT {
    static constexpr int WQCAP = QCAP;
    Footprint fp[PL_WPOSE];
    int16_t rng[PL_WPOSE][4];         // ixlo, ixhi, iylo, iyhi
    uint32_t hit[PL_WPOSE];
    double pose[PL_WPOSE][5];         // x, y, theta, cos, sin of the poses of the pass (staged by the caller)
    int32_t qn, over;
    uint32_t q[QCAP];                 // pose << 26 | ix << 13 | iy   (PL_WPOSE <= 8 poses; nx, ny <= 8191: larger maps take the lane-per-pose form, PlChkEnv::big)
};
typedef PlWaveChkT<PL_WQCAP> PlWaveChk;

struct PlShared {
    // lattice / id space (compute_h.py lattice anchored at the goal)
    int32_t col0, row0, colMin, colMax, rowMin, rowMax, orow0, alias;   // alias: a lattice column has col == S
    int64_t goal_id;
    int32_t regular;
    // sweep
    int32_t E;                        // buckets [0, E) are expanded
    uint32_t qcount[PL_NQ], qbase[PL_NQ];   // pushes so far / entries consumed so far, per rotating bucket queue
    int32_t qover;
    static constexpr bool RELAX_PRECHECK = false;       // (pl_hquery_miss: the sweep of this form is latency bound)
    uint32_t dF; int64_t idF;         // key of the last miss (closed frontier)
    int32_t hasF;
    int64_t h_cells, h_misses;
    // A*
    int32_t nnodes, nheap, nclosed, closed_nonempty;
    int64_t global_index;
    int32_t cur;                      // node position being expanded
    int32_t status, done;
    double goal[3];
    int32_t pid;
    // per pop scratch
    int32_t rs_status, rs_npts, rs_first_coll, in_radius, collision;
    RsPath rs;                        // normalised winner of the shot
    int64_t n_checks, n_rs;
    int64_t snap[5];                  // counters saved before a speculative resolution
    unsigned long long fold_key[PL_THREADS / 64];   // per-wave scratch of pl_rs_fold_wave
    int32_t fold_idx[PL_THREADS / 64];
    MapTabs mt;                       // the map tables as the kernel sees them (LDS copies when staged)
    PlChkEnv env;                     // what the called collision passes read of the map and the vehicle
    // copies of the kernel's arguments for the CALLED parts of plan_kernel (set-up, sweep extension, result record): they
    // take this struct's LDS address and nothing else, so their registers -- and their code -- stay out of the pop loop
    DevMap km; avp_params kp; PlanDims kdims; PlanWs kw; PlLook klook;
    const double* k_starts; const double* k_goals; avp_plan_result_dev* k_results; double* k_paths; int32_t k_max_path, k_pad;
    uint32_t chk_arrived;             // software barrier of the waves that check the shot's samples
    double k_steer[AVP_MAX_STEER], k_dth_dt[AVP_MAX_STEER], k_dth_ddt1[AVP_MAX_STEER], k_travel_ddt1;   // lane-indexed motion-primitive constants (copy of avp_params); sub-step j: x (j + 1)
    int8_t sub_child[PL_MAXSUBS], sub_j[PL_MAXSUBS], sub_steer[PL_MAXSUBS];   // sub-step t -> child, step, steer index (no integer divisions per pose)
    int32_t shot_ready;               // 0 = the shot's arg-min is pending, 1 = s.rs holds its path, 2 = no shot
    long long phase[PH_COUNT];
    uint32_t hq_d;                    // result of the collective query
    int32_t hq_flag;
    // sequential child resolution state machine (thread 0 runs alone between sweep extensions)
    int32_t next_child, need_sweep, have_d, fast;
    int64_t pending_id;
    PlChild child[PL_MAXCHILD];
    // RS word results: [query][word] ok + 5 lengths; kept candidates per query
    RsFrame frame[PL_RSQ];
    int32_t sched_cnt, sched_n;       // lane schedule of the (word, query) items for sched_cnt queries
    uint16_t sched[PL_SCHED_ROUNDS * PL_THREADS];   // [round][wave][lane] -> word << 4 | query, 0xffff = idle lane
    int32_t sched_load[PL_THREADS / 64], sched_rounds[PL_THREADS / 64];   // work arrays of pl_rs_build_schedule
    uint8_t w_ok[PL_RSQ * 46];        // word valid
    uint8_t w_acc[PL_RSQ * 46];       // word accepted by set_path
    uint8_t w_err[PL_RSQ];            // assertion L >= 0.01 failed for an accepted word
    double w_l[PL_RSQ * 46][5];
    double w_Lm[PL_RSQ * 46];         // its length in metres (L / maxc)
    // RS sampling: per output index the last writer (length argument, segment), segment origins
    int32_t smp_hi, smp_point_num;
    double smp_l[PL_RS_CAP];
    int8_t smp_seg[PL_RS_CAP];
    double seg_o[AVP_RS_MAXSEG][3];
    // wave-local cooperative collision passes: every wave owns a scratch area (no workgroup barrier inside)
    int32_t chk_qover;
    PlWaveChk wchk[PL_THREADS / 64];
    static constexpr int RS_CAP = PL_RS_CAP;
    static constexpr bool HEAP_POS = PL_HEAP_POS != 0;
    static constexpr int HEAP_LDS = PL_HEAP_LDS;
    double hl_f[PL_HEAP_LDS > 0 ? PL_HEAP_LDS : 1];      // the first HEAP_LDS entries of the open list's heap: keys,
    uint32_t hl_n[PL_HEAP_LDS > 0 ? PL_HEAP_LDS : 1];    // ... nodes
    static constexpr bool POINT_FAST = false;         // (pl_check_narrow)
    __device__ __forceinline__ PlWaveChk& wave_chk() { return wchk[threadIdx.x >> 6]; }
    uint32_t chk_hit[PL_MAXSUBS];   // hit flag per sub-step pose of the current pop
    int32_t next_cur, have_next;      // node popped ahead by wave 0 at the end of its resolution (see pl_resolve_fast_wave)
    // expansion lookahead
    unsigned long long recb[3][PL_REC_WORDS];   // records (owner): [rec_cur] (0 / 1) = the popped node's, the other of the two = the prefetch target; [2] = the open list's second-best node's (pl_look_second)
    int32_t fetch_go, fetch_nheap;              // pop-ahead -> record fetch hand-over (see pl_resolve_fast_wave)
    int32_t wr_go, wr_done;                     // classification -> writer wave hand-over
    int32_t nf_node, nf_found[16]; int8_t nf_state[16];   // the popped-ahead node's children: pose-hash look-ups done by the fetching wave
    int32_t look_calm, look_live;               // as of the last posting round: helpers alive and at most PL_LOOK_BACKLOG jobs waiting / helpers alive
    int32_t rec_cur, pre_node, pre_ok;          // prefetched: the record of node pre_node sits in recb[rec_cur ^ 1] (pre_ok)
    unsigned long long job[PL_JOB_WORDS];   // the job being served (helper)
    int32_t use_rec, job_skip, helper_reg, n_hits, n_sec[4];
    int32_t n_miss[6];                          // diagnostics (pl_look_load)
    unsigned long long poll_tag; int32_t n_pred, n_busy, n_torn, look_pad;      // the pending record's tag; diagnostics: children posted by pl_look_predict, claims refused (entry busy), copies refused by the seqlock
    static constexpr bool LOOK_SECOND = true;   // (pl_resolve_fast_wave publishes fetch_second)
    double fetch_second;                        // the open list's second-best key as of the pop-ahead that named next_cur (pl_resolve_fast_wave -> pl_look_fetch)
    unsigned long long poll_ri; int32_t poll_on, late_rec, n_late, late_rec2;  // PL_LOOK_LATE: the pending record of the node being expanded the long way (plk_look_late);
                                                                               // late_rec: adopted at the first poll (read at exit #1 only), late_rec2: at the second (read at exit #2 only)
};

static_assert(sizeof(PlShared) <= 160 * 1024, "PlShared must fit the 160 KiB LDS of a CU");
static inline __host__ __device__ size_t pl_lds_tables_offset() { return (sizeof(PlShared) + 15) & ~(size_t)15; }
static inline size_t pl_lds_tables_bytes(const DevMap& m) { return ((size_t)m.nx * m.wpc + m.nx + m.ny) * 8; }

AVP_D int32_t pl_bucket(uint32_t d) { return (int32_t)(d / 10u); }
// AVP_PLAN_BAD_POSE: coordinates that are not finite; headings that are not finite or beyond 1e6 rad -- the reference's pi_2_pi loop (rs_curve.py:648-655)
// takes |theta| / 2 pi trips (> 1.6e5 beyond 1e6 rad; it never returns beyond ~1e16 or on inf / NaN): a narrower domain than the reference's, by choice
AVP_D bool pl_pose_ok(double x, double y, double th) { return fabs(x) <= 1.7e308 && fabs(y) <= 1.7e308 && fabs(th) <= 1e6; }      // (finite coordinates -- the BenchmarkCases' run to 9e9 m --, a heading the wrap loop finishes; NaN: false)

AVP_D uint64_t pl_mix(uint64_t z) { z ^= z >> 33; z *= 0xff51afd7ed558ccdULL; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ULL; z ^= z >> 33; return z; }
AVP_D uint64_t pl_pose_hash(double x, double y, double th)
{
    const double zx = x == 0.0 ? 0.0 : x, zy = y == 0.0 ? 0.0 : y, zt = th == 0.0 ? 0.0 : th;
    return pl_mix(avp_d2u(zx) * 0x9E3779B97F4A7C15ULL ^ pl_mix(avp_d2u(zy) + 0x632BE59BD9B4E019ULL) ^ pl_mix(avp_d2u(zt) * 3));
}
// exact-equality lookup (hybrid_a_star.py:156,170); returns node position or -1
AVP_D int32_t pl_hash_find(const PlanWs& w, int64_t hashCap, double x, double y, double th)
{
    if (x != x || y != y || th != th) return -1;
    uint64_t h = pl_pose_hash(x, y, th) & (uint64_t)(hashCap - 1);
    for (;;) {
        const uint32_t v = w.hash[h];
        if (v == 0) return -1;
        const PlNode& n = w.nodes[v - 1];
        if (n.x == x && n.y == y && n.th == th) return (int32_t)(v - 1);
        h = (h + 1) & (uint64_t)(hashCap - 1);
    }
}
AVP_D void pl_hash_put(const PlanWs& w, int64_t hashCap, int32_t pos)
{
    const PlNode& n = w.nodes[pos];
    if (n.x != n.x || n.y != n.y || n.th != n.th) return;
    uint64_t h = pl_pose_hash(n.x, n.y, n.th) & (uint64_t)(hashCap - 1);
    while (w.hash[h] != 0) h = (h + 1) & (uint64_t)(hashCap - 1);
    w.hash[h] = (uint32_t)pos + 1;
}

// parallel-safe insert (fast-path resolution: several lanes insert at once; order is irrelevant)
AVP_D void pl_hash_put_atomic(const PlanWs& w, int64_t hashCap, int32_t pos, double x, double y, double th)
{
    if (x != x || y != y || th != th) return;
    uint64_t h = pl_pose_hash(x, y, th) & (uint64_t)(hashCap - 1);
    while (atomicCAS(&w.hash[h], 0u, (uint32_t)pos + 1) != 0u) h = (h + 1) & (uint64_t)(hashCap - 1);
}

// ---- CPython heapq on node positions, key = node.f (Node.__lt__ hybrid_a_star.py:61-68) ---------
// The key is stored next to the position (one load per comparison); an in-place improvement of an
// open node updates both copies and, like the reference (:224-230), does NOT restore the heap order.
template <class S>
AVP_D PlHeapEnt pl_heap_get(const PlanWs& w, const S& s, int32_t pos)
{
    if constexpr (S::HEAP_LDS > 0) if (pos < S::HEAP_LDS) { PlHeapEnt e; e.f = s.hl_f[pos]; e.node = s.hl_n[pos]; e.pad = 0; return e; }
    return w.heap[pos];
}
// S::HEAP_POS: every node keeps the slot of its heap entry (a global store per moved entry); without it an open node's
// entry is found by a search when its key is lowered in place (an equal pose reached again at a lower cost: measured
// 0.4 times per pop on the bench workload -- too often for a search, hence 1).
template <class S>
AVP_D void pl_heap_set(const PlanWs& w, S& s, int32_t pos, PlHeapEnt e)
{
    bool lds = false;
    if constexpr (S::HEAP_LDS > 0) if (pos < S::HEAP_LDS) { s.hl_f[pos] = e.f; s.hl_n[pos] = e.node; lds = true; }
    if (!lds) w.heap[pos] = e;
    if constexpr (S::HEAP_POS) w.nodes[e.node].heap_pos = pos;
}
template <class S>
AVP_D void pl_heap_set_key(const PlanWs& w, S& s, int32_t pos, double f)
{
    if constexpr (S::HEAP_LDS > 0) if (pos < S::HEAP_LDS) { s.hl_f[pos] = f; return; }
    w.heap[pos].f = f;
}
// slot of `node`'s entry among the first nheap: by one thread / by a whole wave (every lane gets the answer)
template <class S>
AVP_D int32_t pl_heap_find(const PlanWs& w, const S& s, int32_t nheap, uint32_t node)
{
    for (int32_t i = 0; i < nheap; i++) if (pl_heap_get(w, s, i).node == node) return i;
    return -1;
}
template <class S>
AVP_D int32_t pl_heap_find_wave(const PlanWs& w, const S& s, int32_t nheap, uint32_t node)
{
    const int lane = threadIdx.x & 63;
    for (int32_t base = 0; base < nheap; base += 64) {
        const bool hit = base + lane < nheap && pl_heap_get(w, s, base + lane).node == node;
        const unsigned long long mk = __ballot(hit);
        if (mk) return base + __ffsll((unsigned long long)mk) - 1;
    }
    return -1;
}
// (the item being moved is passed in registers: re-reading a slot this thread has just written would put
// two round trips on the serial path of every push / pop)
template <class S>
AVP_D void pl_siftdown(const PlanWs& w, S& s, int32_t startpos, int32_t pos, const PlHeapEnt newitem)
{
    while (pos > startpos) {
        const int32_t parentpos = (pos - 1) >> 1;
        const PlHeapEnt parent = pl_heap_get(w, s, parentpos);
        if (newitem.f < parent.f) { pl_heap_set(w, s, pos, parent); pos = parentpos; continue; }
        break;
    }
    pl_heap_set(w, s, pos, newitem);
}
template <class S>
AVP_D void pl_siftup(const PlanWs& w, S& s, int32_t pos, int32_t endpos, const PlHeapEnt newitem)
{
    const int32_t startpos = pos;
    int32_t childpos = 2 * pos + 1;
    while (childpos < endpos) {
        const int32_t rightpos = childpos + 1;
        PlHeapEnt c = pl_heap_get(w, s, childpos);
        if (rightpos < endpos) {
            const PlHeapEnt r = pl_heap_get(w, s, rightpos);
            if (!(c.f < r.f)) { childpos = rightpos; c = r; }
        }
        pl_heap_set(w, s, pos, c);
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    pl_siftdown(w, s, startpos, pos, newitem);
}
template <class S>
AVP_D void pl_heap_push(const PlanWs& w, S& s, uint32_t node, double f)
{
    PlHeapEnt e; e.f = f; e.node = node; e.pad = 0;
    s.nheap++;
    pl_siftdown(w, s, 0, s.nheap - 1, e);
}
// heappush by a whole wave (all 64 lanes call it with the same arguments; nheap = the entry count BEFORE the push, held
// in a register by every lane). _siftdown only ever compares the new item with the ancestors of its slot, and those do
// not change while it climbs: lane i fetches ancestor i of the slot (one round trip for the whole root path instead of
// one per level), a ballot finds the first ancestor the item does not beat, the ancestors below it move down one step
// each (in parallel) and the item lands in the freed slot -- the same array the serial loop produces.
template <class S>
AVP_D void pl_heap_push_wave(const PlanWs& w, S& s, int32_t nheap, uint32_t node, double f)
{
    const int lane = threadIdx.x & 63;
    const uint32_t slot1 = (uint32_t)nheap + 1u;                 // 1-based index of the new slot
    const int depth = 31 - __clz(slot1);                          // number of ancestors
    // 1-based ancestor i+1 of the slot = slot1 >> (i + 1); its child on the path = slot1 >> i
    const int32_t mine = (int32_t)(slot1 >> (lane < 31 ? lane : 31)) - 1;           // path[i]   (0-based)
    const int32_t par = (int32_t)(slot1 >> (lane < 30 ? lane + 1 : 31)) - 1;       // path[i+1]
    PlHeapEnt anc; anc.f = 0.0; anc.node = 0; anc.pad = 0;
    const bool act = lane < depth;
    if (act) anc = pl_heap_get(w, s, par);
    const unsigned long long stop = __ballot(act && !(f < anc.f));
    const int j = stop ? __ffsll((unsigned long long)stop) - 1 : depth;   // the item ends at path[j]
    if (lane < j) pl_heap_set(w, s, mine, anc);
    if (lane == j) { PlHeapEnt e; e.f = f; e.node = node; e.pad = 0; pl_heap_set(w, s, mine, e); }
}
template <class S>
AVP_D uint32_t pl_heap_pop(const PlanWs& w, S& s)
{
    const int32_t nh = --s.nheap;
    const PlHeapEnt lastelt = pl_heap_get(w, s, nh);
    if (nh) {
        const PlHeapEnt ret = pl_heap_get(w, s, 0);
        pl_siftup(w, s, 0, nh, lastelt);
        return ret.node;
    }
    return lastelt.node;
}

// heappop by a whole wave (all 64 lanes call it; nheap = the entry count BEFORE the pop, the same in every lane; returns
// the popped node to every lane; the caller stores the new count). heapq's _siftup walks the hole from the root to a leaf
// along the smaller children -- a chain of dependent loads, one round trip per level for a single lane. Here the wave
// fetches six levels of the subtree under the hole at once (lane r - 1 holds relative position r), walks five levels on
// register values (readlane), and repeats from where it stands: two round trips for 1 000 entries instead of ten. Lane k
// remembers path position k and the entry that moves up into it. The final _siftdown of the last element compares it with
// the path's ORIGINAL entries from the leaf upwards (each parent slot holds what was its child on the path) and stops at
// the first one it does not beat: a ballot; the slots below that point end with their original entries (no store), the
// ones above take their path child, the element lands in between -- the array the serial code leaves (PL_HEAP_POP_WAVE 0).
#ifndef PL_HEAP_POP_WAVE
#define PL_HEAP_POP_WAVE 1
#endif
AVP_D double pl_readlane_f64(double v, int l)          // l: the same in every lane (a scalar)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <class S>
AVP_D uint32_t pl_heap_pop_wave(const PlanWs& w, S& s, int32_t nheap)
{
    const int lane = threadIdx.x & 63;
    const uint32_t nh = (uint32_t)__builtin_amdgcn_readfirstlane(nheap - 1);     // (scalar: the walk below is uniform control flow)
    const PlHeapEnt last = pl_heap_get(w, s, (int32_t)nh);
    if (nh == 0) return last.node;
    const uint32_t r = (uint32_t)lane + 1u, k = 31u - (uint32_t)__clz(r);
    uint32_t P = 1, cur1 = 1, root_node = 0;     // 1-based: root of the fetched subtree, the hole
    int level = 0;
    int32_t my_pos = 0;
    PlHeapEnt my_ent; my_ent.f = 0.0; my_ent.node = 0; my_ent.pad = 0;
    for (;;) {
        const uint32_t abs1 = (P << k) + (r - (1u << k));
        PlHeapEnt e; e.f = 0.0; e.node = 0; e.pad = 0;
        if (r < 64u && abs1 <= nh) e = pl_heap_get(w, s, (int32_t)(abs1 - 1u));
        if (P == 1) root_node = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.node);
        uint32_t rr = 1;
        bool leaf = false;
        for (int step = 0; step < 5; step++) {
            const uint32_t left1 = 2u * cur1;
            if (left1 > nh) { leaf = true; break; }
            const int cl = (int)(2u * rr) - 1;                                   // lane of the left child; the right one is next to it
            double cf = pl_readlane_f64(e.f, cl);
            uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)e.node, cl);
            uint32_t right = 0;
            if (left1 + 1u <= nh) {
                const double rf = pl_readlane_f64(e.f, cl + 1);
                if (!(cf < rf)) { right = 1; cf = rf; cn = (uint32_t)__builtin_amdgcn_readlane((int)e.node, cl + 1); }
            }
            if (lane == level) { my_pos = (int32_t)(cur1 - 1u); my_ent.f = cf; my_ent.node = cn; }
            level++;
            cur1 = left1 + right; rr = 2u * rr + right;
        }
        if (leaf) break;
        P = cur1;
    }
    // lane t < level holds orig[pos_(t+1)]: the element stops at the lowest path index whose original entry it does not beat
    const unsigned long long stop = __ballot(lane < level && !(last.f < my_ent.f));
    const int j = stop ? 64 - __clzll((long long)stop) : 0;
    if (lane < j) pl_heap_set(w, s, my_pos, my_ent);
    if (lane == j) pl_heap_set(w, s, j < level ? my_pos : (int32_t)(cur1 - 1u), last);
    return root_node;
}

// ---- heuristic sweep ---------------------------------------------------------------------------
// Relax lattice cell (col, row) with new distance nd, discovered from (srcDist, srcId) via
// neighbour slot nbr (compute_h.py:216-235 add_grid_to_openlist).
template <class S>
AVP_D void pl_relax(const DevMap& m, const PlanWs& w, S& s, int col, int row, uint32_t nd, uint32_t srcDist,
                    int64_t srcId, int nbr)
{
    if (col < s.colMin || col > s.colMax || row < s.rowMin || row > s.rowMax) return;   // one-sided bounds tests :95-191
    // is_obstacle (compute_h.py:237-255): cell (ix-1, iy-1) with Python negative-index wrap
    int ox = col - 1;
    if (ox >= m.S) ox = m.S - 1;
    if (ox < 0) ox += m.nx;
    int oy = s.orow0 + (s.row0 - row);
    if (oy >= m.Sy) oy = m.Sy - 1;
    if (oy < 0) oy += m.ny;
    if ((s.mt.bits[(size_t)ox * m.wpc + (oy >> 6)] >> (oy & 63)) & 1ull) return;      // the column bitmaps (LDS when staged) hold the 255 cells
    const int64_t nid = (int64_t)col + (int64_t)row * m.S;
    if (s.alias && (col == 0 || col == m.S)) {
        const int slot = col == 0 ? row : row + 1;
        const unsigned long long key = ((unsigned long long)srcDist << 44) | ((unsigned long long)srcId << 8) |
                                       ((unsigned long long)nbr << 1) | (col == 0 ? 0ull : 1ull);
        atomicMin(&w.aliasKey[slot], key);
    }
#if PL_RELAX_PRECHECK
    // (distances only ever decrease: a plain load that already shows a value <= nd settles it without the atomic -- about
    // half the relaxations of a ring go backwards or sideways; a stale larger value just takes the atomic as before)
    if (w.dist[nid] <= nd) return;
#endif
    const uint32_t old = atomicMin(&w.dist[nid], nd);
    if (nd < old) {
        const int q = pl_bucket(nd) & (PL_NQ - 1);
        const uint32_t pos = atomicAdd(&s.qcount[q], 1u) - s.qbase[q];       // (qcount counts every push ever, qbase what has been consumed)
        if (pos < PL_QCAP) w.queue[(size_t)q * PL_QCAP + pos] = ((unsigned long long)nd << 32) | (unsigned long long)nid;
        else s.qover = 1;
    }
}

// Heuristic query (hybrid_a_star.py:268-283 + compute_h.py:198-214), split in two:
//  pl_hquery_hit : pure test against the closed frontier, callable by a single thread;
//  pl_hquery_miss: collective sweep extension until the cell's distance is final (all threads).
// force_miss: the initial compute_path(x0, y0) of hybrid_a_star.__init__ (:89-91) always sweeps.
AVP_D bool pl_id_in_range(const DevMap& m, int64_t id) { return id >= 0 && id < (int64_t)m.S * (m.Sy + 3); }
template <class S>
AVP_D bool pl_hquery_hit(const DevMap& m, const S& s, int64_t id, uint32_t d, uint32_t& d_out)
{
    // d = current dist[id] (PL_UNSEEN when the id is outside the id space)
    if (id == s.goal_id) { d_out = 0; return true; }                 // first closedlist entry: the goal Grid, distance 0
    if (!pl_id_in_range(m, id)) { d_out = PL_UNSEEN; return true; }
    if (s.hasF && d != PL_UNSEEN && (d < s.dF || (d == s.dF && id <= s.idF))) { d_out = d; return true; }
    return false;
}

// The sweep loop, ONE group barrier per expanded bucket (round 2 and the first half of round 3 spent three per bucket --
// read the count / expand / reset the count and advance -- and two per EMPTY bucket: ~400 buckets between goal and start,
// the whole cost of a short search). What makes the single barrier enough:
//  * the bucket queues are never reset: qcount[q] counts every push, qbase[q] what has been consumed; the entries of the
//    current use of queue q sit at [0, qcount - qbase). A relaxation out of bucket E pushes into buckets E + 1, E + 2 only
//    (edges weigh 10 .. 14, a bucket spans 10), never into queue E & 3 itself or into E + 3, so every count a thread reads at
//    the top of an iteration was final at the barrier before -- no thread needs to be told;
//  * every thread carries E and the four bases in registers and advances them identically; one thread mirrors the base of
//    the consumed queue to LDS before the barrier (pl_relax reads it two buckets later) and E at the end;
//  * the query cell's distance is not read until E reaches its lower bound -- the obstacle-free octile distance from the
//    goal: the grid distance cannot be smaller -- which saves a dependent global load per bucket for most of the way.
// Relaxations of a bucket commute (atomicMin on distances and alias keys, queue order is irrelevant), so the field, the
// closed frontier and every counter are what the three-barrier loop produced (tests/test_gpu_hfield.py, no tolerance).
template <bool PROFILE = false, class Coop = CoopWG, class S>
AVP_D void pl_hquery_miss(const DevMap& m, const PlanWs& w, S& s, int64_t id)
{
    Coop::sync();
    if (!(id >= 0 && id < (int64_t)m.S * (m.Sy + 3))) { if (Coop::tid() == 0) s.hq_d = PL_UNSEEN; Coop::sync(); return; }
    const long long t_sw = PH_NOW();
    int32_t E = s.E;
    uint32_t b0 = s.qbase[0], b1 = s.qbase[1], b2 = s.qbase[2], b3 = s.qbase[3];
    int32_t lbE = 0;
    {
        const uint32_t id32 = (uint32_t)id, row_u = id32 / (uint32_t)m.S;
        const int col = (int)(id32 - row_u * (uint32_t)m.S), row = (int)row_u;
        if (!(s.alias && col == 0)) {                       // (an aliased id may be the last-column cell of the row above: no bound)
            const int dcl = abs(col - s.col0), drw = abs(row - s.row0);
            lbE = (10 * max(dcl, drw) + 4 * min(dcl, drw)) / 10;
        }
    }
    // this lane's neighbour slot: 0..7 = (-1,-1) (0,-1) (1,-1) (-1,0) (1,0) (-1,1) (0,1) (1,1) in (col, row), y up = row down
    const int nbr = Coop::tid() & 7, nb9 = nbr < 4 ? nbr : nbr + 1;
    const int dcn = nb9 % 3 - 1, drn = nb9 / 3 - 1;
    const uint32_t costn = (dcn != 0 && drn != 0) ? 14u : 10u;
    for (;;) {
        if (E >= lbE) {
            const uint32_t d = w.dist[id];
            if (d != PL_UNSEEN && pl_bucket(d) <= E) break;
        }
        const uint32_t c0 = s.qcount[0] - b0, c1 = s.qcount[1] - b1, c2 = s.qcount[2] - b2, c3 = s.qcount[3] - b3;
        if (c0 + c1 + c2 + c3 == 0) break;
        const int q = E & (PL_NQ - 1);
        const uint32_t full = q == 0 ? c0 : q == 1 ? c1 : q == 2 ? c2 : c3;
        if (full == 0) { E += 1; continue; }              // an empty bucket costs nothing
        // A queue that overflowed (its pushes beyond PL_QCAP were dropped, s.qover set by the pusher) ends the sweep HERE,
        // when its bucket comes up: `full` is final at the barrier before, so every wave of the group takes this exit in
        // the same iteration. (Until round 4 the loop tested s.qover itself at the top: a fast wave could set it during
        // the pushes of the very iteration a slow wave was still entering, the slow wave left, the fast one went on to the
        // barrier below, and the group's software barriers were skewed by one for the rest of the search.) Buckets before
        // it are complete, so a query they settle is still answered; one they do not settle ends with status 5.
        if (full > (uint32_t)PL_QCAP) break;
        const uint32_t cnt = min(full, (uint32_t)PL_QCAP);
        // PL_SWEEP_U (entry, neighbour) pairs per lane and trip, stage by stage: the U queue loads, then the U distance/flag
        // loads, the U pre-check loads and the U atomics are each in flight together -- a trip costs one chain of memory
        // round trips whatever U is (Coop::N is a multiple of 8: a lane keeps its neighbour slot nbr for the whole sweep)
        uint32_t cells = 0;
        for (uint32_t p0 = Coop::tid(); p0 < cnt * 8u; p0 += Coop::N * PL_SWEEP_U) {
            unsigned long long ent[PL_SWEEP_U];
            bool v[PL_SWEEP_U];
#pragma unroll
            for (int k = 0; k < PL_SWEEP_U; k++) {
                const uint32_t p = p0 + (uint32_t)k * Coop::N;
                v[k] = p < cnt * 8u;
                ent[k] = v[k] ? w.queue[(size_t)q * PL_QCAP + (p >> 3)] : 0ull;
            }
            uint32_t cur[PL_SWEEP_U];
            uint8_t fl[PL_SWEEP_U];
#pragma unroll
            for (int k = 0; k < PL_SWEEP_U; k++) {
                const uint32_t eid = (uint32_t)ent[k];               // (0 for an idle pair: a valid index, result unused)
                cur[k] = w.dist[eid];
                fl[k] = w.flags[eid];
            }
            uint32_t nd[PL_SWEEP_U], nid[PL_SWEEP_U];
#pragma unroll
            for (int k = 0; k < PL_SWEEP_U; k++) {
                const uint32_t d = (uint32_t)(ent[k] >> 32), eid = (uint32_t)ent[k];
                // stale entry (distance was lowered later) / terminator: closed but never expanded
                v[k] = v[k] && cur[k] == d && !(fl[k] & PL_FLAG_T);
                // queue ids are in [0, idCap) < 2^31 (the queue entry keeps 32 bits): 32-bit division, not the 64-bit sequence
                const uint32_t row_u = eid / (uint32_t)m.S;
                int col = (int)(eid - row_u * (uint32_t)m.S), row = (int)row_u;
                if (s.alias && v[k] && col == 0) {
                    if (w.aliasKey[row] & 1ull) { col = m.S; row -= 1; }   // owned by the last-column lattice cell
                }
                cells += (v[k] && nbr == 0) ? 1u : 0u;
                // the relaxation of pl_relax, up to its distance load
                col += dcn; row += drn;
                nd[k] = d + costn;
                v[k] = v[k] && !(col < s.colMin || col > s.colMax || row < s.rowMin || row > s.rowMax);
                int ox = col - 1;
                if (ox >= m.S) ox = m.S - 1;
                if (ox < 0) ox += m.nx;
                int oy = s.orow0 + (s.row0 - row);
                if (oy >= m.Sy) oy = m.Sy - 1;
                if (oy < 0) oy += m.ny;
                if (v[k]) v[k] = !((s.mt.bits[(size_t)ox * m.wpc + (oy >> 6)] >> (oy & 63)) & 1ull);
                nid[k] = (uint32_t)col + (uint32_t)row * (uint32_t)m.S;
                if (s.alias && v[k] && (col == 0 || col == m.S)) {
                    const int slot = col == 0 ? row : row + 1;
                    const unsigned long long key = ((unsigned long long)d << 44) | ((unsigned long long)eid << 8) |
                                                   ((unsigned long long)nbr << 1) | (col == 0 ? 0ull : 1ull);
                    atomicMin(&w.aliasKey[slot], key);
                }
            }
            // The pre-check load settles the relaxations that cannot lower a distance without an atomic (about half of a
            // ring's go backwards or sideways). It pays where sixteen searches share a CU and the atomics are what is scarce
            // (group forms: the sweep 15 % shorter); in plan_kernel a bucket is a chain of cold memory round trips -- ~3.3 k
            // cycles each on the bench workload -- and the pre-check is one of them: without it the sweep takes 0.85 x the
            // time (S::RELAX_PRECHECK; round-4 A/B in DESIGN.md section 9).
            if constexpr (PL_RELAX_PRECHECK && S::RELAX_PRECHECK) {
#pragma unroll
                for (int k = 0; k < PL_SWEEP_U; k++) cur[k] = v[k] ? w.dist[nid[k]] : 0u;
#pragma unroll
                for (int k = 0; k < PL_SWEEP_U; k++) v[k] = v[k] && !(cur[k] <= nd[k]);
            }
#pragma unroll
            for (int k = 0; k < PL_SWEEP_U; k++) if (v[k]) cur[k] = atomicMin(&w.dist[nid[k]], nd[k]);
#pragma unroll
            for (int k = 0; k < PL_SWEEP_U; k++) {
                if (v[k] && nd[k] < cur[k]) {
                    const int q2 = pl_bucket(nd[k]) & (PL_NQ - 1);
                    const uint32_t pos = atomicAdd(&s.qcount[q2], 1u) - s.qbase[q2];   // (qcount counts every push ever, qbase what has been consumed)
                    if (pos < PL_QCAP) w.queue[(size_t)q2 * PL_QCAP + pos] = ((unsigned long long)nd[k] << 32) | (unsigned long long)nid[k];
                    else s.qover = 1;
                }
            }
        }
        if (cells) atomicAdd((unsigned long long*)&s.h_cells, (unsigned long long)cells);
        if (q == 0) b0 += full; else if (q == 1) b1 += full; else if (q == 2) b2 += full; else b3 += full;
        if (Coop::tid() == 0) s.qbase[q] += full;         // (read by pl_relax when bucket E + 2 pushes into this queue again)
        E += 1;
        Coop::sync();
    }
    Coop::sync();
    if (Coop::tid() == 0) {
        s.E = E;
        if constexpr (PROFILE) s.phase[PH_SWEEP] += clock64() - t_sw;
        const uint32_t d = w.dist[id];
        s.hq_d = (d != PL_UNSEEN && pl_bucket(d) <= E) ? d : PL_UNSEEN;
        s.h_misses += 1;
        if (s.hq_d != PL_UNSEEN) {
            s.dF = d; s.idF = id; s.hasF = 1;
            w.flags[id] |= PL_FLAG_T;
        }
    }
    Coop::sync();
}

// Heuristic-sweep state for a new goal: clears the id-space arrays, builds the goal-anchored lattice
// description (compute_h.py:89-186 positions are xf +- k*dx by repeated addition) and performs the first
// update_openlist(initial_grid) (:207-210). Sets s.status = 6 when the lattice is not regular w.r.t. the
// id grid or the goal lies outside the map. All threads.
template <class Coop = CoopWG, class S>
AVP_D void pl_sweep_init(const DevMap& m, const PlanWs& w, S& s, const PlanDims& dims, double gx, double gy)
{
    const int tid = Coop::tid();
    for (int64_t i = tid; i < dims.idCap; i += Coop::N) { w.dist[i] = PL_UNSEEN; w.flags[i] = 0; }
    for (int64_t i = tid; i < dims.rowCap; i += Coop::N) w.aliasKey[i] = ~0ull;
    if (tid == 0) {
        s.E = 0; s.qover = 0; s.hasF = 0; s.dF = 0; s.idF = 0;
        for (int q = 0; q < PL_NQ; q++) { s.qcount[q] = 0; s.qbase[q] = 0; }
        s.h_cells = 0; s.h_misses = 0;
        const int col0 = (int)floor((gx - m.b0) / m.dx);
        const int row0 = (int)floor((m.b3 - gy) / m.dy);
        const int orow0 = (int)floor((gy - m.b2) / m.dy) - 1;
        int regular = 1;
        int colMin = col0, colMax = col0, rowMin = row0, rowMax = row0;
        // (a goal outside the map is refused BEFORE the walks below: from a goal far outside they take |g - b| / dx trips)
        if (!(gx >= m.b0 && gx <= m.b1 && gy >= m.b2 && gy <= m.b3)) regular = 0;
        else {
        double v = gx;
        for (int a = 1;; a++) { v = v + m.dx; if (!(v <= m.b1)) break; const int c = (int)floor((v - m.b0) / m.dx); if (c != col0 + a) regular = 0; colMax = col0 + a; }
        v = gx;
        for (int a = 1;; a++) { v = v - m.dx; if (!(v >= m.b0)) break; const int c = (int)floor((v - m.b0) / m.dx); if (c != col0 - a) regular = 0; colMin = col0 - a; }
        v = gy;
        for (int b = 1;; b++) {
            v = v + m.dy; if (!(v <= m.b3)) break;
            const int r = (int)floor((m.b3 - v) / m.dy), o = (int)floor((v - m.b2) / m.dy) - 1;
            if (r != row0 - b || o != orow0 + b) regular = 0;
            rowMin = row0 - b;
        }
        v = gy;
        for (int b = 1;; b++) {
            v = v - m.dy; if (!(v >= m.b2)) break;
            const int r = (int)floor((m.b3 - v) / m.dy), o = (int)floor((v - m.b2) / m.dy) - 1;
            if (r != row0 + b || o != orow0 - b) regular = 0;
            rowMax = row0 + b;
        }
        }
        if (colMin < 0 || colMax > m.S || rowMin < 0 || rowMax > m.Sy + 1) regular = 0;
        s.col0 = col0; s.row0 = row0; s.orow0 = orow0; s.colMin = colMin; s.colMax = colMax; s.rowMin = rowMin; s.rowMax = rowMax;
        s.alias = (colMax == m.S) ? 1 : 0;
        s.goal_id = (int64_t)col0 + (int64_t)row0 * m.S;
        s.regular = regular;
        if (!regular) s.status = 6;
    }
    Coop::sync();
    if (s.status == 0 && tid < 8) {
        const int dc[8] = { -1, 0, 1, -1, 1, -1, 0, 1 };
        const int dr[8] = { -1, -1, -1, 0, 0, 1, 1, 1 };
        const uint32_t cost[8] = { 14, 10, 14, 10, 10, 14, 10, 14 };
        pl_relax(m, w, s, s.col0 + dc[tid], s.row0 + dr[tid], cost[tid], 0, s.goal_id, tid);
    }
    Coop::sync();
}

// Test hook: the heuristic field alone. One workgroup runs the reference's query sequence
// (Dijkstra.compute_path calls, compute_h.py:198-214, preceded by the closed-list lookup of
// hybrid_a_star.py:272-280 unless force[i]) and dumps the distances, terminator flags and the closed frontier.
__global__ __launch_bounds__(PL_THREADS) void hfield_kernel(DevMap m, double gx, double gy, const double* __restrict__ queries,
                                                            const int32_t* __restrict__ force, int32_t nq, int32_t maxNodes,
                                                            char* __restrict__ workspace, int32_t* __restrict__ out_d,
                                                            int32_t* __restrict__ out_miss, uint32_t* __restrict__ out_dist,
                                                            uint8_t* __restrict__ out_flags, int64_t* __restrict__ out_info)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];
    PlShared& s = *reinterpret_cast<PlShared*>(pl_smem);
    const PlanDims dims = plan_dims(m.S, m.Sy, maxNodes);
    const PlanWs w = plan_carve(workspace, dims);
    if (threadIdx.x == 0) { s.status = 0; s.sched_cnt = -1; s.sched_n = 0; for (int k = 0; k < PH_COUNT; k++) s.phase[k] = 0; s.mt.X = m.X; s.mt.Y = m.Y; s.mt.bits = m.colBits; }
    __syncthreads();
    pl_sweep_init(m, w, s, dims, gx, gy);
    for (int i = 0; i < nq && s.status == 0; i++) {
        const int64_t id = avp_pos_to_index(m, queries[2 * i], queries[2 * i + 1]);
        uint32_t hd = PL_UNSEEN;
        const uint32_t d0 = pl_id_in_range(m, id) ? w.dist[id] : PL_UNSEEN;
        const bool hit = !force[i] && pl_hquery_hit(m, s, id, d0, hd);
        if (!hit) { pl_hquery_miss(m, w, s, id); hd = s.hq_d; }
        if (threadIdx.x == 0) { out_d[i] = hd == PL_UNSEEN ? -1 : (int32_t)hd; out_miss[i] = hit ? 0 : 1; }
        __syncthreads();
    }
    for (int64_t i = threadIdx.x; i < dims.idCap; i += PL_THREADS) { out_dist[i] = w.dist[i]; out_flags[i] = w.flags[i]; }
    if (threadIdx.x == 0) {
        out_info[0] = s.status; out_info[1] = s.hasF ? (int64_t)s.dF : -1; out_info[2] = s.idF; out_info[3] = s.goal_id;
        out_info[4] = s.E; out_info[5] = s.h_cells; out_info[6] = s.h_misses; out_info[7] = dims.idCap;
    }
}

// hybrid_a_star.py:243-259
AVP_D double pl_node_cost(const avp_params& p, int node_forward, double node_theta, double father_theta, int father_gear)
{
    double cost_gear = 0;
    if (node_forward != father_gear) cost_gear = p.cost_gear;
    const double cost_heading = fabs(node_theta - father_theta);
    const double cost = cost_gear + p.cost_heading * cost_heading;
    return p.cost_scale * cost;
}

// Evaluate RS words for queries [0, nq): query 0 = current node (the shot), 1.. = children.
// One thread per (query, word); then one thread per query folds the 46 results in source order.
// Lane schedule for the (word, query) items of one RS pass. A wave pays the full instruction stream of
// every word solver any of its lanes runs, so the items are grouped into chunks of <= 64 lanes of ONE solver
// and the chunks are spread over the waves (longest-processing-time first, measured solver costs), each
// chunk in its own round of its wave: sched[round][wave][lane].
static __device__ const int8_t PL_SCHED_WORDS[9][8] = { { 18, 19, 20, 21, -1, -1, -1, -1 }, { 22, 23, 24, 25, -1, -1, -1, -1 }, { 0, 1, -1, -1, -1, -1, -1, -1 },
                                                        { 6, 7, 8, 9, -1, -1, -1, -1 }, { 10, 11, 12, 13, 14, 15, 16, 17 }, { 26, 27, 28, 29, 34, 35, 36, 37 },
                                                        { 30, 31, 32, 33, 38, 39, 40, 41 }, { 2, 3, 4, 5, -1, -1, -1, -1 }, { 42, 43, 44, 45, -1, -1, -1, -1 } };
static __device__ const int32_t PL_SCHED_COST[9] = { 78, 67, 63, 54, 49, 45, 31, 30, 16 };
__device__ __noinline__ void pl_rs_build_schedule(PlShared& s, int nq)      // (a leaf call: runs once, must not be unrolled into the kernel body)
{
    // solvers by descending cost (x100 cycles per call of one wave, scripts/microbench/rs_words.hip on MI355X with round 4's
    // glibc-exact libm): LRLRn, LRLRp (tauOmega: 5 sin/cos + acos + atan2), SLS (two tan, two pow), LRL, LSR, LRSL, LRSR, LSL, LRSLR
    // (runs once per workgroup and per child count, on one thread; its work arrays live in LDS: no stack objects)
    const int nwave = PL_THREADS / 64;
#pragma nounroll
    for (int w = 0; w < nwave; w++) { s.sched_load[w] = 0; s.sched_rounds[w] = 0; }
#pragma nounroll
    for (int i = 0; i < PL_SCHED_ROUNDS * PL_THREADS; i++) s.sched[i] = 0xffff;
    int maxround = 0;
#pragma nounroll
    for (int sv = 0; sv < 9; sv++) {
        int nw = 0;
        while (nw < 8 && PL_SCHED_WORDS[sv][nw] >= 0) nw++;
        const int items = nw * nq;
        for (int base = 0; base < items; base += 64) {
            const int cnt = items - base < 64 ? items - base : 64;
            int best = 0;
            for (int w = 1; w < nwave; w++) if (s.sched_load[w] < s.sched_load[best]) best = w;
            const int r = s.sched_rounds[best]++;
            s.sched_load[best] += PL_SCHED_COST[sv];
            if (r >= PL_SCHED_ROUNDS) continue;           // cannot happen for nq <= PL_RSQ (static_assert below)
            if (r + 1 > maxround) maxround = r + 1;
            for (int k = 0; k < cnt; k++) {
                const int it = base + k;
                s.sched[r * PL_THREADS + best * 64 + k] = (uint16_t)((PL_SCHED_WORDS[sv][it / nq] << 4) | (it % nq));
            }
        }
    }
    s.sched_n = maxround * PL_THREADS;
    s.sched_cnt = nq;
}

template <typename PoseFn>
AVP_D void pl_rs_words(PlShared& s, const avp_params& p, int nq, PoseFn pose, bool frames_ready)
{
    // start-frame normalisation once per query (already done by the caller when frames_ready)
    if (!frames_ready) {
        if ((int)threadIdx.x < nq) {
            double x, y, th;
            pose((int)threadIdx.x, x, y, th);
            s.frame[threadIdx.x] = rs_frame(x, y, th, s.goal[0], s.goal[1], s.goal[2], p.maxc);
        }
        if (threadIdx.x == PL_THREADS - 1 && s.sched_cnt != nq) pl_rs_build_schedule(s, nq);
        __syncthreads();
    }
    for (int t = threadIdx.x; t < s.sched_n; t += PL_THREADS) {
        const uint16_t it = s.sched[t];
        if (it == 0xffff) continue;
        const int wd = it >> 4, q = it & 15;
        const int slot = q * 46 + wd;
        const RsWordOut o = rs_word_fn(wd, s.frame[q]);      // (a called leaf: the nine solvers exist once, outside the pop loop's register budget)
        s.w_ok[slot] = o.ok ? 1 : 0;
        s.w_acc[slot] = 0;
        s.w_l[slot][0] = o.l0; s.w_l[slot][1] = o.l1; s.w_l[slot][2] = o.l2; s.w_l[slot][3] = o.l3; s.w_l[slot][4] = o.l4;
    }
    if ((int)threadIdx.x < nq) s.w_err[threadIdx.x] = 0;
    // (the caller closes the pass with a workgroup barrier)
}

// set_path (rs_curve.py:137-156) for all queries at once: one thread per (query, type group); a
// candidate is only ever compared with kept candidates of the same type sequence, so the groups are
// independent and the within-group order is the source order.
// One (query, type group) of set_path (rs_curve.py:137-156): a candidate is only ever compared with kept
// candidates of the same type sequence, so the groups are independent and the within-group order is the source
// order. w_acc / w_err were cleared by pl_rs_words.
AVP_D void pl_rs_accept_group(PlShared& s, const avp_params& p, int q, int g)
{
    // The unused tail of a word's 5 lengths is 0.0 (rs_word), so the sums below always run over all 5 entries:
    // same values (x + 0.0), but the LDS loads no longer depend on the per-word segment count and issue together.
    // Fully unrolled over the group's (<= 4) words: kept[][] stays in registers (no stack object).
    unsigned accmask = 0;                     // accepted words of this group so far (bit j = RS_GROUPS[g][j])
    double kept[4][5];                        // lengths of the group's words seen so far
    bool live = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int wd = RS_GROUPS[g][j];
        if (wd < 0) live = false;
        const int slot = q * 46 + (wd < 0 ? 0 : wd);
        const bool ok = live && s.w_ok[slot];
        double l[5];
#pragma unroll
        for (int i = 0; i < 5; i++) { l[i] = s.w_l[slot][i]; kept[j][i] = l[i]; }
        bool dup = false;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            if (e >= j) continue;
            double sum = 0;
#pragma unroll
            for (int i = 0; i < 5; i++) sum = sum + (kept[e][i] - l[i]);
            if ((accmask & (1u << e)) && !dup && sum <= 0.01) dup = true;
        }
        double L = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) L = L + fabs(l[i]);
        if (!ok || dup || L >= 1000.0) continue;
        if (!(L >= 0.01)) { s.w_err[q] = 1; continue; }
        accmask |= 1u << j;
        s.w_acc[slot] = 1; s.w_Lm[slot] = L / p.maxc;
    }
}

// arg-min over the accepted words of query q with "<=" (the last of equal minima wins, rs_curve.py:103-108):
// one wave per query, lane = word, butterfly reduction on (length, word index). All lanes of the wave call it;
// the result is valid on every lane.
AVP_D int pl_rs_fold_wave(PlShared& s, int q, RsPath& out)
{
    // min over the accepted words of (length, then the LATER word on a tie: "<=" in calc_optimal_path keeps the last).
    // Lengths are positive doubles, so their bit patterns order like the values: one LDS atomic min on the bits, one
    // atomic max on the word index among the lanes that hold the minimum -- two round trips instead of a 6-step
    // butterfly over three values.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int acc = 0;
    unsigned long long lb = ~0ull;
    if (lane < 46) { acc = s.w_acc[q * 46 + lane]; lb = (unsigned long long)__double_as_longlong(s.w_Lm[q * 46 + lane]); }
    if (lane == 0) { s.fold_key[wv] = ~0ull; s.fold_idx[wv] = -1; }
    wave_sync();
    if (acc) atomicMin(&s.fold_key[wv], lb);
    wave_sync();
    if (acc && lb == s.fold_key[wv]) atomicMax(&s.fold_idx[wv], lane);
    wave_sync();
    const int wd = s.fold_idx[wv];
    acc = wd >= 0;
    out.n = 0; out.L = 0;
    if (s.w_err[q]) return 2;
    if (!acc) return 1;
    const RsWord W = RS_WORDS[wd];
    out.n = W.n;
    out.t[0] = 0 < W.n ? W.a : (int8_t)-1; out.t[1] = 1 < W.n ? W.b : (int8_t)-1; out.t[2] = 2 < W.n ? W.c : (int8_t)-1;
    out.t[3] = 3 < W.n ? W.d : (int8_t)-1; out.t[4] = 4 < W.n ? W.e : (int8_t)-1;
#pragma unroll
    for (int i = 0; i < AVP_RS_MAXSEG; i++) out.l[i] = s.w_l[q * 46 + wd][i];
    double Ln = 0;                                   // the winner's normalised length: the same sum set_path formed (rs_curve.py:145)
#pragma unroll
    for (int i = 0; i < AVP_RS_MAXSEG; i++) Ln = Ln + fabs(out.l[i]);
    out.L = Ln;
    return 0;
}

// ---- parallel Reeds-Shepp sampling (rs_curve.py:537-594 + :125-131), in stages -----------------------
// A (two lanes of two waves): the index bookkeeping of generate_local_course (which output index each
//   interpolate() call writes; later writes overwrite earlier ones) and the chain of segment origins.
// B (the checking waves): each lane evaluates the interpolation of one sample, transforms it to the world frame
//   and hands it to the wave's collision pass (pl_rs_sample_world); the trailing px == 0.0 entries the reference
//   pops are tracked through an atomic max of the last non-zero index.
// generate_local_course (rs_curve.py:537-592) split in two independent serial jobs:
//   pl_rs_sample_book    -- the index bookkeeping: which sample lies at which arc length of which segment;
//   pl_rs_sample_origins -- the chain of segment origins (end pose of the previous segment).
// They touch disjoint state, so two waves run them side by side.
// Returns 0, or 5 when the shot has more samples than the form's buffers hold (the caller stores it: in the pair form
// another wave may be reading s.rs_status meanwhile).
template <class S>
AVP_D int pl_rs_sample_book(S& s, const avp_params& p)
{
    const double maxc = p.maxc;
    const RsPath& rp = s.rs;
    const double step = 0.5 * maxc;
    const int point_num = (int)(rp.L / step) + rp.n + 3;
    s.smp_point_num = point_num;
    s.smp_hi = 0;
    if (point_num > S::RS_CAP || point_num > PL_CHK_MAX - 256) return 5;
    int ind = 1, hi = 0;
    double d = rp.l[0] > 0.0 ? step : -step;
    double pd = d, ll = 0.0;
    for (int i = 0; i < rp.n; i++) {
        const double l = rp.l[i];
        d = l > 0.0 ? step : -step;
        ind -= 1;
        if (i >= 1 && (rp.l[i - 1] * rp.l[i]) > 0) pd = -d - ll; else pd = d - ll;
        while (fabs(pd) <= fabs(l)) {
            ind += 1;
            s.smp_l[ind] = pd; s.smp_seg[ind] = (int8_t)i;
            pd += d;
        }
        ll = l - pd - d;
        ind += 1;
        s.smp_l[ind] = l; s.smp_seg[ind] = (int8_t)i;
        if (ind > hi) hi = ind;
    }
    s.smp_hi = hi;
    return 0;
}
template <class S>
AVP_D void pl_rs_sample_origins(S& s, const avp_params& p)
{
    // Whole wave. The chain "origin of segment i+1 = end pose of segment i" (rs_interpolate at the full segment length)
    // is serial only in its additions: the yaw of every origin is a running sum of +-lengths, and with it every
    // segment's displacement (its two sincos evaluations) is independent of the others. Lane i evaluates segment i
    // with exactly rs_interpolate's expressions; the positions are then accumulated in segment order.
    const int lane = threadIdx.x & 63;
    const int n = s.rs.n;
    double lr[AVP_RS_MAXSEG];
    int tr[AVP_RS_MAXSEG];
#pragma unroll
    for (int i = 0; i < AVP_RS_MAXSEG; i++) { lr[i] = s.rs.l[i]; tr[i] = s.rs.t[i]; }      // one LDS round trip
    double oyaw = 0.0, lmine = 0.0;                      // origin yaw of segment `lane` (px[1] before any write is 0)
    int tmine = RS_S;
#pragma unroll
    for (int i = 0; i < AVP_RS_MAXSEG; i++) {
        if (i < n && i < lane) oyaw = tr[i] == RS_S ? oyaw : (tr[i] == RS_L ? oyaw + lr[i] : oyaw - lr[i]);
        if (i == lane) { lmine = lr[i]; tmine = tr[i]; }
    }
    double gdx = 0.0, gdy = 0.0;
    if (lane < n - 1) {
        double ex, ey, eyaw;
        rs_interpolate(lmine, tmine, p.maxc, 0.0, 0.0, oyaw, ex, ey, eyaw);   // 0 + displacement = the displacement
        gdx = ex; gdy = ey;
    }
    double ox = 0.0, oy = 0.0;
#pragma unroll
    for (int i = 0; i < AVP_RS_MAXSEG; i++) {
        if (i < n) {
            if (lane == i) { s.seg_o[i][0] = ox; s.seg_o[i][1] = oy; s.seg_o[i][2] = oyaw; }
            ox = ox + __shfl(gdx, i, 64);                    // px = ox + gdx (:612 / :601), in segment order
            oy = oy + __shfl(gdy, i, 64);
        }
    }
}
// Sample i of the shot, straight to the world frame (generate_local_course's interpolation :537-624 followed by
// calc_all_paths' rotation :125-131), per sample, so
// that a wave can produce exactly the samples it is about to check. Also maintains the trim bound s.rs_npts.
template <class S>
AVP_D void pl_rs_sample_world(const PlanWs& w, S& s, const avp_params& p, const PlNode& cn, double cm, double sm,
                              int i, double& tx, double& ty, double& tth)
{
    double px = 0.0, py = 0.0, pyaw = 0.0;
    int8_t dr = 0;
    if (i == 0) dr = s.rs.l[0] > 0.0 ? 1 : -1;
    else if (i <= s.smp_hi) {
        const int sg = s.smp_seg[i];
        const double l = s.smp_l[i];
        rs_interpolate(l, s.rs.t[sg], p.maxc, s.seg_o[sg][0], s.seg_o[sg][1], s.seg_o[sg][2], px, py, pyaw);
        dr = l > 0.0 ? 1 : -1;
    }
    tx = cm * px + sm * py + cn.x;
    ty = -sm * px + cm * py + cn.y;
    tth = avp_pi_2_pi(pyaw + cn.th);
    w.rsbuf[3 * i] = tx; w.rsbuf[3 * i + 1] = ty; w.rsbuf[3 * i + 2] = tth; w.rsdir[i] = dr;
    if (px != 0.0) atomicMax(&s.rs_npts, i + 1);
}

// ---- wave-local cooperative collision pass (distance_checker semantics, collision_check.py:144-240) ----
// One wave checks up to PL_WPOSE poses without any workgroup barrier: lanes set up the footprints, one
// lane per (pose, map column under the AABB) gathers the near obstacle points from the column bitmaps into
// the wave's LDS queue, one lane per (pose, point) runs the exact test.
// pl_check_pass is a CALLED function: the poses (x, y, theta, cos, sin) are staged in wc.pose[] by the caller
// (pl_check_wave below), the map / vehicle constants come from the workgroup's PlChkEnv, the hit flags go to
// out_hit[k] (LDS). Its registers are its own -- inlined into the pop loops (twice per kernel) it set their
// register demand.
// the narrow phase of a pass, a called function of its own (round 4): one lane per (pose, point) candidate of the wave's
// queue. Inside pl_check_pass its 22 record values per candidate competed with the pass's set-up and gather state for the
// 128 registers of the group forms (7 - 9 spilled VGPRs, +150 B of scratch per lane once the point test had its
// division-free first look).
// FAST: the point test's division-free first look (avp_footprint_point_hit) or the exact form with its eight divisions.
// Same booleans; which is quicker depends on what bounds the pass (measured in one run, round 4): four waves per problem
// (latency bound: a pass is one dependent chain) 87.5 ms with the exact form against 91.3 ms on the 4 096 batch -- the
// eight divisions overlap, the first look's comparisons and its fall-back code do not pay for themselves --, one wave per
// problem on the saturating batch (issue bound) 194.5 ms with the first look against 197.4 ms. So: the first look where
// sixteen searches share a CU (S::POINT_FAST), the exact form elsewhere.
template <bool STAGE, int QCAP, bool FAST>
__device__ __noinline__ void pl_check_narrow(AVP_LDS const PlChkEnv* envp, AVP_LDS PlWaveChkT<QCAP>* wcp)
{
    const PlTabs<STAGE> mt(*(const PlChkEnv*)envp);
    const int lane = threadIdx.x & 63;
    const int qn = wcp->qn;
    for (int e = lane; e < qn; e += 64) {
        const uint32_t ent = wcp->q[e];
        const int i = ent >> 26, ix = (ent >> 13) & 0x1fff, iy = ent & 0x1fff;
        if (wcp->hit[i]) continue;
        const bool h = FAST ? avp_footprint_point_hit(wcp->fp[i], mt.X[ix], mt.Y[iy]) : avp_footprint_point_hit_exact(wcp->fp[i], mt.X[ix], mt.Y[iy]);
        if (h) wcp->hit[i] = 1;
    }
}

// The gather of a pass whose footprints span more than 64 map columns or more than two bitmap words of rows (fine maps:
// map_discrete_size 0.05 -> a diagonal of 107 cells; the default maps never come here). A called function of its own: its
// loops' registers stay out of pl_check_pass. Lane c owns columns ixlo + c, ixlo + c + 64, ... of every pose of the range
// [lo, hi), all words of the row range; the same two walks (count, reserve queue room with one atomic, write), the same
// queue order within a (pose, column).
template <bool STAGE, int QCAP>
__device__ __noinline__ void pl_check_gather_wide(AVP_LDS const PlChkEnv* envp, AVP_LDS PlWaveChkT<QCAP>* wcp, int lo, int hi)
{
    const PlChkEnv& env = *(const PlChkEnv*)envp;
    PlWaveChkT<QCAP>& wc = *(PlWaveChkT<QCAP>*)wcp;
    const PlTabs<STAGE> mt(env);
    const int lane = threadIdx.x & 63;
    const int wpc = env.wpc;
    int tot = 0;
#pragma nounroll
    for (int i = lo; i < hi; i++) {
        const int ixlo = wc.rng[i][0], ixhi = wc.rng[i][1], iylo = wc.rng[i][2], iyhi = wc.rng[i][3];
        if (iylo > iyhi) continue;
        const int w0 = iylo >> 6, w1 = iyhi >> 6;
        for (int ix = ixlo + lane; ix <= ixhi; ix += 64)
            for (int w = w0; w <= w1; w++) {
                uint64_t b = mt.bits[(size_t)ix * wpc + w];
                if (w == w0) b &= ~0ull << (iylo & 63);
                if (w == w1) b &= ~0ull >> (63 - (iyhi & 63));
                tot += __popcll(b);
            }
    }
    int pos = tot ? atomicAdd(&wc.qn, tot) : 0;
    if (pos + tot > QCAP) { if (tot) wc.over = 1; return; }
    if (!tot) return;
#pragma nounroll
    for (int i = lo; i < hi; i++) {
        const int ixlo = wc.rng[i][0], ixhi = wc.rng[i][1], iylo = wc.rng[i][2], iyhi = wc.rng[i][3];
        if (iylo > iyhi) continue;
        const int w0 = iylo >> 6, w1 = iyhi >> 6;
        for (int ix = ixlo + lane; ix <= ixhi; ix += 64) {
            const uint32_t tag = ((uint32_t)i << 26) | ((uint32_t)ix << 13);
            for (int w = w0; w <= w1; w++) {
                uint64_t b = mt.bits[(size_t)ix * wpc + w];
                if (w == w0) b &= ~0ull << (iylo & 63);
                if (w == w1) b &= ~0ull >> (63 - (iyhi & 63));
                while (b) { const int bpos = __ffsll((unsigned long long)b) - 1; b &= b - 1; wc.q[pos++] = tag | (uint32_t)((w << 6) + bpos); }
            }
        }
    }
}

// The lane-per-pose form of a pass, a called function of its own (round 6: two call sites -- inlined twice it cost the group forms a dozen
// spill slots): lanes lo .. hi-1 each walk the map columns under their pose's footprint serially (pl_check_pose); wc.hit[lane] = the result.
// Used for the two-circle model, for maps beyond the queue's 13-bit cell indices, and for a single pose whose candidates overflow the queue.
template <bool STAGE, int QCAP, int KIND>
__device__ __noinline__ void pl_check_lanes_k(AVP_LDS const PlChkEnv* envp, AVP_LDS PlWaveChkT<QCAP>* wcp, int lo, int hi)
{
    const PlChkEnv& env = *(const PlChkEnv*)envp;
    PlWaveChkT<QCAP>& wc = *(PlWaveChkT<QCAP>*)wcp;
    const PlTabs<STAGE> mt(env);
    const int lane = threadIdx.x & 63;
    if (lane >= lo && lane < hi)
        wc.hit[lane] = pl_check_pose<KIND>(env, mt.X, mt.Y, mt.bits, wc.pose[lane][0], wc.pose[lane][1], wc.pose[lane][2], wc.pose[lane][3], wc.pose[lane][4]) ? 1u : 0u;
}
template <bool STAGE, int QCAP>
__device__ __forceinline__ void pl_check_lanes(AVP_LDS const PlChkEnv* envp, AVP_LDS PlWaveChkT<QCAP>* wcp, int lo, int hi)
{
    if (((const PlChkEnv*)envp)->kind == 1) pl_check_lanes_k<STAGE, QCAP, 1>(envp, wcp, lo, hi);      // (uniform: a per-map constant)
    else pl_check_lanes_k<STAGE, QCAP, 0>(envp, wcp, lo, hi);
}

template <bool STAGE, int QCAP, bool FAST>
__device__ __noinline__ void pl_check_pass(AVP_LDS const PlChkEnv* envp, AVP_LDS PlWaveChkT<QCAP>* wcp, int count, AVP_LDS uint32_t* out_hit_p)
{
    const PlChkEnv& env = *(const PlChkEnv*)envp;
    PlWaveChkT<QCAP>& wc = *(PlWaveChkT<QCAP>*)wcp;
    uint32_t* out_hit = (uint32_t*)out_hit_p;
    const PlTabs<STAGE> mt(env);
    const int lane = threadIdx.x & 63;
    if (env.kind == 1 || env.big) {          // the two-circle model / a map beyond the queue's 13-bit cell indices: a lane per pose (same booleans)
        pl_check_lanes<STAGE, QCAP>(envp, wcp, 0, count);
        wave_sync();
        if (lane < count) out_hit[lane] = wc.hit[lane];
        wave_sync();
        return;
    }
    // Footprint set-up, 8 lanes per pose: every lane of a group evaluates the corners, then the group's lanes split
    // what is data parallel: 4 edges (slope / intercept / norm: the divisions and square roots), 2 side lengths, and
    // the 4 index searches (2 code paths x 2 axes). Same arithmetic per value as avp_footprint_setup /
    // avp_footprint_aabb -- only who computes it changes.
    {
        const int grp = lane >> 3, sub = lane & 7;
        if (grp < count) {
            const double x = wc.pose[grp][0], y = wc.pose[grp][1], cs = wc.pose[grp][3], sn = wc.pose[grp][4];
            const double lx[4] = { env.fp_xr, env.fp_xf, env.fp_xf, env.fp_xr };
            const double ly[4] = { env.fp_yr, env.fp_yr, env.fp_yl, env.fp_yl };
            double cx[4], cy[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                cx[i] = AVP_FMA(-sn, ly[i], cs * lx[i]) + x;
                cy[i] = AVP_FMA(cs, ly[i], sn * lx[i]) + y;
            }
            Footprint& f = wc.fp[grp];
            if (sub < 6) {
                // sub 0..3: edge sub (corner sub -> corner (sub + 1) & 3): slope, intercept, norm sqrt(1 + k*k);
                // sub 4: |rr - lr| (width), sub 5: |lr - lf| (length). One sqrt(u*u + v*v) serves all six lanes
                // (1*1 is exact, so sqrt(1 + k*k) is the same expression).
                const double x1 = sub == 0 ? cx[0] : sub == 1 ? cx[1] : sub == 2 ? cx[2] : cx[3];     // sub 4, 5: corner lr
                const double y1 = sub == 0 ? cy[0] : sub == 1 ? cy[1] : sub == 2 ? cy[2] : cy[3];
                const double x2 = sub == 0 ? cx[1] : sub == 1 ? cx[2] : sub == 2 ? cx[3] : cx[0];
                const double y2 = sub == 0 ? cy[1] : sub == 1 ? cy[2] : sub == 2 ? cy[3] : cy[0];
                double u = 1.0, v = 0.0, k = 0.0;
                if (sub < 4) { k = (y2 - y1) / (x2 - x1); v = k; }            // +-inf / NaN when axis aligned, as numpy
                else if (sub == 4) { u = cx[0] - cx[3]; v = cy[0] - cy[3]; }
                else { u = cx[3] - cx[2]; v = cy[3] - cy[2]; }
                const double r = sqrt(u * u + v * v);
                if (sub < 4) { f.cx[sub] = x1; f.cy[sub] = y1; f.k[sub] = k; f.b[sub] = y1 - k * x1; f.den[sub] = r; if (FAST) f.rden[sub] = avp_footprint_rden(k, r); }   // (1 / den: only the first look reads it)
                else if (sub == 4) f.wthr = r - 0.01;
                else f.lthr = r - 0.01;
            }
            double xmin = cx[0], xmax = cx[0], ymin = cy[0], ymax = cy[0];
#pragma unroll
            for (int i = 1; i < 4; i++) {
                if (cx[i] > xmax) xmax = cx[i];
                if (cx[i] < xmin) xmin = cx[i];
                if (cy[i] > ymax) ymax = cy[i];
                if (cy[i] < ymin) ymin = cy[i];
            }
            if (sub < 4) {
                // ixlo = first node >= xmin, ixhi = last node <= xmax, iylo, iyhi likewise: one code path for the four
                const bool ax = sub < 2, upper = sub & 1;
                wc.rng[grp][sub] = (int16_t)avp_node_search(ax ? mt.X : mt.Y, ax ? env.nx : env.ny, ax ? env.b0 : env.b2, ax ? env.dx : env.dy,
                                                            ax ? (upper ? xmax : xmin) : (upper ? ymax : ymin), upper);
            }
            if (sub == 7) wc.hit[grp] = 0;
        }
    }
    wave_sync();
    // Gather: lane c owns map column ixlo + c of EVERY pose of the range. The lane counts the candidates of all its
    // (pose, column) pairs first (the bitmap words are fetched together: their LDS latencies overlap), reserves queue
    // room for them with ONE atomic, then walks the words again and writes them -- one atomic per range instead of
    // one per pose, and no per-pose words kept in registers between the two walks. The range is the whole pass; when
    // its candidates do not fit the queue it is halved (dense clutter), down to single poses.
    // (uniform: the ranges come from LDS) does a pose of the pass span more than 64 map columns or more than two bitmap
    // words of rows? Then the pass takes the general walk below: any number of column chunks and words per pose -- fine maps
    // (map_discrete_size 0.05: the footprint's diagonal spans 107 cells) and tall footprints. The default maps never do.
    bool wide = false;
    if (env.maybe_wide) {                                 // (a per-map constant: the default maps never look)
        int wide_v = 0;
#pragma nounroll
        for (int i = 0; i < count; i++)
            if (wc.rng[i][2] <= wc.rng[i][3] && (wc.rng[i][1] - wc.rng[i][0] >= 64 || (wc.rng[i][3] >> 6) - (wc.rng[i][2] >> 6) > 1)) wide_v = 1;
        wide = __builtin_amdgcn_readfirstlane(wide_v) != 0;
    }
    const int wpc = env.wpc;
    int lo = 0, span = count;
    while (lo < count) {
        const int hi = min(lo + span, count);
        if (lane == 0) { wc.qn = 0; wc.over = 0; }
        wave_sync();
        if (!wide) {
            int tot = 0;
            bool odd = false;                           // a range this walk does not cover (only a non-finite pose can have one on a map whose footprints fit): the pose goes to the lane-per-pose walk
#pragma unroll
            for (int i = 0; i < PL_WPOSE; i++) {
                if (i >= lo && i < hi) {
                    const int ixlo = wc.rng[i][0], ixhi = wc.rng[i][1], iylo = wc.rng[i][2], iyhi = wc.rng[i][3];
                    if (iylo <= iyhi && lane <= ixhi - ixlo) {
                        const int ix = ixlo + lane, w0 = iylo >> 6, w1 = iyhi >> 6;
                        if (w1 - w0 > 1 || ixhi - ixlo >= 64) odd = true;
                        uint64_t x0 = mt.bits[(size_t)ix * wpc + w0] & (~0ull << (iylo & 63));
                        if (w1 == w0) x0 &= ~0ull >> (63 - (iyhi & 63));
                        const uint64_t x1 = w1 > w0 ? (mt.bits[(size_t)ix * wpc + w1] & (~0ull >> (63 - (iyhi & 63)))) : 0ull;
                        tot += __popcll(x0) + __popcll(x1);
                    }
                }
            }
            int pos = tot ? atomicAdd(&wc.qn, tot) : 0;
            if (odd || pos + tot > QCAP) { if (tot || odd) wc.over = 1; }
            else if (tot) {
#pragma nounroll
                for (int i = lo; i < hi; i++) {
                    const int ixlo = wc.rng[i][0], ixhi = wc.rng[i][1], iylo = wc.rng[i][2], iyhi = wc.rng[i][3];
                    if (iylo <= iyhi && lane <= ixhi - ixlo) {
                        const int ix = ixlo + lane, w0 = iylo >> 6, w1 = iyhi >> 6;
                        uint64_t bits = mt.bits[(size_t)ix * wpc + w0] & (~0ull << (iylo & 63));
                        if (w1 == w0) bits &= ~0ull >> (63 - (iyhi & 63));
                        const uint32_t tag = ((uint32_t)i << 26) | ((uint32_t)ix << 13);
                        while (bits) { const int bpos = __ffsll((unsigned long long)bits) - 1; bits &= bits - 1; wc.q[pos++] = tag | (uint32_t)((w0 << 6) + bpos); }
                        bits = w1 > w0 ? (mt.bits[(size_t)ix * wpc + w1] & (~0ull >> (63 - (iyhi & 63)))) : 0ull;
                        while (bits) { const int bpos = __ffsll((unsigned long long)bits) - 1; bits &= bits - 1; wc.q[pos++] = tag | (uint32_t)(((w0 + 1) << 6) + bpos); }
                    }
                }
            }
        } else pl_check_gather_wide<STAGE, QCAP>(envp, wcp, lo, hi);
        wave_sync();
        if (wc.over) {
            if (hi - lo > 1) { span = (hi - lo + 1) >> 1; continue; }
            // one pose with more candidates than the queue holds: a lane walks its columns serially
            pl_check_lanes<STAGE, QCAP>(envp, wcp, lo, lo + 1);
        } else {
            pl_check_narrow<STAGE, QCAP, FAST>(envp, wcp);
        }
        wave_sync();
        lo = hi;
    }
    if (lane < count) out_hit[lane] = wc.hit[lane];
    wave_sync();
}

// The caller's side of a pass: pose(k, x, y, th, cs, sn) supplies pose k of this wave's chunk (cs, sn = cos / sin of
// th: the pose's own); lane k evaluates it and stages it for the pass. out_hit: LDS.
template <bool STAGE, class S, typename PoseFn>
AVP_D void pl_check_wave(const PlChkEnv& env, S& s, int count, PoseFn pose, uint32_t* out_hit)
{
    if (count <= 0) return;
    const int lane = threadIdx.x & 63;
    auto& wc = s.wave_chk();
    constexpr int QCAP = std::remove_reference<decltype(wc)>::type::WQCAP;
    if (lane < count) {
        double x, y, th, cs, sn;
        pose(lane, x, y, th, cs, sn);
        wc.pose[lane][0] = x; wc.pose[lane][1] = y; wc.pose[lane][2] = th; wc.pose[lane][3] = cs; wc.pose[lane][4] = sn;
    }
    wave_sync();
    pl_check_pass<STAGE, QCAP, S::POINT_FAST>((AVP_LDS const PlChkEnv*)&env, (AVP_LDS PlWaveChkT<QCAP>*)&wc, count, (AVP_LDS uint32_t*)out_hit);
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits until every global store of the wave
// has been acknowledged (a few thousand cycles after the node / heap / hash writes of a resolution). Use where the
// data handed over lives in LDS and the global writes are not read before a later full barrier.
__device__ __forceinline__ void pl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- fast path of the child resolution (expand_node :153-232), run by WAVE 0 alone (wave-level syncs only) ----
// Preconditions (caller): closed list non-empty and the arena has room for every child. Lanes classify and cost
// their child in parallel; if every heuristic query hits the closed frontier, lane 0 then applies, in child
// order, only what is order dependent: arena slots, in-place improvements of open nodes and heap pushes.
// s.fast (preset to 1) reports whether the pop was resolved here; when it is 0 nothing has been modified.
template <bool PROFILE, class S>
AVP_D void pl_resolve_fast_wave(const DevMap& m, const avp_params& p, const PlanWs& w, S& s, const PlanDims& dims,
                                const PlNode& cn, int nchild, bool pop_ahead, bool split = false)
{
    // split (plan_kernel's record pops, where the other waves are idle): the heuristic distances were read ahead by
    // another wave (c.pre_d), and the node / hash writes are handed to a writer wave (pl_resolve_writer_wave) that runs
    // beside the heap pushes of this one.
    const long long t_r0 = PH_NOW();
    // Per-child state stays in the registers of the child's lane; the order dependent parts read it with ballots
    // and shuffles -- the serial sections below would otherwise spend most of their time on LDS round trips.
    const int lane = threadIdx.x & 63;
    int cls = CL_SKIP, found = -1, first_coll = 0x7fffffff;
    double cg = 0.0, ch_ = 0.0, cf = 0.0, cx = 0.0, cy = 0.0, cth = 0.0;
    if (lane < nchild) {
        PlChild& c = s.child[lane];
        if (!split) c.pre_d = pl_id_in_range(m, c.id) ? w.dist[c.id] : PL_UNSEEN;
        found = c.found; first_coll = c.first_coll; cx = c.x; cy = c.y; cth = c.th;
        const int is_forward = lane < p.n_steer ? 1 : 0;
        const bool found_closed = found >= 0 && c.found_state == 2;
        const bool found_open = found >= 0 && c.found_state == 1;
        if (found_closed || c.oob) cls = CL_SKIP;
        else if (!found_open && first_coll != 0x7fffffff) cls = CL_NEW_CLOSED;
        else {
            uint32_t hd = PL_UNSEEN;
            const bool hit = pl_hquery_hit(m, s, c.id, c.pre_d, hd);
            if (!hit || hd == PL_UNSEEN || c.rs_err) { s.fast = 0; cls = CL_SKIP; }
            else {
                const double hv1 = (double)hd / 100, hv2 = c.L;
                const double hval = hv2 > hv1 ? hv2 : hv1;
                if (!found_open) {
                    cg = pl_node_cost(p, is_forward, cth, cn.th, cn.forward);
                    ch_ = hval; cf = cg + hval;
                    cls = CL_NEW_OPEN;
                } else {
                    const PlNode& ch = w.nodes[found];
                    cg = pl_node_cost(p, ch.forward, ch.th, cn.th, cn.forward);
                    ch_ = hval; cf = hval + cg;
                    cls = cf < ch.f ? CL_IMPROVE : CL_KEEP;
                }
            }
        }
    }
    wave_sync();
    if (!s.fast) return;
    const long long t_r1 = PH_NOW();
    // arena slots in child order = prefix count of the children that create a node; counters by ballot / reduction
    const unsigned long long m_closed = __ballot(cls == CL_NEW_CLOSED), m_open = __ballot(cls == CL_NEW_OPEN);
    const unsigned long long m_rs = m_open | __ballot(cls == CL_IMPROVE || cls == CL_KEEP);
    const unsigned long long m_new = m_closed | m_open;
    const int32_t nnodes0 = s.nnodes;
    const int32_t pos = nnodes0 + __popcll(m_new & ((1ull << lane) - 1ull));
    int chk = cls == CL_NEW_CLOSED ? first_coll + 1 : 0;          // checks spent on the children that collided
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) chk += __shfl_xor(chk, d, 64);
    if (lane == 0) {
        s.nnodes = nnodes0 + __popcll(m_new);
        s.nclosed += __popcll(m_closed);
        s.n_checks += chk + (long long)__popcll(m_open) * p.n_sub;
        s.n_rs += __popcll(m_rs);
    }
    const int32_t gidx = (int32_t)s.global_index, cur = s.cur;
    if (split) {
        if (lane < nchild) { PlChild& c = s.child[lane]; c.cls = cls; c.pos = pos; c.g = cg; c.h = ch_; c.f = cf; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        wave_sync();
        if (lane == 0) *(volatile int32_t*)&s.wr_go = 1;
    } else if (cls == CL_NEW_CLOSED || cls == CL_NEW_OPEN) {
        const bool open = cls == CL_NEW_OPEN;
        PlNode& nd = w.nodes[pos];
        nd.x = cx; nd.y = cy; nd.th = cth;
        nd.g = open ? cg : 0.0; nd.h = open ? ch_ : 0.0; nd.f = open ? cf : 0.0;
        nd.index = gidx + lane + 1; nd.parent_index = cn.index; nd.parent_pos = cur;
        nd.forward = (int8_t)(lane < p.n_steer ? 1 : 0); nd.steer_i = (int8_t)(lane % p.n_steer);
        nd.state = open ? 1 : 2; nd.heap_pos = -1;
        pl_hash_put_atomic(w, dims.hashCap, pos, cx, cy, cth);
    }
    if (!split) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    wave_sync();
    const long long t_r2 = PH_NOW();
    // heap pushes / in-place improvements in child order; the operands come from the children's lanes, every push is
    // done by the whole wave (pl_heap_push_wave), an improvement by lane 0
    unsigned long long todo = m_open | __ballot(cls == CL_IMPROVE);
    int32_t nheap = s.nheap;
    while (todo) {
        const int i = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const int icls = __shfl(cls, i, 64), ipos = __shfl(pos, i, 64), ifound = __shfl(found, i, 64);
        const double if_ = __shfl(cf, i, 64), ig = __shfl(cg, i, 64), ih = __shfl(ch_, i, 64);
        if (icls == CL_NEW_OPEN) {
            pl_heap_push_wave(w, s, nheap, (uint32_t)ipos, if_);
            nheap++;
            // The next push reads entries this one has written (other lanes' stores). While the whole root path is in
            // the LDS part of the heap the wave's LDS operations are ordered as issued; entries in the workspace need
            // the stores drained (a global round trip).
            if (nheap > S::HEAP_LDS) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); wave_sync(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
            else wave_sync();
        } else {
            // an improvement reads the heap_pos field earlier pushes of this pop may have written (other lanes' global stores)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            wave_sync();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            int32_t hp = 0;
            if constexpr (!S::HEAP_POS) hp = pl_heap_find_wave(w, s, nheap, (uint32_t)ifound);
            if (lane == 0) {
                PlNode& ch = w.nodes[ifound];
                ch.f = if_; ch.g = ig; ch.h = ih;
                ch.parent_index = cn.index; ch.parent_pos = cur;
                ch.forward = (int8_t)(i < p.n_steer ? 1 : 0); ch.steer_i = (int8_t)(i % p.n_steer);
                if constexpr (S::HEAP_POS) hp = ch.heap_pos;   // current slot: earlier pushes of this pop may have moved it
                pl_heap_set_key(w, s, hp, if_);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            wave_sync();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    if (nheap > S::HEAP_LDS) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (the pop below reads entries in the workspace)
    if (lane == 0) s.nheap = nheap;
    wave_sync();
    const long long t_r3 = PH_NOW();
    // Pop ahead: the open list is final for this pop, so lane 0 takes the next node off it right away instead of at the
    // top of the next iteration behind two workgroup barriers (the other waves are still checking the shot). If the shot
    // then succeeds the search is over and only the open-list COUNT is reported, which the caller restores.
    if (split) {
        // the popped node may be one of this pop's children: its record must be complete before its state is touched
        if (lane == 0) while (*(volatile int32_t*)&s.wr_done == 0) __builtin_amdgcn_s_sleep(1);
        wave_sync();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
#if PL_HEAP_POP_WAVE
    if (pop_ahead && nheap > 0) {                       // (nheap: the count lane 0 has just stored, the same in every lane)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // heappop returns the root; the sift that follows only restores the heap. Publish the node first: on a record pop
        // another wave fetches its expansion record (and waits for a pending one) while this wave walks the heap.
        // (lanes 1 and 2 read the root's children in the same instruction: the smaller key is the list's second best, which the
        //  lookahead's dive prediction compares a child's cost with, pl_look_predict)
        uint32_t root;
        if constexpr (S::LOOK_SECOND) {
            const PlHeapEnt top3 = pl_heap_get(w, s, (lane < 3 && lane < nheap) ? lane : 0);
            root = (uint32_t)__builtin_amdgcn_readfirstlane((int)top3.node);
            const double k1 = pl_readlane_f64(top3.f, 1), k2 = pl_readlane_f64(top3.f, 2);
            if (lane == 0) s.fetch_second = nheap < 2 ? INFINITY : (nheap < 3 || k1 < k2) ? k1 : k2;
        } else root = pl_heap_get(w, s, 0).node;          // (the group forms: exactly the code of rounds 3 - 5)
        if (lane == 0) {
            s.next_cur = (int32_t)root; s.have_next = 1; s.fetch_nheap = nheap - 1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            *(volatile int32_t*)&s.fetch_go = 1;
        }
        const uint32_t c = pl_heap_pop_wave(w, s, nheap);
        if (lane == 0) { s.nheap = nheap - 1; w.nodes[c].state = 3; }
    }
#else
    if (lane == 0 && pop_ahead && s.nheap > 0) {
        const uint32_t root = pl_heap_get(w, s, 0).node;
        s.next_cur = (int32_t)root; s.have_next = 1; s.fetch_nheap = s.nheap - 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        *(volatile int32_t*)&s.fetch_go = 1;
        const uint32_t c = pl_heap_pop(w, s);
        w.nodes[c].state = 3;
    }
#endif
    wave_sync();
    if constexpr (PROFILE) { if (threadIdx.x == 0) { s.phase[PH_RES_CLASSIFY] += t_r1 - t_r0; s.phase[PH_RES_WRITE] += t_r2 - t_r1; s.phase[PH_RES_PUSH] += t_r3 - t_r2; s.phase[PH_SPARE] += clock64() - t_r3; } }
}

// finish_path (hybrid_a_star.py:351-389) + assembly (path_planner.py:100-108): ONE thread writes the record of problem
// pid straight to global memory (no local copy of the struct: its dynamically indexed arrays would be a stack object).
// The counts and the RS fields are filled whether or not the caller asked for way-points (paths == NULL).
// travel_ddt1 / k_dth_ddt1: the motion-primitive constants of one sub-step (LDS copies); sub-step j: x (j + 1), as the reference (:367-374).
// The writer wave of a split resolution (see pl_resolve_fast_wave): once wave 0 has classified the children it creates
// the new nodes (arena records, pose hash) while wave 0 pushes them onto the heap. The two touch different bytes of a node
// record (the pushes its heap_pos), and wave 0 waits for wr_done before it may pop one of these nodes ahead.
template <class S>
AVP_D void pl_resolve_writer_wave(const avp_params& p, const PlanWs& w, S& s, const PlanDims& dims, const PlNode& cn, int nchild)
{
    const int lane = threadIdx.x & 63;
    if (lane == 0) while (*(volatile int32_t*)&s.wr_go == 0) __builtin_amdgcn_s_sleep(1);
    wave_sync();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (*(volatile int32_t*)&s.wr_go == 1) {
        if (lane < nchild) {
            const PlChild& c = s.child[lane];
            if (c.cls == CL_NEW_CLOSED || c.cls == CL_NEW_OPEN) {
                const bool open = c.cls == CL_NEW_OPEN;
                PlNode& nd = w.nodes[c.pos];
                nd.x = c.x; nd.y = c.y; nd.th = c.th;
                nd.g = open ? c.g : 0.0; nd.h = open ? c.h : 0.0; nd.f = open ? c.f : 0.0;
                nd.index = (int32_t)s.global_index + lane + 1; nd.parent_index = cn.index; nd.parent_pos = s.cur;
                nd.forward = (int8_t)(lane < p.n_steer ? 1 : 0); nd.steer_i = (int8_t)(lane % p.n_steer);
                nd.state = open ? 1 : 2;
                if (!open) nd.heap_pos = -1;              // (an open node's heap_pos belongs to the push on wave 0)
                pl_hash_put_atomic(w, dims.hashCap, c.pos, c.x, c.y, c.th);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    wave_sync();
    if (lane == 0) *(volatile int32_t*)&s.wr_done = 1;
}

template <bool PROFILE, class S>
AVP_D void pl_write_result(const avp_params& p, const PlanWs& w, S& s, const double travel_ddt1, const double* k_dth_ddt1,
                           avp_plan_result_dev* __restrict__ results, double* __restrict__ paths, int32_t max_path, int64_t pid,
                           int64_t n_pops, int32_t slot, long long t_fin)
{
            avp_plan_result_dev& r = results[pid];
            int32_t status = s.status;
            r.n_pops = (int32_t)n_pops; r.in_radius_last = s.in_radius; r.rs_collision = s.collision;
            r.n_checks = s.n_checks; r.n_rs = s.n_rs; r.n_closed = s.nclosed; r.n_open = s.nheap;
            r.h_cells = s.h_cells; r.h_misses = s.h_misses; r.global_index = s.global_index; r.n_nodes = s.nnodes;
            r.slot = slot;                  // the slot (persistent workgroup) that ran the problem
            double* out = paths ? paths + (size_t)pid * max_path * 4 : nullptr;
            int32_t n_astar = 0, n_final = 0, rs_n = 0, n_rs_pts = 0, rs_dir0 = 0;
            double rs_L = 0.0, rs0 = 0.0, rs1 = 0.0, rs2 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) r.rs_types[k] = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) r.rs_lengths[k] = 0.0;
            if (s.cur >= 0 && (status == 0 || status == 1)) {
                // chain child -> root
                int32_t len = 0;
                for (int32_t node = s.cur; node >= 0; node = w.nodes[node].parent_pos) { len++; if (w.nodes[node].index == 0) break; }
                int32_t cnt = 0;
                bool over = false;
                auto push = [&](double X, double Y, double T, double D) {
                    if (!out) { cnt++; return; }
                    if (cnt < max_path) { out[4 * cnt] = X; out[4 * cnt + 1] = Y; out[4 * cnt + 2] = T; out[4 * cnt + 3] = D; cnt++; } else over = true;
                };
                // walk from the root: position k of the chain is reached by (len-1-k) parent hops
                int32_t prev = -1;
                for (int32_t k = 0; k < len; k++) {
                    int32_t node = s.cur;
                    for (int32_t hop = 0; hop < len - 1 - k; hop++) node = w.nodes[node].parent_pos;
                    const PlNode& nd = w.nodes[node];
                    if (k == 0) push(nd.x, nd.y, nd.th, 0.0);
                    else if (!out) cnt += p.n_sub;
                    else {
                        const PlNode& par = w.nodes[prev];
                        for (int j = 0; j < p.n_sub; j++) {
                            const double tj = travel_ddt1 * (double)(j + 1);
                            const double td = nd.forward ? tj : -tj;
                            const double th_j = avp_pi_2_pi(par.th + k_dth_ddt1[nd.steer_i] * (double)(j + 1));
                            double s_j, c_j;
                            avp_sincos(th_j, s_j, c_j);
                            push(par.x + td * c_j, par.y + td * s_j, th_j, 0.0);
                        }
                    }
                    prev = node;
                }
                n_astar = cnt;
                if (s.rs.n > 0 && s.rs_status == 0 && s.in_radius) {
                    rs_n = s.rs.n; rs_L = s.rs.L / p.maxc;
                    for (int k = 0; k < s.rs.n; k++) { r.rs_types[k] = s.rs.t[k]; r.rs_lengths[k] = s.rs.l[k] / p.maxc; }
                    n_rs_pts = s.rs_npts;
                    rs0 = w.rsbuf[0]; rs1 = w.rsbuf[1]; rs2 = w.rsbuf[2]; rs_dir0 = w.rsdir[0];
                    for (int k = 1; k < s.rs_npts; k++) push(w.rsbuf[3 * k], w.rsbuf[3 * k + 1], w.rsbuf[3 * k + 2], (double)w.rsdir[k]);
                    n_final = cnt;
                }
                if (over) status = 5;
            }
            r.status = status; r.n_astar = n_astar; r.n_final = n_final; r.rs_n = rs_n; r.n_rs_pts = n_rs_pts;
            r.rs_L = rs_L; r.rs_start[0] = rs0; r.rs_start[1] = rs1; r.rs_start[2] = rs2; r.rs_dir0 = rs_dir0;
            if constexpr (PROFILE) { s.phase[PH_FINISH] += clock64() - t_fin; for (int k = 0; k < PH_COUNT; k++) r.phase_cycles[k] = s.phase[k]; }
            else { for (int k = 0; k < PH_COUNT; k++) r.phase_cycles[k] = 0; }
}

// ---- the record store: tags and state words ------------------------------------------------------------------------------
// A record is named by the POSE it expands (and the problem): tag = 48 bits of the pose hash mixed with the problem index. So a
// node that does not exist yet -- a child somebody expects the search to pop soon -- has the same name as the arena node it
// will be, whoever posts it: the owner from the parent's record, or a HELPER from the children it has just computed (it knows no
// arena index). Two poses that share a tag share an entry: the second post is skipped, the record's key words (exact pose bits)
// turn a wrong reader down.
__device__ __forceinline__ unsigned long long pl_look_tag(int64_t pid, double x, double y, double th)
{
    return (pl_pose_hash(x, y, th) ^ ((unsigned long long)(pid + 1) * 0xD1B54A32D192ED03ULL)) >> 16;
}
#define PL_ST_TAG(st) ((st) >> 8)
#define PL_ST_POSTED(st) (((st) & 1ull) != 0ull)
#define PL_ST_READY(st) (((st) & 6ull) == 6ull)
#ifndef PL_LOOK_PREDICT
#define PL_LOOK_PREDICT 1             // post the child that will be the open list's next root as soon as its parent's record is in the owner's hands (pl_look_predict)
#endif
#ifndef PL_LOOK_PREDICT_FETCH
#define PL_LOOK_PREDICT_FETCH 0       // ... also from a record that only arrives with the fetch at the end of the pop before (one pop of lead instead of two): measured 18.3 vs 18.0 ms -- those jobs come too late and load the helpers
#endif
#ifndef PL_LOOK_SECOND
#define PL_LOOK_SECOND 1              // ... and from the record of the open list's SECOND-best node (one more pop of lead, pl_look_second)
#endif
#ifndef PL_LOOK_CHAIN_TOP
#define PL_LOOK_CHAIN_TOP 0           // ... also below a node that is posted while it already sits in the first three heap slots (pl_look_post): 2 measured no gain (16.9 ms either way)
#endif
#ifndef PL_LOOK_CHAIN_KIDS
#define PL_LOOK_CHAIN_KIDS 0          // ... and below the likely children posted at the start of a long pop: 1 measured no gain
#endif
#ifndef PL_LOOK_PRED2
#define PL_LOOK_PRED2 2               // pl_look_predict / pl_look_chain also post the node's OTHER children that beat the rest of the list (they are popped within a few pops and would be
                                      // posted one pop before their own): 1 = the second-cheapest, 2 = every one of them. They are jobs the owner would post anyway, a pop or two later.
#endif
#ifndef PL_LOOK_ANYSEEN
#define PL_LOOK_ANYSEEN 0             // pl_look_predict: 1 = any distance the sweep has reached counts (as for a helper), 0 = only a query that hits the closed frontier
#endif
#ifndef PL_LOOK_CHAIN
#define PL_LOOK_CHAIN 2               // a helper that has computed a predicted child's children posts the next level of the dive itself, this many levels deep (<= 3)
#endif
// job word 0: tag | owner's workgroup << 48 | gear of the node << 60 | chain depth << 61
#define PL_JOB_W0(tag, blk, gear, depth) ((tag) | ((unsigned long long)(blk) << 48) | ((unsigned long long)((gear) ? 1 : 0) << 60) | ((unsigned long long)(depth) << 61))
#define PL_JOB_BLOCK(w0) ((int)(((w0) >> 48) & 0xfffull))
#define PL_JOB_GEAR(w0) ((int)(((w0) >> 60) & 1ull))
#define PL_JOB_DEPTH(w0) ((int)(((w0) >> 61) & 3ull))
// Owner or helper (one lane per tag): take the tag's entry for a new pair of jobs. True = this lane posts them. False: the tag
// has its jobs already (in flight or done), or the entry is busy with another tag's jobs in flight (*busy counts those).
__device__ __forceinline__ bool pl_look_claim(const PlLook& look, unsigned long long tag, int32_t* busy)
{
    unsigned long long* q = look.state + pl_look_ent(look, tag);
    const unsigned long long st = pl_ld64(q);
    if (PL_ST_TAG(st) == tag && PL_ST_POSTED(st)) return false;
    if (st != 0ull && !PL_ST_READY(st)) { *busy += 1; return false; }
    const unsigned long long nw = (tag << 8) | (((((st >> 3) & 31ull) + 1ull) & 31ull) << 3) | 1ull;
    return atomicCAS(q, st, nw) == st;
}
// One wave copies the record of entry ri (state word st: the caller saw it READY) to rec[] and validates it: the state word
// must be the same after the copy (seqlock: no take-over began meanwhile), and the key words must name this node's exact pose
// bits, this problem and this goal. Not used: a record whose shot failed to solve or came out collision free (that pop ends
// the search or raises: the long way).
template <class S>
__device__ __forceinline__ int pl_look_copy(const PlLook& look, const PlanWs& w, S& s, int64_t pid, int32_t node, int lane, size_t ri, unsigned long long st, unsigned long long* rec)
{
    const unsigned long long* rp = look.recs + ri * PL_REC_WORDS;
    rec[lane] = pl_ld64(rp + lane);
    if (lane + 64 < PL_REC_WORDS) rec[lane + 64] = pl_ld64(rp + lane + 64);
    PL_LOOK_DRAIN();
    wave_sync();
    unsigned long long st2 = 0;
    if (lane == 0) st2 = PL_FLAG_LD64(look.state + ri);
    st2 = __shfl(st2, 0, 64);
    const PlNode& nn = w.nodes[node];
    const unsigned long long fl = rec[84];
    const bool r_in = fl & 1ull, r_err = (fl >> 8) & 0xffull, r_hit = (fl >> 1) & 1ull;
    if (st2 != st) { if (lane == 0) atomicAdd(&s.n_torn, 1); return 0; }
    return rec[80] == pl_bits(nn.x) && rec[81] == pl_bits(nn.y) && rec[82] == pl_bits(nn.th) &&
           rec[83] == pl_look_key3(pid, s.goal[2]) && rec[86] == pl_bits(s.goal[0]) && rec[87] == pl_bits(s.goal[1]) &&
           !(r_in && (r_err || !r_hit));
}
// Owner side of the lookahead (one wave): the expansion record of `node`, if both halves are finished, is copied to
// s.recb[buf]; returns whether it is there and valid (pl_look_copy). Consumer order: state word, payload, state word.
template <class S>
__device__ __forceinline__ int pl_look_load(const PlLook& look, const PlanWs& w, S& s, int64_t pid, int32_t maxNodes, int32_t node, int lane, int buf, bool wait)
{
    int ok = 0;
    if (node >= 0) {
        unsigned long long st = 0;
        size_t ri = 0;
        if (lane == 0) {
            const PlNode& nn = w.nodes[node];
            const unsigned long long tag = pl_look_tag(pid, nn.x, nn.y, nn.th);
            ri = pl_look_ent(look, tag);
            st = PL_FLAG_LD64(look.state + ri);
            if (!(PL_ST_TAG(st) == tag && PL_ST_POSTED(st))) st = 0ull;
            if (wait) {
                atomicAdd(&s.n_sec[st == 0ull ? 0 : PL_ST_READY(st) ? 2 : 1], 1);      // (diagnostics: the next pop's node was not posted / pending / ready)
                // ... and what kind of pop misses: a dive (child of the node being expanded) below a record pop / below a long pop / another node
                if (st == 0ull || !PL_ST_READY(st)) atomicAdd(&s.n_miss[(st == 0ull ? 0 : 3) + (nn.parent_pos == s.cur ? (s.use_rec ? 0 : 1) : 2)], 1);
            }
            if (PL_LOOK_WAIT > 0 && wait && s.look_calm && st != 0ull && !PL_ST_READY(st)) {
                // (an entry whose jobs are in flight is nobody else's to claim: the tag stays while we wait)
                const long long t0 = clock64();
                while (!PL_ST_READY(st) && clock64() - t0 < PL_LOOK_WAIT) { __builtin_amdgcn_s_sleep(4); st = PL_FLAG_LD64(look.state + ri); }
                atomicAdd(&s.n_sec[3], 1);
            }
            // (the next pop's own look-up: a record that is posted but not finished is polled again while that pop goes the long way)
            if (PL_LOOK_LATE && wait) { s.poll_ri = (unsigned long long)ri; s.poll_tag = tag; s.poll_on = (st != 0ull && !PL_ST_READY(st)) ? 1 : 0; }
        }
        st = __shfl(st, 0, 64);
        ri = (size_t)__shfl((unsigned long long)ri, 0, 64);
        if (st != 0ull && PL_ST_READY(st)) ok = pl_look_copy(look, w, s, pid, node, lane, ri, st, s.recb[buf]);
    }
    return ok;
}

// One wave posts the jobs of its lanes that `want` one to BOTH rings (children half, shot half): one ticket range per ring.
// Job words: w0 (PL_JOB_W0), pose, thr = the cost below which a child of this node would be the open list's next root
// (chained prediction, pl_look_chain; +inf for a job that is not part of a predicted dive), the problem.
__device__ __forceinline__ void pl_ring_post2(const PlLook& look, int lane, bool want, unsigned long long w0, double x, double y, double th, double thr, int64_t pid)
{
    const unsigned long long mask = __ballot(want);
    if (!mask) return;
    unsigned long long t = 0;
    if (lane < 2) t = atomicAdd(look.ctrl + 64 * lane, (unsigned long long)__popcll(mask));
    const unsigned long long t0 = __shfl(t, 0, 64), t1 = __shfl(t, 1, 64);
    if (want) {
        const unsigned long long k = (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
        unsigned long long* e0 = look.jobs + ((size_t)((t0 + k) & (PL_JCAP - 1))) * PL_JOB_WORDS;
        unsigned long long* e1 = look.jobs + ((size_t)PL_JCAP + (size_t)((t1 + k) & (PL_JCAP - 1))) * PL_JOB_WORDS;
        const unsigned long long bx = pl_bits(x), by = pl_bits(y), bt = pl_bits(th), b4 = pl_bits(thr), b5 = (unsigned long long)pid;
        pl_st64(e0 + 0, w0); pl_st64(e0 + 1, bx); pl_st64(e0 + 2, by); pl_st64(e0 + 3, bt); pl_st64(e0 + 4, b4); pl_st64(e0 + 5, b5); pl_st64(e0 + 6, 0ull);
        pl_st64(e1 + 0, w0); pl_st64(e1 + 1, bx); pl_st64(e1 + 2, by); pl_st64(e1 + 3, bt); pl_st64(e1 + 4, b4); pl_st64(e1 + 5, b5); pl_st64(e1 + 6, 0ull);
        PL_LOOK_DRAIN();
        PL_FLAG_ST64(e0 + 7, t0 + k + 1ull);
        PL_FLAG_ST64(e1 + 7, t1 + k + 1ull);
    }
}

// Among the lanes' candidate costs (key = bit pattern of a cost >= +0, ~0 = no candidate): the cheapest (the first in lane
// order among equals: child order), and the second-cheapest cost. Uniform results.
__device__ __forceinline__ void pl_look_best2(unsigned long long key, int lane, unsigned long long& mn, int& best, unsigned long long& mn2)
{
    mn = key;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_xor(mn, d, 64); if (o < mn) mn = o; }
    best = mn == ~0ull ? -1 : __ffsll((unsigned long long)__ballot(key == mn)) - 1;
    mn2 = lane == best ? ~0ull : key;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_xor(mn2, d, 64); if (o < mn2) mn2 = o; }
}

// Dive prediction (round 6; one wave, with the validated record of `node` in s.recb[buf]). A quarter of all pops expand a child
// of the node popped just before; until round 5 such a child had a record only if its parent's pop took the long way (which
// posts three likely children) -- three quarters of the long pops that were left were dives below RECORD pops. But a parent's
// record already names its children: poses, first colliding sub-step, Reeds-Shepp length. So as soon as the owner holds the
// record of the node on top of its open list (prefetched during the pop before: two pops of lead) -- or of the list's second-best
// node (three) -- it costs the children as expand_node will (g from calc_node_cost, h = max(field distance / 100, RS length):
// hybrid_a_star.py:243-283) and, when the cheapest new child would be the list's next root -- its f below `second_f`, the
// cheapest key that will be left --, posts THAT child, named by its pose. The job carries the cost its own children have to
// beat (the smaller of second_f and this node's second-cheapest child) and a depth: the helper that computes the child's
// children applies the same rule and posts the next level itself (pl_look_chain) -- a dive of several levels is served at the
// helpers' pace instead of one long pop per level. A guess: children equal to existing nodes, queries that would miss the closed
// frontier and the children of the pop in progress are ignored; a wrong guess is a wasted job, a right one turns the dive's
// long pop (~70 k cycles) into a record pop (~26 k).
template <class S>
__device__ __forceinline__ void pl_look_predict(const PlLook& look, const PlanWs& w, S& s, int64_t pid, int lane, int32_t node, int buf, double second_f)
{
    const DevMap& m = s.km;
    const avp_params& p = s.kp;
    const int nchild = 2 * p.n_steer;
    const unsigned long long* rec = s.recb[buf];
    const PlNode& tn = w.nodes[node];
    const double tth = tn.th;
    const int tfw = tn.forward;
    bool valid = lane < nchild;
    double cx = 0.0, cy = 0.0, cth = 0.0, f = 0.0;
    if (valid) {
        cx = pl_unbits(rec[lane]); cy = pl_unbits(rec[16 + lane]); cth = pl_unbits(rec[32 + lane]);
        const double L = pl_unbits(rec[48 + lane]);
        const uint32_t fc = (uint32_t)(rec[64 + lane] & 0xffffffffull);
        const int err = (int)((rec[64 + lane] >> 32) & 0xffull);
        valid = !(cx > m.b1 || cx < m.b0 || cy > m.b3 || cy < m.b2) && fc == 0x7fffffffu && !err;
        if (valid) {
            const int64_t id = avp_pos_to_index(m, cx, cy);
            const uint32_t d = pl_id_in_range(m, id) ? w.dist[id] : PL_UNSEEN;
            uint32_t hd = PL_UNSEEN;
            valid = pl_hquery_hit(m, s, id, d, hd) && hd != PL_UNSEEN;
            if (PL_LOOK_ANYSEEN && !valid && d != PL_UNSEEN) { valid = true; hd = d; }
            const double hv1 = (double)hd / 100;
            f = pl_node_cost(p, lane < p.n_steer ? 1 : 0, cth, tth, tfw) + (L > hv1 ? L : hv1);
        }
    }
    unsigned long long mn, mn2;
    int best;
    pl_look_best2(valid ? pl_bits(f) : ~0ull, lane, mn, best, mn2);
    if (best < 0 || !(pl_unbits(mn) < second_f)) return;     // (uniform) no candidate / another open node would be popped before it
    const double f2 = mn2 == ~0ull ? INFINITY : pl_unbits(mn2);
    // (the second-cheapest child: when it beats the rest of the list as well it is popped right behind the first one's dive -- a node that
    //  would otherwise be posted one pop before its own)
    const int best2 = (PL_LOOK_PRED2 == 1 && f2 < second_f) ? __ffsll((unsigned long long)__ballot(lane != best && valid && pl_bits(f) == mn2)) - 1 : -1;
    bool want = false;
    unsigned long long w0 = 0;
    if (lane == best || lane == best2 || (PL_LOOK_PRED2 >= 2 && valid && f < second_f)) {
        const unsigned long long tag = pl_look_tag(pid, cx, cy, cth);
        int32_t busy = 0;
        want = pl_look_claim(look, tag, &busy);
        if (busy) atomicAdd(&s.n_busy, busy);
        if (want) atomicAdd(&s.n_pred, 1);
        w0 = PL_JOB_W0(tag, blockIdx.x, lane < p.n_steer, lane == best ? PL_LOOK_CHAIN : 0);
    }
    pl_ring_post2(look, lane, want, w0, cx, cy, cth, f2 < second_f ? f2 : second_f, pid);
}

// The record of the node the next pop expands (wave 0, once that node is known): the prefetched one if it is that node's,
// else loaded now. Not for the only open node: the search may end with that pop and hand back its (colliding) shot,
// which a record does not hold.
template <class S>
__device__ __forceinline__ void pl_look_fetch(const PlLook& look, const PlanWs& w, S& s, int64_t pid, int32_t maxNodes, int32_t node, int lane, int32_t nheap_after,
                                              int64_t hashCap = 0, int nchild = 0, bool lookups = false, bool predict = false)
{
    int ok = 0;
    bool fresh = false;
    if (PL_LOOK_LATE && lane == 0) { s.poll_on = 0; s.late_rec = 0; s.late_rec2 = 0; }
    if (s.look_live && s.status == 0 && nheap_after >= 1 && node >= 0) {       // (no helper yet: no record to look for)
        if (node == s.pre_node && s.pre_ok) { ok = 1; if (lane == 0) atomicAdd(&s.n_sec[2], 1); }
        else { ok = pl_look_load(look, w, s, pid, maxNodes, node, lane, s.rec_cur ^ 1, true); fresh = true; }
    }
    wave_sync();
    if (lane == 0) { s.use_rec = ok; s.n_hits += ok; if (ok) s.rec_cur ^= 1; s.pre_node = -1; s.pre_ok = 0; }
    wave_sync();
    // (a record that was not prefetched has not been looked at for a dive yet; the open list's second-best key as of the pop-ahead
    //  that named `node` was published with it: s.fetch_second)
    if (PL_LOOK_PREDICT && PL_LOOK_PREDICT_FETCH && predict && ok && fresh && s.look_calm) pl_look_predict(look, w, s, pid, lane, node, s.rec_cur, s.fetch_second);
    if (lookups && ok) {
        // (record pop, fetching wave) This pop creates no more nodes -- the writer wave is done before the pop-ahead that
        // named `node` --, so the pose-hash look-ups of the NEXT pop's children are final now: do them here, beside the
        // heap sift on wave 0, instead of at the start of that pop. Two states move until then and are set as they will
        // be: the node being expanded now will be closed, the popped node is marked as such.
        const unsigned long long* rec = s.recb[s.rec_cur];
        if (lane < nchild) {
            const int32_t f = pl_hash_find(w, hashCap, pl_unbits(rec[lane]), pl_unbits(rec[16 + lane]), pl_unbits(rec[32 + lane]));
            int st = f >= 0 ? w.nodes[f].state : 0;
            if (f == s.cur) st = 2;
            if (f == node) st = 3;
            s.nf_found[lane] = f; s.nf_state[lane] = (int8_t)st;
        }
        if (lane == 0) s.nf_node = node;
        wave_sync();
    }
}
// Ahead of that: while the pop is busy elsewhere, an otherwise idle wave copies the record of the node on top of the open
// list -- the next pop's node unless a child of this one beats it -- and looks at it for a dive (pl_look_predict: second_f =
// the list's second-best key, the smaller of the root's two children, read together with the root).
template <class S>
__device__ __forceinline__ int32_t pl_look_prefetch_node(const PlanWs& w, S& s, double& second_f)       // (reads the heap: while nobody changes it)
{
    int32_t node = -1;
    second_f = INFINITY;
    if (s.look_live && s.nheap >= 2) {      // (with one open node left the record would not be used)
        node = (int32_t)pl_heap_get(w, s, 0).node;
        second_f = pl_heap_get(w, s, 1).f;
        if (s.nheap >= 3) { const double f2 = pl_heap_get(w, s, 2).f; if (f2 < second_f) second_f = f2; }
    }
    return node;
}
template <class S>
__device__ __forceinline__ void pl_look_prefetch(const PlLook& look, const PlanWs& w, S& s, int64_t pid, int32_t maxNodes, int lane, int32_t node, double second_f)
{
    const int ok = pl_look_load(look, w, s, pid, maxNodes, node, lane, s.rec_cur ^ 1, false);
    if (lane == 0) { s.pre_node = node; s.pre_ok = ok; }
    if (PL_LOOK_PREDICT && ok && s.look_calm) pl_look_predict(look, w, s, pid, lane, node, s.rec_cur ^ 1, second_f);
}
// ... and one or two more pops ahead: the root's two heap children -- the list's SECOND-best node (the smaller one; the other one's key
// is what its cheapest child has to beat) and, PL_LOOK_SECOND = 2, the other one too (its children have to beat the four nodes of the
// next heap level). Their records, if there, go to the third record buffer one after the other and are looked at for a dive only.
template <class S>
__device__ __forceinline__ void pl_look_second_nodes(const PlanWs& w, S& s, int32_t& n1, double& t1, int32_t& n2, double& t2)       // (reads the heap: while nobody changes it)
{
    n1 = n2 = -1;
    t1 = t2 = INFINITY;
    if (PL_LOOK_SECOND && s.look_live && s.look_calm && s.nheap >= 3) {
        const PlHeapEnt e1 = pl_heap_get(w, s, 1), e2 = pl_heap_get(w, s, 2);
        const bool sw = e2.f < e1.f;
        n1 = (int32_t)(sw ? e2.node : e1.node); t1 = sw ? e1.f : e2.f;
        if (PL_LOOK_SECOND >= 2) {
            n2 = (int32_t)(sw ? e1.node : e2.node);
            double m = INFINITY;
            for (int k = 3; k < 7 && k < s.nheap; k++) { const double f = pl_heap_get(w, s, k).f; if (f < m) m = f; }
            t2 = m;
        }
    }
}
template <class S>
__device__ __forceinline__ void pl_look_second(const PlLook& look, const PlanWs& w, S& s, int64_t pid, int32_t maxNodes, int lane, int32_t node, double thr)
{
    if (node < 0) return;
    const int ok = pl_look_load(look, w, s, pid, maxNodes, node, lane, 2, false);
    if (ok && s.look_calm) pl_look_predict(look, w, s, pid, lane, node, 2, thr);
    wave_sync();
}

// Helper side of a predicted dive (one wave of a children-half helper whose job carries a chain depth; the children of the job's
// node are in s.child[]: poses, first colliding sub-step, Reeds-Shepp length). The same rule as pl_look_predict with what a helper
// can see: the owner's distance field through its workspace slot (agent-scope loads; whether a query would hit the closed frontier is the
// owner's knowledge -- any distance the sweep has reached counts), the cost `thr` a child has to beat from the job. Posts the
// cheapest such child with the depth counted down.
__device__ __forceinline__ void pl_look_chain(const PlLook& look, PlShared& s, const uint32_t* owner_dist, int lane, double nth, int ngear, double thr, int depth, int owner_block, int64_t pid)
{
    const DevMap& m = s.km;
    const avp_params& p = s.kp;
    const int nchild = 2 * p.n_steer;
    // ring counters: only while the helpers keep up (as the owner's posting rounds)
    unsigned long long c = 0;
    if (lane < 4) c = pl_ld64(look.ctrl + (lane == 0 ? 0 : lane == 1 ? 16 : lane == 2 ? 64 : 80));
    const long long backlog = max((long long)(__shfl(c, 0, 64) - __shfl(c, 1, 64)), (long long)(__shfl(c, 2, 64) - __shfl(c, 3, 64)));
    if (backlog > PL_LOOK_BACKLOG) return;
    bool valid = lane < nchild;
    double f = 0.0, cx = 0.0, cy = 0.0, cth = 0.0;
    if (valid) {
        const PlChild& ch = s.child[lane];
        cx = ch.x; cy = ch.y; cth = ch.th;
        valid = !ch.oob && ch.first_coll == 0x7fffffff && !ch.rs_err && pl_id_in_range(m, ch.id);
        if (valid) {
            const uint32_t d = pl_ld32(owner_dist + ch.id);
            valid = d != PL_UNSEEN;
            const double hv1 = (double)d / 100;
            f = pl_node_cost(p, lane < p.n_steer ? 1 : 0, cth, nth, ngear) + (ch.L > hv1 ? ch.L : hv1);
        }
    }
    unsigned long long mn, mn2;
    int best;
    pl_look_best2(valid ? pl_bits(f) : ~0ull, lane, mn, best, mn2);
    if (best < 0 || !(pl_unbits(mn) < thr)) return;
    const double f2 = mn2 == ~0ull ? INFINITY : pl_unbits(mn2);
    bool want = false;
    unsigned long long w0 = 0;
    if (lane == best || (PL_LOOK_PRED2 >= 2 && valid && f < thr)) {
        const unsigned long long tag = pl_look_tag(pid, cx, cy, cth);
        int32_t busy = 0;
        want = pl_look_claim(look, tag, &busy);
        if (want) atomicAdd(look.ctrl + 71, 1ull);           // ([71]: children posted by helpers)
        w0 = PL_JOB_W0(tag, owner_block, lane < p.n_steer, lane == best ? depth - 1 : 0);
    }
    pl_ring_post2(look, lane, want, w0, cx, cy, cth, f2 < thr ? f2 : thr, pid);
}

// Owner side of the lookahead: one wave posts, in ONE round (one look at the ring counters, one ticket range per ring,
// one drain of the payload stores),
//  * lanes 0 .. PL_LOOK_TOP-1: the nodes in the first heap slots that have no job yet (pl_look_candidate reads the heap
//    while nobody changes it; the posting may run later: node poses never change);
//  * lanes 32 .. 34, at the start of a pop that takes the long way (kids): the three children it is most likely to be
//    followed by -- only while the helpers keep up (at most PL_LOOK_BACKLOG jobs waiting in a ring): a child job that
//    queues comes too late anyway, and late records mean more long pops, which post more children; without the gate the
//    system locks into that state (measured: 18 % record pops instead of 75 %).
template <class S>
__device__ __forceinline__ uint32_t pl_look_candidate(const PlanWs& w, S& s, int lane, double& key)
{
    uint32_t node = 0xffffffffu;
    key = INFINITY;
    if (lane < PL_LOOK_TOP && lane < s.nheap) { const PlHeapEnt e = pl_heap_get(w, s, lane); node = e.node; key = e.f; if (node >= (uint32_t)s.nnodes) node = 0xffffffffu; }
    return node;
}
template <class S>
__device__ __forceinline__ void pl_look_post(const PlLook& look, const PlanWs& w, S& s, const avp_params& p, const PlNode& cn, int64_t pid,
                                             int32_t maxNodes, int lane, uint32_t node, double node_key, bool kids)
{
    static_assert(PL_LOOK_TOP <= 32, "lanes 32 .. 34 post the children");
    // ring counters (head counts the tickets drawn: it runs ahead of the tail while helpers wait for work)
    unsigned long long c = 0;
    if (lane < 5) c = pl_ld64(look.ctrl + (lane == 0 ? 48 : lane == 1 ? 0 : lane == 2 ? 16 : lane == 3 ? 64 : 80));
    const int d = lane - 32 - PL_LOOK_KSPAN, sc = (int)cn.steer_i + d;
    bool kid = kids && lane >= 32 && lane < 32 + PL_LOOK_KIDS && cn.steer_i >= 0 && sc >= 0 && sc < p.n_steer;
    const unsigned long long helpers_all = __shfl(c, 0, 64), ta = __shfl(c, 1, 64), ha = __shfl(c, 2, 64), tb = __shfl(c, 3, 64), hb = __shfl(c, 4, 64);
    // (too few helpers for the searches still running -- a large batch before its tail: their records would all come late, and the
    //  bookkeeping of a pop that finds none costs more than it gains: as if there were none yet. The count of helpers only grows.)
    const long long owners_left = (long long)gridDim.x - (long long)helpers_all;      // (every workgroup of the launch that is no helper now: the helper-only workgroups of a launch smaller than the chip are helpers from the start)
    const unsigned long long helpers = 4ll * (long long)helpers_all >= (long long)PL_LOOK_RATIO_X4 * owners_left ? helpers_all : 0ull;
    const long long backlog = max((long long)(ta - ha), (long long)(tb - hb));
    if (lane == 0) { s.look_calm = (helpers != 0 && backlog <= PL_LOOK_BACKLOG) ? 1 : 0; s.look_live = helpers != 0 ? 1 : 0; }   // (calm: the helpers keep up, a pending record is worth a short wait)
    // nobody to serve / a ring is nearly full: every owner may post up to PL_LOOK_TOP + PL_LOOK_KIDS + 2 jobs from the same (stale)
    // reading of the counters, so the margin is several times owners x jobs (512 x 21 = 10 752) -- an unread entry is never overwritten
    if (helpers == 0 || backlog > PL_JCAP - 16384) return;
    if (backlog > PL_LOOK_BACKLOG) kid = false;
    // (helpers behind: only the first PL_LOOK_TOP_BUSY heap slots are worth a job -- the nodes popped next; the rest of the list would
    //  only lengthen the queue the urgent jobs wait in)
    const bool scarce = 4ll * (long long)helpers_all < (long long)PL_LOOK_SCARCE_X4 * owners_left;      // fewer helpers per running search than the blanket posting needs
    const long long backlog2 = PL_LOOK_B2_DIV > 0 ? max(8ll, (long long)helpers_all / max(PL_LOOK_B2_DIV, 1)) : (long long)PL_LOOK_BACKLOG2;
    const bool cand = lane < 32 && node != 0xffffffffu && (!scarce || ((backlog <= PL_LOOK_BACKLOG || lane < PL_LOOK_TOP_BUSY) && (backlog <= backlog2 || lane < PL_LOOK_TOP_BUSY2)));
    // the pose names the record: a node's own, or the child's the children stage will compute (hybrid_a_star.py:134-151)
    double x = 0.0, y = 0.0, th = 0.0;
    int gear = 1;
    if (kid) {
        const double travel = cn.forward ? p.travel_dt : -p.travel_dt;
        th = avp_pi_2_pi(cn.th + s.k_dth_dt[sc]);
        double sth, cth;
        avp_sincos(th, sth, cth);
        x = cn.x + travel * cth;
        y = cn.y + travel * sth;
        gear = cn.forward;
    } else if (cand) {
        const PlNode& nd = w.nodes[node];
        x = nd.x; y = nd.y; th = nd.th; gear = nd.forward;
    }
    // A node in the first three heap slots that has no job YET is a fresh arrival at the top of the list -- a child of the node popped
    // just before, or of the one before that: its own pop will come before its record, but the helper that computes its children can post
    // the next level of the dive at once (pl_look_chain): the job carries a chain depth and the cost a child has to beat, the cheapest
    // key among the other two of the three. Likewise a child posted at the start of a long pop: its children have to beat the list's root.
    const double k0 = __shfl(node_key, 0, 64), k1 = __shfl(node_key, 1, 64), k2 = __shfl(node_key, 2, 64);
    int depth = 0;
    double thr = INFINITY;
    if (PL_LOOK_CHAIN_TOP > 0 && cand && lane < 3) { depth = PL_LOOK_CHAIN_TOP; thr = lane == 0 ? (k1 < k2 ? k1 : k2) : lane == 1 ? (k0 < k2 ? k0 : k2) : (k0 < k1 ? k0 : k1); }
    if (PL_LOOK_CHAIN_KIDS > 0 && kid) { depth = PL_LOOK_CHAIN_KIDS; thr = k0; }
    bool want = false;
    unsigned long long w0 = 0;
    if (kid || cand) {
        const unsigned long long tag = pl_look_tag(pid, x, y, th);
        int32_t busy = 0;
        want = pl_look_claim(look, tag, &busy);
        if (busy) atomicAdd(&s.n_busy, busy);
        w0 = PL_JOB_W0(tag, blockIdx.x, gear, depth);
    }
    pl_ring_post2(look, lane, want, w0, x, y, th, thr, pid);
}

// ---- called parts of plan_kernel (whole workgroup; they read the kernel's arguments through PlShared's LDS copies) ------------
// per-problem set-up: hybrid_a_star.__init__ (hybrid_a_star.py:72-124)
template <bool PROFILE>
__device__ __noinline__ void plk_init(AVP_LDS PlShared* sp, int64_t pid)
{
    PlShared& s = *(PlShared*)sp;
    const DevMap& m = s.km;
    const PlanWs& w = s.kw;
    const int tid = threadIdx.x;
    const double sx = s.k_starts[3 * pid], sy = s.k_starts[3 * pid + 1], sth = s.k_starts[3 * pid + 2];
    const double gx = s.k_goals[3 * pid], gy = s.k_goals[3 * pid + 1], gth = s.k_goals[3 * pid + 2];
    const long long t_init0 = PH_NOW();
    const bool pose_ok = pl_pose_ok(sx, sy, sth) && pl_pose_ok(gx, gy, gth);      // (the same in every thread)
    for (int64_t i = tid; i < s.kdims.hashCap; i += PL_THREADS) w.hash[i] = 0;
    if (tid == 0) {
        s.status = 0; s.done = 0;
        s.nnodes = 0; s.nheap = 0; s.nclosed = 0; s.closed_nonempty = 0; s.have_next = 0; s.next_cur = -1; s.nf_node = -1;
        s.global_index = 0; s.cur = -1; s.n_checks = 0; s.n_rs = 0;
        for (int k = 0; k < PH_COUNT; k++) s.phase[k] = 0;
        s.goal[0] = gx; s.goal[1] = gy; s.goal[2] = pose_ok ? avp_pi_2_pi(gth) : 0.0;
        s.rs_status = 0; s.rs_npts = 0; s.in_radius = 0; s.collision = 0; s.rs.n = 0; s.rs.L = 0;
        if (!pose_ok) { s.status = 7; s.E = 0; s.h_cells = 0; s.h_misses = 0; s.qover = 0; }      // AVP_PLAN_BAD_POSE
    }
    if (pose_ok) pl_sweep_init(m, w, s, s.kdims, gx, gy);
    else __syncthreads();
    if (s.status == 0) {
        // hybrid_a_star.__init__: compute_path(x0, y0) (:89-91)
        const int64_t sid = avp_pos_to_index(m, sx, sy);
        pl_hquery_miss<PROFILE>(m, w, s, sid);
        if (tid == 0) {
            if (s.hq_d == PL_UNSEEN) s.status = s.qover ? 5 : 2;
            else {
                PlNode& nd = w.nodes[0];
                nd.x = sx; nd.y = sy; nd.th = avp_pi_2_pi(sth); nd.g = 0; nd.h = 0; nd.f = 0;
                nd.index = 0; nd.parent_index = -1; nd.parent_pos = -1; nd.forward = 1; nd.steer_i = -1; nd.state = 1;
                s.nnodes = 1;
                pl_heap_push(w, s, 0, 0.0);
                pl_hash_put(w, s.kdims.hashCap, 0);
            }
        }
        __syncthreads();
    }
    PH_ACC(PH_INIT, t_init0);
}
// a heuristic query that misses the closed frontier: the workgroup extends the sweep (compute_h.py:198-214)
template <bool PROFILE>
__device__ __noinline__ void plk_sweep_extend(AVP_LDS PlShared* sp)
{
    PlShared& s = *(PlShared*)sp;
    pl_hquery_miss<PROFILE>(s.km, s.kw, s, s.pending_id);
}
// finish_path + assembly (one thread)
template <bool PROFILE>
__device__ __noinline__ void plk_write_result(AVP_LDS PlShared* sp, int64_t pid, int64_t n_pops, int32_t slot, long long t_fin)
{
    PlShared& s = *(PlShared*)sp;
    pl_write_result<PROFILE>(s.kp, s.kw, s, s.k_travel_ddt1, s.k_dth_ddt1, s.k_results, s.k_paths, s.k_max_path, pid, n_pops, slot, t_fin);
}

// the owner side of the lookahead (one wave each), as called functions: five call sites in the pop loop
__device__ __noinline__ void plk_look_post(AVP_LDS PlShared* sp, int64_t pid, int32_t maxNodes, uint32_t node, double node_key, int kids,
                                           double cnx, double cny, double cnth, int cn_forward, int cn_steer)
{
    PlShared& s = *(PlShared*)sp;
    PlNode cn;
    cn.x = cnx; cn.y = cny; cn.th = cnth; cn.forward = (int8_t)cn_forward; cn.steer_i = (int8_t)cn_steer;     // (what pl_look_post reads of the node)
    pl_look_post(s.klook, s.kw, s, s.kp, cn, pid, maxNodes, threadIdx.x & 63, node, node_key, kids != 0);
}
__device__ __noinline__ void plk_look_fetch(AVP_LDS PlShared* sp, int64_t pid, int32_t maxNodes, int32_t node, int32_t nheap_after, int nchild, int flags)
{
    // flags: bit 0 = do the pose-hash look-ups of the node's children (record pop, fetching wave), bit 1 = the node was named by a pop-ahead
    // (s.fetch_second is that pop-ahead's): look at a freshly loaded record for a dive
    PlShared& s = *(PlShared*)sp;
    pl_look_fetch(s.klook, s.kw, s, pid, maxNodes, node, threadIdx.x & 63, nheap_after, s.kdims.hashCap, nchild, (flags & 1) != 0, (flags & 2) != 0);
}
__device__ __noinline__ void plk_look_prefetch(AVP_LDS PlShared* sp, int64_t pid, int32_t maxNodes, int32_t node, double second_f)
{
    PlShared& s = *(PlShared*)sp;
    pl_look_prefetch(s.klook, s.kw, s, pid, maxNodes, threadIdx.x & 63, node, second_f);
}

__device__ __noinline__ void plk_look_second(AVP_LDS PlShared* sp, int64_t pid, int32_t maxNodes, int32_t n1, double t1, int32_t n2, double t2)
{
    PlShared& s = *(PlShared*)sp;
    pl_look_second(s.klook, s.kw, s, pid, maxNodes, threadIdx.x & 63, n1, t1);
    pl_look_second(s.klook, s.kw, s, pid, maxNodes, threadIdx.x & 63, n2, t2);
}

// Late adoption (round 5). A pop goes the long way when its node's record is not there at the moment the node is popped -- for a
// child of the node popped just before (a dive: a quarter of all pops) its job was posted one pop ago and takes ~42 k cycles,
// the long way ~75 k. So the long way is started AND the record is looked for again while it runs: wave 0, in the slack it has
// ahead of the barrier behind the sub-step checks and of the one behind the Reeds-Shepp words (it arrives 4 - 6 k cycles before
// the other waves at both), polls the pending record's ready bits, and if both halves have landed copies and validates the
// record exactly as pl_look_load does. The pop then leaves the long way at that barrier -- nothing it has done so far has
// touched a counter, the arena, the heap or the hash -- and is resolved from the record like any record pop. Whether and when
// a record lands changes the time of a pop, never a result (tests/test_gpu_lookahead.py).
// The two polls publish through flags of their own (second = 0: s.late_rec, read by every wave behind the barrier of exit #1;
// second = 1: s.late_rec2, read behind the barrier of exit #2). With ONE flag wave 0 could run ahead through its Reeds-Shepp
// words (no barrier inside on the one-pass path) and set it at the second poll while a slower wave had not yet read it for
// exit #1: that wave would leave the long way alone and the workgroup's barrier sequences would no longer match. A flag is
// only ever written before the barrier its readers sit behind, and cleared a barrier later (pl_look_fetch).
__device__ __noinline__ void plk_look_late(AVP_LDS PlShared* sp, int64_t pid, int second)
{
    PlShared& s = *(PlShared*)sp;
    if (!s.poll_on || s.late_rec || s.late_rec2) return;               // (uniform: only wave 0 writes the three, and only here / in pl_look_fetch)
    const PlLook& look = s.klook;
    const int lane = threadIdx.x & 63;
    const size_t ri = (size_t)s.poll_ri;
    unsigned long long st = 0;
    if (lane == 0) st = PL_FLAG_LD64(look.state + ri);
    st = __shfl(st, 0, 64);
    if (PL_ST_TAG(st) != s.poll_tag || !PL_ST_READY(st)) return;      // (not there yet -- or finished AND taken over by another tag since: gone)
    // (the buffer of the popped node's record: free on the long way; the other one is the prefetch target. As pl_look_load / pl_look_fetch: not the record of the only open node)
    const bool ok = pl_look_copy(look, s.kw, s, pid, s.cur, lane, ri, st, s.recb[s.rec_cur]) && s.nheap >= 1;
    if (lane == 0) { s.poll_on = 0; if (ok) { if (second) s.late_rec2 = 1; else s.late_rec = 1; s.n_hits += 1; s.n_late += 1; } }
}

template <bool STAGE, bool PROFILE, bool LOOK = false>
__global__ __launch_bounds__(PL_THREADS) void plan_kernel(DevMap m, avp_params p, const double* __restrict__ starts,
                                                          const double* __restrict__ goals, int64_t n, int32_t maxNodes,
                                                          char* __restrict__ workspace, unsigned int* __restrict__ counter,
                                                          avp_plan_result_dev* __restrict__ results,
                                                          double* __restrict__ paths, int32_t max_path,
                                                          double* __restrict__ trace, int32_t max_trace, int32_t retry_only, PlLook look,
                                                          const int32_t* __restrict__ order)
{
    avp_lds_tables_fill<true>();
    rs_lds_tables_fill();
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];
    PlShared& s = *reinterpret_cast<PlShared*>(pl_smem);
    const PlanDims dims = plan_dims(m.S, m.Sy, maxNodes);
    // with LOOK the workgroups past look.main_blocks own no workspace slot: they only ever serve as helpers
    const bool own_ws = !LOOK || (int32_t)blockIdx.x < look.main_blocks;
    PlanWs w = plan_carve(workspace + (size_t)(own_ws ? blockIdx.x : 0) * dims.bytes, dims);
    if constexpr (LOOK) if (!own_ws) {
        w.rsbuf = (double*)(look.hrs + (size_t)((int32_t)blockIdx.x - look.main_blocks) * PL_LOOK_HRS);
        w.rsdir = (int8_t*)(look.hrs + (size_t)((int32_t)blockIdx.x - look.main_blocks) * PL_LOOK_HRS + pl_al((size_t)PL_RS_CAP * 3 * 8));
    }
    const int tid = threadIdx.x;
    if (tid == 0) { s.sched_cnt = -1; s.sched_n = 0; s.use_rec = 0; s.job_skip = 0; s.helper_reg = 0; s.n_hits = 0; s.look_calm = 0; s.look_live = 0; s.n_sec[0] = s.n_sec[1] = s.n_sec[2] = s.n_sec[3] = 0; s.rec_cur = 0; s.pre_node = -1; s.pre_ok = 0; s.poll_on = 0; s.late_rec = 0; s.late_rec2 = 0; s.n_late = 0; s.n_pred = 0; s.n_busy = 0; s.n_torn = 0; s.poll_tag = 0; s.fetch_second = INFINITY; for (int k = 0; k < 6; k++) s.n_miss[k] = 0; }
    // The lane-indexed constants of avp_params are read through LDS copies only: a dynamically indexed member of the
    // by-value kernel argument would make the compiler copy the whole struct (1 KB) to every lane's scratch.
#pragma unroll
    for (int k = 0; k < AVP_MAX_STEER; k++) if (tid == k) {
        s.k_steer[k] = p.steer[k]; s.k_dth_dt[k] = p.dth_dt[k]; s.k_dth_ddt1[k] = p.dth_ddt1[k];
    }
    if (tid == 0) s.k_travel_ddt1 = p.travel_ddt1;
    if (p.n_sub > 0 && p.n_steer > 0)
        for (int t = tid; t < PL_MAXSUBS && t < 2 * p.n_steer * p.n_sub; t += PL_THREADS) { const int ci = t / p.n_sub; s.sub_child[t] = (int8_t)ci; s.sub_j[t] = (int8_t)(t - ci * p.n_sub); s.sub_steer[t] = (int8_t)(ci % p.n_steer); }
    // STAGE: the column bitmaps and node coordinates of the map live in LDS behind PlShared for the whole
    // (persistent) lifetime of the workgroup; otherwise they are read through L1/L2
    MapTabs mt;
    if (STAGE) {
        uint64_t* lb = reinterpret_cast<uint64_t*>(pl_smem + pl_lds_tables_offset());
        double* lx = reinterpret_cast<double*>(lb + (size_t)m.nx * m.wpc);
        double* ly = lx + m.nx;
        for (int i = tid; i < m.nx * m.wpc; i += PL_THREADS) lb[i] = m.colBits[i];
        for (int i = tid; i < m.nx; i += PL_THREADS) lx[i] = m.X[i];
        for (int i = tid; i < m.ny; i += PL_THREADS) ly[i] = m.Y[i];
        mt.X = lx; mt.Y = ly; mt.bits = lb;
        __syncthreads();
    } else { mt.X = m.X; mt.Y = m.Y; mt.bits = m.colBits; }
    if (tid == 0) {
        s.mt = mt; pl_chk_env_fill(s.env, m, p, STAGE ? mt.X : nullptr, STAGE ? mt.Y : nullptr, STAGE ? mt.bits : nullptr);
        s.km = m; s.kp = p; s.kdims = dims; s.kw = w; s.klook = look;
        s.k_starts = starts; s.k_goals = goals; s.k_results = results; s.k_paths = paths; s.k_max_path = max_path; s.k_pad = 0;
    }
    const int nchild = 2 * p.n_steer;
    const int64_t max_pops = p.max_pops > 0 ? p.max_pops : (int64_t)1 << 40;

    for (;;) {
        __syncthreads();
        if (tid == 0) {
            // problems are taken in the caller's order (order[ticket], e.g. the expected longest first) or by index
            const uint32_t t = own_ws ? atomicAdd(counter, 1u) : 0xffffffffu;
            s.pid = (int64_t)t < n ? (order ? order[t] : (int32_t)t) : 0x7fffffff;
        }
        __syncthreads();
        const int64_t pid = s.pid;
        // no problem left: done -- or, with LOOK, serve the owners of the unfinished problems until all are finished
        const bool helper = LOOK && pid >= n;
        const int hk = (int)(blockIdx.x & 1u);              // a helper serves one kind of job: 0 = children halves, 1 = shot halves (an even split is the optimum: 3 / 8, 5 / 8, 6 / 8 measured, NOTEBOOK round 6)
        const bool hC = helper && hk == 0, hS = helper && hk == 1;
        if (pid >= n && !helper) break;
        // second launch behind plan_wave_kernel: only the problems it handed back (status 100 = AVP_PLAN_RETRY)
        if (!helper && retry_only && results[pid].status != 100) continue;
        if (helper) {
            if (tid == 0) {
                s.status = 0; s.done = 0; s.cur = -1; s.closed_nonempty = 0; s.nnodes = 0; s.nheap = 0; s.have_next = 0; s.use_rec = 0;
                s.n_checks = 0; s.n_rs = 0; s.rs.n = 0;
                if (!s.helper_reg) { s.helper_reg = 1; atomicAdd(look.ctrl + 48, 1ull); }
            }
            __syncthreads();
        } else {
        plk_init<PROFILE>((AVP_LDS PlShared*)&s, pid);
        }
        const int wave = tid >> 6, lane = tid & 63;
        const int nwave = PL_THREADS / 64;
        int64_t n_pops = 0;
        // ---- main loop: path_planner.py:68-98 ------------------------------------------------------
        while (s.status == 0 && !s.done) {
            __syncthreads();
            { const long long t_pop = PH_NOW();
            const int ahead = s.have_next;
            if (tid == 0 && !helper) {
                if (s.have_next) { s.have_next = 0; s.cur = s.next_cur; }     // popped ahead by wave 0 (n_pops < max_pops held there)
                else if (s.nheap == 0) { s.status = 1; }
                else if (n_pops >= max_pops) { s.status = 4; }
                else {
                    const uint32_t c = pl_heap_pop(w, s);
                    s.cur = (int32_t)c;
                    w.nodes[c].state = 3;
                }
            }
            if constexpr (LOOK) {
                if (helper) {
                    // next job of the ring (tickets are served in order; a ticket past the tail waits for its job)
                    if (tid == 0) {
                        const unsigned long long ticket = atomicAdd(look.ctrl + 64 * hk + 16, 1ull);
                        const unsigned long long* e = look.jobs + ((size_t)hk * PL_JCAP + (size_t)(ticket & (PL_JCAP - 1))) * PL_JOB_WORDS;
                        int got = 0;
                        for (int spin = 0; spin < (1 << 23); spin++) {
                            const unsigned long long seq = PL_FLAG_LD64(e + 7);
                            if (seq == ticket + 1ull) { got = 1; break; }
                            if (seq > ticket + 1ull) { got = 2; break; }
                            if (pl_ld64(look.ctrl + 32) >= (unsigned long long)n) break;
                            __builtin_amdgcn_s_sleep(PL_LOOK_SLEEP);
                        }
                        s.job_skip = 0;
                        if (got == 1) {
                            for (int k = 0; k < 7; k++) s.job[k] = pl_ld64(e + k);
                            if (pl_ld64(e + 7) != ticket + 1ull) s.job_skip = 1;
                            // (the goal is the problem's: the heading wrapped as plk_init wraps it -- owners of refused poses post nothing)
                            const int64_t jp = (int64_t)s.job[5];
                            s.goal[0] = goals[3 * jp]; s.goal[1] = goals[3 * jp + 1]; s.goal[2] = avp_pi_2_pi(goals[3 * jp + 2]);
                        } else if (got == 2) s.job_skip = 1;
                        else s.done = 1;
                    }
                } else if (look.on && wave == 0 && !ahead) {
                    // (a node popped ahead had its record fetched right behind the resolution that popped it)
                    wave_sync();
                    plk_look_fetch((AVP_LDS PlShared*)&s, pid, maxNodes, s.cur, s.nheap, 0, 0);
                }
            }
            PH_ACC(PH_POP, t_pop); }
            __syncthreads();
            if (s.status != 0) break;
            if constexpr (LOOK) if (helper) { if (s.done) break; if (s.job_skip) continue; }
            PlNode cn;
            if (!helper) cn = w.nodes[s.cur];
            else {
                cn.x = pl_unbits(s.job[1]); cn.y = pl_unbits(s.job[2]); cn.th = pl_unbits(s.job[3]);
                cn.g = 0; cn.h = 0; cn.f = 0; cn.index = 0; cn.parent_index = -1; cn.parent_pos = -1; cn.forward = (int8_t)PL_JOB_GEAR(s.job[0]); cn.steer_i = -1; cn.state = 3; cn.heap_pos = -1;      // (the gear: what a chained prediction costs this node's children with)
            }
            const bool use_rec = LOOK && !helper && s.use_rec;
#ifndef PL_PH_LONG
#define PL_PH_LONG 0
#endif
            const bool ph_on = !LOOK || (PL_PH_LONG ? !use_rec : use_rec);      // (instrumented lookahead run: the per-wave timeline covers the record pops only; PL_PH_LONG: the long pops only)
            if (PROFILE && LOOK && tid == 0 && ph_on) s.phase[PH_X0 + 7] += 1;
            uint32_t look_node = 0xffffffffu;
            double look_key = INFINITY;
            if constexpr (LOOK) if (!helper && look.on && wave == nwave - 1) {
                look_node = pl_look_candidate(w, s, lane, look_key);
                if (!use_rec) {
                    plk_look_post((AVP_LDS PlShared*)&s, pid, maxNodes, look_node, look_key, 1, cn.x, cn.y, cn.th, cn.forward, cn.steer_i);      // (hidden behind the sub-step checks)
                    if (PL_LOOK_PREDICT && PL_LOOK_SECOND) { double ta, tb; int32_t na, nb; pl_look_second_nodes(w, s, na, ta, nb, tb); plk_look_second((AVP_LDS PlShared*)&s, pid, maxNodes, na, ta, nb, tb); }
                }
            }
            if (trace && tid == 0 && !helper && n_pops < max_trace) {
                double* t = trace + ((size_t)pid * max_trace + n_pops) * PL_TRACE_W;
                t[0] = (double)cn.index; t[1] = (double)cn.parent_index; t[2] = (double)avp_pos_to_index(m, cn.x, cn.y);
                t[3] = cn.x; t[4] = cn.y; t[5] = cn.th; t[6] = cn.g; t[7] = cn.h; t[8] = cn.f;
                t[9] = cn.forward; t[10] = cn.steer_i < 0 ? NAN : s.k_steer[cn.steer_i];
            }
            n_pops++;

            // ---- children poses (expand_node :134-151) + try_reach_goal radius test (:308-312) ----------
            const long long t_d = PH_NOW();
            const long long t_pop0 = t_d;
            const bool one_pass = nchild + 1 <= PL_RSQ;          // shot + all children fit one RS pass
            const double ddx = cn.x - s.goal[0], ddy = cn.y - s.goal[1];
            const bool in_radius = avp_within_radius(ddx, ddy, p.flag_radius);   // np.sqrt(dx ** 2 + dy ** 2) < flag_radius, ** = libm pow (hybrid_a_star.py:308)
            bool can_fast = false;
            long long t_f = 0;
            int32_t pre_cand = -1;
            double pre_second = INFINITY, sec_thr = INFINITY, sec_thr2 = INFINITY;
            int32_t sec_cand = -1, sec_cand2 = -1;
            bool late = false;                 // (PL_LOOK_LATE) the long way was left for the node's record, which landed meanwhile
            if (!use_rec) {
            if (tid == 0) { s.in_radius = in_radius ? 1 : 0; s.collision = 0; s.rs_first_coll = 0x7fffffff; s.rs_npts = 0; s.rs_status = 0; s.rs.n = 0; s.chk_arrived = 0; s.shot_ready = hC ? 2 : 0; }
            // a helper runs its half only: queries = the children (hC), the shot (hS), both (an owner)
            const int qoff = hC ? 0 : 1, nq_all = hC ? nchild : (hS ? 1 : nchild + 1);
            if (tid < nchild && !hS) {
                PlChild& c = s.child[tid];
                const int si = tid % p.n_steer;
                const bool fwd = tid < p.n_steer;       // i < next_index / 2
                const double travel = fwd ? p.travel_dt : -p.travel_dt;
                const double th_ = avp_pi_2_pi(cn.th + s.k_dth_dt[si]);
                c.th = th_;
                double sth, cth;
                avp_sincos(th_, sth, cth);
                c.x = cn.x + travel * cth;
                c.y = cn.y + travel * sth;
                c.oob = (c.x > m.b1 || c.x < m.b0 || c.y > m.b3 || c.y < m.b2) ? 1 : 0;
                c.found = helper ? -1 : pl_hash_find(w, dims.hashCap, c.x, c.y, c.th);     // (a helper has no node table)
                c.found_state = c.found >= 0 ? w.nodes[c.found].state : 0;
                c.id = avp_pos_to_index(m, c.x, c.y);
                c.first_coll = 0x7fffffff;
                c.rs_err = 0;
                c.L = 0;
                if (one_pass) s.frame[tid + qoff] = rs_frame(c.x, c.y, c.th, s.goal[0], s.goal[1], s.goal[2], p.maxc);
            } else if (one_pass && tid == PL_THREADS - 2 && !hC) s.frame[0] = rs_frame(cn.x, cn.y, cn.th, s.goal[0], s.goal[1], s.goal[2], p.maxc);
            else if (one_pass && tid == PL_THREADS - 1 && s.sched_cnt != nq_all) pl_rs_build_schedule(s, nq_all);
            if (PROFILE && tid == 0) s.phase[PH_CHILD_W0] += clock64() - t_d;
            if constexpr (LOOK) if (!helper && look.on && wave == 0) {
                double sec_f;
                const int32_t top = pl_look_prefetch_node(w, s, sec_f);
                plk_look_prefetch((AVP_LDS PlShared*)&s, pid, maxNodes, top, sec_f);                   // (wave 0 idles until the sub-step checks are done)
                if (PL_LOOK_LATE) plk_look_late((AVP_LDS PlShared*)&s, pid, 0);                           // ... and looks for this node's pending record again
            }
            // Meanwhile waves 1 .. nwave-2 check the sub-step poses of every child (:185-204): they depend on the
            // popped node only, not on the children stage that keeps wave 0 (and the last wave) busy.
            const int nsubs = nchild * p.n_sub;
            {
                const int nw = min(nwave - 2, PL_SUB_WAVES);
                const int per = max(1, min(PL_WPOSE, (nsubs + nw - 1) / nw));      // spread the poses evenly over the waves
                if (wave >= 1 && wave <= nw && !hS) {
                    for (int base = (wave - 1) * per; base < nsubs; base += nw * per) {
                        const int cnt = min(per, nsubs - base);
                        pl_check_wave<STAGE>(s.env, s, cnt, [&](int k, double& x, double& y, double& th, double& cs, double& sn) {
                            const int t = base + k;
                            const int ci = s.sub_child[t], j = s.sub_j[t], si = s.sub_steer[t];
                            const double tj = s.k_travel_ddt1 * (double)(j + 1);      // speed * ddt * (i + 1), :188
                            const double td = ci < p.n_steer ? tj : -tj;
                            th = avp_pi_2_pi(cn.th + s.k_dth_ddt1[si] * (double)(j + 1));      // ... / lw * ddt * (i + 1), :189-191
                            avp_sincos(th, sn, cs);
                            x = cn.x + td * cs;
                            y = cn.y + td * sn;
                        }, &s.chk_hit[base]);
                    }
                }
            }
            PH_MARK(0);
            __syncthreads();
#if PL_LOOK_LATE
            if constexpr (LOOK) if (!helper && s.late_rec) { late = true; goto pl_late_record; }      // the record has landed: leave the long way here
#endif
            if (!hS) for (int t = tid; t < nsubs; t += PL_THREADS)
                if (s.chk_hit[t]) { const int ci = t / p.n_sub; atomicMin(&s.child[ci].first_coll, t - ci * p.n_sub); }
            const long long t_e = PH_NOW();
            if (PROFILE && tid == 0) s.phase[PH_CHILD] += t_e - t_d;

            // read here, a full barrier before the speculative resolution on wave 0 starts to move s.nnodes
            can_fast = !helper && s.closed_nonempty && (s.nnodes + nchild <= maxNodes);

            // ---- Reeds-Shepp words: query 0 = the shot from the popped node (:326-332), 1.. = children (:286-294)
            {
                const int nq = (hS && !in_radius) ? 0 : nq_all;      // (a shot half outside the radius has no shot)
                for (int base = 0; base < nq; base += PL_RSQ) {
                    const int cnt = min(PL_RSQ, nq - base);
                    pl_rs_words(s, p, cnt, [&](int q, double& x, double& y, double& th) {
                        const int g = base + q;
                        if (g < qoff) { x = cn.x; y = cn.y; th = cn.th; }
                        else { x = s.child[g - qoff].x; y = s.child[g - qoff].y; th = s.child[g - qoff].th; }
                    }, one_pass);
#if PL_LOOK_LATE
                    if constexpr (LOOK) if (!helper && look.on && wave == 0 && base == 0) plk_look_late((AVP_LDS PlShared*)&s, pid, 1);      // (wave 0 is done with its words ~5 k cycles before the others)
#endif
                    if (base == 0) PH_MARK(1);
                    __syncthreads();
#if PL_LOOK_LATE
                    if constexpr (LOOK) if (!helper && base == 0 && s.late_rec2) { late = true; goto pl_late_record; }
#endif
                    if (PROFILE && tid == 0) s.phase[PH_RS_WORDS] += clock64() - t_e;
                    // set_path and arg-min, a whole query per wave (20 lanes run its type groups, then the wave folds):
                    // no cross-wave hand-over. Wave 0 owns the shot (query 0 of the first pass) and goes straight on to
                    // the sampler's index bookkeeping; the last wave, after its children, walks the chain of segment origins as
                    // soon as wave 0 has published the path -- two serial jobs hidden behind the children's queries.
                    const int q_first = base == 0 ? qoff : 0;           // first query of this pass that is a child
                    if (wave == 0) {
                        if (base == 0 && !hC) {
                            const long long t_a0 = PH_NOW();
                            if (lane < 20) pl_rs_accept_group(s, p, 0, lane);
                            wave_sync();
                            if (PROFILE && lane == 0) PH_X(3, t_a0);
                            RsPath rp;
                            const int st = pl_rs_fold_wave(s, 0, rp);
                            if (PROFILE && lane == 0) PH_X(4, t_a0);
                            if (lane == 0) {
                                s.rs_status = in_radius ? st : 0;
                                const bool shot = in_radius && !st;
                                if (!st) s.rs = rp;
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                *(volatile int32_t*)&s.shot_ready = shot ? 1 : 2;
                                if (shot) { s.n_rs += 1; const int bs = pl_rs_sample_book(s, p); if (bs) s.rs_status = bs; }
                                if (PROFILE) PH_X(5, t_a0);
                            }
                        }
                    } else {
                        // up to two queries per wave at a time: lanes 0..19 / 32..51 run the type groups of one each
                        for (int q = q_first + (wave - 1); q < cnt; q += 2 * (nwave - 1)) {
                            const int q2 = q + (nwave - 1);
                            const int half = lane >> 5, gl = lane & 31;
                            const int qa = half ? q2 : q;
                            if (gl < 20 && qa < cnt) pl_rs_accept_group(s, p, qa, gl);
                            wave_sync();
                            for (int k = 0; k < 2; k++) {
                                const int qq = k ? q2 : q;
                                if (qq >= cnt) break;
                                RsPath rp;
                                const int st = pl_rs_fold_wave(s, qq, rp);
                                if (lane == 0) { const int g = base + qq; s.child[g - qoff].rs_err = (int8_t)st; s.child[g - qoff].L = st ? 0.0 : rp.L / p.maxc; }
                            }
                        }
                        if (wave == nwave - 1 && base == 0 && !hC) {
                            if (lane == 0) while (*(volatile int32_t*)&s.shot_ready == 0) __builtin_amdgcn_s_sleep(1);
                            wave_sync();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            const long long t_o0 = PH_NOW();
                            if (*(volatile int32_t*)&s.shot_ready == 1) pl_rs_sample_origins(s, p);      // (wave-uniform)
                            if (PROFILE && lane == 0) PH_X(6, t_o0);
                        }
                    }
                    if (base == 0) PH_MARK(2);
                    __syncthreads();
                }
            }
            const long long t_f0 = PH_NOW();
            if (PROFILE && tid == 0) s.phase[PH_CHILD_RS] += t_f0 - t_e;
            if (!helper && in_radius && s.rs_status) { if (tid == 0) s.status = (s.rs_status == 4 || s.rs_status == 5) ? 5 : 3; __syncthreads(); break; }
            const long long t_g = t_f0;
            const bool do_shot = in_radius && !s.rs_status && !hC; // (a helper reports a failed solve in its record instead)
            // The outcome of the shot is not an input of the child resolution, so when the resolution can take its
            // fast path, wave 0 runs it SPECULATIVELY while the other waves sample and check the shot; if the shot
            // then turns out collision free (the search ends at this pop, before expand_node), the counters are
            // rolled back -- the arena / heap / hash side effects touch no node of the final path's parent chain.
            // The samples are produced by the wave that checks them (no hand-over through memory, no barrier), in
            // path order, and a wave stops as soon as a collision is known before its next chunk: the reference
            // stops at the first colliding sample (:335-345), typically among the first few.
            if (do_shot) {
                const int total = s.smp_hi + 1;                 // entries past smp_hi are unset = popped by the trim
                const int w0 = can_fast ? 1 : 0, nw = min(nwave - w0, PL_SHOT_WAVES);
                if (wave == 0) {
                    if (lane == 0) {
                        s.fast = can_fast ? 1 : 0;
                        s.snap[0] = s.nnodes; s.snap[1] = s.n_checks; s.snap[2] = s.n_rs; s.snap[3] = s.nclosed; s.snap[4] = s.nheap;
                    }
                    wave_sync();
                    if (can_fast) {
                        pl_resolve_fast_wave<PROFILE>(m, p, w, s, dims, cn, nchild, n_pops < max_pops);
                        if constexpr (LOOK) if (look.on) { wave_sync(); if (s.have_next) plk_look_fetch((AVP_LDS PlShared*)&s, pid, maxNodes, s.next_cur, s.nheap, 0, 2); }
                    }
                }
                if (wave >= w0 && wave < w0 + nw) {
                    double cm, sm;
                    avp_sincos(-cn.th, sm, cm);
                    // chunk = samples base, base + stride, ... (cnt of them); hit flags land in the wave's own wchk.hit[]
                    auto do_chunk = [&](int base, int stride, int cnt) {
                        double tx = 0.0, ty = 0.0, tth = 0.0;
                        const int mine = base + lane * stride;
                        if (lane < cnt) pl_rs_sample_world(w, s, p, cn, cm, sm, mine, tx, ty, tth);
                        // lane k holds pose k and stages it for the pass
                        uint32_t* hits = &s.wave_chk().hit[0];
                        pl_check_wave<STAGE>(s.env, s, cnt, [&](int k, double& x, double& y, double& th, double& cs, double& sn) {
                            x = tx; y = ty; th = avp_pi_2_pi(tth); /* :339 */
                            avp_sincos(th, sn, cs);
                        }, hits);
                        if (lane < cnt && hits[lane]) atomicMin(&s.rs_first_coll, mine);
                        wave_sync();
                    };
                    // Round 0: PL_WPOSE0 samples per wave, interleaved (wave c takes samples c, c + nw, ...: samples get
                    // dearer along the path, towards the obstacles around the goal). On the bench workload 88 % of the
                    // colliding shots hit within their first 21 samples (all within 32); the later chunks then never
                    // run. The checking waves meet at a software barrier so that the decision sees every round-0 result.
                    const int head = min(total, nw * PL_WPOSE0);
                    const long long t_s0 = PH_NOW();
                    {
                        const int first = wave - w0;
                        if (first < head) do_chunk(first, nw, (head - first + nw - 1) / nw);
                    }
                    if (PROFILE && tid == 64) s.phase[PH_SHOT_ROUND0] += clock64() - t_s0;
                    if (head < total) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) {
                            atomicAdd(&s.chk_arrived, 1u);
                            while (*(volatile uint32_t*)&s.chk_arrived < (uint32_t)nw) __builtin_amdgcn_s_sleep(1);
                        }
                        wave_sync();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const int rest = total - head;
                        const int per = max(1, min(PL_WPOSE, (rest + nw - 1) / nw));
                        for (int base = head + (wave - w0) * per; base < total; base += nw * per) {
                            if (*(volatile int32_t*)&s.rs_first_coll < base) break;
                            do_chunk(base, 1, min(per, total - base));
                        }
                    }
                    if (PROFILE && tid == 64) s.phase[PH_SHOT_REST] += clock64() - t_s0;
                }
                pl_lds_barrier();                                   // (the resolution's global stores need not have landed)
                if (tid == 0) {
                    // a hit at or past the trimmed length belongs to a popped entry (rs_curve.py:588-592)
                    if (s.rs_first_coll != 0x7fffffff && s.rs_first_coll >= s.rs_npts) s.rs_first_coll = 0x7fffffff;
                    if (helper) { s.collision = s.rs_first_coll != 0x7fffffff; }
                    else if (s.rs_first_coll == 0x7fffffff) {
                        // success: the reference returns before expand_node -- undo the speculative bookkeeping
                        s.nnodes = (int32_t)s.snap[0]; s.n_checks = s.snap[1]; s.n_rs = s.snap[2]; s.nclosed = (int32_t)s.snap[3]; s.nheap = (int32_t)s.snap[4];
                        s.n_checks += s.rs_npts;
                        s.done = 1;
                    } else { s.collision = 1; s.n_checks += s.rs_first_coll + 1; }
                }
            }
            PH_MARK(3);
            pl_lds_barrier();
            t_f = PH_NOW();
            if (PROFILE && tid == 0) s.phase[PH_SHOT_CHECK] += t_f - t_g;
            if constexpr (LOOK) if (helper) {
                // publish this half of the record: payload, then the half's ready bit
                if (wave == 0) {
                    const unsigned long long j0 = s.job[0];
                    const size_t ri = pl_look_ent(look, PL_JOB_TAG(j0));      // (the entry is this job's until both halves are there: pl_look_claim)
                    unsigned long long* rp = look.recs + ri * PL_REC_WORDS;
                    if (hC && lane < nchild) {
                        const PlChild& c = s.child[lane];
                        pl_st64(rp + lane, pl_bits(c.x)); pl_st64(rp + 16 + lane, pl_bits(c.y)); pl_st64(rp + 32 + lane, pl_bits(c.th));
                        pl_st64(rp + 48 + lane, pl_bits(c.L));
                        pl_st64(rp + 64 + lane, (unsigned long long)(uint32_t)c.first_coll | ((unsigned long long)(uint8_t)c.rs_err << 32));
                    }
                    if (lane == 63) {
                        // (the key words are written by both halves, with the same values)
                        pl_st64(rp + 80, s.job[1]); pl_st64(rp + 81, s.job[2]); pl_st64(rp + 82, s.job[3]); pl_st64(rp + 83, pl_look_key3((int64_t)s.job[5], s.goal[2]));
                        pl_st64(rp + 86, pl_bits(s.goal[0])); pl_st64(rp + 87, pl_bits(s.goal[1]));
                        if (hS) {
                            pl_st64(rp + 84, (unsigned long long)(in_radius ? 1 : 0) | ((unsigned long long)(s.collision ? 1 : 0) << 1) | ((unsigned long long)(uint8_t)s.rs_status << 8));
                            pl_st64(rp + 85, (unsigned long long)(uint32_t)s.rs_first_coll | ((unsigned long long)(uint32_t)s.rs_npts << 32));
                        }
                    }
                    PL_LOOK_DRAIN();
                    wave_sync();
                    if (lane == 0) { PL_FLAG_OR64(look.state + ri, hS ? 4ull : 2ull); atomicAdd(look.ctrl + (hS ? 88 : 24), 1ull); }      // ([24], [88]: halves made)
                    if (PL_LOOK_FAULT > 0 && lane == 0 && PL_JOB_TAG(j0) % (PL_LOOK_FAULT > 0 ? PL_LOOK_FAULT : 1) == 0) pl_st64(rp + 81, s.job[2] ^ 1ull);   // (fault injection: the y key, last bit)
                }
                else if (PL_LOOK_CHAIN > 0 && wave == 1 && hC) {
                    // a predicted dive goes on below this node: post its cheapest child under the job's threshold (beside wave 0's publishing)
                    const unsigned long long j0 = s.job[0];
                    if (PL_JOB_DEPTH(j0) > 0)
                        pl_look_chain(look, s, (const uint32_t*)(workspace + (size_t)PL_JOB_BLOCK(j0) * dims.bytes), lane, cn.th, cn.forward, pl_unbits(s.job[4]),
                                      PL_JOB_DEPTH(j0), PL_JOB_BLOCK(j0), (int64_t)s.job[5]);
                }
                continue;
            }
            if (s.status != 0 || s.done) break;
            }
#if PL_LOOK_LATE
            pl_late_record:
#endif
            if (use_rec || late) {
                // ---- the expansion record of a helper stands in for everything up to the resolution ---------------
                const unsigned long long* rec = s.recb[s.rec_cur];
                if constexpr (LOOK) if (wave == 1) pre_cand = pl_look_prefetch_node(w, s, pre_second);      // (its record is fetched beside the resolution)
                if constexpr (LOOK) if (PL_LOOK_PREDICT && PL_LOOK_SECOND && wave == 3 && !late) pl_look_second_nodes(w, s, sec_cand, sec_thr, sec_cand2, sec_thr2);      // (the long way has looked already)
                const bool nf = s.nf_node == s.cur;              // this node's children were looked up by the fetching wave
                if (wave == 2 && lane < nchild) {
                    // the children's heuristic distances, read ahead of the classification (nothing moves the field meanwhile)
                    const int64_t id = avp_pos_to_index(m, pl_unbits(rec[lane]), pl_unbits(rec[16 + lane]));
                    s.child[lane].pre_d = pl_id_in_range(m, id) ? w.dist[id] : PL_UNSEEN;
                }
                if (tid == 0) {
                    const int32_t fc = (int32_t)(uint32_t)(rec[85] & 0xffffffffull);
                    s.in_radius = in_radius ? 1 : 0; s.collision = in_radius ? 1 : 0; s.rs_first_coll = in_radius ? fc : 0x7fffffff;
                    s.rs_npts = (int32_t)(uint32_t)(rec[85] >> 32); s.rs_status = 0; s.rs.n = 0; s.chk_arrived = 0; s.shot_ready = 2; s.fetch_go = 0; s.wr_go = 0; s.wr_done = 0;
                    if (in_radius) { s.n_rs += 1; s.n_checks += fc + 1; }
                }
                if (tid < nchild) {
                    PlChild& c = s.child[tid];
                    c.x = pl_unbits(rec[tid]); c.y = pl_unbits(rec[16 + tid]); c.th = pl_unbits(rec[32 + tid]);
                    c.oob = (c.x > m.b1 || c.x < m.b0 || c.y > m.b3 || c.y < m.b2) ? 1 : 0;
                    if (nf) { c.found = s.nf_found[tid]; c.found_state = s.nf_state[tid]; }
                    else {
                        c.found = pl_hash_find(w, dims.hashCap, c.x, c.y, c.th);
                        c.found_state = c.found >= 0 ? w.nodes[c.found].state : 0;
                    }
                    c.id = avp_pos_to_index(m, c.x, c.y);
                    c.first_coll = (int32_t)(uint32_t)(rec[64 + tid] & 0xffffffffull);
                    c.rs_err = (int8_t)(rec[64 + tid] >> 32);
                    c.L = pl_unbits(rec[48 + tid]);
                }
                PH_MARK(0);
                __syncthreads();
                can_fast = s.closed_nonempty && (s.nnodes + nchild <= maxNodes);
                t_f = PH_NOW();
                if constexpr (LOOK) if (!can_fast && wave == nwave - 1) plk_look_post((AVP_LDS PlShared*)&s, pid, maxNodes, look_node, look_key, PL_LOOK_KIDS_ON_HIT, cn.x, cn.y, cn.th, cn.forward, cn.steer_i);
            }

            // ---- sequential resolution in child order (:153-232). Thread 0 runs alone; when a heuristic
            // query misses the closed frontier the whole workgroup extends the sweep, then thread 0 resumes.
            const bool rec = use_rec || late;                       // resolved from a record (popped with it, or adopted on the way)
            const bool tried = !rec && in_radius && can_fast;       // the speculative attempt above
            if (tid == 0) { s.next_child = 0; s.have_d = 0; s.need_sweep = 0; if (!tried) s.fast = can_fast ? 1 : 0; }
            if (!can_fast && tid < nchild) s.child[tid].pre_d = pl_id_in_range(m, s.child[tid].id) ? w.dist[s.child[tid].id] : PL_UNSEEN;
            __syncthreads();
            if (!tried && can_fast) {
                if (wave == 0) {
                    pl_resolve_fast_wave<PROFILE>(m, p, w, s, dims, cn, nchild, n_pops < max_pops, LOOK && rec);
                    if constexpr (LOOK) if (look.on) {
                        wave_sync();
                        if (rec) {       // (resolution left early / nothing popped ahead: release the writer and the fetcher)
                            if (lane == 0 && *(volatile int32_t*)&s.wr_go == 0) *(volatile int32_t*)&s.wr_go = 2;
                            if (lane == 0 && *(volatile int32_t*)&s.fetch_go == 0) *(volatile int32_t*)&s.fetch_go = 2;
                        }
                        else if (s.have_next) plk_look_fetch((AVP_LDS PlShared*)&s, pid, maxNodes, s.next_cur, s.nheap, 0, 2);
                    }
                } else if (LOOK && rec && wave == 1) {
                    // record pop: the other waves are idle, so this one fetches the next node's record as soon as wave 0
                    // knows that node (before it sifts the heap), and does the bounded wait for a pending record
                    if constexpr (LOOK) {
                        plk_look_prefetch((AVP_LDS PlShared*)&s, pid, maxNodes, pre_cand, pre_second);
                        if (lane == 0) while (*(volatile int32_t*)&s.fetch_go == 0) __builtin_amdgcn_s_sleep(2);
                        wave_sync();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        if (*(volatile int32_t*)&s.fetch_go == 1) plk_look_fetch((AVP_LDS PlShared*)&s, pid, maxNodes, s.next_cur, s.fetch_nheap, nchild, 3);
                    }
                } else if (LOOK && rec && wave == 2) {
                    pl_resolve_writer_wave(p, w, s, dims, cn, nchild);
                } else if (LOOK && rec && wave == 3) {
                    if constexpr (LOOK) if (PL_LOOK_PREDICT && PL_LOOK_SECOND) plk_look_second((AVP_LDS PlShared*)&s, pid, maxNodes, sec_cand, sec_thr, sec_cand2, sec_thr2);      // (an idle wave of a record pop)
                } else if (LOOK && rec && wave == nwave - 1) {
                    if constexpr (LOOK) plk_look_post((AVP_LDS PlShared*)&s, pid, maxNodes, look_node, look_key, PL_LOOK_KIDS_ON_HIT, cn.x, cn.y, cn.th, cn.forward, cn.steer_i);    // (beside the resolution on wave 0)
                }
                if (LOOK && rec) PH_MARK(3);
                __syncthreads();
            }
            if (s.fast) {
                // resolved by wave 0 (speculatively above, or just now)
            } else
            for (;;) {
                if (tid == 0) {
                    s.need_sweep = 0;
                    int i = s.next_child;
                    for (; i < nchild && s.status == 0; i++) {
                        const PlChild c = s.child[i];
                        const int si = i % p.n_steer;
                        const int is_forward = i < p.n_steer ? 1 : 0;
                        const bool found_closed = c.found >= 0 && c.found_state == 2;
                        if (s.closed_nonempty && (found_closed || c.oob)) continue;          // :155-165
                        const bool found_open = c.found >= 0 && c.found_state == 1;
                        if (!found_open && c.first_coll != 0x7fffffff) {
                            s.n_checks += c.first_coll + 1;
                            if (s.nnodes >= maxNodes) { s.status = 5; break; }
                            const int32_t pos = s.nnodes++;
                            PlNode& nd = w.nodes[pos];
                            nd.x = c.x; nd.y = c.y; nd.th = c.th; nd.g = 0; nd.h = 0; nd.f = 0;
                            nd.index = (int32_t)(s.global_index + i + 1); nd.parent_index = cn.index; nd.parent_pos = s.cur;
                            nd.forward = (int8_t)is_forward; nd.steer_i = (int8_t)si; nd.state = 2; nd.heap_pos = -1;
                            pl_hash_put(w, dims.hashCap, pos);
                            s.nclosed++; s.closed_nonempty = 1;
                            continue;
                        }
                        // heuristic query (hit: answered here; miss: hand over to the workgroup)
                        uint32_t hd;
                        if (s.have_d) { hd = s.hq_d; s.have_d = 0; }
                        else if (!pl_hquery_hit(m, s, c.id, c.pre_d, hd)) { s.pending_id = c.id; s.need_sweep = 1; break; }
                        if (hd == PL_UNSEEN) { if (!found_open) s.n_checks += p.n_sub; s.status = s.qover ? 5 : 2; break; }
                        s.n_rs += 1;
                        if (c.rs_err) { s.status = c.rs_err == 4 ? 5 : 3; break; }
                        const double hv1 = (double)hd / 100, hv2 = c.L;
                        const double hval = hv2 > hv1 ? hv2 : hv1;
                        if (!found_open) {
                            s.n_checks += p.n_sub;
                            if (s.nnodes >= maxNodes) { s.status = 5; break; }
                            const double g = pl_node_cost(p, is_forward, c.th, cn.th, cn.forward);
                            const int32_t pos = s.nnodes++;
                            PlNode& nd = w.nodes[pos];
                            nd.x = c.x; nd.y = c.y; nd.th = c.th; nd.g = g; nd.h = hval; nd.f = g + hval;
                            nd.index = (int32_t)(s.global_index + i + 1); nd.parent_index = cn.index; nd.parent_pos = s.cur;
                            nd.forward = (int8_t)is_forward; nd.steer_i = (int8_t)si; nd.state = 1;
                            pl_heap_push(w, s, (uint32_t)pos, g + hval);
                            pl_hash_put(w, dims.hashCap, pos);
                        } else {
                            PlNode& ch = w.nodes[c.found];
                            const double new_g = pl_node_cost(p, ch.forward, ch.th, cn.th, cn.forward);
                            const double new_f = hval + new_g;
                            if (new_f < ch.f) {
                                ch.f = new_f; ch.g = new_g; ch.h = hval;
                                pl_heap_set_key(w, s, PlShared::HEAP_POS ? ch.heap_pos : pl_heap_find(w, s, s.nheap, (uint32_t)c.found), new_f);
                                ch.parent_index = cn.index; ch.parent_pos = s.cur;
                                ch.forward = (int8_t)is_forward; ch.steer_i = (int8_t)si;
                            }
                        }
                    }
                    s.next_child = i;
                }
                __syncthreads();
                if (!s.need_sweep) break;
                plk_sweep_extend<PROFILE>((AVP_LDS PlShared*)&s);
                if (tid == 0) s.have_d = 1;
                if (tid < nchild) s.child[tid].pre_d = pl_id_in_range(m, s.child[tid].id) ? w.dist[s.child[tid].id] : PL_UNSEEN;
                __syncthreads();
            }
            if (tid == 0) {
                if (s.status == 0) {
                    w.nodes[s.cur].state = 2;
                    s.nclosed++; s.closed_nonempty = 1;
                    s.global_index += nchild;
                }
                if (PROFILE) s.phase[PH_RESOLVE] += clock64() - t_f;
            }
            PH_MARK(4);
            __syncthreads();
        }
        __syncthreads();
        const long long t_fin = PH_NOW();
        if (s.status == 1 && s.cur >= 0 && s.in_radius && s.rs.n > 0 && s.rs_status == 0 && s.collision) {
            // the reference hands back the last (colliding) shot when the open list runs empty
            // (path_planner.py:100-108): the early exit above may have left samples unproduced
            const PlNode cl = w.nodes[s.cur];
            double cm, sm;
            avp_sincos(-cl.th, sm, cm);
            for (int i = tid; i <= s.smp_hi; i += PL_THREADS) { double a, b, c; pl_rs_sample_world(w, s, p, cl, cm, sm, i, a, b, c); }
            __syncthreads();
        }

        if constexpr (LOOK) if (helper) break;             // every problem is finished (or the ring ran dry for good)

        // ---- finish_path (:351-389) + assembly (path_planner.py:100-108) -----------------------------
        if (tid == 0) {
            plk_write_result<PROFILE>((AVP_LDS PlShared*)&s, pid, n_pops, (int32_t)blockIdx.x, t_fin);
            if constexpr (LOOK) { atomicAdd(look.ctrl + 32, 1ull); if (s.n_hits) { atomicAdd(look.ctrl + 8, (unsigned long long)s.n_hits); s.n_hits = 0; } for (int k = 0; k < 4; k++) if (s.n_sec[k]) { atomicAdd(look.ctrl + 72 + k, (unsigned long long)s.n_sec[k]); s.n_sec[k] = 0; } if (s.n_late) { atomicAdd(look.ctrl + 76, (unsigned long long)s.n_late); s.n_late = 0; } if (s.n_pred) { atomicAdd(look.ctrl + 77, (unsigned long long)s.n_pred); s.n_pred = 0; } if (s.n_busy) { atomicAdd(look.ctrl + 78, (unsigned long long)s.n_busy); s.n_busy = 0; } if (s.n_torn) { atomicAdd(look.ctrl + 79, (unsigned long long)s.n_torn); s.n_torn = 0; } for (int k = 0; k < 6; k++) if (s.n_miss[k]) { atomicAdd(look.ctrl + 96 + k, (unsigned long long)s.n_miss[k]); s.n_miss[k] = 0; } }   // ([8]: records used, [76]: of which adopted late -- diagnostics)
        }
        __syncthreads();
    }
}
