/*
 * avp.h -- C-ABI of libavp_hip.so, the MI355X-native hybrid-A* hot path.
 *
 * The reference (wenqing-2021/AutomatedValetParking) is pure Python and has no FFI; the boundary
 * is its class API. Each entry point below replaces the reference
 * interface cited next to it; automatedvaletparking_amd/ binds them with ctypes and re-exposes
 * the reference's own classes (PathPlanner, distance_checker, two_circle_checker, rs_curve.PATH,
 * costmap.Map/Vehicle/Case). INTEGRATION.md shows the binding a maintainer of the reference adds.
 *
 * Conventions: every function returns int32 status (0 = OK, < 0 = error, text via avp_last_error);
 * no exception crosses; the caller owns every buffer; pointers documented "device" are raw HIP
 * device pointers (e.g. torch.Tensor.data_ptr()); launches are asynchronous on the handle's stream
 * (avp_map_set_stream, default = the NULL stream) unless noted. A handle is not thread-safe;
 * distinct handles are independent. All arithmetic is IEEE fp64, ids/distances are integers.
 */
#ifndef AVP_H
#define AVP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVP_VERSION 110
#define AVP_MAX_STEER 32                 /* steering angles: 2 x 32 = 64 children per expansion, one lane of a wave each */
#define AVP_MAX_NODES_PER_AXIS 32767      /* nodes per map axis (3.2 km at 0.1 m; and at most 2^30 cells): beyond 8 191 the lane-per-pose kernel forms
                                            run; the bound keeps heuristic distances (<= 14 per cell) inside the 20 bits of the alias-owner key */
#define AVP_MAX_SUBS 512                 /* children x sub-steps checked per expansion (default: 10 x 3)             */
#define AVP_RS_MAXSEG 5

/* status codes of the library calls */
enum {
    AVP_OK = 0,
    AVP_ERR_ARG = -1,       /* bad argument                                   */
    AVP_ERR_HIP = -2,       /* HIP runtime error (text in avp_last_error)     */
    AVP_ERR_NOGPU = -3,     /* no usable gfx950 device                         */
    AVP_ERR_CAPACITY = -4   /* a caller-provided buffer is too small           */
};

/* per-problem status written by avp_plan_batch (the reference signals these by exceptions/hangs) */
enum {
    AVP_PLAN_OK = 0,            /* goal reached by a collision-free Reeds-Shepp shot            */
    AVP_PLAN_NO_PATH = 1,       /* open list exhausted: AttributeError at path_planner.py:104   */
    AVP_PLAN_H_UNREACHABLE = 2, /* heuristic query unreachable: the reference blocks forever in
                                   PriorityQueue.get(), compute_h.py:77                          */
    AVP_PLAN_RS_ERROR = 3,      /* start == goal: AssertionError rs_curve.py:153                */
    AVP_PLAN_ITER_LIMIT = 4,    /* params.max_pops reached (the reference has no cap)           */
    AVP_PLAN_CAPACITY = 5,      /* node arena / path buffer / sweep queue exhausted             */
    AVP_PLAN_LATTICE = 6,       /* goal outside the map or goal-anchored lattice not regular     */
    AVP_PLAN_BAD_POSE = 7       /* a start / goal coordinate that is not finite, or a heading that is not finite or beyond
                                   1e6 rad. NARROWER than the reference's domain: rs_curve.pi_2_pi's subtract-2-pi loop
                                   (rs_curve.py:648-655, called from hybrid_a_star.py:72-124) does return on any finite
                                   heading below ~1e16 rad (it stops terminating where theta - 2 pi == theta) -- after
                                   |theta| / 2 pi trips: more than 1.6e5 beyond 1e6 rad, 1.6e8 at 1e9. The device runs that
                                   very loop, bit for bit, for |theta| <= 1e6 and refuses the rest before its first loop
                                   (tests/test_gpu_edge_inputs.py: 1e6 and its two neighbours against the oracle). */
};

/*
 * Planner parameters: the hot-path keys of config/config.yaml + Vehicle constants
 * (map/costmap.py:52-63), with every per-config constant that the reference computes through
 * numpy/libm evaluated ON THE HOST in the reference's expression order and passed in, so that the
 * device never has to reproduce numpy's tan / pow (hard part 1).
 */
typedef struct avp_params {
    /* vehicle */
    double lw, lf, lr, lb, max_v, max_steer, min_radius;
    /* inflated footprint in the vehicle frame: rear/front x, right/left y (map/costmap.py:97-101) */
    double fp_xr, fp_xf, fp_yr, fp_yl;
    /* two-circle model (collision_check.py:92-98): radius, front/rear centre offsets */
    double circ_rd, circ_cf, circ_cr;
    /* hybrid A* (hybrid_a_star.py:81-83,145-151,188-194) */
    int32_t n_steer;
    int32_t n_sub;                       /* ceil(dt / trajectory_dt); 2 * n_steer * n_sub <= AVP_MAX_SUBS */
    double steer[AVP_MAX_STEER];         /* np.linspace(-max_steer, max_steer, n)                   */
    double dth_dt[AVP_MAX_STEER];        /* (max_v*np.tan(steer))/lw*dt                             */
    double dth_ddt1[AVP_MAX_STEER];      /* (max_v*np.tan(steer))/lw*ddt: sub-step j turns by dth_ddt1 * (j+1) -- the reference's
                                            left-to-right product (hybrid_a_star.py:189-191), its last factor applied on the
                                            device (an IEEE multiplication by an exact small integer), so n_sub is free        */
    double travel_dt;                    /* max_v*dt (sign applied per gear)                        */
    double travel_ddt1;                  /* max_v*ddt: sub-step j travels travel_ddt1 * (j+1) (:188) */
    double flag_radius;
    double cost_gear, cost_heading, cost_scale;
    double maxc;                         /* 1 / min_radius                                          */
    int32_t extended_num;
    int32_t checker_kind;                /* 0 = distance_checker, 1 = two_circle_checker            */
    int64_t max_pops;                    /* per-problem pop cap, 0 = library default                */
} avp_params;

typedef struct avp_map avp_map;          /* opaque: one costmap resident in HBM + its stream        */

int32_t avp_version(void);
int32_t avp_sizeof_params(void);            /* sizeof(avp_params): binding self-check */
int32_t avp_last_error(char* buf, int32_t n);

/*
 * Replaces: Map (map/costmap.py:159-195) as consumed by the checkers and the heuristic.
 * Host pointers; copied to the device once. occ: nx*ny bytes, [ix*ny + iy], 255 = obstacle edge
 * cell (cost_map == 255). xs/ys: map_position. boundary[4] = Map.boundary. obs_ix/obs_iy: the P
 * obstacle cells in np.where(cost_map == 255) order (row-major). device = HIP ordinal.
 */
int32_t avp_map_create(const avp_params* params, const uint8_t* occ, int32_t nx, int32_t ny,
                       const double* xs, const double* ys, const double boundary[4],
                       const int32_t* obs_ix, const int32_t* obs_iy, int32_t P,
                       int32_t device, avp_map** out);
int32_t avp_map_destroy(avp_map* map);
int32_t avp_map_set_stream(avp_map* map, void* hip_stream);
int32_t avp_sync(avp_map* map);

/*
 * Replaces: distance_checker.check / two_circle_checker.check (collision_check.py:144-240, 88-137),
 * one call per pose in the reference. x, y, th, out: device, n elements; out[i] in {0,1}.
 * kind: 0 distance, 1 circle. variant: 0 = production kernel, 1 = straightforward all-points kernel
 * (kept for cross-checking the production kernel on the GPU).
 */
int32_t avp_check_batch(avp_map* map, int32_t kind, const double* x, const double* y, const double* th,
                        int64_t n, uint8_t* out, int32_t variant);

/*
 * Replaces: the per-way-point body of path_opti.compute_collision_H (optimization/path_optimazition.py:
 * 221-409; the same scan is duplicated in optimization/ocp_optimization.py:36-480): corridor bounds used
 * by the downstream QP / OCP stages. x, y, th: device, n way-points; expand_dis = config['expand_dis'].
 * out: device n x 4 = {x_max + x, y_max + y, x - x_min, y - y_min}, i.e. the rows of H_max and H_min.
 */
int32_t avp_corridor_batch(avp_map* map, double expand_dis, const double* x, const double* y, const double* th,
                           int64_t n, double* out);
/* The same with an explicit kernel choice: variant 0 = production kernel (candidates compacted per wave, running
 * minima by LDS atomics), 1 = lane-per-way-point kernel (kept for cross-checking the production kernel on the GPU). */
int32_t avp_corridor_batch_v(avp_map* map, double expand_dis, const double* x, const double* y, const double* th,
                             int64_t n, double* out, int32_t variant);

/*
 * Replaces: rs_curve.calc_optimal_path (path_plan/rs_curve.py:99-134), one call per (start, goal)
 * pair in the reference. All pointers device. q0, q1: n x 3 (x, y, yaw) row-major; maxc = 1 /
 * min turning radius. Outputs per query i: status[i] (0 ok, 1 no candidate word, 2 the reference's
 * "L >= 0.01" assertion fails, 3 more than maxpts samples (npts[i] = needed), 7 a coordinate that is not finite / a
 * heading that is not finite or beyond 1e6 rad: the reference's pi_2_pi loop never ends on an infinite heading), L[i] total length [m], types[i*5+k] in {0 S, 1 L, 2 R, -1 unused}, lens[i*5+k] signed
 * segment lengths [m], npts[i], xyyaw[(i*maxpts+j)*3 + {0,1,2}] world-frame samples every 0.5 m
 * (yaw wrapped by pi_2_pi), dir[i*maxpts+j] in {+1,-1}. maxpts = 0 (xyyaw/dir NULL) skips sampling.
 * map may be NULL (the reference's function is module-level and needs no map): the launch then goes to the
 * calling thread's current device on the NULL stream.
 */
int32_t avp_rs_optimal_batch(avp_map* map, const double* q0, const double* q1, double maxc, int64_t n,
                             int32_t maxpts, int32_t* status, double* L, int8_t* types, double* lens,
                             int32_t* npts, double* xyyaw, int8_t* dir);

/*
 * Per-problem record written by avp_plan_batch (device memory, n entries).
 * Replaces what PathPlanner.a_star_plan returns/raises (path_planner.py:58-110).
 */
typedef struct avp_plan_result {
    int32_t status;          /* AVP_PLAN_*                                                        */
    int32_t n_pops;          /* nodes popped from the open list (expansions + the final pop)      */
    int32_t n_astar;         /* way-points of finish_path (hybrid_a_star.py:351-389)              */
    int32_t n_rs_pts;        /* samples of the last Reeds-Shepp shot (rs_path.x)                   */
    int32_t n_final;         /* way-points of final_path = astar_path + rs samples[1:]; 0 if none  */
    int32_t rs_n;            /* segments of the last RS shot, 0 = no shot (rs_path is None)        */
    int32_t in_radius_last;  /* info['in_radius'] of the last pop                                  */
    int32_t rs_collision;    /* collision flag of the last shot                                    */
    int64_t n_checks;        /* collision checks the reference would have executed                 */
    int64_t n_rs;            /* calc_optimal_path calls the reference would have executed          */
    int64_t n_closed, n_open;/* len(closed_list), len(open_list.queue) at termination              */
    int64_t h_cells;         /* heuristic-field cells expanded by the sweep                        */
    int64_t h_misses;        /* heuristic queries that extended the sweep (Dijkstra.compute_path calls) */
    int64_t global_index;    /* hybrid_a_star.global_index                                         */
    int64_t n_nodes;         /* nodes created                                                      */
    int8_t rs_types[8];      /* 0 S, 1 L, 2 R                                                      */
    double rs_lengths[5];    /* signed segment lengths [m]                                         */
    double rs_L;             /* total RS length [m]                                                */
    double rs_start[3];      /* sample 0 of the last RS shot = pose of the last popped node        */
    int32_t rs_dir0;         /* its direction flag                                                 */
    int32_t slot;            /* the persistent workgroup (0 .. n_slots-1) that ran the problem      */
    int64_t phase_cycles[64];/* diagnostics, avp_plan_batch_profile only (else 0): shader cycles per phase of the
                                problem: 0 init, 1 heap pop, 2-3 wave-0 resolution: classify, node/hash writes, 4 speculative
                                resolution || shot checks (wall), 5 children || sub-steps (wall), 6 RS words .. replay (wall),
                                7 rest of the resolution, 8 of which sweep, 9 finish, 10 wave-0 resolution: heap pushes,
                                11 RS words until their barrier, 12 wave-0 children stage, 13-14 shot checks round 0 / all
                                rounds (wave 1), 15 pop-ahead; 16 + 5 w + k: arrival of wave w at barrier k of a pop (k = 0 children /
                                sub-steps, 1 RS words, 2 set_path .. replay, 3 resolution / shot, 4 end), cycles since the pop began */
} avp_plan_result;

/*
 * Replaces: PathPlanner.a_star_plan (path_plan/path_planner.py:58-110), one call per (start, goal)
 * in the reference, including hybrid_a_star.__init__'s heuristic sweep (hybrid_a_star.py:72-124).
 * One workgroup per problem; n_slots persistent workgroups pull problems from a counter.
 * All pointers device unless noted. starts/goals: n x 3 (x, y, theta). workspace: at least
 * avp_plan_workspace_bytes(map, n_slots, max_nodes) bytes, caller-owned scratch (no initialisation
 * needed). results: n records. paths: n x max_path x 4 doubles (x, y, theta, RS direction flag or 0) =
 * final_path of each problem; the first n_astar rows are astar_path; may be NULL. trace: n x max_trace x 11 doubles, optional
 * pop trace (node index, parent index, grid id, x, y, theta, g, h, f, forward, steering), may be NULL.
 */
int64_t avp_plan_workspace_bytes(avp_map* map, int32_t n_slots, int32_t max_nodes);
int32_t avp_plan_default_slots(avp_map* map);   /* = number of compute units */
int32_t avp_sizeof_plan_result(void);
int32_t avp_plan_batch(avp_map* map, const double* starts, const double* goals, int64_t n, int32_t n_slots,
                       int32_t max_nodes, void* workspace, int64_t workspace_bytes, avp_plan_result* results,
                       double* paths, int32_t max_path, double* trace, int32_t max_trace);

/*
 * Kernel form. The planner exists in four forms with bit-identical results (tests/test_gpu_plan_wave.py). A form gives
 * every problem a GROUP of waves; avp_plan_group(mode) problems share a workgroup (= a compute unit), avp_plan_slots(map,
 * mode) = avp_plan_group(mode) x CUs problems run at once:
 *   mode 1: one workgroup (512 threads) per problem -- the shortest time per problem; right when the batch is no larger
 *           than the chip (BASELINE config[1]: 256 problems on 256 CUs);
 *   mode 2: one wave per problem, avp_plan_group(2) = 16 problems per workgroup -- the most problems in flight; right for
 *           batches much larger than the chip. Problems a group form cannot hold (a Reeds-Shepp shot of more than 192
 *           samples, more than 16 children) are planned by the mode-1 kernel in a later launch of the same call;
 *   mode 3: a pair of waves per problem, avp_plan_group(3) = 8 per workgroup -- a pop takes ~0.7 x the time of mode 2:
 *           right for a few problems per CU (their long searches all run at once: their latency is the launch time);
 *   mode 4: four waves per problem, avp_plan_group(4) = 4 per workgroup (a pop takes ~0.6 x the time of mode 2);
 *   mode 0: avp_plan_batch's choice by problems per CU: mode 1 below 12, mode 4 below 24, mode 3 below 80, mode 2 from there.
 * n_slots counts problem slots in every form and is a multiple of avp_plan_group(mode) (size it from avp_plan_slots, not
 * from a constant); the workspace is avp_plan_workspace_bytes(map, n_slots, max_nodes) as before. avp_plan_pick_mode
 * returns the form mode 0 would use. IMPLICIT TIME SLICING: when n_slots >= n (a slot for every problem), n exceeds
 * avp_plan_slots(map, mode) and the handle's slice length is non-zero (avp_plan_set_slice_pops below; 64 by default), a
 * group-form launch parks long searches and presets every record's status to AVP_PLAN_UNFINISHED first -- call
 * avp_plan_set_slice_pops(map, 0) before the launch to rule it out. Results are the same either way.
 */
int32_t avp_plan_batch_mode(avp_map* map, const double* starts, const double* goals, int64_t n, int32_t n_slots,
                            int32_t max_nodes, void* workspace, int64_t workspace_bytes, avp_plan_result* results,
                            double* paths, int32_t max_path, double* trace, int32_t max_trace, int32_t mode);
int32_t avp_plan_pick_mode(avp_map* map, int64_t n, int32_t mode);
/*
 * The STAGED call (replaces the same reference code as avp_plan_batch: PathPlanner.a_star_plan, path_plan/path_planner.py:58-110,
 * once per problem). Random pose pairs are bimodal: four searches out of five end within a few pops, the rest run for
 * hundreds of pops or to the cap and hold 95 % of a batch's expansions. Stage 1 runs every problem in the mode-2 kernel
 * and hands back, unfinished, every search still running after stage_pops pops; stage 2 plans those from scratch -- in
 * the mode-4 / 3 / 2 kernel, whichever holds their number in one round, when they outnumber the compute units, else (and
 * whatever those hand back) in the mode-1 kernel.
 * A restarted search is the same search: records, traces and paths are those of avp_plan_batch (tests/test_gpu_staged.py).
 * n_slots / workspace: as for mode 2 (avp_plan_slots(map, 2)). first_stage_only != 0: stop after stage 1; the
 * unfinished searches then carry status AVP_PLAN_DEFERRED and nothing else -- for callers that deal them to other
 * devices themselves (automatedvaletparking_amd.distributed). (A search the wave form cannot hold at all -- see mode 2 --
 * carries the same status after stage 1: it, too, is for a later stage.)
 */
#define AVP_PLAN_DEFERRED 100
int32_t avp_plan_batch_staged(avp_map* map, const double* starts, const double* goals, int64_t n, int32_t n_slots,
                              int32_t max_nodes, void* workspace, int64_t workspace_bytes, avp_plan_result* results,
                              double* paths, int32_t max_path, double* trace, int32_t max_trace, int32_t stage_pops,
                              int32_t first_stage_only, const int32_t* order);
int32_t avp_plan_slots(avp_map* map, int32_t mode);
/*
 * Time slicing of the group forms (modes 2 - 4). With n_slots >= n -- a workspace slot for every problem of the batch --
 * and more problems than the form has groups, a search that is still running after `pops` pops is parked (its state goes
 * to its own slot) whenever another problem is waiting for a group, and taken up again in turn by whichever group is
 * free: the long searches of a batch advance side by side, and the launch ends within a slice of the moment the work
 * runs out instead of within a whole long search of it. Which group runs which part of a search never changes a result
 * (tests/test_gpu_plan_wave.py). pops = 0: never park (the behaviour with n_slots < n); pops < 0: the default, 64.
 * A time-sliced launch presets every record's status to AVP_PLAN_UNFINISHED (-1); none is left after a correct launch.
 * The ticket counters of the planner are per map handle: one planner launch per handle at a time (launches on one stream
 * are ordered anyway; callers with several streams order them, as automatedvaletparking_amd._native.DeviceMap does).
 * No reference counterpart (the reference plans one problem at a time, path_plan/path_planner.py:58-110).
 */
#define AVP_PLAN_UNFINISHED (-1)
int32_t avp_plan_set_slice_pops(avp_map* map, int32_t pops);
/* What the last planner call on this handle launched (the library's own decision -- mode 0 may fall back to the workgroup
 * form, a group-form launch slices only under the conditions above): bit 0 = the group-form launch time-sliced its
 * searches, bits 8 .. 15 = the kernel form that ran (1 .. 4; 16 = the staged call). No reference counterpart. */
int32_t avp_plan_last_launch(avp_map* map);
int32_t avp_plan_group(int32_t mode);       /* problems per workgroup of the kernel form: slot counts are multiples of it (mode 1: 1) */

/*
 * Expansion lookahead (mode 1 only). A batch no larger than the chip leaves compute units without a problem of their
 * own as soon as the short searches finish -- BASELINE config[1] keeps 20 % of the CU time busy. With a lookahead
 * workspace the idle workgroups serve the running searches: the owner of a search posts the nodes at the top of its
 * open list (and children it expects to pop next: the three likely ones of a pop that takes the long way, and -- round 6 -- every
 * child of the list's best nodes that beats the rest of the list, predicted from the parent's record; helpers chain such predictions
 * down a dive themselves), helpers compute
 * everything the expansion of such a node needs before the sequential resolution (children poses, sub-step collision
 * checks, the Reeds-Shepp length of every child | the sampled and checked analytic shot -- two half-jobs, pure functions
 * of node pose, goal, map and parameters, evaluated by the same device code), and the owner reads the record when it
 * pops the node. The record store has a FIXED size since round 6 (262 144 records named by pose hash, 0.2 GB with the job rings --
 * rounds 2 - 5: n x max_nodes x 2 832 bytes, which dropped the lookahead at large pop caps). Whether a record exists changes the time of a pop, never its result
 * (tests/test_gpu_lookahead.py: bit-identical records, paths and traces with and without).
 * avp_plan_look_bytes: bytes of the lookahead workspace (the same for every n and max_nodes; 0 = the library would not use one: a batch it
 * plans in a group form -- 11 problems per CU or more --, more than 16 children). A batch of more problems than CUs has no helpers until
 * its tail: a workgroup serves once the problem counter has run dry, and while helpers are scarce the owners post only the nodes they pop
 * next (rounds 2 - 5: at most two problems per CU). The helpers occupy every CU their launch leaves free until its last problem is done:
 * meant for a launch that has the device to itself, not for several concurrent launches on different streams.
 *
 * Problem order. The persistent workgroups (waves) take problems off a counter; when the batch is larger than the chip the
 * longest searches should start first or the last ones run alone (4 096 random problems in index order keep 76 % of
 * the slot time busy). order: device, n int32, a permutation of 0 .. n-1 -- problem order[k] is the k-th to start;
 * results stay at their problem's index. NULL = index order. For a caller that can tell its long searches: on the bench's
 * random pairs the start-goal distance does NOT (119 vs 116 ms, scripts/order_bench.py), so nothing is reordered by default.
 *
 * avp_plan_batch_ex = avp_plan_batch_mode + the lookahead workspace (look_ws NULL = none) + the order (NULL = none).
 */
int64_t avp_plan_look_bytes(avp_map* map, int64_t n, int32_t max_nodes);
/* Diagnostics: 2^log2_entries records in the lookahead's store on this handle from now on (0 = the default, 2^18; else 6 .. 24). With a
 * small store tags collide and entries are taken over all the time; results must not depend on it. No reference counterpart. */
int32_t avp_plan_set_look_entries(avp_map* map, int32_t log2_entries);
int32_t avp_plan_batch_ex(avp_map* map, const double* starts, const double* goals, int64_t n, int32_t n_slots,
                          int32_t max_nodes, void* workspace, int64_t workspace_bytes, avp_plan_result* results,
                          double* paths, int32_t max_path, double* trace, int32_t max_trace, int32_t mode,
                          void* look_ws, int64_t look_bytes, const int32_t* order);

/* The same call through the instrumented instantiation of the kernel: results[i].phase_cycles holds the shader
 * cycles thread 0 spent in each phase of problem i (avp_plan_batch leaves them 0: the s_memtime reads cost ~10 % of
 * the kernel's wave cycles, so the production kernel carries none). Diagnostics only; results are identical. */
int32_t avp_plan_batch_profile(avp_map* map, const double* starts, const double* goals, int64_t n, int32_t n_slots,
                               int32_t max_nodes, void* workspace, int64_t workspace_bytes, avp_plan_result* results,
                               double* paths, int32_t max_path, double* trace, int32_t max_trace);

/*
 * Replaces: Dijkstra.compute_path / the closed-list lookup of calc_node_heuristic
 * (path_plan/compute_h.py:198-214, hybrid_a_star.py:268-283) for ONE goal and a sequence of queries
 * (test / diagnostic entry; avp_plan_batch runs the same sweep inside the planner). goal_xy: host.
 * queries: device nq x 2 positions; force[i] != 0 skips the closed-list lookup (the initial
 * compute_path(x0, y0) of hybrid_a_star.__init__). Outputs (device): out_dist_q[i] distance (-1
 * unreachable), out_miss[i] 1 if the sweep was extended, out_dist/out_flags: avp_hfield_id_capacity()
 * entries, distance (0x7fffffff unseen) and terminator flag per grid id; out_info[8]: status, frontier
 * distance, frontier id, goal id, expanded buckets, cells expanded, sweep extensions, id capacity.
 * workspace: avp_plan_workspace_bytes(map, 1, 16) bytes.
 */
int64_t avp_hfield_id_capacity(avp_map* map);
int32_t avp_hfield_queries(avp_map* map, const double goal_xy[2], const double* queries, const int32_t* force,
                           int32_t nq, void* workspace, int64_t workspace_bytes, int32_t* out_dist_q,
                           int32_t* out_miss, uint32_t* out_dist, uint8_t* out_flags, int64_t* out_info);

/* Obstacle-edge rasteriser: replaces the per-sample part of Map.detect_obstacle_edge (map/costmap.py:236-261;
 * SURVEY.md section 8(f) rank 3). The host keeps np.unique / the centroid-angle argsort / arctan2, cos, sin of the
 * edge angle (map/costmap.py:203-233, numpy SIMD-dispatch sensitive) and passes one row per polygon edge:
 * edges[e] = {p1x, p1y, cos, sin, rotated length, count = floor(length / dx)}. The device evaluates
 * np.linspace(0, length, count), rotates back, adds p1 and marks occ[ix * ny + iy] = 255 for the unique node with
 * X[ix] < px < X[ix] + dx (strict, :253-257), likewise y. All pointers are device pointers; occ must be zeroed by
 * the caller (it may already hold other obstacles' cells); *multi is incremented for every sample that matches
 * more than one node on an axis (the reference raises TypeError there, :260). No avp_map is needed: this runs
 * before avp_map_create. x0 = xs[0], dx = xs[1] - xs[0], y0 = ys[0], dy = ys[1] - ys[0] (host values of the device
 * tables); max_count = largest count in the table; device < 0 = current device; stream = hipStream_t or NULL. */
int32_t avp_rasterize_edges(int32_t device, void* stream, const double* xs, const double* ys, int32_t nx, int32_t ny,
                            double x0, double dx, double y0, double dy, const double* edges, int64_t n_edges,
                            int32_t max_count, uint8_t* occ, int32_t* multi);

/* Batched ingest: the edge tables of n_maps maps rasterised in ONE launch (a many-map workload -- BASELINE config[2]: 20 scenario
 * files -- otherwise pays one launch and one synchronisation per map). Replaces the per-sample part of map/costmap.py:236-261 for a
 * LIST of maps built from TPCAP files (map/costmap.py:134-156); no reference counterpart for the batching itself. Host arrays:
 * node_off[k] (offset, in doubles, of map k's X table inside `nodes`; its Y table follows at + nx[k]), nx, ny, geo[4 k ..] = {x0, dx,
 * y0, dy}, occ_off[k] (offset of map k's nx * ny occupancy bytes inside `occ`). Device arrays: nodes, edges (n_edges x 6, as for
 * avp_rasterize_edges), edge_map (the map of every edge), occ (zeroed), multi (n_maps zeroed counters), grid_scratch
 * (>= 64 * n_maps bytes). Cells identical to n_maps calls of avp_rasterize_edges (tests/test_raster.py). */
int32_t avp_rasterize_edges_batch(int32_t device, void* stream, int32_t n_maps, const double* nodes, const int64_t* node_off,
                                  const int32_t* nx, const int32_t* ny, const double* geo, const int64_t* occ_off,
                                  const double* edges, const int32_t* edge_map, int64_t n_edges, int32_t max_count,
                                  uint8_t* occ, int32_t* multi, void* grid_scratch, int64_t grid_scratch_bytes);

/* Device evaluation of the shared scalar maths (test hook): out_sin/out_cos = avp_sin/avp_cos(x). */
int32_t avp_trig_batch(avp_map* map, const double* x, int64_t n, double* out_sin, double* out_cos);
/* Device evaluation of the restated glibc libm (test hook; include/avp_glibc_libm.h): kind 0 out = atan2(a, b) as
 * math.atan2 (rs_curve.py:176,313,414,503,664), 1 asin(a) (:190), 2 acos(a) (:332,346), 3 tan(a) (:217-226),
 * 4 pow(a, 2.0) = Python's a ** 2 (:172,220,226; hybrid_a_star.py:308; collision_check.py:131-134). b is read for
 * kind 0 only. */
int32_t avp_libm_batch(avp_map* map, int32_t kind, const double* a, const double* b, int64_t n, double* out);
/* Device IEEE check hook: q = a / b, r = sqrt(|a|), h = hypot(a, b) as the kernels compute them. */
int32_t avp_ieee_batch(avp_map* map, const double* a, const double* b, int64_t n, double* q, double* r, double* h);

#ifdef __cplusplus
}
#endif
#endif /* AVP_H */
