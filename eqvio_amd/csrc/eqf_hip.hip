// eqf_hip.hip — context + C-ABI of the MI355X EqF core (see include/eqf_hip.h). gfx950 only.
#include "eqf_hip.h"
#include "eqf_kernels.hpp"
#include "eqf_lookahead.hpp"
#include "host_prof.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <cerrno>

using namespace eqf;

namespace {

enum KName {
    KN_ASSEMBLE = 0,
    KN_OBSERVER,
    KN_PROP_G,
    KN_PROP_MAIN,
    KN_MEASURE,
    KN_STATS,
    KN_BUILD_Z,
    KN_CHOL_PANEL,
    KN_CHOL_UPDATE,
    KN_GAMMA,
    KN_SYRK,
    KN_LIFT,
    KN_DENSE_GEMM,
    KN_MISC,
    KN_CHOL_LOOKAHEAD,
    KN_COUNT
};
const char* kNames[KN_COUNT] = {"k_assemble_AB", "k_observer",    "k_propagate_G", "k_propagate_main", "k_measure", "k_outlier_stats", "k_build_Z",
                                "k_chol_step",   "k_chol_first", "k_gamma",       "k_syrk_sub",       "k_lift",    "k_gemm_nt",       "misc", "k_chol_lookahead"};

int roundup(int x, int m) { return (x + m - 1) / m * m; }
int pick_ld(int x) {
    int ld = roundup(x, 16);
    if (ld % 128 == 0)
        ld += 16; // avoid power-of-two column strides (channel / cache-set aliasing)
    return ld;
}

struct SensorState { // host-resident 46 doubles
    V3 bgyr, bacc;
    Pose pose;
    V3 vel;
    Pose cam;
};
struct GroupSensor {
    V3 bgyr, bacc; // beta
    Pose A;
    V3 w;
    Pose B;
};
SensorState unpack_sensor(const double* s) {
    SensorState r;
    r.bgyr = v3(s[0], s[1], s[2]);
    r.bacc = v3(s[3], s[4], s[5]);
    r.pose = Pose{Qt{s[6], s[7], s[8], s[9]}, v3(s[10], s[11], s[12])};
    r.vel = v3(s[13], s[14], s[15]);
    r.cam = Pose{Qt{s[16], s[17], s[18], s[19]}, v3(s[20], s[21], s[22])};
    return r;
}
void pack_pose(const Pose& p, double* q, double* x) {
    q[0] = p.R.w;
    q[1] = p.R.x;
    q[2] = p.R.y;
    q[3] = p.R.z;
    x[0] = p.x.x;
    x[1] = p.x.y;
    x[2] = p.x.z;
}
void pack_v3(V3 v, double* d) {
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
}
void pack_sensor(const SensorState& r, double* s) {
    pack_v3(r.bgyr, s);
    pack_v3(r.bacc, s + 3);
    pack_pose(r.pose, s + 6, s + 10);
    pack_v3(r.vel, s + 13);
    pack_pose(r.cam, s + 16, s + 20);
}
GroupSensor unpack_group(const double* s) {
    GroupSensor g;
    g.bgyr = v3(s[0], s[1], s[2]);
    g.bacc = v3(s[3], s[4], s[5]);
    g.A = Pose{Qt{s[6], s[7], s[8], s[9]}, v3(s[10], s[11], s[12])};
    g.w = v3(s[13], s[14], s[15]);
    g.B = Pose{Qt{s[16], s[17], s[18], s[19]}, v3(s[20], s[21], s[22])};
    return g;
}
void pack_group(const GroupSensor& g, double* s) {
    pack_v3(g.bgyr, s);
    pack_v3(g.bacc, s + 3);
    pack_pose(g.A, s + 6, s + 10);
    pack_v3(g.w, s + 13);
    pack_pose(g.B, s + 16, s + 20);
}
// sensorStateGroupAction (src/mathematical/VIOGroup.cpp:25-32)
SensorState sensor_action(const GroupSensor& X, const SensorState& s) {
    SensorState r;
    r.bgyr = s.bgyr + X.bgyr;
    r.bacc = s.bacc + X.bacc;
    r.pose = pose_mul(s.pose, X.A);
    r.vel = q_rot(q_inv(X.A.R), s.vel - X.w);
    r.cam = pose_mul(pose_mul(pose_inv(X.A), s.cam), X.B);
    return r;
}
// VIOGroup::operator* on the sensor part (VIOGroup.cpp:71-92): r = a * b
GroupSensor group_mul(const GroupSensor& a, const GroupSensor& b) {
    GroupSensor r;
    r.bgyr = a.bgyr + b.bgyr;
    r.bacc = a.bacc + b.bacc;
    r.A = pose_mul(a.A, b.A);
    r.B = pose_mul(a.B, b.B);
    r.w = a.w + q_rot(a.A.R, b.w);
    return r;
}

} // namespace

struct eqf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr; // landmark part of the observer steps, concurrent with the Riccati propagation
    hipEvent_t ev_assembled = nullptr, ev_observer = nullptr, ev_early = nullptr;
    bool ev_assembled_early = false; // ev_early marks "k_assemble_AB done" of the latest Riccati call
    bool obs_pending = false;
    bool busy_common = false, busy_steps = false, busy_meas = false; // pinned packets still referenced by queued kernels
    int chart = 0;
    int Ncap = 0, ncap = 0, ld = 0;
    int mcap = 0, ldz = 0;
    int N = 0;
    std::vector<int> ids;
    SensorState xi0;
    GroupSensor X;
    // device
    // Landmark arrays, SoA planes of stride Ncap. Static part: q0 (3 planes) + the chart constants of q0 (27 planes); it changes only
    // when landmarks are added / removed. Dynamic part: Qq (4 planes) + Qa (1 plane). Both double-buffered: compaction is out of place
    // and flips both, the observer blocks of a fused propagation write the other DYNAMIC buffer and flip only that one.
    double* d_st[2] = {nullptr, nullptr};
    double* d_lm[2] = {nullptr, nullptr};
    int stcur = 0, lmcur = 0;
    double* d_sigma[2] = {nullptr, nullptr};
    int cur = 0;
    double *d_Al = nullptr, *d_Bl = nullptr;
    Common* d_common = nullptr;
    ObsStep* d_steps = nullptr;
    double *d_C = nullptr, *d_ytil = nullptr, *d_y = nullptr;
    int *d_lmidx = nullptr, *d_measof = nullptr;
    double *d_Ebuf = nullptr, *d_Yl = nullptr, *d_Fl = nullptr, *d_PhiB = nullptr; // accurate Riccati (lazily allocated)
    int* d_expinfo = nullptr;
    double *d_Zn = nullptr, *d_Wn = nullptr; // NEES factorisation buffers (lazily allocated)
    double* d_pub = nullptr;                 // look-ahead factorisation: published tiles (la_pub_tiles(NJcap) x 8 KB), their flags, the yTilde rows
    int* d_pubf = nullptr;
    int* d_pubfl = nullptr;                  // EQF_OPT_LA_HOME: flags of the hand-offs that stay inside the home XCD's L2
    int la_home = 0;                         // the XCD block 0 of this context's grids runs on (an offset of the stream's hardware queue: learnt from a refused launch)
    long la_home_refused = 0;                // launches that found their blocks elsewhere (the offset is learnt again, the update is redone on the launch chain)
    int opt_la_home = 1;                     // EQF_OPT_LA_HOME
    long la_home_launches = 0;
    bool la_home_force = false;              // the self-test exercises the HOME placement whoever else is on the device
    int reg_slot = -1;                       // this context's slot in the device's registry of contexts (shared memory: device_to_itself)
    unsigned reg_checks = 0;
    char* d_puby = nullptr;
    int la_njcap = 0, la_seq = 0;
    unsigned long long* d_ladbg = nullptr; // EQF_OPT_TRACE: stamps inside the look-ahead kernel (96 x 8)
    int opt_lookahead = 1;                   // EQF_OPT_LOOKAHEAD
    int opt_zb = 1;                          // EQF_OPT_Z_IN_LOOKAHEAD
    int opt_zb_large = 1;                    // ... also with 17 .. 32 panels (round 6); option value 2 = up to 16 panels only (round 5's behaviour, for A/B runs)
    int opt_la_split = 1;                    // EQF_OPT_LA_SPLIT_ROWS
    int opt_prop_tpw = 1;                    // EQF_OPT_TILES_PER_WORKGROUP
    int opt_gather = 1;                      // EQF_OPT_GATHER_IN_PROPAGATE
    int opt_early_door = 1;                  // EQF_OPT_EARLY_DOORBELL
    int opt_la_watch_ahead = 1;              // (debug knob, option 102) LaArgs::watch_ahead
    int early_seq_next = 0, early_armed_seq = 0; // the doorbell sequence the next look-ahead launch carries / the one the launch in flight carries
    bool early_allowed = false;              // the call in progress can take the early doorbell (eqf_stats_then_update / eqf_stats_select_update)
    long early_rings = 0;                    // updates the host took from the early doorbell
    // An update taken from the early doorbell is UNSETTLED until the lift's doorbell has been seen: the sensor lift is applied, the landmark estimates and their
    // invalid flags (the lift kernel's) are not in yet. settle_update() waits for them; every entry point that needs the device or the estimates settles first,
    // eqf_propagate_fast settles behind its launch.
    bool unsettled = false, unsettled_wait = false;
    int settle_seq = 0, settle_N = 0;
    unsigned settle_gen = 0, settle_epoch = 0, est_epoch = 0; // est_epoch: bumped wherever the cached estimates are dropped (the state moved on)
    std::vector<int> settle_ids;             // the landmark ids at the update
    std::vector<int> invalid_ids;            // ... of which the lift flagged these as invalid (VIO_eqf::removeInvalidLandmarks' test): eqf_remove_invalid_at_update
    int opt_hold = 1;                        // EQF_OPT_HOLD_NEW_LANDMARKS
    int opt_sel_one = 1;                     // EQF_OPT_SELECT_ONE_WORKGROUP
    int opt_live_first = 1;                  // EQF_OPT_LIVE_COLUMNS_FIRST
    const int* la_live_cols = nullptr;       // ... and what launch_lookahead passes on
    long live_first_launches = 0;
    // eqf_add_landmarks_held: the last n_held landmarks of the state wait for the next eqf_propagate_fast, which passes them through untouched (GatherArgs)
    int n_held = 0;
    double held_var = 0.0;
    bool held_in_memory = false;             // an entry point other than eqf_propagate_fast came in between: they were appended by an ordinary pass (still passed through untouched)
    double* h_held = nullptr;                // pinned: [0] variance, [1 + 3 t ..] point of held landmark t
    bool held_busy = false;                  // a propagation that reads h_held may still be queued (cleared by every host wait)
    long held_launches = 0;                  // propagation launches that created held landmarks themselves
    long gather_launches = 0;                // propagation launches that applied a removal record themselves
    int opt_measure_prop = 1;                // EQF_OPT_MEASURE_IN_PROPAGATE
    int opt_lift_syrk = 1;                   // EQF_OPT_LIFT_WITH_SYRK
    // ... its state: what the propagation kernel's observer blocks evaluated the output blocks with (camera and C / C* of the LAST update call stand in for the
    // coming one's; the update call checks), for which staged measurement
    bool pred_valid = false, me_valid = false;
    Cam pred_cam{}, me_cam{};
    int pred_star = 0, me_star = 0, me_M = 0;
    unsigned long me_gen = 0;
    long me_used = 0;
    bool obs_one_chunk = false;              // the propagation launch in progress carries ALL observer steps of the call (more than kMaxSteps: further k_observer launches follow)
    bool tail_zb = false;                    // the update tail in flight has no Z in memory (built inside the look-ahead kernel): a retry on the chain builds it first
    double tail_var = 0.0;                   // ... and needs the measurement variance again
    long zb_launches = 0;
    int cu_count = 256;                      // compute units of the device: the look-ahead kernel needs all its workgroups resident at once
    long la_book_timeouts = 0;               // updates that took the launch chain because the compute units did not come free within la_book's bound
    std::vector<char> rm_drop;               // eqf_remove_landmarks' scratch
    std::vector<int> rm_ids, rm_map;
    std::vector<double> rm_est;
    bool own_queue = false;                  // the stream was created with a compute-unit mask (all of them): a hardware queue of its own (create_buffers)
    bool counted_alive = false;
    int la_cus_held = 0;                     // compute units this context has booked for a look-ahead launch in flight (la_book / la_release)
    int la_selftest = 0;                     // look-ahead self-test at creation: 0 not run (never eligible at this capacity), 1 passed, -1 failed (launch chain only)
    long long la_timeout_ticks = LA_TIMEOUT_TICKS; // EQF_OPT_LA_TIMEOUT_US: bound of every device-side wait of the look-ahead kernel (100 MHz ticks)
    long la_launches = 0, la_fallbacks = 0;  // eqf_lookahead_stats: look-ahead launches; of those, stalled ones that were redone on the launch chain
    int la_consecutive_stalls = 0;           // three in a row switch the look-ahead kernel off for this context (the GPU is shared with something long-running)
    int tail_M = 0;                          // measurement count of the update tail in flight (finish_update's retry)
    bool tail_la = false;                    // ... and whether its factorisation was the look-ahead kernel
    int* d_perm = nullptr;                   // 2 x (ncap + 2) row permutations of the NEES elimination fallback
    static constexpr int kRing = 8; // pinned packets of the landmark bookkeeping: a ring, so that a flush need not drain the stream before reusing one
    int ring_pos = 0;
    int ring_inflight = 0; // flushes whose copy may still be queued: reset wherever the host has seen the stream drain past them (sync_ctx, door_wait)
    // Deferred landmark bookkeeping: eqf_remove_landmarks / eqf_add_landmarks only record what they do (ids, N and the estimate cache follow at
    // once); flush_reshape applies everything recorded since the last flush with ONE copy + ONE kernel (k_reshape) when the device state is
    // next needed. A frame's removeOldLandmarks + removeOutliers + addNewLandmarks were 3 copies + 6 launches of host time and three passes over Sigma.
    bool reshape_pending = false;
    int dev_N = 0;                 // landmarks in the device arrays (before the pending reshape)
    std::vector<int> pend_map;     // landmark -> device landmark (>= 0) or -(t + 1): t-th pending new landmark
    std::vector<double> pend_p;    // 3 per pending new landmark
    std::vector<double> pend_var;  // 1 per pending new landmark
    char* h_rs_ring = nullptr;     // kRing pinned packets: map[Ncap] ints | newp[3 Ncap] | var[Ncap] doubles
    char* d_rs = nullptr;
    size_t rs_bytes = 0;
    int spec_backoff = 0, spec_backoff_len = 0; // frames left without speculation after cancelled tails (doubling, <= 16; <= 256 with the device-side outlier decision), see eqf_stats_then_update
    long spec_calls = 0, spec_queued = 0, spec_cancelled = 0; // eqf_stats_then_update: calls, tails queued speculatively, tails cancelled on the device
    long nees_lu_fallbacks = 0;              // computeNEES calls answered by the partial-pivot elimination (Sigma not numerically SPD)
    int ldzn = 0;
    double *d_Z = nullptr, *d_W = nullptr, *d_Linv = nullptr, *d_gamma = nullptr, *d_gpart = nullptr, *d_est = nullptr, *d_stats = nullptr, *d_scratch = nullptr, *d_F = nullptr, *d_tmp = nullptr;
    int* d_flags = nullptr;
    int* d_syrk_order = nullptr;   // k_syrk_sub: block -> tile tables for every tile count nt = 1 .. ntcap (build_syrk_order), table nt at syrk_off[nt]
    std::vector<int> syrk_off;
    // pinned host staging
    Common* h_common = nullptr;
    ObsStep* h_steps = nullptr;
    double* h_buf = nullptr; // general staging, size hbuf_doubles
    bool ocov_valid = false; // h_ocov holds the output covariances of ALL landmarks at the current state for the camera ocov_cam (computed along with a state estimate)
    Cam ocov_cam{};
    bool ocov_hint_valid = false; // a caller has asked for output covariances before (eqf_output_cov_all): the next state estimate computes them along, for that camera
    Cam ocov_hint{};
    size_t hbuf_doubles = 0;
    int* h_ibuf = nullptr;
    int* h_flags = nullptr;
    int* h_lmidx = nullptr;  // pinned measurement packet: lmidx[Ncap], measof[Ncap]
    double* h_y = nullptr;   //   y[2 Ncap]
    double* h_ylm = nullptr; //   the same measurement by landmark: u[Ncap] | v[Ncap] | measurement index or -1 [Ncap]
    // eqf_stage_measurement: HBM copies of the three arrays above, made by a block of the propagation kernel
    double* d_meas = nullptr; //   y[2 Ncap] | ylm[3 Ncap]
    int* d_meas_idx = nullptr;
    bool stage_requested = false, stage_pending = false, staged_valid = false;
    int staged_M = 0;
    unsigned lm_gen = 0, staged_gen = 0; // lm_gen counts changes of the landmark set (indices in a staged measurement go stale)
    std::vector<int> staged_ids;
    std::vector<double> staged_y;
    std::vector<int> map_ids; // ids of the mapping that sits in h_lmidx (map_measurement), valid for map_gen == lm_gen
    unsigned map_gen = ~0u;
    int map_N = -1;
    bool map_all = false;
    bool map_ident = false; // ... and measurement j is landmark j for every j
    bool tail_ident = false; // ... checked against the ids of the update being launched (launch_update_tail)
    std::vector<std::pair<int, int>> lookup; // sorted (id, index), valid for lookup_gen == lm_gen
    std::vector<int> renum_scratch;
    unsigned lookup_gen = ~0u;
    // host-side wait statistics (eqf_host_wait_stats): doorbell waits and the time spent spinning in them
    // EQF_OPT_TRACE: device-side frame timeline (ring of TR_FRAMES frames x TR_SLOTS stamps, 100 MHz device wall clock)
    trace_t* d_trace = nullptr;
    unsigned trace_frame = 0;
    std::vector<long long> h_trace; // host steady_clock stamps (ns), TR_FRAMES x TR_HOST
    long wait_calls = 0, launch_calls = 0;
    double wait_seconds = 0.0, launch_seconds = 0.0;
    double* h_res = nullptr; // pinned result packet: stats[3 Ncap] | est[4 Ncap] | gamma[32]
    double* h_ocov = nullptr; // eqf_output_cov_all: pinned, 4 Ncap (allocated on first use)
    int* h_resflags = nullptr;
    int* h_sel = nullptr; // pinned: k_select_outliers' verdict, [0, N) discarded flags, [Ncap] candidates, [Ncap + 1] discarded
    long sel_frames = 0, sel_discarded = 0; // frames that took the device-side outlier decision, landmarks it discarded
    static constexpr int kMaxSteps = kObsChunk;
    CommonK ck; // kernel-argument form of the last sensor-level packet
    // options
    int opt_dense = 0, opt_check = 0, opt_timing = 0, opt_f32 = 0, opt_early = 1, opt_fuse_asm = 1;
    bool sig32 = false; // Sigma stored as float (EQF_OPT_SIGMA_FP32 = 2)
    int opt_door = 1;   // host doorbell instead of the stream completion signal for the two per-frame waits
    int* d_door = nullptr; // device counters (one per doorbell)
    int* h_door = nullptr; // pinned sequence numbers written by the last workgroup
    unsigned door_seq = 0; // wraps harmlessly: only equality is tested
    int opt_spec = 1;      // speculative frame tail allowed (eqf_stats_then_update)
    int* d_spec = nullptr; // device word the statistics kernel sets to the sequence number to cancel a queued tail
    std::vector<double> last_gamma;
    int n_at_update = 0;
    bool gamma_stale = false; // d_gamma holds a newer Gamma than last_gamma (fetched lazily by eqf_last_gamma)
    std::vector<double> est_cache; // 4 planes of stride N (q_hat xyz, invalid flag), valid after a vision update
    bool est_valid = false;
    // C / yTilde / lmidx already on the device for exactly this measurement (written by k_outlier_stats)
    bool meas_valid = false;
    int meas_star = 0;
    std::vector<int> meas_ids;
    // timing
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> tev;
    std::vector<int> tev_n; // launches inside each span
    std::vector<hipEvent_t> evpool;
    size_t evused = 0;

    double* q0() { return d_st[stcur]; }
    double* Qq() { return d_lm[lmcur]; }
    double* Qa() { return d_lm[lmcur] + 4 * (size_t)Ncap; }
    double* sigma() { return d_sigma[cur]; }
    int n() const { return 21 + 3 * N; }
};

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            std::fprintf(stderr, "[eqf_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (int)_e;                                                                            \
        }                                                                                              \
    } while (0)

namespace {

hipEvent_t get_event(eqf_ctx* c) {
    if (c->evused == c->evpool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        c->evpool.push_back(e);
    }
    return c->evpool[c->evused++];
}
constexpr int TR_FRAMES = 1024, TR_SLOTS = 48;
enum { TR_ASSEMBLE = 0, TR_PROPAGATE = 1, TR_BUILD_Z = 2, TR_STEP0 = 3, TR_LIFT = 40, TR_SYRK = 42 }; // TR_LIFT + 1, TR_SYRK + 1: end times
constexpr int TR_HOST = 8;
enum { TH_DOOR = 0, TH_PROP_ENTRY = 1, TH_ASSEMBLE_OUT = 2, TH_PROP_OUT = 3, TH_TAIL_ENTRY = 4, TH_BUILD_Z_OUT = 5, TH_TAIL_OUT = 6 };
void host_stamp(eqf_ctx* c, int k) {
    if (c->d_trace)
        c->h_trace[(size_t)(c->trace_frame % TR_FRAMES) * TR_HOST + k] = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
trace_t* trace_slot(eqf_ctx* c, int slot) { return c->d_trace ? c->d_trace + (size_t)(c->trace_frame % TR_FRAMES) * TR_SLOTS + slot : nullptr; }
struct KTimer {
    eqf_ctx* c;
    int which;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::chrono::steady_clock::time_point h0;
    int nlaunch; // launches inside the span: reported as that many entries of span / nlaunch each (no events BETWEEN them)
    // option 100 = 2 (round 6): only the factorisation's launches are bracketed and the frame is otherwise the one of the timed region (lift and covariance update one launch,
    // early doorbell): the dominant kernel's span with the GPU as busy around it as it is there
    bool on() const { return c->opt_timing == 1 || (c->opt_timing == 2 && (which == KN_CHOL_LOOKAHEAD || which == KN_CHOL_PANEL || which == KN_CHOL_UPDATE)); }
    KTimer(eqf_ctx* ctx, int w, int n = 1) : c(ctx), which(w), h0(std::chrono::steady_clock::now()), nlaunch(n) {
        if (on()) {
            e0 = get_event(c);
            e1 = get_event(c);
            hipEventRecord(e0, c->stream);
        }
    }
    ~KTimer() {
        if (on()) {
            hipEventRecord(e1, c->stream);
            c->tev.push_back({which, {e0, e1}});
            c->tev_n.push_back(nlaunch);
        }
        c->launch_calls += nlaunch - 1;
        ++c->launch_calls; // host-side cost of the launch calls (eqf_host_wait_stats)
        c->launch_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count();
    }
};
void timing_reset(eqf_ctx* c) {
    c->tev.clear();
    c->tev_n.clear();
    c->evused = 0;
}

int blocks(int n, int b) { return (n + b - 1) / b; }

// full synchronisation of the context: everything queued has finished, every pinned packet is free again
// Busy-wait on the stream: the frame has two host decision points and an interrupt-driven hipStreamSynchronize
// costs tens of microseconds of wake-up latency each time; the caller thread is dedicated to its filter anyway.
int spin_stream(hipStream_t st) {
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess)
            return 0;
        if (e != hipErrorNotReady) {
            std::fprintf(stderr, "[eqf_hip] stream error: %s\n", hipGetErrorString(e));
            return (int)e;
        }
    }
}
// Several filters on one GPU (one context each, any threads of this process): the look-ahead kernel needs ALL its workgroups resident at once, one per compute
// unit (66 at N = 200), and waits for them inside the kernel. Three such launches fit a 256-CU device; a fourth one's workgroups would trickle in as the
// others leave and every kernel involved crawls (measured: 3 filters 22.5 k updates/s aggregate, 4 filters 15 k). The launches are therefore BOOKED against
// the device's compute units: a context whose launch does not fit waits on the host until an earlier one has rung its doorbell. One filter never waits.
namespace {
std::atomic<int> ctx_alive[64]; // contexts of this process per device
thread_local bool replaces_own_queue = false; // set by grow_capacity around eqf_create
std::mutex la_gate_mutex;
std::condition_variable la_gate_cv;
int la_cus_booked[64] = {0}; // per device
} // namespace
// EQF_OPT_LA_HOME needs the device for itself. The HOME placement puts 2 NJ - 1 workgroups (25 at N = 200) of every look-ahead launch on ONE XCD of 32 compute units -
// and which XCD is a property of the stream's hardware queue: the FIRST queue of every process gives the same one (measured: four processes, one filter each, 26.3 k
// updates/s with HOME against 31.3 k without - their launches took turns on that XCD). Filters that share a GPU therefore keep the classic placement, which spreads
// every launch evenly over the XCDs, and HOME is for the filter that has the device to itself - BASELINE's configuration, one filter per GPU. "To itself" is counted
// across processes: every context registers its pid in a small table in shared memory (one per user and device), slots of dead processes are reclaimed.
namespace {
constexpr int REG_SLOTS = 64;
struct DeviceRegistry {
    std::atomic<int> pid[REG_SLOTS];
};
DeviceRegistry* registry_of(int device) {
    static std::mutex mu;
    static DeviceRegistry* maps[64] = {};
    std::lock_guard<std::mutex> g(mu);
    DeviceRegistry*& r = maps[device & 63];
    if (r)
        return r;
    char name[96];
    std::snprintf(name, sizeof(name), "/eqf_hip_contexts_u%u_d%d", (unsigned)getuid(), device);
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0)
        return nullptr;
    if (ftruncate(fd, sizeof(DeviceRegistry)) != 0) {
        close(fd);
        return nullptr;
    }
    void* p = mmap(nullptr, sizeof(DeviceRegistry), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED)
        return nullptr;
    r = static_cast<DeviceRegistry*>(p); // (a fresh object is all zeros: no slot taken)
    return r;
}
bool pid_alive(int pid) { return pid > 0 && (kill(pid, 0) == 0 || errno == EPERM); }
} // namespace
static void registry_enter(eqf_ctx* c) {
    DeviceRegistry* r = registry_of(c->device);
    c->reg_slot = -1;
    if (!r)
        return;
    const int me = (int)getpid();
    for (int pass = 0; pass < 2 && c->reg_slot < 0; ++pass)
        for (int i = 0; i < REG_SLOTS && c->reg_slot < 0; ++i) {
            int cur = r->pid[i].load(std::memory_order_relaxed);
            if (cur != 0 && (pass == 0 || pid_alive(cur)))
                continue; // pass 0: free slots only; pass 1: slots of processes that are gone as well
            if (r->pid[i].compare_exchange_strong(cur, me))
                c->reg_slot = i;
        }
}
static void registry_leave(eqf_ctx* c) {
    DeviceRegistry* r = c->reg_slot >= 0 ? registry_of(c->device) : nullptr;
    if (r)
        r->pid[c->reg_slot].store(0, std::memory_order_relaxed);
    c->reg_slot = -1;
}
// true when no other context - of this or of any other process of this user - is registered on the device
static bool device_to_itself(eqf_ctx* c) {
    DeviceRegistry* r = c->reg_slot >= 0 ? registry_of(c->device) : nullptr;
    if (!r)
        return false; // (no shared memory: nothing is known about the neighbours)
    int others = 0;
    for (int i = 0; i < REG_SLOTS; ++i)
        others += (i != c->reg_slot && r->pid[i].load(std::memory_order_relaxed) != 0) ? 1 : 0;
    if (others == 0)
        return true;
    // somebody else is registered: alive? (a crashed process leaves its slots behind; looked at once per 4096 launches)
    if ((++c->reg_checks & 0xfff) == 1) {
        for (int i = 0; i < REG_SLOTS; ++i) {
            int cur = r->pid[i].load(std::memory_order_relaxed);
            if (i != c->reg_slot && cur != 0 && cur != (int)getpid() && !pid_alive(cur))
                r->pid[i].compare_exchange_strong(cur, 0);
        }
    }
    return false;
}
static void la_release(eqf_ctx* c) {
    if (!c->la_cus_held)
        return;
    {
        std::lock_guard<std::mutex> g(la_gate_mutex);
        la_cus_booked[c->device & 63] -= c->la_cus_held;
    }
    c->la_cus_held = 0;
    la_gate_cv.notify_all();
}
// Returns false when the compute units did not come free within the bound (a context that failed without releasing, or long launches of the others): the caller
// then factorises on the launch chain, which needs no co-residency. The wait is bounded so that no context can block another one's thread for good.
static bool la_book(eqf_ctx* c, int workgroups) {
    la_release(c);
    std::unique_lock<std::mutex> g(la_gate_mutex);
    int& booked = la_cus_booked[c->device & 63];
    const auto bound = std::chrono::microseconds(std::max<long>(2000, 2 * (long)(c->la_timeout_ticks / 100)));
    if (!la_gate_cv.wait_for(g, bound, [&] { return booked == 0 || booked + workgroups <= c->cu_count; }))
        return false;
    booked += workgroups;
    c->la_cus_held = workgroups;
    return true;
}
// releases the booking on every exit of a scope that did not hand it over to a doorbell wait
struct LaBookingGuard {
    eqf_ctx* c;
    bool keep = false;
    ~LaBookingGuard() {
        if (!keep)
            la_release(c);
    }
};
int sync_ctx(eqf_ctx* c) {
    {
        int r = spin_stream(c->stream);
        if (r)
            return r;
    }
    if (c->obs_pending) {
        int r = spin_stream(c->stream2);
        if (r)
            return r;
        c->obs_pending = false;
    }
    c->busy_common = c->busy_steps = c->busy_meas = false;
    c->ring_inflight = 0;
    c->held_busy = false;
    la_release(c);
    return 0;
}
// Wait for the doorbell `which` to show `seq` (written by the last workgroup of the kernel launched with it, after every
// result store has been fenced at system scope). The stream is polled now and then so that a kernel fault is reported
// instead of spinning forever; if the stream completes without the bell (cannot happen) the call fails loudly.
int door_wait(eqf_ctx* c, int which, int seq, bool* early = nullptr) {
    HP_SCOPE("abi.door_wait");
    volatile int* bell = reinterpret_cast<volatile int*>(c->h_door) + which;
    long spins_after_done = 0;
    const auto t0 = std::chrono::steady_clock::now();
    // EQF_OPT_EARLY_DOORBELL: a caller that passes `early` also takes the look-ahead kernel's own doorbell (h_door[3]) - the update WILL be applied, Gamma's sensor rows are in
    // the packet, the lift's results are not - and is told so
    volatile int* bell_early = (which == 1 && early && c->early_armed_seq == seq) ? reinterpret_cast<volatile int*>(c->h_door) + 3 : nullptr;
    if (early)
        *early = false;
    for (long it = 1;; ++it) {
        const bool rang_early = bell_early && *bell_early == seq && *bell != seq;
        if (rang_early)
            *early = true;
        if (*bell == seq || rang_early) {
            std::atomic_thread_fence(std::memory_order_acquire);
            c->busy_common = c->busy_steps = c->busy_meas = false;
            c->ring_inflight = 0; // the kernel that rang was queued behind every flush of this context
            c->held_busy = false;
            if (!(which == 1 && c->unsettled_wait)) // (eqf_host_wait_stats counts frame boundaries: the second wait of an update taken from the early doorbell is not one)
                ++c->wait_calls;
            c->wait_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (which == 1) {
                if (!c->unsettled_wait) // (the wait for the lift's doorbell behind an update taken from the early one is not a frame boundary)
                    host_stamp(c, TH_DOOR);
                la_release(c); // the lift kernel rang: the look-ahead kernel in front of it has left the compute units
            }
            return 0;
        }
        if ((it & 0xfff) == 0 || spins_after_done) {
            const hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) {
                if (++spins_after_done > 1000000) {
                    std::fprintf(stderr, "[eqf_hip] doorbell %d never rang (expected %d, have %d)\n", which, seq, (int)*bell);
                    la_release(c); // the stream is empty: whatever was booked for a look-ahead launch has left the compute units
                    return (int)hipErrorUnknown;
                }
            } else if (e != hipErrorNotReady) {
                std::fprintf(stderr, "[eqf_hip] stream error: %s\n", hipGetErrorString(e));
                la_release(c);
                return (int)e;
            }
        }
    }
}
// make the main stream wait for the observer kernel before anything that reads / writes the landmark arrays
int join_observer(eqf_ctx* c) {
    c->ev_assembled_early = false; // the caller is about to queue a kernel that touches the landmark arrays
    if (c->obs_pending) {
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_observer, 0));
        c->obs_pending = false;
    }
    return 0;
}

Cam make_cam(const eqvio_camera* c) {
    Cam k{c->fx, c->fy, c->cx, c->cy};
    k.model = c->model;
    for (int i = 0; i < 5; ++i)
        k.d[i] = c->dist[i];
    return k;
}
bool camera_ok(const eqvio_camera* c) { return c->model >= EQVIO_CAMERA_PINHOLE && c->model <= EQVIO_CAMERA_EQUIDISTANT; }

// Sensor-level terms of A and B (EqFStateMatrixA_euclid / EqFInputMatrixB_euclid sensor rows and the per-landmark
// common factors; coordinateSuite/euclid.cpp:99-233 — identical for the inverse-depth suite, invdepth.cpp:47-63,152-166)
void compute_common(const eqf_ctx* c, const double* imu13, Common& cm) {
    const SensorState xh = sensor_action(c->X, c->xi0);
    const V3 gyr = v3(imu13[1], imu13[2], imu13[3]) - xh.bgyr; // v_est = imu - bias (IMUVelocity.cpp:52-58)
    const V6 U_I{gyr, xh.vel};
    const M3 R_A = q_mat(c->X.A.R);
    const M3 R_IC = q_mat(xh.cam.R);
    const M3 RTic = transpose(R_IC);
    auto put = [](const M3& A, double* d) {
        const double a[9] = {A.a00, A.a01, A.a02, A.a10, A.a11, A.a12, A.a20, A.a21, A.a22};
        std::memcpy(d, a, sizeof(a));
    };
    put(RTic * transpose(R_A), cm.Mv);
    put(RTic, cm.RTic);
    put(RTic * skew(xh.cam.x), cm.RTicSx);
    // ad( Ad_{T0^-1} Ad_A U_I )
    const V6 U1 = Ad_apply(pose_inv(c->xi0.cam), Ad_apply(c->X.A, U_I));
    const M6 adT = se3_adjoint(U1);
    const M6 CT = m6_mul(se3_Adjoint(pose_inv(c->X.B)), adT);
    std::memcpy(cm.CT, CT.a, sizeof(cm.CT));
    const V6 U_C = Ad_apply(pose_inv(xh.cam), U_I);
    pack_v3(U_C.v, cm.vC);
    // B sensor rows (21 x 12, row-major)
    std::memset(cm.Bs, 0, sizeof(cm.Bs));
    auto setB = [&](int r0, int c0, const M3& A) {
        const double a[9] = {A.a00, A.a01, A.a02, A.a10, A.a11, A.a12, A.a20, A.a21, A.a22};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                cm.Bs[(r0 + i) * 12 + c0 + j] = a[3 * i + j];
    };
    for (int i = 0; i < 6; ++i)
        cm.Bs[i * 12 + 6 + i] = 1.0;
    setB(6, 0, R_A);
    setB(9, 0, skew(c->X.A.x) * R_A);
    setB(12, 0, R_A * skew(xh.vel));
    setB(12, 3, R_A);
    // A sensor block (21 x 21, row-major)
    std::memset(cm.Ass, 0, sizeof(cm.Ass));
    for (int r = 0; r < 21; ++r)
        for (int cc = 0; cc < 6; ++cc)
            cm.Ass[r * 21 + cc] = -cm.Bs[r * 12 + cc];
    for (int i = 0; i < 3; ++i)
        cm.Ass[(9 + i) * 21 + 12 + i] = 1.0;
    const V3 gdir = q_rot(q_inv(c->xi0.pose.R), v3(0, 0, 1)); // xi0.sensor.gravityDir()
    const M3 Gs = (-kGravity) * skew(gdir);
    const double g[9] = {Gs.a00, Gs.a01, Gs.a02, Gs.a10, Gs.a11, Gs.a12, Gs.a20, Gs.a21, Gs.a22};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            cm.Ass[(12 + i) * 21 + 6 + j] = g[3 * i + j];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j)
            cm.Ass[(15 + i) * 21 + 15 + j] = adT.a[i * 6 + j];
    // kernel-argument form
    CommonK& ck = const_cast<eqf_ctx*>(c)->ck;
    std::memcpy(ck.lm, cm.Mv, sizeof(double) * 66); // Mv, RTic, RTicSx, CT, vC are contiguous in Common
    put(R_A, ck.RA);
    put(skew(c->X.A.x) * R_A, ck.SxRA);
    put(R_A * skew(xh.vel), ck.RAsv);
    std::memcpy(ck.G, g, sizeof(g));
    std::memcpy(ck.adT, adT.a, sizeof(ck.adT));
}

int upload_common(eqf_ctx* c, const double* imu13) {
    compute_common(c, imu13, *c->h_common); // h_common is host-only now (debug expansion); the kernels get c->ck by value
    return 0;
}
// record_early: mark "assembly done" for an observer kernel on the second stream (not needed when the observer rides along
// in the propagation kernel on this stream)
int launch_assemble(eqf_ctx* c, bool record_early = true) {
    KTimer t(c, KN_ASSEMBLE);
    int r = join_observer(c);
    if (r)
        return r;
    hipLaunchKernelGGL(k_assemble_AB, dim3(std::max(1, blocks(c->N, 64))), dim3(64), 0, c->stream, c->ck, c->N, c->Ncap, c->chart, c->d_common, c->q0(), c->Qq(),
                       c->Qa(), c->d_Al, c->d_Bl, trace_slot(c, TR_ASSEMBLE));
    if (record_early) {
        HIPCHK(hipEventRecord(c->ev_early, c->stream));
        c->ev_assembled_early = true;
    }
    return (int)hipGetLastError();
}
// d_gamma doubles as a staging buffer (eqf_set_sigma_diag, eqf_compute_nees): if it still holds a Gamma that eqf_last_gamma has not
// fetched yet, fetch it first.
int keep_last_gamma(eqf_ctx* c) {
    if (!c->gamma_stale)
        return 0;
    const int ng = c->n_at_update;
    HIPCHK(hipMemcpyAsync(c->h_buf, c->d_gamma, sizeof(double) * ng, hipMemcpyDeviceToHost, c->stream));
    { int _r = sync_ctx(c); if (_r) return _r; }
    c->last_gamma.assign(c->h_buf, c->h_buf + ng);
    c->gamma_stale = false;
    return 0;
}
int read_flags(eqf_ctx* c) {
    HIPCHK(hipMemcpyAsync(c->h_flags, c->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    { int _r = sync_ctx(c); if (_r) return _r; }
    return 0;
}
// id -> state index. The filter looks up every measured id every frame: a sorted (id, index) table, rebuilt only when the
// landmark set changes, makes that O(M log N) instead of O(M N).
// (round 5: eqf_add_landmarks / eqf_remove_landmarks keep the table up to date in O(N) - a feature source whose ids are not ascending in the state, like the synthetic
// worlds', paid a 200-element sort in every frame with landmark turnover)
void ensure_lookup(eqf_ctx* c) {
    if (c->lookup_gen != c->lm_gen || (int)c->lookup.size() != c->N) {
        c->lookup.resize(c->N);
        for (int i = 0; i < c->N; ++i)
            c->lookup[i] = {c->ids[i], i};
        std::sort(c->lookup.begin(), c->lookup.end());
        c->lookup_gen = c->lm_gen;
    }
}
int index_of(eqf_ctx* c, int id) {
    ensure_lookup(c);
    const auto it = std::lower_bound(c->lookup.begin(), c->lookup.end(), std::make_pair(id, -1));
    return (it != c->lookup.end() && it->first == id) ? it->second : -1;
}

} // namespace

extern "C" {

const char* eqf_error_string(int code) {
    switch (code) {
    case EQF_OK:
        return "ok";
    case EQF_E_NONFINITE:
        return "non-finite value in Sigma or X";
    case EQF_E_NOT_SPD:
        return "innovation covariance S is not positive definite (Cholesky pivot <= 0)";
    case EQF_E_BAD_ARG:
        return "bad argument";
    case EQF_E_CAPACITY:
        return "landmark capacity exceeded";
    case EQF_E_NO_DEVICE:
        return "no gfx950 (MI355X) HIP device available: the EqF path has no CPU fallback";
    case EQF_E_UNSUPPORTED:
        return "option combination not supported by the device path";
    case EQF_E_STALLED:
        return "the look-ahead factorisation stalled and the retry on the launch chain failed as well";
    default:
        return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}
const char* eqf_kernel_name(int which) { return (which >= 0 && which < KN_COUNT) ? kNames[which] : "?"; }

static int create_buffers(eqf_ctx* c, int max_landmarks);
static int grow_capacity(eqf_ctx* c, int new_cap);
static int flush_reshape(eqf_ctx* c);
static int round_sigma(eqf_ctx* c, const int* spec = nullptr, int spec_seq = 0);
// first statement of every entry point that uses the device state: select the device, apply the recorded landmark bookkeeping
static void materialise_held(eqf_ctx* c);
static int settle_update(eqf_ctx* c);
static int enter(eqf_ctx* c, bool settle = true) {
    HIPCHK(hipSetDevice(c->device));
    if (settle) { // an update taken from the early doorbell: the lift's results are waited for now (eqf_propagate_fast does that behind its launch)
        const int r = settle_update(c);
        if (r)
            return r;
    }
    c->ocov_valid = false; // (every call that can change the state or Sigma passes through here)
    materialise_held(c); // held landmarks (eqf_add_landmarks_held) meet an entry point other than eqf_propagate_fast: an ordinary append, still passed through by that propagation
    return flush_reshape(c);
}
static int lookahead_selftest(eqf_ctx* c);
int eqf_create(eqf_ctx** out, int device, int max_landmarks, int coordinate_choice) {
    if (!out || max_landmarks < 1 || (coordinate_choice != EQVIO_COORD_EUCLIDEAN && coordinate_choice != EQVIO_COORD_INVDEPTH && coordinate_choice != EQVIO_COORD_NORMAL))
        return EQF_E_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || device >= ndev)
        return EQF_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return EQF_E_NO_DEVICE;
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        std::fprintf(stderr, "[eqf_hip] device %d is %s, this library is built for gfx950 only\n", device, prop.gcnArchName);
        return EQF_E_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    eqf_ctx* c = new eqf_ctx();
    c->device = device;
    c->chart = coordinate_choice;
    c->cu_count = prop.multiProcessorCount;
    registry_enter(c);
    const int rc = create_buffers(c, max_landmarks);
    if (rc) { // a failed allocation half way (e.g. the pinned Sigma staging at a large capacity): release what exists
        eqf_destroy(c);
        return rc;
    }
    // EQF_OPTIONS="id=value,id=value": options applied to every new context (A/B runs of unchanged callers; a bad entry fails the creation)
    if (const char* env = std::getenv("EQF_OPTIONS")) {
        for (const char* q = env; *q;) {
            char* end = nullptr;
            const long id = std::strtol(q, &end, 10);
            if (end == q || *end != '=') {
                eqf_destroy(c);
                return EQF_E_BAD_ARG;
            }
            q = end + 1;
            const long v = std::strtol(q, &end, 10);
            if (end == q || eqf_set_option(c, (int)id, (int)v) != 0) {
                eqf_destroy(c);
                return EQF_E_BAD_ARG;
            }
            q = (*end == ',') ? end + 1 : end;
            if (*end && *end != ',') {
                eqf_destroy(c);
                return EQF_E_BAD_ARG;
            }
        }
    }
    {   // the persistent look-ahead kernel against the launch chain on a fixed problem, before the context is used (see lookahead_selftest)
        const int st = lookahead_selftest(c);
        if (st) {
            eqf_destroy(c);
            return st;
        }
    }
    *out = c;
    return EQF_OK;
}
// every allocation of a context; on failure the caller destroys the partially built context (eqf_destroy accepts null members)
// Block -> lower tile of Sigma for k_syrk_sub with nt tiles per side, XCD aware: the tiles are sorted by (super-block of g x g tiles, column, row) with
// g = sqrt(tiles / 8), the sorted list is cut into 8 equal runs, and run x is dealt to the blocks b = x, x + 8, x + 16 ... (block b runs on XCD b % 8). An XCD
// then works through about one compact g x g square, column by column: ~2 g row panels of W in its L2 instead of all nt.
static void build_syrk_order(int nt, int* out) {
    const int ntiles = nt * (nt + 1) / 2;
    int g = 1;
    while ((g + 1) * (g + 1) * 8 <= ntiles)
        ++g;
    std::vector<std::pair<long, int>> key;
    key.reserve(ntiles);
    for (int bj = 0; bj < nt; ++bj)
        for (int bi = bj; bi < nt; ++bi) {
            // super-block columns are walked alternately downwards and upwards (boustrophedon), so that a run which crosses from one super-block column
            // into the next stays in neighbouring super-blocks
            const int SJ = bj / g, SI = bi / g, SImax = (nt - 1) / g;
            const int SIk = (SJ & 1) ? SImax - SI : SI;
            key.push_back({(((long)SJ * 4096 + SIk) * 4096 + bj) * 4096 + bi, bi | (bj << 16)});
        }
    std::sort(key.begin(), key.end());
    const int base = ntiles / 8, rem = ntiles % 8;
    int start = 0;
    for (int x = 0; x < 8; ++x) {
        const int len = base + (x < rem ? 1 : 0);
        for (int l = 0; l < len; ++l)
            out[8 * l + x] = key[start + l].second;
        start += len;
    }
}
static int create_buffers(eqf_ctx* c, int max_landmarks) {
    c->Ncap = roundup(max_landmarks, 16);
    c->ncap = 21 + 3 * c->Ncap;
    c->ld = pick_ld(c->ncap);
    c->mcap = 2 * c->Ncap;
    c->ldz = pick_ld(c->mcap + c->ncap + 1);
    // Round 5: the runtime hands a process' plain streams GPU_MAX_HW_QUEUES = 4 hardware queues, and streams that share one run their kernels one after the other. FOUR
    // filters in one process ran at 20.4 k updates/s aggregate (N = 200) where three reached 28.5 k - two pairs of streams on two queues, the numbers GPU_MAX_HW_QUEUES = 2
    // gives. A stream created with a compute-unit mask owns its hardware queue: with EQF_OWN_HW_QUEUES=<n> in the environment the first n contexts of a process on a device
    // get such a stream (mask: all compute units) - four filters at N = 200: 32.5 - 32.8 k in every run. Not the default: such streams are blocking streams, more than four
    // of them were slower than the shared queues (N = 50, 8 / 16 filters: 45 / 58 k against 57 / 67 k), and at N = 50 four of them gave 69 k or 23 k depending on what the
    // process had created before (plain: 44 k). One filter per process (the headline, one rank per GPU) measured the same either way.
    c->own_queue = false;
    const int before = ctx_alive[c->device & 63].fetch_add(1);
    const char* own_env = std::getenv("EQF_OWN_HW_QUEUES");
    // (replaces_own_queue: grow_capacity builds the replacement of a context that HAS a queue of its own while that context is still counted - it keeps one)
    if ((own_env && before < atoi(own_env)) || replaces_own_queue) {
        std::vector<uint32_t> mask((size_t)(c->cu_count + 31) / 32, 0u);
        for (int i = 0; i < c->cu_count; ++i)
            mask[i >> 5] |= 1u << (i & 31);
        if (hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data()) == hipSuccess)
            c->own_queue = true;
        else {
            (void)hipGetLastError();
            c->stream = nullptr;
        }
    }
    c->counted_alive = true;
    if (!c->stream)
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // (the second stream is created by the first stand-alone eqf_integrate_observer call: a stream is a hardware queue, the runtime hands out GPU_MAX_HW_QUEUES = 4 of them per
    //  process by default, and streams that share a queue run one after the other - with two streams per context a GPU saturated at TWO filters)
    HIPCHK(hipEventCreateWithFlags(&c->ev_assembled, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_observer, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_early, hipEventDisableTiming));
    const size_t sig_bytes = sizeof(double) * (size_t)c->ld * c->ncap;
    for (int b = 0; b < 2; ++b) {
        HIPCHK(hipMalloc(&c->d_sigma[b], sig_bytes));
        HIPCHK(hipMemsetAsync(c->d_sigma[b], 0, sig_bytes, c->stream));
        HIPCHK(hipMalloc(&c->d_st[b], sizeof(double) * (CC_OFF + CC_PLANES) * (size_t)c->Ncap));
        HIPCHK(hipMemsetAsync(c->d_st[b], 0, sizeof(double) * (CC_OFF + CC_PLANES) * (size_t)c->Ncap, c->stream));
        HIPCHK(hipMalloc(&c->d_lm[b], sizeof(double) * 5 * (size_t)c->Ncap));
        HIPCHK(hipMemsetAsync(c->d_lm[b], 0, sizeof(double) * 5 * (size_t)c->Ncap, c->stream));
    }
    HIPCHK(hipMalloc(&c->d_Al, sizeof(double) * 45 * (size_t)c->Ncap));
    HIPCHK(hipMalloc(&c->d_Bl, sizeof(double) * 9 * (size_t)c->Ncap));
    HIPCHK(hipMalloc(&c->d_common, sizeof(Common)));
    HIPCHK(hipMalloc(&c->d_steps, sizeof(ObsStep) * eqf_ctx::kMaxSteps));
    HIPCHK(hipMalloc(&c->d_C, sizeof(double) * 6 * (size_t)c->Ncap));
    HIPCHK(hipMalloc(&c->d_ytil, sizeof(double) * c->mcap));
    HIPCHK(hipMalloc(&c->d_y, sizeof(double) * c->mcap));
    HIPCHK(hipMalloc(&c->d_lmidx, sizeof(int) * c->Ncap));
    HIPCHK(hipMalloc(&c->d_measof, sizeof(int) * c->Ncap));
    HIPCHK(hipMalloc(&c->d_Z, sizeof(double) * (size_t)c->ldz * c->mcap));
    HIPCHK(hipMalloc(&c->d_W, sizeof(double) * (size_t)c->ldz * c->mcap));
    HIPCHK(hipMalloc(&c->d_Linv, sizeof(double) * 2048));
    HIPCHK(hipMalloc(&c->d_gamma, sizeof(double) * (c->ncap + 8)));
    HIPCHK(hipMalloc(&c->d_gpart, sizeof(double) * (GAMMA_G + 1) * (size_t)c->ld));
    c->la_njcap = std::min(32, blocks(c->mcap, 32));
    HIPCHK(hipMalloc(&c->d_pub, sizeof(double) * LA_TILE * la_pub_tiles(c->la_njcap)));
    HIPCHK(hipMalloc(&c->d_pubf, sizeof(int) * la_pub_flags(c->la_njcap)));
    HIPCHK(hipMalloc(&c->d_puby, 512 * (size_t)c->la_njcap));
    HIPCHK(hipMemsetAsync(c->d_pubf, 0, sizeof(int) * la_pub_flags(c->la_njcap), c->stream)); // sequence 0 is never used by a launch
    HIPCHK(hipMalloc(&c->d_pubfl, sizeof(int) * la_pub_flags(c->la_njcap)));
    HIPCHK(hipMemsetAsync(c->d_pubfl, 0, sizeof(int) * la_pub_flags(c->la_njcap), c->stream));
    HIPCHK(hipMemsetAsync(c->d_puby, 0, 512 * (size_t)c->la_njcap, c->stream));
    HIPCHK(hipMalloc(&c->d_est, sizeof(double) * 4 * (size_t)c->Ncap));
    HIPCHK(hipMalloc(&c->d_stats, sizeof(double) * 3 * (size_t)c->Ncap));
    HIPCHK(hipMalloc(&c->d_scratch, sizeof(double) * 8 * (size_t)c->Ncap));
    HIPCHK(hipMalloc(&c->d_flags, sizeof(int) * 8)); // [0] pivot not positive, [1] non-finite, [3] look-ahead stalled (sequence valued), [4] HOME placement refused (sequence valued)
    HIPCHK(hipMemsetAsync(c->d_flags, 0, sizeof(int) * 8, c->stream));
    {
        const int ntcap = blocks(c->ncap, 32);
        c->syrk_off.assign(ntcap + 2, 0);
        for (int nt = 1; nt <= ntcap; ++nt)
            c->syrk_off[nt + 1] = c->syrk_off[nt] + nt * (nt + 1) / 2;
        std::vector<int> all(c->syrk_off[ntcap + 1]);
        for (int nt = 1; nt <= ntcap; ++nt)
            build_syrk_order(nt, all.data() + c->syrk_off[nt]);
        HIPCHK(hipMalloc(&c->d_syrk_order, sizeof(int) * all.size()));
        HIPCHK(hipMemcpy(c->d_syrk_order, all.data(), sizeof(int) * all.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(hipHostMalloc(&c->h_common, sizeof(Common)));
    HIPCHK(hipHostMalloc(&c->h_steps, sizeof(ObsStep) * eqf_ctx::kMaxSteps));
    c->hbuf_doubles = (size_t)c->ld * c->ncap; // large enough for a full Sigma transfer
    HIPCHK(hipHostMalloc(&c->h_buf, sizeof(double) * c->hbuf_doubles));
    HIPCHK(hipHostMalloc(&c->h_ibuf, sizeof(int) * 4 * (size_t)c->Ncap));
    c->rs_bytes = (sizeof(int) + 4 * sizeof(double)) * (size_t)c->Ncap + 64;
    HIPCHK(hipHostMalloc(&c->h_rs_ring, eqf_ctx::kRing * c->rs_bytes));
    HIPCHK(hipMalloc(&c->d_rs, c->rs_bytes));
    HIPCHK(hipHostMalloc(&c->h_flags, sizeof(int) * 4));
    HIPCHK(hipHostMalloc(&c->h_lmidx, sizeof(int) * 2 * (size_t)c->Ncap));
    HIPCHK(hipHostMalloc(&c->h_y, sizeof(double) * 2 * (size_t)c->Ncap));
    HIPCHK(hipHostMalloc(&c->h_ylm, sizeof(double) * 3 * (size_t)c->Ncap));
    HIPCHK(hipMalloc(&c->d_meas, sizeof(double) * 5 * (size_t)c->Ncap));
    HIPCHK(hipMalloc(&c->d_meas_idx, sizeof(int) * (size_t)c->Ncap));
    HIPCHK(hipHostMalloc(&c->h_res, sizeof(double) * (7 * (size_t)c->Ncap + 32)));
    HIPCHK(hipHostMalloc(&c->h_resflags, sizeof(int) * 4));
    HIPCHK(hipHostMalloc(&c->h_sel, sizeof(int) * ((size_t)c->Ncap + 2)));
    HIPCHK(hipHostMalloc(&c->h_held, sizeof(double) * (3 * (size_t)c->Ncap + 1)));
    HIPCHK(hipHostMalloc(&c->h_door, sizeof(int) * 4));
    std::memset(c->h_door, 0, sizeof(int) * 4);
    HIPCHK(hipMalloc(&c->d_door, sizeof(int) * 4));
    HIPCHK(hipMemset(c->d_door, 0, sizeof(int) * 4));
    HIPCHK(hipMalloc(&c->d_spec, sizeof(int) * 4));
    HIPCHK(hipMemset(c->d_spec, 0, sizeof(int) * 4));
    // identity state
    const double s0[23] = {0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
    c->xi0 = unpack_sensor(s0);
    c->X = unpack_group(s0);
    return sync_ctx(c);
}

void eqf_destroy(eqf_ctx* c) {
    if (c) {
        la_release(c);
        registry_leave(c);
        if (c->counted_alive)
            ctx_alive[c->device & 63].fetch_sub(1);
        c->counted_alive = false;
    }
    if (c && std::getenv("EQF_DEBUG_STATS"))
        std::fprintf(stderr, "[eqf_hip] look-ahead launches %ld (stalled %ld), of them with Z built inside %ld\n", c->la_launches, c->la_fallbacks, c->zb_launches);
    if (c && std::getenv("EQF_DEBUG_STATS") && c->early_rings)
        std::fprintf(stderr, "[eqf_hip] updates taken from the early doorbell: %ld\n", c->early_rings);
    if (!c)
        return;
    hipSetDevice(c->device);
    if (c->stream)
        hipStreamSynchronize(c->stream);
    if (c->stream2)
        hipStreamSynchronize(c->stream2);
    for (int b = 0; b < 2; ++b) {
        hipFree(c->d_sigma[b]);
        hipFree(c->d_lm[b]);
        hipFree(c->d_st[b]);
    }
    hipFree(c->d_Al);
    hipFree(c->d_Bl);
    hipFree(c->d_common);
    hipFree(c->d_steps);
    hipFree(c->d_C);
    hipFree(c->d_ytil);
    hipFree(c->d_y);
    hipFree(c->d_lmidx);
    hipFree(c->d_measof);
    hipFree(c->d_Z);
    hipFree(c->d_W);
    hipFree(c->d_Linv);
    hipFree(c->d_gamma);
    hipFree(c->d_gpart);
    hipFree(c->d_pub);
    hipFree(c->d_pubf);
    hipFree(c->d_pubfl);
    hipFree(c->d_puby);
    if (c->d_ladbg)
        hipFree(c->d_ladbg);
    if (c->d_trace)
        hipFree(c->d_trace);
    hipFree(c->d_est);
    hipFree(c->d_stats);
    hipFree(c->d_scratch);
    hipFree(c->d_flags);
    hipFree(c->d_syrk_order);
    if (c->d_Ebuf) {
        hipFree(c->d_Ebuf);
        hipFree(c->d_Yl);
        hipFree(c->d_Fl);
        hipFree(c->d_PhiB);
        hipFree(c->d_expinfo);
    }
    if (c->d_Zn)
        hipFree(c->d_Zn);
    if (c->d_Wn)
        hipFree(c->d_Wn);
    if (c->d_perm)
        hipFree(c->d_perm);
    if (c->d_F)
        hipFree(c->d_F);
    if (c->d_tmp)
        hipFree(c->d_tmp);
    hipHostFree(c->h_common);
    hipHostFree(c->h_steps);
    hipHostFree(c->h_buf);
    hipHostFree(c->h_ibuf);
    hipHostFree(c->h_rs_ring);
    hipFree(c->d_rs);
    hipHostFree(c->h_flags);
    hipHostFree(c->h_lmidx);
    hipHostFree(c->h_y);
    hipHostFree(c->h_ylm);
    hipFree(c->d_meas);
    hipFree(c->d_meas_idx);
    hipHostFree(c->h_res);
    if (c->h_ocov)
        hipHostFree(c->h_ocov);
    c->h_ocov = nullptr;
    hipHostFree(c->h_resflags);
    hipHostFree(c->h_sel);
    hipHostFree(c->h_held);
    hipHostFree(c->h_door);
    hipFree(c->d_door);
    hipFree(c->d_spec);
    if (c->ev_assembled)
        hipEventDestroy(c->ev_assembled);
    if (c->ev_observer)
        hipEventDestroy(c->ev_observer);
    if (c->ev_early)
        hipEventDestroy(c->ev_early);
    if (c->stream2)
        hipStreamDestroy(c->stream2);
    for (auto e : c->evpool)
        hipEventDestroy(e);
    if (c->stream)
        hipStreamDestroy(c->stream);
    delete c;
}

// Launch KERNEL<double> or KERNEL<float> according to the storage type of Sigma; Sigma pointer arguments are written (TS*)ptr.
#define LAUNCH_TS(c, KERNEL, grid, block, stream, ...)                                                        \
    do {                                                                                                      \
        if ((c)->sig32) {                                                                                     \
            using TS = float;                                                                                 \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<TS>), grid, block, 0, stream, __VA_ARGS__);             \
        } else {                                                                                              \
            using TS = double;                                                                                \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<TS>), grid, block, 0, stream, __VA_ARGS__);             \
        }                                                                                                     \
    } while (0)

// switch the storage type of Sigma: convert the current buffer into the other one and flip
static int set_sigma_storage(eqf_ctx* c, bool f32) {
    if (c->sig32 == f32)
        return 0;
    const int n = c->n();
    double* in = c->d_sigma[c->cur];
    double* out = c->d_sigma[1 - c->cur];
    if (f32)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_convert_sigma<double, float>), dim3(blocks(n, 256), n), dim3(256), 0, c->stream, n, c->ld, in, (float*)out);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_convert_sigma<float, double>), dim3(blocks(n, 256), n), dim3(256), 0, c->stream, n, c->ld, (const float*)in, out);
    HIPCHK(hipGetLastError());
    c->cur = 1 - c->cur;
    c->sig32 = f32;
    return 0;
}

static int round_sigma(eqf_ctx* c, const int* spec, int spec_seq) {
    if ((c->opt_f32 != 1 && c->opt_f32 != 3) || c->n() == 0) // 2 = real float storage: every store already rounds
        return 0;
    const int n = c->n();
    hipLaunchKernelGGL(k_round_f32, dim3(blocks(n, 256), n), dim3(256), 0, c->stream, n, c->ld, c->sigma(), spec, spec_seq, c->opt_f32 == 3 ? 1 : 0);
    HIPCHK(hipGetLastError());
    return 0;
}

int eqf_get_option(const eqf_ctx* c, int option, int* value) {
    if (!c || !value)
        return EQF_E_BAD_ARG;
    switch (option) {
    case EQF_OPT_RICCATI_DENSE: *value = c->opt_dense; return 0;
    case EQF_OPT_CHECK_FINITE: *value = c->opt_check; return 0;
    case EQF_OPT_SPECULATIVE: *value = c->opt_spec; return 0;
    case EQF_OPT_DOORBELL: *value = c->opt_door; return 0;
    case EQF_OPT_EARLY_LIFT: *value = c->opt_early; return 0;
    case EQF_OPT_FUSED_ASSEMBLY: *value = c->opt_fuse_asm; return 0;
    case EQF_OPT_Z_IN_LOOKAHEAD: *value = c->opt_zb ? (c->opt_zb_large ? 1 : 2) : 0; return 0;
    case EQF_OPT_LA_SPLIT_ROWS: *value = c->opt_la_split; return 0;
    case EQF_OPT_TILES_PER_WORKGROUP: *value = c->opt_prop_tpw; return 0;
    case EQF_OPT_GATHER_IN_PROPAGATE: *value = c->opt_gather; return 0;
    case EQF_OPT_HOLD_NEW_LANDMARKS: *value = c->opt_hold; return 0;
    case EQF_OPT_EARLY_DOORBELL: *value = c->opt_early_door; return 0;
    case EQF_OPT_SELECT_ONE_WORKGROUP: *value = c->opt_sel_one; return 0;
    case EQF_OPT_LIVE_COLUMNS_FIRST: *value = c->opt_live_first; return 0;
    case EQF_OPT_LA_HOME: *value = c->opt_la_home; return 0;
    case EQF_OPT_MEASURE_IN_PROPAGATE: *value = c->opt_measure_prop; return 0;
    case EQF_OPT_LIFT_WITH_SYRK: *value = c->opt_lift_syrk; return 0;
    case EQF_OPT_LOOKAHEAD: *value = c->opt_lookahead; return 0;
    case EQF_OPT_LA_TIMEOUT_US: *value = (int)(c->la_timeout_ticks / 100); return 0;
    case EQF_OPT_TRACE: *value = c->d_trace ? 1 : 0; return 0;
    case EQF_OPT_SIGMA_FP32: *value = c->opt_f32; return 0;
    case 100: *value = c->opt_timing; return 0;
    case 102: *value = c->opt_la_watch_ahead; return 0;
    default: return EQF_E_BAD_ARG;
    }
}
int eqf_set_option(eqf_ctx* c, int option, int value) {
    if (!c)
        return EQF_E_BAD_ARG;
    switch (option) {
    case EQF_OPT_RICCATI_DENSE:
        if (value && c->sig32)
            return EQF_E_UNSUPPORTED;
        c->opt_dense = value;
        return 0;
    case EQF_OPT_CHECK_FINITE:
        c->opt_check = value;
        return 0;
    case EQF_OPT_SPECULATIVE:
        c->opt_spec = value;
        return 0;
    case EQF_OPT_DOORBELL:
        c->opt_door = value;
        return 0;
    case EQF_OPT_EARLY_LIFT:
        c->opt_early = value;
        return 0;
    case EQF_OPT_FUSED_ASSEMBLY:
        c->opt_fuse_asm = value;
        return 0;
    case EQF_OPT_Z_IN_LOOKAHEAD:
        c->opt_zb = value ? 1 : 0, c->opt_zb_large = value == 2 ? 0 : 1;
        return 0;
    case EQF_OPT_LA_SPLIT_ROWS:
        c->opt_la_split = value ? 1 : 0;
        return 0;
    case EQF_OPT_LA_HOME:
        c->opt_la_home = value < 0 ? 0 : std::min(value, 2); // 2: also when the device is shared (tests)
        return 0;
    case EQF_OPT_TILES_PER_WORKGROUP:
        if (value < 0 || value > 8)
            return EQF_E_BAD_ARG;
        c->opt_prop_tpw = value;
        return 0;
    case EQF_OPT_GATHER_IN_PROPAGATE:
        c->opt_gather = value ? 1 : 0;
        return 0;
    case EQF_OPT_HOLD_NEW_LANDMARKS:
        c->opt_hold = value ? 1 : 0;
        return 0;
    case EQF_OPT_SELECT_ONE_WORKGROUP:
        c->opt_sel_one = value ? 1 : 0;
        return 0;
    case EQF_OPT_LIVE_COLUMNS_FIRST:
        c->opt_live_first = value ? 1 : 0;
        return 0;
    case EQF_OPT_EARLY_DOORBELL:
        c->opt_early_door = value ? 1 : 0;
        return 0;
    case EQF_OPT_MEASURE_IN_PROPAGATE:
        c->opt_measure_prop = value ? 1 : 0;
        c->me_valid = false;
        return 0;
    case EQF_OPT_LIFT_WITH_SYRK:
        c->opt_lift_syrk = value ? 1 : 0;
        return 0;
    case EQF_OPT_LOOKAHEAD:
        c->opt_lookahead = value;
        c->la_consecutive_stalls = 0;
        return 0;
    case EQF_OPT_LA_TIMEOUT_US:
        if (value < 0)
            return EQF_E_BAD_ARG;
        c->la_timeout_ticks = 100ll * value; // 100 MHz device wall clock
        return 0;
    case EQF_OPT_TRACE: {
        { int _r = sync_ctx(c); if (_r) return _r; }
        if (value && !c->d_trace) {
            HIPCHK(hipMalloc(&c->d_trace, sizeof(trace_t) * TR_FRAMES * TR_SLOTS));
            HIPCHK(hipMemset(c->d_trace, 0, sizeof(trace_t) * TR_FRAMES * TR_SLOTS));
            if (!c->d_ladbg) {
                HIPCHK(hipMalloc(&c->d_ladbg, 96 * 8 * sizeof(unsigned long long)));
                HIPCHK(hipMemset(c->d_ladbg, 0, 96 * 8 * sizeof(unsigned long long)));
            }
            c->h_trace.assign((size_t)TR_FRAMES * TR_HOST, 0);
        } else if (!value && c->d_trace) {
            hipFree(c->d_trace);
            c->d_trace = nullptr;
        }
        return 0;
    }
    case EQF_OPT_SIGMA_FP32: {
        if (value < 0 || value > 3)
            return EQF_E_BAD_ARG;
        if (value == 2 && c->opt_dense)
            return EQF_E_UNSUPPORTED; // the float store exists for the structured fast path only
        c->opt_f32 = value;
        { int _e = enter(c); if (_e) return _e; } // the live Sigma is converted: pending landmark bookkeeping first
        const int rc = set_sigma_storage(c, value == 2);
        return rc ? rc : round_sigma(c);
    }
    case 100:
        c->opt_timing = value;
        timing_reset(c);
        return 0;
    case 102: // (debug knob, not in the header: same-process A/B of LaArgs::watch_ahead)
        c->opt_la_watch_ahead = value ? 1 : 0;
        return 0;
    default:
        return EQF_E_BAD_ARG;
    }
}
int eqf_synchronize(eqf_ctx* c) {
    if (!c)
        return EQF_E_BAD_ARG;
    { int _r = settle_update(c); if (_r) return _r; }
    { int _r = sync_ctx(c); if (_r) return _r; }
    return 0;
}
int eqf_num_landmarks(const eqf_ctx* c) { return c->N; }
int eqf_get_ids(const eqf_ctx* c, int* ids, int cap) {
    if (!c || (!ids && c->N > 0))
        return EQF_E_BAD_ARG;
    if (c->N > cap)
        return EQF_E_CAPACITY;
    for (int i = 0; i < c->N; ++i)
        ids[i] = c->ids[i];
    return c->N;
}
void* eqf_stream(eqf_ctx* c) { return (void*)c->stream; }

int eqf_set_state(eqf_ctx* c, const double* xi0_sensor, const double* X_sensor, const int* ids, const double* q0, const double* Q, int N) {
    if (!c || N < 0 || (N > 0 && (!ids || !q0 || !Q)))
        return EQF_E_BAD_ARG;
    { int _e = enter(c); if (_e) return _e; }
    if (N > c->Ncap) {
        const int rc = grow_capacity(c, std::max(N, 2 * c->Ncap));
        if (rc)
            return rc;
    }
    { int _r = sync_ctx(c); if (_r) return _r; }
    c->est_valid = false, ++c->est_epoch;
    c->meas_valid = false;
    c->n_held = 0, c->held_in_memory = false;
    c->xi0 = unpack_sensor(xi0_sensor);
    c->X = unpack_group(X_sensor);
    c->ids.assign(ids, ids + N);
    c->N = N;
    c->dev_N = N;
    ++c->lm_gen;
    if (N > 0) {
        std::memcpy(c->h_buf, q0, sizeof(double) * 3 * N);
        std::memcpy(c->h_buf + 3 * N, Q, sizeof(double) * 5 * N);
        HIPCHK(hipMemcpyAsync(c->d_scratch, c->h_buf, sizeof(double) * 8 * N, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_scatter_landmarks, dim3(blocks(N, 64)), dim3(64), 0, c->stream, N, 0, c->Ncap, c->d_scratch, c->d_scratch + 3 * N, c->q0(), c->Qq(),
                           c->Qa());
        HIPCHK(hipGetLastError());
        { int _r = sync_ctx(c); if (_r) return _r; }
    }
    return 0;
}

int eqf_get_state(eqf_ctx* c, double* xi0_sensor, double* X_sensor, int* ids, double* q0, double* Q, int cap) {
    if (!c)
        return EQF_E_BAD_ARG;
    if (xi0_sensor)
        pack_sensor(c->xi0, xi0_sensor);
    if (X_sensor)
        pack_group(c->X, X_sensor);
    const int N = c->N;
    if (N > cap)
        return EQF_E_CAPACITY;
    if (N > 0) {
        { int _e = enter(c); if (_e) return _e; }
        { int _r = join_observer(c); if (_r) return _r; }
        hipLaunchKernelGGL(k_gather_landmarks_aos, dim3(blocks(N, 64)), dim3(64), 0, c->stream, N, c->Ncap, c->q0(), c->Qq(), c->Qa(), c->d_scratch);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->h_buf, c->d_scratch, sizeof(double) * 8 * N, hipMemcpyDeviceToHost, c->stream));
        { int _r = sync_ctx(c); if (_r) return _r; }
        for (int i = 0; i < N; ++i) {
            if (ids)
                ids[i] = c->ids[i];
            if (q0)
                std::memcpy(q0 + 3 * i, c->h_buf + 8 * i, sizeof(double) * 3);
            if (Q)
                std::memcpy(Q + 5 * i, c->h_buf + 8 * i + 3, sizeof(double) * 5);
        }
    }
    return N;
}

int eqf_set_sigma(eqf_ctx* c, const double* sig, int n) {
    if (!c || !sig || n != c->n())
        return EQF_E_BAD_ARG;
    { int _e = enter(c); if (_e) return _e; }
    { int _r = sync_ctx(c); if (_r) return _r; }
    std::memcpy(c->h_buf, sig, sizeof(double) * (size_t)n * n);
    if (c->sig32) { // doubles land in the other buffer, the conversion kernel writes the float store
        double* stage = c->d_sigma[1 - c->cur];
        HIPCHK(hipMemcpy2DAsync(stage, sizeof(double) * c->ld, c->h_buf, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_convert_sigma<double, float>), dim3(blocks(n, 256), n), dim3(256), 0, c->stream, n, c->ld, stage, (float*)c->sigma());
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemcpy2DAsync(c->sigma(), sizeof(double) * c->ld, c->h_buf, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyHostToDevice, c->stream));
    }
    { int _r = round_sigma(c); if (_r) return _r; }
    { int _r = sync_ctx(c); if (_r) return _r; }
    return 0;
}
int eqf_set_sigma_diag(eqf_ctx* c, const double* diag, int n) {
    if (!c || !diag || n != c->n())
        return EQF_E_BAD_ARG;
    { int _e = enter(c); if (_e) return _e; }
    { int _r = sync_ctx(c); if (_r) return _r; }
    { int _r = keep_last_gamma(c); if (_r) return _r; }
    std::memcpy(c->h_buf, diag, sizeof(double) * n);
    HIPCHK(hipMemcpyAsync(c->d_gamma, c->h_buf, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    LAUNCH_TS(c, k_set_diag, dim3(blocks(n, 256), n), dim3(256), c->stream, n, c->ld, c->d_gamma, (TS*)c->sigma());
    HIPCHK(hipGetLastError());
    { int _r = round_sigma(c); if (_r) return _r; }
    { int _r = sync_ctx(c); if (_r) return _r; }
    return 0;
}
int eqf_get_sigma_block(eqf_ctx* c, int r0, int c0, int rows, int cols, double* out) {
    if (!c || !out || r0 < 0 || c0 < 0 || rows < 0 || cols < 0 || r0 + rows > c->n() || c0 + cols > c->n())
        return EQF_E_BAD_ARG;
    if (rows == 0 || cols == 0)
        return 0;
    { int _e = enter(c); if (_e) return _e; }
    const double* src = c->sigma();
    if (c->sig32) { // widen into the other buffer first
        const int n = c->n();
        double* stage = c->d_sigma[1 - c->cur];
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_convert_sigma<float, double>), dim3(blocks(n, 256), n), dim3(256), 0, c->stream, n, c->ld, (const float*)c->sigma(), stage);
        HIPCHK(hipGetLastError());
        src = stage;
    }
    HIPCHK(hipMemcpy2DAsync(c->h_buf, sizeof(double) * rows, src + r0 + (size_t)c0 * c->ld, sizeof(double) * c->ld, sizeof(double) * rows, cols,
                            hipMemcpyDeviceToHost, c->stream));
    { int _r = sync_ctx(c); if (_r) return _r; }
    std::memcpy(out, c->h_buf, sizeof(double) * (size_t)rows * cols);
    return 0;
}
int eqf_get_sigma(eqf_ctx* c, double* sig, int n) {
    if (!c || n != c->n())
        return EQF_E_BAD_ARG;
    return eqf_get_sigma_block(c, 0, 0, n, n, sig);
}

static int fetch_estimates(eqf_ctx* c) { // d_est -> h_buf (4 planes of stride N)
    { int _r = settle_update(c); if (_r) return _r; }
    const int N = c->N;
    if (N == 0)
        return 0;
    if (c->est_valid && (int)c->est_cache.size() == 4 * N) { // (kept current through landmark bookkeeping: no device work, no flush of a pending reshape)
        std::memcpy(c->h_buf, c->est_cache.data(), sizeof(double) * 4 * N);
        return 0;
    }
    { int _e = enter(c); if (_e) return _e; }
    { int _r = join_observer(c); if (_r) return _r; }
    // straight into the pinned staging buffer (no copy command behind the kernel: a blit and its boundary cost more than the 6 KB written across the bus)
    // (round 5) the host waits on a doorbell rung by the last of these kernels instead of the stream's completion signal (~8 us per state estimate read between propagation
    // and update: the reference's removeOutliers, the reference-side binding's member-for-member sequence)
    const bool with_ocov = c->ocov_hint_valid && !c->sig32;
    const bool use_door = c->opt_door && !c->opt_check && !c->obs_pending;
    const int door_seq = use_door ? (int)(++c->door_seq) : 0;
    int* const dcount = use_door ? c->d_door + 2 : nullptr;
    hipLaunchKernelGGL(k_estimate, dim3(blocks(N, 64)), dim3(64), 0, c->stream, N, c->Ncap, c->q0(), c->Qq(), c->Qa(), c->h_buf, with_ocov ? (int*)nullptr : dcount, c->h_door + 2, door_seq);
    HIPCHK(hipGetLastError());
    // A caller that reads the state estimate between propagation and update is the reference's removeOutliers (src/VIOFilter.cpp:304-334), which asks for the output
    // covariance of every measured landmark next: computed here as well, for the camera of the last such request, behind the same wait (eqf_output_cov_all returns it)
    if (with_ocov) {
        if (!c->h_ocov)
            HIPCHK(hipHostMalloc(&c->h_ocov, sizeof(double) * 4 * (size_t)c->Ncap));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_output_cov<double>), dim3(blocks(N, 64)), dim3(64), 0, c->stream, N, c->Ncap, c->ld, c->chart, c->ocov_hint, c->q0(), c->Qq(), c->Qa(),
                           (const double*)c->sigma(), c->h_ocov, dcount, c->h_door + 2, door_seq);
        HIPCHK(hipGetLastError());
    }
    { int _r = use_door ? door_wait(c, 2, door_seq) : sync_ctx(c); if (_r) return _r; }
    if (with_ocov)
        c->ocov_valid = true, c->ocov_cam = c->ocov_hint;
    return 0;
}

int eqf_state_estimate(eqf_ctx* c, double* sensor, int* ids, double* p, int cap) {
    if (!c)
        return EQF_E_BAD_ARG;
    if (sensor)
        pack_sensor(sensor_action(c->X, c->xi0), sensor);
    const int N = c->N;
    if (N > cap)
        return EQF_E_CAPACITY;
    if (N > 0 && (ids || p)) {
        HIPCHK(hipSetDevice(c->device));
        int rc = fetch_estimates(c);
        if (rc)
            return rc;
        for (int i = 0; i < N; ++i) {
            if (ids)
                ids[i] = c->ids[i];
            if (p) {
                p[3 * i] = c->h_buf[i];
                p[3 * i + 1] = c->h_buf[N + i];
                p[3 * i + 2] = c->h_buf[2 * N + i];
            }
        }
    }
    return N;
}

// More landmarks than the context was created for: build a context of the new capacity, move state, Sigma and options across through
// the host (rare: capacities double), and exchange the two. Handles held by the caller stay valid (the eqf_ctx object is the same).
static int grow_capacity(eqf_ctx* c, int new_cap) {
    int rc = keep_last_gamma(c);
    if (rc)
        return rc;
    const int N = c->N, n = c->n();
    double xi0s[23], Xs[23];
    std::vector<int> ids(std::max(N, 1));
    std::vector<double> q0(3 * (size_t)std::max(N, 1)), Q(5 * (size_t)std::max(N, 1)), S((size_t)n * n);
    rc = eqf_get_state(c, xi0s, Xs, ids.data(), q0.data(), Q.data(), std::max(N, 1));
    if (rc < 0)
        return rc;
    rc = eqf_get_sigma(c, S.data(), n);
    if (rc)
        return rc;
    eqf_ctx* t = nullptr;
    replaces_own_queue = c->own_queue;
    rc = eqf_create(&t, c->device, new_cap, c->chart);
    replaces_own_queue = false;
    if (rc)
        return rc == EQF_E_NO_DEVICE ? rc : EQF_E_CAPACITY; // allocation failure at the new size
    // EVERY option of eqf_set_option (tests/test_gpu_edge_cases.py: test_options_and_counters_survive_capacity_growth walks the enum)
    const int opts[][2] = {{EQF_OPT_SIGMA_FP32, c->opt_f32}, {EQF_OPT_RICCATI_DENSE, c->opt_dense}, {EQF_OPT_CHECK_FINITE, c->opt_check}, {EQF_OPT_SPECULATIVE, c->opt_spec},
                           {EQF_OPT_DOORBELL, c->opt_door}, {EQF_OPT_EARLY_LIFT, c->opt_early}, {EQF_OPT_FUSED_ASSEMBLY, c->opt_fuse_asm}, {EQF_OPT_LOOKAHEAD, c->opt_lookahead},
                           {EQF_OPT_LA_TIMEOUT_US, (int)(c->la_timeout_ticks / 100)}, {EQF_OPT_Z_IN_LOOKAHEAD, c->opt_zb ? (c->opt_zb_large ? 1 : 2) : 0}, {EQF_OPT_LA_SPLIT_ROWS, c->opt_la_split}, {EQF_OPT_LA_HOME, c->opt_la_home}, {EQF_OPT_TILES_PER_WORKGROUP, c->opt_prop_tpw}, {EQF_OPT_GATHER_IN_PROPAGATE, c->opt_gather}, {EQF_OPT_HOLD_NEW_LANDMARKS, c->opt_hold}, {EQF_OPT_EARLY_DOORBELL, c->opt_early_door}, {EQF_OPT_SELECT_ONE_WORKGROUP, c->opt_sel_one}, {EQF_OPT_LIVE_COLUMNS_FIRST, c->opt_live_first}, {EQF_OPT_MEASURE_IN_PROPAGATE, c->opt_measure_prop}, {EQF_OPT_LIFT_WITH_SYRK, c->opt_lift_syrk}, {EQF_OPT_TRACE, c->d_trace ? 1 : 0}, {100, c->opt_timing}};
    for (const auto& o : opts)
        if ((rc = eqf_set_option(t, o[0], o[1])) != 0)
            break;
    if (!rc)
        rc = eqf_set_state(t, xi0s, Xs, ids.data(), q0.data(), Q.data(), N);
    if (!rc)
        rc = eqf_set_sigma(t, S.data(), n);
    if (rc) {
        eqf_destroy(t);
        return rc;
    }
    // what is not state of the filter but of the handle: counters, the last Gamma, the generation of the landmark set
    t->last_gamma = c->last_gamma, t->n_at_update = c->n_at_update;
    t->spec_calls = c->spec_calls, t->spec_queued = c->spec_queued, t->spec_cancelled = c->spec_cancelled, t->spec_backoff = c->spec_backoff, t->spec_backoff_len = c->spec_backoff_len;
    t->nees_lu_fallbacks = c->nees_lu_fallbacks, t->wait_calls = c->wait_calls, t->launch_calls = c->launch_calls, t->wait_seconds = c->wait_seconds, t->launch_seconds = c->launch_seconds;
    t->la_launches = c->la_launches, t->la_fallbacks = c->la_fallbacks, t->la_home_launches = c->la_home_launches, t->la_home_refused = c->la_home_refused, t->la_book_timeouts = c->la_book_timeouts, t->zb_launches = c->zb_launches, t->la_consecutive_stalls = c->la_consecutive_stalls, t->early_rings = c->early_rings;
    t->la_selftest = c->la_selftest < 0 ? -1 : (t->la_selftest != 0 ? t->la_selftest : c->la_selftest); // a failure is never forgotten; otherwise the test that ran on the NEW buffers counts
    t->me_used = c->me_used, t->pred_valid = c->pred_valid, t->pred_cam = c->pred_cam, t->pred_star = c->pred_star;
    t->n_held = c->n_held, t->held_var = c->held_var, t->held_in_memory = c->n_held > 0; // (eqf_get_state above went through enter(): held landmarks are in memory now)
    t->lm_gen = c->lm_gen + 1;
    std::swap(*c, *t);
    eqf_destroy(t);
    return 0;
}

// (all three below: see reshape_pending in eqf_ctx)
static void pend_begin(eqf_ctx* c) {
    if (!c->reshape_pending) {
        c->pend_map.resize(c->dev_N);
        for (int i = 0; i < c->dev_N; ++i)
            c->pend_map[i] = i;
        c->pend_p.clear();
        c->pend_var.clear();
        c->reshape_pending = true;
    }
}
// held landmarks that are not in the device arrays yet become an ordinary pending append (flush_reshape writes them); they stay "held" for the propagation
static void materialise_held(eqf_ctx* c) {
    if (c->n_held == 0 || c->held_in_memory)
        return;
    pend_begin(c);
    for (int t = 0; t < c->n_held; ++t) {
        c->pend_map.push_back(-((int)c->pend_var.size() + 1));
        c->pend_p.insert(c->pend_p.end(), c->h_held + 1 + 3 * t, c->h_held + 4 + 3 * t);
        c->pend_var.push_back(c->held_var);
    }
    c->held_in_memory = true;
}
static int flush_reshape(eqf_ctx* c) {
    if (!c->reshape_pending)
        return 0;
    HP_SCOPE("abi.flush_reshape");
    const int Nnew = (int)c->pend_map.size(), knew = (int)c->pend_var.size();
    bool identity = knew == 0 && Nnew == c->dev_N;
    for (int i = 0; identity && i < Nnew; ++i)
        identity = c->pend_map[i] == i;
    if (identity) {
        c->reshape_pending = false;
        return 0;
    }
    { int _r = join_observer(c); if (_r) return _r; } // landmark kernels of a stand-alone observer call run on the second stream
    bool pure_append = knew > 0 && knew <= APPEND_MAX && Nnew - knew == c->dev_N;
    for (int i = 0; pure_append && i < Nnew; ++i)
        pure_append = c->pend_map[i] == (i < c->dev_N ? i : -(i - c->dev_N + 1));
    if (pure_append) { // nothing moves: the new strips and planes are written in place, the numbers travel as kernel arguments
        AppendArgs aa;
        for (int t = 0; t < knew; ++t) {
            aa.p[t][0] = c->pend_p[3 * t], aa.p[t][1] = c->pend_p[3 * t + 1], aa.p[t][2] = c->pend_p[3 * t + 2];
            aa.var[t] = c->pend_var[t];
        }
        {
            KTimer t(c, KN_MISC);
            const int nn = 21 + 3 * Nnew;
            LAUNCH_TS(c, k_append_inplace, dim3(blocks(nn, 256), 3 * knew + 1), dim3(256), c->stream, c->dev_N, knew, c->Ncap, c->ld, aa, (TS*)c->sigma(), c->d_st[c->stcur], c->d_lm[c->lmcur]);
            HIPCHK(hipGetLastError());
        }
        c->dev_N = Nnew;
        c->reshape_pending = false; // only now: a failed launch above leaves the recorded bookkeeping in place
        return round_sigma(c);
    }
    if (Nnew <= RESHAPE_ARG_MAP && knew <= RESHAPE_ARG_NEW && c->Ncap < 32768) {
        // the record travels as kernel arguments: no copy command in front of the pass (k_reshape_args)
        ReshapeArgs ra;
        for (int i = 0; i < Nnew; ++i)
            ra.map[i] = (short)c->pend_map[i];
        for (int t = 0; t < knew; ++t) {
            ra.p[3 * t] = c->pend_p[3 * t], ra.p[3 * t + 1] = c->pend_p[3 * t + 1], ra.p[3 * t + 2] = c->pend_p[3 * t + 2];
            ra.var[t] = c->pend_var[t];
        }
        const int nnew = 21 + 3 * Nnew;
        {
            KTimer t(c, KN_MISC);
            LAUNCH_TS(c, k_reshape_args, dim3(blocks(nnew, 256), nnew + blocks(Nnew, 256)), dim3(256), c->stream, Nnew, c->Ncap, c->ld, ra, (const TS*)c->d_sigma[c->cur],
                      (TS*)c->d_sigma[1 - c->cur], (const double*)c->d_st[c->stcur], (const double*)c->d_lm[c->lmcur], c->d_st[1 - c->stcur], c->d_lm[1 - c->lmcur]);
            HIPCHK(hipGetLastError());
        }
        c->cur = 1 - c->cur;
        c->lmcur = 1 - c->lmcur;
        c->stcur = 1 - c->stcur;
        c->dev_N = Nnew;
        c->reshape_pending = false;
        return knew ? round_sigma(c) : 0;
    }
    // A slot of the ring is rewritten kRing flushes later. Every host wait of the context (sync_ctx, door_wait) proves that all copies queued
    // before it have run; only a caller that queues more than kRing flushes with no wait in between (alternating eqf_remove_landmarks /
    // eqf_add_landmarks with asynchronous Riccati calls) gets here with a slot possibly still unread: drain once.
    if (c->ring_inflight >= eqf_ctx::kRing) {
        const int r = sync_ctx(c);
        if (r)
            return r;
    }
    ++c->ring_inflight;
    char* slot = c->h_rs_ring + (size_t)(c->ring_pos++ % eqf_ctx::kRing) * c->rs_bytes; // no stream drain: a ring of pinned packets
    const size_t off_p = (sizeof(int) * (size_t)c->Ncap + 15) & ~(size_t)15, off_v = off_p + sizeof(double) * 3 * (size_t)c->Ncap;
    std::memcpy(slot, c->pend_map.data(), sizeof(int) * Nnew);
    if (knew) {
        std::memcpy(slot + off_p, c->pend_p.data(), sizeof(double) * 3 * knew);
        std::memcpy(slot + off_v, c->pend_var.data(), sizeof(double) * knew);
    }
    // compact compactly: only what is used travels (two contiguous ranges when there are new landmarks)
    HIPCHK(hipMemcpyAsync(c->d_rs, slot, knew ? off_v + sizeof(double) * knew : sizeof(int) * std::max(Nnew, 1), hipMemcpyHostToDevice, c->stream));
    const int nnew = 21 + 3 * Nnew;
    {
        KTimer t(c, KN_MISC);
        LAUNCH_TS(c, k_reshape, dim3(blocks(nnew, 256), nnew + blocks(Nnew, 256)), dim3(256), c->stream, Nnew, c->Ncap, c->ld, (const int*)c->d_rs, (const double*)(c->d_rs + off_p),
                  (const double*)(c->d_rs + off_v), (const TS*)c->d_sigma[c->cur], (TS*)c->d_sigma[1 - c->cur], (const double*)c->d_st[c->stcur], (const double*)c->d_lm[c->lmcur],
                  c->d_st[1 - c->stcur], c->d_lm[1 - c->lmcur]);
        HIPCHK(hipGetLastError());
    }
    c->cur = 1 - c->cur;
    c->lmcur = 1 - c->lmcur;
    c->stcur = 1 - c->stcur;
    c->dev_N = Nnew;
    c->reshape_pending = false;
    return knew ? round_sigma(c) : 0; // (EQF_OPT_SIGMA_FP32 = 1 rounds after every store of Sigma: the appended variances)
}

int eqf_add_landmarks(eqf_ctx* c, const int* ids, const double* p, int k, double var) {
    HP_SCOPE("abi.add_landmarks");
    if (!c || k < 0 || (k > 0 && (!ids || !p)))
        return EQF_E_BAD_ARG;
    if (k == 0)
        return 0;
    if (c->n_held > 0)
        return EQF_E_UNSUPPORTED; // held landmarks are the LAST ones of the state until the propagation they wait for
    HIPCHK(hipSetDevice(c->device));
    if (c->N + k > c->Ncap) { // the reference has no cap (VIO_eqf.cpp:225-245 resizes Sigma): grow, at least doubling
        const int rc = grow_capacity(c, std::max(c->N + k, 2 * c->Ncap));
        if (rc)
            return rc;
    }
    pend_begin(c);
    for (int t = 0; t < k; ++t) {
        c->pend_map.push_back(-((int)c->pend_var.size() + 1));
        c->pend_p.insert(c->pend_p.end(), p + 3 * t, p + 3 * t + 3);
        c->pend_var.push_back(var);
    }
    if (c->est_valid) { // the estimate of a fresh landmark is its origin point (Q = identity): the cache follows without asking the device
        const int N = c->N;
        std::vector<double> e(4 * (size_t)(N + k));
        for (int pl = 0; pl < 4; ++pl) {
            std::copy(c->est_cache.begin() + (size_t)pl * N, c->est_cache.begin() + (size_t)(pl + 1) * N, e.begin() + (size_t)pl * (N + k));
            for (int t = 0; t < k; ++t)
                e[(size_t)pl * (N + k) + N + t] = pl < 3 ? p[3 * t + pl] : 0.0;
        }
        c->est_cache.swap(e);
    }
    const bool keep_lookup = c->lookup_gen == c->lm_gen && (int)c->lookup.size() == c->N;
    if (keep_lookup) { // the sorted (id, index) table follows: k new pairs sorted and merged in
        const size_t n0 = c->lookup.size();
        for (int t = 0; t < k; ++t)
            c->lookup.push_back({ids[t], c->N + t});
        std::sort(c->lookup.begin() + n0, c->lookup.end());
        std::inplace_merge(c->lookup.begin(), c->lookup.begin() + n0, c->lookup.end());
    }
    c->ids.insert(c->ids.end(), ids, ids + k);
    c->N += k;
    ++c->lm_gen;
    if (keep_lookup)
        c->lookup_gen = c->lm_gen;
    c->meas_valid = false;
    c->ocov_valid = false; // (h_ocov is indexed by the OLD landmark set: eqf_state_estimate -> add / remove -> eqf_output_cov_all must compute afresh)
    return 0;
}

// eqf_hip.h: landmarks that belong to the time BEHIND the next eqf_propagate_fast
int eqf_hold_supported(const eqf_ctx* c) {
    return (c && c->opt_hold && c->opt_gather && c->opt_fuse_asm && !c->opt_dense && c->chart != EQVIO_COORD_NORMAL && !c->sig32 && !c->opt_check && c->h_held) ? 1 : 0;
}
int eqf_add_landmarks_held(eqf_ctx* c, const int* ids, const double* p, int k, double var) {
    HP_SCOPE("abi.add_landmarks_held");
    if (!c || k < 0 || (k > 0 && (!ids || !p)))
        return EQF_E_BAD_ARG;
    if (k == 0)
        return 0;
    // refused (nothing added: the caller appends them behind the propagation, as the reference does) when the options do not allow it, when the capacity would have to
    // grow, or when landmarks with another variance are already held
    if (!eqf_hold_supported(c) || c->N + k > c->Ncap || (c->n_held > 0 && (var != c->held_var || c->held_in_memory)))
        return EQF_E_UNSUPPORTED;
    if (c->n_held == 0 && c->held_busy) { // a propagation that reads the packet may still be queued (no host wait since): wait for it before the packet is rewritten
        HIPCHK(hipSetDevice(c->device));
        const int r = sync_ctx(c);
        if (r)
            return r;
    }
    c->held_var = var;
    c->h_held[0] = var;
    std::memcpy(c->h_held + 1 + 3 * (size_t)c->n_held, p, sizeof(double) * 3 * k);
    if (c->est_valid) { // the estimate of a fresh landmark is its origin point (Q = identity): the cache follows without asking the device
        const int N = c->N;
        std::vector<double> e(4 * (size_t)(N + k));
        for (int pl = 0; pl < 4; ++pl) {
            std::copy(c->est_cache.begin() + (size_t)pl * N, c->est_cache.begin() + (size_t)(pl + 1) * N, e.begin() + (size_t)pl * (N + k));
            for (int t = 0; t < k; ++t)
                e[(size_t)pl * (N + k) + N + t] = pl < 3 ? p[3 * t + pl] : 0.0;
        }
        c->est_cache.swap(e);
    }
    const bool keep_lookup = c->lookup_gen == c->lm_gen && (int)c->lookup.size() == c->N;
    if (keep_lookup) {
        const size_t n0 = c->lookup.size();
        for (int t = 0; t < k; ++t)
            c->lookup.push_back({ids[t], c->N + t});
        std::sort(c->lookup.begin() + n0, c->lookup.end());
        std::inplace_merge(c->lookup.begin(), c->lookup.begin() + n0, c->lookup.end());
    }
    c->ids.insert(c->ids.end(), ids, ids + k);
    c->N += k;
    c->n_held += k;
    ++c->lm_gen;
    if (keep_lookup)
        c->lookup_gen = c->lm_gen;
    c->meas_valid = false;
    c->ocov_valid = false; // (h_ocov is indexed by the OLD landmark set: eqf_state_estimate -> add / remove -> eqf_output_cov_all must compute afresh)
    return 0;
}
int eqf_own_hardware_queue(eqf_ctx* c) { return c ? (c->own_queue ? 1 : 0) : EQF_E_BAD_ARG; }
int eqf_live_columns_stats(eqf_ctx* c, long* launches, int reset) {
    if (!c)
        return EQF_E_BAD_ARG;
    if (launches)
        *launches = c->live_first_launches;
    if (reset)
        c->live_first_launches = 0;
    return 0;
}
int eqf_hold_stats(eqf_ctx* c, long* launches, int reset) {
    if (!c || !launches)
        return EQF_E_BAD_ARG;
    *launches = c->held_launches;
    if (reset)
        c->held_launches = 0;
    return 0;
}

int eqf_remove_landmarks(eqf_ctx* c, const int* indices, int k) {
    HP_SCOPE("abi.remove_landmarks");
    if (!c || k < 0 || (k > 0 && !indices))
        return EQF_E_BAD_ARG;
    if (k == 0)
        return 0;
    // (scratch vectors of the context: a frame of the shipped configurations makes two of these calls, and their allocations were 2 of its ~20 us between doorbell and next launch)
    std::vector<char>& drop = c->rm_drop;
    drop.assign(c->N, 0);
    for (int t = 0; t < k; ++t) {
        if (indices[t] < 0 || indices[t] >= c->N)
            return EQF_E_BAD_ARG;
        drop[indices[t]] = 1;
    }
    if (c->n_held > 0) { // a removal while landmarks are held: they become an ordinary pending append first; a held landmark that is removed again is not held any more
        materialise_held(c);
        for (int i = c->N - c->n_held, held = c->n_held; i < c->N && held > 0; ++i)
            c->n_held -= drop[i] ? 1 : 0;
        if (c->n_held == 0)
            c->held_in_memory = false;
    }
    pend_begin(c);
    const int N = c->N;
    std::vector<int>&newids = c->rm_ids, &newmap = c->rm_map;
    newids.clear(), newmap.clear();
    newids.reserve(N), newmap.reserve(N);
    bool pending_new = false;
    for (int i = 0; i < N; ++i)
        if (!drop[i]) {
            newids.push_back(c->ids[i]);
            newmap.push_back(c->pend_map[i]);
            pending_new |= c->pend_map[i] < 0;
        }
    // pending new landmarks that were removed again: renumber the remaining ones
    if (pending_new || !c->pend_var.empty()) {
        std::vector<int> renum(c->pend_var.size(), -1);
        std::vector<double> np, nv;
        for (int& mo : newmap)
            if (mo < 0) {
                const int t = -mo - 1;
                renum[t] = (int)nv.size();
                np.insert(np.end(), c->pend_p.begin() + 3 * t, c->pend_p.begin() + 3 * t + 3);
                nv.push_back(c->pend_var[t]);
                mo = -(renum[t] + 1);
            }
        c->pend_p.swap(np);
        c->pend_var.swap(nv);
    }
    c->pend_map.swap(newmap);
    const int Nnew = (int)newids.size();
    if (c->est_valid) {
        std::vector<double>& e = c->rm_est;
        e.resize(4 * (size_t)Nnew);
        for (int pl = 0; pl < 4; ++pl) {
            int w = 0;
            for (int i = 0; i < N; ++i)
                if (!drop[i])
                    e[(size_t)pl * Nnew + w++] = c->est_cache[(size_t)pl * N + i];
        }
        c->est_cache.swap(e);
    }
    const bool keep_lookup = c->lookup_gen == c->lm_gen && (int)c->lookup.size() == N;
    if (keep_lookup) { // the sorted (id, index) table follows: dropped entries out, the others renumbered
        std::vector<int>& renum = c->renum_scratch;
        renum.assign(N, -1);
        for (int i = 0, w = 0; i < N; ++i)
            if (!drop[i])
                renum[i] = w++;
        size_t o = 0;
        for (const auto& e : c->lookup)
            if (renum[e.second] >= 0)
                c->lookup[o++] = {e.first, renum[e.second]};
        c->lookup.resize(o);
    }
    c->ids.swap(newids);
    c->N = Nnew;
    ++c->lm_gen;
    if (keep_lookup)
        c->lookup_gen = c->lm_gen;
    c->meas_valid = false;
    c->ocov_valid = false; // (h_ocov is indexed by the OLD landmark set: eqf_state_estimate -> add / remove -> eqf_output_cov_all must compute afresh)
    return 0;
}

// These are exactly the ids of the measurement that was mapped last (map_measurement: validated ascending, every one with a landmark), the landmark set has not changed
// since, and there is one id per landmark: every landmark is measured and every measured id known.
static bool same_as_mapped(const eqf_ctx* c, const int* ids, int M) {
    return M > 0 && M == c->N && c->map_gen == c->lm_gen && c->map_N == c->N && c->map_all && (int)c->map_ids.size() == M &&
           std::memcmp(ids, c->map_ids.data(), sizeof(int) * M) == 0;
}
int eqf_same_as_mapped(const eqf_ctx* c, const int* ids, int M) { return (c && ids && same_as_mapped(c, ids, M)) ? 1 : 0; }
// VIOFilter::removeOldLandmarks (VIOFilter.cpp:280-302) in one call: the landmarks of the state whose id is not among the (strictly ascending) measured ids leave the state
// (recorded like eqf_remove_landmarks). Their indices, ascending, go to removed_idx (room for the current landmark count). O(N + M): one merge pass against the state's ids
// when these ascend as well, else against the sorted (id, index) table.
int eqf_remove_unmeasured_landmarks(eqf_ctx* c, const int* ids, int M, int* removed_idx, int* n_removed) {
    HP_SCOPE("abi.remove_unmeasured");
    if (!c || M < 0 || (M > 0 && !ids) || !removed_idx || !n_removed)
        return EQF_E_BAD_ARG;
    *n_removed = 0;
    if (same_as_mapped(c, ids, M))
        return 0; // the ids of the last mapped measurement, one per landmark: nobody is lost (the steady frame: 0.1 instead of 1 us in front of the propagation's launch)
    for (int j = 1; j < M; ++j)
        if (ids[j] <= ids[j - 1])
            return EQF_E_BAD_ARG;
    const int N = c->N;
    bool state_ascending = true;
    for (int i = 1; i < N && state_ascending; ++i)
        state_ascending = c->ids[i] > c->ids[i - 1];
    int k = 0;
    if (state_ascending) {
        int q = 0;
        for (int i = 0; i < N; ++i) {
            while (q < M && ids[q] < c->ids[i])
                ++q;
            if (q == M || ids[q] != c->ids[i])
                removed_idx[k++] = i;
        }
    } else {
        ensure_lookup(c);
        int q = 0;
        for (const auto& e : c->lookup) { // ascending ids
            while (q < M && ids[q] < e.first)
                ++q;
            if (q == M || ids[q] != e.first)
                removed_idx[k++] = e.second;
        }
        std::sort(removed_idx, removed_idx + k);
    }
    *n_removed = k;
    return k ? eqf_remove_landmarks(c, removed_idx, k) : 0;
}

// The measured ids (strictly ascending) that have no landmark in the state - VIOFilter::addNewLandmarks' membership test (VIOFilter.cpp:258-278) in one merge pass:
// unknown_j (room for M) receives their positions j in `ids`, ascending; *n_unknown their number.
int eqf_find_unknown_ids(eqf_ctx* c, const int* ids, int M, int* unknown_j, int* n_unknown) {
    HP_SCOPE("abi.find_unknown");
    if (!c || M < 0 || (M > 0 && !ids) || !unknown_j || !n_unknown)
        return EQF_E_BAD_ARG;
    *n_unknown = 0;
    if (same_as_mapped(c, ids, M))
        return 0; // (see eqf_remove_unmeasured_landmarks)
    for (int j = 1; j < M; ++j)
        if (ids[j] <= ids[j - 1])
            return EQF_E_BAD_ARG;
    const int N = c->N;
    bool state_ascending = true;
    for (int i = 1; i < N && state_ascending; ++i)
        state_ascending = c->ids[i] > c->ids[i - 1];
    if (!state_ascending)
        ensure_lookup(c);
    int h = 0, k = 0;
    for (int j = 0; j < M; ++j) {
        while (h < N && (state_ascending ? c->ids[h] : c->lookup[h].first) < ids[j])
            ++h;
        if (h == N || (state_ascending ? c->ids[h] : c->lookup[h].first) != ids[j])
            unknown_j[k++] = j;
    }
    *n_unknown = k;
    return 0;
}

// EQF_OPT_EARLY_DOORBELL: 1 while the last update's lift results (landmark estimates, invalid flags) have not been waited for
int eqf_update_unsettled(const eqf_ctx* c) { return (c && c->unsettled) ? 1 : 0; }
// VIO_eqf::removeInvalidLandmarks (VIO_eqf.cpp:213-223) for a caller that deferred it past an unsettled update: the landmarks the UPDATE's lift flagged (Q.a outside
// (1e-8, 1e8]) leave the state, wherever they sit now. Returns their number (>= 0) or an error (< 0).
int eqf_remove_invalid_at_update(eqf_ctx* c) {
    if (!c)
        return EQF_E_BAD_ARG;
    { int _r = settle_update(c); if (_r) return _r < 0 ? _r : EQF_E_STALLED; }
    if (c->invalid_ids.empty())
        return 0;
    std::vector<int> idx;
    for (const int id : c->invalid_ids) {
        const int i = index_of(c, id);
        if (i >= 0)
            idx.push_back(i);
    }
    c->invalid_ids.clear();
    std::sort(idx.begin(), idx.end());
    if (idx.empty())
        return 0;
    const int rc = eqf_remove_landmarks(c, idx.data(), (int)idx.size());
    return rc ? rc : (int)idx.size();
}

int eqf_remove_invalid_landmarks(eqf_ctx* c) {
    HP_SCOPE("abi.remove_invalid");
    if (!c)
        return EQF_E_BAD_ARG;
    if (c->N == 0)
        return 0;
    HIPCHK(hipSetDevice(c->device));
    int rc = fetch_estimates(c);
    if (rc)
        return rc;
    std::vector<int> bad;
    for (int i = 0; i < c->N; ++i)
        if (c->h_buf[3 * c->N + i] != 0.0)
            bad.push_back(i);
    if (bad.empty())
        return 0;
    rc = eqf_remove_landmarks(c, bad.data(), (int)bad.size());
    return rc ? rc : (int)bad.size();
}

static int riccati_after_assemble(eqf_ctx* c, double dt, const double* Qdiag12, const double* Pdiag8, const ObsSteps* obs = nullptr, int obs_k = 0, bool fused = false,
                                  const GatherArgs* gather = nullptr);
int eqf_integrate_riccati_fast(eqf_ctx* c, const double* imu13, double dt, const double* Qdiag12, const double* Pdiag8) {
    if (c && c->n_held > 0)
        return EQF_E_UNSUPPORTED; // landmarks held for eqf_propagate_fast (eqf_add_landmarks_held): only that call knows to leave them alone
    if (!c || !imu13 || !Qdiag12 || !Pdiag8)
        return EQF_E_BAD_ARG;
    { int _e = enter(c); if (_e) return _e; }
    int rc = upload_common(c, imu13);
    if (rc)
        return rc;
    rc = launch_assemble(c);
    if (rc)
        return rc;
    return riccati_after_assemble(c, dt, Qdiag12, Pdiag8);
}
// Sigma' = F Sigma F^T + dt (B Q B^T + P) once A_l / B_l are assembled (arrow form, or the dense GEMM pair)
// obs != nullptr (arrow form only): obs_k observer steps for the landmarks ride along as extra blocks of k_propagate_main
// Normal chart (coordinateSuite/normal.cpp:37-45): Sigma <- T Sigma T^T with T = M (dir > 0, plus dt P on the diagonal) or M^-1 (dir < 0),
// out of place into the other Sigma buffer. See k_congruence_normal.
static int normal_congruence(eqf_ctx* c, int dir, double dt, const double* Pdiag8) {
    NormalM nm{};
    nm.dir = dir;
    nm.v0[0] = c->xi0.vel.x, nm.v0[1] = c->xi0.vel.y, nm.v0[2] = c->xi0.vel.z;
    const M6 Ad = se3_Adjoint(pose_inv(c->xi0.cam));
    for (int r = 0; r < 6; ++r)
        for (int q = 0; q < 6; ++q)
            nm.Ad[6 * r + q] = Ad.a[6 * r + q];
    nm.addP = (dir > 0 && Pdiag8) ? 1 : 0;
    for (int k = 0; k < 8; ++k)
        nm.dtP[k] = Pdiag8 ? dt * Pdiag8[k] : 0.0;
    const int n = c->n();
    LAUNCH_TS(c, k_congruence_normal, dim3(blocks(n, 256), n), dim3(256), c->stream, n, c->Ncap, c->ld, nm, c->q0(), (const TS*)c->d_sigma[c->cur], (TS*)c->d_sigma[1 - c->cur]);
    HIPCHK(hipGetLastError());
    c->cur = 1 - c->cur;
    return 0;
}
static int riccati_after_assemble(eqf_ctx* c, double dt, const double* Qdiag12, const double* Pdiag8, const ObsSteps* obs, int obs_k, bool fused, const GatherArgs* gather) {
    int rc = 0;
    c->me_valid = false; // whatever propagates Sigma (any mode) leaves output blocks of an earlier propagation behind
    static const ObsSteps kNoSteps{};
    const bool normal = c->chart == EQVIO_COORD_NORMAL;
    static const double kZero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double* Pfull = Pdiag8;
    if (normal) { // Euclidean propagation of M^-1 Sigma M^-T without process noise; M ( . ) M^T + dt P follows below
        rc = normal_congruence(c, -1, dt, nullptr);
        if (rc)
            return rc;
        Pdiag8 = kZero8;
    }
    RiccatiArgs ra;
    ra.dt = dt;
    std::memcpy(ra.Qd, Qdiag12, sizeof(ra.Qd));
    std::memcpy(ra.Pd, Pdiag8, sizeof(ra.Pd));
    const int N = c->N, n = c->n();
    double* Sin = c->d_sigma[c->cur];
    double* Sout = c->d_sigma[1 - c->cur];
    if (!c->opt_dense) {
        const int nT = blocks(N, PT);
        const bool sym = nT > 16;           // more tiles than fit the chip in one round: lower triangle only, mirrored (k_propagate_main)
        const int nObs = (obs && obs_k > 0) ? blocks(N, PROP_T) : 0;
        // Lower triangle: tpw consecutive tiles of a block row per workgroup, the smallest tpw with which every workgroup of the launch is resident at once (one of
        // these workgroups per compute unit: a second ROUND costs a whole tile chain, a second tile of the same block row about a third of one; k_propagate_main)
        int tpw = 1, nTiles = sym ? nT * (nT + 1) / 2 : nT * nT;
        if (sym && c->opt_prop_tpw) {
            auto wgs = [&](int t) {
                int w = 0;
                for (int r = 0; r < nT; ++r)
                    w += (r + t) / t;
                return w;
            };
            while (tpw < 8 && wgs(tpw) + 2 + nObs > c->cu_count)
                ++tpw;
            if (c->opt_prop_tpw > 1)
                tpw = c->opt_prop_tpw; // forced (A/B)
            nTiles = wgs(tpw);
        }
        StageArgs sg{};
        if (c->stage_pending) { // one more block copies the staged measurement from the pinned packet to HBM
            sg.M = c->staged_M;
            sg.y_h = c->h_y, sg.ylm_h = c->h_ylm, sg.idx_h = c->h_lmidx;
            sg.y_d = c->d_meas, sg.ylm_d = c->d_meas + 2 * (size_t)c->Ncap, sg.idx_d = c->d_meas_idx;
            c->stage_pending = false;
            c->staged_valid = true;
            c->busy_meas = true;
        }
        // fused assembly (eqf_propagate_fast): every workgroup assembles the rows of A / B it needs from the current Q and c->ck; the
        // observer blocks then write the other landmark buffer, which becomes the current one
        FuseArgs fa{};
        if (fused) {
            fa.on = 1;
            fa.chart = c->chart;
            double* other = c->d_lm[1 - c->lmcur];
            fa.Qqo = other, fa.Qao = other + 4 * (size_t)c->Ncap;
            fa.ck = c->ck;
        }
        // EQF_OPT_MEASURE_IN_PROPAGATE: the observer blocks evaluate the output blocks of the staged measurement with the camera / output choice of the last update call
        MeasEval me{};
        if (c->opt_measure_prop && fused && nObs && c->obs_one_chunk && sg.M > 0 && c->pred_valid && !c->sig32 && !c->opt_f32 && c->opt_zb && !c->opt_check) {
            me.on = 1, me.star = c->pred_star, me.Mcap = c->Ncap, me.cam = c->pred_cam;
            me.ylm = c->h_ylm, me.C = c->d_C, me.ytil = c->d_ytil, me.lmidx_dev = c->d_lmidx;
            c->me_valid = true, c->me_cam = c->pred_cam, c->me_star = c->pred_star, c->me_M = sg.M, c->me_gen = c->staged_gen;
        }
        // EQF_OPT_GATHER_IN_PROPAGATE: the landmarks removed since the last kernel leave inside this launch (eqf_propagate_fast decided; fused assembly + observer blocks)
        GatherArgs ga{};
        if (gather) {
            if (!(fused && nObs))
                return EQF_E_UNSUPPORTED; // (eqf_propagate_fast checks the same conditions before it asks for this)
            ga = *gather;
            ga.st_in = c->d_st[c->stcur], ga.st_out = ga.n ? c->d_st[1 - c->stcur] : c->d_st[c->stcur]; // (nothing removed: the planes stay where they are, only held landmarks are written)
        }
        ga.nprop = N - c->n_held; // (held landmarks that an ordinary pass appended: read from memory, passed through untouched)
        if (c->n_held > 0 && !(fused && nObs))
            return EQF_E_UNSUPPORTED;
        KTimer t(c, KN_PROP_MAIN);
        auto launch = [&](auto kern, auto* sin, auto* sout) {
            hipLaunchKernelGGL(kern, dim3(nTiles + 1 + nObs + (sg.M ? 1 : 0)), dim3(PROP_T), 0, c->stream, N, c->Ncap, c->ld, ra, c->d_common, sin, sout, c->d_Al, c->d_Bl, nT, tpw,
                               nObs ? *obs : kNoSteps, nObs ? obs_k : 0, c->q0(), c->Qq(), c->Qa(), nObs, sg, trace_slot(c, TR_PROPAGATE), fa, me, ga);
        };
#define PROP_LAUNCH(TS_, F_) \
    do { \
        if (sym) \
            launch(k_propagate_main<TS_, F_, true>, (const TS_*)Sin, (TS_*)Sout); \
        else \
            launch(k_propagate_main<TS_, F_, false>, (const TS_*)Sin, (TS_*)Sout); \
    } while (0)
        if (c->sig32) {
            if (fused)
                PROP_LAUNCH(float, true);
            else
                PROP_LAUNCH(float, false);
        } else if (fused && (ga.n > 0 || ga.nprop < N)) { // landmarks leave / are created / pass through inside this launch: the instantiation that knows how
            if (sym)
                launch(k_propagate_main<double, true, true, true>, (const double*)Sin, (double*)Sout);
            else
                launch(k_propagate_main<double, true, false, true>, (const double*)Sin, (double*)Sout);
        } else {
            if (fused)
                PROP_LAUNCH(double, true);
            else
                PROP_LAUNCH(double, false);
        }
#undef PROP_LAUNCH
        HIPCHK(hipGetLastError());
        if (fused && nObs)
            c->lmcur = 1 - c->lmcur;
        if (gather) { // what flush_reshape does behind its pass
            if (gather->n) {
                c->stcur = 1 - c->stcur;
                ++c->gather_launches;
            }
            if (gather->held) {
                ++c->held_launches;
                c->held_busy = true;
            }
            c->dev_N = N;
            c->reshape_pending = false;
        }
        c->n_held = 0, c->held_in_memory = false; // they are ordinary landmarks from here on
    } else {
        // dense: F materialised, tmp = F Sigma (= (Sigma F^T)^T, Sigma symmetric), Sigma' = tmp F^T + noise
        const size_t bytes = sizeof(double) * (size_t)c->ld * c->ncap;
        if (!c->d_F)
            HIPCHK(hipMalloc(&c->d_F, bytes));
        if (!c->d_tmp)
            HIPCHK(hipMalloc(&c->d_tmp, bytes));
        KTimer t(c, KN_DENSE_GEMM);
        hipLaunchKernelGGL(k_build_F, dim3(blocks(n, 256), n), dim3(256), 0, c->stream, N, c->Ncap, n, c->ld, dt, c->d_common, c->d_Al, c->d_F);
        HIPCHK(hipGetLastError());
        // tmp[i][j] = sum_k F[i][k] Sigma[j][k]
        hipLaunchKernelGGL(k_gemm_nt, dim3(blocks(n, 32), blocks(n, 32)), dim3(256), 0, c->stream, n, n, n, c->d_F, c->ld, Sin, c->ld, c->d_tmp, c->ld);
        HIPCHK(hipGetLastError());
        // Sout[i][j] = sum_k tmp[i][k] F[j][k]
        hipLaunchKernelGGL(k_gemm_nt, dim3(blocks(n, 32), blocks(n, 32)), dim3(256), 0, c->stream, n, n, n, c->d_tmp, c->ld, c->d_F, c->ld, Sout, c->ld);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_add_noise, dim3(blocks(n, 256), n), dim3(256), 0, c->stream, N, c->Ncap, n, c->ld, ra, c->d_common, c->d_Bl, Sout);
        HIPCHK(hipGetLastError());
    }
    c->cur = 1 - c->cur;
    if (normal) {
        rc = normal_congruence(c, +1, dt, Pfull);
        if (rc)
            return rc;
    }
    { int _r = round_sigma(c); if (_r) return _r; }
    if (c->opt_check) {
        LAUNCH_TS(c, k_check_finite, dim3(blocks(n, 256), n), dim3(256), c->stream, n, c->ld, (const TS*)c->sigma(), c->d_flags);
        HIPCHK(hipGetLastError());
        rc = read_flags(c);
        if (rc)
            return rc;
        if (c->h_flags[1])
            return EQF_E_NONFINITE;
    }
    return 0;
}

int eqf_integrate_riccati_accurate(eqf_ctx* c, const double* imu13, double dt, const double* Qdiag12, const double* Pdiag8) {
    if (c && c->n_held > 0)
        return EQF_E_UNSUPPORTED; // landmarks held for eqf_propagate_fast (eqf_add_landmarks_held): only that call knows to leave them alone
    if (!c || !imu13 || !Qdiag12 || !Pdiag8 || !(dt > 0.0))
        return EQF_E_BAD_ARG;
    if (c->sig32)
        return EQF_E_UNSUPPORTED; // the float store exists for the structured fast path only
    { int _e = enter(c); if (_e) return _e; }
    const size_t bytes = sizeof(double) * (size_t)c->ld * c->ncap;
    if (!c->d_Ebuf) {
        HIPCHK(hipMalloc(&c->d_Ebuf, sizeof(double) * (EXPM_SMAX + 2) * 21 * 33));
        HIPCHK(hipMalloc(&c->d_Yl, sizeof(double) * 99 * (size_t)c->Ncap));
        HIPCHK(hipMalloc(&c->d_Fl, sizeof(double) * 9 * (size_t)c->Ncap));
        HIPCHK(hipMalloc(&c->d_PhiB, sizeof(double) * (size_t)c->ld * 12));
        HIPCHK(hipMalloc(&c->d_expinfo, sizeof(int) * 4));
    }
    if (!c->d_F)
        HIPCHK(hipMalloc(&c->d_F, bytes));
    if (!c->d_tmp)
        HIPCHK(hipMalloc(&c->d_tmp, bytes));
    int rc = upload_common(c, imu13);
    if (rc)
        return rc;
    rc = launch_assemble(c);
    if (rc)
        return rc;
    const bool normal = c->chart == EQVIO_COORD_NORMAL;
    static const double kZero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (normal) { // expm(dt [[A_n, B_n],[0,0]]) = blkdiag(M, I) expm(dt [[A_e, B_e],[0,0]]) blkdiag(M^-1, I): the same two congruences
        rc = normal_congruence(c, -1, dt, nullptr);
        if (rc)
            return rc;
    }
    RiccatiArgs ra;
    ra.dt = dt;
    std::memcpy(ra.Qd, Qdiag12, sizeof(ra.Qd));
    std::memcpy(ra.Pd, normal ? kZero8 : Pdiag8, sizeof(ra.Pd));
    const int N = c->N, n = c->n();
    double* Sin = c->d_sigma[c->cur];
    double* Sout = c->d_sigma[1 - c->cur];
    KTimer t(c, KN_DENSE_GEMM);
    hipLaunchKernelGGL(k_expm_sensor, dim3(1), dim3(256), 0, c->stream, N, c->Ncap, dt, c->d_common, c->d_Al, c->d_Bl, c->d_Ebuf, c->d_expinfo);
    HIPCHK(hipGetLastError());
    if (N > 0) {
        hipLaunchKernelGGL(k_expm_landmarks, dim3(blocks(N, 4)), dim3(256), 0, c->stream, N, c->Ncap, dt, c->d_Al, c->d_Bl, c->d_Ebuf, c->d_expinfo, c->d_Yl, c->d_Fl);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_build_phi, dim3(blocks(n, 256), n + 12), dim3(256), 0, c->stream, N, c->Ncap, n, c->ld, c->d_Ebuf, c->d_expinfo, c->d_Yl, c->d_Fl, c->d_F, c->d_PhiB);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_gemm_nt, dim3(blocks(n, 32), blocks(n, 32)), dim3(256), 0, c->stream, n, n, n, c->d_F, c->ld, Sin, c->ld, c->d_tmp, c->ld);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_gemm_nt, dim3(blocks(n, 32), blocks(n, 32)), dim3(256), 0, c->stream, n, n, n, c->d_tmp, c->ld, c->d_F, c->ld, Sout, c->ld);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_add_noise_dense, dim3(blocks(n, 256), n), dim3(256), 0, c->stream, n, c->ld, c->ld, ra, c->d_PhiB, Sout);
    HIPCHK(hipGetLastError());
    c->cur = 1 - c->cur;
    if (normal) {
        rc = normal_congruence(c, +1, dt, Pdiag8);
        if (rc)
            return rc;
    }
    return round_sigma(c);
}

// ---- integrateRiccatiStateDiscrete (VIO_eqf.cpp:93-103, EqFMatrices.cpp:24-41) -----------------------------------------------------------
static GroupSensor lift_discrete_sensor(const SensorState& xs, const double* imu, double dt) { // liftVelocityDiscrete, sensor part (VIOGroup.cpp:229-250)
    const V3 gyr = v3(imu[1], imu[2], imu[3]) - xs.bgyr;
    const V3 acc = v3(imu[4], imu[5], imu[6]) - xs.bacc;
    const V3 gdir = q_rot(q_inv(xs.pose.R), v3(0, 0, 1));
    GroupSensor L;
    L.bgyr = dt * v3(imu[7], imu[8], imu[9]);
    L.bacc = dt * v3(imu[10], imu[11], imu[12]);
    L.A.R = so3_exp(dt * gyr);
    const V3 x = dt * q_rot(xs.pose.R, xs.vel) + (0.5 * dt * dt) * (q_rot(xs.pose.R, acc) + v3(0, 0, -kGravity));
    L.A.x = q_rot(q_inv(xs.pose.R), x);
    L.B = pose_mul(pose_mul(pose_inv(xs.cam), L.A), xs.cam);
    L.w = xs.vel - (xs.vel + dt * (acc - kGravity * gdir));
    return L;
}
static GroupSensor group_inv(const GroupSensor& X) { // VIOGroup::inverse, sensor part (VIOGroup.cpp:108-120)
    GroupSensor r;
    r.bgyr = -X.bgyr;
    r.bacc = -X.bacc;
    r.A = pose_inv(X.A);
    r.B = pose_inv(X.B);
    r.w = -q_rot(q_inv(X.A.R), X.w);
    return r;
}
int eqf_integrate_riccati_discrete(eqf_ctx* c, const double* imu13, double dt, const double* Qdiag12, const double* Pdiag8) {
    if (c && c->n_held > 0)
        return EQF_E_UNSUPPORTED; // landmarks held for eqf_propagate_fast (eqf_add_landmarks_held): only that call knows to leave them alone
    if (!c || !imu13 || !Qdiag12 || !Pdiag8 || !(dt > 0.0))
        return EQF_E_BAD_ARG;
    if (c->sig32)
        return EQF_E_UNSUPPORTED; // float store: structured fast path only
    { int _e = enter(c); if (_e) return _e; }
    const size_t bytes = sizeof(double) * (size_t)c->ld * c->ncap;
    if (!c->d_F)
        HIPCHK(hipMalloc(&c->d_F, bytes));
    if (!c->d_tmp)
        HIPCHK(hipMalloc(&c->d_tmp, bytes));
    int rc = upload_common(c, imu13); // B_t at the current X (inputMatrixB), for dt (B Q B^T + P)
    if (rc)
        return rc;
    rc = launch_assemble(c, false); // "the last reader of Q" is k_discrete_A below, not k_assemble_AB: the event is recorded there
    if (rc)
        return rc;
    // Normal chart: A_d,n = M A_d,e M^-1 and B_n = M B_e at the origin (chain rule through the change of coordinates), so the Euclidean
    // discrete propagation sits between the same two congruences as the continuous ones (riccati_after_assemble); the reference differentiates
    // a0Discrete numerically in normal coordinates, which agrees to the differencing's own 1e-9
    const bool normal = c->chart == EQVIO_COORD_NORMAL;
    static const double kZero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (normal) {
        rc = normal_congruence(c, -1, dt, nullptr);
        if (rc)
            return rc;
    }
    // sensor-level part of a0Discrete: nominal + 21 x (+h, -h)
    const double h = std::cbrt(2.220446049250313e-16);
    const SensorState xhat = sensor_action(c->X, c->xi0);
    const GroupSensor Lh = lift_discrete_sensor(xhat, imu13, dt), Lh_inv = group_inv(Lh), Xinv = group_inv(c->X);
    DiscreteAArgs da;
    da.h = h;
    da.cpc[0] = pose_mul(pose_mul(pose_inv(xhat.cam), pose_inv(Lh.A)), xhat.cam);
    double Ass[21 * 21]; // column-major
    double col[2][21];
    for (int j = 0; j < 21; ++j) {
        for (int sgn = 0; sgn < 2; ++sgn) {
            double e[21] = {0};
            e[j] = sgn == 0 ? h : -h;
            // xi_e = sensorChart_std.inv(e, xi0) (VIOState.cpp:114-121)
            SensorState se;
            se.bgyr = c->xi0.bgyr + v3(e[0], e[1], e[2]);
            se.bacc = c->xi0.bacc + v3(e[3], e[4], e[5]);
            se.pose = pose_mul(c->xi0.pose, se3_exp(v3(e[6], e[7], e[8]), v3(e[9], e[10], e[11])));
            se.vel = c->xi0.vel + v3(e[12], e[13], e[14]);
            se.cam = pose_mul(c->xi0.cam, se3_exp(v3(e[15], e[16], e[17]), v3(e[18], e[19], e[20])));
            const SensorState xs = sensor_action(c->X, se);
            const GroupSensor L = lift_discrete_sensor(xs, imu13, dt);
            da.cpc[1 + 2 * j + sgn] = pose_mul(pose_mul(pose_inv(xs.cam), pose_inv(L.A)), xs.cam);
            const GroupSensor G = group_mul(group_mul(c->X, group_mul(L, Lh_inv)), Xinv);
            const SensorState s1 = sensor_action(G, se);
            V3 om, tr, omc, trc;
            se3_log(pose_mul(pose_inv(c->xi0.pose), s1.pose), om, tr);
            se3_log(pose_mul(pose_inv(c->xi0.cam), s1.cam), omc, trc);
            const V3 parts[7] = {s1.bgyr - c->xi0.bgyr, s1.bacc - c->xi0.bacc, om, tr, s1.vel - c->xi0.vel, omc, trc};
            for (int b = 0; b < 7; ++b)
                pack_v3(parts[b], col[sgn] + 3 * b);
        }
        for (int r = 0; r < 21; ++r)
            Ass[r + 21 * j] = (col[0][r] - col[1][r]) / (2.0 * h);
    }
    const int N = c->N, n = c->n();
    { int _r = sync_ctx(c); if (_r) return _r; } // h_buf staging
    std::memcpy(c->h_buf, Ass, sizeof(Ass));
    HIPCHK(hipMemsetAsync(c->d_F, 0, bytes, c->stream));
    HIPCHK(hipMemcpy2DAsync(c->d_F, sizeof(double) * c->ld, c->h_buf, sizeof(double) * 21, sizeof(double) * 21, 21, hipMemcpyHostToDevice, c->stream));
    if (N > 0) {
        hipLaunchKernelGGL(k_discrete_A, dim3(blocks(N, 64)), dim3(64), 0, c->stream, N, c->Ncap, c->ld, normal ? (int)EQVIO_COORD_EUCLIDEAN : c->chart, da, c->q0(), c->Qq(), c->Qa(), c->d_F);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev_early, c->stream)); // an observer call that follows may overwrite Q from here on (observer_launch)
    c->ev_assembled_early = true;
    RiccatiArgs ra;
    ra.dt = dt;
    std::memcpy(ra.Qd, Qdiag12, sizeof(ra.Qd));
    std::memcpy(ra.Pd, normal ? kZero8 : Pdiag8, sizeof(ra.Pd));
    double* Sin = c->d_sigma[c->cur];
    double* Sout = c->d_sigma[1 - c->cur];
    KTimer t(c, KN_DENSE_GEMM);
    hipLaunchKernelGGL(k_gemm_nt, dim3(blocks(n, 32), blocks(n, 32)), dim3(256), 0, c->stream, n, n, n, c->d_F, c->ld, Sin, c->ld, c->d_tmp, c->ld);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_gemm_nt, dim3(blocks(n, 32), blocks(n, 32)), dim3(256), 0, c->stream, n, n, n, c->d_tmp, c->ld, c->d_F, c->ld, Sout, c->ld);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_add_noise, dim3(blocks(n, 256), n), dim3(256), 0, c->stream, N, c->Ncap, n, c->ld, ra, c->d_common, c->d_Bl, Sout);
    HIPCHK(hipGetLastError());
    c->cur = 1 - c->cur;
    if (normal) {
        rc = normal_congruence(c, +1, dt, Pdiag8);
        if (rc)
            return rc;
    }
    return round_sigma(c);
}

// Host half of integrateObserverState for `chunk` consecutive IMU samples: the sensor-level group element Lambda of every
// step (VIOGroup.cpp:190-271), applied to the host-authoritative X right away, and the per-step terms the landmark kernel needs.
static void observer_host_steps(eqf_ctx* c, const double* imu13_k, const double* dt_k, int chunk, int discreteLift, ObsSteps& steps_arg) {
    for (int s = 0; s < chunk; ++s) {
        const double* imu = imu13_k + 13 * s;
        const double dt = dt_k[s];
        // stateEstimate() sensor part and the lift (VIOGroup.cpp:190-271)
        const SensorState xh = sensor_action(c->X, c->xi0);
        const V3 gyr = v3(imu[1], imu[2], imu[3]) - xh.bgyr;
        const V3 acc = v3(imu[4], imu[5], imu[6]) - xh.bacc;
        const V3 gbv = v3(imu[7], imu[8], imu[9]), abv = v3(imu[10], imu[11], imu[12]);
        const V3 gdir = q_rot(q_inv(xh.pose.R), v3(0, 0, 1));
        GroupSensor L;
        ObsStep& st = steps_arg.s[s];
        st.discrete = discreteLift ? 1 : 0;
        st.dt = dt;
        if (discreteLift) {
            L.bgyr = dt * gbv;
            L.bacc = dt * abv;
            L.A.R = so3_exp(dt * gyr);
            V3 x = dt * q_rot(xh.pose.R, xh.vel) + (0.5 * dt * dt) * (q_rot(xh.pose.R, acc) + v3(0, 0, -kGravity));
            L.A.x = q_rot(q_inv(xh.pose.R), x);
            L.B = pose_mul(pose_mul(pose_inv(xh.cam), L.A), xh.cam);
            const V3 bodyVelDiff = acc - kGravity * gdir;
            L.w = xh.vel - (xh.vel + dt * bodyVelDiff);
            st.Tinv = pose_mul(pose_mul(pose_inv(xh.cam), pose_inv(L.A)), xh.cam);
            st.omC = v3(0, 0, 0);
            st.vC = v3(0, 0, 0);
        } else {
            // VIOExp(dt * liftVelocity) (VIOGroup.cpp:190-227, 273-290)
            const V6 U_A{gyr, xh.vel};
            const V6 U_B = Ad_apply(pose_inv(xh.cam), U_A);
            const V3 u_w = -acc + kGravity * gdir;
            L.bgyr = dt * gbv;
            L.bacc = dt * abv;
            const M3 V = so3_V(dt * U_A.w);
            L.A = Pose{so3_exp(dt * U_A.w), V * (dt * U_A.v)};
            L.w = V * (dt * u_w);
            L.B = se3_exp(dt * U_B.w, dt * U_B.v);
            st.Tinv = pose_identity();
            st.omC = U_B.w;
            st.vC = U_B.v;
        }
        c->X = group_mul(c->X, L);
    }
}
// Device half: the landmark part runs on the second stream. It only has to wait for the last kernel that READS Q on the main
// stream (k_assemble_AB of a preceding Riccati call: ev_early, else everything queued so far), so it overlaps the Sigma
// propagation kernels (they touch Sigma / Al / Bl only).
static int observer_launch(eqf_ctx* c, const ObsSteps& steps_arg, int chunk) {
    if (c->N == 0)
        return 0;
    if (!c->stream2)
        HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    if (c->obs_pending)
        HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_observer, 0));
    if (!c->ev_assembled_early)
        HIPCHK(hipEventRecord(c->ev_assembled, c->stream)); // everything queued so far on the main stream
    HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_assembled_early ? c->ev_early : c->ev_assembled, 0));
    c->ev_assembled_early = false;
    hipLaunchKernelGGL(k_observer, dim3(blocks(c->N, 64)), dim3(64), 0, c->stream2, steps_arg, c->N, c->Ncap, chunk, c->q0(), c->Qq(), c->Qa());
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev_observer, c->stream2));
    c->obs_pending = true;
    return 0;
}
int eqf_integrate_observer(eqf_ctx* c, const double* imu13_k, const double* dt_k, int k, int discreteLift) {
    if (c && c->n_held > 0)
        return EQF_E_UNSUPPORTED; // landmarks held for eqf_propagate_fast (eqf_add_landmarks_held): only that call knows to leave them alone
    if (!c || k < 0 || (k > 0 && (!imu13_k || !dt_k)))
        return EQF_E_BAD_ARG;
    if (k == 0)
        return 0;
    { int _e = enter(c); if (_e) return _e; }
    c->est_valid = false, ++c->est_epoch;
    c->meas_valid = false;
    c->me_valid = false; // output blocks the propagation kernel evaluated belong to the Q_i from before these steps
    int done = 0;
    while (done < k) {
        const int chunk = std::min(k - done, eqf_ctx::kMaxSteps);
        ObsSteps steps_arg;
        observer_host_steps(c, imu13_k + 13 * done, dt_k + done, chunk, discreteLift, steps_arg);
        const int rc = observer_launch(c, steps_arg, chunk);
        if (rc)
            return rc;
        done += chunk;
    }
    return 0;
}

// See include/eqf_hip.h: fast Riccati + all observer steps of one frame, two kernels on one stream.
static int stage_prepare(eqf_ctx* c);
int eqf_propagate_fast(eqf_ctx* c, const double* imu13_mean, double dt_total, const double* Qdiag12, const double* Pdiag8, const double* imu13_k, const double* dt_k,
                       int k, int discreteLift) {
    HP_SCOPE("abi.propagate_fast");
    if (!c || !imu13_mean || !Qdiag12 || !Pdiag8 || k < 0 || (k > 0 && (!imu13_k || !dt_k)))
        return EQF_E_BAD_ARG;
    // EQF_OPT_GATHER_IN_PROPAGATE: a pending record of removals only (the frame's lost landmarks, the last frame's discarded outliers) is applied by the propagation kernel
    // itself instead of a compaction pass in front of it - when this call takes the fused kernel with observer blocks and nothing was appended.
    // eqf_add_landmarks_held: the held landmarks (the last n_held of the state) are created by the same kernel (GatherArgs::held) unless an ordinary pass has appended them.
    GatherArgs gather{};
    bool use_gather = false;
    const bool fused_kernel = c->opt_fuse_asm && !c->opt_dense && !c->sig32 && c->chart != EQVIO_COORD_NORMAL && k > 0 && c->N > 0; // (more steps than one chunk holds: the further chunks are k_observer launches on the new buffers, for the propagated landmarks only)
    if (c->n_held > 0 && !fused_kernel)
        return EQF_E_UNSUPPORTED; // (eqf_add_landmarks_held checked the options: one was changed in between, or there are no observer steps to ride along)
    const int Nprop = c->N - c->n_held;
    const bool hold_here = c->n_held > 0 && !c->held_in_memory;
    if (fused_kernel && c->opt_gather && (c->reshape_pending || hold_here) && c->pend_var.empty() && c->dev_N <= GATHER_MAXN &&
        (!c->reshape_pending || ((int)c->pend_map.size() == Nprop && c->dev_N >= Nprop))) {
        bool ok = true;
        if (c->reshape_pending) {
            int prev = -1;
            for (int i = 0; i < Nprop && ok; ++i) {
                const int o = c->pend_map[i];
                ok = o > prev && o < c->dev_N; // survivors keep their order
                if (ok)
                    gather.surv[o >> 6] |= 1ull << (o & 63);
                prev = o;
            }
            gather.n = c->dev_N - Nprop;
        } else
            ok = c->dev_N == Nprop;
        if (ok && (gather.n > 0 || hold_here)) {
            use_gather = true;
            if (hold_here)
                gather.held = c->h_held;
        }
    }
    if (use_gather) { // enter() without the flush
        HIPCHK(hipSetDevice(c->device));
        c->ocov_valid = false;
    } else {
        int _e = enter(c, false); // (an unsettled update is settled behind this call's launch: nothing it queues or computes needs the lift's results)
        if (_e)
            return _e;
    }
    ++c->trace_frame; // EQF_OPT_TRACE: a frame starts here
    host_stamp(c, TH_PROP_ENTRY);
    // 1. A / B terms at the CURRENT X (before the observer steps move it): integrateRiccatiStateFast uses X as it is
    int rc;
    {
        HP_SCOPE("pf.upload_common");
        rc = upload_common(c, imu13_mean);
    }
    if (rc)
        return rc;
    // 2. the assembly kernel goes out first (its terms, c->ck, are fixed now): the host part of the observer steps below then
    //    overlaps it instead of delaying it
    // (with fused assembly there is no such launch: the propagation kernel assembles what it needs itself)
    const bool fuse = c->opt_fuse_asm && !c->opt_dense;
    rc = fuse ? join_observer(c) : launch_assemble(c, false);
    if (rc)
        return rc;
    host_stamp(c, TH_ASSEMBLE_OUT);
    {
        HP_SCOPE("pf.stage_prepare");
        rc = stage_prepare(c);
    }
    if (rc)
        return rc;
    // 3. all observer steps on the host (X advances); chunks of kMaxSteps keep the kernel-argument packet small
    std::vector<ObsSteps> chunks;
    std::vector<int> counts;
    HP_SCOPE("pf.observer_steps_and_launch");
    for (int done = 0; done < k;) {
        const int chunk = std::min(k - done, eqf_ctx::kMaxSteps);
        chunks.emplace_back();
        observer_host_steps(c, imu13_k + 13 * done, dt_k + done, chunk, discreteLift, chunks.back());
        counts.push_back(chunk);
        done += chunk;
    }
    if (k > 0) {
        c->est_valid = false, ++c->est_epoch;
        c->meas_valid = false;
    }
    // 4. Arrow form: the first chunk of observer steps rides along as extra blocks of the Sigma propagation kernel (it does
    //    not touch Q; the assembly before it has read Q, the statistics after it want the new Q): two launches on ONE stream.
    //    Further chunks (k > 24) and the dense mode use the observer kernel, in stream order.
    const bool ride = !c->opt_dense && !chunks.empty() && c->N > 0;
    c->obs_one_chunk = chunks.size() == 1; // (the group elements are final when the propagation kernel's observer blocks are done)
    {
        HP_SCOPE("pf.launch");
        if (use_gather && !(ride && fuse)) { // (cannot happen: the conditions above are the ones of `ride` and `fuse`) - apply the record the ordinary way
            materialise_held(c);
            rc = flush_reshape(c);
            if (rc)
                return rc;
            use_gather = false;
        }
        rc = riccati_after_assemble(c, dt_total, Qdiag12, Pdiag8, ride ? &chunks[0] : nullptr, ride ? counts[0] : 0, fuse, use_gather ? &gather : nullptr);
    }
    if (rc)
        return rc;
    host_stamp(c, TH_PROP_OUT);
    for (size_t q = ride ? 1 : 0; q < chunks.size() && Nprop > 0; ++q) { // (held landmarks are the last ones: the observer kernel does not reach them)
        c->ev_assembled_early = false;
        hipLaunchKernelGGL(k_observer, dim3(blocks(Nprop, 64)), dim3(64), 0, c->stream, chunks[q], Nprop, c->Ncap, counts[q], c->q0(), c->Qq(), c->Qa());
        HIPCHK(hipGetLastError());
    }
    return settle_update(c); // EQF_OPT_EARLY_DOORBELL: the lift's doorbell of the update in front has rung long since
}

// Blocked right-looking factorisation of Z (rows x m, leading dimension ldz): one launch per 32-column panel
// (k_chol_step), preceded by the elimination of the first diagonal tile. Rows >= m of W receive Z[rows >= m] L^-T.
static int launch_chain(eqf_ctx* c, int rows, int m, int ldz, double* Z, double* W, bool first_tile_done = false, const int* spec = nullptr, int spec_seq = 0,
                        double* gpart = nullptr) {
    constexpr int NB = 32;
    if (!first_tile_done) { // the vision update's k_build_Z eliminates the first tile itself
        KTimer t(c, KN_CHOL_UPDATE);
        hipLaunchKernelGGL(k_chol_first, dim3(1), dim3(256), 0, c->stream, std::min(NB, m), ldz, Z, c->d_Linv, c->d_flags);
        HIPCHK(hipGetLastError());
    }
    int step = 0;
    KTimer t(c, KN_CHOL_PANEL, blocks(m, NB)); // one event pair around the whole chain: events between the steps would stretch it
    for (int kb = 0; kb < m; kb += NB, ++step) {
        const int w = std::min(NB, m - kb);
        const int c0 = kb + w;
        double* Lin = c->d_Linv + 1024 * (step & 1);
        double* Lout = c->d_Linv + 1024 * ((step + 1) & 1);
        const int gx = blocks(rows - c0, 32);
        const int nyS = c0 < m ? blocks(m - c0, 32) : 1;
        // gpart: the last launch (c0 == m) also produces Gamma = W z as GAMMA_G + 1 partial vectors (extra grid rows + its own panel)
        double* gp = (gpart && c0 >= m) ? gpart : nullptr;
        {
            auto launch = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(gx, nyS + (gp ? GAMMA_G : 0)), dim3(256), 0, c->stream, rows, m, kb, w, ldz, Z, W, Lin, Lout, c->d_flags, c0 < m ? 1 : 0, nyS, spec,
                                   spec_seq, gp, c->ld, gpart && step < 32 ? trace_slot(c, TR_STEP0 + step) : nullptr);
            };
            if (gp)
                launch(k_chol_step<true>);
            else
                launch(k_chol_step<false>);
        }
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// The same factorisation as launch_chain(first_tile_done = true) in ONE persistent kernel with a look-ahead schedule (eqf_lookahead.hpp):
// bit-identical W; Gamma arrives complete in d_gamma (no partial vectors). Eligible: 3 <= NJ <= 32 panels (64 < m <= 1024, i.e. up to 512 measured landmarks).
static bool lookahead_eligible(const eqf_ctx* c, int m) {
    const int NJ = blocks(m, 32);
    const int NI = (2 * NJ - 1) + blocks(c->n() + 1, 16) + 1; // every workgroup of the launch must be resident at once (+ 1: the statistics workgroup of the ZB = 2 form):
    // one workgroup per compute unit (LDS, registers); a partitioned or smaller device takes the launch chain, which needs no co-residency
    return c->opt_lookahead && c->la_selftest >= 0 && c->d_pub && NJ >= 3 && NJ <= c->la_njcap && NI <= c->cu_count;
}
// Workgroups of a look-ahead launch. With more than 16 panels the half-rows of the block rows >= 16 are split over two workgroups each (la_row2, round 4) where the
// device has the compute units for it: owner + S half-rows + T half-rows (+ 2 per split block row).
constexpr int LA_SPLIT_FROM = 16;
static int la_split_extra(const eqf_ctx* c, int NJ, int base) {
    const int extra = NJ > LA_SPLIT_FROM + 1 ? 2 * (NJ - LA_SPLIT_FROM) : 0;
    return (c->opt_la_split && base + extra <= c->cu_count) ? extra : 0;
}
static int launch_lookahead(eqf_ctx* c, int rows, int m, int ldz, const int* spec, int spec_seq, int zb = 0, const MeasFuse* zb_mf = nullptr) {
    LaArgs a{};
    a.rows = rows;
    a.m = m;
    a.ldz = ldz;
    a.NJ = blocks(m, 32);
    a.NI = (2 * a.NJ - 1) + blocks(rows - m, 16); // the owner + the S half-rows 2 .. 2 NJ - 1 + the T half-rows (16 rows each)
    const int extra = la_split_extra(c, a.NJ, a.NI); // (0 up to 17 panels)
    a.split_from = extra ? LA_SPLIT_FROM : a.NJ;
    a.watch_ahead = c->opt_la_watch_ahead;
    // EQF_OPT_LA_HOME (up to 16 panels, a device of 8 XCDs x 32 compute units): the owner and the 2 NJ - 2 S half-rows are the blocks of ONE XCD (b & 7 == home), the T
    // half-rows (+ the statistics workgroup) are dealt to the other seven; the rest of the 8 x slots grid returns at once (eqf_lookahead.hpp: la_st_l)
    const int nT = a.NI - (2 * a.NJ - 1);
    // (from 6 panels on, unless forced: the 8 x slots grid costs a small kernel more than the shorter hops give it - N = 50: -2 %, N = 100: +3 %, N = 200: +2.3 %)
    const bool home = c->opt_la_home && a.NJ <= 16 && c->cu_count == 256 && c->d_pubfl &&
                      (c->opt_la_home == 2 || c->la_home_force || (a.NJ >= 6 && device_to_itself(c)));
    a.home = home ? c->la_home : -1;
    a.pubfl = c->d_pubfl;
    const int home_grid = 8 * std::max(2 * a.NJ - 1, blocks(nT + (zb >= 2 ? 1 : 0), 7));
    if (home)
        ++c->la_home_launches;
    if (++c->la_seq <= 0) // positive: -1 is "no look-ahead launch in front" for k_lift / k_syrk_sub, 0 the initial state of every flag word
        c->la_seq = 1;
    a.seq = c->la_seq;
    if (c->opt_early_door && c->early_seq_next) { // EQF_OPT_EARLY_DOORBELL: the last T half-row tells the host that the update will be applied (launch_factor_tail set the sequence)
        a.early_cnt = c->d_door + 3, a.early_door = c->h_door + 3, a.early_gamma_host = c->h_res + 7 * (size_t)c->Ncap;
        a.early_seq = c->early_seq_next, a.early_spec = spec, a.early_spec_seq = spec_seq;
        c->early_armed_seq = c->early_seq_next;
    }
    c->early_seq_next = 0;
    a.timeout_ticks = c->la_timeout_ticks;
    a.Z = c->d_Z;
    a.W = c->d_W;
    a.Linv0 = c->d_Linv; // k_build_Z leaves L_0^-1 where step 0 of the launch chain reads it
    a.pub = c->d_pub;
    a.pubf = c->d_pubf;
    a.puby = c->d_puby;
    a.gamma = c->d_gamma;
    a.flags = c->d_flags;
    a.spec = spec;
    a.spec_seq = spec_seq;
    a.live_cols = a.NJ <= 16 ? c->la_live_cols : nullptr; // (the instantiations up to 16 panels)
    if (a.live_cols)
        ++c->live_first_launches;
    a.tr_steps = trace_slot(c, TR_STEP0);
    a.dbg = c->d_trace ? c->d_ladbg : nullptr;
    KTimer t(c, KN_CHOL_LOOKAHEAD); // ONE launch: the whole factorisation
    // MAXT = tiles a wave keeps in registers = ceil(NJ / 4)
    if (zb) { // EQF_OPT_Z_IN_LOOKAHEAD: the half-rows build their rows of Z themselves
        const bool large = a.NJ > 16; // round 6: the 17 .. 32-panel form (la_row2 + la_build_rows2; output blocks from memory: zb = 1 or 3)
        a.zb_sig = (const double*)c->sigma(), a.zb_ld = c->ld, a.zb_M = m / 2, a.zb_Mcap = c->Ncap, a.zb_var = c->tail_var;
        a.zb_C = c->d_C, a.zb_ytil = c->d_ytil, a.zb_lmidx = c->d_lmidx, a.zb_linv0 = c->d_Linv;
        // (the pinned packet's mapping is the one this update was mapped with - map_measurement ran in this call or, for a staged measurement, in stage_prepare)
        a.zb_ident = (zb != 2 && c->tail_ident) ? 1 : 0;
        a.tr_zb = trace_slot(c, TR_BUILD_Z);
        ++c->zb_launches;
        if (zb == 2) { // ... and evaluate the C blocks themselves; one more workgroup for the statistics and the speculation word (the kernel itself does not look at it)
            a.zb_mf = *zb_mf;
            a.spec = nullptr;
            if (home)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_lookahead<4, 2, true>), dim3(home_grid), dim3(LA_T), 0, c->stream, a);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_lookahead<4, 2>), dim3(a.NI + 1), dim3(LA_T), 0, c->stream, a);
        } else if (zb == 3) { // C, yTilde, index map from the propagation kernel's observer blocks; the statistics workgroup as above
            a.zb_mf = *zb_mf;
            a.spec = nullptr;
            if (large)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_lookahead<8, 3>), dim3(a.NI + extra + 1), dim3(LA_T), 0, c->stream, a);
            else if (home)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_lookahead<4, 3, true>), dim3(home_grid), dim3(LA_T), 0, c->stream, a);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_lookahead<4, 3>), dim3(a.NI + 1), dim3(LA_T), 0, c->stream, a);
        } else if (large)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_lookahead<8, 1>), dim3(a.NI + extra), dim3(LA_T), 0, c->stream, a);
        else if (home) // C, yTilde, index map from the measurement kernel
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_lookahead<4, 1, true>), dim3(home_grid), dim3(LA_T), 0, c->stream, a);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_lookahead<4, 1>), dim3(a.NI), dim3(LA_T), 0, c->stream, a);
    } else if (a.NJ <= 16) {
        if (home)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_lookahead<4, 0, true>), dim3(home_grid), dim3(LA_T), 0, c->stream, a);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_lookahead<4>), dim3(a.NI), dim3(LA_T), 0, c->stream, a);
    } else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_lookahead<8>), dim3(a.NI + extra), dim3(LA_T), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// the statistics kernel's input: the measurement by landmark in the pinned packet
static const double* pack_by_landmark(eqf_ctx* c, const int* measof, const double* y) {
    const size_t Ncap = (size_t)c->Ncap;
    for (int i = 0; i < c->N; ++i) {
        const int j = measof[i];
        c->h_ylm[i] = j >= 0 ? y[2 * j] : 0.0;
        c->h_ylm[Ncap + i] = j >= 0 ? y[2 * j + 1] : 0.0;
        c->h_ylm[2 * Ncap + i] = (double)j;
    }
    return c->h_ylm;
}
// map ascending measurement ids to state indices; returns 0 or EQF_E_BAD_ARG
// The per-measurement packets (h_lmidx, h_y, d_meas ...) are sized by the landmark capacity. A frame may carry more features than that
// (the ones without a landmark are about to be added, VIOFilter.cpp:217): grow first.
static int fit_measurement(eqf_ctx* c, int M) { return M > c->Ncap ? grow_capacity(c, std::max(M, 2 * c->Ncap)) : 0; }

static int map_measurement(eqf_ctx* c, const int* ids, int M, bool require_all, int* lmidx, int* measof) {
    HP_SCOPE("abi.map_measurement");
    // Consecutive frames usually measure the same ids: the previous mapping (still in the pinned packet, which only this function
    // writes) is reused when ids and landmark set are unchanged.
    if (c->map_gen == c->lm_gen && c->map_N == c->N && (int)c->map_ids.size() == M && lmidx == c->h_lmidx && (M == 0 || std::memcmp(ids, c->map_ids.data(), sizeof(int) * M) == 0) &&
        (!require_all || c->map_all))
        return 0;
    HP_SCOPE("mm.miss");
    c->map_gen = c->lm_gen - 1; // invalid until this call succeeds
    for (int i = 0; i < c->N; ++i)
        measof[i] = -1;
    bool all = true, ident = true;
    // The landmark ids of the state are usually ascending as well (a tracker numbers its features as they appear, removals keep the order, new landmarks are appended
    // with larger ids): then one merge pass maps the measurement - no sorted lookup table to rebuild (3-4 us at 200 landmarks, on every frame that gains or loses a
    // landmark) and no binary searches
    bool state_ascending = true;
    for (int i = 1; i < c->N && state_ascending; ++i)
        state_ascending = c->ids[i] > c->ids[i - 1];
    int h = 0;
    if (!state_ascending)
        ensure_lookup(c); // (kept up to date by eqf_add_landmarks / eqf_remove_landmarks: no sort in a frame with landmark turnover)
    for (int j = 0; j < M; ++j) {
        if (j > 0 && ids[j] <= ids[j - 1])
            return EQF_E_BAD_ARG; // must be strictly ascending (std::map order)
        int i;
        if (state_ascending) {
            while (h < c->N && c->ids[h] < ids[j])
                ++h;
            i = (h < c->N && c->ids[h] == ids[j]) ? h : -1;
        } else { // the same merge against the sorted (id, index) table
            while (h < c->N && c->lookup[h].first < ids[j])
                ++h;
            i = (h < c->N && c->lookup[h].first == ids[j]) ? c->lookup[h].second : -1;
        }
        lmidx[j] = i;
        ident = ident && i == j;
        if (i >= 0)
            measof[i] = j;
        else if (require_all)
            return EQF_E_BAD_ARG;
        else
            all = false;
    }
    if (lmidx == c->h_lmidx) {
        c->map_ids.assign(ids, ids + M);
        c->map_gen = c->lm_gen;
        c->map_N = c->N;
        c->map_all = all;
        c->map_ident = ident;
    }
    return 0;
}

int eqf_output_cov_all(eqf_ctx* c, const eqvio_camera* cam, double* out4N) {
    if (!c || !cam || !out4N || !camera_ok(cam))
        return EQF_E_BAD_ARG;
    const int N = c->N;
    if (N == 0)
        return 0;
    {
        const Cam k = make_cam(cam);
        if (c->ocov_valid && k.fx == c->ocov_cam.fx && k.fy == c->ocov_cam.fy && k.cx == c->ocov_cam.cx && k.cy == c->ocov_cam.cy && k.model == c->ocov_cam.model &&
            std::equal(k.d, k.d + 5, c->ocov_cam.d)) { // computed with the state estimate that was read just before (fetch_estimates); nothing has touched the state since
            std::memcpy(out4N, c->h_ocov, sizeof(double) * 4 * N);
            return 0;
        }
    }
    c->ocov_hint = make_cam(cam), c->ocov_hint_valid = true;
    { int _e = enter(c); if (_e) return _e; }
    { int _r = join_observer(c); if (_r) return _r; }
    if (!c->h_ocov)
        HIPCHK(hipHostMalloc(&c->h_ocov, sizeof(double) * 4 * (size_t)c->Ncap));
    LAUNCH_TS(c, k_output_cov, dim3(blocks(N, 64)), dim3(64), c->stream, N, c->Ncap, c->ld, c->chart, make_cam(cam), c->q0(), c->Qq(), c->Qa(), (const TS*)c->sigma(), c->h_ocov);
    HIPCHK(hipGetLastError());
    { int _r = sync_ctx(c); if (_r) return _r; }
    std::memcpy(out4N, c->h_ocov, sizeof(double) * 4 * N);
    return 0;
}

int eqf_outlier_stats(eqf_ctx* c, const eqvio_camera* cam, const int* ids, const double* y, int M, double* absErr, double* probErr, double* depth2) {
    if (!c || !cam || M < 0 || (M > 0 && (!ids || !y)) || !camera_ok(cam))
        return EQF_E_BAD_ARG;
    const int N = c->N;
    if (N == 0)
        return 0;
    { int _e = enter(c); if (_e) return _e; }
    { int _r = fit_measurement(c, M); if (_r) return _r; }
    c->staged_valid = c->stage_pending = c->stage_requested = false; // the pinned measurement packet is about to be rewritten
    if (c->busy_meas) {
        int r = sync_ctx(c);
        if (r)
            return r;
    }
    int* lmidx = c->h_lmidx;
    int* measof = c->h_lmidx + c->Ncap;
    int rc = map_measurement(c, ids, M, false, lmidx, measof);
    if (rc)
        return rc;
    std::memcpy(c->h_y, y, sizeof(double) * 2 * M);
    rc = join_observer(c);
    if (rc)
        return rc;
    const bool use_door = c->opt_door && !c->obs_pending;
    const int door_seq = (int)(++c->door_seq);
    {
        KTimer t(c, KN_STATS);
        c->busy_meas = true;
        LAUNCH_TS(c, k_outlier_stats, dim3(blocks(N, 64)), dim3(64), c->stream, N, c->Ncap, c->ld, c->chart, make_cam(cam), pack_by_landmark(c, measof, y), c->q0(), c->Qq(), c->Qa(),
                  (const TS*)c->sigma(), c->h_res, 1, c->d_C, c->d_ytil, c->d_lmidx, c->d_flags, use_door ? c->d_door : nullptr, c->h_door, door_seq, 0.0, 0.0,
                  (int*)nullptr, 0, (double*)nullptr);
        HIPCHK(hipGetLastError());
    }
    {
        bool all_known = true;
        for (int j = 0; j < M; ++j)
            all_known = all_known && (lmidx[j] >= 0);
        c->meas_valid = all_known;
        c->meas_star = 1;
        c->meas_ids.assign(ids, ids + M);
    }
    rc = use_door ? door_wait(c, 0, door_seq) : sync_ctx(c);
    if (rc)
        return rc;
    if (absErr)
        std::memcpy(absErr, c->h_res, sizeof(double) * N);
    if (probErr)
        std::memcpy(probErr, c->h_res + N, sizeof(double) * N);
    if (depth2)
        std::memcpy(depth2, c->h_res + 2 * N, sizeof(double) * N);
    return 0;
}

static int stage_measurement(eqf_ctx* c, const int* ids, const double* y, int M) {
    if (c->busy_meas) {
        int r = sync_ctx(c);
        if (r)
            return r;
    }
    c->staged_valid = c->stage_pending = c->stage_requested = false; // the pinned measurement packet is about to be rewritten
    int* lmidx = c->h_lmidx;
    int* measof = c->h_lmidx + c->Ncap;
    int rc = map_measurement(c, ids, M, true, lmidx, measof);
    if (rc)
        return rc;
    std::memcpy(c->h_y, y, sizeof(double) * 2 * M);
    c->busy_meas = true;
    return join_observer(c);
}

// The device part of the vision update behind the measurement stage: Z, factorisation chain, Sigma update, lift. With
// spec != nullptr every kernel first compares *spec with spec_seq and returns at once if they match (cancelled tail).
static LiftArgs lift_args(eqf_ctx* c, int discreteCorr, const int* spec, int spec_seq, bool use_door, int door_seq, const double* gpart, int stall_seq) {
    return LiftArgs{c->N, c->Ncap, c->chart, discreteCorr, c->d_gamma, c->q0(), c->Qq(), c->Qa(), c->h_res + 3 * (size_t)c->Ncap, c->h_res + 7 * (size_t)c->Ncap, c->d_flags,
                    c->h_resflags, use_door ? c->d_door + 1 : nullptr, c->h_door + 1, door_seq, spec, spec_seq, gpart, c->ld, trace_slot(c, TR_LIFT), stall_seq};
}
static int launch_lift(eqf_ctx* c, int discreteCorr, const int* spec, int spec_seq, bool use_door, int door_seq, const double* gpart, int stall_seq) {
    KTimer t(c, KN_LIFT);
    hipLaunchKernelGGL(k_lift, dim3(blocks(c->N, 64)), dim3(64), 0, c->stream, lift_args(c, discreteCorr, spec, spec_seq, use_door, door_seq, gpart, stall_seq));
    HIPCHK(hipGetLastError());
    return 0;
}
// Self-test at context creation (ADVICE r2 / VERDICT r3, r4): the look-ahead kernel's hand-offs rest on relaxed agent-scope flags, write-through stores,
// s_waitcnt ordering and - EQF_OPT_LA_HOME - on plain stores being visible to the other compute units of one XCD (eqf_lookahead.hpp): measured behaviour of gfx950
// under ROCm 7.2, not a guarantee of the HIP memory model. Before a context is used, fixed problems [S ; T ; y^T] of up to three shapes - 3 panels, the largest shape up to
// 13 panels this capacity allows (ragged last panel: la_row and, when the device qualifies, the HOME placement at the size it was built for) and 17 panels (la_row2, split
// late block rows) - are factorised on the launch chain once and on the look-ahead kernel EIGHT times each (the protocol slip of round 3 showed in ~8 % of the launches): the
// W rows must agree bit for bit every time and no wait may run out. A refused or failing HOME placement is retried in the classic placement (and stays off); a mismatch
// there leaves the context on the launch chain (eqf_lookahead_stats reports it: launches stay 0).
static int lookahead_selftest_pass(eqf_ctx* c, bool& placement_refused) {
    constexpr int nT = 16, REPS = 8;
    placement_refused = false;
    int shapes[3] = {96, std::min(416, c->mcap & ~1), 544};
    if (shapes[1] <= 128)
        shapes[1] = 0;
    int verdict = 0; // 0: nothing ran, 1: every launch agreed, -1: a mismatch
    double* d_z0 = nullptr;
    for (int si = 0; si < 3; ++si) {
        const int m = shapes[si], rows = m + nT + 1;
        if (m == 0 || m > c->mcap || blocks(m, 32) > c->la_njcap || !lookahead_eligible(c, m))
            continue;
        const size_t cnt = (size_t)c->ldz * m;
        std::vector<double> hz(cnt, 0.0), wref(cnt), w(cnt);
        for (int j = 0; j < m; ++j) {
            for (int i = 0; i < m; ++i) // symmetric positive definite S
                hz[i + (size_t)j * c->ldz] = (i == j ? 4.0 + 0.01 * (j % 5) : 0.0) + 1.0 / (1 + std::abs(i - j)) + 1e-3 * ((i + j) % 7);
            for (int t = 0; t < nT; ++t)
                hz[m + t + (size_t)j * c->ldz] = ((t * 31 + j * 17) % 23) / 23.0 - 0.5;
            hz[m + nT + (size_t)j * c->ldz] = ((j * 5) % 13) / 13.0;
        }
        if (!d_z0)
            HIPCHK(hipMalloc(&d_z0, sizeof(double) * (size_t)c->ldz * std::min(c->mcap, 544)));
        auto fail = [&](int rc) {
            hipFree(d_z0);
            return rc;
        };
        if (hipMemcpy(d_z0, hz.data(), sizeof(double) * cnt, hipMemcpyHostToDevice) != hipSuccess)
            return fail(EQF_E_NO_DEVICE);
        for (int rep = -1; rep < REPS; ++rep) { // rep -1: the launch chain
            int fl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            bool stalled = false;
            // A launch that STALLS (its workgroups were not all resident within the bound: other contexts are being created on the same device, or share it) says nothing
            // about the kernel: it is repeated a few times, and if the device stays busy the test counts as not run - the look-ahead kernel stays enabled, every launch of
            // it is bounded and redone on the chain when it stalls (finish_update)
            int learnt = 0;
            for (int attempt = 0; attempt < 4; ++attempt) {
                if (hipMemcpyAsync(c->d_Z, d_z0, sizeof(double) * cnt, hipMemcpyDeviceToDevice, c->stream) != hipSuccess || hipMemsetAsync(c->d_W, 0, sizeof(double) * cnt, c->stream) != hipSuccess)
                    return fail(EQF_E_NO_DEVICE);
                int rc;
                if (rep < 0)
                    rc = launch_chain(c, rows, m, c->ldz, c->d_Z, c->d_W);
                else {
                    hipLaunchKernelGGL(k_chol_first, dim3(1), dim3(256), 0, c->stream, 32, c->ldz, c->d_Z, c->d_Linv, c->d_flags);
                    if (hipGetLastError() != hipSuccess)
                        return fail(EQF_E_NO_DEVICE);
                    const int base = (2 * blocks(m, 32) - 1) + blocks(rows - m, 16);
                    (void)la_book(c, base + 1 + la_split_extra(c, blocks(m, 32), base)); // (result ignored on purpose: a timeout books nothing, the launch is bounded and a stall repeats the attempt)
                    rc = launch_lookahead(c, rows, m, c->ldz, nullptr, 0);
                }
                if (rc)
                    return fail(rc);
                if (hipMemcpyAsync((rep < 0 ? wref : w).data(), c->d_W, sizeof(double) * cnt, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
                    return fail(EQF_E_NO_DEVICE);
                la_release(c);
                if (hipMemcpy(fl, c->d_flags, sizeof(fl), hipMemcpyDeviceToHost) != hipSuccess || hipMemset(c->d_flags, 0, sizeof(int) * 8) != hipSuccess)
                    return fail(EQF_E_NO_DEVICE);
                stalled = rep >= 0 && fl[3] == c->la_seq;
                if (rep >= 0 && fl[4] == c->la_seq) { // the blocks of the grid do not sit on the XCDs the HOME placement assumed: block 0 ran on XCD fl[5]
                    if (++learnt > 2 || fl[5] < 0 || fl[5] > 7) { // (learnt once per stream in practice: the offset belongs to its hardware queue)
                        placement_refused = true;
                        return fail(0);
                    }
                    c->la_home = fl[5];
                    --attempt;
                    continue;
                }
                if (!stalled)
                    break;
            }
            if (stalled) {
                std::fprintf(stderr, "[eqf_hip] look-ahead self-test could not run on device %d (the launch stalled four times: the device is busy); not counted as a failure\n", c->device);
                return fail(0);
            }
            if (rep < 0) {
                if (fl[0]) { // (cannot happen: the problem is positive definite)
                    c->la_selftest = -1;
                    return fail(0);
                }
                continue;
            }
            bool same = !fl[0];
            for (int j = 0; j < m && same; ++j)
                same = std::memcmp(&wref[m + (size_t)j * c->ldz], &w[m + (size_t)j * c->ldz], sizeof(double) * (nT + 1)) == 0;
            if (!same) {
                std::fprintf(stderr, "[eqf_hip] look-ahead self-test: %d columns, launch %d of %d differs from the launch chain on device %d (pivot flag %d)\n", m, rep + 1, REPS, c->device, fl[0]);
                c->la_selftest = -1;
                return fail(0);
            }
            verdict = 1;
        }
    }
    hipFree(d_z0);
    if (verdict > 0)
        c->la_selftest = 1;
    return 0;
}
static int lookahead_selftest(eqf_ctx* c) {
    c->la_selftest = 0;
    if (!c->opt_lookahead || c->mcap < 96 || c->la_njcap < 3 || c->cu_count < 8)
        return 0; // the look-ahead kernel is never eligible at this capacity / on this device
    bool refused = false;
    c->la_home_force = true;
    int rc = lookahead_selftest_pass(c, refused);
    c->la_home_force = false;
    if (rc)
        return rc;
    const bool home_on = c->opt_la_home != 0;
    if (home_on && (refused || c->la_selftest < 0)) {
        std::fprintf(stderr, "[eqf_hip] device %d: the look-ahead kernel's HOME placement %s; this context uses the classic placement\n", c->device,
                     refused ? "was refused (blocks are not dealt to the XCDs round robin)" : "failed its self-test");
        c->opt_la_home = 0;
        c->la_selftest = 0;
        rc = lookahead_selftest_pass(c, refused);
        if (rc)
            return rc;
    }
    if (c->la_selftest < 0)
        std::fprintf(stderr, "[eqf_hip] look-ahead self-test failed on device %d: this context factorises on the launch chain\n", c->device);
    c->la_home_launches = 0; // (the self-test's own launches are not the filter's: eqf_lookahead_home counts what eqf_lookahead_stats counts)
    return 0;
}
static int launch_factor_tail(eqf_ctx* c, int M, int discreteCorr, const int* spec, int spec_seq, bool use_door, int door_seq, bool force_chain, int zb = 0,
                              const MeasFuse* zb_mf = nullptr);
static int launch_update_tail(eqf_ctx* c, const int* ids, int M, double meas_var, int discreteCorr, const int* spec, int spec_seq, bool use_door, int door_seq,
                              const MeasFuse* fuse = nullptr, bool live_first = false) {
    HP_SCOPE("abi.launch_update_tail");
    const int n = c->n(), m = 2 * M;
    const int rows = m + n + 1;
    int rc = 0;
    c->tail_la = false, c->tail_zb = false, c->tail_M = M; // retry state of finish_update: reset before anything of this tail is queued
    // "measurement j is landmark j" lets the Z-building prologue skip the index map: true only if the mapping in the pinned packet is the one of THESE ids
    c->tail_ident = c->map_ident && c->map_gen == c->lm_gen && c->map_N == c->N && (int)c->map_ids.size() == M && std::equal(ids, ids + M, c->map_ids.begin());
    if (live_first) // k_stats_select in front of this tail puts the measurements of the landmarks that stay first: the index map is not the identity
        c->tail_ident = false;
    // The look-ahead kernel's workgroups are booked against the device's compute units BEFORE anything of the tail depends on that kernel (who builds Z); if they
    // do not come free within the bound (la_book), this update takes k_build_Z + the launch chain
    bool chain_only = false;
    LaBookingGuard booking{c}; // an error exit below gives the compute units back
    if (lookahead_eligible(c, m)) {
        const int base = (2 * blocks(m, 32) - 1) + blocks(rows - m, 16);
        chain_only = !la_book(c, base + 1 + la_split_extra(c, blocks(m, 32), base));
        if (chain_only)
            ++c->la_book_timeouts;
    }
    // EQF_OPT_Z_IN_LOOKAHEAD: with the C blocks in memory (k_measure / k_outlier_stats ran), fp64 Sigma and 3 .. 16 panels, the look-ahead kernel builds Z itself
    // (with measurement fusion - the speculative frame tail - it evaluates the C blocks as well, if the measurement has been staged to HBM: ZB = 2)
    const bool zb_ok = c->opt_zb && !c->sig32 && !chain_only && lookahead_eligible(c, m) && (blocks(m, 32) <= 16 || c->opt_zb_large); // (round 6: 17 .. 32 panels too, la_build_rows2)
    // ZB = 2 up to 8 panels (N <= 128) only: measured +2.8 % at N = 50, +1.8 % at N = 100 and neutral at N = 200, where the tail's first launch then reaches
    // the GPU late
    // ZB = 3 (round 4, up to 16 panels): the propagation kernel's observer blocks have evaluated the output blocks of THIS measurement (staged, same landmark set) with
    // the camera and output choice this call asks for: the look-ahead kernel builds Z from them and one more workgroup computes the statistics
    auto same_cam = [](const Cam& x, const Cam& y) {
        return x.fx == y.fx && x.fy == y.fy && x.cx == y.cx && x.cy == y.cy && x.model == y.model && std::equal(x.d, x.d + 5, y.d);
    };
    const bool in_prop = fuse && c->me_valid && c->opt_measure_prop && fuse->y == c->d_meas && c->opt_early && c->me_M == M && c->me_gen == c->lm_gen &&
                         c->me_star == fuse->star && same_cam(c->me_cam, fuse->cam);
    c->me_valid = false;
    const int zb = !zb_ok ? 0 : (!fuse ? 1 : (in_prop ? 3 : ((fuse->y == c->d_meas && c->opt_early && blocks(m, 32) <= 8) ? 2 : 0)));
    if (zb == 3)
        ++c->me_used;
    c->tail_zb = zb != 0;
    c->tail_var = meas_var;
    if (!zb) {
        KTimer t(c, KN_BUILD_Z);
        // one extra grid row eliminates the first diagonal tile of S (no k_chol_first launch in this chain); with measurement fusion
        // the kernel also evaluates C itself, one more grid row computes the outlier statistics and decides about the tail
        MeasFuse mf{};
        if (fuse)
            mf = *fuse;
        auto launch = [&](auto kern, auto* sig) {
            hipLaunchKernelGGL(kern, dim3(blocks(n + M + 1, 256), blocks(M, BZ_JB) + 1 + (mf.enabled ? 1 : 0)), dim3(256), 0, c->stream, n, M, c->Ncap, c->ld, c->ldz, meas_var, c->d_lmidx, sig,
                               c->d_C, c->d_ytil, c->d_Z, c->d_Linv, c->d_flags, mf.enabled ? (const int*)nullptr : spec, spec_seq, mf, trace_slot(c, TR_BUILD_Z));
        };
        if (mf.enabled && in_prop && !c->sig32) { // the propagation kernel has evaluated this measurement's output blocks: nobody evaluates them again
            ++c->me_used;
            launch(k_build_Z<double, true, true>, (const double*)c->sigma());
        } else if (c->sig32) {
            if (mf.enabled)
                launch(k_build_Z<float, true>, (const float*)c->sigma());
            else
                launch(k_build_Z<float, false>, (const float*)c->sigma());
        } else {
            if (mf.enabled)
                launch(k_build_Z<double, true>, (const double*)c->sigma());
            else
                launch(k_build_Z<double, false>, (const double*)c->sigma());
        }
        HIPCHK(hipGetLastError());
    }
    host_stamp(c, TH_BUILD_Z_OUT);
    c->tail_M = M; // what a retry of the factorisation on the launch chain needs to know (finish_update)
    c->la_live_cols = live_first ? c->d_spec + 2 : nullptr;
    rc = launch_factor_tail(c, M, discreteCorr, spec, spec_seq, use_door, door_seq, chain_only, zb, fuse);
    c->la_live_cols = nullptr;
    booking.keep = rc == 0 && c->tail_la; // released by the doorbell wait (door_wait), sync_ctx or eqf_destroy
    return rc;
}
// Everything behind k_build_Z: factorisation of Z (look-ahead kernel or launch chain), lift, covariance update. force_chain: the retry after a stalled
// look-ahead kernel (Z and L_0^-1 are inputs of that kernel only, so the chain can start from them again).
static int launch_factor_tail(eqf_ctx* c, int M, int discreteCorr, const int* spec, int spec_seq, bool use_door, int door_seq, bool force_chain, int zb, const MeasFuse* zb_mf) {
    const int n = c->n(), m = 2 * M;
    const int rows = m + n + 1;
    int rc = 0;
    const bool la = !force_chain && lookahead_eligible(c, m); // one persistent kernel instead of one launch per panel; Gamma complete in d_gamma
    c->tail_la = la; // the retry state of finish_update always describes the tail in flight (with tail_zb / tail_M / tail_var, set by launch_update_tail)
    LaBookingGuard booking{c}; // (booked by launch_update_tail, before it decided who builds Z; handed over to the doorbell wait when everything is queued)
    if (la)
        ++c->la_launches;
    c->early_seq_next = (la && use_door && c->early_allowed && c->opt_early && c->opt_lift_syrk && !c->sig32 && c->opt_timing != 1 && !c->opt_check) ? door_seq : 0;
    c->early_allowed = false;
    rc = la ? launch_lookahead(c, rows, m, c->ldz, spec, spec_seq, zb, zb_mf) : launch_chain(c, rows, m, c->ldz, c->d_Z, c->d_W, true, spec, spec_seq, c->opt_early ? c->d_gpart : nullptr);
    if (rc)
        return rc;
    booking.keep = la;
    const int stall_seq = la ? c->la_seq : -1; // the stall word the kernels behind the factorisation compare (sequence valued: eqf_lookahead.hpp)
    // EQF_OPT_LIFT_WITH_SYRK: lift and covariance update wait for the same kernel and touch different data - one launch, the lift's workgroups in front
    // (not with per-kernel timing, which wants the two spans apart, and not with fp32 storage, whose rounding pass follows the covariance update)
    if (c->opt_early && c->opt_lift_syrk && !c->sig32 && c->opt_timing != 1 && c->N > 0) {
        const int nt = blocks(n, 32), nlift = blocks(c->N, 64);
        KTimer t(c, KN_SYRK);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_syrk_lift<double>), dim3(nlift + nt * (nt + 1) / 2), dim3(64 * SYRK_NW), 0, c->stream, n, m, c->ld, c->ldz, c->d_W, (double*)c->sigma(),
                           c->d_gamma, c->d_flags, trace_slot(c, TR_SYRK), c->d_syrk_order + c->syrk_off[nt], lift_args(c, discreteCorr, spec, spec_seq, use_door, door_seq, la ? nullptr : c->d_gpart, stall_seq),
                           nlift, blocks(m, 32) <= 16 ? c->la_live_cols : (const int*)nullptr, nt <= SYRK_ARITH_TILES ? nt : 0);
        HIPCHK(hipGetLastError());
        { int _r = round_sigma(c, spec, spec_seq); if (_r) return _r; }
        if (c->opt_check) {
            LAUNCH_TS(c, k_check_finite, dim3(blocks(n, 256), n), dim3(256), c->stream, n, c->ld, (const TS*)c->sigma(), c->d_flags);
            HIPCHK(hipGetLastError());
        }
        return 0;
    }
    if (c->opt_early) { // Gamma, landmark lift, result packet and doorbell BEFORE the covariance update: the host round trip overlaps with it
        rc = launch_lift(c, discreteCorr, spec, spec_seq, use_door, door_seq, la ? nullptr : c->d_gpart, stall_seq);
        if (rc)
            return rc;
    }
    {
        const int nt = blocks(n, 32);
        KTimer t(c, KN_SYRK);
        const dim3 sg(nt * (nt + 1) / 2), sb(64 * SYRK_NW);
#define SYRK_LAUNCH(TS_, G_) \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_syrk_sub<TS_, G_>), sg, sb, 0, c->stream, n, m, c->ld, c->ldz, c->d_W, (TS_*)c->sigma(), nt, c->d_gamma, spec, spec_seq, G_ ? 1 : 0, c->d_flags, \
                       trace_slot(c, TR_SYRK), c->d_syrk_order + c->syrk_off[nt], stall_seq)
        const bool wg = !(c->opt_early || la); // EQF_OPT_EARLY_LIFT = 0 on the launch chain: Gamma = W z is a by-product of the diagonal tiles
        if (c->sig32) { if (wg) SYRK_LAUNCH(float, true); else SYRK_LAUNCH(float, false); }
        else { if (wg) SYRK_LAUNCH(double, true); else SYRK_LAUNCH(double, false); }
#undef SYRK_LAUNCH
        HIPCHK(hipGetLastError());
    }
    { int _r = round_sigma(c, spec, spec_seq); if (_r) return _r; }
    if (!c->opt_early) {
        rc = launch_lift(c, discreteCorr, spec, spec_seq, use_door, door_seq, nullptr, stall_seq);
        if (rc)
            return rc;
    }
    if (c->opt_check) {
        LAUNCH_TS(c, k_check_finite, dim3(blocks(n, 256), n), dim3(256), c->stream, n, c->ld, (const TS*)c->sigma(), c->d_flags);
        HIPCHK(hipGetLastError());
    }
    return 0;
}
// Host part after the wait: the lift kernel wrote Gamma's sensor part, the new estimates / invalid flags (4N) and the status
// flags straight into the pinned result packet; the sensor part of Delta is lifted here.
static int apply_sensor_lift(eqf_ctx* c, int discreteCorr);
static int finish_update(eqf_ctx* c, int discreteCorr, bool retried = false) {
    HP_SCOPE("abi.finish_update");
    const int N = c->N, n = c->n();
    c->h_flags[0] = c->h_resflags[0];
    c->h_flags[1] = c->h_resflags[1];
    // A failed factorisation is reported BEFORE anything of the filter changes: the device kept Sigma and the landmarks (k_lift,
    // k_syrk_sub), the sensor lift below is not applied.
    // (round 5, found by the concurrent time-out soak: a launch that gives up half way may ALSO have raised the pivot flag - its pivot wave goes on eliminating
    // whatever is in its LDS until it notices - so the pivot flag of a stalled launch says nothing; the chain decides, from a cleared flag)
    if (c->h_resflags[3] && c->tail_la && !retried) {
        // A bounded wait of the look-ahead kernel ran out: its workgroups were not all resident within the bound (another process or a long kernel
        // of this process holds the CUs). Nothing of the filter was modified, and Z / L_0^-1 are inputs of that kernel only: redo the factorisation
        // on the launch chain (which needs no co-residency), then lift and update Sigma as usual. Three stalls in a row switch the look-ahead
        // kernel off for this context. EQF_E_STALLED reaches the caller only if the chain fails as well (it cannot stall).
        ++c->la_fallbacks;
        if (c->opt_la_home) { // a HOME launch that found its blocks on other XCDs than assumed (the stream moved to another hardware queue) ended at once: learn the offset again
            int fl[8];
            HIPCHK(hipMemcpy(fl, c->d_flags, sizeof(fl), hipMemcpyDeviceToHost));
            if (fl[4] == c->la_seq && fl[5] >= 0 && fl[5] <= 7) {
                c->la_home = fl[5];
                --c->la_consecutive_stalls; // not a stall of the kernel
                if (++c->la_home_refused > 16)
                    c->opt_la_home = 0; // a device that does not keep its offsets: classic placement from here on
            }
        }
        if (++c->la_consecutive_stalls >= 3)
            c->opt_lookahead = 0;
        HIPCHK(hipMemsetAsync(c->d_flags + 3, 0, sizeof(int), c->stream));
        HIPCHK(hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
        const bool use_door = c->opt_door && !c->opt_check && !c->obs_pending;
        const int door_seq = (int)(++c->door_seq);
        if (c->tail_zb) { // the stalled kernel had built Z in its registers: the chain needs it (and L_0^-1) in memory
            const int n_ = c->n(), M_ = c->tail_M;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_build_Z<double, false>), dim3(blocks(n_ + M_ + 1, 256), blocks(M_, BZ_JB) + 1), dim3(256), 0, c->stream, n_, M_, c->Ncap, c->ld, c->ldz,
                               c->tail_var, c->d_lmidx, (const double*)c->sigma(), c->d_C, c->d_ytil, c->d_Z, c->d_Linv, c->d_flags, (const int*)nullptr, 0, MeasFuse{}, (trace_t*)nullptr);
            HIPCHK(hipGetLastError());
            c->tail_zb = false;
        }
        int rc = launch_factor_tail(c, c->tail_M, discreteCorr, nullptr, 0, use_door, door_seq, true);
        if (rc)
            return rc;
        rc = use_door ? door_wait(c, 1, door_seq) : sync_ctx(c);
        if (rc)
            return rc;
        return finish_update(c, discreteCorr, true);
    }
    if (c->h_resflags[3] || c->h_flags[0]) {
        if (std::getenv("EQF_DEBUG_STATS"))
            std::fprintf(stderr, "[eqf_hip] update failed: retried %d tail_la %d tail_zb %d resflags %d %d %d %d la_seq %d\n", (int)retried, (int)c->tail_la, (int)c->tail_zb, c->h_resflags[0],
                         c->h_resflags[1], c->h_resflags[2], c->h_resflags[3], c->la_seq);
        c->est_valid = false, ++c->est_epoch;
        c->meas_valid = false;
        return c->h_resflags[3] ? EQF_E_STALLED : EQF_E_NOT_SPD;
    }
    if (c->tail_la)
        c->la_consecutive_stalls = 0;
    c->gamma_stale = true;
    c->n_at_update = n;
    c->est_cache.assign(c->h_res + 3 * (size_t)c->Ncap, c->h_res + 3 * (size_t)c->Ncap + 4 * N);
    c->est_valid = true;
    return apply_sensor_lift(c, discreteCorr);
}
// An update taken from the early doorbell (EQF_OPT_EARLY_DOORBELL): it WILL be applied - every W row is final, no pivot failed, no wait ran out, the tail was not
// cancelled, and the lift and the covariance update behind the factorisation look at the same words - and Gamma's sensor rows are in the packet: the sensor lift is applied
// now; the landmark estimates and their invalid flags come with the lift's doorbell (settle_update).
static int finish_update_early(eqf_ctx* c, int discreteCorr, int seq) {
    c->h_flags[0] = c->h_flags[1] = 0;
    if (c->tail_la)
        c->la_consecutive_stalls = 0;
    c->gamma_stale = true;
    c->n_at_update = c->n();
    c->est_valid = false, ++c->est_epoch;
    c->unsettled = true;
    c->settle_seq = seq, c->settle_N = c->N, c->settle_gen = c->lm_gen, c->settle_epoch = c->est_epoch;
    c->settle_ids = c->ids;
    ++c->early_rings;
    return apply_sensor_lift(c, discreteCorr);
}
static int settle_update(eqf_ctx* c) {
    if (!c->unsettled)
        return 0;
    HP_SCOPE("abi.settle_update");
    c->unsettled = false;
    HIPCHK(hipSetDevice(c->device));
    c->unsettled_wait = true;
    const int rc = door_wait(c, 1, c->settle_seq);
    c->unsettled_wait = false;
    if (rc)
        return rc;
    if (c->h_resflags[0] || c->h_resflags[2] || c->h_resflags[3]) { // (cannot happen: the early doorbell rings only when none of these can be raised any more)
        std::fprintf(stderr, "[eqf_hip] an update announced by the early doorbell was not applied (flags %d %d %d)\n", c->h_resflags[0], c->h_resflags[2], c->h_resflags[3]);
        return EQF_E_STALLED;
    }
    const int N = c->settle_N;
    const double* est = c->h_res + 3 * (size_t)c->Ncap;
    c->invalid_ids.clear();
    for (int i = 0; i < N; ++i)
        if (est[3 * (size_t)N + i] != 0.0)
            c->invalid_ids.push_back(c->settle_ids[i]);
    if (c->lm_gen == c->settle_gen && c->N == N && c->est_epoch == c->settle_epoch) { // nothing has moved since the update: the estimates are the current ones
        c->est_cache.assign(est, est + 4 * (size_t)N);
        c->est_valid = true;
    }
    return 0;
}
static int apply_sensor_lift(eqf_ctx* c, int discreteCorr) {
    std::memcpy(c->h_buf, c->h_res + 7 * (size_t)c->Ncap, sizeof(double) * 21);
    const double* g = c->h_buf;
    GroupSensor D;
    D.bgyr = v3(g[0], g[1], g[2]);
    D.bacc = v3(g[3], g[4], g[5]);
    V3 gw = v3(g[6], g[7], g[8]), gv = v3(g[9], g[10], g[11]), gvel = v3(g[12], g[13], g[14]);
    V3 cw = v3(g[15], g[16], g[17]), cv = v3(g[18], g[19], g[20]);
    const bool normal = c->chart == EQVIO_COORD_NORMAL;
    if (normal && !discreteCorr) {
        // liftInnovation_normal = liftInnovation_euclid(M^-1 Gamma) (normal.cpp:47-50); sensor block of M^-1: [12:15,6:9] = skew(v0), [15:21,6:12] = -Ad(T0^-1)
        gvel = gvel + cross(c->xi0.vel, gw);
        const V6 UA = Ad_apply(pose_inv(c->xi0.cam), V6{gw, gv});
        cw = cw - UA.w;
        cv = cv - UA.v;
    }
    if (normal && discreteCorr) {
        // liftInnovationDiscrete_normal (normal.cpp:52-55) = the Euclidean discrete lift of chart_euclid(chart_normal^-1(Gamma)). With
        // (R, x0, x1) = SE_2(3).exp(Gamma[6:15]) (sensorChart_normal.inv, VIOState.cpp:138-151) that composition collapses to
        // A = (R, x0), w = -x1, B = SE3.exp(Gamma[15:21]).
        const M3 V = so3_V(gw);
        D.A = Pose{so3_exp(gw), V * gv};
        D.w = -(V * gvel);
        D.B = se3_exp(cw, cv);
    } else if (discreteCorr) {
        // liftInnovationDiscrete sensor part (euclid.cpp:74-79 == invdepth.cpp:228-233)
        D.A = se3_exp(gw, gv);
        D.w = c->xi0.vel - q_rot(D.A.R, c->xi0.vel + gvel);
        D.B = pose_mul(pose_mul(pose_mul(pose_inv(c->xi0.cam), D.A), c->xi0.cam), se3_exp(cw, cv));
    } else {
        // VIOExp(liftInnovation) sensor part (euclid.cpp:40-51, VIOGroup.cpp:273-283)
        const V3 u_w = -gvel - cross(gw, c->xi0.vel);
        const V6 UB0 = Ad_apply(pose_inv(c->xi0.cam), V6{gw, gv});
        const M3 V = so3_V(gw);
        D.A = Pose{so3_exp(gw), V * gv};
        D.w = V * u_w;
        D.B = se3_exp(cw + UB0.w, cv + UB0.v);
    }
    c->X = group_mul(D, c->X);
    if (c->opt_check) { // the finite check ran after the result packet was written: fetch its verdict
        const int r = read_flags(c);
        if (r)
            return r;
    }
    if (c->h_flags[1])
        return EQF_E_NONFINITE;
    for (int i = 0; i < 21; ++i)
        if (!(g[i] - g[i] == 0.0))
            return EQF_E_NONFINITE;
    return 0;
}

int eqf_vision_update(eqf_ctx* c, const eqvio_camera* cam, const int* ids, const double* y, int M, double meas_var, int useEqv, int discreteCorr) {
    if (!c || !cam || M < 0 || (M > 0 && (!ids || !y)) || !camera_ok(cam))
        return EQF_E_BAD_ARG;
    if (M == 0)
        return 0; // VIO_eqf.cpp:108-109
    if (M > c->N)
        return EQF_E_BAD_ARG;
    { int _e = enter(c); if (_e) return _e; }
    const bool reuse = c->meas_valid && c->meas_star == (useEqv ? 1 : 0) && (int)c->meas_ids.size() == M && std::equal(ids, ids + M, c->meas_ids.begin()) &&
                       std::memcmp(c->h_y, y, sizeof(double) * 2 * M) == 0;
    int rc = 0;
    if (!reuse) {
        rc = stage_measurement(c, ids, y, M);
        if (rc)
            return rc;
    }
    c->meas_valid = false;
    if (!reuse) {
        KTimer t(c, KN_MEASURE);
        hipLaunchKernelGGL(k_measure, dim3(blocks(M, 64)), dim3(64), 0, c->stream, M, c->Ncap, c->Ncap, c->chart, make_cam(cam), useEqv, c->h_lmidx, c->h_y, c->q0(), c->Qq(),
                           c->Qa(), c->d_C, c->d_ytil, c->d_lmidx, c->d_flags);
        HIPCHK(hipGetLastError());
    }
    const bool use_door = c->opt_door && !c->opt_check && !c->obs_pending; // a later kernel (finite check) or the observer stream need the full wait
    const int door_seq = (int)(++c->door_seq);
    rc = launch_update_tail(c, ids, M, meas_var, discreteCorr, nullptr, 0, use_door, door_seq);
    if (rc)
        return rc;
    rc = use_door ? door_wait(c, 1, door_seq) : sync_ctx(c);
    if (rc)
        return rc;
    return finish_update(c, discreteCorr);
}

// See include/eqf_hip.h. Optional hint ahead of the propagation call of the same frame.
int eqf_stage_measurement(eqf_ctx* c, const int* ids, const double* y, int M) {
    HP_SCOPE("abi.stage_measurement");
    if (!c || M < 0 || (M > 0 && (!ids || !y)))
        return EQF_E_BAD_ARG;
    c->stage_requested = c->stage_pending = c->staged_valid = false;
    if (M == 0 || c->N == 0 || M > c->N)
        return 0;
    // only a copy here: this call sits between the previous frame's results and this frame's first launch, where every host
    // microsecond is GPU idle time. The id lookup and the packing run in eqf_propagate_fast, behind the first launch.
    c->staged_ids.assign(ids, ids + M);
    c->staged_y.assign(y, y + 2 * M);
    c->staged_M = M;
    c->stage_requested = true;
    return 0;
}
// second half of eqf_stage_measurement (called with at least one kernel of the frame already queued)
static int stage_prepare(eqf_ctx* c) {
    if (!c->stage_requested)
        return 0;
    c->stage_requested = false;
    const int M = c->staged_M;
    if (c->N == 0 || M > c->N)
        return 0;
    if (c->busy_meas) {
        int r = sync_ctx(c);
        if (r)
            return r;
    }
    int* lmidx = c->h_lmidx;
    int* measof = c->h_lmidx + c->Ncap;
    const int rc = map_measurement(c, c->staged_ids.data(), M, false, lmidx, measof);
    if (rc)
        return 0; // ids not ascending: the update call will report it
    for (int j = 0; j < M; ++j)
        if (lmidx[j] < 0)
            return 0; // an id without a landmark: nothing staged, the update call takes its ordinary route
    std::memcpy(c->h_y, c->staged_y.data(), sizeof(double) * 2 * M);
    pack_by_landmark(c, measof, c->staged_y.data());
    c->staged_gen = c->lm_gen;
    c->stage_pending = true;
    return 0;
}

// See include/eqf_hip.h. Statistics and update queued back to back; the statistics kernel cancels the tail on the device if
// the host has an outlier decision to make.
// max_outliers < 0: the caller decides about outliers itself (eqf_stats_then_update); >= 0: eqf_stats_select_update
static int stats_then_update(eqf_ctx* c, const eqvio_camera* cam, const int* ids, const double* y, int M, double thrAbs, double thrProb, double meas_var, int useEqv,
                             int discreteCorr, double* absErr, double* probErr, double* depth2, int* updated, int max_outliers, int* removed_idx, int* n_removed) {
    HP_SCOPE("abi.stats_then_update");
    if (!c || !cam || !updated || M <= 0 || !ids || !y || !camera_ok(cam))
        return EQF_E_BAD_ARG;
    ++c->spec_calls;
    *updated = 0;
    c->pred_cam = make_cam(cam), c->pred_star = useEqv ? 1 : 0, c->pred_valid = true; // what the next propagation call evaluates the output blocks with (EQF_OPT_MEASURE_IN_PROPAGATE)
    const int N = c->N;
    if (N == 0 || M > N) {
        *updated = -1; // not applicable (a landmark has to be added first)
        return 0;
    }
    { int _e = enter(c); if (_e) return _e; }
    host_stamp(c, TH_TAIL_ENTRY);
    { int _r = join_observer(c); if (_r) return _r; } // before the flags below are evaluated, as in eqf_outlier_stats: a stand-alone observer call does not cost the frame its doorbell
    // speculation needs the doorbell-free conditions of both waits and the equivariant-output cache of the statistics kernel
    // A cancelled tail costs the frame a wasted launch sequence on top of the two-round-trip path it falls back to. Where outlier candidates
    // show up frame after frame (tight thresholds) speculation backs off: after a cancellation the next 1, 2, 4 .. 16 frames only compute
    // the statistics; a frame without a candidate resets it. A scheduling choice only: both paths give bit-identical results.
    bool speculate = c->opt_spec && !c->opt_check && !c->obs_pending;
    if (speculate && c->opt_spec == 1 && c->spec_backoff > 0) {
        --c->spec_backoff;
        speculate = false;
    }
    const bool use_door = c->opt_door && !c->opt_check && !c->obs_pending;
    // staged by eqf_stage_measurement and copied to HBM by the propagation kernel: same measurement, same landmark set?
    // (the pinned packet holds exactly this measurement, mapped and packed by stage_prepare: nothing to rewrite - and no wait for the propagation kernel's staging block,
    //  which may still be reading it - whichever route the frame takes)
    const bool same_as_staged = c->staged_valid && c->staged_gen == c->lm_gen && c->staged_M == M && std::equal(ids, ids + M, c->staged_ids.begin()) &&
                                std::memcmp(c->staged_y.data(), y, sizeof(double) * 2 * M) == 0;
    const bool staged = speculate && same_as_staged;
    c->staged_valid = c->stage_pending = c->stage_requested = false;
    int* lmidx = c->h_lmidx;
    int* measof = c->h_lmidx + c->Ncap;
    int rc = 0;
    if (!same_as_staged) {
        if (c->busy_meas) {
            int r = sync_ctx(c);
            if (r)
                return r;
        }
        rc = map_measurement(c, ids, M, false, lmidx, measof);
        if (rc)
            return rc;
        for (int j = 0; j < M; ++j)
            if (lmidx[j] < 0) { // a measurement without a landmark: the caller adds landmarks first (VIOFilter.cpp:217), nothing queued
                *updated = -1;
                return 0;
            }
        std::memcpy(c->h_y, y, sizeof(double) * 2 * M);
    }
    HP_SCOPE("stu.after_map");
    const int seq = (int)(++c->door_seq);
    auto copy_stats = [&]() {
        if (absErr)
            std::memcpy(absErr, c->h_res, sizeof(double) * N);
        if (probErr)
            std::memcpy(probErr, c->h_res + N, sizeof(double) * N);
        if (depth2)
            std::memcpy(depth2, c->h_res + 2 * N, sizeof(double) * N);
    };
    c->meas_star = useEqv ? 1 : 0;
    c->meas_ids.assign(ids, ids + M);
    c->busy_meas = true;
    if (!speculate && max_outliers >= 0 && N <= SEL_MAXN) {
        // Outlier candidates frame after frame (speculation has backed off): statistics, the outlier decision (k_select_outliers: the discarded
        // landmarks' measurements are masked out of C) and the whole update queued at once, ONE host wait. The discarded landmarks leave the
        // state after the update (an unmeasured landmark can be marginalised before or after it).
        // EQF_OPT_LIVE_COLUMNS_FIRST: up to 16 panels the look-ahead kernel ends with the last panel that holds a column of a landmark that stays (k_stats_select orders them)
        const bool live_first = N <= SEL_ONE_WG && c->opt_sel_one && c->opt_live_first && blocks(2 * M, 32) <= 16 && blocks(2 * M, 32) >= 3;
        if (N <= SEL_ONE_WG && c->opt_sel_one) { // statistics and decision as one launch of one workgroup
            KTimer t(c, KN_STATS);
            LAUNCH_TS(c, k_stats_select, dim3(1), dim3(512), c->stream, N, c->Ncap, c->ld, c->chart, make_cam(cam),
                      same_as_staged ? (const double*)(c->d_meas + 2 * (size_t)c->Ncap) /* the staged copy in HBM: no PCIe round trip in front of the statistics */ : pack_by_landmark(c, measof, y), c->q0(), c->Qq(), c->Qa(),
                      (const TS*)c->sigma(), c->h_res, useEqv ? 1 : 0, c->d_C, c->d_ytil, c->d_lmidx, c->d_flags, thrAbs, thrProb, max_outliers, M, c->h_sel, live_first ? c->d_spec + 2 : (int*)nullptr);
            HIPCHK(hipGetLastError());
        } else {
            KTimer t(c, KN_STATS);
            LAUNCH_TS(c, k_outlier_stats, dim3(blocks(N, 64)), dim3(64), c->stream, N, c->Ncap, c->ld, c->chart, make_cam(cam), same_as_staged ? (const double*)c->h_ylm : pack_by_landmark(c, measof, y), c->q0(), c->Qq(),
                      c->Qa(), (const TS*)c->sigma(), c->h_res, useEqv ? 1 : 0, c->d_C, c->d_ytil, c->d_lmidx, c->d_flags, (int*)nullptr, c->h_door, seq, thrAbs, thrProb,
                      (int*)nullptr, seq, c->d_stats);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(k_select_outliers, dim3(1), dim3(256), 0, c->stream, N, c->Ncap, M, c->d_stats, thrAbs, thrProb, max_outliers, c->d_lmidx, c->d_C, c->d_ytil, c->h_sel);
            HIPCHK(hipGetLastError());
        }
        c->meas_valid = false;
        c->early_allowed = c->opt_early_door != 0;
        rc = launch_update_tail(c, ids, M, meas_var, discreteCorr, nullptr, 0, use_door, seq, nullptr, live_first);
        if (rc)
            return rc;
        host_stamp(c, TH_TAIL_OUT);
        bool early = false;
        rc = use_door ? door_wait(c, 1, seq, &early) : sync_ctx(c);
        if (rc)
            return rc;
        copy_stats();
        ++c->sel_frames;
        if (c->h_sel[c->Ncap] == 0) // a frame without an outlier candidate ends the back-off
            c->spec_backoff = c->spec_backoff_len = 0;
        rc = early ? finish_update_early(c, discreteCorr, seq) : finish_update(c, discreteCorr);
        if (rc)
            return rc;
        *updated = 1;
        std::vector<int> idx;
        for (int i = 0; i < N; ++i)
            if (c->h_sel[i])
                idx.push_back(i);
        if (n_removed)
            *n_removed = (int)idx.size();
        if (removed_idx)
            std::copy(idx.begin(), idx.end(), removed_idx);
        c->sel_discarded += (long)idx.size();
        return idx.empty() ? 0 : eqf_remove_landmarks(c, idx.data(), (int)idx.size());
    }
    if (!speculate) { // plain statistics call: the caller decides and calls eqf_vision_update
        {
            KTimer t(c, KN_STATS);
            LAUNCH_TS(c, k_outlier_stats, dim3(blocks(N, 64)), dim3(64), c->stream, N, c->Ncap, c->ld, c->chart, make_cam(cam), same_as_staged ? (const double*)c->h_ylm : pack_by_landmark(c, measof, y), c->q0(), c->Qq(),
                      c->Qa(), (const TS*)c->sigma(), c->h_res, useEqv ? 1 : 0, c->d_C, c->d_ytil, c->d_lmidx, c->d_flags, use_door ? c->d_door : nullptr, c->h_door, seq, thrAbs,
                      thrProb, (int*)nullptr, seq, (double*)nullptr);
            HIPCHK(hipGetLastError());
        }
        c->meas_valid = true;
        rc = use_door ? door_wait(c, 0, seq) : sync_ctx(c);
        if (rc)
            return rc;
        copy_stats();
        bool candidate = false; // a frame without an outlier candidate ends the back-off
        for (int i = 0; i < N && !candidate; ++i)
            candidate = c->h_res[i] > thrAbs || c->h_res[N + i] > thrProb;
        if (!candidate)
            c->spec_backoff = c->spec_backoff_len = 0;
        return 0;
    }
    // Speculative tail: measurement, statistics and Z in ONE kernel (k_build_Z with measurement fusion), then the factorisation,
    // the lift and the covariance update, all queued at once. An outlier candidate cancels everything behind the first kernel.
    MeasFuse mf{};
    mf.enabled = 1;
    mf.N = N, mf.Ncap = c->Ncap, mf.chart = c->chart, mf.star = useEqv ? 1 : 0;
    mf.cam = make_cam(cam);
    if (staged)
        mf.y = c->d_meas, mf.lmidx = c->d_meas_idx, mf.ylm = c->d_meas + 2 * (size_t)c->Ncap;
    else
        mf.y = c->h_y, mf.lmidx = lmidx, mf.ylm = pack_by_landmark(c, measof, y);
    mf.q0 = c->q0(), mf.Qq = c->Qq(), mf.Qa = c->Qa();
    mf.out = c->h_res;
    mf.C = c->d_C, mf.ytil = c->d_ytil, mf.lmidx_dev = c->d_lmidx;
    mf.thrAbs = thrAbs, mf.thrProb = thrProb;
    mf.spec_w = c->d_spec, mf.spec_seq = seq;
    c->meas_valid = false; // consumed by the tail below (restored if the tail is cancelled)
    c->early_allowed = c->opt_early_door != 0;
    rc = launch_update_tail(c, ids, M, meas_var, discreteCorr, c->d_spec, seq, use_door, seq, &mf);
    if (rc)
        return rc;
    host_stamp(c, TH_TAIL_OUT);
    bool early = false;
    rc = use_door ? door_wait(c, 1, seq, &early) : sync_ctx(c);
    if (rc)
        return rc;
    copy_stats();
    ++c->spec_queued;
    if (early) { // (the early doorbell does not ring for a cancelled tail)
        c->spec_backoff = c->spec_backoff_len = 0;
        rc = finish_update_early(c, discreteCorr, seq);
        if (rc == 0)
            *updated = 1;
        return rc;
    }
    if (c->h_resflags[2]) { // cancelled on the device: nothing was modified, C / residuals of the statistics kernel are still valid
        ++c->spec_cancelled;
        // (up to 256 frames when the caller lets the device take the outlier decision: a frame with candidates then costs one round trip anyway, a cancelled tail two and a wasted
        //  launch sequence - with candidates in nearly every frame, the shipped thresholds, the cap of 16 wasted a tail every 17th frame)
        c->spec_backoff_len = std::min(max_outliers >= 0 ? 256 : 16, std::max(1, 2 * c->spec_backoff_len));
        c->spec_backoff = c->spec_backoff_len;
        c->meas_valid = true;
        return 0;
    }
    c->spec_backoff = c->spec_backoff_len = 0;
    rc = finish_update(c, discreteCorr);
    if (rc == 0)
        *updated = 1;
    return rc;
}

int eqf_stats_then_update(eqf_ctx* c, const eqvio_camera* cam, const int* ids, const double* y, int M, double thrAbs, double thrProb, double meas_var, int useEqv,
                          int discreteCorr, double* absErr, double* probErr, double* depth2, int* updated) {
    return stats_then_update(c, cam, ids, y, M, thrAbs, thrProb, meas_var, useEqv, discreteCorr, absErr, probErr, depth2, updated, -1, nullptr, nullptr);
}
int eqf_stats_select_update(eqf_ctx* c, const eqvio_camera* cam, const int* ids, const double* y, int M, double thrAbs, double thrProb, int max_outliers, double meas_var,
                            int useEqv, int discreteCorr, double* absErr, double* probErr, double* depth2, int* updated, int* removed_idx, int* n_removed) {
    if (max_outliers < 0 || !n_removed)
        return EQF_E_BAD_ARG;
    *n_removed = 0;
    return stats_then_update(c, cam, ids, y, M, thrAbs, thrProb, meas_var, useEqv, discreteCorr, absErr, probErr, depth2, updated, max_outliers, removed_idx, n_removed);
}
int eqf_selection_stats(eqf_ctx* c, long* frames, long* discarded, int reset) {
    if (!c)
        return EQF_E_BAD_ARG;
    if (frames)
        *frames = c->sel_frames;
    if (discarded)
        *discarded = c->sel_discarded;
    if (reset)
        c->sel_frames = c->sel_discarded = 0;
    return 0;
}

int eqf_last_gamma(eqf_ctx* c, double* out, int cap) {
    if (!c || !out)
        return EQF_E_BAD_ARG;
    { int _r = keep_last_gamma(c); if (_r) return _r; }
    if ((int)c->last_gamma.size() > cap)
        return EQF_E_CAPACITY;
    std::memcpy(out, c->last_gamma.data(), sizeof(double) * c->last_gamma.size());
    return (int)c->last_gamma.size();
}

int eqf_compute_nees(eqf_ctx* c, const double* ts, const int* tids, const double* tp, int ntrue, double* nees) {
    if (!c || !ts || !nees || ntrue < 0 || (ntrue > 0 && (!tids || !tp)))
        return EQF_E_BAD_ARG;
    { int _e = enter(c); if (_e) return _e; }
    const int N = c->N, n = c->n();
    const int np = n + (n & 1); // even dimension for the 2x2-pivot elimination: pad with a unit diagonal entry
    // landmark group elements from the device
    std::vector<double> q0(3 * (size_t)N + 3), Q(5 * (size_t)N + 5);
    std::vector<int> ids(N + 1);
    double s0[23], g0[23];
    int rc = eqf_get_state(c, s0, g0, ids.data(), q0.data(), Q.data(), N);
    if (rc < 0)
        return rc;
    // stateError = stateGroupAction(X^-1, truncated true state) (VIOGroup.cpp:25-55, 108-120)
    const SensorState tsn = unpack_sensor(ts);
    GroupSensor Xi;
    Xi.bgyr = -c->X.bgyr;
    Xi.bacc = -c->X.bacc;
    Xi.A = pose_inv(c->X.A);
    Xi.B = pose_inv(c->X.B);
    Xi.w = -q_rot(q_inv(c->X.A.R), c->X.w);
    const SensorState se = sensor_action(Xi, tsn);
    std::vector<double> eps(np, 0.0);
    // sensorChart_std (VIOState.cpp:104-113); Normal chart: sensorChart_normal (:123-137)
    const V3 db = se.bgyr - c->xi0.bgyr, da = se.bacc - c->xi0.bacc;
    V3 dv = se.vel - c->xi0.vel;
    V3 om, tr, omc, trc;
    const Pose Arel = pose_mul(pose_inv(c->xi0.pose), se.pose);
    se3_log(Arel, om, tr);
    if (c->chart == EQVIO_COORD_NORMAL) {
        // SE_2(3).log(A.R, A.x, v_A), v_A = R0^T (R v - R0 v0); B = T0^-1 A T
        const V3 vA = q_rot(q_inv(c->xi0.pose.R), q_rot(se.pose.R, se.vel) - q_rot(c->xi0.pose.R, c->xi0.vel));
        dv = so3_Vinv(om) * vA;
        se3_log(pose_mul(pose_mul(pose_inv(c->xi0.cam), Arel), se.cam), omc, trc);
    } else
        se3_log(pose_mul(pose_inv(c->xi0.cam), se.cam), omc, trc);
    const V3 parts[7] = {db, da, om, tr, dv, omc, trc};
    for (int b = 0; b < 7; ++b)
        pack_v3(parts[b], eps.data() + 3 * b);
    for (int i = 0; i < N; ++i) {
        int jt = -1;
        for (int t = 0; t < ntrue; ++t)
            if (tids[t] == ids[i]) {
                jt = t;
                break;
            }
        if (jt < 0)
            return EQF_E_BAD_ARG; // the reference asserts the true state holds every filter landmark
        const V3 ptrue = v3(tp[3 * jt], tp[3 * jt + 1], tp[3 * jt + 2]);
        const Qt qi{Q[5 * i], Q[5 * i + 1], Q[5 * i + 2], Q[5 * i + 3]};
        const V3 pe = Q[5 * i + 4] * q_rot(qi, ptrue); // (Q_i^-1)^-1 * p = a R p
        const V3 q0i = v3(q0[3 * i], q0[3 * i + 1], q0[3 * i + 2]);
        const V3 e = c->chart == EQVIO_COORD_NORMAL ? normal_chart(pe, q0i) : point_chart(c->chart == EQVIO_COORD_INVDEPTH, pe, q0i);
        pack_v3(e, eps.data() + 21 + 3 * i);
    }
    // device: Z = [Sigma ; eps^T], factorise, NEES = |z|^2 / n
    if (!c->d_Zn) {
        const int npcap = c->ncap + 1;
        c->ldzn = pick_ld(npcap + 1);
        HIPCHK(hipMalloc(&c->d_Zn, sizeof(double) * (size_t)c->ldzn * npcap));
        HIPCHK(hipMalloc(&c->d_Wn, sizeof(double) * (size_t)c->ldzn * npcap));
        HIPCHK(hipMalloc(&c->d_perm, sizeof(int) * 2 * (size_t)(c->ncap + 2)));
    }
    { int _r = keep_last_gamma(c); if (_r) return _r; } // d_gamma serves as the staging buffer of eps below
    { int _r = sync_ctx(c); if (_r) return _r; }
    std::memcpy(c->h_buf, eps.data(), sizeof(double) * np);
    HIPCHK(hipMemcpyAsync(c->d_gamma, c->h_buf, sizeof(double) * np, hipMemcpyHostToDevice, c->stream)); // d_gamma as staging (ncap >= np? see below)
    HIPCHK(hipMemsetAsync(c->d_flags, 0, sizeof(int) * 4, c->stream));
    LAUNCH_TS(c, k_build_nees, dim3(blocks(np + 1, 256), np), dim3(256), c->stream, n, np, c->ld, c->ldzn, (const TS*)c->sigma(), c->d_gamma, c->d_Zn);
    HIPCHK(hipGetLastError());
    rc = launch_chain(c, np + 1, np, c->ldzn, c->d_Zn, c->d_Wn);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k_sumsq_row, dim3(1), dim3(256), 0, c->stream, np, c->ldzn, c->d_Wn, np, c->d_stats);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_buf, c->d_stats, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    rc = read_flags(c);
    if (rc)
        return rc;
    if (c->h_flags[0]) {
        // Sigma is positive definite only up to rounding. The reference inverts it by partial-pivot LU and returns a number regardless
        // (VIO_eqf.cpp:166-168): Gaussian elimination with partial pivoting on [Sigma | eps], on the device (k_ge_step, one launch per pivot)
        ++c->nees_lu_fallbacks;
        std::memcpy(c->h_buf, eps.data(), sizeof(double) * np);
        HIPCHK(hipMemcpyAsync(c->d_gamma, c->h_buf, sizeof(double) * np, hipMemcpyHostToDevice, c->stream));
        LAUNCH_TS(c, k_build_nees, dim3(blocks(np + 1, 256), np), dim3(256), c->stream, n, np, c->ld, c->ldzn, (const TS*)c->sigma(), c->d_gamma, c->d_Zn);
        hipLaunchKernelGGL(k_iota, dim3(blocks(np, 256)), dim3(256), 0, c->stream, np, c->d_perm);
        const int pstride = c->ncap + 2;
        for (int k = 0; k < np; ++k)
            hipLaunchKernelGGL(k_ge_step, dim3(std::max(1, blocks(np - k - 1, GE_ROWS))), dim3(256), 0, c->stream, np, c->ldzn, k, c->d_Zn, c->d_perm + (k & 1) * pstride,
                               c->d_perm + ((k + 1) & 1) * pstride);
        hipLaunchKernelGGL(k_ge_back, dim3(1), dim3(1024), sizeof(double) * (np + 16), c->stream, n, np, c->ldzn, c->d_Zn, c->d_perm + (np & 1) * pstride, c->d_gamma, c->d_stats);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->h_buf, c->d_stats, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        { int _r = sync_ctx(c); if (_r) return _r; }
    }
    *nees = c->h_buf[0] / (double)n;
    return 0;
}

int eqf_debug_get_W(eqf_ctx* c, double* out, int rows, int cols) {
    if (!c || !out || rows <= 0 || cols <= 0 || rows > c->ldz || cols > c->mcap)
        return EQF_E_BAD_ARG;
    { int _r = sync_ctx(c); if (_r) return _r; }
    HIPCHK(hipMemcpy2D(out, sizeof(double) * rows, c->d_W, sizeof(double) * c->ldz, sizeof(double) * rows, cols, hipMemcpyDeviceToHost));
    return 0;
}

int eqf_debug_lookahead_stamps(eqf_ctx* c, unsigned long long* out768) {
    if (!c || !out768 || !c->d_ladbg)
        return EQF_E_BAD_ARG;
    { int _r = sync_ctx(c); if (_r) return _r; }
    HIPCHK(hipMemcpy(out768, c->d_ladbg, 96 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

int eqf_measure_in_propagate_stats(eqf_ctx* c, long* used, int reset) {
    if (!c || !used)
        return EQF_E_BAD_ARG;
    *used = c->me_used;
    if (reset)
        c->me_used = 0;
    return 0;
}
int eqf_z_in_lookahead_stats(eqf_ctx* c, long* launches, int reset) {
    if (!c || !launches)
        return EQF_E_BAD_ARG;
    *launches = c->zb_launches;
    if (reset)
        c->zb_launches = 0;
    return 0;
}
int eqf_gather_stats(eqf_ctx* c, long* launches, int reset) {
    if (!c || !launches)
        return EQF_E_BAD_ARG;
    *launches = c->gather_launches;
    if (reset)
        c->gather_launches = 0;
    return 0;
}
int eqf_speculation_stats(eqf_ctx* c, long* calls, long* queued, long* cancelled, int reset) {
    if (!c || !calls || !queued || !cancelled)
        return EQF_E_BAD_ARG;
    *calls = c->spec_calls;
    *queued = c->spec_queued;
    *cancelled = c->spec_cancelled;
    if (reset)
        c->spec_calls = c->spec_queued = c->spec_cancelled = 0;
    return 0;
}

int eqf_debug_syrk_order(int nt, int* tile_of_block) {
    if (nt < 1 || nt > 4095 || !tile_of_block)
        return EQF_E_BAD_ARG;
    build_syrk_order(nt, tile_of_block);
    return 0;
}
int eqf_lookahead_stats(eqf_ctx* c, long* launches, long* fallbacks, int reset) {
    if (!c)
        return EQF_E_BAD_ARG;
    if (launches)
        *launches = c->la_launches;
    if (fallbacks)
        *fallbacks = c->la_fallbacks;
    if (reset)
        c->la_launches = c->la_fallbacks = 0;
    return 0;
}

int eqf_lookahead_selftest(const eqf_ctx* c) { return c ? c->la_selftest : EQF_E_BAD_ARG; }
int eqf_lookahead_home(const eqf_ctx* c, int* home_xcd, long* home_launches) {
    if (!c)
        return EQF_E_BAD_ARG;
    if (home_xcd)
        *home_xcd = (c->opt_la_home && c->cu_count == 256) ? c->la_home : -1;
    if (home_launches)
        *home_launches = c->la_home_launches;
    return 0;
}

int eqf_early_doorbell_stats(eqf_ctx* c, long* updates, int reset) {
    if (!c)
        return EQF_E_BAD_ARG;
    if (updates)
        *updates = c->early_rings;
    if (reset)
        c->early_rings = 0;
    return 0;
}
int eqf_device_to_itself(eqf_ctx* c) { return c ? (device_to_itself(c) ? 1 : 0) : EQF_E_BAD_ARG; }

int eqf_nees_lu_fallbacks(eqf_ctx* c, long* count) {
    if (!c || !count)
        return EQF_E_BAD_ARG;
    *count = c->nees_lu_fallbacks;
    return 0;
}

int eqf_debug_matrices_AB(eqf_ctx* c, const double* imu13, double* A_out, double* B_out) {
    if (!c || !imu13)
        return EQF_E_BAD_ARG;
    { int _e = enter(c); if (_e) return _e; }
    { int _r = sync_ctx(c); if (_r) return _r; }
    int rc = upload_common(c, imu13);
    if (rc)
        return rc;
    rc = launch_assemble(c);
    if (rc)
        return rc;
    const int N = c->N, n = c->n(), Ncap = c->Ncap;
    std::vector<double> Al(45 * (size_t)Ncap), Bl(9 * (size_t)Ncap);
    HIPCHK(hipMemcpyAsync(c->h_buf, c->d_Al, sizeof(double) * 45 * Ncap, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_buf + 45 * (size_t)Ncap, c->d_Bl, sizeof(double) * 9 * Ncap, hipMemcpyDeviceToHost, c->stream));
    { int _r = sync_ctx(c); if (_r) return _r; }
    std::memcpy(Al.data(), c->h_buf, sizeof(double) * 45 * Ncap);
    std::memcpy(Bl.data(), c->h_buf + 45 * (size_t)Ncap, sizeof(double) * 9 * Ncap);
    const Common& cm = *c->h_common;
    if (A_out) {
        std::fill(A_out, A_out + (size_t)n * n, 0.0);
        for (int r = 0; r < 21; ++r)
            for (int cc = 0; cc < 21; ++cc)
                A_out[r + (size_t)cc * n] = cm.Ass[r * 21 + cc];
        for (int i = 0; i < N; ++i)
            for (int r = 0; r < 3; ++r) {
                for (int e = 0; e < 12; ++e)
                    A_out[21 + 3 * i + r + (size_t)al_col(e) * n] = Al[(r * 15 + e) * (size_t)Ncap + i];
                for (int cc = 0; cc < 3; ++cc)
                    A_out[21 + 3 * i + r + (size_t)(21 + 3 * i + cc) * n] = Al[(r * 15 + 12 + cc) * (size_t)Ncap + i];
            }
    }
    if (B_out) {
        std::fill(B_out, B_out + (size_t)n * 12, 0.0);
        for (int r = 0; r < 21; ++r)
            for (int cc = 0; cc < 12; ++cc)
                B_out[r + (size_t)cc * n] = cm.Bs[r * 12 + cc];
        for (int i = 0; i < N; ++i)
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc)
                    B_out[21 + 3 * i + r + (size_t)cc * n] = Bl[(r * 3 + cc) * (size_t)Ncap + i];
    }
    if (c->chart == EQVIO_COORD_NORMAL) {
        // what was expanded above is the Euclidean pair (the device propagates M^-1 Sigma M^-T with it); the Normal suite's matrices are
        // A_n = M A_e M^-1, B_n = M B_e (normal.cpp:37-45) with the closed-form block-diagonal M
        std::vector<double> q0(3 * (size_t)N + 3), Q(5 * (size_t)N + 5);
        std::vector<int> ids(N + 1);
        double s0[23], g0[23];
        rc = eqf_get_state(c, s0, g0, ids.data(), q0.data(), Q.data(), N);
        if (rc < 0)
            return rc;
        std::vector<double> Md((size_t)n * n, 0.0), Mi((size_t)n * n, 0.0); // column-major
        for (int i = 0; i < n; ++i)
            Md[i + (size_t)i * n] = Mi[i + (size_t)i * n] = 1.0;
        const M3 K = skew(c->xi0.vel);
        const M6 Ad = se3_Adjoint(pose_inv(c->xi0.cam));
        const double Kd[9] = {K.a00, K.a01, K.a02, K.a10, K.a11, K.a12, K.a20, K.a21, K.a22};
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) {
                Md[12 + r + (size_t)(6 + q) * n] = -Kd[3 * r + q];
                Mi[12 + r + (size_t)(6 + q) * n] = Kd[3 * r + q];
            }
        for (int r = 0; r < 6; ++r)
            for (int q = 0; q < 6; ++q) {
                Md[15 + r + (size_t)(6 + q) * n] = Ad.a[6 * r + q];
                Mi[15 + r + (size_t)(6 + q) * n] = -Ad.a[6 * r + q];
            }
        for (int i = 0; i < N; ++i) {
            const V3 p0 = v3(q0[3 * i], q0[3 * i + 1], q0[3 * i + 2]);
            const M3 Mb = normal_M(p0), Mv = normal_Minv(p0);
            const double mb[9] = {Mb.a00, Mb.a01, Mb.a02, Mb.a10, Mb.a11, Mb.a12, Mb.a20, Mb.a21, Mb.a22};
            const double mv[9] = {Mv.a00, Mv.a01, Mv.a02, Mv.a10, Mv.a11, Mv.a12, Mv.a20, Mv.a21, Mv.a22};
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) {
                    Md[21 + 3 * i + r + (size_t)(21 + 3 * i + q) * n] = mb[3 * r + q];
                    Mi[21 + 3 * i + r + (size_t)(21 + 3 * i + q) * n] = mv[3 * r + q];
                }
        }
        auto mul = [n](const std::vector<double>& X, const double* Y, int cols, std::vector<double>& Z) { // Z = X Y, X n x n with few non-zeros
            Z.assign((size_t)n * cols, 0.0);
            for (int k = 0; k < n; ++k)
                for (int i = 0; i < n; ++i) {
                    const double x = X[i + (size_t)k * n];
                    if (x != 0.0)
                        for (int j = 0; j < cols; ++j)
                            Z[i + (size_t)j * n] += x * Y[k + (size_t)j * n];
                }
        };
        std::vector<double> T1, T2;
        if (A_out) {
            mul(Md, A_out, n, T1);                        // M A_e
            T2.assign((size_t)n * n, 0.0);                // (M A_e) M^-1
            for (int k = 0; k < n; ++k)
                for (int j = 0; j < n; ++j) {
                    const double x = Mi[k + (size_t)j * n];
                    if (x != 0.0)
                        for (int i = 0; i < n; ++i)
                            T2[i + (size_t)j * n] += T1[i + (size_t)k * n] * x;
                }
            std::copy(T2.begin(), T2.end(), A_out);
        }
        if (B_out) {
            mul(Md, B_out, 12, T1);
            std::copy(T1.begin(), T1.end(), B_out);
        }
    }
    return 0;
}

int eqf_debug_matrix_C(eqf_ctx* c, const eqvio_camera* cam, const int* ids, const double* y, int M, int useEqv, double* C_out, double* ytilde_out) {
    if (!c || !cam || M <= 0 || !ids || !y || M > c->N)
        return EQF_E_BAD_ARG;
    { int _e = enter(c); if (_e) return _e; }
    int rc = stage_measurement(c, ids, y, M);
    if (rc)
        return rc;
    std::vector<int> lmidx(c->h_lmidx, c->h_lmidx + M);
    hipLaunchKernelGGL(k_measure, dim3(blocks(M, 64)), dim3(64), 0, c->stream, M, c->Ncap, c->Ncap, c->chart, make_cam(cam), useEqv, c->h_lmidx, c->h_y, c->q0(), c->Qq(), c->Qa(),
                       c->d_C, c->d_ytil, c->d_lmidx, c->d_flags);
    HIPCHK(hipGetLastError());
    const int Ncap = c->Ncap, n = c->n();
    HIPCHK(hipMemcpyAsync(c->h_buf, c->d_C, sizeof(double) * 6 * Ncap, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_buf + 6 * (size_t)Ncap, c->d_ytil, sizeof(double) * 2 * M, hipMemcpyDeviceToHost, c->stream));
    { int _r = sync_ctx(c); if (_r) return _r; }
    if (C_out) {
        std::fill(C_out, C_out + (size_t)2 * M * n, 0.0);
        for (int j = 0; j < M; ++j)
            for (int k = 0; k < 2; ++k)
                for (int cc = 0; cc < 3; ++cc)
                    C_out[2 * j + k + (size_t)(21 + 3 * lmidx[j] + cc) * (2 * M)] = c->h_buf[(k * 3 + cc) * (size_t)Ncap + j];
    }
    if (ytilde_out)
        std::memcpy(ytilde_out, c->h_buf + 6 * (size_t)Ncap, sizeof(double) * 2 * M);
    return 0;
}

int eqf_mfma_f64_peak_clock(eqf_ctx* c, double* tflops, double* sclk_ghz) {
    if (!c || !tflops)
        return EQF_E_BAD_ARG;
    { int _e = enter(c); if (_e) return _e; }
    const int nblk = 256 * 8, iters = 4096;
    double* d_out = nullptr;
    unsigned long long* d_clk = nullptr;
    HIPCHK(hipMalloc(&d_out, sizeof(double) * nblk * 256));
    HIPCHK(hipMalloc(&d_clk, sizeof(unsigned long long) * 2 * (nblk / 64)));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mfma_peak, dim3(nblk), dim3(256), 0, c->stream, 64, d_out, (unsigned long long*)nullptr); // warm-up
    double best = 0, clock_at_best = 0;
    for (int rep = 0; rep < 5; ++rep) {
        HIPCHK(hipEventRecord(e0, c->stream));
        hipLaunchKernelGGL(k_mfma_peak, dim3(nblk), dim3(256), 0, c->stream, iters, d_out, d_clk);
        HIPCHK(hipEventRecord(e1, c->stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)nblk * 4 /*waves*/ * iters * 4 /*mfma*/ * (2.0 * 16 * 16 * 4);
        const double tf = flops / (ms * 1e-3) / 1e12;
        if (tf > best) {
            unsigned long long h[2 * (nblk / 64)];
            HIPCHK(hipMemcpy(h, d_clk, sizeof(h), hipMemcpyDeviceToHost));
            double cyc = 0, ticks = 0;
            for (int q = 0; q < nblk / 64; ++q)
                cyc += (double)h[2 * q], ticks += (double)h[2 * q + 1];
            best = tf;
            clock_at_best = ticks > 0 ? cyc / (ticks * 10.0) : 0.0; // cycles / (ticks x 1e-8 s) / 1e9 = GHz
        }
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d_out);
    hipFree(d_clk);
    *tflops = best;
    if (sclk_ghz)
        *sclk_ghz = clock_at_best;
    return 0;
}
int eqf_mfma_f64_peak(eqf_ctx* c, double* tflops) { return eqf_mfma_f64_peak_clock(c, tflops, nullptr); }

int eqf_trace_read(eqf_ctx* c, unsigned long long* device_ticks, long long* host_ns, unsigned* last_frame) {
    if (!c || !device_ticks || !host_ns || !last_frame)
        return EQF_E_BAD_ARG;
    if (!c->d_trace)
        return EQF_E_UNSUPPORTED;
    { int _r = sync_ctx(c); if (_r) return _r; }
    HIPCHK(hipMemcpy(device_ticks, c->d_trace, sizeof(trace_t) * TR_FRAMES * TR_SLOTS, hipMemcpyDeviceToHost));
    std::memcpy(host_ns, c->h_trace.data(), sizeof(long long) * TR_FRAMES * TR_HOST);
    *last_frame = c->trace_frame;
    return 0;
}

int eqf_host_wait_stats(eqf_ctx* c, long* calls, double* seconds, int reset) {
    if (!c || !calls || !seconds)
        return EQF_E_BAD_ARG;
    calls[0] = c->wait_calls;
    seconds[0] = c->wait_seconds;
    calls[1] = c->launch_calls;
    seconds[1] = c->launch_seconds;
    if (reset) {
        c->wait_calls = c->launch_calls = 0;
        c->wait_seconds = c->launch_seconds = 0.0;
    }
    return 0;
}

int eqf_last_kernel_times(eqf_ctx* c, int* which, float* usec, int cap) {
    if (!c)
        return EQF_E_BAD_ARG;
    { int _r = sync_ctx(c); if (_r) return _r; }
    int cnt = 0;
    for (size_t q = 0; q < c->tev.size(); ++q) {
        auto& t = c->tev[q];
        float ms = 0;
        hipEventElapsedTime(&ms, t.second.first, t.second.second);
        const int nl = std::max(1, c->tev_n[q]);
        for (int r = 0; r < nl && cnt < cap; ++r) { // a span over nl back-to-back launches: nl entries of span / nl
            which[cnt] = t.first;
            usec[cnt] = ms * 1000.0f / nl;
            ++cnt;
        }
    }
    timing_reset(c);
    return cnt;
}

} // extern "C"
