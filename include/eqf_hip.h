/* eqf_hip.h — C-ABI of the MI355X (gfx950) EqF core: "VIO_eqf on the GPU".
 *
 * One eqf_ctx = one reference `struct VIO_eqf` (include/eqvio/mathematical/VIO_eqf.h:34-134): the origin
 * xi0, the observer state X and the Riccati matrix Sigma. Sigma (n x n fp64, n = 21 + 3N) and the
 * per-landmark arrays (q0_i, Q_i) are DEVICE resident; the 46 doubles of sensor-level state (xi0.sensor,
 * X.{beta,A,w,B}) are host resident inside the context. Every entry point replaces one VIO_eqf member (cited).
 * Plain pointers and sizes only; flat layouts are those of eqvio_types.h. All matrices column-major
 * (Eigen default). Measurement arrays must be sorted by ASCENDING id — the row order the reference gets from
 * std::map (src/mathematical/VisionMeasurement.cpp:72-79, src/mathematical/EqFMatrices.cpp:58-66).
 *
 * Return value: 0 = ok; >0 = hipError_t; <0 = EQF_E_*. No exceptions, no callbacks, no global state; a
 * context is bound to one HIP device + one stream and must be used from one thread at a time.
 * The library FAILS (EQF_E_NO_DEVICE) when no gfx950 device is present: there is no CPU fallback.
 */
#ifndef EQF_HIP_H
#define EQF_HIP_H
#include "eqvio_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct eqf_ctx eqf_ctx;

enum {
    EQF_OK = 0,
    EQF_E_NONFINITE = -1,  /* NaN/Inf detected in Sigma or X (the reference's assert(!hasNaN())) */
    EQF_E_NOT_SPD = -2,    /* Cholesky pivot <= 0 in S = C Sigma C^T + R */
    EQF_E_BAD_ARG = -3,
    EQF_E_CAPACITY = -4,   /* an output array of the caller is too small; or the device buffers could not be grown (allocation failed) */
    EQF_E_NO_DEVICE = -5,
    EQF_E_UNSUPPORTED = -6, /* option combination not implemented on the device path */
    EQF_E_STALLED = -7      /* a bounded device-side wait of the look-ahead factorisation ran out (its workgroups were not all resident, e.g. the
                               GPU is held by another process) AND the automatic retry of the same factorisation on the launch chain failed as
                               well. A stall alone is not an error: the update is redone on the chain (W and Sigma+ bit-identical, Gamma up to rounding) and counted, see
                               eqf_lookahead_stats. Nothing of the filter was modified when this code is returned. */
};

/* options for eqf_set_option */
enum {
    EQF_OPT_RICCATI_DENSE = 1, /* 1: propagate with two dense fp64 MFMA GEMMs (F Sigma F^T, F materialised);
                                  0 (default): structure-exploiting arrow-form kernel */
    EQF_OPT_CHECK_FINITE = 2,  /* 1: scan Sigma/X for non-finite values after propagate/update */
    EQF_OPT_SPECULATIVE = 7,   /* 1 (default): eqf_stats_then_update queues the update behind the statistics kernel, and backs off where that does not
                                  pay: after a tail the device had to cancel, the next 1, 2, 4 .. 16 calls only compute the statistics (*updated = 0),
                                  until a frame shows no outlier candidate; 2: always queue the tail; 0: never (statistics only) */
    EQF_OPT_DOORBELL = 6,      /* 1 (default): the two per-frame host waits poll a sequence number that the last workgroup of the
                                  kernel writes into the pinned result packet (~6 us earlier than the stream's completion signal);
                                  0: wait on the stream */
    EQF_OPT_EARLY_LIFT = 8,    /* 1 (default): the first workgroups of the covariance-update kernel compute Gamma, lift the landmarks and
                                  ring the doorbell, so the host has the frame's results while Sigma -= W W^T is still running;
                                  0: Gamma from the diagonal tiles and a separate lift kernel after it */
    EQF_OPT_TRACE = 9,         /* 1: the frame's kernels stamp the device wall clock (100 MHz) into a ring (eqf_trace_read); 0 (default): off */
    EQF_OPT_FUSED_ASSEMBLY = 11, /* 1 (default): eqf_propagate_fast has no assembly launch: the propagation kernel's workgroups evaluate the rows of
                                  A and B they need themselves and its observer blocks write the second landmark buffer; 0: k_assemble_AB first */
    EQF_OPT_LOOKAHEAD = 12,    /* 1 (default): the factorisation of [S ; T ; y^T] runs as ONE persistent kernel with a look-ahead schedule (one owner
                                  workgroup walks the pivot chain, one workgroup per 16-row half block row keeps its tiles in registers and follows one to
                                  two panels behind; hand-offs as write-through tile stores + one sequence-numbered flag per tile, never cleared) when the update has 3 .. 32
                                  panels (32 < M <= 512 measurements; with 17 .. 32 panels the half-rows run a look-ahead of their own) and all its workgroups
                                  fit the device at once (one per compute unit); W and Sigma+ bit-identical to the chain, Gamma equal up to rounding (summed in
                                  another order). 0: one launch per panel (k_chol_step). Switched to 0 by the library itself after three stalled launches in
                                  a row (eqf_lookahead_stats); setting it again re-arms it. eqf_create runs a self-test of the kernel against the chain on
                                  a fixed problem and keeps a context to the chain if the two disagree */
    EQF_OPT_LA_TIMEOUT_US = 15, /* bound of every device-side wait inside the look-ahead kernel, microseconds of device wall clock; default 20000 (20 ms).
                                  When it runs out the launch is abandoned and the factorisation redone on the launch chain (same Z: W and Sigma+
                                  bit-identical, Gamma up to rounding). 0 makes every look-ahead
                                  launch stall at its first wait: the test hook for that path */
    EQF_OPT_Z_IN_LOOKAHEAD = 17, /* 1 (default): where it applies (fp64 Sigma, 3 .. 32 panels; 2: 3 .. 16 panels only, round 5's behaviour) there is no k_build_Z launch: the look-ahead kernel's half-rows build their
                                  own 16 rows of Z = [S ; T ; yTilde^T] and its owner the first tile, from k_build_Z's expressions (bit-identical W / Sigma). In the
                                  speculative frame tail (eqf_stats_then_update with the measurement staged by the propagation call) its workgroups evaluate the
                                  output blocks C_j themselves as well and one more workgroup computes the outlier statistics and the speculation word; after
                                  eqf_vision_update's measurement kernel they read C from memory. A stalled launch builds Z with k_build_Z before the retry on
                                  the launch chain. Measured for the speculative tail: N = 50 +2.8 %, N = 100 +1.8 %, N = 200 neutral (the kernel's prologue costs
                                  what the launch saved, and the tail's first launch reaches the GPU late), so that form is used up to 8 panels (N <= 128).
                                  0: k_build_Z always */
    EQF_OPT_LA_SPLIT_ROWS = 18, /* 1 (default): with more than 17 panels (N > 272) the look-ahead kernel holds the S half-rows of the block rows >= 16 with TWO workgroups
                                  each (tile columns split: 8 instead of up to 16 MFMA chains per wave and panel in the early panels, where the trailing update is
                                  MFMA-issue bound per compute unit and those rows handed their tiles to the owner late), if the device has the compute units for the
                                  larger launch (N = 500: 191 instead of 159 workgroups). Same products in the same order per tile: bit-identical. 0: one workgroup
                                  per half-row */
    EQF_OPT_MEASURE_IN_PROPAGATE = 19, /* 1 (default): when a measurement has been staged (eqf_stage_measurement) the observer blocks of eqf_propagate_fast's kernel - which have a
                                  landmark's propagated group element in registers when their chain of steps ends - evaluate its output block C_i / C*_i and residual
                                  there and leave them in memory, with the camera and output choice of the LAST eqf_stats_then_update call. If the next such call asks for
                                  the same (and the staged measurement and landmark set still stand) there is no k_build_Z launch (up to 32 panels since round 6): the look-ahead kernel
                                  builds Z from those blocks and one more workgroup of it computes the outlier statistics (on the launch chain k_build_Z runs
                                  but reads the blocks instead of evaluating them again: 25 -> 15 us at N = 500); otherwise the call takes its ordinary route.
                                  Same function and inputs as the update's own evaluation, compiled in another kernel: the compiler contracts the expressions into
                                  fused multiply-adds differently there, so the results agree to rounding (1e-14 on Sigma), not bit for bit (the other routes do
                                  among themselves; with -ffp-contract=off all agree bitwise). 0: never. eqf_measure_in_propagate_stats counts */
    EQF_OPT_LIFT_WITH_SYRK = 20, /* 1 (default): the landmark lift (result packet, doorbell) and Sigma <- Sigma - W W^T, which both only wait for the factorisation and touch
                                  different data, are ONE launch (the lift's workgroups in front) instead of two in a row: the doorbell rings when it did, Sigma - which the
                                  next frame's first kernel waits for - is complete a lift and a kernel boundary earlier. Same arithmetic per thread: bit-identical.
                                  Not with EQF_OPT_SIGMA_FP32 storage or per-kernel timing. 0: two launches */
    EQF_OPT_LA_HOME = 21,      /* 1 (default): up to 16 panels (N <= 256) on a device of 8 XCDs x 32 compute units the look-ahead kernel keeps its owner and all S half-rows on ONE
                                  XCD (the context's "home", claimed at eqf_create) and their hand-offs inside that XCD's L2 (plain stores, a second flag array): 1.1 instead of
                                  2.0 us per hop. What the T half-rows on the other XCDs read is written through a second time. Same arithmetic: bit-identical to 0 and to the
                                  launch chain. Checked at eqf_create (self-test; every home workgroup compares HW_REG_XCC_ID with the XCD it expects) and switched off for the
                                  context if the device deals its blocks differently. eqf_lookahead_home reports the home and the launches that used it. 0: classic placement.
                                  Only for a filter that has the device to itself (eqf_device_to_itself): contexts register their pid in a shared-memory table per user and
                                  device (/dev/shm/eqf_hip_contexts_u<uid>_d<device>). What that table cannot see - another user's processes, a container with its own pid
                                  namespace sharing /dev/shm (its pids look dead and its slots are reclaimed), a process that crashed (its slots stay until the 1-in-4096
                                  liveness sweep) - costs PERFORMANCE only, never correctness: two filters that both take the placement put their home workgroups on the same
                                  XCD (measured: 26.3 k against 31.3 k updates/s for four filters). Set 0 where filters of several users or containers share a GPU */
    EQF_OPT_TILES_PER_WORKGROUP = 22, /* 1 (default): above 256 landmarks (lower-triangle form of the propagation kernel) a workgroup takes as many consecutive tiles of one block
                                  row as it needs for ALL tile workgroups of the launch to be resident at once (N = 500: 3, 187 workgroups instead of 528 in three rounds) and forms
                                  the i side of its tiles once. Same sums per entry: bit-identical. 0: one tile per workgroup; k > 1: k tiles (tests, A/B) */
    EQF_OPT_GATHER_IN_PROPAGATE = 23, /* 1 (default): landmarks removed since the last kernel (eqf_remove_landmarks records; up to 512 landmarks in the state before, nothing appended in between)
                                  leave the device state INSIDE the kernel of eqf_propagate_fast - it reads Sigma and the landmark planes at their old positions and writes the
                                  other buffers at the new ones anyway - instead of in a compaction pass in front of it: one launch and a copy of Sigma less in every frame of a
                                  feature tracker's normal turnover. Same values, same arithmetic: bit-identical. Not with the Normal chart, the dense mode or the float store;
                                  eqf_gather_stats counts. 0: always the pass */
    EQF_OPT_HOLD_NEW_LANDMARKS = 24, /* 1 (default): eqf_add_landmarks_held is available (with EQF_OPT_GATHER_IN_PROPAGATE, fused assembly, fp64 Sigma, not the Normal chart;
                                  eqf_hold_supported says). 0: it returns EQF_E_UNSUPPORTED and the caller appends its new landmarks behind the propagation */
    EQF_OPT_EARLY_DOORBELL = 27, /* 1 (default): eqf_stats_then_update / eqf_stats_select_update return on a doorbell rung by the look-ahead factorisation kernel itself - by its
                                  last T half-row, when every W row is final, no pivot failed, no wait ran out and the tail was not cancelled: the update WILL be applied, and Gamma's
                                  sensor rows are in the result packet - a kernel boundary and the landmark lift earlier than on the lift's doorbell (6 us at 200 landmarks). The
                                  sensor lift is applied on the host at once; the landmark estimates and their invalid flags arrive with the lift's doorbell, which the next entry
                                  point that needs them waits for (eqf_propagate_fast behind its launch; eqf_update_unsettled says whether that is still to come). A caller that
                                  removes invalid landmarks after every update (VIO_eqf::removeInvalidLandmarks) defers that past the next propagation - a landmark can be
                                  marginalised before or after it, bit for bit the same for everybody else - with eqf_remove_invalid_at_update. 0: the lift's doorbell only.
                                  Built in round 5 and not taken (level at 200 landmarks, where the next propagation waits for the covariance update anyway; the counting cost the
                                  factorisation 2 us); taken in round 6 with a count that needs no barrier, acknowledgement or release: +3 .. 4.5 % at 50, +2.8 % at 100 landmarks -
                                  the sizes the reference's own configurations run at, where the GPU idled 10 us per frame behind the host -, level at 200 */
    EQF_OPT_SELECT_ONE_WORKGROUP = 25, /* 1 (default): up to 512 landmarks, the outlier statistics and the device-side outlier decision of eqf_stats_select_update are one
                                  launch of one workgroup (k_stats_select); 0: two launches (k_outlier_stats, k_select_outliers), as above 512 landmarks. Same results */
    EQF_OPT_LIVE_COLUMNS_FIRST = 26, /* 1 (default): in eqf_stats_select_update up to 16 panels (256 measurements), k_stats_select puts the measurements of the landmarks that stay in
                                  front of the discarded ones and the look-ahead kernel ends with the last panel that holds one of them (the discarded landmarks' columns of Z are
                                  decoupled, W = 0 there): the same Sigma+ up to the rounding of a different column order. 0: measurement order kept, every panel factorised */
    EQF_OPT_SIGMA_FP32 = 3     /* fp32-Sigma path (BASELINE config 5); all arithmetic stays fp64.
                                  2: Sigma is STORED as float in HBM (4 bytes per element; loads widen, stores round). Fast-Riccati
                                     path only: dense / accurate Riccati return EQF_E_UNSUPPORTED.
                                  1: numerical model of the same thing on the fp64 store: Sigma rounded to the nearest float after
                                     every store - bit-identical results to mode 2 (tests/test_gpu_fp32_sigma.py), any mode.
                                  3: a ROUNDING MODEL, NOT A STORE (round 5; it saves neither time nor memory - Sigma stays an fp64 buffer): what a MIXED store would do to the numbers: the 21 x 21 sensor block, the sensor-landmark strips and the 3 x 3 landmark
                                     diagonal blocks keep their doubles (4.6 % of Sigma at 200 landmarks), only the landmark-landmark off-diagonal blocks are
                                     rounded to float after every store. Against the fp64 oracle at <= 200 features: Sigma 1.2e-7, pose 1.0e-8, worst landmark
                                     2.6e-6 (all-float store: 1.3e-5 / 2.4e-6 / 2.0e-4) - SURVEY's 1e-5 landmark bound holds. A model on the fp64 store only: the
                                     HBM layout that goes with it (a float matrix + fp64 side arrays for the kept parts) is not built.
                                  0: fp64 (default). Switching converts the live Sigma. */
};

const char* eqf_error_string(int code);

/* lifecycle. coordinate_choice: EQVIO_COORD_EUCLIDEAN | EQVIO_COORD_INVDEPTH | EQVIO_COORD_NORMAL
 * (EqFCoordinateSuite selection, include/eqvio/mathematical/EqFMatrices.h:81-90). */
/* max_landmarks is the INITIAL capacity, not a limit (the reference has none): eqf_add_landmarks and eqf_set_state grow the device
 * buffers (at least doubling, state and Sigma carried over) when more landmarks arrive. The handle stays valid. */
/* Environment: EQF_OPTIONS="id=value,id=value" applies eqf_set_option(id, value) to every context created in the process (A/B runs of an unchanged
 * caller, scripts/ab_bench.sh); a malformed or rejected entry fails the creation with EQF_E_BAD_ARG. The HIP runtime's HIP_FORCE_DEV_KERNARG must stay
 * at its default (1): with kernel arguments in host memory every launch starts ~3 us later (N = 200: 9.5 k -> 8.2 k updates/s, measured). */
int eqf_create(eqf_ctx** out, int device, int max_landmarks, int coordinate_choice);
void eqf_destroy(eqf_ctx* ctx);
int eqf_set_option(eqf_ctx* ctx, int option, int value);
/* the value an option has now (a context that grows its capacity keeps every option and counter: tests/test_gpu_edge_cases.py) */
int eqf_get_option(const eqf_ctx* ctx, int option, int* value);
int eqf_synchronize(eqf_ctx* ctx);
int eqf_num_landmarks(const eqf_ctx* ctx);
/* VIO_eqf::X.id (VIOGroup.h:38): the landmark ids in state order. Host-side only, no device work. Returns N or <0. */
int eqf_get_ids(const eqf_ctx* ctx, int* ids, int cap);
/* the HIP stream the context launches on (hipStream_t as void*), for event timing by the caller */
void* eqf_stream(eqf_ctx* ctx);

/* VIO_eqf::xi0 / VIO_eqf::X (public data members, VIO_eqf.h:37-38). ids/q0/Q are per landmark (3 and 5
 * doubles, AoS on this boundary; SoA on the device). */
int eqf_set_state(eqf_ctx* ctx, const double* xi0_sensor, const double* X_sensor, const int* ids, const double* q0, const double* Q, int N);
int eqf_get_state(eqf_ctx* ctx, double* xi0_sensor, double* X_sensor, int* ids, double* q0, double* Q, int cap); /* returns N or <0 */

/* VIO_eqf::Sigma (VIO_eqf.h:39-40); n must equal 21 + 3N. */
int eqf_set_sigma(eqf_ctx* ctx, const double* sigma_colmajor, int n);
int eqf_set_sigma_diag(eqf_ctx* ctx, const double* diag, int n);
int eqf_get_sigma(eqf_ctx* ctx, double* sigma_colmajor, int n);
int eqf_get_sigma_block(eqf_ctx* ctx, int r0, int c0, int rows, int cols, double* out_colmajor);

/* VIO_eqf::stateEstimate = stateGroupAction(X, xi0) (src/mathematical/VIO_eqf.cpp:137,
 * src/mathematical/VIOGroup.cpp:25-55). Returns N or <0. */
int eqf_state_estimate(eqf_ctx* ctx, double* sensor, int* ids, double* p, int cap);

/* VIO_eqf::addNewLandmarks (VIO_eqf.cpp:225-245): appends k landmarks with Q = identity and
 * newLandmarkCov = var * I (the only form the reference's callers use, src/VIOFilter.cpp:129-130, 274-276). */
int eqf_add_landmarks(eqf_ctx* ctx, const int* ids, const double* p, int k, double var);
/* Round 5: addNewLandmarks for landmarks that belong to the time BEHIND the next eqf_propagate_fast. The reference propagates (src/VIOFilter.cpp:196) and then
 * appends the frame's new landmarks (:217); appending them first and letting the propagation pass them through untouched (F = I, no input and no process noise for their
 * rows, no observer steps; their cross-covariances are exact zeros) gives the same state bit for bit - and every id of the coming measurement is known BEFORE the
 * propagation: the measurement can be staged (eqf_stage_measurement), the propagation kernel evaluates its output blocks and CREATES the held landmarks itself (their
 * rows of Sigma, their planes: no append pass), and the update takes its three-launch route in a frame with landmark turnover as well. The held landmarks are the last
 * ones of the state until that propagation; until then eqf_add_landmarks and the other propagation entry points return EQF_E_UNSUPPORTED; any other entry point turns
 * them into an ordinary pending append (they still pass through the propagation untouched). Returns EQF_E_UNSUPPORTED - nothing added, the caller appends them behind the
 * propagation as the reference does - when the options do not allow it (eqf_hold_supported), the capacity would have to grow, or landmarks with another variance are held.
 * eqf_hold_stats: propagation launches that created held landmarks. */
int eqf_add_landmarks_held(eqf_ctx* ctx, const int* ids, const double* p, int k, double var);
int eqf_hold_supported(const eqf_ctx* ctx);
int eqf_hold_stats(eqf_ctx* ctx, long* launches, int reset);
/* VIO_eqf::removeLandmarkByIndex (VIO_eqf.cpp:172-178) for k indices at once (one compaction pass of Sigma). */
int eqf_remove_landmarks(eqf_ctx* ctx, const int* indices, int k);
/* VIOFilter::removeOldLandmarks (src/VIOFilter.cpp:280-302) in one call: every landmark of the state whose id is not among the M measured ids (strictly ascending, the order of
 * a VisionMeasurement's std::map; EQF_E_BAD_ARG otherwise) leaves the state like eqf_remove_landmarks. removed_idx (room for the current landmark count) receives their
 * indices, ascending, *n_removed their number. O(N + M) whatever the order of the ids in the state. */
int eqf_remove_unmeasured_landmarks(eqf_ctx* ctx, const int* ids, int M, int* removed_idx, int* n_removed);
/* The membership test of VIOFilter::addNewLandmarks (src/VIOFilter.cpp:258-278) in one merge pass: the positions j (ascending) of the measured ids (strictly ascending;
 * EQF_E_BAD_ARG otherwise) that have no landmark in the state go to unknown_j (room for M), their number to *n_unknown. Host only. */
int eqf_find_unknown_ids(eqf_ctx* ctx, const int* ids, int M, int* unknown_j, int* n_unknown);
/* 1 when `ids` are exactly the ids of the measurement the last update mapped, the landmark set has not changed since and there is one id per landmark (no landmark
 * is lost, no id is new: the steady frame of a tracker), else 0. Host only, O(M). */
int eqf_same_as_mapped(const eqf_ctx* ctx, const int* ids, int M);
/* VIO_eqf::removeInvalidLandmarks (VIO_eqf.cpp:213-223). Returns the number removed (>=0) or <0. */
int eqf_remove_invalid_landmarks(eqf_ctx* ctx);
/* EQF_OPT_EARLY_DOORBELL: 1 while the lift results of the last update (estimates, invalid flags) have not been waited for yet; and removeInvalidLandmarks for a caller that
 * deferred it past such an update: the landmarks the UPDATE's lift flagged as invalid leave the state, wherever they sit now. Returns their number (>= 0) or < 0. */
int eqf_update_unsettled(const eqf_ctx* ctx);
/* EQF_OPT_EARLY_DOORBELL: updates the host took from the look-ahead kernel's own doorbell (the others from the lift's) */
int eqf_early_doorbell_stats(eqf_ctx* ctx, long* updates, int reset);
int eqf_remove_invalid_at_update(eqf_ctx* ctx);

/* VIO_eqf::integrateRiccatiStateFast (VIO_eqf.cpp:62-72). Q = diag(Qdiag12) (constructInputGainMatrix,
 * VIOFilterSettings.h:192-201), P = diag: Pdiag8 = the 7 sensor 3-blocks + the per-landmark value
 * (constructStateGainMatrix, :176-190). imu13 = IMUVelocity. */
int eqf_integrate_riccati_fast(eqf_ctx* ctx, const double* imu13, double dt, const double* Qdiag12, const double* Pdiag8);
/* VIO_eqf::integrateRiccatiStateAccurate (VIO_eqf.cpp:74-91), one IMU sample:
 * [Phi, Phi_B] from exp(dt [[A, B],[0, 0]]), Sigma <- Phi Sigma Phi^T + Phi_B (Q/dt) Phi_B^T + dt P.
 * The exponential is evaluated in its block-triangular structure on the device; Phi Sigma Phi^T runs as two dense
 * fp64 MFMA GEMMs. */
int eqf_integrate_riccati_accurate(eqf_ctx* ctx, const double* imu13, double dt, const double* Qdiag12, const double* Pdiag8);
/* VIO_eqf::integrateRiccatiStateDiscrete (VIO_eqf.cpp:93-103): Sigma <- A_d Sigma A_d^T + dt (B Q B^T + P) with A_d = stateMatrixADiscrete
 * (EqFMatrices.cpp:24-41), the central-difference differential (h = cbrt(eps)) of one discrete-lift observer step in error coordinates. A_d has
 * the arrow structure of A: the 43 sensor-level evaluations run on the host, one lane per landmark differentiates its own rows on the device,
 * then two dense fp64 MFMA GEMMs. Normal chart: the Euclidean A_d between the two congruences with the closed-form change of coordinates
 * (A_d,n = M A_d,e M^-1 at the origin). Not with the float Sigma store (EQF_E_UNSUPPORTED). The result carries the differencing's rounding
 * noise (eps / h ~ 4e-11 per unit entry), as the reference's does. */
int eqf_integrate_riccati_discrete(eqf_ctx* ctx, const double* imu13, double dt, const double* Qdiag12, const double* Pdiag8);
/* VIO_eqf::integrateObserverState (VIO_eqf.cpp:47-60) applied for k consecutive samples
 * (the loop of VIOFilter::integrateUpToTime, src/VIOFilter.cpp:160-178). */
int eqf_integrate_observer(eqf_ctx* ctx, const double* imu13_k, const double* dt_k, int k, int discreteLift);

/* Per-landmark statistics VIOFilter::removeOutliers needs (src/VIOFilter.cpp:304-334), for all measured
 * landmarks in one pass: absErr = ||y - yHat||, probErr = yTilde^T (C0 Sigma_ii C0^T)^-1 yTilde with
 * C0 = outputMatrixCi (VIO_eqf::getOutputCovById, VIO_eqf.cpp:196-211); depth2 = |q_hat|^2 for
 * getMedianSceneDepth (src/VIOFilter.cpp:366-380). Outputs are indexed by STATE landmark index (length N);
 * unmeasured landmarks get -1 in absErr/probErr. */
int eqf_outlier_stats(eqf_ctx* ctx, const eqvio_camera* cam, const int* ids, const double* y, int M, double* absErr, double* probErr, double* depth2);
/* VIO_eqf::getOutputCovById (VIO_eqf.cpp:196-211) for ALL landmarks of the state in one kernel and one host wait: out4N[4 i ..] = C0_i Sigma_ii C0_i^T
 * (row-major 2 x 2, STATE landmark index i), C0_i = outputMatrixCi at the current estimate - the measured pixel does not enter (the reference's argument y
 * is [[maybe_unused]]). For a binding that keeps the reference's VIOFilter.cpp unchanged: its removeOutliers asks for one id at a time
 * (src/VIOFilter.cpp:329); the binding fetches all of them on the first such call after the state changed. */
int eqf_output_cov_all(eqf_ctx* ctx, const eqvio_camera* cam, double* out4N);

/* VIO_eqf::performVisionUpdate (VIO_eqf.cpp:105-135): yTilde, C, S = C Sigma C^T + R, K = Sigma C^T S^-1,
 * Gamma = K yTilde, X <- Delta * X, Sigma <- Sigma - K C Sigma; R = meas_var * I
 * (constructOutputGainMatrix, VIOFilterSettings.h:203-206). Every measured id must be a state landmark. */
int eqf_vision_update(eqf_ctx* ctx, const eqvio_camera* cam, const int* ids, const double* y, int M, double meas_var, int useEquivariantOutput, int discreteCorrection);

/* VIOFilter::integrateUpToTime, fast-Riccati branch (VIOFilter.cpp:134-192), in one call: integrateRiccatiStateFast with the
 * mean IMU sample over dt_total (at the current X), then the k observer steps. Same result as eqf_integrate_riccati_fast
 * followed by eqf_integrate_observer (bit-identical); the difference is on the device: the rows of A and B are assembled by the
 * workgroups of the Sigma propagation kernel that need them, and the landmark part of the observer steps rides along as extra
 * workgroups of the same launch (writing a second landmark buffer), so the whole propagation is ONE kernel on one stream, with
 * no second stream and no events (EQF_OPT_FUSED_ASSEMBLY = 0: a separate assembly kernel in front of it). */
int eqf_propagate_fast(eqf_ctx* ctx, const double* imu13_mean, double dt_total, const double* Qdiag12, const double* Pdiag8, const double* imu13_k, const double* dt_k,
                       int k, int discreteLift);
/* Optional hint, to be called BEFORE the propagation call of the same frame (eqf_propagate_fast / eqf_integrate_riccati_fast)
 * with the measurement that eqf_stats_then_update will receive (VIOFilter::processVisionData has it in hand before it propagates,
 * VIOFilter.cpp:194-196). If every id is in the state, the measurement is written to the pinned packet and one extra block of the
 * propagation kernel copies it to HBM, so that the update's first kernel does not fetch it across PCIe. eqf_stats_then_update
 * uses the staged copy only if ids, y and the landmark set are unchanged; otherwise (or without this call) it behaves as before.
 * Never an error for unknown ids (nothing is staged). */
int eqf_stage_measurement(eqf_ctx* ctx, const int* ids, const double* y_px, int M);

/* Speculative frame tail for VIOFilter::processVisionData (VIOFilter.cpp:209-236) when every measurement id is already a
 * landmark of the state: the outlier statistics of removeOutliers (VIOFilter.cpp:304-334) and performVisionUpdate are
 * queued back to back with ONE host wait. The statistics kernel compares each measured landmark with the two thresholds
 * on the device; if any exceeds one (the host then has outliers to rank and remove) it cancels the queued update kernels,
 * which return at their first instruction: nothing is modified, *updated = 0 and the caller continues exactly as without
 * speculation (decide with the returned statistics, remove, eqf_vision_update). Otherwise *updated = 1 and the state is
 * the one eqf_vision_update would have produced (bit-identical). *updated = -1: not applicable, some measurement id is not in
 * the state (nothing was computed). EQF_OPT_SPECULATIVE = 0 turns it into a plain statistics call (*updated = 0). absErr / probErr: -1 for landmarks without a measurement.
 * *updated is 1 only when the call returns 0: a failed update (EQF_E_NOT_SPD, EQF_E_NONFINITE) leaves it 0. */
int eqf_stats_then_update(eqf_ctx* ctx, const eqvio_camera* cam, const int* ids, const double* y_px, int M, double thrAbs, double thrProb, double meas_var,
                          int useEquivariantOutput, int discreteCorrection, double* absErr, double* probErr, double* depth2, int* updated);

/* How the frames went through eqf_stats_then_update since the last reset: calls; calls that queued the update tail speculatively behind the
 * statistics kernel (every measured id already a landmark); of those, tails the device cancelled because a landmark exceeded an outlier
 * threshold (the caller then takes the two-round-trip path: removeOutliers / addNewLandmarks / eqf_vision_update). */
int eqf_speculation_stats(eqf_ctx* ctx, long* calls, long* queued, long* cancelled, int reset);
/* EQF_OPT_MEASURE_IN_PROPAGATE: update calls that used the output blocks evaluated by the propagation kernel in front (no k_build_Z launch where the look-ahead kernel runs). */
int eqf_measure_in_propagate_stats(eqf_ctx* ctx, long* used, int reset);
/* EQF_OPT_Z_IN_LOOKAHEAD: look-ahead launches whose half-rows built Z themselves (no k_build_Z launch in front; 3 .. 32 panels since round 6). */
int eqf_z_in_lookahead_stats(eqf_ctx* ctx, long* launches, int reset);
/* EQF_OPT_GATHER_IN_PROPAGATE: propagation launches that applied a record of removed landmarks themselves */
int eqf_gather_stats(eqf_ctx* ctx, long* launches, int reset);
/* eqf_stats_then_update with VIOFilter::removeOutliers' decision (VIOFilter.cpp:304-364) made on the device where that saves the frame a host round trip:
 * while the speculative tail keeps getting cancelled (outlier candidates frame after frame, as with the shipped thresholds) the call queues the statistics,
 * the decision (candidates ranked absolute outliers first by absErr, then probabilistic ones by probErr, the first max_outliers = (size_t)((1 - featureRetention)
 * * #features) of them discarded) and the update at once, with the discarded landmarks' measurements masked out; it then removes the discarded landmarks
 * from the state (removeLandmarkById) and returns their former state indices in removed_idx (capacity: the landmark count before the call). *updated as for
 * eqf_stats_then_update; when it is 0 the caller runs removeOutliers itself as before. Same results as the reference's erase-then-update order: a landmark
 * without a measurement can be marginalised before or after the update. */
int eqf_stats_select_update(eqf_ctx* ctx, const eqvio_camera* cam, const int* ids, const double* y_px, int M, double thrAbs, double thrProb, int max_outliers, double meas_var,
                            int useEquivariantOutput, int discreteCorrection, double* absErr, double* probErr, double* depth2, int* updated, int* removed_idx, int* n_removed);
/* look-ahead factorisation since the last reset: launches of the persistent kernel; of those, launches whose bounded wait ran out and whose
 * factorisation was redone on the launch chain (same Z: W and Sigma+ bit-identical, Gamma up to rounding; three in a row switch EQF_OPT_LOOKAHEAD off for the context). */
int eqf_lookahead_stats(eqf_ctx* ctx, long* launches, long* fallbacks, int reset);
/* 1 when the context's stream owns its hardware queue. The runtime shares GPU_MAX_HW_QUEUES = 4 hardware queues among a process' plain streams, and two filters whose streams
 * share one take turns kernel by kernel (four filters in one process at N = 200: 20 k updates/s aggregate, three: 28.5 k). With EQF_OWN_HW_QUEUES=<n> in the environment at
 * eqf_create, the first n contexts of the process on a device get a stream created with a compute-unit mask (all compute units), which has a queue of its own (four filters:
 * 32.5 k). 0: plain stream (the default; see create_buffers in eqf_hip.hip for why). */
int eqf_own_hardware_queue(eqf_ctx* ctx);
/* EQF_OPT_LIVE_COLUMNS_FIRST: look-ahead launches that took their panel count from the device (the outlier decision in front of them put the live columns first) */
int eqf_live_columns_stats(eqf_ctx* ctx, long* launches, int reset);
/* Result of the look-ahead kernel's self-test at eqf_create (the persistent kernel against the launch chain on fixed problems of 3, up to 13 and 17 panels, eight launches each, W compared bit
 * for bit): 1 passed, 0 not run (the kernel is never eligible at this capacity / on this device, or its launch stalled four times because the device was busy -
 * e.g. eight processes creating contexts on one device at once: a stall says nothing about the kernel, and every later launch is bounded and redone on the chain
 * if it stalls), -1 failed (W differed): the context factorises on the launch chain. */
int eqf_lookahead_selftest(const eqf_ctx* ctx);
/* EQF_OPT_LA_HOME: the XCD this context's look-ahead launches keep their owner and S half-rows on (-1: none was free at creation, or the placement was refused /
 * switched off), and the look-ahead launches that used it. */
int eqf_lookahead_home(const eqf_ctx* ctx, int* home_xcd, long* home_launches);
/* 1 when no other context - of this or of any other process of the user - is registered on the device (the condition of the HOME placement from 6 panels on), 0 otherwise.
 * A test that has the device to itself asserts home_launches == look-ahead launches with it: a silent return to the classic placement (or to the launch chain) turns it red. */
int eqf_device_to_itself(eqf_ctx* ctx);
/* frames that took the device-side decision, landmarks it discarded */
int eqf_selection_stats(eqf_ctx* ctx, long* frames, long* discarded, int reset);

/* Gamma of the last update (n doubles) — for parity checks. */
int eqf_last_gamma(eqf_ctx* ctx, double* out, int cap);

/* VIO_eqf::computeNEES (VIO_eqf.cpp:153-170): eps = stateChart(stateGroupAction(X^-1, trueState restricted to X.id), xi0),
 * NEES = eps^T Sigma^-1 eps / dim. The true state must contain every landmark id of the filter state. Sigma^-1 eps is
 * never formed: Sigma = L L^T is factorised with the same blocked chain as the vision update and NEES = |L^-1 eps|^2 / n.
 * When that factorisation meets a non-positive pivot (Sigma positive definite only up to rounding) the call does what the
 * reference's LU-based Sigma.inverse() does (VIO_eqf.cpp:166-168) and still returns a number: Gaussian elimination with partial
 * pivoting on [Sigma | eps], on the device. eqf_nees_lu_fallbacks counts those calls. */
int eqf_compute_nees(eqf_ctx* ctx, const double* true_sensor, const int* true_ids, const double* true_p, int n_true, double* nees);
int eqf_nees_lu_fallbacks(eqf_ctx* ctx, long* count);

/* EqF matrices as the device assembled them, expanded to the reference's dense layout for parity tests:
 * A (n x n), B (n x 12) from stateMatrixA / inputMatrixB (coordinateSuite/euclid.cpp:99-233,
 * invdepth.cpp:36-181); C (2M x n) from outputMatrixC (EqFMatrices.cpp:43-82). Column-major. */
int eqf_debug_matrices_AB(eqf_ctx* ctx, const double* imu13, double* A_out, double* B_out);
/* the work matrix of the last vision update, column-major rows x cols: rows [2M, 2M + n) hold W = Sigma C^T L^-T, row 2M + n holds z^T = yTilde^T L^-T
 * (Sigma+ = Sigma - W W^T, Gamma = W z); rows < 2M are scratch. For the bit-identity tests of the factorisation variants. */
int eqf_debug_get_W(eqf_ctx* ctx, double* out, int rows, int cols);
/* The block -> tile table of the covariance update kernel for nt 32 x 32 tiles per side (host-side, no device needed): tile_of_block[b] = bi | bj << 16, nt (nt + 1) / 2
 * entries. Blocks are dealt round robin to the 8 XCDs; the table gives every XCD a compact part of the tile triangle. For the test of that property. */
int eqf_debug_syrk_order(int nt, int* tile_of_block);
int eqf_debug_matrix_C(eqf_ctx* ctx, const eqvio_camera* cam, const int* ids, const double* y, int M, int useEquivariantOutput, double* C_out, double* ytilde_out);

/* fp64 MFMA micro-benchmark (v_mfma_f64_16x16x4_f64 issue rate): returns achieved TFLOP/s over the whole
 * chip; used by bench.py to state the roofline peak it prices against. */
int eqf_mfma_f64_peak(eqf_ctx* ctx, double* tflops);
/* the same, and the shader clock the chip held while that kernel ran (device cycle counter over the 100 MHz wall clock, averaged over 32 waves spread
 * over the grid): the datasheet's 78.6 TFLOP/s are 1024 SIMDs x 32 flop/clock x 2.4 GHz, so tflops / (32768 x sclk) is the issue efficiency at the clock held */
int eqf_mfma_f64_peak_clock(eqf_ctx* ctx, double* tflops, double* sclk_ghz);

/* per-kernel timing of the last frame's launches on the context's stream (HIP events); fills up to cap
 * (name index, microseconds) pairs; see eqf_kernel_name. Enabled with eqf_set_option(ctx, 100, 1). */
int eqf_last_kernel_times(eqf_ctx* ctx, int* which, float* usec, int cap);
/* EQF_OPT_TRACE = 1: device-side timeline of the last 1024 frames, no profiler and no events involved.
 * device_ticks[1024][48] (100 MHz device wall clock; 0 = not stamped): slot 0 k_assemble_AB start, 1 k_propagate_main start, 2 k_build_Z start,
 * 3 + s start of factorisation step s (s < 32), 40 / 41 k_lift start / last workgroup done (= doorbell), 42 / 43 k_syrk_sub start /
 * last workgroup done. host_ns[1024][8] (steady_clock): 0 doorbell seen, 1 eqf_propagate_fast entered, 2 assembly launch returned,
 * 3 propagation launch returned, 4 eqf_stats_then_update entered, 5 k_build_Z launch returned, 6 last launch of the tail returned.
 * Row = frame number mod 1024; *last_frame = number of the newest frame. The doorbell (slot 41 / host 0) ties the two clocks together. */
int eqf_trace_read(eqf_ctx* ctx, unsigned long long* device_ticks, long long* host_ns, unsigned* last_frame);
/* Host-side view of the frame since the last reset: calls[0] / seconds[0] = doorbell waits and the wall time the host spent spinning
 * in them; calls[1] / seconds[1] = kernel launch calls and the wall time spent inside them. (frame period - wait time per frame) is
 * the host's own share; a wait time near zero means the frame is host-bound. Both arrays have two entries. */
/* EQF_OPT_TRACE = 1 and the look-ahead factorisation: device wall-clock stamps (100 MHz) of the LAST frame from inside the persistent
 * kernel, 96 rows of 8: rows [0, 32) the owner's step k (0 pre-work starts, 1 U tiles arrived, 2 b arrived, 3 pre-work done, 4 elimination of
 * D_k done, 5 past the first barrier, 6 D_(k+1) handed to the elimination); rows [32, 64) the first T block row and rows [64, 96) S block row
 * NJ-2, panel p (0 L_p^-1 and the panel tile in LDS, 1 P_I done and published, 2 tile updates done). */
int eqf_debug_lookahead_stamps(eqf_ctx* ctx, unsigned long long* out768);
int eqf_host_wait_stats(eqf_ctx* ctx, long* calls, double* seconds, int reset);
const char* eqf_kernel_name(int which);

#ifdef __cplusplus
}
#endif
#endif
