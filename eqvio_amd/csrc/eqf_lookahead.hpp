// Look-ahead factorisation of Z = [S ; T ; yTilde^T] in ONE persistent kernel (reference arithmetic: S^-1 and K = Sigma C^T S^-1,
// VIO_eqf.cpp:116-119; same blocked right-looking LDL^T / trsm arithmetic, tile by tile and panel by panel, as the one-launch-per-panel
// chain k_chol_step in eqf_kernels.hpp, therefore bit-identical W and L^-1).
//
// The chain of k_chol_step launches walks 13 dependent steps at N = 200, each = operand loads 0.9 + P 0.56 + tile update 0.64 + 32x32
// elimination 3.24 + launch boundary 1.48 us. Only the elimination and two 32^3 products are inherently sequential. Here:
//
//  * workgroup 0, the OWNER, walks the pivot chain and nothing else. For block row I = k + 1 it receives the two tiles next to the
//    diagonal, U1 = Z(I, I-1) and U0 = Z(I, I), with every panel up to I-3 already applied, plus b = P^(I-2)_I; WHILE wave 0 still
//    eliminates D_k, four other waves apply panel I-2 (R1 = U1 - b c_k^T, D' = U0 - b b^T; c_k = P^(k-1)_k is still in LDS from the
//    previous step). When L_k^-1 appears they finish: c = R1 L_k^-T, D = D' - c c^T, and wave 0 eliminates D. Per step on the critical
//    path: elimination + two products + three workgroup barriers; no launch boundary, no memory round trip, no cross-workgroup hop.
//  * one ROW workgroup per 32-row block row I >= 1 keeps ALL tiles Z(I, 0 .. min(I, NJ-1)) in its MFMA accumulator registers for the
//    whole factorisation (8 waves: two groups of four, even / odd tile columns). Per panel p: receive L_p^-1, compute P^(p)_I =
//    Z(I,p) L_p^-T (final W rows for the T block rows; factor rows, published, for the S block rows), receive P^(p)_J of the S block
//    rows J > p and apply Z(I,J) -= P_I P_J^T. S block rows hand U1 / U0 to the owner after panel I-3 and publish b after L_(I-2)^-1.
//    They run one to two panels behind the owner; nothing they do is on the critical path as long as they keep the owner's pace.
//  * HAND-OFF: a published tile is 8 KB of doubles written with write-through stores (global_store sc1: the data is at the agent
//    coherence point when the store completes), then s_waitcnt vmcnt(0), a workgroup barrier and ONE flag word = the launch's sequence
//    number. A consumer polls the flag words it needs with one cache-bypassing load per wave (all flags of a panel are contiguous) and
//    then reads the tiles with ordinary cached loads, every load of the step in flight at once. No release / acquire fence: nothing is
//    dirty in an L2 (write-through), and no line of a tile is ever read before its flag is up, so no L2 or L1 can hold a stale copy of
//    it (tiles are 8 KB aligned; caches are invalidated at the kernel boundary; the sequence number changes with every launch, the
//    buffers are never cleared). The first version published every double as a 16-byte (value, sequence) word read with cache-bypassing
//    loads: 1.0 us per hop in isolation (scripts/ubench/pingpong2.hip) but 30 workgroups fetching the same tiles past the L2 saturate the
//    few memory channels a tile lives in (2.5 us per round trip under load): the cached version lets every XCD fetch a tile once.
//    The yTilde row (32 doubles per panel) still travels as 16-byte words.
//  * Gamma = W z is accumulated by the T block rows on the way (z_p = yTilde_p L_p^-T from the published yTilde row), so the lift
//    kernel finds Gamma complete.
//  * Every poll is bounded (20 ms of device wall clock); a timeout raises flags[3] (EQF_E_STALLED) and the workgroups drain.
//    Dependencies point from higher to lower block rows and to the owner only; the host launches this kernel only when its NI workgroups (66 at
//    N = 200, 161 at N = 512) fit the device at one workgroup per compute unit (eqf_hip.hip: lookahead_eligible).
#pragma once
#include <type_traits>
#include "eqf_kernels.hpp"

namespace eqf {

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int LA_T = 512;                       // threads per workgroup (8 waves)
constexpr int LA_TILE = 1024;                   // doubles of a published 32 x 32 tile, [r + 32 c]
constexpr long long LA_TIMEOUT_TICKS = 2000000; // 20 ms at 100 MHz

struct LaArgs {
    int rows, m, ldz, NJ, NI, seq;
    int watch_ahead;         // round 6 (la_row2): a lane's flag word of the coming panel is requested a panel ahead (debug knob 102: 0 = every panel polls, as before)
    int split_from;          // S block rows I >= split_from are held by TWO workgroups per half-row (la_row2: Jlo, partA); >= NJ: none
    int home;                // HOME instantiations (round 5): the XCD block 0 of this stream's grids runs on (block b runs on XCD (home + b) & 7); -1 otherwise
    int* pubfl;              // HOME: the flags of the hand-offs that stay inside the home XCD's L2 (same indices as pubf)
    long long timeout_ticks; // bound of every device-side wait (100 MHz ticks; default LA_TIMEOUT_TICKS)
    const double* Z;     // [S ; T ; y^T] from k_build_Z (plain memory; complete when this kernel starts)
    double* W;           // out, plain: rows >= m receive W = T L^-T and the z row
    const double* Linv0; // L_0^-1, 32 x 32 column-major, from k_build_Z's first-tile elimination
    double* pub;         // published tiles (offsets below)
    int* pubf;           // their flags
    char* puby;          // the yTilde row of every panel as 16-byte (value, sequence) words
    double* gamma;       // out: Gamma[n]
    int* flags;          // [0] non-positive pivot, [3] = seq: a bounded wait of this launch ran out
    const int* spec;
    int spec_seq;
    trace_t* tr_steps;       // EQF_OPT_TRACE: slot of step 0 (the owner stamps one slot per step), or nullptr
    unsigned long long* dbg; // EQF_OPT_TRACE: per-step stamps inside the owner ([k][8]) and two block rows ([32 + p][8], [64 + p][8]), or nullptr
    // EQF_OPT_Z_IN_LOOKAHEAD (ZB instantiations): no k_build_Z launch in front of this kernel - every half-row builds its own 16 rows of Z = [S ; T ; yTilde^T]
    // from Sigma and the output blocks C_j (the expressions of k_build_Z: bz_T_pair / bz_S_block), the owner builds and eliminates the first tile
    const double* zb_sig; // Sigma (fp64), leading dimension zb_ld
    int zb_ld, zb_M, zb_Mcap;
    double zb_var;        // measurement variance (diagonal of R)
    const double* zb_C;   // C blocks, plane e at [e zb_Mcap + j]
    const double* zb_ytil;
    const int* zb_lmidx;  // measurement -> state landmark index
    double* zb_linv0;     // (unused by the kernel since round 4: L_0^-1 goes straight to the owner's LDS and its published tile)
    int zb_ident;         // measurement j belongs to landmark j for every j (the regular frame: the landmarks without a measurement were removed before the update and both
                          // are in ascending id order) - nobody loads the index map, which would be a memory round trip IN FRONT of every Sigma load of the prologue
    // ZB = 2 (the speculative frame tail, eqf_stats_then_update): no measurement kernel either - every workgroup evaluates the output blocks C_j it needs
    // (measure_j, one lane per measurement, into LDS), workgroup NI computes the outlier statistics, decides about the tail (speculation word) and
    // leaves C / yTilde / the index map in memory for a retry; zb_C / zb_ytil / zb_lmidx are not read
    // ZB = 3 (round 4): like 2, but the output blocks were evaluated by the observer blocks of the propagation kernel in front (EQF_OPT_MEASURE_IN_PROPAGATE): the
    // half-rows and the owner read zb_C / zb_ytil / zb_lmidx like ZB = 1, workgroup NI computes the statistics like ZB = 2
    MeasFuse zb_mf;
    // Up to 16 panels, behind k_stats_select (the outlier decision taken on the device): the columns of the measurements that stay are in front, and this word - written by
    // that kernel - says how many there are. The factorisation ends with the last panel that holds one of them (la_live_panels); W's columns behind it are zero.
    const int* live_cols;
    trace_t* tr_zb; // EQF_OPT_TRACE: k_build_Z's slot - this kernel's start stands for it (the span to step 0 is the prologue that replaces k_build_Z)
    // EQF_OPT_EARLY_DOORBELL (round 5): the T half-row that finishes last - every W row final, no wait ran out, no pivot failed, the tail not cancelled - hands the host Gamma's
    // sensor rows and rings a doorbell of its own: the host hears that the update WILL be applied (lift and covariance update behind this kernel only look at the same words) a
    // kernel boundary and a lift earlier than from k_syrk_lift's doorbell, and starts the next frame's propagation on it
    int* early_cnt;           // device counter of finished T half-rows (low 16 bits) and of those that gave up (high bits); left at 0
    int* early_door;          // pinned
    double* early_gamma_host; // pinned: Gamma[0 .. 20]
    int early_seq;
    const int* early_spec;    // the tail's cancellation word (or nullptr)
    int early_spec_seq;
};
// tiles: [0, NJ) L_p^-1 | [NJ, NJ + NJ^2) P^(p)_J at p NJ + J (a panel's tiles are neighbours) | then U1, U0 of every S block row
// flags: the same indices (one int per tile; U1 / U0 share the flag of U1)
__device__ __forceinline__ int la_i_linv(const LaArgs& a, int p) { return p; }
__device__ __forceinline__ int la_i_p(const LaArgs& a, int J, int p) { return a.NJ + p * a.NJ + J; }
__device__ __forceinline__ int la_i_u(const LaArgs& a, int I, int which) { return a.NJ + a.NJ * a.NJ + 3 * I + which; } // which: 0 U1 = Z(I, I-1), 1 U0 = Z(I, I), 2 U2 = Z(I, I-2)
inline size_t la_pub_tiles(int NJ) { return (size_t)NJ + (size_t)NJ * NJ + 3 * (size_t)NJ; }
// flags: one per L_p^-1 | one per (panel p, half-row h) for the factor rows P^(p)_h (a panel's flags are neighbours) | one per (S block row I, half s) for
// its U1 / U0 hand-off. Half-row h = 2 I + s holds rows 16 h .. 16 h + 15 of S.
__device__ __forceinline__ int la_f_linv(const LaArgs& a, int p) { return p; }
__device__ __forceinline__ int la_f_p(const LaArgs& a, int p, int h) { return a.NJ + 2 * a.NJ * p + h; }
__device__ __forceinline__ int la_f_u(const LaArgs& a, int I, int s) { return a.NJ + 2 * a.NJ * a.NJ + 2 * I + s; }
inline size_t la_pub_flags(int NJ) { return (size_t)NJ + 2 * (size_t)NJ * NJ + 2 * (size_t)NJ; }
__device__ __forceinline__ double* la_tile(const LaArgs& a, int idx) { return a.pub + (size_t)LA_TILE * idx; }
// panels this launch factorises: all of them, or - LaArgs::live_cols - those up to the last live column. One panel is enough: the owner's loops (k + 1 < NJ, k + 2 < NJ) and the
// half-rows' (np, Jmax) are written for any count - a launch that ends after one or two panels is what three-panel problems with most measurements discarded give
// (tests/test_gpu_parity.py::test_outlier_decision_in_one_workgroup_equals_the_two_launches[40-33-*]; the host still takes problems below three panels to the launch chain)
constexpr int LA_MIN_LIVE_PANELS = 1;
__device__ __forceinline__ int la_live_panels(const LaArgs& a) {
    if (!a.live_cols)
        return a.NJ;
    const int me = __builtin_amdgcn_readfirstlane(*a.live_cols);
    return min(a.NJ, max((me + 31) >> 5, LA_MIN_LIVE_PANELS));
}

__device__ __forceinline__ void la_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } // global_store_dwordx2 sc1
__device__ __forceinline__ void la_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void la_raise_f(const LaArgs& a, int fidx) { __hip_atomic_store(a.pubf + fidx, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Round 5, the HOME placement: the owner and every S half-row run on ONE XCD - the blocks b with b & 7 == 0. The hardware deals the blocks of a grid round robin to the
// eight XCDs, block b to XCD (o + b) & 7 with an offset o that belongs to the hardware queue (measured: scripts/ubench/xcd_map.hip, profiles/r05_xcd_map.txt - also with
// two grids in flight at once); the host passes the o it has learnt (LaArgs::home) and EVERY block compares HW_REG_XCC_ID with the XCD that gives it before it does
// anything: a launch whose blocks sit elsewhere ends at once, reports the XCD of block 0 (flags[4], flags[5]) and is redone on the launch chain. What the home
// workgroups hand to EACH OTHER then never has to leave that XCD's L2:
// plain stores (acknowledged by the L2: 0.26 us instead of 0.55 for a write-through store), a flag in a second array (pubfl) written the same way and polled with sc1
// loads (the L2 answers), plain tile loads that hit the L2 (0.24 instead of 0.42 us) - 1.10 us per hop instead of 2.00 between two XCDs (scripts/ubench/xcd_hop2.hip,
// profiles/r05_xcd_hop.txt). What the T half-rows (other XCDs) read as well - L_p^-1, the factor rows P^(p)_J - is stored a SECOND time, written through, behind the
// local flag, with the flag in pubf as before. A plain store is invisible to another XCD (the ubench reads stale tiles there), so no consumer outside the home XCD
// ever looks at pubfl.
__device__ __forceinline__ void la_st_l(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); } // global_store_dwordx2, no cache bits
__device__ __forceinline__ void la_raise_fl(const LaArgs& a, int fidx) { __hip_atomic_store(a.pubfl + fidx, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ int la_xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 0xf; } // HW_REG_XCC_ID, bits 3:0
struct LaPoll {
    long long deadline;
    int seq;
    int* s_abort; // [0]: the word every wave acts on; [1]: raised by a poll that ran out of time
};
// A poll that runs out of time raises the REQUEST word. In a half-row workgroup thread 0 copies it to the abort word in front of a workgroup barrier and
// every wave reads the abort word behind that barrier: all waves of the workgroup leave at the same barrier (a wave reading a word that another wave
// writes at any time could leave one barrier earlier than its neighbours). The owner has no barriers: its waves act on the request word directly.
__device__ __forceinline__ bool la_retry(const LaPoll& pl) {
    if ((long long)wall_clock64() > pl.deadline) {
        pl.s_abort[1] = 1;
        return false;
    }
    __builtin_amdgcn_s_sleep(1);
    return true;
}
// wait until the `count` (<= 64) consecutive flags at f carry the launch's sequence number: lane j watches flag j
__device__ __forceinline__ void la_wait(const int* f, int count, const LaPoll& pl) {
    const int lane = threadIdx.x & 63;
    if (lane < count) {
        for (;;) {
            const int v = __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // global_load_dword sc1: not served by a cache
            if (v == pl.seq || !la_retry(pl))
                break;
        }
    }
    asm volatile("" ::: "memory"); // the tile loads below must not be hoisted over the wait
}
// the yTilde row: 16-byte (value, sequence, ~sequence) words, one store / one load each (32 doubles per panel: no traffic to speak of)
__device__ __forceinline__ void la_put16(char* p, double v, int seq) {
    v4i x;
    x.x = __double2loint(v);
    x.y = __double2hiint(v);
    x.z = seq;
    x.w = ~seq;
    // s_nop: a VALU write of the data registers must not follow a store of more than 64 bits within one wait state, and the compiler
    // cannot see that this asm is such a store (without it: wrong values with a valid sequence number in ~8 % of the launches)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ double la_get16(const char* p0, const LaPoll& pl) {
    v4i r;
    for (;;) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p0) : "memory");
        if ((r.z == pl.seq && r.w == ~pl.seq) || !la_retry(pl))
            break;
    }
    return __hiloint2double(r.y, r.x);
}
// MFMA operand layout: lane (lr, lk) receives v[st] = X[16 h + lr][4 st + lk], st = 0..7, of a published tile X[row + 32 k]
__device__ __forceinline__ void la_operand(const double* __restrict__ tile, int h, double (&v)[8]) {
    const int lane = threadIdx.x & 63;
    const double* p0 = tile + (16 * h + (lane & 15)) + 32 * (lane >> 4);
#pragma unroll
    for (int st = 0; st < 8; ++st)
        v[st] = p0[128 * st];
}

// ---- the owner --------------------------------------------------------------------------------------------------------------------
// Round 3: a dataflow of eight specialised waves, synchronised by monotone LDS counters instead of workgroup barriers, so that no wave ever waits
// for something it does not need - in particular not for the ~1 us acknowledgement of somebody's write-through stores. Roles as of round 4:
//   wave 0      pivot: forms the block (0, 0) of D_(k+1) itself (c rows 0 .. 15 = R1[0:16, :] L_k^-T and D'_00 - c c^T on register-operand MFMAs, with the blocks of
//               L_k^-1 it holds from its last elimination), eliminates it, takes the blocks (1, 0) / (1, 1) from sD when it needs them, finishes the 32 x 32 tile
//               (register-resident LDL^T, eqf_kernels.hpp: ldl_inverse_tile_regs), writes L_(k+1)^-1 to sLk and announces it (LC_L). Only the LAST tile is
//               published from here.
//   waves 5, 7  post-work of step k: rows 16 .. 31 of c = P^(k)_(k+1) -> sX (LC_C), the blocks (1, 0) / (1, 1) of D_(k+1) = D' - c c^T -> sD (LC_D); then tail.
//   waves 2,5,6,7  tail of step k (under the elimination of D_(k+1)): for block row I2 = k + 2 fetch U2 = Z(I2, k), U1 = Z(I2, k+1), U0 = Z(I2, I2) (panels
//               <= k - 1 applied by the block row itself, except two products, below) and form b = P^(k)_I2 = U2 L_k^-T (waves 2 and 7; round 2: b came from
//               block row I2, two dependent hand-offs behind L_k^-1), R1 = U1 - P^(k-1)_I2 b_prev^T - b c^T, D' = U0 - b b^T (LC_T).
//               Quadrant 0 runs on wave 2, not on wave 4: waves sit on SIMD (wave % 4), and fp64 MFMAs of a SIMD-mate slow the pivot wave's fp64 VALU chain
//               (measured: elimination 3.3 -> 4.0 us with 48 tail MFMAs on wave 4); SIMD 0 is left to the pivot wave. Wave 4 has no work since round 4 (a poller
//               there, even asleep between its looks, cost the pivot wave 1 us per factorisation).
//   wave 1      publishes c (sX -> write-through tile, waits for the acknowledgement, raises the two half-row flags of P^(k)_(k+1)), then b the same way
//               (from the LDS copy the tail keeps for the next tail).
//   wave 3      polls the U flags of the block rows, one after the other, so that a tail finds them checked (a flag poll is ~0.9 us of memory latency
//               even when the flag has been up for long), and in between publishes L_k^-1 out of sLk (the pivot wave keeps its issue slots).
// The product Z(I, I-1) -= P^(I-3)_I (P^(I-3)_(I-1))^T is left to the owner (r3 below): its second factor is the owner's own b of the step before, and
// a block row waiting for it closed a cycle b -> block row -> U -> next b of 5.3 us per step (measured). So is (round 4) the block (1, 0) of the diagonal tile's
// product of the same panel, P^(I-3)_bottom (P^(I-3)_top)^T: the bottom half-row would wait for the top one's factor rows.
// Same products in the same order as round 2 and as the launch chain: W and Sigma+ are bit-identical.
// relaxed = true: a wait that is not on the critical path sleeps between its looks, so that five spinning waves do not compete with the pivot wave's
// own LDS traffic (ds_bpermute, operand reads)
template <bool RELAXED = false> __device__ __forceinline__ bool la_lds_wait(const int* c, int target, const int* s_abort) {
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        if (*(volatile const int*)s_abort)
            return false;
        if (RELAXED)
            __builtin_amdgcn_s_sleep(4);
    }
    asm volatile("" ::: "memory");
    return true;
}
__device__ __forceinline__ void la_lds_set(int* c, int v) { // after this wave's LDS stores (LDS operations of a wave execute in order)
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0)
        __hip_atomic_store(c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void la_lds_add(int* c, int by = 1) {
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0)
        __hip_atomic_fetch_add(c, by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
enum { LC_L = 0, LC_T, LC_C, LC_D, LC_B, LC_CCOPIED, LC_BCOPIED, LC_U, LC_LCOPIED, LC_COUNT };

template <int ZB, bool HOME>
__device__ __forceinline__ void la_owner(const LaArgs& a, double* smem, int* s_abort_words, int* cnt, const LaPoll& pl) {
    int* const s_abort = s_abort_words + 1; // the owner has no barriers: its waves act on the request word
    // `wave` as a scalar: the role branches become real (scalar) branches. With a vector condition the compiler predicates short blocks instead of
    // branching around them, and a predicated-off s_sleep still sleeps (measured: every wave took wave 4's nap).
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    double* sLk = smem;                  // L_k^-1, operand layout [r + c CH_LDP]
    double* sX = smem + 32 * CH_LDP;     // c = P^(k)_(k+1): written by the post-work of step k, read by the tail of step k and by wave 1
    double* sY = smem + 2 * 32 * CH_LDP; // R1 of block row k + 1 (written by the tail of step k - 1, read by the post-work of step k)
    double* sD = smem + 3 * 32 * CH_LDP; // the diagonal tile handed to the elimination
    double* swork = smem + 4 * 32 * CH_LDP;
    double* sDq = swork + 288;           // 16 x 16: the block (0, 0) of D' on its way from the tail wave 2 to the post-work wave 4 (a slot of swork the elimination does not use)
    // b of a tail, rows 0 .. 15 / 16 .. 31 in operand layout, parked in the unused rows 32 .. 47 of two of the four operand buffers (leading dimension 48);
    // the pairs alternate with the parity of the step: tail k reads what tail k - 1 kept
    auto b_keep = [&](int k, int half) -> double* { return smem + 32 * CH_LDP * ((k & 1) ? 3 * half : 1 + half) + 32; };
    const int NJ = la_live_panels(a); // (the tiles' and flags' indices are laid out for a.NJ)
    if (ZB) {
        if (a.tr_zb && tid == 0)
            *a.tr_zb = wall_clock64();
        // the first diagonal tile D_0 = (C Sigma C^T + R)[0:32, 0:32] (k_build_Z's first-tile row: one thread per pair of measurements), eliminated here
        const int i = tid & 15, jj = (tid >> 4) & 15;
        const int M = a.zb_M;
        double* sC16 = sX; // ZB = 2: the C blocks of the first 16 measurements
        if (ZB >= 2) { // (no measurement kernel in front that would have done it)
            if (tid == 0) { // this launch's status words start clean (write-through: no dirty line of them may outlive a later write-through set)
                __hip_atomic_store(a.flags + 0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.flags + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (flags[3], the stall word, is sequence valued: never cleared)
            }
        }
        if (ZB == 2) {
            if (tid < 16 && tid < M) {
                int lidx;
                const MeasOut o = measure_j(a.zb_mf, tid, lidx);
#pragma unroll
                for (int e = 0; e < 6; ++e)
                    sC16[tid * 6 + e] = o.c[e];
            }
        }
        double sv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (tid < 256 && i < M && jj < M) { // Sigma's blocks requested before the barrier: in flight while the C blocks are evaluated
            const int* lmg = ZB == 2 ? a.zb_mf.lmidx : a.zb_lmidx;
            const int li = 21 + 3 * (a.zb_ident ? i : lmg[i]), lj2 = 21 + 3 * (a.zb_ident ? jj : lmg[jj]);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    sv[3 * c + r] = a.zb_sig[li + r + (size_t)(lj2 + c) * a.zb_ld];
        }
        if (ZB == 2)
            __syncthreads();
        if (tid < 256) {
            double blk[2][2] = {{(i == jj) ? 1.0 : 0.0, 0.0}, {0.0, (i == jj) ? 1.0 : 0.0}};
            if (i < M && jj < M) {
                double ci[6], cj2[6];
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    ci[e] = ZB == 2 ? sC16[i * 6 + e] : a.zb_C[e * a.zb_Mcap + i];
                    cj2[e] = ZB == 2 ? sC16[jj * 6 + e] : a.zb_C[e * a.zb_Mcap + jj];
                }
                bz_S_block(ci, cj2, sv, i == jj, a.zb_var, blk);
            }
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
                    sD[2 * i + aa + (2 * jj + bb) * 33] = blk[aa][bb];
        }
        __syncthreads();
        // Wave 0 eliminates; L_0^-1 goes straight to sLk and, write-through, to its published tile. No barrier behind it: the other waves take up their roles
        // now (the tail waves fetch block row 1's tiles while the first tile is being eliminated - 1.7 us of the first step otherwise), everything they read of
        // this wave's is behind an LDS counter. The flag of L_0^-1 goes up from the pivot wave's first step, when the stores have been acknowledged (*).
        if (wave == 0) {
            double* l0 = la_tile(a, la_i_linv(a, 0));
            ldl_inverse_tile_put(
                sD, 33, min(32, a.m),
                [sLk, l0](int r, int c, double v) {
                    sLk[r + c * CH_LDP] = v;
                    if (HOME)
                        la_st_l(l0 + r + 32 * c, v); // for the S half-rows (home XCD) ...
                    la_st(l0 + r + 32 * c, v);       // ... and, written through, for the T half-rows
                },
                a.flags, swork);
            la_lds_set(cnt + LC_L, 1);
        }
    } else {
        double* l0 = la_tile(a, la_i_linv(a, 0));
        for (int e = tid; e < 1024; e += LA_T) {
            const double v = a.Linv0[e];
            sLk[(e & 31) + (e >> 5) * CH_LDP] = v;
            if (HOME)
                la_st_l(l0 + e, v);
            la_st(l0 + e, v);
        }
        la_stores_done();
        __syncthreads(); // the only workgroup barrier of the owner
        if (tid == 0) {
            if (HOME)
                la_raise_fl(a, la_f_linv(a, 0));
            la_raise_f(a, la_f_linv(a, 0));
            __hip_atomic_store(cnt + LC_L, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    // ------------------------------------------------------------------------------------------------------------------ wave 0: the pivot chain
    // Round 4 (second half): the pivot wave forms the block it needs FIRST itself. D_(k+1)[0:16, 0:16] = D'_00 - c_0 c_0^T with c_0 = R1[0:16, :] L_k^-T is 20
    // register-operand MFMAs on the blocks of L_k^-1 this wave still holds from its last elimination (elimination layout: no data movement, result in place,
    // ~0.55 us), against ~1.3 us for the four-wave post-work with its two LDS hand-offs and the read-back of the tile. The rows 16 .. 31 of c and the blocks
    // (1, 0), (1, 1) of D_(k+1) are still waves 5 / 7's: they arrive in sD while the first 16 x 16 block is being eliminated. Same products in the same order as
    // before (the zero block of the triangular L_k^-1 skipped): W stays bit-identical to the launch chain.
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3); // this wave is the critical path of the whole frame
        double o1[4], o2[4], xl[4];    // L_k^-1: blocks (0, 0), (1, 1), (1, 0) in elimination layout, lane (lr, lk) holds X[lr][lk + 4 q]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o1[q] = sLk[lr + (lk + 4 * q) * CH_LDP];
            xl[q] = sLk[16 + lr + (lk + 4 * q) * CH_LDP];
            o2[q] = sLk[16 + lr + (16 + lk + 4 * q) * CH_LDP];
        }
        for (int k = 0; k + 1 < NJ; ++k) {
            const int I = k + 1;
            if (a.tr_steps && lane == 0 && k < 32)
                a.tr_steps[k] = wall_clock64();
            // R1 and D'_00 of block row k + 1 (the tail of step k - 1 left them in sY / sDq); sX free: the tail of step k - 1 and wave 1 have read c^(k-1)
            if (!la_lds_wait(cnt + LC_T, 4 * (k + 1), s_abort) || (k >= 1 && !la_lds_wait(cnt + LC_CCOPIED, k, s_abort)))
                return;
            if (a.dbg && lane == 0)
                a.dbg[8 * k + 3] = wall_clock64();
            double r11[4], r12[4], dq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = lk + 4 * q;
                r11[q] = sY[lr + c * CH_LDP];
                r12[q] = sY[lr + (16 + c) * CH_LDP];
                dq[q] = sDq[max(lr, c) + 16 * min(lr, c)]; // symmetric fill from the lower triangle, as the tile read-back did
            }
            d4 c11 = {0, 0, 0, 0}, c12 = {0, 0, 0, 0}, cc = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(o1[q], r11[q], c11, 0, 0, 0); // C = I J^T: first operand J
#pragma unroll
            for (int q = 0; q < 4; ++q)
                c12 = __builtin_amdgcn_mfma_f64_16x16x4f64(xl[q], r11[q], c12, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                c12 = __builtin_amdgcn_mfma_f64_16x16x4f64(o2[q], r12[q], c12, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) { // rows 0 .. 15 of c = P^(k)_(k+1) for waves 5 / 7 (blocks (1, 0), (1, 1) of D), the tail of step k and wave 1
                sX[lr + (lk + 4 * q) * CH_LDP] = c11[q];
                sX[lr + (16 + lk + 4 * q) * CH_LDP] = c12[q];
            }
            la_lds_add(cnt + LC_C, 2);
            if (ZB && k == 0) { // (*) L_0^-1's stores were issued ~0.6 us ago
                la_stores_done();
                if (lane == 0) {
                    if (HOME)
                        la_raise_fl(a, la_f_linv(a, 0));
                    la_raise_f(a, la_f_linv(a, 0));
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                cc = __builtin_amdgcn_mfma_f64_16x16x4f64(c11[q], c11[q], cc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                cc = __builtin_amdgcn_mfma_f64_16x16x4f64(c12[q], c12[q], cc, 0, 0, 0);
            const int w2 = min(32, a.m - 32 * I); // rows / columns >= w2 of the last diagonal tile are identity padding
            double d11[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = lk + 4 * q;
                d11[q] = (lr >= w2 || c >= w2) ? ((lr == c) ? 1.0 : 0.0) : dq[q] - cc[q];
            }
            if (a.dbg && lane == 0)
                a.dbg[8 * k + 6] = wall_clock64();
            bool gone = false;
            auto rest = [&](double (&d21)[4], double (&d22)[4]) {
                // waves 5 / 7 have written the blocks (1, 0) / (1, 1) of D_(k+1) to sD; they, wave 2 and wave 3 have read L_k^-1 out of sLk (rewritten below)
                if (!la_lds_wait(cnt + LC_D, 4 * (k + 1), s_abort) || (k >= 1 && !la_lds_wait(cnt + LC_LCOPIED, k, s_abort)))
                    gone = true;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = lk + 4 * q;
                    d21[q] = sD[16 + lr + c * CH_LDP];
                    d22[q] = sD[16 + max(lr, c) + (16 + min(lr, c)) * CH_LDP];
                }
            };
            // Round 4: L_I^-1 leaves this wave through LDS only; wave 3 publishes it. Only the LAST tile, which the T half-rows are waiting for and behind
            // which nothing of the owner runs, is still published from here.
            if (I + 1 < NJ) {
                ldl_inverse_tile_regs(d11, rest, w2, [sLk](int r, int c, double v) { sLk[r + c * CH_LDP] = v; }, a.flags, swork, o1, o2, xl);
                if (gone)
                    return;
                la_lds_set(cnt + LC_L, I + 1); // L_I^-1 complete in sLk
                if (a.dbg && lane == 0)
                    a.dbg[8 * I + 4] = wall_clock64();
            } else {
                double* lt = la_tile(a, la_i_linv(a, I));
                ldl_inverse_tile_regs(
                    d11, rest, w2,
                    [sLk, lt](int r, int c, double v) {
                        sLk[r + c * CH_LDP] = v;
                        la_st(lt + r + 32 * c, v);
                    },
                    a.flags, swork, o1, o2, xl);
                if (gone)
                    return;
                la_lds_set(cnt + LC_L, I + 1);
                if (a.dbg && lane == 0)
                    a.dbg[8 * I + 4] = wall_clock64();
                la_stores_done();
                if (lane == 0)
                    la_raise_f(a, la_f_linv(a, I));
            }
        }
        if (a.tr_steps && lane == 0 && NJ - 1 < 32)
            a.tr_steps[NJ - 1] = wall_clock64();
        return;
    }
    // ------------------------------------------------------------------------------------------------------------------ wave 1: publishes c and b
    if (wave == 1) {
        auto publish = [&](const double (&v)[16], int I_, int k_) { // 32 x 32 tile of P^(k_)_I_: write through, wait for the acknowledgement, raise both half-row flags
            double* t = la_tile(a, la_i_p(a, I_, k_));
            if (HOME) { // first for the S half-rows next door (L2 of the home XCD), then written through for the T half-rows
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    la_st_l(t + lane + 64 * i, v[i]);
                la_stores_done();
                if (lane < 2)
                    la_raise_fl(a, la_f_p(a, k_, 2 * I_ + lane));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i)
                la_st(t + lane + 64 * i, v[i]);
            la_stores_done();
            if (lane < 2)
                la_raise_f(a, la_f_p(a, k_, 2 * I_ + lane));
        };
        for (int k = 0; k + 1 < NJ; ++k) {
            if (!la_lds_wait<true>(cnt + LC_C, 4 * (k + 1), s_abort))
                return;
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = lane + 64 * i;
                v[i] = sX[(e & 31) + (e >> 5) * CH_LDP];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            la_lds_set(cnt + LC_CCOPIED, k + 1);
            publish(v, k + 1, k);
            if (k + 2 < NJ) { // b of the same step, ~2 us later
                if (!la_lds_wait<true>(cnt + LC_B, 2 * (k + 1), s_abort))
                    return;
#pragma unroll
                for (int i = 0; i < 16; ++i) { // element (r, c) of b: half r >> 4, kept at [(r & 15) + c CH_LDP]
                    const int e = lane + 64 * i, r = e & 31, c = e >> 5;
                    v[i] = b_keep(k, r >> 4)[(r & 15) + c * CH_LDP];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                la_lds_set(cnt + LC_BCOPIED, k + 1);
                publish(v, k + 2, k);
            }
        }
        return;
    }
    // ------------------------------------------------------------------------------------------------------------------ wave 3: polls for the block rows' hand-offs
    // ... and, round 4, publishes L_1^-1 .. L_(NJ-2)^-1 out of sLk (the pivot wave keeps its issue slots for the elimination). The two jobs alternate in the order
    // the data appears: L_k^-1 is complete ~2 us before block row k + 2 hands its tiles over (which needs L_(k-1)^-1 and earlier ones only: no cycle), and the next
    // L^-1 another ~2.5 us later. The flag of L_k^-1 is the first link of the chain L -> block row -> U -> tail, so it goes up as soon as its stores are acknowledged.
    if (wave == 6) {
        // Round 5: ONE loop over both jobs instead of taking them in turns (and on wave 6: wave 3, SIMD 3, is a tail wave now - see the quadrants below). In turns, a hand-off whose flag went up while L_k^-1 was being published (stores, acknowledgement,
        // flag - twice in the HOME placement) was seen 1.8 us late (profiles/r05_h4_lookahead_trace.txt), and L_k^-1 waited for the hand-off in front of it. Every pass:
        // if the next L^-1 is complete in sLk, copy it out (HOME: plain stores + acknowledgement + local flag first; then the written-through copy, whose flag goes up with the
        // next pass' wait); then look at the next block row's flags once. A poller on wave 4 instead costs the pivot wave, its SIMD-mate, 0.8 us per step (measured).
        int next_u = 1, next_l = 1, remote_k = -1;
        const int* const uf = HOME ? a.pubfl : a.pubf;
        while (next_u < NJ || next_l <= NJ - 2 || remote_k >= 0) {
            if (next_l <= NJ - 2 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt + LC_L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= next_l + 1) {
                asm volatile("" ::: "memory");
                const int k = next_l++;
                double v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int e = lane + 64 * i;
                    v[i] = sLk[(e & 31) + (e >> 5) * CH_LDP];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                la_lds_set(cnt + LC_LCOPIED, k); // the pivot wave may overwrite sLk
                double* t = la_tile(a, la_i_linv(a, k));
                if (remote_k >= 0) { // (the written-through copy in front of this one: acknowledged long ago)
                    la_stores_done();
                    if (lane == 0)
                        la_raise_f(a, la_f_linv(a, remote_k));
                }
                if (HOME) {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        la_st_l(t + lane + 64 * i, v[i]);
                    la_stores_done();
                    if (lane == 0)
                        la_raise_fl(a, la_f_linv(a, k));
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    la_st(t + lane + 64 * i, v[i]);
                remote_k = k;
            }
            int fv = pl.seq;
            if (next_u < NJ && lane < 2)
                fv = __hip_atomic_load(uf + la_f_u(a, next_u, 0) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // global_load_dword sc1
            la_stores_done(); // the flags are in - and so is the acknowledgement of every store in front of them
            if (remote_k >= 0) {
                if (lane == 0)
                    la_raise_f(a, la_f_linv(a, remote_k));
                remote_k = -1;
            }
            if (next_u < NJ) {
                if (__builtin_amdgcn_readfirstlane(__builtin_popcountll(__ballot(fv == pl.seq))) == 64) { // both halves of block row next_u have handed their tiles over
                    la_lds_set(cnt + LC_U, next_u);
                    if (a.dbg && lane == 0 && next_u >= 2)
                        a.dbg[8 * (next_u - 2) + 7] = wall_clock64(); // the hand-off the tail of step next_u - 2 needs
                    ++next_u;
                } else if ((long long)wall_clock64() > pl.deadline) {
                    pl.s_abort[1] = 1;
                    return;
                }
            } else if (next_l <= NJ - 2)
                __builtin_amdgcn_s_sleep(2);
            if (*(volatile int*)s_abort)
                return;
        }
        return;
    }
    // ------------------------------------------------------------------------------------------------------------------ waves 2, 4..7: post-work and tail
    const bool post = wave == 5 || wave == 7; // rows 16 .. 31 of c and the blocks (1, 0) / (1, 1) of D_(k+1); rows 0 .. 15 and the block (0, 0) are the pivot wave's
    const int pw = wave & 3;
    const bool tailw = wave == 2 || wave == 3 || wave == 5 || wave == 7;
    if (wave == 4) // no work (round 3: a fourth post-work wave; a poller here, even asleep between its looks, slows the pivot wave, its SIMD-mate)
        return;
    // Quadrants of the tail: wave 2 (0, 0), wave 5 (1, 0), wave 3 (1, 1), wave 7 (0, 1). Round 5: b = P^(k)_I2 is formed by the two waves WITHOUT post-work (2: rows 0 .. 15,
    // 3: rows 16 .. 31; round 4: 2 and 7): they request the block row's tiles when its hand-off is seen and have b ready when c^(k) is, while the post-work waves' requests
    // would have to fly across their post-work (measured: the post-work then takes 2.3 instead of 1.1 us) or go out behind it. The two sit on different SIMDs (2 and 3), and
    // the fp64 MFMAs of a step are spread 48 / 36 / 68 over the SIMDs 1 / 2 / 3 (round 4: 48 / 72 / 44 with both tail-only waves on SIMD 2, whose 1.5 us of tail products
    // were what the pivot wave waited for); the poller / publisher of L^-1 has moved to wave 6.
    const int tq = wave == 2 ? 0 : (wave == 5 ? 1 : (wave == 3 ? 3 : 2)), ihT = tq & 1, jhT = tq >> 1;
    double* sDq2 = swork + 800;          // 16 x 16: the block (1, 1) of D' on its way from the tail wave 3 to the post-work wave 7 (free like sDq)
    double dacc[4] = {0, 0, 0, 0};
    auto put_prepared = [&](const double (&r1)[4], const double (&dp)[4]) { // a tail wave's quadrant of R1 and D' to where the post-work finds them
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sY[16 * ihT + lr + (16 * jhT + lk + 4 * q) * CH_LDP] = r1[q];
            if (wave == 2)
                sDq[lr + 16 * (lk + 4 * q)] = dp[q];
            else if (wave == 3)
                sDq2[lr + 16 * (lk + 4 * q)] = dp[q];
            else
                dacc[q] = dp[q]; // (wave 5: its own post-work's block (1, 0); wave 7: the block (0, 1), never read)
        }
    };
    if (tailw) { // block row 1: no panel to apply
        if (!la_lds_wait<true>(cnt + LC_U, 1, s_abort))
            return;
        const double* t1 = la_tile(a, la_i_u(a, 1, 0)) + (16 * ihT + lr) + 32 * (16 * jhT + lk);
        const double* t0 = la_tile(a, la_i_u(a, 1, 1)) + (16 * ihT + lr) + 32 * (16 * jhT + lk);
        double u1r[4], u0r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u1r[q] = t1[128 * q];
            u0r[q] = t0[128 * q];
        }
        put_prepared(u1r, u0r);
        la_lds_add(cnt + LC_T);
    }
    // Round 5: a tail's tiles are REQUESTED as soon as wave 3 has seen the block row's hand-off - the tail-only waves 2 and 6 wait for it in front of their wait for
    // c^(k), the waves 5 and 7 look once before their post-work - and land while c^(k) is being formed. Before, the requests went out when c^(k) was complete and the tail
    // waited 1.0 - 1.9 us for them with the pivot wave waiting behind it (the loop post-work -> tail -> post-work of DESIGN.md section 3.1).
    double u2i[8], u1r[4], u0r[4], p3[8], p3t[8];
    int have = -1; // block row whose tiles are in (or on their way into) the registers above
    auto fetch_tiles = [&](int I2) {
        if (tq == 0 || tq == 3) // first: b is formed from it, and everybody waits for b (loads return in the order they were requested)
            la_operand(la_tile(a, la_i_u(a, I2, 2)), ihT, u2i);
        if (I2 >= 3) { // complete before the block row raised its U flags (published at its last panel)
            la_operand(la_tile(a, la_i_p(a, I2, I2 - 3)), ihT, p3);
            if (tq == 1)
                la_operand(la_tile(a, la_i_p(a, I2, I2 - 3)), 0, p3t);
        }
        const double* t1 = la_tile(a, la_i_u(a, I2, 0)) + (16 * ihT + lr) + 32 * (16 * jhT + lk);
        const double* t0 = la_tile(a, la_i_u(a, I2, 1)) + (16 * ihT + lr) + 32 * (16 * jhT + lk);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u1r[q] = t1[128 * q];
            u0r[q] = t0[128 * q];
        }
        have = I2;
    };
    for (int k = 0; k + 1 < NJ; ++k) { // step k produces D_(k+1)
        const int I = k + 1;
        // L_k^-1 as B operand, both column halves, for the two waves that form b in the tail (wave 2: rows 0 .. 15, wave 7: rows 16 .. 31): read from sLk
        // now, before the elimination of D_(k+1) rewrites it - the pivot wave writes L_(k+1)^-1 only when wave 2 has counted itself into LC_D as well
        double lkop[2][8];
        if (wave == 2 || wave == 3) {
            if (!la_lds_wait(cnt + LC_L, k + 1, s_abort))
                return;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int st = 0; st < 8; ++st)
                    lkop[ch][st] = sLk[16 * ch + lr + (4 * st + lk) * CH_LDP];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            la_lds_add(cnt + LC_D);
        }
        if (tailw && !post && k + 2 < NJ) { // waves 2, 3: nothing else to do until c^(k) is complete
            if (!la_lds_wait<true>(cnt + LC_U, k + 2, s_abort)) // wave 6 saw both U flags of block row k + 2
                return;
            fetch_tiles(k + 2);
        }
        if (post) {
            if (!la_lds_wait(cnt + LC_L, k + 1, s_abort) || !la_lds_wait(cnt + LC_T, 4 * (k + 1), s_abort))
                return;
            if (k >= 1 && !la_lds_wait(cnt + LC_CCOPIED, k, s_abort)) // wave 1 has read the previous c out of sX (it did, 3 us ago)
                return;
            if (wave == 7) { // the block (1, 1) of D', from wave 3's tail: read before this wave counts itself into LC_C (wave 3's next tail starts behind LC_C and rewrites it)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dacc[q] = sDq2[lr + 16 * (lk + 4 * q)];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            const int ch = pw >> 1; // c = P^(k)_I = R1 L_k^-T : sub-tile (1, ch)
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 8; ++st)
                if (ch == 1 || st < 4) // the triangular L_k^-1 has no columns >= 16 in its rows < 16
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sLk[16 * ch + lr + (4 * st + lk) * CH_LDP], sY[16 + lr + (4 * st + lk) * CH_LDP], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sX[16 + lr + (16 * ch + lk + 4 * q) * CH_LDP] = acc[q];
            la_lds_add(cnt + LC_C);
            if (!la_lds_wait(cnt + LC_C, 4 * (k + 1), s_abort))
                return;
            const int jhU = pw >> 1; // the block (1, jhU) of D_(k+1)
            d4 acc2 = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int c = 4 * st + lk;
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(sX[16 * jhU + lr + c * CH_LDP], sX[16 + lr + c * CH_LDP], acc2, 0, 0, 0);
            }
            const int w2 = min(32, a.m - 32 * I); // rows / columns >= w2 of the last diagonal tile are identity padding
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r_ = 16 + lr, c_ = 16 * jhU + lk + 4 * q;
                sD[r_ + c_ * CH_LDP] = (r_ >= w2 || c_ >= w2) ? ((r_ == c_) ? 1.0 : 0.0) : dacc[q] - acc2[q];
            }
            la_lds_add(cnt + LC_D);
            if (a.dbg && wave == 5 && lane == 0)
                a.dbg[8 * k + 5] = wall_clock64();
        }
        if (tailw && k + 2 < NJ) {
            // tail: block row I2 = k + 2
            const int I2 = k + 2;
            if (!la_lds_wait<true>(cnt + LC_C, 4 * (k + 1), s_abort)) // (wave 2) c complete in sX, every post-work wave done with sY
                return;
            if (a.dbg && wave == 5 && lane == 0)
                a.dbg[8 * k + 0] = wall_clock64();
            if (have != I2) {
                if (!la_lds_wait<true>(cnt + LC_U, I2, s_abort)) // wave 3 saw both U flags of block row I2
                    return;
                fetch_tiles(I2);
            }
            // b = P^(k)_I2 = U2 L_k^-T. Wave 2 forms its rows 0 .. 15, wave 3 its rows 16 .. 31 (the two column halves' accumulators ARE the operand layout:
            // column 16 ch + lk + 4 q); the triangular L_k^-1 has no columns >= 16 in its rows < 16. The rows are kept in LDS for waves 5 and 7, for the
            // next tail and for wave 1, which publishes them.
            double bi[8], bj[8];
            if (tq == 0 || tq == 3) {
                d4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    if (st < 4)
                        b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lkop[0][st], u2i[st], b0, 0, 0, 0);
                    b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lkop[1][st], u2i[st], b1, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bi[q] = b0[q];
                    bi[4 + q] = b1[q];
                    bj[q] = b0[q];
                    bj[4 + q] = b1[q];
                }
                if (k >= 2 && !la_lds_wait<true>(cnt + LC_BCOPIED, k - 1, s_abort)) // wave 1 has read what this slot held two steps ago (long since)
                    return;
                double* keep = b_keep(k, ihT);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    keep[lr + (lk + 4 * q) * CH_LDP] = b0[q];
                    keep[lr + (16 + lk + 4 * q) * CH_LDP] = b1[q];
                }
                la_lds_add(cnt + LC_B);
                if (a.dbg && wave == 3 && lane == 0 && k < 32)
                    a.dbg[8 * (32 + k) + 6] = wall_clock64(); // b complete in LDS
            } else {
                if (!la_lds_wait<true>(cnt + LC_B, 2 * (k + 1), s_abort)) // both halves of this step's b are in LDS
                    return;
                const double *bir = b_keep(k, ihT), *bjr = b_keep(k, jhT);
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    bi[st] = bir[lr + (4 * st + lk) * CH_LDP];
                    bj[st] = bjr[lr + (4 * st + lk) * CH_LDP];
                }
            }
            if (a.dbg && wave == 5 && lane == 0)
                a.dbg[8 * k + 1] = wall_clock64();
            d4 r = {0, 0, 0, 0}, d = {0, 0, 0, 0}, r3 = {0, 0, 0, 0};
            if (I2 >= 3) { // the product the block row left out: P^(I2-3)_I2 (P^(I2-3)_(I2-1))^T, the second factor being the previous tail's b
                const double* bp = b_keep(k - 1, jhT);
#pragma unroll
                for (int st = 0; st < 8; ++st)
                    r3 = __builtin_amdgcn_mfma_f64_16x16x4f64(bp[lr + (4 * st + lk) * CH_LDP], p3[st], r3, 0, 0, 0);
            }
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int c = 4 * st + lk;
                r = __builtin_amdgcn_mfma_f64_16x16x4f64(sX[16 * jhT + lr + c * CH_LDP], bi[st], r, 0, 0, 0); // b c^T (c = P^(k)_(k+1), the post-work's)
                if (tq != 2)                                                                                   // the block (0, 1) of D' is above the diagonal: never read
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(bj[st], bi[st], d, 0, 0, 0);                      // b b^T
            }
            if (tq == 1 && I2 >= 3) {
                // Round 4: the block (1, 0) of the diagonal tile misses its product of the block row's last panel as well, P^(I2-3)_bottom (P^(I2-3)_top)^T: the
                // bottom half-row would have to wait for the top one's factor rows (store acknowledgement + flag poll + load, ~2 us in the middle of the
                // loop L -> block row -> U -> tail that bounds the owner's step); the owner has both halves of that tile in hand anyway
                d4 e = {0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 8; ++st)
                    e = __builtin_amdgcn_mfma_f64_16x16x4f64(p3t[st], p3[st], e, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    u0r[q] -= e[q];
            }
            double r1[4], dp[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                r1[q] = (u1r[q] - r3[q]) - r[q];
                dp[q] = u0r[q] - d[q];
            }
            put_prepared(r1, dp);
            la_lds_add(cnt + LC_T);
            if (a.dbg && wave == 5 && lane == 0)
                a.dbg[8 * k + 2] = wall_clock64();
            if (a.dbg && wave != 5 && lane == 0 && k < 32)
                a.dbg[8 * (32 + k) + (wave == 2 ? 3 : (wave == 3 ? 4 : 5))] = wall_clock64(); // the other three tail waves (the first T half-row uses columns 0 .. 2 of these rows)
        }
    }
}

// ---- a half block row ----------------------------------------------------------------------------------------------------------------
// One workgroup per 16-ROW half of a block row (round 3; round 2 had one per 32-row block row): half-row h = 2 I + s of S block row I (s = 0 top,
// 1 bottom), or the t-th 16 rows of T. The trailing update is fp64-MFMA-throughput bound per CU in the early panels (12 tiles x 32 MFMAs of ~100
// cycles on 4 SIMDs = 4 us per panel for a 32-row block row against an owner step of 4.5 us), so that the late S block rows fell 1-2 panels behind
// and the owner waited for them from panel 8 on; halving the rows per workgroup halves that time and puts 66 instead of 34 CUs to work at N = 200,
// 161 instead of 80 at N = 500. It also halves the tiles a wave keeps in registers (4 / 8 instead of 7 / 16): the 17 .. 32-panel instantiation needs
// neither the LDS operand ring nor spills any more.
//   wave w: jh = w & 1 is the 16-column half of a tile, jr = w >> 1 the tile column modulo 4; acc[t] = Z(h, J = 4 t + jr)[:, 16 jh .. 16 jh + 15].
//   A wave's B operand is rows 16 jh .. 16 jh + 15 of P_J, i.e. what ONE half-row (2 J + jh) published: flags are per (panel, half-row).
//   P^(p)_h = Z(h, p) L_p^-T is formed by waves 0 / 1 (column halves; the zero block of the triangular L_p^-1 skipped).
template <int MAXT, int ZB, bool HOME> // ZB: the rows of Z are built here; HOME: the S half-rows share an XCD with the owner (la_st_l)
__device__ __forceinline__ void la_row(const LaArgs& a, const int hidx, double* smem, int* s_abort, int* row_cnt, const LaPoll& pl) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    double* sLinv = smem;                   // L_p^-1, operand layout [r + c CH_LDP]
    double* sPI = smem + 32 * CH_LDP;       // P_h: 16 rows x 32 columns
    double* sT = smem + 2 * 32 * CH_LDP;    // the panel tile's 16 rows in operand layout; at the very end the Gamma partial sums
    double* sYv = smem + 3 * 32 * CH_LDP;   // yTilde row of the panel (32)
    double* sZp = sYv + 32;                 // z_p as 8 partial sums over 4 columns of L_p^-1 each ([8][32])
    const int NJ = a.NJ, m = a.m, rows = a.rows, ldz = a.ldz, seq = a.seq;
    const int NJe = la_live_panels(a); // panels that are factorised (LaArgs::live_cols; NJ otherwise)
    const bool srow = hidx < 2 * NJ;
    const bool loc = HOME && srow;                 // this half-row and everybody it exchanges tiles with sit on the home XCD
    const int* const fl = loc ? a.pubfl : a.pubf;  // the flags it polls
    const int I = hidx >> 1, s = hidx & 1;
    if (srow && I >= NJe) // rows of S behind the last live column: nobody asks for them
        return;
    const int row0 = srow ? 16 * hidx : m + 16 * (hidx - 2 * NJ);
    const int ilim = srow ? min(m, row0 + 16) : min(rows, row0 + 16);
    const int Jmax = srow ? I : NJe - 1;
    const bool ylast = (!srow) && (rows - 1 >= row0) && (rows - 1 < row0 + 16); // this half-row holds the yTilde row
    const int yloc = rows - 1 - row0;
    const int jh = wave & 1, jr = wave >> 1;
    const int ri = row0 + lr;
    const int ric = min(ri, ilim - 1);
    double acc[MAXT][4];
    if (!ZB) {
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int J = 4 * t + jr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = min(32 * J + 16 * jh + lk + 4 * q, m - 1);
                acc[t][q] = (J <= Jmax) ? a.Z[ric + (size_t)j * ldz] : 0.0;
            }
        }
        if (ylast && tid < 32)
            la_put16(a.puby + 16 * (size_t)tid, a.Z[(rows - 1) + (size_t)min(tid, m - 1) * ldz], seq);
    } else {
        // This half-row's 16 rows of Z, built here (no k_build_Z launch, no Z in memory): chunks of 256 columns are evaluated into LDS in a thread mapping
        // whose Sigma loads are row-coalesced (T: thread = (row, measurement), 3 loads for 2 entries; S: thread = (measurement of the row pair, measurement
        // of the column pair), one 3 x 3 block of Sigma for 4 entries), then read in the accumulator layout with the clamps of the loads above.
        constexpr int CW = 256;
        double* sZ = smem; // [r + 16 c], r < 16, c < CW
        double* sC = smem + 16 * CW;               // ZB = 2: C_j and yTilde_j of every measurement, [8 j + e] (M <= 256)
        const int M = a.zb_M, Mcap = a.zb_Mcap, ldS = a.zb_ld, nS = rows - 1 - m;
        const int ncols = min(m, 32 * (Jmax + 1)); // columns this half-row ever reads
        // A T half-row requests Sigma's entries first, for every measurement of the thread ((row, measurement j = g + 32 k): 3 values): the loads need the
        // index map only, and are in flight while the C blocks are evaluated (ZB = 2: ~2.5 us)
        constexpr int KT = 8, KS = 4; // M <= 256
        const int* lmg = ZB == 2 ? a.zb_mf.lmidx : a.zb_lmidx;
        const bool ident = a.zb_ident != 0;
        const int jmax = min(M, (ncols + 1) / 2); // measurements whose columns this half-row reads
        double preT[KT][3];
        const int r16 = tid & 15, tt = (row0 - m) + r16; // T: row of Sigma / of T; tt == nS: the yTilde row
        const int i8 = tid & 7, iS = 8 * hidx + i8;      // S: measurement of the row pair (2 i, 2 i + 1)
        if (!srow) {
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int jj = (tid >> 4) + 32 * k;
                if (tt < nS && jj < jmax) {
                    const int lj = 21 + 3 * (ident ? jj : lmg[jj]);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        preT[k][c] = a.zb_sig[tt + (size_t)(lj + c) * ldS];
                }
            }
        }
        // (the S half-rows request their 3 x 3 blocks chunk by chunk below: 72 more registers across the evaluation of C spill, and they are not the late ones)
        if (ZB == 2) {
            for (int jj = tid; jj < M; jj += LA_T) {
                int lidx;
                const MeasOut o = measure_j(a.zb_mf, jj, lidx);
#pragma unroll
                for (int e = 0; e < 6; ++e)
                    sC[8 * jj + e] = o.c[e];
                sC[8 * jj + 6] = o.yt[0];
                sC[8 * jj + 7] = o.yt[1];
            }
            __syncthreads();
        }
        auto c_of = [&](int jj, double (&cj)[6]) {
#pragma unroll
            for (int e = 0; e < 6; ++e)
                cj[e] = ZB == 2 ? sC[8 * jj + e] : a.zb_C[e * Mcap + jj];
        };
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[t][q] = 0.0;
        for (int c0 = 0; c0 < ncols; c0 += CW) {
            const int jbeg = c0 / 2, jend = min(M, (min(ncols, c0 + CW) + 1) / 2);
            if (srow) {
                if (iS < M) {
                    double ci[6];
                    c_of(iS, ci);
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        const int jj = (tid >> 3) + 64 * k;
                        if (jj >= jbeg && jj < jend) {
                            double cj[6], blk[2][2], sv[9];
                            const int li = 21 + 3 * (ident ? iS : lmg[iS]), lj = 21 + 3 * (ident ? jj : lmg[jj]);
#pragma unroll
                            for (int c = 0; c < 3; ++c)
#pragma unroll
                                for (int r = 0; r < 3; ++r)
                                    sv[3 * c + r] = a.zb_sig[li + r + (size_t)(lj + c) * ldS];
                            c_of(jj, cj);
                            bz_S_block(ci, cj, sv, iS == jj, a.zb_var, blk);
#pragma unroll
                            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                                for (int bb = 0; bb < 2; ++bb)
                                    sZ[(2 * i8 + aa) + 16 * (2 * jj + bb - c0)] = blk[aa][bb];
                        }
                    }
                }
            } else if (tt <= nS) {
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const int jj = (tid >> 4) + 32 * k;
                    if (jj >= jbeg && jj < jend) {
                        double o0, o1;
                        if (tt < nS) {
                            double cj[6];
                            c_of(jj, cj);
                            bz_T_pair(preT[k][0], preT[k][1], preT[k][2], cj, o0, o1);
                        } else {
                            o0 = ZB == 2 ? sC[8 * jj + 6] : a.zb_ytil[2 * jj];
                            o1 = ZB == 2 ? sC[8 * jj + 7] : a.zb_ytil[2 * jj + 1];
                        }
                        sZ[r16 + 16 * (2 * jj - c0)] = o0;
                        sZ[r16 + 16 * (2 * jj + 1 - c0)] = o1;
                    }
                }
            }
            __syncthreads();
            const int rl = min(lr, ilim - 1 - row0);
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                const int J = 4 * t + jr;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jc = min(32 * J + 16 * jh + lk + 4 * q, m - 1);
                    if (J <= Jmax && jc >= c0 && jc < c0 + CW)
                        acc[t][q] = sZ[rl + 16 * (jc - c0)];
                }
            }
            __syncthreads();
        }
        if (ylast && tid < 32) {
            const int jc = min(tid, m - 1);
            la_put16(a.puby + 16 * (size_t)tid, ZB == 2 ? sC[8 * (jc >> 1) + 6 + (jc & 1)] : a.zb_ytil[jc], seq);
        }
    }
    double gsum = 0.0; // thread (r = tid & 15, c = tid >> 4): Gamma share of row row0 + r from column c of every panel
    // S half-rows: panels 0 .. I-3, then the hand-off of U2 / U1 / U0 to the owner, which forms b = P^(I-2)_I and c = P^(I-1)_I itself. ONE product of the
    // last panel is left to the owner as well: Z(I, I-1) -= P^(I-3)_I (P^(I-3)_(I-1))^T, whose second factor is the owner's own b of the step before -
    // waiting for it here would close a cycle b -> block row -> U -> next b of 5.3 us per step (measured). T half-rows: all panels.
    const int np = srow ? max(I - 2, 0) : NJe;
    auto hand_off = [&]() {
        // to the owner: rows 16 s .. 16 s + 15 of U2 = Z(I, I-2), U1 = Z(I, I-1) and U0 = Z(I, I) with the panels <= I-3 applied (all waves call this)
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int J = 4 * t + jr;
            if (J >= 0 && J >= I - 2 && J <= I) {
                double* u = la_tile(a, la_i_u(a, I, J == I ? 1 : (J == I - 1 ? 0 : 2)));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (loc)
                        la_st_l(u + (16 * s + lr) + 32 * (16 * jh + lk + 4 * q), acc[t][q]);
                    else
                        la_st(u + (16 * s + lr) + 32 * (16 * jh + lk + 4 * q), acc[t][q]);
                }
            }
        }
        la_stores_done();
        __syncthreads();
        if (tid == 0) {
            if (loc)
                la_raise_fl(a, la_f_u(a, I, s));
            else
                la_raise_f(a, la_f_u(a, I, s));
            if (a.dbg && s == 0 && I < 32)
                a.dbg[8 * (64 + I) + 4] = wall_clock64();
        }
    };
    for (int p = 0; p < np; ++p) {
        // laundered once per panel: otherwise the body's address / mask expressions are loop invariant, get hoisted and spilled
        int lrv = lr, lkv = lk;
        asm volatile("" : "+v"(lrv), "+v"(lkv));
        const int w = min(32, m - 32 * p);
        // (a) L_p^-1 -> LDS; this half-row's part of the panel tile Z(h, p) -> LDS in operand layout (masked like the chain's operand loads); yTilde row
        if (a.dbg && tid == 0 && srow && s == 0 && p == np - 1 && I < 32)
            a.dbg[8 * (64 + I) + 5] = wall_clock64();
        la_wait(fl + la_f_linv(a, p), 1, pl);
        {
            const double* lt = la_tile(a, la_i_linv(a, p));
            const double v0 = lt[tid], v1 = lt[tid + LA_T];
            sLinv[(tid & 31) + (tid >> 5) * CH_LDP] = v0;
            sLinv[((tid + LA_T) & 31) + ((tid + LA_T) >> 5) * CH_LDP] = v1;
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (4 * t + jr == p) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 16 * jh + lk + 4 * q;
                    sT[lr + c * CH_LDP] = (ri < ilim && c < w) ? acc[t][q] : 0.0;
                }
            }
        if (!srow && wave == 7 && lane < 32) {
            const double yv = la_get16(a.puby + 512 * (size_t)p + 16 * (size_t)lane, pl);
            sYv[lane] = lane < w ? yv : 0.0;
        }
        if (tid == 0 && s_abort[1])
            s_abort[0] = 1;
        __syncthreads();
        if (*s_abort)
            return;
        // EQF_OPT_TRACE: the first T half-row, every panel ([32 + p]); the top half of every S block row at its LAST panel, the one in front of its hand-off ([64 + I])
        const bool dbg_row = a.dbg && tid == 0 && ((hidx == 2 * NJ && p < 32) || (srow && s == 0 && p == np - 1 && I < 32));
        unsigned long long* dbr = a.dbg + 8 * (hidx == 2 * NJ ? 32 + p : 64 + I);
        if (dbg_row)
            dbr[0] = wall_clock64();
        // (b) P_h = Z(h, p) L_p^-T on waves 0, 1 (column half ch = wave; L_p^-1 is lower triangular: its columns >= 16 are zero in the rows < 16);
        //     z_p partials on waves 4..7
        if (wave < 2) {
            const int ch = wave;
            d4 pacc = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 8; ++st)
                if (ch == 1 || st < 4)
                    pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(sLinv[16 * ch + lr + (4 * st + lk) * CH_LDP], sT[lr + (4 * st + lk) * CH_LDP], pacc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sPI[lr + (16 * ch + lk + 4 * q) * CH_LDP] = pacc[q];
            if (srow) { // the factor rows leave for the other half-rows straight from the accumulators
                double* pt = la_tile(a, la_i_p(a, I, p));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (loc)
                        la_st_l(pt + (16 * s + lr) + 32 * (16 * ch + lk + 4 * q), pacc[q]);
                    else
                        la_st(pt + (16 * s + lr) + 32 * (16 * ch + lk + 4 * q), pacc[q]);
                }
                la_stores_done();
            }
        } else if (!srow && wave >= 4) {
            // z_p[c] = sum_q yTilde_p[q] L_p^-1[c][q] as 8 partial sums (thread (c, h): q = 4 h .. 4 h + 3), summed in a fixed order by the readers
            const int c = tid & 31, h = (tid >> 5) & 7;
            double z = 0.0;
#pragma unroll
            for (int q = 4 * h; q < 4 * h + 4; ++q)
                z = fma(sYv[q], sLinv[c + q * CH_LDP], z); // yTilde entries >= w are zero
            sZp[32 * h + c] = z;
        }
        __syncthreads();
        // (c) P_h: flag for the consumers of an S half-row; final W rows (+ Gamma) for a T half-row
        if (srow) {
            if (tid == 0) {
                if (loc)
                    la_raise_fl(a, la_f_p(a, p, hidx));
                else
                    la_raise_f(a, la_f_p(a, p, hidx));
            }
        } else {
            const int r = tid & 15, c = tid >> 4;
            const double pv = sPI[r + c * CH_LDP];
            if (row0 + r < rows && c < w)
                a.W[(row0 + r) + (size_t)(32 * p + c) * ldz] = pv;
            const double zc = ((sZp[c] + sZp[32 + c]) + (sZp[64 + c] + sZp[96 + c])) + ((sZp[128 + c] + sZp[160 + c]) + (sZp[192 + c] + sZp[224 + c]));
            gsum = fma(pv, c < w ? zc : 0.0, gsum);
        }
        if (dbg_row)
            dbr[1] = wall_clock64();
        double aI[8];
#pragma unroll
        for (int st = 0; st < 8; ++st)
            aI[st] = sPI[lr + (4 * st + lk) * CH_LDP];
        if (loc && wave == 0) { // HOME: the T half-rows' copy of these factor rows, written through; its flag goes up at the end of this panel's trailing update (below)
            double* pt = la_tile(a, la_i_p(a, I, p));
#pragma unroll
            for (int st = 0; st < 8; ++st)
                la_st(pt + (16 * s + lr) + 32 * (4 * st + lk), aI[st]);
        }
        // this wave's operands: rows 16 jh .. of P^(p)_J = what half-row 2 J + jh published; lane t watches the flag of tile t. An S half-row's own
        // P_h (J = I, jh = s) is in LDS; its block above the diagonal (J = I, jh > s) is never used.
        {
            const int J = 4 * lane + jr;
            if (lane < MAXT && J > p && J <= Jmax && !(srow && J == I && jh >= s) && !(srow && p == I - 3 && J >= I - 1)) {
                const int* f = fl + la_f_p(a, p, 2 * J + jh);
                for (;;) {
                    const int v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v == pl.seq || !la_retry(pl))
                        break;
                }
            }
            asm volatile("" ::: "memory");
        }
        // two tiles per round trip: both operand sets are requested before the first product needs one
#pragma unroll
        for (int t0 = 0; t0 < MAXT; t0 += 2) {
            double bjs[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int J = 4 * (t0 + u) + jr;
                if (t0 + u < MAXT && J > p && J <= Jmax && !(srow && J == I && jh > s) && !(srow && p == I - 3 && J == I - 1) && !(srow && p == I - 3 && J == I && jh < s)) {
                    if (srow && J == I && jh == s) { // diagonal block of an S half-row: both operands are P_h
#pragma unroll
                        for (int st = 0; st < 8; ++st)
                            bjs[u][st] = sPI[lrv + (4 * st + lkv) * CH_LDP];
                    } else
                        la_operand(la_tile(a, la_i_p(a, J, p)), jh, bjs[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = (t0 + u < MAXT) ? t0 + u : MAXT - 1;
                const int J = 4 * (t0 + u) + jr;
                if (t0 + u < MAXT && J > p && J <= Jmax && !(srow && J == I && jh > s) && !(srow && p == I - 3 && J == I - 1) && !(srow && p == I - 3 && J == I && jh < s)) {
                    d4 d = {0, 0, 0, 0};
#pragma unroll
                    for (int st = 0; st < 8; ++st)
                        d = __builtin_amdgcn_mfma_f64_16x16x4f64(bjs[u][st], aI[st], d, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[t][q] -= d[q];
                    if (ylast && J == p + 1 && lrv == yloc) {
                        // the yTilde row of the next panel is final now: publish it for every T half-row's z_(p+1)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            la_put16(a.puby + 512 * (size_t)(p + 1) + 16 * (size_t)(16 * jh + lkv + 4 * q), acc[t][q], seq);
                    }
                }
            }
        }
        if (loc && wave == 0) { // the written-through copy of P_h(p) has been on its way for the whole trailing update
            la_stores_done();
            if (lane == 0)
                la_raise_f(a, la_f_p(a, p, hidx));
        }
        if (dbg_row)
            dbr[2] = wall_clock64();
    }
    if (srow)
        hand_off(); // after panel I-3 (block rows 1 and 2 have no panel to wait for: their tiles go to the owner as they are)
    if (!srow) {
        if (NJe < NJ) { // the columns behind the last factorised panel are those of discarded measurements: W = 0 there (T is zero in them)
            const int r = tid & 15;
            if (row0 + r < rows)
                for (int col = 32 * NJe + (tid >> 4); col < m; col += 32)
                    a.W[(row0 + r) + (size_t)col * ldz] = 0.0;
        }
        __syncthreads(); // the last panel's readers of sT / sZp are done
        sT[tid] = gsum;  // [c][r]
        __syncthreads();
        const int row = row0 + tid;
        if (tid < 16 && row >= m && row < rows - 1) {
            double g = 0.0;
#pragma unroll
            for (int c = 0; c < 32; ++c)
                g += sT[tid + 16 * c];
            la_st(a.gamma + (row - m), g);
            if (a.early_door && row - m < 21) { // EQF_OPT_EARLY_DOORBELL: Gamma's sensor rows go to the host's packet from here (fenced at system scope before this half-row counts itself in)
                a.early_gamma_host[row - m] = g;
                __threadfence_system();
            }
        }
    }
}

// ---- the rows of Z built by a half-row of the 17 .. 32-panel form (round 6: EQF_OPT_Z_IN_LOOKAHEAD above 16 panels) -----------------------------------
// What la_row's ZB prologue does up to 256 measurements, for up to 512 and for the tile ranges of split half-rows (part A: tiles 0 .. Jlo - 1, part B: Jlo .. I): the
// columns [32 Jfirst, 32 (Jmax + 1)) of this half-row's 16 rows of Z = [S ; T ; yTilde^T], with k_build_Z's expressions (bz_T_pair / bz_S_block: not a bit may differ),
// the output blocks C_j / yTilde from memory (ZB = 1: the measurement kernel, ZB = 3: the propagation kernel's observer blocks).
// * A T half-row - these end the kernel at 25 .. 32 panels, so what their prologue takes the frame pays - requests every Sigma entry it needs up front (thread = (row,
//   measurement), 16 x 3 entries per thread at 512 measurements: one memory round trip) and stages chunks of 256 columns through LDS into the accumulator layout.
//   (Measured at N = 500, kernel span against the k_build_Z route's 197 us: the next chunk's entries requested under the evaluation of the current one, 4 round trips,
//   +9.6 us; every entry evaluated by the lane that keeps it - 24 loads per tile and lane, no LDS, no barrier - +18.6 us: at 256 registers the loads go out a few at a time.)
// * An S half-row evaluates 3 x 3 blocks of Sigma, one per 2 x 2 block of S (thread = (measurement of the row pair, measurement of the column pair)), into LDS in chunks
//   of 256 columns, the next chunk's Sigma entries requested in front of the evaluation of the current one, and reads them in the accumulator layout.
template <int MAXT, bool srow>
__device__ __forceinline__ void la_build_rows2(const LaArgs& a, const int hidx, double* smem, const int row0, const int ilim, const int Jfirst, const int Jmax, double (&acc)[MAXT][4]) {
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jh = wave & 1, jr = wave >> 1;
    const int m = a.m, rows = a.rows;
    const int M = a.zb_M, Mcap = a.zb_Mcap, ldS = a.zb_ld, nS = rows - 1 - m;
    const int* lmg = a.zb_lmidx;
    const bool ident = a.zb_ident != 0;
    if constexpr (!srow) {
        // thread = (row r16, measurement (tid >> 4) + 32 k): every Sigma entry of the prologue requested up front (KT x 3 per thread: one memory round trip), then chunks of
        // 256 columns through LDS into the accumulator layout
        constexpr int CW = 256, KT = 16; // M <= 512
        double* sZ = smem;               // [r + 16 c], r < 16, c < CW
        const int ncols = min(m, 32 * (Jmax + 1));
        const int r16 = tid & 15, tt = (row0 - m) + r16; // row of Sigma / of T; tt == nS: the yTilde row
        const int jmax = min(M, (ncols + 1) / 2);
        double preT[KT][3];
        if (tt < nS) {
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int jj = (tid >> 4) + 32 * k;
                if (jj < jmax) {
                    const int lj = 21 + 3 * (ident ? jj : lmg[jj]);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        preT[k][c] = a.zb_sig[tt + (size_t)(lj + c) * ldS];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[t][q] = 0.0;
        // the output blocks into LDS, once, behind Sigma's requests ([6 j + e]; read from memory inside the chunk loop they were one more round trip in front of every chunk:
        // the first T half-row's panel 0 began 15.4 us after the kernel and 10.6 with the blocks in LDS, profiles/r06_la_row2_N500_lookahead_trace_{before,after}.txt)
        double* sC = smem + 16 * CW;
        for (int jj = tid; jj < jmax; jj += LA_T)
#pragma unroll
            for (int e = 0; e < 6; ++e)
                sC[6 * jj + e] = a.zb_C[e * Mcap + jj];
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < KT / 4; ++ch) {
            const int c0 = CW * ch;
            if (c0 < ncols) { // (uniform)
                if (tt <= nS) {
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const int k = 4 * ch + k4;
                        const int jj = (tid >> 4) + 32 * k;
                        if (jj < jmax) {
                            double o0, o1;
                            if (tt < nS) {
                                double cj[6];
#pragma unroll
                                for (int e = 0; e < 6; ++e)
                                    cj[e] = sC[6 * jj + e];
                                bz_T_pair(preT[k][0], preT[k][1], preT[k][2], cj, o0, o1);
                            } else {
                                o0 = a.zb_ytil[2 * jj];
                                o1 = a.zb_ytil[2 * jj + 1];
                            }
                            sZ[r16 + 16 * (2 * jj - c0)] = o0;
                            sZ[r16 + 16 * (2 * jj + 1 - c0)] = o1;
                        }
                    }
                }
                __syncthreads();
                const int rl = min(lr, ilim - 1 - row0);
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    const int J = 4 * t + jr;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int jc = min(32 * J + 16 * jh + lk + 4 * q, m - 1);
                        if (J <= Jmax && J >= Jfirst && jc >= c0 && jc < c0 + CW)
                            acc[t][q] = sZ[rl + 16 * (jc - c0)];
                    }
                }
                __syncthreads();
            }
        }
    } else {
        constexpr int CW = 256;
        double* sZ = smem; // [r + 16 c], r < 16, c < CW
        const int ncols = min(m, 32 * (Jmax + 1)); // columns this half-row ever reads
        const int cbeg = ((32 * Jfirst) / CW) * CW; // first chunk that holds one of them
        auto c_of = [&](int jj, double (&cj)[6]) {
#pragma unroll
            for (int e = 0; e < 6; ++e)
                cj[e] = a.zb_C[e * Mcap + jj];
        };
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[t][q] = 0.0;
        constexpr int KS = 2; // column measurements per thread and chunk of 128 measurements (8 row measurements x 64 lanes of column measurements)
        const int i8 = tid & 7, iS = 8 * hidx + i8; // measurement of the row pair (2 i, 2 i + 1)
        double preS[KS][9];
        double ci[6];
        if (iS < M)
            c_of(iS, ci);
        auto request = [&](int c0) {
            const int jbeg = c0 / 2, jend = min(M, (min(ncols, c0 + CW) + 1) / 2);
            if (iS < M) {
                const int li = 21 + 3 * (ident ? iS : lmg[iS]);
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const int jj = jbeg + (tid >> 3) + 64 * k;
                    if (jj < jend) {
                        const int lj = 21 + 3 * (ident ? jj : lmg[jj]);
#pragma unroll
                        for (int c = 0; c < 3; ++c)
#pragma unroll
                            for (int r = 0; r < 3; ++r)
                                preS[k][3 * c + r] = a.zb_sig[li + r + (size_t)(lj + c) * ldS];
                    }
                }
            }
        };
        if (cbeg < ncols)
            request(cbeg);
        // the column measurements' output blocks into LDS, once, behind the first requests (see the T half-rows above)
        double* sC = smem + 16 * CW;
        {
            const int jlo = cbeg / 2, jhi = min(M, (ncols + 1) / 2);
            for (int jj = jlo + tid; jj < jhi; jj += LA_T)
#pragma unroll
                for (int e = 0; e < 6; ++e)
                    sC[6 * jj + e] = a.zb_C[e * Mcap + jj];
            __syncthreads();
        }
        for (int c0 = cbeg; c0 < ncols; c0 += CW) {
            const int jbeg = c0 / 2, jend = min(M, (min(ncols, c0 + CW) + 1) / 2);
            if (iS < M) {
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const int jj = jbeg + (tid >> 3) + 64 * k;
                    if (jj < jend) {
                        double cj[6], blk[2][2];
#pragma unroll
                        for (int e = 0; e < 6; ++e)
                            cj[e] = sC[6 * jj + e];
                        bz_S_block(ci, cj, preS[k], iS == jj, a.zb_var, blk);
#pragma unroll
                        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                            for (int bb = 0; bb < 2; ++bb)
                                sZ[(2 * i8 + aa) + 16 * (2 * jj + bb - c0)] = blk[aa][bb];
                    }
                }
            }
            if (c0 + CW < ncols)
                request(c0 + CW); // in flight across the barrier and the read-out below
            __syncthreads();
            const int rl = min(lr, ilim - 1 - row0);
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                const int J = 4 * t + jr;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jc = min(32 * J + 16 * jh + lk + 4 * q, m - 1);
                    if (J <= Jmax && J >= Jfirst && jc >= c0 && jc < c0 + CW)
                        acc[t][q] = sZ[rl + 16 * (jc - c0)];
                }
            }
            __syncthreads();
        }
    }
}

// ---- a half block row with a look-ahead of its own (17 .. 32 panels) ----------------------------------------------------------------------
// la_row above walks a panel as wait L_p -> barrier -> P_h -> barrier -> wait P_J -> trailing update, one after the other. With 17 .. 32 panels
// the trailing update of the early panels is MFMA bound per CU (31 tiles x 16 MFMAs of ~100 cycles on 4 SIMDs = 5.3 us against an owner step of 4.7 us)
// and the 3.7 us of waits around it are not overlapped with anything: measured 10.5 us per panel at N = 500, the half-rows 40 us behind the owner
// by panel 8 (profiles/r03_v1_N500_lookahead_trace.txt). Here the panel's NEXT tile goes first: the two waves that own tile Z(h, p + 1) apply panel p
// to it before anything else, form P_h(p+1) = Z(h, p+1) L_(p+1)^-T with the L_(p+1)^-1 they fetched on the way (the owner is ahead), publish it
// (S half-rows) and only then turn to their other tiles - while the other six waves are in the trailing update of panel p. One workgroup barrier per
// panel; P_h, L^-1 and z in LDS are double buffered by panel parity. Same products in the same order per tile: bit-identical to la_row and to the chain.
template <int MAXT, bool srow, int ZB = 0>
// Round 4, N > 256: an S half-row of a LATE block row (I >= a.split_from) holds up to 32 tiles, and the trailing update of the early panels is MFMA-issue bound on its
// compute unit (8 tiles per wave: 5.3 us per panel against an owner step of 4.4): those rows handed their tiles over late and the owner's steps were 4.3 - 8.7 us
// (profiles/r04_v2_N500_lookahead_trace.txt). Their tile columns are split over TWO workgroups: part A (partA) holds the tiles 0 .. Jlo - 1, forms and publishes the factor
// rows P_h(p) of the panels p < Jlo and leaves; part B holds the tiles Jlo .. I, takes P_h(p) for p < Jlo from memory like any other operand (no barrier, no LDS: its waves
// walk those panels independently), and is an ordinary half-row from panel Jlo - 1 on. Same products in the same order per tile. Jlo: a multiple of 4, <= I - 2 (the three
// tiles handed to the owner are part B's).
__device__ __forceinline__ void la_row2(const LaArgs& a, const int hidx, double* smem, int* s_abort, int* cnt, const LaPoll& pl, const int Jlo = 0, const bool partA = false) {
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int B = 32 * CH_LDP;
    auto sLinvB = [&](int b) -> double* { return smem + (b & 1) * B; };      // L_p^-1, operand layout [r + c CH_LDP], rows 0 .. 31
    // P_h(p): 16 rows x 32 columns, three buffers (p mod 3: a T half-row stores the W rows of panel p at the top of panel p + 1, while the look-ahead pair
    // already writes P_h(p + 2)): the rows 32 .. 47 of the two L^-1 blocks and the rows 16 .. 31 of the tile block
    auto sPIB = [&](int b) -> double* { return (b % 3) < 2 ? smem + (b % 3) * B + 32 : smem + 2 * B + 16; };
    double* sT = smem + 2 * B;                                               // the next panel tile's 16 rows in operand layout (rows 0 .. 15); at the very end the Gamma partial sums
    auto sZpB = [&](int b) -> double* { return smem + 3 * B + 256 * (b % 3); }; // z_p as 8 partial sums over 4 columns of L_p^-1 each ([8][32]), three buffers like P_h
    int* pair_cnt = cnt;     // the two waves of a look-ahead pair: tile and L^-1 in LDS
    int* pub_cnt = cnt + 1;  // ... their halves of P_h acknowledged (S half-rows)
    const int NJ = a.NJ, m = a.m, rows = a.rows, ldz = a.ldz, seq = a.seq;
    const int I = hidx >> 1, s = hidx & 1;
    const int row0 = srow ? 16 * hidx : m + 16 * (hidx - 2 * NJ);
    const int ilim = srow ? min(m, row0 + 16) : min(rows, row0 + 16);
    const int Jmax = srow ? (partA ? Jlo - 1 : I) : NJ - 1;
    const int Jfirst = (srow && !partA) ? Jlo : 0;
    const bool partB = srow && !partA && Jlo > 0;
    const bool ylast = (!srow) && (rows - 1 >= row0) && (rows - 1 < row0 + 16); // this half-row holds the yTilde row
    const int yloc = rows - 1 - row0;
    const int jh = wave & 1, jr = wave >> 1;
    const int ri = row0 + lr;
    const int ric = min(ri, ilim - 1);
    double acc[MAXT][4];
    if constexpr (ZB == 0) {
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int J = 4 * t + jr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = min(32 * J + 16 * jh + lk + 4 * q, m - 1);
                acc[t][q] = (J <= Jmax && J >= Jfirst) ? a.Z[ric + (size_t)j * ldz] : 0.0;
            }
        }
        if (ylast && tid < 32)
            la_put16(a.puby + 16 * (size_t)tid, a.Z[(rows - 1) + (size_t)min(tid, m - 1) * ldz], seq);
    } else {
        // round 6: no k_build_Z launch in front of the 17 .. 32-panel form either - this half-row's tiles are built here (an S half-row's la_build_rows2 ends with a barrier: smem is free again)
        la_build_rows2<MAXT, srow>(a, hidx, smem, row0, ilim, Jfirst, Jmax, acc);
        if (ylast && tid < 32)
            la_put16(a.puby + 16 * (size_t)tid, a.zb_ytil[min(tid, m - 1)], seq);
    }
    double gsum = 0.0;
    const int np = srow ? (partA ? Jlo : max(I - 2, 0)) : NJ; // see la_row
    auto hand_off = [&]() {
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int J = 4 * t + jr;
            if (J >= 0 && J >= I - 2 && J <= I) {
                double* u = la_tile(a, la_i_u(a, I, J == I ? 1 : (J == I - 1 ? 0 : 2)));
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    la_st(u + (16 * s + lr) + 32 * (16 * jh + lk + 4 * q), acc[t][q]);
            }
        }
        la_stores_done();
        __syncthreads();
        if (tid == 0)
            la_raise_f(a, la_f_u(a, I, s));
    };
    // the panel tile Z(h, q) out of the accumulators into sT, masked like the chain's operand loads (the waves that own it)
    auto tile_to_lds = [&](int q) {
        const int wq = min(32, m - 32 * q);
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (4 * t + jr == q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * jh + lk + 4 * e;
                    sT[lr + c * CH_LDP] = (ri < ilim && c < wq) ? acc[t][e] : 0.0;
                }
            }
    };
    // P_h(q) = Z(h, q) L_q^-T, column half ch, from sT and sLinvB(q) into sPIB(q) (L_q^-1 is lower triangular: its columns >= 16 are zero in the rows < 16)
    auto form_p = [&](int q, int ch, d4& pacc) {
        const double* sL = sLinvB(q);
        pacc = d4{0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < 8; ++st)
            if (ch == 1 || st < 4)
                pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(sL[16 * ch + lr + (4 * st + lk) * CH_LDP], sT[lr + (4 * st + lk) * CH_LDP], pacc, 0, 0, 0);
        double* sP = sPIB(q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            sP[lr + (16 * ch + lk + 4 * e) * CH_LDP] = pacc[e];
    };
    auto publish_p = [&](int q, int ch, const d4& pacc) { // S half-rows: this wave's half of the factor rows, write-through
        double* pt = la_tile(a, la_i_p(a, I, q));
#pragma unroll
        for (int e = 0; e < 4; ++e)
            la_st(pt + (16 * s + lr) + 32 * (16 * ch + lk + 4 * e), pacc[e]);
    };
    // z_q[c] = sum_k yTilde_q[k] L_q^-1[c][k] as 8 partial sums (thread (c, h) of the waves 4 .. 7: k = 4 h .. 4 h + 3), summed in a fixed order by the readers
    // ymine: entry c = lane & 31 of the yTilde row of panel q (one 16-byte word per lane, one round trip per wave); the four this thread needs come from its neighbours
    auto z_partials = [&](int q, const double ymine) {
        const int wq = min(32, m - 32 * q);
        const int c = tid & 31, h = (tid >> 5) & 7;
        const double* sL = sLinvB(q);
        double z = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * h + i;
            const double yv = fetch_lane(ymine, 4 * k);
            z = fma(k < wq ? yv : 0.0, sL[c + k * CH_LDP], z);
        }
        sZpB(q)[32 * h + c] = z;
    };
    // Round 6: the waves 0 .. 3 store the W rows (thread (r, c): the columns c and c + 16, a Gamma share for each - the same sums as one column per thread), and they do it at the
    // END of a panel, in front of the barrier they reach first. The store used to stand at the top of the panel in every wave, and the waves 4 .. 7 - the panel's longest:
    // their yTilde word comes back behind an `s_waitcnt vmcnt(0)` - waited for its acknowledgement there: 2.5 us from the panel's top to their first operand request
    // where the waves 0 .. 3 took 0.8 (profiles/r06_la_row2_N500_lookahead_trace_before.txt), in every panel of every T half-row, which are the ones that end the kernel.
    double gsum2 = 0.0;
    auto store_w = [&](int q) {
        if (tid < 256) {
            const int wq = min(32, m - 32 * q);
            const int r = tid & 15;
            const double* sZp = sZpB(q);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = (tid >> 4) + 16 * u;
                const double pv = sPIB(q)[r + c * CH_LDP];
                if (row0 + r < rows && c < wq)
                    a.W[(row0 + r) + (size_t)(32 * q + c) * ldz] = pv;
                const double zc = ((sZp[c] + sZp[32 + c]) + (sZp[64 + c] + sZp[96 + c])) + ((sZp[128 + c] + sZp[160 + c]) + (sZp[192 + c] + sZp[224 + c]));
                if (u == 0)
                    gsum = fma(pv, c < wq ? zc : 0.0, gsum);
                else
                    gsum2 = fma(pv, c < wq ? zc : 0.0, gsum2);
            }
        }
    };
    if (partB) {
        // part B of a split half-row: the panels 0 .. Jlo - 2 on its own tiles, every wave by itself
        for (int p = 0; p + 1 < Jlo; ++p) {
            {
                // lane t watches the flag of tile t's operand, lane 62 the flag of this half-row's own factor rows (formed by part A)
                const int J = 4 * lane + jr;
                const bool need = lane < MAXT && J >= Jlo && J <= I && !(J == I && jh >= s);
                if (need || lane == 62) {
                    const int* f = a.pubf + la_f_p(a, p, need ? 2 * J + jh : hidx);
                    for (;;) {
                        const int v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v == pl.seq || !la_retry(pl))
                            break;
                    }
                }
                asm volatile("" ::: "memory");
            }
            double aI[8];
            la_operand(la_tile(a, la_i_p(a, I, p)), s, aI);
#pragma unroll
            for (int t0 = 0; t0 < MAXT; t0 += 2) {
                double bjs[2][8];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int J = 4 * (t0 + u) + jr;
                    if (t0 + u < MAXT && J >= Jlo && J <= I && !(J == I && jh > s)) {
                        if (J == I && jh == s) { // diagonal block: both operands are this half-row's factor rows
#pragma unroll
                            for (int st = 0; st < 8; ++st)
                                bjs[u][st] = aI[st];
                        } else
                            la_operand(la_tile(a, la_i_p(a, J, p)), jh, bjs[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = (t0 + u < MAXT) ? t0 + u : MAXT - 1;
                    const int J = 4 * (t0 + u) + jr;
                    if (t0 + u < MAXT && J >= Jlo && J <= I && !(J == I && jh > s)) {
                        d4 d = {0, 0, 0, 0};
#pragma unroll
                        for (int st = 0; st < 8; ++st)
                            d = __builtin_amdgcn_mfma_f64_16x16x4f64(bjs[u][st], aI[st], d, 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[t][e] -= d[e];
                    }
                }
            }
        }
        // P_h(Jlo - 1), the last one part A forms, into the LDS buffer where the loop below expects this half-row's factor rows; the counters of the look-ahead
        // pairs start where a half-row that had walked the panels 0 .. Jlo - 2 itself would have them
        la_wait(a.pubf + la_f_p(a, Jlo - 1, hidx), 1, pl);
        {
            const double* pt = la_tile(a, la_i_p(a, I, Jlo - 1));
            double* sP = sPIB(Jlo - 1);
            const int r = tid & 15, c = tid >> 4;
            sP[r + c * CH_LDP] = pt[(16 * s + r) + 32 * c];
        }
        if (tid == 0) {
            pair_cnt[0] = 2 * (Jlo - 1);
            pub_cnt[0] = 2 * (Jlo - 1);
            if (s_abort[1])
                s_abort[0] = 1;
        }
        __syncthreads();
        if (*s_abort)
            return;
    } else if (np > 0) {
        // prologue: panel 0 the plain way (every thread two entries of L_0^-1; waves 0 / 1 form P_h(0))
        la_wait(a.pubf + la_f_linv(a, 0), 1, pl);
        {
            const double* lt = la_tile(a, la_i_linv(a, 0));
            const double v0 = lt[tid], v1 = lt[tid + LA_T];
            double* sL = sLinvB(0);
            sL[(tid & 31) + (tid >> 5) * CH_LDP] = v0;
            sL[((tid + LA_T) & 31) + ((tid + LA_T) >> 5) * CH_LDP] = v1;
        }
        tile_to_lds(0);
        __syncthreads();
        if (wave < 2) {
            d4 pacc;
            form_p(0, wave, pacc);
            if (srow) {
                publish_p(0, wave, pacc);
                la_stores_done();
            }
        } else if (!srow && wave >= 4)
            z_partials(0, la_get16(a.puby + 16 * (size_t)(tid & 31), pl));
        if (tid == 0 && s_abort[1])
            s_abort[0] = 1;
        __syncthreads();
        if (*s_abort)
            return;
        if (srow && tid == 0)
            la_raise_f(a, la_f_p(a, 0, hidx));
    }
    int pre_flag = 0; // (lane-private) the word this lane watches at the top of the coming panel, as it read a panel before
    for (int p = partB ? Jlo - 1 : 0; p < np; ++p) {
        int lrv = lr, lkv = lk;
        asm volatile("" : "+v"(lrv), "+v"(lkv));
        const int w = min(32, m - 32 * p);
        const double* sPI = sPIB(p);
        const bool dbg_row = a.dbg && tid == 0 && (hidx == 2 * NJ || hidx == 2 * (NJ - 2)) && p < 32;
        unsigned long long* dbr = a.dbg + 8 * ((hidx == 2 * NJ ? 32 : 64) + p);
        if (dbg_row)
            dbr[0] = wall_clock64();
        if (dbg_row)
            dbr[1] = wall_clock64();
        // (d) the trailing update Z(h, J) -= P_h(p) P_J(p)^T; the look-ahead pair first brings the next panel tile forward
        const bool ahead = p + 1 < np && jr == ((p + 1) & 3); // this wave owns half of Z(h, p + 1)
        auto tile_used_at = [&](int q, int t, int& J) -> bool { // tile t of this wave takes part in the trailing update of panel q
            J = 4 * t + jr;
            return t < MAXT && J > q && J <= Jmax && !(srow && J == I && jh > s) && !(srow && q == I - 3 && J == I - 1) && !(srow && q == I - 3 && J == I && jh < s);
        };
        auto tile_used = [&](int t, int& J) -> bool { return tile_used_at(p, t, J); };
        // this wave's operands: rows 16 jh .. of P^(p)_J = what half-row 2 J + jh published; lane t watches the flag of tile t, lane 63 of a look-ahead
        // wave the flag of L_(p+1)^-1
        // Round 6: every lane asks for the word it will watch at panel p + 1 NOW (no wait) and looks at the answer at the top of that panel. A half-row that runs behind
        // the owner and its peers - all of them from panel ~12 on at 25 .. 32 panels, ~30 us behind - finds it raised and goes straight to its operand loads: one memory round
        // trip less per panel, which is what such a half-row needs to catch up (its panel was flag poll -> L^-1 / operands -> products -> barrier at the owner's pace).
        auto watch = [&](int q, bool& want) -> const int* {
            int J;
            const bool need = tile_used_at(q, lane, J) && !(srow && J == I && jh == s);
            const bool look = q + 1 < np && jr == ((q + 1) & 3) && lane == 63;
            want = need || look;
            return a.pubf + (look ? la_f_linv(a, q + 1) : la_f_p(a, q, 2 * (need ? J : 0) + jh));
        };
        {
            bool want;
            const int* f = watch(p, want);
            const bool need = want, look = false; // (one condition below: this lane watches a word)
            const bool seen = want && a.watch_ahead && pre_flag == pl.seq; // raised a panel ago already
            if (p + 1 < np) {
                bool wantn;
                const int* fn = watch(p + 1, wantn);
                pre_flag = wantn ? __hip_atomic_load(fn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            }
            // Round 6: z_p is the work of the four waves that do NOT hold the look-ahead pair of this panel (the pair brings tile p + 1 forward first: ~1.5 us more than the
            // others; yTilde + z_p are ~1 us) - waves 4 .. 7 if the pair sits in 0 .. 3 (jr = (p + 1) & 3 < 2), waves 0 .. 3 otherwise. It was always 4 .. 7, which then
            // ended every second panel 2 us behind the waves 0 .. 3
            const bool zwave = (wave >= 4) == (((p + 1) & 3) < 2 || !a.watch_ahead);
            if (!srow && zwave && p > 0) {
                // four waves of a T half-row: the yTilde row of THIS panel (published a panel ago) on the same round trip as the flags, then z_p
                const char* yp = a.puby + 512 * (size_t)p + 16 * (size_t)(tid & 31);
                v4i r;
                for (;;) {
                    int v = pl.seq;
                    if ((need || look) && !seen)
                        v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(yp) : "memory");
                    if ((v == pl.seq && r.z == pl.seq && r.w == ~pl.seq) || !la_retry(pl))
                        break;
                }
                z_partials(p, __hiloint2double(r.y, r.x));
            } else if ((need || look) && !seen) {
                for (;;) {
                    const int v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v == pl.seq || !la_retry(pl))
                        break;
                }
            }
            asm volatile("" ::: "memory");
        }
        const bool dbg_wave = a.dbg && lane == 0 && (hidx == 2 * NJ || hidx == 2 * (NJ - 2)) && p < 32;
        if (dbg_wave) {
            atomicMax(dbr + 3, (unsigned long long)wall_clock64());
            if (wave == 0)
                dbr[7] = wall_clock64();
        }
        auto apply_tile = [&](int t, const double (&bj)[8]) {
            d4 d = {0, 0, 0, 0};
            // P_h's operand entries are read from LDS again for every tile (index laundered so that the compiler does not keep the 8 doubles in registers
            // across the tiles: the T half-rows of this instantiation sit at the 256-register limit, and 16 registers decide between 0 and 100+ spills)
            int la = lrv + lkv * CH_LDP;
            asm volatile("" : "+v"(la));
#pragma unroll
            for (int st = 0; st < 8; ++st)
                d = __builtin_amdgcn_mfma_f64_16x16x4f64(bj[st], sPI[la + 4 * st * CH_LDP], d, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[t][e] -= d[e];
        };
        bool published = false; // (S look-ahead wave) this wave's half of P_h(p+1) is on its way; the flag is due
        auto flag_when_acknowledged = [&]() {
            if (srow && ahead && !published) {
                la_stores_done();
                if (lane == 0 && __hip_atomic_fetch_add(pub_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 2 * p + 1)
                    la_raise_f(a, la_f_p(a, p + 1, hidx)); // the second of the two halves
                published = true;
            }
        };
        if (ahead) {
            // L_(p+1)^-1: this wave's half of the tile (8 entries per lane), and the operand of tile p + 1, in one round trip
            const double* lt = la_tile(a, la_i_linv(a, p + 1)) + 512 * jh;
            double lv[8], bj[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                lv[i] = lt[lane + 64 * i];
            la_operand(la_tile(a, la_i_p(a, p + 1, p)), jh, bj);
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
                if (4 * t + jr == p + 1) {
                    apply_tile(t, bj);
                    if (ylast && lrv == yloc) {
                        // the yTilde row of the next panel is final now: publish it for every T half-row's z_(p+1) (the only place: the next panel's tile
                        // always goes through here; one copy of the store sequence instead of one per tile - 30 registers in la_row)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            la_put16(a.puby + 512 * (size_t)(p + 1) + 16 * (size_t)(16 * jh + lkv + 4 * e), acc[t][e], seq);
                    }
                }
            tile_to_lds(p + 1);
            double* sL = sLinvB(p + 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = 512 * jh + lane + 64 * i;
                sL[(e & 31) + (e >> 5) * CH_LDP] = lv[i];
            }
            la_lds_add(pair_cnt);
            la_lds_wait(pair_cnt, 2 * (p + 1), s_abort); // the other half of the tile and of L^-1
            d4 pacc;
            form_p(p + 1, jh, pacc);
            if (srow)
                publish_p(p + 1, jh, pacc);
        }
        // The other tiles, from the last one down, ONE per step, the operands of the next requested before the products of the current. Round 4: the requests are
        // unconditional (a tile below the wave's range is a valid address of the same buffer, requested once per panel and never used), so that the number of loads
        // in flight is known at compile time and the wait in front of a tile's products is vmcnt(16): two tiles stay in flight - with requests under run-time conditions the compiler waited
        // for ALL loads in front of every tile (the prefetch bought nothing), and the factor rows P_h were re-read from LDS two entries at a time between the
        // MFMAs (4 exposed LDS round trips per tile: ~150 cycles per MFMA and SIMD where 64 is the issue rate). P_h's operand now stays in 16 registers for
        // the panel (the operand buffers: 3 x 16 registers instead of 4 x 16). A wave's tiles are contiguous in t: what is left out sits at the ends of its range.
        int tlo = MAXT, thi = -1;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            int J;
            if (tile_used(t, J) && !(ahead && J == p + 1)) {
                tlo = min(tlo, t);
                thi = max(thi, t);
            }
        }
        if (thi >= 0) {
            double aI[8], bq0[8], bq1[8], bq2[8]; // two tiles requested ahead of the one whose products run (a round trip is ~1 us, a tile's 8 MFMAs 0.2 us, two waves share a SIMD)
#pragma unroll
            for (int st = 0; st < 8; ++st)
                aI[st] = sPI[lrv + (4 * st + lkv) * CH_LDP];
            bool done = false;
            // (one buffer per t mod 3, chosen at compile time: an array of three indexed by t % 3 went to scratch)
            auto step = [&](auto TT, double (&cur)[8], double (&nxt)[8], double (&nxt2)[8]) {
                constexpr int t = decltype(TT)::value;
                if (done || t > thi)
                    return;
                if (t == thi) {
                    la_operand(la_tile(a, la_i_p(a, 4 * t + jr, p)), jh, cur);
                    if (t > 0)
                        la_operand(la_tile(a, la_i_p(a, 4 * (t > 0 ? t - 1 : 0) + jr, p)), jh, nxt);
                }
                if (t > 1)
                    la_operand(la_tile(a, la_i_p(a, 4 * (t > 1 ? t - 2 : 0) + jr, p)), jh, nxt2);
                flag_when_acknowledged(); // under these round trips (once)
                int J;
                if (tile_used(t, J) && !(ahead && J == p + 1)) {
                    d4 d = {0, 0, 0, 0};
                    if (srow && J == I && jh == s) { // diagonal block of an S half-row: both operands are P_h (its published copy may not have landed yet)
#pragma unroll
                        for (int st = 0; st < 8; ++st)
                            d = __builtin_amdgcn_mfma_f64_16x16x4f64(aI[st], aI[st], d, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int st = 0; st < 8; ++st)
                            d = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[st], aI[st], d, 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[t][e] -= d[e];
                }
                if (t == tlo)
                    done = true;
            };
            // tile t in buffer t mod 3: (7 -> 1), (6 -> 0), (5 -> 2), ...
            if constexpr (MAXT > 4) {
                step(std::integral_constant<int, 7>{}, bq1, bq0, bq2);
                step(std::integral_constant<int, 6>{}, bq0, bq2, bq1);
                step(std::integral_constant<int, 5>{}, bq2, bq1, bq0);
                step(std::integral_constant<int, 4>{}, bq1, bq0, bq2);
            }
            step(std::integral_constant<int, 3>{}, bq0, bq2, bq1);
            step(std::integral_constant<int, 2>{}, bq2, bq1, bq0);
            step(std::integral_constant<int, 1>{}, bq1, bq0, bq2);
            step(std::integral_constant<int, 0>{}, bq0, bq2, bq1);
        }
        flag_when_acknowledged(); // a wave without other tiles
        if (dbg_row)
            dbr[2] = wall_clock64();
        if (dbg_wave) {
            atomicMax(dbr + 4, (unsigned long long)wall_clock64());
            if (wave == 7)
                dbr[5] = wall_clock64();
            if (wave == 3)
                dbr[6] = wall_clock64();
        }
        // (c) final W rows (+ Gamma) of a T half-row - of the panel BEFORE this one (z_(p-1) was completed by the barrier that ended panel p - 1; its buffers are not
        // written again before panel p + 1; nothing waits for W until the very end): see store_w
        if (!srow && p > 0)
            store_w(p - 1);
        if (p + 1 < np) {
            if (tid == 0 && s_abort[1])
                s_abort[0] = 1;
            __syncthreads(); // P_h(p+1) and z_p complete; everybody done with sT and with the buffers of panel p - 1
            if (*s_abort)
                return;
        }
    }
    if (srow && !partA)
        hand_off();
    if (!srow) {
        __syncthreads(); // z of the last panel is complete
        if (np > 0) {
            store_w(np - 1);
        }
        __syncthreads(); // the Gamma partial sums below overwrite the third P_h buffer
        if (tid < 256) {
            sT[tid] = gsum; // [c][r]
            sT[tid + 256] = gsum2;
        }
        __syncthreads();
        const int row = row0 + tid;
        if (tid < 16 && row >= m && row < rows - 1) {
            double g = 0.0;
#pragma unroll
            for (int c = 0; c < 32; ++c)
                g += sT[tid + 16 * c];
            la_st(a.gamma + (row - m), g);
            if (a.early_door && row - m < 21) { // EQF_OPT_EARLY_DOORBELL: Gamma's sensor rows go to the host's packet from here (fenced at system scope before this half-row counts itself in)
                a.early_gamma_host[row - m] = g;
                __threadfence_system();
            }
        }
    }
}

// ZB = 2: the statistics workgroup (block NI). What k_build_Z's statistics row and its first column of measurement groups do in the speculative frame tail:
// absErr / probErr / depth^2 per landmark to the pinned packet, the speculation word if a measured landmark is an outlier candidate (the lift and the
// covariance update behind this kernel then return at once; what this kernel computes is scratch), and C / yTilde / the index map in memory (a retry on the
// launch chain, or the next call with the same measurement, read them).
template <int ZB> // ZB = 3: the C blocks are in memory already (evaluated by the propagation kernel's observer blocks)
__device__ __forceinline__ void la_stats(const LaArgs& a) {
    const MeasFuse& mf = a.zb_mf;
    for (int i = threadIdx.x; i < mf.N; i += LA_T) {
        double abs_err = -1.0, prob_err = -1.0;
        outlier_stats_body<double>(mf.N, mf.Ncap, a.zb_ld, mf.chart, mf.cam, mf.ylm, mf.q0, mf.Qq, mf.Qa, a.zb_sig, mf.out, 0, nullptr, nullptr, nullptr, nullptr, abs_err, prob_err, false, i);
        if (mf.spec_w && abs_err >= 0.0 && (abs_err > mf.thrAbs || prob_err > mf.thrProb)) // the comparisons of VIOFilter.cpp:316-330 (NaN: false)
            *mf.spec_w = mf.spec_seq;
    }
    if (ZB == 3)
        return;
    for (int j = threadIdx.x; j < a.zb_M; j += LA_T) {
        int lidx;
        const MeasOut o = measure_j(mf, j, lidx);
#pragma unroll
        for (int e = 0; e < 6; ++e)
            mf.C[e * a.zb_Mcap + j] = o.c[e];
        mf.ytil[2 * j] = o.yt[0];
        mf.ytil[2 * j + 1] = o.yt[1];
        mf.lmidx_dev[j] = lidx;
    }
}

// A T half-row counts itself in (all threads call this); the last one rings the early doorbell if nobody gave up (LaArgs::early_door). EXTRA: the statistics workgroup of the
// ZB >= 2 forms counts as well (what it wrote to the host's packet is fenced at system scope before it does). The two T half-rows that hold Gamma's sensor rows have written
// them to the host's packet and fenced before they counted in, so the last one's acquire on the counter orders its doorbell store behind them.
// Round 6: no barrier, no store acknowledgement and no release in front of the count (they cost the factorisation 2 us at N = 50: chain 17.6 -> 19.6 us, half of what the earlier
// doorbell gave). The doorbell promises that the update WILL be applied, not that W is in memory (the kernels behind this one wait for the kernel's end as before); what has to be
// visible when it rings - Gamma's sensor rows, the statistics - was written by the counting thread's own wave and fenced at system scope before it got here; `gave_up` is final
// (every wave of a T half-row has passed its last barrier).
template <bool EXTRA>
__device__ __forceinline__ void la_early_count(const LaArgs& a, const bool gave_up) {
    if (threadIdx.x == 0) {
        const int nT = a.NI - (2 * a.NJ - 1) + (EXTRA ? 1 : 0);
        const int add = 1 + (gave_up ? 0x10000 : 0);
        const int tot = __hip_atomic_fetch_add(a.early_cnt, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
        if ((tot & 0xffff) == nT) {
            __hip_atomic_store(a.early_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int f0 = __hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int f3 = __hip_atomic_load(a.flags + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int sp = a.early_spec ? __hip_atomic_load(a.early_spec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            if ((tot >> 16) == 0 && f0 == 0 && f3 != a.seq && !(a.early_spec && sp == a.early_spec_seq)) // (nothing of this thread's to release: what the host reads behind
                *reinterpret_cast<volatile int*>(a.early_door) = a.early_seq;                               //  the bell was fenced at system scope by its writers before they counted in)
        }
    }
}

template <int MAXT, int ZB = 0, bool HOME = false>
__global__ void __launch_bounds__(LA_T) k_chol_lookahead(const LaArgs a) {
    if (a.spec && *a.spec == a.spec_seq) // cancelled speculative tail
        return;
    __shared__ double smem[4 * 32 * CH_LDP + LDL_SBUF]; // static LDS: constant addresses (2.6 us per factorisation at N = 200 against dynamic LDS)
    __shared__ int s_abort[2], s_cnt[LC_COUNT];
    if (threadIdx.x < 2)
        s_abort[threadIdx.x] = 0;
    if (threadIdx.x < LC_COUNT)
        s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const LaPoll pl{(long long)wall_clock64() + a.timeout_ticks, a.seq, s_abort};
    // block 0: the owner; blocks 1 .. 2 NJ - 2: the S half-rows h = 2 .. 2 NJ - 1 (block row 0 is the first diagonal tile, eliminated by k_build_Z);
    // then the T half-rows, numbered on from 2 NJ
    int hidx = (int)blockIdx.x + 1;
    bool owner = blockIdx.x == 0;
    if constexpr (HOME) {
        // HOME placement (up to 16 panels): block b runs on XCD (a.home + b) & 7. The blocks b & 7 == 0 are, in this order, the owner and the S half-rows 2 .. 2 NJ - 1; the
        // blocks of the other seven XCDs are the T half-rows (and the statistics workgroup behind them); what is left of the grid returns at once.
        const int x = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        const int nT = a.NI - (2 * a.NJ - 1);
        const int xcc = la_xcc_id();
        if (xcc != ((a.home + x) & 7)) { // not where the placement assumes: nothing may be exchanged through an L2. Every block of the grid finds the same and leaves.
            if (threadIdx.x == 0) {
                if (blockIdx.x == 0) // (the host learns the offset of this stream's queue from it)
                    __hip_atomic_store(a.flags + 5, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.flags + 4, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.flags + 3, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (a.early_door && x != 0 && 7 * slot + (x - 1) < nT + (ZB >= 2 ? 1 : 0)) // (a T half-row or the statistics workgroup: counted as one that gave up, so that the counter comes back to 0)
                la_early_count<(ZB >= 2)>(a, true);
            return;
        }
        if (x == 0) {
            if (slot > 2 * a.NJ - 2)
                return;
            owner = slot == 0;
            hidx = slot + 1;
        } else {
            const int r = 7 * slot + (x - 1);
            owner = false;
            if (r < nT)
                hidx = 2 * a.NJ + r;
            else if (ZB >= 2 && r == nT) {
                la_stats<ZB>(a);
                if (a.early_door) {
                    __threadfence_system();
                    la_early_count<true>(a, false);
                }
                return;
            } else
                return;
        }
    } else if constexpr (ZB >= 2) {
        // (17 .. 32 panels: the parts A of the split half-rows sit between the T half-rows and the statistics workgroup)
        if ((int)blockIdx.x >= a.NI + (MAXT > 4 && a.split_from < a.NJ ? 2 * (a.NJ - a.split_from) : 0)) { // the statistics workgroup
            la_stats<ZB>(a);
            if (a.early_door) {
                __threadfence_system();
                la_early_count<true>(a, false);
            }
            return;
        }
    }
    if (owner)
        la_owner<ZB, HOME>(a, smem, s_abort, s_cnt, pl);
    else if constexpr (MAXT > 4) { // 17 .. 32 panels: the half-rows with a look-ahead of their own
        const int nbase = a.NI; // owner + S half-rows + T half-rows; behind them the parts A of the split half-rows (block rows >= split_from, two halves each)
        if ((int)blockIdx.x >= nbase) {
            const int e = (int)blockIdx.x - nbase, I = a.split_from + (e >> 1);
            la_row2<MAXT, true, ZB>(a, 2 * I + (e & 1), smem, s_abort, s_cnt, pl, 4 * ((I + 5) >> 3), true);
        } else if (hidx < 2 * a.NJ) {
            const int I = hidx >> 1;
            la_row2<MAXT, true, ZB>(a, hidx, smem, s_abort, s_cnt, pl, I >= a.split_from ? 4 * ((I + 5) >> 3) : 0, false);
        } else
            la_row2<MAXT, false, ZB>(a, hidx, smem, s_abort, s_cnt, pl);
    } else
        la_row<MAXT, ZB, HOME>(a, hidx, smem, s_abort, s_cnt, pl);
    // any wave that saw a timeout reports it (the owner's waves return at different times). The stall word carries the launch's sequence number: nobody has to
    // clear it, so no clear can race with a workgroup that reports early, and a stale word of an earlier launch never matches
    if ((threadIdx.x & 63) == 0 && (s_abort[0] | s_abort[1]))
        __hip_atomic_store(a.flags + 3, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.early_door && !owner && hidx >= 2 * a.NJ && (HOME || (int)blockIdx.x < a.NI)) // a T half-row (not a part A of a split S half-row: those blocks follow the T half-rows)
        la_early_count<(ZB >= 2)>(a, (s_abort[0] | s_abort[1]) != 0);
}

} // namespace eqf
