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
//    Dependencies point from higher to lower block rows and to the owner only, and NI <= 80 workgroups of 64 KB LDS always fit the chip.
#pragma once
#include "eqf_kernels.hpp"

namespace eqf {

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int LA_T = 512;                       // threads per workgroup (8 waves)
constexpr int LA_TILE = 1024;                   // doubles of a published 32 x 32 tile, [r + 32 c]
constexpr long long LA_TIMEOUT_TICKS = 2000000; // 20 ms at 100 MHz

struct LaArgs {
    int rows, m, ldz, NJ, NI, seq;
    long long timeout_ticks; // bound of every device-side wait (100 MHz ticks; default LA_TIMEOUT_TICKS)
    const double* Z;     // [S ; T ; y^T] from k_build_Z (plain memory; complete when this kernel starts)
    double* W;           // out, plain: rows >= m receive W = T L^-T and the z row
    const double* Linv0; // L_0^-1, 32 x 32 column-major, from k_build_Z's first-tile elimination
    double* pub;         // published tiles (offsets below)
    int* pubf;           // their flags
    char* puby;          // the yTilde row of every panel as 16-byte (value, sequence) words
    char* publ;          // unused (an experimental variant published L^-1 as 16-byte words here)
    double* gamma;       // out: Gamma[n]
    int* flags;          // [0] non-positive pivot, [3] stalled
    const int* spec;
    int spec_seq;
    // the frame's results leave from this kernel (what k_lift does behind the launch chain): the last T block row to finish lifts the
    // landmarks, fills the pinned result packet and rings the host doorbell
    int lift_N, lift_Ncap, lift_chart, lift_discrete;
    const double* lift_q0;
    double *lift_Qq, *lift_Qa;
    double *lift_est, *lift_gamma_host; // pinned
    int *lift_flags_host, *lift_done, *lift_door_host;
    int lift_door_seq;
    trace_t* tr_lift;
    trace_t* tr_steps;       // EQF_OPT_TRACE: slot of step 0 (the owner stamps one slot per step), or nullptr
    unsigned long long* dbg; // EQF_OPT_TRACE: per-step stamps inside the owner ([k][8]) and two block rows ([32 + p][8], [64 + p][8]), or nullptr
};
// tiles: [0, NJ) L_p^-1 | [NJ, NJ + NJ^2) P^(p)_J at p NJ + J (a panel's tiles are neighbours) | then U1, U0 of every S block row
// flags: the same indices (one int per tile; U1 / U0 share the flag of U1)
__device__ __forceinline__ int la_i_linv(const LaArgs& a, int p) { return p; }
__device__ __forceinline__ int la_i_p(const LaArgs& a, int J, int p) { return a.NJ + p * a.NJ + J; }
__device__ __forceinline__ int la_i_u(const LaArgs& a, int I, int which) { return a.NJ + a.NJ * a.NJ + 2 * I + which; }
inline size_t la_pub_tiles(int NJ) { return (size_t)NJ + (size_t)NJ * NJ + 2 * (size_t)NJ; }
__device__ __forceinline__ double* la_tile(const LaArgs& a, int idx) { return a.pub + (size_t)LA_TILE * idx; }

__device__ __forceinline__ void la_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } // global_store_dwordx2 sc1
__device__ __forceinline__ void la_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void la_raise(const LaArgs& a, int idx) { __hip_atomic_store(a.pubf + idx, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct LaPoll {
    long long deadline;
    int seq;
    int* s_abort;
};
__device__ __forceinline__ bool la_retry(const LaPoll& pl) {
    if ((long long)wall_clock64() > pl.deadline) {
        *pl.s_abort = 1;
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
__device__ __forceinline__ void la_owner(const LaArgs& a, double* smem, int* s_abort, const LaPoll& pl) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    double* sLk = smem;                  // L_k^-1, operand layout [r + c CH_LDP]
    double* sX = smem + 32 * CH_LDP;     // c_k (read by the pre-work), then c_(k+1) (written by the post-work)
    double* sY = smem + 2 * 32 * CH_LDP; // R1 (written by the pre-work, read by the post-work)
    double* sD = smem + 3 * 32 * CH_LDP; // the diagonal tile handed to the elimination
    double* swork = smem + 4 * 32 * CH_LDP;
    const int NJ = a.NJ;
    {
        double* l0 = la_tile(a, la_i_linv(a, 0));
        for (int e = tid; e < 1024; e += LA_T) {
            const double v = a.Linv0[e];
            sLk[(e & 31) + (e >> 5) * CH_LDP] = v;
            la_st(l0 + e, v);
        }
        la_stores_done();
        __syncthreads();
        if (tid == 0)
            la_raise(a, la_i_linv(a, 0));
    }
    const bool prod = wave >= 4;
    const int pw = wave & 3, ihU = pw & 1, jhU = pw >> 1;
    double dacc[4] = {0, 0, 0, 0};
    for (int k = 0; k + 1 < NJ; ++k) { // this step produces L_(k+1)^-1
        const int I = k + 1;
        if (a.tr_steps && tid == 0 && k < 32)
            a.tr_steps[k] = wall_clock64();
        if (prod) {
            // pre-work (runs while wave 0 eliminates D_k): tiles of block row I with the panels <= I-3 applied, panel I-2 applied here
            if (a.dbg && wave == 4 && lane == 0)
                a.dbg[8 * k + 0] = wall_clock64();
            la_wait(a.pubf + la_i_u(a, I, 0), 1, pl);
            if (k >= 1)
                la_wait(a.pubf + la_i_p(a, I, k - 1), 1, pl);
            if (a.dbg && wave == 4 && lane == 0)
                a.dbg[8 * k + 1] = wall_clock64();
            double u1[4], u0[4];
            {
                const double* t1 = la_tile(a, la_i_u(a, I, 0)) + (16 * ihU + lr) + 32 * (16 * jhU + lk);
                const double* t0 = la_tile(a, la_i_u(a, I, 1)) + (16 * ihU + lr) + 32 * (16 * jhU + lk);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u1[q] = t1[128 * q];
                    u0[q] = t0[128 * q];
                }
            }
            if (k >= 1) {
                double bi[8], bj[8];
                const double* bt = la_tile(a, la_i_p(a, I, k - 1));
                la_operand(bt, ihU, bi);
                la_operand(bt, jhU, bj);
                if (a.dbg && wave == 4 && lane == 0)
                    a.dbg[8 * k + 2] = wall_clock64();
                d4 r = {0, 0, 0, 0}, d = {0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    const int c = 4 * st + lk;
                    r = __builtin_amdgcn_mfma_f64_16x16x4f64(sX[16 * jhU + lr + c * CH_LDP], bi[st], r, 0, 0, 0); // b c_k^T
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(bj[st], bi[st], d, 0, 0, 0);                         // b b^T
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u1[q] -= r[q];
                    u0[q] -= d[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                dacc[q] = u0[q];
                sY[16 * ihU + lr + (16 * jhU + lk + 4 * q) * CH_LDP] = u1[q];
            }
        }
        if (a.dbg && lane == 0 && (wave == 4 || wave == 0))
            a.dbg[8 * k + (wave == 4 ? 3 : 4)] = wall_clock64();
        __syncthreads(); // B1: L_k^-1 in sLk (wave 0), R1 in sY (waves 4..7)
        if (*s_abort)
            return;
        if (a.dbg && tid == 0)
            a.dbg[8 * k + 5] = wall_clock64();
        if (tid == 0 && k >= 1)
            la_raise(a, la_i_linv(a, k)); // wave 0 stored L_k^-1 and waited for its stores before the barrier
        if (prod) {
            // c = P^(k)_I = R1 L_k^-T : sub-tile (ih, ch) = (pw & 1, pw >> 1)
            const int ih = pw & 1, ch = pw >> 1;
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 8; ++st)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sLk[16 * ch + lr + (4 * st + lk) * CH_LDP], sY[16 * ih + lr + (4 * st + lk) * CH_LDP], acc, 0, 0, 0);
            double* ct = la_tile(a, la_i_p(a, I, k));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r_ = 16 * ih + lr, c_ = 16 * ch + lk + 4 * q;
                sX[r_ + c_ * CH_LDP] = acc[q];
                la_st(ct + r_ + 32 * c_, acc[q]);
            }
        }
        __syncthreads(); // B1.5: c in sX
        if (prod) {
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int c = 4 * st + lk;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sX[16 * jhU + lr + c * CH_LDP], sX[16 * ihU + lr + c * CH_LDP], acc, 0, 0, 0);
            }
            const int w2 = min(32, a.m - 32 * I); // rows / columns >= w2 of the last diagonal tile are identity padding
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r_ = 16 * ihU + lr, c_ = 16 * jhU + lk + 4 * q;
                sD[r_ + c_ * CH_LDP] = (r_ >= w2 || c_ >= w2) ? ((r_ == c_) ? 1.0 : 0.0) : dacc[q] - acc[q];
            }
            la_stores_done(); // the write-through stores of c have had the product above to complete
        }
        __syncthreads(); // B2: D in sD; c written through
        if (tid == 0)
            la_raise(a, la_i_p(a, I, k));
        if (a.dbg && tid == 0)
            a.dbg[8 * k + 6] = wall_clock64();
        if (wave == 0) {
            const int w2 = min(32, a.m - 32 * I);
            double* lt = la_tile(a, la_i_linv(a, I));
            ldl_inverse_tile_put(
                sD, CH_LDP, w2,
                [sLk, lt](int r, int c, double v) {
                    sLk[r + c * CH_LDP] = v;
                    la_st(lt + r + 32 * c, v);
                },
                a.flags, swork);
            la_stores_done(); // the flag of L_(k+1)^-1 goes up right after the next barrier
        }
    }
    __syncthreads();
    if (tid == 0)
        la_raise(a, la_i_linv(a, NJ - 1));
    if (a.tr_steps && tid == 0 && NJ - 1 < 32)
        a.tr_steps[NJ - 1] = wall_clock64();
}

// ---- a block row --------------------------------------------------------------------------------------------------------------------
// RING (the instantiations for 17 .. 32 panels): the P^(p)_J operand tiles of a panel do not travel through registers two at a time (one
// memory round trip, ~2 us under load, per pair: with up to 16 tiles per wave that round trip, not the MFMA pipe, sets the pace) but through
// a double-buffered ring in LDS, LA_RC = 8 tiles per half (tile columns 8 c .. 8 c + 7), filled by direct-to-LDS loads (global_load_lds_dwordx4: wave w copies tile w of the
// chunk, no staging registers) while the previous chunk is multiplied. The destination of such a load is wave-uniform base + lane x 16 bytes,
// i.e. linear; the k / k+1 bank separation the operand reads need comes from the SOURCE address instead: LDS slot (r', c) of a tile holds
// element (r' ^ 16 (c & 1), c). Same products in the same order on the same accumulators: bit-identical to the register path.
constexpr int LA_RC = 8;                                               // tiles per ring half (wave w loads tile w of the chunk)
constexpr int LA_ROW_DOUBLES = 2 * 32 * CH_LDP + 32 + 256;             // with a ring: sLinv, sPI, sYv, sZp; sT lives in the ring's last tiles
                                                                       // (dead once P_I exists; the ring's second half is first written after that)
constexpr size_t LA_LDS_PLAIN = sizeof(double) * (4 * 32 * CH_LDP + LDL_SBUF); // the owner's need (and the rows' without a ring)
constexpr size_t LA_LDS_RING = sizeof(double) * (LA_ROW_DOUBLES + 2 * LA_RC * LA_TILE);
static_assert(LA_LDS_RING >= LA_LDS_PLAIN && LA_LDS_RING + 16 <= 160 * 1024, "LDS budget of a row workgroup with the operand ring");
typedef __attribute__((address_space(3))) void la_lds_ptr;

template <int MAXT, bool RING>
__device__ __forceinline__ void la_row(const LaArgs& a, const int I, double* smem, int* s_abort, const LaPoll& pl) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    double* sLinv = smem;
    double* sPI = smem + 32 * CH_LDP;
    double* sYv = smem + 2 * 32 * CH_LDP; // yTilde row of the panel (32)
    double* sZp = sYv + 32;               // z_p as 8 partial sums over 4 columns of L_p^-1 each ([8][32])
    double* ring = smem + LA_ROW_DOUBLES; // RING: 2 x LA_RC tiles
    double* sT = RING ? ring + 2 * LA_RC * LA_TILE - 32 * CH_LDP : sZp + 256; // the panel tile in operand layout
    const int NJ = a.NJ, m = a.m, rows = a.rows, ldz = a.ldz, seq = a.seq;
    const bool srow = I < NJ;
    const int row0 = srow ? 32 * I : m + 32 * (I - NJ);
    const int ilim = srow ? min(m, row0 + 32) : min(rows, row0 + 32);
    const int Jmax = srow ? I : NJ - 1;
    const bool ylast = (!srow) && (rows - 1 >= row0) && (rows - 1 < row0 + 32); // this block row holds the yTilde row
    const int yloc = rows - 1 - row0;
    const int g = wave >> 2, wq = wave & 3, ihU = wq & 1, jhU = wq >> 1;
    const int ri = row0 + 16 * ihU + lr;
    const int ric = min(ri, ilim - 1);
    double acc[MAXT][4];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int J = 2 * t + g;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = min(32 * J + 16 * jhU + lk + 4 * q, m - 1);
            acc[t][q] = (J <= Jmax) ? a.Z[ric + (size_t)j * ldz] : 0.0;
        }
    }
    if (ylast && tid < 32)
        la_put16(a.puby + 16 * (size_t)tid, a.Z[(rows - 1) + (size_t)min(tid, m - 1) * ldz], seq);
    double gsum = 0.0; // thread (r = tid & 31, h = tid >> 5 < 8): Gamma share of row row0 + r over the columns 4 h .. 4 h + 3 of every panel
    // S block rows: panels 0 .. I-3 with updates, then the hand-off of U1 / U0, then panel I-2 (b) without updates. T block rows: all panels.
    const int np = srow ? I - 1 : NJ;
    auto hand_off = [&]() {
        // to the owner: U1 = Z(I, I-1), U0 = Z(I, I) with the panels <= I-3 applied (all waves of the workgroup call this)
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int J = 2 * t + g;
            if (J == I - 1 || J == I) {
                double* u = la_tile(a, la_i_u(a, I, J == I ? 1 : 0));
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    la_st(u + (16 * ihU + lr) + 32 * (16 * jhU + lk + 4 * q), acc[t][q]);
            }
        }
        la_stores_done();
        __syncthreads();
        if (tid == 0)
            la_raise(a, la_i_u(a, I, 0));
    };
    for (int p = 0; p < np; ++p) {
        // laundered once per panel: otherwise the body's address / mask expressions are loop invariant, get hoisted and spilled
        int lrv = lr, lkv = lk;
        asm volatile("" : "+v"(lrv), "+v"(lkv));
        const bool do_update = srow ? (p <= I - 3) : true;
        if (srow && p == I - 2)
            hand_off();
        const int w = min(32, m - 32 * p);
        // (a) L_p^-1 -> LDS; the panel tile Z(I, p) -> LDS in operand layout (masked like the chain's operand loads); yTilde row of the panel
        la_wait(a.pubf + la_i_linv(a, p), 1, pl);
        {
            const double* lt = la_tile(a, la_i_linv(a, p));
            const double v0 = lt[tid], v1 = lt[tid + LA_T];
            sLinv[(tid & 31) + (tid >> 5) * CH_LDP] = v0;
            sLinv[((tid + LA_T) & 31) + ((tid + LA_T) >> 5) * CH_LDP] = v1;
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (2 * t + g == p) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 16 * jhU + lk + 4 * q;
                    sT[16 * ihU + lr + c * CH_LDP] = (ri < ilim && c < w) ? acc[t][q] : 0.0;
                }
            }
        if (!srow && wave == 7 && lane < 32) {
            const double yv = la_get16(a.puby + 512 * (size_t)p + 16 * (size_t)lane, pl);
            sYv[lane] = lane < w ? yv : 0.0;
        }
        __syncthreads();
        if (*s_abort)
            return;
        const bool dbg_row = a.dbg && tid == 0 && (I == NJ || I == NJ - 2) && p < 32;
        unsigned long long* dbr = a.dbg + 8 * ((I == NJ ? 32 : 64) + p);
        if (dbg_row)
            dbr[0] = wall_clock64();
        // (b) P_I = Z(I, p) L_p^-T on waves 0..3 (sub-tile (ih, ch)); z_p partials on waves 4..7
        if (wave < 4) {
            const int ih = wave & 1, ch = wave >> 1;
            d4 pacc = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 8; ++st)
                pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(sLinv[16 * ch + lr + (4 * st + lk) * CH_LDP], sT[16 * ih + lr + (4 * st + lk) * CH_LDP], pacc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sPI[16 * ih + lr + (16 * ch + lk + 4 * q) * CH_LDP] = pacc[q];
            if (srow) { // the factor rows leave for the other block rows straight from the accumulators
                double* pt = la_tile(a, la_i_p(a, I, p));
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    la_st(pt + (16 * ih + lr) + 32 * (16 * ch + lk + 4 * q), pacc[q]);
                la_stores_done();
            }
        } else if (!srow) {
            // z_p[c] = sum_q yTilde_p[q] L_p^-1[c][q] as 8 partial sums (thread (c, h): q = 4 h .. 4 h + 3), summed in a fixed order by the readers
            const int c = tid & 31, h = (tid >> 5) & 7;
            double z = 0.0;
#pragma unroll
            for (int q = 4 * h; q < 4 * h + 4; ++q)
                z = fma(sYv[q], sLinv[c + q * CH_LDP], z); // yTilde entries >= w are zero
            sZp[32 * h + c] = z;
        }
        __syncthreads();
        // (c) P_I: flag for the S block rows; final W rows (+ Gamma) for the T block rows
        if (srow) {
            if (tid == 0)
                la_raise(a, la_i_p(a, I, p));
        } else {
            if constexpr (!RING) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = tid + h * LA_T;
                    const int r = e & 31, c = e >> 5;
                    if (row0 + r < rows && c < w)
                        a.W[(row0 + r) + (size_t)(32 * p + c) * ldz] = sPI[r + c * CH_LDP];
                }
            }
            if (tid < 256) {
                const int r = tid & 31, h = tid >> 5;
#pragma unroll
                for (int c = 4 * h; c < 4 * h + 4; ++c) {
                    const double zc = ((sZp[c] + sZp[32 + c]) + (sZp[64 + c] + sZp[96 + c])) + ((sZp[128 + c] + sZp[160 + c]) + (sZp[192 + c] + sZp[224 + c]));
                    gsum = fma(sPI[r + c * CH_LDP], c < w ? zc : 0.0, gsum);
                }
            }
        }
        if (dbg_row)
            dbr[1] = wall_clock64();
        if (!do_update)
            continue;
        double aI[8];
#pragma unroll
        for (int st = 0; st < 8; ++st)
            aI[st] = sPI[16 * ihU + lr + (4 * st + lk) * CH_LDP];
        // every P^(p)_J this row needs: flags p NJ + (p+1 .. Jmax) are neighbours -> one polling load per wave
        {
            const int jn = min(Jmax, NJ - 1);
            const int cnt = (srow ? jn - 1 : jn) - p; // an S block row's own P_I (J = I) is in LDS
            if (cnt > 0)
                la_wait(a.pubf + la_i_p(a, p + 1, p), cnt, pl);
        }
        if constexpr (RING) {
            // chunk cc = the tiles J = 8 cc .. 8 cc + 7 (static: wave w loads tile 8 cc + w, group g multiplies t = 4 cc + u, J = 2 t + g), in ring half
            // cc & 1; the chunks that hold a tile in (p, Jmax] are walked in order. An S block row's own diagonal tile comes from sPI.
            // (Measured at N = 500: this two-stage ring 311 us per factorisation; four stages of four tiles with three chunks in flight 328 us:
            // the round trip is already hidden, the extra barriers are not free.)
            const int cfirst = (p + 1) >> 3, clast = Jmax >> 3;
            auto issue = [&](int cc) -> bool {
                const int J = 8 * cc + wave;
                if (J <= p || J > Jmax || (srow && J == I))
                    return false;
                const double* src = la_tile(a, la_i_p(a, J, p));
                double* dst = ring + (size_t)((cc & 1) * LA_RC + wave) * LA_TILE;
#pragma unroll
                for (int i = 0; i < 8; ++i) { // 1 KB per instruction: columns 4 i .. 4 i + 3, lane -> (r' = 2 lane & 31, c = 4 i + (lane >> 4))
                    const int cl = 4 * i + (lane >> 4), rp = (2 * lane) & 31;
                    __builtin_amdgcn_global_load_lds((const void*)(src + (rp ^ (16 * (cl & 1))) + 32 * cl), (la_lds_ptr*)(dst + 128 * i), 16, 0, 0);
                }
                return true;
            };
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing of this wave may be in flight besides the ring loads counted below
            issue(cfirst);
#pragma unroll
            for (int cc = 0; cc < MAXT / 4; ++cc) {
                if (cc < cfirst || cc > clast)
                    continue;
                const bool nxt = (cc + 1 <= clast) && issue(cc + 1);
                if (nxt)
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); // loads return in order: chunk cc has landed, chunk cc + 1 may still fly
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const double* half = ring + (size_t)((cc & 1) * LA_RC) * LA_TILE;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = 4 * cc + u, J = 2 * t + g;
                    if (J > p && J <= Jmax) {
                        double bj[8];
                        if (srow && J == I) {
#pragma unroll
                            for (int st = 0; st < 8; ++st)
                                bj[st] = sPI[16 * jhU + lrv + (4 * st + lkv) * CH_LDP];
                        } else {
                            const double* tl = half + (size_t)(2 * u + g) * LA_TILE + ((16 * jhU + lrv) ^ (16 * (lkv & 1))) + 32 * lkv;
#pragma unroll
                            for (int st = 0; st < 8; ++st)
                                bj[st] = tl[128 * st];
                        }
                        d4 d = {0, 0, 0, 0};
#pragma unroll
                        for (int st = 0; st < 8; ++st)
                            d = __builtin_amdgcn_mfma_f64_16x16x4f64(bj[st], aI[st], d, 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[t][q] -= d[q];
                        if (ylast && J == p + 1 && 16 * ihU + lrv == yloc) {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                la_put16(a.puby + 512 * (size_t)(p + 1) + 16 * (size_t)(16 * jhU + lkv + 4 * q), acc[t][q], seq);
                        }
                    }
                }
                if (cc + 2 <= clast) { // ring half cc & 1 is refilled by the next iteration's issue: every wave must be done reading it
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // sT of the next panel shares the ring's last tiles: every wave is done reading the ring
            // the W rows of this panel (final since step (b), still in sPI): stored here so that no store is in flight while the ring loads are counted
            if (!srow) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = tid + h * LA_T;
                    const int r = e & 31, cw = e >> 5;
                    if (row0 + r < rows && cw < w)
                        a.W[(row0 + r) + (size_t)(32 * p + cw) * ldz] = sPI[r + cw * CH_LDP];
                }
            }
        } else {
        // two tiles per round trip: both operand sets are requested before the first product needs one (the loads sit behind branches on
            // the runtime panel index, which the compiler does not hoist them over by itself)
    #pragma unroll
            for (int t0 = 0; t0 < MAXT; t0 += 2) {
                double bjs[2][8];
    #pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int J = 2 * (t0 + u) + g;
                    if (t0 + u < MAXT && J > p && J <= Jmax) {
                        if (J == I) { // diagonal tile of an S block row: both operands are P_I
    #pragma unroll
                            for (int st = 0; st < 8; ++st)
                                bjs[u][st] = sPI[16 * jhU + lrv + (4 * st + lkv) * CH_LDP];
                        } else
                            la_operand(la_tile(a, la_i_p(a, J, p)), jhU, bjs[u]);
                    }
                }
    #pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = (t0 + u < MAXT) ? t0 + u : MAXT - 1;
                    const int J = 2 * (t0 + u) + g;
                    if (t0 + u < MAXT && J > p && J <= Jmax) {
                        d4 d = {0, 0, 0, 0};
    #pragma unroll
                        for (int st = 0; st < 8; ++st)
                            d = __builtin_amdgcn_mfma_f64_16x16x4f64(bjs[u][st], aI[st], d, 0, 0, 0);
    #pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[t][q] -= d[q];
                        if (ylast && J == p + 1 && 16 * ihU + lrv == yloc) {
                            // the yTilde row of the next panel is final now: publish it for every T block row's z_(p+1)
    #pragma unroll
                            for (int q = 0; q < 4; ++q)
                                la_put16(a.puby + 512 * (size_t)(p + 1) + 16 * (size_t)(16 * jhU + lkv + 4 * q), acc[t][q], seq);
                        }
                    }
                }
            }
        }
        if (dbg_row)
            dbr[2] = wall_clock64();
    }
    if (srow && I == 1)
        hand_off(); // block row 1 has no panel of its own to wait for: its two tiles go to the owner as they are
    if (!srow) {
        __syncthreads(); // the last panel's readers of sZp are done
        if (tid < 256)
            sZp[tid] = gsum; // [h][r]
        __syncthreads();
        const int row = row0 + tid;
        if (tid < 32 && row >= m && row < rows - 1)
            la_st(a.gamma + (row - m), ((sZp[tid] + sZp[32 + tid]) + (sZp[64 + tid] + sZp[96 + tid])) + ((sZp[128 + tid] + sZp[160 + tid]) + (sZp[192 + tid] + sZp[224 + tid])));
    }
}

// The end of the frame's device work that the host waits for, run by the T block row that finishes last (every T block row has stored its
// Gamma rows write-through and counted itself in): X <- Delta X for the landmarks (k_lift's arithmetic: lift_load / lift_landmark), the
// estimates, Gamma's sensor rows and the status words into the pinned packet, then the doorbell. A failed factorisation (non-positive pivot,
// stalled wait) lifts nothing, like k_lift. One kernel launch and one kernel boundary less per frame than k_lift behind this kernel, measured
// neutral for the frame rate (the next frame's first kernel is bound by the host's launch): EQF_OPT_FUSED_LIFT, off by default.
__device__ __forceinline__ void la_finish(const LaArgs& a) {
    const int tid = threadIdx.x;
    if (a.tr_lift && tid == 0)
        a.tr_lift[0] = wall_clock64();
    auto ld = [](const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }; // past this XCD's L2: written by other workgroups
    const int f0 = __hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), f1 = a.flags[1];
    const int f3 = __hip_atomic_load(a.flags + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool failed = f0 != 0 || f3 != 0;
    for (int i = tid; i < max(a.lift_N, 21); i += LA_T) {
        const bool lm = i < a.lift_N;
        const int ic = lm ? i : 0;
        const double gs = ld(a.gamma + (i < 21 ? i : 0));
        const double g0 = ld(a.gamma + 21 + 3 * ic), g1 = ld(a.gamma + 21 + 3 * ic + 1), g2 = ld(a.gamma + 21 + 3 * ic + 2);
        const LiftIn in = lift_load(ic, a.lift_Ncap, a.lift_chart, a.lift_discrete, a.lift_q0, a.lift_Qq, a.lift_Qa);
        if (!failed) {
            if (i < 21)
                a.lift_gamma_host[i] = gs;
            if (lm)
                lift_landmark(i, V3{g0, g1, g2}, in, a.lift_N, a.lift_Ncap, a.lift_chart, a.lift_discrete, a.lift_Qq, a.lift_Qa, a.lift_est);
        }
    }
    if (tid == 0) {
        a.lift_flags_host[0] = f0;
        a.lift_flags_host[1] = f1;
        a.lift_flags_host[2] = 0;
        a.lift_flags_host[3] = f3;
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        __threadfence_system();
        *reinterpret_cast<volatile int*>(a.lift_door_host) = a.lift_door_seq;
        if (a.tr_lift)
            a.tr_lift[1] = wall_clock64();
    }
}

template <int MAXT, bool RING = false>
__global__ void __launch_bounds__(LA_T) k_chol_lookahead(const LaArgs a) {
    if (a.spec && *a.spec == a.spec_seq) { // cancelled speculative tail: say so, ring, done
        if (a.lift_door_host && blockIdx.x == 0 && threadIdx.x == 0) {
            a.lift_flags_host[2] = 1;
            a.lift_flags_host[3] = 0;
            __threadfence_system();
            *reinterpret_cast<volatile int*>(a.lift_door_host) = a.lift_door_seq;
        }
        return;
    }
    // static LDS without a ring (constant addresses: 2.6 us per factorisation at N = 200 against the same kernel on dynamic LDS), dynamic
    // (LA_LDS_RING bytes, above the 64 KB a static array may have) with one
    extern __shared__ double la_dyn_smem[];
    __shared__ double la_st_smem[RING ? 1 : 4 * 32 * CH_LDP + LDL_SBUF];
    double* smem = RING ? la_dyn_smem : la_st_smem;
    __shared__ int s_abort;
    if (threadIdx.x == 0)
        s_abort = 0;
    __syncthreads();
    const LaPoll pl{(long long)wall_clock64() + a.timeout_ticks, a.seq, &s_abort};
    if (blockIdx.x == 0)
        la_owner(a, smem, &s_abort, pl);
    else
        la_row<MAXT, RING>(a, (int)blockIdx.x, smem, &s_abort, pl);
    if (threadIdx.x == 0 && s_abort)
        __hip_atomic_store(a.flags + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.lift_door_host && (int)blockIdx.x >= a.NJ) { // a T block row (stalled or not) counts itself in; the last one finishes the frame
        __shared__ int s_last;
        la_stores_done();
        __syncthreads();
        if (threadIdx.x == 0) {
            const int nT = a.NI - a.NJ;
            const int seen = __hip_atomic_fetch_add(a.lift_done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (seen == nT - 1);
            if (s_last)
                __hip_atomic_store(a.lift_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_last)
            la_finish(a);
    }
}

} // namespace eqf
