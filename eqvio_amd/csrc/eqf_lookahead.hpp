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
//    They run one to two panels behind the owner; nothing they do is on the critical path as long as a hand-off takes < ~3 us.
//  * HAND-OFF without flags and without fences: every published double travels as one 16-byte (value, sequence, ~sequence) word written
//    by ONE global_store_dwordx4 sc1 (write-through to the agent coherence point) and read by ONE global_load_dwordx4 sc1; a consumer
//    polls exactly the words it needs, all loads of a tile in flight together, until every one carries the launch's sequence number.
//    A 16-byte aligned access never straddles a 32-byte sector, so value and sequence arrive together (scripts/ubench/pingpong2.hip:
//    1.0 us per 8 KB tile hop against 2.1 us with a separate flag and 2.7 us with release / acquire fences; 20 000 x 2 x 1024 words checked).
//    The sequence number changes with every launch: the buffers are never cleared.
//  * Gamma = W z is accumulated by the T block rows on the way (z_p = yTilde_p L_p^-T from the published yTilde row, one fma chain per
//    row over all columns), so the lift kernel finds Gamma complete.
//  * Every poll is bounded (20 ms of device wall clock); a timeout raises flags[3] (EQF_E_STALLED) and the workgroups drain.
//    Dependencies point from higher to lower block rows and to the owner only, and NI <= 80 workgroups of 64 KB LDS always fit the chip.
#pragma once
#include "eqf_kernels.hpp"

namespace eqf {

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int LA_T = 512;            // threads per workgroup (8 waves)
constexpr int LA_TILE_B = 16 * 1024; // bytes of a published 32 x 32 tile of 16-byte words
constexpr long long LA_TIMEOUT_TICKS = 2000000; // 20 ms at 100 MHz

struct LaArgs {
    int rows, m, ldz, NJ, NI, seq;
    const double* Z;     // [S ; T ; y^T] from k_build_Z (plain memory; complete when this kernel starts)
    double* W;           // out, plain: rows >= m receive W = T L^-T and the z row
    const double* Linv0; // L_0^-1, 32 x 32 column-major, from k_build_Z's first-tile elimination
    char* pub;           // published tiles (see the offsets below)
    double* gamma;       // out: Gamma[n]
    int* flags;          // [0] non-positive pivot, [3] stalled
    const int* spec;
    int spec_seq;
    trace_t* tr_steps;   // EQF_OPT_TRACE: slot of step 0 (the owner stamps one slot per step), or nullptr
    unsigned long long* dbg; // EQF_OPT_TRACE: per-step stamps inside the owner ([k][8]) and two block rows ([32 + p][8], [64 + p][8]), or nullptr
};
// published tiles: [0, NJ) L_p^-1 | [NJ, NJ + NJ^2) P^(p)_J at J NJ + p | then U1, U0 of every S block row | then the yTilde row per panel
__device__ __forceinline__ char* la_linv(const LaArgs& a, int p) { return a.pub + (size_t)LA_TILE_B * p; }
__device__ __forceinline__ char* la_p(const LaArgs& a, int J, int p) { return a.pub + (size_t)LA_TILE_B * (a.NJ + J * a.NJ + p); }
__device__ __forceinline__ char* la_u(const LaArgs& a, int I, int which) { return a.pub + (size_t)LA_TILE_B * (a.NJ + a.NJ * a.NJ + 2 * I + which); }
__device__ __forceinline__ char* la_y(const LaArgs& a, int p) { return a.pub + (size_t)LA_TILE_B * (a.NJ + a.NJ * a.NJ + 2 * a.NJ) + 512 * (size_t)p; }
inline size_t la_pub_bytes(int NJ) { return (size_t)LA_TILE_B * (NJ + (size_t)NJ * NJ + 2 * NJ) + 512 * (size_t)NJ; }

__device__ __forceinline__ void la_put(char* p, double v, int seq) {
    v4i x;
    x.x = __double2loint(v);
    x.y = __double2hiint(v);
    x.z = seq;
    x.w = ~seq;
    // s_nop: a VALU write of the data registers must not follow a store of more than 64 bits within one wait state; the compiler cannot see
    // that this asm is such a store
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
struct LaPoll {
    long long deadline;
    int seq;
    int* s_abort;
};
__device__ __forceinline__ bool la_ok(const v4i& r, int seq) { return r.z == seq && r.w == ~seq; }
__device__ __forceinline__ double la_val(const v4i& r) { return __hiloint2double(r.y, r.x); }
__device__ __forceinline__ bool la_retry(const LaPoll& pl) {
    if ((long long)wall_clock64() > pl.deadline) {
        *pl.s_abort = 1;
        return false;
    }
    __builtin_amdgcn_s_sleep(1);
    return true;
}
__device__ __forceinline__ double la_get1(const char* p0, const LaPoll& pl) {
    v4i r;
    for (;;) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p0) : "memory");
        if (la_ok(r, pl.seq) || !la_retry(pl))
            break;
    }
    return la_val(r);
}
// While a tile is not there yet, poll ONE of its words (every lane the same address: one request per wave) instead of re-requesting
// all of them: 30 workgroups spinning on whole tiles saturate the few memory channels a 16 KB tile lives in and slow down the very
// stores they wait for. The sentinel is a word the producer writes late; it is a hint only - every word is still validated by its own
// sequence number when the tile is fetched.
__device__ __forceinline__ void la_wait_word(const char* word, const LaPoll& pl) {
    v4i r;
    for (;;) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(word) : "memory");
        if (la_ok(r, pl.seq) || !la_retry(pl))
            break;
    }
}
constexpr int LA_SENT_TILE = 16 * 1023; // byte offset of the sentinel word of a published P / U tile (entry (31, 31): stored last)
constexpr int LA_SENT_LINV = 16 * 511;  // ... of L^-1 (entry (31, 15): the elimination's last stage)
__device__ __forceinline__ void la_get2(const char* p0, const char* p1, const LaPoll& pl, double& v0, double& v1) {
    v4i r0, r1;
    for (;;) {
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                     "global_load_dwordx4 %1, %3, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(r0), "=&v"(r1)
                     : "v"(p0), "v"(p1)
                     : "memory");
        if ((la_ok(r0, pl.seq) && la_ok(r1, pl.seq)) || !la_retry(pl))
            break;
    }
    v0 = la_val(r0);
    v1 = la_val(r1);
}
// accumulator layout: the four entries [i][j + 4 q] a lane holds of a 16 x 16 sub-tile; e0 = 16-byte index of entry q = 0, entries 4 columns = 128 words apart
__device__ __forceinline__ void la_get_acc(const char* tile, int e0, const LaPoll& pl, double (&v)[4]) {
    const char* p0 = tile + 16 * (size_t)e0;
    v4i r0, r1, r2, r3;
    for (;;) {
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                     "global_load_dwordx4 %1, %4, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %2, %5, off sc1\n\t"
                     "global_load_dwordx4 %3, %5, off offset:2048 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                     : "v"(p0), "v"(p0 + 4096)
                     : "memory");
        if ((la_ok(r0, pl.seq) && la_ok(r1, pl.seq) && la_ok(r2, pl.seq) && la_ok(r3, pl.seq)) || !la_retry(pl))
            break;
    }
    v[0] = la_val(r0);
    v[1] = la_val(r1);
    v[2] = la_val(r2);
    v[3] = la_val(r3);
}
// MFMA operand layout: lane (lr, lk) receives v[st] = X[16 h + lr][4 st + lk], st = 0..7, of a published tile X[row + 32 k]
__device__ __forceinline__ void la_get_operand(const char* tile, int h, const LaPoll& pl, double (&v)[8]) {
    const int lane = threadIdx.x & 63;
    const char* p0 = tile + 16 * (size_t)((16 * h + (lane & 15)) + 32 * (lane >> 4)); // + 2048 st
    v4i r0, r1, r2, r3, r4, r5, r6, r7;
    for (;;) {
        asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                     "global_load_dwordx4 %1, %8, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %2, %9, off sc1\n\t"
                     "global_load_dwordx4 %3, %9, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %4, %10, off sc1\n\t"
                     "global_load_dwordx4 %5, %10, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %6, %11, off sc1\n\t"
                     "global_load_dwordx4 %7, %11, off offset:2048 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                     : "v"(p0), "v"(p0 + 4096), "v"(p0 + 8192), "v"(p0 + 12288)
                     : "memory");
        const bool ok = la_ok(r0, pl.seq) && la_ok(r1, pl.seq) && la_ok(r2, pl.seq) && la_ok(r3, pl.seq) && la_ok(r4, pl.seq) && la_ok(r5, pl.seq) && la_ok(r6, pl.seq) &&
                        la_ok(r7, pl.seq);
        if (ok || !la_retry(pl))
            break;
    }
    v[0] = la_val(r0);
    v[1] = la_val(r1);
    v[2] = la_val(r2);
    v[3] = la_val(r3);
    v[4] = la_val(r4);
    v[5] = la_val(r5);
    v[6] = la_val(r6);
    v[7] = la_val(r7);
}

// two tiles at once (16 loads in flight, ONE round trip): the block rows' update loop and the owner's pre-work are bound by the number of
// dependent round trips, not by bytes
__device__ __forceinline__ void la_get_operand2(const char* tileA, const char* tileB, int h, const LaPoll& pl, double (&va)[8], double (&vb)[8]) {
    const int lane = threadIdx.x & 63;
    const size_t off = 16 * (size_t)((16 * h + (lane & 15)) + 32 * (lane >> 4));
    const char* pa = tileA + off;
    const char* pb = tileB + off;
    v4i r0, r1, r2, r3, r4, r5, r6, r7, s0, s1, s2, s3, s4, s5, s6, s7;
    for (;;) {
        asm volatile("global_load_dwordx4 %0, %16, off sc1\n\t"
                     "global_load_dwordx4 %1, %16, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %2, %17, off sc1\n\t"
                     "global_load_dwordx4 %3, %17, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %4, %18, off sc1\n\t"
                     "global_load_dwordx4 %5, %18, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %6, %19, off sc1\n\t"
                     "global_load_dwordx4 %7, %19, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %8, %20, off sc1\n\t"
                     "global_load_dwordx4 %9, %20, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %10, %21, off sc1\n\t"
                     "global_load_dwordx4 %11, %21, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %12, %22, off sc1\n\t"
                     "global_load_dwordx4 %13, %22, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %14, %23, off sc1\n\t"
                     "global_load_dwordx4 %15, %23, off offset:2048 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(s4), "=&v"(s5),
                       "=&v"(s6), "=&v"(s7)
                     : "v"(pa), "v"(pa + 4096), "v"(pa + 8192), "v"(pa + 12288), "v"(pb), "v"(pb + 4096), "v"(pb + 8192), "v"(pb + 12288)
                     : "memory");
        const int q = pl.seq;
        const bool ok = la_ok(r0, q) && la_ok(r1, q) && la_ok(r2, q) && la_ok(r3, q) && la_ok(r4, q) && la_ok(r5, q) && la_ok(r6, q) && la_ok(r7, q) && la_ok(s0, q) && la_ok(s1, q) &&
                        la_ok(s2, q) && la_ok(s3, q) && la_ok(s4, q) && la_ok(s5, q) && la_ok(s6, q) && la_ok(s7, q);
        if (ok || !la_retry(pl))
            break;
    }
    va[0] = la_val(r0), va[1] = la_val(r1), va[2] = la_val(r2), va[3] = la_val(r3), va[4] = la_val(r4), va[5] = la_val(r5), va[6] = la_val(r6), va[7] = la_val(r7);
    vb[0] = la_val(s0), vb[1] = la_val(s1), vb[2] = la_val(s2), vb[3] = la_val(s3), vb[4] = la_val(s4), vb[5] = la_val(s5), vb[6] = la_val(s6), vb[7] = la_val(s7);
}
__device__ __forceinline__ void la_get_acc2(const char* tileA, const char* tileB, int e0, const LaPoll& pl, double (&va)[4], double (&vb)[4]) {
    const char* pa = tileA + 16 * (size_t)e0;
    const char* pb = tileB + 16 * (size_t)e0;
    v4i r0, r1, r2, r3, s0, s1, s2, s3;
    for (;;) {
        asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                     "global_load_dwordx4 %1, %8, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %2, %9, off sc1\n\t"
                     "global_load_dwordx4 %3, %9, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %4, %10, off sc1\n\t"
                     "global_load_dwordx4 %5, %10, off offset:2048 sc1\n\t"
                     "global_load_dwordx4 %6, %11, off sc1\n\t"
                     "global_load_dwordx4 %7, %11, off offset:2048 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
                     : "v"(pa), "v"(pa + 4096), "v"(pb), "v"(pb + 4096)
                     : "memory");
        const int q = pl.seq;
        if ((la_ok(r0, q) && la_ok(r1, q) && la_ok(r2, q) && la_ok(r3, q) && la_ok(s0, q) && la_ok(s1, q) && la_ok(s2, q) && la_ok(s3, q)) || !la_retry(pl))
            break;
    }
    va[0] = la_val(r0), va[1] = la_val(r1), va[2] = la_val(r2), va[3] = la_val(r3);
    vb[0] = la_val(s0), vb[1] = la_val(s1), vb[2] = la_val(s2), vb[3] = la_val(s3);
}

// ---- the owner --------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void la_owner(const LaArgs& a, double* smem, int* s_abort, const LaPoll& pl) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    double* sLk = smem;                  // L_k^-1, operand layout [r + c CH_LDP]
    double* sX = smem + 32 * CH_LDP;     // c_k (read by the pre-work), then c_(k+1) (written by the post-work)
    double* sY = smem + 2 * 32 * CH_LDP; // R1 (written by the pre-work, read by the post-work)
    double* sD = smem + 3 * 32 * CH_LDP; // the diagonal tile handed to the elimination
    double* swork = smem + 4 * 32 * CH_LDP;
    const int NJ = a.NJ, seq = a.seq;
    for (int e = tid; e < 1024; e += LA_T) {
        const double v = a.Linv0[e];
        sLk[(e & 31) + (e >> 5) * CH_LDP] = v;
        la_put(la_linv(a, 0) + 16 * (size_t)e, v, seq);
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
            double u1[4], u0[4];
            const int e0 = (16 * ihU + lr) + 32 * (16 * jhU + lk);
            if (a.dbg && wave == 4 && lane == 0)
                a.dbg[8 * k + 0] = wall_clock64();
            la_wait_word(la_u(a, I, 1) + LA_SENT_TILE, pl);
            la_get_acc2(la_u(a, I, 0), la_u(a, I, 1), e0, pl, u1, u0);
            if (a.dbg && wave == 4 && lane == 0)
                a.dbg[8 * k + 1] = wall_clock64();
            if (k >= 1) {
                double bi[8], bj[8];
                la_wait_word(la_p(a, I, k - 1) + LA_SENT_TILE, pl);
                if (jhU == ihU) {
                    la_get_operand(la_p(a, I, k - 1), ihU, pl, bi);
#pragma unroll
                    for (int st = 0; st < 8; ++st)
                        bj[st] = bi[st];
                } else { // the two halves of the same tile: one round trip
                    const char* bt = la_p(a, I, k - 1);
                    la_get_operand2(bt + 16 * 16 * (size_t)ihU, bt + 16 * 16 * (size_t)jhU, 0, pl, bi, bj);
                }
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
        if (prod) {
            // c = P^(k)_I = R1 L_k^-T : sub-tile (ih, ch) = (pw & 1, pw >> 1)
            const int ih = pw & 1, ch = pw >> 1;
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 8; ++st)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sLk[16 * ch + lr + (4 * st + lk) * CH_LDP], sY[16 * ih + lr + (4 * st + lk) * CH_LDP], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r_ = 16 * ih + lr, c_ = 16 * ch + lk + 4 * q;
                sX[r_ + c_ * CH_LDP] = acc[q];
                la_put(la_p(a, I, k) + 16 * (size_t)(r_ + 32 * c_), acc[q], seq);
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
        }
        __syncthreads(); // B2: D in sD
        if (a.dbg && tid == 0)
            a.dbg[8 * k + 6] = wall_clock64();
        if (wave == 0) {
            const int w2 = min(32, a.m - 32 * I);
            char* lp = la_linv(a, I);
            ldl_inverse_tile_put(
                sD, CH_LDP, w2,
                [sLk, lp, seq](int r, int c, double v) {
                    sLk[r + c * CH_LDP] = v;
                    la_put(lp + 16 * (size_t)(r + 32 * c), v, seq);
                },
                a.flags, swork);
        }
    }
    if (a.tr_steps && tid == 0 && NJ - 1 < 32)
        a.tr_steps[NJ - 1] = wall_clock64();
}

// ---- a block row --------------------------------------------------------------------------------------------------------------------
template <int MAXT>
__device__ __forceinline__ void la_row(const LaArgs& a, const int I, double* smem, int* s_abort, const LaPoll& pl) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    double* sLinv = smem;
    double* sT = smem + 32 * CH_LDP;
    double* sPI = smem + 2 * 32 * CH_LDP;
    double* sYv = smem + 3 * 32 * CH_LDP; // yTilde row of the panel (32)
    double* sZp = sYv + 32;               // z_p as 8 partial sums over 4 columns of L_p^-1 each ([8][32])
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
        la_put(la_y(a, 0) + 16 * (size_t)tid, a.Z[(rows - 1) + (size_t)min(tid, m - 1) * ldz], seq);
    double gsum = 0.0; // thread (r = tid & 31, h = tid >> 5 < 8): Gamma share of row row0 + r over the columns 4 h .. 4 h + 3 of every panel
    // S block rows: panels 0 .. I-3 with updates, then the hand-off of U1 / U0, then panel I-2 (b) without updates. T block rows: all panels.
    const int np = srow ? I - 1 : NJ;
    for (int p = 0; p < np; ++p) {
        const bool do_update = srow ? (p <= I - 3) : true;
        if (srow && p == I - 2) {
            // hand-off to the owner: U1 = Z(I, I-1), U0 = Z(I, I) with the panels <= I-3 applied
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                const int J = 2 * t + g;
                if (J == I - 1 || J == I) {
                    char* u = la_u(a, I, J == I ? 1 : 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        la_put(u + 16 * (size_t)((16 * ihU + lr) + 32 * (16 * jhU + lk + 4 * q)), acc[t][q], seq);
                }
            }
        }
        const int w = min(32, m - 32 * p);
        // (a) L_p^-1 -> LDS; the panel tile Z(I, p) -> LDS in operand layout (masked like the chain's operand loads); yTilde row of the panel
        {
            double v0, v1;
            la_wait_word(la_linv(a, p) + (p == 0 ? 16 * 1023 : LA_SENT_LINV), pl);
            la_get2(la_linv(a, p) + 16 * (size_t)tid, la_linv(a, p) + 16 * (size_t)(tid + LA_T), pl, v0, v1);
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
            const double yv = la_get1(la_y(a, p) + 16 * (size_t)lane, pl);
            sYv[lane] = lane < w ? yv : 0.0;
        }
        __syncthreads();
        if (*s_abort)
            return;
        const bool dbg_row = a.dbg && tid == 0 && (I == NJ || I == NJ - 2) && p < 32;
        unsigned long long* dbr = a.dbg + 8 * ((I == NJ ? 32 : 64) + p);
        if (dbg_row)
            dbr[0] = wall_clock64();
        // (b) P_I = Z(I, p) L_p^-T on waves 0..3 (sub-tile (ih, ch)); z_p on wave 4
        if (wave < 4) {
            const int ih = wave & 1, ch = wave >> 1;
            d4 pacc = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 8; ++st)
                pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(sLinv[16 * ch + lr + (4 * st + lk) * CH_LDP], sT[16 * ih + lr + (4 * st + lk) * CH_LDP], pacc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sPI[16 * ih + lr + (16 * ch + lk + 4 * q) * CH_LDP] = pacc[q];
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
        // (c) P_I leaves: published for the S block rows (the factor), stored as final W rows for the T block rows (+ Gamma)
        if (srow) {
            char* pp = la_p(a, I, p);
            la_put(pp + 16 * (size_t)tid, sPI[(tid & 31) + (tid >> 5) * CH_LDP], seq);
            la_put(pp + 16 * (size_t)(tid + LA_T), sPI[((tid + LA_T) & 31) + ((tid + LA_T) >> 5) * CH_LDP], seq);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = tid + h * LA_T;
                const int r = e & 31, c = e >> 5;
                if (row0 + r < rows && c < w)
                    a.W[(row0 + r) + (size_t)(32 * p + c) * ldz] = sPI[r + c * CH_LDP];
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
        // two tiles per round trip: the operands P^(p)_J of both are requested together
#pragma unroll
        for (int tt = 0; tt < MAXT; tt += 2) {
            const int JA = 2 * tt + g, JB = 2 * (tt + 1) + g;
            const bool actA = JA > p && JA <= Jmax;
            const bool actB = (tt + 1 < MAXT) && JB > p && JB <= Jmax;
            if (!actA && !actB)
                continue;
            const bool pollA = actA && JA != I, pollB = actB && JB != I; // the diagonal tile of an S block row takes both operands from P_I
            double bA[8], bB[8];
            if (pollB)
                la_wait_word(la_p(a, JB, p) + LA_SENT_TILE, pl);
            else if (pollA)
                la_wait_word(la_p(a, JA, p) + LA_SENT_TILE, pl);
            if (pollA && pollB)
                la_get_operand2(la_p(a, JA, p), la_p(a, JB, p), jhU, pl, bA, bB);
            else if (pollA)
                la_get_operand(la_p(a, JA, p), jhU, pl, bA);
            else if (pollB)
                la_get_operand(la_p(a, JB, p), jhU, pl, bB);
            if (actA && !pollA) {
#pragma unroll
                for (int st = 0; st < 8; ++st)
                    bA[st] = sPI[16 * jhU + lr + (4 * st + lk) * CH_LDP];
            }
            if (actB && !pollB) {
#pragma unroll
                for (int st = 0; st < 8; ++st)
                    bB[st] = sPI[16 * jhU + lr + (4 * st + lk) * CH_LDP];
            }
            if (actA) {
                d4 d = {0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 8; ++st)
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(bA[st], aI[st], d, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[tt][q] -= d[q];
                if (ylast && JA == p + 1 && 16 * ihU + lr == yloc) {
                    // the yTilde row of the next panel is final now: publish it for every T block row's z_(p+1)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        la_put(la_y(a, p + 1) + 16 * (size_t)(16 * jhU + lk + 4 * q), acc[tt][q], seq);
                }
            }
            if (actB) {
                d4 d = {0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 8; ++st)
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(bB[st], aI[st], d, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[(tt + 1 < MAXT) ? tt + 1 : tt][q] -= d[q];
                if (ylast && JB == p + 1 && 16 * ihU + lr == yloc) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        la_put(la_y(a, p + 1) + 16 * (size_t)(16 * jhU + lk + 4 * q), acc[(tt + 1 < MAXT) ? tt + 1 : tt][q], seq);
                }
            }
        }
        if (dbg_row)
            dbr[2] = wall_clock64();
    }
    if (srow && I == 1) {
        // block row 1 has no panel of its own to wait for: its two tiles go to the owner as they are
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int J = 2 * t + g;
            if (J == 0 || J == 1) {
                char* u = la_u(a, 1, J == 1 ? 1 : 0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    la_put(u + 16 * (size_t)((16 * ihU + lr) + 32 * (16 * jhU + lk + 4 * q)), acc[t][q], seq);
            }
        }
    }
    if (!srow) {
        __syncthreads(); // the last panel's readers of sZp are done
        if (tid < 256)
            sZp[tid] = gsum; // [h][r]
        __syncthreads();
        const int row = row0 + tid;
        if (tid < 32 && row >= m && row < rows - 1)
            a.gamma[row - m] = ((sZp[tid] + sZp[32 + tid]) + (sZp[64 + tid] + sZp[96 + tid])) + ((sZp[128 + tid] + sZp[160 + tid]) + (sZp[192 + tid] + sZp[224 + tid]));
    }
}

template <int MAXT>
__global__ void __launch_bounds__(LA_T) k_chol_lookahead(const LaArgs a) {
    if (a.spec && *a.spec == a.spec_seq)
        return; // cancelled speculative tail
    __shared__ double smem[4 * 32 * CH_LDP + LDL_SBUF];
    __shared__ int s_abort;
    if (threadIdx.x == 0)
        s_abort = 0;
    __syncthreads();
    const LaPoll pl{(long long)wall_clock64() + LA_TIMEOUT_TICKS, a.seq, &s_abort};
    if (blockIdx.x == 0)
        la_owner(a, smem, &s_abort, pl);
    else
        la_row<MAXT>(a, (int)blockIdx.x, smem, &s_abort, pl);
    if (threadIdx.x == 0 && s_abort)
        a.flags[3] = 1;
}

} // namespace eqf
