// HIP kernels of the MI355X EqF path (gfx950 only). Included by eqf_hip.hip.
//
// Data layout in HBM (all fp64):
//   Sigma      : n x n column-major, leading dimension ld (fixed by capacity), two buffers (ping-pong:
//                propagate and landmark compaction are out-of-place, the vision update is in-place)
//   q0, Qq, Qa : per-landmark origin point, SOT(3) quaternion and scale as SoA planes of stride Ncap
//                (q0: 3 planes, Qq: 4 planes (w,x,y,z), Qa: 1 plane) -> lane i reads plane[i]: coalesced
//   Al, Bl     : packed per-landmark rows of the EqF state/input matrices, SoA planes of stride Ncap:
//                Al 45 planes = 3 rows x 15 cols (cols 0:3 | 12:15 | 15:21 of A, then the own 3x3 block),
//                Bl 9 planes  = 3 x 3 (cols 0:3 of B)                      [SURVEY.md §8(a) a3, a4]
//   Z          : (m + n + 1) x m column-major, ld = ldz: rows [0,m) = S, [m,m+n) = T = Sigma C^T,
//                row m+n = yTilde^T. Right-looking blocked Cholesky of the S part applied to all rows
//                leaves [L ; W = T L^-T ; z^T = (L^-1 yTilde)^T].  Then Gamma = W z, Sigma -= W W^T.
#pragma once
#include "eqf_math.hpp"
#include "eqvio_types.h"
#include <hip/hip_runtime.h>

namespace eqf {

// Sensor-level terms of the EqF matrices, computed once per call on the host (O(1) work) and staged to HBM.
struct Common {
    double Mv[9];     // R_IC^T R_A^T                     (euclid.cpp:134-139)
    double RTic[9];   // R_IC^T                            (euclid.cpp:222-230)
    double RTicSx[9]; // R_IC^T skew(x_IC)
    double CT[36];    // Ad_{B^-1} ad(Ad_{T0^-1} Ad_A U_I) (euclid.cpp:141-148)
    double vC[3];     // linear part of Ad_{T_IC^-1} U_I   (euclid.cpp:150-152)
    double Ass[441];  // sensor block of A, row-major 21x21
    double Bs[252];   // sensor rows of B, row-major 21x12
};
// The same terms in the compact form that travels as a KERNEL ARGUMENT (1.1 KB): no staging copy, no PCIe read.
// Workgroup 0 of k_assemble_AB expands it into the Common record in HBM for the propagate kernels.
struct CommonK {
    double lm[66];   // Mv, RTic, RTicSx, CT, vC (the per-landmark factors, in Common's order)
    double RA[9];    // R_A
    double SxRA[9];  // skew(x_A) R_A
    double RAsv[9];  // R_A skew(v_hat)
    double G[9];     // -g skew(R_0^T e3)
    double adT[36];  // ad(Ad_{T0^-1} Ad_A U_I)
};
constexpr int kObsChunk = 20; // observer steps per launch (kernel-argument budget: they share 4 KB with the CommonK terms)
struct RiccatiArgs {
    double dt;
    double Qd[12];
    double Pd[8];
};
// One observer step (integrateObserverState) as the landmark kernel needs it.
struct ObsStep {
    int discrete;
    double dt;
    Pose Tinv; // T_IC^-1 Lambda.A^-1 T_IC   (VIOGroup.cpp:254)
    V3 omC, vC; // U_C = Ad_{T_IC^-1} U_A    (VIOGroup.cpp:207-216), continuous lift only
};

__device__ __forceinline__ V3 ld3(const double* base, int stride, int i) { return V3{base[i], base[stride + i], base[2 * stride + i]}; }
__device__ __forceinline__ Qt ldq(const double* base, int stride, int i) { return Qt{base[i], base[stride + i], base[2 * stride + i], base[3 * stride + i]}; }
__device__ __forceinline__ M3 ldm3(const double* p) { return M3{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]}; }
__device__ __forceinline__ void st_plane9(double* base, int stride, int i, int plane0, const M3& A) {
    base[(plane0 + 0) * stride + i] = A.a00;
    base[(plane0 + 1) * stride + i] = A.a01;
    base[(plane0 + 2) * stride + i] = A.a02;
    base[(plane0 + 3) * stride + i] = A.a10;
    base[(plane0 + 4) * stride + i] = A.a11;
    base[(plane0 + 5) * stride + i] = A.a12;
    base[(plane0 + 6) * stride + i] = A.a20;
    base[(plane0 + 7) * stride + i] = A.a21;
    base[(plane0 + 8) * stride + i] = A.a22;
}

// EQF_OPT_TRACE: the first thread of a launch stamps the device wall clock (100 MHz) into its slot of a per-frame ring; kernels
// whose end matters keep the latest finishing time of their workgroups next to it. tr == nullptr (the default) costs one
// uniform branch.
typedef unsigned long long trace_t;
__device__ __forceinline__ void trace_start(trace_t* tr) {
    if (tr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        *tr = wall_clock64();
}
__device__ __forceinline__ void trace_end(trace_t* tr) {
    if (tr && threadIdx.x == 0)
        atomicMax(tr + 1, (trace_t)wall_clock64());
}

// full-precision reciprocal from v_rcp_f64 + two Newton steps (shorter dependent chain than an IEEE division)
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
}

// full-precision 1/sqrt from v_rsq_f64 + two Newton steps (an IEEE sqrt followed by a division is ~10x longer a chain)
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}
// Chart constants of a landmark's origin point q0_i (InvDepth chart): conv_euc2ind, conv_ind2euc, ind2euc_r0 (invdepth.cpp:65-81,
// 201-207) cost a dozen square roots / divisions and a rotation from two vectors, and q0_i changes only when a landmark is added.
// They are stored once (k_scatter_landmarks) as 27 more SoA planes behind the 3 planes of q0 (the static half of the landmark
// arrays), so every kernel that has the q0 pointer finds them at a fixed offset.
constexpr int CC_OFF = 3, CC_E2I = 0, CC_I2E = 9, CC_R0 = 18, CC_PLANES = 27;
__device__ __forceinline__ M3 ld_cc(const double* __restrict__ q0, int Ncap, int i, int which) {
    const double* b = q0 + (size_t)(CC_OFF + which) * Ncap + i;
    return M3{b[0], b[Ncap], b[2 * (size_t)Ncap], b[3 * (size_t)Ncap], b[4 * (size_t)Ncap], b[5 * (size_t)Ncap], b[6 * (size_t)Ncap], b[7 * (size_t)Ncap], b[8 * (size_t)Ncap]};
}
// The chart constants of a new origin point, computed and stored by ONE function body (noinline): they are computed in four kernels (scatter, the two append / reshape
// passes, the propagation kernel's lanes for held landmarks), and inlined into different surroundings the compiler contracts their multiply-adds differently - the last
// bit of a constant differed for 5 of 200 landmarks between two of them (round 5), and with it every later propagation of those landmarks. r0: also handed back.
__device__ __attribute__((noinline)) void store_chart_constants(double* __restrict__ cc, int Ncap, int i, double px, double py, double pz, M3* r0) {
    const V3 p{px, py, pz};
    st_plane9(cc, Ncap, i, CC_E2I, conv_euc2ind(p));
    st_plane9(cc, Ncap, i, CC_I2E, conv_ind2euc(p));
    const M3 R = ind2euc_r0(p);
    st_plane9(cc, Ncap, i, CC_R0, R);
    if (r0)
        *r0 = R;
}

// column index in A of packed column e (0..11) of the landmark-sensor block
__host__ __device__ __forceinline__ int al_col(int e) { return e < 3 ? e : (e < 6 ? 12 + (e - 3) : 15 + (e - 6)); }

// ---------------------------------------------------------------------------------------------------
// K1: per-landmark rows of A and B (EqFStateMatrixA / EqFInputMatrixB, euclid.cpp:99-233, invdepth.cpp:36-181)
// One lane per landmark; the sensor-level terms are staged in LDS once per workgroup.
// entries of the sensor blocks B_s (21 x 12, row-major) and A_ss (21 x 21) from the compact terms (block layout of
// euclid.cpp:103-109, 186-233)
__device__ __forceinline__ double sensor_Bs_entry(const CommonK& ck, int t) {
    const int r = t / 12, c = t % 12;
    double v = 0.0;
    if (r < 6)
        v = (c == 6 + r) ? 1.0 : 0.0;
    else if (r < 9 && c < 3)
        v = ck.RA[(r - 6) * 3 + c];
    else if (r >= 9 && r < 12 && c < 3)
        v = ck.SxRA[(r - 9) * 3 + c];
    else if (r >= 12 && r < 15 && c < 3)
        v = ck.RAsv[(r - 12) * 3 + c];
    else if (r >= 12 && r < 15 && c >= 3 && c < 6)
        v = ck.RA[(r - 12) * 3 + (c - 3)];
    return v;
}
__device__ __forceinline__ double sensor_Ass_entry(const CommonK& ck, int t) {
    const int r = t / 21, c = t % 21;
    double v = 0.0;
    if (c < 6) { // -B[:, 0:6]
        if (r >= 6 && r < 9 && c < 3)
            v = -ck.RA[(r - 6) * 3 + c];
        else if (r >= 9 && r < 12 && c < 3)
            v = -ck.SxRA[(r - 9) * 3 + c];
        else if (r >= 12 && r < 15 && c < 3)
            v = -ck.RAsv[(r - 12) * 3 + c];
        else if (r >= 12 && r < 15 && c >= 3)
            v = -ck.RA[(r - 12) * 3 + (c - 3)];
    } else if (r >= 9 && r < 12 && c == r + 3) {
        v = 1.0;
    } else if (r >= 12 && r < 15 && c >= 6 && c < 9) {
        v = ck.G[(r - 12) * 3 + (c - 6)];
    } else if (r >= 15 && c >= 15) {
        v = ck.adT[(r - 15) * 6 + (c - 15)];
    }
    return v;
}
// The rows of A and B of ONE landmark: al[r * 15 + c] (packed columns: 0:3 | 12:15 | 15:21 | own 3x3) and bl[9].
// s_cm: the 66 per-landmark factors (Mv, RTic, RTicSx, CT, vC) in LDS.
// PART = -1: everything. PART = 0 / 1 / 2: only the columns 0:6 of the row block and bl / the columns 6:12 / the own 3x3 block
// (columns 12:15): the three parts share a short prefix (R_Q, Q_hat, q_hat) and are otherwise independent, so three lanes in
// three different wavefronts assemble one landmark in a third of the time (the unused results are dead code in each instance).
template <int PART>
__device__ __forceinline__ void assemble_landmark(const double* __restrict__ s_cm, int chart, const V3 p0, const Qt q, const double a, const M3& e2i, const M3& i2e,
                                                  double (&al)[45], double (&bl)[9]) {
    constexpr bool P0 = PART < 0 || PART == 0, P1 = PART < 0 || PART == 1, P2 = PART < 0 || PART == 2;
    const M3 Mv = ldm3(s_cm), RTic = ldm3(s_cm + 9), RTicSx = ldm3(s_cm + 18);
    const double* CT = s_cm + 27;
    const V3 vC{s_cm[63], s_cm[64], s_cm[65]};
    const M3 RQ = q_mat(q);
    const M3 Qhat = a * RQ;
    const V3 qh = (1.0 / a) * (transpose(RQ) * p0); // Q^-1 * q0
    const bool ind = chart == EQVIO_COORD_INVDEPTH; // e2i = conv_euc2ind(p0), i2e = conv_ind2euc(p0): stored chart constants
    if (P0) {
        M3 Bblk = Qhat * (skew(qh) * RTic + RTicSx);
        M3 A_v = (-1.0) * (Qhat * Mv);
        if (ind) {
            Bblk = e2i * Bblk;
            A_v = e2i * A_v;
        }
        const M3 A_b = (-1.0) * Bblk;
        const double ab[9] = {A_b.a00, A_b.a01, A_b.a02, A_b.a10, A_b.a11, A_b.a12, A_b.a20, A_b.a21, A_b.a22};
        const double av[9] = {A_v.a00, A_v.a01, A_v.a02, A_v.a10, A_v.a11, A_v.a12, A_v.a20, A_v.a21, A_v.a22};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                al[r * 15 + c] = ab[r * 3 + c];
                al[r * 15 + 3 + c] = av[r * 3 + c];
            }
        bl[0] = Bblk.a00, bl[1] = Bblk.a01, bl[2] = Bblk.a02, bl[3] = Bblk.a10, bl[4] = Bblk.a11, bl[5] = Bblk.a12, bl[6] = Bblk.a20, bl[7] = Bblk.a21, bl[8] = Bblk.a22;
    }
    if (P1) {
        // [skew(q0) R_Q, -a R_Q] * CT  (3x6 * 6x6)
        const M3 T0 = skew(p0) * RQ;
        const M3 T1 = (-a) * RQ;
        const double t[3][6] = {{T0.a00, T0.a01, T0.a02, T1.a00, T1.a01, T1.a02},
                                {T0.a10, T0.a11, T0.a12, T1.a10, T1.a11, T1.a12},
                                {T0.a20, T0.a21, T0.a22, T1.a20, T1.a21, T1.a22}};
        double Ac[3][6];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    s += t[r][k] * CT[k * 6 + c];
                Ac[r][c] = s;
            }
        if (ind) {
            const double e[3][3] = {{e2i.a00, e2i.a01, e2i.a02}, {e2i.a10, e2i.a11, e2i.a12}, {e2i.a20, e2i.a21, e2i.a22}};
            double Ac2[3][6];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    Ac2[r][c] = e[r][0] * Ac[0][c] + e[r][1] * Ac[1][c] + e[r][2] * Ac[2][c];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    Ac[r][c] = Ac2[r][c];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                al[r * 15 + 6 + c] = Ac[r][c];
    }
    if (P2) {
        const M3 inner = skew(qh) * skew(vC) - 2.0 * outer(vC, qh) + outer(qh, vC);
        const M3 QhatInv = (1.0 / a) * transpose(RQ);
        M3 A_q = (-1.0 / norm2(qh)) * (Qhat * inner * QhatInv);
        if (ind)
            A_q = e2i * A_q * i2e;
        const double aq[9] = {A_q.a00, A_q.a01, A_q.a02, A_q.a10, A_q.a11, A_q.a12, A_q.a20, A_q.a21, A_q.a22};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                al[r * 15 + 12 + c] = aq[r * 3 + c];
    }
}
__global__ void __launch_bounds__(64) k_assemble_AB(const CommonK ck, int N, int Ncap, int chart, Common* __restrict__ cmdev,
                                                    const double* __restrict__ q0, const double* __restrict__ Qq, const double* __restrict__ Qa,
                                                    double* __restrict__ Al, double* __restrict__ Bl, trace_t* tr) {
    trace_start(tr);
    // The sensor-level terms arrive as a kernel argument. The grid expands the sensor blocks A_ss (21x21) and
    // B_s (21x12) into HBM for the propagate kernels.
    __shared__ double s_cm[9 + 9 + 9 + 36 + 3];
    for (int t = threadIdx.x; t < 66; t += blockDim.x)
        s_cm[t] = ck.lm[t];
    {
        const int gt = blockIdx.x * blockDim.x + threadIdx.x, gs = gridDim.x * blockDim.x;
        for (int t = gt; t < 252; t += gs)
            cmdev->Bs[t] = sensor_Bs_entry(ck, t);
        for (int t = gt; t < 441; t += gs)
            cmdev->Ass[t] = sensor_Ass_entry(ck, t);
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    double al[45], bl[9];
    const bool ind = chart == EQVIO_COORD_INVDEPTH;
    assemble_landmark<-1>(s_cm, chart, ld3(q0, Ncap, i), ldq(Qq, Ncap, i), Qa[i], ind ? ld_cc(q0, Ncap, i, CC_E2I) : M3{}, ind ? ld_cc(q0, Ncap, i, CC_I2E) : M3{}, al, bl);
#pragma unroll
    for (int e = 0; e < 45; ++e)
        Al[e * Ncap + i] = al[e];
#pragma unroll
    for (int e = 0; e < 9; ++e)
        Bl[e * Ncap + i] = bl[e];
}

// ---------------------------------------------------------------------------------------------------
// K4: landmark part of k observer steps X <- X * Lambda (VIO_eqf.cpp:47-60, VIOGroup.cpp:190-271, 71-92)
struct ObsSteps {
    ObsStep s[kObsChunk];
};
// landmark i: Q_i <- Q_i * Lambda_1.Q_i * ... * Lambda_k.Q_i for the k steps (one lane per landmark)
// The k steps of a landmark are a serial chain, and with 10-25 IMU samples per frame that chain is as long as the whole Sigma
// propagation it rides along with. The discrete lift is therefore written for a short dependent chain: one reciprocal square
// root gives 1 / (|p1| |q_hat|) for both normalisations and the scale ratio, a second one the half-angle factor; the
// re-normalisation of the unit quaternion uses 1/sqrt(x) = 1.5 - 0.5 x (exact to O((x-1)^2), |x - 1| < 1e-9 here); 1/a is carried
// along instead of divided out each step. Algebraically this is SO3::SO3FromVectors(p1.normalized(), q_hat.normalized()) and
// |q_hat| / |p1| of VIOGroup.cpp:254-262; the antiparallel special case keeps the general routine.
// the chain itself, on values: (q, a) <- (q, a) * Lambda_1 * ... * Lambda_k for the landmark with origin point p0
__device__ __forceinline__ void observer_chain(const ObsStep* __restrict__ steps, int k, const V3 p0, Qt& q, double& a) {
    double inva = 1.0 / a;
    ObsStep nxt = steps[0];
    for (int s = 0; s < k; ++s) {
        const ObsStep st = nxt;
        nxt = steps[min(s + 1, k - 1)]; // the next step's terms (scalar loads from the argument segment) arrive during this step
        const V3 ph = inva * q_rot(q_inv(q), p0); // current estimate q_hat_i
        Qt Lq;
        double La, invLa;
        if (st.discrete) {
            const V3 p1 = pose_act(st.Tinv, ph);
            const double nh2 = norm2(ph), n12 = norm2(p1);
            const double r = fast_rsqrt(nh2 * n12);
            const double c = dot(p1, ph) * r;
            if (c < -1.0 + 1e-12) {
                Lq = so3_from_vectors(normalized(p1), normalized(ph));
            } else {
                const V3 ax = r * cross(p1, ph);
                const double t = (1.0 + c) * 2.0;
                const double is = fast_rsqrt(t);
                const Qt L{0.5 * (t * is), ax.x * is, ax.y * is, ax.z * is};
                const double kk = 1.5 - 0.5 * (L.w * L.w + L.x * L.x + L.y * L.y + L.z * L.z);
                Lq = Qt{L.w * kk, L.x * kk, L.y * kk, L.z * kk};
            }
            La = nh2 * r;
            invLa = n12 * r;
        } else {
            const double ip2 = 1.0 / norm2(ph);
            const V3 Wr = st.omC + ip2 * cross(ph, st.vC);
            const double Ws = ip2 * dot(ph, st.vC);
            Lq = so3_exp(st.dt * Wr);
            La = exp(st.dt * Ws);
            invLa = 1.0 / La;
        }
        if (st.discrete) {
            // product of two quaternions that are unit to a few ulp: the re-normalisation of q_mul without its sqrt and division
            const Qt m{q.w * Lq.w - q.x * Lq.x - q.y * Lq.y - q.z * Lq.z, q.w * Lq.x + q.x * Lq.w + q.y * Lq.z - q.z * Lq.y,
                       q.w * Lq.y + q.y * Lq.w + q.z * Lq.x - q.x * Lq.z, q.w * Lq.z + q.z * Lq.w + q.x * Lq.y - q.y * Lq.x};
            const double km = 1.5 - 0.5 * (m.w * m.w + m.x * m.x + m.y * m.y + m.z * m.z);
            q = Qt{m.w * km, m.x * km, m.y * km, m.z * km};
        } else {
            q = q_mul(q, Lq);
        }
        a = a * La;
        inva = inva * invLa;
    }
}
__device__ __forceinline__ void observer_landmark(const ObsStep* __restrict__ steps, int Ncap, int k, int i, const double* __restrict__ q0, const double* __restrict__ QqIn,
                                                  const double* __restrict__ QaIn, double* __restrict__ Qq, double* __restrict__ Qa) {
    // QqIn / QaIn == Qq / Qa: in place; otherwise the result goes to the other landmark buffer (fused assembly: the tiles of the
    // same launch still read the old Q)
    const V3 p0 = ld3(q0, Ncap, i);
    Qt q = ldq(QqIn, Ncap, i);
    double a = QaIn[i];
    observer_chain(steps, k, p0, q, a);
    Qq[i] = q.w;
    Qq[Ncap + i] = q.x;
    Qq[2 * Ncap + i] = q.y;
    Qq[3 * Ncap + i] = q.z;
    Qa[i] = a;
}
__global__ void __launch_bounds__(64) k_observer(const ObsSteps steps_arg, int N, int Ncap, int k, const double* __restrict__ q0,
                                                 double* __restrict__ Qq, double* __restrict__ Qa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N)
        observer_landmark(steps_arg.s, Ncap, k, i, q0, Qq, Qa, Qq, Qa);
}

// ---------------------------------------------------------------------------------------------------
// K2: G_i = (F Sigma)[l_i, 0:21] = dt*A_ls_i Sigma_ss + (I + dt*A_qi) Sigma_{l_i,s}   (3 x 21 per landmark) is computed
// inside the workgroups that need it (the tile's 16 i-landmarks; a strip workgroup's 12 landmarks): it costs the same 63
// loads per landmark a stored G would, and saves a launch.
// K2b: Sigma' = F Sigma F^T + dt (B Q B^T + P) in arrow form (integrateRiccatiStateFast, VIO_eqf.cpp:62-72).
// Block roles by blockIdx.x: [0, nT (nT + 1) / 2) lower landmark-landmark tiles of 16x16 landmarks (one 3x3 block per lane, mirrored into the upper triangle),
// then strip blocks (landmark-sensor 3x21 blocks and their transposes), then one sensor-sensor block.
constexpr int PT = 16; // landmarks per tile side
constexpr int PROP_T = 3 * PT * PT; // threads per workgroup of k_propagate_main: one per (row of a 3x3 block, landmark pair)
// Fused assembly (eqf_propagate_fast): no k_assemble_AB launch in front of the propagation. Every workgroup evaluates the rows
// of A and B of the landmarks it needs itself (one lane per landmark, results straight into LDS), the sensor blocks come from
// the compact terms in the argument segment.
struct FuseArgs {
    int on, chart;
    double *Qqo, *Qao; // the OTHER dynamic landmark buffer: target of the observer blocks
    CommonK ck;
};
// EQF_OPT_MEASURE_IN_PROPAGATE: the observer blocks of the propagation kernel have a landmark's new group element in registers when their chain ends - they evaluate its
// output block C*_i and residual right there (measure_one: the same function and inputs as everywhere, the same bits) and leave C / yTilde / the index map in
// memory, so that the update's first kernel (the look-ahead kernel building Z itself, ZB = 3) has no evaluation in front of its first tile (~2.5 us)
struct MeasEval {
    int on, star, Mcap;
    Cam cam;
    const double* ylm; // the measurement by landmark, pinned host packet of eqf_stage_measurement: planes u, v, measurement index or -1
    double *C, *ytil;
    int* lmidx_dev;
};
// Round 5: landmarks that left the state since the last kernel (lost features, discarded outliers: a record of removals only) leave INSIDE the propagation kernel - it
// reads Sigma and the landmark planes at their old positions and writes the other buffers at the new ones anyway - instead of in a compaction pass of its own in front of
// it (k_reshape_args: one launch and a 3 MB copy per frame of a feature tracker's normal turnover). old(i) = i + #{r : rem[r] - r <= i} for the ascending old indices
// rem[0 .. n) of the removed landmarks; the observer blocks also move the origin points and chart constants (st planes) to the other buffer. n = 0: nothing to do.
constexpr int GATHER_WORDS = 8, GATHER_MAXN = 64 * GATHER_WORDS; // up to 512 landmarks in the buffers the launch reads
struct GatherArgs {
    int n;                                 // landmarks removed (0: nothing to do)
    int nprop;                             // landmarks that are propagated: the first nprop of the N the launch covers. The others are HELD (eqf_add_landmarks_held, below)
    unsigned long long surv[GATHER_WORDS]; // bit o set: the landmark at the old position o survives; landmark i < nprop of the new state is the (i + 1)-th set bit
    const double* st_in;                   // q0 + chart constants, (CC_OFF + CC_PLANES) planes of stride Ncap
    double* st_out;                        // where the planes go (the other buffer when landmarks are removed, else the same one: only held landmarks are written then)
    const double* held;                    // pinned host packet of the held landmarks: [0] their variance, [1 + 3 t ..] the point of held landmark t; nullptr: they are in
                                           // memory already (appended by an ordinary pass in front of this launch)
};
// HELD landmarks (round 5, eqf_add_landmarks_held): the frame's new landmarks, appended IN FRONT of the propagation that the reference runs before it appends them
// (src/VIOFilter.cpp:217 behind :196). They belong to the time behind this propagation and pass through it untouched - F = I, no input and no process noise for their rows, no
// observer steps - and their cross-covariances are exact zeros, so every sum that involves them adds 0 * x: the result is bit for bit what appending them afterwards
// gives. With `held` set nothing of them is in memory yet: the tile workgroups take Sigma's entries for them as (variance on the diagonal, 0 elsewhere) and WRITE their rows and
// columns of the new Sigma, the observer block's lanes write their planes (origin point, chart constants, Q = identity) and evaluate their output blocks - no append pass.
// position of the k-th (0-based) set bit of x (k < popcount(x))
__device__ __forceinline__ int select64(unsigned long long x, int k) {
    int pos = 0;
    unsigned v = (unsigned)x;
    int c = __popc(v);
    if (k >= c) {
        k -= c;
        pos = 32;
        v = (unsigned)(x >> 32);
    }
    c = __popc(v & 0xffffu);
    if (k >= c) {
        k -= c;
        pos += 16;
        v >>= 16;
    }
    c = __popc(v & 0xffu);
    if (k >= c) {
        k -= c;
        pos += 8;
        v >>= 8;
    }
    c = __popc(v & 0xfu);
    if (k >= c) {
        k -= c;
        pos += 4;
        v >>= 4;
    }
    c = __popc(v & 0x3u);
    if (k >= c) {
        k -= c;
        pos += 2;
        v >>= 2;
    }
    if (k >= (int)(v & 1u))
        pos += 1;
    return pos;
}
// the old position of landmark i of the new state: a few dozen VALU operations on the survivor mask in the argument segment (a table of removed indices walked entry by
// entry cost the propagation kernel 2.3 us: one scalar load per entry in front of every Sigma load)
__device__ __forceinline__ int gather_old(const GatherArgs& ga, int i) {
    if (!ga.n)
        return i;
    int k = i, base = 0;
    unsigned long long word = ga.surv[0];
    bool found = false;
#pragma unroll
    for (int w = 0; w < GATHER_WORDS; ++w) {
        const unsigned long long x = ga.surv[w];
        const int c = __popcll(x);
        if (!found) {
            if (k < c) {
                found = true;
                word = x;
                base = 64 * w;
            } else
                k -= c;
        }
    }
    return found ? base + select64(word, k) : i;
}
struct StageArgs {
    int M; // 0: nothing to stage
    const double *y_h, *ylm_h; // pinned host packet
    const int* idx_h;
    double *y_d, *ylm_d;       // HBM copies
    int* idx_d;
};
// ---------------------------------------------------------------------------------------------------
// Host doorbell. The two kernels whose results the host waits for (outlier statistics, innovation lift) write them into
// the pinned result packet; the LAST workgroup to finish then stores a sequence number next to them. The host polls that
// word instead of the stream's completion signal and sees the results about 6 us earlier (scripts/ubench/doorbell.hip).
// Ordering: every workgroup fences its result stores at system scope before its atomic increment; the workgroup that
// observes all increments fences again and only then writes the sequence number.
__device__ __forceinline__ void ring_doorbell(int* __restrict__ count, int* __restrict__ host_flag, int seq, int nblocks = (int)gridDim.x) {
    if (!count)
        return;
    __threadfence_system();
    if (threadIdx.x == 0) {
        if (atomicAdd(count, 1) == nblocks - 1) {
            atomicExch(count, 0);
            __threadfence_system();
            *reinterpret_cast<volatile int*>(host_flag) = seq;
        }
    }
}
// ---------------------------------------------------------------------------------------------------
// K3: per measurement j: yHat, yTilde and the 2x3 block of C (measureSystemState VIOState.cpp:70-78,
// EqFoutputMatrixCiStar euclid.cpp:162-184, invdepth.cpp:255-266, outputMatrixCi EqFMatrices.cpp:84-89).
// Optionally the outlier statistics of VIOFilter::removeOutliers (VIOFilter.cpp:304-334).
struct MeasOut {
    double c[6];
    double yt[2];
    V3 qh;
};
__device__ __forceinline__ MeasOut measure_one(int chart, const Cam& cam, V3 p0, Qt q, double a, double yu, double yv, bool star, const M3& r0m) {
    MeasOut o;
    const M3 RQ = q_mat(q);
    const V3 qh = (1.0 / a) * (transpose(RQ) * p0);
    o.qh = qh;
    double hu, hv;
    cam_project(cam, qh, hu, hv);
    o.yt[0] = yu - hu;
    o.yt[1] = yv - hv;
    const V3 yHat = normalized(qh);
    if (chart == EQVIO_COORD_NORMAL) {
        // EqFoutputMatrixCiStar_normal (coordinateSuite/normal.cpp:57-65): [J(yHat) R_Q^T chartInvDiff0_normal(q0) | 0], yHat = R_Q^T y0; the pixel is not used
        V3 j0, j1, d0, d1;
        cam_jac(cam, transpose(RQ) * normalized(p0), j0, j1);
        normal_invdiff0(p0, d0, d1);
        const V3 e0 = transpose(RQ) * d0, e1 = transpose(RQ) * d1;
        o.c[0] = dot(j0, e0);
        o.c[1] = dot(j0, e1);
        o.c[2] = 0.0;
        o.c[3] = dot(j1, e0);
        o.c[4] = dot(j1, e1);
        o.c[5] = 0.0;
        return o;
    }
    const V3 yTru = star ? cam_undistort(cam, yu, yv) : cam_undistort(cam, hu, hv);
    V3 a0, a1, b0, b1;
    cam_jac_skew(cam, yTru, a0, a1);
    cam_jac_skew(cam, yHat, b0, b1);
    const V3 g0 = 0.5 * (a0 + b0), g1 = 0.5 * (a1 + b1);
    const double iq2 = 1.0 / norm2(p0);
    V3 c0 = iq2 * cross(p0, RQ * g0);
    V3 c1 = iq2 * cross(p0, RQ * g1);
    if (chart == EQVIO_COORD_INVDEPTH) { // r0m = ind2euc_r0(p0): stored chart constant
        const M3 Mt = transpose(r0m);
        c0 = Mt * c0;
        c1 = Mt * c1;
    }
    o.c[0] = c0.x;
    o.c[1] = c0.y;
    o.c[2] = c0.z;
    o.c[3] = c1.x;
    o.c[4] = c1.y;
    o.c[5] = c1.z;
    return o;
}
__global__ void __launch_bounds__(64) k_measure(int M, int Mcap, int Ncap, int chart, Cam cam, int star, const int* __restrict__ lmidx,
                                                const double* __restrict__ y, const double* __restrict__ q0, const double* __restrict__ Qq,
                                                const double* __restrict__ Qa, double* __restrict__ C, double* __restrict__ ytil,
                                                int* __restrict__ lmidx_dev, int* __restrict__ flags) {
    // lmidx / y live in the pinned host packet (read once, zero-copy); lmidx is mirrored to HBM for k_build_Z.
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) {
        flags[0] = 0;
        flags[1] = 0;
        flags[3] = 0;
    }
    if (j >= M)
        return;
    const int i = lmidx[j];
    lmidx_dev[j] = i;
    const MeasOut o = measure_one(chart, cam, ld3(q0, Ncap, i), ldq(Qq, Ncap, i), Qa[i], y[2 * j], y[2 * j + 1], star != 0,
                                  chart == EQVIO_COORD_INVDEPTH ? ld_cc(q0, Ncap, i, CC_R0) : M3{});
#pragma unroll
    for (int e = 0; e < 6; ++e)
        C[e * Mcap + j] = o.c[e];
    ytil[2 * j] = o.yt[0];
    ytil[2 * j + 1] = o.yt[1];
}
// stats: out[0..N) absErr, out[N..2N) probErr, out[2N..3N) |q_hat|^2 ; unmeasured -> -1
// It also emits what k_measure would (C blocks, residuals, index map) for the same measurement, so that the vision
// update can skip k_measure when the host removes / adds no landmark in between (the common case).
template <typename TS>
__device__ __forceinline__ void outlier_stats_body(int N, int Ncap, int ld, int chart, const Cam& cam, const double* __restrict__ ylm,
                                                   const double* __restrict__ q0, const double* __restrict__ Qq,
                                                   const double* __restrict__ Qa, const TS* __restrict__ Sig, double* __restrict__ out, int star,
                                                   double* __restrict__ C, double* __restrict__ ytil, int* __restrict__ lmidx_dev, int* __restrict__ flags,
                                                   double& abs_err, double& prob_err, bool emit = true, int i_explicit = -1) {
    // emit: also reset the status flags and write C / yTilde / the index map (false inside k_build_Z, which does both itself)
    // i_explicit: the landmark, for a caller whose lanes are not numbered by the grid (the look-ahead kernel's statistics workgroup)
    const int i = i_explicit >= 0 ? i_explicit : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (emit && i == 0) {
        flags[0] = 0;
        flags[1] = 0;
        flags[3] = 0;
    }
    if (i >= N)
        return;
    // ylm: the measurement sorted by LANDMARK in the pinned host packet (planes u, v, measurement index or -1): three independent
    // coalesced zero-copy loads, one PCIe round trip (an index followed by y[index] would be two)
    const double yu = ylm[i], yv = ylm[Ncap + i];
    const int j = (int)ylm[2 * Ncap + i];
    const V3 p0 = ld3(q0, Ncap, i);
    const Qt q = ldq(Qq, Ncap, i);
    const double a = Qa[i];
    if (j < 0) {
        const V3 qh = (1.0 / a) * q_rot(q_inv(q), p0);
        out[i] = -1.0;
        out[N + i] = -1.0;
        out[2 * N + i] = norm2(qh);
        return;
    }
    const M3 r0m = chart == EQVIO_COORD_INVDEPTH ? ld_cc(q0, Ncap, i, CC_R0) : M3{};
    const MeasOut o = measure_one(chart, cam, p0, q, a, yu, yv, false, r0m);
    if (emit) {
        const MeasOut os = star ? measure_one(chart, cam, p0, q, a, yu, yv, true, r0m) : o;
#pragma unroll
        for (int e = 0; e < 6; ++e)
            C[e * Ncap + j] = os.c[e];
        ytil[2 * j] = os.yt[0];
        ytil[2 * j + 1] = os.yt[1];
        lmidx_dev[j] = i;
    }
    const int l = 21 + 3 * i;
    double S[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            S[r][c] = Sig[l + r + (size_t)(l + c) * ld];
    // cov = C0 S C0^T
    double CS[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            CS[r][c] = o.c[r * 3 + 0] * S[0][c] + o.c[r * 3 + 1] * S[1][c] + o.c[r * 3 + 2] * S[2][c];
    const double v00 = CS[0][0] * o.c[0] + CS[0][1] * o.c[1] + CS[0][2] * o.c[2];
    const double v01 = CS[0][0] * o.c[3] + CS[0][1] * o.c[4] + CS[0][2] * o.c[5];
    const double v10 = CS[1][0] * o.c[0] + CS[1][1] * o.c[1] + CS[1][2] * o.c[2];
    const double v11 = CS[1][0] * o.c[3] + CS[1][1] * o.c[4] + CS[1][2] * o.c[5];
    const double det = v00 * v11 - v01 * v10;
    // inverse2 * yt
    const double t0 = (v11 / det) * o.yt[0] + (-v01 / det) * o.yt[1];
    const double t1 = (-v10 / det) * o.yt[0] + (v00 / det) * o.yt[1];
    abs_err = sqrt(o.yt[0] * o.yt[0] + o.yt[1] * o.yt[1]);
    prob_err = o.yt[0] * t0 + o.yt[1] * t1;
    out[i] = abs_err;
    out[N + i] = prob_err;
    out[2 * N + i] = norm2(o.qh);
}
// VIO_eqf::getOutputCovById (VIO_eqf.cpp:196-211) for every landmark of the state in one pass: out[4 i ..] = C0_i Sigma_ii C0_i^T (row-major 2 x 2) with
// C0_i = outputMatrixCi at the current estimate (independent of the measured pixel). The reference's removeOutliers (VIOFilter.cpp:304-334) asks for
// them one id at a time; a binding that keeps the reference's VIOFilter.cpp unchanged fetches all of them on the first call of a frame.
template <typename TS>
__global__ void __launch_bounds__(64) k_output_cov(int N, int Ncap, int ld, int chart, Cam cam, const double* __restrict__ q0, const double* __restrict__ Qq,
                                                   const double* __restrict__ Qa, const TS* __restrict__ Sig, double* __restrict__ out, int* __restrict__ door_count = nullptr,
                                                   int* __restrict__ door_host = nullptr, int door_seq = 0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) {
        ring_doorbell(door_count, door_host, door_seq); // (round 5: a caller that waits for these numbers polls a doorbell instead of the stream's completion signal)
        return;
    }
    const V3 p0 = ld3(q0, Ncap, i);
    const Qt q = ldq(Qq, Ncap, i);
    const double a = Qa[i];
    const M3 r0m = chart == EQVIO_COORD_INVDEPTH ? ld_cc(q0, Ncap, i, CC_R0) : M3{};
    const MeasOut o = measure_one(chart, cam, p0, q, a, 0.0, 0.0, false, r0m);
    const int l = 21 + 3 * i;
    double S[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            S[r][c] = Sig[l + r + (size_t)(l + c) * ld];
    double CS[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            CS[r][c] = o.c[r * 3 + 0] * S[0][c] + o.c[r * 3 + 1] * S[1][c] + o.c[r * 3 + 2] * S[2][c];
    out[4 * i + 0] = CS[0][0] * o.c[0] + CS[0][1] * o.c[1] + CS[0][2] * o.c[2];
    out[4 * i + 1] = CS[0][0] * o.c[3] + CS[0][1] * o.c[4] + CS[0][2] * o.c[5];
    out[4 * i + 2] = CS[1][0] * o.c[0] + CS[1][1] * o.c[1] + CS[1][2] * o.c[2];
    out[4 * i + 3] = CS[1][0] * o.c[3] + CS[1][1] * o.c[4] + CS[1][2] * o.c[5];
    ring_doorbell(door_count, door_host, door_seq);
}
template <typename TS>
__global__ void __launch_bounds__(64) k_outlier_stats(int N, int Ncap, int ld, int chart, Cam cam, const double* __restrict__ ylm,
                                                      const double* __restrict__ q0, const double* __restrict__ Qq,
                                                      const double* __restrict__ Qa, const TS* __restrict__ Sig, double* __restrict__ out, int star,
                                                      double* __restrict__ C, double* __restrict__ ytil, int* __restrict__ lmidx_dev, int* __restrict__ flags,
                                                      int* __restrict__ door_count, int* __restrict__ door_host, int door_seq, double thrAbs, double thrProb,
                                                      int* __restrict__ spec, int spec_seq, double* __restrict__ stats_dev) {
    double abs_err = -1.0, prob_err = -1.0; // stay negative for lanes without a measured landmark
    outlier_stats_body<TS>(N, Ncap, ld, chart, cam, ylm, q0, Qq, Qa, Sig, out, star, C, ytil, lmidx_dev, flags, abs_err, prob_err);
    if (stats_dev) { // k_select_outliers, the next kernel of the stream, decides on the device: keep a copy of the two errors in HBM
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i < N) {
            stats_dev[i] = abs_err;
            stats_dev[N + i] = prob_err;
        }
    }
    // Speculative frame tail (eqf_stats_then_update): the update kernels are already queued behind this one. If any measured
    // landmark is an outlier candidate (VIOFilter.cpp:316-330: absErr > thrAbs or probErr > thrProb) the host has a decision
    // to make, so the queued kernels must not run: they compare this word with their sequence number and return at once.
    if (spec && abs_err >= 0.0 && (abs_err > thrAbs || prob_err > thrProb)) // the comparisons of VIOFilter.cpp:316-330 (NaN: false)
        *spec = spec_seq;
    ring_doorbell(door_count, door_host, door_seq);
}

// VIOFilter::removeOutliers' decision (VIOFilter.cpp:304-364) on the device, so that a frame with outlier candidates needs no host round trip
// between the statistics and the update: candidates are the measured landmarks with absErr > thrAbs, else probErr > thrProb; they are ranked
// absolute outliers first (largest absErr first), then probabilistic ones (largest probErr first), and the first max_outliers of them are
// discarded. A discarded landmark's measurement is taken out of the update by zeroing its C block and its residual (its two columns of Z
// become (0, R_jj, 0): decoupled, W = 0 there), which leaves every other state exactly where the reference's "erase, then update" puts
// it: an unmeasured landmark can be marginalised before or after the update. The host removes the landmark rows after the update, from
// the list in `removed_host` (pinned: [0, N) flags, [Ncap] count of candidates, [Ncap + 1] count of discarded).
// One workgroup; sel (LDS) holds the rank keys of up to SEL_MAXN landmarks.
constexpr int SEL_MAXN = 2048;
__global__ void __launch_bounds__(256) k_select_outliers(int N, int Ncap, int M, const double* __restrict__ stats_dev, double thrAbs, double thrProb, int max_outliers,
                                                         const int* __restrict__ lmidx_dev, double* __restrict__ C, double* __restrict__ ytil,
                                                         int* __restrict__ removed_host) {
    __shared__ double s_val[SEL_MAXN];
    __shared__ signed char s_kind[SEL_MAXN]; // 0: no candidate, 1: probabilistic, 2: absolute
    __shared__ unsigned char s_rm[SEL_MAXN];
    __shared__ int s_cnt[2];
    const int tid = threadIdx.x;
    if (tid < 2)
        s_cnt[tid] = 0;
    for (int i = tid; i < N; i += 256) {
        const double a = stats_dev[i], p = stats_dev[N + i];
        const bool measured = a >= 0.0;
        const bool isabs = measured && a > thrAbs; // the comparisons of VIOFilter.cpp:316, 330 (NaN: false)
        const bool isprob = measured && !isabs && p > thrProb;
        s_kind[i] = isabs ? 2 : (isprob ? 1 : 0);
        s_val[i] = isabs ? a : p;
    }
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        int rank = 0;
        const int ki = s_kind[i];
        const double vi = s_val[i];
        if (ki) {
            for (int j = 0; j < N; ++j) {
                const int kj = s_kind[j];
                rank += (kj > ki) || (kj == ki && (s_val[j] > vi || (s_val[j] == vi && j < i)));
            }
            atomicAdd(&s_cnt[0], 1);
        }
        const bool rm = ki && rank < max_outliers;
        s_rm[i] = rm;
        removed_host[i] = rm ? 1 : 0;
        if (rm)
            atomicAdd(&s_cnt[1], 1);
    }
    __syncthreads();
    if (tid == 0) {
        removed_host[Ncap] = s_cnt[0];
        removed_host[Ncap + 1] = s_cnt[1];
    }
    for (int j = tid; j < M; j += 256) {
        if (s_rm[lmidx_dev[j]]) {
#pragma unroll
            for (int e = 0; e < 6; ++e)
                C[e * Ncap + j] = 0.0;
            ytil[2 * j] = 0.0;
            ytil[2 * j + 1] = 0.0;
        }
    }
}

// k_outlier_stats and k_select_outliers as ONE launch of one workgroup (round 5; up to SEL_ONE_WG landmarks): in a frame whose outlier decision is taken on the device the
// host queues statistics, decision and the whole update back to back, and its launch calls (5 - 6 us each) - not the kernels in front of the factorisation - set the pace
// of that part of the frame. Same arithmetic per landmark (outlier_stats_body's), same ranking: identical results to the two launches.
// A lone workgroup is bound by the latency of its dependent fp64 chains, not by issue, so the work is laid out for latency (first form: 27 us at 200 landmarks):
//   * 512 lanes: waves 0 - 3 evaluate the statistics of a landmark (output block at the estimate, C0 Sigma_ii C0^T, the two errors), waves 4 - 7 at the same time
//     what the update needs of it (the equivariant output block, the residual) - two waves per SIMD, the one's stalls are the other's issue slots;
//   * everything stays in registers until the decision is known: no store to the pinned packet (a PCIe round trip a fence would wait for) and no store of an output block
//     that the mask would have to zero again in front of the ranking;
//   * the candidates go into a list as they are found, and the ranking is among the list's entries only, a candidate against 64 entries per instruction
//     (ballot + population count) - 55 candidates of 200 landmarks with the shipped thresholds - instead of every candidate's lane walking over all landmarks.
constexpr int SEL_ONE_WG = 512;
struct alignas(16) SelCand {
    double val;
    int kind, idx; // 1: probabilistic, 2: absolute; landmark
};
constexpr int SEL_PER_LANE = SEL_ONE_WG / 256;
template <typename TS>
__global__ void __launch_bounds__(512) k_stats_select(int N, int Ncap, int ld, int chart, Cam cam, const double* __restrict__ ylm, const double* __restrict__ q0,
                                                      const double* __restrict__ Qq, const double* __restrict__ Qa, const TS* __restrict__ Sig, double* __restrict__ out, int star,
                                                      double* __restrict__ C, double* __restrict__ ytil, int* __restrict__ lmidx_dev, int* __restrict__ flags, double thrAbs,
                                                      double thrProb, int max_outliers, int M, int* __restrict__ removed_host, int* __restrict__ live_cols) {
    __shared__ signed char s_kind[SEL_ONE_WG]; // 0: no candidate, 1: probabilistic, 2: absolute, -1: beyond N
    __shared__ unsigned char s_rm[SEL_ONE_WG];
    __shared__ SelCand s_cand[SEL_ONE_WG]; // the candidates as a list, in no particular order
    __shared__ int s_cnt[2]; // candidates, discarded
    __shared__ short s_pos[SEL_ONE_WG]; // live_cols: the column pair a measurement's output block goes to
    __shared__ int s_wdead[8];
    const int tid = threadIdx.x, t = tid & 255;
    const bool stat_half = tid < 256; // wave-uniform
    if (tid < 2)
        s_cnt[tid] = 0;
    __syncthreads();
    // registers of a lane: statistics half (abs, prob, depth^2), update half (C block, residual, measurement index)
    double r_a[SEL_PER_LANE], r_p[SEL_PER_LANE], r_d[SEL_PER_LANE], r_c[SEL_PER_LANE][6], r_y[SEL_PER_LANE][2];
    int r_j[SEL_PER_LANE];
#pragma unroll
    for (int it = 0; it < SEL_PER_LANE; ++it) {
        const int i = t + 256 * it;
        r_j[it] = -1;
        if (i >= N) {
            if (stat_half && i < SEL_ONE_WG)
                s_kind[i] = -1;
            continue;
        }
        const double yu = ylm[i], yv = ylm[Ncap + i];
        const int j = (int)ylm[2 * Ncap + i];
        const V3 p0 = ld3(q0, Ncap, i);
        const Qt q = ldq(Qq, Ncap, i);
        const double a = Qa[i];
        r_j[it] = j;
        if (j < 0) { // a landmark without a measurement
            if (stat_half) {
                const V3 qh = (1.0 / a) * q_rot(q_inv(q), p0);
                r_a[it] = -1.0, r_p[it] = -1.0, r_d[it] = norm2(qh);
                s_kind[i] = 0;
            }
            continue;
        }
        const M3 r0m = chart == EQVIO_COORD_INVDEPTH ? ld_cc(q0, Ncap, i, CC_R0) : M3{};
        if (!stat_half) {
            const MeasOut os = measure_one(chart, cam, p0, q, a, yu, yv, star != 0, r0m);
#pragma unroll
            for (int e = 0; e < 6; ++e)
                r_c[it][e] = os.c[e];
            r_y[it][0] = os.yt[0], r_y[it][1] = os.yt[1];
            continue;
        }
        const MeasOut o = measure_one(chart, cam, p0, q, a, yu, yv, false, r0m);
        const int l = 21 + 3 * i;
        double S[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                S[r][c] = Sig[l + r + (size_t)(l + c) * ld];
        double CS[2][3]; // cov = C0 S C0^T
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                CS[r][c] = o.c[r * 3 + 0] * S[0][c] + o.c[r * 3 + 1] * S[1][c] + o.c[r * 3 + 2] * S[2][c];
        const double v00 = CS[0][0] * o.c[0] + CS[0][1] * o.c[1] + CS[0][2] * o.c[2];
        const double v01 = CS[0][0] * o.c[3] + CS[0][1] * o.c[4] + CS[0][2] * o.c[5];
        const double v10 = CS[1][0] * o.c[0] + CS[1][1] * o.c[1] + CS[1][2] * o.c[2];
        const double v11 = CS[1][0] * o.c[3] + CS[1][1] * o.c[4] + CS[1][2] * o.c[5];
        const double det = v00 * v11 - v01 * v10;
        const double t0 = (v11 / det) * o.yt[0] + (-v01 / det) * o.yt[1];
        const double t1 = (-v10 / det) * o.yt[0] + (v00 / det) * o.yt[1];
        const double ae = sqrt(o.yt[0] * o.yt[0] + o.yt[1] * o.yt[1]);
        const double pe = o.yt[0] * t0 + o.yt[1] * t1;
        r_a[it] = ae, r_p[it] = pe, r_d[it] = norm2(o.qh);
        const bool isabs = ae > thrAbs; // the comparisons of VIOFilter.cpp:316, 330 (NaN: false)
        const bool isprob = !isabs && pe > thrProb;
        s_kind[i] = isabs ? 2 : (isprob ? 1 : 0);
        if (isabs || isprob) {
            const int c = atomicAdd(&s_cnt[0], 1);
            s_cand[c] = SelCand{isabs ? ae : pe, isabs ? 2 : 1, i};
        }
    }
    __syncthreads();
    // ranking: candidates ordered absolute outliers first (largest absErr first), then probabilistic ones (largest probErr first), ties by index. Among the list's
    // entries only, by whole waves: wave w takes the candidates c = w, w + 8, ..; its lanes hold 64 entries of the list at a time, and a candidate (one 16-byte
    // broadcast read from LDS) is compared with all of them in one go (ballot + population count). Lane k of the wave accumulates the rank of the wave's k-th candidate.
    {
        const int ncand = s_cnt[0];
        const int w = tid >> 6, lane = tid & 63;
        int myrank = 0;
        for (int r0 = 0; r0 < ncand; r0 += 64) {
            const int e = r0 + lane;
            const SelCand en = e < ncand ? s_cand[e] : SelCand{0.0, -1, 0}; // kind -1: never in front of anything
#pragma unroll 4
            for (int kk = 0; w + 8 * kk < ncand; ++kk) {
                const SelCand ci = s_cand[w + 8 * kk];
                const int part = __popcll(__ballot((en.kind > ci.kind) || (en.kind == ci.kind && (en.val > ci.val || (en.val == ci.val && en.idx < ci.idx)))));
                myrank += lane == kk ? part : 0;
            }
        }
        const int c = w + 8 * lane;
        const bool rm = c < ncand && myrank < max_outliers;
        if (c < ncand)
            s_rm[s_cand[c].idx] = rm ? 1 : 0;
        const int nrm = __popcll(__ballot(rm));
        if (lane == 0 && nrm)
            atomicAdd(&s_cnt[1], nrm);
    }
    __syncthreads();
    // live_cols (the look-ahead kernel takes its panel count from it): the measurements of the landmarks that stay go to the FRONT of C / yTilde / the index map, in their
    // order, the discarded ones behind them (their columns of Z are (0, R_jj, 0): decoupled from everything, W = 0 there) - the factorisation then ends with the last
    // panel that holds a live column instead of walking over the dead ones (55 of 190 measurements with the shipped thresholds: 3 of 12 panels).
    if (live_cols) { // (uniform)
        // s_pos[j] first holds "measurement j is discarded" (written by the landmark's lane of the update half), then its new place
        if (!stat_half) {
#pragma unroll
            for (int it = 0; it < SEL_PER_LANE; ++it) {
                const int i = t + 256 * it;
                if (i < N && r_j[it] >= 0)
                    s_pos[r_j[it]] = (s_kind[i] > 0 && s_rm[i]) ? 1 : 0;
            }
        }
        __syncthreads();
        const int w = tid >> 6, lane = tid & 63;
        const bool dead = tid < M && s_pos[tid] != 0;
        const unsigned long long db = __ballot(dead);
        if (lane == 0)
            s_wdead[w] = __popcll(db);
        __syncthreads();
        int dead_before = __popcll(db & ((1ull << lane) - 1)), dead_all = 0;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) {
            dead_before += ww < w ? s_wdead[ww] : 0;
            dead_all += s_wdead[ww];
        }
        if (tid < M)
            s_pos[tid] = (short)(dead ? (M - dead_all) + dead_before : tid - dead_before);
        if (tid == 0)
            *live_cols = 2 * (M - dead_all);
        __syncthreads();
    }
    // results: the statistics half writes the host's packet, the update half the output blocks (zero for a discarded landmark: its two columns of Z are (0, R_jj, 0))
    if (tid == 0) {
        removed_host[Ncap] = s_cnt[0];
        removed_host[Ncap + 1] = s_cnt[1];
        flags[0] = 0;
        flags[1] = 0;
        flags[3] = 0;
    }
#pragma unroll
    for (int it = 0; it < SEL_PER_LANE; ++it) {
        const int i = t + 256 * it;
        if (i >= N)
            continue;
        const bool rm = r_j[it] >= 0 && s_kind[i] > 0 && s_rm[i];
        if (stat_half) {
            out[i] = r_a[it];
            out[N + i] = r_p[it];
            out[2 * N + i] = r_d[it];
            removed_host[i] = rm ? 1 : 0;
        } else if (r_j[it] >= 0) {
            const int j = live_cols ? (int)s_pos[r_j[it]] : r_j[it];
#pragma unroll
            for (int e = 0; e < 6; ++e)
                C[e * Ncap + j] = rm ? 0.0 : r_c[it][e];
            ytil[2 * j] = rm ? 0.0 : r_y[it][0];
            ytil[2 * j + 1] = rm ? 0.0 : r_y[it][1];
            lmidx_dev[j] = i;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// K8b: blocked right-looking factorisation of Z = [S ; T ; y^T], ONE kernel launch per 32-column panel.
//
// Step k (panel = columns [kb, kb+w), c0 = kb + w): workgroup (bi, bj) owns the 32x32 trailing tile rows
// i0 = c0 + 32 bi, cols j0 = c0 + 32 bj and, with no inter-workgroup synchronisation,
//   1. reads L_kk^-1 (32x32, published by the previous launch) into LDS;
//   2. P_I = Z[I, panel] L^-T and P_J = Z[J, panel] L^-T on fp64 MFMA (Z operand straight from L2) -> LDS;
//   3. Z[I, J] -= P_I P_J^T on fp64 MFMA from LDS;
//   4. workgroups with bj == 0 store the rows >= m of P_I (final W = T L^-T and z^T = y^T L^-T) to Wout;
//   5. the workgroup that owns the NEXT diagonal tile (bi = bj = 0) keeps its updated tile on chip, eliminates it
//      (square-root-free right-looking LDL^T, the tile distributed over the registers of its 256 lanes, pivot
//      column / row exchanged through a double-buffered LDS line: one barrier per pivot) while applying the same row
//      operations to an identity, and publishes L_{k+1}^-1 = diag(d)^-1/2 Lu^-1 for the next launch.
// Only that one workgroup runs the pivot chain, alone on its CU; all others have exited. The chain of 400 pivots
// is the critical path of the whole vision update (SURVEY.md §7 "hard parts").
// The last panel (c0 >= m) runs with update = 0: steps 1, 2 (P_I only) and 4.
// flags[0] is set when a pivot is not positive (EQF_E_NOT_SPD).
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int CH_LDP = 48; // MFMA operand tiles in LDS: k and k+1 columns 32 dwords apart -> conflict-free ds_read_b64

// ---- inverse Cholesky factor of a 32x32 SPD tile, blocked 16 + 16 -----------------------------------------------
// D = [[D11, .],[D21, D22]]:  L11^-1 by a 16x16 elimination;  L21 = D21 L11^-T (MFMA);  S22 = D22 - L21 L21^T (MFMA);
// L22^-1 by a second 16x16 elimination;  L^-1 = [[L11^-1, 0], [-L22^-1 L21 L11^-1, L22^-1]] (two MFMA products).
// The 16x16 eliminations are block LDL^T with 2x2 pivots, ONE element of the tile and one of M = Lu^-1 per lane
// (256 lanes), pivot columns / rows exchanged through a double-buffered LDS line: one barrier per two pivots and
// ~20 VALU ops per round, so a round is bound by its dependent chain (barrier + LDS + reciprocal), not by issue.
struct alignas(16) dpair {
    double x, y;
};
// 16x16x16 product on one wave: C[i][c] = sum_p I[i][p] * J[c][p], I at Ib[i + p*ldi], J at Jb[c + p*ldj].
// Lane (lr = lane & 15, lk = lane >> 4) returns C[lr][lk + 4q], q = 0..3.
__device__ __forceinline__ d4 mfma16_nt(const double* __restrict__ Ib, int ldi, const double* __restrict__ Jb, int ldj) {
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    d4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int p = 4 * st + lk;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Jb[lr + p * ldj], Ib[lr + p * ldi], acc, 0, 0, 0);
    }
    return acc;
}
// ---- cross-lane helpers for the register-resident elimination ----------------------------------------------------
// DPP row_newbcast:J (gfx90a+): every lane of a 16-lane row receives the value lane J of ITS row holds.
template <int J> __device__ __forceinline__ double row_bcast(double v) {
    // 64-bit operand + row_newbcast -> one v_mov_b64_dpp (DP-ALU DPP). Every lane is written, so `old` is never used: an
    // empty asm "defines" it, which spares the copy a tied old = source operand would cost.
    double old;
    asm volatile("" : "=v"(old));
    return __builtin_amdgcn_update_dpp(old, v, 0x150 + J, 0xf, 0xf, false);
}
// x -= m * (the value x holds in lane J of the lane's own 16-lane row) as ONE instruction, v_fmac_f64_dpp: a v_mov_b64_dpp costs a lone wave ~19 cycles
// of issue and the FMA behind it ~9 (scripts/ubench/issue.hip). The hardware does not interlock a DPP read behind a VALU write of the same register (2 wait
// states), the compiler does not see into inline assembly, and an s_nop costs a lone wave a full issue slot: the sequences below are ORDERED so that at
// least two instructions lie between a write of a register and a DPP read of it, and every block opens with s_nop 1 (the compiler may have copied an operand
// into its register right in front of the block).
#define EQF_FNMA_DPP(acc, src, m, lane) "v_fmac_f64_dpp %[" #acc "], -%[" #src "], %[" #m "] row_newbcast:%[" #lane "] row_mask:0xf bank_mask:0xf\n\t"
#define EQF_MOV_DPP(dst, src, lane) "v_mov_b64_dpp %[" #dst "], %[" #src "] row_newbcast:%[" #lane "] row_mask:0xf bank_mask:0xf\n\t"
// rank-2 update of five registers with the multipliers (m0, m1) and the pivot rows J0, J0 + 1 (read in place: a pivot row is not modified by its own pair's
// update, its multipliers are masked to zero)
template <int J0> __device__ __forceinline__ void ldl_rank2_x5(double& x1, double& x2, double& x3, double& x4, double& x5, double m0, double m1) {
    asm volatile("s_nop 1\n\t" EQF_FNMA_DPP(x1, x1, m0, j0) EQF_FNMA_DPP(x2, x2, m0, j0) EQF_FNMA_DPP(x3, x3, m0, j0) EQF_FNMA_DPP(x4, x4, m0, j0) EQF_FNMA_DPP(x5, x5, m0, j0)
                 EQF_FNMA_DPP(x1, x1, m1, j1) EQF_FNMA_DPP(x2, x2, m1, j1) EQF_FNMA_DPP(x3, x3, m1, j1) EQF_FNMA_DPP(x4, x4, m1, j1) EQF_FNMA_DPP(x5, x5, m1, j1)
                 : [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5)
                 : [m0] "v"(m0), [m1] "v"(m1), [j0] "n"(J0), [j1] "n"(J0 + 1));
}
template <int J0> __device__ __forceinline__ void ldl_rank2_x4(double& x1, double& x2, double& x3, double& x4, double m0, double m1) {
    asm volatile("s_nop 1\n\t" EQF_FNMA_DPP(x1, x1, m0, j0) EQF_FNMA_DPP(x2, x2, m0, j0) EQF_FNMA_DPP(x3, x3, m0, j0) EQF_FNMA_DPP(x4, x4, m0, j0)
                 EQF_FNMA_DPP(x1, x1, m1, j1) EQF_FNMA_DPP(x2, x2, m1, j1) EQF_FNMA_DPP(x3, x3, m1, j1) EQF_FNMA_DPP(x4, x4, m1, j1)
                 : [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4)
                 : [m0] "v"(m0), [m1] "v"(m1), [j0] "n"(J0), [j1] "n"(J0 + 1));
}
// First half of a 4x4 round (pivot rows J0, J0 + 1; second pair J0 + 2, J0 + 3): H = G2 - E Q^T with Q[i][j] = A[J0+2+i][J0+j] = g_j of row J0+2+i, the second
// pivot block S = H of the rows J0 + 2, J0 + 3, and the first pair's rank-2 update of five registers.
template <int J0>
__device__ __forceinline__ void ldl_first_pair(double& h1, double& h2, double& s11, double& s21, double& s22, double g0, double g1, double e1, double e2, double& x1,
                                               double& x2, double& x3, double& x4, double& x5) {
    asm volatile("s_nop 1\n\t" EQF_FNMA_DPP(h1, g0, e1, j2) EQF_FNMA_DPP(h2, g0, e1, j3) EQF_FNMA_DPP(h1, g1, e2, j2) EQF_FNMA_DPP(h2, g1, e2, j3)
                 EQF_FNMA_DPP(x1, x1, e1, j0) EQF_FNMA_DPP(x2, x2, e1, j0)
                 EQF_MOV_DPP(s11, h1, j2) EQF_MOV_DPP(s21, h1, j3) EQF_MOV_DPP(s22, h2, j3)
                 EQF_FNMA_DPP(x3, x3, e1, j0) EQF_FNMA_DPP(x4, x4, e1, j0) EQF_FNMA_DPP(x5, x5, e1, j0)
                 EQF_FNMA_DPP(x1, x1, e2, j1) EQF_FNMA_DPP(x2, x2, e2, j1) EQF_FNMA_DPP(x3, x3, e2, j1) EQF_FNMA_DPP(x4, x4, e2, j1) EQF_FNMA_DPP(x5, x5, e2, j1)
                 : [h1] "+v"(h1), [h2] "+v"(h2), [s11] "=&v"(s11), [s21] "=&v"(s21), [s22] "=&v"(s22), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4),
                   [x5] "+v"(x5)
                 : [g0] "v"(g0), [g1] "v"(g1), [e1] "v"(e1), [e2] "v"(e2), [j0] "n"(J0), [j1] "n"(J0 + 1), [j2] "n"(J0 + 2), [j3] "n"(J0 + 3));
}
// odd lanes receive the value of the lane below them (quad_perm [0,0,2,2]); even lanes keep their own
__device__ __forceinline__ double lane_below_for_odd(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0xA0, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0xA0, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int L> __device__ __forceinline__ double read_lane(double v) { // wave-uniform value of lane L
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), L), __builtin_amdgcn_readlane(__double2loint(v), L));
}
__device__ __forceinline__ double fetch_lane(double v, int byte_addr) { // value held by lane byte_addr / 4
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v)));
}
// ---- 4x4 pivot blocks (round 4) ---------------------------------------------------------------------------------
// A single wave issues one instruction per ~4.5 cycles and a dependent fp64 op returns after 11-13 (scripts/ubench/latency.hip: v_fma_f64 11,
// v_mul_f64 13, v_rcp_f64 19, DPP row broadcast 10, two v_readlane 22, ds_bpermute 75 shader cycles), so a round is bound by BOTH its dependent
// chain and its instruction count. The 2x2 rounds of rounds 1-3 paid one cross-lane round trip per TWO pivots and were latency bound (~430 cycles
// for ~48 instructions). A 4x4 round is two 2x2 rounds glued WITHOUT the cross-lane fetch between them: every lane gathers the four pivot-column
// entries G = A[r, J:J+4] of its row once (ds_bpermute, under the first reciprocal), forms the first pair's multipliers E = G1 P^-1 and the
// second pair's multiplier sources H = G2 - E Q^T itself; the second pivot block S = R - Q P^-1 Q^T is then simply H of the rows J+2, J+3 (three DPP
// row broadcasts), F2 = H S^-1. Rank-2 update with (E; pivot rows J, J+1), then rank-2 with (F2; the updated pivot rows J+2, J+3).
// every lane (r, *) receives the values v of the lanes (r, 0), (r, 1), (r, 2), (r, 3): the gfx950 pair v_permlane32_swap + v_permlane16_swap (VALU). ds_bpermute does the
// same through the LDS crossbar at the same issue cost (14 cycles per instruction for a lone wave against ~8 per swap + copy) - but in the look-ahead kernel's owner the
// other waves' operand reads queue in front of it: the elimination took 2.6 us alone and 2.95 us under the tail's LDS traffic (profiles/r04_*_lookahead_trace.txt).
typedef unsigned ldl_v2u __attribute__((ext_vector_type(2)));
template <bool FOUR> __device__ __forceinline__ void row_allgather(double v, double& g0, double& g1, double& g2, double& g3) {
    const unsigned hi = (unsigned)__double2hiint(v), lo = (unsigned)__double2loint(v);
    const ldl_v2u h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false); // .x = rows [0, 1, 0, 1], .y = rows [2, 3, 2, 3]
    const ldl_v2u l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const ldl_v2u h01 = __builtin_amdgcn_permlane16_swap(h32.x, h32.x, false, false); // .x = row 0 everywhere, .y = row 1
    const ldl_v2u l01 = __builtin_amdgcn_permlane16_swap(l32.x, l32.x, false, false);
    g0 = __hiloint2double((int)h01.x, (int)l01.x);
    g1 = __hiloint2double((int)h01.y, (int)l01.y);
    if (FOUR) {
        const ldl_v2u h23 = __builtin_amdgcn_permlane16_swap(h32.y, h32.y, false, false);
        const ldl_v2u l23 = __builtin_amdgcn_permlane16_swap(l32.y, l32.y, false, false);
        g2 = __hiloint2double((int)h23.x, (int)l23.x);
        g3 = __hiloint2double((int)h23.y, (int)l23.y);
    }
}
template <int B> __device__ __forceinline__ void ldl16x4_round(double (&a)[4], double (&mm)[4], int r, const int (&gaddr)[4]) {
    constexpr int J = 4 * B;
    // P straight from the owning lanes: A[J][J] lane (J, 0), A[J+1][J] lane (J+1, 0), A[J+1][J+1] lane (J+1, 1)
    const double p11 = read_lane<J>(a[B]), p21 = read_lane<J + 1>(a[B]), p22 = read_lane<J + 1 + 16>(a[B]);
    double g0u, g1u, h1 = 0.0, h2 = 0.0; // A[r][J], A[r][J+1]; A[r][J+2], A[r][J+3] (which become the columns J+2, J+3 after the first pair's update)
    row_allgather<(B < 3)>(a[B], g0u, g1u, h1, h2);
    const double iP0 = fast_rcp(fma(p11, p22, -p21 * p21));
    const double iP = (r >= J + 2) ? iP0 : 0.0; // rows of the first pair and above are not touched: their multipliers are zero
    const double e1 = (g0u * p22 - g1u * p21) * iP, e2 = (g1u * p11 - g0u * p21) * iP; // (A[r][J], A[r][J+1]) P^-1; the numerators form under the reciprocal
    if (B == 3) { // last block: rows 14, 15 only
        ldl_rank2_x5<J>(a[3], mm[0], mm[1], mm[2], mm[3], e1, e2);
        return;
    }
    double s11, s21, s22;
    // Static pruning: registers of A whose columns are all < J hold eliminated columns (register B keeps the pivot blocks for the final scaling and
    // only needs the first pair's update); registers of M whose columns are all > J + 3 still hold identity columns on which the pivot rows are zero.
    // That leaves register B of A plus four more: a[B+1 .. 3] and mm[0 .. B].
    double& x1 = a[(B + 1) & 3]; // the next pivot columns first
    double& x2 = (B < 2) ? a[(B + 2) & 3] : mm[0];
    double& x3 = (B < 1) ? a[3] : mm[(B < 2) ? 0 : 1];
    double& x4 = mm[B];
    ldl_first_pair<J>(h1, h2, s11, s21, s22, g0u, g1u, e1, e2, x1, x2, x3, x4, a[B]);
    const double iS0 = fast_rcp(fma(s11, s22, -s21 * s21));
    const double iS = (r >= J + 4) ? iS0 : 0.0; // rows of the block and above are not touched by the second pair
    const double f3 = (h1 * s22 - h2 * s21) * iS, f4 = (h2 * s11 - h1 * s21) * iS;
    ldl_rank2_x4<J + 2>(x1, x2, x3, x4, f3, f4);
}
// 16x16 elimination on ONE wave: lane (r = lane & 15, cq = lane >> 4) owns the four entries D[r][cq + 4k], k = 0..3, of the
// full symmetric tile (identity padding outside the tile) and receives Linv[r][cq + 4k] (0 above the diagonal).
// Block LDL^T with 2x2 pivots, two pivot pairs per round, while applying the same row operations to an identity (M = Lu^-1), then
// Linv = blkdiag(chol(D_b)^-1) M. sX: 16 x 17 doubles of LDS, used once after the loop to hand every lane the pivot
// block of its own row pair (rows stop changing after their own pair's step, so the final tile still holds every pivot block).
__device__ __forceinline__ void ldl16_inverse_wave(double (&a)[4], double (&out)[4], int* __restrict__ flags, bool check_row, double* __restrict__ sX) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, cq = lane >> 4;
    double mm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        mm[k] = (r == cq + 4 * k) ? 1.0 : 0.0;
    const int gaddr[4] = {4 * r, 4 * (r + 16), 4 * (r + 32), 4 * (r + 48)};
    ldl16x4_round<0>(a, mm, r, gaddr);
    ldl16x4_round<1>(a, mm, r, gaddr);
    ldl16x4_round<2>(a, mm, r, gaddr);
    ldl16x4_round<3>(a, mm, r, gaddr);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        sX[r * 17 + cq + 4 * k] = a[k];
    const int j = r & ~1;
    const double p11 = sX[j * 17 + j], p21 = sX[(j + 1) * 17 + j], p22 = sX[(j + 1) * 17 + j + 1];
    // Linv = blkdiag(chol(D_b)^-1) M : row j -> M[j]/l11 ; row j+1 -> (M[j+1] - (l21/l11) M[j]) / l22. (1/l22 = p11 rsqrt(p11) rsqrt(p11 p22 - p21^2) would make
    // the two reciprocal square roots independent, ~60 cycles; on the template configuration, cond(S) ~ 1e13, it put Sigma+ 3.5e-9 from the 50-digit truth
    // instead of 1.4e-9 (tests/test_truth_mp.py): not taken.)
    const bool ok = (p11 > 0.0) && (fma(p11, p22, -p21 * p21) > 0.0);
    if (cq == 0 && check_row && !ok)
        __hip_atomic_store(flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // write-through: the look-ahead kernel's own lift reads it from another workgroup
    const double il11 = fast_rsqrt(ok ? p11 : 1.0);
    const double l21 = ok ? p21 * il11 : 0.0;
    const double il22 = fast_rsqrt(ok ? p22 - l21 * l21 : 1.0);
    const bool odd = (r & 1) != 0;
    const double s_self = odd ? il22 : il11;
    const double s_prev = odd ? -(l21 * il11) * il22 : 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double mprev = lane_below_for_odd(mm[k]);
        out[k] = fma(s_prev, mprev, s_self * mm[k]); // M is lower triangular by construction (row operations on an identity): exact zeros above the diagonal
    }
}
// Tile D at sD[r + c*ldd] (lower triangle valid, rows/cols >= w are identity padding, w even). Writes Linv (32x32
// column-major) to LinvOut. swork: LDL_SBUF doubles of LDS. Call after a workgroup barrier; only wave 0 does the work
// (the whole chain is sequential; one wave avoids every barrier), the other waves return immediately.
constexpr int LDL_SBUF = 7 * 256 + 32;
// The result leaves through put(r, c, v) (every entry of the 32x32 factor exactly once, zeros above the diagonal blocks included):
// ldl_inverse_tile stores it column-major to global memory, the look-ahead kernel keeps it in LDS and publishes it.
// The core works on registers: D11 arrives in `a` (elimination layout, symmetric fill, identity padding), D21 / D22 are fetched by `rest(d21, d22)` only
// when the first diagonal block has been inverted (the look-ahead owner's pivot wave forms D11 itself and receives the other two blocks from its neighbours about
// a microsecond later), and the blocks of the factor stay with the caller: o1 = Linv[0:16, 0:16], o2 = Linv[16:32, 16:32], xl = Linv[16:32, 0:16].
template <typename Rest, typename Put>
__device__ __forceinline__ void ldl_inverse_tile_regs(double (&a)[4], Rest rest, int w, Put put, int* __restrict__ flags, double* __restrict__ swork, double (&o1)[4],
                                                      double (&o2)[4], double (&xl)[4]) {
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    double* sA = swork;           // pivot blocks of an elimination (16 x 17)
    double* sLi11 = swork + 544;  // L11inv[r + c*16]: read back transposed (swork + 288 .. 543 is NOT used here: the look-ahead owner parks a block there)
    // Everything lives in the "elimination layout": lane (lr, lk) holds X[lr][lk + 4 q], q = 0..3. A product C = I J^T of two 16 x 16 matrices in that
    // layout needs no data movement at all on fp64 MFMA 16x16x4: step q contracts the columns p = lk + 4 q of BOTH operands (the order of the sum over p is
    // free), and the result comes back in the same layout. Only Y^T = L11inv^T L21^T needs one operand transposed (one LDS round trip, off the chain).
    double d21[4], d22[4];
    // A. first diagonal block
    ldl16_inverse_wave(a, o1, flags, lr < w, sA);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        sLi11[lr + (lk + 4 * k) * 16] = o1[k];
    rest(d21, d22);
    if (w <= 16) {
        // The tile's rows / columns 16 .. 31 are identity padding (the LAST panel of m = 32 p + w columns, w <= 16: N = 200 has m = 400 = 12 x 32 + 16): D21 = 0, D22 = I, so
        // L^-1 = diag(L11^-1, I) without the second elimination and the three products around it (~1.5 us of the frame's pivot chain, which ends with this tile)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lk + 4 * k;
            o2[k] = (lr == c) ? 1.0 : 0.0;
            xl[k] = 0.0;
            put(lr, c, o1[k]);
            put(lr, c + 16, 0.0);
            put(16 + lr, c, 0.0);
            put(16 + lr, 16 + c, o2[k]);
        }
        return;
    }
    // B. L21 = D21 L11inv^T ; S22 = D22 - L21 L21^T
    d4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o1[q], d21[q], acc, 0, 0, 0); // C[i][c] = sum_p I[i][p] J[c][p]: first operand J, second I
    const d4 l21 = acc;
    acc = d4{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[q], l21[q], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        a[q] = d22[q] - acc[q]; // Schur complement entry S22[lr][lk + 4 q] (L21 L21^T is symmetric to rounding)
    // C. second diagonal block
    ldl16_inverse_wave(a, o2, flags, 16 + lr < w, sA);
    // D. Y^T = L11inv^T L21^T (I = L11inv^T read back transposed, J = L21), X = -L22inv Y (I = L22inv, J = Y^T)
    acc = d4{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[q], sLi11[(lk + 4 * q) + lr * 16], acc, 0, 0, 0);
    const d4 yT = acc;
    acc = d4{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(yT[q], o2[q], acc, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lk + 4 * k;
        xl[k] = -acc[k];
        put(lr, c, o1[k]);
        put(lr, c + 16, 0.0);
        put(16 + lr, c, xl[k]);
        put(16 + lr, 16 + c, o2[k]);
    }
}
template <typename Put>
__device__ __forceinline__ void ldl_inverse_tile_put(const double* __restrict__ sD, int ldd, int w, Put put, int* __restrict__ flags, double* __restrict__ swork) {
    if (threadIdx.x >= 64)
        return;
    // this wave is the critical path of the whole frame: win the issue arbitration against co-resident workgroups
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    double a[4], d21r[4], d22r[4], o1[4], o2[4], xl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { // all of the tile requested up front
        const int c = lk + 4 * k;
        a[k] = sD[max(lr, c) + min(lr, c) * ldd]; // symmetric fill from the valid lower triangle
        d21r[k] = sD[16 + lr + c * ldd];
        d22r[k] = sD[16 + max(lr, c) + (16 + min(lr, c)) * ldd];
    }
    ldl_inverse_tile_regs(
        a,
        [&](double (&d21)[4], double (&d22)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d21[k] = d21r[k];
                d22[k] = d22r[k];
            }
        },
        w, put, flags, swork, o1, o2, xl);
}

__device__ __forceinline__ void ldl_inverse_tile(const double* __restrict__ sD, int ldd, int w, double* __restrict__ LinvOut, int* __restrict__ flags,
                                                 double* __restrict__ swork) {
    ldl_inverse_tile_put(sD, ldd, w, [LinvOut](int r, int c, double v) { LinvOut[r + 32 * c] = v; }, flags, swork);
}

// L_00^-1 for the first panel (the only elimination that is not the tail of a step kernel). One workgroup.
__global__ void __launch_bounds__(256) k_chol_first(int w, int ldz, const double* __restrict__ Z, double* __restrict__ LinvOut, int* __restrict__ flags) {
    __shared__ double sD[32 * 33];
    __shared__ double swork[LDL_SBUF];
    const int r = threadIdx.x & 31, g = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = g + 8 * k;
        sD[r + c * 33] = (r < w && c < w && r >= c) ? Z[r + (size_t)c * ldz] : ((r == c) ? 1.0 : 0.0);
    }
    __syncthreads();
    ldl_inverse_tile(sD, 33, w, LinvOut, flags, swork);
}

// ---------------------------------------------------------------------------------------------------
// K8a: build Z = [S ; T ; yTilde^T] from Sigma and the packed C blocks, exploiting the 2x3 block sparsity of C:
//   T[:, 2j:2j+2] = Sigma[:, l_j:l_j+3] C_j^T          (6 n M flops instead of 2 n^2 m)
//   S[2i:2i+2, 2j:2j+2] = C_i Sigma[l_i.., l_j..] C_j^T + delta_ij R
// Grid: x covers "row items" (t < n -> row of T, n <= t < n+M -> block row i of S, t == n+M -> the yTilde row); y = 0 is the
// first-tile row, y = 1 the statistics row (with fusion only), every further y a group of BZ_JB measurements.
// Grid row 0: its first workgroup recomputes the first 32 x 32 tile of S on its own (16 x 16 pairs of 2 x 2 blocks, one per
// thread) and eliminates it, so that the factorisation chain needs no separate first-tile launch.
//
// Measurement fusion (mf.enabled, eqf_stats_then_update): there is no k_measure / k_outlier_stats launch in front of this kernel.
//  * Every thread that needs a block C_i evaluates it itself (measure_one: the same function and inputs, therefore the same
//    bits, wherever it is evaluated) from the measurement in HBM (eqf_stage_measurement) or in the pinned host packet; the
//    workgroups with blockIdx.x == 0 store C_j, yTilde_j and the index map for later reuse (eqf_vision_update after a cancelled
//    tail, debugging).
//  * Grid row 1 computes the per-landmark outlier statistics (VIOFilter.cpp:304-334), writes them to the host packet and, if any
//    landmark is an outlier candidate, stores spec_seq into *spec_w: this kernel only writes scratch (Z, C), the kernels behind
//    it compare that word and return at once.
// The entries of Z = [S ; T ; yTilde^T], written once: k_build_Z stores them, the look-ahead kernel's half-rows can build their own rows from the same
// expressions (EQF_OPT_Z_IN_LOOKAHEAD), and the results must not differ by a bit.
// T[t, 2j + a] = Sigma[t, l_j : l_j + 3] C_j[a, :]^T
// (Round 6: the fused multiply-adds are written out. Left to the compiler, the contraction of a0 b0 + a1 b1 + a2 b2 depended on the code around the inlined copy - the
//  17 .. 32-panel prologue la_build_rows2 came out one bit away from k_build_Z in a third of W's entries.)
__device__ __forceinline__ double bz_dot3(double a0, double b0, double a1, double b1, double a2, double b2) { return __builtin_fma(a2, b2, __builtin_fma(a1, b1, a0 * b0)); }
__device__ __forceinline__ void bz_T_pair(double s0, double s1, double s2, const double (&cj)[6], double& o0, double& o1) {
    o0 = bz_dot3(s0, cj[0], s1, cj[1], s2, cj[2]);
    o1 = bz_dot3(s0, cj[3], s1, cj[4], s2, cj[5]);
}
// S[2i + a, 2j + b] = (C_i Sigma[l_i, l_j] C_j^T)[a, b] (+ the measurement variance on the diagonal); sv[3 c + r] = Sigma[l_i + r, l_j + c]
__device__ __forceinline__ void bz_S_block(const double (&ci)[6], const double (&cj)[6], const double (&sv)[9], bool same_measurement, double meas_var, double (&out)[2][2]) {
    double CS[2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double s0 = sv[3 * c], s1 = sv[3 * c + 1], s2 = sv[3 * c + 2];
        CS[0][c] = bz_dot3(ci[0], s0, ci[1], s1, ci[2], s2);
        CS[1][c] = bz_dot3(ci[3], s0, ci[4], s1, ci[5], s2);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            double v = bz_dot3(CS[a][0], cj[3 * bb], CS[a][1], cj[3 * bb + 1], CS[a][2], cj[3 * bb + 2]);
            if (same_measurement && a == bb)
                v += meas_var;
            out[a][bb] = v;
        }
}
constexpr int BZ_JB = 4; // measurements per workgroup of k_build_Z
struct MeasFuse {
    int enabled;
    int N, Ncap, chart, star;
    Cam cam;
    const double* y;     // y[2 j], y[2 j + 1]        } staged copies in HBM, or the pinned host packet
    const int* lmidx;    // landmark index of measurement j
    const double* ylm;   // the measurement by landmark (statistics row)
    const double *q0, *Qq, *Qa;
    double* out;         // pinned: statistics (3 N)
    double *C, *ytil;    // device, for reuse
    int* lmidx_dev;
    double thrAbs, thrProb;
    int* spec_w;
    int spec_seq;
};
__device__ __forceinline__ MeasOut measure_j(const MeasFuse& mf, int j, int& i_out) {
    const int i = mf.lmidx[j];
    i_out = i;
    return measure_one(mf.chart, mf.cam, ld3(mf.q0, mf.Ncap, i), ldq(mf.Qq, mf.Ncap, i), mf.Qa[i], mf.y[2 * j], mf.y[2 * j + 1], mf.star != 0,
                       mf.chart == EQVIO_COORD_INVDEPTH ? ld_cc(mf.q0, mf.Ncap, i, CC_R0) : M3{});
}
// CMEM (with FUSE, round 4): the output blocks, residuals and the index map of THIS measurement are in memory already - evaluated by the propagation kernel's observer blocks
// (EQF_OPT_MEASURE_IN_PROPAGATE) at a size where the look-ahead kernel does not build Z itself: the statistics row stays, nobody evaluates a block again (N = 500: 25.0 -> 15 us)
template <typename TS, bool FUSE, bool CMEM = false> // FUSE: measurement fusion (FUSE); a template so that neither variant carries the other's code
__global__ void __launch_bounds__(256) k_build_Z(int n, int M, int Mcap, int ld, int ldz, double meas_var, const int* __restrict__ lmidx,
                                                 const TS* __restrict__ Sig, const double* __restrict__ C, const double* __restrict__ ytil,
                                                 double* __restrict__ Z, double* __restrict__ LinvOut, int* __restrict__ flags, const int* __restrict__ spec,
                                                 int spec_seq, const MeasFuse mf, trace_t* tr) {
    trace_start(tr);
    if (spec && *spec == spec_seq)
        return; // cancelled speculative tail
    const int m = 2 * M;
    // grid rows: 0 = first tile, 1 = statistics (with fusion), then one per measurement. The two special rows have the longest
    // dependent chains (evaluation + 32 x 32 elimination; evaluation + stores to the host), so they are dispatched first.
    constexpr bool EVAL = FUSE && !CMEM; // the C blocks are evaluated in this kernel
    const int row0 = FUSE ? 2 : 1;
    if (FUSE && (int)blockIdx.y == 1) {
        // outlier statistics, one lane per landmark
        if ((int)(blockIdx.x * blockDim.x) >= mf.N)
            return;
        double abs_err = -1.0, prob_err = -1.0;
        outlier_stats_body<TS>(mf.N, mf.Ncap, ld, mf.chart, mf.cam, mf.ylm, mf.q0, mf.Qq, mf.Qa, Sig, mf.out, 0, nullptr, nullptr, nullptr, nullptr, abs_err, prob_err, false);
        if (mf.spec_w && abs_err >= 0.0 && (abs_err > mf.thrAbs || prob_err > mf.thrProb)) // the comparisons of VIOFilter.cpp:316-330 (NaN: false)
            *mf.spec_w = mf.spec_seq;
        return;
    }
    if ((int)blockIdx.y == 0) {
        if (blockIdx.x != 0)
            return;
        __shared__ double sD[32 * 33];
        __shared__ double swork[LDL_SBUF];
        __shared__ double sC16[16 * 6];
        const int i = threadIdx.x & 15, jj = threadIdx.x >> 4; // pair (i, jj) of measurements, both < 16
        double sv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // Sigma[l_i.., l_jj..], requested before the C blocks are evaluated
        if (i < M && jj < M) {
            const int li = 21 + 3 * (EVAL ? mf.lmidx[i] : lmidx[i]), lj2 = 21 + 3 * (EVAL ? mf.lmidx[jj] : lmidx[jj]);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    sv[3 * c + r] = Sig[li + r + (size_t)(lj2 + c) * ld];
        }
        if (FUSE) {
            if (threadIdx.x == 0) { // this launch's only writer of the status flags
                flags[0] = 0;
                flags[1] = 0;
                flags[3] = 0;
            }
            if (EVAL && jj == 0 && i < M) {
                int lidx;
                const MeasOut o = measure_j(mf, i, lidx);
#pragma unroll
                for (int e = 0; e < 6; ++e)
                    sC16[i * 6 + e] = o.c[e];
            }
            if (EVAL)
                __syncthreads();
        }
        double blk[2][2] = {{(i == jj) ? 1.0 : 0.0, 0.0}, {0.0, (i == jj) ? 1.0 : 0.0}};
        if (i < M && jj < M) {
            double ci[6], cj2[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                ci[e] = EVAL ? sC16[i * 6 + e] : C[e * Mcap + i];
                cj2[e] = EVAL ? sC16[jj * 6 + e] : C[e * Mcap + jj];
            }
            bz_S_block(ci, cj2, sv, i == jj, meas_var, blk); // identical expression to the S entries written to Z below
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
                sD[2 * i + a + (2 * jj + bb) * 33] = blk[a][bb];
        __syncthreads();
        ldl_inverse_tile(sD, 33, min(32, m), LinvOut, flags, swork);
        return;
    }
    // BZ_JB measurements per workgroup: with fusion one wavefront evaluates their C blocks in BZ_JB lanes at once, and a thread of
    // a block row of S evaluates its own C_i once for all of them (an evaluation costs a few thousand issue cycles per WAVE,
    // however few lanes are active: one measurement per workgroup would make the kernel VALU-bound on redundant evaluations)
    const int j0 = BZ_JB * ((int)blockIdx.y - row0);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ double sCj[BZ_JB][8];
    const bool trow = t < n, srow = t >= n && t < n + M;
    // index map first and the Sigma operands right behind it: those loads are in flight while the C blocks are being evaluated
    int lj[BZ_JB];
#pragma unroll
    for (int q = 0; q < BZ_JB; ++q) {
        const int jc = min(j0 + q, M - 1);
        lj[q] = 21 + 3 * (EVAL ? mf.lmidx[jc] : lmidx[jc]);
    }
    int li = 0;
    double sv[BZ_JB][9];
    if (trow) {
#pragma unroll
        for (int q = 0; q < BZ_JB; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                sv[q][c] = Sig[t + (size_t)(lj[q] + c) * ld];
    } else if (srow) {
        li = 21 + 3 * (EVAL ? mf.lmidx[t - n] : lmidx[t - n]);
#pragma unroll
        for (int q = 0; q < BZ_JB; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    sv[q][3 * c + r] = Sig[li + r + (size_t)(lj[q] + c) * ld];
    }
    if (EVAL) {
        if (threadIdx.x < BZ_JB && j0 + (int)threadIdx.x < M) {
            const int j = j0 + threadIdx.x;
            int lidx;
            const MeasOut o = measure_j(mf, j, lidx);
#pragma unroll
            for (int e = 0; e < 6; ++e)
                sCj[threadIdx.x][e] = o.c[e];
            sCj[threadIdx.x][6] = o.yt[0];
            sCj[threadIdx.x][7] = o.yt[1];
            if (blockIdx.x == 0) {
#pragma unroll
                for (int e = 0; e < 6; ++e)
                    mf.C[e * Mcap + j] = o.c[e];
                mf.ytil[2 * j] = o.yt[0];
                mf.ytil[2 * j + 1] = o.yt[1];
                mf.lmidx_dev[j] = lidx;
            }
        }
    } else if (threadIdx.x < BZ_JB && j0 + (int)threadIdx.x < M) {
        const int j = j0 + threadIdx.x;
#pragma unroll
        for (int e = 0; e < 6; ++e)
            sCj[threadIdx.x][e] = C[e * Mcap + j];
        sCj[threadIdx.x][6] = ytil[2 * j];
        sCj[threadIdx.x][7] = ytil[2 * j + 1];
    }
    // a block row of S needs its own C_i as well: evaluated by the thread that uses it (before the barrier: the evaluations overlap)
    double ci[6] = {0, 0, 0, 0, 0, 0};
    if (srow) {
        if (EVAL) {
            int lidx;
            const MeasOut o = measure_j(mf, t - n, lidx);
#pragma unroll
            for (int e = 0; e < 6; ++e)
                ci[e] = o.c[e];
        } else {
#pragma unroll
            for (int e = 0; e < 6; ++e)
                ci[e] = C[e * Mcap + (t - n)];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BZ_JB; ++q) {
        const int j = j0 + q;
        if (j >= M)
            break;
        double cj[6];
#pragma unroll
        for (int e = 0; e < 6; ++e)
            cj[e] = sCj[q][e];
        if (trow) {
            double o0, o1;
            bz_T_pair(sv[q][0], sv[q][1], sv[q][2], cj, o0, o1);
            Z[m + t + (size_t)(2 * j) * ldz] = o0;
            Z[m + t + (size_t)(2 * j + 1) * ldz] = o1;
        } else if (srow) {
            const int i = t - n;
            double blk[2][2];
            bz_S_block(ci, cj, sv[q], i == j, meas_var, blk);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
                    Z[2 * i + a + (size_t)(2 * j + bb) * ldz] = blk[a][bb];
        } else if (t == n + M) {
            Z[m + n + (size_t)(2 * j) * ldz] = sCj[q][6];
            Z[m + n + (size_t)(2 * j + 1) * ldz] = sCj[q][7];
        }
    }
}

constexpr int GAMMA_G = 4; // column groups of the Gamma partials computed by the last step launch
// Template flag, so that the ordinary step carries none of the optional code: GAM = last launch of the chain, produces the Gamma partials (gpart).
// (Rounds 1-3 also had a fused covariance update and two-phase steps in here; measured slower, removed in round 4: DESIGN_APPENDIX.md, git tag r04-before-prune.)
template <bool GAM>
__global__ void __launch_bounds__(256) k_chol_step(int rows, int m, int kb, int w, int ldz, double* __restrict__ Z, double* __restrict__ Wout,
                                                   const double* __restrict__ LinvIn, double* __restrict__ LinvOut, int* __restrict__ flags, int update, int nyS,
                                                   const int* __restrict__ spec, int spec_seq, double* __restrict__ gpart, int ldg, trace_t* tr) {
    trace_start(tr);
    if (spec && *spec == spec_seq)
        return; // cancelled speculative tail
    const int c0 = kb + w;
    if (GAM && (int)blockIdx.y >= nyS) {
        // Gamma partials (last step of the unfused chain only; c0 == m): the W columns [0, kb) were published by the earlier
        // launches, so Gamma = W z over them can run in the shadow of this step. Grid row nyS + g takes the columns
        // p = g (mod GAMMA_G); workgroup x takes the 32 rows of W starting at 32 x; its 8 lane groups interleave the columns.
        // The last panel's share is added by the workgroups that compute it (below); k_lift sums the GAMMA_G + 1 partials.
        __shared__ double sp[256];
        const int g = (int)blockIdx.y - nyS;
        const int r = threadIdx.x & 31, seg = threadIdx.x >> 5;
        const int n = rows - 1 - m;
        const int wr = 32 * (int)blockIdx.x + r;
        const double* Wr = Wout + m + min(wr, n - 1);
        const double* z = Wout + m + n;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (int p0 = g + GAMMA_G * seg; p0 < kb; p0 += 12 * 8 * GAMMA_G) {
            double wv[12], zv[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int p = p0 + q * 8 * GAMMA_G;
                const int pc = min(p, kb - 1);
                wv[q] = Wr[(size_t)pc * ldz];
                zv[q] = z[(size_t)pc * ldz] * (p < kb ? 1.0 : 0.0);
            }
#pragma unroll
            for (int q = 0; q < 12; q += 4) {
                s0 = fma(wv[q], zv[q], s0);
                s1 = fma(wv[q + 1], zv[q + 1], s1);
                s2 = fma(wv[q + 2], zv[q + 2], s2);
                s3 = fma(wv[q + 3], zv[q + 3], s3);
            }
        }
        sp[threadIdx.x] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (seg == 0 && wr < n) // fixed order: deterministic
            gpart[(size_t)g * ldg + wr] = ((sp[r] + sp[32 + r]) + (sp[64 + r] + sp[96 + r])) + ((sp[128 + r] + sp[160 + r]) + (sp[192 + r] + sp[224 + r]));
        return;
    }
    const int ilim = rows, jlim = m;
    const int i0 = c0 + blockIdx.x * 32, j0 = c0 + blockIdx.y * 32;
    if (update && (i0 + 31 < j0))
        return; // strictly upper tile of the symmetric part: never read
    __shared__ double sLinv[32 * CH_LDP];
    __shared__ double sPI[32 * CH_LDP];
    __shared__ double sPJ[32 * CH_LDP];
    __shared__ double swork[LDL_SBUF];
    const int tid = threadIdx.x;
    const int r = tid & 31, g = tid >> 5;
    const int wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const bool diag_tile = update && (i0 == j0);
    const bool needJ = update && !diag_tile;
    // 0. issue every global load up front: L^-1, the panel rows of I and J in MFMA operand layout
    //    (Zp[row][p], p = 4 st + lk) and the output tile
    double lv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        lv[k] = LinvIn[r + 32 * (g + 8 * k)];
    const int ihP = wave & 1;
    double opI[8], opJ[8];
    {
        const int rowI = i0 + 16 * ihP + lr, rowJ = j0 + 16 * ihP + lr;
        const int rowIc = min(rowI, ilim - 1), rowJc = min(rowJ, jlim - 1);
        const double zI = rowI < ilim ? 1.0 : 0.0, zJ = (needJ && rowJ < jlim) ? 1.0 : 0.0;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int p = 4 * st + lk;
            const int pc = min(p, w - 1);
            const double zp = p < w ? 1.0 : 0.0;
            opI[st] = Z[rowIc + (size_t)(kb + pc) * ldz] * (zI * zp);
            opJ[st] = needJ ? Z[rowJc + (size_t)(kb + pc) * ldz] * (zJ * zp) : 0.0;
        }
    }
    const bool gam_last = GAM && !update; // this launch also produces the last panel's share of Gamma
    const double yv = (gam_last && tid < 32) ? Z[(rows - 1) + (size_t)(kb + min(tid, w - 1)) * ldz] : 0.0;
    const int ihU = wave & 1, jhU = wave >> 1;
    double zt[4];
    {
        const int i = min(i0 + 16 * ihU + lr, ilim - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = min(j0 + 16 * jhU + lk + 4 * q, jlim - 1);
            zt[q] = update ? Z[i + (size_t)j * ldz] : 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        sLinv[r + (g + 8 * k) * CH_LDP] = lv[k];
    if (gam_last && tid < 32)
        swork[32 + tid] = yv;
    __syncthreads();
    // 2. P = Zpanel * Linv^T : P[i][c] = sum_p Zp[i][p] Linv[c][p]; wave -> 16x16 sub-tile (ih, ch) of P_I and of P_J
    {
        const int ch = wave >> 1;
        d4 accI = {0, 0, 0, 0}, accJ = {0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const double b = sLinv[16 * ch + lr + (4 * st + lk) * CH_LDP]; // J operand: Linv[c][p]
            accI = __builtin_amdgcn_mfma_f64_16x16x4f64(b, opI[st], accI, 0, 0, 0);
            if (needJ)
                accJ = __builtin_amdgcn_mfma_f64_16x16x4f64(b, opJ[st], accJ, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sPI[16 * ihP + lr + (16 * ch + lk + 4 * q) * CH_LDP] = accI[q]; // P[i][c], c = 16 ch + lk + 4 q
            if (needJ)
                sPJ[16 * ihP + lr + (16 * ch + lk + 4 * q) * CH_LDP] = accJ[q];
        }
    }
    __syncthreads();
    // 4. final W / z rows of this panel
    if (blockIdx.y == 0) {
        const int row = i0 + r;
        if (row < rows && row >= m) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = g + 8 * k;
                if (c < w)
                    Wout[row + (size_t)(kb + c) * ldz] = sPI[r + c * CH_LDP];
            }
        }
    }
    if (gam_last) {
        // Gamma share of this panel: z_c = sum_p yTilde[p] Linv[c][p], then P_I z for the 32 rows of this workgroup
        if (tid < 32) {
            double z = 0.0;
            for (int p2 = 0; p2 < w; ++p2)
                z = fma(swork[32 + p2], sLinv[tid + p2 * CH_LDP], z);
            swork[tid] = z;
        }
        __syncthreads();
        const int wr = i0 + tid - m;
        if (tid < 32 && wr >= 0 && wr < rows - 1 - m) {
            double gsum = 0.0;
            for (int c = 0; c < w; ++c)
                gsum = fma(sPI[tid + c * CH_LDP], swork[c], gsum);
            gpart[(size_t)GAMMA_G * ldg + wr] = gsum;
        }
    }
    if (!update)
        return;
    // 3. Z[I, J] -= P_I P_J^T : wave -> 16x16 sub-tile (ihU, jhU), K = 32
    const bool next_diag = (blockIdx.x == 0 && blockIdx.y == 0);
    {
        const double* pj = diag_tile ? sPI : sPJ;
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int c = 4 * st + lk;
            const double a = sPI[16 * ihU + lr + c * CH_LDP];
            const double b = pj[16 * jhU + lr + c * CH_LDP];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);
        }
        const int i = i0 + 16 * ihU + lr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + 16 * jhU + lk + 4 * q;
            const double v = zt[q] - acc[q];
            if (i < rows && j < m && (!next_diag || i >= m))
                Z[i + (size_t)j * ldz] = v; // (the next diagonal tile itself stays on chip: nobody reads it from Z again)
            if (next_diag)
                sPJ[16 * ihU + lr + (16 * jhU + lk + 4 * q) * CH_LDP] = v; // keep the updated next-diagonal tile on chip
        }
    }
    if (!next_diag)
        return;
    // 5. eliminate the next diagonal tile D_{k+1} = Z[c0 : c0 + w2, c0 : c0 + w2] (kept in sPJ) and publish its inverse factor
    __syncthreads();
    const int w2 = min(32, m - c0);
    // identity padding outside the w2 x w2 block (rows / cols beyond m belong to T, not to S)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = g + 8 * k;
        if (r >= w2 || c >= w2)
            sPJ[r + c * CH_LDP] = (r == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    ldl_inverse_tile(sPJ, CH_LDP, w2, LinvOut, flags, swork);
}

// ---------------------------------------------------------------------------------------------------
// K2 / K2b, the propagation kernel (described above; it stands behind the measurement helpers because with EQF_OPT_Z_IN_PROPAGATE its tiles build Z themselves)
// TS = storage type of Sigma (double, or float for EQF_OPT_SIGMA_FP32 = 2): loads convert to double, stores round.
template <typename TS, bool FUSED, bool SYM = false, bool GH = false> // FUSED: fused assembly; SYM: lower tiles only, mirrored (an instantiation of its own: as a run-time
                                                      // branch it cost the N = 200 frame 0.6 us); GH: landmarks leave / are created inside this launch (GatherArgs) - an
                                                      // instantiation of its own as well: carried by every launch, that code cost the steady frame's kernel 0.7 us
__global__ void __launch_bounds__(PROP_T) k_propagate_main(int N, int Ncap, int ld, RiccatiArgs ra, const Common* __restrict__ cm,
                                                        const TS* __restrict__ Sig, TS* __restrict__ Sout, const double* __restrict__ Al,
                                                        const double* __restrict__ Bl, int nT, int tpw, const ObsSteps obs, int obs_k,
                                                        const double* __restrict__ q0, double* __restrict__ Qq, double* __restrict__ Qa, int nObs, const StageArgs sg,
                                                        trace_t* tr, const FuseArgs fa, const MeasEval me, const GatherArgs ga_in) {
    GatherArgs ga; // GH = false: constants the compiler folds away (nothing removed, nobody held)
    if constexpr (GH)
        ga = ga_in;
    else {
        ga.n = 0, ga.nprop = N, ga.held = nullptr, ga.st_in = nullptr, ga.st_out = nullptr;
    }
    trace_start(tr);
    const double dt = ra.dt;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    // SYM (large N, chosen by the host): only the lower triangle of landmark tiles is computed, the upper one written as its mirror image - half the
    // tile workgroups (N = 500: 1024 -> 528, one per CU at a time: 45 -> 32 us). Up to 16 tiles per side all tiles fit the chip in one round and the full
    // form is 1 us faster (the block row's 21 strip columns spread over more workgroups, no strided mirror stores).
    constexpr bool sym = SYM;
    constexpr int nStrip = 0;
    // Round 5, SYM only: tpw > 1 consecutive tiles of ONE block row per workgroup. A tile is a dependent chain of ~10 us (loads -> assembly -> G_i -> blocks -> stores) and this
    // kernel's 168 VGPRs allow one workgroup of 12 waves per CU: 528 tiles (N = 500) ran as three rounds of that chain. The host picks the smallest tpw with which all tile
    // workgroups are resident at once; the i side (Sigma strips, assembly, G_i) is then formed once per workgroup. Same sums per entry: bit-identical for every tpw.
    int nTiles = sym ? nT * (nT + 1) / 2 : nT * nT;
    if (sym && tpw > 1) {
        nTiles = 0;
        for (int r_ = 0; r_ < nT; ++r_)
            nTiles += (r_ + tpw) / tpw;
    }
    __shared__ double sm[2 * PT * (63 + 36 + 9 + 9) + 63 * PT + 12 * 21 + 8];
    __shared__ double sJ1[PT * (63 + 36 + 9 + 9)]; // the j side of a workgroup's odd tiles (several tiles per workgroup: tile it + 1 is staged while tile it computes)
    if (b > nTiles + nStrip + nObs) {
        // Staging block (eqf_stage_measurement): the coming frame's measurement moves from the pinned host packet to HBM while Sigma
        // is being propagated, so that the update's first kernel finds it next to the state instead of across PCIe.
        for (int t = tid; t < 2 * sg.M; t += PROP_T)
            sg.y_d[t] = sg.y_h[t];
        for (int t = tid; t < sg.M; t += PROP_T)
            sg.idx_d[t] = sg.idx_h[t];
        for (int t = tid; t < 3 * N; t += PROP_T) {
            const int pl = t / N, i = t - pl * N;
            sg.ylm_d[pl * Ncap + i] = sg.ylm_h[pl * Ncap + i];
        }
        return;
    }
    if (b > nTiles + nStrip) {
        // Observer blocks (eqf_propagate_fast): the landmark part of the frame's observer steps rides along with the Sigma
        // propagation. This kernel touches Sigma / Al / Bl only, the assembly kernel before it has already read Q and the
        // statistics kernel after it wants the new Q: in-stream order gives all three, no second stream, no events.
        // With fused assembly the tiles of THIS launch read Q, so the observer blocks write the other (Qq, Qa) buffer (the host
        // flips to it after the launch; q0 and its chart constants live in a buffer of their own and stay where they are).
        const int i = (b - (nTiles + nStrip + 1)) * PROP_T + tid;
        // The steps' terms go from the argument segment to LDS once: read from the argument segment step by step, every step of the chain - the longest dependent path
        // of this kernel - waited for a scalar load of its own (round 4)
        static_assert(sizeof(ObsStep) % 8 == 0 && kObsChunk * sizeof(ObsStep) <= sizeof(sm), "the steps are copied as doubles into the tile buffer");
        if (FUSED) {
            for (int t = tid; t < obs_k * (int)(sizeof(ObsStep) / 8); t += PROP_T)
                sm[t] = reinterpret_cast<const double*>(obs.s)[t];
            __syncthreads();
        }
        const ObsStep* steps_lds = reinterpret_cast<const ObsStep*>(sm);
        if (i < N) {
            if (FUSED) {
                const bool held = i >= ga.nprop;
                const bool synth = held && ga.held != nullptr; // a held landmark that is not in memory yet: this lane creates it
                const int io = held ? i : gather_old(ga, i); // where landmark i sits in the buffers this launch reads (its origin point and chart constants are moved by the sensor block's workgroup)
                V3 p0;
                Qt q;
                double a_;
                if (synth) {
                    const double* hp = ga.held + 1 + 3 * (i - ga.nprop); // zero-copy across PCIe
                    p0 = V3{hp[0], hp[1], hp[2]};
                    q = Qt{1.0, 0.0, 0.0, 0.0};
                    a_ = 1.0;
                } else {
                    p0 = ld3(q0, Ncap, io);
                    q = ldq(Qq, Ncap, io);
                    a_ = Qa[io];
                }
                double yu = 0.0, yv = 0.0;
                int jm = -1;
                if (me.on) { // requested before the chain: a zero-copy read across PCIe
                    yu = me.ylm[i], yv = me.ylm[Ncap + i];
                    jm = (int)me.ylm[2 * Ncap + i];
                }
                M3 r0_new{};
                if (synth) { // the planes k_append_inplace would have written (the only other place the chart constants are computed)
                    double* st = ga.st_out;
                    st[i] = p0.x;
                    st[Ncap + i] = p0.y;
                    st[2 * (size_t)Ncap + i] = p0.z;
                    double* cc = st + (size_t)CC_OFF * Ncap;
                    store_chart_constants(cc, Ncap, i, p0.x, p0.y, p0.z, &r0_new);
                }
                if (!held)
                    observer_chain(steps_lds, obs_k, p0, q, a_);
                fa.Qqo[i] = q.w;
                fa.Qqo[Ncap + i] = q.x;
                fa.Qqo[2 * Ncap + i] = q.y;
                fa.Qqo[3 * Ncap + i] = q.z;
                fa.Qao[i] = a_;
                if (me.on && jm >= 0) {
                    // (the chain's last products must not be contracted into the evaluation's first sums: the same bits as an evaluation from the stored element)
                    asm volatile("" : "+v"(q.w), "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(a_));
                    const MeasOut o = measure_one(fa.chart, me.cam, p0, q, a_, yu, yv, me.star != 0, fa.chart == EQVIO_COORD_INVDEPTH ? (synth ? r0_new : ld_cc(q0, Ncap, io, CC_R0)) : M3{});
#pragma unroll
                    for (int e = 0; e < 6; ++e)
                        me.C[e * me.Mcap + jm] = o.c[e];
                    me.ytil[2 * jm] = o.yt[0];
                    me.ytil[2 * jm + 1] = o.yt[1];
                    me.lmidx_dev[jm] = i;
                }
            } else {
                observer_landmark(obs.s, Ncap, obs_k, i, q0, Qq, Qa, Qq, Qa);
            }
        }
        return;
    }
    __shared__ double sSens[21 * 33]; // per strip column of this workgroup: row c of A_ss (21) | row c of B_s (12)
    __shared__ double sSens1[21 * 33]; // ... of an odd tile
    __shared__ double s_cm[66];
    if (FUSED) {
        for (int t = tid; t < 66; t += PROP_T)
            s_cm[t] = fa.ck.lm[t];
        __syncthreads();
    }
    if (b < nTiles) {
        // tile (bi, bj), bi >= bj, of the lower triangle in row-major order: b = bi (bi + 1) / 2 + bj
        int bi = b % nT, bj0 = b / nT, ntile = 1;
        if (sym && tpw > 1) {
            int base = 0;
            bi = 0;
            for (;;) {
                const int w_ = (bi + tpw) / tpw; // workgroups of block row bi
                if (b < base + w_)
                    break;
                base += w_;
                ++bi;
            }
            bj0 = (b - base) * tpw;
            ntile = min(tpw, bi + 1 - bj0);
        } else if (sym) {
            bi = (int)((sqrtf(8.0f * (float)b + 1.0f) - 1.0f) * 0.5f);
            while (bi * (bi + 1) / 2 > b)
                --bi;
            while ((bi + 1) * (bi + 2) / 2 <= b)
                ++bi;
            bj0 = b - bi * (bi + 1) / 2;
        }
        const int nb = sym ? bi + 1 : nT; // tiles of this block row: they share its 21 strip columns
        const int Nprop = FUSED ? ga.nprop : N;                      // landmarks that are propagated; the others are held (GatherArgs)
        const int Nmem = (FUSED && ga.held) ? Nprop : N;             // landmarks whose rows of Sigma exist in the buffer this launch reads
        const double held_var = (FUSED && ga.held && Nprop < N) ? ga.held[0] : 0.0; // (zero-copy across PCIe, consumed at the very end of the tile)
        // per-i arrays: G (63), Fls (36), D (9), Bl (9) ; per-j arrays: Ssj (63), Fls (36), D (9), Bl (9). layout [e][PT]
        // The j side exists twice (parity of the tile inside the workgroup): with several tiles per workgroup the loads and the assembly of tile it + 1 are issued
        // in front of the arithmetic of tile it and are in flight during it - one barrier per tile.
        double* sGi = sm;
        double* sFi = sGi + 63 * PT;
        double* sDi = sFi + 36 * PT;
        double* sBi = sDi + 9 * PT;
        double* sJ0 = sBi + 9 * PT;            // j side, parity 0: Ssj (63) | Fls (36) | D (9) | Bl (9)
        double* sSi = sJ0 + 117 * PT;          // Sigma[k][l_i + c'] at [(k*3 + c') * PT + x]
        double* sSs = sSi + 63 * PT;           // Sigma_ss[al_col(e)][k] at [e * 21 + k], e < 12
        const int r = tid / (PT * PT), tp = tid % (PT * PT); // main part: lane = (output row r, landmark pair (ti, tj))
        const int ti = tp % PT, tj = tp / PT;
        // stage A of tile `it`: everything its arithmetic reads from LDS (the i side with the workgroup's first tile)
        auto stage = [&](const int it) {
            const bool first = it == 0;
            const int bj = bj0 + it;
            double* sSj = (it & 1) ? sJ1 : sJ0;
            double* sFj = sSj + 63 * PT;
            double* sDj = sFj + 36 * PT;
            double* sBj = sDj + 9 * PT;
            double* sSn = (it & 1) ? sSens1 : sSens;
            for (int t = tid; t < 63 * PT; t += PROP_T) {
                const int e = t / PT, x = t % PT;
                const int i = bi * PT + x, j = bj * PT + x;
                // Sigma[k][l + c'] with e = k*3 + c'
                const int kk = e / 3, cc = e % 3;
                // (a held landmark that is not in memory yet: its cross-covariances with the sensor states are zeros)
                if (first)
                    sSi[t] = i < Nmem ? Sig[kk + (size_t)(21 + 3 * (i < Nprop ? gather_old(ga, i) : i) + cc) * ld] : 0.0;
                sSj[t] = j < Nmem ? Sig[kk + (size_t)(21 + 3 * (j < Nprop ? gather_old(ga, j) : j) + cc) * ld] : 0.0;
            }
            if (first && tid < 12 * 21)
                sSs[tid] = Sig[al_col(tid / 21) + (size_t)(tid % 21) * ld];
            // the strip columns this tile writes: c = bj, bj + nb, ... ; their rows of the sensor blocks go to LDS
            const int ncol = bj < 21 ? (21 - bj + nb - 1) / nb : 0;
            for (int t = PROP_T - 1 - tid; t < ncol * 33; t += PROP_T) { // taken from the top of the workgroup: the first lanes assemble
                const int m_ = t / 33, e = t % 33;
                const int c = bj + nb * m_;
                sSn[t] = e < 21 ? (FUSED ? sensor_Ass_entry(fa.ck, c * 21 + e) : cm->Ass[c * 21 + e]) : (FUSED ? sensor_Bs_entry(fa.ck, c * 12 + (e - 21)) : cm->Bs[c * 12 + (e - 21)]);
            }
            if (FUSED) {
                // Three wavefronts assemble: lanes 0..2PT-1 of waves 0, 1, 2 take part 0, 1, 2 (assemble_landmark) of the PT i-landmarks and
                // the PT j-landmarks; the other threads are loading Sigma meanwhile.
                const int part = tid >> 6, lane_ = tid & 63;
                if (part < 3 && lane_ < 2 * PT && (first || lane_ >= PT)) {
                    const bool isj = lane_ >= PT;
                    const int x = lane_ % PT;
                    const int l = (isj ? bj : bi) * PT + x;
                    double* dF = isj ? sFj : sFi;
                    double* dD = isj ? sDj : sDi;
                    double* dB = isj ? sBj : sBi;
                    double al[45], bl[9];
#pragma unroll
                    for (int e = 0; e < 45; ++e)
                        al[e] = 0.0;
#pragma unroll
                    for (int e = 0; e < 9; ++e)
                        bl[e] = 0.0;
                    const bool in = l < Nprop; // (a held landmark assembles like the padding of a ragged tile, except for the identity in D below)
                    const bool heldl = l >= Nprop && l < N;
                    const bool ind = fa.chart == EQVIO_COORD_INVDEPTH;
                    const int lc = in ? gather_old(ga, l) : 0;
                    const V3 p0_ = ld3(q0, Ncap, lc);
                    const Qt q_ = ldq(Qq, Ncap, lc);
                    const double a_ = Qa[lc];
                    const M3 e2i = ind ? ld_cc(q0, Ncap, lc, CC_E2I) : M3{};
                    if (part == 0) {
                        if (in)
                            assemble_landmark<0>(s_cm, fa.chart, p0_, q_, a_, e2i, M3{}, al, bl);
#pragma unroll
                        for (int rr = 0; rr < 3; ++rr) {
#pragma unroll
                            for (int c = 0; c < 6; ++c)
                                dF[(rr * 12 + c) * PT + x] = in ? dt * al[rr * 15 + c] : 0.0;
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                dB[(rr * 3 + c) * PT + x] = bl[rr * 3 + c];
                        }
                    } else if (part == 1) {
                        if (in)
                            assemble_landmark<1>(s_cm, fa.chart, p0_, q_, a_, e2i, M3{}, al, bl);
#pragma unroll
                        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                            for (int c = 6; c < 12; ++c)
                                dF[(rr * 12 + c) * PT + x] = in ? dt * al[rr * 15 + c] : 0.0;
                    } else {
                        if (in)
                            assemble_landmark<2>(s_cm, fa.chart, p0_, q_, a_, e2i, ind ? ld_cc(q0, Ncap, lc, CC_I2E) : M3{}, al, bl);
#pragma unroll
                        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                dD[(rr * 3 + c) * PT + x] = in ? dt * al[rr * 15 + 12 + c] + ((rr == c) ? 1.0 : 0.0) : ((heldl && rr == c) ? 1.0 : 0.0);
                    }
                }
            } else {
                for (int t = tid; t < 36 * PT; t += PROP_T) {
                    const int e = t / PT, x = t % PT;
                    const int rr = e / 12, c = e % 12;
                    const int i = bi * PT + x, j = bj * PT + x;
                    if (first)
                        sFi[t] = i < N ? dt * Al[(rr * 15 + c) * Ncap + i] : 0.0;
                    sFj[t] = j < N ? dt * Al[(rr * 15 + c) * Ncap + j] : 0.0;
                }
                for (int t = tid; t < 9 * PT; t += PROP_T) {
                    const int e = t / PT, x = t % PT;
                    const int rr = e / 3, c = e % 3;
                    const int i = bi * PT + x, j = bj * PT + x;
                    const double eye = (rr == c) ? 1.0 : 0.0;
                    if (first) {
                        sDi[t] = i < N ? dt * Al[(rr * 15 + 12 + c) * Ncap + i] + eye : 0.0;
                        sBi[t] = i < N ? Bl[e * Ncap + i] : 0.0;
                    }
                    sDj[t] = j < N ? dt * Al[(rr * 15 + 12 + c) * Ncap + j] + eye : 0.0;
                    sBj[t] = j < N ? Bl[e * Ncap + j] : 0.0;
                }
            }
        };
        stage(0);
        __syncthreads();
        // G_i[r][k] = sum_e (dt A_ls_i)[r][e] Sigma_ss[al_col(e)][k] + sum_c' (I + dt A_qi)[r][c'] Sigma[l_i + c'][k]
        for (int t = tid; t < 63 * PT; t += PROP_T) {
            const int e = t / PT, x = t % PT;
            const int rr = e / 21, k = e % 21;
            double g = 0.0;
#pragma unroll
            for (int q = 0; q < 12; ++q)
                g += sFi[(rr * 12 + q) * PT + x] * sSs[q * 21 + k];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
                g += sDi[(rr * 3 + cc) * PT + x] * sSi[(k * 3 + cc) * PT + x];
            sGi[t] = g;
        }
        __syncthreads();
        for (int it = 0; it < ntile; ++it) {
            const int bj = bj0 + it;
            const double* sSj = (it & 1) ? sJ1 : sJ0;
            const double* sFj = sSj + 63 * PT;
            const double* sDj = sFj + 36 * PT;
            const double* sBj = sDj + 9 * PT;
            const double* sSn = (it & 1) ? sSens1 : sSens;
            const int ncol = bj < 21 ? (21 - bj + nb - 1) / nb : 0;
            const int i = bi * PT + ti, j = bj * PT + tj;
            const bool mine = !(i >= N || j >= N || (sym && bi == bj && i < j)); // (a diagonal tile: the pairs above the diagonal are mirrors too)
            const int li = 21 + 3 * i, lj = 21 + 3 * j;
            // Sigma_ij: requested first ...
            double Sij[3][3];
            if (mine) {
                if (i < Nmem && j < Nmem) {
                    const int lio = 21 + 3 * (i < Nprop ? gather_old(ga, i) : i), ljo = 21 + 3 * (j < Nprop ? gather_old(ga, j) : j);
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            Sij[k][c] = Sig[lio + k + (size_t)(ljo + c) * ld];
                } else { // a held landmark that is not in memory yet: its own block is (variance) I, everything else of its rows and columns zero
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            Sij[k][c] = (i == j && k == c) ? held_var : 0.0;
                }
            }
            // ... then the next tile's stage (other parity: its last readers passed the barrier at the end of tile it - 1) ...
            if (it + 1 < ntile)
                stage(it + 1);
            {
                // Landmark-sensor strips of the PT i-landmarks: Sigma'[l_i + r][c] = sum_k G_i[r][k] Fss[c][k] + dt sum_q Bl_i[r][q] Qd[q] Bs[c][q].
                // G_i is in LDS here anyway; the nT tiles of this block row share the 21 columns (c = bj, bj + nT, ...), at most a few
                // outputs per tile. Same sums, in the same order, as a separate strip pass would evaluate.
                for (int t = tid; t < PT * 3 * ncol; t += PROP_T) {
                    const int x = t % PT, rc = t / PT;
                    const int rr = rc % 3, m_ = rc / 3, c = bj + nb * m_;
                    const int ii = bi * PT + x;
                    if (ii < N) {
                        double sacc = 0;
                        for (int k = 0; k < 21; ++k) {
                            const double f = dt * sSn[m_ * 33 + k] + (k == c ? 1.0 : 0.0);
                            sacc += sGi[(rr * 21 + k) * PT + x] * f;
                        }
                        double bq = 0;
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            bq += sBi[(rr * 3 + q) * PT + x] * ra.Qd[q] * sSn[m_ * 33 + 21 + q];
                        sacc += dt * bq;
                        const int lii = 21 + 3 * ii;
                        Sout[lii + rr + (size_t)c * ld] = sacc;
                        Sout[c + (size_t)(lii + rr) * ld] = sacc;
                    }
                }
            }
            // ... and this tile's blocks: three lanes share a 3x3 block, each produces one row of it. The per-element sums run in the same order as a
            // one-lane-per-block version would (results are bit-identical to it).
            if (mine) {
                // E[r][:] = (Fls_i Sigma_sj + D_i Sigma_ij)[r][:]
                double E[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double s = 0;
#pragma unroll
                    for (int e = 0; e < 12; ++e)
                        s += sFi[(r * 12 + e) * PT + ti] * sSj[(al_col(e) * 3 + c) * PT + tj];
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        s += sDi[(r * 3 + k) * PT + ti] * Sij[k][c];
                    E[c] = s;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double s = 0;
#pragma unroll
                    for (int e = 0; e < 12; ++e)
                        s += sGi[(r * 21 + al_col(e)) * PT + ti] * sFj[(c * 12 + e) * PT + tj];
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        s += E[k] * sDj[(c * 3 + k) * PT + tj];
                    double bq = 0;
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        bq += sBi[(r * 3 + q) * PT + ti] * ra.Qd[q] * sBj[(c * 3 + q) * PT + tj];
                    s += dt * bq;
                    if (i == j && r == c && i < Nprop)
                        s += dt * ra.Pd[7];
                    Sout[li + r + (size_t)(lj + c) * ld] = s;
                    if (sym && i != j)
                        Sout[lj + c + (size_t)(li + r) * ld] = s; // Sigma'_ji = Sigma'_ij^T: exactly symmetric between landmark blocks, half the tiles
                }
            }
            if (it + 1 < ntile)
                __syncthreads(); // the next tile's stage is complete, this tile's readers are done with their parity
        }
        return;
    }
    // sensor-sensor block
    {
        if (FUSED && ga.n) {
            // EQF_OPT_GATHER_IN_PROPAGATE: the origin points and chart constants of the surviving landmarks move to the other buffer (everything else of a landmark is
            // rewritten by this kernel anyway); this workgroup has the shortest chain of the launch
            const int Np = ga.nprop;
            for (int t = tid; t < (CC_OFF + CC_PLANES) * Np; t += PROP_T) {
                const int pl = t / Np, i = t - pl * Np;
                ga.st_out[(size_t)pl * Ncap + i] = ga.st_in[(size_t)pl * Ncap + gather_old(ga, i)];
            }
        }
        double* sF = sm;        // 441
        double* sS = sm + 441;  // 441
        double* sT = sm + 882;  // 441  (F Sigma_ss)
        double* sBs = sm + 1323; // 252 (fused assembly: B_s expanded from the compact terms)
        for (int t = tid; t < 441; t += PROP_T) {
            const int r = t / 21, c = t % 21;
            sF[t] = dt * (FUSED ? sensor_Ass_entry(fa.ck, t) : cm->Ass[t]) + (r == c ? 1.0 : 0.0);
            sS[t] = Sig[r + (size_t)c * ld];
        }
        for (int t = tid; t < 252; t += PROP_T)
            sBs[t] = FUSED ? sensor_Bs_entry(fa.ck, t) : cm->Bs[t];
        __syncthreads();
        for (int t = tid; t < 441; t += PROP_T) {
            const int r = t / 21, c = t % 21;
            double s = 0;
            for (int k = 0; k < 21; ++k)
                s += sF[r * 21 + k] * sS[k * 21 + c];
            sT[t] = s;
        }
        __syncthreads();
        for (int t = tid; t < 441; t += PROP_T) {
            const int r = t / 21, c = t % 21;
            double s = 0;
            for (int k = 0; k < 21; ++k)
                s += sT[r * 21 + k] * sF[c * 21 + k];
            double bq = 0;
            for (int q = 0; q < 12; ++q)
                bq += sBs[r * 12 + q] * ra.Qd[q] * sBs[c * 12 + q];
            s += dt * bq;
            if (r == c)
                s += dt * ra.Pd[r / 3];
            Sout[r + (size_t)c * ld] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// fp64 MFMA tile core. v_mfma_f64_16x16x4_f64: lane l supplies A[row = l&15][k = l>>4] and
// B[k = l>>4][col = l&15]; the 4 results of lane l are D[row = (l>>4) + 4r][col = l&15]
// (cdna_hip_programming.md §3). We feed the "J" operand as A and the "I" operand as B so that a lane's
// results are C[i = i0 + (l&15)][j = j0 + (l>>4) + 4r]; both operand loads P[x0 + (l&15) + (k0 + (l>>4)) * ld]
// are 128-byte contiguous per k, straight from L2 (the working set is L2 / Infinity-Cache resident).
//
// One workgroup (4 waves) owns one 32x32 output tile; the K range is split across the 4 waves (wave w takes the
// k4-steps congruent to w mod 4) and the partial tiles are reduced through LDS in a fixed order (deterministic).
// Out-of-range operand rows / k are clamped for the load and zeroed by a multiplier so every load is unconditional.

#ifndef EQF_SYRK_UNROLL
#define EQF_SYRK_UNROLL 4 // k-steps of a wave whose operand loads are in flight at once (mfma_tile32_splitk)
#endif
struct TileRed {
    double v[4]; // element e of lane t: (i = t & 31, j = (t >> 5) + 8 e)
};
// If zvec != nullptr the workgroup also returns, in gv_out (valid in lanes 0..31 of the workgroup), the GEMV by-product
// g[i0 + t] = sum_k P[i0 + t][k] * zvec[k * ldzv] from the operand values it loads anyway.
template <bool WITH_GEMV, int NW = 4>
__device__ __forceinline__ TileRed mfma_tile32_splitk(const double* __restrict__ P, int ldp, int i0, int rowsP, const double* __restrict__ Qm, int ldq, int j0,
                                                      int rowsQ, int K, double* __restrict__ sred /* 4*1024 doubles */, const double* __restrict__ zvec = nullptr,
                                                      int ldzv = 0, double* gv_out = nullptr) {
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int ia = min(i0 + lr, rowsP - 1), ib = min(i0 + 16 + lr, rowsP - 1);
    const int ja = min(j0 + lr, rowsQ - 1), jb = min(j0 + 16 + lr, rowsQ - 1);
    const double zia = (i0 + lr < rowsP) ? 1.0 : 0.0, zib = (i0 + 16 + lr < rowsP) ? 1.0 : 0.0;
    const double zja = (j0 + lr < rowsQ) ? 1.0 : 0.0, zjb = (j0 + 16 + lr < rowsQ) ? 1.0 : 0.0;
    d4 acc00 = {0, 0, 0, 0}, acc10 = acc00, acc01 = acc00, acc11 = acc00;
    double ga = 0.0, gb = 0.0;
    const int nsteps = (K + 3) >> 2;
#pragma unroll EQF_SYRK_UNROLL
    for (int st = wave; st < nsteps; st += NW) {
        const int kk = 4 * st + lk;
        const int kc = min(kk, K - 1);
        const double zk = kk < K ? 1.0 : 0.0;
        const double pa = P[ia + (size_t)kc * ldp] * (zia * zk);
        const double pb = P[ib + (size_t)kc * ldp] * (zib * zk);
        const double qa = Qm[ja + (size_t)kc * ldq] * zja;
        const double qb = Qm[jb + (size_t)kc * ldq] * zjb;
        if (WITH_GEMV) {
            const double zv = zvec[(size_t)kc * ldzv];
            ga = fma(pa, zv, ga);
            gb = fma(pb, zv, gb);
        }
        acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(qa, pa, acc00, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(qa, pb, acc10, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(qb, pa, acc01, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(qb, pb, acc11, 0, 0, 0);
    }
    // The NW partial tiles are summed four at a time through 4 x 1024 doubles of LDS (NW = 8: two rounds; 64 KB for all eight at once allowed two
    // workgroups per CU, 32 KB allows four). Same additions in the same order as one pass over all of them: sum = 0; sum += (p0 + p1) + (p2 + p3); ...
    TileRed out;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        out.v[e] = 0.0;
#pragma unroll
    for (int q0 = 0; q0 < NW; q0 += 4) {
        if (q0)
            __syncthreads(); // the readers of the previous round are done
        if (wave >= q0 && wave < q0 + 4) {
            double* mine = sred + (wave - q0) * 1024;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = lk + 4 * r;
                mine[lr + 32 * j] = acc00[r];
                mine[16 + lr + 32 * j] = acc10[r];
                mine[lr + 32 * (16 + j)] = acc01[r];
                mine[16 + lr + 32 * (16 + j)] = acc11[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = (tid & 31) + 32 * (((tid & 255) >> 5) + 8 * e); // threads >= 256 (NW = 8) duplicate the first 256
            out.v[e] += (sred[idx] + sred[1024 + idx]) + (sred[2048 + idx] + sred[3072 + idx]); // fixed order: deterministic
        }
    }
    if (WITH_GEMV) {
        // partial sums: per wave, per lk group (4), rows 0..15 (ga) and 16..31 (gb): reduce 16 partials per row via LDS
        __syncthreads();
        sred[(wave * 4 + lk) * 32 + lr] = ga;
        sred[(wave * 4 + lk) * 32 + 16 + lr] = gb;
        __syncthreads();
        if (tid < 32) {
            double g = 0.0;
#pragma unroll
            for (int q = 0; q < 4 * NW; ++q)
                g += sred[q * 32 + tid];
            *gv_out = g;
        }
    }
    return out;
}

// K8d: Gamma = W z  (Gamma = K yTilde = T S^-1 yTilde = W L^-1 yTilde). 64 rows per workgroup, the m columns split
// over 4 lane groups, reduced through LDS.
__global__ void __launch_bounds__(256) k_gamma(int n, int m, int ldz, const double* __restrict__ Wb, double* __restrict__ gamma) {
    __shared__ double sp[256];
    const int r = blockIdx.x * 64 + (threadIdx.x & 63);
    const int seg = threadIdx.x >> 6;
    const double* W = Wb + m;
    const double* z = Wb + m + n;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (r < n) {
        int p = seg;
        for (; p + 12 < m; p += 16) {
            s0 += W[r + (size_t)p * ldz] * z[(size_t)p * ldz];
            s1 += W[r + (size_t)(p + 4) * ldz] * z[(size_t)(p + 4) * ldz];
            s2 += W[r + (size_t)(p + 8) * ldz] * z[(size_t)(p + 8) * ldz];
            s3 += W[r + (size_t)(p + 12) * ldz] * z[(size_t)(p + 12) * ldz];
        }
        for (; p < m; p += 4)
            s0 += W[r + (size_t)p * ldz] * z[(size_t)p * ldz];
    }
    sp[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (seg == 0 && r < n)
        gamma[r] = (sp[threadIdx.x] + sp[64 + threadIdx.x]) + (sp[128 + threadIdx.x] + sp[192 + threadIdx.x]);
}

// K9: Sigma <- Sigma - W W^T  ( = Sigma - K C Sigma, VIO_eqf.cpp:131 ): lower 32x32 tiles computed (one workgroup
// each, K = m split over its 8 waves), the strictly-lower ones mirrored so Sigma stays exactly symmetric.
//
// With with_gamma = 0 the diagonal tiles skip the Gamma by-product: the factorisation's last launch has produced Gamma as
// partial vectors and k_lift, launched BEFORE this kernel, has summed them, lifted the landmarks and rung the host doorbell.
// That order gives the host the frame's results one kernel early: its round trip (results, filter logic, the next frame's
// launches) overlaps with Sigma -= W W^T instead of leaving the GPU idle after it.
constexpr int SYRK_ARITH_TILES = 24; // up to this many tile rows (N <= 249) k_syrk_lift walks the lower triangle row by row (no table look-up in front of its operand loads)
constexpr int SYRK_NW = 8; // waves per workgroup: the K range of a tile is split 8-way (a wave's k-steps are a serial load->MFMA chain)
template <typename TS, bool WITH_GAMMA>
// tile_of_block (round 3): which lower tile (bi | bj << 16) block b works on. Workgroups are dealt round robin to the 8 XCDs (block b runs on XCD b % 8) and every
// XCD has its own L2: with the tiles handed out in plain column order each XCD touched every 32-row panel of W (131 MB fetched for 12 MB of W at N = 500).
// The table (built on the host, eqf_hip.hip: build_syrk_order) gives XCD x a compact square of the tile triangle and walks it column by column, so that an
// XCD fetches ~2 sqrt(tiles / 8) row panels of W instead of all of them.
__device__ __forceinline__ void syrk_sub_tile(int n, int m, int ld, int ldz, const double* __restrict__ Wb, TS* __restrict__ Sig, double* __restrict__ gamma, const int* __restrict__ spec,
                                              int spec_seq, const int* __restrict__ flags, trace_t* tr, const int* __restrict__ tile_of_block, int stall_seq, int blk, const int* __restrict__ live_cols = nullptr, int nt_arith = 0) {
    if (tr && blk == 0 && threadIdx.x == 0)
        *tr = wall_clock64();
    // Round 6: the status words are REQUESTED here and looked at in front of the store (a failed or cancelled update is the rare case: its products are wasted, nothing else), and up to
    // SYRK_ARITH_TILES tile rows the tile comes from the block index in closed form, row by row through the lower triangle, instead of from the XCD-aware table (which pays
    // from N = 256 on, where the XCDs' L2s do not hold W any more): no memory round trip in front of the first operand loads
    const int f0 = flags[0], f3 = flags[3];
    const int specv = spec ? *spec : 0;
    __shared__ double sred[1024 * 4];
    int bi, bj;
    if (nt_arith > 0) {
        bi = (int)((sqrtf(8.0f * (float)blk + 1.0f) - 1.0f) * 0.5f);
        while ((bi + 1) * (bi + 2) / 2 <= blk)
            ++bi;
        while (bi * (bi + 1) / 2 > blk)
            --bi;
        bj = blk - bi * (bi + 1) / 2;
    } else {
        const int code = tile_of_block[blk];
        bi = code & 0xffff, bj = code >> 16;
    }
    const int i0 = bi * 32, j0 = bj * 32;
    const double* W = Wb + m;
    // live_cols (EQF_OPT_LIVE_COLUMNS_FIRST): W is zero behind the last panel that holds a live column (eqf_lookahead.hpp: la_live_panels) - the sums end there
    const int mk = live_cols ? min(m, 32 * max((__builtin_amdgcn_readfirstlane(*live_cols) + 31) >> 5, 1)) : m;
    // Round 6: this tile's entries of Sigma are requested NOW, in front of the products, instead of behind their reduction (one memory round trip, ~1 us, off the end of the
    // kernel that stands between the factorisation and the next frame's propagation)
    const int i = i0 + (threadIdx.x & 31);
    TS sig_pre[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = j0 + (threadIdx.x >> 5) + 8 * e;
        sig_pre[e] = (threadIdx.x < 256 && i < n && j < n && (bi != bj || i >= j)) ? Sig[i + (size_t)j * ld] : TS(0);
    }
    // with_gamma: the diagonal tiles also produce Gamma[i0 : i0+32] = W[rows] z  (Gamma = K yTilde = W L^-1 yTilde, VIO_eqf.cpp:119)
    double gv = 0.0;
    TileRed t;
    if (WITH_GAMMA && bi == bj)
        t = mfma_tile32_splitk<true, SYRK_NW>(W, ldz, i0, n, W, ldz, j0, n, mk, sred, Wb + m + n, ldz, &gv);
    else
        t = mfma_tile32_splitk<false, SYRK_NW>(W, ldz, i0, n, W, ldz, j0, n, mk, sred);
    if (spec && specv == spec_seq)
        return; // cancelled speculative tail
    if (f0 | (f3 == stall_seq ? 1 : 0))
        return; // the factorisation failed (see k_lift): Sigma stays as it was
    if (WITH_GAMMA && bi == bj && threadIdx.x < 32 && i0 + threadIdx.x < n)
        gamma[i0 + threadIdx.x] = gv;
    if (threadIdx.x >= 256)
        return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = j0 + (threadIdx.x >> 5) + 8 * e;
        if (i < n && j < n && (bi != bj || i >= j)) {
            const double v = sig_pre[e] - t.v[e];
            Sig[i + (size_t)j * ld] = v; // (written through instead - no dirty lines for the kernel's end to write back - measured slower: N = 50 45.8 against 44.5 us per frame)
            if (i != j)
                Sig[j + (size_t)i * ld] = v;
        }
    }
    trace_end(tr);
}
template <typename TS, bool WITH_GAMMA>
__global__ void __launch_bounds__(64 * SYRK_NW) k_syrk_sub(int n, int m, int ld, int ldz, const double* __restrict__ Wb, TS* __restrict__ Sig, int nt,
                                                  double* __restrict__ gamma, const int* __restrict__ spec, int spec_seq, int with_gamma, const int* __restrict__ flags,
                                                  trace_t* tr, const int* __restrict__ tile_of_block, int stall_seq) {
    syrk_sub_tile<TS, WITH_GAMMA>(n, m, ld, ldz, Wb, Sig, gamma, spec, spec_seq, flags, tr, tile_of_block, stall_seq, (int)blockIdx.x);
}


// ---------------------------------------------------------------------------------------------------
// K10: landmark part of X <- Delta * X with Delta lifted from Gamma (liftInnovation / liftInnovationDiscrete,
// euclid.cpp:36-97, invdepth.cpp:183-253; VIOExp VIOGroup.cpp:273-290; X = Delta * X VIO_eqf.cpp:130).
// Also writes the new estimates q_hat_i and a flag per landmark with Q_i.a outside (1e-8, 1e8]
// (removeInvalidLandmarks, VIO_eqf.cpp:213-223) into `est` (4 planes of stride N: qx, qy, qz, invalid).
// what the lift of landmark i reads besides Gamma, requested up front (one memory round trip for everything)
struct LiftIn {
    V3 p0;
    Qt q;
    double a;
    M3 r0m;
};
__device__ __forceinline__ LiftIn lift_load(int i, int Ncap, int chart, int discrete, const double* q0, const double* Qq, const double* Qa) {
    LiftIn in;
    in.p0 = ld3(q0, Ncap, i);
    in.q = ldq(Qq, Ncap, i);
    in.a = Qa[i];
    in.r0m = (chart == EQVIO_COORD_INVDEPTH && !discrete) ? ld_cc(q0, Ncap, i, CC_R0) : M3{}; // ind2euc_r0(p0): stored chart constant
    return in;
}
__device__ __forceinline__ void lift_landmark(int i, const V3 g, const LiftIn& in, int N, int Ncap, int chart, int discrete, double* __restrict__ Qq, double* __restrict__ Qa,
                                              double* __restrict__ est) {
    const V3 p0 = in.p0;
    Qt Dq;
    double Da;
    if (discrete) {
        // liftInnovationDiscrete: q1 = chart^-1(gamma_i) about q0 (euclid.cpp:86-91, invdepth.cpp:242-247; normal.cpp:52-55 goes through the Normal chart's inverse)
        const V3 q1 = (chart == EQVIO_COORD_INVDEPTH) ? invdepth_chart_inv(g, p0) : ((chart == EQVIO_COORD_NORMAL) ? normal_chart_inv(g, p0) : p0 + g);
        Dq = so3_from_vectors(normalized(q1), normalized(p0));
        Da = norm(p0) / norm(q1);
    } else {
        // liftInnovation_normal = liftInnovation_euclid(M^-1 gamma) (normal.cpp:47-50)
        const V3 ge = (chart == EQVIO_COORD_INVDEPTH) ? in.r0m * g : ((chart == EQVIO_COORD_NORMAL) ? normal_Minv(p0) * g : g);
        const double iq2 = 1.0 / norm2(p0);
        const V3 Wr = (-iq2) * cross(p0, ge);
        const double Ws = -iq2 * dot(p0, ge);
        Dq = so3_exp(Wr);
        Da = exp(Ws);
    }
    const Qt q = q_mul(Dq, in.q);
    const double a = Da * in.a;
    Qq[i] = q.w;
    Qq[Ncap + i] = q.x;
    Qq[2 * Ncap + i] = q.y;
    Qq[3 * Ncap + i] = q.z;
    Qa[i] = a;
    const V3 qh = (1.0 / a) * q_rot(q_inv(q), p0);
    est[i] = qh.x;
    est[N + i] = qh.y;
    est[2 * N + i] = qh.z;
    est[3 * N + i] = (a <= 1e-8 || a > 1e8 || !(a == a)) ? 1.0 : 0.0;
}
// Gamma row: direct, or the fixed-order sum of the GAMMA_G + 1 partial vectors of k_chol_step's last launch
__device__ __forceinline__ double gamma_row(const double* __restrict__ gamma, const double* __restrict__ gpart, int ldg, int row) {
    static_assert(GAMMA_G == 4, "partial sum order below");
    if (!gpart)
        return gamma[row];
    return ((gpart[row] + gpart[(size_t)ldg + row]) + (gpart[2 * (size_t)ldg + row] + gpart[3 * (size_t)ldg + row])) + gpart[4 * (size_t)ldg + row];
}
struct LiftArgs {
    int N, Ncap, chart, discrete;
    double* gamma;
    const double* q0;
    double *Qq, *Qa, *est, *gamma_host;
    const int* flags;
    int *flags_host, *door_count, *door_host;
    int door_seq;
    const int* spec;
    int spec_seq;
    const double* gpart;
    int ldg;
    trace_t* tr;
    int stall_seq;
};
// one block of 64 landmarks (blk of nblk; threads 0 .. 63 of the workgroup)
__device__ __forceinline__ void lift_block(const LiftArgs& la, const int blk, const int nblk) {
    const int N = la.N, Ncap = la.Ncap, chart = la.chart, discrete = la.discrete, ldg = la.ldg, spec_seq = la.spec_seq, stall_seq = la.stall_seq;
    double* __restrict__ gamma = la.gamma;
    const double* __restrict__ q0 = la.q0;
    double *__restrict__ Qq = la.Qq, *__restrict__ Qa = la.Qa, *__restrict__ est = la.est, *__restrict__ gamma_host = la.gamma_host;
    const int *__restrict__ flags = la.flags, *__restrict__ spec = la.spec;
    int* __restrict__ flags_host = la.flags_host;
    const double* __restrict__ gpart = la.gpart;
    trace_t* tr = la.tr;
    if (tr && blk == 0 && threadIdx.x == 0)
        *tr = wall_clock64();
    // est / gamma_host / flags_host point into the pinned host packet: the results reach the host without copy kernels.
    // Every load this kernel needs is requested before the first result is used (Gamma partials, landmark state, chart constant, status
    // words, cancellation word): one memory round trip instead of three on the path to the doorbell.
    const int i = blk * 64 + threadIdx.x;
    const bool lm = i < N;
    const int ic = lm ? i : 0;
    const double gs = gamma_row(gamma, gpart, ldg, i < 21 ? i : 0);
    const double g0 = gamma_row(gamma, gpart, ldg, 21 + 3 * ic), g1 = gamma_row(gamma, gpart, ldg, 21 + 3 * ic + 1), g2 = gamma_row(gamma, gpart, ldg, 21 + 3 * ic + 2);
    const LiftIn in = lift_load(ic, Ncap, chart, discrete, q0, Qq, Qa);
    const int f0 = flags[0], f1 = flags[1], f3 = flags[3] == stall_seq ? 1 : 0; // the stall word of the look-ahead launch in front (stall_seq = -1: launch chain, cannot stall)
    const int specv = spec ? __hip_atomic_load(spec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const bool aborted = spec && specv == spec_seq; // speculative tail cancelled by the statistics kernel
    // A factorisation that met a non-positive pivot (flags[0], EQF_E_NOT_SPD) or whose bounded wait ran out (flags[3], EQF_E_STALLED) leaves
    // the filter as it was: no landmark is lifted here, k_syrk_sub does not touch Sigma, the host does not apply the sensor lift. The
    // status words are cleared by the first kernel of the next update.
    const bool failed = f0 != 0 || f3 != 0;
    if (i == 0) {
        flags_host[2] = aborted ? 1 : 0;
        flags_host[3] = f3; // look-ahead factorisation: a bounded wait ran out
    }
    if (!aborted && i == 0) {
        flags_host[0] = f0;
        flags_host[1] = f1;
    }
    if (!aborted && !failed) {
        if (i < 21) {
            gamma_host[i] = gs;
            if (gpart)
                gamma[i] = gs;
        }
        if (lm) {
            if (gpart) {
                gamma[21 + 3 * i] = g0;
                gamma[21 + 3 * i + 1] = g1;
                gamma[21 + 3 * i + 2] = g2;
            }
            lift_landmark(i, V3{g0, g1, g2}, in, N, Ncap, chart, discrete, Qq, Qa, est);
        }
    }
    ring_doorbell(la.door_count, la.door_host, la.door_seq, nblk);
    trace_end(tr);
}
__global__ void __launch_bounds__(64) k_lift(const LiftArgs la) { lift_block(la, (int)blockIdx.x, (int)gridDim.x); }
// EQF_OPT_LIFT_WITH_SYRK: k_lift and k_syrk_sub as ONE launch. Both only wait for the factorisation, neither reads what the other writes (landmark elements and the result
// packet here, Sigma there): the first nlift workgroups lift 64 landmarks each with their first wave and ring the doorbell, the others take a tile of Sigma each. The
// doorbell rings when it did with two launches; Sigma - which the NEXT frame's first kernel waits for - is complete one lift and one kernel boundary earlier.
template <typename TS>
__global__ void __launch_bounds__(64 * SYRK_NW) k_syrk_lift(int n, int m, int ld, int ldz, const double* __restrict__ Wb, TS* __restrict__ Sig, double* __restrict__ gamma,
                                                            const int* __restrict__ flags, trace_t* tr, const int* __restrict__ tile_of_block, const LiftArgs la, int nlift, const int* __restrict__ live_cols, int nt_arith) {
    if ((int)blockIdx.x < nlift) {
        if (threadIdx.x < 64)
            lift_block(la, (int)blockIdx.x, nlift);
        return;
    }
    syrk_sub_tile<TS, false>(n, m, ld, ldz, Wb, Sig, gamma, la.spec, la.spec_seq, flags, tr, tile_of_block, la.stall_seq, (int)blockIdx.x - nlift, live_cols, nt_arith);
}
// q_hat_i = Q_i^-1 q0_i for all landmarks (stateGroupAction, VIOGroup.cpp:44-52)
__global__ void __launch_bounds__(64) k_estimate(int N, int Ncap, const double* __restrict__ q0, const double* __restrict__ Qq, const double* __restrict__ Qa,
                                                 double* __restrict__ est, int* __restrict__ door_count = nullptr, int* __restrict__ door_host = nullptr, int door_seq = 0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) {
        const double a = Qa[i];
        const V3 qh = (1.0 / a) * q_rot(q_inv(ldq(Qq, Ncap, i)), ld3(q0, Ncap, i));
        est[i] = qh.x;
        est[N + i] = qh.y;
        est[2 * N + i] = qh.z;
        est[3 * N + i] = (a <= 1e-8 || a > 1e8 || !(a == a)) ? 1.0 : 0.0;
    }
    ring_doorbell(door_count, door_host, door_seq);
}

// ---------------------------------------------------------------------------------------------------
// landmark bookkeeping on the device
// AoS (host boundary) <-> SoA planes
__global__ void k_scatter_landmarks(int k, int dst0, int Ncap, const double* __restrict__ p_aos, const double* __restrict__ Q_aos, double* __restrict__ q0,
                                    double* __restrict__ Qq, double* __restrict__ Qa) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k)
        return;
    const int i = dst0 + t;
    for (int c = 0; c < 3; ++c)
        q0[c * Ncap + i] = p_aos[3 * t + c];
    { // chart constants of the new origin point (ld_cc): the only place they are computed
        const V3 p{p_aos[3 * t], p_aos[3 * t + 1], p_aos[3 * t + 2]};
        double* cc = q0 + (size_t)CC_OFF * Ncap;
        store_chart_constants(cc, Ncap, i, p.x, p.y, p.z, nullptr);
    }
    if (Q_aos) {
        for (int c = 0; c < 4; ++c)
            Qq[c * Ncap + i] = Q_aos[5 * t + c];
        Qa[i] = Q_aos[5 * t + 4];
    } else {
        Qq[i] = 1.0;
        Qq[Ncap + i] = 0.0;
        Qq[2 * Ncap + i] = 0.0;
        Qq[3 * Ncap + i] = 0.0;
        Qa[i] = 1.0;
    }
}
__global__ void k_gather_landmarks_aos(int N, int Ncap, const double* __restrict__ q0, const double* __restrict__ Qq, const double* __restrict__ Qa,
                                       double* __restrict__ out /* 8 doubles per landmark: p[3], q[4], a */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    for (int c = 0; c < 3; ++c)
        out[8 * i + c] = q0[c * Ncap + i];
    for (int c = 0; c < 4; ++c)
        out[8 * i + 3 + c] = Qq[c * Ncap + i];
    out[8 * i + 7] = Qa[i];
}
// integrateRiccatiStateDiscrete (VIO_eqf.cpp:93-103): A_d = numericalDifferential of a0Discrete (EqFMatrices.cpp:24-41) at 0, central
// differences with h = cbrt(eps) (Geometry.cpp:25-36). a0Discrete has the arrow structure of A: a landmark's coordinates move only its own
// three rows; the 21 sensor coordinates move everything. The sensor-level part of the 43 evaluations (nominal + 21 x +-h) is O(1) and done
// on the host; what a landmark needs of each is the camera-frame pose change cpc = T^-1 A^-1 T of the discrete lift (VIOGroup.cpp:252).
// One lane per landmark: its 3 x 21 sensor columns and its own 3 x 3 block of the dense A_d.
struct DiscreteAArgs {
    double h;
    Pose cpc[43]; // [0] nominal, [1 + 2 j + s] sensor coordinate j perturbed by +h (s = 0) / -h (s = 1)
};
struct Sot3 {
    Qt R;
    double a;
};
__device__ __forceinline__ Sot3 sot3_mul(const Sot3& x, const Sot3& y) { return Sot3{q_mul(x.R, y.R), x.a * y.a}; }
__device__ __forceinline__ Sot3 sot3_inv(const Sot3& x) { return Sot3{q_inv(x.R), 1.0 / x.a}; }
__device__ __forceinline__ Sot3 lift_q(const Pose& cpc, V3 p0) { // liftVelocityDiscrete landmark part (VIOGroup.cpp:255-268)
    const V3 p1 = pose_act(cpc, p0);
    return Sot3{so3_from_vectors(normalized(p1), normalized(p0)), norm(p0) / norm(p1)};
}
__device__ __forceinline__ V3 chart_fwd(int chart, V3 q, V3 q0) { return chart == EQVIO_COORD_NORMAL ? normal_chart(q, q0) : point_chart(chart == EQVIO_COORD_INVDEPTH, q, q0); }
__device__ __forceinline__ V3 chart_bwd(int chart, V3 e, V3 q0) {
    return chart == EQVIO_COORD_INVDEPTH ? invdepth_chart_inv(e, q0) : (chart == EQVIO_COORD_NORMAL ? normal_chart_inv(e, q0) : q0 + e);
}
// epsilon_1 of landmark i for the perturbed origin point qe (= q0 for a sensor perturbation) and the lift's pose change cpc
__device__ __forceinline__ V3 a0_discrete_landmark(int chart, const Sot3& Qi, const Sot3& QLh_inv, const Pose& cpc, V3 qe, V3 q0) {
    const V3 q = (1.0 / Qi.a) * q_rot(q_inv(Qi.R), qe);              // phi(X, xi_e): Q_i^-1 q_e
    const Sot3 Lt = sot3_mul(lift_q(cpc, q), QLh_inv);               // LambdaTilde_i
    const Sot3 G = sot3_mul(sot3_mul(Qi, Lt), sot3_inv(Qi));         // (X LambdaTilde X^-1)_i
    const V3 q1 = (1.0 / G.a) * q_rot(q_inv(G.R), qe);               // phi(., xi_e)
    return chart_fwd(chart, q1, q0);
}
__global__ void __launch_bounds__(64) k_discrete_A(int N, int Ncap, int ldf, int chart, DiscreteAArgs da, const double* __restrict__ q0p, const double* __restrict__ Qq,
                                                  const double* __restrict__ Qa, double* __restrict__ F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    const V3 q0 = ld3(q0p, Ncap, i);
    const Sot3 Qi{ldq(Qq, Ncap, i), Qa[i]};
    const V3 qh = (1.0 / Qi.a) * q_rot(q_inv(Qi.R), q0);
    const Sot3 QLh_inv = sot3_inv(lift_q(da.cpc[0], qh)); // liftVelocityDiscrete(xi_hat)^-1, landmark i
    const double ih2 = 1.0 / (2.0 * da.h);
    const int r0 = 21 + 3 * i;
    for (int j = 0; j < 21; ++j) {
        const V3 ep = a0_discrete_landmark(chart, Qi, QLh_inv, da.cpc[1 + 2 * j], q0, q0);
        const V3 em = a0_discrete_landmark(chart, Qi, QLh_inv, da.cpc[2 + 2 * j], q0, q0);
        const V3 d = ih2 * (ep - em);
        F[r0 + (size_t)j * ldf] = d.x;
        F[r0 + 1 + (size_t)j * ldf] = d.y;
        F[r0 + 2 + (size_t)j * ldf] = d.z;
    }
    for (int k = 0; k < 3; ++k) {
        const V3 e = V3{k == 0 ? da.h : 0.0, k == 1 ? da.h : 0.0, k == 2 ? da.h : 0.0};
        const V3 ep = a0_discrete_landmark(chart, Qi, QLh_inv, da.cpc[0], chart_bwd(chart, e, q0), q0);
        const V3 em = a0_discrete_landmark(chart, Qi, QLh_inv, da.cpc[0], chart_bwd(chart, -e, q0), q0);
        const V3 d = ih2 * (ep - em);
        F[r0 + (size_t)(r0 + k) * ldf] = d.x;
        F[r0 + 1 + (size_t)(r0 + k) * ldf] = d.y;
        F[r0 + 2 + (size_t)(r0 + k) * ldf] = d.z;
    }
}

// Normal chart: A_n = M A_e M^-1, B_n = M B_e with the block-diagonal change of coordinates M (eqf_math.hpp, normal_M), so a Riccati step is
//   Sigma' = M [ F_e (M^-1 Sigma M^-T) F_e^T + dt B_e Q B_e^T (or the accurate form) ] M^T + dt P
// i.e. the Euclidean propagation between two congruences. This kernel is one of them: out = T Sigma T^T, T = M (dir > 0) or M^-1
// (dir < 0), plus dt P on the diagonal if addP. One thread per output entry; a row of T has at most 7 non-zeros.
struct NormalM {
    double v0[3];  // xi0 velocity
    double Ad[36]; // Adjoint(T0^-1), row-major 6 x 6, T0 = xi0 camera offset
    double dtP[8]; // dt * process variances by class (bias omega / accel, attitude, position, velocity, camera attitude / position, point)
    int dir, addP;
};
__device__ __forceinline__ int normal_row(const NormalM& nm, int i, int Ncap, const double* __restrict__ q0, int (&idx)[7], double (&co)[7]) {
    if (i < 21) {
        idx[0] = i;
        co[0] = 1.0;
        int k = 1;
        const double sg = nm.dir > 0 ? 1.0 : -1.0;
        if (i >= 12 && i < 15) { // [12:15, 6:9] = -skew(v0) (inverse: +skew(v0))
            const M3 K = skew(V3{nm.v0[0], nm.v0[1], nm.v0[2]});
            const V3 r = row(K, i - 12);
            idx[1] = 6, idx[2] = 7, idx[3] = 8;
            co[1] = -sg * r.x, co[2] = -sg * r.y, co[3] = -sg * r.z;
            k = 4;
        } else if (i >= 15) { // [15:21, 6:12] = Ad(T0^-1) (inverse: -Ad(T0^-1))
            for (int c = 0; c < 6; ++c) {
                idx[1 + c] = 6 + c;
                co[1 + c] = sg * nm.Ad[6 * (i - 15) + c];
            }
            k = 7;
        }
        return k;
    }
    const int l = (i - 21) / 3, r = (i - 21) % 3;
    const V3 p0 = ld3(q0, Ncap, l);
    const M3 T = nm.dir > 0 ? normal_M(p0) : normal_Minv(p0);
    const V3 tr = row(T, r);
    idx[0] = 21 + 3 * l, idx[1] = idx[0] + 1, idx[2] = idx[0] + 2;
    co[0] = tr.x, co[1] = tr.y, co[2] = tr.z;
    return 3;
}
template <typename TS>
__global__ void __launch_bounds__(256) k_congruence_normal(int n, int Ncap, int ld, NormalM nm, const double* __restrict__ q0, const TS* __restrict__ Sin, TS* __restrict__ Sout) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= n || j >= n || i < j)
        return; // lower triangle, mirrored below: Sigma stays exactly symmetric
    int ii[7], jj[7];
    double ci[7], cj[7];
    const int ni = normal_row(nm, i, Ncap, q0, ii, ci), nj = normal_row(nm, j, Ncap, q0, jj, cj);
    double acc = 0.0;
    for (int b = 0; b < nj; ++b) {
        double t = 0.0;
        for (int a = 0; a < ni; ++a)
            t = fma(ci[a], (double)Sin[ii[a] + (size_t)jj[b] * ld], t);
        acc = fma(cj[b], t, acc);
    }
    if (nm.addP && i == j)
        acc += nm.dtP[i < 21 ? i / 3 : 7];
    Sout[i + (size_t)j * ld] = (TS)acc;
    Sout[j + (size_t)i * ld] = (TS)acc;
}
// Landmark bookkeeping of a frame in ONE pass: removeLandmarkByIndex (VIO_eqf.cpp:172-178) and addNewLandmarks (:225-245) are recorded by the
// host as a map new landmark -> old landmark (>= 0) or -(t + 1) for the t-th appended landmark, and applied here when the state is next needed:
// Sigma_new = Sigma_old[map, map] with zero strips and var_t on the diagonal of appended landmarks, the landmark planes gathered / initialised
// (Q = identity, chart constants of the new origin point), everything written to the other buffers. Pure data movement: the same bits as
// one gather / scatter / append pass per call would give.
// grid: (ceil(nnew / 256), nnew + ceil(Nnew / 256)); rows blockIdx.y >= nnew handle the landmark planes.
template <typename TS, typename MapT>
__device__ __forceinline__ void reshape_body(int Nnew, int Ncap, int ld, const MapT* __restrict__ map, const double* __restrict__ newp, const double* __restrict__ newvar,
                                             const TS* __restrict__ Sin, TS* __restrict__ Sout, const double* __restrict__ st_in, const double* __restrict__ lm_in,
                                             double* __restrict__ st_out, double* __restrict__ lm_out) {
    const int nnew = 21 + 3 * Nnew;
    if ((int)blockIdx.y < nnew) {
        const int r = blockIdx.x * blockDim.x + threadIdx.x;
        const int c = blockIdx.y;
        if (r >= nnew)
            return;
        const int mr = r < 21 ? 0 : map[(r - 21) / 3], mc = c < 21 ? 0 : map[(c - 21) / 3];
        TS v;
        if (mr >= 0 && mc >= 0) {
            const int ro = r < 21 ? r : 21 + 3 * mr + (r - 21) % 3;
            const int co = c < 21 ? c : 21 + 3 * mc + (c - 21) % 3;
            v = Sin[ro + (size_t)co * ld];
        } else
            v = (TS)((r == c) ? newvar[-mr - 1] : 0.0);
        Sout[r + (size_t)c * ld] = v;
        return;
    }
    if (blockIdx.x != 0)
        return;
    const int i = ((int)blockIdx.y - nnew) * 256 + threadIdx.x;
    if (i >= Nnew)
        return;
    const int o = map[i];
    double* Qqo = lm_out;
    double* Qao = lm_out + 4 * (size_t)Ncap;
    if (o >= 0) {
        for (int c = 0; c < CC_OFF + CC_PLANES; ++c) // origin point and its chart constants travel with the landmark
            if (c < 3 || c >= CC_OFF)
                st_out[(size_t)c * Ncap + i] = st_in[(size_t)c * Ncap + o];
        for (int c = 0; c < 4; ++c)
            Qqo[c * Ncap + i] = lm_in[c * Ncap + o];
        Qao[i] = lm_in[4 * (size_t)Ncap + o];
    } else {
        const int t = -o - 1;
        const V3 p{newp[3 * t], newp[3 * t + 1], newp[3 * t + 2]};
        st_out[i] = p.x;
        st_out[Ncap + i] = p.y;
        st_out[2 * Ncap + i] = p.z;
        double* cc = st_out + (size_t)CC_OFF * Ncap;
        store_chart_constants(cc, Ncap, i, p.x, p.y, p.z, nullptr);
        Qqo[i] = 1.0;
        Qqo[Ncap + i] = 0.0;
        Qqo[2 * Ncap + i] = 0.0;
        Qqo[3 * Ncap + i] = 0.0;
        Qao[i] = 1.0;
    }
}
template <typename TS>
__global__ void __launch_bounds__(256) k_reshape(int Nnew, int Ncap, int ld, const int* __restrict__ map, const double* __restrict__ newp, const double* __restrict__ newvar,
                                                 const TS* __restrict__ Sin, TS* __restrict__ Sout, const double* __restrict__ st_in, const double* __restrict__ lm_in,
                                                 double* __restrict__ st_out, double* __restrict__ lm_out) {
    reshape_body(Nnew, Ncap, ld, map, newp, newvar, Sin, Sout, st_in, lm_in, st_out, lm_out);
}
// The same pass with the record in the kernel's argument segment (round 3): a frame of the realistic mix has two of these (old landmarks out before the
// statistics, outliers out + new landmarks in before the update), and the copy command in front of each was a host call of 4-5 us plus a 4 us blit
// kernel on the stream. Up to RESHAPE_ARG_MAP landmarks and RESHAPE_ARG_NEW new ones (2.3 KB of arguments); larger records take the copy.
constexpr int RESHAPE_ARG_MAP = 768, RESHAPE_ARG_NEW = 24;
struct ReshapeArgs {
    short map[RESHAPE_ARG_MAP]; // >= 0: old index; -(t + 1): new landmark t
    double p[RESHAPE_ARG_NEW * 3];
    double var[RESHAPE_ARG_NEW];
};
template <typename TS>
__global__ void __launch_bounds__(256) k_reshape_args(int Nnew, int Ncap, int ld, const ReshapeArgs ra, const TS* __restrict__ Sin, TS* __restrict__ Sout,
                                                      const double* __restrict__ st_in, const double* __restrict__ lm_in, double* __restrict__ st_out, double* __restrict__ lm_out) {
    reshape_body(Nnew, Ncap, ld, ra.map, ra.p, ra.var, Sin, Sout, st_in, lm_in, st_out, lm_out);
}
// k_reshape for the commonest record of all - nothing removed, up to APPEND_MAX landmarks appended (a frame's addNewLandmarks): nothing
// moves, only the new strips of Sigma (zeros, the variances on the diagonal) and the new landmarks' planes are written, in place, and the few
// numbers travel as kernel arguments (no copy command). grid: (ceil(nnew / 256), 3 k + 1): row y < 3 k writes column nold + y and row nold + y
// of Sigma, the last row the landmark planes.
constexpr int APPEND_MAX = 24;
struct AppendArgs {
    double p[APPEND_MAX][3];
    double var[APPEND_MAX];
};
template <typename TS>
__global__ void __launch_bounds__(256) k_append_inplace(int Nold, int k, int Ncap, int ld, const AppendArgs aa, TS* __restrict__ Sig, double* __restrict__ st, double* __restrict__ lm) {
    const int nold = 21 + 3 * Nold, nnew = nold + 3 * k;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if ((int)blockIdx.y < 3 * k) {
        if (x >= nnew)
            return;
        const int c = nold + blockIdx.y;
        const TS d = (TS)aa.var[blockIdx.y / 3];
        Sig[x + (size_t)c * ld] = (x == c) ? d : (TS)0.0;
        if (x < nold)
            Sig[c + (size_t)x * ld] = (TS)0.0;
        return;
    }
    if (x >= k)
        return;
    const int i = Nold + x;
    const V3 p{aa.p[x][0], aa.p[x][1], aa.p[x][2]};
    st[i] = p.x;
    st[Ncap + i] = p.y;
    st[2 * Ncap + i] = p.z;
    double* cc = st + (size_t)CC_OFF * Ncap;
    store_chart_constants(cc, Ncap, i, p.x, p.y, p.z, nullptr);
    lm[i] = 1.0;
    lm[Ncap + i] = 0.0;
    lm[2 * Ncap + i] = 0.0;
    lm[3 * Ncap + i] = 0.0;
    lm[4 * (size_t)Ncap + i] = 1.0;
}
template <typename TS>
__global__ void __launch_bounds__(256) k_set_diag(int n, int ld, const double* __restrict__ diag, TS* __restrict__ Sig) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (r >= n)
        return;
    Sig[r + (size_t)c * ld] = (r == c) ? diag[r] : 0.0;
}
// count non-finite entries of Sigma (the reference's assert(!Sigma.hasNaN()))
// EQF_OPT_SIGMA_FP32: numerical model of an fp32 Sigma store - every element rounded to the nearest float
// mixed = 1 (EQF_OPT_SIGMA_FP32 = 3): only the landmark-landmark OFF-DIAGONAL 3 x 3 blocks are rounded; the 21 x 21 sensor block, the sensor-landmark
// strips and the 3 x 3 landmark diagonal blocks (21 (n + 3 N) - 441 + 9 N entries: 4.6 % of Sigma at N = 200) keep their doubles - the numerical model of a store that
// holds those separately in fp64
__global__ void __launch_bounds__(256) k_round_f32(int n, int ld, double* __restrict__ Sig, const int* __restrict__ spec, int spec_seq, int mixed) {
    if (spec && *spec == spec_seq)
        return;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (r >= n)
        return;
    if (mixed && (r < 21 || c < 21 || (r - 21) / 3 == (c - 21) / 3))
        return;
    Sig[r + (size_t)c * ld] = (double)(float)Sig[r + (size_t)c * ld];
}
// storage conversion between the two Sigma buffers (same leading dimension in elements)
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) k_convert_sigma(int n, int ld, const TI* __restrict__ in, TO* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (r < n)
        out[r + (size_t)c * ld] = (TO)in[r + (size_t)c * ld];
}
template <typename TS>
__global__ void __launch_bounds__(256) k_check_finite(int n, int ld, const TS* __restrict__ Sig, int* __restrict__ flags) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (r >= n)
        return;
    const double v = Sig[r + (size_t)c * ld];
    if (!(v - v == 0.0))
        flags[1] = 1;
}

// ---------------------------------------------------------------------------------------------------
// Dense fp64 MFMA GEMM used by the dense-Riccati mode (EQF_OPT_RICCATI_DENSE): C = A * B^T with A (Mr x K),
// B (Nc x K), all column-major. One workgroup per 32x32 tile (K split over its 4 waves).
__global__ void __launch_bounds__(256) k_gemm_nt(int Mr, int Nc, int K, const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                                 double* __restrict__ Cm, int ldc) {
    __shared__ double sred[4096];
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const TileRed t = mfma_tile32_splitk<false>(A, lda, i0, Mr, B, ldb, j0, Nc, K, sred);
    const int i = i0 + (threadIdx.x & 31);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = j0 + (threadIdx.x >> 5) + 8 * e;
        if (i < Mr && j < Nc)
            Cm[i + (size_t)j * ldc] = t.v[e];
    }
}
// F = I + dt*A materialised dense (row i, col j) column-major, from the packed blocks (dense-Riccati mode)
__global__ void __launch_bounds__(256) k_build_F(int N, int Ncap, int n, int ldf, double dt, const Common* __restrict__ cm, const double* __restrict__ Al,
                                                 double* __restrict__ F) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (r >= n)
        return;
    double v = 0.0;
    if (r < 21) {
        if (c < 21)
            v = dt * cm->Ass[r * 21 + c];
    } else {
        const int i = (r - 21) / 3, rr = (r - 21) % 3;
        if (c < 21) {
            int e = -1;
            if (c < 3)
                e = c;
            else if (c >= 12 && c < 15)
                e = 3 + (c - 12);
            else if (c >= 15)
                e = 6 + (c - 15);
            if (e >= 0)
                v = dt * Al[(rr * 15 + e) * Ncap + i];
        } else if ((c - 21) / 3 == i) {
            v = dt * Al[(rr * 15 + 12 + (c - 21) % 3) * Ncap + i];
        }
    }
    if (r == c)
        v += 1.0;
    F[r + (size_t)c * ldf] = v;
}
// Sigma' = M + dt (B Q B^T + P) elementwise finish for the dense mode (M = F Sigma F^T already in Sout)
__global__ void __launch_bounds__(256) k_add_noise(int N, int Ncap, int n, int ld, RiccatiArgs ra, const Common* __restrict__ cm, const double* __restrict__ Bl,
                                                   double* __restrict__ Sout) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (r >= n)
        return;
    auto brow = [&](int x, int q) -> double {
        if (x < 21)
            return cm->Bs[x * 12 + q];
        if (q >= 3)
            return 0.0;
        return Bl[(((x - 21) % 3) * 3 + q) * Ncap + (x - 21) / 3];
    };
    double bq = 0;
    for (int q = 0; q < 12; ++q)
        bq += brow(r, q) * ra.Qd[q] * brow(c, q);
    double v = Sout[r + (size_t)c * ld] + ra.dt * bq;
    if (r == c)
        v += ra.dt * (r < 21 ? ra.Pd[r / 3] : ra.Pd[7]);
    Sout[r + (size_t)c * ld] = v;
}

// ---------------------------------------------------------------------------------------------------
// Accurate Riccati (integrateRiccatiStateAccurate, VIO_eqf.cpp:74-91): exp(dt [[A, B],[0, 0]]) with the structure
//   A = [[A_ss, 0],[A_ls, blkdiag(D_i)]]  =>  the exponential of the (n+12)^2 matrix is block lower triangular too:
//   sensor part  E = exp(P),  P = dt [[A_ss, B_s],[0, 0]]  (33 x 33, only its first 21 rows are non-trivial),
//   landmark i   [Y_i | F_i] with  exp(dt [[P/dt, 0],[R_i, D_i]]) = [[E, 0],[Y_i, F_i]],  R_i = [A_ls_i | B_l_i]  (3 x 33).
// Both by Taylor series (degree EXPM_K) of the matrix scaled by 2^-s, then s squarings
//   [[E,0],[Y,F]]^2 = [[E^2, 0],[Y E + F Y, F^2]].
// expinfo[0] = s (chosen on the device from the infinity norm so that no host round trip is needed).
constexpr int EXPM_K = 14;
constexpr int EXPM_SMAX = 12;
// One workgroup. Ebuf: (EXPM_SMAX + 1) x (21 x 33) row-major: Ebuf[0] = Ps = P / 2^s (scaled generator, top 21 rows),
// Ebuf[1 + j] = E after j squarings (top 21 rows; rows 21..32 of every E are [0 I]). The final E is Ebuf[1 + s].
__global__ void __launch_bounds__(256) k_expm_sensor(int N, int Ncap, double dt, const Common* __restrict__ cm, const double* __restrict__ Al,
                                                     const double* __restrict__ Bl, double* __restrict__ Ebuf, int* __restrict__ expinfo) {
    __shared__ double sP[21 * 33], sT[21 * 33], sE[21 * 33], sN[21 * 33];
    __shared__ double sred[256];
    __shared__ int s_s;
    const int tid = threadIdx.x;
    // infinity norm (max abs row sum) of dt [[A, B]]
    double mx = 0.0;
    for (int r = tid; r < 21; r += 256) {
        double a = 0.0;
        for (int c = 0; c < 21; ++c)
            a += fabs(cm->Ass[r * 21 + c]);
        for (int c = 0; c < 12; ++c)
            a += fabs(cm->Bs[r * 12 + c]);
        mx = fmax(mx, a);
    }
    for (int t = tid; t < 3 * N; t += 256) {
        const int i = t / 3, r = t % 3;
        double a = 0.0;
        for (int e = 0; e < 15; ++e)
            a += fabs(Al[(r * 15 + e) * Ncap + i]);
        for (int e = 0; e < 3; ++e)
            a += fabs(Bl[(r * 3 + e) * Ncap + i]);
        mx = fmax(mx, a);
    }
    sred[tid] = mx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st)
            sred[tid] = fmax(sred[tid], sred[tid + st]);
        __syncthreads();
    }
    if (tid == 0) {
        const double nrm = dt * sred[0];
        int sc = 0;
        while (sc < EXPM_SMAX && ldexp(nrm, -sc) > 0.25)
            ++sc;
        s_s = sc;
        expinfo[0] = sc;
    }
    __syncthreads();
    const int sc = s_s;
    const double scale = ldexp(dt, -sc);
    for (int t = tid; t < 21 * 33; t += 256) {
        const int r = t / 33, c = t % 33;
        const double v = scale * (c < 21 ? cm->Ass[r * 21 + c] : cm->Bs[r * 12 + (c - 21)]);
        sP[t] = v;
        sT[t] = v;                   // current term Ps^k / k!
        sE[t] = v + (r == c ? 1.0 : 0.0); // I + Ps
        Ebuf[t] = v;
    }
    __syncthreads();
    // Taylor: term_k = term_{k-1} Ps / k  (rows 21..32 of Ps are zero: the inner index runs over 21)
    for (int k = 2; k <= EXPM_K; ++k) {
        for (int t = tid; t < 21 * 33; t += 256) {
            const int r = t / 33, c = t % 33;
            double a = 0.0;
            for (int q = 0; q < 21; ++q)
                a += sT[r * 33 + q] * sP[q * 33 + c];
            sN[t] = a / k;
        }
        __syncthreads();
        for (int t = tid; t < 21 * 33; t += 256) {
            sT[t] = sN[t];
            sE[t] += sN[t];
        }
        __syncthreads();
    }
    for (int t = tid; t < 21 * 33; t += 256)
        Ebuf[21 * 33 + t] = sE[t];
    // squarings: E <- E E with E = [[E_top],[0 I]]  =>  top rows: E_top[:, 0:21] E_top + [0 | E_top[:, 21:33]]
    for (int j = 0; j < sc; ++j) {
        __syncthreads();
        for (int t = tid; t < 21 * 33; t += 256) {
            const int r = t / 33, c = t % 33;
            double a = (c >= 21) ? sE[r * 33 + c] : 0.0;
            for (int q = 0; q < 21; ++q)
                a += sE[r * 33 + q] * sE[q * 33 + c];
            sN[t] = a;
        }
        __syncthreads();
        for (int t = tid; t < 21 * 33; t += 256) {
            sE[t] = sN[t];
            Ebuf[(2 + j) * 21 * 33 + t] = sN[t];
        }
    }
}
// One wave per landmark (4 per workgroup). Output per landmark: Yl 3 x 33 planes (Phi_ls | PhiB_l), Fl 3 x 3 planes.
__global__ void __launch_bounds__(256) k_expm_landmarks(int N, int Ncap, double dt, const double* __restrict__ Al, const double* __restrict__ Bl,
                                                        const double* __restrict__ Ebuf, const int* __restrict__ expinfo, double* __restrict__ Yl,
                                                        double* __restrict__ Fl) {
    __shared__ double sM[21 * 33];      // Ps, then E_j
    __shared__ double sR[4][3 * 33 + 3]; // per wave: current 3 x 33 row block
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = blockIdx.x * 4 + wave;
    const bool live = i < N;
    const int sc = expinfo[0];
    const double scale = ldexp(dt, -sc);
    for (int t = tid; t < 21 * 33; t += 256)
        sM[t] = Ebuf[t];
    // R_s (3 x 33): cols 0:3 | 12:15 | 15:21 from Al, cols 21:24 from Bl; S_s = scale * D
    double R[3] = {0, 0, 0};
    const int c = lane; // column owned by this lane (c < 33)
    if (live && c < 33) {
        int e = -1;
        if (c < 3)
            e = c;
        else if (c >= 12 && c < 15)
            e = 3 + (c - 12);
        else if (c >= 15 && c < 21)
            e = 6 + (c - 15);
        for (int r = 0; r < 3; ++r) {
            if (e >= 0)
                R[r] = scale * Al[(r * 15 + e) * Ncap + i];
            else if (c >= 21 && c < 24)
                R[r] = scale * Bl[(r * 3 + (c - 21)) * Ncap + i];
        }
    }
    double S[9], Sp[9], F[9];
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) {
            S[r * 3 + q] = live ? scale * Al[(r * 15 + 12 + q) * Ncap + i] : 0.0;
            Sp[r * 3 + q] = (r == q) ? 1.0 : 0.0; // S^(k-1) / (k-1)!  (k = 1)
            F[r * 3 + q] = ((r == q) ? 1.0 : 0.0) + S[r * 3 + q];
        }
    double T[3] = {R[0], R[1], R[2]}; // current term (3 x 33 block column c) : R_1
    double Y[3] = {R[0], R[1], R[2]};
    __syncthreads();
    // term_k = (term_{k-1} Ps + Spow_{k-1} R) / k with Spow_{k-1} = S^(k-1)/(k-1)!
    double Spow[9];
    for (int q = 0; q < 9; ++q)
        Spow[q] = S[q]; // S^1/1!
    for (int k = 2; k <= EXPM_K; ++k) {
        if (c < 33)
            for (int r = 0; r < 3; ++r)
                sR[wave][r * 33 + c] = T[r];
        __syncthreads();
        double nt[3] = {0, 0, 0};
        if (c < 33) {
            for (int q = 0; q < 21; ++q) {
                const double p = sM[q * 33 + c];
                nt[0] += sR[wave][q] * p;
                nt[1] += sR[wave][33 + q] * p;
                nt[2] += sR[wave][66 + q] * p;
            }
            for (int r = 0; r < 3; ++r)
                nt[r] = (nt[r] + Spow[r * 3 + 0] * R[0] + Spow[r * 3 + 1] * R[1] + Spow[r * 3 + 2] * R[2]) / k;
        }
        // Spow_k = Spow_{k-1} S / k
        double ns[9];
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q)
                ns[r * 3 + q] = (Spow[r * 3 + 0] * S[0 * 3 + q] + Spow[r * 3 + 1] * S[1 * 3 + q] + Spow[r * 3 + 2] * S[2 * 3 + q]) / k;
        __syncthreads();
        for (int r = 0; r < 3; ++r) {
            T[r] = nt[r];
            Y[r] += nt[r];
        }
        for (int q = 0; q < 9; ++q) {
            Spow[q] = ns[q];
            F[q] += ns[q];
        }
    }
    // squarings: Y <- Y E + F Y, F <- F F
    for (int j = 0; j < sc; ++j) {
        __syncthreads();
        for (int t = tid; t < 21 * 33; t += 256)
            sM[t] = Ebuf[(1 + j) * 21 * 33 + t];
        if (c < 33)
            for (int r = 0; r < 3; ++r)
                sR[wave][r * 33 + c] = Y[r];
        __syncthreads();
        if (c < 33) {
            double ny[3];
            for (int r = 0; r < 3; ++r)
                ny[r] = (c >= 21) ? sR[wave][r * 33 + c] : 0.0; // rows 21..32 of E are [0 I]
            for (int q = 0; q < 21; ++q) {
                const double p = sM[q * 33 + c];
                ny[0] += sR[wave][q] * p;
                ny[1] += sR[wave][33 + q] * p;
                ny[2] += sR[wave][66 + q] * p;
            }
            for (int r = 0; r < 3; ++r)
                ny[r] += F[r * 3 + 0] * Y[0] + F[r * 3 + 1] * Y[1] + F[r * 3 + 2] * Y[2];
            for (int r = 0; r < 3; ++r)
                Y[r] = ny[r];
        }
        double nf[9];
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q)
                nf[r * 3 + q] = F[r * 3 + 0] * F[0 * 3 + q] + F[r * 3 + 1] * F[1 * 3 + q] + F[r * 3 + 2] * F[2 * 3 + q];
        for (int q = 0; q < 9; ++q)
            F[q] = nf[q];
    }
    if (live) {
        if (c < 33)
            for (int r = 0; r < 3; ++r)
                Yl[(r * 33 + c) * Ncap + i] = Y[r];
        if (lane < 9)
            Fl[lane * Ncap + i] = F[lane];
    }
}
// Dense Phi (n x n) and Phi_B (n x 12), column-major, from the structured exponential.
__global__ void __launch_bounds__(256) k_build_phi(int N, int Ncap, int n, int ldf, const double* __restrict__ Ebuf, const int* __restrict__ expinfo,
                                                   const double* __restrict__ Yl, const double* __restrict__ Fl, double* __restrict__ Phi,
                                                   double* __restrict__ PhiB) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y; // 0 .. n + 12
    if (r >= n)
        return;
    const double* E = Ebuf + (size_t)(1 + expinfo[0]) * 21 * 33;
    double v = 0.0;
    if (c < n) {
        if (r < 21) {
            if (c < 21)
                v = E[r * 33 + c];
        } else {
            const int i = (r - 21) / 3, rr = (r - 21) % 3;
            if (c < 21)
                v = Yl[(rr * 33 + c) * Ncap + i];
            else if ((c - 21) / 3 == i)
                v = Fl[(rr * 3 + (c - 21) % 3) * Ncap + i];
        }
        Phi[r + (size_t)c * ldf] = v;
    } else {
        const int q = c - n;
        if (r < 21)
            v = E[r * 33 + 21 + q];
        else
            v = Yl[(((r - 21) % 3) * 33 + 21 + q) * Ncap + (r - 21) / 3];
        PhiB[r + (size_t)q * ldf] = v;
    }
}
// Sigma' = M + Phi_B diag(Q / dt) Phi_B^T + dt P   (M = Phi Sigma Phi^T already in Sout)
__global__ void __launch_bounds__(256) k_add_noise_dense(int n, int ld, int ldf, RiccatiArgs ra, const double* __restrict__ PhiB, double* __restrict__ Sout) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (r >= n)
        return;
    double bq = 0;
    for (int q = 0; q < 12; ++q)
        bq += PhiB[r + (size_t)q * ldf] * (ra.Qd[q] / ra.dt) * PhiB[c + (size_t)q * ldf];
    double v = Sout[r + (size_t)c * ld] + bq;
    if (r == c)
        v += ra.dt * (r < 21 ? ra.Pd[r / 3] : ra.Pd[7]);
    Sout[r + (size_t)c * ld] = v;
}

// NEES support: Z = [Sigma (lower, padded to even dimension np with a unit diagonal) ; eps^T] for the factorisation chain
template <typename TS>
__global__ void __launch_bounds__(256) k_build_nees(int n, int np, int ld, int ldzn, const TS* __restrict__ Sig, const double* __restrict__ eps,
                                                    double* __restrict__ Z) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (r > np || c >= np)
        return;
    double v;
    if (r == np)
        v = c < n ? eps[c] : 0.0;
    else if (r < n && c < n)
        v = Sig[r + (size_t)c * ld];
    else
        v = (r == c) ? 1.0 : 0.0;
    Z[r + (size_t)c * ldzn] = v;
}
__global__ void __launch_bounds__(256) k_sumsq_row(int np, int ldzn, const double* __restrict__ Wb, int row, double* __restrict__ out) {
    __shared__ double sp[256];
    double s = 0;
    for (int c = threadIdx.x; c < np; c += 256) {
        const double v = Wb[row + (size_t)c * ldzn];
        s += v * v;
    }
    sp[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st)
            sp[threadIdx.x] += sp[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[0] = sp[0];
}

// computeNEES when Sigma is positive definite only up to rounding (the factorisation chain reports a non-positive pivot): the reference
// inverts Sigma by partial-pivot LU and always returns a number (VIO_eqf.cpp:166-168), so the fallback is Gaussian elimination with
// partial pivoting on the augmented matrix [Sigma | eps], one launch per pivot. Sigma is symmetric, so the column-major Z = [Sigma ; eps^T]
// of k_build_nees is read as ROW-major M[i][j] = Z[j + i ldzn] (j = np is the right-hand side): rows are contiguous. Rows are never
// moved: a permutation (ping-pong between launches) maps logical to physical rows. Every workgroup repeats the pivot search (np strided
// reads); the pivot row is read-only in its launch and every other row belongs to exactly one workgroup.
constexpr int GE_ROWS = 4;
__global__ void __launch_bounds__(256) k_ge_step(int np, int ldzn, int k, double* __restrict__ Z, const int* __restrict__ perm_in, int* __restrict__ perm_out) {
    __shared__ double sv[256];
    __shared__ int si[256];
    double best = -1.0;
    int bi = k;
    for (int i = k + threadIdx.x; i < np; i += 256) {
        const double v = fabs(Z[k + (size_t)perm_in[i] * ldzn]);
        if (v > best) { // strict: the first maximum wins, as in LAPACK / Eigen
            best = v;
            bi = i;
        }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            const double o = sv[threadIdx.x + st];
            const int oi = si[threadIdx.x + st];
            if (o > sv[threadIdx.x] || (o == sv[threadIdx.x] && oi < si[threadIdx.x])) {
                sv[threadIdx.x] = o;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const int p = si[0];
    const int rk = perm_in[p]; // physical pivot row
    const double piv = Z[k + (size_t)rk * ldzn];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < np; i += 256)
            perm_out[i] = i == k ? rk : (i == p ? perm_in[k] : perm_in[i]);
    for (int r = 0; r < GE_ROWS; ++r) {
        const int i = k + 1 + blockIdx.x * GE_ROWS + r;
        if (i >= np)
            break;
        const int phys = i == p ? perm_in[k] : perm_in[i];
        double* row = Z + (size_t)phys * ldzn;
        const double* prow = Z + (size_t)rk * ldzn;
        const double l = row[k] / piv;
        for (int j = k + 1 + threadIdx.x; j <= np; j += 256)
            row[j] -= l * prow[j];
    }
}
__global__ void __launch_bounds__(256) k_iota(int n, int* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] = i;
}
// back substitution U x = b and NEES * n = eps . x (one workgroup; x in LDS)
__global__ void __launch_bounds__(1024) k_ge_back(int n, int np, int ldzn, const double* __restrict__ Z, const int* __restrict__ perm, const double* __restrict__ eps,
                                                  double* __restrict__ out) {
    extern __shared__ double xs[]; // np + 16
    __shared__ double red[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = np - 1; k >= 0; --k) {
        const double* row = Z + (size_t)perm[k] * ldzn;
        double s = 0.0;
        for (int j = k + 1 + threadIdx.x; j < np; j += 1024)
            s += row[j] * xs[j];
        for (int o = 32; o > 0; o >>= 1)
            s += __shfl_down(s, o, 64);
        if (lane == 0)
            red[wv] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < 16; ++w)
                t += red[w];
            xs[k] = (row[np] - t) / row[k];
        }
        __syncthreads();
    }
    double s = 0.0;
    for (int j = threadIdx.x; j < n; j += 1024)
        s += eps[j] * xs[j];
    for (int o = 32; o > 0; o >>= 1)
        s += __shfl_down(s, o, 64);
    if (lane == 0)
        red[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w)
            t += red[w];
        out[0] = t;
    }
}

// fp64 MFMA issue-rate micro-benchmark: 4 independent accumulators per wave, no memory traffic.
// clk (optional): the first wave of every 64th block leaves the shader-clock cycles (s_memtime) and the 100 MHz wall-clock ticks it ran for: their ratio is the
// shader clock the chip held UNDER THIS LOAD (the datasheet peak assumes 2.4 GHz).
__global__ void __launch_bounds__(256) k_mfma_peak(int iters, double* __restrict__ out, unsigned long long* __restrict__ clk) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    const d4 s = a0 + a1 + a2 + a3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (clk && threadIdx.x == 0 && (blockIdx.x & 63) == 0) {
        clk[2 * (blockIdx.x >> 6)] = clock64() - c0;
        clk[2 * (blockIdx.x >> 6) + 1] = wall_clock64() - w0;
    }
}

} // namespace eqf
