// engine.hpp -- per-env articulated-body physics step, compile-time specialised per robot model.
//
// One environment per SIMD lane: every function here is straight-line code for ONE env whose loops are
// statically unrolled over the generated `Model` tables (csrc/gen/model_*.h), so every per-lane array index is a
// literal and the compiler keeps the state in VGPRs (spilling the tail to scratch for the big robots).
//
// Replaces the closed `gym.simulate()` of the reference (call sites reference vec_task.py:382, ant.py:233-235).
// Same maths as oracle/physics.c, deliberately different formulation:
//   * branch-sparse joint-space inertia H (only ancestor pairs stored), factorised H = L^T L in place
//     (Featherstone's LTL ordering, which creates no fill-in on a kinematic tree);
//   * constraints solved in the *whitened* velocity w = L qd, where a row's Jacobian and its M^-1 J^T collapse
//     into one chain-sparse vector g = L^-T J^T  (J qd = g.w,  qd += M^-1 J^T dl  <=>  w += g dl);
//   * PGS sweeps touch only the kinematic chain of the contact body.
//
// The file is host+device: hipcc builds it into the kernels; tests/ also build it with g++ to debug the
// specialised code path on CPU against the oracle (never used by the product path).
#pragma once
#include <cmath>
#include <utility>

#if defined(__HIPCC__)
#define MI_HD __host__ __device__ __forceinline__
#define MI_LAMBDA __attribute__((always_inline))
#else
#define MI_HD inline __attribute__((always_inline))
#define MI_LAMBDA __attribute__((always_inline))
#endif

namespace mi {

template <class F, int... I>
MI_HD void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// statically unrolled loop: f(integral_constant<int,0>) ... f(integral_constant<int,N-1>)
template <int N, class F>
MI_HD void sfor(F&& f) {
    if constexpr (N > 0) sfor_impl(f, std::make_integer_sequence<int, N>{});
}
// descending: N-1 ... 0
template <int N, class F>
MI_HD void sfor_rev(F&& f) {
    sfor<N>([&](auto I) MI_LAMBDA { f(std::integral_constant<int, N - 1 - decltype(I)::value>{}); });
}

struct SimParams {
    float dt;
    int substeps, iters;
    float g[3];
    float contact_offset, rest_offset, max_depen_vel, erp, plane_mu, ground_z, cfm, warm;
};

MI_HD void cross3(const float* a, const float* b, float* o) {
    float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
MI_HD float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
MI_HD float dot6(const float* a, const float* b) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
MI_HD void quat2mat(const float* q, float* R) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
MI_HD void matvec3(const float* R, const float* v, float* o) {
    float x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
          z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
MI_HD void matTvec3(const float* R, const float* v, float* o) {
    float x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
          z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
MI_HD void matmul3(const float* A, const float* B, float* o) {
    float t[9];
    sfor<3>([&](auto I) MI_LAMBDA {
        sfor<3>([&](auto J) MI_LAMBDA {
            constexpr int i = I, j = J;
            t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
        });
    });
    sfor<9>([&](auto K) MI_LAMBDA { o[K] = t[K]; });
}
// spatial vectors are [ang(3); lin(3)], forces [moment(3); force(3)], all in world axes about O = root origin
MI_HD void crm(const float* V, const float* S, float* o) {
    float a[3], b[3], c[3];
    cross3(V, S, a); cross3(V, S + 3, b); cross3(V + 3, S, c);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
MI_HD void crf(const float* V, const float* F, float* o) {
    float a[3], b[3], c[3];
    cross3(V, F, a); cross3(V + 3, F + 3, b); cross3(V, F + 3, c);
    o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
struct SpI {  // spatial inertia about O: mass, h = m*c, I (xx yy zz xy xz yz)
    float m, h[3], I[6];
};
MI_HD void spi_mul(const SpI& I, const float* X, float* F) {
    const float *al = X, *a = X + 3;
    float Ia0 = I.I[0] * al[0] + I.I[3] * al[1] + I.I[4] * al[2];
    float Ia1 = I.I[3] * al[0] + I.I[1] * al[1] + I.I[5] * al[2];
    float Ia2 = I.I[4] * al[0] + I.I[5] * al[1] + I.I[2] * al[2];
    float hxa[3], hxal[3];
    cross3(I.h, a, hxa); cross3(I.h, al, hxal);
    F[0] = Ia0 + hxa[0]; F[1] = Ia1 + hxa[1]; F[2] = Ia2 + hxa[2];
    F[3] = I.m * a[0] - hxal[0]; F[4] = I.m * a[1] - hxal[1]; F[5] = I.m * a[2] - hxal[2];
}

template <class M>
struct Sim {
    static constexpr int NB = M::NB, ND = M::ND, NV = M::NV, OFF = M::OFF, NSPH = M::NSPH, NSENS = M::NSENS;
    static constexpr int NLIM = []() constexpr { int n = 0; for (int d = 0; d < ND; ++d) n += M::dof_limited[d] ? 1 : 0; return n; }();
    static constexpr int NROWG = (NLIM + 3 * NSPH) > 0 ? (NLIM + 3 * NSPH) : 1;

    // ---- persistent per-env state (lives in HBM between steps, SoA [field][env])
    float root[13];               // pos3, quat xyzw, linvel3, angvel3 (world)
    float q[M::NDA], qd[M::NDA];
    float lamc[3 * M::NSPHA];     // warm-start contact impulses (n, t1, t2)
    float laml[M::NDA];           // warm-start limit impulses (signed)
    // ---- per-step outputs
    float sensor[6 * M::NSENSA];  // force3, torque3 in the sensor body's frame
    float dof_force[M::NDA];

    static constexpr bool brot_is_identity(int b) {
        for (int k = 0; k < 9; ++k)
            if (M::brot[b][k] != ((k % 4 == 0) ? 1.f : 0.f)) return false;
        return true;
    }
    static constexpr int limrow(int d) {  // row slot of dof d's limit row (limited dofs only)
        int n = 0;
        for (int k = 0; k < d; ++k) n += M::dof_limited[k] ? 1 : 0;
        return n;
    }

    MI_HD void step(const SimParams& P, const float* tau) {
        const float h = P.dt / (float)P.substeps;
        for (int ss = 0; ss < P.substeps; ++ss) substep(P, tau, h);
    }

    MI_HD void substep(const SimParams& P, const float* tau, const float h) {
        float R[NB][9], r[NB][3];
        float S[M::NDA][6];
        // ------------------------------------------------------------ forward kinematics
        sfor<NB>([&](auto B) MI_LAMBDA {
            constexpr int b = B;
            float Rb[9], rb[3];
            if constexpr (b == 0) {
                quat2mat(root + 3, Rb);
                rb[0] = rb[1] = rb[2] = 0.f;
            } else {
                constexpr int p = M::parent[b];
                if constexpr (brot_is_identity(b)) {
                    sfor<9>([&](auto K) MI_LAMBDA { Rb[K] = R[p][K]; });
                } else {
                    matmul3(R[p], M::brot[b], Rb);
                }
                float t[3];
                matvec3(R[p], M::bpos[b], t);
                rb[0] = r[p][0] + t[0]; rb[1] = r[p][1] + t[1]; rb[2] = r[p][2] + t[2];
            }
            sfor<M::body_ndof[b]>([&](auto K) MI_LAMBDA {
                constexpr int d = M::body_dof0[b] + K;
                constexpr float ax = M::dof_axis[d][0], ay = M::dof_axis[d][1], az = M::dof_axis[d][2];
                const float al[3] = {ax, ay, az};
                const float anl[3] = {M::dof_anchor[d][0], M::dof_anchor[d][1], M::dof_anchor[d][2]};
                float a[3], ta[3], pt[3];
                matvec3(Rb, al, a);
                matvec3(Rb, anl, ta);
                pt[0] = rb[0] + ta[0]; pt[1] = rb[1] + ta[1]; pt[2] = rb[2] + ta[2];
                if constexpr (M::dof_type[d] == 0) {
                    float s, c;
                    sincosf(q[d], &s, &c);
                    const float t = 1.f - c;
                    // rotation about the (constant) local axis: Rb <- Rb * Q_local
                    const float Q[9] = {c + ax * ax * t, ax * ay * t - az * s, ax * az * t + ay * s,
                                        ay * ax * t + az * s, c + ay * ay * t, ay * az * t - ax * s,
                                        az * ax * t - ay * s, az * ay * t + ax * s, c + az * az * t};
                    matmul3(Rb, Q, Rb);
                    float tb[3];
                    matvec3(Rb, anl, tb);
                    rb[0] = pt[0] - tb[0]; rb[1] = pt[1] - tb[1]; rb[2] = pt[2] - tb[2];
                    S[d][0] = a[0]; S[d][1] = a[1]; S[d][2] = a[2];
                    cross3(pt, a, &S[d][3]);
                } else {
                    rb[0] += a[0] * q[d]; rb[1] += a[1] * q[d]; rb[2] += a[2] * q[d];
                    S[d][0] = S[d][1] = S[d][2] = 0.f;
                    S[d][3] = a[0]; S[d][4] = a[1]; S[d][5] = a[2];
                }
            });
            sfor<9>([&](auto K) MI_LAMBDA { R[b][K] = Rb[K]; });
            r[b][0] = rb[0]; r[b][1] = rb[1]; r[b][2] = rb[2];
        });
        // ------------------------------------------------------------ world spatial inertias, bias forces
        SpI Ic[NB];
        float bias[NV > 0 ? NV : 1];
        {
            float V[NB][6], A[NB][6], F[NB][6];
            sfor<NB>([&](auto B) MI_LAMBDA {
                constexpr int b = B;
                float t[3], c[3];
                matvec3(R[b], M::com[b], t);
                c[0] = r[b][0] + t[0]; c[1] = r[b][1] + t[1]; c[2] = r[b][2] + t[2];
                constexpr float ixx = M::inertia[b][0], iyy = M::inertia[b][1], izz = M::inertia[b][2],
                                ixy = M::inertia[b][3], ixz = M::inertia[b][4], iyz = M::inertia[b][5];
                const float Il[9] = {ixx, ixy, ixz, ixy, iyy, iyz, ixz, iyz, izz};
                float T[9];
                matmul3(R[b], Il, T);
                const float* Rb = R[b];
                // Iw = T * Rb^T (symmetric)
                float Iw0 = T[0] * Rb[0] + T[1] * Rb[1] + T[2] * Rb[2];
                float Iw4 = T[3] * Rb[3] + T[4] * Rb[4] + T[5] * Rb[5];
                float Iw8 = T[6] * Rb[6] + T[7] * Rb[7] + T[8] * Rb[8];
                float Iw1 = T[0] * Rb[3] + T[1] * Rb[4] + T[2] * Rb[5];
                float Iw2 = T[0] * Rb[6] + T[1] * Rb[7] + T[2] * Rb[8];
                float Iw5 = T[3] * Rb[6] + T[4] * Rb[7] + T[5] * Rb[8];
                constexpr float mm = M::mass[b];
                const float cc = dot3(c, c);
                SpI& I = Ic[b];
                I.m = mm; I.h[0] = mm * c[0]; I.h[1] = mm * c[1]; I.h[2] = mm * c[2];
                I.I[0] = Iw0 + mm * (cc - c[0] * c[0]); I.I[1] = Iw4 + mm * (cc - c[1] * c[1]);
                I.I[2] = Iw8 + mm * (cc - c[2] * c[2]);
                I.I[3] = Iw1 - mm * c[0] * c[1]; I.I[4] = Iw2 - mm * c[0] * c[2]; I.I[5] = Iw5 - mm * c[1] * c[2];
                // velocity / bias acceleration recursion
                float Vc[6], Ac[6];
                if constexpr (b == 0) {
                    if constexpr (M::FIXED) {
                        sfor<6>([&](auto K) MI_LAMBDA { Vc[K] = 0.f; Ac[K] = 0.f; });
                        Ac[3] = -P.g[0]; Ac[4] = -P.g[1]; Ac[5] = -P.g[2];
                    } else {
                        float wxv[3];
                        cross3(root + 10, root + 7, wxv);
                        Vc[0] = root[10]; Vc[1] = root[11]; Vc[2] = root[12];
                        Vc[3] = root[7]; Vc[4] = root[8]; Vc[5] = root[9];
                        Ac[0] = Ac[1] = Ac[2] = 0.f;
                        Ac[3] = -wxv[0] - P.g[0]; Ac[4] = -wxv[1] - P.g[1]; Ac[5] = -wxv[2] - P.g[2];
                    }
                } else {
                    constexpr int p = M::parent[b];
                    sfor<6>([&](auto K) MI_LAMBDA { Vc[K] = V[p][K]; Ac[K] = A[p][K]; });
                }
                sfor<M::body_ndof[b]>([&](auto K) MI_LAMBDA {
                    constexpr int d = M::body_dof0[b] + K;
                    float Sd[6];
                    crm(Vc, S[d], Sd);
                    sfor<6>([&](auto C) MI_LAMBDA { Ac[C] += Sd[C] * qd[d]; Vc[C] += S[d][C] * qd[d]; });
                });
                sfor<6>([&](auto K) MI_LAMBDA { V[b][K] = Vc[K]; A[b][K] = Ac[K]; });
                float IA[6], IV[6], X[6];
                spi_mul(I, Ac, IA); spi_mul(I, Vc, IV); crf(Vc, IV, X);
                sfor<6>([&](auto K) MI_LAMBDA { F[b][K] = IA[K] + X[K]; });
            });
            // subtree accumulation (forces and composite inertias), children before parents
            sfor_rev<NB>([&](auto B) MI_LAMBDA {
                constexpr int b = B;
                if constexpr (b > 0) {
                    constexpr int p = M::parent[b];
                    sfor<6>([&](auto K) MI_LAMBDA { F[p][K] += F[b][K]; });
                    Ic[p].m += Ic[b].m;
                    sfor<3>([&](auto K) MI_LAMBDA { Ic[p].h[K] += Ic[b].h[K]; });
                    sfor<6>([&](auto K) MI_LAMBDA { Ic[p].I[K] += Ic[b].I[K]; });
                }
            });
            sfor<ND>([&](auto D) MI_LAMBDA {
                constexpr int d = D;
                bias[OFF + d] = dot6(S[d], F[M::dof_body[d]]);
            });
            if constexpr (!M::FIXED) {
                bias[0] = F[0][3]; bias[1] = F[0][4]; bias[2] = F[0][5];
                bias[3] = F[0][0]; bias[4] = F[0][1]; bias[5] = F[0][2];
            }
        }
        // ------------------------------------------------------------ branch-sparse joint-space inertia H
        float L[M::NM];
        float Ldi[NV > 0 ? NV : 1];  // 1 / L_ii
        if constexpr (!M::FIXED) {
            const SpI& I = Ic[0];
            sfor<6>([&](auto A_) MI_LAMBDA {
                sfor<6>([&](auto B_) MI_LAMBDA {
                    constexpr int i = A_, j = B_;
                    if constexpr (j <= i) {
                        float v = 0.f;
                        if constexpr (i < 3) v = (i == j) ? I.m : 0.f;
                        else if constexpr (j < 3) {  // M_wv = [h]x : row i-3, col j
                            constexpr int rr = i - 3, cc = j;
                            if constexpr (rr == cc) v = 0.f;
                            else if constexpr (rr == 0 && cc == 1) v = -I.h[2];
                            else if constexpr (rr == 0 && cc == 2) v = I.h[1];
                            else if constexpr (rr == 1 && cc == 0) v = I.h[2];
                            else if constexpr (rr == 1 && cc == 2) v = -I.h[0];
                            else if constexpr (rr == 2 && cc == 0) v = -I.h[1];
                            else v = I.h[0];
                        } else {
                            constexpr int rr = i - 3, cc = j - 3;
                            constexpr int idx = (rr == cc) ? rr : ((rr + cc == 1) ? 3 : ((rr + cc == 2) ? 4 : 5));
                            v = I.I[idx];
                        }
                        L[M::midx[i][j]] = v;
                    }
                });
            });
        }
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            float F[6];
            spi_mul(Ic[M::dof_body[d]], S[d], F);
            L[M::midx[gi][gi]] = dot6(S[d], F);
            sfor<M::nanc[gi]>([&](auto K) MI_LAMBDA {
                constexpr int gj = M::anc[gi][K];
                float v;
                if constexpr (gj >= OFF) v = dot6(S[gj - OFF], F);
                else if constexpr (gj < 3) v = F[3 + gj];
                else v = F[gj - 3];
                L[M::midx[gi][gj]] = v;
            });
        });
        // ------------------------------------------------------------ rhs, implicit spring/damper on the diagonal
        float y[NV > 0 ? NV : 1];
        sfor<OFF>([&](auto I) MI_LAMBDA { y[I] = -bias[I]; });
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            constexpr float K = M::dof_stiffness[d], Dm = M::dof_damping[d];
            L[M::midx[gi][gi]] += M::dof_armature[d] + h * Dm + h * h * K;
            y[gi] = tau[d] - bias[gi] - K * (q[d] - M::dof_springref[d]) - (Dm + h * K) * qd[d];
        });
        // ------------------------------------------------------------ H = L^T L in place (no fill-in on a tree)
        sfor_rev<NV>([&](auto K_) MI_LAMBDA {
            constexpr int k = K_;
            const float dkk = sqrtf(fmaxf(L[M::midx[k][k]], 1e-30f));
            const float inv = 1.f / dkk;
            L[M::midx[k][k]] = dkk;
            Ldi[k] = inv;
            sfor<M::nanc[k]>([&](auto A_) MI_LAMBDA {
                constexpr int i = M::anc[k][A_];
                L[M::midx[k][i]] *= inv;
            });
            sfor<M::nanc[k]>([&](auto A_) MI_LAMBDA {
                constexpr int i = M::anc[k][A_];
                const float lki = L[M::midx[k][i]];
                L[M::midx[i][i]] -= lki * lki;
                sfor<M::nanc[i]>([&](auto B_) MI_LAMBDA {
                    constexpr int j = M::anc[i][B_];
                    L[M::midx[i][j]] -= lki * L[M::midx[k][j]];
                });
            });
        });
        // ------------------------------------------------------------ whitened velocity  w = L qd + h L^-T rhs
        float w[NV > 0 ? NV : 1];
        {
            float v[NV > 0 ? NV : 1];
            if constexpr (!M::FIXED) {
                v[0] = root[7]; v[1] = root[8]; v[2] = root[9]; v[3] = root[10]; v[4] = root[11]; v[5] = root[12];
            }
            sfor<ND>([&](auto D) MI_LAMBDA { v[OFF + D] = qd[D]; });
            sfor_rev<NV>([&](auto I_) MI_LAMBDA {
                constexpr int i = I_;
                const float z = y[i] * Ldi[i];
                sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA {
                    constexpr int j = M::anc[i][A_];
                    y[j] -= L[M::midx[i][j]] * z;
                });
                float s = L[M::midx[i][i]] * v[i];
                sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA {
                    constexpr int j = M::anc[i][A_];
                    s += L[M::midx[i][j]] * v[j];
                });
                w[i] = s + h * z;
            });
        }
        // ------------------------------------------------------------ constraint rows in whitened space
        float G[NROWG][M::MAXCHAIN];
        float Ainv[NROWG], vt[NROWG], lam[NROWG];
        bool act[M::NSPHA];
        // solve L^T g = J^T restricted to a chain (descending generalized indices), in place in g[]
        auto chain_solve = [&](auto B, float* g) MI_LAMBDA {
            constexpr int b = decltype(B)::value;
            sfor<M::chain_len[b]>([&](auto K) MI_LAMBDA {
                constexpr int k = K, i = M::chain[b][k];
                const float z = g[k] * Ldi[i];
                g[k] = z;
                // ancestors of i are exactly the later chain entries
                sfor<M::chain_len[b] - 1 - k>([&](auto T) MI_LAMBDA {
                    constexpr int kk = k + 1 + T, j = M::chain[b][kk];
                    g[kk] -= L[M::midx[i][j]] * z;
                });
            });
        };
        // limits: one speculative row per limited dof (nearest bound), chain = dof + its ancestors
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (M::dof_limited[d]) {
                constexpr int row = limrow(d);
                const float dl = q[d] - M::dof_lower[d], du = M::dof_upper[d] - q[d];
                const bool lower = dl < du;
                const float C = lower ? dl : du, s = lower ? 1.f : -1.f;
                if (laml[d] * s < 0.f) laml[d] = 0.f;
                // g over [gi, anc(gi)...]
                float g[M::MAXCHAIN];
                g[0] = s * Ldi[gi];
                sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA {
                    constexpr int a = A_;
                    g[1 + a] = 0.f;
                });
                // propagate: same recursion as chain_solve along [gi, anc...]
                sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA {
                    constexpr int k = K;
                    constexpr int i = (k == 0) ? gi : M::anc[gi][k == 0 ? 0 : k - 1];
                    if constexpr (k > 0) g[k] *= Ldi[i];
                    const float z = g[k];
                    sfor<M::nanc[gi] - k>([&](auto T) MI_LAMBDA {
                        constexpr int kk = k + 1 + T, j = M::anc[gi][kk - 1];
                        g[kk] -= L[M::midx[i][j]] * z;
                    });
                });
                float a = P.cfm;
                sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { a += g[K] * g[K]; G[row][K] = g[K]; });
                Ainv[row] = 1.f / a;
                vt[row] = (C >= 0.f) ? -C / h : fminf(-C * P.erp / h, P.max_depen_vel);
                const float l0 = fabsf(laml[d]) * P.warm;
                lam[row] = l0;
                w[gi] += g[0] * l0;
                sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += g[1 + A_] * l0; });
            } else {
                laml[d] = 0.f;
            }
        });
        // ground contacts: 3 rows per sphere (normal +z, tangents x, y)
        float xcs[M::NSPHA][3];
        sfor<NSPH>([&](auto S_) MI_LAMBDA {
            constexpr int s = S_, b = M::sph_body[s], row0 = NLIM + 3 * s;
            float t[3], x[3];
            matvec3(R[b], M::sph_pos[s], t);
            x[0] = r[b][0] + t[0]; x[1] = r[b][1] + t[1]; x[2] = r[b][2] + t[2];
            const float dist = (root[2] + x[2]) - M::sph_rad[s] - P.ground_z;
            const bool on = dist < P.contact_offset;
            act[s] = on;
            xcs[s][0] = x[0]; xcs[s][1] = x[1]; xcs[s][2] = x[2] - M::sph_rad[s];
            if (!on) {
                lamc[3 * s] = lamc[3 * s + 1] = lamc[3 * s + 2] = 0.f;
                sfor<3>([&](auto K) MI_LAMBDA { lam[row0 + K] = 0.f; });
            } else {
                const float gap = dist - P.rest_offset;
                const float* xc = xcs[s];
                sfor<3>([&](auto K) MI_LAMBDA {
                    constexpr int k = K, row = row0 + k;
                    // unit force u at xc as a spatial force [xc x u; u]; u = z, x, y
                    constexpr int ax = (k == 0) ? 2 : (k == 1 ? 0 : 1);
                    float W[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    W[3 + ax] = 1.f;
                    if constexpr (ax == 0) { W[1] = xc[2]; W[2] = -xc[1]; }
                    else if constexpr (ax == 1) { W[0] = -xc[2]; W[2] = xc[0]; }
                    else { W[0] = xc[1]; W[1] = -xc[0]; }
                    float g[M::MAXCHAIN];
                    sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA {
                        constexpr int gi = M::chain[b][C];
                        if constexpr (gi >= OFF) g[C] = dot6(S[gi - OFF], W);
                        else if constexpr (gi < 3) g[C] = W[3 + gi];
                        else g[C] = W[gi - 3];
                    });
                    chain_solve(std::integral_constant<int, b>{}, g);
                    float a = P.cfm;
                    sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { a += g[C] * g[C]; G[row][C] = g[C]; });
                    Ainv[row] = 1.f / a;
                    vt[row] = (k == 0) ? ((gap >= 0.f) ? -gap / h : fminf(-gap * P.erp / h, P.max_depen_vel)) : 0.f;
                    const float l0 = lamc[3 * s + k] * P.warm;
                    lam[row] = l0;
                    sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += g[C] * l0; });
                });
            }
        });
        // ------------------------------------------------------------ projected Gauss-Seidel sweeps
        for (int it = 0; it < P.iters; ++it) {
            sfor<ND>([&](auto D) MI_LAMBDA {
                constexpr int d = D, gi = OFF + d;
                if constexpr (M::dof_limited[d]) {
                    constexpr int row = limrow(d);
                    float vn = G[row][0] * w[gi];
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { vn += G[row][1 + A_] * w[M::anc[gi][A_]]; });
                    const float nl = fmaxf(lam[row] - (vn - vt[row]) * Ainv[row], 0.f);
                    const float dl = nl - lam[row];
                    lam[row] = nl;
                    w[gi] += G[row][0] * dl;
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += G[row][1 + A_] * dl; });
                }
            });
            sfor<NSPH>([&](auto S_) MI_LAMBDA {
                constexpr int s = S_, b = M::sph_body[s], row0 = NLIM + 3 * s;
                if (act[s]) {
                    const float mu = 0.5f * (M::sph_mu[s] + P.plane_mu);
                    {
                        float vn = 0.f;
                        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { vn += G[row0][C] * w[M::chain[b][C]]; });
                        const float nl = fmaxf(lam[row0] - (vn - vt[row0]) * Ainv[row0], 0.f);
                        const float dl = nl - lam[row0];
                        lam[row0] = nl;
                        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += G[row0][C] * dl; });
                    }
                    float lt[2];
                    sfor<2>([&](auto K) MI_LAMBDA {
                        constexpr int row = row0 + 1 + K;
                        float vn = 0.f;
                        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { vn += G[row][C] * w[M::chain[b][C]]; });
                        const float dl = -(vn - vt[row]) * Ainv[row];
                        lt[K] = lam[row] + dl;
                        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += G[row][C] * dl; });
                    });
                    const float lim = mu * lam[row0];
                    const float nrm = sqrtf(lt[0] * lt[0] + lt[1] * lt[1]);
                    const float sc = (nrm > lim) ? lim / fmaxf(nrm, 1e-30f) : 1.f;
                    sfor<2>([&](auto K) MI_LAMBDA {
                        constexpr int row = row0 + 1 + K;
                        const float nl = lt[K] * sc, dl = nl - lt[K];
                        lam[row] = nl;
                        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += G[row][C] * dl; });
                    });
                }
            });
        }
        // ------------------------------------------------------------ back to generalised velocity: qd = L^-1 w
        float v[NV > 0 ? NV : 1];
        sfor<NV>([&](auto I_) MI_LAMBDA {
            constexpr int i = I_;
            float s = w[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA {
                constexpr int j = M::anc[i][A_];
                s -= L[M::midx[i][j]] * v[j];
            });
            v[i] = s * Ldi[i];
        });
        // ------------------------------------------------------------ impulses -> warm start, sensors, dof forces
        const float invh = 1.f / h;
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            float ll = 0.f;
            if constexpr (M::dof_limited[d]) {
                constexpr int row = limrow(d);
                const float dl = q[d] - M::dof_lower[d], du = M::dof_upper[d] - q[d];
                ll = (dl < du) ? lam[row] : -lam[row];
                laml[d] = ll;
            }
            dof_force[d] = tau[d] - M::dof_stiffness[d] * (q[d] - M::dof_springref[d]) - M::dof_damping[d] * v[OFF + d] + ll * invh;
        });
        sfor<6 * NSENS>([&](auto K) MI_LAMBDA { sensor[K] = 0.f; });
        sfor<NSPH>([&](auto S_) MI_LAMBDA {
            constexpr int s = S_, b = M::sph_body[s], row0 = NLIM + 3 * s;
            if (act[s]) {
                lamc[3 * s] = lam[row0]; lamc[3 * s + 1] = lam[row0 + 1]; lamc[3 * s + 2] = lam[row0 + 2];
                sfor<NSENS>([&](auto K) MI_LAMBDA {
                    constexpr int k = K;
                    if constexpr (M::sens_body[k] == b) {
                        const float f[3] = {lam[row0 + 1] * invh, lam[row0 + 2] * invh, lam[row0] * invh};
                        const float arm[3] = {xcs[s][0] - r[b][0], xcs[s][1] - r[b][1], xcs[s][2] - r[b][2]};
                        float tq[3], fl[3], tl[3];
                        cross3(arm, f, tq);
                        matTvec3(R[b], f, fl); matTvec3(R[b], tq, tl);
                        sfor<3>([&](auto C) MI_LAMBDA { sensor[6 * k + C] += fl[C]; sensor[6 * k + 3 + C] += tl[C]; });
                    }
                });
            }
        });
        // ------------------------------------------------------------ integrate (semi-implicit Euler)
        sfor<ND>([&](auto D) MI_LAMBDA { qd[D] = v[OFF + D]; q[D] += h * qd[D]; });
        if constexpr (!M::FIXED) {
            sfor<3>([&](auto K) MI_LAMBDA { root[7 + K] = v[K]; root[10 + K] = v[3 + K]; root[K] += h * v[K]; });
            const float om[3] = {v[3], v[4], v[5]};
            const float an = sqrtf(dot3(om, om)), th = an * h;
            float dq[4];
            if (th > 1e-12f) {
                float sn, cs;
                sincosf(0.5f * th, &sn, &cs);
                const float k = sn / an;
                dq[0] = om[0] * k; dq[1] = om[1] * k; dq[2] = om[2] * k; dq[3] = cs;
            } else {
                dq[0] = om[0] * h * 0.5f; dq[1] = om[1] * h * 0.5f; dq[2] = om[2] * h * 0.5f; dq[3] = 1.f;
            }
            float* Q = root + 3;
            const float x = dq[3] * Q[0] + dq[0] * Q[3] + dq[1] * Q[2] - dq[2] * Q[1];
            const float yy = dq[3] * Q[1] - dq[0] * Q[2] + dq[1] * Q[3] + dq[2] * Q[0];
            const float z = dq[3] * Q[2] + dq[0] * Q[1] - dq[1] * Q[0] + dq[2] * Q[3];
            const float ww = dq[3] * Q[3] - dq[0] * Q[0] - dq[1] * Q[1] - dq[2] * Q[2];
            const float n = 1.f / sqrtf(x * x + yy * yy + z * z + ww * ww);
            Q[0] = x * n; Q[1] = yy * n; Q[2] = z * n; Q[3] = ww * n;
        }
    }
};

}  // namespace mi
