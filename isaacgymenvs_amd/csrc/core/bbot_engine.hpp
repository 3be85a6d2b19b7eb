// bbot_engine.hpp -- physics sub-step of the BallBalance task: a free tray on three two-joint legs whose feet are pinned, and ONE
// free ball that lands on the tray.
//
// Replaces gym.simulate() for reference isaacgymenvs/tasks/ball_balance.py.  Built on the pieces of core/engine.hpp (tree pass,
// branch-sparse L^T L factor, whitened velocity, chain-sparse rows, PGS); what is specific here:
//   * implicit PD position drives on the three lower-leg joints only (DOF_MODE_POS, stiffness 4000, damping 100; the upper-leg
//     joints are DOF_MODE_NONE, ball_balance.py:271-281);
//   * the rigid-body attractors that hold the far end of every lower leg at a world point (stiffness k = 5e7, damping c = 5e3,
//     AXIS_TRANSLATION, :285-300) as implicit spring-dampers = soft equality rows
//         J v+ + gamma lam = -beta x,    gamma = 1 / (h (h k + c)),    beta = k / (h k + c),    lam = h * force,
//     three rows (world x, y, z) per foot over the chain [lower-leg joint, upper-leg joint, 6 tray dofs], warm started; the
//     closed loops tray - leg - ground are thus solved by the same Gauss-Seidel sweeps as the joint limits;
//   * the ball: a free sphere (whitening = constant diagonal), gravity on; contact = ball against the tray's solid cylinder
//     (exact closest point), 3 rows (normal + friction disc) over [6 tray dofs | 6 ball dofs], no warm start;
//   * the three force sensors of the reference sit on the tray (:254-260): the net non-gravity wrench on the tray over the
//     sub-step (leg joints + ball contact) from its momentum balance, in the tray frame about each sensor origin.
// The model is small (12 generalised velocities, 18 rows of at most 8 + 6 entries): every row stays in registers, no LDS.
// Same maths as oracle/bbot.py (dense, numpy, fp64).
#pragma once
#include "engine.hpp"

namespace mi {

struct BallState {          // world frame
    float pos[3], quat[4], vel[3], angvel[3];
};
struct BbotPhys {           // the physics part of MiBallBalanceParams (same layout, include/mi_engine.h)
    float pin_stiffness, pin_damping;      // attractor (ball_balance.py:287-288)
    float drive_kp, drive_kd;              // DOF_MODE_POS drive of the actuated dofs (:276-277)
    int actuated_mask;                     // bit d set: dof d is position driven (:271: dofs 1, 3, 5)
    float ball_radius, ball_mass, ball_inertia, mu;
    float tray_radius, tray_half;          // the tray's collision cylinder
    float pin_offset[3];                   // attractor offset in the lower leg's frame (:299)
    float pin_target[3][3];                // attractor targets, env frame (:293-297)
    float sensor_pos[3][3];                // force-sensor origins in the tray frame (:256-259)
};

template <class M>
struct BbotSim : Sim<M> {
    using B = Sim<M>;
    static constexpr int NB = M::NB, ND = M::ND, NV = M::NV, OFF = M::OFF, NLIM = B::NLIM, NVA = B::NVA, NPIN = 9, CH = M::MAXCHAIN;
    static_assert(M::FIXED == 0 && M::NSPH == 0 && M::NSENS == 3 && OFF == 6, "BbotSim: floating base, three pinned feet as sensor bodies");

    BallState ball;

    // sphere (centre c in the cylinder's frame, axis z) vs solid cylinder: signed distance, outward normal (cylinder frame)
    MI_HD static void sphere_cylinder(const float* c, float r, float radius, float half, float* dist, float* n) {
        const float rho2 = c[0] * c[0] + c[1] * c[1];
        const float s = fminf(1.f, radius * MI_RSQ(fmaxf(rho2, 1e-30f)));
        const float d[3] = {c[0] - c[0] * s, c[1] - c[1] * s, c[2] - fminf(fmaxf(c[2], -half), half)};
        const float d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const bool outside = d2 > 1e-24f;
        const float inv = MI_RSQ(fmaxf(d2, 1e-30f));
        // centre inside the solid: leave through the nearer flat face
        *dist = outside ? d2 * inv - r : -(half - fabsf(c[2])) - r;
        n[0] = outside ? d[0] * inv : 0.f;
        n[1] = outside ? d[1] * inv : 0.f;
        n[2] = outside ? d[2] * inv : (c[2] >= 0.f ? 1.f : -1.f);
    }

    // one sub-step of length h.  target [ND]: position targets; laml [ND] / lamp [9]: warm-start impulses (read and rewritten);
    // sensor [18]: force (3) + torque (3) of the three tray sensors; *ncontact: 1 while the ball touches the tray
    MI_HD void substep(const SimParams& P, const BbotPhys& bp, const float h, const float* target, const Strided laml, const Strided lamp,
                       const Strided sensor, int* ncontact) {
        float (&root)[13] = this->root;      // (gcc does not look through `using B::q` inside generic lambdas)
        float (&q)[M::NDA] = this->q;
        float (&qd)[M::NDA] = this->qd;
        const float invh = MI_RCP(h);
        typename B::Ctx c;
        float (&S)[M::NDA][6] = c.S;
        float (&L)[M::NM] = c.L;
        {
            SpI Iroot;
            float Froot[6];
            this->template body_pass<0>(P, c, nullptr, nullptr, nullptr, nullptr, Iroot, Froot);
        }
        MI_PHASE();
        const float v0[6] = {root[7], root[8], root[9], root[10], root[11], root[12]};
        // tray inertia about its origin (= its centre of mass) in world axes, before the factorisation overwrites H's root block.
        // The root block of H holds the COMPOSITE inertia, so take the tray's own from the model constants instead.
        float Rt[9];
        quat2mat(root + 3, Rt);
        // ------------------------------------------------------------ rhs, implicit passive terms and position drives
        float Ldi[NVA], y[NVA];
        sfor<OFF>([&](auto I) MI_LAMBDA { y[I] = -c.bias[I]; });
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            const bool act = (bp.actuated_mask >> d) & 1;
            const float kp = act ? bp.drive_kp : 0.f, kd = M::dof_damping[d] + (act ? bp.drive_kd : 0.f);
            L[M::midx[gi][gi]] += M::dof_armature[d] + h * kd + h * h * kp;
            y[gi] = -c.bias[gi] + kp * (target[d] - q[d]) - (kd + h * kp) * qd[d];
        });
        // ------------------------------------------------------------ H = L^T L in place (no fill-in on a tree)
        sfor_rev<NV>([&](auto K_) MI_LAMBDA {
            constexpr int k = K_;
            const float dk2 = fmaxf(L[M::midx[k][k]], 1e-30f);
            const float inv = MI_RSQ(dk2);
            L[M::midx[k][k]] = dk2 * inv;
            Ldi[k] = inv;
            sfor<M::nanc[k]>([&](auto A_) MI_LAMBDA { L[M::midx[k][M::anc[k][A_]]] *= inv; });
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
        MI_PHASE();
        // ------------------------------------------------------------ whitened velocities: bot w = L v + h L^-T rhs, ball wb
        float w[NVA];
        {
            float v[NVA];
            sfor<6>([&](auto K) MI_LAMBDA { v[K] = v0[K]; });
            sfor<ND>([&](auto D) MI_LAMBDA { v[OFF + D] = qd[D]; });
            sfor_rev<NV>([&](auto I_) MI_LAMBDA {
                constexpr int i = I_;
                const float z = y[i] * Ldi[i];
                sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { y[M::anc[i][A_]] -= L[M::midx[i][M::anc[i][A_]]] * z; });
                float s = L[M::midx[i][i]] * v[i];
                sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { s += L[M::midx[i][M::anc[i][A_]]] * v[M::anc[i][A_]]; });
                w[i] = s + h * z;
            });
        }
        const float sm = MI_SQRT(bp.ball_mass), si = MI_SQRT(bp.ball_inertia);
        const float ism = MI_RCP(sm), isi = MI_RCP(si);
        float wb[6];
        sfor<3>([&](auto K) MI_LAMBDA { wb[K] = sm * (ball.vel[K] + h * P.g[K]); wb[3 + K] = si * ball.angvel[K]; });
        MI_PHASE();
        // solve L^T g = J^T restricted to a chain (descending generalized indices), in place in g[]
        auto chain_solve = [&](auto B_, float* g) MI_LAMBDA {
            constexpr int b = decltype(B_)::value;
            sfor<M::chain_len[b]>([&](auto K) MI_LAMBDA {
                constexpr int k = K, i = M::chain[b][k];
                const float z = g[k] * Ldi[i];
                g[k] = z;
                sfor<M::chain_len[b] - 1 - k>([&](auto T) MI_LAMBDA {
                    constexpr int kk = k + 1 + T, j = M::chain[b][kk];
                    g[kk] -= L[M::midx[i][j]] * z;
                });
            });
        };
        // Jacobian row of a unit force u at the point x (relative to O) of body b, over the chain of b
        auto point_row = [&](auto B_, const float* x, const float* u, float* g) MI_LAMBDA {
            constexpr int b = decltype(B_)::value;
            float W[6];
            cross3(x, u, W);
            W[3] = u[0]; W[4] = u[1]; W[5] = u[2];
            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA {
                constexpr int gi = M::chain[b][C];
                if constexpr (gi >= OFF) g[C] = dot6(S[gi - OFF], W);
                else if constexpr (gi < 3) g[C] = W[3 + gi];
                else g[C] = W[gi - 3];
            });
        };
        // ------------------------------------------------------------ joint limit rows (as core/engine.hpp)
        float Gl[NLIM][CH], Al[NLIM], vl[NLIM], ll[NLIM];
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (M::dof_limited[d]) {
                constexpr int row = B::limrow(d);
                const float dl = q[d] - M::dof_lower[d], du = M::dof_upper[d] - q[d];
                const bool lower = dl < du;
                const float C = lower ? dl : du, s = lower ? 1.f : -1.f;
                const float lw = laml(d);
                float* g = Gl[row];
                g[0] = s * Ldi[gi];
                sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { g[1 + A_] = 0.f; });
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
                sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { a += g[K] * g[K]; });
                Al[row] = MI_RCP(a);
                vl[row] = (C >= 0.f) ? -C * invh : fminf(-C * P.erp * invh, P.max_depen_vel);
                ll[row] = ((lw * s < 0.f) ? 0.f : fabsf(lw)) * P.warm;
            }
        });
        MI_PHASE();
        // ------------------------------------------------------------ attractor rows: foot J, world axis K
        const float gamma = MI_RCP(h * (h * bp.pin_stiffness + bp.pin_damping));
        const float beta = bp.pin_stiffness * MI_RCP(h * bp.pin_stiffness + bp.pin_damping);
        float Gp[NPIN][CH], Ap[NPIN], vp[NPIN], lp[NPIN];
        sfor<3>([&](auto J_) MI_LAMBDA {
            constexpr int j = J_, b = M::sens_body[j];
            static_assert(M::chain_len[b] == CH, "a lower leg's chain: its joint, the upper-leg joint, 6 tray dofs");
            float t[3], x[3];
            matvec3(c.Rs[j], bp.pin_offset, t);
            sfor<3>([&](auto K) MI_LAMBDA { x[K] = c.rs[j][K] + t[K]; });
            sfor<3>([&](auto K) MI_LAMBDA {
                constexpr int k = K, row = 3 * j + k;
                float u[3] = {0.f, 0.f, 0.f};
                u[k] = 1.f;
                point_row(std::integral_constant<int, b>{}, x, u, Gp[row]);
                chain_solve(std::integral_constant<int, b>{}, Gp[row]);
                float a = P.cfm + gamma;
                sfor<CH>([&](auto C) MI_LAMBDA { a += Gp[row][C] * Gp[row][C]; });
                Ap[row] = MI_RCP(a);
                vp[row] = -beta * ((root[k] + x[k]) - bp.pin_target[j][k]);
                lp[row] = lamp(row) * P.warm;
            });
            MI_PHASE();
        });
        // ------------------------------------------------------------ the ball against the tray: 3 rows over [6 tray dofs | 6 ball dofs]
        float Gt[3][6], Gb[3][6], Ac[3], vtn, lc[3] = {0.f, 0.f, 0.f};
        bool on;
        {
            const float rel[3] = {ball.pos[0] - root[0], ball.pos[1] - root[1], ball.pos[2] - root[2]};
            float cl[3], nl[3], dist, fr[3][3];
            matTvec3(Rt, rel, cl);
            sphere_cylinder(cl, bp.ball_radius, bp.tray_radius, bp.tray_half, &dist, nl);
            on = dist < P.contact_offset;
            matvec3(Rt, nl, fr[0]);                        // from the tray towards the ball
            contact_frame(fr[0], fr[1], fr[2]);
            float pc[3], rc[3];
            sfor<3>([&](auto K) MI_LAMBDA { rc[K] = -bp.ball_radius * fr[0][K]; pc[K] = rel[K] + rc[K]; });
            const float gap = dist - P.rest_offset;
            vtn = (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
            // an env whose ball is away builds the rows like any other and makes them inert with Ainv = 0 (no divergence)
            const float onf = on ? 1.f : 0.f;
            sfor<3>([&](auto K) MI_LAMBDA {
                constexpr int k = K;
                static_assert(M::chain_len[0] == 6, "tray chain = its 6 root dofs");
                point_row(std::integral_constant<int, 0>{}, pc, fr[k], Gt[k]);
                sfor<6>([&](auto C) MI_LAMBDA { Gt[k][C] = -Gt[k][C]; });
                chain_solve(std::integral_constant<int, 0>{}, Gt[k]);
                float cx[3];
                cross3(rc, fr[k], cx);
                sfor<3>([&](auto I_) MI_LAMBDA { Gb[k][I_] = fr[k][I_] * ism; Gb[k][3 + I_] = cx[I_] * isi; });
                float a = P.cfm;
                sfor<6>([&](auto C) MI_LAMBDA { a += Gt[k][C] * Gt[k][C] + Gb[k][C] * Gb[k][C]; });
                Ac[k] = onf * MI_RCP(a);
            });
        }
        *ncontact = on ? 1 : 0;
        MI_PHASE();
        // ------------------------------------------------------------ warm start: w += G^T lam0
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (M::dof_limited[d]) {
                constexpr int row = B::limrow(d);
                w[gi] += Gl[row][0] * ll[row];
                sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += Gl[row][1 + A_] * ll[row]; });
            }
        });
        sfor<3>([&](auto J_) MI_LAMBDA {
            constexpr int b = M::sens_body[J_];
            sfor<3>([&](auto K) MI_LAMBDA {
                constexpr int row = 3 * J_ + K;
                sfor<CH>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += Gp[row][C] * lp[row]; });
            });
        });
        MI_PHASE();
        // ------------------------------------------------------------ projected Gauss-Seidel sweeps: limits, attractors, contact
        for (int it = 0; it < P.iters; ++it) {
            sfor<ND>([&](auto D) MI_LAMBDA {
                constexpr int d = D, gi = OFF + d;
                if constexpr (M::dof_limited[d]) {
                    constexpr int row = B::limrow(d);
                    float vn = Gl[row][0] * w[gi];
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { vn += Gl[row][1 + A_] * w[M::anc[gi][A_]]; });
                    const float nl = fmaxf(ll[row] - (vn - vl[row]) * Al[row], 0.f);
                    const float dl = nl - ll[row];
                    ll[row] = nl;
                    w[gi] += Gl[row][0] * dl;
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += Gl[row][1 + A_] * dl; });
                }
            });
            sfor<3>([&](auto J_) MI_LAMBDA {
                constexpr int b = M::sens_body[J_];
                sfor<3>([&](auto K) MI_LAMBDA {
                    constexpr int row = 3 * J_ + K;
                    float vn = 0.f;
                    sfor<CH>([&](auto C) MI_LAMBDA { vn += Gp[row][C] * w[M::chain[b][C]]; });
                    const float dl = -((vn - vp[row]) + gamma * lp[row]) * Ap[row];
                    lp[row] += dl;
                    sfor<CH>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += Gp[row][C] * dl; });
                });
            });
            {
                auto vrow = [&](auto K_) MI_LAMBDA -> float {
                    constexpr int k = decltype(K_)::value;
                    float vn = 0.f;
                    sfor<6>([&](auto C) MI_LAMBDA { vn += Gt[k][C] * w[M::chain[0][C]] + Gb[k][C] * wb[C]; });
                    return vn;
                };
                auto push = [&](auto K_, float dl) MI_LAMBDA {
                    constexpr int k = decltype(K_)::value;
                    sfor<6>([&](auto C) MI_LAMBDA { w[M::chain[0][C]] += Gt[k][C] * dl; wb[C] += Gb[k][C] * dl; });
                };
                using I0 = std::integral_constant<int, 0>;
                const float ln = fmaxf(lc[0] - (vrow(I0{}) - vtn) * Ac[0], 0.f);
                push(I0{}, ln - lc[0]);
                lc[0] = ln;
                float lt[2], vtg[2];
                sfor<2>([&](auto K) MI_LAMBDA {      // both tangent rows from the same velocity, then the disc (core/engine.hpp friction_disc), one application
                    using IK = std::integral_constant<int, 1 + decltype(K)::value>;
                    vtg[K] = vrow(IK{});
                    lt[K] = lc[1 + K] - vtg[K] * Ac[1 + K];
                });
                friction_disc(lt, lc[1], lc[2], vtg[0], vtg[1], Ac[1], Ac[2], bp.mu * ln);
                sfor<2>([&](auto K) MI_LAMBDA {
                    using IK = std::integral_constant<int, 1 + decltype(K)::value>;
                    const float nl = lt[K], dl = nl - lc[1 + K];
                    lc[1 + K] = nl;
                    push(IK{}, dl);
                });
            }
        }
        MI_PHASE();
        // ------------------------------------------------------------ back to generalised velocity: v = L^-1 w
        float v[NVA];
        sfor<NV>([&](auto I_) MI_LAMBDA {
            constexpr int i = I_;
            float s = w[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { s -= L[M::midx[i][M::anc[i][A_]]] * v[M::anc[i][A_]]; });
            v[i] = s * Ldi[i];
        });
        // ------------------------------------------------------------ impulses -> warm start; tray sensors
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            float x = 0.f;
            if constexpr (M::dof_limited[d]) {
                const float dl = q[d] - M::dof_lower[d], du = M::dof_upper[d] - q[d];
                x = (dl < du) ? ll[B::limrow(d)] : -ll[B::limrow(d)];
            }
            laml(d) = x;
        });
        sfor<NPIN>([&](auto R) MI_LAMBDA { lamp(R) = lp[R]; });
        {
            // net non-gravity wrench on the tray (its centre of mass is the body origin): F = m (dv / h - g), T = I dw / h + w x I w
            constexpr float mt = M::mass[0];
            const float Il[3] = {M::inertia[0][0], M::inertia[0][1], M::inertia[0][2]};
            static_assert(M::inertia[0][3] == 0.f && M::inertia[0][4] == 0.f && M::inertia[0][5] == 0.f && M::com[0][0] == 0.f && M::com[0][1] == 0.f && M::com[0][2] == 0.f,
                          "tray: principal axes, centre of mass at the origin");
            float F[3], dw[3], T[3], t0[3], t1[3];
            sfor<3>([&](auto K) MI_LAMBDA { F[K] = mt * ((v[K] - v0[K]) * invh - P.g[K]); dw[K] = (v[3 + K] - v0[3 + K]) * invh; });
            auto Iw = [&](const float* a, float* o) MI_LAMBDA {      // R diag(I) R^T a
                float l[3];
                matTvec3(Rt, a, l);
                sfor<3>([&](auto K) MI_LAMBDA { l[K] *= Il[K]; });
                matvec3(Rt, l, o);
            };
            Iw(dw, t0);
            Iw(v0 + 3, t1);
            cross3(v0 + 3, t1, T);
            sfor<3>([&](auto K) MI_LAMBDA { T[K] += t0[K]; });
            float Fl[3];
            matTvec3(Rt, F, Fl);
            sfor<3>([&](auto I_) MI_LAMBDA {
                float rw[3], rxF[3], Ti[3], Tl[3];
                matvec3(Rt, bp.sensor_pos[I_], rw);
                cross3(rw, F, rxF);
                sfor<3>([&](auto K) MI_LAMBDA { Ti[K] = T[K] - rxF[K]; });
                matTvec3(Rt, Ti, Tl);
                sfor<3>([&](auto K) MI_LAMBDA { sensor(6 * I_ + K) = Fl[K]; sensor(6 * I_ + 3 + K) = Tl[K]; });
            });
        }
        MI_PHASE();
        // ------------------------------------------------------------ integrate (semi-implicit Euler)
        sfor<ND>([&](auto D) MI_LAMBDA { qd[D] = v[OFF + D]; q[D] += h * qd[D]; });
        sfor<3>([&](auto K) MI_LAMBDA { root[7 + K] = v[K]; root[10 + K] = v[3 + K]; root[K] += h * v[K]; });
        integrate_quat(root + 3, v + 3, h);
        sfor<3>([&](auto K) MI_LAMBDA {
            ball.vel[K] = wb[K] * ism; ball.angvel[K] = wb[3 + K] * isi;
            ball.pos[K] += h * ball.vel[K];
        });
        integrate_quat(ball.quat, ball.angvel, h);
    }

    MI_HD static void integrate_quat(float* Q, const float* om, const float h) {
        const float an = MI_SQRT(dot3(om, om)), th = an * h;
        float sn, cs;
        sincosf(0.5f * th, &sn, &cs);
        const bool big = th > 1e-12f;
        const float k = big ? sn * MI_RCP(fmaxf(an, 1e-30f)) : 0.5f * h;
        const float dq[4] = {om[0] * k, om[1] * k, om[2] * k, big ? cs : 1.f};
        const float x = dq[3] * Q[0] + dq[0] * Q[3] + dq[1] * Q[2] - dq[2] * Q[1];
        const float yy = dq[3] * Q[1] - dq[0] * Q[2] + dq[1] * Q[3] + dq[2] * Q[0];
        const float z = dq[3] * Q[2] + dq[0] * Q[1] - dq[1] * Q[0] + dq[2] * Q[3];
        const float ww = dq[3] * Q[3] - dq[0] * Q[0] - dq[1] * Q[1] - dq[2] * Q[2];
        const float n = MI_RSQ(x * x + yy * yy + z * z + ww * ww);
        Q[0] = x * n; Q[1] = yy * n; Q[2] = z * n; Q[3] = ww * n;
    }
};

}  // namespace mi
