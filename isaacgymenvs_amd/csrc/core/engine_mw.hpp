// engine_mw.hpp -- the multi-wave ("limb per wave") form of the physics sub-step.
//
// Why: at the benchmark sizes (Ant@4096: 64 waves of 64 envs on a chip with 1024 SIMDs) a sub-step is bound by the instruction
// issue rate of ONE wave -- 10.8 k instructions at 4-5 cycles each -- while 15 of 16 SIMDs idle.  More envs per SIMD cannot help;
// fewer instructions per wave can.  Here one env's sub-step is spread over the NROLE = 4 waves of a workgroup (one per SIMD of a
// CU), each wave running DIFFERENT code for the same E envs (wave-level MIMD, which suits the compile-time-specialised engine:
// every role is its own straight-line instruction stream):
//
//     role r  owns the limbs dealt to it (Ant / ANYmal: one leg each; Humanoid: a leg or an arm), and
//     the trunk (the limb of the root body) is recomputed by every wave, so nothing has to be fetched to start a limb.
//
//   P1  all roles   trunk going down (pose, velocity, bias acceleration), own limbs down + up, joint-space inertia rows of the own
//                   dofs, their L^T L factor and whitened velocity.  The factor of a limb touches the trunk block only through its
//                   Schur complement: that, the limb root's composite inertia / force and the right-hand-side carry go to LDS.
//   --- barrier ---
//   P2  all roles   (redundantly) trunk coming up with every limb's contribution, trunk factor, trunk part of the whitened velocity
//   P3  all roles   constraint rows of the own limbs (joint limits, ground contacts) into the LDS row store, warm start
//   --- barrier ---
//   P4  all roles   the constraint sweeps, BLOCK by block: every wave runs Gauss-Seidel over its OWN rows (their limb part of the
//                   whitened velocity never leaves its registers); the waves couple only through the trunk part of w, which is
//                   treated Jacobi-fashion with mass splitting -- a trunk shared by n >= 2 active waves answers each of them with the
//                   weight (n + 1) / 2, and after every sweep the waves' true trunk contributions are summed in role order
//                   (one barrier per sweep, exchange area double buffered).  oracle/physics.c solve_blocks() states the order.
//   P5  all roles   generalised velocity of trunk + own limbs, impulses / sensors / joint forces of the own rows, integration
//
// Same arithmetic per row as Sim<M>::substep; the sums over roles have a fixed order (role 0, 1, 2, 3), so results do not depend on
// wave timing.  Static row store only.  (Until round 3 P4 was ONE role sweeping all rows in the single-wave kernel's Gauss-Seidel
// order -- 37 % of the sub-step with three waves idle; tools/solver_convergence.py compares the two orders.)
#pragma once
#include "engine.hpp"

namespace mi {

template <class M>
struct SimMW : Sim<M> {
    using B = Sim<M>;
    using typename B::Ctx;
    using typename B::BodyTmp;
    static constexpr int NB = B::NB, ND = B::ND, NV = B::NV, OFF = B::OFF, NSPH = B::NSPH, NSENS = B::NSENS, NLIM = B::NLIM,
                         NROWG = B::NROWG, NVA = B::NVA, NR = M::NROLE;
    // (the ownership tables below also serve fixed-base manipulators -- core/hand_engine_mw.hpp; substep_role itself is for free bases)
    // (substep_role below is the static-row-store form; core/engine_mwc.hpp derives the compact-store form from this struct)

    // ---- who owns what
    static constexpr int role_of_body(int b) { return M::role_of_limb[M::limb_of_body[b]]; }       // -1: trunk
    static constexpr int role_of_gi(int gi) { return gi < OFF ? -1 : role_of_body(M::dof_body[gi - OFF]); }
    static constexpr bool trunk_body(int b) { return role_of_body(b) < 0; }
    static constexpr bool trunk_gi(int gi) { return role_of_gi(gi) < 0; }
    template <int R> static constexpr bool owns_body(int b) { return role_of_body(b) == R || (trunk_body(b) && R == M::TRUNK_ROLE); }
    template <int R> static constexpr bool owns_gi(int gi) { return role_of_gi(gi) == R || (trunk_gi(gi) && R == M::TRUNK_ROLE); }
    template <int R> static constexpr bool sees_gi(int gi) { return role_of_gi(gi) == R || trunk_gi(gi); }   // holds valid L / w entries
    // trunk bookkeeping: bodies, generalised indices, trunk x trunk entries of L, limb roots
    static constexpr int NTB = []() constexpr { int n = 0; for (int b = 0; b < NB; ++b) n += trunk_body(b) ? 1 : 0; return n; }();
    static constexpr int tslot(int b) { int n = 0; for (int k = 0; k < b; ++k) n += trunk_body(k) ? 1 : 0; return n; }
    static constexpr int NVT = []() constexpr { int n = 0; for (int i = 0; i < NV; ++i) n += trunk_gi(i) ? 1 : 0; return n; }();
    static constexpr int tidx(int gi) { int n = 0; for (int k = 0; k < gi; ++k) n += trunk_gi(k) ? 1 : 0; return n; }
    static constexpr int entry_row(int e) {         // row i of the L entry e = (i, j)
        for (int i = 0; i < NV; ++i)
            for (int j = 0; j <= i; ++j)
                if (M::midx[i][j] == e) return i;
        return 0;
    }
    static constexpr bool trunk_entry(int e) { return trunk_gi(entry_row(e)); }      // (i, j) with i a trunk index: then j, an ancestor, is one too
    static constexpr int NTE = []() constexpr { int n = 0; for (int e = 0; e < M::NM; ++e) n += trunk_entry(e) ? 1 : 0; return n; }();
    static constexpr int teidx(int e) { int n = 0; for (int k = 0; k < e; ++k) n += trunk_entry(k) ? 1 : 0; return n; }
    static constexpr bool limb_root(int b) { return b > 0 && !trunk_body(b) && trunk_body(M::parent[b]); }
    static constexpr int NLR = []() constexpr { int n = 0; for (int b = 0; b < NB; ++b) n += limb_root(b) ? 1 : 0; return n; }();
    static constexpr int lridx(int b) { int n = 0; for (int k = 0; k < b; ++k) n += limb_root(k) ? 1 : 0; return n; }
    // every non-trunk body hangs below a trunk body through bodies of its own limb only (limbs are paths below the trunk)
    // ---- exchange area in LDS, behind the static row store (slots of E floats)
    static constexpr int X_LR = B::ROW_SLOTS_STATIC;             // [NLR][16]  composite inertia (10) + force (6) of every limb root
    static constexpr int X_DT = X_LR + 16 * NLR;                 // [NR][NTE]  Schur complement of the role's limbs on the trunk block
    static constexpr int X_DY = X_DT + NR * NTE;                 // [NR][NVT]  right-hand-side carry of the role's limbs
    static constexpr int X_DW = X_DY + NR * NVT;                 // [2][NR][NVT] the role's contribution to the trunk part of w: warm start, then one per sweep (double buffered)
    static constexpr int X_FLG = X_DW + 2 * NR * NVT;            // [2][NR]    "this role's block is active" (1.f / 0.f), double buffered like X_DW
    static constexpr int X_AT = X_FLG + 2 * NR;                  // [NROWG]    trunk part |g_T|^2 of every row's diagonal (Ainv slot: cfm + limb part, +inf = inert row)
    static constexpr int MW_SLOTS = X_AT + NROWG;

    // units of the constraint sweeps (one limit row, or the 3 rows of a sphere) in row order, and which role sweeps them
    static constexpr int NUNIT = NLIM + NSPH;
    template <int R> static constexpr bool own_unit(int u) {
        if (u < 0 || u >= NUNIT) return false;
        return u < NLIM ? owns_gi<R>(OFF + B::limdof(u)) : owns_body<R>(M::sph_body[u - NLIM]);
    }
    template <int R> static constexpr int next_own(int u) {       // first unit of role R after u (NUNIT: none)
        for (int k = u + 1; k < NUNIT; ++k) if (own_unit<R>(k)) return k;
        return NUNIT;
    }
    template <int R> static constexpr int own_pos(int u) { int n = 0; for (int k = 0; k < u; ++k) n += own_unit<R>(k) ? 1 : 0; return n; }

    // ------------------------------------------------------------------------------------------------ trunk, going down
    template <int R, int b, int XLR, int RS>
    MI_HD void trunk_down(const SimParams& P, Ctx& c, BodyTmp (&tb)[NTB], const float* Rp, const float* rp, const float* Vp,
                          const float* Ap, const RowStore<RS> rows) {
        constexpr int ts = tslot(b);
        BodyTmp& t = tb[ts];
        this->template body_down<b>(P, c, Rp, rp, Vp, Ap, t);
        sfor<NB>([&](auto C_) MI_LAMBDA {
            constexpr int ch = C_;
            if constexpr (ch > b) if constexpr (M::parent[ch] == b) {
                if constexpr (trunk_body(ch)) {
                    trunk_down<R, ch, XLR>(P, c, tb, t.Rb, t.rb, t.Vc, t.Ac, rows);
                } else if constexpr (role_of_body(ch) == R) {     // root of one of my limbs: the whole subtree, down and up
                    SpI Ic;
                    float Fc[6];
                    this->template body_pass<ch>(P, c, t.Rb, t.rb, t.Vc, t.Ac, Ic, Fc);
                    constexpr int o = XLR + 16 * lridx(ch);
                    rows(o) = Ic.m;
                    sfor<3>([&](auto K) MI_LAMBDA { rows(o + 1 + K) = Ic.h[K]; });
                    sfor<6>([&](auto K) MI_LAMBDA { rows(o + 4 + K) = Ic.I[K]; rows(o + 10 + K) = Fc[K]; });
                }
            }
        });
    }

    // ------------------------------------------------------------------------------------------------ one role of a sub-step
    // tau: efforts of all dofs.  BAR: callable workgroup barrier (device: __syncthreads; host tests: a thread barrier).
    // stage: where last sub-step's impulses of the own rows are -- 0: in lamc / laml (read here), 1: on their way into the row store (LDS-direct
    // loads issued by the caller), 2: already IN the row store, left there by the previous sub-step of the same launch (KEEP).
    // KEEP: leave the signed limit impulses in the row store for a following sub-step of the same launch (mw_kernels.hpp, fused sub-steps);
    // the contact impulses are there anyway.
    template <int R, bool KEEP = false, int RS, class GND, class BAR>
    MI_HD void substep_role(const SimParams& P, const float* tau, const float h, const RowStore<RS> rows, const Strided lamc,
                            const Strided laml, const Strided sensor, const Strided dof_force, const GND& gnd, const float mu_env,
                            const Strided netf, const int stage, const BAR& bar) {
        static_assert(!M::FIXED, "multi-wave sub-step: free-base models");
        auto G = [&](int row, int cc) MI_LAMBDA -> float& { return rows(row * M::MAXCHAIN + cc); };
        auto Ainv = [&](int row) MI_LAMBDA -> float& { return rows(NROWG * M::MAXCHAIN + row); };
        auto vt = [&](int row) MI_LAMBDA -> float& { return rows(NROWG * M::MAXCHAIN + NROWG + row); };
        auto lam = [&](int row) MI_LAMBDA -> float& { return rows(NROWG * M::MAXCHAIN + 2 * NROWG + row); };
        const float invh = MI_RCP(h);
        float (&root)[13] = this->root;
        float (&q)[M::NDA] = this->q;
        float (&qd)[M::NDA] = this->qd;
#if defined(MI_TIMING)
        unsigned long long* const tstamp = this->tstamp;     // tools/debug/mw_phases.py: [8] per sub-step
#endif
        MI_STAMP(0);
        Ctx c;
        float (&S)[M::NDA][6] = c.S;
        float (&L)[M::NM] = c.L;
        // ============================================================ P1
        if (stage == 0) {
            sfor<ND>([&](auto D) MI_LAMBDA {
                constexpr int d = D;
                if constexpr (M::dof_limited[d] && owns_gi<R>(OFF + d)) { constexpr int o = B::stage_slot_lim(d); rows(o) = laml(d); }
            });
            sfor<3 * NSPH>([&](auto K) MI_LAMBDA {
                if constexpr (owns_body<R>(M::sph_body[K / 3])) { constexpr int o = B::stage_slot_con(K); rows(o) = lamc(K); }
            });
        }
        BodyTmp tb[NTB];
        static_assert(!B::COMPACT && B::LAM_IN_ROWS, "substep_role: models on the static row store");
        trunk_down<R, 0, X_LR>(P, c, tb, nullptr, nullptr, nullptr, nullptr, rows);
        MI_PHASE();
        float Ldi[NVA], y[NVA], w[NVA], v[NVA];
        v[0] = root[7]; v[1] = root[8]; v[2] = root[9]; v[3] = root[10]; v[4] = root[11]; v[5] = root[12];
        sfor<ND>([&](auto D) MI_LAMBDA { v[OFF + D] = qd[D]; });
        // the trunk block of L and the trunk part of y start from zero in every wave: after the own limbs have been eliminated they
        // hold exactly this role's Schur complement / right-hand-side carry
        sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (trunk_entry(E_)) L[E_] = 0.f; });
        sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (trunk_gi(I)) y[I] = 0.f; });
        // `actor_params` scale factors (see Sim::actor_scale): H and the bias forces are linear in the masses -> scaled after the tree pass
        // per-dof `actor_params` factors of the dofs this role sees (the bodies' mass factors are applied where the tree pass forms their inertias)
        float sc_damp[M::NDA], sc_stiff[M::NDA], sc_arm[M::NDA];
        sfor<ND>([&](auto D) MI_LAMBDA { sc_damp[D] = 1.f; sc_stiff[D] = 1.f; sc_arm[D] = 1.f; });
        if constexpr (B::SCALED) {
            if (this->actor_scale.p != nullptr) {
                sfor<ND>([&](auto D) MI_LAMBDA {
                    if constexpr (sees_gi<R>(OFF + D)) {
                        sc_damp[D] = this->actor_scale(B::AS_DAMP + D); sc_stiff[D] = this->actor_scale(B::AS_STIFF + D); sc_arm[D] = this->actor_scale(B::AS_ARM + D);
                    }
                });
            }
        }
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (role_of_gi(gi) == R) {
                const float K = M::dof_stiffness[d] * sc_stiff[d], Dm = M::dof_damping[d] * sc_damp[d];
                L[M::midx[gi][gi]] += M::dof_armature[d] * sc_arm[d] + h * Dm + h * h * K;
                y[gi] = tau[d] - c.bias[gi] - K * (q[d] - M::dof_springref[d]) - (Dm + h * K) * qd[d];
            }
        });
        // L^T L of the own limb dofs (descending indices); updates that land on trunk entries accumulate the Schur complement
        auto factor = [&](auto K_) MI_LAMBDA {
            constexpr int k = decltype(K_)::value;
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
        };
        // z = L^-T y along the way down, w_i = (L v)_i + h z_i
        auto whiten = [&](auto I_) MI_LAMBDA {
            constexpr int i = decltype(I_)::value;
            const float z = y[i] * Ldi[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { y[M::anc[i][A_]] -= L[M::midx[i][M::anc[i][A_]]] * z; });
            float s = L[M::midx[i][i]] * v[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { s += L[M::midx[i][M::anc[i][A_]]] * v[M::anc[i][A_]]; });
            w[i] = s + h * z;
        };
        sfor_rev<NV>([&](auto K_) MI_LAMBDA { if constexpr (role_of_gi(K_) == R) factor(K_); });
        MI_PHASE();
        sfor_rev<NV>([&](auto I_) MI_LAMBDA { if constexpr (role_of_gi(I_) == R) whiten(I_); });
        // (every table look-up below is forced into a constant expression: left to the optimiser, the constexpr helpers became
        //  run-time loops over the model tables -- 580 KB of scalar code per kernel instead of 60)
        sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (trunk_entry(E_)) { constexpr int o = X_DT + R * NTE + teidx(E_); rows(o) = L[E_]; } });
        sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (trunk_gi(I)) { constexpr int o = X_DY + R * NVT + tidx(I); rows(o) = y[I]; } });
        MI_STAMP(1);
        bar();
        MI_STAMP(2);
        // ============================================================ P2 (every role, redundantly): trunk coming up, trunk factor
        sfor_rev<NB>([&](auto B_) MI_LAMBDA {
            constexpr int b = B_;
            if constexpr (trunk_body(b)) {
                constexpr int ts = tslot(b);
                BodyTmp& t = tb[ts];
                sfor<NB>([&](auto C_) MI_LAMBDA {
                    constexpr int ch = C_;
                    if constexpr (ch > b) if constexpr (M::parent[ch] == b) {
                        if constexpr (trunk_body(ch)) {
                            constexpr int tsc = tslot(ch);
                            const BodyTmp& tc = tb[tsc];
                            t.I.m += tc.I.m;
                            sfor<3>([&](auto K) MI_LAMBDA { t.I.h[K] += tc.I.h[K]; });
                            sfor<6>([&](auto K) MI_LAMBDA { t.I.I[K] += tc.I.I[K]; t.F[K] += tc.F[K]; });
                        } else {
                            constexpr int o = X_LR + 16 * lridx(ch);
                            t.I.m += rows(o);
                            sfor<3>([&](auto K) MI_LAMBDA { t.I.h[K] += rows(o + 1 + K); });
                            sfor<6>([&](auto K) MI_LAMBDA { t.I.I[K] += rows(o + 4 + K); t.F[K] += rows(o + 10 + K); });
                        }
                    }
                });
                this->template body_up<b>(c, t);
            }
        });
        sfor<OFF>([&](auto I) MI_LAMBDA { y[I] = -c.bias[I]; });
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (trunk_gi(gi)) {
                const float K = M::dof_stiffness[d] * sc_stiff[d], Dm = M::dof_damping[d] * sc_damp[d];
                L[M::midx[gi][gi]] += M::dof_armature[d] * sc_arm[d] + h * Dm + h * h * K;
                y[gi] = tau[d] - c.bias[gi] - K * (q[d] - M::dof_springref[d]) - (Dm + h * K) * qd[d];
            }
        });
        sfor<NR>([&](auto R_) MI_LAMBDA {     // fixed order of the roles: the sum does not depend on which wave got here first
            sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (trunk_entry(E_)) { constexpr int o = X_DT + R_ * NTE + teidx(E_); L[E_] += rows(o); } });
            sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (trunk_gi(I)) { constexpr int o = X_DY + R_ * NVT + tidx(I); y[I] += rows(o); } });
        });
        sfor_rev<NV>([&](auto K_) MI_LAMBDA { if constexpr (trunk_gi(K_)) factor(K_); });
        sfor_rev<NV>([&](auto I_) MI_LAMBDA { if constexpr (trunk_gi(I_)) whiten(I_); });
        MI_PHASE();
        // ============================================================ P3: constraint rows of the own limbs (static store)
        float dw[NVT];                      // this role's warm-start contribution to the trunk part of w
        sfor<NVT>([&](auto I) MI_LAMBDA { dw[I] = 0.f; });
        // is this role's block active in the first sweep: a row with a warm-start impulse, a violated joint limit, a contact
        float act = 0.f;
        auto wadd = [&](auto GI, const float val) MI_LAMBDA {
            constexpr int gi = decltype(GI)::value;
            if constexpr (trunk_gi(gi)) { constexpr int ti = tidx(gi); dw[ti] += val; } else w[gi] += val;
        };
#if defined(__HIP_DEVICE_COMPILE__)
        if (stage == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        auto chain_solve = [&](auto Bd, float* g) MI_LAMBDA {
            constexpr int b = decltype(Bd)::value;
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
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (M::dof_limited[d] && owns_gi<R>(gi)) {
                constexpr int row = B::limrow(d);
                MI_PHASE();
                const float dl = q[d] - this->template limit_lower<d>(), du = this->template limit_upper<d>() - q[d];
                const bool lower = dl < du;
                const float C = lower ? dl : du, s = lower ? 1.f : -1.f;
                const float lw = lam(row);
                const float l0 = ((lw * s < 0.f) ? 0.f : fabsf(lw)) * P.warm;
                float g[M::MAXCHAIN];
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
                float al = P.cfm, at = 0.f;      // diagonal of the row, limb and trunk part apart (the sweeps weight the trunk part)
                sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA {
                    constexpr int k = K, i = (k == 0) ? gi : M::anc[gi][k == 0 ? 0 : k - 1];
                    if constexpr (trunk_gi(i)) at += g[k] * g[k]; else al += g[k] * g[k];
                    G(row, k) = g[k];
                });
                Ainv(row) = al;
                rows(X_AT + row) = at;
                const float vtl = (C >= 0.f) ? -C * invh : fminf(-C * P.erp * invh, P.max_depen_vel);
                vt(row) = vtl;
                lam(row) = l0;
                act = ((l0 > 0.f) || (vtl > 0.f)) ? 1.f : act;
                wadd(std::integral_constant<int, gi>{}, g[0] * l0);
                sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { wadd(std::integral_constant<int, M::anc[gi][A_]>{}, g[1 + A_] * l0); });
            }
        });
        MI_PHASE();
        unsigned long long sph_active = 0ull;
        sfor<NSPH>([&](auto S_) MI_LAMBDA {
            constexpr int s = S_, b = M::sph_body[s], row0 = NLIM + 3 * s;
            if constexpr (owns_body<R>(b)) {
                MI_PHASE();
                const float* cs = c.xcs[s];
                float xc[3], dist;
                float fr[3][3];
                if constexpr (GND::HEIGHTFIELD) {
                    gnd.contact(root[0] + cs[0], root[1] + cs[1], root[2] + cs[2], M::sph_rad[s], &dist, fr[0]);
                    contact_frame(fr[0], fr[1], fr[2]);
                    sfor<3>([&](auto K) MI_LAMBDA { xc[K] = cs[K] - M::sph_rad[s] * fr[0][K]; });
                } else {
                    xc[0] = cs[0]; xc[1] = cs[1]; xc[2] = cs[2] - M::sph_rad[s];
                    dist = (root[2] + xc[2]) - P.ground_z;
                }
                const bool on = dist < P.contact_offset;
                const float onf = on ? 1.f : 0.f;
                const float gap = dist - P.rest_offset;
                if (!MI_WAVE_ANY(on)) {
                    sfor<3>([&](auto K) MI_LAMBDA { lam(row0 + K) = 0.f; });
                    return;
                }
                sph_active |= 1ull << s;
                act = on ? 1.f : act;
                sfor<3>([&](auto K) MI_LAMBDA {
                    constexpr int k = K, row = row0 + k;
                    float W[6];
                    if constexpr (GND::HEIGHTFIELD) {
                        cross3(xc, fr[k], W);
                        W[3] = fr[k][0]; W[4] = fr[k][1]; W[5] = fr[k][2];
                    } else {
                        constexpr int ax = (k == 0) ? 2 : (k == 1 ? 0 : 1);
                        sfor<6>([&](auto I_) MI_LAMBDA { W[I_] = 0.f; });
                        W[3 + ax] = 1.f;
                        if constexpr (ax == 0) { W[1] = xc[2]; W[2] = -xc[1]; }
                        else if constexpr (ax == 1) { W[0] = -xc[2]; W[2] = xc[0]; }
                        else { W[0] = xc[1]; W[1] = -xc[0]; }
                    }
                    float g[M::MAXCHAIN];
                    sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA {
                        constexpr int gi = M::chain[b][C];
                        if constexpr (gi >= OFF) g[C] = dot6(S[gi - OFF], W);
                        else if constexpr (gi < 3) g[C] = W[3 + gi];
                        else g[C] = W[gi - 3];
                    });
                    chain_solve(std::integral_constant<int, b>{}, g);
                    float al = P.cfm, at = 0.f;
                    sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA {
                        if constexpr (trunk_gi(M::chain[b][C])) at += g[C] * g[C]; else al += g[C] * g[C];
                        G(row, C) = g[C];
                    });
                    Ainv(row) = on ? al : __builtin_inff();          // inert row: 1 / inf = 0 in every sweep
                    rows(X_AT + row) = at;
                    const float vtn = (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
                    if constexpr (GND::HEIGHTFIELD) vt(row) = (k == 0) ? vtn : fr[0][k - 1];
                    else vt(row) = (k == 0) ? vtn : 0.f;
                    const float l0 = lam(row) * P.warm * onf;
                    lam(row) = l0;
                    sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { wadd(std::integral_constant<int, M::chain[b][C]>{}, g[C] * l0); });
                });
            }
        });
        sfor<NVT>([&](auto I) MI_LAMBDA { rows(X_DW + R * NVT + I) = dw[I]; });
        rows(X_FLG + R) = act;
        MI_STAMP(3);
        bar();
        MI_STAMP(4);
        // ============================================================ P4 (every role): block sweeps
        // trunk part of w: every role adds all roles' warm-start contributions, in role order
        sfor<NV>([&](auto I) MI_LAMBDA {
            constexpr int i = I;
            if constexpr (trunk_gi(i)) { sfor<NR>([&](auto R_) MI_LAMBDA { constexpr int o = X_DW + R_ * NVT + tidx(i); w[i] += rows(o); }); }
        });
        {
            struct UBuf { float g[3][M::MAXCHAIN]; float al[3], at[3], vt, lam[3]; };
            UBuf ub[2];
            float wtl[NVT];                  // trunk part of w as this block sees it during a sweep
            for (int it = 0; it < P.iters; ++it) {
                int zero;
                MI_OPAQUE_ZERO(zero);
                const RowStore<RS> rit = rows.shifted(zero);
                const int par = it & 1;
                const RowStore<RS> xout = rows.shifted((X_DW + (par ^ 1) * NR * NVT) * RowStore<RS>::stride);
                const RowStore<RS> fin = rows.shifted((X_FLG + par * NR) * RowStore<RS>::stride), fout = rows.shifted((X_FLG + (par ^ 1) * NR) * RowStore<RS>::stride);
                float nact = 0.f;
                sfor<NR>([&](auto R_) MI_LAMBDA { nact += fin(R_); });
                const float om = (nact > 1.5f) ? 0.5f * (nact + 1.f) : 1.f;      // weight of the shared trunk coordinates in this sweep
                const float iom = 1.f / om;
                sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (trunk_gi(I)) { constexpr int ti = tidx(I); wtl[ti] = w[I]; } });
                auto wget = [&](auto GI) MI_LAMBDA -> float {
                    constexpr int gi = decltype(GI)::value;
                    if constexpr (trunk_gi(gi)) { constexpr int ti = tidx(gi); return wtl[ti]; } else return w[gi];
                };
                auto wupd = [&](auto GI, const float val) MI_LAMBDA {
                    constexpr int gi = decltype(GI)::value;
                    if constexpr (trunk_gi(gi)) { constexpr int ti = tidx(gi); wtl[ti] += om * val; } else w[gi] += val;
                };
                float actn = 0.f;
                auto load_unit = [&](auto U_, UBuf& Bf) MI_LAMBDA {
                    constexpr int u = decltype(U_)::value;
                    if constexpr (u < NLIM) {
                        constexpr int d = B::limdof(u), gi = OFF + d, row = u;
                        sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { Bf.g[0][K] = rit(row * M::MAXCHAIN + K); });
                        Bf.al[0] = rit(NROWG * M::MAXCHAIN + row);
                        Bf.at[0] = rit(X_AT + row);
                        Bf.vt = rit(NROWG * M::MAXCHAIN + NROWG + row);
                        Bf.lam[0] = rit(NROWG * M::MAXCHAIN + 2 * NROWG + row);
                    } else if constexpr (u < NUNIT) {
                        constexpr int s = u - NLIM, b = M::sph_body[s], row0 = NLIM + 3 * s;
                        sfor<3>([&](auto K) MI_LAMBDA {
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { Bf.g[K][C] = rit((row0 + K) * M::MAXCHAIN + C); });
                            Bf.al[K] = rit(NROWG * M::MAXCHAIN + row0 + K);
                            Bf.at[K] = rit(X_AT + row0 + K);
                            Bf.lam[K] = rit(NROWG * M::MAXCHAIN + 2 * NROWG + row0 + K);
                        });
                        Bf.vt = rit(NROWG * M::MAXCHAIN + NROWG + row0);
                    }
                };
                auto unit_on = [&](auto U_) MI_LAMBDA -> bool {
                    constexpr int u = decltype(U_)::value;
                    if constexpr (u < NLIM) return true;
                    else if constexpr (u < NUNIT) return (sph_active >> (u - NLIM) & 1ull) != 0ull;
                    else return false;
                };
                constexpr int U0 = next_own<R>(-1);
                if constexpr (U0 < NUNIT) load_unit(std::integral_constant<int, U0>{}, ub[0]);
                sfor<NUNIT>([&](auto U_) MI_LAMBDA {
                    constexpr int u = U_;
                    if constexpr (own_unit<R>(u)) {
                        // position of u among the own units picks the register buffer
                        constexpr int pos = own_pos<R>(u), un = next_own<R>(u);
                        UBuf& Bf = ub[pos & 1];
                        if (unit_on(std::integral_constant<int, un>{}))
                            load_unit(std::integral_constant<int, un>{}, ub[(pos + 1) & 1]);  // prefetch (no-op past the end)
                        MI_PHASE();
                        if (!unit_on(std::integral_constant<int, u>{})) return;
                        if constexpr (u < NLIM) {
                            constexpr int d = B::limdof(u), gi = OFF + d, row = u;
                            float vn = Bf.g[0][0] * wget(std::integral_constant<int, gi>{});
                            sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { vn += Bf.g[0][1 + A_] * wget(std::integral_constant<int, M::anc[gi][A_]>{}); });
                            const float lo = Bf.lam[0];
                            const float nl = fmaxf(lo - (vn - Bf.vt) * MI_RCP(Bf.al[0] + om * Bf.at[0]), 0.f);
                            const float dl = nl - lo;
                            lam(row) = nl;
                            actn = (nl > 0.f) ? 1.f : actn;
                            wupd(std::integral_constant<int, gi>{}, Bf.g[0][0] * dl);
                            sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { wupd(std::integral_constant<int, M::anc[gi][A_]>{}, Bf.g[0][1 + A_] * dl); });
                        } else {
                            constexpr int s = u - NLIM, b = M::sph_body[s], row0 = NLIM + 3 * s;
                            const float mu = 0.5f * ((mu_env >= 0.f ? mu_env : M::sph_mu[s]) + P.plane_mu);
                            float ainv[3];
                            sfor<3>([&](auto K) MI_LAMBDA { ainv[K] = MI_RCP(Bf.al[K] + om * Bf.at[K]); });
                            float ln;
                            {
                                float vn = 0.f;
                                sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { vn += Bf.g[0][C] * wget(std::integral_constant<int, M::chain[b][C]>{}); });
                                const float lo = Bf.lam[0];
                                ln = fmaxf(lo - (vn - Bf.vt) * ainv[0], 0.f);
                                const float dl = ln - lo;
                                lam(row0) = ln;
                                actn = (ln > 0.f) ? 1.f : actn;
                                sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { wupd(std::integral_constant<int, M::chain[b][C]>{}, Bf.g[0][C] * dl); });
                            }
                            float lt[2], vtg[2];
                            // both tangent rows from the SAME velocity, the disc (core/engine.hpp friction_disc; oracle/physics.c solve_blocks), ONE application
                            sfor<2>([&](auto K) MI_LAMBDA {
                                float vn = 0.f;
                                sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { vn += Bf.g[1 + K][C] * wget(std::integral_constant<int, M::chain[b][C]>{}); });
                                vtg[K] = vn;
                                lt[K] = Bf.lam[1 + K] - vn * ainv[1 + K];
                            });
                            friction_disc(lt, Bf.lam[1], Bf.lam[2], vtg[0], vtg[1], ainv[1], ainv[2], mu * ln);
                            sfor<2>([&](auto K) MI_LAMBDA {
                                constexpr int row = row0 + 1 + K;
                                const float nl = lt[K], dl = nl - Bf.lam[1 + K];
                                lam(row) = nl;
                                sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { wupd(std::integral_constant<int, M::chain[b][C]>{}, Bf.g[1 + K][C] * dl); });
                            });
                        }
                    }
                });
                // this block's true contribution to the trunk part, its activity in the next sweep; then everybody's, in role order
                sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (trunk_gi(I)) { constexpr int ti = tidx(I); xout(R * NVT + ti) = (wtl[ti] - w[I]) * iom; } });
                fout(R) = actn;
                bar();
                sfor<NV>([&](auto I) MI_LAMBDA {
                    constexpr int i = I;
                    if constexpr (trunk_gi(i)) { sfor<NR>([&](auto R_) MI_LAMBDA { constexpr int o = R_ * NVT + tidx(i); w[i] += xout(o); }); }
                });
            }
        }
        MI_STAMP(5);
        // ============================================================ P5: back to generalised velocity, outputs, integration
        sfor<NV>([&](auto I_) MI_LAMBDA {       // ascending: ancestors (trunk or own limb) first
            constexpr int i = I_;
            if constexpr (sees_gi<R>(i)) {
                float s = w[i];
                sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { s -= L[M::midx[i][M::anc[i][A_]]] * v[M::anc[i][A_]]; });
                v[i] = s * Ldi[i];
            }
        });
        MI_PHASE();
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            if constexpr (owns_gi<R>(OFF + d)) {
                float ll = 0.f;
                if constexpr (M::dof_limited[d]) {
                    constexpr int row = B::limrow(d);
                    const float dl = q[d] - this->template limit_lower<d>(), du = this->template limit_upper<d>() - q[d];
                    ll = (dl < du) ? lam(row) : -lam(row);
                    if constexpr (KEEP) lam(row) = ll;
                }
                laml(d) = ll;
                dof_force(d) = tau[d] - M::dof_stiffness[d] * sc_stiff[d] * (q[d] - M::dof_springref[d]) - M::dof_damping[d] * sc_damp[d] * v[OFF + d] + ll * invh;
            }
        });
        float sens[6 * M::NSENSA];
        sfor<6 * NSENS>([&](auto K) MI_LAMBDA { sens[K] = 0.f; });
        float nf[GND::NETF ? NB : 1][3];
        if constexpr (GND::NETF) sfor<NB>([&](auto B_) MI_LAMBDA { nf[B_][0] = nf[B_][1] = nf[B_][2] = 0.f; });
        sfor<NSPH>([&](auto S_) MI_LAMBDA {
            constexpr int s = S_, b = M::sph_body[s], row0 = NLIM + 3 * s;
            if constexpr (owns_body<R>(b)) {
                if (!(sph_active >> s & 1ull)) {
                    lamc(3 * s) = 0.f; lamc(3 * s + 1) = 0.f; lamc(3 * s + 2) = 0.f;
                    return;
                }
                const float ln = lam(row0), l1 = lam(row0 + 1), l2 = lam(row0 + 2);
                lamc(3 * s) = ln; lamc(3 * s + 1) = l1; lamc(3 * s + 2) = l2;
                float f[3], xc[3];
                if constexpr (GND::HEIGHTFIELD) {
                    float n[3], t1[3], t2[3];
                    n[0] = vt(row0 + 1); n[1] = vt(row0 + 2);
                    n[2] = MI_SQRT(fmaxf(1.f - n[0] * n[0] - n[1] * n[1], 0.f));
                    contact_frame(n, t1, t2);
                    sfor<3>([&](auto K) MI_LAMBDA {
                        f[K] = (n[K] * ln + t1[K] * l1 + t2[K] * l2) * invh;
                        xc[K] = c.xcs[s][K] - M::sph_rad[s] * n[K];
                        nf[b][K] += f[K];
                    });
                } else {
                    f[0] = l1 * invh; f[1] = l2 * invh; f[2] = ln * invh;
                    xc[0] = c.xcs[s][0]; xc[1] = c.xcs[s][1]; xc[2] = c.xcs[s][2] - M::sph_rad[s];
                    if constexpr (GND::NETF) sfor<3>([&](auto K) MI_LAMBDA { nf[b][K] += f[K]; });
                }
                if constexpr (B::sensor_of(b) >= 0) {
                    constexpr int k = B::sensor_of(b);
                    const float arm[3] = {xc[0] - c.rs[k][0], xc[1] - c.rs[k][1], xc[2] - c.rs[k][2]};
                    float tq[3], fl[3], tl[3];
                    cross3(arm, f, tq);
                    matTvec3(c.Rs[k], f, fl); matTvec3(c.Rs[k], tq, tl);
                    sfor<3>([&](auto C) MI_LAMBDA { sens[6 * k + C] += fl[C]; sens[6 * k + 3 + C] += tl[C]; });
                }
            }
        });
        if constexpr (GND::NETF) sfor<NB>([&](auto B_) MI_LAMBDA {
            if constexpr (owns_body<R>(B_)) sfor<3>([&](auto K) MI_LAMBDA { netf(3 * B_ + K) = nf[B_][K]; });
        });
        sfor<NSENS>([&](auto K_) MI_LAMBDA {
            if constexpr (owns_body<R>(M::sens_body[K_])) sfor<6>([&](auto C) MI_LAMBDA { sensor(6 * K_ + C) = sens[6 * K_ + C]; });
        });
        MI_PHASE();
        sfor<ND>([&](auto D) MI_LAMBDA {
            if constexpr (owns_gi<R>(OFF + D)) { qd[D] = v[OFF + D]; q[D] += h * qd[D]; }
        });
        if constexpr (R == M::TRUNK_ROLE) {
#if !defined(MI_NO_VEL_CLAMP)   // (measurement builds only)
            {   // AssetOptions.max_angular_velocity / max_linear_velocity (core/engine.hpp kMax*Velocity)
                const float w2 = v[3] * v[3] + v[4] * v[4] + v[5] * v[5], l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
                const float sw = (w2 > kMaxAngularVelocity * kMaxAngularVelocity) ? kMaxAngularVelocity * MI_RSQ(w2) : 1.f;
                const float sl = (l2 > kMaxLinearVelocity * kMaxLinearVelocity) ? kMaxLinearVelocity * MI_RSQ(l2) : 1.f;
                v[0] *= sl; v[1] *= sl; v[2] *= sl; v[3] *= sw; v[4] *= sw; v[5] *= sw;
            }
#endif
            sfor<3>([&](auto K) MI_LAMBDA { root[7 + K] = v[K]; root[10 + K] = v[3 + K]; root[K] += h * v[K]; });
            const float om[3] = {v[3], v[4], v[5]};
            const float an = MI_SQRT(dot3(om, om)), th = an * h;
            float dq[4];
            {
                float sn, cs;
                sincosf(0.5f * th, &sn, &cs);
                const bool big = th > 1e-12f;
                const float k = big ? sn * MI_RCP(fmaxf(an, 1e-30f)) : 0.5f * h;
                dq[0] = om[0] * k; dq[1] = om[1] * k; dq[2] = om[2] * k; dq[3] = big ? cs : 1.f;
            }
            float* Q = root + 3;
            const float x = dq[3] * Q[0] + dq[0] * Q[3] + dq[1] * Q[2] - dq[2] * Q[1];
            const float yy = dq[3] * Q[1] - dq[0] * Q[2] + dq[1] * Q[3] + dq[2] * Q[0];
            const float z = dq[3] * Q[2] + dq[0] * Q[1] - dq[1] * Q[0] + dq[2] * Q[3];
            const float ww = dq[3] * Q[3] - dq[0] * Q[0] - dq[1] * Q[1] - dq[2] * Q[2];
            const float n = MI_RSQ(x * x + yy * yy + z * z + ww * ww);
            Q[0] = x * n; Q[1] = yy * n; Q[2] = z * n; Q[3] = ww * n;
        }
        MI_STAMP(6);
    }
};

}  // namespace mi
