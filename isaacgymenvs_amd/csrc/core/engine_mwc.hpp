// engine_mwc.hpp -- the limb-per-wave sub-step for robots on the COMPACT contact store (Humanoid: 126 potential rows, self-collision).
//
// Round 2 ran this robot on one main wave (plus a helper for the self-collision phase): 49 % of the sub-step was a serial tail on that
// wave (warm start, Gauss-Seidel sweeps, write-back), it executed the UNION of its 32 envs' active contact sets, and its live state (the
// whole factor L, all joint axes, 105 sphere coordinates) overflowed the register file into 1 KB of scratch per lane.  Here the four
// waves of the workgroup each own ONE limb (engine_mw.hpp roles; the trunk is recomputed by all) -- tree pass, factor, constraint
// rows, SWEEPS and outputs of that limb -- so a wave's live state is one limb's L / S / rows and nothing spills by construction:
//
//   P1  all     trunk down, own limb down + up, limb factor + whitened velocity; Schur complement / carries -> LDS          | B1
//   P2  all     trunk up with every limb's contribution, trunk factor (redundantly)
//   P3  all     own joint-limit rows; own ground contacts into the role's OWN contact slots (per-role caps M::wave_kcap, rows in a
//               FIXED shape [limb dofs | trunk dofs]: every body of a limb shares it, so the sweeps walk a lane's actual contacts in
//               a run-time loop instead of the union of the wave's spheres).  The pair role (the lightest one) also runs a
//               positions-only forward kinematics of the whole body + the self-collision narrow phase and publishes <= KPAIR contacts | B2
//   S1  all     self-contact half rows, side a: the wave that owns body a builds g_a = L^-T J_a^T (limb part, trunk part)            | B3
//   S2  all     side b: -g_b into the other limb part (or folded into side a's when both bodies sit on one limb), trunk part -=     | B4
//   W   pair    warm start of the self-contact rows                                                                                 | B5
//   P4  all     block sweeps (oracle/physics.c solve_blocks): own rows Gauss-Seidel; the self contacts are a fifth block swept by the
//               pair role after its own; coordinates shared by n >= 2 active blocks (the trunk: all; a limb: its owner + the pair
//               block) answer with weight (n + 1) / 2; true contributions exchanged through LDS after every sweep                  | 1 per sweep
//   P5  all     velocities, impulses / sensors / joint forces of the own rows, integration of the own dofs
//
// LDS per env: <= 1280 floats (32 envs per workgroup, one workgroup per CU).  The exchange areas of P1 / P2 and everything that is only
// needed after B2 (self-contact rows, sweep exchange) share one region.
#pragma once
#include "engine_mw.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
#define MI_ATOMIC_ADD_INT(p, v) atomicAdd((p), (v))
#else
#define MI_ATOMIC_ADD_INT(p, v) __atomic_fetch_add((p), (v), __ATOMIC_RELAXED)
#endif

namespace mi {

template <class M>
struct SimMWC : SimMW<M> {
    using MW = SimMW<M>;
    using B = Sim<M>;
    using typename B::Ctx;
    using typename B::BodyTmp;
    static constexpr int NB = M::NB, ND = M::ND, NV = M::NV, OFF = M::OFF, NSPH = M::NSPH, NSENS = M::NSENS, NLIM = B::NLIM, NVA = B::NVA,
                         NR = M::NROLE, NVT = MW::NVT, NTE = MW::NTE, NLR = MW::NLR, NTB = MW::NTB, NPG = M::NPG, KPAIR = 3, NBLK = NR + 1;
    static_assert(M::NLIMB == NR + 1 && !M::FIXED && B::COMPACT, "one limb per role, free base, compact-store model");
    // ---- the limb of a role: generalised indices lfirst(r) .. lfirst(r) + nl(r) - 1
    static constexpr int lfirst(int r) { for (int i = OFF; i < NV; ++i) if (MW::role_of_gi(i) == r) return i; return NV; }
    static constexpr int nl(int r) { int n = 0; for (int i = 0; i < NV; ++i) n += (MW::role_of_gi(i) == r) ? 1 : 0; return n; }
    static constexpr bool limbs_contiguous() {
        for (int r = 0; r < NR; ++r) for (int i = lfirst(r); i < lfirst(r) + nl(r); ++i) if (MW::role_of_gi(i) != r) return false;
        return true;
    }
    static_assert(limbs_contiguous(), "a limb's dofs are numbered consecutively (depth-first numbering)");
    static constexpr int NLMAX = []() constexpr { int m = 0; for (int r = 0; r < NR; ++r) m = nl(r) > m ? nl(r) : m; return m; }();
    static constexpr int NVL = NV - NVT;                                                   // limb coordinates of all roles
    static constexpr int loff(int r) { int o = 0; for (int k = 0; k < r; ++k) o += nl(k); return o; }   // role r's place in an [NVL] vector
    static constexpr int rlen(int r) { return nl(r) + NVT; }                               // fixed row shape of role r: [limb | trunk]
    static constexpr int gcsz(int r) { return 3 * rlen(r) + 7; }                           // ground slot: 3 rows, |g_limb|^2 x3, vt, lam x3
    static constexpr int kcap(int r) { return M::wave_kcap[r]; }
    static constexpr int PAIR_ROLE = []() constexpr {                                      // the lightest role hosts the self-contact block
        int best = 0, bl = 1 << 30;
        for (int r = 0; r < NR; ++r) { const int l = nl(r) + (r == M::TRUNK_ROLE ? NVT : 0); if (l <= bl) { bl = l; best = r; } }
        return best;
    }();
    // ---- LDS layout (floats per env)
    static constexpr int C_LIMG = B::C_LIMG;                                               // limit rows' G, packed by B::limoff
    static constexpr int L_SL = C_LIMG, L_VT = C_LIMG + NLIM, L_LAM = C_LIMG + 2 * NLIM;   // |g_limb|^2, velocity target, impulse
    static constexpr int gcb(int r) { int o = C_LIMG + 3 * NLIM; for (int k = 0; k < r; ++k) o += kcap(k) * gcsz(k); return o; }
    static constexpr int C_SLOTOF = gcb(NR);                                               // one byte per sphere: its slot in the owner's range, -1
    static constexpr int C_X = C_SLOTOF + (NSPH + 3) / 4;                                  // group -> slot map, then per self contact 12 floats:
    static constexpr int XI = 12;                                                          //   point (3), normal (3), bodies / sides word, mu, vt, lam0 (3)
    static constexpr int A0 = C_X + 1 + XI * KPAIR;                                        // shared region: before B2 ...
    static constexpr int X_LR = A0, X_DT = X_LR + 16 * NLR, X_DY = X_DT + NR * NTE, X_END = X_DY + NR * NVT;
    static constexpr int P_CSZ = 3 * (2 * NLMAX + NVT) + 4;                                // ... after B2: self-contact rows LA, LB, T; vt, lam x3
    static constexpr int P_B = A0, PW = P_B + KPAIR * P_CSZ;                               // [NVL]        limb part of w as the pair block starts a sweep
    static constexpr int DOWN = PW + NVL;                                                  // [2][NVL]     true contribution of every owner's block to its limb part
    static constexpr int DPAIR = DOWN + 2 * NVL;                                           // [2][NVL]     ... of the pair block
    static constexpr int X_DW = DPAIR + 2 * NVL;                                           // [2][NBLK][NVT] ... of every block to the trunk part
    static constexpr int X_FLG = X_DW + 2 * NBLK * NVT;                                    // [2][NBLK]    block active
    static constexpr int X_TOUCH = X_FLG + 2 * NBLK;                                       // [1]          roles whose limbs the self contacts touch (bit mask)
    static constexpr int S_END = X_TOUCH + 1;
    static constexpr int MWC_SLOTS = X_END > S_END ? X_END : S_END;
    static constexpr int LANES = 32;
    static_assert((size_t)MWC_SLOTS * LANES * sizeof(float) <= 160 * 1024, "row store + exchange areas fit the LDS of a CU at 32 envs per workgroup");
    static constexpr int PLA = 0, PLB = 3 * NLMAX, PT = 6 * NLMAX, PVT = 6 * NLMAX + 3 * NVT, PLAM = PVT + 1;   // inside a self-contact slot

    // position of generalised index gi in role R's fixed row shape
    template <int R> static constexpr int shape_idx(int gi) { return MW::trunk_gi(gi) ? nl(R) + MW::tidx(gi) : gi - lfirst(R); }
    template <int R> static constexpr bool in_chain(int b, int idx) {
        for (int c = 0; c < M::chain_len[b]; ++c) if (shape_idx<R>(M::chain[b][c]) == idx) return true;
        return false;
    }
    static constexpr bool chain_has_t(int b, int t) {
        for (int c = 0; c < M::chain_len[b]; ++c) if (MW::trunk_gi(M::chain[b][c]) && MW::tidx(M::chain[b][c]) == t) return true;
        return false;
    }
    static constexpr bool chain_has_l(int b, int k) {      // limb-local index k of the body's own limb
        const int r = MW::role_of_body(b);
        if (r < 0) return false;
        for (int c = 0; c < M::chain_len[b]; ++c) if (!MW::trunk_gi(M::chain[b][c]) && M::chain[b][c] - lfirst(r) == k) return true;
        return false;
    }

    // ---------------------------------------------------------------- positions-only forward kinematics (pair role: all sphere centres)
    template <int b>
    MI_HD void fk_pos(const float* Rp, const float* rp, float (*xa)[3]) {
        float Rb[9], rb[3];
        if constexpr (b == 0) {
            quat2mat(this->root + 3, Rb);
            rb[0] = rb[1] = rb[2] = 0.f;
        } else {
            if constexpr (B::brot_is_identity(b)) sfor<9>([&](auto K) MI_LAMBDA { Rb[K] = Rp[K]; });
            else matmul3(Rp, M::brot[b], Rb);
            float t[3];
            matvec3(Rp, M::bpos[b], t);
            rb[0] = rp[0] + t[0]; rb[1] = rp[1] + t[1]; rb[2] = rp[2] + t[2];
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
                float s, cs;
                MI_SINCOS(this->q[d], &s, &cs);
                const float t = 1.f - cs;
                const float Q[9] = {cs + ax * ax * t, ax * ay * t - az * s, ax * az * t + ay * s,
                                    ay * ax * t + az * s, cs + ay * ay * t, ay * az * t - ax * s,
                                    az * ax * t - ay * s, az * ay * t + ax * s, cs + az * az * t};
                matmul3(Rb, Q, Rb);
                float tb[3];
                matvec3(Rb, anl, tb);
                rb[0] = pt[0] - tb[0]; rb[1] = pt[1] - tb[1]; rb[2] = pt[2] - tb[2];
            } else {
                rb[0] += a[0] * this->q[d]; rb[1] += a[1] * this->q[d]; rb[2] += a[2] * this->q[d];
            }
        });
        sfor<NSPH>([&](auto S_) MI_LAMBDA {
            constexpr int s = S_;
            if constexpr (M::sph_body[s] == b) {
                float t[3];
                matvec3(Rb, M::sph_pos[s], t);
                xa[s][0] = rb[0] + t[0]; xa[s][1] = rb[1] + t[1]; xa[s][2] = rb[2] + t[2];
            }
        });
        sfor<NB>([&](auto C_) MI_LAMBDA {
            constexpr int ch = C_;
            if constexpr (ch > b) if constexpr (M::parent[ch] == b) fk_pos<ch>(Rb, rb, xa);
        });
    }

    // ---------------------------------------------------------------- one role of a sub-step
    template <int R, int RS, class BAR>
    MI_HD void substep_role_c(const SimParams& P, const float* tau, const float h, const RowStore<RS> rows, const Strided lamc,
                              const Strided laml, const Strided sensor, const Strided dof_force, const float mu_env, const SelfCol* scol,
                              const BAR& bar) {
        constexpr int ST = RowStore<RS>::stride;
        constexpr int NLR_ = nl(R), RLEN = rlen(R), GCB = gcb(R), GCSZ = gcsz(R), KCAP = kcap(R), LF = lfirst(R);
        constexpr bool PAIRW = (R == PAIR_ROLE) && (NPG > 0);
        auto slot8 = [&](int s) MI_LAMBDA -> signed char& { return reinterpret_cast<signed char*>(rows.ptr(C_SLOTOF + (s >> 2)))[s & 3]; };
        const float invh = MI_RCP(h);
        float (&root)[13] = this->root;
        float (&q)[M::NDA] = this->q;
        float (&qd)[M::NDA] = this->qd;
        const bool selfcol = (NPG > 0) && (scol != nullptr);
        Ctx c;
        float (&S)[M::NDA][6] = c.S;
        float (&L)[M::NM] = c.L;
        // ============================================================ P1 (as engine_mw.hpp)
        BodyTmp tb[NTB];
        this->template trunk_down<R, 0, X_LR>(P, c, tb, nullptr, nullptr, nullptr, nullptr, rows);
        MI_PHASE();
        float Ldi[NVA], y[NVA], w[NVA], v[NVA];
        v[0] = root[7]; v[1] = root[8]; v[2] = root[9]; v[3] = root[10]; v[4] = root[11]; v[5] = root[12];
        sfor<ND>([&](auto D) MI_LAMBDA { v[OFF + D] = qd[D]; });
        sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::trunk_entry(E_)) L[E_] = 0.f; });
        sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) y[I] = 0.f; });
        float sc_mass = 1.f, sc_damp = 1.f, sc_stiff = 1.f, sc_arm = 1.f;
        if constexpr (B::SCALED) {
            if (this->actor_scale.p != nullptr) {
                sc_mass = this->actor_scale(0); sc_damp = this->actor_scale(1); sc_stiff = this->actor_scale(2); sc_arm = this->actor_scale(3);
                sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::role_of_gi(MW::entry_row(E_)) == R) L[E_] *= sc_mass; });
                sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::role_of_gi(I) == R) c.bias[I] *= sc_mass; });
            }
        }
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (MW::role_of_gi(gi) == R) {
                const float K = M::dof_stiffness[d] * sc_stiff, Dm = M::dof_damping[d] * sc_damp;
                L[M::midx[gi][gi]] += M::dof_armature[d] * sc_arm + h * Dm + h * h * K;
                y[gi] = tau[d] - c.bias[gi] - K * (q[d] - M::dof_springref[d]) - (Dm + h * K) * qd[d];
            }
        });
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
        auto whiten = [&](auto I_) MI_LAMBDA {
            constexpr int i = decltype(I_)::value;
            const float z = y[i] * Ldi[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { y[M::anc[i][A_]] -= L[M::midx[i][M::anc[i][A_]]] * z; });
            float s = L[M::midx[i][i]] * v[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { s += L[M::midx[i][M::anc[i][A_]]] * v[M::anc[i][A_]]; });
            w[i] = s + h * z;
        };
        sfor_rev<NV>([&](auto K_) MI_LAMBDA { if constexpr (MW::role_of_gi(K_) == R) factor(K_); });
        MI_PHASE();
        sfor_rev<NV>([&](auto I_) MI_LAMBDA { if constexpr (MW::role_of_gi(I_) == R) whiten(I_); });
        sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::trunk_entry(E_)) { constexpr int o = X_DT + R * NTE + MW::teidx(E_); rows(o) = L[E_]; } });
        sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int o = X_DY + R * NVT + MW::tidx(I); rows(o) = y[I]; } });
        bar();                                                                                       // ---- B1
        // ============================================================ P2 (every role, redundantly): trunk coming up, trunk factor
        sfor_rev<NB>([&](auto B_) MI_LAMBDA {
            constexpr int b = B_;
            if constexpr (MW::trunk_body(b)) {
                constexpr int ts = MW::tslot(b);
                BodyTmp& t = tb[ts];
                sfor<NB>([&](auto C_) MI_LAMBDA {
                    constexpr int ch = C_;
                    if constexpr (ch > b) if constexpr (M::parent[ch] == b) {
                        if constexpr (MW::trunk_body(ch)) {
                            constexpr int tsc = MW::tslot(ch);
                            const BodyTmp& tc = tb[tsc];
                            t.I.m += tc.I.m;
                            sfor<3>([&](auto K) MI_LAMBDA { t.I.h[K] += tc.I.h[K]; });
                            sfor<6>([&](auto K) MI_LAMBDA { t.I.I[K] += tc.I.I[K]; t.F[K] += tc.F[K]; });
                        } else {
                            constexpr int o = X_LR + 16 * MW::lridx(ch);
                            t.I.m += rows(o);
                            sfor<3>([&](auto K) MI_LAMBDA { t.I.h[K] += rows(o + 1 + K); });
                            sfor<6>([&](auto K) MI_LAMBDA { t.I.I[K] += rows(o + 4 + K); t.F[K] += rows(o + 10 + K); });
                        }
                    }
                });
                this->template body_up<b>(c, t);
            }
        });
        if constexpr (B::SCALED) {
            if (this->actor_scale.p != nullptr) {
                sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::trunk_entry(E_)) L[E_] *= sc_mass; });
                sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) c.bias[I] *= sc_mass; });
            }
        }
        sfor<OFF>([&](auto I) MI_LAMBDA { y[I] = -c.bias[I]; });
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (MW::trunk_gi(gi)) {
                const float K = M::dof_stiffness[d] * sc_stiff, Dm = M::dof_damping[d] * sc_damp;
                L[M::midx[gi][gi]] += M::dof_armature[d] * sc_arm + h * Dm + h * h * K;
                y[gi] = tau[d] - c.bias[gi] - K * (q[d] - M::dof_springref[d]) - (Dm + h * K) * qd[d];
            }
        });
        sfor<NR>([&](auto R_) MI_LAMBDA {
            sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::trunk_entry(E_)) { constexpr int o = X_DT + R_ * NTE + MW::teidx(E_); L[E_] += rows(o); } });
            sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int o = X_DY + R_ * NVT + MW::tidx(I); y[I] += rows(o); } });
        });
        sfor_rev<NV>([&](auto K_) MI_LAMBDA { if constexpr (MW::trunk_gi(K_)) factor(K_); });
        sfor_rev<NV>([&](auto I_) MI_LAMBDA { if constexpr (MW::trunk_gi(I_)) whiten(I_); });
        MI_PHASE();
        // ============================================================ P3: own constraint rows
        float dw[NVT];
        sfor<NVT>([&](auto I) MI_LAMBDA { dw[I] = 0.f; });
        float act = 0.f;
        auto wadd = [&](auto GI, const float val) MI_LAMBDA {
            constexpr int gi = decltype(GI)::value;
            if constexpr (MW::trunk_gi(gi)) { constexpr int ti = MW::tidx(gi); dw[ti] += val; } else w[gi] += val;
        };
        // three rows over the chain of body b for the unit spatial forces W[k] (J from the joint axes, then L^-T along the chain)
        auto rows3 = [&](auto Bd, const float (&W)[3][6], float (&g)[3][M::MAXCHAIN]) MI_LAMBDA {
            constexpr int b = decltype(Bd)::value;
            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA {
                constexpr int gi = M::chain[b][C];
                if constexpr (gi >= OFF) sfor<3>([&](auto K) MI_LAMBDA { g[K][C] = dot6(S[gi - OFF], W[K]); });
                else if constexpr (gi < 3) sfor<3>([&](auto K) MI_LAMBDA { g[K][C] = W[K][3 + gi]; });
                else sfor<3>([&](auto K) MI_LAMBDA { g[K][C] = W[K][gi - 3]; });
            });
            sfor<M::chain_len[b]>([&](auto K) MI_LAMBDA {
                constexpr int k = K, i = M::chain[b][k];
                const float di = Ldi[i];
                const float z0 = g[0][k] * di, z1 = g[1][k] * di, z2 = g[2][k] * di;
                g[0][k] = z0; g[1][k] = z1; g[2][k] = z2;
                sfor<M::chain_len[b] - 1 - k>([&](auto T) MI_LAMBDA {
                    constexpr int kk = k + 1 + T, j = M::chain[b][kk];
                    const float l = L[M::midx[i][j]];
                    g[0][kk] -= l * z0; g[1][kk] -= l * z1; g[2][kk] -= l * z2;
                });
            });
        };
        // last sub-step's impulses of the own ground spheres: issued together, consumed sphere by sphere below
        float lprev[NSPH > 0 ? NSPH : 1][3];
        sfor<NSPH>([&](auto S_) MI_LAMBDA {
            if constexpr (MW::template owns_body<R>(M::sph_body[S_])) sfor<3>([&](auto K) MI_LAMBDA { lprev[S_][K] = lamc(3 * S_ + K); });
        });
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (M::dof_limited[d] && MW::template owns_gi<R>(gi)) {
                constexpr int row = B::limrow(d), g0 = B::limoff(row);
                MI_PHASE();
                const float dl = q[d] - this->template limit_lower<d>(), du = this->template limit_upper<d>() - q[d];
                const bool lower = dl < du;
                const float C = lower ? dl : du, s = lower ? 1.f : -1.f;
                const float lw = laml(d);
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
                float sl = 0.f;
                sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA {
                    constexpr int k = K, i = (k == 0) ? gi : M::anc[gi][k == 0 ? 0 : k - 1];
                    if constexpr (!MW::trunk_gi(i)) sl += g[k] * g[k];
                    rows(g0 + k) = g[k];
                });
                rows(L_SL + row) = sl;
                const float vtl = (C >= 0.f) ? -C * invh : fminf(-C * P.erp * invh, P.max_depen_vel);
                rows(L_VT + row) = vtl;
                rows(L_LAM + row) = l0;
                act = ((l0 > 0.f) || (vtl > 0.f)) ? 1.f : act;
                wadd(std::integral_constant<int, gi>{}, g[0] * l0);
                sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { wadd(std::integral_constant<int, M::anc[gi][A_]>{}, g[1 + A_] * l0); });
            }
        });
        MI_PHASE();
        int cnt = 0, ndrop = 0;
        sfor<NSPH>([&](auto S_) MI_LAMBDA {
            constexpr int s = S_, b = M::sph_body[s];
            if constexpr (MW::template owns_body<R>(b)) {
                MI_PHASE();
                const float* cs = c.xcs[s];
                const float xc[3] = {cs[0], cs[1], cs[2] - M::sph_rad[s]};
                const float dist = (root[2] + xc[2]) - P.ground_z;
                const bool near = dist < P.contact_offset;
                const bool on = near && (cnt < KCAP);
                ndrop += (near && !on) ? 1 : 0;
                const int j = on ? cnt : -1;
                if (MI_WAVE_ANY(on)) {
                    if (on) {
                        float* cb = rows.ptr(GCB + j * GCSZ);
                        float W[3][6];
                        sfor<3>([&](auto K) MI_LAMBDA {
                            constexpr int k = K, ax = (k == 0) ? 2 : (k == 1 ? 0 : 1);       // normal z, tangents x, y
                            sfor<6>([&](auto I_) MI_LAMBDA { W[k][I_] = 0.f; });
                            W[k][3 + ax] = 1.f;
                            if constexpr (ax == 0) { W[k][1] = xc[2]; W[k][2] = -xc[1]; }
                            else if constexpr (ax == 1) { W[k][0] = -xc[2]; W[k][2] = xc[0]; }
                            else { W[k][0] = xc[1]; W[k][1] = -xc[0]; }
                        });
                        float g[3][M::MAXCHAIN];
                        rows3(std::integral_constant<int, b>{}, W, g);
                        const float gap = dist - P.rest_offset;
                        sfor<3>([&](auto K) MI_LAMBDA {
                            constexpr int k = K;
                            float sl = 0.f;
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA {
                                constexpr int gi = M::chain[b][C], idx = shape_idx<R>(gi);
                                if constexpr (!MW::trunk_gi(gi)) sl += g[k][C] * g[k][C];
                                cb[(k * RLEN + idx) * ST] = g[k][C];
                            });
                            sfor<RLEN>([&](auto I_) MI_LAMBDA { if constexpr (!in_chain<R>(b, I_)) cb[(k * RLEN + I_) * ST] = 0.f; });   // the rest of the fixed shape
                            cb[(3 * RLEN + k) * ST] = sl;
                            const float l0 = lprev[s][k] * P.warm;
                            cb[(3 * RLEN + 4 + k) * ST] = l0;
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { wadd(std::integral_constant<int, M::chain[b][C]>{}, g[k][C] * l0); });
                        });
                        cb[(3 * RLEN + 3) * ST] = (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
                        act = 1.f;
                    }
                }
                cnt += on ? 1 : 0;
                slot8(s) = (signed char)j;
            }
        });
        if (scol != nullptr && scol->dropped != nullptr && ndrop > 0) MI_ATOMIC_ADD_INT(scol->dropped, ndrop);
        // ---- pair role: positions of all spheres, narrow phase of the self-collision groups, <= KPAIR contacts -> C_X
        unsigned pmap = 0xFFFFFFFFu;
        if constexpr (PAIRW) { if (selfcol) {
            MI_PHASE();
            float xa[NSPH][3];
            fk_pos<0>(nullptr, nullptr, xa);
            int cntp = 0, pdrop = 0;
            sfor<KPAIR>([&](auto J_) MI_LAMBDA { rows(C_X + 1 + XI * J_ + 6) = __builtin_bit_cast(float, 0xFFFFFFFFu); });
            sfor<NPG>([&](auto G_) MI_LAMBDA {
                constexpr int g = G_;
                MI_PHASE();
                float best = 3.0e38f, bca[3] = {0.f, 0.f, 0.f}, bcb[3] = {0.f, 0.f, 0.f};
                int bk = 0;
                sfor<M::pg_count[g]>([&](auto K_) MI_LAMBDA {
                    constexpr int k = M::pg_first[g] + K_, ia = M::gp_a[k], ib = M::gp_b[k];
                    constexpr float reach = B::cap_bound(ia) + B::cap_bound(ib);
                    float dm[3];
                    sfor<3>([&](auto I_) MI_LAMBDA {
                        dm[I_] = 0.5f * (xa[M::cap_s0[ia]][I_] + xa[M::cap_s1[ia]][I_]) - 0.5f * (xa[M::cap_s0[ib]][I_] + xa[M::cap_s1[ib]][I_]);
                    });
                    const float rr = reach + P.contact_offset;
                    if (!MI_WAVE_ANY(dot3(dm, dm) < rr * rr)) return;
                    float ca[3], cb[3];
                    seg_seg_closest<B::cap_is_point(ia), B::cap_is_point(ib)>(xa[M::cap_s0[ia]], xa[M::cap_s1[ia]], xa[M::cap_s0[ib]], xa[M::cap_s1[ib]], ca, cb);
                    const float dv[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
                    const float dist = MI_SQRT(dot3(dv, dv)) - (M::cap_rad[ia] + M::cap_rad[ib]);
                    const bool better = dist < best;
                    best = better ? dist : best;
                    sfor<3>([&](auto I_) MI_LAMBDA { bca[I_] = better ? ca[I_] : bca[I_]; bcb[I_] = better ? cb[I_] : bcb[I_]; });
                    bk = better ? K_ : bk;
                });
                const bool near = best < P.contact_offset;
                const bool on = near && (cntp < KPAIR);
                pdrop += (near && !on) ? 1 : 0;
                if (MI_WAVE_ANY(on)) {
                    if (on) {
                        int bab = 0;
                        float brb = 0.f, bmu = 0.f;
                        sfor<M::pg_count[g]>([&](auto K_) MI_LAMBDA {
                            constexpr int k = M::pg_first[g] + K_, ia = M::gp_a[k], ib = M::gp_b[k], ba = M::cap_body[ia], bb = M::cap_body[ib];
                            constexpr int ra = MW::role_of_body(ba), rb = MW::role_of_body(bb);
                            constexpr bool fold = (ra >= 0) && (ra == rb);                    // both bodies on one limb: side b folds into side a's limb part
                            constexpr int word = ba | (bb << 8) | ((ra + 1) << 16) | ((fold ? 0 : rb + 1) << 20) | ((fold ? 1 : 0) << 24);
                            const bool me = bk == K_;
                            bab = me ? word : bab;
                            brb = me ? M::cap_rad[ib] : brb;
                            bmu = me ? 0.5f * (M::cap_mu[ia] + M::cap_mu[ib]) : bmu;
                        });
                        float n[3];
                        {
                            const float dv[3] = {bca[0] - bcb[0], bca[1] - bcb[1], bca[2] - bcb[2]};
                            const float d2 = dot3(dv, dv);
                            const bool okd = d2 > 1e-18f;
                            const float inv = MI_RSQ(fmaxf(d2, 1e-30f));
                            n[0] = okd ? dv[0] * inv : 0.f; n[1] = okd ? dv[1] * inv : 0.f; n[2] = okd ? dv[2] * inv : 1.f;
                        }
                        float* xi = rows.ptr(C_X + 1 + XI * cntp);
                        sfor<3>([&](auto I_) MI_LAMBDA { xi[I_ * ST] = bcb[I_] + n[I_] * (brb + 0.5f * best); xi[(3 + I_) * ST] = n[I_]; });
                        xi[6 * ST] = __builtin_bit_cast(float, bab);
                        xi[7 * ST] = bmu;
                        const float gap = best - P.rest_offset;
                        xi[8 * ST] = (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
                        sfor<3>([&](auto K) MI_LAMBDA { xi[(9 + K) * ST] = scol->lamp(3 * g + K) * P.warm; });
                    }
                }
                pmap = (pmap & ~(3u << (2 * g))) | ((unsigned)(on ? cntp : 3) << (2 * g));
                cntp += on ? 1 : 0;
            });
            rows(C_X) = __builtin_bit_cast(float, pmap);
            if (scol->dropped != nullptr && pdrop > 0) MI_ATOMIC_ADD_INT(scol->dropped + scol->dstride, pdrop);
        } }
        bar();                                                                                       // ---- B2: tree-pass exchange is dead, C_X is published
        // ============================================================ exchange for the sweeps; self-contact half rows
        sfor<NVT>([&](auto I) MI_LAMBDA { rows(X_DW + R * NVT + I) = dw[I]; });
        sfor<NLR_>([&](auto K) MI_LAMBDA { rows(DOWN + loff(R) + K) = w[LF + K]; });          // (round 0 carries the values themselves)
        rows(X_FLG + R) = act;
        if (selfcol) {
            // side a (SIDE 0, stage S1) / side b (SIDE 1, stage S2) of every self contact whose body on that side is one of mine
            auto pair_side = [&](auto SIDE_) MI_LAMBDA {
                constexpr int SIDE = decltype(SIDE_)::value;
                for (int j = 0; j < KPAIR; ++j) {
                    const float* xi = rows.ptr(C_X + 1 + XI * j);
                    const unsigned bab = __builtin_bit_cast(unsigned, xi[6 * ST]);
                    const bool onj = bab != 0xFFFFFFFFu;
                    if (!MI_WAVE_ANY(onj)) break;                    // slots fill from the front
                    const int bs = (SIDE == 0) ? (int)(bab & 255u) : (int)((bab >> 8) & 255u);
                    const bool fold = ((bab >> 24) & 1u) != 0u;
                    float* pb = rows.ptr(P_B + j * P_CSZ);
                    sfor<NB>([&](auto B_) MI_LAMBDA {
                        constexpr int b = B_;
                        if constexpr (MW::template owns_body<R>(b)) {
                            const bool me = onj && (bs == b);
                            if (MI_WAVE_ANY(me)) {
                                if (me) {
                                    float x[3], fr[3][3], W[3][6];
                                    sfor<3>([&](auto I_) MI_LAMBDA { x[I_] = xi[I_ * ST]; fr[0][I_] = xi[(3 + I_) * ST]; });
                                    contact_frame(fr[0], fr[1], fr[2]);
                                    sfor<3>([&](auto K) MI_LAMBDA {
                                        cross3(x, fr[K], W[K]);
                                        W[K][3] = fr[K][0]; W[K][4] = fr[K][1]; W[K][5] = fr[K][2];
                                    });
                                    float g[3][M::MAXCHAIN];
                                    rows3(std::integral_constant<int, b>{}, W, g);
                                    constexpr int rb_ = MW::role_of_body(b);
                                    sfor<3>([&](auto K) MI_LAMBDA {
                                        constexpr int k = K;
                                        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA {
                                            constexpr int gi = M::chain[b][C];
                                            if constexpr (MW::trunk_gi(gi)) {
                                                constexpr int t = MW::tidx(gi);
                                                if constexpr (SIDE == 0) pb[(PT + k * NVT + t) * ST] = g[k][C];
                                                else pb[(PT + k * NVT + t) * ST] -= g[k][C];
                                            } else {
                                                constexpr int kk = gi - lfirst(rb_ < 0 ? 0 : rb_);
                                                if constexpr (SIDE == 0) pb[(PLA + k * NLMAX + kk) * ST] = g[k][C];
                                                else { if (fold) pb[(PLA + k * NLMAX + kk) * ST] -= g[k][C]; else pb[(PLB + k * NLMAX + kk) * ST] = -g[k][C]; }
                                            }
                                        });
                                        // the rest of the fixed shape: zeros (side a owns the trunk part and its limb part, side b its limb part)
                                        if constexpr (SIDE == 0) sfor<NVT>([&](auto T_) MI_LAMBDA { if constexpr (!chain_has_t(b, T_)) pb[(PT + k * NVT + T_) * ST] = 0.f; });
                                        if constexpr (rb_ >= 0) sfor<NLMAX>([&](auto K2) MI_LAMBDA {
                                            if constexpr (!chain_has_l(b, K2)) {
                                                if constexpr (SIDE == 0) pb[(PLA + k * NLMAX + K2) * ST] = 0.f;
                                                else { if (!fold) pb[(PLB + k * NLMAX + K2) * ST] = 0.f; }
                                            }
                                        });
                                    });
                                }
                            }
                        }
                    });
                }
            };
            pair_side(std::integral_constant<int, 0>{});
            bar();                                                                                   // ---- B3
            pair_side(std::integral_constant<int, 1>{});
            bar();                                                                                   // ---- B4: rows complete
        }
        // limb part of a self-contact row / of w for the limb role word `r1` (1 + role, 0: none): run-time offsets into [NVL] vectors
        auto lofs = [&](const unsigned r1) MI_LAMBDA -> int {
            int o = 0;
            sfor<NR>([&](auto R_) MI_LAMBDA { o = (r1 == (unsigned)(R_ + 1)) ? loff(R_) : o; });
            return o;
        };
        auto lnum = [&](const unsigned r1) MI_LAMBDA -> int {
            int n = 0;
            sfor<NR>([&](auto R_) MI_LAMBDA { n = (r1 == (unsigned)(R_ + 1)) ? nl(R_) : n; });
            return n;
        };
        if constexpr (PAIRW) { if (selfcol) {
            // warm start of the self-contact rows: their contribution to the trunk part (block NR of X_DW) and to the limbs (DPAIR)
            float dtp[NVT];
            sfor<NVT>([&](auto I) MI_LAMBDA { dtp[I] = 0.f; });
            sfor<NVL>([&](auto I) MI_LAMBDA { rows(DPAIR + I) = 0.f; });
            float actp = 0.f;
            unsigned touch = 0u;
            for (int j = 0; j < KPAIR; ++j) {
                const float* xi = rows.ptr(C_X + 1 + XI * j);
                const unsigned bab = __builtin_bit_cast(unsigned, xi[6 * ST]);
                const bool onj = bab != 0xFFFFFFFFu;
                if (!MI_WAVE_ANY(onj)) break;
                if (onj) {
                    float* pb = rows.ptr(P_B + j * P_CSZ);
                    const unsigned ra1 = (bab >> 16) & 15u, rb1 = (bab >> 20) & 15u;
                    const int oa = lofs(ra1), ob = lofs(rb1), na = lnum(ra1), nb_ = lnum(rb1);
                    touch |= (ra1 ? 1u << (ra1 - 1) : 0u) | (rb1 ? 1u << (rb1 - 1) : 0u);
                    actp = 1.f;
                    pb[PVT * ST] = xi[8 * ST];
                    sfor<3>([&](auto K) MI_LAMBDA {
                        constexpr int k = K;
                        const float l0 = xi[(9 + k) * ST];
                        pb[(PLAM + k) * ST] = l0;
                        sfor<NVT>([&](auto T_) MI_LAMBDA { dtp[T_] += pb[(PT + k * NVT + T_) * ST] * l0; });
                        sfor<NLMAX>([&](auto K2) MI_LAMBDA {
                            if (K2 < na) rows.ptr(DPAIR + oa + K2)[0] += pb[(PLA + k * NLMAX + K2) * ST] * l0;
                            if (K2 < nb_) rows.ptr(DPAIR + ob + K2)[0] += pb[(PLB + k * NLMAX + K2) * ST] * l0;
                        });
                    });
                }
            }
            sfor<NVT>([&](auto I) MI_LAMBDA { rows(X_DW + NR * NVT + I) = dtp[I]; });
            rows(X_FLG + NR) = actp;
            rows(X_TOUCH) = __builtin_bit_cast(float, touch);
        } }
        if constexpr (PAIRW) { if (!selfcol) { rows(X_FLG + NR) = 0.f; rows(X_TOUCH) = 0.f; sfor<NVT>([&](auto I) MI_LAMBDA { rows(X_DW + NR * NVT + I) = 0.f; }); sfor<NVL>([&](auto I) MI_LAMBDA { rows(DPAIR + I) = 0.f; }); } }
        if constexpr (NPG == 0 && R == PAIR_ROLE) { rows(X_FLG + NR) = 0.f; rows(X_TOUCH) = 0.f; sfor<NVT>([&](auto I) MI_LAMBDA { rows(X_DW + NR * NVT + I) = 0.f; }); sfor<NVL>([&](auto I) MI_LAMBDA { rows(DPAIR + I) = 0.f; }); }
        bar();                                                                                       // ---- B5: round 0 of the exchange is complete
        // ============================================================ P4: block sweeps
        // round 0: every block's warm-start contribution (trunk part in block order; own limb part: the pair block's)
        sfor<NV>([&](auto I) MI_LAMBDA {
            constexpr int i = I;
            if constexpr (MW::trunk_gi(i)) { sfor<NBLK>([&](auto B_) MI_LAMBDA { constexpr int o = X_DW + B_ * NVT + MW::tidx(i); w[i] += rows(o); }); }
            else if constexpr (MW::role_of_gi(i) == R) { constexpr int o = DPAIR + loff(R) + (i - LF); w[i] += rows(o); }
        });
        if constexpr (PAIRW) sfor<NVL>([&](auto I) MI_LAMBDA { rows(PW + I) = rows(DOWN + I) + rows(DPAIR + I); });
        const unsigned touch = __builtin_bit_cast(unsigned, (float)rows(X_TOUCH));
        {
            float wtl[NVT], wll[NLR_ > 0 ? NLR_ : 1];
            for (int it = 0; it < P.iters; ++it) {
                int zero;
                MI_OPAQUE_ZERO(zero);
                const RowStore<RS> rit = rows.shifted(zero);
                const int par = it & 1;
                const RowStore<RS> xdw = rows.shifted((X_DW + (par ^ 1) * NBLK * NVT) * ST), fin = rows.shifted((X_FLG + par * NBLK) * ST),
                                   fout = rows.shifted((X_FLG + (par ^ 1) * NBLK) * ST), down = rows.shifted((DOWN + (par ^ 1) * NVL) * ST),
                                   dpair = rows.shifted((DPAIR + (par ^ 1) * NVL) * ST);
                // weights of this sweep: the trunk is shared by all active blocks, a limb by its owner's block and the pair block
                float fl[NBLK];
                sfor<NBLK>([&](auto B_) MI_LAMBDA { fl[B_] = fin(B_); });
                float nact = 0.f;
                sfor<NBLK>([&](auto B_) MI_LAMBDA { nact += fl[B_]; });
                const float omT = (nact > 1.5f) ? 0.5f * (nact + 1.f) : 1.f, iomT = 1.f / omT;
                float oml[NR];      // weight of every role's limb coordinates
                sfor<NR>([&](auto R_) MI_LAMBDA { oml[R_] = (fl[R_] + (((touch >> R_) & 1u) ? fl[NR] : 0.f) > 1.5f) ? 1.5f : 1.f; });
                const float omL = oml[R], iomL = 1.f / omL;
                sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int ti = MW::tidx(I); wtl[ti] = w[I]; } });
                sfor<NLR_>([&](auto K) MI_LAMBDA { wll[K] = w[LF + K]; });
                auto wget = [&](auto GI) MI_LAMBDA -> float {
                    constexpr int gi = decltype(GI)::value;
                    if constexpr (MW::trunk_gi(gi)) { constexpr int ti = MW::tidx(gi); return wtl[ti]; } else { constexpr int k = gi - LF; return wll[k]; }
                };
                auto wupd = [&](auto GI, const float val) MI_LAMBDA {
                    constexpr int gi = decltype(GI)::value;
                    if constexpr (MW::trunk_gi(gi)) { constexpr int ti = MW::tidx(gi); wtl[ti] += omT * val; } else { constexpr int k = gi - LF; wll[k] += omL * val; }
                };
                float actn = 0.f;
                // ---- own limit rows
                sfor<ND>([&](auto D) MI_LAMBDA {
                    constexpr int d = D, gi = OFF + d;
                    if constexpr (M::dof_limited[d] && MW::template owns_gi<R>(gi)) {
                        constexpr int row = B::limrow(d), g0 = B::limoff(row);
                        float g[M::MAXCHAIN];
                        sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { g[K] = rit(g0 + K); });
                        float at = 0.f;
                        sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA {
                            constexpr int k = K, i = (k == 0) ? gi : M::anc[gi][k == 0 ? 0 : k - 1];
                            if constexpr (MW::trunk_gi(i)) at += g[k] * g[k];
                        });
                        float vn = g[0] * wget(std::integral_constant<int, gi>{});
                        sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { vn += g[1 + A_] * wget(std::integral_constant<int, M::anc[gi][A_]>{}); });
                        const float lo = rit(L_LAM + row);
                        const float nl_ = fmaxf(lo - (vn - rit(L_VT + row)) * MI_RCP(P.cfm + omL * rit(L_SL + row) + omT * at), 0.f);
                        const float dl = nl_ - lo;
                        rit(L_LAM + row) = nl_;
                        actn = (nl_ > 0.f) ? 1.f : actn;
                        wupd(std::integral_constant<int, gi>{}, g[0] * dl);
                        sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { wupd(std::integral_constant<int, M::anc[gi][A_]>{}, g[1 + A_] * dl); });
                    }
                });
                // ---- own ground contacts: a lane's j-th contact, whichever sphere it is (fixed row shape [limb | trunk])
                for (int j = 0; j < KCAP; ++j) {
                    const bool onj = j < cnt;
                    if (!MI_WAVE_ANY(onj)) break;
                    if (onj) {
                        float* cb = rit.ptr(GCB + j * GCSZ);
                        const float mu = 0.5f * ((mu_env >= 0.f ? mu_env : M::sph_mu[0]) + P.plane_mu);
                        float g[3][RLEN], ainv[3], lm[3];
                        sfor<3>([&](auto K) MI_LAMBDA {
                            sfor<RLEN>([&](auto C) MI_LAMBDA { g[K][C] = cb[(K * RLEN + C) * ST]; });
                            float at = 0.f;
                            sfor<NVT>([&](auto T_) MI_LAMBDA { at += g[K][NLR_ + T_] * g[K][NLR_ + T_]; });
                            ainv[K] = MI_RCP(P.cfm + omL * cb[(3 * RLEN + K) * ST] + omT * at);
                            lm[K] = cb[(3 * RLEN + 4 + K) * ST];
                        });
                        const float vtn = cb[(3 * RLEN + 3) * ST];
                        auto dotw = [&](const float (&gr)[RLEN]) MI_LAMBDA -> float {
                            float s = 0.f;
                            sfor<NLR_>([&](auto K) MI_LAMBDA { s += gr[K] * wll[K]; });
                            sfor<NVT>([&](auto T_) MI_LAMBDA { s += gr[NLR_ + T_] * wtl[T_]; });
                            return s;
                        };
                        auto addw = [&](const float (&gr)[RLEN], const float dl) MI_LAMBDA {
                            sfor<NLR_>([&](auto K) MI_LAMBDA { wll[K] += omL * gr[K] * dl; });
                            sfor<NVT>([&](auto T_) MI_LAMBDA { wtl[T_] += omT * gr[NLR_ + T_] * dl; });
                        };
                        const float ln = fmaxf(lm[0] - (dotw(g[0]) - vtn) * ainv[0], 0.f);
                        addw(g[0], ln - lm[0]);
                        float lt[2];
                        sfor<2>([&](auto K) MI_LAMBDA {
                            const float dl = -dotw(g[1 + K]) * ainv[1 + K];
                            lt[K] = lm[1 + K] + dl;
                            addw(g[1 + K], dl);
                        });
                        const float lim = mu * ln;
                        const float n2 = lt[0] * lt[0] + lt[1] * lt[1];
                        const float sc = (n2 > lim * lim) ? lim * MI_RSQ(fmaxf(n2, 1e-30f)) : 1.f;
                        cb[(3 * RLEN + 4) * ST] = ln;
                        actn = (ln > 0.f) ? 1.f : actn;
                        sfor<2>([&](auto K) MI_LAMBDA {
                            const float nl_ = lt[K] * sc;
                            cb[(3 * RLEN + 5 + K) * ST] = nl_;
                            addw(g[1 + K], nl_ - lt[K]);
                        });
                    }
                }
                // this block's true contributions
                sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int ti = MW::tidx(I); xdw(R * NVT + ti) = (wtl[ti] - w[I]) * iomT; } });
                float dlo[NLR_ > 0 ? NLR_ : 1];
                sfor<NLR_>([&](auto K) MI_LAMBDA { dlo[K] = (wll[K] - w[LF + K]) * iomL; down(loff(R) + K) = dlo[K]; });
                fout(R) = actn;
                // ---- the self contacts: a block of their own, swept by this role from the same sweep-start velocity
                if constexpr (PAIRW) { if (selfcol) {
                    float wtp[NVT];
                    sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int ti = MW::tidx(I); wtp[ti] = w[I]; } });
                    sfor<NVL>([&](auto I) MI_LAMBDA { dpair(I) = 0.f; });
                    float actp = 0.f;
                    for (int j = 0; j < KPAIR; ++j) {
                        const float* xi = rit.ptr(C_X + 1 + XI * j);
                        const unsigned bab = __builtin_bit_cast(unsigned, xi[6 * ST]);
                        const bool onj = bab != 0xFFFFFFFFu;
                        if (!MI_WAVE_ANY(onj)) break;
                        if (onj) {
                            float* pb = rit.ptr(P_B + j * P_CSZ);
                            const unsigned ra1 = (bab >> 16) & 15u, rb1 = (bab >> 20) & 15u;
                            const int oa = lofs(ra1), ob = lofs(rb1), na = lnum(ra1), nb_ = lnum(rb1);
                            float omA = 1.f, omB = 1.f;
                            sfor<NR>([&](auto R_) MI_LAMBDA { omA = (ra1 == (unsigned)(R_ + 1)) ? oml[R_] : omA; omB = (rb1 == (unsigned)(R_ + 1)) ? oml[R_] : omB; });
                            const float mu = xi[7 * ST];
                            // the two limbs' part of w as this block sees it: sweep-start value + weight x what the block has contributed so far
                            float wa[NLMAX], wb[NLMAX], da[NLMAX], db[NLMAX];
                            sfor<NLMAX>([&](auto K2) MI_LAMBDA {
                                wa[K2] = (K2 < na) ? rit.ptr(PW + oa + K2)[0] + omA * dpair.ptr(oa + K2)[0] : 0.f;
                                wb[K2] = (K2 < nb_) ? rit.ptr(PW + ob + K2)[0] + omB * dpair.ptr(ob + K2)[0] : 0.f;
                                da[K2] = 0.f; db[K2] = 0.f;
                            });
                            float ga[3][NLMAX], gb[3][NLMAX], gt[3][NVT], ainv[3], lm[3];
                            sfor<3>([&](auto K) MI_LAMBDA {
                                float sa = 0.f, sb = 0.f, st = 0.f;
                                sfor<NLMAX>([&](auto K2) MI_LAMBDA {
                                    ga[K][K2] = (K2 < na) ? pb[(PLA + K * NLMAX + K2) * ST] : 0.f;
                                    gb[K][K2] = (K2 < nb_) ? pb[(PLB + K * NLMAX + K2) * ST] : 0.f;
                                    sa += ga[K][K2] * ga[K][K2]; sb += gb[K][K2] * gb[K][K2];
                                });
                                sfor<NVT>([&](auto T_) MI_LAMBDA { gt[K][T_] = pb[(PT + K * NVT + T_) * ST]; st += gt[K][T_] * gt[K][T_]; });
                                ainv[K] = MI_RCP(P.cfm + omA * sa + omB * sb + omT * st);
                                lm[K] = pb[(PLAM + K) * ST];
                            });
                            const float vtn = pb[PVT * ST];
                            auto dotw = [&](const int r) MI_LAMBDA -> float {
                                float s = 0.f;
                                sfor<3>([&](auto K) MI_LAMBDA { if (r == K) {
                                    sfor<NLMAX>([&](auto K2) MI_LAMBDA { s += ga[K][K2] * wa[K2] + gb[K][K2] * wb[K2]; });
                                    sfor<NVT>([&](auto T_) MI_LAMBDA { s += gt[K][T_] * wtp[T_]; });
                                } });
                                return s;
                            };
                            auto addw = [&](const int r, const float dl) MI_LAMBDA {
                                sfor<3>([&](auto K) MI_LAMBDA { if (r == K) {
                                    sfor<NLMAX>([&](auto K2) MI_LAMBDA {
                                        wa[K2] += omA * ga[K][K2] * dl; da[K2] += ga[K][K2] * dl;
                                        wb[K2] += omB * gb[K][K2] * dl; db[K2] += gb[K][K2] * dl;
                                    });
                                    sfor<NVT>([&](auto T_) MI_LAMBDA { wtp[T_] += omT * gt[K][T_] * dl; });
                                } });
                            };
                            const float ln = fmaxf(lm[0] - (dotw(0) - vtn) * ainv[0], 0.f);
                            addw(0, ln - lm[0]);
                            float lt[2];
                            sfor<2>([&](auto K) MI_LAMBDA {
                                const float dl = -dotw(1 + K) * ainv[1 + K];
                                lt[K] = lm[1 + K] + dl;
                                addw(1 + K, dl);
                            });
                            const float lim = mu * ln;
                            const float n2 = lt[0] * lt[0] + lt[1] * lt[1];
                            const float sc = (n2 > lim * lim) ? lim * MI_RSQ(fmaxf(n2, 1e-30f)) : 1.f;
                            pb[PLAM * ST] = ln;
                            actp = (ln > 0.f) ? 1.f : actp;
                            sfor<2>([&](auto K) MI_LAMBDA {
                                const float nl_ = lt[K] * sc;
                                pb[(PLAM + 1 + K) * ST] = nl_;
                                addw(1 + K, nl_ - lt[K]);
                            });
                            sfor<NLMAX>([&](auto K2) MI_LAMBDA {
                                if (K2 < na) dpair.ptr(oa + K2)[0] += da[K2];
                                if (K2 < nb_) dpair.ptr(ob + K2)[0] += db[K2];
                            });
                        }
                    }
                    sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int ti = MW::tidx(I); xdw(NR * NVT + ti) = (wtp[ti] - w[I]) * iomT; } });
                    fout(NR) = actp;
                } else { sfor<NVT>([&](auto I) MI_LAMBDA { xdw(NR * NVT + I) = 0.f; }); sfor<NVL>([&](auto I) MI_LAMBDA { dpair(I) = 0.f; }); fout(NR) = 0.f; } }
                if constexpr (NPG == 0 && R == PAIR_ROLE) { sfor<NVT>([&](auto I) MI_LAMBDA { xdw(NR * NVT + I) = 0.f; }); sfor<NVL>([&](auto I) MI_LAMBDA { dpair(I) = 0.f; }); fout(NR) = 0.f; }
                bar();                                                                               // ---- one barrier per sweep
                sfor<NV>([&](auto I) MI_LAMBDA {
                    constexpr int i = I;
                    if constexpr (MW::trunk_gi(i)) { sfor<NBLK>([&](auto B_) MI_LAMBDA { constexpr int o = B_ * NVT + MW::tidx(i); w[i] += xdw(o); }); }
                    else if constexpr (MW::role_of_gi(i) == R) { constexpr int k = i - LF; w[i] += dlo[k] + dpair(loff(R) + k); }
                });
                if constexpr (PAIRW) sfor<NVL>([&](auto I) MI_LAMBDA { rit(PW + I) += down(I) + dpair(I); });
            }
        }
        // ============================================================ P5: back to generalised velocity, outputs, integration
        sfor<NV>([&](auto I_) MI_LAMBDA {
            constexpr int i = I_;
            if constexpr (MW::template sees_gi<R>(i)) {
                float s = w[i];
                sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { s -= L[M::midx[i][M::anc[i][A_]]] * v[M::anc[i][A_]]; });
                v[i] = s * Ldi[i];
            }
        });
        MI_PHASE();
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            if constexpr (MW::template owns_gi<R>(OFF + d)) {
                float ll = 0.f;
                if constexpr (M::dof_limited[d]) {
                    constexpr int row = B::limrow(d);
                    const float dl = q[d] - this->template limit_lower<d>(), du = this->template limit_upper<d>() - q[d];
                    const float lr = rows(L_LAM + row);
                    ll = (dl < du) ? lr : -lr;
                }
                laml(d) = ll;
                dof_force(d) = tau[d] - M::dof_stiffness[d] * sc_stiff * (q[d] - M::dof_springref[d]) - M::dof_damping[d] * sc_damp * v[OFF + d] + ll * invh;
            }
        });
        float sens[6 * M::NSENSA];
        sfor<6 * NSENS>([&](auto K) MI_LAMBDA { sens[K] = 0.f; });
        sfor<NSPH>([&](auto S_) MI_LAMBDA {
            constexpr int s = S_, b = M::sph_body[s];
            if constexpr (MW::template owns_body<R>(b)) {
                const int j = (int)slot8(s);
                const bool onj = j >= 0;
                const float* cb = rows.ptr(GCB + (onj ? j : 0) * GCSZ);
                const float ln = onj ? cb[(3 * RLEN + 4) * ST] : 0.f, l1 = onj ? cb[(3 * RLEN + 5) * ST] : 0.f, l2 = onj ? cb[(3 * RLEN + 6) * ST] : 0.f;
                lamc(3 * s) = ln; lamc(3 * s + 1) = l1; lamc(3 * s + 2) = l2;
                if constexpr (B::sensor_of(b) >= 0) {
                    constexpr int k = B::sensor_of(b);
                    const float f[3] = {l1 * invh, l2 * invh, ln * invh};
                    const float arm[3] = {c.xcs[s][0] - c.rs[k][0], c.xcs[s][1] - c.rs[k][1], c.xcs[s][2] - M::sph_rad[s] - c.rs[k][2]};
                    float tq[3], flo[3], tl[3];
                    cross3(arm, f, tq);
                    matTvec3(c.Rs[k], f, flo); matTvec3(c.Rs[k], tq, tl);
                    sfor<3>([&](auto C) MI_LAMBDA { sens[6 * k + C] += flo[C]; sens[6 * k + 3 + C] += tl[C]; });
                }
            }
        });
        if constexpr (NPG > 0) { if (selfcol) {
            // self contacts on the own sensor bodies: +f on side a, -f on side b, at the contact point
            sfor<KPAIR>([&](auto J_) MI_LAMBDA {
                const float* xi = rows.ptr(C_X + 1 + XI * J_);
                const unsigned bab = __builtin_bit_cast(unsigned, xi[6 * ST]);
                const bool onj = bab != 0xFFFFFFFFu;
                const float* pb = rows.ptr(P_B + J_ * P_CSZ);
                const float ln = onj ? pb[PLAM * ST] : 0.f, l1 = onj ? pb[(PLAM + 1) * ST] : 0.f, l2 = onj ? pb[(PLAM + 2) * ST] : 0.f;
                float x[3], n[3], t1[3], t2[3], f[3];
                sfor<3>([&](auto I_) MI_LAMBDA { x[I_] = xi[I_ * ST]; n[I_] = onj ? xi[(3 + I_) * ST] : (I_ == 2 ? 1.f : 0.f); });
                contact_frame(n, t1, t2);
                sfor<3>([&](auto K) MI_LAMBDA { f[K] = (n[K] * ln + t1[K] * l1 + t2[K] * l2) * invh; });
                sfor<NSENS>([&](auto K_) MI_LAMBDA {
                    constexpr int k = K_, sb = M::sens_body[k];
                    if constexpr (MW::template owns_body<R>(sb)) {
                        const float cf = onj ? (((int)(bab & 255u) == sb ? 1.f : 0.f) - ((int)((bab >> 8) & 255u) == sb ? 1.f : 0.f)) : 0.f;
                        const float arm[3] = {x[0] - c.rs[k][0], x[1] - c.rs[k][1], x[2] - c.rs[k][2]};
                        float tq[3], flo[3], tl[3];
                        cross3(arm, f, tq);
                        matTvec3(c.Rs[k], f, flo); matTvec3(c.Rs[k], tq, tl);
                        sfor<3>([&](auto C) MI_LAMBDA { sens[6 * k + C] += cf * flo[C]; sens[6 * k + 3 + C] += cf * tl[C]; });
                    }
                });
            });
            if constexpr (PAIRW) {
                // impulses of the groups -> warm start of the next sub-step, world force on side a
                sfor<NPG>([&](auto G_) MI_LAMBDA {
                    constexpr int g = G_;
                    const int j = (int)((pmap >> (2 * g)) & 3u);
                    const bool onj = j != 3;
                    const float* pb = rows.ptr(P_B + (onj ? j : 0) * P_CSZ);
                    const float* xi = rows.ptr(C_X + 1 + XI * (onj ? j : 0));
                    const float ln = onj ? pb[PLAM * ST] : 0.f, l1 = onj ? pb[(PLAM + 1) * ST] : 0.f, l2 = onj ? pb[(PLAM + 2) * ST] : 0.f;
                    scol->lamp(3 * g) = ln; scol->lamp(3 * g + 1) = l1; scol->lamp(3 * g + 2) = l2;
                    if (scol->pairf.p) {
                        float n[3], t1[3], t2[3];
                        sfor<3>([&](auto I_) MI_LAMBDA { n[I_] = onj ? xi[(3 + I_) * ST] : (I_ == 2 ? 1.f : 0.f); });
                        contact_frame(n, t1, t2);
                        sfor<3>([&](auto K) MI_LAMBDA { scol->pairf(3 * g + K) = (n[K] * ln + t1[K] * l1 + t2[K] * l2) * invh; });
                    }
                });
            }
        } }
        sfor<NSENS>([&](auto K_) MI_LAMBDA {
            if constexpr (MW::template owns_body<R>(M::sens_body[K_])) sfor<6>([&](auto C) MI_LAMBDA { sensor(6 * K_ + C) = sens[6 * K_ + C]; });
        });
        MI_PHASE();
        sfor<ND>([&](auto D) MI_LAMBDA {
            if constexpr (MW::template owns_gi<R>(OFF + D)) { qd[D] = v[OFF + D]; q[D] += h * qd[D]; }
        });
        if constexpr (R == M::TRUNK_ROLE) {
#if !defined(MI_NO_VEL_CLAMP)
            {
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
    }
};

}  // namespace mi
