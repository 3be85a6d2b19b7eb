// hand_engine_mw.hpp -- the finger-per-wave form of the Shadow-Hand sub-step (replaces gym.simulate() for reference
// isaacgymenvs/tasks/shadow_hand.py; the one-wave form is core/hand_engine.hpp).
//
// Why: the one-wave sub-step spends 56 % of its time in the 8 Gauss-Seidel sweeps and 24 % in narrow phase + contact rows, all on ONE
// wavefront per 32 envs that executes the union of its envs' contact sets, with its live state (the whole factor L, 24 joint axes, 216
// pose floats) overflowing into 1.4 KB of scratch per lane -- while three SIMDs of four idle.  The hand is fixed-base: its fingers
// couple only through the two wrist dofs and, when they touch it, through the object's six.  So, as in core/engine_mw.hpp, one env's
// sub-step is spread over the NROLE waves of a workgroup, every wave running different code for the same 32 envs:
//
//     role r   owns the limbs (fingers) dealt to it -- their tree pass, factor, joint-limit rows, object contacts, sweeps, outputs;
//     the trunk (forearm, wrist, palm: limb 0, two dofs) is recomputed by every wave; its rows (wrist limits, palm contacts) belong to
//     role TRUNK_ROLE, which also integrates the object.
//
//   P1  all roles   trunk going down, own limbs down + up (gravity off on the hand); the NARROW PHASE of a body's spheres runs right
//                   where the tree pass has its pose (conservative bounding tests skip far bodies / spheres for the whole wave; no
//                   pose is handed on): the geometry of the kept contacts (normal, lever, velocity target) goes into the limb's OWN
//                   slots (caps M::limb_kcap, <= BODY_CAP per body); drives / tendons, limb factor + whitened velocity; Schur
//                   complement of the limbs on the wrist block, limb-root composite inertias, right-hand-side carry -> LDS          | B1
//   P2  all roles   (redundantly) trunk coming up, wrist factor and whitened velocity; the object's whitened velocity                 | B1b
//                   (B1b frees the exchange area of P1 / P2: the contacts' rows live in the same LDS)
//   P3  all roles   joint-limit rows of the own dofs -- kept in REGISTERS (<= 9 rows per role); the rows of the own contacts, in the
//                   fixed shape [limb dofs | wrist dofs] shared by every body of the limb (the object part of a row follows from the
//                   stored normal and lever)                                                                                        | B2
//   P4  all roles   block sweeps (oracle/physics.c solve_blocks, oracle/hand.c solver 1): every wave Gauss-Seidel over its own rows --
//                   limb by limb the limit rows, then the limb's contacts in a run-time loop over the lane's ACTUAL contacts; the
//                   coordinates the blocks share are treated Jacobi-fashion with mass splitting: the wrist pair answers the n_W active
//                   blocks with weight (n_W + 1) / 2, the object's six the n_O active blocks that hold a contact with (n_O + 1) / 2;
//                   after every sweep the blocks' true contributions are summed in role order (one barrier per sweep, double buffered)
//   P5  all roles   velocities of wrist + own limbs, limit impulses / joint forces / fingertip sensors of the own rows, integration;
//                   TRUNK_ROLE: the object, the contact counters
//
// Same arithmetic per row as HandSim<M>::substep_hand; sums over roles run in role order, so results do not depend on wave timing.
// LDS per env: <= 640 floats (two 32-env workgroups per CU, as the one-wave form).
#pragma once
#include "hand_engine.hpp"
#include "engine_mw.hpp"

namespace mi {

template <class M>
struct HandSimMW : HandSim<M> {
    using HB = HandSim<M>;
    using B = Sim<M>;
    using MW = SimMW<M>;          // ownership tables only (static members)
    using typename B::Ctx;
    using typename B::BodyTmp;
    static constexpr int NB = M::NB, ND = M::ND, NV = M::NV, OFF = M::OFF, NSENS = M::NSENS, NLIM = B::NLIM, NVA = B::NVA, NR = M::NROLE,
                         NLIMB = M::NLIMB, NVT = MW::NVT, NTE = MW::NTE, NLR = MW::NLR, NTB = MW::NTB, BODY_CAP = HB::BODY_CAP;
    static_assert(M::FIXED == 1 && OFF == 0 && M::NSPH == 0 && M::limb_of_body[0] == 0, "fixed-base manipulator, limb 0 = the trunk");
    static constexpr int NSH = NVT + 6;                                   // coordinates the blocks share: wrist dofs | object
    static constexpr int LANES = 32;                                      // envs per workgroup of the default launch shape (64: one workgroup per CU)
    // ---- limbs
    static constexpr int limb_of_gi(int gi) { return M::limb_of_body[M::dof_body[gi]]; }
    static constexpr int ldof0(int l) { for (int d = 0; d < ND; ++d) if (limb_of_gi(d) == l) return d; return ND; }
    static constexpr int lnd(int l) { int n = 0; for (int d = 0; d < ND; ++d) n += (limb_of_gi(d) == l) ? 1 : 0; return n; }
    static constexpr bool limbs_contiguous() {
        for (int l = 0; l < NLIMB; ++l) for (int d = ldof0(l); d < ldof0(l) + lnd(l); ++d) if (limb_of_gi(d) != l) return false;
        return true;
    }
    static_assert(limbs_contiguous(), "a limb's dofs are numbered consecutively (depth-first numbering)");
    static constexpr int limb_role(int l) { return l == 0 ? M::TRUNK_ROLE : M::role_of_limb[l]; }
    static constexpr int nl(int l) { return l == 0 ? 0 : lnd(l); }      // limb part of a row (the trunk limb's rows have none)
    static constexpr int rl(int l) { return nl(l) + NVT; }              // fixed row shape of limb l: [limb | wrist]
    static constexpr int csz(int l) { return 3 * rl(l) + 10; }          // slot: 3 rows | n 3, rc 3 | vt_n | lam x3
    static constexpr int kcap(int l) { return M::limb_kcap[l]; }
    static constexpr int shape_idx(int l, int gi) { return MW::trunk_gi(gi) ? nl(l) + MW::tidx(gi) : gi - ldof0(l); }
    static constexpr bool in_chain(int b, int l, int idx) {
        for (int c = 0; c < M::chain_len[b]; ++c) if (shape_idx(l, M::chain[b][c]) == idx) return true;
        return false;
    }
    // ---- LDS layout (floats per env): per contact slot 10 floats of geometry / state (n 3, rc 3, vt_n, lam x3) in one array over all
    //      limbs, the 3 rows [limb | wrist] in per-limb arrays behind it.  The exchange area of P1 / P2 (region A) aliases the ROWS arrays
    //      only: geometry is written during P1, rows after B1b.
    static constexpr int GSZ = 10, G_VT = 6, G_LAM = 7;
    static constexpr int gslot(int l) { int n = 0; for (int k = 0; k < l; ++k) n += kcap(k); return n; }
    static constexpr int KTOTAL = gslot(NLIMB);
    static constexpr int G0 = 0, R0 = G0 + GSZ * KTOTAL;
    static constexpr int rbase(int l) { int o = R0; for (int k = 0; k < l; ++k) o += kcap(k) * 3 * rl(k); return o; }
    static constexpr int R_END = rbase(NLIMB);
    static constexpr int X_LR = R0, X_DT = X_LR + 16 * NLR, X_DY = X_DT + NR * NTE, A_END = X_DY + NR * NVT;
    static_assert(A_END <= R_END, "the P1 / P2 exchange fits behind the geometry array");
    static constexpr int X_DW = R_END;                                    // [2][NR][NSH] the block's true contribution to the shared coordinates
    static constexpr int X_FLG = X_DW + 2 * NR * NSH;                     // [2][NR]      block active in this sweep
    static constexpr int X_HASC = X_FLG + 2 * NR;                         // [NR]         block holds a contact (touches the object's coordinates)
    static constexpr int X_CNT = X_HASC + NR;                             // [NLIMB]      contacts kept | refused << 16 (int bits)
    static constexpr int MW_SLOTS = X_CNT + NLIMB;
    // ---- the asset's hand-to-hand pairs (core/hand_engine.hpp pair_side): the axis end points of the pair capsules that sit on LIMB bodies go
    //      through LDS once per sub-step (6 floats each; the palm is trunk: every wave has its pose).  They alias the sweeps' exchange area,
    //      which is dead from the start of the sub-step until the end of P3.
    static constexpr int NHP = M::NHP;
    static constexpr int hp_slot(int b) {            // slot of limb body b among the pair capsules' bodies, or -1
        if (!HB::hp_capsule_body(b) || MW::trunk_body(b)) return -1;
        int n = 0;
        for (int k = 0; k < b; ++k) n += (HB::hp_capsule_body(k) && !MW::trunk_body(k)) ? 1 : 0;
        return n;
    }
    static constexpr int hp_nslots() { int n = 0; for (int b = 0; b < NB; ++b) n += (hp_slot(b) >= 0) ? 1 : 0; return n; }
    static constexpr int X_PE = X_DW;
    static_assert(X_PE + 6 * hp_nslots() <= MW_SLOTS, "the pair capsules' end points fit in the sweeps' exchange area");
    static constexpr bool hp_pairs_ok() {
        for (int p = 0; p < NHP; ++p) {
            if (M::hp_box[p] && !MW::trunk_body(M::hp_ba[p])) return false;          // a box side's pose must be known to every wave
            if (!M::hp_box[p] && MW::trunk_body(M::hp_ba[p])) return false;          // (capsules on trunk bodies: not needed by the Shadow Hand)
            if (MW::trunk_body(M::hp_bb[p])) return false;
        }
        return true;
    }
    static_assert(hp_pairs_ok(), "pair shapes: boxes on trunk bodies, capsules on limb bodies");
    static constexpr int role_of_body_h(int b) { return limb_role(M::limb_of_body[b]); }
    static_assert((size_t)MW_SLOTS * LANES * sizeof(float) <= 80 * 1024, "two 32-env hand workgroups per CU, or one of 64 envs");
    // ---- conservative bounds for the narrow phase: bounding sphere (centre in the body frame, radius) of the spheres of body b
    static constexpr float csqrt(float x) { float r = x > 1.f ? x : 1.f; for (int i = 0; i < 40; ++i) r = 0.5f * (r + x / r); return r; }
    static constexpr float bs_centre(int b, int k) {
        float s = 0.f; int n = 0;
        for (int i = 0; i < M::NOS; ++i) if (M::os_body[i] == b) { s += M::os_pos[i][k]; ++n; }
        return n ? s / (float)n : 0.f;
    }
    static constexpr float bs_radius(int b) {
        float r = 0.f;
        for (int i = 0; i < M::NOS; ++i) if (M::os_body[i] == b) {
            const float dx = M::os_pos[i][0] - bs_centre(b, 0), dy = M::os_pos[i][1] - bs_centre(b, 1), dz = M::os_pos[i][2] - bs_centre(b, 2);
            const float d = csqrt(dx * dx + dy * dy + dz * dz) + M::os_rad[i];
            r = d > r ? d : r;
        }
        return r * 1.001f + 1e-5f;
    }
    // what the narrow phase carries through the tree pass
    struct Narrow {
        float Ro[9], xo[3];                 // object frame, COM relative to O
        float reach;                        // bounding radius of the object + contact offset, with a safety margin
        int cnt[NLIMB], refused[NLIMB];     // contacts kept / refused for want of a slot, per own limb
        unsigned bfc[B::NOSB > 0 ? B::NOSB : 1];   // per sphere-carrying body: first slot (inside its limb's range) | count << 8
    };
    // ---- own joint-limit rows (registers)
    template <int R> static constexpr int nownl() { int n = 0; for (int d = 0; d < ND; ++d) n += (M::dof_limited[d] && MW::template owns_gi<R>(d)) ? 1 : 0; return n; }
    template <int R> static constexpr int own_lim_idx(int d) { int n = 0; for (int k = 0; k < d; ++k) n += (M::dof_limited[k] && MW::template owns_gi<R>(k)) ? 1 : 0; return n; }
    struct LimReg { float g[M::MAXCHAIN]; float al, at, vt, lam, rho; };      // rho: the drive clamp's impulse on the dof (core/hand_engine.hpp drive_clamp_update)

    // ------------------------------------------------------------------------------------------------ narrow phase of one body's spheres
    // Spheres in the body's (farthest-point) order; kept: distance below the contact offset, < BODY_CAP on this body, a free slot in the
    // limb's range.  The two bounding tests only skip work no lane of the wave needs: every evaluated sphere goes through the same
    // arithmetic as in the one-wave form.
    template <int SHAPE, int b, int RS>
    MI_HD void narrow_body(const SimParams& P, const ObjectParams& OP, const float invh, Narrow& nw, const float* Rb, const float* rb, const RowStore<RS> rows) {
        constexpr int ST = RowStore<RS>::stride;
        constexpr int l = M::limb_of_body[b], KCAP = kcap(l), GS0 = gslot(l), bs = B::os_slot(b);
        int& cnt = nw.cnt[l];
        const int first = cnt;
        int nbody = 0;
        nw.bfc[bs] = (unsigned)first;
        // object COM in the body frame
        float xb[3];
        {
            const float d[3] = {nw.xo[0] - rb[0], nw.xo[1] - rb[1], nw.xo[2] - rb[2]};
            matTvec3(Rb, d, xb);
        }
        {
            constexpr float bc[3] = {bs_centre(b, 0), bs_centre(b, 1), bs_centre(b, 2)};
            constexpr float br = bs_radius(b);
            const float d[3] = {xb[0] - bc[0], xb[1] - bc[1], xb[2] - bc[2]};
            const float lim = br + nw.reach;
            if (!MI_WAVE_ANY(dot3(d, d) < lim * lim)) return;
        }
        constexpr int OS_N = B::os_count(b), OS_0 = B::os_first(b);     // (constant expressions on purpose: see hand_engine.hpp)
        for (int i = 0; i < OS_N; ++i) {
            const int s = OS_0 + i;
            const float pl[3] = {M::os_pos[s][0], M::os_pos[s][1], M::os_pos[s][2]};
            const float rad = M::os_rad[s];
            {
                const float d[3] = {xb[0] - pl[0], xb[1] - pl[1], xb[2] - pl[2]};
                const float lim = rad * 1.001f + nw.reach;
                if (!MI_WAVE_ANY(dot3(d, d) < lim * lim)) continue;
            }
            float t[3], cs[3];
            matvec3(Rb, pl, t);
            sfor<3>([&](auto K) MI_LAMBDA { cs[K] = rb[K] + t[K]; });
            const float rel[3] = {cs[0] - nw.xo[0], cs[1] - nw.xo[1], cs[2] - nw.xo[2]};
            float cl[3], nloc[3], dist;
            matTvec3(nw.Ro, rel, cl);
            HB::template sphere_object<SHAPE>(cl, rad, OP, &dist, nloc);
            const bool nearc = (dist < P.contact_offset) && (nbody < BODY_CAP);
            const bool on = nearc && (cnt < KCAP);
            nw.refused[l] += (nearc && !on) ? 1 : 0;                      // all slots of the limb taken
            if (on) {
                float n[3], rc[3];
                matvec3(nw.Ro, nloc, n);                                 // from the object towards the sphere
                sfor<3>([&](auto K) MI_LAMBDA { rc[K] = (cs[K] - rad * n[K]) - nw.xo[K]; });
                float* gp = rows.ptr(G0 + (GS0 + cnt) * GSZ);
                const float gap = dist - P.rest_offset;
                sfor<3>([&](auto I_) MI_LAMBDA { gp[I_ * ST] = n[I_]; gp[(3 + I_) * ST] = rc[I_]; });
                gp[G_VT * ST] = (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
                sfor<3>([&](auto K) MI_LAMBDA { gp[(G_LAM + K) * ST] = 0.f; });     // no warm start for object contacts
            }
            nbody += on ? 1 : 0;
            cnt += on ? 1 : 0;
        }
        nw.bfc[bs] = (unsigned)first | ((unsigned)nbody << 8);
    }

    // ------------------------------------------------------------------------------------------------ a body of an own limb: down, narrow phase, children, up
    template <int SHAPE, int b, int RS>
    MI_HD void limb_pass(const SimParams& P, const ObjectParams& OP, const float invh, Ctx& c, Narrow& nw, const float* Rp, const float* rp,
                         const float* Vp, const float* Ap, SpI& Iout, float* Fout, const RowStore<RS> rows) {
        BodyTmp t;
        this->template body_down<b>(P, c, Rp, rp, Vp, Ap, t);
        if constexpr (hp_slot(b) >= 0) {       // this body's pair capsule, for the roles that own the other sides (read after barrier B0)
            if (this->pair_k > 0.f) {
                float e0[3], e1[3];
                HB::template hp_endpoints<b>(t.Rb, t.rb, e0, e1);
                sfor<3>([&](auto K) MI_LAMBDA { rows(X_PE + 6 * hp_slot(b) + K) = e0[K]; rows(X_PE + 6 * hp_slot(b) + 3 + K) = e1[K]; });
            }
        }
        if constexpr (B::os_count(b) > 0) narrow_body<SHAPE, b>(P, OP, invh, nw, t.Rb, t.rb, rows);
        sfor<NB>([&](auto C_) MI_LAMBDA {
            constexpr int ch = C_;
            if constexpr (ch > b) if constexpr (M::parent[ch] == b) {
                SpI Ic;
                float Fc[6];
                limb_pass<SHAPE, ch>(P, OP, invh, c, nw, t.Rb, t.rb, t.Vc, t.Ac, Ic, Fc, rows);
                sfor<6>([&](auto K) MI_LAMBDA { t.F[K] += Fc[K]; });
                t.I.m += Ic.m;
                sfor<3>([&](auto K) MI_LAMBDA { t.I.h[K] += Ic.h[K]; });
                sfor<6>([&](auto K) MI_LAMBDA { t.I.I[K] += Ic.I[K]; });
            }
        });
        this->template body_up<b>(c, t);
        Iout = t.I;
        sfor<6>([&](auto K) MI_LAMBDA { Fout[K] = t.F[K]; });
    }

    // ------------------------------------------------------------------------------------------------ trunk, going down (as SimMW::trunk_down)
    // P: the hand's parameters (gravity off).  The trunk's own spheres (wrist, palm) are looked at by TRUNK_ROLE.
    template <int R, int SHAPE, int b, int RS>
    MI_HD void trunk_down_h(const SimParams& P, const ObjectParams& OP, const float invh, Ctx& c, Narrow& nw, BodyTmp (&tb)[NTB], const float* Rp,
                            const float* rp, const float* Vp, const float* Ap, const RowStore<RS> rows) {
        constexpr int ts = MW::tslot(b);
        BodyTmp& t = tb[ts];
        this->template body_down<b>(P, c, Rp, rp, Vp, Ap, t);
        if constexpr (R == M::TRUNK_ROLE && B::os_count(b) > 0) narrow_body<SHAPE, b>(P, OP, invh, nw, t.Rb, t.rb, rows);
        sfor<NB>([&](auto C_) MI_LAMBDA {
            constexpr int ch = C_;
            if constexpr (ch > b) if constexpr (M::parent[ch] == b) {
                if constexpr (MW::trunk_body(ch)) {
                    trunk_down_h<R, SHAPE, ch>(P, OP, invh, c, nw, tb, t.Rb, t.rb, t.Vc, t.Ac, rows);
                } else if constexpr (MW::role_of_body(ch) == R) {     // root of one of my limbs: the whole subtree, down and up
                    SpI Ic;
                    float Fc[6];
                    limb_pass<SHAPE, ch>(P, OP, invh, c, nw, t.Rb, t.rb, t.Vc, t.Ac, Ic, Fc, rows);
                    constexpr int o = X_LR + 16 * MW::lridx(ch);
                    rows(o) = Ic.m;
                    sfor<3>([&](auto K) MI_LAMBDA { rows(o + 1 + K) = Ic.h[K]; });
                    sfor<6>([&](auto K) MI_LAMBDA { rows(o + 4 + K) = Ic.I[K]; rows(o + 10 + K) = Fc[K]; });
                }
            }
        });
    }

    // ------------------------------------------------------------------------------------------------ one role of a sub-step
    // target[ND]: drive targets; laml / sensor / dof_force: the own entries are read / written; ncontact: written by TRUNK_ROLE only
    // (contacts kept | refused for want of a slot << 16).  BAR: workgroup barrier (device: s_barrier; host tests: a thread barrier).
    // PSENS false: the pairs' forces on the fingertip sensors are left out at COMPILE time -- the instantiation for every sub-step launch of a call
    // but the last, whose sensor values are overwritten unseen (with the run-time flag alone the skipped launch cost as much as the other one:
    // the code's presence, not its execution, is what the register allocation pays for; profiles/r6_hand_pair_sensors_ab.txt)
    template <int R, int RS, int SHAPE, bool PSENS = true, class BAR = void>
    MI_HD void substep_hand_role(const SimParams& P, const ObjectParams& OP, const float* target, const float h, const RowStore<RS> rows,
                                 const Strided laml, const Strided sensor, const Strided dof_force, int* ncontact, const BAR& bar) {
        constexpr int ST = RowStore<RS>::stride;
        float (&q)[M::NDA] = this->q;
        float (&qd)[M::NDA] = this->qd;
        float (&root)[13] = this->root;
        FreeBody& obj = this->obj;
        const float invh = MI_RCP(h);
        const Strided limit_shift = this->limit_shift;
        Ctx c;
        float (&S)[M::NDA][6] = c.S;
        float (&L)[M::NM] = c.L;
#if defined(MI_TIMING)
        unsigned long long* const tstamp = this->tstamp;
#endif
        MI_STAMP(0);
        // the object's frame and whitened velocity (every role alike)
        const float sm = MI_SQRT(OP.mass), si = MI_SQRT(SHAPE == OBJ_BOX ? OP.inertia : 1.f);
        const float ism = MI_RCP(sm), isi = MI_RCP(si);
        float wo[6];
        sfor<3>([&](auto K) MI_LAMBDA { wo[K] = sm * (obj.vel[K] + h * (P.g[K] + OP.fw[K] * (ism * ism))); wo[3 + K] = si * obj.angvel[K]; });
        Narrow nw;
        float (&Ro)[9] = nw.Ro;
        float (&xo)[3] = nw.xo;
        quat2mat(obj.quat, Ro);
        float isqI[3] = {1.f, 1.f, 1.f};
        if constexpr (SHAPE != OBJ_BOX) {
            float sqI[3];
            sfor<3>([&](auto K) MI_LAMBDA { sqI[K] = MI_SQRT(OP.inertia3[K]); isqI[K] = MI_RCP(sqI[K]); });
            HB::body_diag(Ro, sqI, obj.angvel, wo + 3);
        }
        sfor<3>([&](auto K) MI_LAMBDA { xo[K] = obj.pos[K] - root[K]; });                         // object COM rel O
        {   // bounding radius of the object (box: half diagonal; capsule: radius + half length; ellipsoid: largest semi-axis)
            float ro;
            if constexpr (SHAPE == OBJ_BOX) ro = 1.7320508f * OP.half;
            else if constexpr (SHAPE == OBJ_CAPSULE) ro = OP.dims[0] + OP.dims[1];
            else ro = fmaxf(OP.dims[0], fmaxf(OP.dims[1], OP.dims[2]));
            nw.reach = ro * 1.001f + P.contact_offset + 1e-5f;
        }
        sfor<NLIMB>([&](auto L_) MI_LAMBDA { nw.cnt[L_] = 0; nw.refused[L_] = 0; });
        sfor<B::NOSB>([&](auto K) MI_LAMBDA { nw.bfc[K] = 0u; });
        // ============================================================ P1: tree pass with gravity off (disable_gravity on the hand) + narrow phase
        BodyTmp tb[NTB];
        {
            SimParams P0 = P;
            P0.g[0] = P0.g[1] = P0.g[2] = 0.f;
            float pose_sink;                      // (the one-wave form's pose hand-off: nobody reads it here)
            c.pose_out = &pose_sink;
            c.pose_stride = 0;
            trunk_down_h<R, SHAPE, 0>(P0, OP, invh, c, nw, tb, nullptr, nullptr, nullptr, nullptr, rows);
        }
        MI_PHASE();
        float Ldi[NVA], y[NVA], w[NVA], v[NVA];
        sfor<ND>([&](auto D) MI_LAMBDA { v[D] = qd[D]; });
        sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::trunk_entry(E_)) L[E_] = 0.f; });
        sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) y[I] = 0.f; });
        // `actor_params.hand` factors (core/hand_engine.hpp HS_*): H and the bias forces are linear in the link masses
        float sc_mass = 1.f, sc_damp = 1.f, sc_kp = 1.f, sc_tk = 1.f, sc_td = 1.f;
        if (this->actor_scale.p != nullptr) {
            sc_mass = this->actor_scale(HS_MASS); sc_damp = this->actor_scale(HS_DAMPING); sc_kp = this->actor_scale(HS_STIFFNESS);
            sc_tk = this->actor_scale(HS_TENDON_STIFFNESS); sc_td = this->actor_scale(HS_TENDON_DAMPING);
            sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::role_of_gi(MW::entry_row(E_)) == R) L[E_] *= sc_mass; });
            sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::role_of_gi(I) == R) c.bias[I] *= sc_mass; });
        }
        auto drive = [&](auto D) MI_LAMBDA {      // implicit PD drive + passive damping of dof D
            constexpr int d = decltype(D)::value;
            const float kp = M::dof_kp[d] * sc_kp, Dm = M::dof_damping[d] * sc_damp;
            L[M::midx[d][d]] += M::dof_armature[d] + h * Dm + h * h * kp;
            y[d] = -c.bias[d] - kp * (q[d] - target[d]) - (Dm + h * kp) * qd[d];
        };
        sfor<ND>([&](auto D) MI_LAMBDA { if constexpr (MW::role_of_gi(D) == R) drive(D); });
        sfor<M::NTEND>([&](auto T_) MI_LAMBDA {
            constexpr int t = T_, d0 = M::tend_d0[t], d1 = M::tend_d1[t];
            static_assert(!MW::trunk_gi(d0) && limb_of_gi(d0) == limb_of_gi(d1), "a tendon couples two joints of one finger");
            if constexpr (MW::role_of_gi(d0) == R) {
                constexpr float c0 = M::tend_c0[t], c1 = M::tend_c1[t];
                const float Lt = c0 * q[d0] + c1 * q[d1], Ld = c0 * qd[d0] + c1 * qd[d1];
                const float viol = Lt - fminf(fmaxf(Lt, M::tend_lo[t]), M::tend_hi[t]);
                const float k = (viol != 0.f) ? M::tend_stiffness * sc_tk : 0.f;
                const float td = M::tend_damping * sc_td;
                const float a = h * td + h * h * k;
                const float f = k * viol + (td + h * k) * Ld;
                L[M::midx[d0][d0]] += a * c0 * c0;
                L[M::midx[d1][d1]] += a * c1 * c1;
                if constexpr (M::midx[d0][d1] >= 0) L[M::midx[d0][d1]] += a * c0 * c1; else L[M::midx[d1][d0]] += a * c0 * c1;
                y[d0] -= c0 * f;
                y[d1] -= c1 * f;
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
        // ------------------------------------------------------------ the asset's hand-to-hand pairs: every role evaluates the pairs with a side on
        // one of its bodies (both owners compute the same geometry from the same end points) and adds ITS side's implicit spring to its own H
        // entries / right-hand side; the wrist part rides on the Schur-complement / carry exchange below (X_DT / X_DY, summed in role order)
        this->pair_active = 0;
        float psPN[M::NSENSA][2];                               // the pairs on this role's fingertips: sum of pen, sides (core/hand_engine.hpp pair_sensor_acc)
        sfor<NSENS>([&](auto K_) MI_LAMBDA { psPN[K_][0] = 0.f; psPN[K_][1] = 0.f; });
        float psG[M::NSENSA][M::MAXCHAIN];                      // ... and A . S_c over the fingertip's chain (pair_sensor_g)
        if constexpr (NHP > 0) {
            if (this->pair_k > 0.f) {                                                                 // (uniform over the workgroup: a kernel argument)
                bar();                                                                               // ---- B0: everybody's pair capsules are in LDS
                int npa = 0;
                float psA[M::NSENSA][8];
                sfor<NSENS>([&](auto K_) MI_LAMBDA {
                    if constexpr (role_of_body_h(M::sens_body[K_]) == R && HB::pair_sensor_body(M::sens_body[K_])) {
                        sfor<8>([&](auto I) MI_LAMBDA { psA[K_][I] = 0.f; });
                    }
                });
                sfor<NHP>([&](auto P_) MI_LAMBDA {
                    constexpr int pp = P_, ba = M::hp_ba[pp], bb = M::hp_bb[pp];
                    constexpr bool mine_a = role_of_body_h(ba) == R, mine_b = role_of_body_h(bb) == R;
                    if constexpr (mine_a || mine_b) {
                        MI_PHASE();
                        float b0[3], b1[3], n[3], pc[3], pen;
                        sfor<3>([&](auto K) MI_LAMBDA { b0[K] = rows(X_PE + 6 * hp_slot(bb) + K); b1[K] = rows(X_PE + 6 * hp_slot(bb) + 3 + K); });
                        if constexpr (M::hp_box[pp]) {
                            const BodyTmp& ta = tb[MW::tslot(ba)];
                            pen = HB::template pair_box<pp>(ta.Rb, ta.rb, b0, b1, n, pc);
                        } else {
                            float a0[3], a1[3];
                            sfor<3>([&](auto K) MI_LAMBDA { a0[K] = rows(X_PE + 6 * hp_slot(ba) + K); a1[K] = rows(X_PE + 6 * hp_slot(ba) + 3 + K); });
                            pen = HB::template pair_capsules<pp>(a0, a1, b0, b1, n, pc);
                        }
                        const bool on = pen > 0.f;
                        if (MI_WAVE_ANY(on)) {
                            const float pe = on ? pen : 0.f;
                            if constexpr (mine_a) {
                                this->template pair_side<ba>(pc, n, pe, h, S, L, y); npa += on ? 1 : 0;
                                if constexpr (PSENS && HB::pair_sensor_body(ba)) { if (this->pair_sens) this->pair_sensor_acc(pc, n, pe, h, psA[HB::sensor_of(ba)]); }
                            }
                            if constexpr (mine_b) {
                                const float nm[3] = {-n[0], -n[1], -n[2]};
                                this->template pair_side<bb>(pc, nm, pe, h, S, L, y); npa += on ? 1 : 0;
                                if constexpr (PSENS && HB::pair_sensor_body(bb)) { if (this->pair_sens) this->pair_sensor_acc(pc, nm, pe, h, psA[HB::sensor_of(bb)]); }
                            }
                        }
                    }
                });
                this->pair_active = npa;
                // the six numbers of A wait in the fingertip's own slots of the `sensor` tensor (this lane rewrites them in the output phase; a load after
                // the own store of the same address sees it): carried in registers through the contact phases and the sweeps they cost 4 % of the step
                // (ShadowHand@16384 0.1857 -> 0.1933 ms, profiles/r6_hand_pair_sensors_ab.txt); P and n stay
                if constexpr (PSENS) if (this->pair_sens) sfor<NSENS>([&](auto K_) MI_LAMBDA {
                    if constexpr (role_of_body_h(M::sens_body[K_]) == R && HB::pair_sensor_body(M::sens_body[K_])) {
                        psPN[K_][0] = psA[K_][6]; psPN[K_][1] = psA[K_][7];
                        this->template pair_sensor_g<M::sens_body[K_]>(psA[K_], S, psG[K_]);
                        if (MI_WAVE_ANY(psA[K_][6] > 0.f)) sfor<6>([&](auto C) MI_LAMBDA { sensor(6 * K_ + C) = psA[K_][C]; });
                    }
                });
            }
        }
        sfor_rev<NV>([&](auto K_) MI_LAMBDA { if constexpr (MW::role_of_gi(K_) == R) factor(K_); });
        MI_PHASE();
        sfor_rev<NV>([&](auto I_) MI_LAMBDA { if constexpr (MW::role_of_gi(I_) == R) whiten(I_); });
        sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::trunk_entry(E_)) { constexpr int o = X_DT + R * NTE + MW::teidx(E_); rows(o) = L[E_]; } });
        sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int o = X_DY + R * NVT + MW::tidx(I); rows(o) = y[I]; } });
        MI_STAMP(1);
        bar();                                                                                       // ---- B1
        MI_STAMP(2);
        // ============================================================ P2 (every role, redundantly): trunk coming up, wrist factor
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
        if (this->actor_scale.p != nullptr) {       // the trunk's own H entries and bias forces, just written by body_up
            sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::trunk_entry(E_)) L[E_] *= sc_mass; });
            sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) c.bias[I] *= sc_mass; });
        }
        sfor<ND>([&](auto D) MI_LAMBDA { if constexpr (MW::trunk_gi(D)) drive(D); });
        sfor<NR>([&](auto R_) MI_LAMBDA {     // fixed order of the roles: the sum does not depend on which wave got here first
            sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::trunk_entry(E_)) { constexpr int o = X_DT + R_ * NTE + MW::teidx(E_); L[E_] += rows(o); } });
            sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int o = X_DY + R_ * NVT + MW::tidx(I); y[I] += rows(o); } });
        });
        sfor_rev<NV>([&](auto K_) MI_LAMBDA { if constexpr (MW::trunk_gi(K_)) factor(K_); });
        sfor_rev<NV>([&](auto I_) MI_LAMBDA { if constexpr (MW::trunk_gi(I_)) whiten(I_); });
        MI_STAMP(3);
        bar();                                                                                       // ---- B1b: region A is dead
        MI_STAMP(4);
        const bool clamp_on = HB::any_clamped() && this->drive_clamp != 0;
        // ============================================================ P3: own joint-limit rows (registers)
        float dw[NVT > 0 ? NVT : 1];                      // this role's warm-start contribution to the wrist part of w
        sfor<NVT>([&](auto I) MI_LAMBDA { dw[I] = 0.f; });
        float act = 0.f;
        auto wadd = [&](auto GI, const float val) MI_LAMBDA {
            constexpr int gi = decltype(GI)::value;
            if constexpr (MW::trunk_gi(gi)) { constexpr int ti = MW::tidx(gi); dw[ti] += val; } else w[gi] += val;
        };
        constexpr int NOWNL = nownl<R>();
        LimReg lr[NOWNL > 0 ? NOWNL : 1];
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = d;
            if constexpr (M::dof_limited[d] && MW::template owns_gi<R>(gi)) {
                constexpr int li = own_lim_idx<R>(d);
                LimReg& Rw = lr[li];
                MI_PHASE();
                const float dl = q[d] - (M::dof_lower[d] + limit_shift(d)), du = (M::dof_upper[d] + limit_shift(ND + d)) - q[d];
                const bool lower = dl < du;
                const float C = lower ? dl : du, s = lower ? 1.f : -1.f;
                const float lw = laml(d);
                const float l0 = ((lw * s < 0.f) ? 0.f : fabsf(lw)) * P.warm;
                float (&g)[M::MAXCHAIN] = Rw.g;
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
                float al = P.cfm, at = 0.f;      // diagonal of the row, limb and wrist part apart (the sweeps weight the wrist part)
                sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA {
                    constexpr int k = K, i = (k == 0) ? gi : M::anc[gi][k == 0 ? 0 : k - 1];
                    if constexpr (MW::trunk_gi(i)) at += g[k] * g[k]; else al += g[k] * g[k];
                });
                Rw.al = al; Rw.at = at;
                Rw.vt = (C >= 0.f) ? -C * invh : fminf(-C * P.erp * invh, P.max_depen_vel);
                Rw.lam = l0;
                Rw.rho = 0.f;
                act = ((l0 > 0.f) || (Rw.vt > 0.f)) ? 1.f : act;
                if constexpr (HB::clamped(d)) {      // a drive that starts the sub-step beyond its force range makes the block active in the first sweep
                    const float kp_ = M::dof_kp[d] * sc_kp, c_ = M::dof_damping[d] * sc_damp + h * kp_;
                    act = (clamp_on && fabsf(-kp_ * (q[d] - target[d]) - c_ * qd[d]) > M::dof_force_limit[d]) ? 1.f : act;
                }
                wadd(std::integral_constant<int, gi>{}, g[0] * l0);
                sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { wadd(std::integral_constant<int, M::anc[gi][A_]>{}, g[1 + A_] * l0); });
            }
        });
        MI_PHASE();
        MI_STAMP(5);
        // ============================================================ P3: rows of the own contacts (geometry in the slots since P1)
        float hasc = 0.f;
        sfor<NLIMB>([&](auto L_) MI_LAMBDA {
            constexpr int l = L_;
            if constexpr (limb_role(l) == R) {
                constexpr int RL = rl(l), GS0 = gslot(l), RB0 = rbase(l);
                sfor<NB>([&](auto B_) MI_LAMBDA {
                    constexpr int b = B_;
                    if constexpr (M::limb_of_body[b] == l && B::os_count(b) > 0) {
                        constexpr int CL = M::chain_len[b];
                        MI_PHASE();
                        constexpr int OS_SLOT = B::os_slot(b);
                        const int first = (int)(nw.bfc[OS_SLOT] & 255u), nbody = (int)(nw.bfc[OS_SLOT] >> 8);
                        // walked by contact (at most BODY_CAP, left by the whole wave as soon as no env has an (i+1)-th one on this body)
                        for (int i = 0; i < BODY_CAP; ++i) {
                            if (!MI_WAVE_ANY(i < nbody)) break;
                            if (i < nbody) {
                                const float* gp = rows.ptr(G0 + (GS0 + first + i) * GSZ);
                                float* rp_ = rows.ptr(RB0 + (first + i) * 3 * RL);
                                float fr[3][3], pc[3];
                                sfor<3>([&](auto I_) MI_LAMBDA { fr[0][I_] = gp[I_ * ST]; pc[I_] = gp[(3 + I_) * ST] + xo[I_]; });
                                contact_frame(fr[0], fr[1], fr[2]);
                                sfor<3>([&](auto K) MI_LAMBDA {
                                    constexpr int k = K;
                                    float W[6];
                                    cross3(pc, fr[k], W);
                                    W[3] = fr[k][0]; W[4] = fr[k][1]; W[5] = fr[k][2];
                                    float g[M::MAXCHAIN];
                                    sfor<CL>([&](auto C) MI_LAMBDA { g[C] = dot6(S[M::chain[b][C]], W); });
                                    // chain solve (descending indices; the later entries of a chain are exactly the ancestors)
                                    sfor<CL>([&](auto C) MI_LAMBDA {
                                        constexpr int kk0 = C, ii = M::chain[b][kk0];
                                        const float z = g[kk0] * Ldi[ii];
                                        g[kk0] = z;
                                        sfor<CL - 1 - kk0>([&](auto T) MI_LAMBDA {
                                            constexpr int kk = kk0 + 1 + T, jj = M::chain[b][kk];
                                            g[kk] -= L[M::midx[ii][jj]] * z;
                                        });
                                    });
                                    sfor<CL>([&](auto C) MI_LAMBDA { constexpr int idx = shape_idx(l, M::chain[b][C]); rp_[(k * RL + idx) * ST] = g[C]; });
                                    sfor<RL>([&](auto I_) MI_LAMBDA { if constexpr (!in_chain(b, l, I_)) rp_[(k * RL + I_) * ST] = 0.f; });   // the rest of the fixed shape
                                });
                            }
                        }
                    }
                });
                hasc = (nw.cnt[l] > 0) ? 1.f : hasc;
                rows(X_CNT + l) = __builtin_bit_cast(float, nw.cnt[l] | (nw.refused[l] << 16));
            }
        });
        act = (hasc > 0.f) ? 1.f : act;
        sfor<NVT>([&](auto I) MI_LAMBDA { rows(X_DW + R * NSH + I) = dw[I]; });
        rows(X_FLG + R) = act;
        rows(X_HASC + R) = hasc;
        MI_STAMP(6);
        bar();                                                                                       // ---- B2
        MI_STAMP(7);
        // ============================================================ P4: block sweeps
        // wrist part of w: every role adds all roles' warm-start contributions, in role order (object contacts are not warm started)
        sfor<NV>([&](auto I) MI_LAMBDA {
            constexpr int i = I;
            if constexpr (MW::trunk_gi(i)) { sfor<NR>([&](auto R_) MI_LAMBDA { constexpr int o = X_DW + R_ * NSH + MW::tidx(i); w[i] += rows(o); }); }
        });
        float hc[NR];
        sfor<NR>([&](auto R_) MI_LAMBDA { hc[R_] = rows(X_HASC + R_); });
        {
            float wtl[NVT > 0 ? NVT : 1], wol[6];           // the shared coordinates as this block sees them during a sweep
            for (int it = 0; it < P.iters; ++it) {
                int zero;
                MI_OPAQUE_ZERO(zero);
                const RowStore<RS> rit = rows.shifted(zero);
                const int par = it & 1;
                const RowStore<RS> xout = rows.shifted((X_DW + (par ^ 1) * NR * NSH) * ST);
                const RowStore<RS> fin = rows.shifted((X_FLG + par * NR) * ST), fout = rows.shifted((X_FLG + (par ^ 1) * NR) * ST);
                float nW = 0.f, nO = 0.f;
                sfor<NR>([&](auto R_) MI_LAMBDA { const float f = fin(R_); nW += f; nO += f * hc[R_]; });
                const float omW = (nW > 1.5f) ? 0.5f * (nW + 1.f) : 1.f, omO = (nO > 1.5f) ? 0.5f * (nO + 1.f) : 1.f;
                sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int ti = MW::tidx(I); wtl[ti] = w[I]; } });
                sfor<6>([&](auto K) MI_LAMBDA { wol[K] = wo[K]; });
                auto wget = [&](auto GI) MI_LAMBDA -> float {
                    constexpr int gi = decltype(GI)::value;
                    if constexpr (MW::trunk_gi(gi)) { constexpr int ti = MW::tidx(gi); return wtl[ti]; } else return w[gi];
                };
                auto wupd = [&](auto GI, const float val) MI_LAMBDA {
                    constexpr int gi = decltype(GI)::value;
                    if constexpr (MW::trunk_gi(gi)) { constexpr int ti = MW::tidx(gi); wtl[ti] += omW * val; } else w[gi] += val;
                };
                float actn = 0.f;
                sfor<NLIMB>([&](auto L_) MI_LAMBDA {
                    constexpr int l = L_;
                    if constexpr (limb_role(l) == R) {
                        // ---- the limit rows of the limb's dofs
                        sfor<ND>([&](auto D) MI_LAMBDA {
                            constexpr int d = D, gi = d;
                            if constexpr (M::dof_limited[d] && limb_of_gi(d) == l) {
                                constexpr int li = own_lim_idx<R>(d);
                                LimReg& Rw = lr[li];
                                float vn = Rw.g[0] * wget(std::integral_constant<int, gi>{});
                                sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { vn += Rw.g[1 + A_] * wget(std::integral_constant<int, M::anc[gi][A_]>{}); });
                                if constexpr (HB::clamped(d)) {
                                    if (clamp_on) {      // the dof's drive clamp, ahead of its limit row: g = s L^-1 e_d, v_d = s vn
                                        const float kp_ = M::dof_kp[d] * sc_kp, c_ = M::dof_damping[d] * sc_damp + h * kp_;
                                        // the closed form takes the dof's TRUE response g . g, not the block's wrist-weighted one: the weighted one
                                        // (larger) would over-relax this row -- its slope 1 / h - c a shrinks with a, unlike a limit row's -- and the
                                        // block order then oscillates on the wrist's clamps (oracle/physics.c solve_blocks)
                                        const float sg = (Rw.g[0] > 0.f) ? 1.f : -1.f, a_true = (Rw.al - P.cfm) + Rw.at, a_loc = (Rw.al - P.cfm) + omW * Rw.at;
                                        const float dr = sg * drive_clamp_update(-kp_ * (q[d] - target[d]), c_, M::dof_force_limit[d], invh, sg * vn,
                                                                                 (Rw.lam > 0.f) ? 0.f : a_true, Rw.rho);
                                        wupd(std::integral_constant<int, gi>{}, Rw.g[0] * dr);
                                        sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { wupd(std::integral_constant<int, M::anc[gi][A_]>{}, Rw.g[1 + A_] * dr); });
                                        vn += a_loc * dr;
                                        actn = (Rw.rho != 0.f) ? 1.f : actn;
                                    }
                                }
                                const float lo = Rw.lam;
                                const float nl_ = fmaxf(lo - (vn - Rw.vt) * MI_RCP(Rw.al + omW * Rw.at), 0.f);
                                const float dl = nl_ - lo;
                                Rw.lam = nl_;
                                actn = (nl_ > 0.f) ? 1.f : actn;
                                wupd(std::integral_constant<int, gi>{}, Rw.g[0] * dl);
                                sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { wupd(std::integral_constant<int, M::anc[gi][A_]>{}, Rw.g[1 + A_] * dl); });
                            }
                        });
                        // ---- the limb's contacts: a lane's j-th contact, whichever body it is on (fixed row shape [limb | wrist])
                        constexpr int NL = nl(l), RL = rl(l), GS0 = gslot(l), RB0 = rbase(l), KCAP = kcap(l), D0 = ldof0(l);
                        const int cnt = nw.cnt[l];
                        for (int j = 0; j < KCAP; ++j) {
                            const bool onj = j < cnt;
                            if (!MI_WAVE_ANY(onj)) break;
                            if (onj) {
                                float* gp = rit.ptr(G0 + (GS0 + j) * GSZ);
                                const float* rp_ = rit.ptr(RB0 + j * 3 * RL);
                                float g[3][RL], go[3][6], lm[3];
                                sfor<3>([&](auto K) MI_LAMBDA {
                                    sfor<RL>([&](auto C) MI_LAMBDA { g[K][C] = rp_[(K * RL + C) * ST]; });
                                    lm[K] = gp[(G_LAM + K) * ST];
                                });
                                {   // object part of the three rows from the stored normal and lever (unscaled: the whitening scales go in below)
                                    float fr[3][3], rc[3];
                                    sfor<3>([&](auto I_) MI_LAMBDA { fr[0][I_] = gp[I_ * ST]; rc[I_] = gp[(3 + I_) * ST]; });
                                    contact_frame(fr[0], fr[1], fr[2]);
                                    sfor<3>([&](auto K) MI_LAMBDA {
                                        float cx[3];
                                        cross3(rc, fr[K], cx);
                                        if constexpr (SHAPE != OBJ_BOX) { float cw[3]; HB::body_diag(Ro, isqI, cx, cw); sfor<3>([&](auto I_) MI_LAMBDA { cx[I_] = cw[I_]; }); }
                                        sfor<3>([&](auto I_) MI_LAMBDA { go[K][I_] = -fr[K][I_]; go[K][3 + I_] = -cx[I_]; });
                                    });
                                }
                                const float vtn = gp[G_VT * ST];
                                float ainv[3];
                                sfor<3>([&](auto K) MI_LAMBDA {
                                    float a = P.cfm, at = 0.f;
                                    sfor<NL>([&](auto C) MI_LAMBDA { a += g[K][C] * g[K][C]; });
                                    sfor<NVT>([&](auto T_) MI_LAMBDA { at += g[K][NL + T_] * g[K][NL + T_]; });
                                    const float aol = go[K][0] * go[K][0] + go[K][1] * go[K][1] + go[K][2] * go[K][2];
                                    const float aoa = go[K][3] * go[K][3] + go[K][4] * go[K][4] + go[K][5] * go[K][5];
                                    ainv[K] = MI_RCP(a + omW * at + omO * ((ism * ism) * aol + (isi * isi) * aoa));
                                });
                                auto rowvel = [&](int k) MI_LAMBDA {
                                    float vn = 0.f;
                                    sfor<NL>([&](auto C) MI_LAMBDA { vn += g[k][C] * w[D0 + C]; });
                                    sfor<NVT>([&](auto T_) MI_LAMBDA { vn += g[k][NL + T_] * wtl[T_]; });
                                    float vl = 0.f, va = 0.f;
                                    sfor<3>([&](auto C) MI_LAMBDA { vl += go[k][C] * wol[C]; va += go[k][3 + C] * wol[3 + C]; });
                                    return vn + (ism * vl + isi * va);
                                };
                                auto apply = [&](int k, float dl) MI_LAMBDA {
                                    sfor<NL>([&](auto C) MI_LAMBDA { w[D0 + C] += g[k][C] * dl; });
                                    const float dlw = omW * dl;
                                    sfor<NVT>([&](auto T_) MI_LAMBDA { wtl[T_] += g[k][NL + T_] * dlw; });
                                    const float dll = (omO * ism) * dl, dla = (omO * isi) * dl;
                                    sfor<3>([&](auto C) MI_LAMBDA { wol[C] += go[k][C] * dll; wol[3 + C] += go[k][3 + C] * dla; });
                                };
                                const float ln = fmaxf(lm[0] - (rowvel(0) - vtn) * ainv[0], 0.f);
                                apply(0, ln - lm[0]);
                                float lt[2];
                                // both tangent rows from the SAME velocity, the disc projection, ONE application (core/hand_engine.hpp)
                                float vtg[2];
                                sfor<2>([&](auto K) MI_LAMBDA { vtg[K] = rowvel(1 + K); lt[K] = lm[1 + K] - vtg[K] * ainv[1 + K]; });
                                friction_disc(lt, lm[1], lm[2], vtg[0], vtg[1], ainv[1], ainv[2], OP.mu * ln);
                                gp[G_LAM * ST] = ln;
                                actn = (ln > 0.f) ? 1.f : actn;
                                sfor<2>([&](auto K) MI_LAMBDA { gp[(G_LAM + 1 + K) * ST] = lt[K]; });
                                sfor<2>([&](auto K) MI_LAMBDA { apply(1 + K, lt[K] - lm[1 + K]); });
                            }
                        }
                    }
                });
                // this block's true contribution to the shared coordinates, its activity in the next sweep; then everybody's, in role order
                {
                    const float iomW = 1.f / omW, iomO = 1.f / omO;
                    sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int ti = MW::tidx(I); xout(R * NSH + ti) = (wtl[ti] - w[I]) * iomW; } });
                    sfor<6>([&](auto K) MI_LAMBDA { xout(R * NSH + NVT + K) = (wol[K] - wo[K]) * iomO; });
                }
                fout(R) = actn;
                bar();                                                                               // ---- one barrier per sweep
                sfor<NV>([&](auto I) MI_LAMBDA {
                    constexpr int i = I;
                    if constexpr (MW::trunk_gi(i)) { sfor<NR>([&](auto R_) MI_LAMBDA { constexpr int o = R_ * NSH + MW::tidx(i); w[i] += xout(o); }); }
                });
                sfor<6>([&](auto K) MI_LAMBDA { sfor<NR>([&](auto R_) MI_LAMBDA { constexpr int o = R_ * NSH + NVT + K; wo[K] += xout(o); }); });
            }
        }
        MI_STAMP(8);
        // ============================================================ P5: back to generalised velocity, outputs, integration
        sfor<NV>([&](auto I_) MI_LAMBDA {       // ascending: ancestors (wrist or own limb) first
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
            if constexpr (MW::template owns_gi<R>(d)) {
                float ll = 0.f;
                if constexpr (M::dof_limited[d]) {
                    constexpr int li = own_lim_idx<R>(d);
                    // which limit the row was built for: g[0] = s / L_dd, s = +1 lower, -1 upper
                    ll = (lr[li].g[0] > 0.f) ? lr[li].lam : -lr[li].lam;
                }
                laml(d) = ll;
                float df = -M::dof_kp[d] * sc_kp * (q[d] - target[d]) - M::dof_damping[d] * sc_damp * v[d] + ll * invh;
                if constexpr (HB::clamped(d)) {      // a force-limited drive reports the end-of-step force the clamp acts on: fa - c v + rho / h
                    if (clamp_on) df += lr[own_lim_idx<R>(d)].rho * invh - h * M::dof_kp[d] * sc_kp * v[d];
                }
                dof_force(d) = df;
            }
        });
        sfor<NSENS>([&](auto K_) MI_LAMBDA {
            constexpr int k = K_, b = M::sens_body[k], l = M::limb_of_body[b];
            if constexpr (limb_role(l) == R) {
                constexpr int GS0 = gslot(l);
                float sens[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if constexpr (B::os_count(b) > 0) {
                    const float (&Rb)[9] = c.Rs[k];          // the fingertip's frame, kept by the tree pass (force-sensor body)
                    const float (&rb)[3] = c.rs[k];
                    constexpr int OS_SLOT = B::os_slot(b);
                    const int first = (int)(nw.bfc[OS_SLOT] & 255u), nb_ = (int)(nw.bfc[OS_SLOT] >> 8);
                    for (int i = 0; i < BODY_CAP; ++i) {
                        if (!MI_WAVE_ANY(i < nb_)) break;
                        if (i < nb_) {
                            const float* gp = rows.ptr(G0 + (GS0 + first + i) * GSZ);
                            const float ln = gp[G_LAM * ST], l1 = gp[(G_LAM + 1) * ST], l2 = gp[(G_LAM + 2) * ST];
                            float n[3], t1[3], t2[3], rc[3];
                            sfor<3>([&](auto K) MI_LAMBDA { n[K] = gp[K * ST]; rc[K] = gp[(3 + K) * ST]; });
                            contact_frame(n, t1, t2);
                            float f[3], arm[3], tq[3], fl[3], tl[3];
                            sfor<3>([&](auto K) MI_LAMBDA {
                                f[K] = (n[K] * ln + t1[K] * l1 + t2[K] * l2) * invh;
                                arm[K] = (rc[K] + xo[K]) - rb[K];
                            });
                            cross3(arm, f, tq);
                            matTvec3(Rb, f, fl); matTvec3(Rb, tq, tl);
                            sfor<3>([&](auto C) MI_LAMBDA { sens[C] += fl[C]; sens[3 + C] += tl[C]; });
                        }
                    }
                }
                if constexpr (NHP > 0 && PSENS && HB::pair_sensor_body(b)) {
                    if (MI_WAVE_ANY(psPN[k][0] > 0.f)) {       // the hand's own contacts on this fingertip (core/hand_engine.hpp pair_sensor_wrench)
                        const float (&Rb)[9] = c.Rs[k];
                        const float (&rb)[3] = c.rs[k];
                        float wr[6], tq[3], fl[3], tl[3], A8[8];
                        const bool any = psPN[k][0] > 0.f;
                        sfor<6>([&](auto C) MI_LAMBDA { const float a_ = sensor(6 * k + C); A8[C] = any ? a_ : 0.f; });       // (parked there by the pair phase)
                        A8[6] = psPN[k][0]; A8[7] = psPN[k][1];
                        this->template pair_sensor_wrench_g<b>(A8, psG[k], h, v, wr);
                        const float f[3] = {wr[3], wr[4], wr[5]};
                        cross3(rb, f, tq);
                        sfor<3>([&](auto C) MI_LAMBDA { tq[C] = wr[C] - tq[C]; });
                        matTvec3(Rb, f, fl); matTvec3(Rb, tq, tl);
                        sfor<3>([&](auto C) MI_LAMBDA { sens[C] += fl[C]; sens[3 + C] += tl[C]; });
                    }
                }
                sfor<6>([&](auto C) MI_LAMBDA { sensor(6 * k + C) = sens[C]; });
            }
        });
        MI_PHASE();
        sfor<ND>([&](auto D) MI_LAMBDA {
            if constexpr (MW::template owns_gi<R>(D)) { qd[D] = v[D]; q[D] += h * qd[D]; }
        });
        if constexpr (R == M::TRUNK_ROLE) {
            // contact counters (written by their owners before B2)
            int tot = 0, ref = 0;
            sfor<NLIMB>([&](auto L_) MI_LAMBDA { const int cw = __builtin_bit_cast(int, (float)rows(X_CNT + L_)); tot += cw & 0xFFFF; ref += cw >> 16; });
            *ncontact = tot | (ref << 16);
            // the object (semi-implicit Euler)
            sfor<3>([&](auto K) MI_LAMBDA {
                obj.vel[K] = wo[K] * ism; obj.angvel[K] = wo[3 + K] * isi;
                obj.pos[K] += h * obj.vel[K];
            });
            if constexpr (SHAPE != OBJ_BOX) { float om[3]; HB::body_diag(Ro, isqI, wo + 3, om); sfor<3>([&](auto K) MI_LAMBDA { obj.angvel[K] = om[K]; }); }
            {
                const float* om = obj.angvel;
                const float an = MI_SQRT(dot3(om, om)), th = an * h;
                float sn, cs;
                sincosf(0.5f * th, &sn, &cs);
                const bool big = th > 1e-12f;
                const float k = big ? sn * MI_RCP(fmaxf(an, 1e-30f)) : 0.5f * h;
                const float dq[4] = {om[0] * k, om[1] * k, om[2] * k, big ? cs : 1.f};
                float* Q = obj.quat;
                const float x = dq[3] * Q[0] + dq[0] * Q[3] + dq[1] * Q[2] - dq[2] * Q[1];
                const float yy = dq[3] * Q[1] - dq[0] * Q[2] + dq[1] * Q[3] + dq[2] * Q[0];
                const float z = dq[3] * Q[2] + dq[0] * Q[1] - dq[1] * Q[0] + dq[2] * Q[3];
                const float ww = dq[3] * Q[3] - dq[0] * Q[0] - dq[1] * Q[1] - dq[2] * Q[2];
                const float n = MI_RSQ(x * x + yy * yy + z * z + ww * ww);
                Q[0] = x * n; Q[1] = yy * n; Q[2] = z * n; Q[3] = ww * n;
            }
        }
        MI_STAMP(9);
    }
};

}  // namespace mi
