// engine_mwc.hpp -- the limb-per-wave sub-step for robots on the COMPACT contact store (Humanoid: 126 potential rows, self-collision).
//
// Round 2 ran this robot on one main wave (plus a helper for the self-collision phase): 49 % of the sub-step was a serial tail on that
// wave (warm start, Gauss-Seidel sweeps, write-back), it executed the UNION of its 32 envs' active contact sets, and its live state (the
// whole factor L, all joint axes, 105 sphere coordinates) overflowed the register file into 1 KB of scratch per lane.  Here the limb
// roles of the workgroup (engine_mw.hpp; Humanoid: right leg | left leg | trunk rows + both arms) each own their limbs' tree pass,
// factor, constraint rows, SWEEPS and outputs, and the LAST wave owns no limb: it is the pair role -- positions-only forward kinematics
// and the self-collision narrow phase while the others work through P1 .. P3, then the sweeps of the self-contact rows:
//
//   P1  limb roles  trunk down, own limbs down + up, limb factor + whitened velocity; Schur complement / carries -> LDS     | B1
//       pair role   forward kinematics (positions), first groups of the narrow phase
//   P2  limb roles  trunk up with every limb's contribution, trunk factor (redundantly)
//   P3  limb roles  own joint-limit rows; own ground contacts into the role's OWN contact slots (caps M::wave_kcap, rows in a FIXED
//                   shape [limb dofs | trunk dofs] shared by every body of the role, so the sweeps walk a lane's actual contacts in a
//                   run-time loop instead of the union of the wave's spheres)
//       pair role   the other groups, <= KPAIR self contacts (point, normal, bodies) -> C_X                                  | B2
//   Z   pair role   zeroes the dense rows [3][NV] of the contacts, stores their velocity targets and warm-start impulses      | B3
//   S   limb roles  for every self contact with a body of mine: g = +-L^-T J^T over that body's chain, ADDED to the dense rows
//                   (ds_add_f32: at most two waves add to an entry of a zeroed row -- commutative, so bit-reproducible)      | B4
//   W   limb roles  warm start of the self-contact rows on the own limb coordinates (and, redundantly, the trunk's)            | B5
//   P4  all         block sweeps (oracle/physics.c solve_blocks): every limb role Gauss-Seidel over its own rows, the pair role over the
//                   self contacts; coordinates shared by n >= 2 active blocks (the trunk: all; a limb: its owner + the pair block)
//                   answer with weight (n + 1) / 2; true contributions exchanged through LDS (double buffered)                | 1 per sweep
//   P5  limb roles  velocities, impulses / sensors / joint forces of the own rows, integration; pair role: impulses of the groups
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

// float add into the shared row store: ds_add_f32 on the device (no return value); a CAS loop in the host build of the tests
MI_HD void lds_add(float* p, const float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    do { nw = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, old) + v); } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
#endif
}

template <class M>
struct SimMWC : SimMW<M> {
    using MW = SimMW<M>;
    using B = Sim<M>;
    using typename B::Ctx;
    using typename B::BodyTmp;
    static constexpr int NB = M::NB, ND = M::ND, NV = M::NV, OFF = M::OFF, NSPH = M::NSPH, NSENS = M::NSENS, NLIM = B::NLIM, NVA = B::NVA,
                         NR = M::NROLE, NVT = MW::NVT, NTE = MW::NTE, NLR = MW::NLR, NTB = MW::NTB, NPG = M::NPG, KPAIR = 3, NLIMB = M::NLIMB;
    static_assert(!M::FIXED && B::COMPACT, "free base, compact-store model");
    // ---- roles.  A limb role owns the limbs dealt to it (consecutive generalised indices lfirst(r) .. lfirst(r) + nl(r) - 1); a model
    //      with self-collision tables keeps its last role free of limbs: the pair role.
    static constexpr int lfirst(int r) { for (int i = OFF; i < NV; ++i) if (MW::role_of_gi(i) == r) return i; return NV; }
    static constexpr int nl(int r) { int n = 0; for (int i = 0; i < NV; ++i) n += (MW::role_of_gi(i) == r) ? 1 : 0; return n; }
    static constexpr bool owns_any(int r) { for (int b = 0; b < NB; ++b) if (MW::role_of_body(b) == r || (MW::trunk_body(b) && r == M::TRUNK_ROLE)) return true; return false; }
    static constexpr bool HAS_PAIR_ROLE = NPG > 0;
    static constexpr int PAIR_ROLE = NR - 1;
    static_assert(!HAS_PAIR_ROLE || !owns_any(PAIR_ROLE), "self-colliding model: the last role owns no body (assets/model.py wave_roles)");
    static constexpr int NRL = HAS_PAIR_ROLE ? NR - 1 : NR;                               // limb roles
    static constexpr bool limbs_contiguous() {
        for (int r = 0; r < NR; ++r) for (int i = lfirst(r); i < lfirst(r) + nl(r); ++i) if (MW::role_of_gi(i) != r) return false;
        return true;
    }
    static_assert(limbs_contiguous(), "a role's limb dofs are numbered consecutively (depth-first numbering)");
    static constexpr int NVL = NV - NVT;                                                   // limb coordinates of all roles
    static constexpr int lidx(int gi) { int n = 0; for (int k = 0; k < gi; ++k) n += MW::trunk_gi(k) ? 0 : 1; return n; }   // place in an [NVL] vector
    static constexpr int limb_of_gi(int gi) { return gi < OFF ? 0 : M::limb_of_body[M::dof_body[gi - OFF]]; }               // coordinate group (0: trunk)
    static constexpr int rlen(int r) { return nl(r) + NVT; }                               // fixed row shape of role r: [limb | trunk]
    static constexpr int gcsz(int r) { return 3 * rlen(r) + 4; }                           // ground slot: 3 rows, vt, lam x3
    static constexpr int kcap(int r) { return M::wave_kcap[r]; }
    static constexpr bool uniform_mu() { for (int s = 1; s < NSPH; ++s) if (M::sph_mu[s] != M::sph_mu[0]) return false; return true; }
    static_assert(uniform_mu(), "the ground slots do not carry their sphere's friction: one value per model");
    // ---- LDS layout (floats per env)
    static constexpr int C_LIMG = B::C_LIMG;                                               // limit rows' G, packed by B::limoff
    static constexpr int L_VT = C_LIMG, L_LAM = C_LIMG + NLIM;                             // velocity target, impulse
    static constexpr int gcb(int r) { int o = C_LIMG + 2 * NLIM; for (int k = 0; k < r; ++k) o += kcap(k) * gcsz(k); return o; }
    static constexpr int C_SLOTOF = gcb(NR);                                               // one byte per sphere: its slot in the owner's range, -1
    static constexpr int C_X = C_SLOTOF + (NSPH + 3) / 4;                                  // one spare word (until round 6: the group -> slot map), then per self contact 8 floats:
    static constexpr int XI = 8;                                                           //   point (3), normal (3), bodies word, mu
    static constexpr int A0 = C_X + 1 + XI * KPAIR;                                        // shared region: before B2 ...
    static constexpr int X_LR = A0, X_DT = X_LR + 16 * NLR, X_DY = X_DT + NR * NTE, X_END = X_DY + NR * NVT;
    static constexpr int P_CSZ = 3 * NV + 4;                                               // ... after B2: self-contact rows, dense [3][NV]; vt, lam x3
    static constexpr int P_B = A0, XW0 = P_B + KPAIR * P_CSZ;                              // [NVT]        trunk part of w before any warm start (for the pair role)
    static constexpr int DOWN = XW0 + NVT;                                                 // [2][NVL]     true contribution of the owners' blocks to their limb coordinates
    static constexpr int DPAIR = DOWN + 2 * NVL;                                           // [2][NVL]     ... of the pair block
    static constexpr int X_DW = DPAIR + 2 * NVL;                                           // [2][NR][NVT] ... of every block to the trunk part
    static constexpr int X_FLG = X_DW + 2 * NR * NVT;                                      // [2][NR]      block active
    static constexpr int X_TOUCH = X_FLG + 2 * NR;                                         // [1]          limbs the self contacts touch (bit l - 1 for limb l)
    static constexpr int S_END = X_TOUCH + 1;
    static constexpr int MWC_SLOTS = X_END > S_END ? X_END : S_END;
    static constexpr int LANES = 32;
    static_assert((size_t)MWC_SLOTS * LANES * sizeof(float) <= 160 * 1024, "row store + exchange areas fit the LDS of a CU at 32 envs per workgroup");
    static constexpr int PVT = 3 * NV, PLAM = PVT + 1;                                     // inside a self-contact slot

    // ---- chain groups of a role.  The chain of a body is a SUFFIX of the chains of the bodies below it (chains list a body's coordinates from its
    //      own joint up to the root): a row over the chain of a group's deepest body (its leaf) is the row of ANY body of the group once the
    //      entries of the joints below that body are set to zero.  The self-contact rows are built once per (contact, group) that way instead of
    //      once per (contact, body): a lane's body is run-time data, the union of the wave's bodies used to be executed.
    static constexpr bool chain_suffix(int b, int lf) {
        if (M::chain_len[b] > M::chain_len[lf]) return false;
        for (int c = 0; c < M::chain_len[b]; ++c) if (M::chain[b][c] != M::chain[lf][c + M::chain_len[lf] - M::chain_len[b]]) return false;
        return true;
    }
    template <int R> static constexpr bool owned(int b) { return MW::role_of_body(b) == R || (MW::trunk_body(b) && R == M::TRUNK_ROLE); }
    template <int R> static constexpr bool chain_leaf(int b) {
        if (!owned<R>(b)) return false;
        for (int c = 0; c < NB; ++c) if (c != b && owned<R>(c) && M::chain_len[c] > M::chain_len[b] && chain_suffix(b, c)) return false;
        return true;
    }
    template <int R> static constexpr int chain_group(int b) {        // the leaf whose group body b belongs to (the first one that fits), -1: not this role's
        if (!owned<R>(b)) return -1;
        for (int c = 0; c < NB; ++c) if (chain_leaf<R>(c) && chain_suffix(b, c)) return c;
        return -1;
    }
    template <int R> static constexpr int shape_idx(int gi) { return MW::trunk_gi(gi) ? nl(R) + MW::tidx(gi) : gi - lfirst(R); }
    template <int R> static constexpr bool in_chain(int b, int idx) {
        for (int c = 0; c < M::chain_len[b]; ++c) if (shape_idx<R>(M::chain[b][c]) == idx) return true;
        return false;
    }
    // weight of a coordinate in a sweep: trunk omT, limb l oml[l - 1]
    template <int GI> static MI_HD float om_of(const float omT, const float (&oml)[NLIMB > 1 ? NLIMB - 1 : 1]) {
        if constexpr (MW::trunk_gi(GI)) return omT; else { constexpr int l = limb_of_gi(GI); return oml[l - 1]; }
    }
    // per-sweep weights from the blocks' activity flags and the limbs the self contacts touch
    template <int RS>
    MI_HD void sweep_weights(const RowStore<RS>& fin, const unsigned touch, float& omT, float (&oml)[NLIMB > 1 ? NLIMB - 1 : 1]) const {
        float fl[NR];
        sfor<NR>([&](auto B_) MI_LAMBDA { fl[B_] = fin(B_); });
        float nact = 0.f;
        sfor<NR>([&](auto B_) MI_LAMBDA { nact += fl[B_]; });
        omT = (nact > 1.5f) ? 0.5f * (nact + 1.f) : 1.f;
        sfor<NLIMB - 1>([&](auto L_) MI_LAMBDA {
            constexpr int l = L_ + 1, r = M::role_of_limb[l];
            const float fp = HAS_PAIR_ROLE ? (((touch >> L_) & 1u) ? fl[PAIR_ROLE] : 0.f) : 0.f;
            oml[L_] = (fl[r] + fp > 1.5f) ? 1.5f : 1.f;
        });
    }

    // ---------------------------------------------------------------- who examines a self-collision group
    // A group whose two limbs BOTH belong to the trunk role (the trunk itself, or a limb that role owns: the Humanoid's arms) is examined by THAT role,
    // from the sphere centres its own tree pass leaves in c.xcs, in the slack it has before B2; the pair role examines the others.  (Until round 6 the
    // pair role examined all of them and arrived last at B1 and at B2: every limb role waited ~60 units per sub-step for it, tools/debug/mwc_phases.py.)
    // The model's table lists the trunk role's groups LAST (assets/model.py self_collision_groups), and slots are dealt in table order: the pair role
    // fills slots 0 .. before B2, the trunk role continues behind them after B2 -- the order of every other form and of the oracle.
    static constexpr bool pg_trunk_role(int g) {
        if (NPG == 0) return false;
        const int la = M::limb_of_body[M::pg_tip_a[g]], lb = M::limb_of_body[M::pg_tip_b[g]];
        return (la == 0 || M::role_of_limb[la] == M::TRUNK_ROLE) && (lb == 0 || M::role_of_limb[lb] == M::TRUNK_ROLE);
    }
    static constexpr int NPGP = []() constexpr { int n = 0; for (int g = 0; g < (NPG > 0 ? NPG : 0); ++g) { if (pg_trunk_role(g)) break; ++n; } return n; }();
    static constexpr int NTG = (NPG > 0 ? NPG : 0) - NPGP;                                    // groups of the trunk role
    static constexpr bool pg_suffix() { for (int g = NPGP; g < (NPG > 0 ? NPG : 0); ++g) if (!pg_trunk_role(g)) return false; return true; }
    static_assert(pg_suffix(), "the trunk role's self-collision groups are the last ones of the table");
    // the deepest capsule pair of group g: distance (3e38: none was near), closest points, its index in the group
    template <int g>
    MI_HD static void pair_group_search(const SimParams& P, const float (*xa)[3], const float (*capm)[3], float& best, float (&bca)[3], float (&bcb)[3], int& bk) {
        best = 3.0e38f; bk = 0;
        sfor<3>([&](auto I_) MI_LAMBDA { bca[I_] = 0.f; bcb[I_] = 0.f; });
        sfor<M::pg_count[g]>([&](auto K_) MI_LAMBDA {
            constexpr int k = M::pg_first[g] + K_, ia = M::gp_a[k], ib = M::gp_b[k];
            constexpr float reach = B::cap_bound(ia) + B::cap_bound(ib);
            const float dm[3] = {capm[ia][0] - capm[ib][0], capm[ia][1] - capm[ib][1], capm[ia][2] - capm[ib][2]};
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
    }
    // what a contact slot says about pair bk of group g: bodies word (body a | body b << 8 | limb a << 16 | limb b << 20 | g << 24), radius of side b,
    // friction; and the normal from b towards a
    template <int g>
    MI_HD static void pair_group_pick(const int bk, const float (&bca)[3], const float (&bcb)[3], int& bab, float& brb, float& bmu, float (&n)[3]) {
        bab = 0; brb = 0.f; bmu = 0.f;
        sfor<M::pg_count[g]>([&](auto K_) MI_LAMBDA {
            constexpr int k = M::pg_first[g] + K_, ia = M::gp_a[k], ib = M::gp_b[k], ba = M::cap_body[ia], bb = M::cap_body[ib];
            constexpr int la = M::limb_of_body[ba], lb = M::limb_of_body[bb];     // 0: trunk
            constexpr int word = ba | (bb << 8) | (la << 16) | (lb << 20) | (g << 24);
            const bool me = bk == K_;
            bab = me ? word : bab;
            brb = me ? M::cap_rad[ib] : brb;
            bmu = me ? 0.5f * (M::cap_mu[ia] + M::cap_mu[ib]) : bmu;
        });
        const float dv[3] = {bca[0] - bcb[0], bca[1] - bcb[1], bca[2] - bcb[2]};
        const float d2 = dot3(dv, dv);
        const bool okd = d2 > 1e-18f;
        const float inv = MI_RSQ(fmaxf(d2, 1e-30f));
        n[0] = okd ? dv[0] * inv : 0.f; n[1] = okd ? dv[1] * inv : 0.f; n[2] = okd ? dv[2] * inv : 1.f;
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

    // ================================================================ the pair role: narrow phase, sweeps of the self contacts
    // groups 0 .. G_SPLIT - 1 are examined before B1 (while the limb roles run their tree pass), the others before B2
    static constexpr int G_SPLIT = []() constexpr {
        int tot = 0, acc = 0;
        for (int g = 0; g < NPGP; ++g) tot += M::pg_count[g];
        for (int g = 0; g < NPGP; ++g) { if (5 * (acc + M::pg_count[g]) > tot) return g; acc += M::pg_count[g]; }
        return NPGP;
    }();
    template <int RS, class BAR>
    MI_HD void substep_pair(const SimParams& P, const float h, const RowStore<RS> rows, const SelfCol* scol, const BAR& bar) {
        constexpr int ST = RowStore<RS>::stride;
        const float invh = MI_RCP(h);
        const bool selfcol = (NPG > 0) && (scol != nullptr);
        float pvt[KPAIR], pl0[KPAIR][3];       // velocity targets / warm-start impulses of the contacts, kept until the shared region is free
        sfor<KPAIR>([&](auto J_) MI_LAMBDA { pvt[J_] = 0.f; sfor<3>([&](auto K) MI_LAMBDA { pl0[J_][K] = 0.f; }); });
        int cntp = 0;                          // slots this role filled (its own groups); the trunk role continues behind them after B2
        unsigned prevm = 0u;                   // groups whose normal impulse in memory (last sub-step's) is not zero: bit g
#if defined(MI_TIMING)
        unsigned long long* const tstamp = this->tstamp;
#endif
        MI_STAMP(0);
        if constexpr (NPG > 0) {
            float xa[NSPH][3], capm[M::NCAP][3];
            int pdrop = 0;
            auto group = [&](auto G_) MI_LAMBDA {
                constexpr int g = decltype(G_)::value;
                MI_PHASE();
                float best, bca[3], bcb[3];
                int bk;
                pair_group_search<g>(P, xa, capm, best, bca, bcb, bk);
                const bool near = best < P.contact_offset;
                const bool on = near && (cntp < KPAIR);
                pdrop += (near && !on) ? 1 : 0;
                if (MI_WAVE_ANY(on)) {
                    if (on) {
                        int bab;
                        float brb, bmu, n[3];
                        pair_group_pick<g>(bk, bca, bcb, bab, brb, bmu, n);
                        float* xi = rows.ptr(C_X + 1 + XI * cntp);
                        sfor<3>([&](auto I_) MI_LAMBDA { xi[I_ * ST] = bcb[I_] + n[I_] * (brb + 0.5f * best); xi[(3 + I_) * ST] = n[I_]; });
                        xi[6 * ST] = __builtin_bit_cast(float, bab);
                        xi[7 * ST] = bmu;
                        // (velocity target and warm-start impulses go straight into the slot once the shared region is free: kept per slot)
                        const float gap = best - P.rest_offset;
                        sfor<KPAIR>([&](auto J_) MI_LAMBDA {
                            const bool mej = cntp == J_;
                            pvt[J_] = mej ? ((gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel)) : pvt[J_];
                            sfor<3>([&](auto K) MI_LAMBDA { pl0[J_][K] = mej ? scol->lamp(3 * g + K) * P.warm : pl0[J_][K]; });
                        });
                    }
                }
                cntp += on ? 1 : 0;
            };
            if (selfcol) {
                sfor<NPG>([&](auto G_) MI_LAMBDA { prevm |= (scol->lamp(3 * G_) != 0.f) ? (1u << G_) : 0u; });
                fk_pos<0>(nullptr, nullptr, xa);
                sfor<M::NCAP>([&](auto C_) MI_LAMBDA {
                    sfor<3>([&](auto I_) MI_LAMBDA { capm[C_][I_] = 0.5f * (xa[M::cap_s0[C_]][I_] + xa[M::cap_s1[C_]][I_]); });
                });
                sfor<KPAIR>([&](auto J_) MI_LAMBDA { rows(C_X + 1 + XI * J_ + 6) = __builtin_bit_cast(float, 0xFFFFFFFFu); });
                sfor<G_SPLIT>([&](auto G_) MI_LAMBDA { group(G_); });
            }
            MI_STAMP(1);
            bar();                                                                                   // ---- B1
            MI_STAMP(2);
            if (selfcol) {
                sfor<NPGP - G_SPLIT>([&](auto G_) MI_LAMBDA { group(std::integral_constant<int, G_SPLIT + decltype(G_)::value>{}); });
                if (scol->dropped != nullptr && pdrop > 0) MI_ATOMIC_ADD_INT(scol->dropped + scol->dstride, pdrop);
            }
            MI_STAMP(4);
        } else {
            bar();
        }
        bar();                                                                                       // ---- B2
        MI_STAMP(5);
        if constexpr (NPG > 0) { if (selfcol) {
            // Z: zero the dense rows of the contacts this role found; their velocity targets and warm-start impulses.  (Its OWN slots, by count: the
            // trunk role is filling the slots behind them right now -- with its own Z part -- and their bodies words must not be looked at before B3.)
            sfor<KPAIR>([&](auto J_) MI_LAMBDA {
                constexpr int j = J_;
                // the trunk entries, which every limb role ADDS to (a limb's entries are STORED by the role that owns the limb): of every slot, used or
                // not -- the trunk role does not have to know, and an unused slot's rows are never read
                sfor<3>([&](auto K) MI_LAMBDA { sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) rows(P_B + j * P_CSZ + K * NV + I) = 0.f; }); });
                const bool onj = j < cntp;
                if (onj) {
                    rows(P_B + j * P_CSZ + PVT) = pvt[j];
                    sfor<3>([&](auto K) MI_LAMBDA { rows(P_B + j * P_CSZ + PLAM + K) = pl0[j][K]; });
                }
            });
            MI_STAMP(6);
            bar();                                                                                   // ---- B3
            MI_STAMP(7);
            MI_STAMP(8);
            bar();                                                                                   // ---- B4: rows complete
            MI_STAMP(9);
        } }
        // ---- round 0 of the exchange: this block contributes nothing to it (the warm start of its rows is added by the limb roles)
        unsigned touch = 0u;
        float actp = 0.f;
        if constexpr (NPG > 0) { if (selfcol) {
            sfor<KPAIR>([&](auto J_) MI_LAMBDA {
                const unsigned bab = __builtin_bit_cast(unsigned, (float)rows(C_X + 1 + XI * J_ + 6));
                const bool onj = bab != 0xFFFFFFFFu;
                const unsigned la = (bab >> 16) & 15u, lb = (bab >> 20) & 15u;
                touch |= onj ? ((la ? 1u << (la - 1) : 0u) | (lb ? 1u << (lb - 1) : 0u)) : 0u;
                actp = onj ? 1.f : actp;
            });
        } }
        // warm start of the self-contact rows on the trunk part: computed HERE, from the complete rows, BEFORE B5 -- afterwards the sweeps overwrite the
        // impulses in the slots -- and published as this block's round-0 contribution to the trunk part, which every role adds last (block order).
        // (Until round 6 every limb role recomputed it from the rows: 81 LDS loads and 70 units of the limb roles' critical path per sub-step.)
        float dwp[NVT];
        sfor<NVT>([&](auto T_) MI_LAMBDA { dwp[T_] = 0.f; });
        if constexpr (NPG > 0) { if (selfcol) {
            sfor<KPAIR>([&](auto J_) MI_LAMBDA {
                constexpr int j = J_;
                const unsigned bab = __builtin_bit_cast(unsigned, (float)rows(C_X + 1 + XI * j + 6));
                if (bab != 0xFFFFFFFFu) {
                    sfor<3>([&](auto K) MI_LAMBDA {
                        const float l0 = rows(P_B + j * P_CSZ + PLAM + K);
                        sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int ti = MW::tidx(I); dwp[ti] += rows(P_B + j * P_CSZ + K * NV + I) * l0; } });
                    });
                }
            });
        } }
        sfor<NVT>([&](auto I) MI_LAMBDA { rows(X_DW + PAIR_ROLE * NVT + I) = dwp[I]; });
        rows(X_FLG + PAIR_ROLE) = actp;
        rows(X_TOUCH) = __builtin_bit_cast(float, touch);
        MI_STAMP(10);
        bar();                                                                                       // ---- B5
        MI_STAMP(11);
        // ============================================================ P4: the self contacts' block
        float wt[NVT], pw[NVL > 0 ? NVL : 1];
        sfor<NVT>([&](auto T_) MI_LAMBDA {
            wt[T_] = rows(XW0 + T_);
            sfor<NR>([&](auto B_) MI_LAMBDA { constexpr int o = X_DW + B_ * NVT + T_; wt[T_] += rows(o); });
        });
        sfor<NVL>([&](auto I) MI_LAMBDA { pw[I] = rows(DOWN + I); });
        for (int it = 0; it < P.iters; ++it) {
            MI_STAMP(15);
            int zero;
            MI_OPAQUE_ZERO(zero);
            const RowStore<RS> rit = rows.shifted(zero);
            const int par = it & 1;
            const RowStore<RS> xdw = rows.shifted((X_DW + (par ^ 1) * NR * NVT) * ST), fin = rows.shifted((X_FLG + par * NR) * ST),
                               fout = rows.shifted((X_FLG + (par ^ 1) * NR) * ST), down = rows.shifted((DOWN + (par ^ 1) * NVL) * ST),
                               dpair = rows.shifted((DPAIR + (par ^ 1) * NVL) * ST);
            float omT, oml[NLIMB > 1 ? NLIMB - 1 : 1];
            sweep_weights(fin, touch, omT, oml);
            float wtl[NVT], pwl[NVL > 0 ? NVL : 1];
            sfor<NVT>([&](auto T_) MI_LAMBDA { wtl[T_] = wt[T_]; });
            sfor<NVL>([&](auto I) MI_LAMBDA { pwl[I] = pw[I]; });
            float actn = 0.f;
            MI_STAMP(16);
            MI_STAMP(17);
            if constexpr (NPG > 0) { if (selfcol) {
                // The rows of slot j + 1 are loaded while slot j is worked on (as the limb roles' ground slots: this wave is alone on its SIMD).
                float gq[KPAIR][3][NV], gx[KPAIR][6];                 // dense rows | vt, lam x3, bodies word, mu: statically indexed
                auto gload_n = [&](auto J_) MI_LAMBDA {              // (ahead: the normal row and the six scalars, as the limb roles' ground slots)
                    constexpr int j = decltype(J_)::value;
                    sfor<NV>([&](auto I) MI_LAMBDA { gq[j][0][I] = rit(P_B + j * P_CSZ + I); });
                    sfor<4>([&](auto I_) MI_LAMBDA { gx[j][I_] = rit(P_B + j * P_CSZ + PVT + I_); });
                    gx[j][4] = rit(C_X + 1 + XI * j + 6); gx[j][5] = rit(C_X + 1 + XI * j + 7);
                };
                auto gload_t = [&](auto J_) MI_LAMBDA {
                    constexpr int j = decltype(J_)::value;
                    sfor<2>([&](auto K) MI_LAMBDA { sfor<NV>([&](auto I) MI_LAMBDA { gq[j][1 + K][I] = rit(P_B + j * P_CSZ + (1 + K) * NV + I); }); });
                };
#ifndef MI_MWC_PF_PAIR
#define MI_MWC_PF_PAIR 1
#endif
                if constexpr (MI_MWC_PF_PAIR) gload_n(std::integral_constant<int, 0>{});
                bool open = true;                                // uniform: no empty slot seen yet
                sfor<KPAIR>([&](auto J_) MI_LAMBDA {
                    constexpr int j = J_;
                    if (!open) return;
                    if constexpr (!MI_MWC_PF_PAIR) gload_n(std::integral_constant<int, j>{});
                    const bool onj = __builtin_bit_cast(unsigned, gx[j][4]) != 0xFFFFFFFFu;
                    if (!MI_WAVE_ANY(onj)) { open = false; return; }          // slots fill from the front: nothing behind an empty slot is read
                    gload_t(std::integral_constant<int, j>{});
                    if constexpr (MI_MWC_PF_PAIR && j + 1 < KPAIR) gload_n(std::integral_constant<int, j + 1>{});
                    MI_PHASE();
                    if (onj) {
                        const float mu = gx[j][5];
                        float (&g)[3][NV] = gq[j];
                        float ainv[3], lm[3];
                        // the weights of a row's coordinates: one for the trunk part, one per limb -- onto the partial sums of the diagonal and onto the impulse
                        sfor<3>([&](auto K) MI_LAMBDA {
                            float ag[NLIMB];
                            sfor<NLIMB>([&](auto L_) MI_LAMBDA { ag[L_] = 0.f; });
                            sfor<NV>([&](auto I) MI_LAMBDA { constexpr int l = limb_of_gi(decltype(I)::value); ag[l] += g[K][I] * g[K][I]; });
                            float a = P.cfm + omT * ag[0];
                            sfor<NLIMB - 1>([&](auto L_) MI_LAMBDA { a += oml[L_] * ag[L_ + 1]; });
                            ainv[K] = MI_RCP(a);
                            lm[K] = gx[j][1 + K];
                        });
                        const float vtn = gx[j][0];
                        auto wref = [&](auto I) MI_LAMBDA -> float& {
                            constexpr int i = decltype(I)::value;
                            if constexpr (MW::trunk_gi(i)) { constexpr int ti = MW::tidx(i); return wtl[ti]; } else { constexpr int li = lidx(i); return pwl[li]; }
                        };
                        auto dotw = [&](const float (&gr)[NV]) MI_LAMBDA -> float {
                            float s = 0.f;
                            sfor<NV>([&](auto I) MI_LAMBDA { s += gr[I] * wref(I); });
                            return s;
                        };
                        auto addw = [&](const float (&gr)[NV], const float dl) MI_LAMBDA {
                            float dg[NLIMB];
                            dg[0] = omT * dl;
                            sfor<NLIMB - 1>([&](auto L_) MI_LAMBDA { dg[L_ + 1] = oml[L_] * dl; });
                            sfor<NV>([&](auto I) MI_LAMBDA { constexpr int l = limb_of_gi(decltype(I)::value); wref(I) += gr[I] * dg[l]; });
                        };
                        const float ln = fmaxf(lm[0] - (dotw(g[0]) - vtn) * ainv[0], 0.f);
                        addw(g[0], ln - lm[0]);
                        float lt[2], vtg[2];
                        // both tangent rows from the SAME velocity, the disc (core/engine.hpp friction_disc; oracle/physics.c solve_blocks), ONE application
                        sfor<2>([&](auto K) MI_LAMBDA { vtg[K] = dotw(g[1 + K]); lt[K] = lm[1 + K] - vtg[K] * ainv[1 + K]; });
                        friction_disc(lt, lm[1], lm[2], vtg[0], vtg[1], ainv[1], ainv[2], mu * ln);
                        rit(P_B + j * P_CSZ + PLAM) = ln;
                        actn = (ln > 0.f) ? 1.f : actn;
                        sfor<2>([&](auto K) MI_LAMBDA {
                            const float nl_ = lt[K];
                            rit(P_B + j * P_CSZ + PLAM + 1 + K) = nl_;
                            addw(g[1 + K], nl_ - lm[1 + K]);
                        });
                    }
                });
            } }
            MI_STAMP(18);
            const float iomT = 1.f / omT;
            sfor<NVT>([&](auto T_) MI_LAMBDA { xdw(PAIR_ROLE * NVT + T_) = (wtl[T_] - wt[T_]) * iomT; });
            float dp[NVL > 0 ? NVL : 1];
            sfor<NV>([&](auto I) MI_LAMBDA {
                if constexpr (!MW::trunk_gi(I)) {
                    constexpr int li = lidx(I);
                    dp[li] = (pwl[li] - pw[li]) / om_of<decltype(I)::value>(omT, oml);
                    dpair(li) = dp[li];
                }
            });
            fout(PAIR_ROLE) = actn;
            MI_STAMP(19);
            bar();                                                                                   // ---- one barrier per sweep
            MI_STAMP(20);
            sfor<NVT>([&](auto T_) MI_LAMBDA { sfor<NR>([&](auto B_) MI_LAMBDA { constexpr int o = B_ * NVT + T_; wt[T_] += xdw(o); }); });
            sfor<NVL>([&](auto I) MI_LAMBDA { pw[I] += down(I) + dp[I]; });
            MI_STAMP(21);
        }
        MI_STAMP(12);
        // ============================================================ P5: impulses of the groups -> warm start of the next sub-step, world force on side a
        // The <= KPAIR slots are read once (static addresses, one frame per occupied slot); every group then SELECTS its values from them by the group
        // number in the slot's bodies word.  (Until round 6 each of the NPG groups looked its slot up through pmap at a run-time address and built its
        // own frame: 2.5 k instructions on the wave that finishes last -- this loop is the tail of the kernel.)
        if constexpr (NPG > 0) { if (selfcol) {
            int sg[KPAIR];
            float sl[KPAIR][3], sf[KPAIR][3];
            sfor<KPAIR>([&](auto J_) MI_LAMBDA {
                constexpr int j = J_;
                const unsigned bab = __builtin_bit_cast(unsigned, (float)rows(C_X + 1 + XI * j + 6));
                const bool onj = bab != 0xFFFFFFFFu;
                sg[j] = onj ? (int)(bab >> 24) : -1;
                sfor<3>([&](auto K) MI_LAMBDA { sl[j][K] = onj ? (float)rows(P_B + j * P_CSZ + PLAM + K) : 0.f; sf[j][K] = 0.f; });
                if (scol->pairf.p) {
                    if (MI_WAVE_ANY(onj)) {
                        float n[3], t1[3], t2[3];
                        sfor<3>([&](auto I_) MI_LAMBDA { n[I_] = onj ? (float)rows(C_X + 1 + XI * j + 3 + I_) : (I_ == 2 ? 1.f : 0.f); });
                        contact_frame(n, t1, t2);
                        sfor<3>([&](auto K) MI_LAMBDA { sf[j][K] = (n[K] * sl[j][0] + t1[K] * sl[j][1] + t2[K] * sl[j][2]) * invh; });
                    }
                }
            });
            sfor<NPG>([&](auto G_) MI_LAMBDA {
                constexpr int g = G_;
                float l[3] = {0.f, 0.f, 0.f}, f[3] = {0.f, 0.f, 0.f};
                bool mine = false;
                sfor<KPAIR>([&](auto J_) MI_LAMBDA {
                    const bool me = sg[J_] == g;
                    mine = mine || me;
                    sfor<3>([&](auto K) MI_LAMBDA { l[K] = me ? sl[J_][K] : l[K]; f[K] = me ? sf[J_][K] : f[K]; });
                });
                // (a group without a contact now whose impulse in memory is zero already -- nearly all of them -- is left alone: an impulse of zero
                //  normal part has zero tangential parts and a zero force, friction_disc; 78 stores per env were the tail of the kernel)
                if (MI_WAVE_ANY(mine || ((prevm >> g) & 1u))) {
                    sfor<3>([&](auto K) MI_LAMBDA { scol->lamp(3 * g + K) = l[K]; });
                    if (scol->pairf.p) sfor<3>([&](auto K) MI_LAMBDA { scol->pairf(3 * g + K) = f[K]; });
                }
            });
        } }
        MI_STAMP(13);
    }

    // ---------------------------------------------------------------- one role of a sub-step
    // FUSED: the sub-step is one of several inside one launch (substeps_fused below): the role also integrates the trunk's dofs and the root
    // state -- every limb role holds the trunk part of the whitened velocity, identical bit for bit (the sweeps sum the blocks' contributions
    // in role order), so each of them arrives at the same new root / trunk state by itself and nothing of it has to be handed over.
    template <int R, bool FUSED = false, int RS, class BAR>
    MI_HD void substep_role_c(const SimParams& P, const float* tau, const float h, const RowStore<RS> rows, const Strided lamc,
                              const Strided laml, const Strided sensor, const Strided dof_force, const float mu_env, const SelfCol* scol,
                              const BAR& bar) {
        constexpr int ST = RowStore<RS>::stride;
        constexpr int NLR_ = nl(R), RLEN = rlen(R), GCB = gcb(R), GCSZ = gcsz(R), KCAP = kcap(R), LF = lfirst(R);
        static_assert(R < NRL, "limb role");
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
#if defined(MI_TIMING)
        unsigned long long* const tstamp = this->tstamp;
#endif
        MI_STAMP(0);
        BodyTmp tb[NTB];
        this->template trunk_down<R, 0, X_LR>(P, c, tb, nullptr, nullptr, nullptr, nullptr, rows);
        MI_PHASE();
        float Ldi[NVA], y[NVA], w[NVA], v[NVA];
        v[0] = root[7]; v[1] = root[8]; v[2] = root[9]; v[3] = root[10]; v[4] = root[11]; v[5] = root[12];
        sfor<ND>([&](auto D) MI_LAMBDA { v[OFF + D] = qd[D]; });
        sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::trunk_entry(E_)) L[E_] = 0.f; });
        sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) y[I] = 0.f; });
        // per-dof `actor_params` factors of the dofs this role sees (the bodies' mass factors are applied where the tree pass forms their inertias)
        float sc_damp[M::NDA], sc_stiff[M::NDA], sc_arm[M::NDA];
        sfor<ND>([&](auto D) MI_LAMBDA { sc_damp[D] = 1.f; sc_stiff[D] = 1.f; sc_arm[D] = 1.f; });
        if constexpr (B::SCALED) {
            if (this->actor_scale.p != nullptr) {
                sfor<ND>([&](auto D) MI_LAMBDA {
                    if constexpr (MW::template sees_gi<R>(OFF + D)) {
                        sc_damp[D] = this->actor_scale(B::AS_DAMP + D); sc_stiff[D] = this->actor_scale(B::AS_STIFF + D); sc_arm[D] = this->actor_scale(B::AS_ARM + D);
                    }
                });
            }
        }
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (MW::role_of_gi(gi) == R) {
                const float K = M::dof_stiffness[d] * sc_stiff[d], Dm = M::dof_damping[d] * sc_damp[d];
                L[M::midx[gi][gi]] += M::dof_armature[d] * sc_arm[d] + h * Dm + h * h * K;
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
        MI_STAMP(1);
        bar();                                                                                       // ---- B1
        MI_STAMP(2);
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
        // register-allocation fence (Sim::alloc_fence): without this never-taken block the kernel spills 220 registers instead of 60
        this->alloc_fence([&](float f) MI_LAMBDA {
            sfor<M::NM>([&](auto E_) MI_LAMBDA { if constexpr (MW::trunk_entry(E_)) L[E_] *= f; });
            sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) c.bias[I] *= f; });
        });
        sfor<OFF>([&](auto I) MI_LAMBDA { y[I] = -c.bias[I]; });
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (MW::trunk_gi(gi)) {
                const float K = M::dof_stiffness[d] * sc_stiff[d], Dm = M::dof_damping[d] * sc_damp[d];
                L[M::midx[gi][gi]] += M::dof_armature[d] * sc_arm[d] + h * Dm + h * h * K;
                y[gi] = tau[d] - c.bias[gi] - K * (q[d] - M::dof_springref[d]) - (Dm + h * K) * qd[d];
            }
        });
        sfor<NRL>([&](auto R_) MI_LAMBDA {      // (the limb roles: the pair role has no limb to eliminate)
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
        // last sub-step's impulses of the own ground spheres are loaded where a contact is built.  (Round 3 let the leg roles issue all their
        // loads together up front -- 24 registers that pushed those roles into 24 spilled VGPRs / 100 B of scratch; without it they compile with
        // no spill at all and the step takes the same time, profiles/r4b_humanoid_prefetch_and_hand_slp_ab.txt.  -DMI_MWC_PREFETCH_KCAP=4 restores it.)
#ifndef MI_MWC_PREFETCH_KCAP
#define MI_MWC_PREFETCH_KCAP 99
#endif
        constexpr bool PREFETCH_LAMC = KCAP >= MI_MWC_PREFETCH_KCAP;
        float lprev[PREFETCH_LAMC ? (NSPH > 0 ? NSPH : 1) : 1][3];
        if constexpr (PREFETCH_LAMC) sfor<NSPH>([&](auto S_) MI_LAMBDA {
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
                sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { rows(g0 + K) = g[K]; });
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
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA {
                                constexpr int gi = M::chain[b][C], idx = shape_idx<R>(gi);
                                cb[(k * RLEN + idx) * ST] = g[k][C];
                            });
                            sfor<RLEN>([&](auto I_) MI_LAMBDA { if constexpr (!in_chain<R>(b, I_)) cb[(k * RLEN + I_) * ST] = 0.f; });   // the rest of the fixed shape
                            float l0;
                            if constexpr (PREFETCH_LAMC) l0 = lprev[s][k] * P.warm; else l0 = lamc(3 * s + k) * P.warm;
                            cb[(3 * RLEN + 1 + k) * ST] = l0;
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { wadd(std::integral_constant<int, M::chain[b][C]>{}, g[k][C] * l0); });
                        });
                        cb[(3 * RLEN) * ST] = (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
                        act = 1.f;
                    }
                }
                cnt += on ? 1 : 0;
                slot8(s) = (signed char)j;
            }
        });
        if (scol != nullptr && scol->dropped != nullptr && ndrop > 0) MI_ATOMIC_ADD_INT(scol->dropped, ndrop);
        // ---- the trunk role's own self-collision groups (pg_trunk_role: trunk x trunk, trunk x arm, arm x arm): narrow phase on the sphere centres of
        //      this role's tree pass, in its slack before B2.  Everything a slot holds -- point, normal, bodies word, friction, velocity target, warm-start
        //      impulses (their loads are in flight across the barrier) -- waits in registers until the pair role's slot count is visible.
        float tx[NTG > 0 ? NTG : 1][8], tvt[NTG > 0 ? NTG : 1], tl0[NTG > 0 ? NTG : 1][3];
        bool tnear[NTG > 0 ? NTG : 1];
        if constexpr (R == M::TRUNK_ROLE && NTG > 0) {
            if (selfcol) {
                float capm[M::NCAP][3];
                sfor<M::NCAP>([&](auto C_) MI_LAMBDA {
                    if constexpr (owned<R>(M::cap_body[C_])) sfor<3>([&](auto I_) MI_LAMBDA { capm[C_][I_] = 0.5f * (c.xcs[M::cap_s0[C_]][I_] + c.xcs[M::cap_s1[C_]][I_]); });
                });
                sfor<NTG>([&](auto T_) MI_LAMBDA {
                    constexpr int g = NPGP + T_;
                    MI_PHASE();
                    float best, bca[3], bcb[3], brb, n[3];
                    int bk, bab;
                    pair_group_search<g>(P, c.xcs, capm, best, bca, bcb, bk);
                    tnear[T_] = best < P.contact_offset;
                    sfor<8>([&](auto I_) MI_LAMBDA { tx[T_][I_] = 0.f; });
                    tvt[T_] = 0.f;
                    sfor<3>([&](auto K) MI_LAMBDA { tl0[T_][K] = 0.f; });
                    if (MI_WAVE_ANY(tnear[T_])) {
                        pair_group_pick<g>(bk, bca, bcb, bab, brb, tx[T_][7], n);
                        sfor<3>([&](auto I_) MI_LAMBDA { tx[T_][I_] = bcb[I_] + n[I_] * (brb + 0.5f * best); tx[T_][3 + I_] = n[I_]; });
                        tx[T_][6] = __builtin_bit_cast(float, bab);
                        const float gap = best - P.rest_offset;
                        tvt[T_] = (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
                        sfor<3>([&](auto K) MI_LAMBDA { tl0[T_][K] = scol->lamp(3 * g + K) * P.warm; });
                    }
                });
            }
        }
        MI_STAMP(3);
        MI_STAMP(4);
        bar();                                                                                       // ---- B2: tree-pass exchange is dead, C_X is published
        MI_STAMP(5);
        if constexpr (R == M::TRUNK_ROLE && NTG > 0) {
            if (selfcol) {
                // slots behind the pair role's, in table order (the trunk entries of every slot's dense rows are zeroed by the pair role meanwhile)
                int cntp = 0, pdrop = 0;
                sfor<KPAIR>([&](auto J_) MI_LAMBDA { cntp += (__builtin_bit_cast(unsigned, (float)rows(C_X + 1 + XI * J_ + 6)) != 0xFFFFFFFFu) ? 1 : 0; });
                sfor<NTG>([&](auto T_) MI_LAMBDA {
                    const bool on = tnear[T_] && (cntp < KPAIR);
                    pdrop += (tnear[T_] && !on) ? 1 : 0;
                    if (MI_WAVE_ANY(on)) {
                        if (on) {
                            float* xi = rows.ptr(C_X + 1 + XI * cntp);
                            sfor<8>([&](auto I_) MI_LAMBDA { xi[I_ * ST] = tx[T_][I_]; });
                            float* pb = rows.ptr(P_B + cntp * P_CSZ);
                            pb[PVT * ST] = tvt[T_];
                            sfor<3>([&](auto K) MI_LAMBDA { pb[(PLAM + K) * ST] = tl0[T_][K]; });
                        }
                    }
                    cntp += on ? 1 : 0;
                });
                if (scol->dropped != nullptr && pdrop > 0) MI_ATOMIC_ADD_INT(scol->dropped + scol->dstride, pdrop);
            }
        }
        // ============================================================ self-contact rows: every body of mine adds its half
        if constexpr (R == M::TRUNK_ROLE) sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int o = XW0 + MW::tidx(I); rows(o) = w[I]; } });
        if (selfcol) {
            MI_STAMP(6);
            bar();                                                                                   // ---- B3: the pair role has zeroed the dense rows
            MI_STAMP(7);
            for (int j = 0; j < KPAIR; ++j) {
                const float* xi = rows.ptr(C_X + 1 + XI * j);
                const unsigned bab = __builtin_bit_cast(unsigned, xi[6 * ST]);
                const bool onj = bab != 0xFFFFFFFFu;
                if (!MI_WAVE_ANY(onj)) break;                    // slots fill from the front
                const int ba = (int)(bab & 255u), bb = (int)((bab >> 8) & 255u);
                float* pb = rows.ptr(P_B + j * P_CSZ);
                // one evaluation per chain GROUP of this role (a leg: one; the trunk role: the trunk's chain and one per arm) for +lambda on side a and
                // -lambda on side b together: entry C of the row over the group's deepest chain carries [C moves a's body] - [C moves b's body]
                sfor<NB>([&](auto LF_) MI_LAMBDA {
                    constexpr int lf = LF_;
                    if constexpr (chain_leaf<R>(lf)) {
                        constexpr int CLF = M::chain_len[lf];
                        int sa = CLF, sb = CLF;                 // first chain entry that moves the side's body (CLF: the body is not in this group)
                        sfor<NB>([&](auto B_) MI_LAMBDA {
                            constexpr int b = B_;
                            if constexpr (chain_group<R>(b) == lf) { constexpr int sk = CLF - M::chain_len[b]; sa = (ba == b) ? sk : sa; sb = (bb == b) ? sk : sb; }
                        });
                        const bool me = onj && ((sa < CLF) || (sb < CLF));
                        float g[3][M::MAXCHAIN];
                        sfor<3>([&](auto K) MI_LAMBDA { sfor<CLF>([&](auto C) MI_LAMBDA { g[K][C] = 0.f; }); });
                        if (MI_WAVE_ANY(me)) {
                            if (me) {
                                float x[3], fr[3][3], W[3][6];
                                sfor<3>([&](auto I_) MI_LAMBDA { x[I_] = xi[I_ * ST]; fr[0][I_] = xi[(3 + I_) * ST]; });
                                contact_frame(fr[0], fr[1], fr[2]);
                                sfor<3>([&](auto K) MI_LAMBDA {
                                    cross3(x, fr[K], W[K]);
                                    W[K][3] = fr[K][0]; W[K][4] = fr[K][1]; W[K][5] = fr[K][2];
                                });
                                sfor<CLF>([&](auto C) MI_LAMBDA {
                                    constexpr int c = C, gi = M::chain[lf][c];
                                    const float cc = ((c >= sa) ? 1.f : 0.f) - ((c >= sb) ? 1.f : 0.f);
                                    if constexpr (gi >= OFF) sfor<3>([&](auto K) MI_LAMBDA { g[K][c] = cc * dot6(S[gi - OFF], W[K]); });
                                    else if constexpr (gi < 3) sfor<3>([&](auto K) MI_LAMBDA { g[K][c] = cc * W[K][3 + gi]; });
                                    else sfor<3>([&](auto K) MI_LAMBDA { g[K][c] = cc * W[K][gi - 3]; });
                                });
                                sfor<CLF>([&](auto K) MI_LAMBDA {
                                    constexpr int k = K, i = M::chain[lf][k];
                                    const float di = Ldi[i];
                                    const float z0 = g[0][k] * di, z1 = g[1][k] * di, z2 = g[2][k] * di;
                                    g[0][k] = z0; g[1][k] = z1; g[2][k] = z2;
                                    sfor<CLF - 1 - k>([&](auto T) MI_LAMBDA {
                                        constexpr int kk = k + 1 + T, jj = M::chain[lf][kk];
                                        const float l = L[M::midx[i][jj]];
                                        g[0][kk] -= l * z0; g[1][kk] -= l * z1; g[2][kk] -= l * z2;
                                    });
                                });
                                // warm start of these rows on the group's own limb coordinates, from the registers (their trunk part: the pair role)
                                sfor<3>([&](auto K) MI_LAMBDA {
                                    const float l0 = pb[(PLAM + K) * ST];
                                    sfor<CLF>([&](auto C) MI_LAMBDA { constexpr int gi = M::chain[lf][C]; if constexpr (!MW::trunk_gi(gi)) w[gi] += g[K][C] * l0; });
                                });
                                // trunk entries: added (zeroed by the pair role, other roles add to them too)
                                sfor<3>([&](auto K) MI_LAMBDA {
                                    sfor<CLF>([&](auto C) MI_LAMBDA { constexpr int gi = M::chain[lf][C]; if constexpr (MW::trunk_gi(gi)) lds_add(&pb[(K * NV + gi) * ST], g[K][C]); });
                                });
                            }
                        }
                        // the group's own limb entries: stored by this role alone, zero where the contact does not touch the limb
                        if (onj) sfor<3>([&](auto K) MI_LAMBDA {
                            sfor<CLF>([&](auto C) MI_LAMBDA { constexpr int gi = M::chain[lf][C]; if constexpr (!MW::trunk_gi(gi)) pb[(K * NV + gi) * ST] = g[K][C]; });
                        });
                    }
                });
            }
            MI_STAMP(8);
            bar();                                                                                   // ---- B4: rows complete
            MI_STAMP(9);
            // (the warm start of the self-contact rows: own limb coordinates above, trunk part by the pair role -- substep_pair)
        }
        sfor<NVT>([&](auto I) MI_LAMBDA { rows(X_DW + R * NVT + I) = dw[I]; });
        sfor<NLR_>([&](auto K) MI_LAMBDA { constexpr int o = DOWN + lidx(LF + K); rows(o) = w[LF + K]; });     // (round 0 carries the values themselves)
        rows(X_FLG + R) = act;
        if constexpr (!HAS_PAIR_ROLE && R == NR - 1) { rows(X_TOUCH) = 0.f; }
        MI_STAMP(10);
        bar();                                                                                       // ---- B5: round 0 of the exchange is complete
        MI_STAMP(11);
        // ============================================================ P4: block sweeps
        // round 0: every block's warm-start contribution to the trunk part, in block order (the pair block's: the self-contact rows' warm start)
        sfor<NV>([&](auto I) MI_LAMBDA {
            constexpr int i = I;
            if constexpr (MW::trunk_gi(i)) { sfor<NR>([&](auto B_) MI_LAMBDA { constexpr int o = X_DW + B_ * NVT + MW::tidx(i); w[i] += rows(o); }); }
        });
        const unsigned touch = __builtin_bit_cast(unsigned, (float)rows(X_TOUCH));
        {
            float wtl[NVT], wll[NLR_ > 0 ? NLR_ : 1];
            for (int it = 0; it < P.iters; ++it) {
                MI_STAMP(15);
                int zero;
                MI_OPAQUE_ZERO(zero);
                const RowStore<RS> rit = rows.shifted(zero);
                const int par = it & 1;
                const RowStore<RS> xdw = rows.shifted((X_DW + (par ^ 1) * NR * NVT) * ST), fin = rows.shifted((X_FLG + par * NR) * ST),
                                   fout = rows.shifted((X_FLG + (par ^ 1) * NR) * ST), down = rows.shifted((DOWN + (par ^ 1) * NVL) * ST),
                                   dpair = rows.shifted((DPAIR + (par ^ 1) * NVL) * ST);
                // weights of this sweep: the trunk is shared by all active blocks, a limb by its owner's block and the pair block
                float omT, oml[NLIMB > 1 ? NLIMB - 1 : 1];
                this->sweep_weights(fin, touch, omT, oml);
                const float iomT = 1.f / omT;
                sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int ti = MW::tidx(I); wtl[ti] = w[I]; } });
                sfor<NLR_>([&](auto K) MI_LAMBDA { wll[K] = w[LF + K]; });
                auto wget = [&](auto GI) MI_LAMBDA -> float {
                    constexpr int gi = decltype(GI)::value;
                    if constexpr (MW::trunk_gi(gi)) { constexpr int ti = MW::tidx(gi); return wtl[ti]; } else { constexpr int k = gi - LF; return wll[k]; }
                };
                float actn = 0.f;
                MI_STAMP(16);
                // ---- own limit rows
                sfor<ND>([&](auto D) MI_LAMBDA {
                    constexpr int d = D, gi = OFF + d;
                    if constexpr (M::dof_limited[d] && MW::template owns_gi<R>(gi)) {
                        constexpr int row = B::limrow(d), g0 = B::limoff(row);
                        float g[M::MAXCHAIN];
                        sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { g[K] = rit(g0 + K); });
                        // a row's coordinates: its dof's own limb (one weight: a limb is a path below the trunk) and the trunk -- the weights go onto the
                        // two partial sums of the diagonal and onto the impulse, not onto every coordinate
                        float al = 0.f, at = 0.f;
                        sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA {
                            constexpr int k = K, i = (k == 0) ? gi : M::anc[gi][k == 0 ? 0 : k - 1];
                            static_assert(MW::trunk_gi(i) || limb_of_gi(i) == limb_of_gi(gi), "a limit row touches its own limb and the trunk");
                            if constexpr (MW::trunk_gi(i)) at += g[k] * g[k]; else al += g[k] * g[k];
                        });
                        const float omL = om_of<gi>(omT, oml);
                        const float a = P.cfm + omL * al + omT * at;
                        float vn = g[0] * wget(std::integral_constant<int, gi>{});
                        sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { vn += g[1 + A_] * wget(std::integral_constant<int, M::anc[gi][A_]>{}); });
                        const float lo = rit(L_LAM + row);
                        const float nl_ = fmaxf(lo - (vn - rit(L_VT + row)) * MI_RCP(a), 0.f);
                        const float dl = nl_ - lo;
                        rit(L_LAM + row) = nl_;
                        actn = (nl_ > 0.f) ? 1.f : actn;
                        const float dlL = omL * dl, dlT = omT * dl;
                        sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA {
                            constexpr int k = K, i = (k == 0) ? gi : M::anc[gi][k == 0 ? 0 : k - 1];
                            if constexpr (MW::trunk_gi(i)) { constexpr int ti = MW::tidx(i); wtl[ti] += g[k] * dlT; } else { constexpr int kk = i - LF; wll[kk] += g[k] * dlL; }
                        });
                    }
                });
                MI_STAMP(17);
                // ---- own ground contacts: a lane's j-th contact, whichever sphere it is (fixed row shape [limb | trunk])
                // The rows of slot j + 1 are loaded while slot j is worked on (this wave is alone on its SIMD: nobody else hides the LDS latency, and a slot's
                // loads used to start only when the slot before it was finished -- a third of the time of this loop); slot j + 1 is not written by slot j.
                float gq[KCAP > 0 ? KCAP : 1][3][RLEN], gx[KCAP > 0 ? KCAP : 1][4];          // rows | vt, lam x3: statically indexed
                // (what is in flight across a slot: the NORMAL row and the four scalars of the next one -- 19 registers; its tangent rows are asked for when
                //  its turn comes and arrive while the normal row is worked on.  All three rows ahead cost the tree pass 50 units in registers: r6n.)
                auto gload_n = [&](auto J_) MI_LAMBDA {
                    constexpr int j = decltype(J_)::value;
                    sfor<RLEN>([&](auto C) MI_LAMBDA { gq[j][0][C] = rit(GCB + j * GCSZ + C); });
                    sfor<4>([&](auto I_) MI_LAMBDA { gx[j][I_] = rit(GCB + j * GCSZ + 3 * RLEN + I_); });
                };
                auto gload_t = [&](auto J_) MI_LAMBDA {
                    constexpr int j = decltype(J_)::value;
                    sfor<2>([&](auto K) MI_LAMBDA { sfor<RLEN>([&](auto C) MI_LAMBDA { gq[j][1 + K][C] = rit(GCB + j * GCSZ + (1 + K) * RLEN + C); }); });
                };
#ifndef MI_MWC_PF_LEG
#define MI_MWC_PF_LEG 1
#endif
                if constexpr (KCAP > 0 && MI_MWC_PF_LEG) { if (MI_WAVE_ANY(0 < cnt)) gload_n(std::integral_constant<int, 0>{}); }
                sfor<KCAP>([&](auto J_) MI_LAMBDA {
                    constexpr int j = J_;
                    const bool onj = j < cnt;
                    if (!MI_WAVE_ANY(onj)) return;                // (slots fill from the front)
                    if constexpr (!MI_MWC_PF_LEG) gload_n(std::integral_constant<int, j>{});
                    gload_t(std::integral_constant<int, j>{});
                    if constexpr (MI_MWC_PF_LEG && j + 1 < KCAP) { if (MI_WAVE_ANY(j + 1 < cnt)) gload_n(std::integral_constant<int, j + 1>{}); }
                    MI_PHASE();
                    if (onj) {
                        const float mu = 0.5f * ((mu_env >= 0.f ? mu_env : M::sph_mu[0]) + P.plane_mu);
                        float (&g)[3][RLEN] = gq[j];
                        float ainv[3], lm[3];
                        sfor<3>([&](auto K) MI_LAMBDA {
                            float alm[NLIMB > 1 ? NLIMB - 1 : 1];                 // the diagonal's partial sums per limb of this role and for the trunk: weights applied once
                            sfor<NLIMB - 1>([&](auto L_) MI_LAMBDA { alm[L_] = 0.f; });
                            sfor<NLR_>([&](auto K2) MI_LAMBDA { constexpr int l = limb_of_gi(LF + decltype(K2)::value); alm[l - 1] += g[K][K2] * g[K][K2]; });
                            float at = 0.f;
                            sfor<NVT>([&](auto T_) MI_LAMBDA { at += g[K][NLR_ + T_] * g[K][NLR_ + T_]; });
                            float a = P.cfm + omT * at;
                            sfor<NLIMB - 1>([&](auto L_) MI_LAMBDA { if constexpr (M::role_of_limb[L_ + 1] == R) a += oml[L_] * alm[L_]; });
                            ainv[K] = MI_RCP(a);
                            lm[K] = gx[j][1 + K];
                        });
                        const float vtn = gx[j][0];
                        auto dotw = [&](const float (&gr)[RLEN]) MI_LAMBDA -> float {
                            float s = 0.f;
                            sfor<NLR_>([&](auto K) MI_LAMBDA { s += gr[K] * wll[K]; });
                            sfor<NVT>([&](auto T_) MI_LAMBDA { s += gr[NLR_ + T_] * wtl[T_]; });
                            return s;
                        };
                        auto addw = [&](const float (&gr)[RLEN], const float dl) MI_LAMBDA {      // (the weights go onto the impulse: one fma per coordinate)
                            float dlim[NLIMB > 1 ? NLIMB - 1 : 1];
                            sfor<NLIMB - 1>([&](auto L_) MI_LAMBDA { dlim[L_] = oml[L_] * dl; });
                            const float dT = omT * dl;
                            sfor<NLR_>([&](auto K) MI_LAMBDA { constexpr int l = limb_of_gi(LF + decltype(K)::value); wll[K] += gr[K] * dlim[l - 1]; });
                            sfor<NVT>([&](auto T_) MI_LAMBDA { wtl[T_] += gr[NLR_ + T_] * dT; });
                        };
                        const float ln = fmaxf(lm[0] - (dotw(g[0]) - vtn) * ainv[0], 0.f);
                        addw(g[0], ln - lm[0]);
                        float lt[2], vtg[2];
                        // both tangent rows from the SAME velocity, the disc (core/engine.hpp friction_disc; oracle/physics.c solve_blocks), ONE application
                        sfor<2>([&](auto K) MI_LAMBDA { vtg[K] = dotw(g[1 + K]); lt[K] = lm[1 + K] - vtg[K] * ainv[1 + K]; });
                        friction_disc(lt, lm[1], lm[2], vtg[0], vtg[1], ainv[1], ainv[2], mu * ln);
                        rit(GCB + j * GCSZ + 3 * RLEN + 1) = ln;
                        actn = (ln > 0.f) ? 1.f : actn;
                        sfor<2>([&](auto K) MI_LAMBDA {
                            const float nl_ = lt[K];
                            rit(GCB + j * GCSZ + 3 * RLEN + 2 + K) = nl_;
                            addw(g[1 + K], nl_ - lm[1 + K]);
                        });
                    }
                });
                MI_STAMP(18);
                // this block's true contributions
                sfor<NV>([&](auto I) MI_LAMBDA { if constexpr (MW::trunk_gi(I)) { constexpr int ti = MW::tidx(I); xdw(R * NVT + ti) = (wtl[ti] - w[I]) * iomT; } });
                float dlo[NLR_ > 0 ? NLR_ : 1];
                sfor<NLR_>([&](auto K) MI_LAMBDA {
                    constexpr int gi = LF + K, li = lidx(gi);
                    dlo[K] = (wll[K] - w[gi]) / om_of<gi>(omT, oml);
                    down(li) = dlo[K];
                });
                fout(R) = actn;
                if constexpr (!HAS_PAIR_ROLE) { /* no pair block: nobody writes dpair; keep it zero */ if constexpr (R == 0) sfor<NVL>([&](auto I) MI_LAMBDA { dpair(I) = 0.f; }); }
                MI_STAMP(19);
                bar();                                                                               // ---- one barrier per sweep
                MI_STAMP(20);
                sfor<NV>([&](auto I) MI_LAMBDA {
                    constexpr int i = I;
                    if constexpr (MW::trunk_gi(i)) { sfor<NR>([&](auto B_) MI_LAMBDA { constexpr int o = B_ * NVT + MW::tidx(i); w[i] += xdw(o); }); }
                    else if constexpr (MW::role_of_gi(i) == R) { constexpr int k = i - LF, li = lidx(i); w[i] += dlo[k] + dpair(li); }
                });
                MI_STAMP(21);
            }
        }
        MI_STAMP(12);
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
                dof_force(d) = tau[d] - M::dof_stiffness[d] * sc_stiff[d] * (q[d] - M::dof_springref[d]) - M::dof_damping[d] * sc_damp[d] * v[OFF + d] + ll * invh;
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
                const float ln = onj ? cb[(3 * RLEN + 1) * ST] : 0.f, l1 = onj ? cb[(3 * RLEN + 2) * ST] : 0.f, l2 = onj ? cb[(3 * RLEN + 3) * ST] : 0.f;
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
                // (an unused pair slot holds whatever the LDS held before this launch: its contact point must not reach the lever arm below --
                //  garbage x 0 is 0 for a finite pattern and NaN for an Inf / NaN one; found by poisoning the LDS, tests/test_gpu_fullsize.py)
                sfor<3>([&](auto I_) MI_LAMBDA { x[I_] = onj ? xi[I_ * ST] : 0.f; n[I_] = onj ? xi[(3 + I_) * ST] : (I_ == 2 ? 1.f : 0.f); });
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
        } }
        sfor<NSENS>([&](auto K_) MI_LAMBDA {
            if constexpr (MW::template owns_body<R>(M::sens_body[K_])) sfor<6>([&](auto C) MI_LAMBDA { sensor(6 * K_ + C) = sens[6 * K_ + C]; });
        });
        MI_PHASE();
        sfor<ND>([&](auto D) MI_LAMBDA {
            if constexpr (FUSED ? MW::template sees_gi<R>(OFF + D) : MW::template owns_gi<R>(OFF + D)) { qd[D] = v[OFF + D]; q[D] += h * qd[D]; }
        });
        if constexpr (FUSED || R == M::TRUNK_ROLE) {
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
        MI_STAMP(13);
    }

    // ================================================================ all sub-steps of a control step inside one launch
    // Between two sub-steps nothing goes through HBM but the warm-start impulses, which each wave reads back from where it wrote them itself
    // (joint limits and ground spheres: the owner role; the limb-pair groups: the pair role).  Every limb role keeps its own q / qd in registers
    // and integrates the trunk's dofs and the root redundantly (FUSED above).  The one wave that has no dynamics of its own, the pair role, gets
    // the new pose -- root position + orientation and the 21 joint angles, for its positions-only forward kinematics -- through H: the half of
    // the trunk exchange area X_DW that the LAST sweep did not use (parity (iters - 1) & 1: written and read for the last time one barrier
    // earlier; next written before B5 of the following sub-step, long after the pair role has read it).  One barrier per sub-step boundary.
    static constexpr int H_Q = 7, H_SIZE = 7 + ND;
    static_assert(H_SIZE <= NR * NVT, "the pose hand-over fits one half of the trunk exchange area");
    // NSUB > 0: the sub-steps as straight-line code, one copy per sub-step (no loop for the optimiser to hoist out of); 0: a run-time loop
    template <int R, int NSUB = 0, int RS, class BAR>
    MI_HD void substeps_fused(const SimParams& P, const float* tau, const float h, const RowStore<RS> rows, const Strided lamc, const Strided laml,
                              const Strided sensor, const Strided dof_force, const float mu_env, const SelfCol* scol, const BAR& bar, const int n_sub) {
        constexpr int ST = RowStore<RS>::stride;
        const RowStore<RS> H = rows.shifted((X_DW + ((P.iters - 1) & 1) * NR * NVT) * ST);
        if constexpr (NSUB > 0) {
            sfor<NSUB>([&](auto I_) MI_LAMBDA {
                constexpr int i = decltype(I_)::value;
                if constexpr (HAS_PAIR_ROLE && R == PAIR_ROLE) {
                    if constexpr (i > 0) {
                        sfor<7>([&](auto K) MI_LAMBDA { this->root[K] = H(K); });
                        sfor<ND>([&](auto D) MI_LAMBDA { this->q[D] = H(H_Q + D); });
                    }
                    substep_pair(P, h, rows, scol, bar);
                } else {
                    this->template substep_role_c<R, true>(P, tau, h, rows, lamc, laml, sensor, dof_force, mu_env, scol, bar);
                }
                if constexpr (i + 1 < NSUB) {
                    if constexpr (!(HAS_PAIR_ROLE && R == PAIR_ROLE)) {
                        sfor<ND>([&](auto D) MI_LAMBDA { if constexpr (MW::template owns_gi<R>(OFF + D)) H(H_Q + D) = this->q[D]; });
                        if constexpr (R == M::TRUNK_ROLE) sfor<7>([&](auto K) MI_LAMBDA { H(K) = this->root[K]; });
                    }
                    bar();
                }
            });
            return;
        }
        for (int i = 0; i < n_sub; ++i) {
            // An integer zero the optimiser cannot see through, added to every global pointer of the sub-step: the ~200 per-lane addresses
            // of the impulse / sensor / joint-force tensors are loop invariant, and hoisted out of this loop they would stay live in VGPR
            // pairs through the whole body (the first build of this loop spilled 306 VGPRs where the one-sub-step kernel spills 50).
            int zero;
            MI_OPAQUE_ZERO(zero);
            // (and the step size, the friction and the parameter block: whatever is computed from them stays inside the iteration)
            const SimParams* Pp = &P;
            MI_OPAQUE_SPTR(Pp);
            float h_i = h, mu_i = mu_env;
            MI_OPAQUE_VF(h_i);
            MI_OPAQUE_VF(mu_i);
            // (the stride too: k * stride, the offset of row k of a [k][N] tensor, is a uniform 64-bit value per k -- ~170 of them)
            int stride_i = lamc.stride;
            MI_OPAQUE_SINT(stride_i);
            const Strided lamc_i{lamc.p + zero, stride_i}, laml_i{laml.p + zero, stride_i}, sensor_i{sensor.p + zero, stride_i},
                          dof_force_i{dof_force.p + zero, stride_i};
            SelfCol sc_i{Strided{nullptr, 1}, Strided{nullptr, 1}, nullptr, 0};
            if (scol != nullptr)
                sc_i = SelfCol{Strided{scol->lamp.p + zero, stride_i}, Strided{scol->pairf.p ? scol->pairf.p + zero : nullptr, stride_i},
                               scol->dropped ? scol->dropped + zero : nullptr, scol->dstride};
            const SelfCol* scp = scol != nullptr ? &sc_i : nullptr;
            if constexpr (HAS_PAIR_ROLE && R == PAIR_ROLE) {
                if (i > 0) {
                    sfor<7>([&](auto K) MI_LAMBDA { this->root[K] = H(K); });
                    sfor<ND>([&](auto D) MI_LAMBDA { this->q[D] = H(H_Q + D); });
                }
                substep_pair(*Pp, h_i, rows, scp, bar);
            } else {
                this->template substep_role_c<R, true>(*Pp, tau, h_i, rows, lamc_i, laml_i, sensor_i, dof_force_i, mu_i, scp, bar);
            }
            if (i + 1 < n_sub) {
                if constexpr (!(HAS_PAIR_ROLE && R == PAIR_ROLE)) {
                    sfor<ND>([&](auto D) MI_LAMBDA { if constexpr (MW::template owns_gi<R>(OFF + D)) H(H_Q + D) = this->q[D]; });
                    if constexpr (R == M::TRUNK_ROLE) sfor<7>([&](auto K) MI_LAMBDA { H(K) = this->root[K]; });
                }
                bar();
            }
        }
    }
};

}  // namespace mi
