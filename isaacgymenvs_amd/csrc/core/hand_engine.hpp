// hand_engine.hpp -- physics sub-step of a fixed-base manipulator + ONE free isotropic rigid object (ShadowHand task).
//
// Replaces gym.simulate() for reference isaacgymenvs/tasks/shadow_hand.py.  Built on the pieces of core/engine.hpp (tree
// pass, branch-sparse L^T L factor, whitened PGS, compact LDS row store); what is specific here:
//   * gravity is disabled on the hand (shadow_hand.py:239) and enabled on the object;
//   * implicit PD position drives: tau = kp (target - q) - D qd with kp from the MJCF position actuators
//     (shared.xml:250-269), targets = cur_targets of the task (shadow_hand.py:683-698);
//   * the four fixed tendons coupling J0/J1 (shared.xml:53-69) as soft two-sided limits on c0 q0 + c1 q1
//     (limit_stiffness / damping set by the task, shadow_hand.py:256-266), implicit like the drives;
//   * the manipulated cube: a free body with isotropic inertia (cube_multicolor.urdf: 5 cm box, density 567), so its
//     whitening is a constant diagonal; contact = hand collision geometry sampled by spheres (generated os_* tables)
//     against the exact box; every active contact takes a data-dependent slot of the compact LDS store and contributes
//     3 rows over [chain of the hand body | 6 object dofs].
// Same maths as oracle/hand.py (dense, numpy, fp64).
#pragma once
#include "engine.hpp"

namespace mi {

struct FreeBody {           // world frame
    float pos[3], quat[4], vel[3], angvel[3];
};
struct ObjectParams {       // mirrors the object part of MiHandParams
    float half, mass, inertia, mu;   // cube half size, mass, isotropic inertia, combined friction
    float fw[3] = {0.f, 0.f, 0.f};   // external world-frame force on the object for this step (apply_rigid_body_force_tensors)
    // objects other than the cube (SHAPE != 0): semi-axes of the ellipsoid and the principal inertias about the body axes
    float dims[3] = {0.f, 0.f, 0.f};
    float inertia3[3] = {0.f, 0.f, 0.f};
    // `actor_params.object` factors (ShadowHand.yaml:132-159): mass -- with it the inertia -- and size (gym.set_actor_scale leaves the
    // mass alone)
    MI_HD void randomise(float mass_factor, float size_factor) {
        mass *= mass_factor; inertia *= mass_factor; half *= size_factor;
        for (int k = 0; k < 3; ++k) { inertia3[k] *= mass_factor; dims[k] *= size_factor; }
    }
};
constexpr int OBJ_BOX = 0, OBJ_CAPSULE = 1, OBJ_ELLIPSOID = 2;   // objectType "block" / "pen" / "egg" (shadow_hand.py:86-96)
// columns of the ShadowHand's per-env `actor_scale` tensor (`actor_params` domain randomisation, ShadowHand.yaml:104-159)
constexpr int HS_MASS = 0, HS_DAMPING = 1, HS_STIFFNESS = 2, HS_TENDON_STIFFNESS = 3, HS_TENDON_DAMPING = 4, HS_OBJECT_MASS = 5,
              HS_OBJECT_SCALE = 6, HS_COLUMNS = 8;

// Effort-limited position drive of one dof, one Gauss-Seidel visit (oracle/physics.c drive_clamp_update).  The drive's implicit force at the end
// of the sub-step is F = fa - c v_d (fa = kp (target - q), c = D + h kp: both already in the factor and the right-hand side); the actuator delivers
// clamp(F, +-fmax).  What the clamp removes is an impulse rho on the dof with fa - c v_d + rho / h in [-fmax, fmax], rho = 0 inside; for the row's
// own response a (v_d changes by a per unit impulse) the update is closed-form; while the dof's joint limit holds it (limit impulse > 0) the caller
// passes a = 0: the limit row absorbs rho, rho = h (clamp(F) - F).  Returns the change of rho.
MI_HD float drive_clamp_update(const float fa, const float c, const float fmax, const float invh, const float v_d, const float a, float& rho) {
    const float k = fmaxf(invh - c * a, 0.1f * invh);
    const float Ff = fa - c * v_d + rho * c * a;
    const float rn = (Ff > fmax) ? (fmax - Ff) * MI_RCP(k) : ((Ff < -fmax) ? (-fmax - Ff) * MI_RCP(k) : 0.f);
    const float dr = rn - rho;
    rho = rn;
    return dr;
}

template <class M>
struct HandSim : Sim<M> {
    using B = Sim<M>;
    static constexpr int NB = M::NB, ND = M::ND, NV = M::NV, OFF = M::OFF, NSENS = M::NSENS, NOS = M::NOS, NLIM = B::NLIM, NVA = B::NVA;
    static_assert(M::FIXED == 1 && M::NSPH == 0, "HandSim: fixed-base model without ground spheres");
#ifndef MI_HAND_KMAX
#define MI_HAND_KMAX 12
#endif
    static constexpr int KMAX = MI_HAND_KMAX;                // active object contacts kept per env
    Strided limit_shift{nullptr, 1};                         // [2 * ND] per-env shifts of the lower / upper joint limits (required)
    int drive_clamp = 1;                                     // effort-limited drives on (HandView::drive_clamp)
    // a dof whose position drive has a force range (MJCF forcerange, shared.xml:250-269; allegro_hand.py:264 effort).  Its clamp rides on
    // the dof's joint-limit row -- the same whitened vector g = s L^-1 e_d -- so every such dof has one
    static constexpr bool clamped(int d) { return M::dof_force_limit[d] > 0.f && M::dof_kp[d] > 0.f && M::dof_limited[d]; }
    static constexpr bool any_clamped() { for (int d = 0; d < ND; ++d) if (clamped(d)) return true; return false; }
    // (a driven dof with a force range but NO joint limit would silently get an unlimited drive -- refuse such a model at compile time)
    static constexpr bool clamps_have_rows() {
        for (int d = 0; d < ND; ++d) if (M::dof_force_limit[d] > 0.f && M::dof_kp[d] > 0.f && !M::dof_limited[d]) return false;
        return true;
    }
    static_assert(clamps_have_rows(), "a force-limited position drive rides on its dof's joint-limit row: every such dof must be limited");
    static constexpr int HCH = M::MAXCHAIN;                  // stored row width: the hand chain; the 6 object entries are re-derived
    static constexpr int H_LIMG = B::limoff(NLIM);
    static constexpr int H_CB = H_LIMG + 4 * NLIM;           // limit G | Ainv, vt, lam, rho (drive clamp impulse) | contact slots
    // one contact slot: 3 rows over the hand chain | normal n (3), lever rc = contact point - object COM (3) | Ainv x3, vt_n, lam x3.
    // The object part of row k, -[u_k; rc x u_k] whitened, follows from (n, rc): 6 floats stored instead of 18.
    static constexpr int H_GEO = 3 * HCH, H_AUX = H_GEO + 6;
    static constexpr int H_CSZ = H_AUX + 7;
    static constexpr int BODY_CAP = 4;                        // contacts admitted per hand body (manifold size)
    // The contacts of one hand body take consecutive slots (slots are handed out body by body): one dword per sphere-carrying body,
    // first slot | count << 8.  The sweeps and the sensor pass walk a body's <= BODY_CAP contacts instead of its (up to 30) spheres.
    static constexpr int H_BODYSLOT = H_CB + KMAX * H_CSZ;
    static constexpr int ROW_SLOTS = H_BODYSLOT + B::NOSB;
    // Round 2: the store is 605 floats per env (77 KB per 32-env workgroup) -- TWO workgroups fit a CU's 160 KB, so the 512
    // workgroups of ShadowHand@16384 are resident at once instead of running in two rounds (it was 1224 floats: contact slots with
    // the object part stored, KMAX 16, a slot index per sphere and a 216-float block of body poses handed from the tree
    // pass to the narrow phase, which now lives in a per-lane local array).  Measured before the rewrite with a KMAX = 3 build and
    // its padded twin (tools/hand_residency_ab.sh): two resident workgroups per CU run the same 16384 envs 1.64x faster.
    static_assert((size_t)ROW_SLOTS * 32 * sizeof(float) <= 80 * 1024, "two hand workgroups per CU");
    static constexpr int LANES = 32;

    FreeBody obj;

    static constexpr int sensor_of(int b) { return B::sensor_of(b); }

    // sphere (centre c in box frame, radius r) vs cube of half size a: signed distance, outward normal (box frame)
    MI_HD static void sphere_box(const float* c, float r, float a, float* dist, float* n) {
        const float qx = fminf(fmaxf(c[0], -a), a), qy = fminf(fmaxf(c[1], -a), a), qz = fminf(fmaxf(c[2], -a), a);
        const float dx = c[0] - qx, dy = c[1] - qy, dz = c[2] - qz;
        const float d2 = dx * dx + dy * dy + dz * dz;
        const float px = a - fabsf(c[0]), py = a - fabsf(c[1]), pz = a - fabsf(c[2]);
        // inside: push out along the axis of least penetration
        const bool ix = (px <= py) && (px <= pz), iy = !ix && (py <= pz);
        const float pen = ix ? px : (iy ? py : pz);
        const bool outside = d2 > 1e-24f;
        const float inv = MI_RSQ(fmaxf(d2, 1e-30f));
        const float nd = d2 * inv;
        *dist = outside ? nd - r : -pen - r;
        n[0] = outside ? dx * inv : (ix ? (c[0] >= 0.f ? 1.f : -1.f) : 0.f);
        n[1] = outside ? dy * inv : (iy ? (c[1] >= 0.f ? 1.f : -1.f) : 0.f);
        n[2] = outside ? dz * inv : ((!ix && !iy) ? (c[2] >= 0.f ? 1.f : -1.f) : 0.f);
    }

    // sphere (centre c in the object frame, radius r) vs ellipsoid with semi-axes a: first-order signed distance f / |grad f| of
    // f = |c / a| - 1 scaled back to length (exact on the surface and along the axes, a few % off one radius away -- contacts live within
    // contact_offset = 2 mm of the surface), outward normal = normalised gradient (exact direction on the surface)
    MI_HD static void sphere_ellipsoid(const float* c, float r, const float* a, float* dist, float* n) {
        const float u[3] = {c[0] / a[0], c[1] / a[1], c[2] / a[2]};
        const float g[3] = {u[0] / a[0], u[1] / a[1], u[2] / a[2]};
        const float k0 = MI_SQRT(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        const float k1sq = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
        const bool ok = k1sq > 1e-20f;
        const float ik1 = MI_RSQ(fmaxf(k1sq, 1e-30f));
        *dist = (ok ? k0 * (k0 - 1.f) * ik1 : -fminf(a[0], fminf(a[1], a[2]))) - r;
        n[0] = ok ? g[0] * ik1 : 0.f; n[1] = ok ? g[1] * ik1 : 0.f; n[2] = ok ? g[2] * ik1 : 1.f;
    }
    // sphere vs capsule along the object's z axis (radius rc, half length hl of the cylindrical part): exact
    MI_HD static void sphere_capsule(const float* c, float r, float rc, float hl, float* dist, float* n) {
        const float pz = fminf(fmaxf(c[2], -hl), hl);
        const float d[3] = {c[0], c[1], c[2] - pz};
        const float d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const bool ok = d2 > 1e-24f;
        const float inv = MI_RSQ(fmaxf(d2, 1e-30f));
        *dist = d2 * inv - rc - r;
        n[0] = ok ? d[0] * inv : 1.f; n[1] = ok ? d[1] * inv : 0.f; n[2] = ok ? d[2] * inv : 0.f;
    }
    template <int SHAPE>
    MI_HD static void sphere_object(const float* c, float r, const ObjectParams& OP, float* dist, float* n) {
        if constexpr (SHAPE == OBJ_BOX) sphere_box(c, r, OP.half, dist, n);
        else if constexpr (SHAPE == OBJ_CAPSULE) sphere_capsule(c, r, OP.dims[0], OP.dims[1], dist, n);
        else sphere_ellipsoid(c, r, OP.dims, dist, n);
    }
    // ------------------------------------------------------------------------------------------------ hand-to-hand contact pairs
    // The asset lists its hand-to-hand contacts explicitly (MJCF <contact><pair>, shared.xml:31-51: 18 distinct geom pairs -- every finger's
    // distal link, and the first finger's other two and the middle finger's proximal one, against the thumb's distal link; the palm against it;
    // neighbouring distal / proximal links; four more around the little finger), all condim 1: frictionless.  The hand's shapes are contype 1 /
    // conaffinity 0, so these are its ONLY self contacts.  They are COMPLIANT contacts here (a soft row each, like the fixed tendons), not rows
    // of the Gauss-Seidel solve: for a pair whose shapes overlap by pen > 0 at the start of the sub-step (capsule axes: exact closest points;
    // the palm box against the thumb tip's capsule sampled by three spheres), EACH side s is pushed along its outward direction u_s
    // (u_a = n: from b towards a, u_b = -n) at the contact point by the implicit spring force
    //         F_s = k (pen - h J_s qd+),    J_s = u_s^T (velocity Jacobian of the side's body at the contact point),
    // i.e. H += h^2 k J_s^T J_s and the right-hand side gets J_s^T k (pen - h J_s qd): the side's own motion is implicit (unconditionally
    // stable), the other side is taken where the sub-step found it.  Why not rows: a pair couples the private coordinates of two fingers, which
    // in the finger-per-wave form live on different wavefronts whose LDS is full (637 of 640 floats per env) -- a compliant pair needs only the
    // two capsule axes exchanged once per sub-step and no solver state; the same model in every form (one wave, finger waves, CPU backend,
    // oracle/hand.c, oracle/hand.py), independent of the solver order.  k = pair_k (option "hand_pair_stiffness", N/m; 0 switches the pairs
    // off): 2e4 holds a finger at its drive's force limit (~10 N at the tip) 0.5 mm inside the other shape.
    float pair_k = 0.f;                                      // stiffness of the compliant pairs (HandView::pair_k); 0 = pairs off
    int pair_active = 0;                                     // out: pair sides this instance pushed in the last sub-step
    int pair_sens = 1;                                       // 0: skip the pairs' forces on the fingertip sensors (finger waves: every sub-step launch of a call but the last)
    static constexpr int NHP = M::NHP;
    // sphere (centre c in the box frame, radius r) vs box of half sizes a[3]: signed distance, outward normal (box frame)
    MI_HD static void sphere_box3(const float* c, float r, const float* a, float* dist, float* n) {
        float qd_[3], d[3], pn[3];
        sfor<3>([&](auto K) MI_LAMBDA { qd_[K] = fminf(fmaxf(c[K], -a[K]), a[K]); d[K] = c[K] - qd_[K]; pn[K] = a[K] - fabsf(c[K]); });
        const float d2 = dot3(d, d);
        const bool ix = (pn[0] <= pn[1]) && (pn[0] <= pn[2]), iy = !ix && (pn[1] <= pn[2]);
        const float pen = ix ? pn[0] : (iy ? pn[1] : pn[2]);
        const bool outside = d2 > 1e-24f;
        const float inv = MI_RSQ(fmaxf(d2, 1e-30f));
        *dist = outside ? d2 * inv - r : -pen - r;
        n[0] = outside ? d[0] * inv : (ix ? (c[0] >= 0.f ? 1.f : -1.f) : 0.f);
        n[1] = outside ? d[1] * inv : (iy ? (c[1] >= 0.f ? 1.f : -1.f) : 0.f);
        n[2] = outside ? d[2] * inv : ((!ix && !iy) ? (c[2] >= 0.f ? 1.f : -1.f) : 0.f);
    }
    // capsule pair P: world axis end points of both sides -> overlap pen (> 0: touching), n (from b towards a), contact point pc (middle of the overlap)
    template <int P>
    MI_HD static float pair_capsules(const float* a0, const float* a1, const float* b0, const float* b1, float* n, float* pc) {
        float ca[3], cb[3], dv[3];
        seg_seg_closest<false, false>(a0, a1, b0, b1, ca, cb);
        sfor<3>([&](auto K) MI_LAMBDA { dv[K] = ca[K] - cb[K]; });
        const float d2 = dot3(dv, dv);
        const bool ok = d2 > 1e-18f;
        const float inv = MI_RSQ(fmaxf(d2, 1e-30f));
        const float dist = d2 * inv - M::hp_ra[P] - M::hp_rb[P];
        n[0] = ok ? dv[0] * inv : 0.f; n[1] = ok ? dv[1] * inv : 0.f; n[2] = ok ? dv[2] * inv : 1.f;
        sfor<3>([&](auto K) MI_LAMBDA { pc[K] = cb[K] + n[K] * (M::hp_rb[P] + 0.5f * dist); });
        return -dist;
    }
    // box pair P (side a: the box of hp_a0 = centre, hp_a1 = half sizes in the frame Ra, ra of its body; side b: capsule with world end points):
    // the capsule sampled by the spheres at its two ends and its middle, the deepest one counts
    template <int P>
    MI_HD static float pair_box(const float* Ra, const float* ra, const float* b0, const float* b1, float* n, float* pc) {
        constexpr float ctr[3] = {M::hp_a0[P][0], M::hp_a0[P][1], M::hp_a0[P][2]}, half[3] = {M::hp_a1[P][0], M::hp_a1[P][1], M::hp_a1[P][2]};
        float best = 1e30f;
        sfor<3>([&](auto T) MI_LAMBDA {
            constexpr float t = 0.5f * (float)decltype(T)::value;
            float cw[3], rel[3], cl[3], nl[3], dist;
            sfor<3>([&](auto K) MI_LAMBDA { cw[K] = b0[K] + t * (b1[K] - b0[K]); rel[K] = cw[K] - ra[K]; });
            matTvec3(Ra, rel, cl);
            sfor<3>([&](auto K) MI_LAMBDA { cl[K] -= ctr[K]; });
            sphere_box3(cl, M::hp_rb[P], half, &dist, nl);
            if (dist < best) {
                float nw[3];
                matvec3(Ra, nl, nw);                        // from the box towards the sphere: side b's push direction, n = -nw
                best = dist;
                sfor<3>([&](auto K) MI_LAMBDA { n[K] = -nw[K]; pc[K] = cw[K] - nw[K] * (M::hp_rb[P] + 0.5f * dist); });
            }
        });
        return -best;
    }
    // one side of an active pair: body b pushed along u at pc (relative to O) -- the implicit spring's terms in H (before its factor) and in
    // the right-hand side.  S: the joints' motion subspaces about O, qd: joint velocities; H entries as L[midx[descendant][ancestor]].
    template <int b>
    MI_HD void pair_side(const float* pc, const float* u, const float pen, const float h, const float (&S)[M::NDA][6], float* L, float* y) const {
        constexpr int CL = M::chain_len[b];
        float W[6];
        cross3(pc, u, W);
        W[3] = u[0]; W[4] = u[1]; W[5] = u[2];
        float g[M::MAXCHAIN], vs = 0.f;
        sfor<CL>([&](auto C) MI_LAMBDA { g[C] = dot6(S[M::chain[b][C] - OFF], W); vs += g[C] * this->qd[M::chain[b][C] - OFF]; });
        const float kk = (pen > 0.f) ? pair_k : 0.f;
        const float a = h * h * kk, f = kk * (pen - h * vs);
        sfor<CL>([&](auto C1) MI_LAMBDA {
            constexpr int i = M::chain[b][C1];
            y[i] += g[C1] * f;
            const float ag = a * g[C1];
            sfor<CL - C1>([&](auto T) MI_LAMBDA { constexpr int c2 = C1 + T, j = M::chain[b][c2]; L[M::midx[i][j]] += ag * g[c2]; });
        });
    }
    // ---- the pairs on the force-sensor bodies (the fingertips; shadow_hand.py:291-297): a sensor sees every constraint force on its body, the hand's
    // own contacts included.  A pushed side's force is F_s = k (pen_s - h W_s . V+), W_s = [pc x u_s; u_s], V+ the body's twist about O once the solve
    // has the new joint velocities.  Per fingertip the pair phase leaves 8 numbers: A = sum k pen_s W_s (the static part of the wrench, exact for any
    // number of sides), P = sum pen_s and the number of sides n; the output phase applies the velocity part along the RESULTANT,
    //         wrench = A (1 - n h (A . V+) / (k P^2)),
    // which is k W (pen - h W . V+) exactly for one side -- what a fingertip almost always has -- and a stated approximation of the damping term when a
    // fingertip is pushed by two pairs at once (the thumb tip between two fingers); oracle/hand.c / hand.py state the same formula.  (The exact form for
    // any number of sides, A - M V+ with M = sum k h W_s W_s^T, was built first: 27 live registers per fingertip took the 64-env finger waves from 0 to
    // 448 B of scratch and ShadowHand@16384 from 0.1866 to 0.2010 ms per step.)
    MI_HD void pair_sensor_acc(const float* pc, const float* u, const float pen, const float h, float (&A)[8]) const {
        float W[6];
        cross3(pc, u, W);
        W[3] = u[0]; W[4] = u[1]; W[5] = u[2];
        const bool on = pen > 0.f;
        const float kp_ = on ? pair_k * pen : 0.f;
        sfor<6>([&](auto I) MI_LAMBDA { A[I] += kp_ * W[I]; });
        A[6] += on ? pen : 0.f;
        A[7] += on ? 1.f : 0.f;
        (void)h;
    }
    // the wrench (about O: torque, force) the pairs put on body b after the solve: vnew[gi] the new joint velocities
    template <int b>
    MI_HD void pair_sensor_wrench(const float (&A)[8], const float h, const float (&S)[M::NDA][6], const float* vnew, float (&wr)[6]) const {
        float V[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { constexpr int gi = M::chain[b][C]; sfor<6>([&](auto K) MI_LAMBDA { V[K] += S[gi - OFF][K] * vnew[gi]; }); });
        float av = 0.f;
        sfor<6>([&](auto I) MI_LAMBDA { av += A[I] * V[I]; });
        const float P2 = A[6] * A[6];
        const float fac = (P2 > 0.f) ? 1.f - A[7] * h * av * MI_RCP(pair_k * P2) : 0.f;
        sfor<6>([&](auto I) MI_LAMBDA { wr[I] = A[I] * fac; });
    }
    // the same in two halves for a form that cannot keep the motion subspaces alive until its output phase (the finger waves: S of a fingertip's chain is
    // 42 registers through the sweeps, 4 % of the step): g_c = A . S_c while S is there, then A . V+ = sum_c g_c v_c
    template <int b>
    MI_HD void pair_sensor_g(const float (&A)[8], const float (&S)[M::NDA][6], float (&g)[M::MAXCHAIN]) const {
        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA {
            constexpr int gi = M::chain[b][C];
            float s_ = 0.f;
            sfor<6>([&](auto K) MI_LAMBDA { s_ += A[K] * S[gi - OFF][K]; });
            g[C] = s_;
        });
    }
    template <int b>
    MI_HD void pair_sensor_wrench_g(const float (&A)[8], const float (&g)[M::MAXCHAIN], const float h, const float* vnew, float (&wr)[6]) const {
        float av = 0.f;
        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { av += g[C] * vnew[M::chain[b][C]]; });
        const float P2 = A[6] * A[6];
        const float fac = (P2 > 0.f) ? 1.f - A[7] * h * av * MI_RCP(pair_k * P2) : 0.f;
        sfor<6>([&](auto I) MI_LAMBDA { wr[I] = A[I] * fac; });
    }
#ifndef MI_HAND_PAIR_SENSORS
#define MI_HAND_PAIR_SENSORS 1          // 0: A/B builds without the pairs' forces in the fingertip sensors (tools/debug)
#endif
    static constexpr bool pair_sensor_body(int b) {      // a force-sensor body that takes part in a pair
        if (!MI_HAND_PAIR_SENSORS || sensor_of(b) < 0) return false;
        for (int p = 0; p < NHP; ++p) if (M::hp_ba[p] == b || M::hp_bb[p] == b) return true;
        return false;
    }
    // the segment of body b's pair capsule in its own frame (every pair a body takes part in uses the same shape of it)
    static constexpr bool hp_capsule_body(int b) {
        for (int p = 0; p < NHP; ++p) if ((M::hp_ba[p] == b && !M::hp_box[p]) || M::hp_bb[p] == b) return true;
        return false;
    }
    static constexpr float hp_seg(int b, int end, int k) {
        for (int p = 0; p < NHP; ++p) {
            if (M::hp_ba[p] == b && !M::hp_box[p]) return end ? M::hp_a1[p][k] : M::hp_a0[p][k];
            if (M::hp_bb[p] == b) return end ? M::hp_b1[p][k] : M::hp_b0[p][k];
        }
        return 0.f;
    }
    // world end points (relative to O) of body b's pair capsule from its pose
    template <int b>
    MI_HD static void hp_endpoints(const float* Rb, const float* rb, float* e0, float* e1) {
        constexpr float l0[3] = {hp_seg(b, 0, 0), hp_seg(b, 0, 1), hp_seg(b, 0, 2)}, l1[3] = {hp_seg(b, 1, 0), hp_seg(b, 1, 1), hp_seg(b, 1, 2)};
        float t0[3], t1[3];
        matvec3(Rb, l0, t0); matvec3(Rb, l1, t1);
        sfor<3>([&](auto K) MI_LAMBDA { e0[K] = rb[K] + t0[K]; e1[K] = rb[K] + t1[K]; });
    }

    // y = Ro diag(s) Ro^T x: a body-diagonal operator (inertia^{+-1/2}) applied to a world-frame vector
    MI_HD static void body_diag(const float* Ro, const float* s, const float* x, float* y) {
        float t[3];
        matTvec3(Ro, x, t);
        sfor<3>([&](auto K) MI_LAMBDA { t[K] *= s[K]; });
        matvec3(Ro, t, y);
    }

    // one sub-step of length h.  target[ND]: drive targets; laml: warm-start limit impulses; sensor: 6*NSENS fingertip
    // force/torque (body frame); dof_force[ND]; ncontact: number of object contacts taken | number refused for want of a slot << 16
    // SHAPE: OBJ_BOX (isotropic inertia OP.inertia, the benchmark configuration -- code unchanged) or OBJ_ELLIPSOID (OP.dims, OP.inertia3:
    // the angular part of the object's whitened velocity / rows goes through Ro diag(I^{+-1/2}) Ro^T; no gyroscopic torque, as PhysX's default)
    template <int RS, int SHAPE = OBJ_BOX>
    MI_HD void substep_hand(const SimParams& P, const ObjectParams& OP, const float* target, const float h, const RowStore<RS> rows,
                            const Strided laml, const Strided sensor, const Strided dof_force, int* ncontact) {
        constexpr int ST = RowStore<RS>::stride;
#if defined(MI_TIMING)
        unsigned long long* tstamp = this->tstamp;   // MI_STAMP (debug builds: tools/debug/phase_timing_live.py)
#endif
        float (&q)[M::NDA] = this->q;        // (dependent base: make the state names visible inside the generic lambdas)
        float (&qd)[M::NDA] = this->qd;
        float (&root)[13] = this->root;
        auto G = [&](int row, int c) MI_LAMBDA -> float& { return rows(B::limoff(row) + c); };
        auto Ainv = [&](int row) MI_LAMBDA -> float& { return rows(H_LIMG + row); };
        auto vt = [&](int row) MI_LAMBDA -> float& { return rows(H_LIMG + NLIM + row); };
        auto lam = [&](int row) MI_LAMBDA -> float& { return rows(H_LIMG + 2 * NLIM + row); };
        auto rho = [&](int row) MI_LAMBDA -> float& { return rows(H_LIMG + 3 * NLIM + row); };
        const bool clamp_on = any_clamped() && this->drive_clamp != 0;
        const float invh = MI_RCP(h);
        // `actor_params.hand.dof_properties.lower / upper` (ShadowHand.yaml:118-129): per-env shifts of the joint limits, [ND] lower then
        // [ND] upper, read once where the limit rows are built (the caller always provides them: zeros = the model's limits)
        const Strided limit_shift = this->limit_shift;
        typename B::Ctx c;
        float (&S)[M::NDA][6] = c.S;
        float (&L)[M::NM] = c.L;
        // pose (R 9, r 3) of the sphere-carrying bodies, handed from the tree pass to the narrow phase in per-lane memory (scratch),
        // not in LDS.  The opaque zero in every index keeps the array in memory: promoted to registers its 216 values would be
        // spilled one by one around the tree pass (601 instead of 360 spilled registers, the sub-step 1.5x slower).
        float pose[12 * (B::NOSB > 0 ? B::NOSB : 1)];
        int pz;
        MI_OPAQUE_ZERO(pz);
        // ------------------------------------------------------------ stage the limit impulses of the last sub-step
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            if constexpr (M::dof_limited[d]) lam(B::limrow(d)) = laml(d);
        });
        MI_STAMP(0);
        // ------------------------------------------------------------ tree pass with gravity off (disable_gravity on the hand)
        {
            SimParams P0 = P;
            P0.g[0] = P0.g[1] = P0.g[2] = 0.f;
            c.pose_out = pose + pz;
            c.pose_stride = 1;
            SpI Iroot;
            float Froot[6];
            this->template body_pass<0>(P0, c, nullptr, nullptr, nullptr, nullptr, Iroot, Froot);
        }
        MI_PHASE();
        MI_STAMP(1);
        // ------------------------------------------------------------ rhs: implicit PD drives, passive damping, tendons
        float Ldi[NVA], y[NVA];
        // `actor_params.hand` domain randomisation (ShadowHand.yaml:104-131): per-env factors of the link masses, the joints' damping and
        // drive stiffness and the tendons' limit stiffness / damping, loaded here where they are used (H and the bias forces are linear
        // in the link masses: the tree pass ran on the model's own).  this->actor_scale.p == nullptr: the model's constants.
        float sc_damp = 1.f, sc_kp = 1.f, sc_tk = 1.f, sc_td = 1.f;
        if (this->actor_scale.p != nullptr) {
            const float sc_mass = this->actor_scale(HS_MASS);
            sc_damp = this->actor_scale(HS_DAMPING); sc_kp = this->actor_scale(HS_STIFFNESS);
            sc_tk = this->actor_scale(HS_TENDON_STIFFNESS); sc_td = this->actor_scale(HS_TENDON_DAMPING);
            sfor<M::NM>([&](auto E_) MI_LAMBDA { L[E_] *= sc_mass; });
            sfor<NV>([&](auto I) MI_LAMBDA { c.bias[I] *= sc_mass; });
        }
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            const float kp = M::dof_kp[d] * sc_kp, Dm = M::dof_damping[d] * sc_damp;
            L[M::midx[gi][gi]] += M::dof_armature[d] + h * Dm + h * h * kp;
            y[gi] = -c.bias[gi] - kp * (q[d] - target[d]) - (Dm + h * kp) * qd[d];
        });
        sfor<M::NTEND>([&](auto T_) MI_LAMBDA {
            constexpr int t = T_, d0 = M::tend_d0[t], d1 = M::tend_d1[t], g0 = OFF + d0, g1 = OFF + d1;
            constexpr float c0 = M::tend_c0[t], c1 = M::tend_c1[t];
            static_assert(M::midx[g0][g1] >= 0 || M::midx[g1][g0] >= 0, "tendon joints must be on one kinematic chain");
            const float Lt = c0 * q[d0] + c1 * q[d1], Ld = c0 * qd[d0] + c1 * qd[d1];
            const float viol = Lt - fminf(fmaxf(Lt, M::tend_lo[t]), M::tend_hi[t]);
            const float k = (viol != 0.f) ? M::tend_stiffness * sc_tk : 0.f;
            const float td = M::tend_damping * sc_td;
            const float a = h * td + h * h * k;
            const float f = k * viol + (td + h * k) * Ld;
            L[M::midx[g0][g0]] += a * c0 * c0;
            L[M::midx[g1][g1]] += a * c1 * c1;
            if constexpr (M::midx[g0][g1] >= 0) L[M::midx[g0][g1]] += a * c0 * c1; else L[M::midx[g1][g0]] += a * c0 * c1;
            y[g0] -= c0 * f;
            y[g1] -= c1 * f;
        });
        // ------------------------------------------------------------ the asset's hand-to-hand pairs: compliant contacts (pair_side)
        this->pair_active = 0;
        float psPN[M::NSENSA][2];                               // the pairs on the fingertips: sum of pen, sides (pair_sensor_acc)
        sfor<NSENS>([&](auto K_) MI_LAMBDA { psPN[K_][0] = 0.f; psPN[K_][1] = 0.f; });
        const bool pairs_on = (NHP > 0) && MI_WAVE_ANY(this->pair_k > 0.f);
        if constexpr (NHP > 0) {
            if (pairs_on) {
                int npa = 0;
                float psA[M::NSENSA][8];                        // ... and their wrench A x6 (parked in the `sensor` tensor below)
                sfor<NSENS>([&](auto K_) MI_LAMBDA { sfor<8>([&](auto I) MI_LAMBDA { psA[K_][I] = 0.f; }); });
                sfor<NHP>([&](auto P_) MI_LAMBDA {
                    constexpr int pp = P_, ba = M::hp_ba[pp], bb = M::hp_bb[pp];
                    static_assert(B::os_count(ba) > 0 && B::os_count(bb) > 0, "pair bodies carry object spheres: their poses are in the pose array");
                    MI_PHASE();
                    float Ra[9], ra[3], Rb2[9], rb2[3];
                    sfor<9>([&](auto I_) MI_LAMBDA { Ra[I_] = pose[pz + 12 * B::os_slot(ba) + I_]; Rb2[I_] = pose[pz + 12 * B::os_slot(bb) + I_]; });
                    sfor<3>([&](auto I_) MI_LAMBDA { ra[I_] = pose[pz + 12 * B::os_slot(ba) + 9 + I_]; rb2[I_] = pose[pz + 12 * B::os_slot(bb) + 9 + I_]; });
                    float b0[3], b1[3], n[3], pc[3], pen;
                    hp_endpoints<bb>(Rb2, rb2, b0, b1);
                    if constexpr (M::hp_box[pp]) pen = pair_box<pp>(Ra, ra, b0, b1, n, pc);
                    else { float a0[3], a1[3]; hp_endpoints<ba>(Ra, ra, a0, a1); pen = pair_capsules<pp>(a0, a1, b0, b1, n, pc); }
                    const bool on = (pen > 0.f) && (this->pair_k > 0.f);
                    if (MI_WAVE_ANY(on)) {
                        const float pe = on ? pen : 0.f;
                        const float nm[3] = {-n[0], -n[1], -n[2]};
                        this->template pair_side<ba>(pc, n, pe, h, S, L, y);
                        this->template pair_side<bb>(pc, nm, pe, h, S, L, y);
                        if constexpr (pair_sensor_body(ba)) this->pair_sensor_acc(pc, n, pe, h, psA[sensor_of(ba)]);
                        if constexpr (pair_sensor_body(bb)) this->pair_sensor_acc(pc, nm, pe, h, psA[sensor_of(bb)]);
                        npa += on ? 2 : 0;
                    }
                });
                this->pair_active = npa;
                // the six numbers of A wait in the fingertip's own slots of the `sensor` tensor until the output phase (this lane rewrites them there; a
                // load after the own store of the same address sees it): kept in registers through the solve they took this form from 232 to 426
                // spilled VGPRs; P and n stay
                sfor<NSENS>([&](auto K_) MI_LAMBDA {
                    if constexpr (pair_sensor_body(M::sens_body[K_])) {
                        psPN[K_][0] = psA[K_][6]; psPN[K_][1] = psA[K_][7];
                        if (MI_WAVE_ANY(psA[K_][6] > 0.f)) sfor<6>([&](auto C) MI_LAMBDA { sensor(6 * K_ + C) = psA[K_][C]; });
                    }
                });
            }
        }
        MI_PHASE();
        // ------------------------------------------------------------ H = L^T L
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
        // ------------------------------------------------------------ whitened velocities: hand w = L qd + h L^-T rhs, object wo
        float w[NVA];
        sfor_rev<NV>([&](auto I_) MI_LAMBDA {
            constexpr int i = I_;
            const float z = y[i] * Ldi[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { y[M::anc[i][A_]] -= L[M::midx[i][M::anc[i][A_]]] * z; });
            float s = L[M::midx[i][i]] * qd[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { s += L[M::midx[i][M::anc[i][A_]]] * qd[M::anc[i][A_]]; });
            w[i] = s + h * z;
        });
        const float sm = MI_SQRT(OP.mass), si = MI_SQRT(SHAPE == OBJ_BOX ? OP.inertia : 1.f);
        const float ism = MI_RCP(sm), isi = MI_RCP(si);
        float wo[6];
        sfor<3>([&](auto K) MI_LAMBDA { wo[K] = sm * (obj.vel[K] + h * (P.g[K] + OP.fw[K] * (ism * ism))); wo[3 + K] = si * obj.angvel[K]; });
        float Ro[9];
        quat2mat(obj.quat, Ro);
        float isqI[3] = {1.f, 1.f, 1.f};     // SHAPE != OBJ_BOX: 1 / sqrt of the principal inertias
        if constexpr (SHAPE != OBJ_BOX) {
            float sqI[3];
            sfor<3>([&](auto K) MI_LAMBDA { sqI[K] = MI_SQRT(OP.inertia3[K]); isqI[K] = MI_RCP(sqI[K]); });
            body_diag(Ro, sqI, obj.angvel, wo + 3);
        }
        const float xo[3] = {obj.pos[0] - root[0], obj.pos[1] - root[1], obj.pos[2] - root[2]};   // object COM rel O
        MI_PHASE();
        MI_STAMP(2);
        // ------------------------------------------------------------ joint limit rows (as core/engine.hpp)
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (M::dof_limited[d]) {
                constexpr int row = B::limrow(d);
                MI_PHASE();
                const float dl = q[d] - (M::dof_lower[d] + limit_shift(d)), du = (M::dof_upper[d] + limit_shift(ND + d)) - q[d];
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
                float a = P.cfm;
                sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { a += g[K] * g[K]; G(row, K) = g[K]; });
                Ainv(row) = MI_RCP(a);
                vt(row) = (C >= 0.f) ? -C * invh : fminf(-C * P.erp * invh, P.max_depen_vel);
                lam(row) = l0;
                if constexpr (clamped(d)) rho(row) = 0.f;
            }
        });
        MI_PHASE();
        MI_STAMP(3);
        // ------------------------------------------------------------ object contacts -> compact slots
        // Static over the sphere-carrying bodies (the kinematic chain of a row is compile-time), a run-time loop over the
        // spheres of one body with their positions / radii read from the constant tables (scalar loads): one copy of the
        // narrow phase + row build per BODY, not per sphere -- the per-sphere unrolled version had ~1000 distinct literals
        // in SGPRs, spilled 472 of them and miscomputed on gfx950 ("compiler regime").
        int cnt = 0, refused = 0;
        sfor<NB>([&](auto B_) MI_LAMBDA {
            constexpr int b = B_;
            if constexpr (B::os_count(b) > 0) {
                constexpr int CL = M::chain_len[b];
                // (the table look-ups as constant expressions: left to the optimiser the helpers' loops over the model's sphere table ran at
                //  RUN time for the Allegro hand's 119 spheres -- in the loop condition, in every pose index -- and its sub-step took 5.3 ms)
                constexpr int OS_N = B::os_count(b), OS_0 = B::os_first(b), OS_SLOT = B::os_slot(b);
                MI_PHASE();
                float Rb[9], rb[3];
                sfor<9>([&](auto I_) MI_LAMBDA { Rb[I_] = pose[pz + 12 * OS_SLOT + I_]; });
                sfor<3>([&](auto I_) MI_LAMBDA { rb[I_] = pose[pz + 12 * OS_SLOT + 9 + I_]; });
                int nbody = 0;
                const int first = cnt;
                for (int i = 0; i < OS_N; ++i) {
                    const int s = OS_0 + i;
                    const float pl[3] = {M::os_pos[s][0], M::os_pos[s][1], M::os_pos[s][2]};
                    const float rad = M::os_rad[s];
                    float t[3], cs[3];
                    matvec3(Rb, pl, t);
                    sfor<3>([&](auto K) MI_LAMBDA { cs[K] = rb[K] + t[K]; });
                    const float rel[3] = {cs[0] - xo[0], cs[1] - xo[1], cs[2] - xo[2]};
                    float cl[3], nl[3], dist;
                    matTvec3(Ro, rel, cl);
                    sphere_object<SHAPE>(cl, rad, OP, &dist, nl);
                    // contact manifold: at most BODY_CAP contacts per hand body, taken in the body's (spread-out, farthest-point) sphere
                    // order, so that a cube lying on the 30-sphere palm cannot use up all KMAX slots before the fingers are looked at
                    const bool on = (dist < P.contact_offset) && (cnt < KMAX) && (nbody < BODY_CAP);
                    refused += ((dist < P.contact_offset) && (nbody < BODY_CAP) && (cnt >= KMAX)) ? 1 : 0;   // all KMAX slots taken
                    nbody += on ? 1 : 0;
                    const int j = on ? cnt : -1;
                    if (on) {      // narrow phase only: the contact's geometry goes into its slot, the rows are built below
                        float n[3], rc[3];
                        matvec3(Ro, nl, n);                     // from the object towards the sphere
                        sfor<3>([&](auto K) MI_LAMBDA { rc[K] = (cs[K] - rad * n[K]) - xo[K]; });
                        float* cb = rows.ptr(H_CB + j * H_CSZ);
                        const float gap = dist - P.rest_offset;
                        sfor<3>([&](auto I_) MI_LAMBDA { cb[(H_GEO + I_) * ST] = n[I_]; cb[(H_GEO + 3 + I_) * ST] = rc[I_]; });
                        cb[(H_AUX + 3) * ST] = (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
                    }
                    cnt += on ? 1 : 0;
                }
                // rows of this body's contacts, walked by contact (at most BODY_CAP, left by the whole wave as soon as no env has an
                // (i+1)-th one) instead of being built inside the sphere loop: one wave executes the UNION of its envs' work, and the
                // union of touched spheres of a body is several times larger than the largest per-env contact count on it
                for (int i = 0; i < BODY_CAP; ++i) {
                    if (!MI_WAVE_ANY(i < nbody)) break;
                    if (i < nbody) {
                        float* cb = rows.ptr(H_CB + (first + i) * H_CSZ);
                        float fr[3][3], rc[3], pc[3];
                        sfor<3>([&](auto I_) MI_LAMBDA { fr[0][I_] = cb[(H_GEO + I_) * ST]; rc[I_] = cb[(H_GEO + 3 + I_) * ST]; pc[I_] = rc[I_] + xo[I_]; });
                        contact_frame(fr[0], fr[1], fr[2]);
                        sfor<3>([&](auto K) MI_LAMBDA {
                            constexpr int k = K;
                            float W[6];
                            cross3(pc, fr[k], W);
                            W[3] = fr[k][0]; W[4] = fr[k][1]; W[5] = fr[k][2];
                            float g[HCH + 6];
                            sfor<CL>([&](auto C) MI_LAMBDA { g[C] = dot6(S[M::chain[b][C] - OFF], W); });
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
                            // object part: J_o = -[u; rc x u], whitened by the constant diagonal
                            float cx[3];
                            cross3(rc, fr[k], cx);
                            if constexpr (SHAPE != OBJ_BOX) { float cw[3]; body_diag(Ro, isqI, cx, cw); sfor<3>([&](auto I_) MI_LAMBDA { cx[I_] = cw[I_]; }); }
                            sfor<3>([&](auto I_) MI_LAMBDA { g[CL + I_] = -fr[k][I_] * ism; g[CL + 3 + I_] = -cx[I_] * isi; });
                            float a = P.cfm;
                            sfor<CL + 6>([&](auto C) MI_LAMBDA { a += g[C] * g[C]; });
                            sfor<CL>([&](auto C) MI_LAMBDA { cb[(k * HCH + C) * ST] = g[C]; });
                            cb[(H_AUX + k) * ST] = MI_RCP(a);
                            cb[(H_AUX + 4 + k) * ST] = 0.f;    // no warm start for object contacts
                        });
                    }
                }
                rows(H_BODYSLOT + OS_SLOT) = __builtin_bit_cast(float, first | (nbody << 8));
            }
        });
        *ncontact = cnt | (refused << 16);
        MI_PHASE();
        MI_STAMP(4);
        // ------------------------------------------------------------ warm start (limit rows only)
        {
            int zero;
            MI_OPAQUE_ZERO(zero);
            const RowStore<RS> rit = rows.shifted(zero);
            sfor<ND>([&](auto D) MI_LAMBDA {
                constexpr int d = D, gi = OFF + d;
                if constexpr (M::dof_limited[d]) {
                    constexpr int row = B::limrow(d), g0 = B::limoff(row);
                    const float l0 = rit(H_LIMG + 2 * NLIM + row);
                    w[gi] += rit(g0) * l0;
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += rit(g0 + 1 + A_) * l0; });
                }
            });
        }
        MI_PHASE();
        MI_STAMP(5);
        // ------------------------------------------------------------ projected Gauss-Seidel sweeps
        float cl_kp = 1.f, cl_damp = 1.f;     // `actor_params` factors of the drive gains, for the drive clamps (re-loaded here)
        if (clamp_on && this->actor_scale.p != nullptr) { cl_damp = this->actor_scale(HS_DAMPING); cl_kp = this->actor_scale(HS_STIFFNESS); }
        for (int it = 0; it < P.iters; ++it) {
            int zero;
            MI_OPAQUE_ZERO(zero);
            const RowStore<RS> rit = rows.shifted(zero);
            sfor<ND>([&](auto D) MI_LAMBDA {
                constexpr int d = D, gi = OFF + d;
                if constexpr (M::dof_limited[d]) {
                    constexpr int row = B::limrow(d), g0 = B::limoff(row);
                    float g[M::MAXCHAIN];
                    sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { g[K] = rit(g0 + K); });
                    float vn = g[0] * w[gi];
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { vn += g[1 + A_] * w[M::anc[gi][A_]]; });
                    if constexpr (clamped(d)) {
                        if (clamp_on) {      // the dof's drive clamp, ahead of its limit row: g = s L^-1 e_d, v_d = s vn
                            const float kp_ = M::dof_kp[d] * cl_kp, c_ = M::dof_damping[d] * cl_damp + h * kp_;
                            const float sg = (g[0] > 0.f) ? 1.f : -1.f, a_ = MI_RCP(rit(H_LIMG + row)) - P.cfm;
                            float r_ = rit(H_LIMG + 3 * NLIM + row);
                            const float dr = sg * drive_clamp_update(-kp_ * (q[d] - target[d]), c_, M::dof_force_limit[d], invh, sg * vn,
                                                                     (rit(H_LIMG + 2 * NLIM + row) > 0.f) ? 0.f : a_, r_);
                            rit(H_LIMG + 3 * NLIM + row) = r_;
                            w[gi] += g[0] * dr;
                            sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += g[1 + A_] * dr; });
                            vn += a_ * dr;           // g . g = a
                        }
                    }
                    const float lo = rit(H_LIMG + 2 * NLIM + row);
                    const float nl_ = fmaxf(lo - (vn - rit(H_LIMG + NLIM + row)) * rit(H_LIMG + row), 0.f);
                    const float dl = nl_ - lo;
                    rit(H_LIMG + 2 * NLIM + row) = nl_;
                    w[gi] += g[0] * dl;
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += g[1 + A_] * dl; });
                }
            });
            sfor<NB>([&](auto B_) MI_LAMBDA {
                constexpr int b = B_;
                if constexpr (B::os_count(b) > 0) {
                    constexpr int CL = M::chain_len[b];
                    constexpr int OS_SLOT = B::os_slot(b);
                    const int fc = __builtin_bit_cast(int, rit(H_BODYSLOT + OS_SLOT));
                    const int first = fc & 255, nb_ = fc >> 8;
                    for (int i = 0; i < BODY_CAP; ++i) {
                        if (!MI_WAVE_ANY(i < nb_)) break;           // no env of the wave has an (i+1)-th contact on this body
                        if (i < nb_) {
                            float* cb = rit.ptr(H_CB + (first + i) * H_CSZ);
                            float g[3][HCH + 6], ainv[3], lm[3];
                            sfor<3>([&](auto K) MI_LAMBDA {
                                sfor<CL>([&](auto C) MI_LAMBDA { g[K][C] = cb[(K * HCH + C) * ST]; });
                                ainv[K] = cb[(H_AUX + K) * ST];
                                lm[K] = cb[(H_AUX + 4 + K) * ST];
                            });
                            {   // object part of the three rows from the stored normal and lever (as in the row build)
                                float fr[3][3], rc[3];
                                sfor<3>([&](auto I_) MI_LAMBDA { fr[0][I_] = cb[(H_GEO + I_) * ST]; rc[I_] = cb[(H_GEO + 3 + I_) * ST]; });
                                contact_frame(fr[0], fr[1], fr[2]);
                                sfor<3>([&](auto K) MI_LAMBDA {
                                    float cx[3];
                                    cross3(rc, fr[K], cx);
                                    if constexpr (SHAPE != OBJ_BOX) { float cw[3]; body_diag(Ro, isqI, cx, cw); sfor<3>([&](auto I_) MI_LAMBDA { cx[I_] = cw[I_]; }); }
                                    sfor<3>([&](auto I_) MI_LAMBDA { g[K][CL + I_] = -fr[K][I_]; g[K][CL + 3 + I_] = -cx[I_]; });   // whitening scales: below
                                });
                            }
                            const float vtn = cb[(H_AUX + 3) * ST];
                            auto rowvel = [&](int k) MI_LAMBDA {
                                float vn = 0.f;
                                sfor<CL>([&](auto C) MI_LAMBDA { vn += g[k][C] * w[M::chain[b][C]]; });
                                float vl = 0.f, va = 0.f;
                                sfor<3>([&](auto C) MI_LAMBDA { vl += g[k][CL + C] * wo[C]; va += g[k][CL + 3 + C] * wo[3 + C]; });
                                return vn + (ism * vl + isi * va);
                            };
                            auto apply = [&](int k, float dl) MI_LAMBDA {
                                sfor<CL>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += g[k][C] * dl; });
                                const float dll = ism * dl, dla = isi * dl;
                                sfor<3>([&](auto C) MI_LAMBDA { wo[C] += g[k][CL + C] * dll; wo[3 + C] += g[k][CL + 3 + C] * dla; });
                            };
                            const float ln = fmaxf(lm[0] - (rowvel(0) - vtn) * ainv[0], 0.f);
                            apply(0, ln - lm[0]);
                            float lt[2];
                            // both tangent rows from the SAME velocity, the disc projection, ONE application (round 6: as core/scene_engine.hpp; until
                            // then t1 was solved and applied before t2 was looked at -- a fast-sliding contact's friction pointed off the sliding direction)
                            float vtg[2];
                            sfor<2>([&](auto K) MI_LAMBDA { vtg[K] = rowvel(1 + K); lt[K] = lm[1 + K] - vtg[K] * ainv[1 + K]; });
                            friction_disc(lt, lm[1], lm[2], vtg[0], vtg[1], ainv[1], ainv[2], OP.mu * ln);
                            cb[(H_AUX + 4) * ST] = ln;
                            sfor<2>([&](auto K) MI_LAMBDA {
                                const float nl_ = lt[K];
                                cb[(H_AUX + 5 + K) * ST] = nl_;
                                apply(1 + K, nl_ - lm[1 + K]);
                            });
                        }
                    }
                }
            });
        }
        MI_PHASE();
        MI_STAMP(6);
        // ------------------------------------------------------------ back to generalised velocity
        float v[NVA];
        sfor<NV>([&](auto I_) MI_LAMBDA {
            constexpr int i = I_;
            float s = w[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { s -= L[M::midx[i][M::anc[i][A_]]] * v[M::anc[i][A_]]; });
            v[i] = s * Ldi[i];
        });
        // ------------------------------------------------------------ outputs: limit impulses, dof forces, fingertip sensors
        float fs_kp = 1.f, fs_damp = 1.f;     // (re-loaded: not kept live through the contact phases)
        if (this->actor_scale.p != nullptr) { fs_damp = this->actor_scale(HS_DAMPING); fs_kp = this->actor_scale(HS_STIFFNESS); }
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            float ll = 0.f;
            if constexpr (M::dof_limited[d]) {
                constexpr int row = B::limrow(d);
                // which limit the row was built for: G(row, 0) = s / L_dd, s = +1 lower, -1 upper
                ll = (G(row, 0) > 0.f) ? lam(row) : -lam(row);
            }
            laml(d) = ll;
            float df = -M::dof_kp[d] * fs_kp * (q[d] - target[d]) - M::dof_damping[d] * fs_damp * v[OFF + d] + ll * invh;
            if constexpr (clamped(d)) {      // a force-limited drive reports the end-of-step force the clamp acts on: fa - c v + rho / h (+- fmax when saturated)
                if (clamp_on) df += rho(B::limrow(d)) * invh - h * M::dof_kp[d] * fs_kp * v[OFF + d];
            }
            dof_force(d) = df;
        });
        float sens[6 * M::NSENSA];
        sfor<6 * NSENS>([&](auto K) MI_LAMBDA { sens[K] = 0.f; });
        sfor<NB>([&](auto B_) MI_LAMBDA {
            constexpr int b = B_;
            if constexpr (sensor_of(b) >= 0 && B::os_count(b) > 0) {
                constexpr int k = sensor_of(b);
                // the fingertip's pose from the per-lane pose array (keeping the tree pass's own sensor frames c.Rs / c.rs alive
                // until here costs 100 more spilled registers)
                float Rb[9], rb[3];
                constexpr int OS_SLOT = B::os_slot(b);
                sfor<9>([&](auto I_) MI_LAMBDA { Rb[I_] = pose[pz + 12 * OS_SLOT + I_]; });
                sfor<3>([&](auto I_) MI_LAMBDA { rb[I_] = pose[pz + 12 * OS_SLOT + 9 + I_]; });
                const int fc = __builtin_bit_cast(int, rows(H_BODYSLOT + OS_SLOT));
                const int first = fc & 255, nb_ = fc >> 8;
                for (int i = 0; i < BODY_CAP; ++i) {
                    if (!MI_WAVE_ANY(i < nb_)) break;
                    if (i < nb_) {
                        const float* cb = rows.ptr(H_CB + (first + i) * H_CSZ);
                        const float ln = cb[(H_AUX + 4) * ST], l1 = cb[(H_AUX + 5) * ST], l2 = cb[(H_AUX + 6) * ST];
                        // contact frame and point from the slot (neither body has moved yet): n, lever rc = point - object COM
                        float n[3], t1[3], t2[3], rc[3];
                        sfor<3>([&](auto K) MI_LAMBDA { n[K] = cb[(H_GEO + K) * ST]; rc[K] = cb[(H_GEO + 3 + K) * ST]; });
                        contact_frame(n, t1, t2);
                        float f[3], arm[3], tq[3], fl[3], tl[3];
                        sfor<3>([&](auto K) MI_LAMBDA {
                            f[K] = (n[K] * ln + t1[K] * l1 + t2[K] * l2) * invh;
                            arm[K] = (rc[K] + xo[K]) - rb[K];
                        });
                        cross3(arm, f, tq);
                        matTvec3(Rb, f, fl); matTvec3(Rb, tq, tl);
                        sfor<3>([&](auto C) MI_LAMBDA { sens[6 * k + C] += fl[C]; sens[6 * k + 3 + C] += tl[C]; });
                    }
                }
            }
        });
        if constexpr (NHP > 0) {
            if (pairs_on) {
                sfor<NB>([&](auto B_) MI_LAMBDA {
                    constexpr int b = B_;
                    if constexpr (pair_sensor_body(b) && B::os_count(b) > 0) {
                        constexpr int k = sensor_of(b), OS_SLOT = B::os_slot(b);
                        float Rb[9], rb[3], wr[6], tq[3], fl[3], tl[3];
                        sfor<9>([&](auto I_) MI_LAMBDA { Rb[I_] = pose[pz + 12 * OS_SLOT + I_]; });
                        sfor<3>([&](auto I_) MI_LAMBDA { rb[I_] = pose[pz + 12 * OS_SLOT + 9 + I_]; });
                        float A8[8];
                        const bool any = psPN[k][0] > 0.f;
                        sfor<6>([&](auto C) MI_LAMBDA { const float a_ = sensor(6 * k + C); A8[C] = any ? a_ : 0.f; });       // (parked there by the pair phase)
                        A8[6] = psPN[k][0]; A8[7] = psPN[k][1];
                        this->template pair_sensor_wrench<b>(A8, h, S, v, wr);
                        const float f[3] = {wr[3], wr[4], wr[5]};
                        cross3(rb, f, tq);                                   // torque about the sensor origin: tau_O - rb x f
                        sfor<3>([&](auto C) MI_LAMBDA { tq[C] = wr[C] - tq[C]; });
                        matTvec3(Rb, f, fl); matTvec3(Rb, tq, tl);
                        sfor<3>([&](auto C) MI_LAMBDA { sens[6 * k + C] += fl[C]; sens[6 * k + 3 + C] += tl[C]; });
                    }
                });
            }
        }
        sfor<6 * NSENS>([&](auto K) MI_LAMBDA { sensor(K) = sens[K]; });
        MI_PHASE();
        MI_STAMP(7);
        // ------------------------------------------------------------ integrate hand and object (semi-implicit Euler)
        sfor<ND>([&](auto D) MI_LAMBDA { qd[D] = v[OFF + D]; q[D] += h * qd[D]; });
        sfor<3>([&](auto K) MI_LAMBDA {
            obj.vel[K] = wo[K] * ism; obj.angvel[K] = wo[3 + K] * isi;
            obj.pos[K] += h * obj.vel[K];
        });
        if constexpr (SHAPE != OBJ_BOX) { float om[3]; body_diag(Ro, isqI, wo + 3, om); sfor<3>([&](auto K) MI_LAMBDA { obj.angvel[K] = om[K]; }); }
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
        MI_STAMP(8);
    }

    // world pose and velocity of the force-sensor (fingertip) bodies at the CURRENT state: [NSENS][13] = pos3, quat xyzw,
    // linvel3, angvel3 -- what gym.refresh_rigid_body_state_tensor exposes (shadow_hand.py:440,456-457)
    MI_HD void fingertip_states(float (*out)[13]) {
        sfor<NSENS>([&](auto K_) MI_LAMBDA { this->template fingertip_state<decltype(K_)::value>(out[K_]); });
    }
    // one fingertip: reads q / qd of the dofs on its chain only
    template <int k>
    MI_HD void fingertip_state(float* o) {
        {
            constexpr int tip = M::sens_body[k];
            // walk the chain root -> tip (static): bodies on the path, in order
            float Rb[9], rb[3] = {0.f, 0.f, 0.f}, om[3] = {0.f, 0.f, 0.f}, vl[3] = {0.f, 0.f, 0.f};
            quat2mat(this->root + 3, Rb);
            sfor<NB>([&](auto B_) MI_LAMBDA {
                constexpr int b = B_;
                if constexpr (b > 0 && is_ancestor_or_self(b, tip)) {
                    // parent frame -> body frame (same steps as body_pass)
                    float t[3];
                    matvec3(Rb, M::bpos[b], t);
                    // velocity of the new origin: v += om x t
                    float cx[3];
                    cross3(om, t, cx);
                    sfor<3>([&](auto I_) MI_LAMBDA { rb[I_] += t[I_]; vl[I_] += cx[I_]; });
                    if constexpr (!B::brot_is_identity(b)) matmul3(Rb, M::brot[b], Rb);
                    sfor<M::body_ndof[b]>([&](auto J_) MI_LAMBDA {
                        constexpr int d = M::body_dof0[b] + J_;
                        constexpr float ax = M::dof_axis[d][0], ay = M::dof_axis[d][1], az = M::dof_axis[d][2];
                        const float al[3] = {ax, ay, az}, anl[3] = {M::dof_anchor[d][0], M::dof_anchor[d][1], M::dof_anchor[d][2]};
                        float a[3], ta[3];
                        matvec3(Rb, al, a);
                        matvec3(Rb, anl, ta);
                        static_assert(M::dof_type[d] == 0, "fingertip chains are hinge-only");
                        float s_, c_;
                        sincosf(this->q[d], &s_, &c_);
                        const float tt = 1.f - c_;
                        const float Q[9] = {c_ + ax * ax * tt, ax * ay * tt - az * s_, ax * az * tt + ay * s_,
                                            ay * ax * tt + az * s_, c_ + ay * ay * tt, ay * az * tt - ax * s_,
                                            az * ax * tt - ay * s_, az * ay * tt + ax * s_, c_ + az * az * tt};
                        matmul3(Rb, Q, Rb);
                        float tb[3];
                        matvec3(Rb, anl, tb);
                        // the body origin moves on a circle about the anchor: r = pt - tb ; its velocity picks up the joint
                        // rate about the anchor
                        const float dr[3] = {ta[0] - tb[0], ta[1] - tb[1], ta[2] - tb[2]};
                        float c1[3], c2[3];
                        cross3(om, dr, c1);
                        const float wj[3] = {a[0] * this->qd[d], a[1] * this->qd[d], a[2] * this->qd[d]};
                        const float mtb[3] = {-tb[0], -tb[1], -tb[2]};
                        cross3(wj, mtb, c2);
                        sfor<3>([&](auto I_) MI_LAMBDA { rb[I_] += dr[I_]; vl[I_] += c1[I_] + c2[I_]; om[I_] += wj[I_]; });
                    });
                }
            });
            sfor<3>([&](auto I_) MI_LAMBDA { o[I_] = this->root[I_] + rb[I_]; o[7 + I_] = vl[I_]; o[10 + I_] = om[I_]; });
            mat2quat(Rb, o + 3);
        }
    }
    static constexpr bool is_ancestor_or_self(int a, int b) {
        while (b >= 0) { if (b == a) return true; b = M::parent[b]; }
        return false;
    }
    // rotation matrix -> quaternion xyzw (Shepperd), w >= 0 branch first
    MI_HD static void mat2quat(const float* R, float* qo) {
        const float tr = R[0] + R[4] + R[8];
        float x, y, z, w;
        if (tr > 0.f) {
            const float s = sqrtf(tr + 1.f) * 2.f;
            w = 0.25f * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
        } else if (R[0] > R[4] && R[0] > R[8]) {
            const float s = sqrtf(1.f + R[0] - R[4] - R[8]) * 2.f;
            w = (R[7] - R[5]) / s; x = 0.25f * s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
        } else if (R[4] > R[8]) {
            const float s = sqrtf(1.f + R[4] - R[0] - R[8]) * 2.f;
            w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = 0.25f * s; z = (R[5] + R[7]) / s;
        } else {
            const float s = sqrtf(1.f + R[8] - R[0] - R[4]) * 2.f;
            w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = 0.25f * s;
        }
        qo[0] = x; qo[1] = y; qo[2] = z; qo[3] = w;
    }
};

}  // namespace mi
