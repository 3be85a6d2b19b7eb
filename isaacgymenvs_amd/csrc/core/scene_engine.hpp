// scene_engine.hpp -- physics sub-step of a fixed-base articulated actor in a SCENE: free rigid boxes and static boxes beside it.
//
// Replaces gym.simulate() for the reference's table-top manipulation tasks, whose envs hold more than one actor: reference
// isaacgymenvs/tasks/franka_cube_stack.py:204-233,323-339 (the Franka arm, a table and its stand -- static boxes, gym.create_box with
// fix_base_link -- and two free cubes).  Built on the pieces of core/engine.hpp (tree pass, branch-sparse L^T L factor, whitened PGS) the way
// core/hand_engine.hpp is; what is specific here:
//   * up to kSceneMaxFree free boxes (three half sizes, principal inertias along the box axes, gravity on) and kSceneMaxStatic static boxes
//     (pose + half sizes), both RUN-TIME parameters (SceneParams = MiScene of include/mi_engine.h): one compiled robot serves any scene;
//   * the actor's own gravity follows asset option disable_gravity (franka_cube_stack.py:199), the boxes' follows the sim;
//   * contacts, every one 3 rows (normal + friction disc) of ONE Gauss-Seidel sequence with the actor's joint-limit rows:
//       actor sphere (the model's collision spheres, meshes sampled by assets/mesh.py) vs free box / static box   [actor chain | 6 box dofs]
//       free box corner vs ground plane / static box / other free box (both directions)                            [6 | 6 box dofs]
//       static box corner vs free box                                                                               [6 box dofs]
//       free box edge vs free / static box edge (round 6; scene_box_edge: one contact per pair whose least-penetration axis is edge x edge)
//       outline of the incident face vs outline of the reference face of a free box and a free / static box (round 6; scene_face_crossings)
//     box-box and box-static manifolds are the CORNER-IN-BOX contacts of both boxes (exact signed distance of a point to a box) -- face-face
//     and corner-face configurations (a cube on a table, a cube stacked on a cube, a cube pushed against a cube, a plate on a smaller stand) --
//     plus the edge-edge contact of crossed edges and the outline crossings of a face contact (crossed planks, a plank on a knife edge).  A stated approximation like every contact model here: physics parity against PhysX is
//     unpinned.
//   * contact slots are data dependent: at most KARM actor contacts (taken in sphere order, grouped by actor body) and KBOX box contacts
//     (box order, corner order, then ground / static boxes / free boxes); refusals are counted.
//   * contacts are WARM STARTED although their slots are data dependent: a contact is identified by its feature (actor sphere x target, or box
//     corner x target -- static numbers), the sub-step leaves (feature, impulses) of every slot in tensor scene_warm, the next one looks its
//     features up there (<= 48 compares per contact) and starts each contact's sweep from last sub-step's impulses: a cube at rest, a stack, a
//     cube held between two fingers start every solve from the force they already carry.
//   * free boxes live in plain velocity space (their mass matrix is block diagonal: v += M^-1 J^T dl costs 2 cross products), the actor
//     in the whitened space of its factor.
// Same maths as oracle/scene.py (dense, numpy, fp64).
#pragma once
#include "engine.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
#define MI_OPAQUE_VI(x) asm volatile("" : "+v"(x))          // a per-lane integer the optimiser cannot see through
#else
#define MI_OPAQUE_VI(x) do { } while (0)
#endif

namespace mi {

constexpr int kSceneMaxFree = 4, kSceneMaxStatic = 4;
struct SceneParams {        // mirrors MiScene (include/mi_engine.h)
    int n_free, n_static;
    int arm_gravity;        // 0: asset option disable_gravity on the articulated actor
    int pad;
    float free_half[kSceneMaxFree][3], free_mass[kSceneMaxFree], free_inertia[kSceneMaxFree][3], free_mu[kSceneMaxFree];
    float free_init[kSceneMaxFree][7];      // start pose (create_actor): what reset leaves
    float static_pos[kSceneMaxStatic][3], static_quat[kSceneMaxStatic][4], static_half[kSceneMaxStatic][3], static_mu[kSceneMaxStatic];
    float arm_mu;           // friction of the actor's shapes (combined with the other side's by averaging, PhysX's default combine mode)
};

// point / sphere (centre c in the box frame, radius r) vs box of half sizes a[3]: signed distance, outward normal (box frame)
MI_HD void scene_sphere_box(const float* c, float r, const float* a, float* dist, float* n) {
    float d[3], pn[3];
    sfor<3>([&](auto K) MI_LAMBDA { d[K] = c[K] - fminf(fmaxf(c[K], -a[K]), a[K]); pn[K] = a[K] - fabsf(c[K]); });
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

// EDGE-EDGE contact of two boxes (A: rotation Ra, centre xa, half sizes ha; B likewise), separating-axis test: of the 15 axes (3 + 3 face normals,
// 9 edge x edge) the one of least penetration says which features touch.  False unless that axis is an edge x edge one by a margin (5 % + 0.1 mm:
// the ties of face contacts -- a yawed cube flat on a table, a_x x b_y is the table's normal again -- stay face contacts, which the corner-in-box
// tests cover); else the separation *dist along it, the unit normal n from B towards A, the contact point pc (the middle of the closest points
// of the two supporting edges) and the pair's edge axes 3 i + j.  In A's frame with R = Ra^T Rb: the axis a_i x b_j is e_i x R[:, j], its length
// sqrt(1 - R_ij^2), the boxes' radii along it the classic |R| sums.  Same rule as oracle/scene.py box_edge_contact (which tests the axes in world
// coordinates through a generic support function).
// Returns 1 with that contact; 2 when a FACE axis is the one of least penetration: *axes = 4 * (the face is A's) + its axis -- B's unless one
// of A's is better by the same margin (scene_face_crossings below takes it from there); 0 when the boxes are further apart than `reach`.
MI_HD int scene_box_edge(const float* Ra, const float* xa, const float* ha, const float* Rb, const float* xb, const float* hb, const float reach,
                         float* dist, float* n, float* pc, int* axes) {
    float R[3][3], Q[3][3], dA[3], dB[3];
    const float d[3] = {xa[0] - xb[0], xa[1] - xb[1], xa[2] - xb[2]};
    matTvec3(Ra, d, dA);
    matTvec3(Rb, d, dB);
    sfor<3>([&](auto I) MI_LAMBDA {
        sfor<3>([&](auto J) MI_LAMBDA {
            constexpr int i = I, j = J;
            R[i][j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j];          // a_i . b_j (columns of row-major rotations)
            Q[i][j] = fabsf(R[i][j]);
        });
    });
    float sA = -1e30f, sB = -1e30f;
    int kA = 0, kB = 0;
    sfor<3>([&](auto K) MI_LAMBDA {
        constexpr int k = K;
        const float a_ = fabsf(dA[k]) - ha[k] - (hb[0] * Q[k][0] + hb[1] * Q[k][1] + hb[2] * Q[k][2]);
        const float b_ = fabsf(dB[k]) - hb[k] - (ha[0] * Q[0][k] + ha[1] * Q[1][k] + ha[2] * Q[2][k]);
        kA = (a_ > sA) ? k : kA; sA = fmaxf(sA, a_);
        kB = (b_ > sB) ? k : kB; sB = fmaxf(sB, b_);
    });
    const float s_face = fmaxf(sA, sB);
    float best = -1e30f, cA[3] = {0.f, 0.f, 0.f};
    int bi = -1;
    sfor<3>([&](auto I) MI_LAMBDA {
        sfor<3>([&](auto J) MI_LAMBDA {
            constexpr int i = I, j = J, i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const float l2 = R[i1][j] * R[i1][j] + R[i2][j] * R[i2][j];
            const float il = MI_RSQ(fmaxf(l2, 1e-30f));
            const float s_ = (fabsf(R[i1][j] * dA[i2] - R[i2][j] * dA[i1]) - (ha[i1] * Q[i2][j] + ha[i2] * Q[i1][j]) - (hb[j1] * Q[i][j2] + hb[j2] * Q[i][j1])) * il;
            const bool take = (l2 > 1e-6f) && (s_ > best);
            best = take ? s_ : best;
            bi = take ? 3 * i + j : bi;
            float c[3];
            c[i] = 0.f; c[i1] = -R[i2][j] * il; c[i2] = R[i1][j] * il;
            sfor<3>([&](auto C) MI_LAMBDA { cA[C] = take ? c[C] : cA[C]; });
        });
    });
    if (fmaxf(s_face, best) > reach) return 0;
    if (bi < 0 || !(best > s_face + 0.05f * fabsf(s_face) + 1e-4f)) {
        const bool ref_a = sA > sB + 0.05f * fabsf(sB) + 1e-4f;
        *axes = ref_a ? 4 + kA : kB;
        return 2;
    }
    const float sg = (cA[0] * dA[0] + cA[1] * dA[1] + cA[2] * dA[2]) > 0.f ? 1.f : -1.f;
    sfor<3>([&](auto C) MI_LAMBDA { cA[C] *= sg; });
    matvec3(Ra, cA, n);
    float nB[3], u[3], w[3], pa[3], pb[3];
    matTvec3(Rb, n, nB);
    const int ei = bi / 3, ej = bi - 3 * ei;
    // the supporting edges: the corner of A farthest along -n / of B farthest along +n, minus its component along the edge's own axis
    float ca[3], cb_[3], hai = 0.f, hbj = 0.f;
    sfor<3>([&](auto K) MI_LAMBDA {
        constexpr int k = K;
        ca[k] = (k == ei) ? 0.f : ((cA[k] >= 0.f) ? -ha[k] : ha[k]);
        cb_[k] = (k == ej) ? 0.f : ((nB[k] >= 0.f) ? hb[k] : -hb[k]);
        hai = (k == ei) ? ha[k] : hai;
        hbj = (k == ej) ? hb[k] : hbj;
        u[k] = Ra[3 * k + 0] * (ei == 0 ? 1.f : 0.f) + Ra[3 * k + 1] * (ei == 1 ? 1.f : 0.f) + Ra[3 * k + 2] * (ei == 2 ? 1.f : 0.f);
        w[k] = Rb[3 * k + 0] * (ej == 0 ? 1.f : 0.f) + Rb[3 * k + 1] * (ej == 1 ? 1.f : 0.f) + Rb[3 * k + 2] * (ej == 2 ? 1.f : 0.f);
    });
    matvec3(Ra, ca, pa);
    matvec3(Rb, cb_, pb);
    sfor<3>([&](auto K) MI_LAMBDA { pa[K] += xa[K]; pb[K] += xb[K]; });
    const float r[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    const float uw = dot3(u, w), ru = dot3(r, u), rw = dot3(r, w);
    const float iden = MI_RCP(fmaxf(1.f - uw * uw, 1e-6f));
    const float al = fminf(fmaxf((ru - rw * uw) * iden, -hai), hai), be = fminf(fmaxf((ru * uw - rw) * iden, -hbj), hbj);
    sfor<3>([&](auto K) MI_LAMBDA { pc[K] = 0.5f * ((pa[K] + al * u[K]) + (pb[K] + be * w[K])); });
    *dist = best;
    *axes = bi;
    return 1;
}

// column k (run-time) of a row-major rotation / entry k of a 3-vector, without indexing an array at run time
MI_HD void scene_col(const float* R, const int k, float* o) {
    sfor<3>([&](auto C) MI_LAMBDA { o[C] = (k == 0) ? R[3 * C] : ((k == 1) ? R[3 * C + 1] : R[3 * C + 2]); });
}
MI_HD float scene_pick(const float* v, const int k) { return (k == 0) ? v[0] : ((k == 1) ? v[1] : v[2]); }

// The part of a FACE contact of two boxes that no corner-in-box test sees: where the outline of the incident face crosses the outline of the
// reference face (two planks lying crossed on each other touch in a rectangle none of whose corners is a corner of either box; a plank on a
// knife edge).  Reference box (Rr, xr, hr), its face axis k (scene_box_edge's answer 2); incident box (Ri, xi, hi): its face most anti-parallel to
// the reference face; each of that face's four edges is clipped to the reference face's rectangle (Liang-Barsky in the reference box's frame);
// a clip point that is not an end of the edge is a contact: emit(point, separation from the reference face, outward normal of the reference
// face, 2 * edge + end).  Same rule as oracle/scene.py box_face_crossings.
template <class F>
MI_HD void scene_face_crossings(const float* Rr, const float* xr, const float* hr, const int k, const float* Ri, const float* xi, const float* hi, F&& emit) {
    float nr[3], t1[3], t2[3];
    scene_col(Rr, k, nr);
    scene_col(Rr, (k + 1) % 3, t1);
    scene_col(Rr, (k + 2) % 3, t2);
    const float dx[3] = {xi[0] - xr[0], xi[1] - xr[1], xi[2] - xr[2]};
    const float sr = dot3(nr, dx) >= 0.f ? 1.f : -1.f;
    sfor<3>([&](auto C) MI_LAMBDA { nr[C] *= sr; });
    const float hk = scene_pick(hr, k), h1 = scene_pick(hr, (k + 1) % 3), h2 = scene_pick(hr, (k + 2) % 3);
    float nI[3];
    matTvec3(Ri, nr, nI);
    const float a0 = fabsf(nI[0]), a1 = fabsf(nI[1]), a2 = fabsf(nI[2]);
    const int m = (a0 >= a1 && a0 >= a2) ? 0 : ((a1 >= a2) ? 1 : 2);          // (first of the largest, as numpy's argmax)
    float am[3], u1[3], u2[3];
    scene_col(Ri, m, am);
    scene_col(Ri, (m + 1) % 3, u1);
    scene_col(Ri, (m + 2) % 3, u2);
    const float sI = scene_pick(nI, m) >= 0.f ? -1.f : 1.f;
    const float g0 = sI * scene_pick(hi, m), g1 = scene_pick(hi, (m + 1) % 3), g2 = scene_pick(hi, (m + 2) % 3);
    // the incident face's centre and its two half edges in the reference box's frame (normal, t1, t2 coordinates)
    float c3[3], e1[3], e2[3];
    {
        const float cw[3] = {dx[0] + am[0] * g0, dx[1] + am[1] * g0, dx[2] + am[2] * g0};
        c3[0] = dot3(nr, cw); c3[1] = dot3(t1, cw); c3[2] = dot3(t2, cw);
        e1[0] = dot3(nr, u1) * g1; e1[1] = dot3(t1, u1) * g1; e1[2] = dot3(t2, u1) * g1;
        e2[0] = dot3(nr, u2) * g2; e2[1] = dot3(t1, u2) * g2; e2[2] = dot3(t2, u2) * g2;
    }
    for (int e = 0; e < 4; ++e) {
        // vertices in the order (-,-), (+,-), (+,+), (-,+): edge e runs from vertex e to vertex e + 1
        const float s1a = (e == 0 || e == 3) ? -1.f : 1.f, s2a = (e < 2) ? -1.f : 1.f;
        const float s1b = (e == 0 || e == 1) ? 1.f : -1.f, s2b = (e == 1 || e == 2) ? 1.f : -1.f;
        float q0[3], dq[3];
        sfor<3>([&](auto C) MI_LAMBDA {
            q0[C] = c3[C] + s1a * e1[C] + s2a * e2[C];
            dq[C] = (s1b - s1a) * e1[C] + (s2b - s2a) * e2[C];
        });
        float ta = 0.f, tb = 1.f;
        sfor<2>([&](auto C) MI_LAMBDA {
            constexpr int c = 1 + C;
            const float hb_ = (c == 1) ? h1 : h2;
            sfor<2>([&](auto S_) MI_LAMBDA {
                const float sg = (S_ == 0) ? 1.f : -1.f;
                const float num = hb_ - sg * q0[c], den = sg * dq[c];
                if (fabsf(den) < 1e-12f) { if (num < 0.f) { ta = 1.f; tb = 0.f; } }
                else if (den > 0.f) tb = fminf(tb, num / den);
                else ta = fmaxf(ta, num / den);
            });
        });
        if (ta > tb) continue;
        for (int end = 0; end < 2; ++end) {
            const float t = end ? tb : ta;
            if (end ? !(t < 1.f - 1e-6f) : !(t > 1e-6f)) continue;
            const float qn = q0[0] + t * dq[0], qa = q0[1] + t * dq[1], qb = q0[2] + t * dq[2];
            float p[3];
            sfor<3>([&](auto C) MI_LAMBDA { p[C] = xr[C] + nr[C] * qn + t1[C] * qa + t2[C] * qb; });
            emit(p, qn - hk, nr, 2 * e + end);
        }
    }
}

template <class M>
struct SceneSim : Sim<M> {
    using B = Sim<M>;
    static constexpr int NB = M::NB, ND = M::ND, NV = M::NV, OFF = M::OFF, NSPH = M::NSPH, NSENS = M::NSENS, NLIM = B::NLIM, NVA = B::NVA;
    static_assert(M::FIXED == 1, "SceneSim: a fixed-base actor (the scene's free bodies are boxes)");
    static constexpr int KARM = 24, KBOX = 24;              // contact slots: actor spheres, box corners
    // the chain part of an actor contact's rows is stored DENSE over the actor's coordinates (zeros off the body's chain): the sweeps then visit the
    // actor slots in ONE loop with one copy of the contact code.  (Until round 6 the rows were packed along the body's chain and the sweeps went body
    // by body, each with its own unrolled copy of the contact code: a wavefront walked through all of those copies in every sweep -- the code of a
    // 10-body arm's sweep loop alone outgrew the instruction cache -- and ran every body's loop to the largest count among its lanes.)
    static constexpr int HCH = NV;
    // one contact slot: 3 rows over the actor chain (zero for box contacts) | normal n (3), contact point pc rel. O (3) |
    // Ainv x3, vt_n, lam x3, mu, side A's free box (int bits; -1: the actor / nobody), side B's free box (-1: static)
    // ... | feature id (int bits, > 0)
    static constexpr int S_GEO = 3 * HCH, S_AUX = S_GEO + 6, S_CSZ = S_AUX + 11;
    static constexpr int KSLOT = KARM + KBOX;              // entries of the warm-start tensor: (feature, lam_n, lam_t1, lam_t2) per slot
    static constexpr int NTGT = kSceneMaxFree + kSceneMaxStatic;
    static_assert(kSceneMaxFree * NTGT <= 32, "the box pairs' broad-phase mask is one word");
    static_assert(KSLOT == 48, "tensor scene_warm holds MI_SCENE_WARM_SLOTS = 48 entries per env (include/mi_engine.h; checked against it in tasks/articulation.hpp)");
    static constexpr int R_LIMG = B::limoff(NLIM);
    static constexpr int R_CB = R_LIMG + 3 * NLIM;          // limit G | Ainv, vt, lam | contact slots
    // a box contact's rows span no actor coordinate: its slot is the geometry + scalars only (S_BSZ floats), addressed through a pointer S_GEO floats
    // before it so that both kinds of slot use the same field offsets (without this the dense rows' 144 more floats per env took the Franka's
    // workgroup from 80 to 86 KB of LDS: one workgroup per CU instead of two, 0.88 -> 1.31 ms per gym.simulate() at 4096 envs)
    static constexpr int S_BSZ = S_CSZ - S_GEO, R_BX = R_CB + KARM * S_CSZ - S_GEO;
    static constexpr int R_BODY = R_CB + KARM * S_CSZ + KBOX * S_BSZ;     // per actor body: first slot | count << 8
    // work area behind the slots: what the narrow phase and the sweeps index with a RUN-TIME box number -- the boxes' velocities (lin, ang), centres
    // rel. O, world inverse inertias, inverse masses, rotations; the static boxes' rotations and centres; every box's half sizes (round 6: a pointer into the kernel's parameter struct selected at run
    // time made the compiler copy the whole struct into scratch memory, 416 -> 1392 B per lane); the actor's sphere centres.  In the row store
    // (LDS on the device) because a per-lane array indexed at run time lives in scratch memory, and the sweeps read every velocity right after the
    // previous row wrote it: with the boxes in scratch each of those ~10^4 read-after-write pairs per sub-step was a round trip to memory (1.7 ms
    // per sub-step at ANY batch size; profiles/r5t_scene_time.txt)
    static constexpr int W_VB = R_BODY + NB, W_XF = W_VB + 6 * kSceneMaxFree, W_IINV = W_XF + 3 * kSceneMaxFree, W_IM = W_IINV + 9 * kSceneMaxFree,
                         W_RF = W_IM + kSceneMaxFree, W_RS = W_RF + 9 * kSceneMaxFree, W_XST = W_RS + 9 * kSceneMaxStatic,
                         W_HF = W_XST + 3 * kSceneMaxStatic, W_HS = W_HF + 3 * kSceneMaxFree, W_XS = W_HS + 3 * kSceneMaxStatic;
    static constexpr int ROW_SLOTS = W_XS + 3 * M::NSPHA;

    float box[kSceneMaxFree][13];                           // free boxes: pos3, quat xyzw, linvel3, angvel3 (world)

    // the spheres of one actor body are consecutive (generated sph_body is non-decreasing)
    static constexpr bool sph_grouped() { for (int s = 1; s < NSPH; ++s) if (M::sph_body[s] < M::sph_body[s - 1]) return false; return true; }
    static_assert(sph_grouped(), "collision spheres are listed body by body");
    static constexpr int sph_first(int b) { for (int s = 0; s < NSPH; ++s) if (M::sph_body[s] == b) return s; return 0; }
    static constexpr int sph_count(int b) { int n = 0; for (int s = 0; s < NSPH; ++s) n += (M::sph_body[s] == b) ? 1 : 0; return n; }

    // one sub-step of length h.  tau[ND]: efforts; drv: per-dof position drives; laml: warm-start limit impulses; ncontact: contacts taken
    // (actor + box) | refused for want of a slot << 16
    // warm: [4 * KSLOT] last sub-step's (feature, impulses) per slot, rewritten at the end (p == nullptr: no warm start)
    // vmax: [ND] the asset's joint velocity limits (<= 0: none): the solved joint velocities are clamped to them (the simulator enforces
    // maxJointVelocity; without it an arm under random operational-space commands reaches 70 rad/s and its task's matrix inverses blow up)
    template <int RS>
    MI_HD void substep_scene(const SimParams& P, const SceneParams& SP, const float* tau, const Drive& drv, const float h, const RowStore<RS> rows,
                             const Strided laml, const Strided dof_force, int* ncontact, const Strided warm = Strided{nullptr, 1},
                             const float* vmax = nullptr, const Strided netf = Strided{nullptr, 1}) {
        constexpr int ST = RowStore<RS>::stride;
        float (&q)[M::NDA] = this->q;
        float (&qd)[M::NDA] = this->qd;
        float (&root)[13] = this->root;
        auto G = [&](int row, int c) MI_LAMBDA -> float& { return rows(B::limoff(row) + c); };
        auto Ainv = [&](int row) MI_LAMBDA -> float& { return rows(R_LIMG + row); };
        auto vt = [&](int row) MI_LAMBDA -> float& { return rows(R_LIMG + NLIM + row); };
        auto lam = [&](int row) MI_LAMBDA -> float& { return rows(R_LIMG + 2 * NLIM + row); };
        const float invh = MI_RCP(h);
        const int nf = SP.n_free < kSceneMaxFree ? SP.n_free : kSceneMaxFree, ns = SP.n_static < kSceneMaxStatic ? SP.n_static : kSceneMaxStatic;
        typename B::Ctx c;
        float (&S)[M::NDA][6] = c.S;
        float (&L)[M::NM] = c.L;
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            if constexpr (M::dof_limited[d]) lam(B::limrow(d)) = laml(d);
        });
        // ------------------------------------------------------------ tree pass (the actor's own gravity: asset option disable_gravity)
        {
            SimParams P0 = P;
            if (!SP.arm_gravity) P0.g[0] = P0.g[1] = P0.g[2] = 0.f;
            SpI Iroot;
            float Froot[6];
            this->template body_pass<0>(P0, c, nullptr, nullptr, nullptr, nullptr, Iroot, Froot);
        }
        MI_PHASE();
        // sphere centres (rel. O) in per-lane memory: the narrow phase walks a body's spheres in a run-time loop (one copy of the code per
        // BODY, not per sphere -- core/hand_engine.hpp's reason)
        auto W = [&](int base, int idx) MI_LAMBDA -> float& { return *rows.ptr(base + idx); };
        auto ld3 = [&](int base, int i, float* o) MI_LAMBDA { sfor<3>([&](auto K) MI_LAMBDA { o[K] = W(base, 3 * i + K); }); };
        auto ld9 = [&](int base, int i, float* o) MI_LAMBDA { sfor<9>([&](auto K) MI_LAMBDA { o[K] = W(base, 9 * i + K); }); };
        sfor<NSPH>([&](auto S_) MI_LAMBDA { sfor<3>([&](auto K) MI_LAMBDA { W(W_XS, 3 * S_ + K) = c.xcs[S_][K]; }); });
        // ------------------------------------------------------------ rhs: efforts, passive spring / damper, implicit position drives
        float Ldi[NVA], y[NVA];
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            const float K = M::dof_stiffness[d], Dm = M::dof_damping[d];
            const float kpd = drv.gain_p(d), kdd = drv.gain_d(d);
            L[M::midx[gi][gi]] += M::dof_armature[d] + h * (Dm + kdd) + h * h * (K + kpd);
            y[gi] = tau[d] - c.bias[gi] - K * (q[d] - M::dof_springref[d]) - (Dm + h * K) * qd[d] + kpd * (drv.target[d] - q[d]) - (kdd + h * kpd) * qd[d];
        });
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
        // ------------------------------------------------------------ whitened actor velocity w = L qd + h L^-T rhs; free boxes: v + h g
        float w[NVA];
        sfor_rev<NV>([&](auto I_) MI_LAMBDA {
            constexpr int i = I_;
            const float z = y[i] * Ldi[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { y[M::anc[i][A_]] -= L[M::midx[i][M::anc[i][A_]]] * z; });
            float s = L[M::midx[i][i]] * qd[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { s += L[M::midx[i][M::anc[i][A_]]] * qd[M::anc[i][A_]]; });
            w[i] = s + h * z;
        });
        for (int i = 0; i < nf; ++i) {          // velocities (lin, ang), centre rel. O, rotation, world inverse inertia, inverse mass -> work area
            float R[9];
            quat2mat(box[i] + 3, R);
            sfor<3>([&](auto K) MI_LAMBDA {
                W(W_VB, 6 * i + K) = box[i][7 + K] + h * P.g[K]; W(W_VB, 6 * i + 3 + K) = box[i][10 + K];
                W(W_XF, 3 * i + K) = box[i][K] - root[K];
            });
            sfor<9>([&](auto K) MI_LAMBDA { W(W_RF, 9 * i + K) = R[K]; });
            W(W_IM, i) = MI_RCP(SP.free_mass[i]);
            sfor<3>([&](auto K) MI_LAMBDA { W(W_HF, 3 * i + K) = SP.free_half[i][K]; });
            const float id[3] = {MI_RCP(SP.free_inertia[i][0]), MI_RCP(SP.free_inertia[i][1]), MI_RCP(SP.free_inertia[i][2])};
            sfor<3>([&](auto R_) MI_LAMBDA {
                sfor<3>([&](auto C_) MI_LAMBDA {
                    constexpr int r = R_, cc = C_;
                    W(W_IINV, 9 * i + 3 * r + cc) = R[3 * r] * id[0] * R[3 * cc] + R[3 * r + 1] * id[1] * R[3 * cc + 1] + R[3 * r + 2] * id[2] * R[3 * cc + 2];
                });
            });
        }
        for (int i = 0; i < ns; ++i) {
            float R[9];
            quat2mat(SP.static_quat[i], R);
            sfor<9>([&](auto K) MI_LAMBDA { W(W_RS, 9 * i + K) = R[K]; });
            sfor<3>([&](auto K) MI_LAMBDA { W(W_XST, 3 * i + K) = SP.static_pos[i][K] - root[K]; W(W_HS, 3 * i + K) = SP.static_half[i][K]; });
        }
        MI_PHASE();
        // ------------------------------------------------------------ joint limit rows (as core/engine.hpp)
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (M::dof_limited[d]) {
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
                float a = P.cfm;
                sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { a += g[K] * g[K]; G(row, K) = g[K]; });
                Ainv(row) = MI_RCP(a);
                vt(row) = (C >= 0.f) ? -C * invh : fminf(-C * P.erp * invh, P.max_depen_vel);
                lam(row) = l0;
                w[gi] += g[0] * l0;
                sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += g[1 + A_] * l0; });
            }
        });
        MI_PHASE();
        // the box part of a row: lever x direction, the response of the box to a unit impulse along u at the lever, its share of the diagonal
        auto box_diag = [&](int i, const float* r, const float* u) MI_LAMBDA -> float {
            float rx[3], t[3], Ii[9];
            cross3(r, u, rx);
            ld9(W_IINV, i, Ii);
            matvec3(Ii, rx, t);
            return W(W_IM, i) + dot3(rx, t);
        };
        // last sub-step's impulses of the contact with feature id fid (0, 0, 0 if it did not exist)
        // (actor contacts live in entries [0, KARM), box contacts in [KARM, KSLOT): only the own region is searched)
        auto warm_lookup = [&](int fid, float* l0, int k0, int k1) MI_LAMBDA {
            l0[0] = l0[1] = l0[2] = 0.f;
            if (warm.p == nullptr) return;
            // (a region's used entries come first, the others carry feature 0: a lane's search is over at the first of those or at its match; the
            //  loop ends when every lane's is -- a wave-uniform exit: per-lane `break`s in this loop, inlined at five places, cost 1400 spilled SGPRs)
            bool open_ = true;
            for (int k = k0; k < k1; ++k) {
                if (!MI_WAVE_ANY(open_)) break;
                const int f = __builtin_bit_cast(int, warm(4 * k));
                const bool hit = open_ && (f == fid);
                if (hit) { l0[0] = warm(4 * k + 1) * P.warm; l0[1] = warm(4 * k + 2) * P.warm; l0[2] = warm(4 * k + 3) * P.warm; }
                open_ = open_ && (f != 0) && !hit;
            }
        };
        auto target_velocity = [&](float dist) MI_LAMBDA -> float {
            const float gap = dist - P.rest_offset;
            return (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
        };
        // ------------------------------------------------------------ actor contacts: sphere vs free boxes, then static boxes
        int cnt = 0, refused = 0;
        sfor<NB>([&](auto B_) MI_LAMBDA {
            constexpr int b = B_;
            if constexpr (sph_count(b) > 0) {
                constexpr int CL = M::chain_len[b], S0 = sph_first(b), SN = sph_count(b);
                MI_PHASE();
                const int first = cnt;
                // broad phase (round 6): the targets the body's bounding sphere (around the box of its collision spheres) reaches.  Conservative -- the
                // signed distance to a box is 1-Lipschitz, so a culled (body, target) has no sphere within contact_offset of it --: the contacts
                // and their order are what they were; most links of an arm are nowhere near a cube or the table
                int tmask = 0;
                {
                    float lo[3], hi[3];
                    sfor<3>([&](auto K) MI_LAMBDA { lo[K] = 1e30f; hi[K] = -1e30f; });
                    for (int si = 0; si < SN; ++si) {
                        const float rad = M::sph_rad[S0 + si];
                        float cs[3];
                        ld3(W_XS, S0 + si, cs);
                        sfor<3>([&](auto K) MI_LAMBDA { lo[K] = fminf(lo[K], cs[K] - rad); hi[K] = fmaxf(hi[K], cs[K] + rad); });
                    }
                    const float cb3[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
                    const float hd[3] = {0.5f * (hi[0] - lo[0]), 0.5f * (hi[1] - lo[1]), 0.5f * (hi[2] - lo[2])};
                    const float rbound = MI_SQRT(dot3(hd, hd));
                    for (int t = 0; t < nf + ns; ++t) {
                        const bool fr_ = t < nf;
                        const int ib = fr_ ? t : t - nf;
                        if (CL == 0 && !fr_) continue;
                        float Rb_[9], xb_[3], cl[3], nl[3], dist;
                        ld9(fr_ ? W_RF : W_RS, ib, Rb_);
                        ld3(fr_ ? W_XF : W_XST, ib, xb_);
                        const float rel[3] = {cb3[0] - xb_[0], cb3[1] - xb_[1], cb3[2] - xb_[2]};
                        matTvec3(Rb_, rel, cl);
                        float hb_[3];
                        ld3(fr_ ? W_HF : W_HS, ib, hb_);
                        scene_sphere_box(cl, rbound, hb_, &dist, nl);
                        tmask |= (dist < P.contact_offset + 1e-4f) ? (1 << t) : 0;
                    }
                }
                if (MI_WAVE_ANY(tmask != 0))
                for (int si = 0; si < SN; ++si) {
                    const int s = S0 + si;
                    const float rad = M::sph_rad[s];
                    float cs[3];
                    ld3(W_XS, s, cs);
                    for (int t = 0; t < nf + ns; ++t) {
                        const bool fr_ = t < nf;
                        const int ib = fr_ ? t : t - nf;
                        // a body that no dof moves (the fixed base link) against a static box: the row would act on nothing (a = cfm only) and only
                        // take one of the KARM slots from a finger or a cube
                        if (CL == 0 && !fr_) continue;
                        if (!((tmask >> t) & 1)) continue;
                        float Rb_[9], xb_[3];
                        ld9(fr_ ? W_RF : W_RS, ib, Rb_);
                        ld3(fr_ ? W_XF : W_XST, ib, xb_);
                        float hb_[3];
                        ld3(fr_ ? W_HF : W_HS, ib, hb_);
                        const float rel[3] = {cs[0] - xb_[0], cs[1] - xb_[1], cs[2] - xb_[2]};
                        float cl[3], nl[3], dist;
                        matTvec3(Rb_, rel, cl);
                        scene_sphere_box(cl, rad, hb_, &dist, nl);
                        if (!(dist < P.contact_offset)) continue;
                        if (cnt >= KARM) { refused += 1; continue; }
                        float fr[3][3], pc[3];
                        matvec3(Rb_, nl, fr[0]);                    // from the box towards the sphere
                        contact_frame(fr[0], fr[1], fr[2]);
                        sfor<3>([&](auto K) MI_LAMBDA { pc[K] = cs[K] - rad * fr[0][K]; });
                        float* cb = rows.ptr(R_CB + cnt * S_CSZ);
                        const float rB[3] = {pc[0] - xb_[0], pc[1] - xb_[1], pc[2] - xb_[2]};
                        const int fid = 1 + s * NTGT + (fr_ ? ib : kSceneMaxFree + ib);
                        float l0[3];
                        warm_lookup(fid, l0, 0, KARM);
                        sfor<3>([&](auto K) MI_LAMBDA {
                            constexpr int k = K;
                            float W[6];
                            cross3(pc, fr[k], W);
                            W[3] = fr[k][0]; W[4] = fr[k][1]; W[5] = fr[k][2];
                            float g[M::MAXCHAIN];
                            sfor<CL>([&](auto C) MI_LAMBDA { g[C] = dot6(S[M::chain[b][C] - OFF], W); });
                            sfor<CL>([&](auto C) MI_LAMBDA {
                                constexpr int kk0 = C, ii = M::chain[b][kk0];
                                const float z = g[kk0] * Ldi[ii];
                                g[kk0] = z;
                                sfor<CL - 1 - kk0>([&](auto T) MI_LAMBDA {
                                    constexpr int kk = kk0 + 1 + T, jj = M::chain[b][kk];
                                    g[kk] -= L[M::midx[ii][jj]] * z;
                                });
                            });
                            float a = P.cfm;
                            sfor<HCH>([&](auto C) MI_LAMBDA { cb[(k * HCH + C) * ST] = 0.f; });
                            sfor<CL>([&](auto C) MI_LAMBDA { a += g[C] * g[C]; cb[(k * HCH + M::chain[b][C]) * ST] = g[C]; });
                            if (fr_) a += box_diag(ib, rB, fr[k]);
                            cb[(S_AUX + k) * ST] = MI_RCP(a);
                            cb[(S_AUX + 4 + k) * ST] = l0[k];
                        });
                        cb[(S_AUX + 10) * ST] = __builtin_bit_cast(float, fid);
                        sfor<3>([&](auto I_) MI_LAMBDA { cb[(S_GEO + I_) * ST] = fr[0][I_]; cb[(S_GEO + 3 + I_) * ST] = pc[I_]; });
                        cb[(S_AUX + 3) * ST] = target_velocity(dist);
                        cb[(S_AUX + 7) * ST] = 0.5f * (SP.arm_mu + (fr_ ? SP.free_mu[ib] : SP.static_mu[ib]));
                        cb[(S_AUX + 8) * ST] = __builtin_bit_cast(float, (int)-1);
                        cb[(S_AUX + 9) * ST] = __builtin_bit_cast(float, fr_ ? ib : (int)-1);
                        cnt += 1;
                    }
                }
                rows(R_BODY + b) = __builtin_bit_cast(float, first | ((cnt - first) << 8));
            }
        });
        const int narm = cnt;
        MI_PHASE();
        // ------------------------------------------------------------ box contacts: the corners of every free box vs the ground plane, the
        // static boxes, the other free boxes (side A: the corner's box, pushed along n; side B: the box it is in / on)
        int nbox = 0;
        // broad phase of the box pairs (round 6): bit (i * NTGT + t) -- free box i and target t (static boxes first, then the free ones) -- is set when
        // their bounding spheres come within contact_offset; the corner, edge and outline tests below skip the other pairs (conservative: same contacts)
        unsigned pmask = 0u;
        for (int i = 0; i < nf; ++i) {
            float xi[3];
            ld3(W_XF, i, xi);
            float hi_[3];
            ld3(W_HF, i, hi_);
            const float ri = MI_SQRT(dot3(hi_, hi_));
            for (int t = 0; t < ns + nf; ++t) {
                const bool st_ = t < ns;
                const int j = st_ ? t : t - ns;
                if (!st_ && j == i) continue;
                float xj[3];
                ld3(st_ ? W_XST : W_XF, j, xj);
                float hj[3];
                ld3(st_ ? W_HS : W_HF, j, hj);
                const float d[3] = {xi[0] - xj[0], xi[1] - xj[1], xi[2] - xj[2]};
                const float reach = ri + MI_SQRT(dot3(hj, hj)) + P.contact_offset + 1e-4f;
                pmask |= (dot3(d, d) < reach * reach) ? (1u << (i * NTGT + t)) : 0u;
            }
        }
        for (int i = 0; i < nf; ++i) {
            float Ri[9], xi[3];
            ld9(W_RF, i, Ri);
            ld3(W_XF, i, xi);
            float hi_[3];
            ld3(W_HF, i, hi_);
            for (int cr = 0; cr < 8; ++cr) {
                const float pl[3] = {(cr & 1) ? hi_[0] : -hi_[0], (cr & 2) ? hi_[1] : -hi_[1], (cr & 4) ? hi_[2] : -hi_[2]};
                float pr[3], pc[3];
                matvec3(Ri, pl, pr);
                sfor<3>([&](auto K) MI_LAMBDA { pc[K] = xi[K] + pr[K]; });
                for (int t = -1; t < ns + nf; ++t) {
                    if (t >= ns && t - ns == i) continue;
                    if (t >= 0 && !((pmask >> (i * NTGT + t)) & 1u)) continue;
                    float n[3], dist, mu_b;
                    int ib = -1;
                    if (t < 0) {
                        n[0] = 0.f; n[1] = 0.f; n[2] = 1.f;
                        dist = (root[2] + pc[2]) - P.ground_z;
                        mu_b = P.plane_mu;
                    } else {
                        const bool st_ = t < ns;
                        const int j = st_ ? t : t - ns;
                        float Rb_[9], xb_[3];
                        ld9(st_ ? W_RS : W_RF, j, Rb_);
                        ld3(st_ ? W_XST : W_XF, j, xb_);
                        float hb_[3];
                        ld3(st_ ? W_HS : W_HF, j, hb_);
                        const float rel[3] = {pc[0] - xb_[0], pc[1] - xb_[1], pc[2] - xb_[2]};
                        float cl[3], nl[3];
                        matTvec3(Rb_, rel, cl);
                        scene_sphere_box(cl, 0.f, hb_, &dist, nl);
                        matvec3(Rb_, nl, n);
                        mu_b = st_ ? SP.static_mu[j] : SP.free_mu[j];
                        ib = st_ ? -1 : j;
                    }
                    if (!(dist < P.contact_offset)) continue;
                    if (nbox >= KBOX) { refused += 1; continue; }
                    float fr[3][3];
                    sfor<3>([&](auto K) MI_LAMBDA { fr[0][K] = n[K]; });
                    contact_frame(fr[0], fr[1], fr[2]);
                    float* cb = rows.ptr(R_BX + nbox * S_BSZ);
                    float rB[3] = {0.f, 0.f, 0.f};
                    const int fid = 1 + NSPH * NTGT + (i * 8 + cr) * (NTGT + 1) + (t + 1);
                    float l0[3];
                    warm_lookup(fid, l0, KARM, KSLOT);
                    if (ib >= 0) sfor<3>([&](auto K) MI_LAMBDA { rB[K] = pc[K] - W(W_XF, 3 * ib + K); });
                    sfor<3>([&](auto K) MI_LAMBDA {
                        constexpr int k = K;
                        float a = P.cfm + box_diag(i, pr, fr[k]);
                        if (ib >= 0) a += box_diag(ib, rB, fr[k]);
                        cb[(S_AUX + k) * ST] = MI_RCP(a);
                        cb[(S_AUX + 4 + k) * ST] = l0[k];
                    });
                    cb[(S_AUX + 10) * ST] = __builtin_bit_cast(float, fid);
                    sfor<3>([&](auto I_) MI_LAMBDA { cb[(S_GEO + I_) * ST] = n[I_]; cb[(S_GEO + 3 + I_) * ST] = pc[I_]; });
                    cb[(S_AUX + 3) * ST] = target_velocity(dist);
                    cb[(S_AUX + 7) * ST] = 0.5f * (SP.free_mu[i] + mu_b);
                    cb[(S_AUX + 8) * ST] = __builtin_bit_cast(float, i);
                    cb[(S_AUX + 9) * ST] = __builtin_bit_cast(float, ib);
                    nbox += 1;
                }
            }
        }
        // one more box contact in the next slot: side A = free box ia pushed along n at pc (rel. O), side B = free box ib or nobody (-1)
        auto add_box_contact = [&](const int ia, const int ib, const float* n, const float* pc, const float dist, const float mu, const int fid) MI_LAMBDA {
            float fr[3][3];
            sfor<3>([&](auto K) MI_LAMBDA { fr[0][K] = n[K]; });
            contact_frame(fr[0], fr[1], fr[2]);
            float* cb = rows.ptr(R_BX + nbox * S_BSZ);
            float rA[3], rB[3] = {0.f, 0.f, 0.f}, l0[3];
            warm_lookup(fid, l0, KARM, KSLOT);
            sfor<3>([&](auto K) MI_LAMBDA { rA[K] = pc[K] - W(W_XF, 3 * ia + K); });
            if (ib >= 0) sfor<3>([&](auto K) MI_LAMBDA { rB[K] = pc[K] - W(W_XF, 3 * ib + K); });
            sfor<3>([&](auto K) MI_LAMBDA {
                constexpr int k = K;
                float a = P.cfm + box_diag(ia, rA, fr[k]);
                if (ib >= 0) a += box_diag(ib, rB, fr[k]);
                cb[(S_AUX + k) * ST] = MI_RCP(a);
                cb[(S_AUX + 4 + k) * ST] = l0[k];
            });
            cb[(S_AUX + 10) * ST] = __builtin_bit_cast(float, fid);
            sfor<3>([&](auto I_) MI_LAMBDA { cb[(S_GEO + I_) * ST] = n[I_]; cb[(S_GEO + 3 + I_) * ST] = pc[I_]; });
            cb[(S_AUX + 3) * ST] = target_velocity(dist);
            cb[(S_AUX + 7) * ST] = mu;
            cb[(S_AUX + 8) * ST] = __builtin_bit_cast(float, ia);
            cb[(S_AUX + 9) * ST] = __builtin_bit_cast(float, ib);
            nbox += 1;
        };
        // the corners of the STATIC boxes inside free boxes (a plate lying on a stand smaller than itself has no corner of its own in the stand):
        // the free box is pushed back along the inward normal of the face the corner is nearest to
        constexpr int FID_SC = 1 + NSPH * NTGT + kSceneMaxFree * 8 * (NTGT + 1), FID_EE = FID_SC + kSceneMaxStatic * 8 * kSceneMaxFree,
                      FID_FC = FID_EE + kSceneMaxFree * NTGT * 9;
        for (int t = 0; t < ns; ++t) {
            float Rt[9], xt[3];
            ld9(W_RS, t, Rt);
            ld3(W_XST, t, xt);
            float ht_[3];
            ld3(W_HS, t, ht_);
            for (int cr = 0; cr < 8; ++cr) {
                const float pl[3] = {(cr & 1) ? ht_[0] : -ht_[0], (cr & 2) ? ht_[1] : -ht_[1], (cr & 4) ? ht_[2] : -ht_[2]};
                float pr[3], pc[3];
                matvec3(Rt, pl, pr);
                sfor<3>([&](auto K) MI_LAMBDA { pc[K] = xt[K] + pr[K]; });
                for (int j = 0; j < nf; ++j) {
                    if (!((pmask >> (j * NTGT + t)) & 1u)) continue;
                    float Rb_[9], xb_[3], cl[3], nl[3], n[3], dist;
                    ld9(W_RF, j, Rb_);
                    ld3(W_XF, j, xb_);
                    const float rel[3] = {pc[0] - xb_[0], pc[1] - xb_[1], pc[2] - xb_[2]};
                    matTvec3(Rb_, rel, cl);
                    float hj_[3];
                    ld3(W_HF, j, hj_);
                    scene_sphere_box(cl, 0.f, hj_, &dist, nl);
                    if (!(dist < P.contact_offset)) continue;
                    if (nbox >= KBOX) { refused += 1; continue; }
                    matvec3(Rb_, nl, n);
                    sfor<3>([&](auto K) MI_LAMBDA { n[K] = -n[K]; });
                    add_box_contact(j, -1, n, pc, dist, 0.5f * (SP.free_mu[j] + SP.static_mu[t]), FID_SC + (t * 8 + cr) * kSceneMaxFree + j);
                }
            }
        }
        // per box pair (free box i against the static boxes, then against the free boxes behind it), by the separating-axis test: EDGE-EDGE -- one
        // contact when the least-penetration axis is the cross product of an edge of each (scene_box_edge) --, or, when a face axis wins, the points
        // where the incident face's outline crosses the reference face's (scene_face_crossings)
        for (int i = 0; i < nf; ++i) {
            float Ri[9], xi[3], hi_[3];
            ld9(W_RF, i, Ri);
            ld3(W_XF, i, xi);
            ld3(W_HF, i, hi_);
            for (int t = 0; t < ns + nf - 1 - i; ++t) {
                const bool st_ = t < ns;
                const int j = st_ ? t : i + 1 + (t - ns);
                if (!((pmask >> (i * NTGT + (st_ ? j : ns + j))) & 1u)) continue;
                float Rb_[9], xb_[3], n_[3], pc_[3], dist_;
                int axes;
                ld9(st_ ? W_RS : W_RF, j, Rb_);
                ld3(st_ ? W_XST : W_XF, j, xb_);
                float hj[3];
                ld3(st_ ? W_HS : W_HF, j, hj);
                const int kind = scene_box_edge(Ri, xi, hi_, Rb_, xb_, hj, P.contact_offset, &dist_, n_, pc_, &axes);
                if (kind == 0) continue;
                const float mu_ = 0.5f * (SP.free_mu[i] + (st_ ? SP.static_mu[j] : SP.free_mu[j]));
                const int pair = i * NTGT + (st_ ? j : kSceneMaxStatic + j);
                if (kind == 1) {
                    if (!(dist_ < P.contact_offset)) continue;
                    if (nbox >= KBOX) { refused += 1; continue; }
                    add_box_contact(i, st_ ? -1 : j, n_, pc_, dist_, mu_, FID_EE + pair * 9 + axes);
                    continue;
                }
                const bool ref_a = axes >= 4;
                auto emit = [&](const float* p, const float dist, const float* nr, const int cid) MI_LAMBDA {
                    if (!(dist < P.contact_offset)) return;
                    if (nbox >= KBOX) { refused += 1; return; }
                    const float n[3] = {ref_a ? -nr[0] : nr[0], ref_a ? -nr[1] : nr[1], ref_a ? -nr[2] : nr[2]};       // from B towards A
                    add_box_contact(i, st_ ? -1 : j, n, p, dist, mu_, FID_FC + pair * 8 + cid);
                };
                if (ref_a) scene_face_crossings(Ri, xi, hi_, axes - 4, Rb_, xb_, hj, emit);
                else scene_face_crossings(Rb_, xb_, hj, axes, Ri, xi, hi_, emit);
            }
        }
        *ncontact = (narm + nbox) | (refused << 16);
        MI_PHASE();
        // ------------------------------------------------------------ projected Gauss-Seidel sweeps: limits, actor contacts, box contacts
        // one contact: the normal row, the two tangent rows, then the friction disc (the order of core/hand_engine.hpp).  Btag: 0 for an actor
        // contact, -1 for a box contact
        // `first` (the first sweep): the slot's impulses are last sub-step's and have not acted yet -- they are applied before the contact is solved
        auto solve_contact = [&](auto Btag, float* cb, const bool first) MI_LAMBDA {
            constexpr int b = decltype(Btag)::value;                        // 0: an actor contact (rows over the actor's coordinates), -1: a box contact
            constexpr int CL = b >= 0 ? HCH : 0;
            float g[3][HCH > 0 ? HCH : 1], ainv[3], lm[3], fr[3][3], pc[3];
            sfor<3>([&](auto K) MI_LAMBDA {
                sfor<CL>([&](auto C) MI_LAMBDA { g[K][C] = cb[(K * HCH + C) * ST]; });
                ainv[K] = cb[(S_AUX + K) * ST];
                lm[K] = cb[(S_AUX + 4 + K) * ST];
            });
            sfor<3>([&](auto I_) MI_LAMBDA { fr[0][I_] = cb[(S_GEO + I_) * ST]; pc[I_] = cb[(S_GEO + 3 + I_) * ST]; });
            contact_frame(fr[0], fr[1], fr[2]);
            const float vtn = cb[(S_AUX + 3) * ST], mu = cb[(S_AUX + 7) * ST];
            const int ia = __builtin_bit_cast(int, cb[(S_AUX + 8) * ST]), ib = __builtin_bit_cast(int, cb[(S_AUX + 9) * ST]);
            float rA[3] = {0.f, 0.f, 0.f}, rB[3] = {0.f, 0.f, 0.f};
            if (ia >= 0) sfor<3>([&](auto K) MI_LAMBDA { rA[K] = pc[K] - W(W_XF, 3 * ia + K); });
            if (ib >= 0) sfor<3>([&](auto K) MI_LAMBDA { rB[K] = pc[K] - W(W_XF, 3 * ib + K); });
            // the (at most two) boxes' velocities, inverse inertias and inverse masses: loaded once per contact visit, written back at its end
            float vA[6], vB[6], IA[9], IB[9], imA = 0.f, imB = 0.f;
            if (ia >= 0) { sfor<6>([&](auto K) MI_LAMBDA { vA[K] = W(W_VB, 6 * ia + K); }); ld9(W_IINV, ia, IA); imA = W(W_IM, ia); }
            if (ib >= 0) { sfor<6>([&](auto K) MI_LAMBDA { vB[K] = W(W_VB, 6 * ib + K); }); ld9(W_IINV, ib, IB); imB = W(W_IM, ib); }
            auto rowvel = [&](int k) MI_LAMBDA {
                float vn = 0.f;
                if constexpr (b >= 0) sfor<CL>([&](auto C) MI_LAMBDA { vn += g[k][C] * w[C]; });
                if (ia >= 0) { float rx[3]; cross3(rA, fr[k], rx); vn += dot3(fr[k], vA) + dot3(rx, vA + 3); }
                if (ib >= 0) { float rx[3]; cross3(rB, fr[k], rx); vn -= dot3(fr[k], vB) + dot3(rx, vB + 3); }
                return vn;
            };
            auto apply = [&](int k, float dl) MI_LAMBDA {
                if constexpr (b >= 0) sfor<CL>([&](auto C) MI_LAMBDA { w[C] += g[k][C] * dl; });
                if (ia >= 0) {
                    float rx[3], t[3];
                    cross3(rA, fr[k], rx); matvec3(IA, rx, t);
                    sfor<3>([&](auto C) MI_LAMBDA { vA[C] += fr[k][C] * (imA * dl); vA[3 + C] += t[C] * dl; });
                }
                if (ib >= 0) {
                    float rx[3], t[3];
                    cross3(rB, fr[k], rx); matvec3(IB, rx, t);
                    sfor<3>([&](auto C) MI_LAMBDA { vB[C] -= fr[k][C] * (imB * dl); vB[3 + C] -= t[C] * dl; });
                }
            };
            if (first) { apply(0, lm[0]); apply(1, lm[1]); apply(2, lm[2]); }
            const float ln = fmaxf(lm[0] - (rowvel(0) - vtn) * ainv[0], 0.f);
            apply(0, ln - lm[0]);
            // the two tangent rows together: both corrections from the SAME velocity, the disc projection, then one application.  (Solving t1 to
            // zero velocity and applying it before t2 is looked at -- the order of core/hand_engine.hpp -- lets the unclamped t1 impulse of a fast
            // sliding corner spin the box, t2 then cancels a lateral velocity that is not there, and after the projection the friction points 20
            // degrees off the sliding direction: a cube on a 40 degree ramp slid with mu_eff = 0.468 instead of 0.5, tests/test_scene.py.)
            // (Round 6: the other engine forms let a sliding contact repeat the step with one step size for both rows -- friction_disc<true>; here the
            //  rows keep their own: friction_disc<false>, see core/engine.hpp for what the isotropic step did to a grasp.)
            float lt[2], vtg[2];
            sfor<2>([&](auto K) MI_LAMBDA { vtg[K] = rowvel(1 + K); lt[K] = lm[1 + K] - vtg[K] * ainv[1 + K]; });
            friction_disc<false>(lt, lm[1], lm[2], vtg[0], vtg[1], ainv[1], ainv[2], mu * ln);
            cb[(S_AUX + 4) * ST] = ln;
            sfor<2>([&](auto K) MI_LAMBDA {
                const float nl_ = lt[K];
                cb[(S_AUX + 5 + K) * ST] = nl_;
                apply(1 + K, nl_ - lm[1 + K]);
            });
            if (ia >= 0) sfor<6>([&](auto K) MI_LAMBDA { W(W_VB, 6 * ia + K) = vA[K]; });
            if (ib >= 0) sfor<6>([&](auto K) MI_LAMBDA { W(W_VB, 6 * ib + K) = vB[K]; });
        };
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
                    const float lo = rit(R_LIMG + 2 * NLIM + row);
                    const float nl_ = fmaxf(lo - (vn - rit(R_LIMG + NLIM + row)) * rit(R_LIMG + row), 0.f);
                    const float dl = nl_ - lo;
                    rit(R_LIMG + 2 * NLIM + row) = nl_;
                    w[gi] += g[0] * dl;
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += g[1 + A_] * dl; });
                }
            });
            // (the slot number is laundered through a register: with a visible induction variable the loop optimiser gives every one of the
            //  contact's ~100 row-store accesses a strength-reduced address of its own and spills 1400 SGPRs)
            for (int i = 0; i < narm; ++i) { int ii = i; MI_OPAQUE_VI(ii); solve_contact(std::integral_constant<int, 0>{}, rit.ptr(R_CB + ii * S_CSZ), it == 0); }
            for (int i = 0; i < nbox; ++i) { int ii = i; MI_OPAQUE_VI(ii); solve_contact(std::integral_constant<int, -1>{}, rit.ptr(R_BX + ii * S_BSZ), it == 0); }
        }
        MI_PHASE();
        // ------------------------------------------------------------ back to generalised velocity, outputs
        float v[NVA];
        sfor<NV>([&](auto I_) MI_LAMBDA {
            constexpr int i = I_;
            float s = w[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA { s -= L[M::midx[i][M::anc[i][A_]]] * v[M::anc[i][A_]]; });
            v[i] = s * Ldi[i];
        });
        if (vmax != nullptr) sfor<ND>([&](auto D) MI_LAMBDA { if (vmax[D] > 0.f) v[OFF + D] = fminf(fmaxf(v[OFF + D], -vmax[D]), vmax[D]); });
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            float ll = 0.f;
            if constexpr (M::dof_limited[d]) {
                constexpr int row = B::limrow(d);
                ll = (G(row, 0) > 0.f) ? lam(row) : -lam(row);          // G(row, 0) = s / L_dd, s = +1 lower, -1 upper
            }
            laml(d) = ll;
            dof_force(d) = tau[d] - M::dof_stiffness[d] * (q[d] - M::dof_springref[d]) - M::dof_damping[d] * v[OFF + d] + ll * invh
                           + drv.gain_p(d) * (drv.target[d] - q[d]) - drv.gain_d(d) * v[OFF + d];
        });
        // gym's net contact force tensor, the actor's rows [NB][3]: per actor body the sum of its contacts' forces (world frame, this sub-step)
        if (netf.p != nullptr) {
            sfor<NB>([&](auto B_) MI_LAMBDA {
                constexpr int b = B_;
                float f[3] = {0.f, 0.f, 0.f};
                if constexpr (sph_count(b) > 0) {
                    const int fc = __builtin_bit_cast(int, rows(R_BODY + b));
                    const int first = fc & 255, nb_ = fc >> 8;
                    for (int i = 0; i < nb_; ++i) {
                        const float* cb = rows.ptr(R_CB + (first + i) * S_CSZ);
                        float fr[3][3];
                        sfor<3>([&](auto I_) MI_LAMBDA { fr[0][I_] = cb[(S_GEO + I_) * ST]; });
                        contact_frame(fr[0], fr[1], fr[2]);
                        sfor<3>([&](auto K) MI_LAMBDA {
                            const float l_ = cb[(S_AUX + 4 + K) * ST] * invh;
                            sfor<3>([&](auto C) MI_LAMBDA { f[C] += fr[K][C] * l_; });
                        });
                    }
                }
                sfor<3>([&](auto C) MI_LAMBDA { netf(3 * b + C) = f[C]; });
            });
        }
        if (warm.p != nullptr) {
            for (int k = 0; k < KSLOT; ++k) {
                const bool used = (k < KARM) ? (k < narm) : (k - KARM < nbox);
                const float* cb = (k < KARM) ? rows.ptr(R_CB + k * S_CSZ) : rows.ptr(R_BX + (k - KARM) * S_BSZ);
                warm(4 * k) = used ? cb[(S_AUX + 10) * ST] : 0.f;
                sfor<3>([&](auto J) MI_LAMBDA { warm(4 * k + 1 + J) = used ? cb[(S_AUX + 4 + J) * ST] : 0.f; });
            }
        }
        // ------------------------------------------------------------ integrate the actor and the boxes (semi-implicit Euler)
        sfor<ND>([&](auto D) MI_LAMBDA { qd[D] = v[OFF + D]; q[D] += h * qd[D]; });
        for (int i = 0; i < nf; ++i) {
            float vv[6];
            sfor<6>([&](auto K) MI_LAMBDA { vv[K] = W(W_VB, 6 * i + K); });
            const float w2 = vv[3] * vv[3] + vv[4] * vv[4] + vv[5] * vv[5], l2 = vv[0] * vv[0] + vv[1] * vv[1] + vv[2] * vv[2];
            const float sw = (w2 > kMaxAngularVelocity * kMaxAngularVelocity) ? kMaxAngularVelocity * MI_RSQ(w2) : 1.f;
            const float sl = (l2 > kMaxLinearVelocity * kMaxLinearVelocity) ? kMaxLinearVelocity * MI_RSQ(l2) : 1.f;
            for (int k = 0; k < 3; ++k) { vv[k] *= sl; vv[3 + k] *= sw; box[i][7 + k] = vv[k]; box[i][10 + k] = vv[3 + k]; box[i][k] += h * vv[k]; }
            const float* om = vv + 3;
            const float an = MI_SQRT(dot3(om, om)), th = an * h;
            float sn, cs;
            sincosf(0.5f * th, &sn, &cs);
            const bool big = th > 1e-12f;
            const float k = big ? sn * MI_RCP(fmaxf(an, 1e-30f)) : 0.5f * h;
            const float dq[4] = {om[0] * k, om[1] * k, om[2] * k, big ? cs : 1.f};
            float* Q = box[i] + 3;
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
