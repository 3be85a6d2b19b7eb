/* oracle/hand.c -- CPU oracle of the Shadow-Hand physics step in C (fixed-base 24-DoF hand + one free object).
 * TEST INFRASTRUCTURE ONLY: nothing under isaacgymenvs_amd/ links or loads this file; tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg are its only users.
 *
 * What it restates: gym.simulate() for reference isaacgymenvs/tasks/shadow_hand.py (closed PhysX => PARITY UNPINNED);
 * the stated algorithm is the one of csrc/core/hand_engine.hpp / hand_engine_mw.hpp, written differently: dense generalised-coordinate
 * matrices from physics.c (fk / rnea_bias / crba / point_jac, #included below), fp64, one env per OpenMP iteration.  It is the C twin
 * of oracle/hand.py (numpy; kept as the independent cross-check of this file at small sizes, tests/test_oracle_physics.py) so that
 * ShadowHand@16384 can be compared on EVERY env (tests/test_gpu_fullsize.py) and solver orders can be studied on thousands of states
 * (tools/solver_convergence.py hand).
 *
 *   hand    joint-space dynamics with gravity off (shadow_hand.py:239), implicit PD position drives tau = kp (target - q) - D qd
 *           (kp: MJCF position actuators, shared.xml:250-269), joint-limit rows, 4 fixed tendons as soft two-sided limits
 *           (limit_stiffness / damping set by the task, shadow_hand.py:256-266)
 *   object  free rigid body (cube: isotropic inertia; egg / pen: principal inertias about the body axes), gravity on
 *   contact hand collision geometry sampled by spheres against the exact box / capsule, first-order ellipsoid; 3 rows per contact
 *           (normal + friction disc), no warm start of object contacts
 *   pairs   the asset's 18 explicit hand-to-hand contact pairs (shared.xml:31-51, frictionless) as compliant contacts: h_pairs()
 *   solver 0  one Gauss-Seidel sequence: all limit rows, then the contacts in sphere order; at most kmax contacts per env and
 *             body_cap per hand body (the single-wave kernel hand_substep_kernel)
 *   solver 1  BLOCK sweeps (the finger-per-wave kernel hand_substep_mw_kernel): the hand's limbs (0 = forearm / wrist / palm, 1..5 =
 *             the fingers) are dealt to blocks (body_block of the OrModel); a limb keeps at most limb_cap[l] contacts (and body_cap per
 *             body); a block sweeps, limb by limb, the limit rows of the limb's dofs and then the limb's contacts, Gauss-Seidel inside
 *             the block, Jacobi with mass splitting across blocks on the coordinates they share -- the two wrist dofs (group 0, touched
 *             by every block) and the object's six (group nlimb, touched by the blocks that hold a contact): physics.c solve_blocks().
 */
#include "physics.c"

typedef struct {
    int32_t nos, ntend, kmax, body_cap, shape, solver, nlimb, pad;   /* shape: 0 box, 1 capsule (pen), 2 ellipsoid (egg) */
    const int32_t *os_body;                  /* [nos] non-decreasing */
    const real *os_pos, *os_rad;             /* [nos*3], [nos] */
    const int32_t *tend_d0, *tend_d1;        /* [ntend] */
    const real *tend_c0, *tend_c1, *tend_lo, *tend_hi;
    real tend_stiffness, tend_damping;
    const real *kp;                          /* [nd] drive stiffness */
    real obj_mass, obj_inertia[3], obj_dims[3], mu;   /* box: dims[0] = half size; capsule: radius, half length; ellipsoid: semi-axes */
    const int32_t *limb_of_body;             /* [nb] (solver 1) */
    const int32_t *limb_cap;                 /* [nlimb] (solver 1) */
    const real *fmax;                        /* [nd] drive force limits (0: none), or NULL -- physics.c OrDriveClamp */
    /* the asset's explicit hand-to-hand contact pairs (MJCF <contact><pair>, reference assets/mjcf/open_ai_assets/hand/shared.xml:31-51;
     * models/shadow_hand_extras.json "pairs"): side a a capsule (axis end points a0, a1 in the body frame, radius ra) or a box (pair_box: a0 =
     * centre, a1 = half sizes), side b a capsule; COMPLIANT contacts of stiffness pair_k (0: off) -- h_pairs() below */
    int32_t npair, pad2;
    const int32_t *pair_ba, *pair_bb, *pair_box;   /* [npair] */
    const real *pair_a0, *pair_a1, *pair_ra, *pair_b0, *pair_b1, *pair_rb;   /* [npair*3] / [npair] */
    real pair_k;
    int32_t *pair_sides;                     /* [nenv] out (or NULL): pair sides pushed in the last sub-step */
} OrHand;

static void h_contact_frame(const real *n, real *t1, real *t2) {       /* oracle/hand.py contact_frame */
    real a[3] = {1 - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
    real na = RSQRT(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (na > (real)1e-6) { t1[0] = a[0] / na; t1[1] = a[1] / na; t1[2] = a[2] / na; }
    else { t1[0] = 0; t1[1] = 1; t1[2] = 0; }
    v3cross(n, t1, t2);
}

static real h_sphere_box(const real *c, real r, real half, real *n) {
    real qc[3], d[3], nd = 0;
    for (int k = 0; k < 3; k++) { qc[k] = c[k] < -half ? -half : (c[k] > half ? half : c[k]); d[k] = c[k] - qc[k]; nd += d[k] * d[k]; }
    nd = RSQRT(nd);
    if (nd > (real)1e-12) { for (int k = 0; k < 3; k++) n[k] = d[k] / nd; return nd - r; }
    int i = 0;
    real pen[3];
    for (int k = 0; k < 3; k++) pen[k] = half - RFABS(c[k]);
    for (int k = 1; k < 3; k++) if (pen[k] < pen[i]) i = k;
    n[0] = n[1] = n[2] = 0;
    n[i] = c[i] >= 0 ? 1 : -1;
    return -pen[i] - r;
}
static real h_sphere_ellipsoid(const real *c, real r, const real *a, real *n) {
    real u[3], g[3], k0 = 0, k1 = 0;
    for (int k = 0; k < 3; k++) { u[k] = c[k] / a[k]; g[k] = u[k] / a[k]; k0 += u[k] * u[k]; k1 += g[k] * g[k]; }
    k0 = RSQRT(k0);
    if (k1 <= (real)1e-20) { real mn = a[0] < a[1] ? a[0] : a[1]; mn = mn < a[2] ? mn : a[2]; n[0] = 0; n[1] = 0; n[2] = 1; return -mn - r; }
    k1 = RSQRT(k1);
    for (int k = 0; k < 3; k++) n[k] = g[k] / k1;
    return k0 * (k0 - 1) / k1 - r;
}
static real h_sphere_capsule(const real *c, real r, real rc, real hl, real *n) {
    real pz = c[2] < -hl ? -hl : (c[2] > hl ? hl : c[2]);
    real d[3] = {c[0], c[1], c[2] - pz};
    real nn = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (nn <= (real)1e-24) { n[0] = 1; n[1] = 0; n[2] = 0; return -rc - r; }
    nn = RSQRT(nn);
    for (int k = 0; k < 3; k++) n[k] = d[k] / nn;
    return nn - rc - r;
}

/* sphere (centre c in the box frame, radius r) against a box of half sizes a[3]: signed distance, outward normal */
static real h_sphere_box3(const real *c, real r, const real *a, real *n) {
    real qc[3], d[3], nd = 0;
    for (int k = 0; k < 3; k++) { qc[k] = c[k] < -a[k] ? -a[k] : (c[k] > a[k] ? a[k] : c[k]); d[k] = c[k] - qc[k]; nd += d[k] * d[k]; }
    nd = RSQRT(nd);
    if (nd > (real)1e-12) { for (int k = 0; k < 3; k++) n[k] = d[k] / nd; return nd - r; }
    int i = 0;
    real pen[3];
    for (int k = 0; k < 3; k++) pen[k] = a[k] - RFABS(c[k]);
    for (int k = 1; k < 3; k++) if (pen[k] < pen[i]) i = k;
    n[0] = n[1] = n[2] = 0;
    n[i] = c[i] >= 0 ? 1 : -1;
    return -pen[i] - r;
}

/* The hand-to-hand pairs as compliant contacts (csrc/core/hand_engine.hpp pair_side states the model): for every listed pair whose shapes
 * overlap by pen > 0 at the start of the sub-step -- capsule axes: exact closest points; the palm box against the thumb tip's capsule sampled by
 * the spheres at its ends and its middle, the deepest one -- each side s is pushed along u_s (u_a = n from b towards a, u_b = -n) at the contact
 * point (the middle of the overlap) by F_s = k (pen - h J_s qd+): M += h^2 k J_s^T J_s, rhs += J_s^T k (pen - h J_s qd).  Frictionless (condim 1).
 * Returns the number of sides pushed. */
/* The pairs on the force-sensor bodies (the fingertips; shadow_hand.py:291-297 -- PhysX's sensors see every constraint force on their body, the
 * hand's own contacts included).  A pushed side's force is F_s = k (pen_s - h W_s . V+), W_s = [pc x u; u] about the root, V+ the body's twist after
 * the solve.  csrc/core/hand_engine.hpp pair_sensor_acc / pair_sensor_wrench state what the sensor reports: per fingertip A = sum k pen_s W_s, P = sum
 * pen_s, n = sides, and wrench = A (1 - n h (A . V+) / (k P^2)) -- exact for one side, the damping term along the resultant for more. */
typedef struct { real A[6], P, n; } PairSens;
#define MAXSENS 8
static int h_pairs(const OrModel *m, const OrHand *hd, const Work *w, real h, const real *qd, real (*M)[MAXV], real *rhs, PairSens *ps) {
    int sides = 0;
    static _Thread_local real Jr[MAXV];
    for (int k = 0; k < m->nsens && k < MAXSENS; k++) { for (int c = 0; c < 6; c++) ps[k].A[c] = 0; ps[k].P = 0; ps[k].n = 0; }
    for (int p = 0; p < hd->npair; p++) {
        const int ba = hd->pair_ba[p], bb = hd->pair_bb[p];
        real b0[3], b1[3], t[3], n[3], pc[3], dist;
        m3v(w->R[bb], hd->pair_b0 + 3 * p, t); for (int c = 0; c < 3; c++) b0[c] = w->r[bb][c] + t[c];
        m3v(w->R[bb], hd->pair_b1 + 3 * p, t); for (int c = 0; c < 3; c++) b1[c] = w->r[bb][c] + t[c];
        if (hd->pair_box[p]) {
            dist = (real)1e30;
            for (int s = 0; s < 3; s++) {
                real cw[3], rel[3], cl[3], nl[3], nw[3];
                for (int c = 0; c < 3; c++) { cw[c] = b0[c] + (real)0.5 * s * (b1[c] - b0[c]); rel[c] = cw[c] - w->r[ba][c]; }
                m3tv(w->R[ba], rel, cl);
                for (int c = 0; c < 3; c++) cl[c] -= hd->pair_a0[3 * p + c];
                real ds = h_sphere_box3(cl, hd->pair_rb[p], hd->pair_a1 + 3 * p, nl);
                if (ds < dist) {
                    dist = ds;
                    m3v(w->R[ba], nl, nw);        /* from the box towards the sphere */
                    for (int c = 0; c < 3; c++) { n[c] = -nw[c]; pc[c] = cw[c] - nw[c] * (hd->pair_rb[p] + (real)0.5 * ds); }
                }
            }
        } else {
            real a0[3], a1[3], ca[3], cb[3];
            m3v(w->R[ba], hd->pair_a0 + 3 * p, t); for (int c = 0; c < 3; c++) a0[c] = w->r[ba][c] + t[c];
            m3v(w->R[ba], hd->pair_a1 + 3 * p, t); for (int c = 0; c < 3; c++) a1[c] = w->r[ba][c] + t[c];
            seg_seg_closest(a0, a1, b0, b1, ca, cb);
            real dv[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
            real d = RSQRT(v3dot(dv, dv));
            if (d > (real)1e-9) { n[0] = dv[0] / d; n[1] = dv[1] / d; n[2] = dv[2] / d; } else { n[0] = 0; n[1] = 0; n[2] = 1; }
            dist = d - hd->pair_ra[p] - hd->pair_rb[p];
            for (int c = 0; c < 3; c++) pc[c] = cb[c] + n[c] * (hd->pair_rb[p] + (real)0.5 * dist);
        }
        const real pen = -dist;
        if (!(pen > 0)) continue;
        for (int s = 0; s < 2; s++) {
            const real u[3] = {s ? -n[0] : n[0], s ? -n[1] : n[1], s ? -n[2] : n[2]};
            point_jac(m, w, s ? bb : ba, pc, u, Jr);
            real vs = 0;
            for (int i = 0; i < m->nd; i++) vs += Jr[i] * qd[i];
            const real a = h * h * hd->pair_k, f = hd->pair_k * (pen - h * vs);
            for (int i = 0; i < m->nd; i++) {
                if (Jr[i] == 0) continue;
                rhs[i] += Jr[i] * f;
                for (int j = 0; j < m->nd; j++) M[i][j] += a * Jr[i] * Jr[j];
            }
            for (int k = 0; k < m->nsens && k < MAXSENS; k++) {
                if (m->sens_body[k] != (s ? bb : ba)) continue;
                real Wt[3];
                v3cross(pc, u, Wt);
                for (int c = 0; c < 3; c++) { ps[k].A[c] += hd->pair_k * pen * Wt[c]; ps[k].A[3 + c] += hd->pair_k * pen * u[c]; }
                ps[k].P += pen; ps[k].n += 1;
            }
            sides++;
        }
    }
    return sides;
}

#define HMAXC 64      /* contacts per env */
#define HNV 30        /* 24 hand dofs + 6 object dofs */

typedef struct { int b, si; real pc[3], n[3], t1[3], t2[3]; int row0; } HContact;

/* one sub-step of one env.  st: root13 | q | qd | laml ; obj: pos3 quat4 vel3 angvel3 */
static void hand_substep(const OrModel *m, const OrParams *p, const OrHand *hd, real h, real *st, real *obj, const real *tgt,
                         const real *fobj, const real *scale, const real *lshift, real mu, real *sensor, real *dof_force, int32_t *ncontact,
                         int32_t *limb_count, int32_t *pair_sides) {
    static _Thread_local Work w;
    static _Thread_local real J[MAXROWS][MAXV], Bm[MAXROWS][MAXV];
    const int nd = m->nd;
    static _Thread_local PairSens psens[MAXSENS];
    int have_pairs = 0;
    real *root = st, *q = st + 13, *qd = st + 13 + nd, *laml = st + 13 + 2 * nd;
    const real zero3[3] = {0, 0, 0};
    const real s_mass = scale[0], s_damp = scale[1], s_kp = scale[2], s_tk = scale[3], s_td = scale[4], s_om = scale[5], s_os = scale[6];
    fk(m, root, q, &w);
    rnea_bias(m, root, qd, zero3, &w);           /* gravity is disabled on the hand */
    crba(m, &w);
    real rhs[MAXV], D[MAXD], kp[MAXD];
    for (int i = 0; i < nd; i++) { w.bias[i] *= s_mass; for (int j = 0; j < nd; j++) w.M[i][j] *= s_mass; }
    for (int d = 0; d < nd; d++) {
        D[d] = m->dof_damping[d] * s_damp; kp[d] = hd->kp[d] * s_kp;
        w.M[d][d] += m->dof_armature[d] + h * D[d] + h * h * kp[d];
        rhs[d] = -w.bias[d] - kp[d] * (q[d] - tgt[d]) - (D[d] + h * kp[d]) * qd[d];
    }
    for (int t = 0; t < hd->ntend; t++) {        /* soft two-sided limit on the tendon length */
        int d0 = hd->tend_d0[t], d1 = hd->tend_d1[t];
        real c0 = hd->tend_c0[t], c1 = hd->tend_c1[t];
        real Lt = c0 * q[d0] + c1 * q[d1], Ld = c0 * qd[d0] + c1 * qd[d1];
        real cl = Lt < hd->tend_lo[t] ? hd->tend_lo[t] : (Lt > hd->tend_hi[t] ? hd->tend_hi[t] : Lt);
        real viol = Lt - cl;
        real k = viol != 0 ? hd->tend_stiffness * s_tk : 0, dmp = hd->tend_damping * s_td;
        real a = h * dmp + h * h * k, f = k * viol + (dmp + h * k) * Ld;
        w.M[d0][d0] += a * c0 * c0; w.M[d1][d1] += a * c1 * c1; w.M[d0][d1] += a * c0 * c1; w.M[d1][d0] += a * c0 * c1;
        rhs[d0] -= c0 * f; rhs[d1] -= c1 * f;
    }
    {   /* the asset's hand-to-hand contact pairs: compliant contacts */
        int sides = (hd->npair > 0 && hd->pair_k > 0) ? h_pairs(m, hd, &w, h, qd, w.M, rhs, psens) : 0;
        have_pairs = hd->npair > 0 && hd->pair_k > 0;
        if (pair_sides) *pair_sides = sides;
    }
    /* object: mass matrix in world axes */
    real Ro[9], xo[3] = {obj[0], obj[1], obj[2]};
    quat2mat(obj + 3, Ro);
    const real omass = hd->obj_mass * s_om;
    real Io[9];       /* world-frame inertia Ro diag(I) Ro^T */
    {
        real T[9], Rt[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T[i * 3 + j] = Ro[i * 3 + j] * hd->obj_inertia[j] * s_om; Rt[i * 3 + j] = Ro[j * 3 + i]; }
        m3m(T, Rt, Io);
    }
    for (int i = 0; i < HNV; i++) for (int j = 0; j < HNV; j++) if (i >= nd || j >= nd) w.M[i][j] = 0;
    for (int k = 0; k < 3; k++) w.M[nd + k][nd + k] = omass;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) w.M[nd + 3 + i][nd + 3 + j] = Io[i * 3 + j];
    const int nv = nd + 6;
    chol(nv, w.M, w.L);
    /* free velocity */
    real v[MAXV], dv[MAXV];
    for (int k = 0; k < 3; k++) rhs[nd + k] = omass * p->gravity[k] + fobj[k];
    for (int k = 0; k < 3; k++) rhs[nd + 3 + k] = 0;      /* no gyroscopic torque (PhysX default) */
    chol_solve(nv, w.L, rhs, dv);
    for (int d = 0; d < nd; d++) v[d] = qd[d] + h * dv[d];
    for (int k = 0; k < 6; k++) v[nd + k] = obj[7 + k] + h * dv[nd + k];
    /* ---- rows */
    int nrow = 0, lim_row[MAXD], lim_sign[MAXD];
    real vt[MAXROWS], lam[MAXROWS];
    /* drive clamps ride on the limit row of their dof (every driven dof of the hands is limited) */
    static _Thread_local real cl_fmax[MAXROWS], cl_fa[MAXROWS], cl_c[MAXROWS], cl_sgn[MAXROWS], cl_rho[MAXROWS];
    static _Thread_local int cl_pred[MAXROWS];
    for (int d = 0; d < nd; d++) {
        lim_row[d] = -1;
        if (!m->dof_limited[d]) { laml[d] = 0; continue; }
        {
            real fm = (hd->fmax && kp[d] > 0) ? hd->fmax[d] : 0;
            cl_fmax[nrow] = fm; cl_fa[nrow] = -kp[d] * (q[d] - tgt[d]); cl_c[nrow] = D[d] + h * kp[d]; cl_rho[nrow] = 0;
            cl_pred[nrow] = fm > 0 && RFABS(cl_fa[nrow] - cl_c[nrow] * qd[d]) > fm;
        }
        real lo = m->dof_lower[d] < m->dof_upper[d] ? m->dof_lower[d] : m->dof_upper[d], up = m->dof_lower[d] < m->dof_upper[d] ? m->dof_upper[d] : m->dof_lower[d];
        real dl = q[d] - (lo + lshift[d]), du = (up + lshift[nd + d]) - q[d];
        real Cc = dl < du ? dl : du, s = dl < du ? 1 : -1;
        real lw = laml[d];
        real l0 = (lw * s < 0 ? 0 : RFABS(lw)) * p->warm;
        for (int i = 0; i < nv; i++) J[nrow][i] = 0;
        J[nrow][d] = s;
        vt[nrow] = Cc >= 0 ? -Cc / h : (-Cc * p->erp / h < p->max_depen_vel ? -Cc * p->erp / h : p->max_depen_vel);
        lam[nrow] = l0;
        lim_row[d] = nrow; lim_sign[d] = (int)s; cl_sgn[nrow] = s;
        nrow++;
    }
    const OrDriveClamp clamp = {cl_fmax, cl_fa, cl_c, cl_sgn, cl_pred, cl_rho, h};
    HContact con[HMAXC];
    int ncon = 0, per_body[MAXB], per_limb[16];
    for (int b = 0; b < m->nb; b++) per_body[b] = 0;
    for (int l = 0; l < 16; l++) per_limb[l] = 0;
    for (int si = 0; si < hd->nos; si++) {
        int b = hd->os_body[si];
        real c[3], t[3], rel[3], cl[3], nl[3], dist;
        m3v(w.R[b], hd->os_pos + 3 * si, t);
        for (int k = 0; k < 3; k++) { c[k] = root[k] + w.r[b][k] + t[k]; rel[k] = c[k] - xo[k]; }
        m3tv(Ro, rel, cl);
        if (hd->shape == 1) dist = h_sphere_capsule(cl, hd->os_rad[si], hd->obj_dims[0] * s_os, hd->obj_dims[1] * s_os, nl);
        else if (hd->shape == 2) { real a[3] = {hd->obj_dims[0] * s_os, hd->obj_dims[1] * s_os, hd->obj_dims[2] * s_os}; dist = h_sphere_ellipsoid(cl, hd->os_rad[si], a, nl); }
        else dist = h_sphere_box(cl, hd->os_rad[si], hd->obj_dims[0] * s_os, nl);
        if (dist >= p->contact_offset || per_body[b] >= hd->body_cap) continue;
        if (hd->solver == 0) { if (ncon >= hd->kmax) continue; }
        else { int l = hd->limb_of_body[b]; if (per_limb[l] >= hd->limb_cap[l]) continue; per_limb[l]++; }
        if (ncon >= HMAXC) continue;
        per_body[b]++;
        HContact *cc = &con[ncon];
        cc->b = b; cc->si = si;
        m3v(Ro, nl, cc->n);                      /* from the object towards the sphere */
        h_contact_frame(cc->n, cc->t1, cc->t2);
        for (int k = 0; k < 3; k++) cc->pc[k] = c[k] - hd->os_rad[si] * cc->n[k];
        real xc[3] = {cc->pc[0] - root[0], cc->pc[1] - root[1], cc->pc[2] - root[2]}, rc[3] = {cc->pc[0] - xo[0], cc->pc[1] - xo[1], cc->pc[2] - xo[2]};
        real gap = dist - p->rest_offset;
        real vtn = gap >= 0 ? -gap / h : (-gap * p->erp / h < p->max_depen_vel ? -gap * p->erp / h : p->max_depen_vel);
        cc->row0 = nrow;
        const real *us[3] = {cc->n, cc->t1, cc->t2};
        for (int k = 0; k < 3; k++) {
            real cx[3];
            point_jac(m, &w, b, xc, us[k], J[nrow]);
            v3cross(rc, us[k], cx);
            for (int i = 0; i < 3; i++) { J[nrow][nd + i] = -us[k][i]; J[nrow][nd + 3 + i] = -cx[i]; }
            vt[nrow] = k == 0 ? vtn : 0;
            lam[nrow] = 0;
            nrow++;
        }
        ncon++;
    }
    *ncontact = ncon;
    if (limb_count) for (int l = 0; l < hd->nlimb; l++) limb_count[l] = per_limb[l];
    if (hd->solver == 1) {
        /* units in the block order: limb by limb, the limit rows of the limb's dofs, then the limb's contacts */
        int nunit = 0, u_row[MAXROWS], u_kind[MAXROWS], u_blk[MAXROWS], u_ga[MAXROWS], u_gb[MAXROWS], grp[MAXV];
        real u_mu[MAXROWS];
        for (int d = 0; d < nd; d++) grp[d] = hd->limb_of_body[m->dof_body[d]];
        for (int k = 0; k < 6; k++) grp[nd + k] = hd->nlimb;
        for (int l = 0; l < hd->nlimb; l++) {
            for (int d = 0; d < nd; d++) {
                if (lim_row[d] < 0 || grp[d] != l) continue;
                u_row[nunit] = lim_row[d]; u_kind[nunit] = 0; u_blk[nunit] = m->body_block[m->dof_body[d]]; u_mu[nunit] = 0; u_ga[nunit] = u_gb[nunit] = -1;
                nunit++;
            }
            for (int c = 0; c < ncon; c++) {
                if (hd->limb_of_body[con[c].b] != l) continue;
                u_row[nunit] = con[c].row0; u_kind[nunit] = 1; u_blk[nunit] = m->body_block[con[c].b]; u_mu[nunit] = mu;
                u_ga[nunit] = l; u_gb[nunit] = hd->nlimb;      /* touches its limb's group (and, as every block, the wrist) and the object's */
                nunit++;
            }
        }
        OrModel mm = *m;
        mm.gi_group = grp;
        g_drive_clamp = hd->fmax ? &clamp : 0;
        solve_blocks(&mm, p, &w, nv, nrow, J, vt, lam, v, nunit, u_row, u_kind, u_blk, u_mu, u_ga, u_gb);
        g_drive_clamp = 0;
    } else {
        real Ainv[MAXROWS];
        for (int r = 0; r < nrow; r++) {
            chol_solve(nv, w.L, J[r], Bm[r]);
            real a = p->cfm;
            for (int i = 0; i < nv; i++) a += J[r][i] * Bm[r][i];
            Ainv[r] = 1 / a;
            if (lam[r] != 0) for (int i = 0; i < nv; i++) v[i] += Bm[r][i] * lam[r];
        }
        for (int it = 0; it < p->iters; it++) {
            for (int d = 0; d < nd; d++) {
                int r = lim_row[d];
                if (r < 0) continue;
                if (hd->fmax && cl_fmax[r] > 0) {          /* the drive clamp of the dof, ahead of its limit row (physics.c OrDriveClamp) */
                    real dr = drive_clamp_update(&clamp, r, v[d], lam[r] > 0 ? 0 : 1 / Ainv[r] - p->cfm, 0);
                    if (dr != 0) for (int i = 0; i < nv; i++) v[i] += Bm[r][i] * (cl_sgn[r] * dr);
                }
                real vn = 0;
                for (int i = 0; i < nv; i++) vn += J[r][i] * v[i];
                real nl = lam[r] - (vn - vt[r]) * Ainv[r];
                if (nl < 0) nl = 0;
                real dl = nl - lam[r];
                lam[r] = nl;
                for (int i = 0; i < nv; i++) v[i] += Bm[r][i] * dl;
            }
            for (int c = 0; c < ncon; c++) {
                int r0 = con[c].row0;
                real vn = 0;
                for (int i = 0; i < nv; i++) vn += J[r0][i] * v[i];
                real ln = lam[r0] - (vn - vt[r0]) * Ainv[r0];
                if (ln < 0) ln = 0;
                real dl = ln - lam[r0];
                lam[r0] = ln;
                for (int i = 0; i < nv; i++) v[i] += Bm[r0][i] * dl;
                real lt[2], vtan[2];
                for (int k = 1; k <= 2; k++) {      /* both tangent corrections from the same velocity, then the disc, one application (physics.c friction_step) */
                    int r = r0 + k;
                    real vv = 0;
                    for (int i = 0; i < nv; i++) vv += J[r][i] * v[i];
                    vtan[k - 1] = vv;
                    lt[k - 1] = lam[r] - vv * Ainv[r];
                }
                friction_step(lt, lam[r0 + 1], lam[r0 + 2], vtan, Ainv[r0 + 1], Ainv[r0 + 2], mu * ln);
                for (int k = 1; k <= 2; k++) {
                    int r = r0 + k;
                    real nl = lt[k - 1];
                    dl = nl - lam[r];
                    lam[r] = nl;
                    for (int i = 0; i < nv; i++) v[i] += Bm[r][i] * dl;
                }
            }
        }
    }
    /* ---- outputs */
    for (int d = 0; d < nd; d++) {
        real ll = lim_row[d] >= 0 ? lam[lim_row[d]] * lim_sign[d] : 0;
        laml[d] = ll;
        /* (with a clamped drive: what the actuator delivers, clamp(F) = F + rho / h, plus the limit force) */
        dof_force[d] = -kp[d] * (q[d] - tgt[d]) - D[d] * v[d] + ll / h;
        /* a force-limited drive reports what the actuator delivers: the end-of-step force the clamp acts on, fa - c v + rho / h (+- fmax when saturated) */
        if (hd->fmax && lim_row[d] >= 0 && cl_fmax[lim_row[d]] > 0) dof_force[d] += cl_rho[lim_row[d]] / h - h * kp[d] * v[d];
    }
    for (int k = 0; k < 6 * m->nsens; k++) sensor[k] = 0;
    for (int c = 0; c < ncon; c++) {
        for (int k = 0; k < m->nsens; k++) {
            if (m->sens_body[k] != con[c].b) continue;
            int b = con[c].b, r0 = con[c].row0;
            real f[3], arm[3], tq[3], fl[3], tl[3];
            for (int i = 0; i < 3; i++) {
                f[i] = (con[c].n[i] * lam[r0] + con[c].t1[i] * lam[r0 + 1] + con[c].t2[i] * lam[r0 + 2]) / h;
                arm[i] = con[c].pc[i] - (root[i] + w.r[b][i]);
            }
            v3cross(arm, f, tq);
            m3tv(w.R[b], f, fl); m3tv(w.R[b], tq, tl);
            for (int i = 0; i < 3; i++) { sensor[6 * k + i] += fl[i]; sensor[6 * k + 3 + i] += tl[i]; }
        }
    }
    /* the hand's own contacts on the fingertips (PairSens above): the body's twist about the root from six probes of point_jac -- the velocity of the
       point p is v_O + omega x p --, the wrench A (1 - n h (A . V) / (k P^2)), then into the sensor's frame like a contact force */
    for (int k = 0; have_pairs && k < m->nsens && k < MAXSENS; k++) {
        const PairSens *ps = psens + k;
        if (!(ps->P > 0)) continue;
        const int b = m->sens_body[k];
        static _Thread_local real Jp[MAXV];
        const real zero[3] = {0, 0, 0}, ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0};
        const real dirs[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        real vO[3], vex_y, vex_z, vey_z;
#define PROBE(pt, dir, out) do { point_jac(m, &w, b, pt, dir, Jp); real s_ = 0; for (int i = 0; i < nd; i++) s_ += Jp[i] * v[i]; out = s_; } while (0)
        for (int c = 0; c < 3; c++) PROBE(zero, dirs[c], vO[c]);
        PROBE(ex, dirs[1], vex_y); PROBE(ex, dirs[2], vex_z); PROBE(ey, dirs[2], vey_z);
#undef PROBE
        const real om[3] = {vey_z - vO[2], vO[2] - vex_z, vex_y - vO[1]};
        real av = 0;
        for (int c = 0; c < 3; c++) av += ps->A[c] * om[c] + ps->A[3 + c] * vO[c];
        const real fac = 1 - ps->n * h * av / (hd->pair_k * ps->P * ps->P);
        real f[3], tq[3], rxf[3], fl[3], tl[3];
        for (int c = 0; c < 3; c++) f[c] = ps->A[3 + c] * fac;
        v3cross(w.r[b], f, rxf);
        for (int c = 0; c < 3; c++) tq[c] = ps->A[c] * fac - rxf[c];          /* torque about the sensor origin: tau_O - r_b x f */
        m3tv(w.R[b], f, fl); m3tv(w.R[b], tq, tl);
        for (int c = 0; c < 3; c++) { sensor[6 * k + c] += fl[c]; sensor[6 * k + 3 + c] += tl[c]; }
    }
    /* ---- integrate */
    for (int d = 0; d < nd; d++) { qd[d] = v[d]; q[d] += h * v[d]; }
    for (int k = 0; k < 6; k++) obj[7 + k] = v[nd + k];
    for (int k = 0; k < 3; k++) obj[k] = xo[k] + h * v[nd + k];
    {
        const real *om = v + nd + 3;
        real an = RSQRT(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]), th = an * h, dq[4];
        if (th > (real)1e-12) { real s = RSIN(th / 2) / an; dq[0] = om[0] * s; dq[1] = om[1] * s; dq[2] = om[2] * s; dq[3] = RCOS(th / 2); }
        else { dq[0] = om[0] * h / 2; dq[1] = om[1] * h / 2; dq[2] = om[2] * h / 2; dq[3] = 1; }
        real *Q = obj + 3;
        real x = dq[3] * Q[0] + dq[0] * Q[3] + dq[1] * Q[2] - dq[2] * Q[1];
        real y = dq[3] * Q[1] - dq[0] * Q[2] + dq[1] * Q[3] + dq[2] * Q[0];
        real z = dq[3] * Q[2] + dq[0] * Q[1] - dq[1] * Q[0] + dq[2] * Q[3];
        real ww = dq[3] * Q[3] - dq[0] * Q[0] - dq[1] * Q[1] - dq[2] * Q[2];
        real n = RSQRT(x * x + y * y + z * z + ww * ww);
        Q[0] = x / n; Q[1] = y / n; Q[2] = z / n; Q[3] = ww / n;
    }
}

/* p->substeps sub-steps of every env.  state [nenv][13 + 3 nd] (root | q | qd | laml), obj [nenv][13], targets [nenv][nd],
 * obj_force [nenv][3] (world frame), scale [nenv][8] (csrc/core/hand_engine.hpp HS_*), limit_shift [nenv][2 nd], env_mu [nenv] or NULL
 * (negative entries: hd->mu); outputs sensor [nenv][6 nsens], dof_force [nenv][nd], ncontacts [nenv] (of the last sub-step) */
void or_hand_step(const OrModel *m, const OrParams *p, const OrHand *hd, int nenv, real *state, real *obj, const real *targets,
                  const real *obj_force, const real *scale, const real *limit_shift, const real *env_mu, real *sensor, real *dof_force,
                  int32_t *ncontacts, int32_t *limb_counts /* [nenv][nlimb] or NULL (solver 1: contacts kept per limb) */) {
    const int nd = m->nd, ss = 13 + 3 * nd;
    const real h = p->dt / p->substeps;
#pragma omp parallel for schedule(dynamic, 4)
    for (int e = 0; e < nenv; e++) {
        real mu = (env_mu && env_mu[e] >= 0) ? env_mu[e] : hd->mu;
        for (int s = 0; s < p->substeps; s++)
            hand_substep(m, p, hd, h, state + (size_t)e * ss, obj + (size_t)e * 13, targets + (size_t)e * nd, obj_force + (size_t)e * 3,
                         scale + (size_t)e * 8, limit_shift + (size_t)e * 2 * nd, mu, sensor + (size_t)e * 6 * m->nsens,
                         dof_force + (size_t)e * nd, ncontacts + e, limb_counts ? limb_counts + (size_t)e * hd->nlimb : NULL,
                         hd->pair_sides ? hd->pair_sides + e : NULL);
    }
}

/* world pos, quat xyzw, linear and angular velocity of the sensor (fingertip) bodies of every env: out [nenv][nsens][13]
 * (gym.refresh_rigid_body_state_tensor, shadow_hand.py:440,456-457; the batch form of oracle/hand.py fingertip_states) */
static void h_mat2quat(const real *R, real *q) {       /* isaacgymenvs_amd/assets/model.py mat_to_quat: Shepperd, w >= 0 branch first */
    real tr = R[0] + R[4] + R[8], x, y, z, w;
    if (tr > 0) { real s = RSQRT(tr + 1) * 2; w = s / 4; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s; }
    else if (R[0] > R[4] && R[0] > R[8]) { real s = RSQRT(1 + R[0] - R[4] - R[8]) * 2; w = (R[7] - R[5]) / s; x = s / 4; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s; }
    else if (R[4] > R[8]) { real s = RSQRT(1 + R[4] - R[0] - R[8]) * 2; w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = s / 4; z = (R[5] + R[7]) / s; }
    else { real s = RSQRT(1 + R[8] - R[0] - R[4]) * 2; w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = s / 4; }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}
void or_hand_fingertips(const OrModel *m, int nenv, const real *state, real *out) {
    const int nd = m->nd, ss = 13 + 3 * nd;
    const real zero3[3] = {0, 0, 0};
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nenv; e++) {
        static _Thread_local Work w;
        const real *st = state + (size_t)e * ss;
        fk(m, st, st + 13, &w);
        rnea_bias(m, st, st + 13 + nd, zero3, &w);          /* fills the body velocities V = [omega; v_O] about O */
        for (int k = 0; k < m->nsens; k++) {
            int b = m->sens_body[k];
            real *o = out + ((size_t)e * m->nsens + k) * 13, c[3];
            for (int i = 0; i < 3; i++) o[i] = st[i] + w.r[b][i];
            h_mat2quat(w.R[b], o + 3);
            v3cross(w.V[b], w.r[b], c);
            for (int i = 0; i < 3; i++) { o[7 + i] = w.V[b][3 + i] + c[i]; o[10 + i] = w.V[b][i]; }
        }
    }
}
