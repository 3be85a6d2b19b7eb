/*
 * oracle/physics.c -- CPU restatement of the per-env physics step.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.  The product
 * path (isaacgymenvs_amd/csrc) never links or calls it.
 *
 * What it replaces: the closed `gym.simulate(sim)` + `gym.refresh_*_tensor` calls of the reference
 * (call sites /root/reference/isaacgymenvs/tasks/base/vec_task.py:382,386; tasks/ant.py:233-235;
 * tasks/humanoid.py:240-245).  The arithmetic of that path lives in the third-party, closed-source
 * `isaacgym` Preview-4 binary (PhysX 5), which is not in /root/reference => PARITY UNPINNED against
 * PhysX.  This file instead pins *our* engine's stated algorithm; it is itself pinned by first-principles
 * known-answer tests (tests/test_oracle_physics.py: free fall, cart-pole ODE, energy, linear / angular momentum
 * first-order convergence, static equilibrium weight, Coulomb cone on a tilted plane, joint limits under a constant
 * effort, asset constants).
 *
 * Algorithm (same maths as the HIP kernels, deliberately different formulation: table driven runtime
 * loops, dense mass matrix, dense Cholesky, PGS in generalised-velocity space):
 *   per sub-step h = dt/substeps
 *     FK -> world-frame spatial quantities about O = root origin
 *     bias  = RNEA(q, qd, qdd=0) incl. gravity              (world-frame spatial algebra)
 *     M     = CRBA composite inertias
 *     Mh    = M + diag(armature + h*damping + h^2*stiffness)        (implicit joint spring/damper)
 *     qd*   = qd + h Mh^-1 (tau - bias - K(q-ref) - (D+hK) qd)
 *     rows  = joint limits (1 row / limited dof), ground contacts (3 rows / active sphere)
 *     PGS   iters sweeps, warm started, cone friction
 *     optional (or_step_drive): implicit PD position drives on the dofs, external forces at body centres of mass
 *     integrate q with the new qd (semi-implicit Euler, exponential map for the root quaternion)
 *
 * Build: oracle/Makefile -> oracle/_build/liboracle_f64.so (real=double), liboracle_f32.so (real=float)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef REAL_FLOAT
typedef float real;
#define RSQRT(x) sqrtf(x)
#define RSIN(x) sinf(x)
#define RCOS(x) cosf(x)
#define RFABS(x) fabsf(x)
#else
typedef double real;
#define RSQRT(x) sqrt(x)
#define RSIN(x) sin(x)
#define RCOS(x) cos(x)
#define RFABS(x) fabs(x)
#endif

#define MAXB 40
#define MAXD 40
#define MAXV 46
#define MAXS 96
#define MAXG 32
#define MAXROWS (MAXD + 3 * MAXS + 3 * MAXG)

typedef struct {
    int32_t nb, nd, fixed_base, nsph, nsens, pad0;
    const int32_t *parent;       /* [nb] */
    const real *bpos, *bquat;    /* [nb*3], [nb*4] xyzw */
    const real *mass, *com, *inertia; /* [nb], [nb*3], [nb*6] xx yy zz xy xz yz */
    const int32_t *dof_body, *dof_type; /* [nd] ; type 0 hinge 1 slide */
    const real *dof_axis, *dof_anchor;  /* [nd*3] */
    const real *dof_lower, *dof_upper;
    const int32_t *dof_limited;
    const real *dof_armature, *dof_damping, *dof_stiffness, *dof_springref;
    const int32_t *sph_body;     /* [nsph] */
    const real *sph_pos, *sph_rad, *sph_mu;
    const int32_t *sens_body;    /* [nsens] */
    /* self-collision (actors created with collision filter 0, reference humanoid.py:194): capsules (a sphere is a zero-length
     * capsule) in body frames, capsule pairs gp_* listed group by group (a group = a pair of limbs; pg_first/pg_count index the
     * pair list).  A group carries at most ONE contact per sub-step, its deepest capsule pair.  kmax / kpair > 0 cap the
     * number of ground contacts / group contacts per env (first come first served in sphere / group order), as the engine's
     * LDS-resident contact store does. */
    int32_t ncap, npg, ngp, kmax, kpair, warm_slots;   /* warm_slots > 0: only the first warm_slots ground-contact slots of an env are warm started (the engine's compact store parks last step's impulses in the tail of its slot region, csrc/core/engine.hpp C_WARM_OK) */
    const int32_t *cap_body;     /* [ncap] */
    const real *cap_p0, *cap_p1, *cap_rad, *cap_mu; /* [ncap*3] x2, [ncap] x2 */
    const int32_t *gp_a, *gp_b;  /* [ngp] capsule indices; side a receives +lambda n, side b -lambda n */
    const int32_t *pg_first, *pg_count; /* [npg] */
    /* ---- solver order.  solver 0: one Gauss-Seidel sequence over all rows (limits, ground contacts, self contacts), the order of the
     * single-wave kernels.  solver 1: BLOCK sweeps, the order of the limb-per-wave kernels (csrc/core/engine_mw.hpp, engine_mwc.hpp): the
     * rows are dealt to nblk blocks (a block = what one wavefront sweeps: the limit rows of the dofs and the ground contacts of the
     * bodies with body_block[b] == k; all self contacts form block nblk - 1 when npg > 0); inside a block Gauss-Seidel in the usual row
     * order, across blocks Jacobi with mass splitting (Tonge et al. 2012, the scheme of PhysX's GPU solver): the whitened velocity
     * coordinates come in groups (gi_group: 0 = the trunk incl. the floating base, 1.. = one per limb); a group shared by n armed blocks
     * coordinates come in groups (gi_group: 0 = the trunk incl. the floating base, 1.. = one per limb); a group shared by n active blocks
     * answers each of them with the weight (n + 1) / 2, and after every sweep the blocks' true contributions are summed in block
     * order -- solve_blocks() below. */
    int32_t solver, nblk, pad1, pad2;
    const int32_t *gi_group;     /* [nv] coordinate group of every generalised velocity index */
    const int32_t *body_block;   /* [nb] */
    const int32_t *kmax_blk;     /* [nblk] or NULL: ground contacts kept per BLOCK (the limb-per-wave kernels of the compact store give every
                                  * wave its own contact slots, csrc/core/engine_mwc.hpp), first come first served in sphere order */
} OrModel;

typedef struct {
    real dt;
    int32_t substeps, iters;
    real gravity[3];
    real contact_offset, rest_offset, max_depen_vel, erp, plane_mu, ground_z, cfm, warm;
} OrParams;

/* ------------------------------------------------------------------ small vector helpers */
static inline void v3cross(const real *a, const real *b, real *o) {
    real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline real v3dot(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void m3v(const real *R, const real *v, real *o) {
    real x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
         z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3tv(const real *R, const real *v, real *o) {
    real x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
         z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3m(const real *A, const real *B, real *o) {
    real t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    memcpy(o, t, sizeof(t));
}
static inline void quat2mat(const real *q, real *R) { /* xyzw */
    real x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
static inline void axisangle2mat(const real *a, real th, real *R) {
    real c = RCOS(th), s = RSIN(th), t = 1 - c;
    R[0] = c + a[0] * a[0] * t; R[1] = a[0] * a[1] * t - a[2] * s; R[2] = a[0] * a[2] * t + a[1] * s;
    R[3] = a[1] * a[0] * t + a[2] * s; R[4] = c + a[1] * a[1] * t; R[5] = a[1] * a[2] * t - a[0] * s;
    R[6] = a[2] * a[0] * t - a[1] * s; R[7] = a[2] * a[1] * t + a[0] * s; R[8] = c + a[2] * a[2] * t;
}
/* spatial: X = [ang(3); lin(3)] */
static inline void crm(const real *V, const real *S, real *o) { /* V x S (motion) */
    real a[3], b[3], c[3];
    v3cross(V, S, a); v3cross(V, S + 3, b); v3cross(V + 3, S, c);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static inline void crf(const real *V, const real *F, real *o) { /* V x* F (force: [n; f]) */
    real a[3], b[3], c[3];
    v3cross(V, F, a); v3cross(V + 3, F + 3, b); v3cross(V, F + 3, c);
    o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
typedef struct { real m, h[3], I[6]; } SpI; /* I: xx yy zz xy xz yz about O */
static inline void spi_mul(const SpI *I, const real *X, real *F) { /* F=[n;f] = I*[alpha;a] */
    const real *al = X, *a = X + 3;
    real Ia[3] = {I->I[0] * al[0] + I->I[3] * al[1] + I->I[4] * al[2], I->I[3] * al[0] + I->I[1] * al[1] + I->I[5] * al[2],
                  I->I[4] * al[0] + I->I[5] * al[1] + I->I[2] * al[2]};
    real hxa[3], hxal[3];
    v3cross(I->h, a, hxa); v3cross(I->h, al, hxal);
    F[0] = Ia[0] + hxa[0]; F[1] = Ia[1] + hxa[1]; F[2] = Ia[2] + hxa[2];
    F[3] = I->m * a[0] - hxal[0]; F[4] = I->m * a[1] - hxal[1]; F[5] = I->m * a[2] - hxal[2];
}

/* ------------------------------------------------------------------ per-env workspace */
typedef struct {
    real R[MAXB][9], r[MAXB][3];      /* body frames (r relative to O) */
    real ax[MAXD][3], an[MAXD][3];    /* world joint axes / anchors */
    real S[MAXD][6];
    SpI I[MAXB], Ic[MAXB];
    real V[MAXB][6], A[MAXB][6], F[MAXB][6];
    real M[MAXV][MAXV], L[MAXV][MAXV];
    real bias[MAXV];
} Work;

static void fk(const OrModel *m, const real *root, const real *q, Work *w) {
    for (int b = 0; b < m->nb; b++) {
        real R[9], r[3];
        if (b == 0) {
            quat2mat(root + 3, R);
            r[0] = r[1] = r[2] = 0;
        } else {
            int p = m->parent[b];
            real Rl[9], t[3];
            quat2mat(m->bquat + 4 * b, Rl);
            m3m(w->R[p], Rl, R);
            m3v(w->R[p], m->bpos + 3 * b, t);
            r[0] = w->r[p][0] + t[0]; r[1] = w->r[p][1] + t[1]; r[2] = w->r[p][2] + t[2];
        }
        for (int d = 0; d < m->nd; d++) {
            if (m->dof_body[d] != b) continue;
            real a[3], pt[3], t[3];
            m3v(R, m->dof_axis + 3 * d, a);
            m3v(R, m->dof_anchor + 3 * d, t);
            pt[0] = r[0] + t[0]; pt[1] = r[1] + t[1]; pt[2] = r[2] + t[2];
            memcpy(w->ax[d], a, sizeof(a)); memcpy(w->an[d], pt, sizeof(pt));
            if (m->dof_type[d] == 0) {
                real Q[9], d3[3] = {r[0] - pt[0], r[1] - pt[1], r[2] - pt[2]}, dd[3];
                axisangle2mat(a, q[d], Q);
                m3m(Q, R, R);
                m3v(Q, d3, dd);
                r[0] = pt[0] + dd[0]; r[1] = pt[1] + dd[1]; r[2] = pt[2] + dd[2];
                w->S[d][0] = a[0]; w->S[d][1] = a[1]; w->S[d][2] = a[2];
                v3cross(pt, a, w->S[d] + 3);
            } else {
                r[0] += a[0] * q[d]; r[1] += a[1] * q[d]; r[2] += a[2] * q[d];
                w->S[d][0] = w->S[d][1] = w->S[d][2] = 0;
                w->S[d][3] = a[0]; w->S[d][4] = a[1]; w->S[d][5] = a[2];
            }
        }
        memcpy(w->R[b], R, sizeof(R)); memcpy(w->r[b], r, sizeof(r));
    }
    /* world spatial inertias about O */
    for (int b = 0; b < m->nb; b++) {
        real c[3], t[3];
        m3v(w->R[b], m->com + 3 * b, t);
        c[0] = w->r[b][0] + t[0]; c[1] = w->r[b][1] + t[1]; c[2] = w->r[b][2] + t[2];
        const real *il = m->inertia + 6 * b;
        real Il[9] = {il[0], il[3], il[4], il[3], il[1], il[5], il[4], il[5], il[2]}, T[9], Iw[9], Rt[9];
        const real *R = w->R[b];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[i * 3 + j] = R[j * 3 + i];
        m3m(R, Il, T); m3m(T, Rt, Iw);
        real mm = m->mass[b], cc = v3dot(c, c);
        SpI *I = &w->I[b];
        I->m = mm; I->h[0] = mm * c[0]; I->h[1] = mm * c[1]; I->h[2] = mm * c[2];
        I->I[0] = Iw[0] + mm * (cc - c[0] * c[0]); I->I[1] = Iw[4] + mm * (cc - c[1] * c[1]);
        I->I[2] = Iw[8] + mm * (cc - c[2] * c[2]);
        I->I[3] = Iw[1] - mm * c[0] * c[1]; I->I[4] = Iw[2] - mm * c[0] * c[2]; I->I[5] = Iw[5] - mm * c[1] * c[2];
    }
}

static inline int nvof(const OrModel *m) { return m->nd + (m->fixed_base ? 0 : 6); }
static inline int jo(const OrModel *m) { return m->fixed_base ? 0 : 6; }

/* bias = RNEA(q, qd, 0) incl. gravity; also fills body velocities V */
static void rnea_bias(const OrModel *m, const real *root, const real *qd, const real *g, Work *w) {
    int off = jo(m);
    for (int b = 0; b < m->nb; b++) {
        real Vc[6], Ac[6];
        if (b == 0) {
            if (m->fixed_base) {
                for (int k = 0; k < 6; k++) Vc[k] = 0, Ac[k] = 0;
                Ac[3] = -g[0]; Ac[4] = -g[1]; Ac[5] = -g[2];
            } else {
                const real *v = root + 7, *om = root + 10;
                real wxv[3];
                v3cross(om, v, wxv);
                Vc[0] = om[0]; Vc[1] = om[1]; Vc[2] = om[2]; Vc[3] = v[0]; Vc[4] = v[1]; Vc[5] = v[2];
                Ac[0] = Ac[1] = Ac[2] = 0;
                Ac[3] = -wxv[0] - g[0]; Ac[4] = -wxv[1] - g[1]; Ac[5] = -wxv[2] - g[2];
            }
        } else {
            memcpy(Vc, w->V[m->parent[b]], sizeof(Vc)); memcpy(Ac, w->A[m->parent[b]], sizeof(Ac));
        }
        for (int d = 0; d < m->nd; d++) {
            if (m->dof_body[d] != b) continue;
            real Sd[6];
            crm(Vc, w->S[d], Sd);
            for (int k = 0; k < 6; k++) { Ac[k] += Sd[k] * qd[d]; Vc[k] += w->S[d][k] * qd[d]; }
        }
        memcpy(w->V[b], Vc, sizeof(Vc)); memcpy(w->A[b], Ac, sizeof(Ac));
        real IA[6], IV[6], VxIV[6];
        spi_mul(&w->I[b], Ac, IA); spi_mul(&w->I[b], Vc, IV); crf(Vc, IV, VxIV);
        for (int k = 0; k < 6; k++) w->F[b][k] = IA[k] + VxIV[k];
    }
    for (int b = m->nb - 1; b > 0; b--)
        for (int k = 0; k < 6; k++) w->F[m->parent[b]][k] += w->F[b][k];
    for (int d = 0; d < m->nd; d++) {
        const real *F = w->F[m->dof_body[d]], *S = w->S[d];
        real s = 0;
        for (int k = 0; k < 6; k++) s += S[k] * F[k];
        w->bias[off + d] = s;
    }
    if (!m->fixed_base) {
        w->bias[0] = w->F[0][3]; w->bias[1] = w->F[0][4]; w->bias[2] = w->F[0][5];
        w->bias[3] = w->F[0][0]; w->bias[4] = w->F[0][1]; w->bias[5] = w->F[0][2];
    }
}

static void crba(const OrModel *m, Work *w) {
    int nv = nvof(m), off = jo(m);
    for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) w->M[i][j] = 0;
    for (int b = 0; b < m->nb; b++) w->Ic[b] = w->I[b];
    for (int b = m->nb - 1; b > 0; b--) {
        SpI *P = &w->Ic[m->parent[b]], *C = &w->Ic[b];
        P->m += C->m;
        for (int k = 0; k < 3; k++) P->h[k] += C->h[k];
        for (int k = 0; k < 6; k++) P->I[k] += C->I[k];
    }
    for (int i = 0; i < m->nd; i++) {
        real F[6];
        int bi = m->dof_body[i];
        spi_mul(&w->Ic[bi], w->S[i], F);
        /* ancestors: earlier dofs on the same body, then every dof of every ancestor body */
        for (int j = 0; j <= i; j++) {
            int bj = m->dof_body[j], anc = 0;
            if (bj == bi) anc = 1;
            else { int b = m->parent[bi]; while (b >= 0) { if (b == bj) { anc = 1; break; } b = m->parent[b]; } }
            if (!anc) continue;
            real s = 0;
            for (int k = 0; k < 6; k++) s += w->S[j][k] * F[k];
            w->M[off + i][off + j] = w->M[off + j][off + i] = s;
        }
        if (!m->fixed_base) {
            for (int k = 0; k < 3; k++) {
                w->M[k][off + i] = w->M[off + i][k] = F[3 + k];
                w->M[3 + k][off + i] = w->M[off + i][3 + k] = F[k];
            }
        }
    }
    if (!m->fixed_base) {
        const SpI *I = &w->Ic[0];
        for (int k = 0; k < 3; k++) w->M[k][k] = I->m;
        /* M_vw = -[h]x ; M_wv = [h]x */
        real hx[9] = {0, -I->h[2], I->h[1], I->h[2], 0, -I->h[0], -I->h[1], I->h[0], 0};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { w->M[i][3 + j] = -hx[i * 3 + j]; w->M[3 + i][j] = hx[i * 3 + j]; }
        w->M[3][3] = I->I[0]; w->M[4][4] = I->I[1]; w->M[5][5] = I->I[2];
        w->M[3][4] = w->M[4][3] = I->I[3]; w->M[3][5] = w->M[5][3] = I->I[4]; w->M[4][5] = w->M[5][4] = I->I[5];
    }
}

static void chol(int n, real A[MAXV][MAXV], real L[MAXV][MAXV]) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            real s = A[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) L[i][i] = RSQRT(s > (real)1e-30 ? s : (real)1e-30);
            else L[i][j] = s / L[j][j];
        }
}
static void chol_solve(int n, real L[MAXV][MAXV], const real *b, real *x) {
    real y[MAXV];
    for (int i = 0; i < n; i++) { real s = b[i]; for (int k = 0; k < i; k++) s -= L[i][k] * y[k]; y[i] = s / L[i][i]; }
    for (int i = n - 1; i >= 0; i--) { real s = y[i]; for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k]; x[i] = s / L[i][i]; }
}

/* Ground: the z = ground_z plane (hs == NULL) or a height field sampled on a regular grid, int16 heights, laid out
 * like the reference's `Terrain.height_field_raw` (tasks/anymal_terrain.py:569, converted to a triangle mesh by
 * `convert_heightfield_to_trimesh`, :575, each cell split along the (i,j)-(i+1,j+1) diagonal; vertex (i,j) sits at
 * world (i*hscale - border, j*hscale - border, h*vscale), :208-210).  The surface used here is that piecewise-linear
 * mesh; the generator's slope correction (slopeTreshold: the lower vertex of a steep edge slides under the upper one) is
 * applied per query by levelling every cell edge that rises by more than thr raw units to its lower end -- see
 * oracle/terrain_mesh.py for the corrected mesh itself and how close this comes to it. */
typedef struct {
    const int16_t *hs; /* [rows*cols], row-major: hs[i*cols + j] */
    int32_t rows, cols;
    real hscale, vscale, border;
    real thr;          /* slope_threshold * hscale / vscale; <= 0: no correction */
    int32_t walls;     /* with the correction on: the risers it creates collide from the side (ground_contact) */
} OrGround;

/* height z and unit normal n of the surface under world (x, y) */
static void ground_query(const OrGround *g, real ground_z, real x, real y, real *z, real *n) {
    if (g == 0 || g->hs == 0) { *z = ground_z; n[0] = 0; n[1] = 0; n[2] = 1; return; }
    real gx = (x + g->border) / g->hscale, gy = (y + g->border) / g->hscale;
    int i = (int)floor(gx), j = (int)floor(gy);
    if (i < 0) i = 0; if (i > g->rows - 2) i = g->rows - 2;
    if (j < 0) j = 0; if (j > g->cols - 2) j = g->cols - 2;
    real fx = gx - i, fy = gy - j;
    if (fx < 0) fx = 0; if (fx > 1) fx = 1; if (fy < 0) fy = 0; if (fy > 1) fy = 1;
    real h00 = g->hs[i * g->cols + j], h10 = g->hs[(i + 1) * g->cols + j], h01 = g->hs[i * g->cols + j + 1],
         h11 = g->hs[(i + 1) * g->cols + j + 1];
    real dzx, dzy, zz;
    if (g->thr > 0) {
        real m;
        if (RFABS(h10 - h00) > g->thr) { m = h00 < h10 ? h00 : h10; h00 = m; h10 = m; }
        if (RFABS(h11 - h01) > g->thr) { m = h01 < h11 ? h01 : h11; h01 = m; h11 = m; }
        if (RFABS(h01 - h00) > g->thr) { m = h00 < h01 ? h00 : h01; h00 = m; h01 = m; }
        if (RFABS(h11 - h10) > g->thr) { m = h10 < h11 ? h10 : h11; h10 = m; h11 = m; }
    }
    if (fx >= fy) { dzx = h10 - h00; dzy = h11 - h10; } else { dzx = h11 - h01; dzy = h01 - h00; }
    zz = h00 + dzx * fx + dzy * fy;
    *z = zz * g->vscale;
    real sx = dzx * g->vscale / g->hscale, sy = dzy * g->vscale / g->hscale;
    real inv = 1 / RSQRT(sx * sx + sy * sy + 1);
    n[0] = -sx * inv; n[1] = -sy * inv; n[2] = inv;
}
/* Contact of a sphere (centre (x, y, z), radius r <= hscale) with the terrain: distance dist and unit normal n of the NEARER of
 *  (a) the tangent plane of the surface below the centre (ground_query), and
 *  (b) the wall of a riser.  The mesh generator's slope correction slides the lower vertex of a steep edge under the upper one
 *      (oracle/terrain_mesh.py), so a cell whose two x edges (y edges) both rise by more than thr in the same direction is floor at
 *      the lower level with a vertical wall on the cell boundary of the higher vertices, as tall as they are.  From inside that cell
 *      the wall is a contact candidate: below its top with a horizontal normal and distance (distance to the cell boundary) - r, above
 *      its top against the top EDGE (distance to the edge line - r, normal from the edge to the centre), which continues into the
 *      surface candidate of the cell behind the wall.  x walls on the raw heights, y walls on the x-levelled ones, like the levelling.
 * A sphere of radius <= hscale cannot reach a wall from another cell than the steep one, so only the centre's cell is looked at.  One
 * contact per sphere: the nearer candidate; a sphere inside BOTH (a foot pressed into the corner of tread and riser) gets one contact along
 * the sum of the two penetration vectors, as deep as that sum is long -- the direction adjusts itself to the ratio of the forces it has to
 * carry and the position correction drives both penetrations to zero ("the deeper one wins" let a foot that carries weight creep through
 * the wall on every other sub-step).
 * The reference collides against the triangle mesh itself (anymal_terrain.py:198-211); tests/test_terrain.py compares this with the
 * restated corrected mesh of oracle/terrain_mesh.py. */
static void ground_contact(const OrGround *g, real ground_z, real x, real y, real z, real r, real *dist, real *n) {
    real zt;
    ground_query(g, ground_z, x, y, &zt, n);
    real d = (z - zt) * n[2] - r;
    *dist = d;
    if (g == 0 || g->hs == 0 || !(g->thr > 0) || !g->walls) return;
    real gx = (x + g->border) / g->hscale, gy = (y + g->border) / g->hscale;
    int i = (int)floor(gx), j = (int)floor(gy);
    if (i < 0) i = 0; if (i > g->rows - 2) i = g->rows - 2;
    if (j < 0) j = 0; if (j > g->cols - 2) j = g->cols - 2;
    real fx = gx - i, fy = gy - j;
    if (fx < 0) fx = 0; if (fx > 1) fx = 1; if (fy < 0) fy = 0; if (fy > 1) fy = 1;
    real h00 = g->hs[i * g->cols + j], h10 = g->hs[(i + 1) * g->cols + j], h01 = g->hs[i * g->cols + j + 1],
         h11 = g->hs[(i + 1) * g->cols + j + 1];
    real dw = 1e30, nw[3] = {0, 0, 0};      /* the nearer wall candidate */
    {   /* x walls */
        int s0 = RFABS(h10 - h00) > g->thr, s1 = RFABS(h11 - h01) > g->thr, up0 = h10 > h00, up1 = h11 > h01;
        if (s0 && s1 && up0 == up1) {
            real t0 = up0 ? h10 : h00, t1 = up1 ? h11 : h01;
            real top = (t0 + (t1 - t0) * fy) * g->vscale;
            real dx = (up0 ? 1 - fx : fx) * g->hscale, dz = z - top > 1e-4 ? z - top : 0;     /* above the top (by more than 0.1 mm): its edge */
            real len = RSQRT(dx * dx + dz * dz), il = 1 / (len > 1e-12 ? len : 1e-12);
            if (len - r < dw) { dw = len - r; nw[0] = (up0 ? -dx : dx) * il; nw[1] = 0; nw[2] = dz * il; }
        }
        real m;
        if (s0) { m = h00 < h10 ? h00 : h10; h00 = m; h10 = m; }
        if (s1) { m = h01 < h11 ? h01 : h11; h01 = m; h11 = m; }
    }
    {   /* y walls, on the x-levelled heights */
        int s2 = RFABS(h01 - h00) > g->thr, s3 = RFABS(h11 - h10) > g->thr, up0 = h01 > h00, up1 = h11 > h10;
        if (s2 && s3 && up0 == up1) {
            real t0 = up0 ? h01 : h00, t1 = up1 ? h11 : h10;
            real top = (t0 + (t1 - t0) * fx) * g->vscale;
            real dy = (up0 ? 1 - fy : fy) * g->hscale, dz = z - top > 1e-4 ? z - top : 0;
            real len = RSQRT(dy * dy + dz * dz), il = 1 / (len > 1e-12 ? len : 1e-12);
            if (len - r < dw) { dw = len - r; nw[0] = 0; nw[1] = (up0 ? -dy : dy) * il; nw[2] = dz * il; }
        }
    }
    if (d < 0 && dw < 0) {          /* in the corner of tread and riser, inside both: one contact along the summed penetrations */
        real v[3] = {-d * n[0] - dw * nw[0], -d * n[1] - dw * nw[1], -d * n[2] - dw * nw[2]};
        real L = RSQRT(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        n[0] = v[0] / L; n[1] = v[1] / L; n[2] = v[2] / L;
        d = -L;
    } else if (dw < d) {
        d = dw; n[0] = nw[0]; n[1] = nw[1]; n[2] = nw[2];
    }
    *dist = d;
}
void or_ground_contact(const OrGround *g, real ground_z, real x, real y, real z, real r, real *dist, real *n) {
    ground_contact(g, ground_z, x, y, z, r, dist, n);
}
/* the query by itself (tests/test_terrain.py compares it with oracle/terrain_mesh.py) */
void or_ground_query(const OrGround *g, real ground_z, real x, real y, real *z, real *n) { ground_query(g, ground_z, x, y, z, n); }

/* contact frame: n, t1 = normalize(x - n (n.x)), t2 = n x t1   (n = z gives t1 = x, t2 = y) */
static void contact_frame(const real *n, real *t1, real *t2) {
    real a[3] = {1 - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
    real a2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    if (a2 > (real)1e-12) { real inv = 1 / RSQRT(a2); t1[0] = a[0] * inv; t1[1] = a[1] * inv; t1[2] = a[2] * inv; }
    else { t1[0] = 0; t1[1] = 1; t1[2] = 0; }
    v3cross(n, t1, t2);
}

/* J row of a world direction u at point xc (rel O) on body b */
static void point_jac(const OrModel *m, const Work *w, int b, const real *xc, const real *u, real *J) {
    int nv = nvof(m), off = jo(m);
    for (int i = 0; i < nv; i++) J[i] = 0;
    if (!m->fixed_base) {
        real t[3];
        v3cross(xc, u, t);
        J[0] = u[0]; J[1] = u[1]; J[2] = u[2]; J[3] = t[0]; J[4] = t[1]; J[5] = t[2];
    }
    for (int bb = b; bb >= 0; bb = m->parent[bb])
        for (int d = 0; d < m->nd; d++) {
            if (m->dof_body[d] != bb) continue;
            if (m->dof_type[d] == 0) {
                real rr[3] = {xc[0] - w->an[d][0], xc[1] - w->an[d][1], xc[2] - w->an[d][2]}, t[3];
                v3cross(w->ax[d], rr, t);
                J[off + d] = v3dot(u, t);
            } else J[off + d] = v3dot(u, w->ax[d]);
        }
}

/* closest points of the segments [a0,a1], [b0,b1] (clamped closest-point construction; same operation sequence as
 * isaacgymenvs_amd/assets/model.py::segment_distance and csrc/core/engine.hpp::seg_seg_closest) */
static void seg_seg_closest(const real *a0, const real *a1, const real *b0, const real *b1, real *ca, real *cb) {
    real d1[3], d2[3], rr[3];
    for (int k = 0; k < 3; k++) { d1[k] = a1[k] - a0[k]; d2[k] = b1[k] - b0[k]; rr[k] = a0[k] - b0[k]; }
    const real A = v3dot(d1, d1), E = v3dot(d2, d2), F = v3dot(d2, rr), C = v3dot(d1, rr), B = v3dot(d1, d2);
    const real eps = (real)1e-12;
    const real den = A * E - B * B;
    real s = 0, t;
    if (den > eps && A > eps) { s = (B * F - C * E) / den; s = s < 0 ? 0 : (s > 1 ? 1 : s); }
    t = (E > eps) ? (B * s + F) / E : 0;
    real tc = t < 0 ? 0 : (t > 1 ? 1 : t);
    if ((t != tc || !(E > eps)) && A > eps) { s = (B * tc - C) / A; s = s < 0 ? 0 : (s > 1 ? 1 : s); }
    for (int k = 0; k < 3; k++) { ca[k] = a0[k] + d1[k] * s; cb[k] = b0[k] + d2[k] * tc; }
}


/* ------------------------------------------------------------------ solver 1: block sweeps with mass splitting (OrModel.solver)
 * Whitened coordinates w = L^T P v with P H P^T = L L^T, P ordering every limb's coordinates before the trunk's (group 0): then the
 * trunk part of w is the trunk's own velocity under the inertia the trunk shows with its limbs free, a limb's part is the limb's
 * velocity relative to what the trunk imposes on it -- the same split of the kinetic energy into groups as the engine's branch-sparse
 * L^T L factor produces (inside a group the two bases differ by a rotation, which none of the quantities below notices).
 *
 * One sweep: every block starts from the same w and runs Gauss-Seidel over its own rows in the usual order; a coordinate group that
 * n >= 2 ACTIVE blocks touch answers each of them with the weight om = (n + 1) / 2 (the row's diagonal is cfm + sum_G om_G |g_G|^2,
 * its local velocity update om_G g_G dlam); afterwards the blocks' true contributions g_G dlam are added up in block order.  With
 * D = the block diagonal of these weighted rows, A <= D_n (Cauchy-Schwarz over the n blocks sharing a coordinate) gives A < 2 D_om:
 * the scaled projected-Jacobi condition.  A block is active in a sweep when one of its rows carries an impulse, and before the first
 * sweep also when it holds a contact or a violated joint limit.  tools/solver_convergence.py compares the order with plain
 * Gauss-Seidel. */
#define MAXGRP 16
#define MAXBLK 16
/* Effort-limited position drives (gym dof property `effort`; MJCF `forcerange`, shared.xml:250-269; allegro_hand.py:264).  The drive's
 * implicit force on dof d at the end of the sub-step is F = fa - c v_d (fa = kp (target - q), c = D + h kp, both already in M' and the
 * right-hand side); the actuator delivers clamp(F, +-fmax).  The part the clamp removes is an impulse rho on the dof with
 *      fa - c v_d + rho / h  in [-fmax, fmax],   rho = 0 inside   (rho <= 0 at the upper bound, >= 0 at the lower),
 * solved with the other rows: the unit of a dof's joint-limit row first updates rho in closed form for the row's own response a
 * (g = L^-1 e_d is the limit row's, up to its sign s), then the limit impulse.  While the limit holds the dof (its impulse > 0) the
 * dof's velocity does not answer to rho -- the limit row absorbs it -- so the update then takes a = 0: rho = h (clamp(F) - F).
 * (Taken with the free response there, the pair (rho, limit impulse) on one and the same g oscillates with factor x / (1 - x),
 * x = h c a, and diverges for light joints with stiff drives, x > 1/2.)  In the block order the closed form takes the dof's true
 * response g . g, not the block's weighted one: the row's slope 1 / h - c a shrinks as a grows, so the larger weighted response
 * would over-relax it (the wrist's clamps then oscillate between the blocks).  Arrays are indexed by the limit row. */
typedef struct {
    const real *fmax, *fa, *c, *sgn;   /* [nrow]: fmax 0 = no clamp on this row; sgn = the limit row's s */
    const int *pred;                   /* [nrow]: |fa - c qd| > fmax at the start of the sub-step (block activity in the first sweep) */
    real *rho;                         /* [nrow] in: 0, out: the impulses */
    real h;
} OrDriveClamp;
static _Thread_local const OrDriveClamp *g_drive_clamp = 0;     /* set by the caller around solve_blocks (oracle/hand.c) */

static real drive_clamp_update(const OrDriveClamp *cl, int r, real v_d, real a, real cfm_unused) {
    (void)cfm_unused;
    real c = cl->c[r], fm = cl->fmax[r], k = 1 / cl->h - c * a;
    if (k < (real)0.1 / cl->h) k = (real)0.1 / cl->h;
    real Ff = cl->fa[r] - c * v_d + cl->rho[r] * c * a;
    real rn = Ff > fm ? (fm - Ff) / k : (Ff < -fm ? (-fm - Ff) / k : 0);
    real dr = rn - cl->rho[r];
    cl->rho[r] = rn;
    return dr;
}

/* The tangential update of one contact: lt[] comes in as the per-row step r_k = lam_k - v_k / a_kk.  Inside the friction disc (|r| <= lim, the
   contact sticks) it stands.  A contact that slides takes one step size for both rows instead -- s_k = lam_k - v_k / max(a_11, a_22): never a longer
   step than either row would take alone -- scaled back onto the disc.  Why: the fixed point of "step, radial projection" satisfies
   lam_t || -(D^-1 v_t); with the per-row D = diag(a_11, a_22) that is not Coulomb's law -- measured before round 6: a Humanoid lying on the ground
   and sliding at 31 degrees to the tangent axes was braked along 15 degrees --; with D a multiple of the identity it is: friction antiparallel to the
   sliding velocity (tests/friction_util.py).  Between the regimes (|s| < lim < |r|): the point of the segment s -> r on the circle, which makes the
   update continuous in its inputs. */
static void friction_step(real *lt, real lam1, real lam2, const real *vtan, real ainv1, real ainv2, real lim) {
    real r0 = lt[0], r1 = lt[1], l2 = lim * lim;
    if (r0 * r0 + r1 * r1 <= l2) return;
    real ac = ainv1 < ainv2 ? ainv1 : ainv2;
    real s0 = lam1 - vtan[0] * ac, s1 = lam2 - vtan[1] * ac;
    real n2 = s0 * s0 + s1 * s1;
    if (n2 >= l2) {
        real nrm = RSQRT(n2), sc = lim / (nrm > (real)1e-30 ? nrm : (real)1e-30);
        lt[0] = s0 * sc; lt[1] = s1 * sc;
        return;
    }
    real d0 = r0 - s0, d1 = r1 - s1;
    real a = d0 * d0 + d1 * d1, b = s0 * d0 + s1 * d1, c = n2 - l2;
    if (a < (real)1e-30) a = (real)1e-30;
    real disc = b * b - a * c;
    real t = (RSQRT(disc > 0 ? disc : 0) - b) / a;
    lt[0] = s0 + t * d0; lt[1] = s1 + t * d1;
}

static void solve_blocks(const OrModel *m, const OrParams *p, Work *wk, int nv, int nrow, real (*J)[MAXV], const real *vt, real *lam,
                         real *v, int nunit, const int *u_row, const int *u_kind, const int *u_blk, const real *u_mu,
                         const int *u_ga, const int *u_gb) {
    static _Thread_local real Hp[MAXV][MAXV], Lp[MAXV][MAXV], G[MAXROWS][MAXV];
    int perm[MAXV], grp[MAXV], np_ = 0, ngrp = 0;
    for (int i = 0; i < nv; i++) if (m->gi_group[i] + 1 > ngrp) ngrp = m->gi_group[i] + 1;
    for (int g = ngrp - 1; g >= 0; g--)       /* limbs first (any order: they do not couple), the trunk (group 0) last */
        for (int i = 0; i < nv; i++) if (m->gi_group[i] == g) { perm[np_] = i; grp[np_] = g; np_++; }
    for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) Hp[i][j] = wk->M[perm[i]][perm[j]];
    chol(nv, Hp, Lp);
    real wv[MAXV], wloc[MAXV], dsum[MAXV];
    for (int i = 0; i < nv; i++) { real s = 0; for (int j = i; j < nv; j++) s += Lp[j][i] * v[perm[j]]; wv[i] = s; }
    for (int r = 0; r < nrow; r++)            /* g = L^-1 P J^T */
        for (int i = 0; i < nv; i++) { real s = J[r][perm[i]]; for (int k = 0; k < i; k++) s -= Lp[i][k] * G[r][k]; G[r][i] = s / Lp[i][i]; }
    /* which coordinate groups a block's rows touch: a limb block the trunk and the groups of its own bodies, the self contacts the
     * trunk and the groups of the two bodies of every contact */
    int touch[MAXBLK][MAXGRP];
    for (int b = 0; b < m->nblk; b++) for (int g = 0; g < ngrp; g++) touch[b][g] = 0;
    for (int b = 0; b < m->nb; b++) {
        int g = 0;
        for (int bb = b; bb >= 0 && g == 0; bb = m->parent[bb])
            for (int d = 0; d < m->nd; d++) if (m->dof_body[d] == bb) g = m->gi_group[jo(m) + d];
        touch[m->body_block[b]][g] = 1; touch[m->body_block[b]][0] = 1;
    }
    for (int u = 0; u < nunit; u++)
        if (u_ga[u] >= 0) { touch[u_blk[u]][0] = 1; touch[u_blk[u]][u_ga[u]] = 1; touch[u_blk[u]][u_gb[u]] = 1; }
    for (int r = 0; r < nrow; r++) if (lam[r] != 0) for (int i = 0; i < nv; i++) wv[i] += G[r][i] * lam[r];   /* warm start */
    for (int it = 0; it < p->iters; it++) {
        int active[MAXBLK];
        real om[MAXGRP];
        for (int b = 0; b < m->nblk; b++) active[b] = 0;
        for (int u = 0; u < nunit; u++) {
            int r0 = u_row[u];
            if (lam[r0] > 0 || (it == 0 && (u_kind[u] == 1 || vt[r0] > 0))) active[u_blk[u]] = 1;
            if (g_drive_clamp && u_kind[u] == 0 && g_drive_clamp->fmax[r0] > 0 && (g_drive_clamp->rho[r0] != 0 || (it == 0 && g_drive_clamp->pred[r0])))
                active[u_blk[u]] = 1;
        }
        for (int g = 0; g < ngrp; g++) {
            int n = 0;
            for (int b = 0; b < m->nblk; b++) n += (active[b] && touch[b][g]) ? 1 : 0;
            om[g] = n > 1 ? (real)0.5 * (real)(n + 1) : 1;
        }
        for (int i = 0; i < nv; i++) dsum[i] = 0;
        for (int b = 0; b < m->nblk; b++) {
            for (int i = 0; i < nv; i++) wloc[i] = wv[i];
            for (int u = 0; u < nunit; u++) {
                if (u_blk[u] != b) continue;
                int r0 = u_row[u];
                real Ain[3];
                for (int k = 0; k < (u_kind[u] ? 3 : 1); k++) {
                    real a = p->cfm;
                    for (int i = 0; i < nv; i++) a += om[grp[i]] * G[r0 + k][i] * G[r0 + k][i];
                    Ain[k] = 1 / a;
                }
                if (g_drive_clamp && u_kind[u] == 0 && g_drive_clamp->fmax[r0] > 0) {   /* the dof's drive clamp, ahead of its limit row */
                    real vn = 0, sg = g_drive_clamp->sgn[r0];
                    for (int i = 0; i < nv; i++) vn += G[r0][i] * wloc[i];
                    real a_true = 0;            /* the dof's true response g . g: NOT the block's weighted one (which over-relaxes this row, see below) */
                    for (int i = 0; i < nv; i++) a_true += G[r0][i] * G[r0][i];
                    real dr = drive_clamp_update(g_drive_clamp, r0, sg * vn, lam[r0] > 0 ? 0 : a_true, 0);
                    if (dr != 0) for (int i = 0; i < nv; i++) wloc[i] += om[grp[i]] * G[r0][i] * (sg * dr);
                }
                {   /* limit row / contact normal */
                    real vn = 0;
                    for (int i = 0; i < nv; i++) vn += G[r0][i] * wloc[i];
                    real nl = lam[r0] - (vn - vt[r0]) * Ain[0];
                    if (nl < 0) nl = 0;
                    real dl = nl - lam[r0];
                    lam[r0] = nl;
                    for (int i = 0; i < nv; i++) wloc[i] += om[grp[i]] * G[r0][i] * dl;
                }
                if (u_kind[u] == 0) continue;
                /* the two tangent rows TOGETHER: both corrections from the same velocity, the disc projection, one application (round 6; until then
                   t1 was solved and applied before t2 was looked at: a fast-sliding contact's unclamped t1 impulse turned the body, t2 cancelled a
                   lateral velocity only that impulse had created, and the projected friction pointed off the sliding direction) */
                real lt[2], vtan[2];
                for (int k = 1; k <= 2; k++) {
                    int r = r0 + k;
                    real vn = 0;
                    for (int i = 0; i < nv; i++) vn += G[r][i] * wloc[i];
                    vtan[k - 1] = vn - vt[r];
                    lt[k - 1] = lam[r] - vtan[k - 1] * Ain[k];
                }
                real lim = u_mu[u] * lam[r0];
                friction_step(lt, lam[r0 + 1], lam[r0 + 2], vtan, Ain[1], Ain[2], lim);
                real sc = 1;
                for (int k = 1; k <= 2; k++) {
                    int r = r0 + k;
                    real nl = lt[k - 1] * sc, dl = nl - lam[r];
                    lam[r] = nl;
                    if (dl != 0) for (int i = 0; i < nv; i++) wloc[i] += om[grp[i]] * G[r][i] * dl;
                }
            }
            for (int i = 0; i < nv; i++) dsum[i] += (wloc[i] - wv[i]) / om[grp[i]];     /* the block's true contribution, in block order */
        }
        for (int i = 0; i < nv; i++) wv[i] += dsum[i];
    }
    /* back to generalised velocities: P v = L^-T w */
    real x[MAXV];
    for (int i = nv - 1; i >= 0; i--) { real s = wv[i]; for (int k = i + 1; k < nv; k++) s -= Lp[k][i] * x[k]; x[i] = s / Lp[i][i]; }
    for (int i = 0; i < nv; i++) v[perm[i]] = x[i];
}

/* ------------------------------------------------------------------ one env, one full step of dt
 * state layout per env:  root[13] | q[nd] | qd[nd] | lam_c[3*nsph] | lam_l[nd]
 * outputs per env:       sensor[6*nsens] | dof_force[nd] | sph_force[3*nsph] (world)
 * optional: gnd (height field), mu_env >= 0 (per-env friction replacing the per-sphere model value), netf[3*nb]
 * (net contact force per body, world frame, last sub-step = `contact_collection: 1`, reference AnymalTerrain.yaml:148)
 */
/* optional per-env extras of a step: implicit PD position drives on every dof (gym DOF_MODE_POS: stiffness kp, damping kd,
 * targets target[nd]) and external forces fext[3*nb] applied at the bodies' centres of mass, given in each body's LOCAL
 * frame (gym.apply_rigid_body_force_tensors(..., LOCAL_SPACE), reference quadcopter.py:291-292) */
typedef struct {
    real kp, kd;
    const real *target; /* [nd] or NULL */
    const real *fext;   /* [3*nb] or NULL */
    const real *kpv, *kdv; /* [nd] per-dof gains instead of kp / kd (gym dof properties stiffness / damping differ per dof:
                            * amp/humanoid_amp_base.py:219-222 keeps the MJCF's), or NULL */
} OrExtra;
#define EX_KP(ex, d) ((ex)->kpv ? (ex)->kpv[d] : (ex)->kp)
#define EX_KD(ex, d) ((ex)->kdv ? (ex)->kdv[d] : (ex)->kd)

static void step_env(const OrModel *m, const OrParams *p, const OrGround *gnd, real mu_env, real *root, real *q, real *qd,
                     real *lam_c, real *lam_l, real *lam_p, const real *tau, real *sensor, real *dof_force, real *sph_force,
                     real *pair_out, real *netf, const OrExtra *ex) {
    static _Thread_local Work w;
    int nv = nvof(m), off = jo(m), nd = m->nd;
    real h = p->dt / p->substeps;
    for (int ss = 0; ss < p->substeps; ss++) {
        fk(m, root, q, &w);
        rnea_bias(m, root, qd, p->gravity, &w);
        crba(m, &w);
        real rhs[MAXV], v[MAXV], dv[MAXV];
        for (int i = 0; i < off; i++) rhs[i] = -w.bias[i];
        for (int d = 0; d < nd; d++) {
            real K = m->dof_stiffness[d], D = m->dof_damping[d];
            w.M[off + d][off + d] += m->dof_armature[d] + h * D + h * h * K;
            rhs[off + d] = tau[d] - w.bias[off + d] - K * (q[d] - m->dof_springref[d]) - (D + h * K) * qd[d];
            if (ex && ex->target) {   /* implicit PD drive: same linearisation as the passive spring/damper */
                real kp = EX_KP(ex, d), kd = EX_KD(ex, d);
                w.M[off + d][off + d] += h * kd + h * h * kp;
                rhs[off + d] += kp * (ex->target[d] - q[d]) - (kd + h * kp) * qd[d];
            }
        }
        if (ex && ex->fext) {         /* generalised force J^T f of every externally forced body */
            for (int b = 0; b < m->nb; b++) {
                const real *fl = ex->fext + 3 * b;
                if (fl[0] == 0 && fl[1] == 0 && fl[2] == 0) continue;
                real fw[3], cw[3], xc[3], Jr[MAXV];
                m3v(w.R[b], fl, fw);
                m3v(w.R[b], m->com + 3 * b, cw);
                for (int k = 0; k < 3; k++) xc[k] = w.r[b][k] + cw[k];
                const real dirs[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
                for (int k = 0; k < 3; k++) {
                    point_jac(m, &w, b, xc, dirs[k], Jr);
                    for (int i = 0; i < nv; i++) rhs[i] += Jr[i] * fw[k];
                }
            }
        }
        chol(nv, w.M, w.L);
        chol_solve(nv, w.L, rhs, dv);
        if (!m->fixed_base) { for (int k = 0; k < 3; k++) { v[k] = root[7 + k]; v[3 + k] = root[10 + k]; } }
        for (int d = 0; d < nd; d++) v[off + d] = qd[d];
        for (int i = 0; i < nv; i++) v[i] += h * dv[i];

        /* ---------------- constraint rows */
        static _Thread_local real J[MAXROWS][MAXV], B[MAXROWS][MAXV];
        static _Thread_local real Ainv[MAXROWS], vt[MAXROWS], lam[MAXROWS];
        int nrow = 0;
        int lim_row[MAXD], sph_row[MAXS], grp_row[MAXG], grp_sel[MAXG];
        real lim_sign[MAXD], grp_x[MAXG][3], grp_fr[MAXG][3][3], grp_dist[MAXG];
        int nground = 0, ngrp = 0, dropped = 0, nblkc[16] = {0};
        for (int d = 0; d < nd; d++) {
            lim_row[d] = -1;
            if (!m->dof_limited[d]) { lam_l[d] = 0; continue; }
            real lo = m->dof_lower[d], up = m->dof_upper[d];
            real dl = q[d] - lo, du = up - q[d], C, s;
            if (dl < du) { C = dl; s = 1; } else { C = du; s = -1; }
            /* warm-start impulse only survives if the same side stays active */
            if (lam_l[d] * s < 0) lam_l[d] = 0;
            int r = nrow++;
            lim_row[d] = r; lim_sign[d] = s;
            for (int i = 0; i < nv; i++) J[r][i] = 0;
            J[r][off + d] = s;
            vt[r] = (C >= 0) ? -C / h : fmin(-C * p->erp / h, p->max_depen_vel);
            lam[r] = RFABS(lam_l[d]) * p->warm;
        }
        for (int s = 0; s < m->nsph; s++) {
            sph_row[s] = -1;
            int b = m->sph_body[s];
            real t[3], x[3];
            m3v(w.R[b], m->sph_pos + 3 * s, t);
            x[0] = w.r[b][0] + t[0]; x[1] = w.r[b][1] + t[1]; x[2] = w.r[b][2] + t[2];
            real dirs[3][3], dist;
            /* distance of the sphere to the local tangent plane of the surface, or to the wall of a riser beside it */
            ground_contact(gnd, p->ground_z, root[0] + x[0], root[1] + x[1], root[2] + x[2], m->sph_rad[s], &dist, dirs[0]);
            contact_frame(dirs[0], dirs[1], dirs[2]);
            if (dist >= p->contact_offset) { lam_c[3 * s] = lam_c[3 * s + 1] = lam_c[3 * s + 2] = 0; continue; }
            if (m->kmax > 0 && nground >= m->kmax) { lam_c[3 * s] = lam_c[3 * s + 1] = lam_c[3 * s + 2] = 0; dropped++; continue; }
            if (m->solver == 1 && m->kmax_blk) {
                int blk = m->body_block[b];
                if (nblkc[blk] >= m->kmax_blk[blk]) { lam_c[3 * s] = lam_c[3 * s + 1] = lam_c[3 * s + 2] = 0; dropped++; continue; }
                nblkc[blk]++;
            }
            nground++;
            real xc[3] = {x[0] - m->sph_rad[s] * dirs[0][0], x[1] - m->sph_rad[s] * dirs[0][1], x[2] - m->sph_rad[s] * dirs[0][2]};
            real gap = dist - p->rest_offset;
            sph_row[s] = nrow;
            for (int k = 0; k < 3; k++) {
                int r = nrow++;
                point_jac(m, &w, b, xc, dirs[k], J[r]);
                vt[r] = (k == 0) ? ((gap >= 0) ? -gap / h : fmin(-gap * p->erp / h, p->max_depen_vel)) : 0;
                lam[r] = (m->warm_slots > 0 && nground > m->warm_slots) ? 0 : lam_c[3 * s + k] * p->warm;
            }
        }
        /* self-collision: per group the deepest capsule pair; one contact (normal + 2 tangents) between the two bodies */
        for (int g = 0; g < m->npg; g++) {
            grp_row[g] = -1; grp_sel[g] = -1; grp_dist[g] = 0;
            real best = 0, bca[3] = {0, 0, 0}, bcb[3] = {0, 0, 0};
            int sel = -1;
            for (int k = m->pg_first[g]; k < m->pg_first[g] + m->pg_count[g]; k++) {
                int ia = m->gp_a[k], ib = m->gp_b[k], ba = m->cap_body[ia], bb = m->cap_body[ib];
                real a0[3], a1[3], b0[3], b1[3], t[3], ca[3], cb[3];
                m3v(w.R[ba], m->cap_p0 + 3 * ia, t); for (int c = 0; c < 3; c++) a0[c] = w.r[ba][c] + t[c];
                m3v(w.R[ba], m->cap_p1 + 3 * ia, t); for (int c = 0; c < 3; c++) a1[c] = w.r[ba][c] + t[c];
                m3v(w.R[bb], m->cap_p0 + 3 * ib, t); for (int c = 0; c < 3; c++) b0[c] = w.r[bb][c] + t[c];
                m3v(w.R[bb], m->cap_p1 + 3 * ib, t); for (int c = 0; c < 3; c++) b1[c] = w.r[bb][c] + t[c];
                seg_seg_closest(a0, a1, b0, b1, ca, cb);
                real dv3[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
                real dist = RSQRT(v3dot(dv3, dv3)) - m->cap_rad[ia] - m->cap_rad[ib];
                if (sel < 0 || dist < best) { best = dist; sel = k; memcpy(bca, ca, sizeof(bca)); memcpy(bcb, cb, sizeof(bcb)); }
            }
            if (sel < 0 || best >= p->contact_offset) { lam_p[3 * g] = lam_p[3 * g + 1] = lam_p[3 * g + 2] = 0; continue; }
            if (m->kpair > 0 && ngrp >= m->kpair) { lam_p[3 * g] = lam_p[3 * g + 1] = lam_p[3 * g + 2] = 0; dropped++; continue; }
            ngrp++;
            int ia = m->gp_a[sel], ib = m->gp_b[sel], ba = m->cap_body[ia], bb = m->cap_body[ib];
            real dv3[3] = {bca[0] - bcb[0], bca[1] - bcb[1], bca[2] - bcb[2]};
            real d = RSQRT(v3dot(dv3, dv3));
            real (*fr)[3] = grp_fr[g];
            if (d > (real)1e-9) { fr[0][0] = dv3[0] / d; fr[0][1] = dv3[1] / d; fr[0][2] = dv3[2] / d; }
            else { fr[0][0] = 0; fr[0][1] = 0; fr[0][2] = 1; }
            contact_frame(fr[0], fr[1], fr[2]);
            /* contact point: middle of the gap (or of the overlap) on the line between the closest points */
            for (int c = 0; c < 3; c++) grp_x[g][c] = bcb[c] + fr[0][c] * (m->cap_rad[ib] + (real)0.5 * best);
            grp_sel[g] = sel; grp_dist[g] = best;
            real gap = best - p->rest_offset;
            grp_row[g] = nrow;
            for (int k = 0; k < 3; k++) {
                int r = nrow++;
                real Jb[MAXV];
                point_jac(m, &w, ba, grp_x[g], fr[k], J[r]);
                point_jac(m, &w, bb, grp_x[g], fr[k], Jb);
                for (int i = 0; i < nv; i++) J[r][i] -= Jb[i];
                vt[r] = (k == 0) ? ((gap >= 0) ? -gap / h : fmin(-gap * p->erp / h, p->max_depen_vel)) : 0;
                lam[r] = lam_p[3 * g + k] * p->warm;
            }
        }
        if (m->solver == 1) {
            /* units in the canonical order: limit rows by dof, ground contacts by sphere, self contacts by group */
            int nunit = 0;
            static _Thread_local int u_row[MAXROWS], u_kind[MAXROWS], u_blk[MAXROWS], u_ga[MAXROWS], u_gb[MAXROWS];
            static _Thread_local real u_mu[MAXROWS];
            int bgrp[MAXB];
            for (int b = 0; b < m->nb; b++) {
                bgrp[b] = b == 0 ? 0 : bgrp[m->parent[b]];
                for (int d = 0; d < nd; d++) if (m->dof_body[d] == b) bgrp[b] = m->gi_group[off + d];
            }
            for (int d = 0; d < nd; d++) {
                int r = lim_row[d];
                if (r < 0) continue;
                u_row[nunit] = r; u_kind[nunit] = 0; u_blk[nunit] = m->body_block[m->dof_body[d]];
                u_mu[nunit] = 0; u_ga[nunit] = u_gb[nunit] = -1;
                nunit++;
            }
            for (int s = 0; s < m->nsph; s++) {
                if (sph_row[s] < 0) continue;
                u_row[nunit] = sph_row[s]; u_kind[nunit] = 1; u_blk[nunit] = m->body_block[m->sph_body[s]];
                u_mu[nunit] = (real)0.5 * ((mu_env >= 0 ? mu_env : m->sph_mu[s]) + p->plane_mu); u_ga[nunit] = u_gb[nunit] = -1;
                nunit++;
            }
            for (int g = 0; g < m->npg; g++) {
                if (grp_row[g] < 0) continue;
                u_row[nunit] = grp_row[g]; u_kind[nunit] = 1; u_blk[nunit] = m->nblk - 1;
                u_mu[nunit] = (real)0.5 * (m->cap_mu[m->gp_a[grp_sel[g]]] + m->cap_mu[m->gp_b[grp_sel[g]]]);
                u_ga[nunit] = bgrp[m->cap_body[m->gp_a[grp_sel[g]]]]; u_gb[nunit] = bgrp[m->cap_body[m->gp_b[grp_sel[g]]]];
                nunit++;
            }
            solve_blocks(m, p, &w, nv, nrow, J, vt, lam, v, nunit, u_row, u_kind, u_blk, u_mu, u_ga, u_gb);
        } else {
        for (int r = 0; r < nrow; r++) {
            chol_solve(nv, w.L, J[r], B[r]);
            real a = p->cfm;
            for (int i = 0; i < nv; i++) a += J[r][i] * B[r][i];
            Ainv[r] = 1 / a;
            if (lam[r] != 0) for (int i = 0; i < nv; i++) v[i] += B[r][i] * lam[r];
        }
        for (int it = 0; it < p->iters; it++) {
            for (int d = 0; d < nd; d++) {
                int r = lim_row[d];
                if (r < 0) continue;
                real vn = 0;
                for (int i = 0; i < nv; i++) vn += J[r][i] * v[i];
                real nl = lam[r] - (vn - vt[r]) * Ainv[r];
                if (nl < 0) nl = 0;
                real dl = nl - lam[r];
                lam[r] = nl;
                for (int i = 0; i < nv; i++) v[i] += B[r][i] * dl;
            }
            for (int s = 0; s < m->nsph; s++) {
                int r0 = sph_row[s];
                if (r0 < 0) continue;
                real mu = (real)0.5 * ((mu_env >= 0 ? mu_env : m->sph_mu[s]) + p->plane_mu);
                { /* normal */
                    int r = r0;
                    real vn = 0;
                    for (int i = 0; i < nv; i++) vn += J[r][i] * v[i];
                    real nl = lam[r] - (vn - vt[r]) * Ainv[r];
                    if (nl < 0) nl = 0;
                    real dl = nl - lam[r];
                    lam[r] = nl;
                    for (int i = 0; i < nv; i++) v[i] += B[r][i] * dl;
                }
                real lt[2], vtan[2];
                for (int k = 1; k <= 2; k++) { /* both tangential corrections from the same velocity (see solve_blocks) */
                    int r = r0 + k;
                    real vn = 0;
                    for (int i = 0; i < nv; i++) vn += J[r][i] * v[i];
                    vtan[k - 1] = vn - vt[r];
                    lt[k - 1] = lam[r] - vtan[k - 1] * Ainv[r];
                }
                /* project onto the friction disc |lt| <= mu * ln, apply once.  A contact that SLIDES (the per-row step leaves the disc) repeats the step
                   with ONE step size for both rows before it is scaled back: the fixed point of "row step, radial projection" has the friction
                   antiparallel to D^-1 v (D = the rows' diagonal), which is Coulomb's law only when D is a multiple of the identity (friction_step) */
                real lim = mu * lam[r0];
                friction_step(lt, lam[r0 + 1], lam[r0 + 2], vtan, Ainv[r0 + 1], Ainv[r0 + 2], lim);
                real sc = 1;
                for (int k = 1; k <= 2; k++) {
                    int r = r0 + k;
                    real nl = lt[k - 1] * sc, dl = nl - lam[r];
                    lam[r] = nl;
                    if (dl != 0) for (int i = 0; i < nv; i++) v[i] += B[r][i] * dl;
                }
            }
            for (int g = 0; g < m->npg; g++) {   /* self-collision contacts: same update, friction = mean of the two shapes */
                int r0 = grp_row[g];
                if (r0 < 0) continue;
                real mu = (real)0.5 * (m->cap_mu[m->gp_a[grp_sel[g]]] + m->cap_mu[m->gp_b[grp_sel[g]]]);
                {
                    int r = r0;
                    real vn = 0;
                    for (int i = 0; i < nv; i++) vn += J[r][i] * v[i];
                    real nl = lam[r] - (vn - vt[r]) * Ainv[r];
                    if (nl < 0) nl = 0;
                    real dl = nl - lam[r];
                    lam[r] = nl;
                    for (int i = 0; i < nv; i++) v[i] += B[r][i] * dl;
                }
                real lt[2], vtan[2];
                for (int k = 1; k <= 2; k++) { /* both tangential corrections from the same velocity (see solve_blocks) */
                    int r = r0 + k;
                    real vn = 0;
                    for (int i = 0; i < nv; i++) vn += J[r][i] * v[i];
                    vtan[k - 1] = vn - vt[r];
                    lt[k - 1] = lam[r] - vtan[k - 1] * Ainv[r];
                }
                /* project onto the friction disc |lt| <= mu * ln, apply once.  A contact that SLIDES (the per-row step leaves the disc) repeats the step
                   with ONE step size for both rows before it is scaled back: the fixed point of "row step, radial projection" has the friction
                   antiparallel to D^-1 v (D = the rows' diagonal), which is Coulomb's law only when D is a multiple of the identity (friction_step) */
                real lim = mu * lam[r0];
                friction_step(lt, lam[r0 + 1], lam[r0 + 2], vtan, Ainv[r0 + 1], Ainv[r0 + 2], lim);
                real sc = 1;
                for (int k = 1; k <= 2; k++) {
                    int r = r0 + k;
                    real nl = lt[k - 1] * sc, dl = nl - lam[r];
                    lam[r] = nl;
                    if (dl != 0) for (int i = 0; i < nv; i++) v[i] += B[r][i] * dl;
                }
            }
        }
        }
        /* ---------------- write back impulses, sensors */
        for (int d = 0; d < nd; d++) {
            real ll = 0;
            if (lim_row[d] >= 0) { ll = lam[lim_row[d]] * lim_sign[d]; lam_l[d] = ll; }
            dof_force[d] = tau[d] - m->dof_stiffness[d] * (q[d] - m->dof_springref[d]) - m->dof_damping[d] * v[off + d] + ll / h;
            if (ex && ex->target) dof_force[d] += EX_KP(ex, d) * (ex->target[d] - q[d]) - EX_KD(ex, d) * v[off + d];
        }
        for (int k = 0; k < 6 * m->nsens; k++) sensor[k] = 0;
        if (netf) for (int k = 0; k < 3 * m->nb; k++) netf[k] = 0;
        for (int s = 0; s < m->nsph; s++) {
            real f[3] = {0, 0, 0}, nrm[3] = {0, 0, 1};
            int b = m->sph_body[s];
            if (sph_row[s] >= 0) {
                int r0 = sph_row[s];
                lam_c[3 * s] = lam[r0]; lam_c[3 * s + 1] = lam[r0 + 1]; lam_c[3 * s + 2] = lam[r0 + 2];
                real t[3], x[3], dd, t1[3], t2[3];
                m3v(w.R[b], m->sph_pos + 3 * s, t);
                x[0] = w.r[b][0] + t[0]; x[1] = w.r[b][1] + t[1]; x[2] = w.r[b][2] + t[2];
                ground_contact(gnd, p->ground_z, root[0] + x[0], root[1] + x[1], root[2] + x[2], m->sph_rad[s], &dd, nrm);
                contact_frame(nrm, t1, t2);
                for (int c = 0; c < 3; c++) f[c] = (nrm[c] * lam[r0] + t1[c] * lam[r0 + 1] + t2[c] * lam[r0 + 2]) / h;
                if (netf) for (int c = 0; c < 3; c++) netf[3 * b + c] += f[c];
            }
            sph_force[3 * s] = f[0]; sph_force[3 * s + 1] = f[1]; sph_force[3 * s + 2] = f[2];
            if (sph_row[s] < 0) continue;
            for (int k = 0; k < m->nsens; k++) {
                if (m->sens_body[k] != b) continue;
                real t[3], x[3], arm[3], tq[3], fl[3], tl[3];
                m3v(w.R[b], m->sph_pos + 3 * s, t);
                x[0] = w.r[b][0] + t[0] - m->sph_rad[s] * nrm[0]; x[1] = w.r[b][1] + t[1] - m->sph_rad[s] * nrm[1];
                x[2] = w.r[b][2] + t[2] - m->sph_rad[s] * nrm[2];
                arm[0] = x[0] - w.r[b][0]; arm[1] = x[1] - w.r[b][1]; arm[2] = x[2] - w.r[b][2];
                v3cross(arm, f, tq);
                m3tv(w.R[b], f, fl); m3tv(w.R[b], tq, tl);
                for (int c = 0; c < 3; c++) { sensor[6 * k + c] += fl[c]; sensor[6 * k + 3 + c] += tl[c]; }
            }
        }
        for (int g = 0; g < m->npg; g++) {
            real f[3] = {0, 0, 0};
            if (grp_row[g] >= 0) {
                int r0 = grp_row[g];
                lam_p[3 * g] = lam[r0]; lam_p[3 * g + 1] = lam[r0 + 1]; lam_p[3 * g + 2] = lam[r0 + 2];
                for (int c = 0; c < 3; c++)
                    f[c] = (grp_fr[g][0][c] * lam[r0] + grp_fr[g][1][c] * lam[r0 + 1] + grp_fr[g][2][c] * lam[r0 + 2]) / h;
                int bs[2] = {m->cap_body[m->gp_a[grp_sel[g]]], m->cap_body[m->gp_b[grp_sel[g]]]};
                for (int side = 0; side < 2; side++) {     /* force sensors see +f on side a, -f on side b */
                    int b = bs[side];
                    real sg = side == 0 ? 1 : -1;
                    if (netf) for (int c = 0; c < 3; c++) netf[3 * b + c] += sg * f[c];
                    for (int k = 0; k < m->nsens; k++) {
                        if (m->sens_body[k] != b) continue;
                        real arm[3] = {grp_x[g][0] - w.r[b][0], grp_x[g][1] - w.r[b][1], grp_x[g][2] - w.r[b][2]}, fs[3] = {sg * f[0], sg * f[1], sg * f[2]};
                        real tq[3], fl[3], tl[3];
                        v3cross(arm, fs, tq);
                        m3tv(w.R[b], fs, fl); m3tv(w.R[b], tq, tl);
                        for (int c = 0; c < 3; c++) { sensor[6 * k + c] += fl[c]; sensor[6 * k + 3 + c] += tl[c]; }
                    }
                }
            }
            if (pair_out) {
                real *po = pair_out + 9 * g;   /* world force on side a, selected capsule pair (-1: none), its distance, #dropped, contact point rel. O */
                po[0] = f[0]; po[1] = f[1]; po[2] = f[2]; po[3] = (real)grp_sel[g]; po[4] = grp_dist[g]; po[5] = (real)dropped;
                for (int c = 0; c < 3; c++) po[6 + c] = grp_row[g] >= 0 ? grp_x[g][c] : 0;
            }
        }
        /* ---------------- integrate */
        for (int d = 0; d < nd; d++) { qd[d] = v[off + d]; q[d] += h * qd[d]; }
        if (!m->fixed_base) {
            {   /* gymapi.AssetOptions defaults max_angular_velocity = 64 rad/s, max_linear_velocity = 1000 m/s (the tasks restated here
                 * set neither): the simulator clamps the actor's velocities */
                real w2 = v[3] * v[3] + v[4] * v[4] + v[5] * v[5], l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
                real sw = w2 > (real)(64.0 * 64.0) ? (real)64.0 / RSQRT(w2) : 1, sl = l2 > (real)(1000.0 * 1000.0) ? (real)1000.0 / RSQRT(l2) : 1;
                v[0] *= sl; v[1] *= sl; v[2] *= sl; v[3] *= sw; v[4] *= sw; v[5] *= sw;
            }
            for (int k = 0; k < 3; k++) { root[7 + k] = v[k]; root[10 + k] = v[3 + k]; root[k] += h * v[k]; }
            real om[3] = {v[3], v[4], v[5]};
            real an = RSQRT(v3dot(om, om)), th = an * h;
            real dq[4];
            if (th > (real)1e-12) {
                real s = RSIN(th / 2) / an;
                dq[0] = om[0] * s; dq[1] = om[1] * s; dq[2] = om[2] * s; dq[3] = RCOS(th / 2);
            } else { dq[0] = om[0] * h / 2; dq[1] = om[1] * h / 2; dq[2] = om[2] * h / 2; dq[3] = 1; }
            real *Q = root + 3;
            real x = dq[3] * Q[0] + dq[0] * Q[3] + dq[1] * Q[2] - dq[2] * Q[1];
            real y = dq[3] * Q[1] - dq[0] * Q[2] + dq[1] * Q[3] + dq[2] * Q[0];
            real z = dq[3] * Q[2] + dq[0] * Q[1] - dq[1] * Q[0] + dq[2] * Q[3];
            real ww = dq[3] * Q[3] - dq[0] * Q[0] - dq[1] * Q[1] - dq[2] * Q[2];
            real n = 1 / RSQRT(x * x + y * y + z * z + ww * ww);
            Q[0] = x * n; Q[1] = y * n; Q[2] = z * n; Q[3] = ww * n;
        }
    }
}

/* ------------------------------------------------------------------ exported API (ctypes) */
int or_state_size(const OrModel *m) { return 13 + 2 * m->nd + 3 * m->nsph + m->nd + 3 * m->npg; }
int or_out_size(const OrModel *m) { return 6 * m->nsens + m->nd + 3 * m->nsph + 9 * m->npg; }

void or_step(const OrModel *m, const OrParams *p, int nenv, real *state, const real *tau, real *out) {
    int ss = or_state_size(m), os = or_out_size(m), nd = m->nd;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nenv; e++) {
        real *s = state + (size_t)e * ss, *o = out + (size_t)e * os;
        step_env(m, p, 0, (real)-1, s, s + 13, s + 13 + nd, s + 13 + 2 * nd, s + 13 + 2 * nd + 3 * m->nsph,
                 s + 13 + 3 * nd + 3 * m->nsph, tau + (size_t)e * nd, o, o + 6 * m->nsens, o + 6 * m->nsens + nd,
                 o + 6 * m->nsens + nd + 3 * m->nsph, 0, 0);
    }
}

/* as or_step, on a height field (gnd may be NULL), with per-env friction env_mu[nenv] (NULL = model friction) and the
 * per-body net contact forces netf[nenv][3*nb] (NULL = not wanted) */
void or_step_ex(const OrModel *m, const OrParams *p, const OrGround *gnd, const real *env_mu, int nenv, real *state,
                const real *tau, real *out, real *netf) {
    int ss = or_state_size(m), os = or_out_size(m), nd = m->nd;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nenv; e++) {
        real *s = state + (size_t)e * ss, *o = out + (size_t)e * os;
        step_env(m, p, gnd, env_mu ? env_mu[e] : (real)-1, s, s + 13, s + 13 + nd, s + 13 + 2 * nd,
                 s + 13 + 2 * nd + 3 * m->nsph, s + 13 + 3 * nd + 3 * m->nsph, tau + (size_t)e * nd, o, o + 6 * m->nsens,
                 o + 6 * m->nsens + nd, o + 6 * m->nsens + nd + 3 * m->nsph, netf ? netf + (size_t)e * 3 * m->nb : 0, 0);
    }
}

/* as or_step, with PD position drives (kp, kd, target[nenv][nd]; target may be NULL) and local-frame external body forces
 * fext[nenv][3*nb] (may be NULL) */
void or_step_drive(const OrModel *m, const OrParams *p, int nenv, real *state, const real *tau, real *out, real kp, real kd,
                   const real *target, const real *fext) {
    int ss = or_state_size(m), os = or_out_size(m), nd = m->nd;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nenv; e++) {
        real *s = state + (size_t)e * ss, *o = out + (size_t)e * os;
        OrExtra ex = {kp, kd, target ? target + (size_t)e * nd : 0, fext ? fext + (size_t)e * 3 * m->nb : 0, 0, 0};
        step_env(m, p, 0, (real)-1, s, s + 13, s + 13 + nd, s + 13 + 2 * nd, s + 13 + 2 * nd + 3 * m->nsph,
                 s + 13 + 3 * nd + 3 * m->nsph, tau + (size_t)e * nd, o, o + 6 * m->nsens, o + 6 * m->nsens + nd,
                 o + 6 * m->nsens + nd + 3 * m->nsph, 0, &ex);
    }
}

/* as or_step_drive with one gain pair per dof (kpv[nd], kdv[nd], the same for every env) and the net contact force per body */
void or_step_drive_v(const OrModel *m, const OrParams *p, int nenv, real *state, const real *tau, real *out, const real *kpv,
                     const real *kdv, const real *target, real *netf) {
    int ss = or_state_size(m), os = or_out_size(m), nd = m->nd;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nenv; e++) {
        real *s = state + (size_t)e * ss, *o = out + (size_t)e * os;
        OrExtra ex = {0, 0, target + (size_t)e * nd, 0, kpv, kdv};
        step_env(m, p, 0, (real)-1, s, s + 13, s + 13 + nd, s + 13 + 2 * nd, s + 13 + 2 * nd + 3 * m->nsph,
                 s + 13 + 3 * nd + 3 * m->nsph, tau + (size_t)e * nd, o, o + 6 * m->nsens, o + 6 * m->nsens + nd,
                 o + 6 * m->nsens + nd + 3 * m->nsph, netf ? netf + (size_t)e * 3 * m->nb : 0, &ex);
    }
}

/* dense M (nv*nv row-major) and bias (nv) for one env: used to cross-check the HIP host build */
void or_dynamics(const OrModel *m, const OrParams *p, const real *state, real *Mout, real *bias_out) {
    static _Thread_local Work w;
    int nv = nvof(m);
    fk(m, state, state + 13, &w);
    rnea_bias(m, state, state + 13 + m->nd, p->gravity, &w);
    crba(m, &w);
    for (int i = 0; i < nv; i++) { bias_out[i] = w.bias[i]; for (int j = 0; j < nv; j++) Mout[i * nv + j] = w.M[i][j]; }
}

/* kinetic + potential energy and world-frame body poses (pos[3]+R[9] per body), for known-answer tests */
void or_energy(const OrModel *m, const OrParams *p, const real *state, real *ke, real *pe, real *body_pose) {
    static _Thread_local Work w;
    const real zero[3] = {0, 0, 0};
    fk(m, state, state + 13, &w);
    rnea_bias(m, state, state + 13 + m->nd, zero, &w);
    real K = 0, P = 0;
    for (int b = 0; b < m->nb; b++) {
        real IV[6];
        spi_mul(&w.I[b], w.V[b], IV);
        for (int k = 0; k < 6; k++) K += (real)0.5 * w.V[b][k] * IV[k];
        real c[3] = {w.I[b].h[0], w.I[b].h[1], w.I[b].h[2]}; /* m*c rel O */
        real O[3] = {state[0], state[1], state[2]};
        for (int k = 0; k < 3; k++) P -= p->gravity[k] * (c[k] + m->mass[b] * O[k]);
        if (body_pose) {
            for (int k = 0; k < 3; k++) body_pose[12 * b + k] = O[k] + w.r[b][k];
            for (int k = 0; k < 9; k++) body_pose[12 * b + 3 + k] = w.R[b][k];
        }
    }
    *ke = K; *pe = P;
}

/* linear Jacobian (3 x nv, rows = world x, y, z) of a point fixed on body b, given relative to O (oracle/hand.py) */
void or_point_jac(const OrModel *m, const real *state, int b, const real *xc, real *J3) {
    static _Thread_local Work w;
    int nv = nvof(m);
    fk(m, state, state + 13, &w);
    const real dirs[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int k = 0; k < 3; k++) point_jac(m, &w, b, xc, dirs[k], J3 + k * nv);
}

/* spatial velocity [omega; v_O] of body b about O, world axes (oracle/hand.py: fingertip states) */
void or_body_vel(const OrModel *m, const real *state, int b, real *out6) {
    static _Thread_local Work w;
    const real zero[3] = {0, 0, 0};
    fk(m, state, state + 13, &w);
    rnea_bias(m, state, state + 13 + m->nd, zero, &w);
    for (int k = 0; k < 6; k++) out6[k] = w.V[b][k];
}

void or_seg_seg_closest(const real *a0, const real *a1, const real *b0, const real *b1, real *ca, real *cb) {
    seg_seg_closest(a0, a1, b0, b1, ca, cb);
}
int or_sizeof_real(void) { return (int)sizeof(real); }
