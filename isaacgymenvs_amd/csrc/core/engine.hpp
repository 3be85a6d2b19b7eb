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
#include <type_traits>

#if defined(MI_TIMING) && defined(__HIP_DEVICE_COMPILE__)
// tools/debug only: s_memtime stamps at the phase boundaries of a sub-step (never defined in the product build)
#define MI_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
                         if (tstamp) tstamp[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#elif defined(MI_MARKERS) && defined(__HIP_DEVICE_COMPILE__)
// tools/debug only: assembler comments at the phase boundaries (tools/debug/phasecount.py counts spill traffic per phase)
#define MI_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MI_MARK %0" ::"n"(i)); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MI_STAMP(i) do { } while (0)
#endif
#if defined(__HIPCC__)
#define MI_HD __host__ __device__ __forceinline__
#define MI_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#define MI_LAMBDA __attribute__((always_inline))
#if defined(__HIP_DEVICE_COMPILE__)
// nothing may be scheduled across this point: keeps the phases of the (one-basic-block) sub-step apart so the
// machine scheduler cannot stretch live ranges across them
#define MI_PHASE() __builtin_amdgcn_sched_barrier(0)
// an integer zero the optimiser cannot see through: added to the row-store base inside the PGS loop so that LICM
// does not hoist the (loop-invariant) loads of all G rows out of the sweep loop and keep them live in registers
#define MI_OPAQUE_ZERO(z) asm volatile("s_mov_b32 %0, 0" : "=s"(z))
// wave-uniform "does any env of this wavefront ...": lets the whole wave branch around work no lane needs
#define MI_WAVE_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
// a value the optimiser cannot see through (uniform pointer / per-lane float): re-defined inside a loop body, nothing computed from it can be
// hoisted out of the loop (the fused sub-steps: hundreds of loop-invariant expressions of the step size and the sim parameters)
#define MI_OPAQUE_SPTR(p) asm volatile("" : "+s"(p))
#define MI_OPAQUE_SINT(x) asm volatile("" : "+s"(x))
#define MI_OPAQUE_VF(x) asm volatile("" : "+v"(x))
#ifndef MI_EXACT_SINCOS
// joint rotations in the tree pass: hardware v_sin_f32 / v_cos_f32 (input in revolutions; ~1e-6 absolute error for |q| <= pi, the
// same order as the 1-ulp v_rcp / v_rsq the solver already uses) instead of libm's range-reducing sincosf (~40 instructions per
// joint).  Measured: Ant step -3..6 %, AnymalTerrain -8 %, ShadowHand -3.5 %.  -DMI_EXACT_SINCOS restores sincosf.
#define MI_SINCOS(x, s, c) do { const float _r = (x) * 0.15915494309189535f; *(s) = __builtin_amdgcn_sinf(_r); *(c) = __builtin_amdgcn_cosf(_r); } while (0)
#endif
#else
#define MI_PHASE() do { } while (0)
#define MI_OPAQUE_ZERO(z) (z) = 0
#define MI_WAVE_ANY(x) (x)
#define MI_OPAQUE_SPTR(p) do { } while (0)
#define MI_OPAQUE_SINT(x) do { } while (0)
#define MI_OPAQUE_VF(x) do { } while (0)
#endif
#else
#define MI_PHASE() do { } while (0)
#define MI_OPAQUE_ZERO(z) (z) = 0
#define MI_WAVE_ANY(x) (x)
#define MI_OPAQUE_SPTR(p) do { } while (0)
#define MI_OPAQUE_SINT(x) do { } while (0)
#define MI_OPAQUE_VF(x) do { } while (0)
#define MI_HD inline __attribute__((always_inline))
#define MI_HD_NOINLINE __attribute__((noinline))
#define MI_LAMBDA __attribute__((always_inline))
#endif

#ifndef MI_SINCOS
#define MI_SINCOS(x, s, c) sincosf((x), (s), (c))
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

// Constraint-row store.  On the device the rows of the 64 envs of a wave live in LDS as [slot][lane] (one
// ds_read/ds_write per access, conflict-free, addresses are compile-time offsets from a per-lane base); models
// whose rows do not fit the 160 KB of LDS (Humanoid) and the host build use a private array with stride 1.
template <int STRIDE>
struct RowStore {
    float* p;
    MI_HD float& operator()(int slot) const { return p[slot * STRIDE]; }
    MI_HD RowStore shifted(int off) const { return RowStore{p + off}; }
    MI_HD float* ptr(int slot) const { return p + slot * STRIDE; }   // data-dependent slot (compact contact store)
    static constexpr int stride = STRIDE;
};
// LDS flavour: slot k of this lane sits at byte k*256 + lane*4.  DS instructions only encode a 16-bit byte offset,
// so one base register per 64 KB segment is kept (3 VGPRs for up to 192 KB); with the segment picked at compile
// time every access is `ds_read/ds_write base_seg offset:imm`, and neighbouring slots pair into ds_read2st64_b32
// (whose offset unit is exactly our 256-byte slot stride).
template <>
struct RowStore<64> {
    float* seg[3];
    MI_HD explicit RowStore(float* p) : seg{p, p + 256 * 64, p + 512 * 64} {}
    MI_HD float& operator()(int slot) const { return seg[slot >> 8][(slot & 255) * 64]; }
    MI_HD RowStore shifted(int off) const { return RowStore(seg[0] + off); }
    MI_HD float* ptr(int slot) const { return seg[0] + slot * 64; }
    static constexpr int stride = 64;
};

// reciprocal / reciprocal square root / square root: the 1-ulp hardware ops on the device (v_rcp_f32, v_rsq_f32,
// v_sqrt_f32) instead of the 10-15 instruction IEEE sequences -- these sit on the per-row critical path of the solver
#if defined(__HIP_DEVICE_COMPILE__)
#define MI_RCP(x) __builtin_amdgcn_rcpf(x)
#define MI_RSQ(x) __builtin_amdgcn_rsqf(x)
#define MI_SQRT(x) __builtin_amdgcn_sqrtf(x)
#else
#define MI_RCP(x) (1.f / (x))
#define MI_RSQ(x) (1.f / sqrtf(x))
#define MI_SQRT(x) sqrtf(x)
#endif

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

// ---- ground policies.  PlaneGround: the z = ground_z plane of the Ant / Humanoid / Cartpole tasks (reference
// ant.py:128-133).  HeightfieldGround: the rough terrain of AnymalTerrain -- the reference turns an int16 height grid
// into a triangle mesh (anymal_terrain.py:569-575, each cell split along the (i,j)-(i+1,j+1) diagonal; vertex (i,j) at
// world (i*hscale - border, j*hscale - border, h*vscale), :208-210); the surface queried here is that piecewise-linear
// mesh, read straight from the int16 grid.  The mesh generator's slope correction (`slopeTreshold`: the lower vertex of an
// edge steeper than the threshold slides under the upper one, so a stair riser becomes a vertical wall at the upper vertex and
// the lower tread reaches up to it) is applied per query: every edge of the cell that rises by more than `thr` raw height units
// is levelled to its lower end before the cell's two triangles are evaluated -- within 1 mm of the corrected mesh on 97 % of
// the AnymalTerrain map (88 % without; oracle/terrain_mesh.py, tests/test_terrain.py).  The walls themselves collide through
// contact(): a sphere inside a steep cell, below the wall's top, takes the wall as its contact when that is the nearer surface.
// Optional extras of a sub-step (nullptr = none: every call site that passes nullptr compiles to exactly the code it had
// before this existed, the branches below fold away after inlining).
struct Drive {
    float kp, kd;          // implicit PD position drive on every dof (gym DOF_MODE_POS: stiffness, damping)
    const float* target;   // [ND] position targets (gym.set_dof_position_target_tensor)
    const float* fsens;    // [NSENS][3] external force at the centre of mass of each force-sensor body, in that body's own
                           // frame (gym.apply_rigid_body_force_tensors(..., LOCAL_SPACE)), or nullptr
    const float* kpv = nullptr;   // [ND] per-dof gains instead of the one pair above (the Articulation task: every dof its own drive), or nullptr
    const float* kdv = nullptr;
    MI_HD float gain_p(int d) const { return kpv ? kpv[d] : kp; }
    MI_HD float gain_d(int d) const { return kdv ? kdv[d] : kd; }
};
// gymapi.AssetOptions defaults the reference's tasks leave alone (ant.py / humanoid.py / anymal*.py set neither): the simulator clamps
// every actor's linear / angular velocity to these
constexpr float kMaxAngularVelocity = 64.f, kMaxLinearVelocity = 1000.f;
struct PlaneGround {
    static constexpr bool HEIGHTFIELD = false;
    static constexpr bool NETF = false;   // per-body net contact forces (gym.acquire_net_contact_force_tensor) not wanted
};
struct PlaneGroundNF {                     // flat ground, net contact forces reported (Anymal: anymal.py:110)
    static constexpr bool HEIGHTFIELD = false;
    static constexpr bool NETF = true;
};
struct HeightfieldGround {
    static constexpr bool HEIGHTFIELD = true;
    static constexpr bool NETF = true;
    const short* hs;  // [rows * cols], row-major
    int rows, cols;
    float hscale, vscale, border;
    float thr = 3.0e38f;   // slope_threshold * hscale / vscale (raw height units per cell), huge = no correction
    int walls = 1;         // with the correction on: the risers it creates collide from the side (contact()); 0: the surface below only
    // the cell under world (x, y): position inside it and its four raw corner heights h00, h10, h01, h11
    MI_HD void locate(float x, float y, float* fx, float* fy, float* h) const {
        const float gx = (x + border) / hscale, gy = (y + border) / hscale;
        int i = (int)floorf(gx), j = (int)floorf(gy);
        i = i < 0 ? 0 : (i > rows - 2 ? rows - 2 : i);
        j = j < 0 ? 0 : (j > cols - 2 ? cols - 2 : j);
        *fx = fminf(fmaxf(gx - (float)i, 0.f), 1.f); *fy = fminf(fmaxf(gy - (float)j, 0.f), 1.f);
        h[0] = (float)hs[i * cols + j]; h[1] = (float)hs[(i + 1) * cols + j]; h[2] = (float)hs[i * cols + j + 1]; h[3] = (float)hs[(i + 1) * cols + j + 1];
    }
    // surface of the cell: height and unit normal at (fx, fy); hx: the corner heights after the x edges have been levelled
    MI_HD void surface(const float* h, float fx, float fy, float* z, float* n, float* hx) const {
        float h00 = h[0], h10 = h[1], h01 = h[2], h11 = h[3];
        {   // risers: x edges first, then y edges (on the levelled values)
            const float m0 = fminf(h00, h10), m1 = fminf(h01, h11);
            const bool s0 = fabsf(h10 - h00) > thr, s1 = fabsf(h11 - h01) > thr;
            h00 = s0 ? m0 : h00; h10 = s0 ? m0 : h10; h01 = s1 ? m1 : h01; h11 = s1 ? m1 : h11;
            hx[0] = h00; hx[1] = h10; hx[2] = h01; hx[3] = h11;
            const float m2 = fminf(h00, h01), m3 = fminf(h10, h11);
            const bool s2 = fabsf(h01 - h00) > thr, s3 = fabsf(h11 - h10) > thr;
            h00 = s2 ? m2 : h00; h01 = s2 ? m2 : h01; h10 = s3 ? m3 : h10; h11 = s3 ? m3 : h11;
        }
        const bool lower = fx >= fy;
        const float dzx = lower ? h10 - h00 : h11 - h01, dzy = lower ? h11 - h10 : h01 - h00;
        *z = (h00 + dzx * fx + dzy * fy) * vscale;
        const float sx = dzx * vscale / hscale, sy = dzy * vscale / hscale;
        const float inv = 1.f / sqrtf(sx * sx + sy * sy + 1.f);
        n[0] = -sx * inv; n[1] = -sy * inv; n[2] = inv;
    }
    // height z and unit normal n of the surface under world (x, y); same arithmetic as oracle/physics.c ground_query
    MI_HD void query(float x, float y, float* z, float* n) const {
        float fx, fy, h[4], hx[4];
        locate(x, y, &fx, &fy, h);
        surface(h, fx, fy, z, n, hx);
    }
    // Contact of a sphere (centre (x, y, z), radius r <= hscale) with the terrain: distance and unit normal of the NEARER of the tangent
    // plane of the surface below the centre and the wall of a riser in the centre's cell (both x edges, or both y edges, of the cell rise
    // by more than thr in the same direction: floor at the lower level, a vertical wall on the boundary of the higher vertices; a candidate
    // -- its face below the wall's top, the top's edge above it); inside both, one contact along the summed penetration vectors.  Statement and reasoning:
    // oracle/physics.c ground_contact (same arithmetic, branch-free here).
    MI_HD void contact(float x, float y, float z, float r, float* dist, float* n) const {
        float fx, fy, h[4], hx[4], zt;
        locate(x, y, &fx, &fy, h);
        surface(h, fx, fy, &zt, n, hx);
        float d = (z - zt) * n[2] - r;
#if !defined(MI_NO_TERRAIN_WALLS)   // (measurement builds only)
        // (branch-free on purpose: a wave-uniform skip of the wall block when no env is inside a steep cell made the fused AnymalTerrain
        //  sub-step kernel 101 -> 139 us: the block is short, the branch breaks the schedule around every sphere)
        const bool s0 = fabsf(h[1] - h[0]) > thr, s1 = fabsf(h[3] - h[2]) > thr, ux0 = h[1] > h[0], ux1 = h[3] > h[2];
        const bool s2 = fabsf(hx[2] - hx[0]) > thr, s3 = fabsf(hx[3] - hx[1]) > thr, uy0 = hx[2] > hx[0], uy1 = hx[3] > hx[1];
        const bool wx = (walls != 0) && s0 && s1 && (ux0 == ux1), wy = (walls != 0) && s2 && s3 && (uy0 == uy1);
        {
            float dw = 1e30f, nwx = 0.f, nwy = 0.f, nwz = 0.f;     // the nearer wall candidate: the wall's face, above its top the top's edge
            {   // x walls, raw heights
                const float t0 = ux0 ? h[1] : h[0], t1 = ux1 ? h[3] : h[2];
                const float top = (t0 + (t1 - t0) * fy) * vscale;
                // (below the top -- and up to 0.1 mm above it -- the normal is EXACTLY +-x: the output stage re-derives the contact frame from the
                //  normal's x, y parked in the row store, and a normal that is x only to rounding would pick the other tangent pair there)
                const float dx = (ux0 ? 1.f - fx : fx) * hscale, dzr = z - top, dz = dzr > 1e-4f ? dzr : 0.f;
                const float l2 = dx * dx + dz * dz, il = MI_RSQ(fmaxf(l2, 1e-24f)), len = dz > 0.f ? l2 * il : dx;
                const bool use = wx && (len - r < dw);
                dw = use ? len - r : dw;
                nwx = use ? (dz > 0.f ? (ux0 ? -dx : dx) * il : (ux0 ? -1.f : 1.f)) : nwx; nwz = use ? dz * il : nwz;
            }
            {   // y walls, x-levelled heights
                const float t0 = uy0 ? hx[2] : hx[0], t1 = uy1 ? hx[3] : hx[1];
                const float top = (t0 + (t1 - t0) * fx) * vscale;
                const float dy = (uy0 ? 1.f - fy : fy) * hscale, dzr = z - top, dz = dzr > 1e-4f ? dzr : 0.f;
                const float l2 = dy * dy + dz * dz, il = MI_RSQ(fmaxf(l2, 1e-24f)), len = dz > 0.f ? l2 * il : dy;
                const bool use = wy && (len - r < dw);
                dw = use ? len - r : dw;
                nwx = use ? 0.f : nwx; nwy = use ? (dz > 0.f ? (uy0 ? -dy : dy) * il : (uy0 ? -1.f : 1.f)) : nwy; nwz = use ? dz * il : nwz;
            }
            // inside both the surface below and a wall: one contact along the summed penetrations; else the nearer of the two
            const bool both = (d < 0.f) && (dw < 0.f), wall = !both && (dw < d);
            const float vx = -d * n[0] - dw * nwx, vy = -d * n[1] - dw * nwy, vz = -d * n[2] - dw * nwz;
            const float L2 = vx * vx + vy * vy + vz * vz, iL = MI_RSQ(fmaxf(L2, 1e-30f));
            n[0] = both ? vx * iL : (wall ? nwx : n[0]);
            n[1] = both ? vy * iL : (wall ? nwy : n[1]);
            n[2] = both ? vz * iL : (wall ? nwz : n[2]);
            d = both ? -L2 * iL : (wall ? dw : d);
        }
#endif
        *dist = d;
    }
};
// contact frame: n, t1 = normalize(x - n (n.x)), t2 = n x t1   (n = z gives t1 = x, t2 = y)
MI_HD void contact_frame(const float* n, float* t1, float* t2) {
    const float a[3] = {1.f - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
    const float a2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    const bool ok = a2 > 1e-12f;                       // n = +-x (possible for object contacts): fall back to y
    const float inv = 1.f / sqrtf(ok ? a2 : 1.f);
    t1[0] = ok ? a[0] * inv : 0.f; t1[1] = ok ? a[1] * inv : 1.f; t1[2] = ok ? a[2] * inv : 0.f;
    cross3(n, t1, t2);
}

// The tangential update of one contact (every engine form; oracle/physics.c friction_step states the same rule).  lt[] comes in as the per-row
// step r_k = lam_k - v_k / a_kk of the two tangent rows, BOTH taken from the same velocity.  Inside the friction disc (|r| <= lim = mu ln: the
// contact sticks) it stands.  A contact that slides takes ONE step size for both rows instead -- s_k = lam_k - v_k / max(a_11, a_22), never a longer
// step than either row would take alone -- scaled back onto the disc.  Why: the fixed point of "step, radial projection" has the friction
// antiparallel to D^-1 v_t, D = diag(a_11, a_22) -- Coulomb's law only when D is a multiple of the identity.  Until round 6 the rows kept their own
// step sizes (and t1 was applied before t2 was looked at): a Humanoid lying on the ground, sliding at 31 degrees to the tangent axes, was braked along
// 15 degrees; a cube on a 40 degree ramp slid with mu_eff 0.468 for mu 0.5 (tests/friction_util.py, tests/test_scene.py).  Between the two regimes
// (|s| < lim < |r|) the result is the point of the segment s -> r that lies ON the circle, so that the update is a continuous function of its inputs
// (a jump there would let fp32 and fp64 runs part at every stick / slip transition).  Branch-free.
// ISO = false (core/scene_engine.hpp): both rows keep their own step sizes and the result is scaled radially onto the disc -- the scene engine's rule
// since round 5.  The isotropic step was tried there in round 6 and gave the scripted grasp of tests/test_scene.py away: a cube held between the two
// finger pads of the Franka (contacts that should stick, whose rows' diagonals differ by an order of magnitude) slid out during the carry within the
// scene's nine sweeps, where the per-row steps hold it; the scene's sliding known answers (ramp, stop distance) hold to 2 % with either.
template <bool ISO = true>
MI_HD void friction_disc(float (&lt)[2], const float lm1, const float lm2, const float v1, const float v2, const float ainv1, const float ainv2,
                         const float lim) {
    if constexpr (!ISO) {
        const float n2r = lt[0] * lt[0] + lt[1] * lt[1];
        const float scr = (n2r > lim * lim) ? lim * MI_RSQ(fmaxf(n2r, 1e-30f)) : 1.f;
        lt[0] *= scr; lt[1] *= scr;
        return;
    }
    const float l2 = lim * lim;
    const float r0 = lt[0], r1 = lt[1];
    const bool stick = r0 * r0 + r1 * r1 <= l2;
    const float ac = fminf(ainv1, ainv2);
    const float s0 = lm1 - v1 * ac, s1 = lm2 - v2 * ac;
    const float n2 = s0 * s0 + s1 * s1;
    const bool slide = n2 >= l2;
    const float sc = lim * MI_RSQ(fmaxf(n2, 1e-30f));
    // in between: s + t (r - s) with |.| = lim, t in (0, 1]
    const float d0 = r0 - s0, d1 = r1 - s1;
    const float a = fmaxf(d0 * d0 + d1 * d1, 1e-30f), b = s0 * d0 + s1 * d1, c = n2 - l2;
    const float t = (MI_SQRT(fmaxf(b * b - a * c, 0.f)) - b) * MI_RCP(a);
    const float m0 = slide ? s0 * sc : s0 + t * d0, m1 = slide ? s1 * sc : s1 + t * d1;
    lt[0] = stick ? r0 : m0;
    lt[1] = stick ? r1 : m1;
}

// strided view of a per-env vector that lives in HBM as SoA [k][env] (device: stride = num_envs) or in a plain
// array (host build: stride 1)
struct Strided {
    float* p;
    int stride;
    MI_HD float& operator()(int k) const { return p[(size_t)k * stride]; }
};
// self-collision state of an actor created with collision filter 0 (reference humanoid.py:194); nullptr = the actor ignores itself
struct SelfCol {
    Strided lamp;          // [3 * NPG] warm-start impulses of the limb-pair groups (normal, two tangents)
    Strided pairf;         // [3 * NPG] world force on side a of each group's contact, this sub-step (p == nullptr: not wanted)
    int* dropped = nullptr;      // [2] running counts of contacts refused because the env's slots were taken: ground (KMAX), self (KPAIR); or null
    int dstride = 1;
};

// closest points ca, cb of the segments [a0,a1], [b0,b1] (capsule axes; A_PT / B_PT: that side is a sphere, i.e. a point).  Clamped
// closest-point construction, branch-free; the same operation sequence as oracle/physics.c::seg_seg_closest.
template <bool A_PT, bool B_PT>
MI_HD void seg_seg_closest(const float* a0, const float* a1, const float* b0, const float* b1, float* ca, float* cb) {
    if constexpr (A_PT && B_PT) {
        sfor<3>([&](auto K) MI_LAMBDA { ca[K] = a0[K]; cb[K] = b0[K]; });
    } else if constexpr (A_PT) {        // point vs segment: t = clamp(d2.(a0 - b0) / |d2|^2)
        float d2[3], rr[3];
        sfor<3>([&](auto K) MI_LAMBDA { d2[K] = b1[K] - b0[K]; rr[K] = a0[K] - b0[K]; });
        const float E = dot3(d2, d2), F = dot3(d2, rr);
        const float t = (E > 1e-12f) ? fminf(fmaxf(F * MI_RCP(fmaxf(E, 1e-30f)), 0.f), 1.f) : 0.f;
        sfor<3>([&](auto K) MI_LAMBDA { ca[K] = a0[K]; cb[K] = b0[K] + d2[K] * t; });
    } else if constexpr (B_PT) {
        seg_seg_closest<true, false>(b0, b1, a0, a1, cb, ca);
    } else {
        float d1[3], d2[3], rr[3];
        sfor<3>([&](auto K) MI_LAMBDA { d1[K] = a1[K] - a0[K]; d2[K] = b1[K] - b0[K]; rr[K] = a0[K] - b0[K]; });
        const float A = dot3(d1, d1), E = dot3(d2, d2), F = dot3(d2, rr), C = dot3(d1, rr), B = dot3(d1, d2);
        const float den = A * E - B * B;
        const bool okA = A > 1e-12f, okE = E > 1e-12f, okD = (den > 1e-12f) && okA;
        const float rA = MI_RCP(fmaxf(A, 1e-30f)), rE = MI_RCP(fmaxf(E, 1e-30f));
        float sp = okD ? fminf(fmaxf((B * F - C * E) * MI_RCP(fmaxf(den, 1e-30f)), 0.f), 1.f) : 0.f;
        const float t = okE ? (B * sp + F) * rE : 0.f;
        const float tc = fminf(fmaxf(t, 0.f), 1.f);
        const float s2 = fminf(fmaxf((B * tc - C) * rA, 0.f), 1.f);
        sp = ((t != tc || !okE) && okA) ? s2 : sp;      // clamped t (or a point-like b) moves the closest point on a
        sfor<3>([&](auto K) MI_LAMBDA { ca[K] = a0[K] + d1[K] * sp; cb[K] = b0[K] + d2[K] * tc; });
    }
}

// `actor_params` domain randomisation of a model's masses and joint constants: Scaled<M> is M with the per-env factor tensors switched on
// (see Sim::SCALED); only models with M::ACTOR_SCALES != 0 (Ant, Humanoid) are ever wrapped.
template <class M> struct Scaled : M { static constexpr bool MI_SCALED = true; };
template <class M, class = void> struct is_scaled : std::false_type {};
template <class M> struct is_scaled<M, std::void_t<decltype(M::MI_SCALED)>> : std::true_type {};

template <class M>
struct Sim {
    static constexpr int NB = M::NB, ND = M::ND, NV = M::NV, OFF = M::OFF, NSPH = M::NSPH, NSENS = M::NSENS;
    static constexpr int NLIM = []() constexpr { int n = 0; for (int d = 0; d < ND; ++d) n += M::dof_limited[d] ? 1 : 0; return n; }();
    static constexpr int NROWG = (NLIM + 3 * NSPH) > 0 ? (NLIM + 3 * NSPH) : 1;
    static constexpr int NVA = NV > 0 ? NV : 1;
    // slots of the row store: G rows (NROWG x MAXCHAIN), 1/A_ii, velocity targets and (big models) the impulses
    static constexpr bool LAM_IN_ROWS = NROWG > 16;
    static constexpr int ROW_SLOTS_STATIC = NROWG * M::MAXCHAIN + (LAM_IN_ROWS ? 3 : 2) * NROWG;
    // ---- compact contact store.  When the static store (a row set for EVERY contact sphere) does not fit the LDS of a
    // CU even at 64 envs per wave (Humanoid: 126 rows x 15 = 7.5 KB per env), only ACTIVE contacts get a slot: at most
    // KMAX contacts per env, 32 envs per wave (half-filled waves cost nothing while there are more CUs than waves), the
    // slot index of sphere s is data dependent.  Layout [slot][lane] makes any per-lane slot bank-conflict free.
    static constexpr bool COMPACT = (size_t)ROW_SLOTS_STATIC * 64 * sizeof(float) > 152 * 1024;
#ifndef MI_COMPACT_LANES
#define MI_COMPACT_LANES 32
#endif
    // envs per workgroup (= per wave).  Compact store: 32.  Measured (profiles/r2d_lanes_ab.txt, Humanoid@8192): 16 envs per wave
    // -- two 79 KB workgroups per CU, more wave-uniform skipping of inactive spheres / self-contact groups -- is 1.2-1.3x SLOWER
    // (0.425 vs 0.326 ms per step without, 0.630 vs 0.526 ms with self-collision): twice the waves stream the same 250 KB of
    // straight-line code through the instruction caches.
    static constexpr int LANES = COMPACT ? MI_COMPACT_LANES : 64;
#ifndef MI_INLINE_WARM
#define MI_INLINE_WARM 2
#endif
    // static store: the warm-start impulses of a sphere's rows are applied to w right where the rows are built (their g is in
    // registers) instead of re-reading the rows from LDS in a separate pass; the MI_PHASE() fence per sphere keeps the
    // scheduler from batching these updates (which once made it hold every row alive, see DESIGN.md "compiler regime").
    // Measured (A/B in one session): spheres inline -3..-6 % Ant step, -9 % AnymalTerrain step; doing the same for the limit
    // rows (MI_INLINE_WARM=1) costs more spills than it saves, limit rows only (=3) is slower than the separate pass (=0).
    static constexpr bool INLINE_WARM = MI_INLINE_WARM && !COMPACT;
    static constexpr bool INLINE_WARM_LIM = INLINE_WARM && (MI_INLINE_WARM != 2), INLINE_WARM_SPH = INLINE_WARM && (MI_INLINE_WARM != 3);
    // ---- self-collision (compact store only): at most one contact per limb-pair group, KPAIR of them per env, each in a slot of
    // 3 rows over the union of the two limb tips' chains (PCHAIN entries) behind the ground-contact slots
    static constexpr int NPG = COMPACT ? M::NPG : 0, PCH = M::PCHAIN, KPAIR = 3;
    static constexpr int P_CSZ = 3 * PCH + 7;
#ifndef MI_KMAX_SELFCOL
#define MI_KMAX_SELFCOL 12
#endif
    static constexpr int KMAX = NPG > 0 ? MI_KMAX_SELFCOL : 16;   // active ground contacts kept per env (compact store only)
    static constexpr int limoff(int r) {                  // tight packing of the limit rows: offset of row r
        int n = 0;
        for (int k = 0; k < r; ++k) n += M::nanc[OFF + limdof_c(k)] + 1;
        return n;
    }
    static constexpr int limdof_c(int r) {
        int n = 0;
        for (int d = 0; d < ND; ++d) {
            if (M::dof_limited[d]) { if (n == r) return d; ++n; }
        }
        return 0;
    }
    static constexpr int C_LIMG = limoff(NLIM);           // floats of limit-row G
    static constexpr int C_CB = C_LIMG + 3 * NLIM;        // first contact slot (after limit Ainv, vt, lam)
    static constexpr int C_CSZ = 3 * M::MAXCHAIN + 7;     // 3 rows + Ainv x3, vt_n, lam x3
    static constexpr int C_PB = C_CB + KMAX * C_CSZ;      // first self-contact slot
    static constexpr int C_SLOTOF = C_PB + (NPG > 0 ? KPAIR * P_CSZ : 0);  // [NSPH] slot index of each sphere (-1: inactive), one BYTE per sphere
    // [1 + 8 KPAIR] self-contact bookkeeping (group -> slot map, per slot contact point / normal / bodies / friction) handed from
    // the helper wave to the main wave when the sub-step runs on two waves (role 1 -> role 0, below)
    static constexpr int C_X = C_SLOTOF + (NSPH + 3) / 4;
    // [ND][6] joint motion subspaces, parked here by the tree pass for the contact-row build: with S out of the register file
    // after the tree pass the Humanoid sub-step keeps ~126 fewer values live (it overflows the 512 registers of its lane)
    static constexpr int C_S = C_X + (NPG > 0 ? 1 + 8 * KPAIR : 0);
    static constexpr bool S_IN_ROWS = (size_t)(C_S + 6 * ND) * 32 * sizeof(float) <= 160 * 1024;
    static constexpr int ROW_SLOTS_COMPACT = C_S + (S_IN_ROWS ? 6 * ND : 0);
    // last sub-step's contact impulses are parked in the last 3 NSPH floats of the (still empty) GROUND-slot region -- not further
    // up, over the self-contact slots: on two waves those are being filled while the ground spheres still read their parked values
    static constexpr int C_PARK = C_PB;
    static constexpr int C_WARM_OK = (C_PARK - C_CB - 3 * NSPH) / C_CSZ - 1;   // last slot whose write cannot reach the staged warm-start values
    static_assert(!COMPACT || NPG == 0 || C_WARM_OK == 8, "the oracle is told warm_slots = C_WARM_OK + 1 = 9 (tests, bench.py)");
    static constexpr int ROW_SLOTS = COMPACT ? ROW_SLOTS_COMPACT : ROW_SLOTS_STATIC;

    // ---- per-env state carried in registers through a sub-step; the warm-start impulses and the sensor outputs
    //      stay in memory (Strided views) and are touched exactly once per sub-step
    float root[13];               // pos3, quat xyzw, linvel3, angvel3 (world)
    float q[M::NDA], qd[M::NDA];
    // `actor_params` domain randomisation (reference vec_task.py:752-828: rigid_body_properties.mass, dof_properties.damping /
    // stiffness / armature): per-env scale factors of the model's link masses + inertias and joint constants.  Only the models that
    // carry the "actor_scale" tensor (M::ACTOR_SCALES, Ant and Humanoid) multiply by them; for the others the constants stay literals.
    // The factors are compiled into a SEPARATE instantiation, Sim<Scaled<M>>: the launchers pick it when the arena holds the tensors
    // (option actor_tensors) and the plain Sim<M> otherwise, so the benchmark kernels carry none of this code.  (Rounds 1-2 compiled
    // it into the one kernel behind a null-pointer test; the tree pass of the Humanoid then kept the factors live and its register
    // allocation swung by hundreds of spilled registers with every change to this block.)
    static constexpr bool SCALED = is_scaled<M>::value;
    // [NB + 3 ND] factors of this env (strided like every per-env vector), or p == nullptr: one per BODY for its mass and inertia, then
    // one per DOF for the joint's damping, its stiffness and its armature -- the granularity the reference draws them with (it walks
    // every body / dof property struct of the actor, vec_task.py:783-828).  Each factor is loaded where it is used (a body's in the
    // downward half of the tree pass, where its spatial inertia is formed; the joints' at the start of the right-hand-side phase and
    // again for the joint forces at the end) instead of living in registers through the sub-step.  (Until round 3: one factor per
    // actor for each of the four, applied to H and the bias forces after the tree pass.)
    Strided actor_scale{nullptr, 1};
    // Register-allocation fence of the plain kernels of those models.  The sub-step is one huge basic block; LLVM's allocator does far
    // better on the Humanoid when a conditional block that rewrites H and the bias forces splits it between the tree pass and the
    // right-hand side (limb-wave kernel: 60 spilled registers with it, 220 without; DESIGN.md "compiler regime").  Up to round 3 the
    // null-tested application of the mass factor happened to be that block; now that the factors live in Sim<Scaled<M>> the kernels
    // keep a block of the same shape on purpose.  `fence` is View::alloc_fence, a pointer no arena ever sets, so the block is never
    // executed; the compiler cannot know that.
    static constexpr bool FENCED = M::ACTOR_SCALES != 0;
    Strided fence{nullptr, 1};
    template <class F> MI_HD void alloc_fence(F&& rewrite) const {
        if constexpr (FENCED) { if (fence.p != nullptr) rewrite(fence(0)); }
    }
    static constexpr int AS_BODY = 0, AS_DAMP = M::NB, AS_STIFF = M::NB + M::ND, AS_ARM = M::NB + 2 * M::ND, AS_COLS = M::NB + 3 * M::ND;
    // manipulators (M::NOS > 0: the hands, whose `actor_scale` holds the hand_engine.hpp HS_* columns instead): one mass factor per body in a tensor
    // of its own, `hand_body_mass_scale` [NB] (round 5: the reference draws rigid_body_properties.mass per BODY, vec_task.py:783-828)
    Strided body_mass{nullptr, 1};
    template <int b> MI_HD float body_scale() const {
        float x = 1.f;
        if constexpr (SCALED) {
            if constexpr (M::NOS > 0) { if (body_mass.p != nullptr) x = body_mass(b); }
            else { if (actor_scale.p != nullptr) x = actor_scale(AS_BODY + b); }
        }
        return x;
    }
    // `actor_params.<actor>.dof_properties.lower / upper` (Ant.yaml:94-101): per-env shifts of the joint limits, [ND] lower then [ND] upper,
    // or p == nullptr; same models as actor_scale
    Strided limit_shift{nullptr, 1};
    template <int D> MI_HD float limit_lower() const {
        float x = M::dof_lower[D];
        if constexpr (SCALED) { if (limit_shift.p != nullptr) x += limit_shift(D); }
        return x;
    }
    template <int D> MI_HD float limit_upper() const {
        float x = M::dof_upper[D];
        if constexpr (SCALED) { if (limit_shift.p != nullptr) x += limit_shift(ND + D); }
        return x;
    }
#if defined(MI_TIMING)
    unsigned long long* tstamp = nullptr;
#endif

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
    static constexpr int limdof(int r) {  // inverse of limrow: dof of the r-th limit row
        int n = 0;
        for (int d = 0; d < ND; ++d) {
            if (M::dof_limited[d]) { if (n == r) return d; ++n; }
        }
        return 0;
    }
    static constexpr bool cap_is_point(int c) { return M::cap_s0[c] == M::cap_s1[c]; }
    static constexpr float cap_bound(int c) {   // radius of the capsule's bounding sphere about the middle of its axis
        const float d[3] = {M::cap_p0[c][0] - M::cap_p1[c][0], M::cap_p0[c][1] - M::cap_p1[c][1], M::cap_p0[c][2] - M::cap_p1[c][2]};
        const float l2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        float r = l2 > 0.f ? 0.5f : 0.f;       // constexpr square root of l2 / 4 by Newton iteration
        if (l2 > 0.f) { r = l2; for (int it = 0; it < 40; ++it) r = 0.5f * (r + l2 / r); r *= 0.5f; }
        return r * 1.0001f + M::cap_rad[c];
    }
    static constexpr bool group_has_body(int g, int b) {  // can body b take part in a contact of group g?
        for (int k = M::pg_first[g]; k < M::pg_first[g] + M::pg_count[g]; ++k)
            if (M::cap_body[M::gp_a[k]] == b || M::cap_body[M::gp_b[k]] == b) return true;
        return false;
    }
    static constexpr int sensor_of(int b) {  // index of the force sensor on body b, -1 if none
        for (int k = 0; k < NSENS; ++k)
            if (M::sens_body[k] == b) return k;
        return -1;
    }

    // whole simulate() on plain arrays (host build of the tests; the kernels call substep() directly).
    // state layout = oracle/physics.c: lamc[3*NSPH], laml[ND], sensor[6*NSENS], dof_force[ND]
    // lamp [3 * NPG] (self-contact warm start; nullptr: the actor ignores itself), pairf [stride 6 per group]: world force per group
    MI_HD void step(const SimParams& P, const float* tau, float* lamc, float* laml, float* sensor, float* dof_force,
                    float* lamp = nullptr, float* pairf = nullptr) {
        const float h = P.dt / (float)P.substeps;
        float rows[ROW_SLOTS];
        for (int ss = 0; ss < P.substeps; ++ss)
            substep_noinline(P, tau, h, rows, lamc, laml, sensor, dof_force, lamp, pairf);
    }
    MI_HD_NOINLINE void substep_noinline(const SimParams& P, const float* tau, const float h, float* rows, float* lamc,
                                         float* laml, float* sensor, float* dof_force, float* lamp, float* pairf) {
        const SelfCol sc{Strided{lamp, 1}, Strided{pairf, 1}};
        substep(P, tau, h, RowStore<1>{rows}, Strided{lamc, 1}, Strided{laml, 1}, Strided{sensor, 1}, Strided{dof_force, 1},
                PlaneGround{}, -1.f, Strided{nullptr, 1}, nullptr, false, lamp ? &sc : nullptr);
    }
    // same on a height field with per-env friction and per-body net contact forces netf[3*NB]
    MI_HD void step_terrain(const SimParams& P, const float* tau, float* lamc, float* laml, float* sensor, float* dof_force,
                            const HeightfieldGround& gnd, float mu_env, float* netf) {
        const float h = P.dt / (float)P.substeps;
        float rows[ROW_SLOTS];
        for (int ss = 0; ss < P.substeps; ++ss)
            substep(P, tau, h, RowStore<1>{rows}, Strided{lamc, 1}, Strided{laml, 1}, Strided{sensor, 1}, Strided{dof_force, 1},
                    gnd, mu_env, Strided{netf, 1});
    }

    // row-store slot in which the sub-step expects last sub-step's impulse of limit row `d` / of contact row k (= 3 * sphere +
    // direction) when it starts -- staged there either by the sub-step itself or, on the GPU, by LDS-direct loads issued by the
    // kernel before it calls the sub-step (`prestaged`)
    static constexpr bool STAGES_LAM = COMPACT || LAM_IN_ROWS;
    static constexpr int stage_slot_lim(int d) { return COMPACT ? C_LIMG + 2 * NLIM + limrow(d) : NROWG * M::MAXCHAIN + 2 * NROWG + limrow(d); }
    static constexpr int stage_slot_con(int k) { return COMPACT ? C_PARK - 3 * (k / 3 + 1) + (k % 3) : NROWG * M::MAXCHAIN + 2 * NROWG + NLIM + k; }
    // working set shared by the phases of one sub-step
    struct Ctx {
        float S[M::NDA][6];           // joint motion subspaces, world axes about O = root origin
        float bias[NVA];              // C(q, qd) + gravity terms (RNEA with zero acceleration)
        float L[M::NM];               // branch-sparse H, later its L^T L factor
        float xcs[M::NSPHA][3];       // centre of every contact sphere, relative to O
        float Rs[M::NSENSA][9], rs[M::NSENSA][3];  // pose of the force-sensor bodies
        // manipulation models: pose (R row-major 9, r 3) of every body that carries object-contact spheres is written to
        // pose_out[(12 * os_slot(b) + i) * pose_stride] during the tree pass (the hand engine points this into LDS)
        float* pose_out = nullptr;
        int pose_stride = 1;
        // compact-store models: S[d][k] is also written to s_out[(6 * d + k) * s_stride] (row store, C_S) when non-null
        float* s_out = nullptr;
        int s_stride = 1;
    };
    // object-contact spheres are grouped by body (generated os_body is non-decreasing)
    static constexpr int os_first(int b) { for (int s = 0; s < M::NOS; ++s) if (M::os_body[s] == b) return s; return 0; }
    static constexpr int os_count(int b) { int n = 0; for (int s = 0; s < M::NOS; ++s) n += (M::os_body[s] == b) ? 1 : 0; return n; }
    static constexpr int os_slot(int b) {  // index among the bodies that carry spheres, -1 if none
        if (os_count(b) == 0) return -1;
        int k = 0;
        for (int bb = 0; bb < b; ++bb) k += os_count(bb) > 0 ? 1 : 0;
        return k;
    }
    static constexpr int NOSB = []() constexpr { int k = 0; for (int b = 0; b < NB; ++b) k += os_count(b) > 0 ? 1 : 0; return k; }();

    // ---------------------------------------------------------------- one body of the depth-first tree pass
    // Going down: pose, joint axes, velocity / bias acceleration.  Coming back up: subtree force and composite
    // inertia, from which the bias force and the H entries of this body's dofs follow at once -- so per-body
    // quantities only live while their subtree is being processed (live state ~ tree depth, not body count).
    // per-body quantities handed from the downward half of the tree pass to the upward half
    struct BodyTmp {
        float Rb[9], rb[3];     // pose (rb relative to O = root origin)
        SpI I;                  // own spatial inertia about O; after the children have been added: composite inertia of the subtree
        float Vc[6], Ac[6];     // spatial velocity / bias acceleration
        float F[6];             // own force; after the children have been added: subtree force
    };
    // going down: pose, joint axes S, contact-sphere centres, sensor frame, own inertia, velocity / bias acceleration, own force
    template <int b>
    MI_HD void body_down(const SimParams& P, Ctx& c, const float* Rp, const float* rp, const float* Vp, const float* Ap, BodyTmp& t) {
        float (&Rb)[9] = t.Rb;
        float (&rb)[3] = t.rb;
        if constexpr (b == 0) {
            quat2mat(root + 3, Rb);
            rb[0] = rb[1] = rb[2] = 0.f;
        } else {
            if constexpr (brot_is_identity(b)) {
                sfor<9>([&](auto K) MI_LAMBDA { Rb[K] = Rp[K]; });
            } else {
                matmul3(Rp, M::brot[b], Rb);
            }
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
            float* Sd = c.S[d];
            if constexpr (M::dof_type[d] == 0) {
                float s, cs;
                MI_SINCOS(q[d], &s, &cs);
                const float t = 1.f - cs;
                // rotation about the (constant) local axis: Rb <- Rb * Q_local
                const float Q[9] = {cs + ax * ax * t, ax * ay * t - az * s, ax * az * t + ay * s,
                                    ay * ax * t + az * s, cs + ay * ay * t, ay * az * t - ax * s,
                                    az * ax * t - ay * s, az * ay * t + ax * s, cs + az * az * t};
                matmul3(Rb, Q, Rb);
                float tb[3];
                matvec3(Rb, anl, tb);
                rb[0] = pt[0] - tb[0]; rb[1] = pt[1] - tb[1]; rb[2] = pt[2] - tb[2];
                Sd[0] = a[0]; Sd[1] = a[1]; Sd[2] = a[2];
                cross3(pt, a, Sd + 3);
            } else {
                rb[0] += a[0] * q[d]; rb[1] += a[1] * q[d]; rb[2] += a[2] * q[d];
                Sd[0] = Sd[1] = Sd[2] = 0.f;
                Sd[3] = a[0]; Sd[4] = a[1]; Sd[5] = a[2];
            }
            if constexpr (COMPACT && S_IN_ROWS && M::NOS == 0) {
                if (c.s_out) sfor<6>([&](auto I_) MI_LAMBDA { c.s_out[(6 * d + I_) * c.s_stride] = Sd[I_]; });
            }
        });
        // contact spheres and force sensor riding on this body
        sfor<NSPH>([&](auto S_) MI_LAMBDA {
            constexpr int s = S_;
            if constexpr (M::sph_body[s] == b) {
                float t[3];
                matvec3(Rb, M::sph_pos[s], t);
                c.xcs[s][0] = rb[0] + t[0]; c.xcs[s][1] = rb[1] + t[1]; c.xcs[s][2] = rb[2] + t[2];
            }
        });
        if constexpr (os_slot(b) >= 0) {
            constexpr int k = os_slot(b);
            sfor<9>([&](auto I_) MI_LAMBDA { c.pose_out[(12 * k + I_) * c.pose_stride] = Rb[I_]; });
            sfor<3>([&](auto I_) MI_LAMBDA { c.pose_out[(12 * k + 9 + I_) * c.pose_stride] = rb[I_]; });
        }
        if constexpr (sensor_of(b) >= 0) {
            constexpr int k = sensor_of(b);
            sfor<9>([&](auto K) MI_LAMBDA { c.Rs[k][K] = Rb[K]; });
            c.rs[k][0] = rb[0]; c.rs[k][1] = rb[1]; c.rs[k][2] = rb[2];
        }
        // world spatial inertia about O
        SpI& I = t.I;
        {
            float t[3], cm[3];
            matvec3(Rb, M::com[b], t);
            cm[0] = rb[0] + t[0]; cm[1] = rb[1] + t[1]; cm[2] = rb[2] + t[2];
            constexpr float ixx = M::inertia[b][0], iyy = M::inertia[b][1], izz = M::inertia[b][2],
                            ixy = M::inertia[b][3], ixz = M::inertia[b][4], iyz = M::inertia[b][5];
            const float Il[9] = {ixx, ixy, ixz, ixy, iyy, iyz, ixz, iyz, izz};
            float T[9];
            matmul3(Rb, Il, T);
            // Iw = T * Rb^T (symmetric)
            const float Iw0 = T[0] * Rb[0] + T[1] * Rb[1] + T[2] * Rb[2];
            const float Iw4 = T[3] * Rb[3] + T[4] * Rb[4] + T[5] * Rb[5];
            const float Iw8 = T[6] * Rb[6] + T[7] * Rb[7] + T[8] * Rb[8];
            const float Iw1 = T[0] * Rb[3] + T[1] * Rb[4] + T[2] * Rb[5];
            const float Iw2 = T[0] * Rb[6] + T[1] * Rb[7] + T[2] * Rb[8];
            const float Iw5 = T[3] * Rb[6] + T[4] * Rb[7] + T[5] * Rb[8];
            const float cc = dot3(cm, cm);
            if constexpr (SCALED) {
                // `actor_params` rigid_body_properties.mass: this body's mass (and with it its inertia) times its factor.  Only in the
                // Sim<Scaled<M>> instantiation: in the Humanoid's kernels this multiply costs ~300 more spilled registers (limb waves
                // 154 -> 467, one wave 327 -> 727), which the runs without randomised masses must not pay.
                const float sb = this->template body_scale<b>();
                const float mm = M::mass[b] * sb;
                I.m = mm; I.h[0] = mm * cm[0]; I.h[1] = mm * cm[1]; I.h[2] = mm * cm[2];
                I.I[0] = sb * Iw0 + mm * (cc - cm[0] * cm[0]); I.I[1] = sb * Iw4 + mm * (cc - cm[1] * cm[1]);
                I.I[2] = sb * Iw8 + mm * (cc - cm[2] * cm[2]);
                I.I[3] = sb * Iw1 - mm * cm[0] * cm[1]; I.I[4] = sb * Iw2 - mm * cm[0] * cm[2]; I.I[5] = sb * Iw5 - mm * cm[1] * cm[2];
            } else {
                constexpr float mm = M::mass[b];
                I.m = mm; I.h[0] = mm * cm[0]; I.h[1] = mm * cm[1]; I.h[2] = mm * cm[2];
                I.I[0] = Iw0 + mm * (cc - cm[0] * cm[0]); I.I[1] = Iw4 + mm * (cc - cm[1] * cm[1]);
                I.I[2] = Iw8 + mm * (cc - cm[2] * cm[2]);
                I.I[3] = Iw1 - mm * cm[0] * cm[1]; I.I[4] = Iw2 - mm * cm[0] * cm[2]; I.I[5] = Iw5 - mm * cm[1] * cm[2];
            }
        }
        // velocity / bias acceleration recursion
        float (&Vc)[6] = t.Vc;
        float (&Ac)[6] = t.Ac;
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
            sfor<6>([&](auto K) MI_LAMBDA { Vc[K] = Vp[K]; Ac[K] = Ap[K]; });
        }
        sfor<M::body_ndof[b]>([&](auto K) MI_LAMBDA {
            constexpr int d = M::body_dof0[b] + K;
            float Sd[6];
            crm(Vc, c.S[d], Sd);
            sfor<6>([&](auto C) MI_LAMBDA { Ac[C] += Sd[C] * qd[d]; Vc[C] += c.S[d][C] * qd[d]; });
        });
        float (&F)[6] = t.F;
        {
            float IA[6], IV[6], X[6];
            spi_mul(I, Ac, IA); spi_mul(I, Vc, IV); crf(Vc, IV, X);
            sfor<6>([&](auto K) MI_LAMBDA { F[K] = IA[K] + X[K]; });
        }
    }
    // coming back up (t.I / t.F hold the subtree's composite inertia / force): bias force and H entries of this body's dofs
    template <int b>
    MI_HD void body_up(Ctx& c, BodyTmp& t) {
        SpI& I = t.I;
        float (&F)[6] = t.F;
        sfor<M::body_ndof[b]>([&](auto K) MI_LAMBDA {
            constexpr int d = M::body_dof0[b] + K, gi = OFF + d;
            c.bias[gi] = dot6(c.S[d], F);
            float Fd[6];
            spi_mul(I, c.S[d], Fd);
            c.L[M::midx[gi][gi]] = dot6(c.S[d], Fd);
            sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA {
                constexpr int gj = M::anc[gi][A_];
                float v;
                if constexpr (gj >= OFF) v = dot6(c.S[gj - OFF], Fd);
                else if constexpr (gj < 3) v = Fd[3 + gj];
                else v = Fd[gj - 3];
                c.L[M::midx[gi][gj]] = v;
            });
        });
        if constexpr (b == 0 && !M::FIXED) {
            c.bias[0] = F[3]; c.bias[1] = F[4]; c.bias[2] = F[5];
            c.bias[3] = F[0]; c.bias[4] = F[1]; c.bias[5] = F[2];
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
                        c.L[M::midx[i][j]] = v;
                    }
                });
            });
        }
    }
    template <int b>
    MI_HD void body_pass(const SimParams& P, Ctx& c, const float* Rp, const float* rp, const float* Vp, const float* Ap,
                         SpI& Iout, float* Fout) {
        BodyTmp t;
        body_down<b>(P, c, Rp, rp, Vp, Ap, t);
        // children (bodies are numbered depth-first, so every child index is > b)
        sfor<NB>([&](auto C_) MI_LAMBDA {
            constexpr int ch = C_;
            if constexpr (ch > b) if constexpr (M::parent[ch] == b) {
                SpI Ic;
                float Fc[6];
                body_pass<ch>(P, c, t.Rb, t.rb, t.Vc, t.Ac, Ic, Fc);
                sfor<6>([&](auto K) MI_LAMBDA { t.F[K] += Fc[K]; });
                t.I.m += Ic.m;
                sfor<3>([&](auto K) MI_LAMBDA { t.I.h[K] += Ic.h[K]; });
                sfor<6>([&](auto K) MI_LAMBDA { t.I.I[K] += Ic.I[K]; });
            }
        });
        body_up<b>(c, t);
        if constexpr (b > 0) {
            Iout = t.I;
            sfor<6>([&](auto K) MI_LAMBDA { Fout[K] = t.F[K]; });
        }
    }

    // ---------------------------------------------------------------- world state of ONE body at the current q / qd
    // o[13] = position, quaternion xyzw, linear velocity of the body frame's origin, angular velocity (world frame): one row of
    // gym.acquire_rigid_body_state_tensor (reference shadow_hand.py:150-175,456-457).  Walks the chain root -> body; reads q / qd of the
    // dofs on that chain only.
    static constexpr bool on_chain(int a, int b) { while (b >= 0) { if (b == a) return true; b = M::parent[b]; } return false; }
    MI_HD static void rot2quat(const float* R, float* qo) {      // Shepperd, w >= 0 branch first
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
    template <int TIP>
    MI_HD void body_state(float* o) {
        float Rb[9], p[3], vl[3], om[3];
        quat2mat(root + 3, Rb);
        sfor<3>([&](auto I_) MI_LAMBDA { p[I_] = root[I_]; vl[I_] = M::FIXED ? 0.f : root[7 + I_]; om[I_] = M::FIXED ? 0.f : root[10 + I_]; });
        sfor<NB>([&](auto B_) MI_LAMBDA {
            constexpr int b = B_;
            if constexpr (b > 0 && on_chain(b, TIP)) {
                float t[3], cx[3];
                matvec3(Rb, M::bpos[b], t);
                cross3(om, t, cx);
                sfor<3>([&](auto I_) MI_LAMBDA { p[I_] += t[I_]; vl[I_] += cx[I_]; });
                if constexpr (!brot_is_identity(b)) matmul3(Rb, M::brot[b], Rb);
                sfor<M::body_ndof[b]>([&](auto J_) MI_LAMBDA {
                    constexpr int d = M::body_dof0[b] + J_;
                    constexpr float ax = M::dof_axis[d][0], ay = M::dof_axis[d][1], az = M::dof_axis[d][2];
                    const float al[3] = {ax, ay, az}, anl[3] = {M::dof_anchor[d][0], M::dof_anchor[d][1], M::dof_anchor[d][2]};
                    float a[3];
                    matvec3(Rb, al, a);
                    if constexpr (M::dof_type[d] == 0) {
                        float ta[3], s_, c_;
                        matvec3(Rb, anl, ta);
                        sincosf(q[d], &s_, &c_);
                        const float tt = 1.f - c_;
                        const float Q[9] = {c_ + ax * ax * tt, ax * ay * tt - az * s_, ax * az * tt + ay * s_,
                                            ay * ax * tt + az * s_, c_ + ay * ay * tt, ay * az * tt - ax * s_,
                                            az * ax * tt - ay * s_, az * ay * tt + ax * s_, c_ + az * az * tt};
                        matmul3(Rb, Q, Rb);
                        float tb[3];
                        matvec3(Rb, anl, tb);
                        // the body origin moves on a circle about the anchor; its velocity picks up the joint rate about the anchor
                        const float dr[3] = {ta[0] - tb[0], ta[1] - tb[1], ta[2] - tb[2]};
                        float c1[3], c2[3];
                        cross3(om, dr, c1);
                        const float wj[3] = {a[0] * qd[d], a[1] * qd[d], a[2] * qd[d]};
                        const float mtb[3] = {-tb[0], -tb[1], -tb[2]};
                        cross3(wj, mtb, c2);
                        sfor<3>([&](auto I_) MI_LAMBDA { p[I_] += dr[I_]; vl[I_] += c1[I_] + c2[I_]; om[I_] += wj[I_]; });
                    } else {
                        const float dr[3] = {a[0] * q[d], a[1] * q[d], a[2] * q[d]};
                        float c1[3];
                        cross3(om, dr, c1);
                        sfor<3>([&](auto I_) MI_LAMBDA { p[I_] += dr[I_]; vl[I_] += c1[I_] + a[I_] * qd[d]; });
                    }
                });
            }
        });
        sfor<3>([&](auto I_) MI_LAMBDA { o[I_] = p[I_]; o[7 + I_] = vl[I_]; o[10 + I_] = om[I_]; });
        rot2quat(Rb, o + 3);
    }

    // ---------------------------------------------------------------- Jacobian of ONE body, joint-space inertia matrix
    // gym.acquire_jacobian_tensor / acquire_mass_matrix_tensor (reference franka_cube_stack.py:388-392, the operational-space controller
    // of :595-612 reads them every step).  Generalised velocity: [root linear velocity (of the root frame's origin), root angular velocity,
    // both world frame -- floating bases only] ++ qd, NV entries.
    // J[6][NV], row-major: rows 0-2 the linear velocity of body b's frame origin, rows 3-5 its angular velocity (world frame) per unit of
    // each generalised velocity: column c is body_state's velocity block at the c-th unit velocity, so J qd == the velocities
    // gym.refresh_rigid_body_state_tensor reports, by construction.
    template <int b>
    MI_HD void body_jacobian(float* J) const {
        Sim s = *this;
        sfor<6>([&](auto K) MI_LAMBDA { s.root[7 + K] = 0.f; });
        sfor<ND>([&](auto D) MI_LAMBDA { s.qd[D] = 0.f; });
        sfor<NV>([&](auto C_) MI_LAMBDA {
            constexpr int c = C_;
            constexpr bool moves = (c < OFF) || on_chain(M::dof_body[c < OFF ? 0 : c - OFF], b);
            if constexpr (!moves) {
                sfor<6>([&](auto K) MI_LAMBDA { J[K * NV + c] = 0.f; });
            } else {
                if constexpr (c < OFF) s.root[7 + c] = 1.f; else s.qd[c - OFF] = 1.f;
                float o[13];
                s.template body_state<b>(o);
                sfor<3>([&](auto K) MI_LAMBDA { J[K * NV + c] = o[7 + K]; J[(3 + K) * NV + c] = o[10 + K]; });
                if constexpr (c < OFF) s.root[7 + c] = 0.f; else s.qd[c - OFF] = 0.f;
            }
        });
    }
    // H[NV][NV], row-major, symmetric: the composite-rigid-body inertia the sub-step factors (tree pass), joint armatures on the diagonal
    // (asset_options.use_physx_armature / dof_props['armature'], shadow_hand.py:243, allegro_hand.py:263)
    MI_HD void mass_matrix(const SimParams& P, float* H) {
        Ctx c;
        SpI Iroot;
        float Froot[6];
        float pose[12 * (NOSB > 0 ? NOSB : 1)];      // manipulators: the tree pass hands the poses of the sphere-carrying bodies on (unused here)
        c.pose_out = pose;
        c.pose_stride = 1;
        this->template body_pass<0>(P, c, nullptr, nullptr, nullptr, nullptr, Iroot, Froot);
        sfor<NV * NV>([&](auto K) MI_LAMBDA { H[K] = 0.f; });
        sfor<NV>([&](auto I_) MI_LAMBDA {
            constexpr int i = I_;
            H[i * NV + i] = c.L[M::midx[i][i]] + (i >= OFF ? M::dof_armature[i >= OFF ? i - OFF : 0] : 0.f);
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA {
                constexpr int j = M::anc[i][A_];
                H[i * NV + j] = c.L[M::midx[i][j]];
                H[j * NV + i] = c.L[M::midx[i][j]];
            });
        });
    }

    // ---------------------------------------------------------------- one physics sub-step of length h
    // gnd: ground policy; mu_env >= 0 replaces the per-sphere model friction (per-env friction buckets of
    // anymal_terrain.py:236-239,279-281); netf: per-body net contact force [3*NB] (world, this sub-step), written only on
    // height fields (gym.acquire_net_contact_force_tensor, anymal_terrain.py:119)
    // role / nroles / bar: the sub-step of a self-colliding robot on TWO or THREE waves of one workgroup that share the row store.
    // role -1: one wave does everything (bar unused).  role 1, the self-collision helper: tree pass and factorisation like the main
    // wave, then ONLY the self-collision phase (broad + narrow phase, rows into the self-contact slots), publishes its bookkeeping
    // at C_X, meets the others at bar() and is done.  role 2 (nroles == 3), the limit-row helper: tree pass, factorisation, the
    // joint-limit rows (it stages their warm-start impulses itself), bar(), done.  role 0, the main wave: everything the helpers do
    // not do; it meets them at bar() after its ground rows and picks the bookkeeping up.  All waves compute bit-identical L, S,
    // sphere centres (same code, same inputs), and write disjoint parts of the store.
    struct NoBarrier { MI_HD void operator()() const {} };
    template <int RS, class GND, class BAR = NoBarrier>
    MI_HD void substep(const SimParams& P, const float* tau, const float h, const RowStore<RS> rows, const Strided lamc,
                       const Strided laml, const Strided sensor, const Strided dof_force, const GND& gnd, const float mu_env,
                       const Strided netf, const Drive* drv = nullptr, const bool prestaged = false, const SelfCol* scol = nullptr,
                       const int role = -1, const BAR& bar = BAR{}, const int nroles = 2) {
        const bool main_wave = role <= 0;                                   // right-hand side, w, ground rows, everything after the barrier
        const bool do_limits = role < 0 || role == (nroles == 3 ? 2 : 0);   // who builds the joint-limit rows
        const bool do_pairs = role < 0 || role == 1;
        auto slot8 = [&](const RowStore<RS>& r, int s) MI_LAMBDA -> signed char& { return reinterpret_cast<signed char*>(r.ptr(C_SLOTOF + (s >> 2)))[s & 3]; };
        // static store: row r at r*MAXCHAIN; compact store: only the limit rows (r < NLIM) live at fixed, tightly packed places
        auto G = [&](int row, int c) MI_LAMBDA -> float& { return rows(COMPACT ? limoff(row) + c : row * M::MAXCHAIN + c); };
        auto Ainv = [&](int row) MI_LAMBDA -> float& { return rows(COMPACT ? C_LIMG + row : NROWG * M::MAXCHAIN + row); };
        auto vt = [&](int row) MI_LAMBDA -> float& { return rows(COMPACT ? C_LIMG + NLIM + row : NROWG * M::MAXCHAIN + NROWG + row); };
        const float invh = MI_RCP(h);
        Ctx c;
        float (&S)[M::NDA][6] = c.S;
        float (&L)[M::NM] = c.L;
        MI_STAMP(0);
        // ------------------------------------------------------------ stage last sub-step's impulses (warm start) in the row store
        // all loads are issued back to back here, far ahead of their use in the row build, instead of one exposed
        // HBM round trip per row (a wave has nobody to switch to while it waits)
        static_assert(LAM_IN_ROWS || NROWG <= 16, "small models keep lam in registers");
        // compact store: contact impulses are parked at the END of the (still empty) contact-slot region, sphere 0 last: slots
        // fill from the front and sphere s is read before any slot > s can be written
        if constexpr (STAGES_LAM) {
            if (!prestaged) {     // (a kernel that prestages does it for exactly the rows this wave owns)
                if (do_limits) sfor<ND>([&](auto D) MI_LAMBDA {
                    constexpr int d = D;
                    if constexpr (M::dof_limited[d]) rows(stage_slot_lim(d)) = laml(d);
                });
                if (main_wave) sfor<3 * NSPH>([&](auto K) MI_LAMBDA { rows(stage_slot_con(K)) = lamc(K); });
            }
        }
        MI_STAMP(1);
        // ------------------------------------------------------------ kinematics + dynamics, one depth-first tree pass
        if constexpr (COMPACT && S_IN_ROWS && M::NOS == 0) {
            c.s_out = rows.ptr(C_S);
            c.s_stride = RowStore<RS>::stride;
        }
        {
            SpI Iroot;
            float Froot[6];
            body_pass<0>(P, c, nullptr, nullptr, nullptr, nullptr, Iroot, Froot);
        }
#if defined(MI_STOP_AFTER) && MI_STOP_AFTER == 1
        { float acc = 0.f; sfor<M::NM>([&](auto K) MI_LAMBDA { acc += L[K]; }); sfor<NV>([&](auto K) MI_LAMBDA { acc += c.bias[K]; });
          sfor<ND>([&](auto K) MI_LAMBDA { sfor<6>([&](auto J) MI_LAMBDA { acc += S[K][J]; }); });
          sfor<NSPH>([&](auto K) MI_LAMBDA { sfor<3>([&](auto J) MI_LAMBDA { acc += c.xcs[K][J]; }); });
          sfor<NSENS>([&](auto K) MI_LAMBDA { sfor<9>([&](auto J) MI_LAMBDA { acc += c.Rs[K][J]; }); sfor<3>([&](auto J) MI_LAMBDA { acc += c.rs[K][J]; }); });
          root[0] = acc; return; }
#endif
        MI_PHASE();
        MI_STAMP(2);
        // ------------------------------------------------------------ rhs, implicit spring/damper on the diagonal
        float Ldi[NVA];  // 1 / L_ii
        float y[NVA];
        float sc_damp[M::NDA], sc_stiff[M::NDA], sc_arm[M::NDA];      // per-dof `actor_params` factors (folded away for unscaled models)
        sfor<ND>([&](auto D) MI_LAMBDA { sc_damp[D] = 1.f; sc_stiff[D] = 1.f; sc_arm[D] = 1.f; });
        if constexpr (SCALED) {
            if (actor_scale.p != nullptr) {
                sfor<ND>([&](auto D) MI_LAMBDA { sc_damp[D] = actor_scale(AS_DAMP + D); sc_stiff[D] = actor_scale(AS_STIFF + D); sc_arm[D] = actor_scale(AS_ARM + D); });
            }
        }
        alloc_fence([&](float f) MI_LAMBDA {      // never taken (see alloc_fence)
            sfor<M::NM>([&](auto E_) MI_LAMBDA { L[E_] *= f; });
            sfor<NV>([&](auto I) MI_LAMBDA { c.bias[I] *= f; });
        });
        sfor<OFF>([&](auto I) MI_LAMBDA { y[I] = -c.bias[I]; });
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            const float K = M::dof_stiffness[d] * sc_stiff[d], Dm = M::dof_damping[d] * sc_damp[d];
            L[M::midx[gi][gi]] += M::dof_armature[d] * sc_arm[d] + h * Dm + h * h * K;
            y[gi] = tau[d] - c.bias[gi] - K * (q[d] - M::dof_springref[d]) - (Dm + h * K) * qd[d];
            if (drv) {   // position drive: the same implicit linearisation as the passive spring / damper
                const float kpd = drv->gain_p(d), kdd = drv->gain_d(d);
                L[M::midx[gi][gi]] += h * kdd + h * h * kpd;
                y[gi] += kpd * (drv->target[d] - q[d]) - (kdd + h * kpd) * qd[d];
            }
        });
        if (drv && drv->fsens) {   // generalised force J^T f of the externally forced bodies (chain-sparse, like a contact row)
            sfor<NSENS>([&](auto K_) MI_LAMBDA {
                constexpr int k = K_, b = M::sens_body[k];
                const float fl[3] = {drv->fsens[3 * k], drv->fsens[3 * k + 1], drv->fsens[3 * k + 2]};
                float fw[3], cm[3], W[6];
                matvec3(c.Rs[k], fl, fw);
                matvec3(c.Rs[k], M::com[b], cm);
                sfor<3>([&](auto I_) MI_LAMBDA { cm[I_] += c.rs[k][I_]; });
                cross3(cm, fw, W);
                W[3] = fw[0]; W[4] = fw[1]; W[5] = fw[2];
                sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA {
                    constexpr int gi = M::chain[b][C];
                    if constexpr (gi >= OFF) y[gi] += dot6(S[gi - OFF], W);
                    else if constexpr (gi < 3) y[gi] += W[3 + gi];
                    else y[gi] += W[gi - 3];
                });
            });
        }
        MI_PHASE();
        // ------------------------------------------------------------ H = L^T L in place (no fill-in on a tree)
        sfor_rev<NV>([&](auto K_) MI_LAMBDA {
            constexpr int k = K_;
            const float dk2 = fmaxf(L[M::midx[k][k]], 1e-30f);
            const float inv = MI_RSQ(dk2);
            const float dkk = dk2 * inv;
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
        MI_PHASE();
        // ------------------------------------------------------------ whitened velocity  w = L qd + h L^-T rhs
        float w[NVA];
        if (main_wave) {
            float v[NVA];
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
#if defined(MI_STOP_AFTER) && MI_STOP_AFTER == 2
        { float acc = 0.f; sfor<M::NM>([&](auto K) MI_LAMBDA { acc += L[K]; }); sfor<NV>([&](auto K) MI_LAMBDA { acc += w[K] + Ldi[K]; });
          sfor<ND>([&](auto K) MI_LAMBDA { sfor<6>([&](auto J) MI_LAMBDA { acc += S[K][J]; }); });
          sfor<NSPH>([&](auto K) MI_LAMBDA { sfor<3>([&](auto J) MI_LAMBDA { acc += c.xcs[K][J]; }); });
          sfor<NSENS>([&](auto K) MI_LAMBDA { sfor<9>([&](auto J) MI_LAMBDA { acc += c.Rs[K][J]; }); sfor<3>([&](auto J) MI_LAMBDA { acc += c.rs[K][J]; }); });
          root[0] = acc; return; }
#endif
        MI_PHASE();
        MI_STAMP(3);
        // ------------------------------------------------------------ constraint rows in whitened space
#if defined(__HIP_DEVICE_COMPILE__)
        // LDS-direct staging loads were issued before the tree pass; they are long done, but the compiler cannot know
        if constexpr (STAGES_LAM) { if (prestaged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
        float lam_reg[LAM_IN_ROWS ? 1 : NROWG];
        auto lam = [&](int row) MI_LAMBDA -> float& {
            if constexpr (COMPACT) return rows(C_LIMG + 2 * NLIM + row);   // limit rows only
            else if constexpr (LAM_IN_ROWS) return rows(NROWG * M::MAXCHAIN + 2 * NROWG + row);
            else return lam_reg[row];
        };
        // the same for the three rows of one contact at once: every L entry and 1/L_ii is fetched once for the three right-hand sides
        auto chain_solve3 = [&](auto B, float (*g)[M::MAXCHAIN]) MI_LAMBDA {
            constexpr int b = decltype(B)::value;
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
        if (do_limits) sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D, gi = OFF + d;
            if constexpr (M::dof_limited[d]) {
                constexpr int row = limrow(d);
                MI_PHASE();
                const float dl = q[d] - this->template limit_lower<d>(), du = this->template limit_upper<d>() - q[d];
                const bool lower = dl < du;
                const float C = lower ? dl : du, s = lower ? 1.f : -1.f;
                float lw;
                if constexpr (LAM_IN_ROWS) lw = lam(row); else lw = laml(d);
                const float l0 = ((lw * s < 0.f) ? 0.f : fabsf(lw)) * P.warm;
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
                sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { a += g[K] * g[K]; G(row, K) = g[K]; });
                Ainv(row) = MI_RCP(a);
                vt(row) = (C >= 0.f) ? -C * invh : fminf(-C * P.erp * invh, P.max_depen_vel);
                lam(row) = l0;
                if constexpr (INLINE_WARM_LIM) {
                    w[gi] += g[0] * l0;
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += g[1 + A_] * l0; });
                    MI_PHASE();
                }
            }
        });
        MI_PHASE();
        MI_STAMP(4);
        // bit s set <=> some env of this wave has sphere s within contact_offset (wave-uniform, lives in an SGPR): rows of
        // a sphere no env touches are neither built nor swept -- their contribution would be exactly zero anyway
        unsigned long long sph_active = 0ull;
        static_assert(COMPACT || NSPH <= 64, "sph_active is a 64-bit mask");
        // self-collision bookkeeping: 2 bits per group = the slot of its contact (3: none); per slot the contact point (rel. O), the
        // normal (side b -> side a), the two bodies (a | b << 8, as int bits) and the friction coefficient
        static_assert(NPG <= 16, "pmap holds 2 bits per group");
        unsigned pmap = 0xFFFFFFFFu;
        float pinf[NPG > 0 ? KPAIR : 1][8];
        if constexpr (NPG > 0) sfor<KPAIR>([&](auto J_) MI_LAMBDA { sfor<8>([&](auto I_) MI_LAMBDA { pinf[J_][I_] = 0.f; }); });
        const bool selfcol = (NPG > 0) && (scol != nullptr);
        if constexpr (!COMPACT) {
        // ground contacts: 3 rows per sphere (normal, two tangents; +z, x, y on the plane)
        sfor<NSPH>([&](auto S_) MI_LAMBDA {
            constexpr int s = S_, b = M::sph_body[s], row0 = NLIM + 3 * s;
            MI_PHASE();
            const float* cs = c.xcs[s];
            float xc[3], dist;
            float fr[3][3];  // contact frame n, t1, t2 (height field only)
            if constexpr (GND::HEIGHTFIELD) {
                gnd.contact(root[0] + cs[0], root[1] + cs[1], root[2] + cs[2], M::sph_rad[s], &dist, fr[0]);  // distance to the local tangent plane (or a riser's wall)
                contact_frame(fr[0], fr[1], fr[2]);
                sfor<3>([&](auto K) MI_LAMBDA { xc[K] = cs[K] - M::sph_rad[s] * fr[0][K]; });
            } else {
                xc[0] = cs[0]; xc[1] = cs[1]; xc[2] = cs[2] - M::sph_rad[s];
                dist = (root[2] + xc[2]) - P.ground_z;
            }
            const bool on = dist < P.contact_offset;
            // Within an active sphere the build is branch-free: rows of an env that does not touch are built like any
            // other and made inert with Ainv = 0 and lam = 0 (every PGS update then multiplies by zero) -- no EXEC-mask
            // divergence; the only branch is the wave-uniform one.
            const float onf = on ? 1.f : 0.f;
            const float gap = dist - P.rest_offset;
            if (!MI_WAVE_ANY(on)) {
                sfor<3>([&](auto K) MI_LAMBDA { lam(row0 + K) = 0.f; });
                return;
            }
            sph_active |= 1ull << s;
            sfor<3>([&](auto K) MI_LAMBDA {
                constexpr int k = K, row = row0 + k;
                // unit force u at xc as a spatial force [xc x u; u]
                float W[6];
                if constexpr (GND::HEIGHTFIELD) {
                    cross3(xc, fr[k], W);
                    W[3] = fr[k][0]; W[4] = fr[k][1]; W[5] = fr[k][2];
                } else {  // u = z, x, y: exploit the zeros
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
                float a = P.cfm;
                sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { a += g[C] * g[C]; G(row, C) = g[C]; });
                Ainv(row) = onf * MI_RCP(a);
                const float vtn = (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
                // the velocity target of a tangent row is zero and the sweeps never read its slot: on a height field the two slots keep the
                // surface normal (x, y; z > 0 follows) for the net-contact-force pass after the solve, which would otherwise repeat the query
                if constexpr (GND::HEIGHTFIELD) vt(row) = (k == 0) ? vtn : fr[0][k - 1];
                else vt(row) = (k == 0) ? vtn : 0.f;
                float lprev;
                if constexpr (LAM_IN_ROWS) lprev = lam(row); else lprev = lamc(3 * s + k);
                const float l0 = lprev * P.warm * onf;
                lam(row) = l0;
                if constexpr (INLINE_WARM_SPH) sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += g[C] * l0; });
            });
        });
        } else {
        // ground contacts, compact store: a sphere within contact_offset takes the next free slot of its env (at most
        // KMAX); the three rows are only built -- by the lanes that need them -- when some env of the wave has the
        // sphere active (EXEC-masked region, skipped by the whole wave otherwise)
        int cnt = 0, ndrop = 0;
        if (main_wave) sfor<NSPH>([&](auto S_) MI_LAMBDA {
            constexpr int s = S_, b = M::sph_body[s];
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
            const bool on = (dist < P.contact_offset) && (cnt < KMAX);
            ndrop += ((dist < P.contact_offset) && (cnt >= KMAX)) ? 1 : 0;
            const int j = on ? cnt : -1;
            // the parked warm-start impulses of this sphere must be read before its slot (possibly) overwrites them
            float lprev[3];
            sfor<3>([&](auto K) MI_LAMBDA { lprev[K] = rows(C_PARK - 3 * (s + 1) + K); });
            if (on) {
                float* cb = rows.ptr(C_CB + j * C_CSZ);
                constexpr int ST = RowStore<RS>::stride;
                const float gap = dist - P.rest_offset;
                // unit forces of the three rows (normal, two tangents) at xc as spatial forces [xc x u; u]
                float W[3][6];
                sfor<3>([&](auto K) MI_LAMBDA {
                    constexpr int k = K;
                    if constexpr (GND::HEIGHTFIELD) {
                        cross3(xc, fr[k], W[k]);
                        W[k][3] = fr[k][0]; W[k][4] = fr[k][1]; W[k][5] = fr[k][2];
                    } else {
                        constexpr int ax = (k == 0) ? 2 : (k == 1 ? 0 : 1);
                        sfor<6>([&](auto I_) MI_LAMBDA { W[k][I_] = 0.f; });
                        W[k][3 + ax] = 1.f;
                        if constexpr (ax == 0) { W[k][1] = xc[2]; W[k][2] = -xc[1]; }
                        else if constexpr (ax == 1) { W[k][0] = -xc[2]; W[k][2] = xc[0]; }
                        else { W[k][0] = xc[1]; W[k][1] = -xc[0]; }
                    }
                });
                // Jacobian entries of the three rows share the joint subspace S_i (read once from the row store where the tree
                // pass parked it), the chain solve shares the L entries
                float g[3][M::MAXCHAIN];
                sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA {
                    constexpr int gi = M::chain[b][C];
                    if constexpr (gi >= OFF) {
                        float Sd[6];
                        if constexpr (S_IN_ROWS && M::NOS == 0) sfor<6>([&](auto I_) MI_LAMBDA { Sd[I_] = rows(C_S + 6 * (gi - OFF) + I_); });
                        else sfor<6>([&](auto I_) MI_LAMBDA { Sd[I_] = S[gi - OFF][I_]; });
                        sfor<3>([&](auto K) MI_LAMBDA { g[K][C] = dot6(Sd, W[K]); });
                    } else if constexpr (gi < 3) {
                        sfor<3>([&](auto K) MI_LAMBDA { g[K][C] = W[K][3 + gi]; });
                    } else {
                        sfor<3>([&](auto K) MI_LAMBDA { g[K][C] = W[K][gi - 3]; });
                    }
                });
                chain_solve3(std::integral_constant<int, b>{}, g);
                sfor<3>([&](auto K) MI_LAMBDA {
                    constexpr int k = K;
                    float a = P.cfm;
                    sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { a += g[k][C] * g[k][C]; cb[(k * M::MAXCHAIN + C) * ST] = g[k][C]; });
                    cb[(3 * M::MAXCHAIN + k) * ST] = MI_RCP(a);
                    // slots past C_WARM_OK may already have overwritten parked values of later spheres: no warm start there
                    cb[(3 * M::MAXCHAIN + 4 + k) * ST] = (j <= C_WARM_OK) ? lprev[k] * P.warm : 0.f;
                });
                cb[(3 * M::MAXCHAIN + 3) * ST] = (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
            }
            cnt += on ? 1 : 0;
            slot8(rows, s) = (signed char)j;
        });
        // ---- self-collision: per limb-pair group the deepest capsule pair becomes one contact between the two bodies.  The capsule
        // axes end on contact spheres, whose centres the tree pass left in c.xcs.  The Jacobian of the row is J_a - J_b: dofs that
        // carry both bodies drop out, the others enter with the sign of their side -- which side(s) a dof moves depends on the bodies
        // the env's deepest pair happens to join, so it is read from the per-lane chain masks instead of being unrolled per body pair
        // (13 row-build code paths for the Humanoid instead of 66).
        MI_STAMP(9);      // (debug stamps: 4 .. 9 = ground contact rows, 9 .. 5 = self-collision phase)
        if (scol != nullptr && scol->dropped != nullptr && main_wave && ndrop > 0) scol->dropped[0] += ndrop;
        if constexpr (NPG > 0) { if (selfcol && do_pairs) {
        int cntp = 0, pdrop = 0;
        // broad phase: bounding sphere of every capsule (centre = middle of its axis, radius = half length + capsule radius); the
        // narrow phase of a capsule pair is skipped by the whole wave when no env has the two spheres within reach
        float capm[M::NCAP][3];
        sfor<M::NCAP>([&](auto C_) MI_LAMBDA {
            constexpr int cc = C_;
            sfor<3>([&](auto I_) MI_LAMBDA { capm[cc][I_] = 0.5f * (c.xcs[M::cap_s0[cc]][I_] + c.xcs[M::cap_s1[cc]][I_]); });
        });
        sfor<NPG>([&](auto G_) MI_LAMBDA {
            constexpr int g = G_, LEN = M::pg_chain_len[g];
            MI_PHASE();
            float best = 3.0e38f, bca[3] = {0.f, 0.f, 0.f}, bcb[3] = {0.f, 0.f, 0.f};
            int bk = 0;
            sfor<M::pg_count[g]>([&](auto K_) MI_LAMBDA {
                constexpr int k = M::pg_first[g] + K_, ia = M::gp_a[k], ib = M::gp_b[k];
                constexpr float reach = cap_bound(ia) + cap_bound(ib);
                const float dm[3] = {capm[ia][0] - capm[ib][0], capm[ia][1] - capm[ib][1], capm[ia][2] - capm[ib][2]};
                const float rr = reach + P.contact_offset;
                if (!MI_WAVE_ANY(dot3(dm, dm) < rr * rr)) return;
                float ca[3], cb[3];
                seg_seg_closest<cap_is_point(ia), cap_is_point(ib)>(c.xcs[M::cap_s0[ia]], c.xcs[M::cap_s1[ia]], c.xcs[M::cap_s0[ib]],
                                                                      c.xcs[M::cap_s1[ib]], ca, cb);
                const float dv[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
                const float dist = MI_SQRT(dot3(dv, dv)) - (M::cap_rad[ia] + M::cap_rad[ib]);
                const bool better = dist < best;                 // the first pair of the list wins ties
                best = better ? dist : best;
                sfor<3>([&](auto I_) MI_LAMBDA { bca[I_] = better ? ca[I_] : bca[I_]; bcb[I_] = better ? cb[I_] : bcb[I_]; });
                bk = better ? K_ : bk;
            });
            const bool on = (best < P.contact_offset) && (cntp < KPAIR);
            pdrop += ((best < P.contact_offset) && (cntp >= KPAIR)) ? 1 : 0;
            if (MI_WAVE_ANY(on)) {
                if (on) {
                    // what the chosen pair implies (bodies, chain masks, radius of side b, friction): looked up by its index
                    int bab = 0;
                    unsigned mA = 0u, mB = 0u;
                    float brb = 0.f, bmu = 0.f;
                    sfor<M::pg_count[g]>([&](auto K_) MI_LAMBDA {
                        constexpr int k = M::pg_first[g] + K_, ia = M::gp_a[k], ib = M::gp_b[k];
                        const bool me = bk == K_;
                        bab = me ? (M::cap_body[ia] | (M::cap_body[ib] << 8)) : bab;
                        mA = me ? M::chain_mask[M::cap_body[ia]] : mA;
                        mB = me ? M::chain_mask[M::cap_body[ib]] : mB;
                        brb = me ? M::cap_rad[ib] : brb;
                        bmu = me ? 0.5f * (M::cap_mu[ia] + M::cap_mu[ib]) : bmu;
                    });
                    float* pb = rows.ptr(C_PB + cntp * P_CSZ);
                    constexpr int ST = RowStore<RS>::stride;
                    float fr[3][3], x[3];
                    {
                        const float dv[3] = {bca[0] - bcb[0], bca[1] - bcb[1], bca[2] - bcb[2]};
                        const float d2 = dot3(dv, dv);
                        const bool okd = d2 > 1e-18f;
                        const float inv = MI_RSQ(fmaxf(d2, 1e-30f));
                        fr[0][0] = okd ? dv[0] * inv : 0.f; fr[0][1] = okd ? dv[1] * inv : 0.f; fr[0][2] = okd ? dv[2] * inv : 1.f;
                    }
                    contact_frame(fr[0], fr[1], fr[2]);
                    sfor<3>([&](auto I_) MI_LAMBDA { x[I_] = bcb[I_] + fr[0][I_] * (brb + 0.5f * best); });   // middle of the gap / overlap
                    float W[3][6];
                    sfor<3>([&](auto K) MI_LAMBDA {
                        cross3(x, fr[K], W[K]);
                        W[K][3] = fr[K][0]; W[K][4] = fr[K][1]; W[K][5] = fr[K][2];
                    });
                    // Jacobian entries J_a - J_b over the union chain.  A dof below the two contact bodies (a thigh contact does not
                    // involve the knee) has no entry: such dofs are skipped by the whole wave when no env needs them.
                    float gg[3][PCH];
                    sfor<LEN>([&](auto C) MI_LAMBDA {
                        constexpr int gi = M::pg_chain[g][C];
                        sfor<3>([&](auto K) MI_LAMBDA { gg[K][C] = 0.f; });
                        if constexpr (!((M::pg_common[g] >> gi) & 1u)) {       // (a dof that moves both bodies alike has none either)
                            const float cf = (float)((mA >> gi) & 1u) - (float)((mB >> gi) & 1u);
                            if (MI_WAVE_ANY(cf != 0.f)) {
                                if constexpr (gi >= OFF) {
                                    float Sd[6];
                                    if constexpr (S_IN_ROWS && M::NOS == 0) sfor<6>([&](auto I_) MI_LAMBDA { Sd[I_] = rows(C_S + 6 * (gi - OFF) + I_); });
                                    else sfor<6>([&](auto I_) MI_LAMBDA { Sd[I_] = S[gi - OFF][I_]; });
                                    sfor<3>([&](auto K) MI_LAMBDA { gg[K][C] = cf * dot6(Sd, W[K]); });
                                } else if constexpr (gi < 3) {
                                    sfor<3>([&](auto K) MI_LAMBDA { gg[K][C] = cf * W[K][3 + gi]; });
                                } else {
                                    sfor<3>([&](auto K) MI_LAMBDA { gg[K][C] = cf * W[K][gi - 3]; });
                                }
                            }
                        }
                    });
                    // L^T g = J^T over the union chain (descending indices): an entry feeds exactly its ancestors among the later ones;
                    // entries that are zero in every env of the wave feed nothing
                    sfor<LEN>([&](auto C) MI_LAMBDA {
                        constexpr int k = C, i = M::pg_chain[g][k];
                        if (!MI_WAVE_ANY((gg[0][k] != 0.f) || (gg[1][k] != 0.f) || (gg[2][k] != 0.f))) return;
                        const float di = Ldi[i];
                        const float z0 = gg[0][k] * di, z1 = gg[1][k] * di, z2 = gg[2][k] * di;
                        gg[0][k] = z0; gg[1][k] = z1; gg[2][k] = z2;
                        sfor<LEN - 1 - k>([&](auto T) MI_LAMBDA {
                            constexpr int kk = k + 1 + T, jj = M::pg_chain[g][kk];
                            if constexpr (M::midx[i][jj] >= 0) {
                                const float l = L[M::midx[i][jj]];
                                gg[0][kk] -= l * z0; gg[1][kk] -= l * z1; gg[2][kk] -= l * z2;
                            }
                        });
                    });
                    const float gap = best - P.rest_offset;
                    sfor<3>([&](auto K) MI_LAMBDA {
                        constexpr int k = K;
                        float a = P.cfm;
                        sfor<LEN>([&](auto C) MI_LAMBDA { a += gg[k][C] * gg[k][C]; pb[(k * PCH + C) * ST] = gg[k][C]; });
                        pb[(3 * PCH + k) * ST] = MI_RCP(a);
                        pb[(3 * PCH + 4 + k) * ST] = scol->lamp(3 * g + k) * P.warm;
                    });
                    pb[(3 * PCH + 3) * ST] = (gap >= 0.f) ? -gap * invh : fminf(-gap * P.erp * invh, P.max_depen_vel);
                    sfor<KPAIR>([&](auto J_) MI_LAMBDA {
                        const bool me = cntp == J_;
                        sfor<3>([&](auto I_) MI_LAMBDA { pinf[J_][I_] = me ? x[I_] : pinf[J_][I_]; pinf[J_][3 + I_] = me ? fr[0][I_] : pinf[J_][3 + I_]; });
                        pinf[J_][6] = me ? __builtin_bit_cast(float, bab) : pinf[J_][6];
                        pinf[J_][7] = me ? bmu : pinf[J_][7];
                    });
                }
            }
            pmap = (pmap & ~(3u << (2 * g))) | ((unsigned)(on ? cntp : 3) << (2 * g));
            cntp += on ? 1 : 0;
        });
        if (scol->dropped != nullptr && pdrop > 0) scol->dropped[scol->dstride] += pdrop;
        } }
        if constexpr (NPG > 0) {
            // two waves: the helper hands its bookkeeping over through the row store and is done; the main wave waits for it here,
            // with its own limit and ground rows already built
            if (role >= 1) {
                if (selfcol && role == 1) {
                    rows(C_X) = __builtin_bit_cast(float, pmap);
                    sfor<KPAIR>([&](auto J_) MI_LAMBDA { sfor<8>([&](auto I_) MI_LAMBDA { rows(C_X + 1 + 8 * J_ + I_) = pinf[J_][I_]; }); });
                }
                bar();
                return;
            }
            if (role == 0) {
                bar();
                if (selfcol) {
                    pmap = __builtin_bit_cast(unsigned, rows(C_X));
                    sfor<KPAIR>([&](auto J_) MI_LAMBDA { sfor<8>([&](auto I_) MI_LAMBDA { pinf[J_][I_] = rows(C_X + 1 + 8 * J_ + I_); }); });
                }
            }
        }
        }
        MI_PHASE();
        MI_STAMP(5);
        // ------------------------------------------------------------ warm start: w += G^T lam0, rows read back from the store
        // (a separate pass on purpose: accumulating into w while the rows are being built makes the compiler keep
        // every row's g alive until one big batched update -- hundreds of spilled registers)
        {
            int zero;
            MI_OPAQUE_ZERO(zero);
            const RowStore<RS> rit = rows.shifted(zero);
            if constexpr (!INLINE_WARM_LIM) sfor<ND>([&](auto D) MI_LAMBDA {
                constexpr int d = D, gi = OFF + d;
                if constexpr (M::dof_limited[d]) {
                    constexpr int row = limrow(d);
                    const float l0 = lam(row);
                    constexpr int g0 = COMPACT ? limoff(row) : row * M::MAXCHAIN;
                    w[gi] += rit(g0) * l0;
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += rit(g0 + 1 + A_) * l0; });
                }
            });
            if constexpr (INLINE_WARM_SPH) {
            } else if constexpr (!COMPACT) {
                sfor<NSPH>([&](auto S_) MI_LAMBDA {
                    constexpr int s = S_, b = M::sph_body[s], row0 = NLIM + 3 * s;
                    if (sph_active >> s & 1ull) {
                        sfor<3>([&](auto K) MI_LAMBDA {
                            constexpr int row = row0 + K;
                            const float l0 = lam(row);
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += rit(row * M::MAXCHAIN + C) * l0; });
                        });
                    }
                });
            } else {
                sfor<NSPH>([&](auto S_) MI_LAMBDA {
                    constexpr int s = S_, b = M::sph_body[s];
                    const int j = (int)slot8(rit, s);
                    if (j >= 0) {
                        const float* cb = rit.ptr(C_CB + j * C_CSZ);
                        constexpr int ST = RowStore<RS>::stride;
                        sfor<3>([&](auto K) MI_LAMBDA {
                            const float l0 = cb[(3 * M::MAXCHAIN + 4 + K) * ST];
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += cb[(K * M::MAXCHAIN + C) * ST] * l0; });
                        });
                    }
                });
                if constexpr (NPG > 0) { if (selfcol) {
                    sfor<NPG>([&](auto G_) MI_LAMBDA {
                        constexpr int g = G_, LEN = M::pg_chain_len[g];
                        const int j = (int)((pmap >> (2 * g)) & 3u);
                        if (j != 3) {
                            const float* pb = rit.ptr(C_PB + j * P_CSZ);
                            constexpr int ST = RowStore<RS>::stride;
                            sfor<3>([&](auto K) MI_LAMBDA {
                                const float l0 = pb[(3 * PCH + 4 + K) * ST];
                                sfor<LEN>([&](auto C) MI_LAMBDA { w[M::pg_chain[g][C]] += pb[(K * PCH + C) * ST] * l0; });
                            });
                        }
                    });
                } }
            }
        }
#if defined(MI_STOP_AFTER) && MI_STOP_AFTER == 3
        { float acc = 0.f; sfor<M::NM>([&](auto K) MI_LAMBDA { acc += L[K]; }); sfor<NV>([&](auto K) MI_LAMBDA { acc += w[K] + Ldi[K]; });
          sfor<NROWG>([&](auto K) MI_LAMBDA { acc += lam(K); });
          sfor<NSPH>([&](auto K) MI_LAMBDA { sfor<3>([&](auto J) MI_LAMBDA { acc += c.xcs[K][J]; }); });
          sfor<NSENS>([&](auto K) MI_LAMBDA { sfor<9>([&](auto J) MI_LAMBDA { acc += c.Rs[K][J]; }); sfor<3>([&](auto J) MI_LAMBDA { acc += c.rs[K][J]; }); });
          root[0] = acc; return; }
#endif
        MI_PHASE();
        MI_STAMP(6);
        if constexpr (!COMPACT) {
        // ------------------------------------------------------------ projected Gauss-Seidel sweeps
        // Software-pipelined over "units" (one limit row, or the 3 rows of one sphere): while unit u is being
        // solved, the rows of unit u+1 are already being read from the store into the other register buffer.  A wave
        // owns its SIMD alone (512 VGPRs), so nobody else hides the LDS round trip -- the prefetch has to.
        {
            constexpr int NUNIT = NLIM + NSPH;
            struct UBuf { float g[3][M::MAXCHAIN]; float ainv[3], vt[3], lam[3]; };
            UBuf ub[2];
            for (int it = 0; it < P.iters; ++it) {
                int zero;
                MI_OPAQUE_ZERO(zero);
                const RowStore<RS> rit = rows.shifted(zero);
                auto load_unit = [&](auto U_, UBuf& B) MI_LAMBDA {
                    constexpr int u = decltype(U_)::value;
                    if constexpr (u < NLIM) {
                        constexpr int d = limdof(u), gi = OFF + d, row = u;
                        sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { B.g[0][K] = rit(row * M::MAXCHAIN + K); });
                        B.ainv[0] = rit(NROWG * M::MAXCHAIN + row);
                        B.vt[0] = rit(NROWG * M::MAXCHAIN + NROWG + row);
                        B.lam[0] = lam(row);
                    } else if constexpr (u < NUNIT) {
                        constexpr int s = u - NLIM, b = M::sph_body[s], row0 = NLIM + 3 * s;
                        sfor<3>([&](auto K) MI_LAMBDA {
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { B.g[K][C] = rit((row0 + K) * M::MAXCHAIN + C); });
                            B.ainv[K] = rit(NROWG * M::MAXCHAIN + row0 + K);
                            B.lam[K] = lam(row0 + K);
                        });
                        B.vt[0] = rit(NROWG * M::MAXCHAIN + NROWG + row0);  // tangential targets are zero
                    }
                };
                auto unit_on = [&](auto U_) MI_LAMBDA -> bool {    // wave-uniform
                    constexpr int u = decltype(U_)::value;
                    if constexpr (u < NLIM) return true;
                    else if constexpr (u < NUNIT) return (sph_active >> (u - NLIM) & 1ull) != 0ull;
                    else return false;
                };
                if constexpr (NUNIT > 0) load_unit(std::integral_constant<int, 0>{}, ub[0]);
                sfor<NUNIT>([&](auto U_) MI_LAMBDA {
                    constexpr int u = U_;
                    UBuf& B = ub[u & 1];
                    if (unit_on(std::integral_constant<int, u + 1>{}))
                        load_unit(std::integral_constant<int, u + 1>{}, ub[(u + 1) & 1]);  // prefetch (no-op past the end)
                    MI_PHASE();
                    if (!unit_on(std::integral_constant<int, u>{})) return;
                    if constexpr (u < NLIM) {
                        constexpr int d = limdof(u), gi = OFF + d, row = u;
                        float vn = B.g[0][0] * w[gi];
                        sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { vn += B.g[0][1 + A_] * w[M::anc[gi][A_]]; });
                        const float lo = B.lam[0];
                        const float nl = fmaxf(lo - (vn - B.vt[0]) * B.ainv[0], 0.f);
                        const float dl = nl - lo;
                        lam(row) = nl;
                        w[gi] += B.g[0][0] * dl;
                        sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += B.g[0][1 + A_] * dl; });
                    } else {
                        constexpr int s = u - NLIM, b = M::sph_body[s], row0 = NLIM + 3 * s;
                        // inactive spheres have Ainv = lam = 0: every update below is then exactly zero
                        const float mu = 0.5f * ((mu_env >= 0.f ? mu_env : M::sph_mu[s]) + P.plane_mu);
                        float ln;
                        {
                            float vn = 0.f;
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { vn += B.g[0][C] * w[M::chain[b][C]]; });
                            const float lo = B.lam[0];
                            ln = fmaxf(lo - (vn - B.vt[0]) * B.ainv[0], 0.f);
                            const float dl = ln - lo;
                            lam(row0) = ln;
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += B.g[0][C] * dl; });
                        }
                        float lt[2], vtg[2];
                        // both tangent rows from the SAME velocity, the disc (friction_disc), ONE application
                        sfor<2>([&](auto K) MI_LAMBDA {
                            float vn = 0.f;
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { vn += B.g[1 + K][C] * w[M::chain[b][C]]; });
                            vtg[K] = vn;
                            lt[K] = B.lam[1 + K] - vn * B.ainv[1 + K];
                        });
                        friction_disc(lt, B.lam[1], B.lam[2], vtg[0], vtg[1], B.ainv[1], B.ainv[2], mu * ln);
                        sfor<2>([&](auto K) MI_LAMBDA {
                            constexpr int row = row0 + 1 + K;
                            const float nl = lt[K], dl = nl - B.lam[1 + K];
                            lam(row) = nl;
                            sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += B.g[1 + K][C] * dl; });
                        });
                    }
                });
            }
        }
        } else {
        // projected Gauss-Seidel sweeps, compact store: limit rows at their fixed places, then every sphere that holds a slot
        for (int it = 0; it < P.iters; ++it) {
            int zero;
            MI_OPAQUE_ZERO(zero);
            const RowStore<RS> rit = rows.shifted(zero);
            constexpr int ST = RowStore<RS>::stride;
            sfor<ND>([&](auto D) MI_LAMBDA {
                constexpr int d = D, gi = OFF + d;
                if constexpr (M::dof_limited[d]) {
                    constexpr int row = limrow(d), g0 = limoff(row);
                    float g[M::MAXCHAIN];
                    sfor<M::nanc[gi] + 1>([&](auto K) MI_LAMBDA { g[K] = rit(g0 + K); });
                    float vn = g[0] * w[gi];
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { vn += g[1 + A_] * w[M::anc[gi][A_]]; });
                    const float lo = rit(C_LIMG + 2 * NLIM + row);
                    const float nl = fmaxf(lo - (vn - rit(C_LIMG + NLIM + row)) * rit(C_LIMG + row), 0.f);
                    const float dl = nl - lo;
                    rit(C_LIMG + 2 * NLIM + row) = nl;
                    w[gi] += g[0] * dl;
                    sfor<M::nanc[gi]>([&](auto A_) MI_LAMBDA { w[M::anc[gi][A_]] += g[1 + A_] * dl; });
                }
            });
            sfor<NSPH>([&](auto S_) MI_LAMBDA {
                constexpr int s = S_, b = M::sph_body[s];
                const int j = (int)slot8(rit, s);
                if (j >= 0) {
                    float* cb = rit.ptr(C_CB + j * C_CSZ);
                    const float mu = 0.5f * ((mu_env >= 0.f ? mu_env : M::sph_mu[s]) + P.plane_mu);
                    float g[3][M::MAXCHAIN], ainv[3], lm[3];
                    sfor<3>([&](auto K) MI_LAMBDA {
                        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { g[K][C] = cb[(K * M::MAXCHAIN + C) * ST]; });
                        ainv[K] = cb[(3 * M::MAXCHAIN + K) * ST];
                        lm[K] = cb[(3 * M::MAXCHAIN + 4 + K) * ST];
                    });
                    const float vtn = cb[(3 * M::MAXCHAIN + 3) * ST];
                    float ln;
                    {
                        float vn = 0.f;
                        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { vn += g[0][C] * w[M::chain[b][C]]; });
                        ln = fmaxf(lm[0] - (vn - vtn) * ainv[0], 0.f);
                        const float dl = ln - lm[0];
                        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += g[0][C] * dl; });
                    }
                    float lt[2], vtg[2];
                    sfor<2>([&](auto K) MI_LAMBDA {      // (both tangent rows from the same velocity, as above)
                        float vn = 0.f;
                        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { vn += g[1 + K][C] * w[M::chain[b][C]]; });
                        vtg[K] = vn;
                        lt[K] = lm[1 + K] - vn * ainv[1 + K];
                    });
                    friction_disc(lt, lm[1], lm[2], vtg[0], vtg[1], ainv[1], ainv[2], mu * ln);
                    cb[(3 * M::MAXCHAIN + 4) * ST] = ln;
                    sfor<2>([&](auto K) MI_LAMBDA {
                        const float nl = lt[K], dl = nl - lm[1 + K];
                        cb[(3 * M::MAXCHAIN + 5 + K) * ST] = nl;
                        sfor<M::chain_len[b]>([&](auto C) MI_LAMBDA { w[M::chain[b][C]] += g[1 + K][C] * dl; });
                    });
                }
            });
            if constexpr (NPG > 0) { if (selfcol) {
                sfor<NPG>([&](auto G_) MI_LAMBDA {
                    constexpr int g = G_, LEN = M::pg_chain_len[g];
                    const int j = (int)((pmap >> (2 * g)) & 3u);
                    if (j != 3) {
                        float* pb = rit.ptr(C_PB + j * P_CSZ);
                        float mu = pinf[0][7];
                        sfor<KPAIR - 1>([&](auto J_) MI_LAMBDA { mu = (j == J_ + 1) ? pinf[J_ + 1][7] : mu; });
                        float gq[3][PCH], ainv[3], lm[3];
                        sfor<3>([&](auto K) MI_LAMBDA {
                            sfor<LEN>([&](auto C) MI_LAMBDA { gq[K][C] = pb[(K * PCH + C) * ST]; });
                            ainv[K] = pb[(3 * PCH + K) * ST];
                            lm[K] = pb[(3 * PCH + 4 + K) * ST];
                        });
                        const float vtn = pb[(3 * PCH + 3) * ST];
                        float ln;
                        {
                            float vn = 0.f;
                            sfor<LEN>([&](auto C) MI_LAMBDA { vn += gq[0][C] * w[M::pg_chain[g][C]]; });
                            ln = fmaxf(lm[0] - (vn - vtn) * ainv[0], 0.f);
                            const float dl = ln - lm[0];
                            sfor<LEN>([&](auto C) MI_LAMBDA { w[M::pg_chain[g][C]] += gq[0][C] * dl; });
                        }
                        float lt[2], vtg[2];
                        sfor<2>([&](auto K) MI_LAMBDA {      // (both tangent rows from the same velocity, as above)
                            float vn = 0.f;
                            sfor<LEN>([&](auto C) MI_LAMBDA { vn += gq[1 + K][C] * w[M::pg_chain[g][C]]; });
                            vtg[K] = vn;
                            lt[K] = lm[1 + K] - vn * ainv[1 + K];
                        });
                        friction_disc(lt, lm[1], lm[2], vtg[0], vtg[1], ainv[1], ainv[2], mu * ln);
                        pb[(3 * PCH + 4) * ST] = ln;
                        sfor<2>([&](auto K) MI_LAMBDA {
                            const float nl = lt[K], dl = nl - lm[1 + K];
                            pb[(3 * PCH + 5 + K) * ST] = nl;
                            sfor<LEN>([&](auto C) MI_LAMBDA { w[M::pg_chain[g][C]] += gq[1 + K][C] * dl; });
                        });
                    }
                });
            } }
        }
        }
#if defined(MI_STOP_AFTER) && MI_STOP_AFTER == 4
        { float acc = 0.f; sfor<M::NM>([&](auto K) MI_LAMBDA { acc += L[K]; }); sfor<NV>([&](auto K) MI_LAMBDA { acc += w[K] + Ldi[K]; });
          sfor<NSPH>([&](auto K) MI_LAMBDA { sfor<3>([&](auto J) MI_LAMBDA { acc += c.xcs[K][J]; }); });
          sfor<NSENS>([&](auto K) MI_LAMBDA { sfor<9>([&](auto J) MI_LAMBDA { acc += c.Rs[K][J]; }); sfor<3>([&](auto J) MI_LAMBDA { acc += c.rs[K][J]; }); });
          root[0] = acc; return; }
#endif
        MI_PHASE();
        MI_STAMP(7);
        // ------------------------------------------------------------ back to generalised velocity: qd = L^-1 w
        float v[NVA];
        sfor<NV>([&](auto I_) MI_LAMBDA {
            constexpr int i = I_;
            float s = w[i];
            sfor<M::nanc[i]>([&](auto A_) MI_LAMBDA {
                constexpr int j = M::anc[i][A_];
                s -= L[M::midx[i][j]] * v[j];
            });
            v[i] = s * Ldi[i];
        });
        MI_PHASE();
        // ------------------------------------------------------------ impulses -> warm start, sensors, dof forces
        float fs_damp[M::NDA], fs_stiff[M::NDA];
        sfor<ND>([&](auto D) MI_LAMBDA { fs_damp[D] = 1.f; fs_stiff[D] = 1.f; });
        if constexpr (SCALED) { if (actor_scale.p != nullptr) { sfor<ND>([&](auto D) MI_LAMBDA { fs_damp[D] = actor_scale(AS_DAMP + D); fs_stiff[D] = actor_scale(AS_STIFF + D); }); } }
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            float ll = 0.f;
            if constexpr (M::dof_limited[d]) {
                constexpr int row = limrow(d);
                const float dl = q[d] - this->template limit_lower<d>(), du = this->template limit_upper<d>() - q[d];
                ll = (dl < du) ? lam(row) : -lam(row);
            }
            laml(d) = ll;
            float df = tau[d] - M::dof_stiffness[d] * fs_stiff[d] * (q[d] - M::dof_springref[d]) - M::dof_damping[d] * fs_damp[d] * v[OFF + d] + ll * invh;
            if (drv) df += drv->gain_p(d) * (drv->target[d] - q[d]) - drv->gain_d(d) * v[OFF + d];
            dof_force(d) = df;
        });
        float sens[6 * M::NSENSA];
        sfor<6 * NSENS>([&](auto K) MI_LAMBDA { sens[K] = 0.f; });
        float nf[GND::NETF ? NB : 1][3];
        if constexpr (GND::NETF) sfor<NB>([&](auto B_) MI_LAMBDA { nf[B_][0] = nf[B_][1] = nf[B_][2] = 0.f; });
        sfor<NSPH>([&](auto S_) MI_LAMBDA {
            constexpr int s = S_, b = M::sph_body[s], row0 = NLIM + 3 * s;
            // inactive spheres carry lam = 0 => zero warm start and zero sensor / net-force contribution
            float ln, l1, l2;
            if constexpr (COMPACT) {
                const int j = (int)slot8(rows, s);
                const float* cb = rows.ptr(C_CB + (j >= 0 ? j : 0) * C_CSZ);
                constexpr int ST = RowStore<RS>::stride;
                const bool onj = j >= 0;
                ln = onj ? cb[(3 * M::MAXCHAIN + 4) * ST] : 0.f;
                l1 = onj ? cb[(3 * M::MAXCHAIN + 5) * ST] : 0.f;
                l2 = onj ? cb[(3 * M::MAXCHAIN + 6) * ST] : 0.f;
                (void)row0;
            } else {
                if (!(sph_active >> s & 1ull)) {   // wave-uniform: nobody touches with this sphere
                    lamc(3 * s) = 0.f; lamc(3 * s + 1) = 0.f; lamc(3 * s + 2) = 0.f;
                    return;
                }
                ln = lam(row0); l1 = lam(row0 + 1); l2 = lam(row0 + 2);
            }
            lamc(3 * s) = ln; lamc(3 * s + 1) = l1; lamc(3 * s + 2) = l2;
            float f[3], xc[3];
            if constexpr (GND::HEIGHTFIELD) {
                // the frame is re-derived here instead of being kept live through the solve: static store -- from the normal parked in the
                // tangent rows' unused target slots (row build above); compact store -- by repeating the query (root has not moved yet)
                float n[3], t1[3], t2[3];
                if constexpr (!COMPACT) {
                    n[0] = vt(row0 + 1); n[1] = vt(row0 + 2);
                    n[2] = MI_SQRT(fmaxf(1.f - n[0] * n[0] - n[1] * n[1], 0.f));
                } else {
                    float dd;
                    gnd.contact(root[0] + c.xcs[s][0], root[1] + c.xcs[s][1], root[2] + c.xcs[s][2], M::sph_rad[s], &dd, n);
                }
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
            if constexpr (sensor_of(b) >= 0) {
                constexpr int k = sensor_of(b);
                const float arm[3] = {xc[0] - c.rs[k][0], xc[1] - c.rs[k][1], xc[2] - c.rs[k][2]};
                float tq[3], fl[3], tl[3];
                cross3(arm, f, tq);
                matTvec3(c.Rs[k], f, fl); matTvec3(c.Rs[k], tq, tl);
                sfor<3>([&](auto C) MI_LAMBDA { sens[6 * k + C] += fl[C]; sens[6 * k + 3 + C] += tl[C]; });
            }
        });
        // self-contacts: impulses -> warm start, forces on the sensor bodies (+f on side a, -f on side b, at the contact point)
        if constexpr (NPG > 0) { if (selfcol) {
            sfor<NPG>([&](auto G_) MI_LAMBDA {
                constexpr int g = G_;
                const int j = (int)((pmap >> (2 * g)) & 3u);
                const bool onj = j != 3;
                const float* pb = rows.ptr(C_PB + (onj ? j : 0) * P_CSZ);
                constexpr int ST = RowStore<RS>::stride;
                const float ln = onj ? pb[(3 * PCH + 4) * ST] : 0.f, l1 = onj ? pb[(3 * PCH + 5) * ST] : 0.f, l2 = onj ? pb[(3 * PCH + 6) * ST] : 0.f;
                scol->lamp(3 * g) = ln; scol->lamp(3 * g + 1) = l1; scol->lamp(3 * g + 2) = l2;
                float pi[7];
                sfor<7>([&](auto I_) MI_LAMBDA {
                    pi[I_] = pinf[0][I_];
                    sfor<KPAIR - 1>([&](auto J_) MI_LAMBDA { pi[I_] = (j == J_ + 1) ? pinf[J_ + 1][I_] : pi[I_]; });
                });
                float t1[3], t2[3], f[3];
                contact_frame(pi + 3, t1, t2);
                sfor<3>([&](auto K) MI_LAMBDA { f[K] = (pi[3 + K] * ln + t1[K] * l1 + t2[K] * l2) * invh; });
                if (scol->pairf.p) sfor<3>([&](auto K) MI_LAMBDA { scol->pairf(3 * g + K) = f[K]; });
                const int bab = __builtin_bit_cast(int, pi[6]);
                sfor<NSENS>([&](auto K_) MI_LAMBDA {
                    constexpr int k = K_, sb = M::sens_body[k];
                    if constexpr (group_has_body(g, sb)) {
                        const float cf = ((bab & 255) == sb ? 1.f : 0.f) - (((bab >> 8) & 255) == sb ? 1.f : 0.f);
                        const float arm[3] = {pi[0] - c.rs[k][0], pi[1] - c.rs[k][1], pi[2] - c.rs[k][2]};
                        float tq[3], fl[3], tl[3];
                        cross3(arm, f, tq);
                        matTvec3(c.Rs[k], f, fl); matTvec3(c.Rs[k], tq, tl);
                        sfor<3>([&](auto C) MI_LAMBDA { sens[6 * k + C] += cf * fl[C]; sens[6 * k + 3 + C] += cf * tl[C]; });
                    }
                });
            });
        } }
        if constexpr (GND::NETF) sfor<NB>([&](auto B_) MI_LAMBDA { sfor<3>([&](auto K) MI_LAMBDA { netf(3 * B_ + K) = nf[B_][K]; }); });
        sfor<6 * NSENS>([&](auto K) MI_LAMBDA { sensor(K) = sens[K]; });
        MI_PHASE();
        // ------------------------------------------------------------ integrate (semi-implicit Euler)
        sfor<ND>([&](auto D) MI_LAMBDA { qd[D] = v[OFF + D]; q[D] += h * qd[D]; });
        if constexpr (!M::FIXED) {
#if !defined(MI_NO_VEL_CLAMP)   // (measurement builds only)
            {   // AssetOptions.max_angular_velocity / max_linear_velocity, Isaac Gym's defaults (64 rad/s, 1000 m/s): PhysX clamps the body
                // velocities; without it a robot flung into a fast spin (100 rad/s = 0.8 rad per sub-step) gains energy until it is NaN
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
        MI_STAMP(8);
    }
};

}  // namespace mi
