// kernels_jit_twins.hip -- stand-alone C-ABI replacements of the reference's remaining @torch.jit.script task functions
// (SURVEY 8a-ext): same argument order and row-major [n, k] tensors as the jitted signatures, one thread per env.  The
// per-env maths lives in tasks/jit_twins.hpp (line-cited); these are on-demand entry points for tasks whose physics the
// engine does not run yet, so that a reference task file can swap its jitted call for the HIP one.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string>

#include "../../include/mi_engine.h"
#include "tasks/jit_twins.hpp"

using namespace mi;

namespace mi { int abi_fail(const char* msg); }   // mi_engine.hip: records the message for mi_last_error(), returns -1

static_assert(sizeof(MiFrankaCabinetRewardParams) == sizeof(FrankaCabinetRewardParams), "MiFrankaCabinetRewardParams layout");
static_assert(sizeof(MiFrankaCubeStackRewardParams) == sizeof(FrankaCubeStackRewardParams), "MiFrankaCubeStackRewardParams layout");
static_assert(sizeof(MiTrifingerRewardParams) == sizeof(TrifingerRewardParams), "MiTrifingerRewardParams layout");
static_assert(sizeof(MiDextremeRewardParams) == sizeof(DextremeRewardParams), "MiDextremeRewardParams layout");

#define TWIN_OK(name) do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) return abi_fail((std::string(name) + ": " + hipGetErrorString(_e)).c_str()); } while (0)
#define TWIN_LAUNCH(kernel, n, stream, ...) hipLaunchKernelGGL(kernel, dim3(((n) + 127) / 128), dim3(128), 0, (hipStream_t)(stream), n, __VA_ARGS__)

// ------------------------------------------------------------------------------------------------ BallBalance / Ingenuity
__global__ void bbot_reward_kernel(int n, const float* ball_pos, const float* ball_vel, float ball_radius, const long long* reset_in,
                                   const long long* progress, float max_len, float* rew, long long* reset) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float p[3] = {ball_pos[3 * e], ball_pos[3 * e + 1], ball_pos[3 * e + 2]}, v[3] = {ball_vel[3 * e], ball_vel[3 * e + 1], ball_vel[3 * e + 2]};
    bbot_reward(p, v, ball_radius, reset_in[e], progress[e], max_len, rew + e, reset + e);
}
extern "C" int mi_compute_bbot_reward(int n, const float* tray_positions, const float* ball_positions, const float* ball_velocities,
                                      float ball_radius, const int64_t* reset_buf_in, const int64_t* progress_buf, float max_episode_length,
                                      float* rew_buf, int64_t* reset_buf_out, void* stream) {
    if (n <= 0) return 0;
    (void)tray_positions;   // unused by the reference as well (ball_balance.py:459-476)
    if (!ball_positions || !ball_velocities || !reset_buf_in || !progress_buf || !rew_buf || !reset_buf_out)
        return abi_fail("mi_compute_bbot_reward: null argument");
    TWIN_LAUNCH(bbot_reward_kernel, n, stream, ball_positions, ball_velocities, ball_radius, (const long long*)reset_buf_in,
                (const long long*)progress_buf, max_episode_length, rew_buf, (long long*)reset_buf_out);
    TWIN_OK("mi_compute_bbot_reward");
    return 0;
}

__global__ void ingenuity_reward_kernel(int n, const float* pos, const float* target, const float* quat, const float* angvel,
                                        const long long* progress, float max_len, float* rew, long long* reset) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float p[3], t[3], q[4], w[3];
    for (int k = 0; k < 3; ++k) { p[k] = pos[3 * e + k]; t[k] = target[3 * e + k]; w[k] = angvel[3 * e + k]; }
    for (int k = 0; k < 4; ++k) q[k] = quat[4 * e + k];
    ingenuity_reward(p, t, q, w, progress[e], max_len, rew + e, reset + e);
}
extern "C" int mi_compute_ingenuity_reward(int n, const float* root_positions, const float* target_root_positions, const float* root_quats,
                                           const float* root_linvels, const float* root_angvels, const int64_t* reset_buf_in,
                                           const int64_t* progress_buf, float max_episode_length, float* rew_buf, int64_t* reset_buf_out,
                                           void* stream) {
    if (n <= 0) return 0;
    (void)root_linvels; (void)reset_buf_in;   // the reference reads neither (ingenuity.py:410-442; reset_buf only for its shape)
    if (!root_positions || !target_root_positions || !root_quats || !root_angvels || !progress_buf || !rew_buf || !reset_buf_out)
        return abi_fail("mi_compute_ingenuity_reward: null argument");
    TWIN_LAUNCH(ingenuity_reward_kernel, n, stream, root_positions, target_root_positions, root_quats, root_angvels,
                (const long long*)progress_buf, max_episode_length, rew_buf, (long long*)reset_buf_out);
    TWIN_OK("mi_compute_ingenuity_reward");
    return 0;
}

// ------------------------------------------------------------------------------------------------ FrankaCabinet
struct CabinetArgs {
    const long long* reset_in; const long long* progress; const float* actions; int na; const float* cabinet_dof_pos; int ncd;
    const float *fgp, *dgp, *fgr, *dgr, *lf, *rf, *gfa, *dia, *gua, *dua;
    float* rew; long long* reset;
};
__global__ void franka_cabinet_reward_kernel(int n, FrankaCabinetRewardParams p, CabinetArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float act[16];
    const int na = a.na < 16 ? a.na : 16;
    for (int k = 0; k < na; ++k) act[k] = a.actions[(size_t)e * a.na + k];
    float v3[10][3], q[2][4];
    const float* src3[8] = {a.fgp, a.dgp, a.lf, a.rf, a.gfa, a.dia, a.gua, a.dua};
    for (int i = 0; i < 8; ++i) for (int k = 0; k < 3; ++k) v3[i][k] = src3[i][3 * e + k];
    for (int k = 0; k < 4; ++k) { q[0][k] = a.fgr[4 * e + k]; q[1][k] = a.dgr[4 * e + k]; }
    franka_cabinet_reward(p, a.reset_in[e], a.progress[e], act, na, a.cabinet_dof_pos[(size_t)e * a.ncd + 3], v3[0], v3[1], q[0], q[1], v3[2], v3[3],
                          v3[4], v3[5], v3[6], v3[7], a.rew + e, a.reset + e);
}
extern "C" int mi_compute_franka_cabinet_reward(int n, const MiFrankaCabinetRewardParams* p, const int64_t* reset_buf_in, const int64_t* progress_buf,
                                                const float* actions, int num_actions, const float* cabinet_dof_pos, int num_cabinet_dofs,
                                                const float* franka_grasp_pos, const float* drawer_grasp_pos, const float* franka_grasp_rot,
                                                const float* drawer_grasp_rot, const float* franka_lfinger_pos, const float* franka_rfinger_pos,
                                                const float* gripper_forward_axis, const float* drawer_inward_axis, const float* gripper_up_axis,
                                                const float* drawer_up_axis, float* rew_buf, int64_t* reset_buf_out, void* stream) {
    if (n <= 0) return 0;
    if (!p || !reset_buf_in || !progress_buf || !actions || !cabinet_dof_pos || !franka_grasp_pos || !drawer_grasp_pos || !franka_grasp_rot ||
        !drawer_grasp_rot || !franka_lfinger_pos || !franka_rfinger_pos || !gripper_forward_axis || !drawer_inward_axis || !gripper_up_axis ||
        !drawer_up_axis || !rew_buf || !reset_buf_out)
        return abi_fail("mi_compute_franka_cabinet_reward: null argument");
    if (num_cabinet_dofs < 4) return abi_fail("mi_compute_franka_cabinet_reward: cabinet_dof_pos needs the drawer_top_joint column (index 3)");
    if (num_actions < 0 || num_actions > 16) return abi_fail("mi_compute_franka_cabinet_reward: num_actions must be within 0..16");
    FrankaCabinetRewardParams fp;
    memcpy(&fp, p, sizeof(fp));
    CabinetArgs a{(const long long*)reset_buf_in, (const long long*)progress_buf, actions, num_actions, cabinet_dof_pos, num_cabinet_dofs,
                  franka_grasp_pos, drawer_grasp_pos, franka_grasp_rot, drawer_grasp_rot, franka_lfinger_pos, franka_rfinger_pos,
                  gripper_forward_axis, drawer_inward_axis, gripper_up_axis, drawer_up_axis, rew_buf, (long long*)reset_buf_out};
    TWIN_LAUNCH(franka_cabinet_reward_kernel, n, stream, fp, a);
    TWIN_OK("mi_compute_franka_cabinet_reward");
    return 0;
}

__global__ void grasp_transforms_kernel(int n, const float* hand_rot, const float* hand_pos, const float* flr, const float* flp,
                                        const float* drawer_rot, const float* drawer_pos, const float* dlr, const float* dlp,
                                        float* gfr, float* gfp, float* gdr, float* gdp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float q1[4], t1[3], q2[4], t2[3], q[4], t[3];
    for (int k = 0; k < 4; ++k) { q1[k] = hand_rot[4 * e + k]; q2[k] = flr[4 * e + k]; }
    for (int k = 0; k < 3; ++k) { t1[k] = hand_pos[3 * e + k]; t2[k] = flp[3 * e + k]; }
    tf_combine(q1, t1, q2, t2, q, t);
    for (int k = 0; k < 4; ++k) gfr[4 * e + k] = q[k];
    for (int k = 0; k < 3; ++k) gfp[3 * e + k] = t[k];
    for (int k = 0; k < 4; ++k) { q1[k] = drawer_rot[4 * e + k]; q2[k] = dlr[4 * e + k]; }
    for (int k = 0; k < 3; ++k) { t1[k] = drawer_pos[3 * e + k]; t2[k] = dlp[3 * e + k]; }
    tf_combine(q1, t1, q2, t2, q, t);
    for (int k = 0; k < 4; ++k) gdr[4 * e + k] = q[k];
    for (int k = 0; k < 3; ++k) gdp[3 * e + k] = t[k];
}
extern "C" int mi_compute_grasp_transforms(int n, const float* hand_rot, const float* hand_pos, const float* franka_local_grasp_rot,
                                           const float* franka_local_grasp_pos, const float* drawer_rot, const float* drawer_pos,
                                           const float* drawer_local_grasp_rot, const float* drawer_local_grasp_pos, float* global_franka_rot,
                                           float* global_franka_pos, float* global_drawer_rot, float* global_drawer_pos, void* stream) {
    if (n <= 0) return 0;
    if (!hand_rot || !hand_pos || !franka_local_grasp_rot || !franka_local_grasp_pos || !drawer_rot || !drawer_pos || !drawer_local_grasp_rot ||
        !drawer_local_grasp_pos || !global_franka_rot || !global_franka_pos || !global_drawer_rot || !global_drawer_pos)
        return abi_fail("mi_compute_grasp_transforms: null argument");
    TWIN_LAUNCH(grasp_transforms_kernel, n, stream, hand_rot, hand_pos, franka_local_grasp_rot, franka_local_grasp_pos, drawer_rot, drawer_pos,
                drawer_local_grasp_rot, drawer_local_grasp_pos, global_franka_rot, global_franka_pos, global_drawer_rot, global_drawer_pos);
    TWIN_OK("mi_compute_grasp_transforms");
    return 0;
}

// ------------------------------------------------------------------------------------------------ FrankaCubeStack
__global__ void axisangle2quat_kernel(int n, const float* vec, float eps, float* quat) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float v[3] = {vec[3 * e], vec[3 * e + 1], vec[3 * e + 2]};
    float q[4];
    axisangle2quat(v, eps, q);
    for (int k = 0; k < 4; ++k) quat[4 * e + k] = q[k];
}
extern "C" int mi_axisangle2quat(int n, const float* vec, float eps, float* quat, void* stream) {
    if (n <= 0) return 0;
    if (!vec || !quat) return abi_fail("mi_axisangle2quat: null argument");
    TWIN_LAUNCH(axisangle2quat_kernel, n, stream, vec, eps, quat);
    TWIN_OK("mi_axisangle2quat");
    return 0;
}
__global__ void franka_cube_stack_reward_kernel(int n, FrankaCubeStackRewardParams p, const long long* reset_in, const long long* progress,
                                                const float* cubeA_size, const float* cubeB_size, const float* cubeA_pos, const float* cubeA_rel,
                                                const float* lf, const float* rf, const float* a2b, float* rew, long long* reset) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float v[5][3];
    const float* src[5] = {cubeA_pos, cubeA_rel, lf, rf, a2b};
    for (int i = 0; i < 5; ++i) for (int k = 0; k < 3; ++k) v[i][k] = src[i][3 * e + k];
    franka_cube_stack_reward(p, reset_in[e], progress[e], cubeA_size[e], cubeB_size[e], v[0], v[1], v[2], v[3], v[4], rew + e, reset + e);
}
extern "C" int mi_compute_franka_cube_stack_reward(int n, const MiFrankaCubeStackRewardParams* p, const int64_t* reset_buf_in,
                                                   const int64_t* progress_buf, const float* cubeA_size, const float* cubeB_size,
                                                   const float* cubeA_pos, const float* cubeA_pos_relative, const float* eef_lf_pos,
                                                   const float* eef_rf_pos, const float* cubeA_to_cubeB_pos, float* rew_buf,
                                                   int64_t* reset_buf_out, void* stream) {
    if (n <= 0) return 0;
    if (!p || !reset_buf_in || !progress_buf || !cubeA_size || !cubeB_size || !cubeA_pos || !cubeA_pos_relative || !eef_lf_pos || !eef_rf_pos ||
        !cubeA_to_cubeB_pos || !rew_buf || !reset_buf_out)
        return abi_fail("mi_compute_franka_cube_stack_reward: null argument");
    FrankaCubeStackRewardParams fp;
    memcpy(&fp, p, sizeof(fp));
    TWIN_LAUNCH(franka_cube_stack_reward_kernel, n, stream, fp, (const long long*)reset_buf_in, (const long long*)progress_buf, cubeA_size, cubeB_size,
                cubeA_pos, cubeA_pos_relative, eef_lf_pos, eef_rf_pos, cubeA_to_cubeB_pos, rew_buf, (long long*)reset_buf_out);
    TWIN_OK("mi_compute_franka_cube_stack_reward");
    return 0;
}

// ------------------------------------------------------------------------------------------------ AllegroHand
__global__ void randomize_rotation_pen_kernel(int n, const float* rand0, const float* rand1, float max_angle, const float* x_unit,
                                              const float* y_unit, const float* z_unit, float* out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float x[3] = {x_unit[3 * e], x_unit[3 * e + 1], x_unit[3 * e + 2]}, y[3] = {y_unit[3 * e], y_unit[3 * e + 1], y_unit[3 * e + 2]},
                z[3] = {z_unit[3 * e], z_unit[3 * e + 1], z_unit[3 * e + 2]};
    float q[4];
    randomize_rotation_pen(rand0[e], rand1[e], max_angle, x, y, z, q);
    for (int k = 0; k < 4; ++k) out[4 * e + k] = q[k];
}
extern "C" int mi_randomize_rotation_pen(int n, const float* rand0, const float* rand1, float max_angle, const float* x_unit, const float* y_unit,
                                         const float* z_unit, float* out_quat, void* stream) {
    if (n <= 0) return 0;
    if (!rand0 || !rand1 || !x_unit || !y_unit || !z_unit || !out_quat) return abi_fail("mi_randomize_rotation_pen: null argument");
    TWIN_LAUNCH(randomize_rotation_pen_kernel, n, stream, rand0, rand1, max_angle, x_unit, y_unit, z_unit, out_quat);
    TWIN_OK("mi_randomize_rotation_pen");
    return 0;
}

// ------------------------------------------------------------------------------------------------ Trifinger
__global__ void lgsk_kernel(int n, const float* x, float scale, float eps, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = lgsk(x[i], scale, eps);
}
extern "C" int mi_lgsk_kernel(int n, const float* x, float scale, float eps, float* out, void* stream) {
    if (n <= 0) return 0;
    if (!x || !out) return abi_fail("mi_lgsk_kernel: null argument");
    TWIN_LAUNCH(lgsk_kernel, n, stream, x, scale, eps, out);
    TWIN_OK("mi_lgsk_kernel");
    return 0;
}
struct Size3 { float v[3]; };
__global__ void gen_keypoints_kernel(int n, const float* pose, int pose_stride, Size3 size, float* out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float p[7], kp[24];
    for (int k = 0; k < 7; ++k) p[k] = pose[(size_t)e * pose_stride + k];
    gen_keypoints(p, size.v, kp);
    for (int k = 0; k < 24; ++k) out[(size_t)e * 24 + k] = kp[k];
}
extern "C" int mi_gen_keypoints(int n, const float* pose, int pose_stride, const float* size3, float* keypoints, void* stream) {
    if (n <= 0) return 0;
    if (!pose || !size3 || !keypoints || pose_stride < 7) return abi_fail("mi_gen_keypoints: bad argument");
    Size3 s{{size3[0], size3[1], size3[2]}};
    TWIN_LAUNCH(gen_keypoints_kernel, n, stream, pose, pose_stride, s, keypoints);
    TWIN_OK("mi_gen_keypoints");
    return 0;
}
__global__ void trifinger_reward_kernel(int n, TrifingerRewardParams p, const long long* progress, const float* goal, const float* obj,
                                        const float* last_obj, const float* ft, const float* last_ft, float* rew, long long* reset,
                                        float* info_move, float* info_reach) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float g[7], o[13], lo[13], f[39], lf[39];
    for (int k = 0; k < 7; ++k) g[k] = goal[(size_t)e * 7 + k];
    for (int k = 0; k < 13; ++k) { o[k] = obj[(size_t)e * 13 + k]; lo[k] = last_obj[(size_t)e * 13 + k]; }
    for (int k = 0; k < 39; ++k) { f[k] = ft[(size_t)e * 39 + k]; lf[k] = last_ft[(size_t)e * 39 + k]; }
    float r, mv, rc;
    long long rs;
    trifinger_reward(p, progress[e], g, o, lo, f, lf, &r, &rs, &mv, &rc);
    rew[e] = r; reset[e] = rs;
    if (info_move) info_move[e] = mv;
    if (info_reach) info_reach[e] = rc;
}
extern "C" int mi_compute_trifinger_reward(int n, const MiTrifingerRewardParams* p, const int64_t* progress_buf, const float* object_goal_poses,
                                           const float* object_state, const float* last_object_state, const float* fingertip_state,
                                           const float* last_fingertip_state, float* rew_buf, int64_t* reset_buf_out,
                                           float* info_finger_movement_penalty, float* info_finger_reach_object_reward, void* stream) {
    if (n <= 0) return 0;
    if (!p || !progress_buf || !object_goal_poses || !object_state || !last_object_state || !fingertip_state || !last_fingertip_state || !rew_buf ||
        !reset_buf_out)
        return abi_fail("mi_compute_trifinger_reward: null argument");
    if (p->dt <= 0.f) return abi_fail("mi_compute_trifinger_reward: dt must be positive");
    TrifingerRewardParams tp;
    memcpy(&tp, p, sizeof(tp));
    TWIN_LAUNCH(trifinger_reward_kernel, n, stream, tp, (const long long*)progress_buf, object_goal_poses, object_state, last_object_state,
                fingertip_state, last_fingertip_state, rew_buf, (long long*)reset_buf_out, info_finger_movement_penalty,
                info_finger_reach_object_reward);
    TWIN_OK("mi_compute_trifinger_reward");
    return 0;
}
// compute_trifinger_observations_states (trifinger.py:1386-1420): obs = dof_pos 9 | dof_vel 9 | object pose 7 | goal pose 7 | actions 9;
// states = obs | object vel 6 | fingertip_state 39 | joint_torques 9 | tip_wrenches 18 (asymmetric) or obs
__global__ void trifinger_obs_states_kernel(int n, int asym, int nd, int na, int nft, int nw, const float* dof_pos, const float* dof_vel, const float* obj,
                                            const float* goal, const float* actions, const float* ft, const float* tau, const float* wrench,
                                            float* obs, float* states) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int nobs = 2 * nd + 14 + na, nst = asym ? nobs + 6 + nft + nd + nw : nobs;
    float* o = obs + (size_t)e * nobs;
    float* s = states ? states + (size_t)e * nst : nullptr;
    int w = 0;
    auto put = [&](float x) { o[w] = x; if (s) s[w] = x; ++w; };
    for (int k = 0; k < nd; ++k) put(dof_pos[(size_t)e * nd + k]);
    for (int k = 0; k < nd; ++k) put(dof_vel[(size_t)e * nd + k]);
    for (int k = 0; k < 7; ++k) put(obj[(size_t)e * 13 + k]);
    for (int k = 0; k < 7; ++k) put(goal[(size_t)e * 7 + k]);
    for (int k = 0; k < na; ++k) put(actions[(size_t)e * na + k]);
    if (s && asym) {
        for (int k = 0; k < 6; ++k) s[w++] = obj[(size_t)e * 13 + 7 + k];
        for (int k = 0; k < nft; ++k) s[w++] = ft[(size_t)e * nft + k];
        for (int k = 0; k < nd; ++k) s[w++] = tau[(size_t)e * nd + k];
        for (int k = 0; k < nw; ++k) s[w++] = wrench[(size_t)e * nw + k];
    }
}
extern "C" int mi_compute_trifinger_observations_states(int n, int asymmetric_obs, int num_dofs, int num_actions, int fingertip_state_cols,
                                                        int tip_wrench_cols, const float* dof_position, const float* dof_velocity,
                                                        const float* object_state, const float* object_goal_poses, const float* actions,
                                                        const float* fingertip_state, const float* joint_torques, const float* tip_wrenches,
                                                        float* obs_buf, float* states_buf, void* stream) {
    if (n <= 0) return 0;
    if (!dof_position || !dof_velocity || !object_state || !object_goal_poses || !actions || !obs_buf)
        return abi_fail("mi_compute_trifinger_observations_states: null argument");
    if (asymmetric_obs && states_buf && (!fingertip_state || !joint_torques || !tip_wrenches))
        return abi_fail("mi_compute_trifinger_observations_states: asymmetric states need fingertip_state, joint_torques, tip_wrenches");
    if (num_dofs < 0 || num_actions < 0 || fingertip_state_cols < 0 || tip_wrench_cols < 0)
        return abi_fail("mi_compute_trifinger_observations_states: negative size");
    TWIN_LAUNCH(trifinger_obs_states_kernel, n, stream, asymmetric_obs, num_dofs, num_actions, fingertip_state_cols, tip_wrench_cols, dof_position,
                dof_velocity, object_state, object_goal_poses, actions, fingertip_state, joint_torques, tip_wrenches, obs_buf, states_buf);
    TWIN_OK("mi_compute_trifinger_observations_states");
    return 0;
}

// cuboid-pose samplers (trifinger.py:1427-1512) on injected draws: kind 0 random_xy (rand [n,2] -> out [n,2]), 1 random_z (rand [n] -> [n]),
// 2 default_orientation (-> [n,4]), 3 random_orientation (randn [n,4] -> [n,4]), 4 random_orientation_within_angle (rand [n,3], base [n,4]
// -> [n,4]), 5 random_angular_vel (randn [n,4] = axis 3 + magnitude 1 -> [n,3]), 6 random_yaw_orientation (rand [n] -> [n,4])
__global__ void trifinger_sample_kernel(int n, int kind, const float* draws, const float* base, float a0, float a1, float* out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float q[4];
    switch (kind) {
        case 0: { const float u[2] = {draws[2 * e], draws[2 * e + 1]}; tri_random_xy(u, a0, out + 2 * e, out + 2 * e + 1); return; }
        case 1: out[e] = tri_random_z(draws[e], a0, a1); return;
        case 2: q[0] = q[1] = q[2] = 0.f; q[3] = 1.f; break;
        case 3: { const float g[4] = {draws[4 * e], draws[4 * e + 1], draws[4 * e + 2], draws[4 * e + 3]}; tri_random_orientation(g, q); break; }
        case 4: { const float u[3] = {draws[3 * e], draws[3 * e + 1], draws[3 * e + 2]};
                  const float b[4] = {base[4 * e], base[4 * e + 1], base[4 * e + 2], base[4 * e + 3]};
                  tri_random_orientation_within_angle(u, b, a0, q); break; }
        case 5: { const float g[4] = {draws[4 * e], draws[4 * e + 1], draws[4 * e + 2], draws[4 * e + 3]}; float w[3]; tri_random_angular_vel(g, a0, w);
                  for (int k = 0; k < 3; ++k) out[3 * e + k] = w[k]; return; }
        default: tri_random_yaw_orientation(draws[e], q); break;
    }
    for (int k = 0; k < 4; ++k) out[4 * e + k] = q[k];
}
static int trifinger_sample(const char* name, int n, int kind, const float* draws, const float* base, float a0, float a1, float* out, void* stream) {
    if (n <= 0) return 0;
    if (!out || (kind != 2 && !draws) || (kind == 4 && !base)) return abi_fail((std::string(name) + ": null argument").c_str());
    TWIN_LAUNCH(trifinger_sample_kernel, n, stream, kind, draws, base, a0, a1, out);
    TWIN_OK(name);
    return 0;
}
extern "C" int mi_trifinger_random_xy(int n, const float* rand2, float max_com_distance_to_center, float* xy, void* stream) {
    return trifinger_sample("mi_trifinger_random_xy", n, 0, rand2, nullptr, max_com_distance_to_center, 0.f, xy, stream);
}
extern "C" int mi_trifinger_random_z(int n, const float* rand1, float min_height, float max_height, float* z, void* stream) {
    return trifinger_sample("mi_trifinger_random_z", n, 1, rand1, nullptr, min_height, max_height, z, stream);
}
extern "C" int mi_trifinger_default_orientation(int n, float* quat, void* stream) {
    return trifinger_sample("mi_trifinger_default_orientation", n, 2, nullptr, nullptr, 0.f, 0.f, quat, stream);
}
extern "C" int mi_trifinger_random_orientation(int n, const float* randn4, float* quat, void* stream) {
    return trifinger_sample("mi_trifinger_random_orientation", n, 3, randn4, nullptr, 0.f, 0.f, quat, stream);
}
extern "C" int mi_trifinger_random_orientation_within_angle(int n, const float* rand3, const float* base, float max_angle, float* quat, void* stream) {
    return trifinger_sample("mi_trifinger_random_orientation_within_angle", n, 4, rand3, base, max_angle, 0.f, quat, stream);
}
extern "C" int mi_trifinger_random_angular_vel(int n, const float* randn4, float magnitude_stdev, float* angvel, void* stream) {
    return trifinger_sample("mi_trifinger_random_angular_vel", n, 5, randn4, nullptr, magnitude_stdev, 0.f, angvel, stream);
}
extern "C" int mi_trifinger_random_yaw_orientation(int n, const float* rand1, float* quat, void* stream) {
    return trifinger_sample("mi_trifinger_random_yaw_orientation", n, 6, rand1, nullptr, 0.f, 0.f, quat, stream);
}

// ------------------------------------------------------------------------------------------------ HumanoidAMP
__global__ void amp_dof_to_obs_kernel(int n, const float* pose, float* out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float p[kAmpDof], o[kAmpDofObs];
    for (int k = 0; k < kAmpDof; ++k) p[k] = pose[(size_t)e * kAmpDof + k];
    amp_dof_to_obs(p, o);
    for (int k = 0; k < kAmpDofObs; ++k) out[(size_t)e * kAmpDofObs + k] = o[k];
}
extern "C" int mi_amp_dof_to_obs(int n, const float* pose, float* dof_obs, void* stream) {
    if (n <= 0) return 0;
    if (!pose || !dof_obs) return abi_fail("mi_amp_dof_to_obs: null argument");
    TWIN_LAUNCH(amp_dof_to_obs_kernel, n, stream, pose, dof_obs);
    TWIN_OK("mi_amp_dof_to_obs");
    return 0;
}
constexpr int kAmpMaxKey = 8;
__global__ void amp_observations_kernel(int n, const float* root_states, const float* dof_pos, const float* dof_vel, const float* key_body_pos, int nk,
                                        int local_root_obs, float* obs) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float root[13], q[kAmpDof], qd[kAmpDof], key[3 * kAmpMaxKey], o[13 + kAmpDofObs + kAmpDof + 3 * kAmpMaxKey];
    for (int k = 0; k < 13; ++k) root[k] = root_states[(size_t)e * 13 + k];
    for (int k = 0; k < kAmpDof; ++k) { q[k] = dof_pos[(size_t)e * kAmpDof + k]; qd[k] = dof_vel[(size_t)e * kAmpDof + k]; }
    for (int k = 0; k < 3 * nk; ++k) key[k] = key_body_pos[(size_t)e * 3 * nk + k];
    amp_observations(root, q, qd, key, nk, local_root_obs != 0, o);
    const int nobs = 13 + kAmpDofObs + kAmpDof + 3 * nk;
    for (int k = 0; k < nobs; ++k) obs[(size_t)e * nobs + k] = o[k];
}
extern "C" int mi_compute_humanoid_amp_observations(int n, const float* root_states, const float* dof_pos, const float* dof_vel,
                                                    const float* key_body_pos, int num_key_bodies, int local_root_obs, float* obs, void* stream) {
    if (n <= 0) return 0;
    if (!root_states || !dof_pos || !dof_vel || !key_body_pos || !obs) return abi_fail("mi_compute_humanoid_amp_observations: null argument");
    if (num_key_bodies < 0 || num_key_bodies > kAmpMaxKey) return abi_fail("mi_compute_humanoid_amp_observations: num_key_bodies must be within 0..8");
    TWIN_LAUNCH(amp_observations_kernel, n, stream, root_states, dof_pos, dof_vel, key_body_pos, num_key_bodies, local_root_obs, obs);
    TWIN_OK("mi_compute_humanoid_amp_observations");
    return 0;
}
__global__ void amp_reward_kernel(int n, float* rew) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) rew[e] = 1.f;
}
extern "C" int mi_compute_humanoid_amp_reward(int n, const float* obs_buf, float* rew_buf, void* stream) {
    if (n <= 0) return 0;
    (void)obs_buf;   // shape only (humanoid_amp_base.py:530-534: the task reward is constant, the style reward comes from the discriminator)
    if (!rew_buf) return abi_fail("mi_compute_humanoid_amp_reward: null argument");
    TWIN_LAUNCH(amp_reward_kernel, n, stream, rew_buf);
    TWIN_OK("mi_compute_humanoid_amp_reward");
    return 0;
}
__global__ void amp_reset_kernel(int n, const long long* progress, const float* contact, const float* body_pos, int nb, unsigned long long mask,
                                 float max_len, int early, float term_h, long long* reset, long long* terminated) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    amp_reset(progress[e], contact + (size_t)e * 3 * nb, body_pos + (size_t)e * 3 * nb, nb, mask, max_len, early != 0, term_h, reset + e, terminated + e);
}
extern "C" int mi_compute_humanoid_amp_reset(int n, const int64_t* reset_buf_in, const int64_t* progress_buf, const float* contact_buf,
                                             const int64_t* contact_body_ids, int num_contact_body_ids, const float* rigid_body_pos, int num_bodies,
                                             float max_episode_length, int enable_early_termination, float termination_height,
                                             int64_t* reset_buf_out, int64_t* terminated_out, void* stream) {
    if (n <= 0) return 0;
    (void)reset_buf_in;   // shape only (humanoid_amp_base.py:539,562)
    if (!progress_buf || !contact_buf || !rigid_body_pos || !reset_buf_out || !terminated_out || (num_contact_body_ids > 0 && !contact_body_ids))
        return abi_fail("mi_compute_humanoid_amp_reset: null argument");
    if (num_bodies < 1 || num_bodies > 64) return abi_fail("mi_compute_humanoid_amp_reset: num_bodies must be within 1..64");
    unsigned long long mask = 0;   // contact_body_ids is a short HOST list (the reference builds it once at start-up, humanoid_amp_base.py:308-321)
    for (int i = 0; i < num_contact_body_ids; ++i) {
        if (contact_body_ids[i] < 0 || contact_body_ids[i] >= num_bodies) return abi_fail("mi_compute_humanoid_amp_reset: contact body id out of range");
        mask |= 1ull << contact_body_ids[i];
    }
    TWIN_LAUNCH(amp_reset_kernel, n, stream, (const long long*)progress_buf, contact_buf, rigid_body_pos, num_bodies, mask, max_episode_length,
                enable_early_termination, termination_height, (long long*)reset_buf_out, (long long*)terminated_out);
    TWIN_OK("mi_compute_humanoid_amp_reset");
    return 0;
}

// ------------------------------------------------------------------------------------------------ AllegroHand DeXtreme
struct DextremeArgs {
    long long *reset_buf, *reset_goal_buf, *progress_buf, *hold_count_buf;
    const float *cur_targets, *prev_targets, *hand_dof_vel; int nd;
    float* successes;
    const float *object_pos, *object_rot, *target_pos, *target_rot, *actions; int na;
    float* rew; float* terms8; float* ws;
};
__global__ void dextreme_reward_kernel(int n, DextremeRewardParams p, DextremeArgs a) {
    const int e0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e0 < n;
    const int e = valid ? e0 : n - 1;
    float ct[32], pt[32], dv[32], act[32];
    for (int k = 0; k < a.nd; ++k) { ct[k] = a.cur_targets[(size_t)e * a.nd + k]; pt[k] = a.prev_targets[(size_t)e * a.nd + k]; dv[k] = a.hand_dof_vel[(size_t)e * a.nd + k]; }
    for (int k = 0; k < a.na; ++k) act[k] = a.actions[(size_t)e * a.na + k];
    float op[3], tp[3], orot[4], trot[4];
    for (int k = 0; k < 3; ++k) { op[k] = a.object_pos[3 * e + k]; tp[k] = a.target_pos[3 * e + k]; }
    for (int k = 0; k < 4; ++k) { orot[k] = a.object_rot[4 * e + k]; trot[k] = a.target_rot[4 * e + k]; }
    long long prog = a.progress_buf[e], hold = a.hold_count_buf[e], rs, gr;
    float succ = a.successes[e], r, out8[8];
    dextreme_reward(p, a.reset_buf[e], a.reset_goal_buf[e], &prog, &hold, ct, pt, dv, a.nd, &succ, op, orot, tp, trot, act, a.na, &r, &rs, &gr, out8);
    float nres = valid ? (float)rs : 0.f, fin = valid ? succ * (float)rs : 0.f;
    for (int o = 32; o > 0; o >>= 1) { nres += __shfl_xor(nres, o, 64); fin += __shfl_xor(fin, o, 64); }
    if ((threadIdx.x & 63) == 0 && nres > 0.f) { atomicAdd(a.ws, nres); atomicAdd(a.ws + 1, fin); }
    if (!valid) return;
    a.rew[e] = r; a.reset_buf[e] = rs; a.reset_goal_buf[e] = gr; a.progress_buf[e] = prog; a.hold_count_buf[e] = hold; a.successes[e] = succ;
    if (a.terms8) for (int k = 0; k < 8; ++k) a.terms8[(size_t)k * n + e] = out8[k];
}
__global__ void dextreme_finalize_kernel(float av_factor, const float* ws, float* consecutive_successes) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float num_resets = ws[0], finished = ws[1], cs = consecutive_successes[0];
        consecutive_successes[0] = (num_resets > 0.f) ? av_factor * finished / num_resets + (1.0f - av_factor) * cs : cs;
    }
}
extern "C" int mi_compute_hand_reward_dextreme(int n, const MiDextremeRewardParams* p, float* rew_buf, int64_t* reset_buf, int64_t* reset_goal_buf,
                                               int64_t* progress_buf, int64_t* hold_count_buf, const float* cur_targets, const float* prev_targets,
                                               const float* hand_dof_vel, int num_dofs, float* successes, float* consecutive_successes,
                                               const float* object_pos, const float* object_rot, const float* target_pos, const float* target_rot,
                                               const float* actions, int num_actions, float* reward_terms8, float* workspace2, void* stream) {
    if (n <= 0) return 0;
    if (!p || !rew_buf || !reset_buf || !reset_goal_buf || !progress_buf || !hold_count_buf || !cur_targets || !prev_targets || !hand_dof_vel ||
        !successes || !consecutive_successes || !object_pos || !object_rot || !target_pos || !target_rot || !actions || !workspace2)
        return abi_fail("mi_compute_hand_reward_dextreme: null argument");
    if (num_dofs < 0 || num_dofs > 32 || num_actions < 0 || num_actions > 32) return abi_fail("mi_compute_hand_reward_dextreme: num_dofs / num_actions must be within 0..32");
    DextremeRewardParams dp;
    memcpy(&dp, p, sizeof(dp));
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(workspace2, 0, 2 * sizeof(float), s) != hipSuccess) return abi_fail("mi_compute_hand_reward_dextreme: hipMemsetAsync failed");
    DextremeArgs a{(long long*)reset_buf, (long long*)reset_goal_buf, (long long*)progress_buf, (long long*)hold_count_buf, cur_targets, prev_targets,
                   hand_dof_vel, num_dofs, successes, object_pos, object_rot, target_pos, target_rot, actions, num_actions, rew_buf, reward_terms8,
                   workspace2};
    hipLaunchKernelGGL(dextreme_reward_kernel, dim3((n + 63) / 64), dim3(64), 0, s, n, dp, a);
    hipLaunchKernelGGL(dextreme_finalize_kernel, dim3(1), dim3(64), 0, s, dp.av_factor, workspace2, consecutive_successes);
    TWIN_OK("mi_compute_hand_reward_dextreme");
    return 0;
}
