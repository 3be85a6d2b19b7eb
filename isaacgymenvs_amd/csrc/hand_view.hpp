// hand_view.hpp -- arena view of the in-hand manipulation tasks' extra tensors (ShadowHand, AllegroHand): device or host pointers, SoA [k][N]
// unless noted.  One definition for the HIP library, its kernels and the CPU backend; where the tensors sit: task_views.hpp build_hand_layout.
#pragma once

namespace mi {

struct HandView {
    float* cur_targets;    // [ND][N]
    float* prev_targets;   // [ND][N]
    float* object_state;   // [13][N]  root state of the object
    float* goal_state;     // [7][N]   goal pose (pos, quat)
    float* fingertip;      // [5*13][N] fingertip body states
    float* successes;      // [N]
    long long* reset_goal; // [N]
    int* goal_count;       // [N] number of goal resets so far (RNG counter)
    float* cons;           // [1] consecutive_successes (shadow_hand.py:795-798)
    float* ws;             // [4] per-step scratch of the cross-env sums
    int* ncontact;         // [N] object contacts of the last sub-step (diagnostic)
    float* full_state;     // [N][NFULL] row-major: compute_full_state's vector when it is not obs_buf itself (states_buf, :584)
    float* obj_force;      // [3][N] world-frame force on the object during this control step (apply_rigid_body_force_tensors)
    float* rb_force;       // [3][N] rb_forces[:, object] in the object's local frame (:201, 700-708)
    float* force_prob;     // [N] random_force_prob (:198-199)
    float* mu_env;         // [N] per-env hand-object contact friction for actor_params friction randomisation; negative = HandParams.mu
    float* scale;          // [8][N] per-env `actor_params` factors (core/hand_engine.hpp HS_*): hand link masses, joint damping, drive stiffness,
                           //        tendon limit stiffness / damping, object mass, object size; 1 = the model's own values
    float* limit_shift;    // [2*ND][N] per-env shifts of the lower / upper joint limits (`actor_params.hand.dof_properties.lower / upper`)
    int* ndropped;         // [N] contacts refused since init because all slots were taken (diagnostic; a manifold's 5th+ contact does not count)
    int pre_parts = 4;     // lanes per env of the pre kernel (option "pre_parts": 4 = hand_pre4_kernel, 1 = one lane per env, the A/B of round 4)
    int tips_in_post = 1;  // the post kernel's fingertip groups walk the fingertip chains themselves (option "tips_in_post"; 0: hand_tips_kernel in a launch of its own, the A/B of round 4)
    float* body_mass;      // [NB][N] per-env, per-BODY factors of the hand's link masses (`actor_params.hand.rigid_body_properties.mass`), or nullptr: the plain
                           //         kernels run (option "hand_body_mass": the Sim<Scaled<M>> instantiations read them in the tree pass)
    float* body_mass_arena;// where that tensor sits (always; `body_mass` is this pointer or null)
    int* npair;            // [NROLE][N] sides of the asset's hand-to-hand contact pairs that were pushed in the last sub-step, per role wave (one-wave form / CPU: column 0)
    int pair_sens = 1;     // 0: this launch is not the last sub-step of its call -- the pairs' forces on the fingertip sensors are skipped (its sensor values are overwritten unseen)
    float pair_k = 0.f;    // stiffness (N/m) of the compliant hand-to-hand pairs (option "hand_pair_stiffness"; 0 = off; core/hand_engine.hpp pair_side)
    int drive_clamp = 1;   // the position drives deliver at most M::dof_force_limit (option "drive_force_limit"; core/hand_engine.hpp drive_clamp_update)
};

}  // namespace mi
