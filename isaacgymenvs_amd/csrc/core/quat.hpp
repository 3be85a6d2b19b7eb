// quat.hpp -- scalar restatement of the quaternion helpers the reference tasks use
// (reference isaacgymenvs/utils/torch_jit_utils.py; xyzw convention, :48).  Operation order follows the
// reference expression by expression so that fp32 results agree to the last bits; FP contraction is disabled
// in here for the same reason (torch CPU evaluates each op separately).
#pragma once
#include "engine.hpp"

namespace mi {
#if defined(__clang__)
#define MI_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define MI_NO_CONTRACT
#endif

// torch_jit_utils.py:42-63
MI_HD void quat_mul(const float* a, const float* b, float* o) {
    MI_NO_CONTRACT
    const float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3];
    const float x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    const float ww = (z1 + x1) * (x2 + y2);
    const float yy = (w1 - y1) * (w2 + z2);
    const float zz = (w1 + y1) * (w2 - z2);
    const float xx = ww + yy + zz;
    const float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    const float w = qq - ww + (z1 - y1) * (y2 - z2);
    const float x = qq - xx + (x1 + w1) * (x2 + w2);
    const float y = qq - yy + (w1 - x1) * (y2 + z2);
    const float z = qq - zz + (z1 + y1) * (w2 - x2);
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
// torch_jit_utils.py:80-104 ; sign=+1 quat_rotate, sign=-1 quat_rotate_inverse
MI_HD void quat_rotate_s(const float* q, const float* v, float sign, float* o) {
    MI_NO_CONTRACT
    const float qw = q[3];
    const float k = 2.0f * qw * qw - 1.0f;
    const float c3[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    const float d = (q[0] * v[0] + q[1] * v[1]) + q[2] * v[2];
    for (int i = 0; i < 3; ++i) {
        const float a = v[i] * k;
        const float b = c3[i] * qw * 2.0f;
        const float c = q[i] * d * 2.0f;
        o[i] = (sign > 0.f) ? (a + b + c) : (a - b + c);
    }
}
// ATen CPU remainder semantics (python-style modulo), used by get_euler_xyz's `% (2*np.pi)` (:195)
MI_HD float py_mod(float a, float b) {
    float m = fmodf(a, b);
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
    return m;
}
// torch_jit_utils.py:175-195 (pitch is never consumed by the five tasks)
MI_HD void euler_roll_yaw(const float* q, float* roll, float* yaw) {
    MI_NO_CONTRACT
    const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    const float sinr_cosp = 2.0f * (qw * qx + qy * qz);
    const float cosr_cosp = qw * qw - qx * qx - qy * qy + qz * qz;
    const float siny_cosp = 2.0f * (qw * qz + qx * qy);
    const float cosy_cosp = qw * qw + qx * qx - qy * qy - qz * qz;
    const float TWO_PI = 6.283185307179586f;
    *roll = py_mod(atan2f(sinr_cosp, cosr_cosp), TWO_PI);
    *yaw = py_mod(atan2f(siny_cosp, cosy_cosp), TWO_PI);
}
MI_HD float normalize_angle(float x) { return atan2f(sinf(x), cosf(x)); }  // :126-128

}  // namespace mi
