// TEST INFRASTRUCTURE — stand-in for <ceres/rotation.h>: initial/initial_sfm.h names QuaternionRotatePoint inside a
// functor template that this build never instantiates (global SfM is out of scope); the definition is the textbook one.
#ifndef VINS_REF_STUB_CERES_ROTATION_H
#define VINS_REF_STUB_CERES_ROTATION_H
#include <cmath>
namespace ceres {
template <typename T> inline void UnitQuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
    const T t2 = q[0] * q[1], t3 = q[0] * q[2], t4 = q[0] * q[3], t5 = -q[1] * q[1], t6 = q[1] * q[2];
    const T t7 = q[1] * q[3], t8 = -q[2] * q[2], t9 = q[2] * q[3], t1 = -q[3] * q[3];
    result[0] = T(2) * ((t8 + t1) * pt[0] + (t6 - t4) * pt[1] + (t3 + t7) * pt[2]) + pt[0];
    result[1] = T(2) * ((t4 + t6) * pt[0] + (t5 + t1) * pt[1] + (t9 - t2) * pt[2]) + pt[1];
    result[2] = T(2) * ((t7 - t3) * pt[0] + (t2 + t9) * pt[1] + (t5 + t8) * pt[2]) + pt[2];
}
template <typename T> inline void QuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
    const T scale = T(1) / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const T unit[4] = {scale * q[0], scale * q[1], scale * q[2], scale * q[3]};
    UnitQuaternionRotatePoint(unit, pt, result);
}
}  // namespace ceres
#endif
