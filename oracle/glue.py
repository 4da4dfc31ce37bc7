"""ctypes loader for oracle/libglue_oracle.so -- TEST INFRASTRUCTURE ONLY.

Batched numpy front end of the float restatement (glue_oracle.c) of the reference's per-tick
glue: leg kinematics / Jacobian, leg command (Cartesian PD + J^T f + IK), swing-foot Bezier.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
GEOM = np.array([0.062, 0.209, 0.195, 0.004], np.float32)   # Dynamics/MiniCheetah.h:31-37
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libglue_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-C", _HERE, "-j8"], check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(path)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def leg_update(q, qd, geom=GEOM):
    """q, qd [B,12] -> J [B,4,9], p [B,12], v [B,12]  (LegController::updateData)."""
    q, qd = _f(q), _f(qd)
    B = q.shape[0]
    J = np.zeros((B, 4, 9), np.float32)
    p = np.zeros((B, 12), np.float32)
    v = np.zeros((B, 12), np.float32)
    g = _f(geom)
    lib().oracle_leg_update_batch(_p(g), C.c_int(B), _p(q), _p(qd), _p(J), _p(p), _p(v))
    return J, p, v


def leg_command(c, geom=GEOM):
    """dict with tau_ff, force_ff, kp_cart, kd_cart, p_des, v_des, q, qd, J, p, v, kp_joint, kd_joint
    -> tau [B,12], q_des [B,12]  (LegController::updateCommand)."""
    B = np.asarray(c["q"]).shape[0]
    a = {k: _f(c[k]) for k in ("tau_ff", "force_ff", "kp_cart", "kd_cart", "p_des", "v_des", "q", "qd", "J", "p", "v")}
    tau = np.zeros((B, 12), np.float32)
    qdes = np.zeros((B, 12), np.float32)
    g = _f(geom)
    lib().oracle_leg_command_batch(_p(g), C.c_int(B), _p(a["tau_ff"]), _p(a["force_ff"]), _p(a["kp_cart"]), _p(a["kd_cart"]),
                                   _p(a["p_des"]), _p(a["v_des"]), _p(a["q"]), _p(a["qd"]), _p(a["J"]), _p(a["p"]), _p(a["v"]),
                                   C.c_float(c["kp_joint"]), C.c_float(c["kd_joint"]), _p(tau), _p(qdes))
    return tau, qdes


def swing(p0, pf, height, phase, swing_time):
    """[n,3], [n,3], [n], [n], [n] -> p, v, a [n,3]  (computeSwingTrajectoryBezier)."""
    p0, pf, height, phase, swing_time = _f(p0), _f(pf), _f(height), _f(phase), _f(swing_time)
    n = p0.shape[0]
    p, v, a = (np.zeros((n, 3), np.float32) for _ in range(3))
    lib().oracle_swing_batch(C.c_int(n), _p(p0), _p(pf), _p(height), _p(phase), _p(swing_time), _p(p), _p(v), _p(a))
    return p, v, a


HIP = np.array([0.19, 0.049, 0.0], np.float32)   # _abadLocation, Dynamics/MiniCheetah.h:25-26,105


def kf_init(batch):
    """LinearKFPositionVelocityEstimator::setup: xhat = 0, P = 100 I."""
    return np.zeros((batch, 18), np.float32), np.tile((100 * np.eye(18, dtype=np.float32)).reshape(1, 324), (batch, 1))


def kf_step(xhat, P, r_body, a_world, omega_body, contact, leg_p, leg_v):
    """One run() for every robot; xhat, P updated IN PLACE (float32 contiguous).  -> position, v_world, v_body."""
    B = xhat.shape[0]
    assert xhat.dtype == np.float32 and P.dtype == np.float32 and xhat.flags.c_contiguous and P.flags.c_contiguous
    a = [_f(x) for x in (r_body, a_world, omega_body, contact, leg_p, leg_v)]
    pos, vw, vb = (np.zeros((B, 3), np.float32) for _ in range(3))
    lib().oracle_kf_step_batch(_p(HIP), C.c_int(B), _p(xhat), _p(P), *[_p(x) for x in a], _p(pos), _p(vw), _p(vb))
    return pos, vw, vb
