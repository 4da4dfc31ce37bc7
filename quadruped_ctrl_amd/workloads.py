"""Synthetic batched robot states for the five BASELINE.json configs.

Generators follow SURVEY.md section 8(d) literally (seed = 20260928 + config
index, numpy default_rng).  Output is a dict of numpy arrays in the layout
the C-ABI (include/qmpc.h) consumes: one row per robot, row-major.

    p[B,3] v[B,3] q[B,4](w,x,y,z) w[B,3] r[B,12](axis-major, r[axis*4+foot])
    yaw[B] weights[B,12] traj[B,12h] alpha[B] x_drag[B] gait[B,4h](u8)

Contact tables come from the OffsetDurationGait rule of the reference
(Gait.cpp:142-166), re-implemented in gait.mpc_table.
"""
import numpy as np

from .gait import mpc_table

SEED0 = 20260928
DT_MPC = 13 * 0.002            # GaitCtrller.cpp:6, config freq 500 Hz
MU = 0.4                       # ConvexMPCLocomotion.cpp:630
F_MAX = 120.0
ALPHA = 4e-5                   # ConvexMPCLocomotion.cpp:604
Q_WEIGHTS = np.array([2.5, 2.5, 10, 50, 50, 100, 0, 0, 0.5, 0.2, 0.2, 0.1],
                     np.float32)  # ConvexMPCLocomotion.cpp:598

# h=10 rescalings of the reference's 14-segment gaits
# (ConvexMPCLocomotion.cpp:27-29,40), SURVEY.md 8(d) config 3.
GAITS_H10 = {
    "trot": ((0, 5, 5, 0), (5, 5, 5, 5)),
    "bounding": ((5, 5, 0, 0), (4, 4, 4, 4)),
    "pacing": ((5, 0, 5, 0), (5, 5, 5, 5)),
    "standing": ((0, 0, 0, 0), (10, 10, 10, 10)),
}
GAITS_H16 = {"trot": ((0, 8, 8, 0), (8, 8, 8, 8))}


def _quat_from_rpy(rpy):
    """ZYX Euler -> (w,x,y,z); inverse of quat_to_rpy (SolverMPC.cpp:257-267)."""
    r, p, y = rpy[:, 0] / 2, rpy[:, 1] / 2, rpy[:, 2] / 2
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.stack([cr * cp * cy + sr * sp * sy,
                     sr * cp * cy - cr * sp * sy,
                     cr * sp * cy + sr * cp * sy,
                     cr * cp * sy - sr * sp * cy], 1)


def _states(rng, B, h, stairs=False):
    p = np.array([0, 0, 0.29]) + rng.normal(0, 0.01, (B, 3))
    v = np.array([0.5, 0, 0]) + rng.normal(0, 0.1, (B, 3))
    rpy = rng.normal(0, 0.05, (B, 3))
    w = rng.normal(0, 0.2, (B, 3))
    feet = np.array([[.19, .19, -.19, -.19],
                     [-.111, .111, -.111, .111],
                     [-.29, -.29, -.29, -.29]])
    r = feet[None] + rng.normal(0, 0.02, (B, 3, 4))
    if stairs:
        rpy[:, 0:2] = rng.uniform(-0.3, 0.3, (B, 2))
        v[:, 2] = rng.normal(0, 0.2, B)
        r[:, 2, :] += rng.choice([0, 0.02, 0.04, 0.06, 0.08], (B, 4))
    q = _quat_from_rpy(rpy)
    traj = np.zeros((B, h, 12))
    traj[:, :, 2] = rpy[:, 2:3]
    traj[:, :, 3] = p[:, 0:1] + v[:, 0:1] * DT_MPC * np.arange(h)[None]
    traj[:, :, 4] = p[:, 1:2]
    traj[:, :, 5] = 0.25
    traj[:, :, 9] = 0.5
    f32 = np.float32
    return dict(p=p.astype(f32), v=v.astype(f32), q=q.astype(f32),
                w=w.astype(f32), r=r.reshape(B, 12).astype(f32),
                yaw=rpy[:, 2].astype(f32),
                traj=traj.reshape(B, 12 * h).astype(f32))


def _finish(d, B, h, gait):
    d.update(batch=B, horizon=h, dt=DT_MPC, mu=MU, f_max=F_MAX,
             weights=np.tile(Q_WEIGHTS, (B, 1)),
             alpha=np.full(B, ALPHA, np.float32),
             x_drag=np.zeros(B, np.float32),
             gait=np.ascontiguousarray(gait, np.uint8))
    return d


def _gait_tables(rng, B, h, names, table):
    which = rng.integers(0, len(names), B)
    it = rng.integers(0, h, B)
    g = np.zeros((B, 4 * h), np.uint8)
    for i in range(B):
        off, dur = table[names[which[i]]]
        g[i] = mpc_table(h, off, dur, int(it[i]))
    return g


def make_config(idx, batch=None):
    """BASELINE.json configs[idx] (0..4); `batch` overrides the batch size."""
    rng = np.random.default_rng(SEED0 + idx)
    if idx in (0, 1):
        B, h = (batch or (1 if idx == 0 else 1024)), 10
        d = _states(rng, B, h)
        return _finish(d, B, h, _gait_tables(rng, B, h, ["trot"], GAITS_H10))
    if idx == 2:
        B, h = (batch or 4096), 10
        d = _states(rng, B, h)
        return _finish(d, B, h, _gait_tables(
            rng, B, h, ["trot", "bounding", "pacing"], GAITS_H10))
    if idx == 3:
        B, h = (batch or 16384), 16
        d = _states(rng, B, h)
        return _finish(d, B, h, _gait_tables(rng, B, h, ["trot"], GAITS_H16))
    if idx == 4:
        B, h = (batch or 65536), 10
        d = _states(rng, B, h, stairs=True)
        g = (rng.random((B, h, 4)) < 0.5).astype(np.uint8)
        none0 = g[:, 0, :].sum(1) == 0      # >= 1 stance foot at step 0
        g[none0, 0, rng.integers(0, 4, none0.sum())] = 1
        return _finish(d, B, h, g.reshape(B, 4 * h))
    raise ValueError(idx)


def make_standing(batch, horizon=10, seed=99, calm=False):
    """All four feet in stance for every step (n_r = 12h): the reference's
    'Standing' gait (ConvexMPCLocomotion.cpp:35) -- largest reduced QP.
    Default states are SURVEY 8d's (v_x ~ 0.5 m/s) against a zero-velocity reference, i.e. a
    robot BRAKING to a stand: the backward friction limit binds on most foot-steps (30+ active
    constraints, the solver's worst case).  calm=True: a robot that already stands
    (|v| ~ 2 cm/s, small attitude / rate errors), which is what the gait is used for."""
    rng = np.random.default_rng(SEED0 + seed)
    d = _states(rng, batch, horizon)
    if calm:
        f32 = np.float32
        d["v"] = rng.normal(0, 0.02, (batch, 3)).astype(f32)
        d["w"] = rng.normal(0, 0.05, (batch, 3)).astype(f32)
    d["traj"].reshape(batch, horizon, 12)[:, :, 9] = 0
    d["traj"].reshape(batch, horizon, 12)[:, :, 3] = d["p"][:, 0:1]
    return _finish(d, batch, horizon, np.ones((batch, 4 * horizon), np.uint8))


def make_trot(batch, horizon, seed=55):
    """Trot at an arbitrary even horizon (offsets 0, h/2, h/2, 0; durations h/2) -- the
    reference's own trot has 14 / 16 segments (ConvexMPCLocomotion.cpp:25,27)."""
    rng = np.random.default_rng(SEED0 + seed + horizon)
    d = _states(rng, batch, horizon)
    hh = horizon // 2
    table = {"trot": ((0, hh, hh, 0), (hh,) * 4)}
    return _finish(d, batch, horizon, _gait_tables(rng, batch, horizon, ["trot"], table))


def make_long_horizon(batch, horizon, gait="trot", seed=77):
    """Long horizons (17 .. 36 = K_MAX_GAIT_SEGMENTS, convexMPC_interface.h:3): `gait` = "trot" (offsets 0, h/2, h/2, 0,
    durations h/2: n_r = 6 h, the 192-row class up to horizon 32), "bound" (the reference's bounding rescaled:
    offsets h/2, h/2, 0, 0, durations 0.4 h: n_r ~ 4.7 h, <= 192 up to horizon 36) or "stand" (n_r = 12 h)."""
    rng = np.random.default_rng(SEED0 + seed + horizon)
    d = _states(rng, batch, horizon)
    hh = horizon // 2
    if gait == "trot":
        table = {"g": ((0, hh, hh, 0), (hh,) * 4)}
    elif gait == "stand":   # all four feet down for the whole horizon: n_r = 12 h (432 at horizon 36: the large-problem path)
        table = {"g": ((0, 0, 0, 0), (horizon,) * 4)}
    else:
        dur = (2 * horizon) // 5
        table = {"g": ((hh, hh, 0, 0), (dur,) * 4)}
    return _finish(d, batch, horizon, _gait_tables(rng, batch, horizon, ["g"], table))


def shard(d, rank, world):
    """Contiguous batch slice [rank*ceil(B/world), ...) -- SURVEY.md 8(e)."""
    B = d["batch"]
    per = -(-B // world)
    lo, hi = min(rank * per, B), min((rank + 1) * per, B)
    out = {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v)
           for k, v in d.items()}
    out["batch"] = hi - lo
    return out


def make_commands(batch, horizon=10, seed=7, stand_fraction=0.1, omni_mode=0, calm=False):
    """Synthetic controller commands + estimator states for the caller-side
    packer (include/qmpc.h qmpc_command; ConvexMPCLocomotion.cpp:498-640).
    Gaits: the h-segment rescalings of trot / bounding / pacing used by
    make_config; `stand_fraction` of the robots are in current_gait == 4.
    calm=True: small commands that the robot already tracks (few active
    constraints, like configs[1]); default: aggressive commands and up to 0.25 m
    of position error (many active constraints, exercises every clamp branch)."""
    rng = np.random.default_rng(seed)
    B, h = batch, horizon
    f32 = np.float32
    rpy = rng.normal(0, 0.05, (B, 3)).astype(f32)
    rpy[:, 2] = rng.uniform(-3.0, 3.0, B)
    quat = _quat_from_rpy(rpy.astype(np.float64)).astype(f32)
    cy, sy = np.cos(rpy[:, 2].astype(np.float64)), np.sin(rpy[:, 2].astype(np.float64))
    r_body = np.zeros((B, 3, 3))
    r_body[:, 0, 0], r_body[:, 0, 1] = cy, sy        # world -> body for a yaw-only attitude
    r_body[:, 1, 0], r_body[:, 1, 1] = -sy, cy
    r_body[:, 2, 2] = 1.0
    pos = (np.array([0, 0, 0.29]) + rng.normal(0, [0.5, 0.5, 0.01], (B, 3))).astype(f32)
    hip = np.array([[.19, -.111, -.29], [.19, .111, -.29], [-.19, -.111, -.29], [-.19, .111, -.29]])
    # feet under the hips of the yawed body (body -> world = rBody^T), 2 cm of scatter
    hip_w = np.einsum("bji,lj->bli", r_body, hip)
    p_foot = (pos[:, None, :].astype(np.float64) + hip_w + rng.normal(0, 0.02, (B, 4, 3))).astype(f32)
    amp = 0.3 if calm else 1.0
    vel_des = np.stack([rng.uniform(-1, 1, B), rng.uniform(-0.3, 0.3, B), rng.uniform(-1, 1, B)], 1) * amp
    v_des_w = np.einsum("bji,bj->bi", r_body, np.concatenate([vel_des[:, :2], np.zeros((B, 1))], 1))
    gaits = {"trot": ((0, h // 2, h // 2, 0), (h // 2,) * 4),
             "bound": ((h // 2, h // 2, 0, 0), (max(h // 2 - 1, 1),) * 4),
             "pace": ((h // 2, 0, h // 2, 0), (h // 2,) * 4),
             "stand": ((0, 0, 0, 0), (h,) * 4)}
    names = rng.choice(["trot", "bound", "pace"], B)
    stand = rng.random(B) < stand_fraction
    names = np.where(stand, "stand", names)
    cmd = {
        "batch": B, "horizon": h,
        "position": pos,
        "v_world": (v_des_w + rng.normal(0, 0.02 if calm else 0.1, (B, 3))).astype(f32),   # tracking the command
        "omega_world": (rng.normal(0, 0.2, (B, 3)) * (0.2 if calm else 1.0)).astype(f32),
        "orientation": quat, "rpy": rpy,
        "r_body": r_body.reshape(B, 9).astype(f32),
        "p_foot": p_foot.reshape(B, 12),
        "vel_des": vel_des.astype(f32),
        "yaw_des_true": (rpy[:, 2] + rng.normal(0, 0.05, B)).astype(f32),
        "rpy_comp": rng.normal(0, 0.02, (B, 2)).astype(f32),
        "stand_traj": np.concatenate([pos, rpy], 1).astype(f32),
        "rp_des": rng.normal(0, 0.01, (B, 2)).astype(f32),
        "gait_type": np.where(stand, 4, 0).astype(np.int32),
        "gait_offsets": np.array([gaits[n][0] for n in names], np.int32),
        "gait_durations": np.array([gaits[n][1] for n in names], np.int32),
        "gait_iteration": rng.integers(0, h, B).astype(np.int32),
        # desired position up to 0.25 m away from the estimate: exercises all four clamp branches
        "world_position_desired": (pos[:, :2] + rng.uniform(-0.25, 0.25, (B, 2)) * (0.05 if calm else 1.0)).astype(f32),
        "x_comp_integral": rng.normal(0, 0.01, B).astype(f32),
        "body_height": 0.29, "omni_mode": int(omni_mode),
    }
    return cmd


def make_leg_states(batch, seed=3):
    """Synthetic joint states and leg commands for the per-tick glue (include/qmpc.h
    qmpc_leg_kinematics / qmpc_leg_command; LegController.cpp:89-160): joint angles around the
    Mini Cheetah's stance posture, joint rates, Cartesian gains of the reference's magnitude
    (ConvexMPCLocomotion.cpp Kp ~ diag(700, 700, 150), Kd ~ diag(7, 7, 7)) with small off-diagonal
    terms so that every matrix entry is exercised, foot targets a few centimetres from the foot."""
    rng = np.random.default_rng(SEED0 + 1000 + seed)
    f32 = np.float32
    B = batch
    q = np.tile(np.array([0.0, -0.8, 1.6]), (B, 4)) + rng.normal(0, [0.15, 0.3, 0.3] * 4, (B, 12))
    qd = rng.normal(0, 2.0, (B, 12))
    kp = np.tile(np.diag([700.0, 700.0, 150.0]).reshape(9), (B, 4, 1)) + rng.normal(0, 5.0, (B, 4, 9))
    kd = np.tile(np.diag([7.0, 7.0, 7.0]).reshape(9), (B, 4, 1)) + rng.normal(0, 0.1, (B, 4, 9))
    return {"batch": B, "q": q.astype(f32), "qd": qd.astype(f32),
            "kp_cart": kp.astype(f32), "kd_cart": kd.astype(f32),
            "dp_des": rng.normal(0, 0.03, (B, 12)).astype(f32), "v_des": rng.normal(0, 0.5, (B, 12)).astype(f32),
            "tau_ff": rng.normal(0, 0.5, (B, 12)).astype(f32), "force_ff": rng.normal(0, 30.0, (B, 12)).astype(f32),
            "kp_joint": 3.0, "kd_joint": 0.5}


def make_swing_states(n, seed=4):
    """n swing feet: start / landing points ~0.2 m apart, apex 6-10 cm, phase in [0, 1] (including
    exactly 0, 0.5 and 1), swing times of 5-9 MPC steps (FootSwingTrajectory.cpp:17-37)."""
    rng = np.random.default_rng(SEED0 + 2000 + seed)
    f32 = np.float32
    p0 = rng.normal(0, 0.3, (n, 3))
    p0[:, 2] = rng.normal(0, 0.02, n)
    pf = p0 + rng.normal(0, 0.15, (n, 3))
    pf[:, 2] = rng.normal(0, 0.03, n)
    phase = rng.random(n)
    phase[:3] = [0.0, 0.5, 1.0][:min(3, n)]
    return {"p0": p0.astype(f32), "pf": pf.astype(f32), "height": rng.uniform(0.06, 0.1, n).astype(f32),
            "phase": phase.astype(f32), "swing_time": (0.026 * rng.integers(5, 10, n)).astype(f32)}


class Rollout:
    """Closed-loop sliding-window rollout of B robots for the warm-start measurements:
    every MPC cycle the contact table advances by one step (OffsetDurationGait iteration + 1,
    Gait.cpp:187-193), the single-rigid-body state is integrated over dtMPC with the first-step
    forces the solver returned (the model of SolverMPC.cpp:235-254, explicit Euler, fp64 on the host),
    stance feet stay where they are and feet that lift off are re-placed under their hips.  Host-side
    test / bench scaffolding: it only PRODUCES input records, all solving is the GPU's."""

    HIP = np.array([[.19, -.111], [.19, .111], [-.19, -.111], [-.19, .111]])

    def __init__(self, batch, horizon=10, gait="trot", seed=0, v_des=0.5, kick=1.0):
        rng = np.random.default_rng(SEED0 + 5000 + seed)
        self.rng = rng
        self.kick = kick       # per-cycle disturbance scale (pushes), so that constraints keep binding
        self.B, self.h = batch, horizon
        h = horizon
        table = {"trot": ((0, h // 2, h // 2, 0), (h // 2,) * 4),
                 "bound": ((h // 2, h // 2, 0, 0), (max(h // 2 - 1, 1),) * 4),
                 "pace": ((h // 2, 0, h // 2, 0), (h // 2,) * 4),
                 "stand": ((0, 0, 0, 0), (h,) * 4)}
        names = [gait] * batch if gait != "mixed" else list(rng.choice(["trot", "bound", "pace"], batch))
        self.off = np.array([table[n][0] for n in names])
        self.dur = np.array([table[n][1] for n in names])
        self.it = rng.integers(0, h, batch)
        self.p = np.array([0, 0, 0.29]) + rng.normal(0, 0.01, (batch, 3))
        self.v = np.zeros((batch, 3))
        self.v[:, 0] = (0.0 if gait == "stand" else v_des) + rng.normal(0, 0.05, batch)
        self.rpy = rng.normal(0, 0.03, (batch, 3))
        self.w = rng.normal(0, 0.1, (batch, 3))
        self.vdes = np.zeros((batch, 2))
        self.vdes[:, 0] = 0.0 if gait == "stand" else v_des
        self.v_cmd = v_des
        self.yaw_rate = rng.normal(0, 0.2, batch) * (0.0 if gait == "stand" else 1.0)
        self.feet = np.zeros((batch, 4, 3))
        self._place(np.ones((batch, 4), bool))
        self.feet[:, :, :2] += rng.normal(0, 0.01, (batch, 4, 2))

    def _place(self, mask):
        cy, sy = np.cos(self.rpy[:, 2]), np.sin(self.rpy[:, 2])
        hx = cy[:, None] * self.HIP[None, :, 0] - sy[:, None] * self.HIP[None, :, 1]
        hy = sy[:, None] * self.HIP[None, :, 0] + cy[:, None] * self.HIP[None, :, 1]
        tgt = np.stack([self.p[:, None, 0] + hx + 0.5 * 0.13 * self.v[:, None, 0],
                        self.p[:, None, 1] + hy + 0.5 * 0.13 * self.v[:, None, 1],
                        np.zeros_like(hx)], -1)
        self.feet[mask] = tgt[mask]

    def contact_table(self):
        g = np.zeros((self.B, 4 * self.h), np.uint8)
        for i in range(self.B):
            g[i] = mpc_table(self.h, tuple(self.off[i]), tuple(self.dur[i]), int(self.it[i]))
        return g

    def record(self):
        """The update_data_t record of this cycle (workloads layout)."""
        B, h = self.B, self.h
        f32 = np.float32
        q = _quat_from_rpy(self.rpy)
        traj = np.zeros((B, h, 12))
        k = np.arange(h)[None]
        traj[:, :, 2] = self.rpy[:, 2:3] + self.yaw_rate[:, None] * DT_MPC * k
        traj[:, :, 3] = self.p[:, 0:1] + self.vdes[:, 0:1] * DT_MPC * k
        traj[:, :, 4] = self.p[:, 1:2] + self.vdes[:, 1:2] * DT_MPC * k
        traj[:, :, 5] = 0.29
        traj[:, :, 8] = self.yaw_rate[:, None]
        traj[:, :, 9] = self.vdes[:, 0:1]
        traj[:, :, 10] = self.vdes[:, 1:2]
        r = (self.feet - self.p[:, None, :]).transpose(0, 2, 1).reshape(B, 12)      # axis-major
        d = dict(p=self.p.astype(f32), v=self.v.astype(f32), q=q.astype(f32), w=self.w.astype(f32),
                 r=r.astype(f32), yaw=self.rpy[:, 2].astype(f32), traj=traj.reshape(B, 12 * h).astype(f32))
        return _finish(d, B, h, self.contact_table())

    def demand(self, vx, vy=0.0, yaw_rate=None):
        """Sustained command change (the robot has to accelerate / turn for many cycles: friction
        limits bind persistently instead of by chance)."""
        self.vdes[:, 0], self.vdes[:, 1] = vx, vy
        if yaw_rate is not None:
            self.yaw_rate[:] = yaw_rate

    def advance(self, grf):
        """Integrate one MPC step with the first-step forces grf[B,12] (world frame, foot-major)."""
        f = np.asarray(grf, np.float64).reshape(self.B, 4, 3)
        m, ib = 9.0, np.array([.07, .26, .242])
        cy, sy = np.cos(self.rpy[:, 2]), np.sin(self.rpy[:, 2])
        R = np.zeros((self.B, 3, 3))
        R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = cy, -sy, sy, cy, 1.0
        Iinv = np.einsum("bij,j,bkj->bik", R, 1.0 / ib, R)
        tau = np.cross(self.feet - self.p[:, None, :], f).sum(1)
        acc = f.sum(1) / m + np.array([0, 0, -9.8])
        self.p = self.p + self.v * DT_MPC
        self.v = self.v + acc * DT_MPC
        self.rpy = self.rpy + np.einsum("bji,bj->bi", R, self.w) * DT_MPC
        self.w = self.w + np.einsum("bij,bj->bi", Iinv, tau) * DT_MPC
        if self.kick:          # pushes: the spread of SURVEY 8d's synthetic states, re-injected every cycle
            self.v += self.rng.normal(0, 0.06 * self.kick, (self.B, 3))
            self.w += self.rng.normal(0, 0.15 * self.kick, (self.B, 3))
        before = self.contact_table()[:, :4] != 0
        self.it = (self.it + 1) % self.h
        after = self.contact_table()[:, :4] != 0
        self._place(before & ~after)          # feet that just lifted off: re-placed under the hips


class ConfigRollout:
    """Closed-loop continuation of a BASELINE config (bench.py's `closed_loop` leg): cycle 0 IS the config's
    record, bit for bit; every later MPC cycle the contact table advances by one step (ConvexMPCLocomotion.cpp:498-590 solves
    once per 13 ticks with the table of `iteration + 1`, Gait.cpp:187-193 -- for a periodic gait of h segments that is the table
    rolled by one row; a random contact table of configs[4] is shifted and gets a fresh Bernoulli(0.5) last row), the
    single-rigid-body state is integrated over dtMPC with the first-step forces the solver returned (the model of
    SolverMPC.cpp:235-254, explicit Euler, fp64 on the host), SURVEY 8d's state noise is re-injected as pushes, stance feet
    stay where they are, feet that lift off are re-placed under their hips, and the reference trajectory is rebuilt from the
    new state by SURVEY 8d's own rule.  Host-side scaffolding: it only PRODUCES input records, all solving is the GPU's."""

    def __init__(self, b, seed=0, kick=1.0, periodic=True):
        self.b0 = b
        self.B, self.h = int(b["batch"]), int(b["horizon"])
        self.rng = np.random.default_rng(SEED0 + 7000 + seed)
        self.kick, self.periodic = kick, periodic
        self.cycle = 0
        f64 = np.float64
        self.p, self.v, self.w = b["p"].astype(f64), b["v"].astype(f64), b["w"].astype(f64)
        q = b["q"].astype(f64)
        qw, qx, qy, qz = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        self.rpy = np.stack([np.arctan2(2 * (qy * qz + qw * qx), qw * qw - qx * qx - qy * qy + qz * qz),
                             np.arcsin(np.clip(-2 * (qx * qz - qw * qy), -1, 1)),
                             np.arctan2(2 * (qx * qy + qw * qz), qw * qw + qx * qx - qy * qy - qz * qz)], 1)
        self.feet = self.p[:, None, :] + b["r"].astype(f64).reshape(self.B, 3, 4).transpose(0, 2, 1)
        self.hip = (self.feet - self.p[:, None, :]).copy()       # a lifted foot goes back to where it stood at cycle 0 (body frame, yaw-rotated)
        self.yaw0 = self.rpy[:, 2].copy()
        self.table = b["gait"].reshape(self.B, self.h, 4).copy()
        self.vx_des = b["traj"].reshape(self.B, self.h, 12)[:, 0, 9].astype(f64)
        self.z_des = b["traj"].reshape(self.B, self.h, 12)[:, 0, 5].astype(f64)

    def record(self):
        if self.cycle == 0:
            return self.b0
        B, h = self.B, self.h
        f32 = np.float32
        traj = np.zeros((B, h, 12))
        k = np.arange(h)[None]
        traj[:, :, 2] = self.rpy[:, 2:3]                                     # [0, 0, yaw, p_x + v_x dt i, p_y, z_des, 0, 0, 0, vx_des, 0, 0]
        traj[:, :, 3] = self.p[:, 0:1] + self.v[:, 0:1] * DT_MPC * k
        traj[:, :, 4] = self.p[:, 1:2]
        traj[:, :, 5] = self.z_des[:, None]
        traj[:, :, 9] = self.vx_des[:, None]
        r = (self.feet - self.p[:, None, :]).transpose(0, 2, 1).reshape(B, 12)
        d = dict(p=self.p.astype(f32), v=self.v.astype(f32), q=_quat_from_rpy(self.rpy).astype(f32), w=self.w.astype(f32),
                 r=r.astype(f32), yaw=self.rpy[:, 2].astype(f32), traj=traj.reshape(B, 12 * h).astype(f32))
        out = _finish(d, B, h, self.table.reshape(B, 4 * h))
        for key in ("dt", "mu", "f_max", "weights", "alpha", "x_drag"):
            out[key] = self.b0[key]
        return out

    def advance(self, grf):
        f = np.asarray(grf, np.float64).reshape(self.B, 4, 3)
        m, ib = 9.0, np.array([.07, .26, .242])
        cy, sy = np.cos(self.rpy[:, 2]), np.sin(self.rpy[:, 2])
        R = np.zeros((self.B, 3, 3))
        R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = cy, -sy, sy, cy, 1.0
        Iinv = np.einsum("bij,j,bkj->bik", R, 1.0 / ib, R)
        tau = np.cross(self.feet - self.p[:, None, :], f).sum(1)
        acc = f.sum(1) / m + np.array([0, 0, -9.8])
        self.p = self.p + self.v * DT_MPC
        self.v = self.v + acc * DT_MPC
        self.rpy = self.rpy + np.einsum("bji,bj->bi", R, self.w) * DT_MPC
        self.w = self.w + np.einsum("bij,bj->bi", Iinv, tau) * DT_MPC
        if self.kick:
            self.v += self.rng.normal(0, 0.06 * self.kick, (self.B, 3))
            self.w += self.rng.normal(0, 0.15 * self.kick, (self.B, 3))
        before = self.table[:, 0, :] != 0
        if self.periodic:
            self.table = np.roll(self.table, -1, axis=1)
        else:
            new = (self.rng.random((self.B, 4)) < 0.5).astype(np.uint8)
            self.table = np.concatenate([self.table[:, 1:], new[:, None, :]], 1)
            none0 = self.table[:, 0, :].sum(1) == 0
            self.table[none0, 0, self.rng.integers(0, 4, int(none0.sum()))] = 1
        after = self.table[:, 0, :] != 0
        lift = before & ~after
        dy = self.rpy[:, 2] - self.yaw0
        c, s_ = np.cos(dy)[:, None], np.sin(dy)[:, None]
        tgt = self.p[:, None, :] + np.stack([c * self.hip[..., 0] - s_ * self.hip[..., 1], s_ * self.hip[..., 0] + c * self.hip[..., 1],
                                             self.hip[..., 2]], -1)
        tgt[..., 2] = self.feet[..., 2]                     # the ground (stairs: the step the foot stood on) does not move with the body
        tgt[..., 0] += 0.5 * 0.13 * self.v[:, None, 0]
        tgt[..., 1] += 0.5 * 0.13 * self.v[:, None, 1]
        self.feet[lift] = tgt[lift]
        self.cycle += 1


def make_kf_stream(batch, steps, seed=5):
    """Synthetic estimator inputs for the Kalman filter (include/qmpc.h qmpc_kf_state): robots moving at a
    constant world velocity with a yawed body, feet planted (so the leg kinematics see the body move
    over them), gait phases sweeping 0..1 per leg, accelerometer = gravity + noise.  Yields one dict per
    control tick (dt = 2 ms) plus the true position / velocity."""
    rng = np.random.default_rng(SEED0 + 3000 + seed)
    f32 = np.float32
    B = batch
    yaw = rng.uniform(-1, 1, B)
    cy, sy = np.cos(yaw), np.sin(yaw)
    R = np.zeros((B, 3, 3))                     # rBody: world -> body
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = cy, sy, -sy, cy, 1.0
    vel = np.concatenate([rng.normal(0, 0.3, (B, 2)), np.zeros((B, 1))], 1)
    pos0 = np.array([0, 0, 0.29]) + rng.normal(0, 0.01, (B, 3))
    hips = np.array([[.19, -.049, 0], [.19, .049, 0], [-.19, -.049, 0], [-.19, .049, 0]])
    feet_w = pos0[:, None, :] + np.einsum("bji,lj->bli", R, hips + np.array([0, 0, -0.29]))  # under the hips, on the ground
    feet_w[:, :, 2] = 0.0
    phase0 = rng.random((B, 4))
    out = []
    for t in range(steps):
        pos = pos0 + vel * 0.002 * t
        rel_w = feet_w - pos[:, None, :]                              # foot relative to the body, world frame
        p_body = np.einsum("bij,blj->bli", R, rel_w) - hips[None]     # hip frame (legControllerData.p)
        v_body = np.einsum("bij,bj->bi", R, -vel)[:, None, :].repeat(4, 1)
        out.append({
            "r_body": R.reshape(B, 9).astype(f32),
            "a_world": (np.array([0, 0, 9.81]) + rng.normal(0, 0.05, (B, 3))).astype(f32),
            "omega_body": rng.normal(0, 0.01, (B, 3)).astype(f32),
            "contact_phase": ((phase0 + 0.004 * t) % 1.0).astype(f32),
            "leg_p": (p_body + rng.normal(0, 0.001, (B, 4, 3))).reshape(B, 12).astype(f32),
            "leg_v": (v_body + rng.normal(0, 0.02, (B, 4, 3))).reshape(B, 12).astype(f32),
            "true_position": pos, "true_velocity": vel})
    return out
