"""ctypes binding of the C ABI in include/qmpc.h (libqmpc.so).

PyTorch is used only for device memory and streams; every solve goes through
the hand-written HIP kernels in csrc/.  There is no CPU or PyTorch fallback:
if libqmpc.so is missing or no HIP device is present this module raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libqmpc.so")

QMPC_OK = 0
ABI_VERSION = 21              # qmpc_abi_version() this binding was written against
ST_MAXITER, ST_NOT_PD, ST_INFEASIBLE, ST_WS_FULL, ST_FALLBACK = 1, 2, 4, 8, 16
ST_COMPACTED, ST_SPILLED = 64, 128
ST_NONFINITE = 32
ST_ERROR_MASK = 15 | 32
EXPORTS = ["qmpc_abi_version", "qmpc_last_error", "qmpc_create", "qmpc_destroy",
           "qmpc_setup", "qmpc_set_robot", "qmpc_settings", "qmpc_solve",
           "qmpc_solve_host", "qmpc_set_debug", "qmpc_debug_ld",
           "qmpc_set_debug_clock", "qmpc_set_max_stance", "qmpc_pack",
           "qmpc_forces_to_body", "qmpc_solve_commands", "qmpc_set_min_stance",
           "qmpc_set_debug_aux", "qmpc_set_debug_overflow_slices", "qmpc_solve_sharded", "qmpc_set_leg_geometry",
           "qmpc_leg_kinematics", "qmpc_leg_torques", "qmpc_swing_trajectory", "qmpc_set_warm_start", "qmpc_settings_jcqp", "qmpc_kf_init", "qmpc_kf_step", "qmpc_set_model",
           "qmpc_max_horizon", "qmpc_set_debug_pool_busy", "qmpc_set_split", "qmpc_reserve", "qmpc_set_debug_engine_events", "qmpc_set_chunks", "qmpc_set_block_start", "qmpc_debug_read_item", "qmpc_debug_read_counts", "qmpc_set_dense", "qmpc_set_size_order", "qmpc_debug_keys", "qmpc_set_order_hint", "qmpc_set_debug_balance",
           "qmpc_set_warm_start_min_iters", "qmpc_set_debug_overflow_spin"]

KF_FIELDS = ("xhat", "P", "r_body", "a_world", "omega_body", "contact_phase", "leg_p", "leg_v", "position", "v_world", "v_body")

# qmpc_leg_command fields (include/qmpc.h), in declaration order
LEG_F32 = ("tau_ff", "force_ff", "kp_cart", "kd_cart", "p_des", "v_des", "q", "qd", "J", "p", "v")

# qmpc_command fields (include/qmpc.h), in declaration order
CMD_F32 = ("position", "v_world", "omega_world", "orientation", "rpy", "r_body", "p_foot",
           "vel_des", "yaw_des_true", "rpy_comp", "stand_traj", "rp_des")
CMD_I32 = ("gait_type", "gait_offsets", "gait_durations", "gait_iteration")
CMD_STATE = ("world_position_desired", "x_comp_integral")
REC_FIELDS = ("p", "v", "q", "w", "r", "yaw", "traj", "gait", "x_drag", "weights", "alpha")


class Inputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("p", "v", "q", "w", "r", "yaw", "traj", "gait", "weights",
                 "alpha", "x_drag")] + [("weights_stride", C.c_int),
                                        ("alpha_stride", C.c_int),
                                        ("x_drag_stride", C.c_int)]


class Outputs(C.Structure):
    _fields_ = [("grf", C.c_void_p), ("soln", C.c_void_p),
                ("status", C.c_void_p), ("iters", C.c_void_p)]


class Command(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in CMD_F32 + CMD_I32 + CMD_STATE] + [
        ("body_height", C.c_float), ("omni_mode", C.c_int)]


class Record(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in REC_FIELDS]


class KfState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("xhat", "P", "r_body", "a_world", "omega_body", "contact_phase", "leg_p", "leg_v",
                                          "position", "v_world", "v_body")]


class LegCommand(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in LEG_F32] + [("kp_joint", C.c_float), ("kd_joint", C.c_float)]


_lib = None


def load_library():
    """Load libqmpc.so (built in-tree by __graft_entry__.build()).  Loud
    failure when absent -- the product has no other compute path."""
    global _lib
    if _lib is None:
        # PyTorch ships its own copy of the HIP runtime: it has to be the one this process loads FIRST.  Loaded after
        # libqmpc.so (which would pull /opt/rocm's), the process ends up with two runtimes and the kernels registered with
        # the wrong one -- qmpc_create then fails with QMPC_ERR_DEVICE (build() followed by smoke() in one process did)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        # Development only: QMPC_LIB names a VARIANT BUILD of this library made by tools/build_variant.sh.  It is honoured only
        # for a file inside this checkout's own (git-ignored) variants/ directory -- the environment cannot point a production
        # load at an arbitrary shared object -- and anything else is refused loudly
        LIB_PATH = globals()["LIB_PATH"]
        override = os.environ.get("QMPC_LIB")
        if override:
            vdir = os.path.realpath(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "variants"))
            real = os.path.realpath(override)
            if os.path.commonpath([real, vdir]) != vdir or os.path.basename(real) != "libqmpc.so":
                raise RuntimeError(f"QMPC_LIB={override!r} refused: only <checkout>/variants/<name>/libqmpc.so "
                                   "(tools/build_variant.sh) may replace the in-tree library")
            LIB_PATH = real
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()')")
        lib = C.CDLL(LIB_PATH)
        lib.qmpc_last_error.restype = C.c_char_p
        lib.qmpc_last_error.argtypes = [C.c_void_p]
        lib.qmpc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.qmpc_destroy.argtypes = [C.c_void_p]
        lib.qmpc_setup.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double]
        lib.qmpc_set_robot.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_double), C.c_double]
        lib.qmpc_settings.argtypes = [C.c_void_p, C.c_int, C.c_double]
        lib.qmpc_solve.argtypes = [C.c_void_p, C.c_int, C.POINTER(Inputs),
                                   C.POINTER(Outputs), C.c_void_p]
        lib.qmpc_solve_host.argtypes = [C.c_void_p, C.c_int, C.POINTER(Inputs),
                                        C.POINTER(Outputs)]
        lib.qmpc_set_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.qmpc_debug_ld.argtypes = [C.c_void_p]
        lib.qmpc_set_debug_aux.argtypes = [C.c_void_p, C.c_void_p]
        lib.qmpc_set_debug_overflow_slices.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_set_debug_overflow_spin.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_set_debug_pool_busy.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_debug_read_item.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.qmpc_set_split.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_reserve.argtypes = [C.c_void_p]
        lib.qmpc_set_debug_engine_events.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_set_chunks.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_set_dense.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_set_size_order.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_debug_keys.argtypes = [C.c_void_p, C.c_int, C.POINTER(Inputs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.qmpc_set_order_hint.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_set_debug_balance.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_set_block_start.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_set_warm_start.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.qmpc_set_warm_start_min_iters.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_set_model.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_kf_init.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.qmpc_kf_step.argtypes = [C.c_void_p, C.c_int, C.POINTER(KfState), C.c_void_p]
        lib.qmpc_settings_jcqp.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_double] * 4
        lib.qmpc_set_leg_geometry.argtypes = [C.c_void_p] + [C.c_double] * 4
        lib.qmpc_leg_kinematics.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        lib.qmpc_leg_torques.argtypes = [C.c_void_p, C.c_int, C.POINTER(LegCommand), C.c_void_p, C.c_void_p, C.c_void_p]
        lib.qmpc_swing_trajectory.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 9
        lib.qmpc_solve_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(Inputs),
                                           C.POINTER(Outputs)]
        lib.qmpc_set_debug_clock.argtypes = [C.c_void_p, C.c_void_p]
        lib.qmpc_debug_read_counts.argtypes = [C.c_void_p, C.c_void_p]
        lib.qmpc_set_max_stance.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_set_min_stance.argtypes = [C.c_void_p, C.c_int]
        lib.qmpc_pack.argtypes = [C.c_void_p, C.c_int, C.POINTER(Command), C.POINTER(Record), C.c_void_p]
        lib.qmpc_solve_commands.argtypes = [C.c_void_p, C.c_int, C.POINTER(Command), C.POINTER(Outputs), C.c_void_p,
                                            C.c_void_p]
        lib.qmpc_forces_to_body.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


class QmpcError(RuntimeError):
    pass


class BatchedConvexMPC:
    """Batched MPC solver on one GPU.

    Host-side mirror of the reference's MPC interface
    (src/MPC_Ctrl/convexMPC_interface.h:40-48) for B robots at once:
    setup_problem -> setup(), update_problem_data_floats -> solve(),
    get_solution(0..11) -> the returned grf[B,12].
    """

    def __init__(self, device=0, max_batch=65536, max_horizon=16):
        import torch
        if not torch.cuda.is_available():
            raise QmpcError("no HIP device visible: quadruped_ctrl_amd has no CPU path")
        self.torch = torch
        self.lib = load_library()
        self.device = torch.device("cuda", device)
        self.h = C.c_void_p()
        rc = self.lib.qmpc_create(device, max_batch, max_horizon, C.byref(self.h))
        if rc != QMPC_OK:
            raise QmpcError(f"qmpc_create failed rc={rc}")
        self.horizon = None
        self._dbg = None

    def close(self):
        if self.h:
            self.lib.qmpc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != QMPC_OK:
            err = self.lib.qmpc_last_error(self.h).decode()
            raise QmpcError(f"{what} failed rc={rc} {err}")

    def setup(self, dt, horizon, mu, f_max):
        self._check(self.lib.qmpc_setup(self.h, dt, horizon, mu, f_max), "qmpc_setup")
        self.horizon = horizon

    def set_robot(self, mass, ibody, gravity):
        arr = (C.c_double * 3)(*ibody)
        self._check(self.lib.qmpc_set_robot(self.h, mass, arr, gravity), "qmpc_set_robot")

    def set_model(self, model):
        """0 = the dense path's zero-order-hold model, 1 = SparseCMPC's (QMPC_MODEL_SPARSE)."""
        self._check(self.lib.qmpc_set_model(self.h, int(model)), "qmpc_set_model")

    def settings_jcqp(self, use_jcqp, max_iter=10000, rho=1e-7, sigma=1e-8, alpha=1.5, terminate=0.1):
        """The reference's JCQP/ADMM alternate (update_solver_settings' use_jcqp = 1 / 2); 0 = exact solve.
        Defaults are the caller's settings (ConvexMPCLocomotion.cpp:644-648)."""
        self._check(self.lib.qmpc_settings_jcqp(self.h, int(use_jcqp), int(max_iter), rho, sigma, alpha, terminate),
                    "qmpc_settings_jcqp")

    def warm_start(self, batch=None, shift_steps=1):
        """Enable warm starting across MPC cycles; returns the [batch, 64] int32 working-set tensor
        (all -1 = cold).  warm_start(None) switches it off."""
        if batch is None:
            self._check(self.lib.qmpc_set_warm_start(self.h, None, 1), "qmpc_set_warm_start")
            self._ws = None
            return None
        t = self.torch
        ws = t.full((batch, 64), -1, dtype=t.int32, device=self.device)
        self._check(self.lib.qmpc_set_warm_start(self.h, ws.data_ptr(), int(shift_steps)), "qmpc_set_warm_start")
        self._ws = ws
        return ws

    def warm_start_min_iters(self, n):
        """Selective warm start: only robots with at least n iterations in the previous call start warm (0: all)."""
        self._check(self.lib.qmpc_set_warm_start_min_iters(self.h, int(n)), "qmpc_set_warm_start_min_iters")

    def set_max_stance(self, max_stance_footsteps):
        """Caller's bound on stance foot-steps per robot (0 = unknown)."""
        self._check(self.lib.qmpc_set_max_stance(self.h, int(max_stance_footsteps)), "qmpc_set_max_stance")

    def set_min_stance(self, min_stance_footsteps):
        """Caller's lower bound on stance foot-steps per robot (0 = unknown)."""
        self._check(self.lib.qmpc_set_min_stance(self.h, int(min_stance_footsteps)), "qmpc_set_min_stance")

    def settings(self, max_iter=1000, tol=1e-9):
        self._check(self.lib.qmpc_settings(self.h, max_iter, tol), "qmpc_settings")

    # ---- device-resident path -------------------------------------------
    def upload(self, b):
        """numpy batch dict (workloads layout) -> dict of device tensors."""
        t = self.torch
        d = {}
        for k in ("p", "v", "q", "w", "r", "yaw", "traj", "weights", "alpha", "x_drag"):
            d[k] = t.from_numpy(np.ascontiguousarray(b[k], np.float32)).to(self.device)
        d["gait"] = t.from_numpy(np.ascontiguousarray(b["gait"], np.uint8)).to(self.device)
        d["batch"] = int(b["batch"])
        return d

    def alloc_outputs(self, batch, full=False, iters=True):
        t = self.torch
        o = {"grf": t.empty((batch, 12), dtype=t.float32, device=self.device),
             "status": t.empty((batch,), dtype=t.int32, device=self.device)}
        o["soln"] = (t.empty((batch, 12 * self.horizon), dtype=t.float64, device=self.device)
                     if full else None)
        o["iters"] = t.empty((batch,), dtype=t.int32, device=self.device) if iters else None
        return o

    def make_args(self, d, o):
        """Pack ctypes argument structs once (for launch-only timing loops)."""
        inp = Inputs()
        for k in ("p", "v", "q", "w", "r", "yaw", "traj", "gait", "weights", "alpha", "x_drag"):
            setattr(inp, k, d[k].data_ptr())
        B = d["batch"]
        inp.weights_stride = 12 if d["weights"].dim() == 2 else 0
        inp.alpha_stride = 1 if d["alpha"].numel() == B else 0
        inp.x_drag_stride = 1 if d["x_drag"].numel() == B else 0
        out = Outputs(o["grf"].data_ptr(),
                      o["soln"].data_ptr() if o["soln"] is not None else None,
                      o["status"].data_ptr(),
                      o["iters"].data_ptr() if o["iters"] is not None else None)
        return inp, out

    def solve_async(self, batch, inp, out, stream=None):
        """Enqueue one batched solve on `stream` (torch current stream by default)."""
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        rc = self.lib.qmpc_solve(self.h, batch, C.byref(inp), C.byref(out),
                                 C.c_void_p(s.cuda_stream))
        self._check(rc, "qmpc_solve")

    def solve(self, b, full=False):
        """Convenience: numpy batch dict in, numpy results out (device path)."""
        d = self.upload(b)
        o = self.alloc_outputs(d["batch"], full=full)
        inp, out = self.make_args(d, o)
        self.solve_async(d["batch"], inp, out)
        self.torch.cuda.synchronize(self.device)
        res = {"grf": o["grf"].cpu().numpy(), "status": o["status"].cpu().numpy(),
               "iters": o["iters"].cpu().numpy()}
        if full:
            res["soln"] = o["soln"].cpu().numpy()
        return res

    # ---- caller side on the GPU (ConvexMPCLocomotion.cpp:498-640, :672-680) --
    def upload_command(self, cmd):
        """numpy command dict (workloads.make_commands layout) -> device tensors."""
        t = self.torch
        d = {}
        for k in CMD_F32 + CMD_STATE:
            d[k] = None if cmd.get(k) is None else t.from_numpy(np.ascontiguousarray(cmd[k], np.float32)).to(self.device)
        for k in CMD_I32:
            d[k] = None if cmd.get(k) is None else t.from_numpy(np.ascontiguousarray(cmd[k], np.int32)).to(self.device)
        d["body_height"] = float(cmd["body_height"])
        d["omni_mode"] = int(cmd["omni_mode"])
        d["batch"] = int(cmd["batch"])
        return d

    def alloc_record(self, batch):
        """Device arrays of the update_data_t record for `batch` robots."""
        t, h = self.torch, self.horizon
        f = lambda *shape: t.empty(shape, dtype=t.float32, device=self.device)
        return {"p": f(batch, 3), "v": f(batch, 3), "q": f(batch, 4), "w": f(batch, 3), "r": f(batch, 12),
                "yaw": f(batch), "traj": f(batch, 12 * h),
                "gait": t.empty((batch, 4 * h), dtype=t.uint8, device=self.device),
                "x_drag": f(batch), "weights": f(batch, 12), "alpha": f(batch), "batch": batch}

    def pack_async(self, dcmd, rec, stream=None):
        """Enqueue the record build (qmpc_pack) for the uploaded command."""
        cs = Command()
        for k in CMD_F32 + CMD_I32 + CMD_STATE:
            setattr(cs, k, None if dcmd[k] is None else dcmd[k].data_ptr())
        cs.body_height = dcmd["body_height"]
        cs.omni_mode = dcmd["omni_mode"]
        rs = Record()
        for k in REC_FIELDS:
            setattr(rs, k, rec[k].data_ptr())
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        self._check(self.lib.qmpc_pack(self.h, dcmd["batch"], C.byref(cs), C.byref(rs),
                                       C.c_void_p(s.cuda_stream)), "qmpc_pack")

    def make_command_args(self, dcmd):
        cs = Command()
        for k in CMD_F32 + CMD_I32 + CMD_STATE:
            setattr(cs, k, None if dcmd[k] is None else dcmd[k].data_ptr())
        cs.body_height = dcmd["body_height"]
        cs.omni_mode = dcmd["omni_mode"]
        return cs

    def solve_commands_async(self, batch, cs, out, f_ff=None, stream=None):
        """One fused launch: command -> (record in registers) -> solve -> grf (+ body-frame forces)."""
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        self._check(self.lib.qmpc_solve_commands(self.h, batch, C.byref(cs), C.byref(out),
                                                 None if f_ff is None else f_ff.data_ptr(),
                                                 C.c_void_p(s.cuda_stream)), "qmpc_solve_commands")

    def forces_to_body_async(self, batch, r_body, grf, f_ff, stream=None):
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        self._check(self.lib.qmpc_forces_to_body(self.h, batch, r_body.data_ptr(), grf.data_ptr(), f_ff.data_ptr(),
                                                 C.c_void_p(s.cuda_stream)), "qmpc_forces_to_body")

    # ---- host-pointer path (what the single-robot shim uses) -------------
    def solve_host(self, b, full=False):
        B, h = int(b["batch"]), self.horizon
        keep = {k: np.ascontiguousarray(b[k], np.float32) for k in
                ("p", "v", "q", "w", "r", "yaw", "traj", "weights", "alpha", "x_drag")}
        keep["gait"] = np.ascontiguousarray(b["gait"], np.uint8)
        inp = Inputs()
        for k, a in keep.items():
            setattr(inp, k, a.ctypes.data)
        inp.weights_stride = 12 if keep["weights"].size == 12 * B else 0
        inp.alpha_stride = 1 if keep["alpha"].size == B else 0
        inp.x_drag_stride = 1 if keep["x_drag"].size == B else 0
        grf = np.zeros((B, 12), np.float32)
        st = np.zeros(B, np.int32)
        it = np.zeros(B, np.int32)
        soln = np.zeros((B, 12 * h)) if full else None
        out = Outputs(grf.ctypes.data, soln.ctypes.data if full else None,
                      st.ctypes.data, it.ctypes.data)
        self._check(self.lib.qmpc_solve_host(self.h, B, C.byref(inp), C.byref(out)),
                    "qmpc_solve_host")
        res = {"grf": grf, "status": st, "iters": it}
        if full:
            res["soln"] = soln
        return res

    # ---- per-tick glue either side of the solve ---------------------------
    def _dev32(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)

    def _stream_ptr(self, stream):
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        return C.c_void_p(s.cuda_stream)

    def leg_kinematics(self, q, qd, stream=None):
        """LegController::updateData on device tensors q, qd [B,12] -> J [B,4,9], p [B,12], v [B,12]."""
        t = self.torch
        B = q.shape[0]
        J = t.empty((B, 4, 9), dtype=t.float32, device=self.device)
        p = t.empty((B, 12), dtype=t.float32, device=self.device)
        v = t.empty((B, 12), dtype=t.float32, device=self.device)
        self._check(self.lib.qmpc_leg_kinematics(self.h, B, q.data_ptr(), qd.data_ptr(), J.data_ptr(), p.data_ptr(),
                                                 v.data_ptr(), self._stream_ptr(stream)), "qmpc_leg_kinematics")
        return J, p, v

    def leg_torques(self, c, stream=None):
        """LegController::updateCommand: dict of device tensors (LEG_F32 keys; tau_ff / force_ff may be
        None) + kp_joint, kd_joint -> tau [B,12], q_des [B,12]."""
        t = self.torch
        B = c["q"].shape[0]
        lc = LegCommand()
        for k in LEG_F32:
            setattr(lc, k, None if c.get(k) is None else c[k].data_ptr())
        lc.kp_joint, lc.kd_joint = float(c["kp_joint"]), float(c["kd_joint"])
        tau = t.empty((B, 12), dtype=t.float32, device=self.device)
        qdes = t.empty((B, 12), dtype=t.float32, device=self.device)
        self._check(self.lib.qmpc_leg_torques(self.h, B, C.byref(lc), tau.data_ptr(), qdes.data_ptr(),
                                              self._stream_ptr(stream)), "qmpc_leg_torques")
        return tau, qdes

    def kf_init(self, batch, stream=None):
        """LinearKFPositionVelocityEstimator::setup -> (xhat [B,18], P [B,324]) device tensors."""
        t = self.torch
        xhat = t.empty((batch, 18), dtype=t.float32, device=self.device)
        P = t.empty((batch, 324), dtype=t.float32, device=self.device)
        self._check(self.lib.qmpc_kf_init(self.h, batch, xhat.data_ptr(), P.data_ptr(), self._stream_ptr(stream)), "qmpc_kf_init")
        return xhat, P

    def kf_step(self, xhat, P, r_body, a_world, omega_body, contact_phase, leg_p, leg_v, stream=None):
        """LinearKFPositionVelocityEstimator::run on device tensors; xhat, P are updated in place.
        Returns (position, v_world, v_body) [B,3]."""
        t = self.torch
        B = xhat.shape[0]
        pos, vw, vb = (t.empty((B, 3), dtype=t.float32, device=self.device) for _ in range(3))
        st = KfState(xhat.data_ptr(), P.data_ptr(), r_body.data_ptr(), a_world.data_ptr(), omega_body.data_ptr(),
                     contact_phase.data_ptr(), leg_p.data_ptr(), leg_v.data_ptr(), pos.data_ptr(), vw.data_ptr(), vb.data_ptr())
        self._check(self.lib.qmpc_kf_step(self.h, B, C.byref(st), self._stream_ptr(stream)), "qmpc_kf_step")
        return pos, vw, vb

    def swing_trajectory(self, p0, pf, height, phase, swing_time, stream=None):
        """computeSwingTrajectoryBezier for n feet: device tensors [n,3], [n,3], [n], [n], [n] -> p, v, a."""
        t = self.torch
        n = p0.shape[0]
        p, v, a = (t.empty((n, 3), dtype=t.float32, device=self.device) for _ in range(3))
        self._check(self.lib.qmpc_swing_trajectory(self.h, n, p0.data_ptr(), pf.data_ptr(), height.data_ptr(),
                                                   phase.data_ptr(), swing_time.data_ptr(), p.data_ptr(), v.data_ptr(),
                                                   a.data_ptr(), self._stream_ptr(stream)), "qmpc_swing_trajectory")
        return p, v, a

    @staticmethod
    def solve_sharded(solvers, b, full=False):
        """qmpc_solve_sharded: ONE host thread drives several handles (one per device in
        production; several may share a device): contiguous shards of the host batch `b`, all
        devices busy at once, results collected into one set of host arrays."""
        first = solvers[0]
        B, h = int(b["batch"]), first.horizon
        keep = {k: np.ascontiguousarray(b[k], np.float32) for k in
                ("p", "v", "q", "w", "r", "yaw", "traj", "weights", "alpha", "x_drag")}
        keep["gait"] = np.ascontiguousarray(b["gait"], np.uint8)
        inp = Inputs()
        for k, a in keep.items():
            setattr(inp, k, a.ctypes.data)
        inp.weights_stride = 12 if keep["weights"].size == 12 * B else 0
        inp.alpha_stride = 1 if keep["alpha"].size == B else 0
        inp.x_drag_stride = 1 if keep["x_drag"].size == B else 0
        grf = np.zeros((B, 12), np.float32)
        st = np.zeros(B, np.int32)
        it = np.zeros(B, np.int32)
        soln = np.zeros((B, 12 * h)) if full else None
        out = Outputs(grf.ctypes.data, soln.ctypes.data if full else None, st.ctypes.data, it.ctypes.data)
        hs = (C.c_void_p * len(solvers))(*[m.h for m in solvers])
        rc = first.lib.qmpc_solve_sharded(hs, len(solvers), B, C.byref(inp), C.byref(out))
        if rc != QMPC_OK:
            raise QmpcError(f"qmpc_solve_sharded failed rc={rc}: " +
                            "; ".join(m.lib.qmpc_last_error(m.h).decode() for m in solvers))
        res = {"grf": grf, "status": st, "iters": it}
        if full:
            res["soln"] = soln
        return res

    # ---- test hook ---------------------------------------------------------
    def debug_dump(self, batch):
        """Enable the assembled-QP dump; returns (H_dev, g_dev, ld) tensors."""
        t = self.torch
        ld = self.lib.qmpc_debug_ld(self.h)
        H = t.zeros((batch, ld, ld), dtype=t.float64, device=self.device)
        g = t.zeros((batch, ld), dtype=t.float64, device=self.device)
        self._check(self.lib.qmpc_set_debug(self.h, H.data_ptr(), g.data_ptr()), "qmpc_set_debug")
        self._dbg = (H, g)
        return H, g, ld

    def debug_aux(self, batch):
        """Enable the dump of the kernel's float transcendentals; returns the [batch,8] tensor
        (cos yaw, sin yaw, roll, pitch, yaw)."""
        t = self.torch
        aux = t.zeros((batch, 8), dtype=t.float64, device=self.device)
        self._check(self.lib.qmpc_set_debug_aux(self.h, aux.data_ptr()), "qmpc_set_debug_aux")
        self._dbg_aux = aux
        return aux

    def debug_overflow_slices(self, n):
        """Test hook: use only n slices of the overflow event pool (negative: all)."""
        self._check(self.lib.qmpc_set_debug_overflow_slices(self.h, int(n)), "qmpc_set_debug_overflow_slices")

    def debug_overflow_spin(self, probes):
        """Test hook: probes for a free overflow slice before a robot falls back (negative: default)."""
        self._check(self.lib.qmpc_set_debug_overflow_spin(self.h, int(probes)), "qmpc_set_debug_overflow_spin")

    def set_split(self, mode):
        """Decoupled sweep / engine kernels for the 128- and 192-row classes: 0 / False off, 1 automatic by batch
        size (default), 2 / True always."""
        mode = 2 if mode is True else (0 if mode is False else int(mode))
        self._check(self.lib.qmpc_set_split(self.h, mode), "qmpc_set_split")

    def set_block_start(self, on):
        self._check(self.lib.qmpc_set_block_start(self.h, int(bool(on))), "qmpc_set_block_start")

    def set_dense(self, mode):
        """0 / 1 / 2: the 64-row class's five-workgroups-per-CU instantiation never / automatic (handles of 2048+ robots; see
        include/qmpc.h) / whenever that class is the whole chain."""
        self._check(self.lib.qmpc_set_dense(self.h, int(mode)), "qmpc_set_dense")

    def debug_keys(self, b):
        """The scheduling keys of DESIGN 13 as the kernels evaluate them: (stance foot-steps, score, demand) per robot (numpy)."""
        t = self.torch
        d = self.upload(b)
        B = d["batch"]
        o = self.alloc_outputs(B)
        inp, _ = self.make_args(d, o)
        nst = t.empty((B,), dtype=t.int32, device=self.device)
        score = t.empty((B,), dtype=t.float32, device=self.device)
        demand = t.empty((B,), dtype=t.float32, device=self.device)
        s = t.cuda.current_stream(self.device)
        self._check(self.lib.qmpc_debug_keys(self.h, B, C.byref(inp), nst.data_ptr(), score.data_ptr(), demand.data_ptr(),
                                             C.c_void_p(s.cuda_stream)), "qmpc_debug_keys")
        t.cuda.synchronize(self.device)
        return nst.cpu().numpy(), score.cpu().numpy(), demand.cpu().numpy()

    def set_size_order(self, on):
        """0 / 1: without a usable order hint, multi-round launches take the robots in blockIdx order / the robots that fit
        the first class largest first by their contact tables (default; include/qmpc_expert.h)."""
        self._check(self.lib.qmpc_set_size_order(self.h, int(bool(on))), "qmpc_set_size_order")

    def set_order_hint(self, mode):
        """0 / 1: multi-round launches take the robots in blockIdx order / hardest first by the previous call's
        iteration counts (default; include/qmpc.h)."""
        self._check(self.lib.qmpc_set_order_hint(self.h, int(mode)), "qmpc_set_order_hint")

    def set_debug_balance(self, mode):
        self._check(self.lib.qmpc_set_debug_balance(self.h, int(mode)), "qmpc_set_debug_balance")

    def set_chunks(self, n):
        self._check(self.lib.qmpc_set_chunks(self.h, int(n)), "qmpc_set_chunks")

    def set_debug_engine_events(self, n):
        self._check(self.lib.qmpc_set_debug_engine_events(self.h, int(n)), "qmpc_set_debug_engine_events")

    def reserve(self):
        self._check(self.lib.qmpc_reserve(self.h), "qmpc_reserve")

    def debug_read_item(self, which, item):
        """Test hook: (H^-1 [ld, ld], x_u [ld], (rid, n, nst, status)) of a work item of the decoupled path after a solve."""
        ld = (128, 192, 448)[which]
        hinv = np.zeros((ld, ld), np.float64)
        xu = np.zeros(ld, np.float64)
        hdr = np.zeros(4, np.int32)
        self._check(self.lib.qmpc_debug_read_item(self.h, int(which), int(item), hinv.ctypes.data_as(C.c_void_p),
                                                  xu.ctypes.data_as(C.c_void_p), hdr.ctypes.data_as(C.c_void_p)),
                    "qmpc_debug_read_item")
        return hinv, xu, hdr

    def debug_read_counts(self):
        """Test hook: the handle's three per-call counter sets as an int32 [3, 256] array (layout: csrc/qmpc_device.h;
        sets 0 / 1 alternate between consecutive calls, set 2 serves calls captured into a graph).  Synchronises."""
        buf = np.zeros((3, 256), np.int32)
        self._check(self.lib.qmpc_debug_read_counts(self.h, buf.ctypes.data_as(C.c_void_p)), "qmpc_debug_read_counts")
        return buf

    def set_debug_pool_busy(self, on):
        self._check(self.lib.qmpc_set_debug_pool_busy(self.h, int(bool(on))), "qmpc_set_debug_pool_busy")

    def debug_clock(self, batch):
        """Enable per-phase shader-clock stamps; returns the [batch,16] tensor."""
        t = self.torch
        clk = t.zeros((batch, 16), dtype=t.int64, device=self.device)
        self._check(self.lib.qmpc_set_debug_clock(self.h, clk.data_ptr()), "qmpc_set_debug_clock")
        self._clk = clk
        return clk

    def debug_off(self):
        self.lib.qmpc_set_debug_clock(self.h, None)
        self.lib.qmpc_set_debug(self.h, None, None)
        self.lib.qmpc_set_debug_aux(self.h, None)
        self._dbg = None
        self._dbg_aux = None
