"""PyTorch custom operator over the C ABI: `torch.ops.qmpc.solve` for batched RL /
simulation users who hold their robot states in torch tensors on the GPU.

    import quadruped_ctrl_amd.torch_op            # registers the ops
    grf, soln, status, iters = torch.ops.qmpc.solve(p, v, q, w, r, yaw, traj, gait, weights, alpha, x_drag,
                                                    0.026, 0.4, 120.0, False)

The operator is plumbing only: tensors stay where they are (device pointers go straight into
qmpc_solve), the work is enqueued on torch's CURRENT stream, nothing synchronises, and all compute is
the hand-written HIP kernels of libqmpc.so.  There is no CPU implementation: the op is registered for
the "cuda" device type only, so a CPU tensor fails loudly in the dispatcher.

Handles are cached per (device, horizon, dt, max_stance_hint) -- what the handle's tables and pools depend on; mu and f_max are
plain parameters of the next launch, so sweeping them (domain randomisation) reuses ONE handle -- and
grown when a larger batch arrives.  The cache holds at most MAX_HANDLES entries (least recently used
is destroyed: a handle owns a few hundred MB of event pools).  One handle serialises its calls
(include/qmpc.h, "Streams"), so concurrent streams on one device are ordered, not raced.
"""
from collections import OrderedDict
import ctypes as C

import torch

from . import binding as _b

_handles = OrderedDict()
MAX_HANDLES = 4
# Caller's bound on stance foot-steps per robot (qmpc_set_max_stance; 0 = unknown), applied to handles created from now
# on BEFORE their setup: qmpc_setup allocates the pools of every size class the bound leaves reachable -- at horizons
# above 16 an unhinted 1024-robot handle owns 1.5 GiB of large-problem work items that a trot (2 feet x h <= 64
# foot-steps) never uses.  A robot beyond the bound is reported (QMPC_ST_WS_FULL), not solved.
max_stance_hint = 0


def _solver(device, horizon, dt, mu, f_max, batch):
    key = (device.index, int(horizon), float(dt), int(max_stance_hint))
    ent = _handles.get(key)
    if ent is None or ent[1] < batch:
        if ent is not None:
            ent[0].close()
        cap = max(int(batch), 1024)
        m = _b.BatchedConvexMPC(device.index, max_batch=cap, max_horizon=_b_max_horizon())
        if max_stance_hint:
            m.set_max_stance(int(max_stance_hint))
        ent = (m, cap)
        _handles[key] = ent
        while len(_handles) > MAX_HANDLES:            # least recently used first
            _, (old, _cap) = _handles.popitem(last=False)
            old.close()
    _handles.move_to_end(key)
    # qmpc_setup returns at once when (dt, horizon) are unchanged: mu / f_max only update the parameter block
    ent[0].setup(dt, horizon, mu, f_max)
    return ent[0]


def _b_max_horizon():
    return int(_b.load_library().qmpc_max_horizon())


def _chk(t, name, dtype, shape):
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
        raise ValueError(f"qmpc::solve: {name} must be a contiguous {dtype} tensor of shape {tuple(shape)}, "
                         f"got {t.dtype} {tuple(t.shape)}")


@torch.library.custom_op("qmpc::solve", mutates_args=(), device_types="cuda")
def solve(p: torch.Tensor, v: torch.Tensor, q: torch.Tensor, w: torch.Tensor, r: torch.Tensor, yaw: torch.Tensor,
          traj: torch.Tensor, gait: torch.Tensor, weights: torch.Tensor, alpha: torch.Tensor, x_drag: torch.Tensor,
          dt: float, mu: float, f_max: float, full: bool) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Batched convex-MPC solve (the reference's update_problem_data_floats -> solve_mpc -> get_solution for B
    robots).  Layouts as include/qmpc.h: p v w [B,3], q [B,4] (w,x,y,z), r [B,12] axis-major, yaw [B],
    traj [B,12h], gait [B,4h] uint8, weights [B,12] or [12], alpha / x_drag [B] or [1].
    Returns grf [B,12] float32, soln [B,12h] float64 (empty [0,12h] unless `full`), status [B] int32,
    iters [B] int32."""
    B = p.shape[0]
    if traj.dim() != 2 or traj.shape[1] % 12:
        raise ValueError("qmpc::solve: traj must be [B, 12*horizon]")
    h = traj.shape[1] // 12
    f32 = torch.float32
    _chk(p, "p", f32, (B, 3)); _chk(v, "v", f32, (B, 3)); _chk(q, "q", f32, (B, 4)); _chk(w, "w", f32, (B, 3))
    _chk(r, "r", f32, (B, 12)); _chk(yaw, "yaw", f32, (B,)); _chk(traj, "traj", f32, (B, 12 * h))
    _chk(gait, "gait", torch.uint8, (B, 4 * h))
    for name, t, sizes in (("weights", weights, (12, 12 * B)), ("alpha", alpha, (1, B)), ("x_drag", x_drag, (1, B))):
        # (a float64 tensor, e.g. one built from numpy, would be reinterpreted as float32 by the kernel)
        if t.dtype != f32 or t.numel() not in sizes or t.device != p.device:
            raise ValueError(f"qmpc::solve: {name} must be float32 on {p.device} with {sizes[0]} or {sizes[1]} elements "
                             f"(weights [B,12] or [12]; alpha, x_drag [B] or [1]), got {t.dtype} {tuple(t.shape)} on {t.device}")
    if weights.numel() == 12 * B and B > 1 and tuple(weights.shape) != (B, 12):
        raise ValueError(f"qmpc::solve: per-robot weights must have shape {(B, 12)}, got {tuple(weights.shape)}")
    dev = p.device
    m = _solver(dev, h, dt, mu, f_max, B)
    grf = torch.empty((B, 12), dtype=f32, device=dev)
    soln = torch.empty((B if full else 0, 12 * h), dtype=torch.float64, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    if B == 0:
        return grf, soln, status, iters
    inp = _b.Inputs()
    for name, t in (("p", p), ("v", v), ("q", q), ("w", w), ("r", r), ("yaw", yaw), ("traj", traj), ("gait", gait),
                    ("weights", weights.contiguous()), ("alpha", alpha.contiguous()), ("x_drag", x_drag.contiguous())):
        setattr(inp, name, t.data_ptr())
    # stride 0 = one row shared by the batch (with B == 1 both readings address the same row)
    inp.weights_stride = 0 if weights.numel() == 12 else 12
    inp.alpha_stride = 0 if alpha.numel() == 1 else 1
    inp.x_drag_stride = 0 if x_drag.numel() == 1 else 1
    out = _b.Outputs(grf.data_ptr(), soln.data_ptr() if full else None, status.data_ptr(), iters.data_ptr())
    stream = torch.cuda.current_stream(dev)
    rc = m.lib.qmpc_solve(m.h, B, C.byref(inp), C.byref(out), C.c_void_p(stream.cuda_stream))
    m._check(rc, "qmpc_solve")
    return grf, soln, status, iters


@solve.register_fake
def _solve_fake(p, v, q, w, r, yaw, traj, gait, weights, alpha, x_drag, dt, mu, f_max, full):
    B, h12 = p.shape[0], traj.shape[1]
    return (p.new_empty((B, 12)), p.new_empty((B if full else 0, h12), dtype=torch.float64),
            p.new_empty((B,), dtype=torch.int32), p.new_empty((B,), dtype=torch.int32))


def release_handles():
    """Destroy the cached solver handles (tests / interpreter shutdown)."""
    for m, _ in _handles.values():
        m.close()
    _handles.clear()
