"""quadruped_ctrl_amd -- MI355X-native batched convex-MPC solver."""
