// oracle/qpoases_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" driver around the reference's OWN vendored qpOASES 3.2.0
// (compiled unmodified from /root/reference/src/qpOASES by oracle/Makefile
// into oracle/_ref/).  It performs exactly the call sequence of
// SolverMPC.cpp:527-541:
//     QProblem problem_red(new_vars, new_cons);
//     Options op; op.setToMPC(); op.printLevel = PL_NONE;
//     problem_red.setOptions(op);
//     problem_red.init(H_red, g_red, A_red, NULL, NULL, lb_red, ub_red, nWSR);
//     problem_red.getPrimalSolution(q_red);
// Nothing here is shipped or measured as product; bench.py times it only as
// the "cpu_baseline" (kind "reference").
#include <qpOASES.hpp>

extern "C" {

// Returns 0 iff getPrimalSolution succeeded (the reference's only check,
// SolverMPC.cpp:539-541).  *init_rc receives init()'s return value (the
// reference discards it); *nwsr_used the working-set recalculations used.
int qpoases_ref_solve_ex(int nv, int nc, const double* H, const double* g,
                         const double* A, const double* lb, const double* ub,
                         int nwsr_max, double* x, int* nwsr_used,
                         int* init_rc, double* y_dual) {
  qpOASES::QProblem problem_red(nv, nc);
  qpOASES::Options op;
  op.setToMPC();
  op.printLevel = qpOASES::PL_NONE;
  problem_red.setOptions(op);
  qpOASES::int_t nWSR = nwsr_max;
  int rval = problem_red.init(H, g, A, NULL, NULL, lb, ub, nWSR);
  int rval2 = problem_red.getPrimalSolution(x);
  if (y_dual) problem_red.getDualSolution(y_dual);  // nv + nc entries
  if (nwsr_used) *nwsr_used = (int)nWSR;
  if (init_rc) *init_rc = rval;
  return rval2 == qpOASES::SUCCESSFUL_RETURN ? 0 : 1;
}

// oracle_qp_fn-compatible entry (see mpc_oracle.h)
int qpoases_ref_solve(int nv, int nc, const double* H, const double* g,
                      const double* A, const double* lb, const double* ub,
                      int nwsr_max, double* x, int* nwsr_used) {
  return qpoases_ref_solve_ex(nv, nc, H, g, A, lb, ub, nwsr_max, x, nwsr_used,
                              nullptr, nullptr);
}

}  // extern "C"
