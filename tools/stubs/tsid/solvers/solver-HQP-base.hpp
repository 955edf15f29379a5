// COMPILE-CHECK STUB (tools/stubs/README.md): declarations only.  Not tsid.
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include <tsid/math/constraint-base.hpp>
namespace tsid { namespace solvers {
enum HQPStatus { HQP_STATUS_UNKNOWN = -1, HQP_STATUS_OPTIMAL = 0, HQP_STATUS_INFEASIBLE = 1, HQP_STATUS_UNBOUNDED = 2, HQP_STATUS_MAX_ITER_REACHED = 3, HQP_STATUS_ERROR = 4 };
enum SolverHQP { SOLVER_HQP_EIQUADPROG = 0, SOLVER_HQP_EIQUADPROG_FAST = 1 };
typedef std::vector<std::pair<double, std::shared_ptr<math::ConstraintBase>>> ConstraintLevel;
typedef std::vector<ConstraintLevel> HQPData;
struct HQPOutput {
    HQPStatus status;
    Eigen::VectorXd x, lambda;
    Eigen::VectorXi activeSet;
    int iterations;
};
struct SolverHQPBase {
    virtual ~SolverHQPBase();
    virtual void resize(unsigned int n, unsigned int neq, unsigned int nin) = 0;
    virtual const HQPOutput& solve(const HQPData& problemData) = 0;
    virtual double getObjectiveValue() = 0;
};
}} // namespace tsid::solvers
