// COMPILE-CHECK STUB (tools/stubs/README.md): declarations only.  Not tsid.
#pragma once
#include <tsid/solvers/solver-HQP-base.hpp>
namespace tsid { namespace solvers {
struct SolverHQPFactory {
    static SolverHQPBase* createNewSolver(const SolverHQP solverType, const std::string& name);
};
}} // namespace tsid::solvers
