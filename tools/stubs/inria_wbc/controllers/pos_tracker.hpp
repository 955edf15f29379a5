// COMPILE-CHECK STUB (tools/stubs/README.md): declarations only (reference: include/inria_wbc/controllers/pos_tracker.hpp, controller.hpp).
#pragma once
#include <memory>
#include <string>
#include <tsid/formulations/inverse-dynamics-formulation-acc-force.hpp>
namespace YAML {
struct Node {
    Node operator[](const char*) const;
    template <typename T> T as() const;
};
Node LoadFile(const std::string&);
} // namespace YAML
namespace tsid { namespace tasks {
struct TaskSE3Equality { const Eigen::VectorXd& position_error() const; const math::ConstraintBase& getConstraint() const; };
struct TaskJointPosVelAccBounds { const math::ConstraintBase& getConstraint() const; };
}} // namespace tsid::tasks
namespace inria_wbc { namespace controllers {
struct PosTracker {
    explicit PosTracker(const YAML::Node&);
    std::shared_ptr<tsid::InverseDynamicsFormulationAccForce> tsid();
    std::shared_ptr<tsid::robots::RobotWrapper> robot();
    double dt() const; double t() const;
    const Eigen::VectorXd& q_tsid() const;
    Eigen::VectorXd dq(bool filter_mimics = true) const;
    Eigen::VectorXd ddq(bool filter_mimics = true) const;
    bool has_task(const std::string&) const;
    std::shared_ptr<tsid::tasks::TaskSE3Equality> se3_task(const std::string&);
    std::shared_ptr<tsid::tasks::TaskJointPosVelAccBounds> bound_task();
};
}} // namespace inria_wbc::controllers
