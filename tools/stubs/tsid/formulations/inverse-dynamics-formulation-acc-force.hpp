// COMPILE-CHECK STUB (tools/stubs/README.md): declarations only.  Not tsid / pinocchio.
#pragma once
#include <string>
#include <vector>
#include <tsid/solvers/solver-HQP-base.hpp>
namespace pinocchio {
struct Force { Eigen::VectorXd toVector() const; };
struct Frame { std::string name; };
struct Model { std::vector<Frame> frames; std::size_t getFrameId(const std::string&) const; };
struct Data { Force hg; };
const Force& computeCentroidalMomentumTimeVariation(const Model&, Data&);
} // namespace pinocchio
namespace tsid {
namespace robots {
struct RobotWrapper {
    int nq() const; int nv() const; int na() const;
    const pinocchio::Model& model() const;
    const Eigen::MatrixXd& mass(const pinocchio::Data&);
    const Eigen::VectorXd& nonLinearEffects(const pinocchio::Data&) const;
    void frameJacobianWorld(const pinocchio::Data&, std::size_t, Eigen::Matrix<double, 6, Eigen::Dynamic>&) const;
};
} // namespace robots
struct InverseDynamicsFormulationAccForce {
    unsigned int nVar() const; unsigned int nEq() const; unsigned int nIn() const;
    const pinocchio::Data& data() const;
    const solvers::HQPData& computeProblemData(double t, const Eigen::VectorXd& q, const Eigen::VectorXd& v);
    const Eigen::VectorXd& getActuatorForces(const solvers::HQPOutput&);
    const Eigen::VectorXd& getAccelerations(const solvers::HQPOutput&);
};
} // namespace tsid
