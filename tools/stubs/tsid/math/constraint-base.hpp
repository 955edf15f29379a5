// COMPILE-CHECK STUB (tools/stubs/README.md): declarations only.  Not tsid.
#pragma once
#include <string>
#include <Eigen/Core>
namespace tsid { namespace math {
struct ConstraintBase {
    const std::string& name() const;
    unsigned int rows() const; unsigned int cols() const;
    bool isEquality() const; bool isInequality() const; bool isBound() const;
    const Eigen::MatrixXd& matrix() const;
    const Eigen::VectorXd& vector() const;
    const Eigen::VectorXd& lowerBound() const;
    const Eigen::VectorXd& upperBound() const;
};
}} // namespace tsid::math
