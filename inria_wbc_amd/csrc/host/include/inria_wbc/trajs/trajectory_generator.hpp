// Minimum-jerk polynomials and sampled trajectories (closed forms of
// /root/reference/include/inria_wbc/trajs/trajectory_generator.hpp:23-78) on std::vector<double>.
#ifndef IWBC_HIP_TRAJECTORY_GENERATOR_HPP
#define IWBC_HIP_TRAJECTORY_GENERATOR_HPP

#include <cmath>
#include <vector>

#include <inria_wbc/exceptions.hpp>

namespace inria_wbc {
    namespace trajs {
        namespace d_order {
            constexpr unsigned ZERO = 0, FIRST = 1, SECOND = 2;
        }
        using Vec = std::vector<double>;

        template <unsigned DO>
        inline Vec minimum_jerk_polynom(const Vec& x0, const Vec& xf, double t, double trajectory_duration)
        {
            IWBC_ASSERT(x0.size() == xf.size(), "minimum_jerk_polynom x0 and xf should have the same size");
            static_assert(DO <= d_order::SECOND, "minimum_jerk_polynom is implemented up to the second derivative");
            const double s = t / trajectory_duration;
            double k;
            if (DO == d_order::ZERO)
                k = 6 * s * s * s * s * s - 15 * s * s * s * s + 10 * s * s * s;
            else if (DO == d_order::FIRST)
                k = (30 * s * s * s * s - 60 * s * s * s + 30 * s * s) / trajectory_duration;
            else
                k = (120 * s * s * s - 180 * s * s + 60 * s) / (trajectory_duration * trajectory_duration);
            Vec out(x0.size());
            for (size_t i = 0; i < x0.size(); ++i) out[i] = (DO == d_order::ZERO ? x0[i] : 0.0) + (xf[i] - x0[i]) * k;
            return out;
        }

        template <unsigned ORDER = d_order::ZERO>
        inline std::vector<Vec> min_jerk_trajectory(const Vec& start, const Vec& dest, double dt, double trajectory_duration)
        {
            const unsigned n_steps = (unsigned)std::floor(trajectory_duration / dt);
            std::vector<Vec> trajectory(n_steps);
            for (unsigned i = 0; i < n_steps; ++i) trajectory[i] = minimum_jerk_polynom<ORDER>(start, dest, dt * i, trajectory_duration);
            return trajectory;
        }
    } // namespace trajs
} // namespace inria_wbc
#endif
