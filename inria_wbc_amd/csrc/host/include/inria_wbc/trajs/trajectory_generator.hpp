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

        // ---- SE(3): translation by the polynomial, rotation about the fixed axis of start^-1 dest with a min-jerk angle
        //      (/root/reference/include/inria_wbc/trajs/trajectory_generator.hpp:80-147).  Poses are 12 numbers in tsid's
        //      SE3ToVector order (translation, rotation column-major); derivatives are 6 numbers (linear, angular). ----
        struct AngleAxis { double angle; double axis[3]; };
        // Eigen::AngleAxisd(Matrix3d): matrix -> quaternion -> angle, axis
        inline AngleAxis angle_axis_from_rotation(const double* R /* row-major */)
        {
            double w, x, y, z;
            const double t = R[0] + R[4] + R[8];
            if (t > 0.0) {
                double s = std::sqrt(t + 1.0);
                w = 0.5 * s;
                s = 0.5 / s;
                x = (R[7] - R[5]) * s; y = (R[2] - R[6]) * s; z = (R[3] - R[1]) * s;
            }
            else {
                int i = 0;
                if (R[4] > R[0]) i = 1;
                if (R[8] > R[4 * i]) i = 2;
                const int j = (i + 1) % 3, k = (j + 1) % 3;
                double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
                double q[3];
                q[i] = 0.5 * s;
                s = 0.5 / s;
                w = (R[3 * k + j] - R[3 * j + k]) * s;
                q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
                q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
                x = q[0]; y = q[1]; z = q[2];
            }
            AngleAxis a;
            double n = std::sqrt(x * x + y * y + z * z);
            if (n != 0.0) {
                a.angle = 2.0 * std::atan2(n, std::fabs(w));
                if (w < 0.0) n = -n;
                a.axis[0] = x / n; a.axis[1] = y / n; a.axis[2] = z / n;
            }
            else {
                a.angle = 0.0;
                a.axis[0] = 1.0; a.axis[1] = 0.0; a.axis[2] = 0.0;
            }
            return a;
        }
        namespace detail {
            inline void pose_to_rowmajor(const Vec& pose, double* R)
            {
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) R[3 * i + j] = pose[3 + 3 * j + i];
            }
            inline AngleAxis relative_rotation(const Vec& start, const Vec& dest, double* Rs)
            {
                double Rd[9], Rrel[9];
                pose_to_rowmajor(start, Rs);
                pose_to_rowmajor(dest, Rd);
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) Rrel[3 * i + j] = Rs[i] * Rd[j] + Rs[3 + i] * Rd[3 + j] + Rs[6 + i] * Rd[6 + j]; // Rs' Rd
                return angle_axis_from_rotation(Rrel);
            }
        } // namespace detail

        inline std::vector<Vec> min_jerk_trajectory_se3(const Vec& start, const Vec& dest, double dt, double trajectory_duration)
        {
            IWBC_ASSERT(start.size() == 12 && dest.size() == 12, "an SE3 pose holds 12 numbers");
            double Rs[9];
            const AngleAxis aa = detail::relative_rotation(start, dest, Rs);
            const Vec p0(start.begin(), start.begin() + 3), p1(dest.begin(), dest.begin() + 3);
            const unsigned n_steps = (unsigned)std::floor(trajectory_duration / dt);
            std::vector<Vec> trajectory(n_steps);
            for (unsigned i = 0; i < n_steps; ++i) {
                const Vec pos = minimum_jerk_polynom<d_order::ZERO>(p0, p1, dt * i, trajectory_duration);
                const double ang = minimum_jerk_polynom<d_order::ZERO>(Vec{0.0}, Vec{aa.angle}, dt * i, trajectory_duration)[0];
                // Eigen::AngleAxisd(ang, axis).toRotationMatrix()
                const double c = std::cos(ang), s = std::sin(ang), x = aa.axis[0], y = aa.axis[1], z = aa.axis[2];
                const double cx = (1 - c) * x, cy = (1 - c) * y, cz = (1 - c) * z;
                const double Ra[9] = {cx * x + c, cx * y - s * z, cx * z + s * y, cy * x + s * z, cy * y + c, cy * z - s * x,
                                      cz * x - s * y, cz * y + s * x, cz * z + c};
                Vec pose(12);
                for (int k = 0; k < 3; ++k) pose[k] = pos[k];
                for (int r = 0; r < 3; ++r)
                    for (int col = 0; col < 3; ++col)
                        pose[3 + 3 * col + r] = Rs[3 * r] * Ra[col] + Rs[3 * r + 1] * Ra[3 + col] + Rs[3 * r + 2] * Ra[6 + col];
                trajectory[i] = pose;
            }
            return trajectory;
        }
        template <unsigned ORDER>
        inline std::vector<Vec> min_jerk_trajectory_se3_d(const Vec& start, const Vec& dest, double dt, double trajectory_duration)
        {
            static_assert(ORDER == d_order::FIRST || ORDER == d_order::SECOND, "min_jerk_trajectory is not implemented for this derivative order");
            IWBC_ASSERT(start.size() == 12 && dest.size() == 12, "an SE3 pose holds 12 numbers");
            double Rs[9];
            const AngleAxis aa = detail::relative_rotation(start, dest, Rs);
            const Vec p0(start.begin(), start.begin() + 3), p1(dest.begin(), dest.begin() + 3);
            const unsigned n_steps = (unsigned)std::floor(trajectory_duration / dt);
            std::vector<Vec> trajectory(n_steps);
            for (unsigned i = 0; i < n_steps; ++i) {
                const Vec pd = minimum_jerk_polynom<ORDER>(p0, p1, dt * i, trajectory_duration);
                const double ad = minimum_jerk_polynom<ORDER>(Vec{0.0}, Vec{aa.angle}, dt * i, trajectory_duration)[0];
                Vec d(6);
                for (int k = 0; k < 3; ++k) {
                    d[k] = pd[k];
                    d[3 + k] = Rs[3 * k] * (ad * aa.axis[0]) + Rs[3 * k + 1] * (ad * aa.axis[1]) + Rs[3 * k + 2] * (ad * aa.axis[2]);
                }
                trajectory[i] = d;
            }
            return trajectory;
        }
    } // namespace trajs
} // namespace inria_wbc
#endif
