// robots::RobotWrapper: the part of tsid::robots::RobotWrapper / pinocchio::Model the reference's controllers touch at
// construction time (/root/reference/src/controllers/controller.cpp:104-118 builds it from the URDF,
// /root/reference/src/controllers/pos_tracker.cpp:51-75 adds the virtual frames and the reference configurations,
// /root/reference/src/controllers/tasks.cpp:64-80,109,194-195,287-292,315,361 read frames, placements, limits).
//
// There is no URDF parser and no pinocchio in this build: the tree arrives already parsed, as a YAML file in the
// facade's own subset (written by inria_wbc_amd/model.py::to_yaml) -- joints in depth-first order with parent, type,
// placement, inertia and limits, link frames, reference configurations.  Everything that happens per tick happens on the
// device (wbcqp_problem_data); what is here is initialisation-time forward kinematics on the host.
#ifndef IWBC_HIP_ROBOT_WRAPPER_HPP
#define IWBC_HIP_ROBOT_WRAPPER_HPP

#include <array>
#include <cmath>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include <inria_wbc/exceptions.hpp>
#include <inria_wbc/utils/yaml_lite.hpp>

#include "wbcqp.h"

namespace inria_wbc {
    namespace robots {
        // rigid placement: rotation row-major + translation (the 12-double layout of include/wbcqp.h)
        struct SE3 {
            std::array<double, 9> R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
            std::array<double, 3> p{{0, 0, 0}};
            SE3 operator*(const SE3& b) const
            {
                SE3 c;
                for (int i = 0; i < 3; ++i) {
                    for (int j = 0; j < 3; ++j) c.R[3 * i + j] = R[3 * i] * b.R[j] + R[3 * i + 1] * b.R[3 + j] + R[3 * i + 2] * b.R[6 + j];
                    c.p[i] = R[3 * i] * b.p[0] + R[3 * i + 1] * b.p[1] + R[3 * i + 2] * b.p[2] + p[i];
                }
                return c;
            }
            // tsid SE3ToVector / src/trajs/loader.cpp:11-53: translation, then rotation column-major
            std::array<double, 12> to_vector() const
            {
                std::array<double, 12> v;
                for (int i = 0; i < 3; ++i) v[i] = p[i];
                for (int j = 0; j < 3; ++j)
                    for (int i = 0; i < 3; ++i) v[3 + 3 * j + i] = R[3 * i + j];
                return v;
            }
        };

        class RobotWrapper {
        public:
            explicit RobotWrapper(const std::string& model_path, bool verbose = false)
            {
                yaml::Node root = IWBC_CHECK(yaml::LoadFile(model_path));
                name_ = IWBC_CHECK(root["name"].as<std::string>());
                floating_base_ = IWBC_CHECK(root["floating_base"].as<bool>());
                auto g = root["gravity"] ? root["gravity"].as<std::vector<double>>() : std::vector<double>{0.0, 0.0, -9.81};
                IWBC_ASSERT(g.size() == 3, "gravity needs 3 entries");
                for (int k = 0; k < 3; ++k) gravity_[k] = g[k];
                yaml::Node joints = IWBC_CHECK(root["joints"]);
                for (const auto& kv : joints) {
                    const yaml::Node& j = kv.second;
                    const auto parent = IWBC_CHECK(j["parent"].as<std::string>());
                    const auto type = IWBC_CHECK(j["type"].as<std::string>());
                    int pi = -1;
                    if (parent != "universe") {
                        pi = joint_id(parent);
                        IWBC_ASSERT(pi >= 0, "joint ", kv.first, " names a parent that does not precede it: ", parent);
                    }
                    static const char* kinds[7] = {"freeflyer", "RX", "RY", "RZ", "PX", "PY", "PZ"};
                    int jt = -1;
                    for (int k = 0; k < 7; ++k)
                        if (type == kinds[k]) jt = k;
                    IWBC_ASSERT(jt >= 0, "unknown joint type ", type, " for joint ", kv.first);
                    auto pl = IWBC_CHECK(j["placement"].as<std::vector<double>>());
                    auto in = IWBC_CHECK(j["inertia"].as<std::vector<double>>());
                    IWBC_ASSERT(pl.size() == 12 && in.size() == 10, "joint ", kv.first, ": placement needs 12 numbers, inertia 10");
                    joint_names_.push_back(kv.first);
                    parent_.push_back(pi);
                    jtype_.push_back(jt);
                    placement_.insert(placement_.end(), pl.begin(), pl.end());
                    inertia_.insert(inertia_.end(), in.begin(), in.end());
                    if (jt != WBCQP_J_FREEFLYER) {
                        auto lim = IWBC_CHECK(j["limits"].as<std::vector<double>>()); // lower, upper, velocity, effort
                        IWBC_ASSERT(lim.size() == 4, "joint ", kv.first, ": limits needs 4 numbers");
                        q_lb_.push_back(lim[0]); q_ub_.push_back(lim[1]); dq_max_.push_back(lim[2]); tau_max_.push_back(lim[3]);
                    }
                    // pinocchio adds a JOINT frame per joint
                    frame_names_.push_back(kv.first);
                    frame_body_.push_back((int)joint_names_.size() - 1);
                    const SE3 id;
                    push_placement(id);
                }
                IWBC_ASSERT(!joint_names_.empty(), "the model has no joint");
                IWBC_ASSERT((jtype_[0] == WBCQP_J_FREEFLYER) == floating_base_, "floating_base and the first joint disagree");
                if (root["frames"])
                    for (const auto& kv : root["frames"]) {
                        auto pl = IWBC_CHECK(kv.second["placement"].as<std::vector<double>>());
                        IWBC_ASSERT(pl.size() == 12, "frame ", kv.first, ": placement needs 12 numbers");
                        const int b = joint_id(IWBC_CHECK(kv.second["parent"].as<std::string>()));
                        IWBC_ASSERT(b >= 0, "frame ", kv.first, " hangs on an unknown joint");
                        frame_names_.push_back(kv.first);
                        frame_body_.push_back(b);
                        frame_placement_.insert(frame_placement_.end(), pl.begin(), pl.end());
                    }
                if (root["reference_configurations"])
                    for (const auto& kv : root["reference_configurations"]) {
                        auto q = kv.second.as<std::vector<double>>();
                        IWBC_ASSERT((int)q.size() == nq(), "reference configuration ", kv.first, " has ", q.size(), " entries, nq = ", nq());
                        reference_configurations_[kv.first] = q;
                    }
                if (verbose) std::cout << "model " << name_ << ": " << joint_names_.size() << " joints, " << frame_names_.size() << " frames" << std::endl;
            }

            const std::string& name() const { return name_; }
            bool floating_base() const { return floating_base_; }
            int nbody() const { return (int)joint_names_.size(); }
            int nq() const { return nbody() + (floating_base_ ? 6 : 0); }
            int nv() const { return nbody() + (floating_base_ ? 5 : 0); }
            int na() const { return nv() - (floating_base_ ? 6 : 0); }
            int nframes() const { return (int)frame_names_.size(); }
            bool existJointName(const std::string& n) const { return joint_id(n) >= 0; }
            bool existFrame(const std::string& n) const { return find_frame(n) >= 0; }
            int getFrameId(const std::string& n) const
            {
                const int f = find_frame(n);
                if (f < 0) IWBC_ERROR("Unknown frame or joint [", n, "]"); // tasks.cpp:68-69
                return f;
            }
            const std::vector<std::string>& frame_names() const { return frame_names_; }
            const std::vector<std::string>& joint_names() const { return joint_names_; }
            const std::vector<double>& effortLimit() const { return tau_max_; }
            const std::map<std::string, std::vector<double>>& referenceConfigurations() const { return reference_configurations_; }

            // PosTracker::parse_frames (pos_tracker.cpp:191-209): a frame `pos` away from an existing frame, same parent joint
            void addFrame(const std::string& name, const std::string& ref, const std::array<double, 3>& pos)
            {
                const int f = getFrameId(ref);
                SE3 t;
                t.p = pos;
                const SE3 pl = frame_placement(f) * t;
                frame_names_.push_back(name);
                frame_body_.push_back(frame_body_[f]);
                push_placement(pl);
            }

            SE3 frame_placement(int f) const
            {
                SE3 s;
                for (int k = 0; k < 9; ++k) s.R[k] = frame_placement_[12 * f + k];
                for (int k = 0; k < 3; ++k) s.p[k] = frame_placement_[12 * f + 9 + k];
                return s;
            }

            // ---- initialisation-time forward kinematics (the reference asks pinocchio: tasks.cpp:72-76,109,361) ----
            std::vector<SE3> bodyPlacements(const double* q) const
            {
                std::vector<SE3> o(nbody());
                for (int i = 0; i < nbody(); ++i) {
                    SE3 P, J;
                    for (int k = 0; k < 9; ++k) P.R[k] = placement_[12 * i + k];
                    for (int k = 0; k < 3; ++k) P.p[k] = placement_[12 * i + 9 + k];
                    const int jt = jtype_[i];
                    const double* qi = q + (floating_base_ ? (i == 0 ? 0 : 6 + i) : i);
                    if (jt == WBCQP_J_FREEFLYER) {
                        const double x = qi[3], y = qi[4], z = qi[5], w = qi[6];
                        J.R = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                                2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
                        J.p = {{qi[0], qi[1], qi[2]}};
                    }
                    else if (jt <= WBCQP_J_RZ) {
                        const double c = std::cos(qi[0]), s = std::sin(qi[0]);
                        if (jt == WBCQP_J_RX) J.R = {{1, 0, 0, 0, c, -s, 0, s, c}};
                        if (jt == WBCQP_J_RY) J.R = {{c, 0, s, 0, 1, 0, -s, 0, c}};
                        if (jt == WBCQP_J_RZ) J.R = {{c, -s, 0, s, c, 0, 0, 0, 1}};
                    }
                    else
                        J.p[jt - WBCQP_J_PX] = qi[0];
                    const SE3 l = P * J;
                    o[i] = parent_[i] >= 0 ? o[parent_[i]] * l : l;
                }
                return o;
            }
            SE3 framePosition(const double* q, int frame) const { return bodyPlacements(q)[frame_body_[frame]] * frame_placement(frame); }
            std::array<double, 3> com(const double* q) const
            {
                const auto o = bodyPlacements(q);
                std::array<double, 3> c{{0, 0, 0}};
                double m = 0.0;
                for (int i = 0; i < nbody(); ++i) {
                    const double mi = inertia_[10 * i];
                    for (int r = 0; r < 3; ++r)
                        c[r] += mi * (o[i].R[3 * r] * inertia_[10 * i + 1] + o[i].R[3 * r + 1] * inertia_[10 * i + 2] + o[i].R[3 * r + 2] * inertia_[10 * i + 3] + o[i].p[r]);
                    m += mi;
                }
                for (int r = 0; r < 3; ++r) c[r] /= m;
                return c;
            }

            // pointers stay valid while this object lives and is not modified
            wbcqp_model c_model() const
            {
                wbcqp_model m;
                m.nbody = nbody();
                m.floating_base = floating_base_ ? 1 : 0;
                m.parent = parent_.data(); m.jtype = jtype_.data();
                m.placement = placement_.data(); m.inertia = inertia_.data();
                for (int k = 0; k < 3; ++k) m.gravity[k] = gravity_[k];
                m.nframe = nframes();
                m.frame_body = frame_body_.data(); m.frame_placement = frame_placement_.data();
                m.q_lb = q_lb_.data(); m.q_ub = q_ub_.data(); m.dq_max = dq_max_.data();
                return m;
            }

        private:
            int joint_id(const std::string& n) const
            {
                for (size_t i = 0; i < joint_names_.size(); ++i)
                    if (joint_names_[i] == n) return (int)i;
                return -1;
            }
            int find_frame(const std::string& n) const
            {
                for (size_t i = 0; i < frame_names_.size(); ++i)
                    if (frame_names_[i] == n) return (int)i;
                return -1;
            }
            void push_placement(const SE3& s)
            {
                frame_placement_.insert(frame_placement_.end(), s.R.begin(), s.R.end());
                frame_placement_.insert(frame_placement_.end(), s.p.begin(), s.p.end());
            }

            std::string name_;
            bool floating_base_ = false;
            double gravity_[3] = {0.0, 0.0, -9.81};
            std::vector<std::string> joint_names_, frame_names_;
            std::vector<int32_t> parent_, jtype_, frame_body_;
            std::vector<double> placement_, inertia_, frame_placement_, q_lb_, q_ub_, dq_max_, tau_max_;
            std::map<std::string, std::vector<double>> reference_configurations_;
        };
    } // namespace robots
} // namespace inria_wbc
#endif
