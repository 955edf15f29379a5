// humanoid::move-feet -- generic::cartesian for the feet, with the matching contact's motion task following the same sample
// (pose, velocity, acceleration): the contact constraint itself moves (/root/reference/src/behaviors/humanoid/move_feet.cpp:8-102,
// /root/reference/etc/talos/move_feet.yaml; PosTracker::set_contact_se3_ref(sample, name), pos_tracker.cpp:240-244).
#ifndef IWBC_HIP_MOVE_FEET_HPP
#define IWBC_HIP_MOVE_FEET_HPP

#include <inria_wbc/behaviors/behavior.hpp>
#include <inria_wbc/trajs/trajectory_generator.hpp>

namespace inria_wbc {
    namespace behaviors {
        namespace generic {
            class MoveFeet : public Behavior {
            public:
                MoveFeet(const controller_ptr_t& controller, const yaml::Node& config) : Behavior(controller, config)
                {
                    auto tracker = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                    IWBC_ASSERT(tracker, "Need a PosTracker for MoveFeet");
                    auto c = IWBC_CHECK(config["BEHAVIOR"]);
                    trajectory_duration_ = IWBC_CHECK(c["trajectory_duration"].as<double>());
                    behavior_type_ = this->behavior_type();
                    controller_->set_behavior_type(behavior_type_);
                    loop_ = IWBC_CHECK(c["loop"].as<bool>());
                    task_names_ = IWBC_CHECK(c["task_names"].as<std::vector<std::string>>());
                    contact_names_ = IWBC_CHECK(c["contact_names"].as<std::vector<std::string>>());
                    for (auto& task_name : task_names_) IWBC_ASSERT(tracker->has_task(task_name), "active_walk: a " + task_name + " task is required");
                    for (auto& contact_name : contact_names_)
                        IWBC_ASSERT(tracker->has_contact(contact_name), "active_walk: a " + contact_name + " task is required");
                    auto ts = IWBC_CHECK(c["relative_targets_pos"].as<std::vector<std::vector<double>>>());
                    auto to = IWBC_CHECK(c["relative_targets_rpy"].as<std::vector<std::vector<double>>>());
                    if (task_names_.size() != ts.size()) IWBC_ERROR("MoveFeet behavior needs the same number of tasks and targets");
                    IWBC_ASSERT(contact_names_.size() == task_names_.size(), "MoveFeet behavior needs one contact per task");
                    for (size_t i = 0; i < task_names_.size(); ++i) {
                        const trajs::Vec task_init = tracker->get_se3_ref(task_names_[i]);
                        trajs::Vec task_final = task_init;
                        if (ts[i].size() == 3)
                            for (int k = 0; k < 3; ++k) task_final[k] = ts[i][k] + task_init[k];
                        if (i < to.size() && to[i].size() == 3) {
                            const double cr = std::cos(to[i][0]), sr = std::sin(to[i][0]), cp = std::cos(to[i][1]), sp = std::sin(to[i][1]),
                                         cy = std::cos(to[i][2]), sy = std::sin(to[i][2]);
                            const double rot[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr,
                                                   sy * sp * cr - cy * sr, -sp, cp * sr, cp * cr};
                            for (int col = 0; col < 3; ++col)
                                for (int r = 0; r < 3; ++r) {
                                    double v = 0.0;
                                    for (int k = 0; k < 3; ++k) v += rot[3 * r + k] * task_init[3 + 3 * col + k];
                                    task_final[3 + 3 * col + r] = v;
                                }
                        }
                        const double dt = controller_->dt();
                        std::vector<std::vector<trajs::Vec>> tr, tr_d, tr_dd;
                        tr.push_back(trajs::min_jerk_trajectory_se3(task_init, task_final, dt, trajectory_duration_));
                        tr_d.push_back(trajs::min_jerk_trajectory_se3_d<trajs::d_order::FIRST>(task_init, task_final, dt, trajectory_duration_));
                        tr_dd.push_back(trajs::min_jerk_trajectory_se3_d<trajs::d_order::SECOND>(task_init, task_final, dt, trajectory_duration_));
                        if (loop_) {
                            tr.push_back(trajs::min_jerk_trajectory_se3(task_final, task_init, dt, trajectory_duration_));
                            tr_d.push_back(trajs::min_jerk_trajectory_se3_d<trajs::d_order::FIRST>(task_final, task_init, dt, trajectory_duration_));
                            tr_dd.push_back(trajs::min_jerk_trajectory_se3_d<trajs::d_order::SECOND>(task_final, task_init, dt, trajectory_duration_));
                        }
                        trajectories_.push_back(tr);
                        trajectories_d_.push_back(tr_d);
                        trajectories_dd_.push_back(tr_dd);
                    }
                }
                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    auto tracker = std::static_pointer_cast<controllers::PosTracker>(controller_);
                    for (size_t i = 0; i < task_names_.size(); ++i)
                        if (traj_selector_ < trajectories_[i].size()) {
                            controllers::TrajectorySample sample_ref(0);
                            sample_ref.pos = trajectories_[i][traj_selector_][time_];
                            sample_ref.vel = trajectories_d_[i][traj_selector_][time_];
                            sample_ref.acc = trajectories_dd_[i][traj_selector_][time_];
                            tracker->set_se3_ref(sample_ref, task_names_[i]);
                            tracker->set_contact_se3_ref(sample_ref, contact_names_[i]);
                        }
                    controller_->update(sensor_data);
                    ++time_;
                    if (!trajectories_.empty() && traj_selector_ < trajectories_[0].size() && time_ == trajectories_[0][traj_selector_].size()) {
                        time_ = 0;
                        ++traj_selector_;
                        if (loop_) traj_selector_ = traj_selector_ % trajectories_[0].size();
                    }
                }
                std::string behavior_type() const override { return controllers::behavior_types::DOUBLE_SUPPORT; }

            private:
                size_t time_ = 0, traj_selector_ = 0;
                double trajectory_duration_ = 0.0;
                bool loop_ = false;
                std::vector<std::string> task_names_, contact_names_;
                std::vector<std::vector<std::vector<trajs::Vec>>> trajectories_, trajectories_d_, trajectories_dd_;
            };
        } // namespace generic
    } // namespace behaviors
} // namespace inria_wbc
#endif
