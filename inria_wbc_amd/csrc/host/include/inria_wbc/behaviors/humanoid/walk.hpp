// humanoid::walk -- num_of_cycles steps forward (/root/reference/src/behaviors/humanoid/walk.cpp:8-246,
// /root/reference/etc/talos/walk.yaml): CoM over the right foot, left foot up, then per cycle {left foot down step_length ahead,
// CoM over it, right foot up and forward, down two step lengths ahead of where it was, CoM over it, left foot up}, a closing
// half step and the CoM back between the feet; the hands follow the feet.  Contacts are removed when a foot leaves the ground
// and added back on its last tick in the air, so the QP changes size twice per step.  Pose-only references.
// The reference wants a HumanoidPosTracker for the cast only; a PosTracker does.
#ifndef IWBC_HIP_WALK_HPP
#define IWBC_HIP_WALK_HPP

#include <algorithm>

#include <inria_wbc/behaviors/behavior.hpp>
#include <inria_wbc/trajs/trajectory_generator.hpp>

namespace inria_wbc {
    namespace behaviors {
        namespace humanoid {
            class Walk : public Behavior {
            public:
                enum States { INIT = 0, LF_INIT, LIFT_DOWN_LF, MOVE_COM_LEFT, LIFT_UP_RF, LIFT_DOWN_RF, MOVE_COM_RIGHT, LIFT_UP_LF,
                              LIFT_DOWN_LF_FINAL, MOVE_COM_CENTER_FINAL };

                Walk(const controller_ptr_t& controller, const yaml::Node& config) : Behavior(controller, config)
                {
                    auto h_controller = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                    IWBC_ASSERT(h_controller != NULL, "Walk: the controllers needs to be a PosTracker (or related)!");
                    for (const char* t : {"lf", "rf", "lh", "rh", "com"}) IWBC_ASSERT(h_controller->has_task(t), "Walk: a ", t, " task is required");
                    IWBC_ASSERT(h_controller->has_contact("contact_lfoot"), "Walk: a contact_lfoot task is required");
                    IWBC_ASSERT(h_controller->has_contact("contact_rfoot"), "Walk: a contact_rfoot task is required");
                    auto c = IWBC_CHECK(config["BEHAVIOR"]);
                    traj_com_duration_ = IWBC_CHECK(c["traj_com_duration"].as<double>());
                    traj_foot_duration_ = IWBC_CHECK(c["traj_foot_duration"].as<double>());
                    step_height_ = IWBC_CHECK(c["step_height"].as<double>());
                    step_length_ = IWBC_CHECK(c["step_length"].as<double>());
                    num_of_cycles_ = IWBC_CHECK(c["num_of_cycles"].as<int>());
                    if (num_of_cycles_ <= 0) IWBC_ERROR("num_of_cycles needs to be more than 0");
                    behavior_type_ = this->behavior_type();
                    controller_->set_behavior_type(behavior_type_);
                    dt_ = controller_->dt();
                    _generate_trajectories(num_of_cycles_);
                }

                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    auto controller = std::static_pointer_cast<controllers::PosTracker>(controller_);
                    if (run_) {
                        const int last = (int)com_trajs_[current_traj_].size() - 1;
                        // add and remove contacts (walk.cpp:190-209)
                        if (time_ == 0 && (state_ == LIFT_UP_LF || state_ == LF_INIT)) {
                            controller->set_behavior_type(controllers::behavior_types::SINGLE_SUPPORT);
                            controller->remove_contact("contact_lfoot");
                        }
                        if (time_ == 0 && state_ == LIFT_UP_RF) {
                            controller->set_behavior_type(controllers::behavior_types::SINGLE_SUPPORT);
                            controller->remove_contact("contact_rfoot");
                        }
                        if (time_ == last && (state_ == LIFT_DOWN_LF || state_ == LIFT_DOWN_LF_FINAL)) {
                            controller->set_behavior_type(controllers::behavior_types::DOUBLE_SUPPORT);
                            controller->add_contact("contact_lfoot");
                        }
                        if (time_ == last && state_ == LIFT_DOWN_RF) {
                            controller->set_behavior_type(controllers::behavior_types::DOUBLE_SUPPORT);
                            controller->add_contact("contact_rfoot");
                        }
                        auto at = [&](const std::vector<std::vector<trajs::Vec>>& tr) -> const trajs::Vec& {
                            const auto& seg = tr[current_traj_];
                            return seg[std::min((size_t)time_, seg.size() - 1)]; // the hands' segments are sized by traj_com_duration (walk.cpp:112-113)
                        };
                        controllers::TrajectorySample com(3);
                        com.pos = at(com_trajs_);
                        controller->set_com_ref(com);
                        auto pose = [&](const trajs::Vec& p) {
                            controllers::TrajectorySample s(0);
                            s.pos = p;
                            s.vel.assign(6, 0.0);
                            s.acc.assign(6, 0.0);
                            return s;
                        };
                        controller->set_se3_ref(pose(at(lf_trajs_)), "lf");
                        controller->set_se3_ref(pose(at(rf_trajs_)), "rf");
                        controller->set_contact_se3_ref(at(lf_trajs_), "contact_lfoot");
                        controller->set_contact_se3_ref(at(rf_trajs_), "contact_rfoot");
                        controller->set_se3_ref(pose(at(lh_trajs_)), "lh");
                        controller->set_se3_ref(pose(at(rh_trajs_)), "rh");
                    }
                    controller_->update(sensor_data);
                    if (run_) {
                        time_++;
                        if (time_ == (int)com_trajs_[current_traj_].size()) {
                            time_ = 0;
                            ++current_traj_;
                            if (current_traj_ < (int)cycle_.size()) state_ = cycle_[current_traj_];
                            else run_ = false;
                        }
                    }
                }
                std::string behavior_type() const override { return controllers::behavior_types::DOUBLE_SUPPORT; }
                bool running() const { return run_; }

            private:
                void _generate_trajectories(int num_of_cycles)
                {
                    const std::vector<int> cycle_to_repeat = {LIFT_DOWN_LF, MOVE_COM_LEFT, LIFT_UP_RF, LIFT_DOWN_RF, MOVE_COM_RIGHT, LIFT_UP_LF};
                    cycle_ = {INIT, LF_INIT};
                    for (int i = 0; i < num_of_cycles; ++i) cycle_.insert(cycle_.end(), cycle_to_repeat.begin(), cycle_to_repeat.end());
                    cycle_.push_back(LIFT_DOWN_LF_FINAL);
                    cycle_.push_back(MOVE_COM_CENTER_FINAL);
                    state_ = cycle_[0];
                    auto controller = std::static_pointer_cast<controllers::PosTracker>(controller_);
                    auto translate = [](trajs::Vec p, double v, int index) { p[index] += v; return p; };
                    trajs::Vec lf_low = controller->get_se3_ref("lf"), lf_high = translate(lf_low, step_height_, 2);
                    trajs::Vec rf_low = controller->get_se3_ref("rf"), rf_high = translate(rf_low, step_height_, 2);
                    trajs::Vec com_init = controller->get_com_ref();
                    trajs::Vec com_lf = {lf_low[0], lf_low[1], com_init[2]}, com_rf = {rf_low[0], rf_low[1], com_init[2]};
                    trajs::Vec lh_init = controller->get_se3_ref("lh"), rh_init = controller->get_se3_ref("rh");
                    trajs::Vec lh_forward = lh_init, rh_forward = rh_init;
                    auto constant = [&](const trajs::Vec& p, double T) { return std::vector<trajs::Vec>((size_t)std::floor(T / dt_), p); };
                    auto se3 = [&](const trajs::Vec& a, const trajs::Vec& b, double T) { return trajs::min_jerk_trajectory_se3(a, b, dt_, T); };
                    auto com = [&](const trajs::Vec& a, const trajs::Vec& b) { return trajs::min_jerk_trajectory<trajs::d_order::ZERO>(a, b, dt_, traj_com_duration_); };
                    const double Tc = traj_com_duration_, Tf = traj_foot_duration_;
                    double diff = 0.0;
                    for (int c : cycle_) {
                        switch (c) {
                        case INIT:
                            rf_trajs_.push_back(constant(rf_low, Tc)); lf_trajs_.push_back(constant(lf_low, Tc)); com_trajs_.push_back(com(com_init, com_rf));
                            lh_trajs_.push_back(constant(lh_init, Tc)); rh_trajs_.push_back(constant(rh_init, Tc));
                            break;
                        case LF_INIT:
                            rf_trajs_.push_back(constant(rf_low, Tf)); lf_trajs_.push_back(se3(lf_low, lf_high, Tf)); com_trajs_.push_back(constant(com_rf, Tf));
                            lh_trajs_.push_back(constant(lh_init, Tf)); rh_trajs_.push_back(constant(rh_init, Tf));
                            break;
                        case LIFT_DOWN_LF:
                            diff = rf_low[0] - lf_low[0];
                            lf_low = translate(lf_low, diff + step_length_, 0);
                            lh_forward = translate(lh_init, diff + step_length_, 0);
                            rf_trajs_.push_back(constant(rf_low, Tf)); lf_trajs_.push_back(se3(lf_high, lf_low, Tf)); com_trajs_.push_back(constant(com_rf, Tf));
                            lh_trajs_.push_back(se3(lh_init, lh_forward, Tc)); rh_trajs_.push_back(constant(rh_init, Tc));
                            lh_init = lh_forward;
                            break;
                        case MOVE_COM_LEFT:
                            com_lf[0] = lf_low[0];
                            rf_trajs_.push_back(constant(rf_low, Tc)); lf_trajs_.push_back(constant(lf_low, Tc)); com_trajs_.push_back(com(com_rf, com_lf));
                            lh_trajs_.push_back(constant(lh_init, Tc)); rh_trajs_.push_back(constant(rh_init, Tc));
                            break;
                        case LIFT_UP_RF:
                            rf_high[0] = lf_low[0];
                            rf_trajs_.push_back(se3(rf_low, rf_high, Tf)); lf_trajs_.push_back(constant(lf_low, Tf)); com_trajs_.push_back(constant(com_lf, Tf));
                            lh_trajs_.push_back(constant(lh_init, Tc)); rh_trajs_.push_back(constant(rh_init, Tc));
                            break;
                        case LIFT_DOWN_RF:
                            rf_low = translate(rf_low, 2 * step_length_, 0);
                            rh_forward = translate(rh_init, 2 * step_length_, 0);
                            rf_trajs_.push_back(se3(rf_high, rf_low, Tf)); lf_trajs_.push_back(constant(lf_low, Tf)); com_trajs_.push_back(constant(com_lf, Tf));
                            lh_trajs_.push_back(constant(lh_init, Tc)); rh_trajs_.push_back(se3(rh_init, rh_forward, Tc));
                            rh_init = rh_forward;
                            break;
                        case MOVE_COM_RIGHT:
                            com_rf[0] = rf_low[0];
                            rf_trajs_.push_back(constant(rf_low, Tc)); lf_trajs_.push_back(constant(lf_low, Tc)); com_trajs_.push_back(com(com_lf, com_rf));
                            lh_trajs_.push_back(constant(lh_init, Tc)); rh_trajs_.push_back(constant(rh_init, Tc));
                            break;
                        case LIFT_UP_LF:
                            lf_high[0] = rf_low[0];
                            rf_trajs_.push_back(constant(rf_low, Tf)); lf_trajs_.push_back(se3(lf_low, lf_high, Tf)); com_trajs_.push_back(constant(com_rf, Tf));
                            lh_trajs_.push_back(constant(lh_init, Tc)); rh_trajs_.push_back(constant(rh_init, Tc));
                            break;
                        case LIFT_DOWN_LF_FINAL:
                            lf_low = translate(lf_low, step_length_, 0);
                            lh_forward = translate(lh_init, step_length_, 0);
                            rf_trajs_.push_back(constant(rf_low, Tf)); lf_trajs_.push_back(se3(lf_high, lf_low, Tf)); com_trajs_.push_back(constant(com_rf, Tf));
                            lh_trajs_.push_back(se3(lh_init, lh_forward, Tc)); rh_trajs_.push_back(constant(rh_init, Tc));
                            break;
                        case MOVE_COM_CENTER_FINAL:
                            com_init[0] = (rf_low[0] + lf_low[0]) / 2.0;
                            com_init[1] = (rf_low[1] + lf_low[1]) / 2.0;
                            rf_trajs_.push_back(constant(rf_low, Tc)); lf_trajs_.push_back(constant(lf_low, Tc)); com_trajs_.push_back(com(com_rf, com_init));
                            lh_trajs_.push_back(constant(lh_forward, Tc)); rh_trajs_.push_back(constant(rh_forward, Tc));
                            break;
                        }
                    }
                }

                int time_ = 0, current_traj_ = 0, state_ = INIT, num_of_cycles_ = 1;
                bool run_ = true;
                double dt_ = 0.001, traj_com_duration_ = 1.0, traj_foot_duration_ = 1.0, step_height_ = 0.1, step_length_ = 0.2;
                std::vector<int> cycle_;
                std::vector<std::vector<trajs::Vec>> lf_trajs_, rf_trajs_, com_trajs_, lh_trajs_, rh_trajs_;
            };
        } // namespace humanoid
    } // namespace behaviors
} // namespace inria_wbc
#endif
