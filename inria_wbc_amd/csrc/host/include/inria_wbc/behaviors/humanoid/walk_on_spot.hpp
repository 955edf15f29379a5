// humanoid::walk-on-spot -- the behaviour that changes the QP's size while it runs (what it has to do is set by
// /root/reference/src/behaviors/humanoid/walk_on_spot.cpp:8-211 and /root/reference/etc/talos/walk_on_spot.yaml): the CoM moves over one foot, the
// other foot's contact is removed (n 74 -> 62, nEq 18 -> 12, nIn 122 -> 105 on Talos), the foot goes up and down on a min-jerk path, its contact comes
// back, and the same on the other side.
//
// Here the gait is DATA: a table of phases, each with its three reference paths (CoM, left foot, right foot), the contact it lets go of on its first
// tick and the contact it takes back on its last; update() only walks the table.  The first phase (from the stance's middle over the right foot)
// runs once, the other six repeat.  A PosTracker is enough as controller -- nothing of the humanoid tracker's stabiliser is used.
#ifndef IWBC_HIP_WALK_ON_SPOT_HPP
#define IWBC_HIP_WALK_ON_SPOT_HPP

#include <inria_wbc/behaviors/behavior.hpp>
#include <inria_wbc/trajs/trajectory_generator.hpp>

namespace inria_wbc {
    namespace behaviors {
        namespace humanoid {
            class WalkOnSpot : public Behavior {
            public:
                // (the reference's state names, in its order: state() is compared with them by callers)
                enum States { INIT = 0, LIFT_UP_LF, LIFT_DOWN_LF, MOVE_COM_LEFT, LIFT_UP_RF, LIFT_DOWN_RF, MOVE_COM_RIGHT };

                WalkOnSpot(const controller_ptr_t& controller, const yaml::Node& config) : Behavior(controller, config)
                {
                    tracker_ = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                    IWBC_ASSERT(tracker_ != nullptr, "walk-on-spot drives a PosTracker (or a controller derived from it)");
                    for (const char* task : {"lf", "rf", "com"}) IWBC_ASSERT(tracker_->has_task(task), "walk-on-spot: the stack has no task '", task, "'");
                    for (const char* contact : {kLeft, kRight}) IWBC_ASSERT(tracker_->has_contact(contact), "walk-on-spot: the stack has no contact '", contact, "'");
                    const auto params = IWBC_CHECK(config["BEHAVIOR"]);
                    const double t_com = IWBC_CHECK(params["traj_com_duration"].as<double>());
                    const double t_foot = IWBC_CHECK(params["traj_foot_duration"].as<double>());
                    const double lift = IWBC_CHECK(params["step_height"].as<double>());
                    behavior_type_ = this->behavior_type();
                    controller_->set_behavior_type(behavior_type_);
                    build_table(controller_->dt(), t_com, t_foot, lift);
                }

                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    const Phase& ph = table_[phase_];
                    const int last = (int)ph.com.size() - 1;
                    if (tick_ == 0 && ph.lets_go) { // the QP loses a contact: 12 variables, 6 equalities, 17 inequality rows
                        tracker_->set_behavior_type(controllers::behavior_types::SINGLE_SUPPORT);
                        tracker_->remove_contact(ph.lets_go);
                    }
                    if (tick_ == last && ph.takes_back) { // ... and gets it back with the foot on the ground again
                        tracker_->set_behavior_type(controllers::behavior_types::DOUBLE_SUPPORT);
                        tracker_->add_contact(ph.takes_back);
                    }
                    // position references only: velocity and acceleration references stay zero (the reference samples its paths the same way, :79-84)
                    controllers::TrajectorySample com(3), lf(0), rf(0);
                    com.pos = ph.com[tick_];
                    for (auto* foot : {&lf, &rf}) {
                        foot->vel.assign(6, 0.0);
                        foot->acc.assign(6, 0.0);
                    }
                    lf.pos = ph.lf[tick_];
                    rf.pos = ph.rf[tick_];
                    tracker_->set_com_ref(com);
                    tracker_->set_se3_ref(lf, "lf");
                    tracker_->set_se3_ref(rf, "rf");
                    tracker_->set_contact_se3_ref(lf.pos, kLeft);
                    tracker_->set_contact_se3_ref(rf.pos, kRight);
                    controller_->update(sensor_data);
                    if (++tick_ > last) {
                        tick_ = 0;
                        phase_ = (phase_ + 1 < (int)table_.size()) ? phase_ + 1 : 1; // the opening phase is not part of the cycle
                    }
                }
                std::string behavior_type() const override { return controllers::behavior_types::DOUBLE_SUPPORT; }
                int state() const { return table_[phase_].state; }

            private:
                static constexpr const char* kLeft = "contact_lfoot";
                static constexpr const char* kRight = "contact_rfoot";
                using Path = std::vector<trajs::Vec>;
                struct Phase {
                    int state;
                    const char* lets_go;    // contact removed on the phase's first tick (nullptr: none)
                    const char* takes_back; // contact added on its last tick
                    Path com, lf, rf;       // one sample per tick, all three of the same length
                };

                void build_table(double dt, double t_com, double t_foot, double lift)
                {
                    const trajs::Vec lf_down = tracker_->get_se3_ref("lf"), rf_down = tracker_->get_se3_ref("rf"), com_mid = tracker_->get_com_ref();
                    auto raised = [lift](trajs::Vec pose) { pose[2] += lift; return pose; };
                    const trajs::Vec lf_up = raised(lf_down), rf_up = raised(rf_down);
                    // the CoM stands over a foot at the height it starts from (walk_on_spot.cpp:62-67)
                    const trajs::Vec over_lf = {lf_down[0], lf_down[1], com_mid[2]}, over_rf = {rf_down[0], rf_down[1], com_mid[2]};
                    auto hold = [dt](const trajs::Vec& p, double T) { return Path((size_t)std::floor(T / dt), p); };
                    auto swing = [dt, t_foot](const trajs::Vec& a, const trajs::Vec& b) { return trajs::min_jerk_trajectory_se3(a, b, dt, t_foot); };
                    auto shift = [dt, t_com](const trajs::Vec& a, const trajs::Vec& b) { return trajs::min_jerk_trajectory<trajs::d_order::ZERO>(a, b, dt, t_com); };
                    //            state            lets go  takes back  CoM                     left foot              right foot
                    table_ = {{INIT,           nullptr, nullptr, shift(com_mid, over_rf), hold(lf_down, t_com),  hold(rf_down, t_com)},
                              {LIFT_UP_LF,     kLeft,   nullptr, hold(over_rf, t_foot),   swing(lf_down, lf_up), hold(rf_down, t_foot)},
                              {LIFT_DOWN_LF,   nullptr, kLeft,   hold(over_rf, t_foot),   swing(lf_up, lf_down), hold(rf_down, t_foot)},
                              {MOVE_COM_LEFT,  nullptr, nullptr, shift(over_rf, over_lf), hold(lf_down, t_com),  hold(rf_down, t_com)},
                              {LIFT_UP_RF,     kRight,  nullptr, hold(over_lf, t_foot),   hold(lf_down, t_foot), swing(rf_down, rf_up)},
                              {LIFT_DOWN_RF,   nullptr, kRight,  hold(over_lf, t_foot),   hold(lf_down, t_foot), swing(rf_up, rf_down)},
                              {MOVE_COM_RIGHT, nullptr, nullptr, shift(over_lf, over_rf), hold(lf_down, t_com),  hold(rf_down, t_com)}};
                    for (const Phase& ph : table_)
                        IWBC_ASSERT(!ph.com.empty() && ph.com.size() == ph.lf.size() && ph.com.size() == ph.rf.size(),
                                    "walk-on-spot: the paths of phase ", ph.state, " differ in length (", ph.com.size(), " / ", ph.lf.size(), " / ", ph.rf.size(), ")");
                }

                std::shared_ptr<controllers::PosTracker> tracker_;
                std::vector<Phase> table_;
                int phase_ = 0, tick_ = 0;
            };
        } // namespace humanoid
    } // namespace behaviors
} // namespace inria_wbc
#endif
