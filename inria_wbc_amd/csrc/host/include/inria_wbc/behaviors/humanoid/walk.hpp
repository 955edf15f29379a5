// humanoid::walk -- num_of_cycles double steps forward (what it has to do is set by /root/reference/src/behaviors/humanoid/walk.cpp:8-246 and
// /root/reference/etc/talos/walk.yaml): CoM over the right foot, left foot up, then per cycle {left foot down a step ahead of the right, CoM over it,
// right foot up and forward, down two step lengths ahead of where it was, CoM over it, left foot up}, a closing half step and the CoM back between
// the feet; each hand follows its foot.  A foot's contact is removed on the first tick of its way up and added on the last tick of its way down, so the
// QP changes size twice per step.  Pose-only references.
//
// Here the walk is a PLAN written once by a small planner that knows where feet, hands and CoM stand (shift / raise / lower) and played back tick by
// tick: a phase carries its five reference paths and the contact it lets go of or takes back.  A PosTracker is enough as controller.
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
                // (the reference's state names, in its order)
                enum States { INIT = 0, LF_INIT, LIFT_DOWN_LF, MOVE_COM_LEFT, LIFT_UP_RF, LIFT_DOWN_RF, MOVE_COM_RIGHT, LIFT_UP_LF,
                              LIFT_DOWN_LF_FINAL, MOVE_COM_CENTER_FINAL };

                Walk(const controller_ptr_t& controller, const yaml::Node& config) : Behavior(controller, config)
                {
                    tracker_ = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                    IWBC_ASSERT(tracker_ != nullptr, "walk drives a PosTracker (or a controller derived from it)");
                    for (const char* task : {"lf", "rf", "lh", "rh", "com"}) IWBC_ASSERT(tracker_->has_task(task), "walk: the stack has no task '", task, "'");
                    for (const Side& s : kSides) IWBC_ASSERT(tracker_->has_contact(s.contact), "walk: the stack has no contact '", s.contact, "'");
                    const auto params = IWBC_CHECK(config["BEHAVIOR"]);
                    Gait g;
                    g.t_com = IWBC_CHECK(params["traj_com_duration"].as<double>());
                    g.t_foot = IWBC_CHECK(params["traj_foot_duration"].as<double>());
                    g.lift = IWBC_CHECK(params["step_height"].as<double>());
                    g.step = IWBC_CHECK(params["step_length"].as<double>());
                    g.cycles = IWBC_CHECK(params["num_of_cycles"].as<int>());
                    if (g.cycles <= 0) IWBC_ERROR("num_of_cycles needs to be more than 0");
                    g.dt = controller_->dt();
                    behavior_type_ = this->behavior_type();
                    controller_->set_behavior_type(behavior_type_);
                    write_plan(g);
                }

                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    if (running()) {
                        const Phase& ph = plan_[phase_];
                        const int last = (int)ph.com.size() - 1;
                        if (tick_ == 0 && ph.lets_go) {
                            tracker_->set_behavior_type(controllers::behavior_types::SINGLE_SUPPORT);
                            tracker_->remove_contact(ph.lets_go);
                        }
                        if (tick_ == last && ph.takes_back) {
                            tracker_->set_behavior_type(controllers::behavior_types::DOUBLE_SUPPORT);
                            tracker_->add_contact(ph.takes_back);
                        }
                        // (a hand's path is as long as a CoM shift even in a foot's phase, the reference's :112-113: past its end it rests on its last sample)
                        auto now = [this](const Path& p) -> const trajs::Vec& { return p[std::min((size_t)tick_, p.size() - 1)]; };
                        auto pose_only = [](const trajs::Vec& p) {
                            controllers::TrajectorySample s(0);
                            s.pos = p;
                            s.vel.assign(6, 0.0);
                            s.acc.assign(6, 0.0);
                            return s;
                        };
                        controllers::TrajectorySample com(3);
                        com.pos = now(ph.com);
                        tracker_->set_com_ref(com);
                        for (const Side& s : kSides) {
                            const Path& foot = (&s == &kSides[0]) ? ph.lf : ph.rf;
                            const Path& hand = (&s == &kSides[0]) ? ph.lh : ph.rh;
                            tracker_->set_se3_ref(pose_only(now(foot)), s.foot);
                            tracker_->set_contact_se3_ref(now(foot), s.contact);
                            tracker_->set_se3_ref(pose_only(now(hand)), s.hand);
                        }
                    }
                    const bool played = running();
                    controller_->update(sensor_data);
                    if (played && ++tick_ == (int)plan_[phase_].com.size()) {
                        tick_ = 0;
                        ++phase_; // past the plan's end the last references stand
                    }
                }
                std::string behavior_type() const override { return controllers::behavior_types::DOUBLE_SUPPORT; }
                bool running() const { return phase_ < (int)plan_.size(); }

            private:
                using Path = std::vector<trajs::Vec>;
                struct Side { const char *foot, *hand, *contact; };
                static constexpr Side kSides[2] = {{"lf", "lh", "contact_lfoot"}, {"rf", "rh", "contact_rfoot"}};
                struct Gait { double dt = 0.001, t_com = 1.0, t_foot = 1.0, lift = 0.1, step = 0.2; int cycles = 1; };
                struct Phase {
                    int state;
                    const char* lets_go;    // contact removed on the phase's first tick (nullptr: none)
                    const char* takes_back; // contact added on its last tick
                    Path com, lf, rf, lh, rh;
                };

                void write_plan(const Gait& g)
                {
                    enum { L = 0, R = 1 };
                    // where things stand while the plan is written
                    trajs::Vec ground[2] = {tracker_->get_se3_ref("lf"), tracker_->get_se3_ref("rf")}; // a foot's pose on the floor
                    trajs::Vec air[2] = {ground[L], ground[R]};                                          // ... and at the top of its swing
                    trajs::Vec hand[2] = {tracker_->get_se3_ref("lh"), tracker_->get_se3_ref("rh")};
                    trajs::Vec hand_ahead[2] = {hand[L], hand[R]}; // where a hand was last sent to
                    trajs::Vec com = tracker_->get_com_ref();
                    const double com_height = com[2];
                    air[L][2] += g.lift;
                    air[R][2] += g.lift;
                    trajs::Vec over[2] = {{ground[L][0], ground[L][1], com_height}, {ground[R][0], ground[R][1], com_height}}; // the CoM above a foot

                    auto hold = [&g](const trajs::Vec& p, double T) { return Path((size_t)std::floor(T / g.dt), p); };
                    auto move = [&g](const trajs::Vec& a, const trajs::Vec& b, double T) { return trajs::min_jerk_trajectory_se3(a, b, g.dt, T); };
                    auto emit = [this](int state, const char* lets_go, const char* takes_back, Path c, Path lf, Path rf, Path lh, Path rh) {
                        plan_.push_back({state, lets_go, takes_back, std::move(c), std::move(lf), std::move(rf), std::move(lh), std::move(rh)});
                    };
                    // the CoM goes from where it is to `to`; feet and hands rest
                    auto shift = [&](int state, const trajs::Vec& to, const trajs::Vec& lh_at, const trajs::Vec& rh_at) {
                        emit(state, nullptr, nullptr, trajs::min_jerk_trajectory<trajs::d_order::ZERO>(com, to, g.dt, g.t_com), hold(ground[L], g.t_com),
                             hold(ground[R], g.t_com), hold(lh_at, g.t_com), hold(rh_at, g.t_com));
                        com = to;
                    };
                    // foot s leaves the floor for the top of its swing (above x_top when given), its contact goes on the first tick
                    auto raise = [&](int state, int s, const double* x_top) {
                        if (x_top) air[s][0] = *x_top;
                        Path swing = move(ground[s], air[s], g.t_foot), rest = hold(ground[1 - s], g.t_foot);
                        emit(state, kSides[s].contact, nullptr, hold(com, g.t_foot), s == L ? swing : rest, s == L ? rest : swing, hold(hand[L], g.t_com),
                             hold(hand[R], g.t_com));
                    };
                    // foot s comes down dx further ahead than it left, its hand goes dx ahead with it (over a CoM shift's time), its contact is back on the last tick
                    auto lower = [&](int state, int s, double dx, bool hand_stays_there) {
                        ground[s][0] += dx;
                        hand_ahead[s] = hand[s];
                        hand_ahead[s][0] += dx;
                        Path swing = move(air[s], ground[s], g.t_foot), rest = hold(ground[1 - s], g.t_foot);
                        Path reach = move(hand[s], hand_ahead[s], g.t_com), still = hold(hand[1 - s], g.t_com);
                        emit(state, nullptr, kSides[s].contact, hold(com, g.t_foot), s == L ? swing : rest, s == L ? rest : swing, s == L ? reach : still,
                             s == L ? still : reach);
                        if (hand_stays_there) hand[s] = hand_ahead[s];
                    };

                    shift(INIT, over[R], hand[L], hand[R]);
                    raise(LF_INIT, L, nullptr);
                    for (int cycle = 0; cycle < g.cycles; ++cycle) {
                        lower(LIFT_DOWN_LF, L, (ground[R][0] - ground[L][0]) + g.step, true); // level with the right foot, then a step ahead
                        over[L][0] = ground[L][0];
                        shift(MOVE_COM_LEFT, over[L], hand[L], hand[R]);
                        raise(LIFT_UP_RF, R, &ground[L][0]);
                        lower(LIFT_DOWN_RF, R, 2 * g.step, true);
                        over[R][0] = ground[R][0];
                        shift(MOVE_COM_RIGHT, over[R], hand[L], hand[R]);
                        raise(LIFT_UP_LF, L, &ground[R][0]);
                    }
                    lower(LIFT_DOWN_LF_FINAL, L, g.step, false);
                    const trajs::Vec between = {(ground[R][0] + ground[L][0]) / 2.0, (ground[R][1] + ground[L][1]) / 2.0, com_height};
                    shift(MOVE_COM_CENTER_FINAL, between, hand_ahead[L], hand_ahead[R]);
                }

                std::shared_ptr<controllers::PosTracker> tracker_;
                std::vector<Phase> plan_;
                int phase_ = 0, tick_ = 0;
            };
        } // namespace humanoid
    } // namespace behaviors
} // namespace inria_wbc
#endif
