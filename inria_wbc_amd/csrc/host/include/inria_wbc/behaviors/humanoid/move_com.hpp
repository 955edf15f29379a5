// humanoid::move_com -- the squat of BASELINE config 4 (/root/reference/src/behaviors/humanoid/move_com.cpp:8-61,
// /root/reference/etc/talos/squat.yaml): the CoM reference visits BEHAVIOR.targets one after the other (relative to the initial
// CoM or absolute, only on the masked axes) on min-jerk segments, back to the start when looping; position, velocity and
// acceleration of the current sample go to the controller every tick.
#ifndef IWBC_HIP_MOVE_COM_HPP
#define IWBC_HIP_MOVE_COM_HPP

#include <inria_wbc/behaviors/behavior.hpp>
#include <inria_wbc/behaviors/reference_streams.hpp>

namespace inria_wbc {
    namespace behaviors {
        namespace humanoid {
            class MoveCom : public Behavior {
            public:
                MoveCom(const controller_ptr_t& controller, const yaml::Node& config) : Behavior(controller, config), stream_(controller->dt(), true)
                {
                    tracker_ = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                    IWBC_ASSERT(tracker_, "Need a PosTracker for MovCom");
                    const yaml::Node c = IWBC_CHECK(config["BEHAVIOR"]);
                    const double duration = IWBC_CHECK(c["trajectory_duration"].as<double>());
                    loop_ = IWBC_CHECK(c["loop"].as<bool>());
                    auto waypoints = IWBC_CHECK(c["targets"].as<std::vector<std::vector<double>>>());
                    const auto mask = IWBC_CHECK(c["mask"].as<std::string>());
                    const bool absolute = IWBC_CHECK(c["absolute"].as<bool>());
                    IWBC_ASSERT(mask.size() == 3, "The mask for the CoM should be 3-dimensional");
                    behavior_type_ = controllers::behavior_types::DOUBLE_SUPPORT;
                    controller_->set_behavior_type(behavior_type_);
                    const trajs::Vec home = tracker_->get_com_ref();
                    if (loop_) waypoints.push_back(absolute ? home : trajs::Vec(3, 0.0)); // the way back closes the loop
                    trajs::Vec from = home;
                    for (const auto& w : waypoints) {
                        IWBC_ASSERT(w.size() == 3, "references need to be 3-dimensional");
                        trajs::Vec to = home;
                        for (size_t axis = 0; axis < 3; ++axis)
                            if (mask[axis] == '1') to[axis] = absolute ? w[axis] : home[axis] + w[axis];
                        stream_.move(from, to, duration);
                        from = to;
                        ticks_ += stream_.length(stream_.segments() - 1);
                    }
                }
                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    tracker_->set_com_ref_tracking(stream_.sample(cursor_.segment, cursor_.tick));
                    controller_->update(sensor_data);
                    const bool at_the_end = cursor_.segment + 1 == stream_.segments() && cursor_.tick + 1 == stream_.length(cursor_.segment);
                    if (loop_ || !at_the_end) cursor_.step(stream_.length(cursor_.segment), stream_.segments(), loop_); // otherwise: stay on the last sample
                }
                std::string behavior_type() const override { return behavior_type_; }
                // jump to sample `tick` of the whole stream (harnesses start a batch in the middle of the squat)
                void set_time(int tick)
                {
                    size_t k = (size_t)tick % ticks_;
                    cursor_ = SegmentCursor();
                    while (k >= stream_.length(cursor_.segment)) k -= stream_.length(cursor_.segment++);
                    cursor_.tick = k;
                }
                size_t trajectory_size() const { return ticks_; }

            private:
                std::shared_ptr<controllers::PosTracker> tracker_;
                Vec3Stream stream_;
                SegmentCursor cursor_;
                size_t ticks_ = 0;
                bool loop_ = false;
            };
        } // namespace humanoid
    } // namespace behaviors
} // namespace inria_wbc
#endif
