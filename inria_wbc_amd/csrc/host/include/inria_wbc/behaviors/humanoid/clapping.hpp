// humanoid::clapping -- the hands go motion_size towards each other along y and back, for ever
// (/root/reference/src/behaviors/humanoid/clapping.cpp:8-52, /root/reference/etc/talos/clapping.yaml).  Pose-only references.
#ifndef IWBC_HIP_CLAPPING_HPP
#define IWBC_HIP_CLAPPING_HPP

#include <inria_wbc/behaviors/behavior.hpp>
#include <inria_wbc/behaviors/reference_streams.hpp>

namespace inria_wbc {
    namespace behaviors {
        namespace humanoid {
            class Clapping : public Behavior {
            public:
                Clapping(const controller_ptr_t& controller, const yaml::Node& config)
                    : Behavior(controller, config), lh_(controller->dt(), false), rh_(controller->dt(), false)
                {
                    tracker_ = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                    IWBC_ASSERT(tracker_, "we need a pos tracker here");
                    const yaml::Node c = IWBC_CHECK(config["BEHAVIOR"]);
                    const double duration = IWBC_CHECK(c["trajectory_duration"].as<double>());
                    const double size = IWBC_CHECK(c["motion_size"].as<double>());
                    behavior_type_ = controllers::behavior_types::DOUBLE_SUPPORT;
                    controller_->set_behavior_type(behavior_type_);
                    const trajs::Vec lh_open = tracker_->get_se3_ref("lh"), rh_open = tracker_->get_se3_ref("rh");
                    const trajs::Vec lh_closed = displaced(lh_open, {0.0, -size, 0.0}, {}), rh_closed = displaced(rh_open, {0.0, size, 0.0}, {});
                    lh_.move(lh_open, lh_closed, duration); lh_.move(lh_closed, lh_open, duration);
                    rh_.move(rh_open, rh_closed, duration); rh_.move(rh_closed, rh_open, duration);
                }
                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    tracker_->set_se3_ref(lh_.sample(cursor_.segment, cursor_.tick), "lh");
                    tracker_->set_se3_ref(rh_.sample(cursor_.segment, cursor_.tick), "rh");
                    controller_->update(sensor_data);
                    cursor_.step(lh_.length(cursor_.segment), lh_.segments(), true);
                }
                std::string behavior_type() const override { return behavior_type_; }

            private:
                std::shared_ptr<controllers::PosTracker> tracker_;
                Se3Stream lh_, rh_;
                SegmentCursor cursor_;
            };
        } // namespace humanoid
    } // namespace behaviors
} // namespace inria_wbc
#endif
