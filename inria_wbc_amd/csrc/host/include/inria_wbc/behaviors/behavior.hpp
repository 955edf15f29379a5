// What a behaviour is in inria_wbc (/root/reference/include/inria_wbc/behaviors/behavior.hpp:9-30, src/behaviors/behavior.cpp:7-22):
// an object that owns a controller pointer, may override task weights at construction (BEHAVIOR.customize_task_weights, e.g.
// etc/talos/walk.yaml:8-9), and on every update() pushes this tick's references into the controller and ticks it.
// Behaviours are created by name through behaviors::Factory; a plugin registers itself with `static Register<T> r("name");`.
#ifndef IWBC_HIP_BEHAVIOR_HPP
#define IWBC_HIP_BEHAVIOR_HPP

#include <iostream>
#include <map>
#include <memory>
#include <string>

#include <inria_wbc/controllers/pos_tracker.hpp>
#include <inria_wbc/utils/factory.hpp>

namespace inria_wbc {
    namespace behaviors {
        class Behavior {
        public:
            using controller_ptr_t = std::shared_ptr<controllers::Controller>;

            virtual ~Behavior() = default;
            // one control tick: set the references that belong to this instant, then controller->update(sensor_data)
            virtual void update(const controllers::SensorData& sensor_data = {}) = 0;
            virtual std::string behavior_type() const = 0;
            virtual controller_ptr_t controller() { return controller_; }
            virtual std::shared_ptr<const controllers::Controller> controller() const { return controller_; }

        protected:
            Behavior(const controller_ptr_t& controller, const yaml::Node& config) : controller_(controller)
            {
                IWBC_ASSERT(controller_, "Invalid controller pointer");
                override_weights(config["BEHAVIOR"]["customize_task_weights"]);
            }

            controller_ptr_t controller_;
            std::string behavior_type_;

        private:
            // {task: weight} from the behaviour file wins over tasks.yaml
            void override_weights(const yaml::Node& table)
            {
                if (!table) return;
                auto tracker = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                IWBC_ASSERT(tracker, "Task customization requires a controllers::PosTracker or a derivative");
                std::map<std::string, double> weights;
                for (const auto& entry : table) {
                    std::cout << "\x1B[33mWarning: overriding weight of task: " << entry.first << "\x1B[0m" << std::endl;
                    weights[entry.first] = entry.second.as<double>();
                }
                tracker->update_task_weights(weights);
            }
        };

        using Factory = utils::Factory<Behavior, Behavior::controller_ptr_t, yaml::Node>;
        template <typename T>
        using Register = Factory::AutoRegister<T>;
    } // namespace behaviors
} // namespace inria_wbc
#endif
