// Static registrations under the reference's string keys (pos_tracker.cpp:38, humanoid_pos_tracker.cpp:35, talos_pos_tracker.cpp:35, move_com.cpp:6, cartesian.cpp:6, cartesian_traj.cpp:6, walk_on_spot.cpp:7).
#include <inria_wbc/behaviors/generic/cartesian.hpp>
#include <inria_wbc/behaviors/generic/cartesian_traj.hpp>
#include <inria_wbc/behaviors/humanoid/clapping.hpp>
#include <inria_wbc/behaviors/humanoid/move_com.hpp>
#include <inria_wbc/behaviors/humanoid/move_feet.hpp>
#include <inria_wbc/behaviors/humanoid/walk.hpp>
#include <inria_wbc/behaviors/humanoid/walk_on_spot.hpp>
#include <inria_wbc/controllers/humanoid_pos_tracker.hpp>
#include <inria_wbc/controllers/pos_tracker.hpp>

namespace inria_wbc {
    namespace controllers {
        static Register<PosTracker> __generic_pos_tracker("pos-tracker");
        static Register<HumanoidPosTracker> __humanoid_pos_tracker("humanoid-pos-tracker");
        static Register<TalosPosTracker> __talos_pos_tracking("talos-pos-tracker");
    }
    namespace behaviors {
        namespace generic {
            static Register<Cartesian> __talos_move_arm("generic::cartesian");
            static Register<CartesianTraj> __generic_cartesian_trajectory("generic::cartesian_traj");
            static Register<MoveFeet> __talos_move_feet("humanoid::move-feet"); // the reference keeps it in namespace generic (move_feet.cpp:5-6)
        }
        namespace humanoid {
            static Register<Clapping> __talos_clapping("humanoid::clapping");
            static Register<MoveCom> __talos_move_com("humanoid::move_com");
            static Register<Walk> __walk("humanoid::walk");
            static Register<WalkOnSpot> __walk_on_spot("humanoid::walk-on-spot");
        }
    } // namespace behaviors
} // namespace inria_wbc
