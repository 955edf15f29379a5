// Batched counterpart of the reference's solver timing harness (/root/reference/src/robot_dart/qp_timer_test.cpp:15-70):
// builds a PosTracker from argv[1] and a behavior from argv[2], then loops behavior->update() with empty sensor data
// (open loop) under the "solver" timer.  Differences: B robot instances per tick (from the batch file argv[3]), and the
// last tick's torques can be written to argv[5] so that a test can compare them with the CPU oracle.
//   qp_timer_test <controller.yaml> <behavior.yaml> <batch.bin | -> [n_ticks=10] [tau_out.bin] [first_tick=0] [q_out.bin]
//   qp_timer_test --dense <dense_qp.bin> <batch.bin> <tasks.yaml> <nv> <na> [reps=500]
//       ONE QP through the two host-pointer seams of the C ABI, timed from C with the same Timer (no Python, no ctypes between two
//       calls): wbcqp_solve_dense_host on the dense QP of the first file (int64 n, neq, nin, then H, g, CE, ce0, CI, ci0 as doubles:
//       what tsid's solver stacks from the HQPData, controller.cpp:247) and wbcqp_solve_batch_host on QP 0 of the batch file with the
//       task stack of tasks.yaml.  The single-robot comparison of bench.py (dense_seam) reads the two "us" lines.
// With `-` for the batch file the controller must carry its own model (CONTROLLER.model): the rows come from the robot
// state on the device (ModelSource), the loop is closed through the integrated state, and the final q can be written too.
#include <algorithm>
#include <cmath>
#include <csignal>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include <inria_wbc/behaviors/generic/cartesian.hpp>
#include <inria_wbc/behaviors/humanoid/clapping.hpp>
#include <inria_wbc/behaviors/humanoid/move_com.hpp>
#include <inria_wbc/behaviors/humanoid/move_feet.hpp>
#include <inria_wbc/behaviors/humanoid/walk.hpp>
#include <inria_wbc/behaviors/humanoid/walk_on_spot.hpp>
#include <inria_wbc/controllers/file_source.hpp>
#include <inria_wbc/utils/timer.hpp>

static volatile sig_atomic_t stop = 0;
static void stopsig(int) { stop = 1; }

static int dense_mode(int argc, char** argv)
{
    using namespace inria_wbc;
    if (argc < 7) {
        std::cerr << "usage: " << argv[0] << " --dense <dense_qp.bin> <batch.bin> <tasks.yaml> <nv> <na> [reps]" << std::endl;
        return 2;
    }
    const int nv = std::atoi(argv[5]), na = std::atoi(argv[6]), reps = argc > 7 ? std::atoi(argv[7]) : 500;
    std::ifstream f(argv[2], std::ios::binary);
    IWBC_ASSERT(f.good(), "cannot open ", argv[2]);
    int64_t hdr[3];
    f.read(reinterpret_cast<char*>(hdr), sizeof hdr);
    const int n = (int)hdr[0], neq = (int)hdr[1], nin = (int)hdr[2];
    std::vector<double> H((size_t)n * n), g(n), CE((size_t)neq * n), ce0(neq), CI((size_t)nin * n), ci0(nin);
    auto rd = [&](std::vector<double>& a) { f.read(reinterpret_cast<char*>(a.data()), (std::streamsize)(a.size() * sizeof(double))); };
    rd(H); rd(g); rd(CE); rd(ce0); rd(CI); rd(ci0);
    IWBC_ASSERT(f.good(), "short read in ", argv[2]);
    tasks::TaskStack stack(IWBC_CHECK(yaml::LoadFile(argv[4])), nv, na);
    wbcqp_structure st = stack.c_struct();
    wbcqp_layout L;
    IWBC_ASSERT(wbcqp_layout_of(&st, &L) == WBCQP_OK, wbcqp_last_error(nullptr));
    IWBC_ASSERT(L.n == n && L.neq == neq && L.nin2 == nin, "the dense QP and the task stack disagree on the sizes");
    controllers::FileSource src(argv[3]);
    controllers::TickInputs in;
    in.resize(src.batch(), L);
    src.compute(0.0, controllers::MatrixXd(), controllers::MatrixXd(), stack, L, in);
    wbcqp_handle* h = nullptr;
    wbcqp_desc desc = {0, WBCQP_F64, 0};
    if (wbcqp_create(&desc, &h) != WBCQP_OK) IWBC_ERROR("wbcqp_create: ", wbcqp_last_error(nullptr));
    IWBC_ASSERT(wbcqp_set_structure(h, 0, &st) == WBCQP_OK, wbcqp_last_error(h));
    wbcqp_dense_inputs di = {H.data(), g.data(), CE.data(), ce0.data(), CI.data(), ci0.data()};
    wbcqp_inputs bi = {in.M.data(), in.h.data(), in.A.data(), in.b1.data(), in.Ac.data(), in.bc.data(), in.blb.data(), in.bub.data(),
                       in.tlb.data(), in.tub.data(), in.w.data(), in.Acop.data()};
    std::vector<double> x(n), tau(std::max(na, 1)), obj(1);
    int32_t status = -99, iters = 0;
    wbcqp_outputs bo = {x.data(), tau.data(), &status, &iters, obj.data(), nullptr, nullptr};
    const wbcqp_dense_output* res = nullptr;
    utils::Timer timer;
    for (int r = 0; r < 20; ++r) { // warm: allocations, page-locked staging, the first launch of each kernel
        IWBC_ASSERT(wbcqp_solve_dense_host(h, 1, n, neq, nin, 0, &di, &res) == WBCQP_OK, wbcqp_last_error(h));
        IWBC_ASSERT(wbcqp_solve_batch_host(h, 0, 1, &bi, &bo) == WBCQP_OK, wbcqp_last_error(h));
    }
    for (int r = 0; r < reps; ++r) {
        timer.begin("dense");
        wbcqp_solve_dense_host(h, 1, n, neq, nin, 0, &di, &res);
        timer.end("dense");
    }
    for (int r = 0; r < reps; ++r) {
        timer.begin("batch");
        wbcqp_solve_batch_host(h, 0, 1, &bi, &bo);
        timer.end("batch");
    }
    double dmax = 0.0;
    for (int i = 0; i < n; ++i) dmax = std::max(dmax, std::fabs(res->x[i] - x[i]));
    std::cout.precision(6);
    std::cout << "dense_host_us: " << timer["dense"].time / timer["dense"].iterations << " min " << timer["dense"].min_time << " max "
              << timer["dense"].max_time << " reps " << reps << std::endl;
    std::cout << "batch_host_us: " << timer["batch"].time / timer["batch"].iterations << " min " << timer["batch"].min_time << " max "
              << timer["batch"].max_time << " reps " << reps << std::endl;
    std::cout << "status " << status << " / " << res->status[0] << " iters " << iters << " / " << res->iters[0] << " max |dx| " << dmax << std::endl;
    const bool ok = status == 0 && res->status[0] == 0; // (res lives in the handle: read it before the handle goes)
    wbcqp_destroy(h);
    return ok ? 0 : 1;
}

int main(int argc, char** argv)
{
    using namespace inria_wbc;
    if (argc > 1 && std::string(argv[1]) == "--dense") {
        try { return dense_mode(argc, argv); }
        catch (std::exception& e) {
            std::cerr << "Exception (dense):" << e.what() << std::endl;
            return 1;
        }
    }
    if (argc < 4) {
        std::cerr << "usage: " << argv[0] << " <controller.yaml> <behavior.yaml> <batch.bin> [n_ticks] [tau_out.bin] [first_tick]" << std::endl;
        return 2;
    }
    std::signal(SIGINT, stopsig);
    try {
        const std::string ctrl_path = argv[1];
        yaml::Node c_config = IWBC_CHECK(yaml::LoadFile(ctrl_path));
        c_config["CONTROLLER"].set("base_path", ctrl_path.substr(0, ctrl_path.find_last_of('/')));
        // IWBC_SENSOR_LOOP=1: closed loop (controller.cpp:161-205) on sensor data in the reference's shape, made from the
        // controller's own integrated state: positions / joint_velocities hold the joints only, the base travels in
        // floating_base_position / floating_base_velocity.  IWBC_SENSOR_LOOP=missing: closed loop with no sensor data at all.
        const char* sensor_loop = std::getenv("IWBC_SENSOR_LOOP");
        if (sensor_loop) c_config["CONTROLLER"].set("closed_loop", "true");
        auto controller_name = IWBC_CHECK(c_config["CONTROLLER"]["name"].as<std::string>());
        if (const char* as = std::getenv("IWBC_CONTROLLER_NAME")) controller_name = as; // e.g. talos-pos-tracker on the same file
        auto controller = controllers::Factory::instance().create(controller_name, c_config);
        if (std::string(argv[3]) != "-") controller->set_problem_source(std::make_shared<controllers::FileSource>(argv[3]));

        yaml::Node b_config = IWBC_CHECK(yaml::LoadFile(argv[2]));
        auto behavior_name = IWBC_CHECK(b_config["BEHAVIOR"]["name"].as<std::string>());
        auto behavior = behaviors::Factory::instance().create(behavior_name, controller, b_config);
        const int n_ticks = argc > 4 ? std::atoi(argv[4]) : 10;
        if (argc > 6)
            if (auto mc = std::dynamic_pointer_cast<behaviors::humanoid::MoveCom>(behavior)) mc->set_time(std::atoi(argv[6]));

        utils::Timer timer;
        int it = 0;
        while (!stop && it < n_ticks) {
            controllers::SensorData sensors;
            if (sensor_loop && std::string(sensor_loop) == "1") {
                const auto &q = controller->q_tsid(), &v = controller->dq();
                const int B = q.rows, nq = q.cols, nvv = v.cols, fb = (nq == nvv + 1) ? 7 : 0, fbv = fb ? 6 : 0;
                controllers::MatrixXd pos(B, nq - fb), vel(B, nvv - fbv), fpos(B, fb), fvel(B, fbv);
                for (int i = 0; i < B; ++i) {
                    std::copy(q.row(i), q.row(i) + fb, fpos.row(i));
                    std::copy(q.row(i) + fb, q.row(i) + nq, pos.row(i));
                    std::copy(v.row(i), v.row(i) + fbv, fvel.row(i));
                    std::copy(v.row(i) + fbv, v.row(i) + nvv, vel.row(i));
                }
                sensors["positions"] = pos;
                sensors["joint_velocities"] = vel;
                if (fb) {
                    sensors["floating_base_position"] = fpos;
                    sensors["floating_base_velocity"] = fvel;
                }
            }
            timer.begin("solver");
            behavior->update(sensors);
            timer.end("solver");
            timer.report(std::cout, it++, 1);
        }
        std::cout << "instances per tick: " << controller->batch_size() << std::endl;
        if (auto pt = std::dynamic_pointer_cast<controllers::PosTracker>(controller))
            if (pt->has_task("com")) {
                std::cout.precision(17);
                std::cout << "cost com: " << pt->cost("com") << std::endl; // |A ddq - b| of the CoM rows, last tick (controller.hpp:148-152)
            }
        {   // what a robot-side consumer reads each tick (reference: examples and robot_dart glue): the filtered command and momentum()
            const auto tau_cmd = controller->tau(), q_cmd = controller->q();
            std::cout << "command columns: " << tau_cmd.cols << " of " << controller->tau(false).cols << " dofs ("
                      << controller->mimic_names().size() << " mimic joints filtered)" << std::endl;
            const auto& mom = controller->momentum();
            if (mom.rows > 0) {
                std::cout.precision(17);
                std::cout << "momentum[0]: " << mom(0, 0) << " " << mom(0, 1) << " " << mom(0, 2) << std::endl;
            }
            if (const char* fp = std::getenv("IWBC_DUMP_COMMAND")) {
                std::ofstream f(fp, std::ios::binary);
                f.write(reinterpret_cast<const char*>(tau_cmd.data.data()), (std::streamsize)(tau_cmd.data.size() * sizeof(double)));
                f.write(reinterpret_cast<const char*>(q_cmd.data.data()), (std::streamsize)(q_cmd.data.size() * sizeof(double)));
            }
            if (std::getenv("IWBC_STEP_BACK")) { // qp_step_back(): the next tick starts from the state this one started from
                const auto q_before = controller->q_tsid();
                controller->qp_step_back();
                const auto& q_after = controller->q_tsid();
                double moved = 0.0;
                for (size_t i = 0; i < q_after.data.size(); ++i) moved = std::max(moved, std::fabs(q_after.data[i] - q_before.data[i]));
                controller->update(controllers::SensorData{}); // same references as the tick just undone (the behavior is not advanced)
                const auto& q_redo = controller->q_tsid();
                double diff = 0.0;
                for (size_t i = 0; i < q_redo.data.size(); ++i) diff = std::max(diff, std::fabs(q_redo.data[i] - q_before.data[i]));
                std::cout << "step back moved q by " << moved << ", redoing the tick differs from the first time by " << diff << std::endl;
            }
        }
        if (argc > 5) {
            std::ofstream f(argv[5], std::ios::binary);
            const auto& tau = controller->tau_tsid(); // na entries per instance (tau() pads a floating base with six zeros)
            f.write(reinterpret_cast<const char*>(tau.data.data()), (std::streamsize)(tau.data.size() * sizeof(double)));
        }
        if (argc > 7) {
            std::ofstream f(argv[7], std::ios::binary);
            const auto& q = controller->q_tsid(); // quaternion form, nq entries (q() is the reference's angle-axis form)
            f.write(reinterpret_cast<const char*>(q.data.data()), (std::streamsize)(q.data.size() * sizeof(double)));
        }
    }
    catch (std::exception& e) {
        std::cerr << "Exception (solver):" << e.what() << std::endl;
        return 1;
    }
    return 0;
}
