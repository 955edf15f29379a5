// CPU-only checks of the host facade, in the spirit of the reference's utest programs
// (/root/reference/tests/utest.hpp:154-212, test_all_robots.cpp:135-229): factories, config parsing, task stacks,
// error behaviour of the solver switch.  Exit code 0 = all checks passed.
#include <cmath>
#include <iostream>

#include <inria_wbc/behaviors/humanoid/move_com.hpp>
#include <inria_wbc/controllers/pos_tracker.hpp>
#include <inria_wbc/robots/robot_wrapper.hpp>
#include <inria_wbc/trajs/loader.hpp>
#include <inria_wbc/utils/timer.hpp>
#include <cstdio>
#include <fstream>

static int failures = 0;
#define UTEST_CHECK(cond)                                                              \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            ++failures;                                                                \
            std::cerr << "CHECK FAILED " << __FILE__ << ":" << __LINE__ << " " #cond << std::endl; \
        }                                                                              \
    } while (0)
#define UTEST_CHECK_EXCEPTION(expr, needle)                                            \
    do {                                                                               \
        bool thrown_ = false;                                                          \
        try { expr; }                                                                  \
        catch (const inria_wbc::Exception& e) { thrown_ = std::string(e.what()).find(needle) != std::string::npos; \
            if (!thrown_) std::cerr << "unexpected message: " << e.what() << std::endl; } \
        UTEST_CHECK(thrown_);                                                          \
    } while (0)

using namespace inria_wbc;

int main(int argc, char** argv)
{
    const std::string cfg = argc > 1 ? argv[1] : "configs";
    // ---- task stacks: sizes of SURVEY.md Appendix B ----
    struct Exp { const char* robot; int nv, na, n, neq, nin, r1, dense; };
    const Exp exps[] = {{"talos", 50, 44, 74, 18, 122, 97, 41}, {"icub", 38, 32, 62, 18, 66, 83, 39}, {"franka", 9, 9, 9, 0, 0, 15, 6}, {"tiago", 12, 12, 12, 0, 12, 25, 13}};
    for (const auto& e : exps) {
        yaml::Node tasks = yaml::LoadFile(cfg + "/" + e.robot + "/tasks.yaml");
        tasks::TaskStack st(tasks, e.nv, e.na);
        UTEST_CHECK(st.nVar() == e.n && st.nEq() == e.neq && st.nIn() == e.nin && st.level1_rows() == e.r1 && st.n_dense() == e.dense);
        wbcqp_structure s = st.c_struct();
        wbcqp_layout L;
        UTEST_CHECK(wbcqp_layout_of(&s, &L) == WBCQP_OK);
        UTEST_CHECK(L.n == e.n && L.neq == e.neq && L.nin == e.nin && L.nin2 == 2 * e.nin && L.r1 == e.r1);
    }
    {
        yaml::Node tasks = yaml::LoadFile(cfg + "/talos/tasks.yaml");
        tasks::TaskStack st(tasks, 50, 44);
        UTEST_CHECK(st.task("com").kp == 30.0 && std::fabs(st.task("com").kd - 2.0 * std::sqrt(30.0)) < 1e-12);
        UTEST_CHECK(st.task("self_collision-left").kd == 250.0 && st.task("posture").weight == 1.75);
        // single support: n 62, nEq 12, nIn 105
        tasks::TaskStack ss = st.without_contact("contact_lfoot");
        UTEST_CHECK(ss.nVar() == 62 && ss.nEq() == 12 && ss.nIn() == 105 && ss.level1_rows() == 91);
        UTEST_CHECK_EXCEPTION(st.task("nope"), "not found");
    }
    // ---- the task types no shipped stack uses (tasks.cpp:156-178 cop, :227-271 torque) and the posture mask (:205-214) ----
    {
        yaml::Node tasks = yaml::LoadFile(cfg + "/talos/tasks.yaml");
        std::string mask(44, '1'), pmask(44, '1');
        for (int j = 12; j < 44; ++j) mask[j] = '0'; // the legs' torques only
        pmask[3] = pmask[40] = '0';
        yaml::Node tq = yaml::Load("type: torque\nweight: 0.01\nmask: " + mask + "\n");
        yaml::Node cop = yaml::Load("type: cop\nweight: 10.0\n");
        tasks.set("torque", tq);
        tasks.set("cop", cop);
        yaml::Node post = tasks["posture"];
        post.set("mask", pmask);
        tasks.set("posture", post);
        tasks::TaskStack st(tasks, 50, 44);
        // tsid's nVar / nEq / nIn do not move (level-1 tasks); level-1 rows: 97 - 2 (posture mask) + 12 (torque) + 3 (cop)
        UTEST_CHECK(st.nVar() == 74 && st.nEq() == 18 && st.nIn() == 122 && st.n_acteq() == 12 && st.has_cop() && st.n_sel() == 42);
        UTEST_CHECK(st.level1_rows() == 97 - 2 + 12 + 3 && st.task("torque").rows == 12 && st.task("cop").rows == 3);
        wbcqp_structure s = st.c_struct();
        wbcqp_layout L;
        UTEST_CHECK(wbcqp_layout_of(&s, &L) == WBCQP_OK);
        UTEST_CHECK(L.r1 == st.level1_rows() && L.len_b1 == L.r1 && L.dense_h == 1 && L.len_Acop == 72 && L.waves_per_cu == 1);
        UTEST_CHECK(s.acteq_joint[11] == 11 && s.acteq_scale[0] == 1.0 && s.cop_task == st.task("cop").weight_index);
        // scaling: one entry per actuated joint
        yaml::Node tq2 = yaml::Load("type: torque\nweight: 1\nscaling: [1, 2, 3]\n");
        yaml::Node t2 = yaml::LoadFile(cfg + "/talos/tasks.yaml");
        t2.set("torque", tq2);
        UTEST_CHECK_EXCEPTION(tasks::TaskStack(t2, 50, 44), "wrong size in torque scaling");
        post.set("mask", "101");
        t2 = yaml::LoadFile(cfg + "/talos/tasks.yaml");
        t2.set("posture", post);
        UTEST_CHECK_EXCEPTION(tasks::TaskStack(t2, 50, 44), "wrong size in posture mask");
    }
    // a cop task without a contact, a second torque task, an unknown type and an unknown KEY are refused with the task's name
    UTEST_CHECK_EXCEPTION(tasks::TaskStack(yaml::Load("t:\n  type: cop\n  weight: 1\n"), 9, 9), "needs a contact");
    UTEST_CHECK_EXCEPTION(tasks::TaskStack(yaml::Load("a:\n  type: torque\n  weight: 1\nb:\n  type: torque\n  weight: 1\n"), 9, 9), "one torque task");
    UTEST_CHECK_EXCEPTION(tasks::TaskStack(yaml::Load("t:\n  type: banana\n  weight: 1\n"), 9, 9), "is not registered");
    UTEST_CHECK_EXCEPTION(tasks::TaskStack(yaml::Load("t:\n  type: posture\n  weight: 1\n  kp: 10\n  ref: x\n  maskk: 111111111\n"), 9, 9), "unknown key [maskk]");
    UTEST_CHECK_EXCEPTION(tasks::TaskStack(yaml::Load("t:\n  type: se3\n  weight: 1\n  mask: 11\n"), 9, 9), "mask");
    // ---- yaml subset ----
    {
        yaml::Node n = yaml::Load("BEHAVIOR:\n  name: humanoid::move_com # comment\n  targets: [[0, 0, -0.2], [1, 2, 3]]\n  loop: true\n  mask: 001\n");
        UTEST_CHECK(n["BEHAVIOR"]["name"].as<std::string>() == "humanoid::move_com");
        auto t = n["BEHAVIOR"]["targets"].as<std::vector<std::vector<double>>>();
        UTEST_CHECK(t.size() == 2 && t[0][2] == -0.2 && t[1][1] == 2.0);
        UTEST_CHECK(n["BEHAVIOR"]["loop"].as<bool>() && n["BEHAVIOR"]["mask"].as<std::string>() == "001");
        UTEST_CHECK(!n["BEHAVIOR"]["missing"]);
        UTEST_CHECK_EXCEPTION(IWBC_CHECK(n["BEHAVIOR"]["missing"].as<double>()), "when calling");
    }
    // ---- factories: unknown name lists the known ones; known names are registered ----
    UTEST_CHECK(controllers::Factory::instance().has("pos-tracker"));
    // the reference's two humanoid controllers load under their own names (humanoid_pos_tracker.cpp:35, talos_pos_tracker.cpp:35)
    UTEST_CHECK(controllers::Factory::instance().has("humanoid-pos-tracker") && controllers::Factory::instance().has("talos-pos-tracker"));
    UTEST_CHECK(behaviors::Factory::instance().has("humanoid::move_com"));
    UTEST_CHECK(behaviors::Factory::instance().has("generic::cartesian") && behaviors::Factory::instance().has("generic::cartesian_traj"));
    UTEST_CHECK(behaviors::Factory::instance().has("humanoid::walk-on-spot") && behaviors::Factory::instance().has("humanoid::move-feet") &&
                behaviors::Factory::instance().has("humanoid::clapping") && behaviors::Factory::instance().has("humanoid::walk"));
    UTEST_CHECK_EXCEPTION(controllers::Factory::instance().create("no-such-controller", yaml::Node()), "is not in the factory");
    const bool stacks_only = argc > 3 && std::string(argv[3]) == "stacks-only"; // a directory with task stacks only
    // ---- solver switch (pos_tracker.cpp:88-100) ----
    if (!stacks_only) {
        yaml::Node c = yaml::LoadFile(cfg + "/talos/pos_tracker.yaml");
        c["CONTROLLER"].set("base_path", cfg + "/talos");
        c["CONTROLLER"].set("solver", "banana");
        UTEST_CHECK_EXCEPTION(controllers::Factory::instance().create("pos-tracker", c), "must be either");
        c["CONTROLLER"].set("solver", "eiquadprog");
        UTEST_CHECK_EXCEPTION(controllers::Factory::instance().create("pos-tracker", c), "not available");
        UTEST_CHECK_EXCEPTION(controllers::Factory::instance().create("talos-pos-tracker", c), "not available");
    }
    // ---- min jerk ----
    {
        auto p = trajs::min_jerk_trajectory<trajs::d_order::ZERO>({0, 0, 0.9}, {0, 0, 0.7}, 1e-3, 2.0);
        auto v = trajs::min_jerk_trajectory<trajs::d_order::FIRST>({0, 0, 0.9}, {0, 0, 0.7}, 1e-3, 2.0);
        UTEST_CHECK(p.size() == 2000 && std::fabs(p[1000][2] - 0.8) < 1e-12 && p[0][2] == 0.9);
        UTEST_CHECK(std::fabs(v[1000][2] - (-0.2 * 30.0 / 16.0 / 2.0)) < 1e-12);
    }
    // ---- trajectory files (loader.cpp:11-88): SE3 = 3 + 9 numbers, rotation column-major; line breaks carry no meaning ----
    {
        const std::string dir = argc > 2 ? std::string(argv[2]) : std::string("/tmp");
        {
            std::ofstream(dir + "/lh.csv") << "0.1 0.2 0.3  1 0 0 0 0 1 0 -1 0\n0.4 0.5 0.6\n 0 1 0 -1 0 0 0 0 1\n";
            std::ofstream(dir + "/rh.csv") << "0 0 0 1 0 0 0 1 0 0 0 1 1 1 1 1 0 0 0 1 0 0 0 1";
            std::ofstream(dir + "/com.csv") << "0 0 0.9\n0 0 0.8\n";
            std::ofstream(dir + "/q.csv") << "1 2 3 4\n5 6 7 8\n";
            std::ofstream(dir + "/refs.yaml") << "refs:\n  lh: lh.csv\n  rh: rh.csv\n  com: com.csv\n  posture:\n    posture: q.csv\n    size: 4\n";
            std::ofstream(dir + "/short.csv") << "0 0 0 1 0 0 0 1 0 0 0 1";
            std::ofstream(dir + "/bad.yaml") << "refs:\n  lh: lh.csv\n  rh: short.csv\n";
        }
        trajs::Loader ld(dir + "/refs.yaml");
        UTEST_CHECK(ld.size() == 2 && ld.size_vec() == 2 && ld.has_com_refs() && ld.ref_names().size() == 2);
        UTEST_CHECK(ld.ref_names_vec().size() == 1 && ld.ref_names_vec()[0] == "posture" && ld.task_ref_vec("posture", 1)[2] == 7.0);
        UTEST_CHECK(ld.task_ref("lh", 1).translation[1] == 0.5 && ld.com_ref(1)[2] == 0.8);
        // sample 0 of lh: columns (1,0,0), (0,0,1), (0,-1,0) => R(2,1) = 1, R(1,2) = -1
        UTEST_CHECK(ld.task_ref("lh", 0).R(2, 1) == 1.0 && ld.task_ref("lh", 0).R(1, 2) == -1.0 && ld.task_ref("lh", 0).R(0, 0) == 1.0);
        UTEST_CHECK(ld.task_ref("rh", 1).translation[0] == 1.0);
        UTEST_CHECK_EXCEPTION(trajs::Loader(dir + "/bad.yaml"), "wrong number of rows");
        UTEST_CHECK_EXCEPTION(trajs::Loader(dir + "/missing.yaml"), "");
    }
    // ---- timer ----
    {
        utils::Timer timer;
        timer.begin("solver");
        timer.end("solver");
        UTEST_CHECK(timer["solver"].iterations == 1 && timer["solver"].min_time <= timer["solver"].max_time);
    }
    // ---- the robot as data: tree (stand-in for the URDF), virtual frames in the reference's frames.yaml schema, host FK ----
    // (skipped with a third argument "stacks-only": a directory that holds task stacks but none of this repository's model files)
    if (!stacks_only) {
        robots::RobotWrapper robot(cfg + "/talos/talos_like.model.yaml");
        UTEST_CHECK(robot.nq() == 51 && robot.nv() == 50 && robot.na() == 44 && robot.floating_base());
        UTEST_CHECK(robot.existJointName("leg_left_6_joint") && robot.existFrame("torso_2_link") && !robot.existFrame("v_leg_left_3"));
        const int before = robot.nframes();
        yaml::Node fr = yaml::LoadFile(cfg + "/talos/frames.yaml");
        for (const auto& kv : fr) {
            auto pos = kv.second["pos"].as<std::vector<double>>();
            robot.addFrame(kv.first, kv.second["ref"].as<std::string>(), {{pos[0], pos[1], pos[2]}});
        }
        UTEST_CHECK(robot.nframes() == before + 4 && robot.existFrame("v_base_link_right"));
        UTEST_CHECK_EXCEPTION(robot.getFrameId("nope"), "Unknown frame or joint");
        const auto& q0 = robot.referenceConfigurations().at("inria_start");
        auto base = robot.framePosition(q0.data(), robot.getFrameId("root_joint"));
        UTEST_CHECK(base.p[0] == q0[0] && base.p[1] == q0[1] && base.p[2] == q0[2]);
        // a virtual frame sits `pos` away from its reference frame, in that frame's axes
        auto a = robot.framePosition(q0.data(), robot.getFrameId("base_link"));
        auto b = robot.framePosition(q0.data(), robot.getFrameId("v_base_link_left"));
        double d2 = 0.0;
        for (int k = 0; k < 3; ++k) d2 += (a.p[k] - b.p[k]) * (a.p[k] - b.p[k]);
        UTEST_CHECK(std::fabs(std::sqrt(d2) - 0.1) < 1e-12);
        auto c = robot.com(q0.data());
        UTEST_CHECK(c[2] > 0.8 && c[2] < 1.0 && std::fabs(c[1]) < 0.01);
        // feet flat and level at the reference posture
        auto lf = robot.framePosition(q0.data(), robot.getFrameId("leg_left_6_joint"));
        auto rf = robot.framePosition(q0.data(), robot.getFrameId("leg_right_6_joint"));
        UTEST_CHECK(std::fabs(lf.p[2] - rf.p[2]) < 1e-6 && lf.R[8] > 0.999999);
        wbcqp_model m = robot.c_model();
        UTEST_CHECK(m.nbody == 45 && m.nframe == robot.nframes() && m.parent[0] == -1 && m.jtype[0] == WBCQP_J_FREEFLYER);
        auto v = base.to_vector();
        UTEST_CHECK(v[0] == q0[0] && v[3] == base.R[0] && v[4] == base.R[3] && v[6] == base.R[1]); // rotation column-major
    }
    std::cout << (failures ? "FAILED" : "OK") << " (" << failures << " failures)" << std::endl;
    return failures ? 1 : 0;
}
