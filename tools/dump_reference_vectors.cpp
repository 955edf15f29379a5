// dump_reference_vectors.cpp -- makes "parity pinned" a one-command job on a machine that HAS the reference's solver stack.
//
// This repository's oracle (oracle/wbc_oracle.c, oracle/rbd_oracle.c) restates algorithms that live in third-party libraries
// the reference neither vendors nor pins (tsid, eiquadprog, pinocchio: /root/reference/docs/installation.md:37-76) and that
// do not exist in the build image -- so every parity claim here is against a restatement ("parity unpinned", DESIGN.md
// section 2).  This program is written against the reference's REAL API and, linked against an installed inria_wbc + tsid,
// writes for N control ticks of a shipped controller / behavior pair
//
//     (q, v)  ->  (M, h, every level-0 / level-1 constraint of tsid's HQPData)  ->  (x, tau, status, iterations)
//
// as .npy files; tools/pack_reference_vectors.py turns one such directory into tests/golden/reference/<name>.npz in the schema
// of tests/golden/*.npz, and tests/test_reference_vectors.py then checks the oracle and the HIP path against THOSE numbers
// (it skips, loudly, while no such file exists).
//
// NOT BUILT HERE (no Eigen / tsid / pinocchio / yaml-cpp in this image; syntax-checked against declaration stubs, see below).  It follows the reference's own harness
// (/root/reference/src/robot_dart/qp_timer_test.cpp:15-70) and calls, per tick, exactly what Controller::_solve calls
// (/root/reference/src/controllers/controller.cpp:231-313):
//     tsid_->computeProblemData(t, q, dq)      controller.cpp:244
//     solver_->solve(HQPData)                  controller.cpp:247
//     tsid_->getActuatorForces / getAccelerations   controller.cpp:250-251
// on its OWN solver instance and at the controller's CURRENT state, before letting behavior->update() advance the
// controller; the two must agree (checked: max |ddq_dump - controller->ddq(false)|).
//
// Build (on a machine with the reference installed as docs/installation.md describes):
//     g++ -std=c++14 -O2 tools/dump_reference_vectors.cpp -o dump_reference_vectors
//         $(pkg-config --cflags --libs tsid pinocchio eigen3 yaml-cpp) -linria_wbc -lboost_system -lboost_filesystem      (one command line)
// Compile check in THIS image (declarations only, tools/stubs/README.md; run by tests/test_host_cpp.py):
//     g++ -std=c++14 -fsyntax-only -Wall -I tools/stubs tools/dump_reference_vectors.cpp
// Run:
//     ./dump_reference_vectors etc/talos/talos_pos_tracker.yaml etc/talos/squat.yaml out_dir/talos_squat 200
//     ./dump_reference_vectors etc/franka/pos_tracker.yaml etc/franka/cartesian_line.yaml out_dir/franka_line 200
//     ./dump_reference_vectors etc/icub/humanoid_pos_tracker.yaml etc/icub/squat.yaml out_dir/icub_squat 200
//     python tools/pack_reference_vectors.py out_dir/talos_squat talos tests/golden/reference/talos_squat.npz
//
// First things to look at in the output (the [UPSTREAM-RECALL] items of oracle/rbd_oracle.c, docs/HISTORY.md section 6c): the
// program also writes, per tick, what those four recalled pieces produce upstream so that the packer can compare them with
// the oracle's versions one by one:
//   recall_se3_<task>.npy      the SE(3) task's position error (errorInSE3: translation + log3 of M^-1 M_ref) and its constraint b
//   recall_jac_<frame>.npy     robot->frameJacobianWorld of every self-collision frame (pinocchio WORLD, origin dependent)
//   recall_bounds.npy          TaskJointPosVelAccBounds' lower / upper acceleration limits (computeAccLimits) at (q, v)
//   recall_momentum.npy        the centroidal momentum and its drift term as TaskMEquality used them
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <sys/stat.h>

#include <Eigen/Core>

#include <tsid/formulations/inverse-dynamics-formulation-acc-force.hpp>
#include <tsid/math/constraint-base.hpp>
#include <tsid/solvers/solver-HQP-base.hpp>
#include <tsid/solvers/solver-HQP-factory.hxx>
#include <tsid/solvers/utils.hpp>

#include "inria_wbc/behaviors/behavior.hpp"
#include "inria_wbc/controllers/pos_tracker.hpp"
#include "inria_wbc/exceptions.hpp"

namespace {

// ---- minimal .npy (format 1.0, little-endian f8 / i4, C order) ----
void write_npy(const std::string& path, const char* descr, size_t itemsize, const void* data, const std::vector<size_t>& shape)
{
    std::ostringstream hs;
    hs << "{'descr': '" << descr << "', 'fortran_order': False, 'shape': (";
    size_t count = 1;
    for (size_t i = 0; i < shape.size(); ++i) {
        hs << shape[i] << (shape.size() == 1 ? "," : (i + 1 < shape.size() ? ", " : ""));
        count *= shape[i];
    }
    hs << "), }";
    std::string header = hs.str();
    const size_t unpadded = 10 + header.size() + 1;
    header += std::string((64 - unpadded % 64) % 64, ' ');
    header += '\n';
    std::ofstream f(path, std::ios::binary);
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    f.write(reinterpret_cast<const char*>(magic), 8);
    const unsigned short hl = static_cast<unsigned short>(header.size());
    f.write(reinterpret_cast<const char*>(&hl), 2);
    f.write(header.data(), static_cast<std::streamsize>(header.size()));
    f.write(reinterpret_cast<const char*>(data), static_cast<std::streamsize>(count * itemsize));
}

void dump_matrix(const std::string& path, const Eigen::MatrixXd& m)
{
    // Eigen is column-major: hand numpy a row-major copy
    Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> r = m;
    write_npy(path, "<f8", 8, r.data(), {static_cast<size_t>(r.rows()), static_cast<size_t>(r.cols())});
}

void dump_vector(const std::string& path, const Eigen::VectorXd& v)
{
    write_npy(path, "<f8", 8, v.data(), {static_cast<size_t>(v.size())});
}

void dump_ints(const std::string& path, const std::vector<int>& v)
{
    write_npy(path, "<i4", 4, v.data(), {v.size()});
}

std::string tick_dir(const std::string& root, int k)
{
    std::ostringstream s;
    s << root << "/tick" << std::setw(5) << std::setfill('0') << k;
    mkdir(s.str().c_str(), 0755);
    return s.str();
}

std::string clean(std::string name)
{
    for (auto& ch : name)
        if (ch == '/' || ch == ' ') ch = '_';
    return name;
}

} // namespace

int main(int argc, char** argv)
{
    if (argc < 5) {
        std::cerr << "usage: " << argv[0] << " <controller.yaml> <behavior.yaml> <out_dir> <n_ticks> [stride=1]" << std::endl;
        return 2;
    }
    try {
        const std::string out = argv[3];
        const int n_ticks = std::atoi(argv[4]);
        const int stride = argc > 5 ? std::atoi(argv[5]) : 1;
        mkdir(out.c_str(), 0755);

        auto controller_config = IWBC_CHECK(YAML::LoadFile(argv[1]));
        // the plain tick: no stabiliser, no torque safety (they act on references between ticks, not on the QP of a tick)
        auto controller = std::make_shared<inria_wbc::controllers::PosTracker>(controller_config);
        auto behavior_config = IWBC_CHECK(YAML::LoadFile(argv[2]));
        auto behavior_name = IWBC_CHECK(behavior_config["BEHAVIOR"]["name"].as<std::string>());
        auto behavior = inria_wbc::behaviors::Factory::instance().create(behavior_name, controller, behavior_config);

        auto tsid = controller->tsid();
        auto robot = controller->robot();
        // our own solver of the kind the controller uses (pos_tracker.cpp:88-102)
        auto solver = tsid::solvers::SolverHQPFactory::createNewSolver(tsid::solvers::SOLVER_HQP_EIQUADPROG_FAST, "dump-solver");
        solver->resize(tsid->nVar(), tsid->nEq(), tsid->nIn());

        {   // sizes, once: what pos_tracker.cpp:150-158 prints under `verbose`
            std::ofstream f(out + "/sizes.txt");
            f << "nq " << robot->nq() << "\nnv " << robot->nv() << "\nna " << robot->na() << "\nnVar " << tsid->nVar() << "\nnEq "
              << tsid->nEq() << "\nnIn " << tsid->nIn() << "\ndt " << controller->dt() << "\n";
        }

        double worst = 0.0;
        for (int k = 0; k < n_ticks; ++k) {
            // the state the coming tick starts from (open loop: the controller's own integrated state, controller.cpp:203-205)
            const Eigen::VectorXd q = controller->q_tsid();
            const Eigen::VectorXd v = controller->dq(false); // dq_ = v_tsid_ (controller.cpp:279)
            const double t = controller->t();

            // behavior->update() = set the references of this tick, then Controller::update -> _solve(q, v).  Let it run
            // FIRST: the references it sets are what the tasks use; the state (q, v, t) above is what it solves at.
            behavior->update();
            const Eigen::VectorXd ddq_ctrl = controller->ddq(false);

            if (k % stride != 0) continue;
            // the same tick again on our own solver, with everything in between kept for the dump: references are unchanged
            // (the behavior set them above and nothing has touched them since), so this is the QP the controller just solved
            const tsid::solvers::HQPData& hqp = tsid->computeProblemData(t, q, v);
            const tsid::solvers::HQPOutput& sol = solver->solve(hqp);
            const Eigen::VectorXd tau = tsid->getActuatorForces(sol);
            const Eigen::VectorXd dv = tsid->getAccelerations(sol);
            worst = std::max(worst, (dv - ddq_ctrl).cwiseAbs().maxCoeff());

            const std::string d = tick_dir(out, k);
            dump_vector(d + "/q.npy", q);
            dump_vector(d + "/v.npy", v);
            dump_vector(d + "/x.npy", sol.x);
            dump_vector(d + "/tau.npy", tau);
            dump_vector(d + "/dv.npy", dv);
            dump_ints(d + "/status_iters.npy", {static_cast<int>(sol.status), static_cast<int>(sol.iterations)});
            // the rest of HQPOutput ("identical active set"; pos_tracker.hpp:44 getObjectiveValue): written as tsid hands them out --
            // [UPSTREAM-RECALL] SolverHQuadProgFast::solve copies eiquadprog's getActiveSet() / getLagrangeMultipliers() (entries: equality i tagged
            // -i-1, otherwise the one-sided CI row; which slice of them lands in HQPOutput is for the packer to find out from the numbers)
            dump_ints(d + "/active_set.npy", std::vector<int>(sol.activeSet.data(), sol.activeSet.data() + sol.activeSet.size()));
            dump_vector(d + "/lambda.npy", sol.lambda);
            {
                Eigen::VectorXd fv(1);
                fv << solver->getObjectiveValue();
                dump_vector(d + "/objective.npy", fv);
            }
            dump_matrix(d + "/M.npy", robot->mass(tsid->data()));
            dump_vector(d + "/h.npy", robot->nonLinearEffects(tsid->data()));
            // every constraint of every level, in tsid's own order: that order IS the row order of CE / CI / the level-1 sum
            std::ofstream index(d + "/hqp_index.txt");
            for (size_t level = 0; level < hqp.size(); ++level) {
                int idx = 0;
                for (const auto& wc : hqp[level]) {
                    const double weight = wc.first;
                    const auto& c = wc.second;
                    std::ostringstream stem;
                    stem << "L" << level << "_" << std::setw(3) << std::setfill('0') << idx++ << "_" << clean(c->name());
                    const char* kind = c->isEquality() ? "eq" : (c->isInequality() ? "ineq" : "bound");
                    index << stem.str() << " " << kind << " " << c->rows() << " " << c->cols() << " " << std::setprecision(17) << weight << "\n";
                    if (!c->isBound()) dump_matrix(d + "/" + stem.str() + "_A.npy", c->matrix());
                    if (c->isEquality()) dump_vector(d + "/" + stem.str() + "_b.npy", c->vector());
                    else {
                        dump_vector(d + "/" + stem.str() + "_lb.npy", c->lowerBound());
                        dump_vector(d + "/" + stem.str() + "_ub.npy", c->upperBound());
                    }
                }
            }
            // ---- the four recalled pieces (see the header).  Accessors are the ones PosTracker exposes (pos_tracker.hpp:44-70).
            {
                const auto& data = tsid->data();
                std::ofstream rc(d + "/recall_index.txt");
                for (const auto& name : {"lh", "rh", "lf", "rf", "torso", "head", "ee"}) {
                    if (!controller->has_task(name)) continue;
                    auto task = controller->se3_task(name);
                    Eigen::VectorXd rec(task->position_error().size() + task->getConstraint().vector().size());
                    rec << task->position_error(), task->getConstraint().vector();
                    dump_vector(d + "/recall_se3_" + std::string(name) + ".npy", rec);
                    rc << "se3 " << name << " " << task->position_error().size() << "\n";
                }
                if (controller->has_task("momentum")) {
                    Eigen::VectorXd rec(12);
                    rec << data.hg.toVector(), pinocchio::computeCentroidalMomentumTimeVariation(robot->model(), const_cast<pinocchio::Data&>(data)).toVector();
                    dump_vector(d + "/recall_momentum.npy", rec);
                }
                if (controller->has_task("bounds")) {
                    const auto& bc = controller->bound_task()->getConstraint();
                    Eigen::VectorXd rec(2 * bc.rows());
                    rec << bc.lowerBound(), bc.upperBound();
                    dump_vector(d + "/recall_bounds.npy", rec);
                }
                for (const auto& fr : robot->model().frames) {
                    if (fr.name.find("v_") != 0) continue; // the virtual frames of frames.yaml: the self-collision spheres sit there
                    Eigen::Matrix<double, 6, Eigen::Dynamic> J(6, robot->nv());
                    robot->frameJacobianWorld(data, robot->model().getFrameId(fr.name), J);
                    dump_matrix(d + "/recall_jac_" + clean(fr.name) + ".npy", J);
                }
            }
        }
        std::cout << "dumped " << (n_ticks + stride - 1) / stride << " ticks into " << out << "; max |ddq(own solve) - ddq(controller)| = " << worst
                  << (worst < 1e-10 ? "  (consistent)" : "  (INCONSISTENT: the dump is not the QP the controller solved)") << std::endl;
        return worst < 1e-10 ? 0 : 3;
    }
    catch (std::exception& e) {
        std::cerr << "Exception: " << e.what() << std::endl;
        return 1;
    }
}
