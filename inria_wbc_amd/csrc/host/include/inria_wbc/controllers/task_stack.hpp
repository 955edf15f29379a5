// TaskStack: what PosTracker::parse_tasks (/root/reference/src/controllers/pos_tracker.cpp:161-189) and the task
// factories (/root/reference/src/controllers/tasks.cpp:38-404) establish at construction time, expressed as the
// constant structure the batched QP needs (include/wbcqp.h: wbcqp_structure).  Tasks are taken in file order;
// level-1 weights are indexed in the order tsid would have received addMotionTask / addRigidContact calls.
#ifndef IWBC_HIP_TASK_STACK_HPP
#define IWBC_HIP_TASK_STACK_HPP

#include <array>
#include <cmath>
#include <string>
#include <vector>

#include <inria_wbc/exceptions.hpp>
#include <inria_wbc/utils/yaml_lite.hpp>

#include "wbcqp.h"

namespace inria_wbc {
    namespace tasks {
        namespace cst {
            static constexpr double w_force_feet = 1e-3; // regularization force for contacts (reference tasks.hpp:23)
        }

        struct ContactSpec {
            std::string name, joint;
            double lxp, lxn, lyp, lyn, lz, mu, fmin, fmax, kp;
            std::array<double, 3> normal;
        };

        struct TaskSpec {
            std::string name, type, mask;
            double weight = 0.0, kp = 0.0, kd = 0.0;
            int rows = 0;       // level-1 rows this task contributes
            int first_row = -1; // first dense row (dense tasks) or first selection row (posture)
            int weight_index = -1;
        };

        inline int mask_rows(const std::string& mask, size_t expected, const std::string& task)
        {
            IWBC_ASSERT(mask.size() == expected, "wrong size for the mask of task ", task, ": ", mask.size());
            int r = 0;
            for (char ch : mask) r += (ch == '1');
            return r;
        }

        class TaskStack {
        public:
            TaskStack() = default;
            // `tasks`: the tasks.yaml tree; nv / na come from the robot model (the URDF is not read here)
            TaskStack(const yaml::Node& tasks, int nv, int na) { parse(tasks, nv, na); }

            int nv() const { return nv_; }
            int na() const { return na_; }
            int nc() const { return (int)contacts_.size(); }
            int nVar() const { return nv_ + 12 * nc(); }
            int nEq() const { return (nv_ - na_) + 6 * nc(); }
            int nIn() const
            {
                int t = 0;
                for (size_t b = 0; b < ineq_kind_.size(); ++b)
                    t += ineq_kind_[b] == WBCQP_INEQ_BOUNDS ? (int)bound_col_.size() : (ineq_kind_[b] == WBCQP_INEQ_ACTUATION ? na_ : 17);
                return t;
            }
            int n_dense() const { return (int)dense_row_task_.size(); }
            int n_sel() const { return (int)sel_col_.size(); }
            int n_tasks() const { return (int)weights_.size(); }
            int n_acteq() const { return (int)acteq_joint_.size(); }
            bool has_cop() const { return cop_task_ >= 0; }
            int level1_rows() const { return n_dense() + n_sel() + 6 * nc() + n_acteq() + (has_cop() ? 3 : 0); }
            bool has_task(const std::string& name) const
            {
                for (const auto& t : tasks_)
                    if (t.name == name) return true;
                return false;
            }
            const TaskSpec& task(const std::string& name) const
            {
                for (const auto& t : tasks_)
                    if (t.name == name) return t;
                IWBC_ERROR("Task [", name, "] not found");
            }
            const std::vector<TaskSpec>& tasks() const { return tasks_; }
            const yaml::Node& source() const { return source_; } // the tasks.yaml tree, in file order
            const std::vector<ContactSpec>& contacts() const { return contacts_; }
            const std::vector<double>& default_weights() const { return weights_; }
            const std::vector<std::string>& weight_names() const { return weight_names_; }
            int contact_index(const std::string& name) const
            {
                for (size_t c = 0; c < contacts_.size(); ++c)
                    if (contacts_[c].name == name) return (int)c;
                return -1;
            }

            // the same stack with one contact removed (PosTracker::remove_contact, pos_tracker.cpp:246-254)
            TaskStack without_contact(const std::string& name) const
            {
                yaml::Node pruned = yaml::Node::MakeMap();
                for (const auto& kv : source_)
                    if (kv.first != name) pruned.set(kv.first, kv.second);
                return TaskStack(pruned, nv_, na_);
            }

            // pointers stay valid while this object lives and is not modified
            wbcqp_structure c_struct() const
            {
                wbcqp_structure s;
                s.nv = nv_; s.na = na_; s.nc = nc();
                s.n_dense = n_dense(); s.n_tasks = n_tasks();
                s.dense_row_task = dense_row_task_.data();
                s.n_sel = n_sel(); s.sel_col = sel_col_.data(); s.sel_task = sel_task_.data();
                s.forcereg_mat = forcereg_.data(); s.forcereg_task = forcereg_task_.data();
                s.force_gen = force_gen_.data();
                s.fric_mat = fric_.data(); s.fric_lb = fric_lb_.data(); s.fric_ub = fric_ub_.data();
                s.n_bound = (int)bound_col_.size(); s.bound_col = bound_col_.data();
                s.act_bounds = act_bounds_ ? 1 : 0;
                s.n_ineq_blocks = (int)ineq_kind_.size(); s.ineq_kind = ineq_kind_.data(); s.ineq_arg = ineq_arg_.data();
                s.hessian_reg = 1e-8; // tsid DEFAULT_HESSIAN_REGULARIZATION
                s.max_iter = 1000;    // eiquadprog-fast DEFAULT_MAX_ITER
                s.n_acteq = n_acteq(); s.acteq_joint = acteq_joint_.data(); s.acteq_scale = acteq_scale_.data();
                s.acteq_task = acteq_task_; s.cop_task = cop_task_;
                return s;
            }

        private:
            void parse(const yaml::Node& tasks, int nv, int na)
            {
                IWBC_ASSERT(tasks.IsMap(), "the task list must be a map of named tasks");
                nv_ = nv;
                na_ = na;
                source_ = tasks;
                for (const auto& kv : tasks) {
                    const std::string& name = kv.first;
                    const yaml::Node& node = kv.second;
                    const auto type = IWBC_CHECK(node["type"].as<std::string>());
                    check_keys(name, type, node);
                    TaskSpec t;
                    t.name = name;
                    t.type = type;
                    if (type == "contact") {
                        add_contact(name, node);
                        continue;
                    }
                    t.weight = IWBC_CHECK(node["weight"].as<double>());
                    if (node["kp"]) t.kp = node["kp"].as<double>();
                    t.kd = node["kd"] ? node["kd"].as<double>() : 2.0 * std::sqrt(t.kp); // Kd = 2 sqrt(Kp) (tasks.cpp:57,106)
                    if (type == "se3" || type == "momentum") {
                        t.mask = IWBC_CHECK(node["mask"].as<std::string>());
                        t.rows = mask_rows(t.mask, 6, name);
                        add_dense(t);
                    }
                    else if (type == "com") {
                        t.mask = IWBC_CHECK(node["mask"].as<std::string>());
                        t.rows = mask_rows(t.mask, 3, name);
                        add_dense(t);
                    }
                    else if (type == "self-collision") {
                        t.rows = 1; // one soft-repulsion row (task-self-collision.cpp:195-197)
                        add_dense(t);
                    }
                    else if (type == "posture") {
                        // tasks.cpp:205-214: without `mask:` every actuated joint, with it one character per actuated joint
                        t.mask = node["mask"] ? IWBC_CHECK(node["mask"].as<std::string>()) : std::string((size_t)na_, '1');
                        IWBC_ASSERT((int)t.mask.size() == na_, "wrong size in posture mask, expected:", na_, " got:", t.mask.size());
                        t.rows = mask_rows(t.mask, (size_t)na_, name);
                        t.weight_index = (int)weights_.size();
                        t.first_row = (int)sel_col_.size();
                        weights_.push_back(t.weight);
                        weight_names_.push_back(name);
                        for (int j = 0; j < na_; ++j)
                            if (t.mask[j] == '1') {
                                sel_col_.push_back(nv_ - na_ + j);
                                sel_task_.push_back(t.weight_index);
                            }
                    }
                    else if (type == "torque") {
                        // tasks.cpp:227-271: tsid TaskActuationEquality, mask over the actuated joints (default all), `scaling:` = its
                        // weight vector (default ones), reference zero, level 1.  Rows scale_j [M_a(j,:) | -J_a(:,j)']: H becomes one
                        // n x n matrix (wbcqp_layout.dense_h)
                        IWBC_ASSERT(acteq_task_ < 0, "one torque task per stack (task ", name, " is the second)");
                        t.mask = node["mask"] ? IWBC_CHECK(node["mask"].as<std::string>()) : std::string((size_t)na_, '1');
                        IWBC_ASSERT((int)t.mask.size() == na_, "wrong size in torque mask, expected:", na_, " got:", t.mask.size());
                        std::vector<double> scaling((size_t)na_, 1.0);
                        if (node["scaling"]) {
                            scaling = IWBC_CHECK(node["scaling"].as<std::vector<double>>());
                            IWBC_ASSERT((int)scaling.size() == na_, "wrong size in torque scaling, expected:", na_, " got:", scaling.size());
                        }
                        t.rows = mask_rows(t.mask, (size_t)na_, name);
                        t.weight_index = acteq_task_ = (int)weights_.size();
                        weights_.push_back(t.weight);
                        weight_names_.push_back(name);
                        for (int j = 0; j < na_; ++j)
                            if (t.mask[j] == '1') {
                                acteq_joint_.push_back(j);
                                acteq_scale_.push_back(scaling[j]);
                            }
                    }
                    else if (type == "cop") {
                        // tasks.cpp:156-178: tsid TaskCopEquality with reference (0, 0, 0), a force task on level 1 over all contact forces
                        IWBC_ASSERT(cop_task_ < 0, "one cop task per stack (task ", name, " is the second)");
                        t.rows = 3;
                        t.weight_index = cop_task_ = (int)weights_.size();
                        weights_.push_back(t.weight);
                        weight_names_.push_back(name);
                    }
                    else if (type == "bounds") {
                        for (int j = 0; j < na_; ++j) bound_col_.push_back(nv_ - na_ + j);
                        ineq_kind_.push_back(WBCQP_INEQ_BOUNDS);
                        ineq_arg_.push_back(0);
                    }
                    else if (type == "actuation-bounds") {
                        act_bounds_ = true;
                        ineq_kind_.push_back(WBCQP_INEQ_ACTUATION);
                        ineq_arg_.push_back(0);
                    }
                    else
                        IWBC_ERROR("task type [", type, "] of task ", name, " is not registered (known: se3, com, momentum, cop, posture, torque, "
                                   "bounds, actuation-bounds, self-collision, contact: the reference's factory, tasks.cpp:85-404)");
                    tasks_.push_back(t);
                }
                IWBC_ASSERT(cop_task_ < 0 || !contacts_.empty(), "a cop task needs a contact");
            }

            // A key the factory of that type does not read is a typo in the stack, not a setting: the reference's yaml-cpp lookups
            // would ignore it silently and solve another QP than the file describes; here it raises.
            static void check_keys(const std::string& name, const std::string& type, const yaml::Node& node)
            {
                static const std::vector<std::pair<std::string, std::vector<std::string>>> known = {
                    {"se3", {"type", "tracked", "weight", "kp", "kd", "mask"}},
                    {"com", {"type", "weight", "kp", "kd", "mask"}},
                    {"momentum", {"type", "weight", "kp", "kd", "mask"}},
                    {"cop", {"type", "weight"}},
                    {"posture", {"type", "weight", "kp", "kd", "ref", "mask"}},
                    {"torque", {"type", "weight", "mask", "scaling"}},
                    {"bounds", {"type", "weight"}},
                    {"actuation-bounds", {"type", "weight"}},
                    {"self-collision", {"type", "tracked", "weight", "kp", "kd", "radius", "margin", "m", "avoided"}},
                    {"contact", {"type", "joint", "kp", "kd", "lxp", "lxn", "lyp", "lyn", "lz", "mu", "normal", "fmin", "fmax"}}};
                for (const auto& k : known) {
                    if (k.first != type) continue;
                    for (const auto& kv : node) {
                        bool ok = false;
                        for (const auto& key : k.second) ok = ok || key == kv.first;
                        if (!ok) IWBC_ERROR("task ", name, " (type ", type, "): unknown key [", kv.first, "]");
                    }
                }
            }

            void add_dense(TaskSpec& t)
            {
                t.weight_index = (int)weights_.size();
                t.first_row = (int)dense_row_task_.size();
                weights_.push_back(t.weight);
                weight_names_.push_back(t.name);
                for (int r = 0; r < t.rows; ++r) dense_row_task_.push_back(t.weight_index);
            }

            void add_contact(const std::string& name, const yaml::Node& node)
            {
                ContactSpec c;
                c.name = name;
                c.kp = IWBC_CHECK(node["kp"].as<double>());
                c.joint = IWBC_CHECK(node["joint"].as<std::string>());
                c.lxn = IWBC_CHECK(node["lxn"].as<double>());
                c.lyn = IWBC_CHECK(node["lyn"].as<double>());
                c.lxp = IWBC_CHECK(node["lxp"].as<double>());
                c.lyp = IWBC_CHECK(node["lyp"].as<double>());
                c.lz = IWBC_CHECK(node["lz"].as<double>());
                c.mu = IWBC_CHECK(node["mu"].as<double>());
                auto normal = IWBC_CHECK(node["normal"].as<std::vector<double>>());
                c.fmin = IWBC_CHECK(node["fmin"].as<double>());
                c.fmax = IWBC_CHECK(node["fmax"].as<double>());
                IWBC_ASSERT(normal.size() == 3, "normal size:", normal.size());
                c.normal = {normal[0], normal[1], normal[2]};
                const int ci = (int)contacts_.size();
                contacts_.push_back(c);
                // 4 contact points in the order of tasks.cpp:353-358
                const double px[4] = {-c.lxn, -c.lxn, c.lxp, c.lxp}, py[4] = {-c.lyn, c.lyp, -c.lyn, c.lyp};
                // force generator T = per point [I3; skew(p)]   (tsid Contact6d::updateForceGeneratorMatrix)
                std::vector<double> T(72, 0.0);
                for (int p = 0; p < 4; ++p) {
                    const double x = px[p], y = py[p], z = c.lz;
                    for (int d = 0; d < 3; ++d) T[d * 12 + 3 * p + d] = 1.0;
                    const double sk[9] = {0, -z, y, z, 0, -x, -y, x, 0};
                    for (int r = 0; r < 3; ++r)
                        for (int d = 0; d < 3; ++d) T[(3 + r) * 12 + 3 * p + d] = sk[r * 3 + d];
                }
                force_gen_.insert(force_gen_.end(), T.begin(), T.end());
                // force regularisation matrix diag(w_f) T with tsid's default w_f = (1,1,1e-3,2,2,2)
                const double wf[6] = {1.0, 1.0, 1e-3, 2.0, 2.0, 2.0};
                for (int r = 0; r < 6; ++r)
                    for (int m = 0; m < 12; ++m) forcereg_.push_back(wf[r] * T[r * 12 + m]);
                // friction pyramid + normal force rows (tsid Contact6d::updateForceInequalityConstraints)
                const std::array<double, 3> n = c.normal;
                auto cross = [](const std::array<double, 3>& a, const std::array<double, 3>& b) {
                    return std::array<double, 3>{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
                };
                auto norm = [](const std::array<double, 3>& a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); };
                std::array<double, 3> t1 = cross(n, {1.0, 0.0, 0.0});
                if (norm(t1) < 1e-5) t1 = cross(n, {0.0, 1.0, 0.0});
                std::array<double, 3> t2 = cross(n, t1);
                const double n1 = norm(t1), n2 = norm(t2);
                for (int d = 0; d < 3; ++d) {
                    t1[d] /= n1;
                    t2[d] /= n2;
                }
                std::vector<double> B(17 * 12, 0.0), lb(17, -1e10), ub(17, 0.0);
                for (int p = 0; p < 4; ++p)
                    for (int d = 0; d < 3; ++d) {
                        B[(4 * p + 0) * 12 + 3 * p + d] = -t1[d] - c.mu * n[d];
                        B[(4 * p + 1) * 12 + 3 * p + d] = t1[d] - c.mu * n[d];
                        B[(4 * p + 2) * 12 + 3 * p + d] = -t2[d] - c.mu * n[d];
                        B[(4 * p + 3) * 12 + 3 * p + d] = t2[d] - c.mu * n[d];
                        B[16 * 12 + 3 * p + d] = n[d];
                    }
                lb[16] = c.fmin;
                ub[16] = c.fmax;
                fric_.insert(fric_.end(), B.begin(), B.end());
                fric_lb_.insert(fric_lb_.end(), lb.begin(), lb.end());
                fric_ub_.insert(fric_ub_.end(), ub.begin(), ub.end());
                // addRigidContact(contact, w_force_feet): force inequality (level 0), motion equality (level 0), force reg (level 1)
                ineq_kind_.push_back(WBCQP_INEQ_FORCE);
                ineq_arg_.push_back(ci);
                forcereg_task_.push_back((int)weights_.size());
                weights_.push_back(cst::w_force_feet);
                weight_names_.push_back("forcereg_" + name);
            }

            int nv_ = 0, na_ = 0;
            bool act_bounds_ = false;
            int acteq_task_ = -1, cop_task_ = -1;
            std::vector<int32_t> acteq_joint_;
            std::vector<double> acteq_scale_;
            yaml::Node source_;
            std::vector<TaskSpec> tasks_;
            std::vector<ContactSpec> contacts_;
            std::vector<double> weights_;
            std::vector<std::string> weight_names_;
            std::vector<int32_t> dense_row_task_, sel_col_, sel_task_, forcereg_task_, bound_col_, ineq_kind_, ineq_arg_;
            std::vector<double> forcereg_, force_gen_, fric_, fric_lb_, fric_ub_;
        };
    } // namespace tasks
} // namespace inria_wbc
#endif
