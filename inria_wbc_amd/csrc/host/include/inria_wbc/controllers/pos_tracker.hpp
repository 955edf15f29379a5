// Batched stand-in for inria_wbc::controllers::PosTracker (/root/reference/src/controllers/pos_tracker.cpp:40-326,
// /root/reference/include/inria_wbc/controllers/pos_tracker.hpp:12-82): YAML task stack -> QP structure, the
// `solver:` switch (a third value, "hip-batched", selects libwbcqp), reference setters, contact add/remove
// (changes the QP dimension), task-weight updates.  Registered as "pos-tracker".
#ifndef IWBC_HIP_POS_TRACKER_HPP
#define IWBC_HIP_POS_TRACKER_HPP

#include <iostream>
#include <map>

#include <inria_wbc/controllers/controller.hpp>
#include <inria_wbc/controllers/model_source.hpp>

namespace inria_wbc {
    namespace controllers {
        class PosTracker : public Controller {
        public:
            explicit PosTracker(const yaml::Node& config) : Controller(config)
            {
                yaml::Node c = IWBC_CHECK(config["CONTROLLER"]);
                solver_to_use_ = IWBC_CHECK(c["solver"].as<std::string>());
                // robot dimensions come from the URDF in the reference (controller.cpp:104-118).  Here: from `model:` (the parsed
                // tree, robots/robot_wrapper.hpp) when the step before the path runs on the device too, else from nv / na
                int nv, na;
                if (c["model"]) {
                    auto mf = c["model"].as<std::string>();
                    robot_ = std::make_shared<robots::RobotWrapper>(mf.size() && mf[0] == '/' ? mf : base_path_ + "/" + mf, verbose_);
                    nv = robot_->nv();
                    na = robot_->na();
                    IWBC_ASSERT(robot_->floating_base() == floating_base_, "floating_base and the model disagree");
                    {   // the 1-dof joints in model order: what tsid_joint_names_ holds behind the floating base (controller.cpp:157)
                        const auto& jn = robot_->joint_names();
                        _set_joint_names(std::vector<std::string>(jn.begin() + (floating_base_ ? 1 : 0), jn.end()));
                    }
                    // create additional frames if needed (optional) (pos_tracker.cpp:51-55)
                    if (c["frames"]) {
                        auto ff = c["frames"].as<std::string>();
                        parse_frames(ff.size() && ff[0] == '/' ? ff : base_path_ + "/" + ff);
                    }
                }
                else {
                    nv = IWBC_CHECK(c["nv"].as<int>());
                    na = IWBC_CHECK(c["na"].as<int>());
                    IWBC_ASSERT(mimic_dof_names_.empty(), "mimic_dof_names needs the joint names of a model (CONTROLLER.model)");
                }
                closed_loop_ = c["closed_loop"] ? c["closed_loop"].as<bool>() : false;
                // eiquadprog-fast's iteration bound (DEFAULT_MAX_ITER = 1000; upstream: tsid SolverHQuadProgFast::setMaximumIterations).  The
                // reference never sets it; the key exists so that a caller who bounds tick time gets the reference's own failure,
                // "Status : 3 => Max iter reached" (controller.cpp:297-299), instead of a late tick
                solver_max_iter_ = c["solver_max_iter"] ? c["solver_max_iter"].as<int>() : 1000;
                IWBC_ASSERT(solver_max_iter_ >= 1, "solver_max_iter must be positive");

                // qp solver to be used: the reference accepts 'eiquadprog' or 'qpmad' (pos_tracker.cpp:88-100)
                if (solver_to_use_ == "hip-batched") {
                    wbcqp_desc desc = {c["device"] ? c["device"].as<int>() : 0, WBCQP_F64, 0};
                    int rc = wbcqp_create(&desc, &handle_);
                    if (rc != WBCQP_OK) IWBC_ERROR("'hip-batched' solver is not available: ", wbcqp_last_error(nullptr));
                }
                else if (solver_to_use_ == "eiquadprog" || solver_to_use_ == "qpmad") {
                    IWBC_ERROR("'", solver_to_use_, "' solver is not available: this build does not link tsid; use 'hip-batched'.");
                }
                else {
                    IWBC_ERROR("solver in configuration file must be either 'eiquadprog', 'qpmad' or 'hip-batched'.");
                }

                auto task_file = IWBC_CHECK(c["tasks"].as<std::string>());
                auto p = task_file.size() && task_file[0] == '/' ? task_file : base_path_ + "/" + task_file;
                parse_tasks(p, nv, na);
                if (robot_) {
                    // q_tsid_ = the named reference configuration (pos_tracker.cpp:60-67); one controller instance per batch row
                    auto ref_config = IWBC_CHECK(c["ref_config"].as<std::string>());
                    const auto& ref_map = robot_->referenceConfigurations();
                    IWBC_ASSERT(ref_map.find(ref_config) != ref_map.end(), "The following reference config is not in ref_map : ", ref_config);
                    q0_ = ref_map.at(ref_config);
                    const int batch = c["batch"] ? c["batch"].as<int>() : 1;
                    set_problem_source(std::make_shared<ModelSource>(robot_, batch, q0_));
                }

                if (verbose_) {
                    std::cout << "--------- Solver size info ---------" << std::endl;
                    std::cout << "Solver : " << solver_to_use_ << std::endl;
                    std::cout << "total number of variable (acceleration + contact-force) : " << stack_.nVar() << std::endl;
                    std::cout << "number of equality constraints : " << stack_.nEq() << std::endl;
                    std::cout << "number of inequality constraints : " << stack_.nIn() << std::endl;
                    std::cout << "--------- ------------- ---------" << std::endl;
                }
            }

            // ---- reference setters / getters (pos_tracker.hpp:44-70); one sample drives every instance ----
            void set_com_ref(const TrajectorySample& sample)
            {
                com_ref_ = sample;
                if (source_ && source_->handles_references()) source_->set_com_ref(sample);
            }
            void set_posture_ref(const std::vector<double>& ref)
            {
                IWBC_ASSERT(source_ && source_->handles_references(), "set_posture_ref needs a source that evaluates the task laws");
                source_->set_posture_ref(ref);
            }
            const std::shared_ptr<robots::RobotWrapper>& robot() const { return robot_; }
            void set_se3_ref(const TrajectorySample& sample, const std::string& task_name)
            {
                IWBC_ASSERT(stack_.has_task(task_name), "Task [", task_name, "] not found");
                se3_refs_[task_name] = sample;
                if (source_ && source_->handles_references()) source_->set_se3_ref(task_name, sample);
            }
            bool has_task(const std::string& str) const { return full_stack_.has_task(str); }
            bool has_contact(const std::string& str) const { return full_stack_.contact_index(str) >= 0; }
            void set_contact_se3_ref(const std::vector<double>& pose, const std::string& contact_name)
            {
                IWBC_ASSERT(has_contact(contact_name), "Contact [", contact_name, "] not found");
                IWBC_ASSERT(source_ && source_->handles_references(), "set_contact_se3_ref needs a source that holds the references (CONTROLLER.model)");
                source_->set_contact_se3_ref(contact_name, pose);
            }
            // the sample form (pos_tracker.cpp:240-244): the contact's motion task follows pose, velocity and acceleration
            void set_contact_se3_ref(const TrajectorySample& sample, const std::string& contact_name)
            {
                IWBC_ASSERT(sample.pos.size() == 12 && sample.vel.size() == 6 && sample.acc.size() == 6, "an SE3 sample holds 12 + 6 + 6 numbers");
                std::vector<double> r(sample.pos);
                r.insert(r.end(), sample.vel.begin(), sample.vel.end());
                r.insert(r.end(), sample.acc.begin(), sample.acc.end());
                set_contact_se3_ref(r, contact_name);
            }
            const std::vector<double>& get_com_ref() const { return com_init_; }
            // PosTracker::get_se3_ref (pos_tracker.cpp:211-218): the task's current reference placement, 12 numbers in
            // SE3ToVector order (translation, rotation column-major)
            std::vector<double> get_se3_ref(const std::string& task_name) const
            {
                IWBC_ASSERT(stack_.has_task(task_name), "Task [", task_name, "] not found");
                IWBC_ASSERT(source_ && source_->handles_references(), "get_se3_ref needs a source that holds the task references (CONTROLLER.model)");
                return source_->get_se3_ref(task_name);
            }
            double objective_value(int instance = 0) const { return objective_.at(instance); }
            double cost(const std::string& task_name) const override
            {
                // |A ddq - b| of the first instance (controller.hpp:148-152)
                const auto& t = stack_.task(task_name);
                if (t.type == "posture" || in_.batch == 0) return 0.0;
                _ensure_rows();
                const int nv = stack_.nv();
                double s = 0.0;
                for (int r = 0; r < t.rows; ++r) {
                    double v = -in_.b1[t.first_row + r];
                    for (int j = 0; j < nv; ++j) v += in_.A[(size_t)(t.first_row + r) * nv + j] * a_tsid_(0, j);
                    s += v * v;
                }
                return std::sqrt(s);
            }

            // PosTracker::update_task_weights (pos_tracker.cpp:314-324)
            void update_task_weights(const std::map<std::string, double>& new_weights)
            {
                for (const auto& kv : new_weights) {
                    const auto& t = stack_.task(kv.first);
                    IWBC_ASSERT(t.weight_index >= 0, "Task [", kv.first, "] carries no level-1 weight");
                    weights_[t.weight_index] = kv.second;
                }
            }
            // remove_contact / add_contact (pos_tracker.cpp:246-263): n, nEq, nIn change with the contact set
            void remove_contact(const std::string& contact_name)
            {
                IWBC_ASSERT(stack_.contact_index(contact_name) >= 0, "Trying to remove an contact:", contact_name);
                stack_ = stack_.without_contact(contact_name);
                _install_stack();
            }
            void add_contact(const std::string& contact_name)
            {
                IWBC_ASSERT(full_stack_.contact_index(contact_name) >= 0, "Trying to add an unknown contact:", contact_name);
                stack_ = full_stack_; // every shipped stack has two contacts: adding one back restores the full stack
                _install_stack();
            }
            const tasks::TaskStack& stack() const { return stack_; }

        protected:
            void parse_tasks(const std::string& path, int nv, int na)
            {
                if (verbose_) std::cout << "parsing task file:" << path << std::endl;
                yaml::Node task_list = IWBC_CHECK(yaml::LoadFile(path));
                stack_ = tasks::TaskStack(task_list, nv, na);
                full_stack_ = stack_;
                _install_stack();
                if (verbose_) std::cout << "Number of parsed tasks " << task_list.size() << std::endl;
            }

            // PosTracker::parse_frames (pos_tracker.cpp:191-209): virtual frames, `ref:` an existing frame, `pos:` the offset
            void parse_frames(const std::string& path)
            {
                if (verbose_) std::cout << "Parsing virtual frame file:" << path << std::endl;
                yaml::Node node = IWBC_CHECK(yaml::LoadFile(path));
                for (const auto& kv : node) {
                    auto ref = IWBC_CHECK(kv.second["ref"].as<std::string>());
                    auto pos = IWBC_CHECK(kv.second["pos"].as<std::vector<double>>());
                    IWBC_ASSERT(pos.size() == 3, "frame ", kv.first, ": pos needs 3 numbers");
                    robot_->addFrame(kv.first, ref, {{pos[0], pos[1], pos[2]}});
                }
            }

            // One library slot per set of active contacts: the first time a set is seen its structure goes to the device (the
            // analogue of solver_->resize(nVar, nEq, nIn), pos_tracker.cpp:102, which tsid repeats on every contact change);
            // coming back to a set seen before costs nothing on the device.
            void _install_stack()
            {
                std::string key;
                for (const auto& c : stack_.contacts()) key += c.name + ";";
                auto known = slots_.find(key);
                if (known == slots_.end()) {
                    const int slot = (int)slots_.size();
                    IWBC_ASSERT(slot < WBCQP_MAX_STRUCTURES, "more contact sets than the library has slots");
                    wbcqp_structure s = stack_.c_struct();
                    s.max_iter = solver_max_iter_;
                    int rc = wbcqp_set_structure(handle_, slot, &s);
                    if (rc != WBCQP_OK) IWBC_ERROR("wbcqp_set_structure failed: ", wbcqp_last_error(handle_));
                    wbcqp_layout L;
                    rc = wbcqp_layout_of(&s, &L);
                    if (rc != WBCQP_OK) IWBC_ERROR("wbcqp_layout_of failed");
                    known = slots_.emplace(key, std::make_pair(slot, L)).first;
                }
                slot_ = known->second.first;
                layout_ = known->second.second;
                std::vector<double> w = stack_.default_weights();
                if (!weights_.empty()) // keep user-updated weights across a contact switch, by name
                    for (size_t i = 0; i < w.size(); ++i)
                        for (size_t j = 0; j < weight_names_.size(); ++j)
                            if (weight_names_[j] == stack_.weight_names()[i]) w[i] = weights_[j];
                weights_ = w;
                weight_names_ = stack_.weight_names();
                activated_contacts_.clear();
                for (const auto& c : stack_.contacts()) activated_contacts_.push_back(c.name);
                if (all_contacts_.empty()) all_contacts_ = activated_contacts_;
                if (source_ && source_->handles_references()) source_->bind(handle_, slot_, stack_, dt_);
            }

            void _reset() override
            {
                const int nv = stack_.nv(), nq = floating_base_ ? nv + 1 : nv;
                q_tsid_ = MatrixXd(batch_, nq);
                if (!q0_.empty()) {
                    for (int i = 0; i < batch_; ++i)
                        for (int j = 0; j < nq; ++j) q_tsid_(i, j) = q0_[j];
                }
                else if (floating_base_)
                    for (int i = 0; i < batch_; ++i) q_tsid_(i, 6) = 1.0; // unit quaternion
                if (source_->handles_references()) source_->bind(handle_, slot_, stack_, dt_);
                v_tsid_ = MatrixXd(batch_, nv);
                a_tsid_ = MatrixXd(batch_, nv);
                MatrixXd cp, cv;
                source_->com(cp, cv);
                com_init_.assign(3, 0.0);
                if (cp.rows > 0)
                    for (int d = 0; d < 3; ++d) com_init_[d] = cp(0, d);
                com_ref_ = TrajectorySample(3);
                com_ref_.pos = com_init_;
                com_ref_set_ = false;
            }

            void _build_inputs(const MatrixXd& q, const MatrixXd& v) override
            {
                in_.resize(batch_, layout_);
                if (source_->handles_references())
                    source_->fill_limits(layout_, in_); // the rows themselves are computed inside the tick (Controller::_solve)
                else
                    source_->compute(t_, q, v, stack_, layout_, in_);
                for (int i = 0; i < batch_; ++i)
                    for (int k = 0; k < layout_.len_w; ++k) in_.w[(size_t)i * layout_.len_w + k] = weights_[k];
                // CoM task PD law: b += Kp (x_ref - x) + Kd (v_ref - v) + a_ref on the masked axes (tsid TaskComEquality)
                // -- unless the source evaluates the task laws itself (ModelSource: the reference went to it in set_com_ref)
                if (com_ref_set_ && stack_.has_task("com") && !source_->handles_references()) {
                    const auto& t = stack_.task("com");
                    MatrixXd cp, cv;
                    source_->com(cp, cv);
                    for (int i = 0; i < batch_; ++i) {
                        int r = 0;
                        for (int d = 0; d < 3; ++d) {
                            if (t.mask[d] != '1') continue;
                            in_.b1[(size_t)i * layout_.len_b1 + t.first_row + r] +=
                                t.kp * (com_ref_.pos[d] - cp(i, d)) + t.kd * (com_ref_.vel[d] - cv(i, d)) + com_ref_.acc[d];
                            ++r;
                        }
                    }
                }
            }
            const tasks::TaskStack& _stack() const override { return stack_; }
            int _slot() const override { return slot_; }
            const wbcqp_layout& _layout() const override { return layout_; }

        public:
            // set_com_ref with tracking enabled (MoveCom::update calls this every tick)
            void set_com_ref_tracking(const TrajectorySample& sample)
            {
                com_ref_ = sample;
                com_ref_set_ = true;
                if (source_ && source_->handles_references()) source_->set_com_ref(sample);
            }

        protected:
            tasks::TaskStack stack_, full_stack_;
            wbcqp_layout layout_{};
            std::vector<double> weights_;
            std::vector<std::string> weight_names_;
            TrajectorySample com_ref_{3};
            bool com_ref_set_ = false;
            std::vector<double> com_init_, q0_;
            std::map<std::string, std::pair<int, wbcqp_layout>> slots_; // active-contact set -> (library slot, layout)
            int slot_ = 0;
            std::shared_ptr<robots::RobotWrapper> robot_;
            std::map<std::string, TrajectorySample> se3_refs_;
        };
    } // namespace controllers
} // namespace inria_wbc
#endif
