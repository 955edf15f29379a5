// ModelSource: the step before the path on the device.  Holds what the reference's task objects hold after construction
// (/root/reference/src/controllers/tasks.cpp:38-404: tracked frame, mask, gains, reference per task) and turns the robot
// state of every instance into the rows of the QP with wbcqp_problem_data -- the batched counterpart of
// tsid_->computeProblemData(t, q, dq) at /root/reference/src/controllers/controller.cpp:244.
#ifndef IWBC_HIP_MODEL_SOURCE_HPP
#define IWBC_HIP_MODEL_SOURCE_HPP

#include <map>
#include <memory>

#include <inria_wbc/controllers/controller.hpp>
#include <inria_wbc/robots/robot_wrapper.hpp>

namespace inria_wbc {
    namespace controllers {
        class ModelSource : public ProblemSource {
            // what one (slot, task stack) pair needs: the task table handed to wbcqp_set_model and the reference vector laid out
            // for it.  One per slot: a controller that switches between contact sets keeps them all and only re-points.
            struct Bound {
                int nref_ = 0, posture_ref_ = 0;
                std::vector<double> ref_;
                std::vector<wbcqp_task> tasks_;
                std::vector<std::string> names_, contact_names_;
                std::vector<int32_t> avoided_frames_, contact_frame_, contact_ref_;
                std::vector<double> avoided_r0_, contact_kp_, contact_kd_;
            };
#define IWBC_BOUND_ALIASES(B)                                                                                                         \
    auto& nref_ = (B).nref_; auto& posture_ref_ = (B).posture_ref_; auto& ref_ = (B).ref_; auto& tasks_ = (B).tasks_;                  \
    auto& names_ = (B).names_; auto& contact_names_ = (B).contact_names_; auto& avoided_frames_ = (B).avoided_frames_;                 \
    auto& contact_frame_ = (B).contact_frame_; auto& contact_ref_ = (B).contact_ref_; auto& avoided_r0_ = (B).avoided_r0_;             \
    auto& contact_kp_ = (B).contact_kp_; auto& contact_kd_ = (B).contact_kd_;                                                         \
    (void)nref_; (void)posture_ref_; (void)ref_; (void)tasks_; (void)names_; (void)contact_names_; (void)avoided_frames_;              \
    (void)contact_frame_; (void)contact_ref_; (void)avoided_r0_; (void)contact_kp_; (void)contact_kd_;


        public:
            // q0: the configuration the task references are initialised at (the controller's ref_config, pos_tracker.cpp:60-67)
            ModelSource(const std::shared_ptr<robots::RobotWrapper>& robot, int batch, const std::vector<double>& q0)
                : robot_(robot), batch_(batch), q0_(q0)
            {
                IWBC_ASSERT(robot_, "Invalid robot");
                IWBC_ASSERT(batch_ > 0, "batch must be positive");
                IWBC_ASSERT((int)q0_.size() == robot_->nq(), "q0 must hold nq entries");
            }
            int batch() const override { return batch_; }
            bool handles_references() const override { return true; }

            void bind(wbcqp_handle* h, int slot, const tasks::TaskStack& stack, double dt) override
            {
                handle_ = h;
                slot_ = slot;
                auto known = bounds_.find(slot);
                if (known != bounds_.end()) {
                    cur_ = &known->second;
                    refresh(*cur_);
                    return;
                }
                cur_ = &bounds_[slot];
                IWBC_BOUND_ALIASES(*cur_)
                const int na = robot_->na();
                IWBC_ASSERT(stack.nv() == robot_->nv() && stack.na() == na, "task stack and robot disagree on nv / na");
                tasks_.clear(); avoided_frames_.clear(); avoided_r0_.clear(); names_.clear();
                contact_frame_.clear(); contact_kp_.clear(); contact_kd_.clear(); contact_ref_.clear(); contact_names_.clear();
                std::vector<size_t> av_begin;
                int off = 0;
                bool bounds = false;
                double pkp = 0.0, pkd = 0.0;
                int posture_ref = 0;
                std::string posture_ref_name;
                std::vector<std::pair<std::string, int>> contacts; // name, frame
                std::vector<double> ckp;
                for (const auto& kv : stack.source()) {
                    const yaml::Node& node = kv.second;
                    const auto type = IWBC_CHECK(node["type"].as<std::string>());
                    wbcqp_task t{};
                    const size_t begin = avoided_frames_.size();
                    auto mask_bits = [](const std::string& m) { int b = 0; for (size_t i = 0; i < m.size(); ++i) if (m[i] != '0') b |= 1 << i; return b; };
                    if (type == "se3" || type == "com" || type == "momentum") {
                        const double kp = IWBC_CHECK(node["kp"].as<double>());
                        t.kind = type == "se3" ? WBCQP_T_SE3 : (type == "com" ? WBCQP_T_COM : WBCQP_T_MOMENTUM);
                        t.frame = type == "se3" ? robot_->getFrameId(IWBC_CHECK(node["tracked"].as<std::string>())) : 0;
                        t.mask = mask_bits(IWBC_CHECK(node["mask"].as<std::string>()));
                        t.kp = kp;
                        t.kd = 2.0 * std::sqrt(kp); // tasks.cpp:57,107,140
                        t.ref = off;
                        off += type == "se3" ? 24 : (type == "com" ? 9 : 12);
                    }
                    else if (type == "self-collision") {
                        t.kind = WBCQP_T_SELFCOLLISION;
                        t.frame = robot_->getFrameId(IWBC_CHECK(node["tracked"].as<std::string>()));
                        t.mask = 1;
                        t.kp = IWBC_CHECK(node["kp"].as<double>());
                        t.kd = IWBC_CHECK(node["kd"].as<double>());
                        t.radius = IWBC_CHECK(node["radius"].as<double>());
                        t.margin = IWBC_CHECK(node["margin"].as<double>());
                        t.m = IWBC_CHECK(node["m"].as<double>());
                        for (const auto& a : IWBC_CHECK(node["avoided"])) {
                            IWBC_ASSERT(robot_->existFrame(a.first), "Frame ", a.first, " in ", kv.first, " does not exists."); // tasks.cpp:388
                            avoided_frames_.push_back(robot_->getFrameId(a.first));
                            avoided_r0_.push_back(a.second.as<double>());
                            ++t.n_avoided;
                        }
                    }
                    else if (type == "posture") {
                        pkp = IWBC_CHECK(node["kp"].as<double>());
                        pkd = 2.0 * std::sqrt(pkp);
                        posture_ref_name = IWBC_CHECK(node["ref"].as<std::string>());
                        continue;
                    }
                    else if (type == "contact") {
                        const auto joint = IWBC_CHECK(node["joint"].as<std::string>());
                        IWBC_ASSERT(robot_->existFrame(joint), joint, " does not exist!"); // tasks.cpp:349
                        contacts.emplace_back(kv.first, robot_->getFrameId(joint));
                        ckp.push_back(IWBC_CHECK(node["kp"].as<double>()));
                        continue;
                    }
                    else if (type == "bounds") { bounds = true; continue; }
                    else continue; // actuation-bounds: constant limits; torque / cop: the rows are the structure's (task_stack.hpp), the
                                   // cop rows come out of the rows kernel with the contact frames, the torque reference is zero (tasks.cpp:263)
                    av_begin.push_back(begin);
                    tasks_.push_back(t);
                    names_.push_back(kv.first);
                }
                for (size_t i = 0; i < tasks_.size(); ++i)
                    if (tasks_[i].kind == WBCQP_T_SELFCOLLISION) {
                        tasks_[i].avoided_frame = avoided_frames_.data() + av_begin[i];
                        tasks_[i].avoided_r0 = avoided_r0_.data() + av_begin[i];
                    }
                posture_ref = off;
                if (stack.n_sel() > 0) off += na;
                for (size_t c = 0; c < contacts.size(); ++c) {
                    contact_frame_.push_back(contacts[c].second);
                    contact_kp_.push_back(ckp[c]);
                    contact_kd_.push_back(2.0 * std::sqrt(ckp[c])); // tasks.cpp:360
                    contact_ref_.push_back(off);
                    off += 24; // a full sample, like an SE3 task
                }
                nref_ = off;
                // references of a freshly constructed controller: every frame where it is at q0 (tasks.cpp:64-80,361), the CoM
                // where it is (tasks.cpp:109), zero momentum (tasks.cpp:144-145), the named posture (tasks.cpp:194-217)
                std::vector<double> one(nref_, 0.0);
                for (size_t i = 0; i < tasks_.size(); ++i) {
                    const wbcqp_task& t = tasks_[i];
                    if (t.kind == WBCQP_T_SE3) {
                        auto v = robot_->framePosition(q0_.data(), t.frame).to_vector();
                        std::copy(v.begin(), v.end(), one.begin() + t.ref);
                    }
                    else if (t.kind == WBCQP_T_COM) {
                        auto c = robot_->com(q0_.data());
                        std::copy(c.begin(), c.end(), one.begin() + t.ref);
                    }
                    auto it = named_.find(names_[i]);
                    if (it != named_.end() && t.kind != WBCQP_T_SELFCOLLISION) std::copy(it->second.begin(), it->second.end(), one.begin() + t.ref);
                }
                if (stack.n_sel() > 0) {
                    const auto& refs = robot_->referenceConfigurations();
                    IWBC_ASSERT(refs.count(posture_ref_name) == 1, "Reference name ", posture_ref_name, " not found"); // tasks.cpp:194
                    const auto& rq = refs.at(posture_ref_name);
                    std::copy(rq.end() - na, rq.end(), one.begin() + posture_ref);
                    if (!posture_user_.empty()) std::copy(posture_user_.begin(), posture_user_.end(), one.begin() + posture_ref);
                }
                for (size_t c = 0; c < contacts.size(); ++c) {
                    contact_names_.push_back(contacts[c].first);
                    auto v = robot_->framePosition(q0_.data(), contact_frame_[c]).to_vector();
                    std::copy(v.begin(), v.end(), one.begin() + contact_ref_[c]);
                    auto it = named_.find(contacts[c].first);
                    if (it != named_.end()) std::copy(it->second.begin(), it->second.end(), one.begin() + contact_ref_[c]);
                }
                ref_.assign((size_t)batch_ * nref_, 0.0);
                for (int i = 0; i < batch_; ++i) std::copy(one.begin(), one.end(), ref_.begin() + (size_t)i * nref_);
                posture_ref_ = posture_ref;

                wbcqp_model md = robot_->c_model();
                wbcqp_taskmap tm{};
                tm.n_task = (int)tasks_.size();
                tm.task = tasks_.data();
                tm.posture_kp = pkp; tm.posture_kd = pkd; tm.posture_ref = posture_ref;
                tm.n_contact = (int)contact_frame_.size();
                tm.contact_frame = contact_frame_.data(); tm.contact_kp = contact_kp_.data(); tm.contact_kd = contact_kd_.data();
                tm.contact_ref = contact_ref_.data();
                tm.bounds = bounds ? 1 : 0;
                tm.dt = dt;
                tm.nref = nref_;
                if (wbcqp_set_model(h, slot, &md, &tm) != WBCQP_OK) IWBC_ERROR("wbcqp_set_model failed: ", wbcqp_last_error(h));
                auto c0 = robot_->com(q0_.data());
                com_pos_ = MatrixXd(batch_, 3);
                com_vel_ = MatrixXd(batch_, 3);
                for (int i = 0; i < batch_; ++i)
                    for (int d = 0; d < 3; ++d) com_pos_(i, d) = c0[d];
            }

            // the controller re-set the structure of a slot (wbcqp_set_structure drops the model): build again on the next bind
            void forget(int slot)
            {
                if (cur_ && bounds_.count(slot) && cur_ == &bounds_[slot]) cur_ = nullptr;
                bounds_.erase(slot);
            }

            void compute(double, const MatrixXd& q, const MatrixXd& v, const tasks::TaskStack& stack, const wbcqp_layout& L, TickInputs& in) override
            {
                IWBC_ASSERT(cur_, "ModelSource is not bound to a solver");
                IWBC_BOUND_ALIASES(*cur_)
                IWBC_ASSERT(handle_, "ModelSource is not bound to a solver");
                IWBC_ASSERT(q.rows == batch_ && q.cols == robot_->nq() && v.rows == batch_ && v.cols == robot_->nv(), "one state row per instance");
                wbcqp_state st = {q.data.data(), v.data.data(), ref_.data()};
                wbcqp_inputs rows{};
                rows.M = in.M.data(); rows.h = in.h.data(); rows.A = in.A.data(); rows.b1 = in.b1.data(); rows.Ac = in.Ac.data();
                rows.bc = in.bc.data(); rows.blb = in.blb.data(); rows.bub = in.bub.data(); rows.Acop = in.Acop.data();
                if (wbcqp_problem_data_host(handle_, slot_, batch_, &st, &rows) != WBCQP_OK)
                    IWBC_ERROR("wbcqp_problem_data_host failed: ", wbcqp_last_error(handle_));
                fill_limits(L, in);
                (void)stack;
            }
            // actuation bounds: -tau_max, tau_max (tasks.cpp:315-316)
            void fill_limits(const wbcqp_layout& L, TickInputs& in) const override
            {
                const auto& tmax = robot_->effortLimit();
                for (int i = 0; i < batch_; ++i)
                    for (int j = 0; j < L.len_tlb; ++j) {
                        in.tlb[(size_t)i * L.len_tlb + j] = -tmax[j];
                        in.tub[(size_t)i * L.len_tub + j] = tmax[j];
                    }
            }
            const double* reference_data() const override { return cur_ ? cur_->ref_.data() : nullptr; }
            void com(MatrixXd& pos, MatrixXd& vel) const override { pos = com_pos_; vel = com_vel_; }

            void set_com_ref(const TrajectorySample& s) override
            {
                IWBC_ASSERT(cur_, "ModelSource is not bound to a solver");
                IWBC_BOUND_ALIASES(*cur_)
                for (size_t i = 0; i < tasks_.size(); ++i)
                    if (tasks_[i].kind == WBCQP_T_COM) {
                        std::vector<double> r(9);
                        for (int d = 0; d < 3; ++d) { r[d] = s.pos[d]; r[3 + d] = s.vel[d]; r[6 + d] = s.acc[d]; }
                        store(names_[i], tasks_[i].ref, r);
                    }
            }
            // sample.pos: 12 numbers (translation, rotation column-major), vel / acc: 6 each (PosTracker::set_se3_ref, pos_tracker.cpp:221-237)
            void set_se3_ref(const std::string& name, const TrajectorySample& s) override
            {
                IWBC_ASSERT(cur_, "ModelSource is not bound to a solver");
                IWBC_BOUND_ALIASES(*cur_)
                for (size_t i = 0; i < tasks_.size(); ++i)
                    if (names_[i] == name && tasks_[i].kind == WBCQP_T_SE3) {
                        IWBC_ASSERT(s.pos.size() == 12 && s.vel.size() == 6 && s.acc.size() == 6, "an SE3 sample holds 12 + 6 + 6 numbers");
                        std::vector<double> r(s.pos);
                        r.insert(r.end(), s.vel.begin(), s.vel.end());
                        r.insert(r.end(), s.acc.begin(), s.acc.end());
                        store(name, tasks_[i].ref, r);
                        return;
                    }
                IWBC_ERROR("Task [", name, "] not found");
            }
            std::vector<double> get_se3_ref(const std::string& name) const override
            {
                IWBC_ASSERT(cur_, "ModelSource is not bound to a solver");
                IWBC_BOUND_ALIASES(*cur_)
                for (size_t i = 0; i < tasks_.size(); ++i)
                    if (names_[i] == name && tasks_[i].kind == WBCQP_T_SE3) return std::vector<double>(ref_.begin() + tasks_[i].ref, ref_.begin() + tasks_[i].ref + 12);
                IWBC_ERROR("Task [", name, "] not found");
            }
            // PosTracker::set_contact_se3_ref (pos_tracker.cpp:227-232): kept for contacts that are currently removed too
            void set_contact_se3_ref(const std::string& name, const std::vector<double>& sample) override
            {
                IWBC_ASSERT(cur_, "ModelSource is not bound to a solver");
                IWBC_BOUND_ALIASES(*cur_)
                IWBC_ASSERT(sample.size() == 12 || sample.size() == 24, "a contact reference holds 12 numbers (placement) or 24 (with velocity and acceleration)");
                std::vector<double> pose(sample);
                pose.resize(24, 0.0); // to_sample(SE3): zero derivatives (pos_tracker.cpp:227-232)
                named_[name] = pose;
                for (size_t c = 0; c < contact_names_.size(); ++c)
                    if (contact_names_[c] == name)
                        for (int i = 0; i < batch_; ++i) std::copy(pose.begin(), pose.end(), ref_.begin() + (size_t)i * nref_ + contact_ref_[c]);
            }
            void set_posture_ref(const std::vector<double>& q_actuated) override
            {
                IWBC_ASSERT(cur_, "ModelSource is not bound to a solver");
                IWBC_BOUND_ALIASES(*cur_)
                IWBC_ASSERT((int)q_actuated.size() == robot_->na(), "the posture reference holds na entries");
                posture_user_ = q_actuated;
                for (int i = 0; i < batch_; ++i) std::copy(q_actuated.begin(), q_actuated.end(), ref_.begin() + (size_t)i * nref_ + posture_ref_);
            }
            const std::vector<double>& references() const { return cur_->ref_; }
            int nref() const { return cur_ ? cur_->nref_ : 0; }

        private:
            void store(const std::string& name, int off, const std::vector<double>& r)
            {
                IWBC_ASSERT(cur_, "ModelSource is not bound to a solver");
                IWBC_BOUND_ALIASES(*cur_)
                named_[name] = r;
                for (int i = 0; i < batch_; ++i) std::copy(r.begin(), r.end(), ref_.begin() + (size_t)i * nref_ + off);
            }

            // a slot seen before: nothing goes to the device, the references set meanwhile are laid out for this stack
            void refresh(Bound& b)
            {
                IWBC_BOUND_ALIASES(b)
                auto put = [&](int off, const std::vector<double>& r) {
                    for (int i = 0; i < batch_; ++i) std::copy(r.begin(), r.end(), ref_.begin() + (size_t)i * nref_ + off);
                };
                for (size_t i = 0; i < tasks_.size(); ++i) {
                    auto it = named_.find(names_[i]);
                    if (it != named_.end() && tasks_[i].kind != WBCQP_T_SELFCOLLISION) put(tasks_[i].ref, it->second);
                }
                for (size_t c = 0; c < contact_names_.size(); ++c) {
                    auto it = named_.find(contact_names_[c]);
                    if (it != named_.end()) put(contact_ref_[c], it->second);
                }
                if (!posture_user_.empty()) put(posture_ref_, posture_user_);
            }

            std::shared_ptr<robots::RobotWrapper> robot_;
            int batch_ = 0, slot_ = 0;
            std::vector<double> q0_, posture_user_;
            wbcqp_handle* handle_ = nullptr;
            std::map<int, Bound> bounds_;
            Bound* cur_ = nullptr;
            std::map<std::string, std::vector<double>> named_;
            MatrixXd com_pos_, com_vel_;
        };
#undef IWBC_BOUND_ALIASES
    } // namespace controllers
} // namespace inria_wbc
#endif
