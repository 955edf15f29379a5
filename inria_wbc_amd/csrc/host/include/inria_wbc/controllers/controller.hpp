// Batched stand-in for inria_wbc::controllers::Controller (/root/reference/include/inria_wbc/controllers/controller.hpp:46-233,
// /root/reference/src/controllers/controller.cpp:161-313): same method names and tick contract, but every getter
// returns one row per robot instance.  The per-tick hot path -- computeProblemData's assembly + solver_->solve +
// decode (controller.cpp:244-251) -- goes through the C ABI of libwbcqp (include/wbcqp.h) in ONE call for all
// B instances.  What the reference gets from pinocchio and from each task's compute() (the step before the path)
// is supplied by a ProblemSource.
#ifndef IWBC_HIP_CONTROLLER_HPP
#define IWBC_HIP_CONTROLLER_HPP

#include <algorithm>
#include <cmath>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include <inria_wbc/controllers/task_stack.hpp>
#include <inria_wbc/exceptions.hpp>
#include <inria_wbc/utils/factory.hpp>
#include <inria_wbc/utils/yaml_lite.hpp>

#include "wbcqp.h"

namespace inria_wbc {
    namespace controllers {

        // dense row-major matrix: [rows = instances][cols]
        struct MatrixXd {
            int rows = 0, cols = 0;
            std::vector<double> data;
            MatrixXd() = default;
            MatrixXd(int r, int c, double v = 0.0) : rows(r), cols(c), data((size_t)r * c, v) {}
            double& operator()(int r, int c) { return data[(size_t)r * cols + c]; }
            double operator()(int r, int c) const { return data[(size_t)r * cols + c]; }
            double* row(int r) { return data.data() + (size_t)r * cols; }
            const double* row(int r) const { return data.data() + (size_t)r * cols; }
        };
        using VectorXi = std::vector<int>;

        // same keys as the reference (controller.cpp:165-187): "positions", "joint_velocities", "floating_base_position", ...
        using SensorData = std::unordered_map<std::string, MatrixXd>;

        struct behavior_types {
            static constexpr const char* FIXED_BASE = "FIXED_BASE";
            static constexpr const char* SINGLE_SUPPORT = "SINGLE_SUPPORT";
            static constexpr const char* DOUBLE_SUPPORT = "DOUBLE_SUPPORT";
        };

        // Per-tick inputs of the path for B instances, laid out exactly as wbcqp_inputs wants them ([B][len], row-major).
        struct TickInputs {
            int batch = 0;
            std::vector<double> M, h, A, b1, Ac, bc, blb, bub, tlb, tub, w, Acop;
            // sizes the arrays for B instances; contents are left alone when nothing changed (every tick overwrites what it uses)
            void resize(int B, const wbcqp_layout& L)
            {
                batch = B;
                auto fit = [B](std::vector<double>& a, int len) {
                    if (a.size() != (size_t)B * len) a.assign((size_t)B * len, 0.0);
                };
                fit(M, L.len_M); fit(h, L.len_h); fit(A, L.len_A); fit(b1, L.len_b1); fit(Ac, L.len_Ac); fit(bc, L.len_bc);
                fit(blb, L.len_blb); fit(bub, L.len_bub); fit(tlb, L.len_tlb); fit(tub, L.len_tub); fit(w, L.len_w); fit(Acop, L.len_Acop);
            }
        };

        // TrajectorySample of tsid: value / first / second derivative
        struct TrajectorySample {
            std::vector<double> pos, vel, acc;
            explicit TrajectorySample(int n = 0) : pos(n, 0.0), vel(n, 0.0), acc(n, 0.0) {}
        };

        // "The step before the path": rigid-body terms and task rows for every instance at time t (pinocchio computeAllTerms +
        // each task's compute() in the reference, controller.cpp:244 upstream half).  Two kinds: FileSource replays rows
        // computed elsewhere and leaves the reference-driven right-hand sides to the controller; ModelSource
        // (model_source.hpp) computes everything on the device from the robot state (wbcqp_problem_data) and therefore
        // owns the task references too.
        class ProblemSource {
        public:
            virtual ~ProblemSource() {}
            virtual int batch() const = 0;
            // fills every array of `in` except the reference-driven parts the controller adds afterwards
            virtual void compute(double t, const MatrixXd& q, const MatrixXd& v, const tasks::TaskStack& stack, const wbcqp_layout& L, TickInputs& in) = 0;
            // current CoM position / velocity per instance (3 columns) for the CoM task's PD law
            virtual void com(MatrixXd& pos, MatrixXd& vel) const = 0;
            // sources that evaluate the task laws themselves
            virtual bool handles_references() const { return false; }
            virtual void bind(wbcqp_handle*, int, const tasks::TaskStack&, double) {}
            virtual void set_com_ref(const TrajectorySample&) {}
            virtual void set_se3_ref(const std::string&, const TrajectorySample&) {}
            virtual void set_posture_ref(const std::vector<double>&) {}
            virtual std::vector<double> get_se3_ref(const std::string&) const { return {}; } // 12 numbers, SE3ToVector order
            virtual void set_contact_se3_ref(const std::string&, const std::vector<double>&) {}  // 12 numbers, or 24 with derivatives
            virtual const double* reference_data() const { return nullptr; }                   // [batch][nref] for wbcqp_tick_host
            virtual void fill_limits(const wbcqp_layout&, TickInputs&) const {}                // constant tlb / tub
        };

        class Controller {
        public:
            explicit Controller(const yaml::Node& config)
            {
                yaml::Node c = IWBC_CHECK(config["CONTROLLER"]);
                base_path_ = c["base_path"] ? c["base_path"].as<std::string>() : std::string(".");
                dt_ = IWBC_CHECK(c["dt"].as<double>());
                floating_base_ = IWBC_CHECK(c["floating_base"].as<bool>());
                verbose_ = c["verbose"] ? c["verbose"].as<bool>() : false;
                // controller.cpp:63,72 (the reference requires both keys; here they default to what a model without mimic joints has)
                fb_joint_name_ = c["floating_base_joint_name"] ? c["floating_base_joint_name"].as<std::string>() : std::string("root_joint");
                if (c["mimic_dof_names"]) mimic_dof_names_ = c["mimic_dof_names"].as<std::vector<std::string>>();
                t_ = 0.0;
            }
            Controller(const Controller&) = delete;
            Controller& operator=(const Controller&) = delete;
            virtual ~Controller()
            {
                if (handle_) wbcqp_destroy(handle_);
            }

            const std::string& base_path() const { return base_path_; }
            // Controller::update (controller.cpp:161-205): open loop integrates the controller's own state; closed loop REQUIRES
            // the sensor keys the reference requires and builds q = [floating_base_position, positions],
            // dq = [floating_base_velocity, joint_velocities] per instance (one row per instance instead of one vector)
            virtual void update(const SensorData& sensor_data = {})
            {
                if (!closed_loop_) {
                    _solve();
                    return;
                }
                auto qi = sensor_data.find("positions"), vi = sensor_data.find("joint_velocities");
                IWBC_ASSERT(qi != sensor_data.end(), "we need the joint positions in closed loop mode!");
                IWBC_ASSERT(vi != sensor_data.end(), "we need the joint velocities in closed loop mode!");
                const MatrixXd &pos = qi->second, &vel = vi->second;
                IWBC_ASSERT(pos.rows == batch_ && vel.rows == batch_, "closed loop: one sensor row per instance (", batch_, ")");
                if (!floating_base_) {
                    IWBC_ASSERT(vel.cols == v_tsid_.cols, "Joint velocities do not have the correct size:", vel.cols, " vs (expected)", v_tsid_.cols);
                    IWBC_ASSERT(pos.cols == q_tsid_.cols, "Joint positions do not have the correct size:", pos.cols, " vs (expected)", q_tsid_.cols);
                    _solve(pos, vel);
                    return;
                }
                auto fqi = sensor_data.find("floating_base_position"), fvi = sensor_data.find("floating_base_velocity");
                IWBC_ASSERT(fqi != sensor_data.end(), "we need the floating base position in closed loop mode!");
                IWBC_ASSERT(fvi != sensor_data.end(), "we need the floating base velocity in closed loop mode!");
                const MatrixXd &fb_pos = fqi->second, &fb_vel = fvi->second;
                IWBC_ASSERT(fb_pos.rows == batch_ && fb_vel.rows == batch_, "closed loop: one floating-base row per instance (", batch_, ")");
                IWBC_ASSERT(vel.cols + fb_vel.cols == v_tsid_.cols, "Joint velocities do not have the correct size:", vel.cols + fb_vel.cols,
                            " vs (expected)", v_tsid_.cols);
                IWBC_ASSERT(pos.cols + fb_pos.cols == q_tsid_.cols, "Joint positions do not have the correct size:", pos.cols + fb_pos.cols,
                            " vs (expected)", q_tsid_.cols);
                MatrixXd q(batch_, q_tsid_.cols), dq(batch_, v_tsid_.cols);
                for (int i = 0; i < batch_; ++i) {
                    std::copy(fb_pos.row(i), fb_pos.row(i) + fb_pos.cols, q.row(i));
                    std::copy(pos.row(i), pos.row(i) + pos.cols, q.row(i) + fb_pos.cols);
                    std::copy(fb_vel.row(i), fb_vel.row(i) + fb_vel.cols, dq.row(i));
                    std::copy(vel.row(i), vel.row(i) + vel.cols, dq.row(i) + fb_vel.cols);
                }
                _solve(q, dq);
            }

            int batch_size() const { return batch_; }
            double t() const { return t_; }
            double dt() const { return dt_; }
            bool verbose() const { return verbose_; }
            void set_verbose(bool b) { verbose_ = b; }
            virtual void set_behavior_type(const std::string& bt) { behavior_type_ = bt; }
            const std::string& behavior_type() const { return behavior_type_; }

            // one row per instance (the reference returns one Eigen::VectorXd), in the reference's "DART" format
            // (controller.cpp:262-281): ndofs = nv entries; a floating base is [position(3), angle * axis(3)] in q and six
            // leading zeros in tau.  filter_mimics (default true, as in the reference, controller.cpp:369-397) drops the columns of
            // the joints named in CONTROLLER.mimic_dof_names: slice_vec(x, non_mimic_indexes_) per instance.
            MatrixXd tau(bool filter_mimics = true) const { return filter_mimics ? filter_cmd(tau_dart_) : tau_dart_; }
            MatrixXd ddq(bool filter_mimics = true) const { return filter_mimics ? filter_cmd(a_tsid_) : a_tsid_; }
            MatrixXd dq(bool filter_mimics = true) const { return filter_mimics ? filter_cmd(v_tsid_) : v_tsid_; }
            MatrixXd q(bool filter_mimics = true) const { return filter_mimics ? filter_cmd(q_solver_) : q_solver_; }
            MatrixXd q_solver(bool filter_mimics = true) const { return filter_mimics ? filter_cmd(q_solver_) : q_solver_; }
            // controller.cpp:245: momentum_ = momentumJacobian(data).bottomRows(3) * dq -- the angular momentum about the CoM of the
            // state the last tick was solved at, one row per instance (zero until a tick has run on a model-driven source)
            const MatrixXd& momentum() const { return momentum_; }
            // controller.hpp:60-64,114-115 of the reference
            const std::vector<std::string>& mimic_names() const { return mimic_dof_names_; }
            const std::vector<int>& non_mimic_indexes() const { return non_mimic_indexes_; }
            MatrixXd filter_cmd(const MatrixXd& cmd) const
            {
                if (non_mimic_indexes_.empty() || (int)non_mimic_indexes_.size() == cmd.cols) return cmd;
                MatrixXd out(cmd.rows, (int)non_mimic_indexes_.size());
                for (int i = 0; i < cmd.rows; ++i)
                    for (size_t j = 0; j < non_mimic_indexes_.size(); ++j) out(i, (int)j) = cmd(i, non_mimic_indexes_[j]);
                return out;
            }
            // Removes the universe and root (floating base) joint names (controller.cpp:316-331)
            std::vector<std::string> controllable_dofs(bool filter_mimics = true) const
            {
                std::vector<std::string> out;
                for (const auto& n : joint_names_)
                    if (!filter_mimics || std::find(mimic_dof_names_.begin(), mimic_dof_names_.end(), n) == mimic_dof_names_.end()) out.push_back(n);
                return out;
            }
            // Order of the floating base in q_ according to dart naming convention (controller.cpp:333-363)
            std::vector<std::string> floating_base_dofs() const
            {
                if (!floating_base_) return {};
                const std::string b = (fb_joint_name_ == "root_joint") ? std::string("rootJoint") : fb_joint_name_;
                return {b + "_pos_x", b + "_pos_y", b + "_pos_z", b + "_rot_x", b + "_rot_y", b + "_rot_z"};
            }
            std::vector<std::string> all_dofs(bool filter_mimics = true) const
            {
                std::vector<std::string> all = floating_base_dofs(), ctl = controllable_dofs(filter_mimics);
                all.insert(all.end(), ctl.begin(), ctl.end());
                return all;
            }
            // controller.cpp:445-450 / controller.hpp:170-171: back to the state the last tick started from (the reference also
            // restores pinocchio's data; here every tick recomputes the rigid-body terms from (q, dq) on the device)
            void qp_step_back(const MatrixXd& q, const MatrixXd& dq)
            {
                IWBC_ASSERT(q.rows == batch_ && dq.rows == batch_ && q.cols == q_tsid_.cols && dq.cols == v_tsid_.cols, "qp_step_back: wrong size");
                q_tsid_ = q;
                v_tsid_ = dq;
            }
            void qp_step_back()
            {
                if (q_tsid_prev_.rows == batch_ && batch_ > 0) qp_step_back(q_tsid_prev_, v_tsid_prev_);
            }
            // tsid's own forms: q with the base quaternion (nq = nv + 1 entries), tau of the actuated joints only (na entries)
            const MatrixXd& q_tsid() const { return q_tsid_; }
            const MatrixXd& tau_tsid() const { return tau_; }
            const std::vector<std::string>& activated_contacts() const { return activated_contacts_; }
            // 12 force components per activated contact and instance (tsid getContactForces, controller.cpp:258-261)
            const std::unordered_map<std::string, MatrixXd>& activated_contacts_forces() const { return activated_contacts_forces_; }
            const VectorXi& qp_status() const { return status_; }
            const VectorXi& qp_iterations() const { return iters_; }
            virtual double cost(const std::string& task_name) const = 0;

            void set_problem_source(const std::shared_ptr<ProblemSource>& src)
            {
                IWBC_ASSERT(src, "Invalid problem source");
                source_ = src;
                batch_ = src->batch();
                _reset();
            }

        protected:
            virtual void _reset() {}
            // builds this tick's inputs (source + references) -- implemented by PosTracker
            virtual void _build_inputs(const MatrixXd& q, const MatrixXd& v) = 0;
            virtual const tasks::TaskStack& _stack() const = 0;
            virtual int _slot() const = 0;
            virtual const wbcqp_layout& _layout() const = 0;

            void _solve() { _solve(MatrixXd(q_tsid_), MatrixXd(v_tsid_)); }

            // tsid_joint_names_ = all_dofs(false), non_mimic_indexes_ = get_non_mimics_indexes() (controller.cpp:157-158,208-229):
            // called by the derived controller once the model's joint names are known (without a model: no names, no mimic joints)
            void _set_joint_names(const std::vector<std::string>& actuated_joint_names)
            {
                joint_names_ = actuated_joint_names;
                const std::vector<std::string> tsid_joint_names = all_dofs(false);
                std::vector<int> mimic_indexes;
                for (const auto& m : mimic_dof_names_) {
                    auto it = std::find(tsid_joint_names.begin(), tsid_joint_names.end(), m);
                    IWBC_ASSERT(it != tsid_joint_names.end(), " joint ", m, " not found");
                    mimic_indexes.push_back((int)std::distance(tsid_joint_names.begin(), it));
                }
                non_mimic_indexes_.clear();
                for (int i = 0; i < (int)tsid_joint_names.size(); ++i)
                    if (std::find(mimic_indexes.begin(), mimic_indexes.end(), i) == mimic_indexes.end()) non_mimic_indexes_.push_back(i);
            }

            // Controller::_solve (controller.cpp:231-313) for B instances
            void _solve(const MatrixXd& q, const MatrixXd& dq)
            {
                IWBC_ASSERT(handle_, "no solver: the controller was not fully constructed");
                IWBC_ASSERT(source_, "no problem source set (set_problem_source)");
                const tasks::TaskStack& st = _stack();
                const wbcqp_layout& L = _layout();
                _build_inputs(q, dq);
                const int B = batch_, n = L.n, na = st.na(), nv = st.nv();
                x_.assign((size_t)B * n, 0.0);
                tau_ = MatrixXd(B, na);
                status_.assign(B, WBCQP_HQP_UNKNOWN);
                iters_.assign(B, 0);
                objective_.assign(B, 0.0);
                wbcqp_inputs in = {in_.M.data(), in_.h.data(), in_.A.data(), in_.b1.data(), in_.Ac.data(), in_.bc.data(),
                                   in_.blb.data(), in_.bub.data(), in_.tlb.data(), in_.tub.data(), in_.w.data(), in_.Acop.data()};
                wbcqp_outputs out = {x_.data(), tau_.data.data(), status_.data(), iters_.data(), objective_.data(), nullptr};
                IWBC_ASSERT(q.cols == (floating_base_ ? nv + 1 : nv), "q must hold ", floating_base_ ? nv + 1 : nv, " entries per instance");
                q_tsid_prev_ = q; // controller.cpp:237-241 (the reference keeps them when send_cmd_ is set; qp_step_back() returns here)
                v_tsid_prev_ = dq;
                momentum_ = MatrixXd(B, 3);
                std::vector<double> mom6((size_t)B * 6, 0.0);
                MatrixXd vnew(B, nv);
                MatrixXd qnew(B, q.cols);
                MatrixXd qsol(B, nv);
                const bool whole_tick = source_->handles_references();
                if (whole_tick) {
                    // rows, QP and integration in one trip to the device: only the state and the references go up, the solution
                    // and the integrated state come back.  The rows stay on the device; cost() fetches them when somebody asks.
                    wbcqp_tick_io io{};
                    io.state = {q.data.data(), dq.data.data(), source_->reference_data(), mom6.data()};
                    io.rows = in;
                    io.rows.M = io.rows.h = io.rows.A = io.rows.b1 = io.rows.Ac = io.rows.bc = io.rows.blb = io.rows.bub = io.rows.Acop = nullptr;
                    last_q_ = q;
                    last_v_ = dq;
                    rows_valid_ = false;
                    io.out = out;
                    io.q_next = qnew.data.data();
                    io.v_next = vnew.data.data();
                    io.q_solver = qsol.data.data();
                    io.dt = dt_;
                    if (wbcqp_tick_host(handle_, _slot(), B, &io) != WBCQP_OK) IWBC_ERROR("wbcqp_tick_host failed: ", wbcqp_last_error(handle_));
                    for (int i = 0; i < B; ++i)
                        for (int k = 0; k < 3; ++k) momentum_(i, k) = mom6[(size_t)i * 6 + 3 + k];
                }
                else {
                    int rc = wbcqp_solve_batch_host(handle_, _slot(), B, &in, &out);
                    if (rc != WBCQP_OK) IWBC_ERROR("wbcqp_solve_batch_host failed: ", wbcqp_last_error(handle_));
                    rows_valid_ = true;
                }
                for (int i = 0; i < B; ++i) {
                    if (status_[i] != WBCQP_HQP_OPTIMAL) {
                        // same text as controller.cpp:285-307, with the instance appended
                        std::string error = "Controller failed, can't solve problem. ";
                        error += "Status : " + std::to_string(status_[i]);
                        switch (status_[i]) {
                        case -1: error += " => Unknown"; break;
                        case 1: error += " => Infeasible "; break;
                        case 2: error += " => Unbounded "; break;
                        case 3: error += " => Max iter reached "; break;
                        case 4: error += " => Error "; break;
                        default: error += " => Uknown status";
                        }
                        error += " (t=" + std::to_string(t_) + ")";
                        error += " [instance " + std::to_string(i) + "]";
                        throw IWBC_EXCEPTION(error);
                    }
                }
                // a_tsid = dv ; v_tsid = dq + dt dv ; q = integrate(q, dt v)  (controller.cpp:250-256): on the device
                a_tsid_ = MatrixXd(B, nv);
                for (int i = 0; i < B; ++i)
                    for (int j = 0; j < nv; ++j) a_tsid_(i, j) = x_[(size_t)i * n + j];
                if (!whole_tick &&
                    wbcqp_integrate_host(handle_, B, nv, floating_base_ ? 1 : 0, dt_, q.data.data(), dq.data.data(), x_.data(), n,
                                         status_.data(), qnew.data.data(), vnew.data.data(), qsol.data.data()) != WBCQP_OK)
                    IWBC_ERROR(wbcqp_last_error(handle_));
                v_tsid_ = vnew;
                q_tsid_ = qnew;
                q_solver_ = qsol;
                // tau_ << 0, 0, 0, 0, 0, 0, tau_tsid_ for a floating base (controller.cpp:269); tau_tsid_ itself otherwise
                tau_dart_ = MatrixXd(B, nv);
                for (int i = 0; i < B; ++i)
                    for (int j = 0; j < na; ++j) tau_dart_(i, nv - na + j) = tau_(i, j);
                t_ += dt_;
                activated_contacts_forces_.clear();
                for (size_t c = 0; c < st.contacts().size(); ++c) {
                    MatrixXd f(B, 12);
                    for (int i = 0; i < B; ++i)
                        for (int m = 0; m < 12; ++m) f(i, m) = x_[(size_t)i * n + nv + 12 * c + m];
                    activated_contacts_forces_[st.contacts()[c].name] = f;
                }
            }

            bool verbose_ = false;
            double t_ = 0.0, dt_ = 0.001;
            bool floating_base_ = true;
            bool closed_loop_ = false;
            int solver_max_iter_ = 1000; // eiquadprog-fast DEFAULT_MAX_ITER
            std::string base_path_, behavior_type_, solver_to_use_;
            int batch_ = 0;

            MatrixXd q_tsid_, v_tsid_, a_tsid_, tau_, tau_dart_, q_solver_, momentum_, q_tsid_prev_, v_tsid_prev_;
            std::string fb_joint_name_;
            std::vector<std::string> mimic_dof_names_, joint_names_; // joint_names_: the 1-dof joints in model order (no universe, no root)
            std::vector<int> non_mimic_indexes_;
            std::vector<double> x_, objective_;
            VectorXi status_, iters_;
            std::vector<std::string> activated_contacts_, all_contacts_;
            std::unordered_map<std::string, MatrixXd> activated_contacts_forces_;

            // the rows of the last tick on the host, for cost(): fetched on demand when the tick computed them on the device
            void _ensure_rows() const
            {
                if (rows_valid_ || !source_ || in_.batch == 0) return;
                auto self = const_cast<Controller*>(this);
                self->source_->compute(t_ - dt_, last_q_, last_v_, _stack(), _layout(), self->in_);
                self->rows_valid_ = true;
            }

            std::shared_ptr<ProblemSource> source_;
            TickInputs in_;
            MatrixXd last_q_, last_v_;
            bool rows_valid_ = false;
            wbcqp_handle* handle_ = nullptr; // the reference's solver_ (controller.hpp:224)
        };

        using Factory = utils::Factory<Controller, yaml::Node>;
        template <typename T>
        using Register = Factory::AutoRegister<T>;
    } // namespace controllers
} // namespace inria_wbc
#endif
