// copra.h -- host-side C++ mirror of copra's controller API on top of the C ABI (include/copra_hip.h).
//
// Same names, constructor arguments and error behaviour as the reference headers
//   include/PreviewSystem.h, costFunctions.h, constraints.h, LMPC.h, SolverInterface.h, solverUtils.h, AutoSpan.h
// so that code (and tests) written against copra compile against this header; all arithmetic of LMPC::solve runs on
// the GPU (one fused launch), nothing is computed on the host.  Differences, all documented in DESIGN.md:
//   * the nine built-in cost / constraint classes hand the device a DESCRIPTOR (their constructor arguments): the
//     kernels evaluate them, nothing of them is computed on the host.  Their update() / Q() c() E() f() / A() b() Y() z()
//     exist with the reference's signatures and, when a caller asks for them, are evaluated ON THE DEVICE as well;
//   * user subclasses of CostFunction / EqIneqConstraint / ControlBoundConstraint (plug-in point 2) are supported the way
//     the reference runs them: LMPC::solve calls their update(const PreviewSystem&) on the host, then hands the resulting
//     dense Q c E f / A b Y z to the device (COPRA_COST_DENSE / COPRA_CSTR_DENSE), where they join the same fused solve;
//   * LMPC::useSolver installs a user SolverInterface (plug-in point 1): the QP is then condensed on the device, copied
//     out and handed to SI_problem / SI_solve / SI_result exactly as src/LMPC.cpp:79-101 does;
//   * PreviewSystem::updateSystem() computes Phi / Psi / xi on the device (copra_preview_update); the solve path itself
//     never materialises them;
//   * every solve() has fresh-controller semantics (reference quirk Q2 -- accumulation across solves -- is not kept).
#pragma once

#include "../../../../include/copra_hip.h"
#include "../../../csrc/plan_builder.hpp" // dimension checks identical to copra_batch_create (pure host code)
#include "EigenCompat.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <limits>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace copra {

// include/debugUtils.h:32-42
#define COPRA_DOMAIN_ERROR(msg)                                                                                       \
    throw std::domain_error(std::string("In file ") + __FILE__ + " (line " + std::to_string(__LINE__)                 \
        + "): [In function: " + __func__ + "]\n" + (msg))
#define COPRA_RUNTIME_ERROR(msg)                                                                                      \
    throw std::runtime_error(std::string("In file ") + __FILE__ + " (line " + std::to_string(__LINE__)                \
        + "): [In function: " + __func__ + "]\n" + (msg))

inline void throw_status(copra_status_t rc)
{
    if (rc == COPRA_OK) return;
    const std::string msg = copra_last_error();
    if (rc == COPRA_ERR_DOMAIN) throw std::domain_error(msg);
    throw std::runtime_error(msg);
}

// ---------------------------------------------------------------------------------------------- AutoSpan.h
struct AutoSpan {
    AutoSpan() = delete;
    static void spanMatrix(Eigen::MatrixXd& mat, Eigen::Index new_dim, int addCols = 0) // src/AutoSpan.cpp:10-28
    {
        const auto rows = mat.rows(), cols = mat.cols();
        if (new_dim == rows) return;
        const auto steps = new_dim / rows;
        if (steps * rows != new_dim) COPRA_DOMAIN_ERROR("spanMatrix: the new dimension is not a multiple of the rows");
        Eigen::MatrixXd out = Eigen::MatrixXd::Zero(new_dim, cols * (steps + addCols));
        for (Eigen::Index s = 0; s < steps; ++s)
            for (Eigen::Index j = 0; j < cols; ++j)
                for (Eigen::Index i = 0; i < rows; ++i) out(s * rows + i, s * cols + j) = mat(i, j);
        mat = out;
    }
    static void spanVector(Eigen::VectorXd& vec, Eigen::Index new_dim) // src/AutoSpan.cpp:30-47
    {
        const auto rows = vec.rows();
        if (new_dim == rows) return;
        const auto steps = new_dim / rows;
        if (steps * rows != new_dim) COPRA_DOMAIN_ERROR("spanVector: the new dimension is not a multiple of the size");
        Eigen::VectorXd out(new_dim);
        for (Eigen::Index s = 0; s < steps; ++s)
            for (Eigen::Index i = 0; i < rows; ++i) out(s * rows + i) = vec(i);
        vec = out;
    }
};

// ---------------------------------------------------------------------------------------------- PreviewSystem.h
struct PreviewSystem {
    PreviewSystem() = default;
    PreviewSystem(const Eigen::MatrixXd& state, const Eigen::MatrixXd& control, const Eigen::VectorXd& bias,
        const Eigen::VectorXd& xInit, int numberOfSteps)
    {
        system(state, control, bias, xInit, numberOfSteps);
    }
    void system(const Eigen::MatrixXd& state, const Eigen::MatrixXd& control, const Eigen::VectorXd& bias,
        const Eigen::VectorXd& xInit, int numberOfSteps) // src/PreviewSystem.cpp:16-55
    {
        if (xInit.rows() != state.rows()) COPRA_DOMAIN_ERROR("xInit and state should have the same number of rows");
        if (state.rows() != state.cols()) COPRA_DOMAIN_ERROR("state should be a square matrix");
        if (xInit.rows() != control.rows()) COPRA_DOMAIN_ERROR("xInit and control should have the same number of rows");
        if (xInit.rows() != bias.rows()) COPRA_DOMAIN_ERROR("xInit and bias should have the same number of rows");
        if (numberOfSteps <= 0) COPRA_DOMAIN_ERROR("The number of step sould be a positive number! ");
        isUpdated = false;
        previewOnHost = false;
        nrUStep = numberOfSteps;
        nrXStep = numberOfSteps + 1;
        xDim = (int)state.cols();
        uDim = (int)control.cols();
        fullXDim = xDim * nrXStep;
        fullUDim = uDim * nrUStep;
        x0 = xInit;
        A = state;
        B = control;
        d = bias;
    }
    // src/PreviewSystem.cpp:57-74 -- Phi, Psi, xi through the device (copra_preview_update).  LMPC::solve only calls this
    // when a host-evaluated user cost / constraint / solver needs the matrices.
    void updateSystem() noexcept
    {
        Phi.resize(fullXDim, xDim);
        Psi.resize(fullXDim, fullUDim);
        xi.resize(fullXDim);
        if (copra_preview_update(xDim, uDim, nrUStep, A.data(), B.data(), d.data(), Phi.data(), Psi.data(), xi.data()) == COPRA_OK) {
            isUpdated = true;
            previewOnHost = true;
        }
    }
    void xInit(const Eigen::VectorXd& xInit) { x0 = xInit; }

    bool isUpdated = false;
    bool previewOnHost = false; // Phi / Psi / xi below hold the matrices of the current (A, B, d)
    int nrUStep = 0, nrXStep = 0, xDim = 0, uDim = 0, fullXDim = 0, fullUDim = 0;
    Eigen::VectorXd x0;
    Eigen::MatrixXd A, B;
    Eigen::VectorXd d;
    Eigen::VectorXd xi; // include/PreviewSystem.h:58-60
    Eigen::MatrixXd Phi, Psi;
};

namespace detail {
    struct HandleGuard {
        copra_batch_t* h = nullptr;
        ~HandleGuard()
        {
            if (h) copra_batch_destroy(h);
        }
    };
    inline copra_dims_t dims_of(const PreviewSystem& ps) { return copra_dims_t { ps.xDim, ps.uDim, ps.nrUStep, 1 }; }
    // the checks of initializeCost / initializeConstraint (plan_builder.hpp == copra_batch_create), without the device
    inline void check_pieces(const PreviewSystem& ps, const std::vector<copra_cost_desc_t>& costs,
        const std::vector<copra_cstr_desc_t>& cstrs, copra_hip::HostPlan* out = nullptr)
    {
        copra_hip::HostPlan hp;
        const copra_dims_t dims = dims_of(ps);
        const copra_status_t rc = copra_hip::build_plan(hp, dims, (int)costs.size(), costs.data(), (int)cstrs.size(), cstrs.data());
        if (rc == COPRA_ERR_DOMAIN) throw std::domain_error(hp.error);
        if (rc == COPRA_ERR_RUNTIME) throw std::runtime_error(hp.error);
        if (out) *out = hp;
    }
    // Q_, c_, E_, f_ of ONE built-in cost, evaluated by the device's condense code (the parity hook copra_batch_dump_qp):
    // the LMPC form gives Q (minus LMPC::updateSystem's 1e-6 I, src/LMPC.cpp:228-230) and c, the InitialStateLMPC form
    // gives E (top-right block of its Hessian) and f (tail of its linear term), src/InitialStateLMPC.cpp:80-84
    inline void evaluate_cost_on_device(const PreviewSystem& ps, const copra_cost_desc_t& d, Eigen::MatrixXd& Q,
        Eigen::VectorXd& c, Eigen::MatrixXd& E, Eigen::VectorXd& f)
    {
        const int n = ps.fullUDim, nx = ps.xDim;
        const copra_dims_t dims = dims_of(ps);
        Q.resize(n, n), c.resize(n), E.resize(nx, n), f.resize(n);
        {
            HandleGuard g;
            throw_status(copra_batch_create(&g.h, &dims, 1, &d, 0, nullptr));
            throw_status(copra_batch_set_system(g.h, ps.A.data(), ps.B.data(), ps.d.data(), ps.x0.data(), 0));
            throw_status(copra_batch_dump_qp(g.h, 0, Q.data(), c.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
            for (int i = 0; i < n; ++i) Q(i, i) -= 1e-6;
        }
        Eigen::MatrixXd R = Eigen::MatrixXd::Identity(nx, nx), H(nx + n, nx + n);
        Eigen::VectorXd r = Eigen::VectorXd::Zero(nx), g2(nx + n);
        copra_initial_state_desc_t is { R.data(), r.data() };
        HandleGuard g;
        if (copra_batch_create_initial_state(&g.h, &dims, 1, &d, 0, nullptr, &is) != COPRA_OK) return; // (xDim > 16: E, f stay zero)
        throw_status(copra_batch_set_system(g.h, ps.A.data(), ps.B.data(), ps.d.data(), ps.x0.data(), 0));
        throw_status(copra_batch_dump_qp(g.h, 0, H.data(), g2.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
        for (int j = 0; j < n; ++j) {
            f(j) = g2(nx + j);
            for (int a = 0; a < nx; ++a) E(a, j) = H(a, nx + j);
        }
    }
    // A_, b_, Y_, z_ of ONE built-in equality / inequality constraint, the same way: [A | b] from the LMPC form
    // (src/LMPC.cpp:257-271), [Y, A | z] from the InitialStateLMPC form (src/InitialStateLMPC.cpp:88-102)
    inline void evaluate_constraint_on_device(const PreviewSystem& ps, const copra_cstr_desc_t& d, bool ineq, Eigen::MatrixXd& A,
        Eigen::VectorXd& b, Eigen::MatrixXd& Y, Eigen::VectorXd& z)
    {
        const int n = ps.fullUDim, nx = ps.xDim;
        const copra_dims_t dims = dims_of(ps);
        int rows = 0;
        {
            HandleGuard g;
            throw_status(copra_batch_create(&g.h, &dims, 0, nullptr, 1, &d));
            int nv, ne, ni;
            throw_status(copra_batch_qp_sizes(g.h, &nv, &ne, &ni));
            rows = ineq ? ni : ne;
            A.resize(rows, n), b.resize(rows), Y.resize(rows, nx), z.resize(rows);
            throw_status(copra_batch_set_system(g.h, ps.A.data(), ps.B.data(), ps.d.data(), ps.x0.data(), 0));
            throw_status(copra_batch_dump_qp(g.h, 0, nullptr, nullptr, ineq ? nullptr : A.data(), ineq ? nullptr : b.data(),
                ineq ? A.data() : nullptr, ineq ? b.data() : nullptr, nullptr, nullptr));
        }
        Eigen::MatrixXd R = Eigen::MatrixXd::Identity(nx, nx), YA(rows, nx + n);
        Eigen::VectorXd r = Eigen::VectorXd::Zero(nx);
        copra_initial_state_desc_t is { R.data(), r.data() };
        HandleGuard g;
        if (copra_batch_create_initial_state(&g.h, &dims, 0, nullptr, 1, &d, &is) != COPRA_OK) return;
        throw_status(copra_batch_set_system(g.h, ps.A.data(), ps.B.data(), ps.d.data(), ps.x0.data(), 0));
        throw_status(copra_batch_dump_qp(g.h, 0, nullptr, nullptr, ineq ? nullptr : YA.data(), ineq ? nullptr : z.data(),
            ineq ? YA.data() : nullptr, ineq ? z.data() : nullptr, nullptr, nullptr));
        for (int i = 0; i < rows; ++i)
            for (int a = 0; a < nx; ++a) Y(i, a) = YA(i, a);
    }
    // true while LMPC::solve() evaluates its costs: the only moment a cost in reference-accumulation mode (below) accumulates
    inline bool& solveInProgress()
    {
        static thread_local bool v = false;
        return v;
    }
} // namespace detail

// ---------------------------------------------------------------------------------------------- costFunctions.h
// include/costFunctions.h:22-97.  A user subclass overrides autoSpan / initializeCost / update and fills Q_, c_ (read by
// LMPC) and E_, f_ (read by InitialStateLMPC) in update(); the nine built-in classes instead describe themselves to the
// device (deviceDescriptor) and are evaluated there.
class CostFunction {
public:
    explicit CostFunction(std::string&& name)
        : name_(std::move(name))
    {
    }
    virtual ~CostFunction() = default;
    virtual void autoSpan() {}
    virtual void initializeCost(const PreviewSystem& ps) // costFunctions.cpp:24-30
    {
        Q_.resize(ps.fullUDim, ps.fullUDim);
        c_.resize(ps.fullUDim);
        E_.resize(ps.xDim, ps.fullUDim);
        f_.resize(ps.fullUDim);
    }
    virtual void update(const PreviewSystem& ps) = 0;
    // engine hook: true + the constructor arguments when the device evaluates this cost itself (built-in classes)
    virtual bool deviceDescriptor(copra_cost_desc_t&) const { return false; }
    // costFunctions.h:54-67
    void weights(const Eigen::VectorXd& w)
    {
        if (w.rows() == weights_.rows()) {
            weights_ = w;
        } else if (w.rows() > 0 && weights_.rows() % w.rows() == 0) {
            const auto reps = weights_.rows() / w.rows();
            for (Eigen::Index i = 0; i < reps; ++i)
                for (Eigen::Index k = 0; k < w.rows(); ++k) weights_(i * w.rows() + k) = w(k);
        } else {
            COPRA_DOMAIN_ERROR("weights: bad dimension");
        }
    }
    void weight(double w) { weights_.setConstant(w); } // costFunctions.h:72-76
    const std::string& name() const noexcept { return name_; }
    const Eigen::MatrixXd& Q() const noexcept { return Q_; } // costFunctions.h:79-90
    const Eigen::VectorXd& c() const noexcept { return c_; }
    const Eigen::MatrixXd& E() const noexcept { return E_; }
    const Eigen::VectorXd& f() const noexcept { return f_; }

protected:
    std::string name_;
    bool fullSizeEntry_ = false;
    Eigen::MatrixXd Q_, E_;
    Eigen::VectorXd c_, f_, weights_;
};

// the four built-in classes: M / N / p / weights are the descriptor the kernels read
class BuiltinCost : public CostFunction {
public:
    BuiltinCost(std::string&& name, int kind)
        : CostFunction(std::move(name))
        , kind_(kind)
    {
    }
    copra_cost_desc_t desc() const
    {
        copra_cost_desc_t d {};
        d.kind = kind_;
        d.rows = (int)p_.rows();
        d.m_cols = (int)M_.cols();
        d.n_cols = (int)N_.cols();
        d.M = M_.size() ? M_.data() : nullptr;
        d.N = N_.size() ? N_.data() : nullptr;
        d.p = p_.data();
        d.weights = weights_.data();
        return d;
    }
    // Reference quirk Q2 (src/costFunctions.cpp:73-80 and :205-213): the per-step entries of TrajectoryCost and MixedCost ADD each
    // solve's Q, E, f to members that are only zeroed in initializeCost (:52-55, :184-187), and MixedCost also adds to c -- the n-th
    // solve() on one controller sees n x the trajectory Hessian.  The engine's default is a fresh controller's first solve (what a
    // batched solve means).  A caller who solves repeatedly on one copra::LMPC and wants the reference's numbers turns
    // this on (LMPC::referenceAccumulation): the cost is then evaluated per solve by the device's condense code, accumulated HERE in the
    // reference's own order, and handed to the fused solve as the dense members Q_, c_, E_, f_ (the plug-in route of user subclasses).
    void referenceAccumulation(bool on) { accumulate_ = on; }
    bool accumulates() const
    {
        return accumulate_ && !fullSizeEntry_ && (kind_ == COPRA_COST_TRAJECTORY || kind_ == COPRA_COST_MIXED);
    }
    bool deviceDescriptor(copra_cost_desc_t& d) const override
    {
        if (accumulates()) return false;
        d = desc();
        return true;
    }
    void initializeCost(const PreviewSystem& ps) override // costFunctions.cpp:44-61, 88-98, 122-137, 173-193
    {
        if (M_.size() && M_.rows() != p_.rows()) COPRA_DOMAIN_ERROR("M and p should have the same number of rows (try autoSpan)");
        if (N_.size() && N_.rows() != p_.rows()) COPRA_DOMAIN_ERROR("N and p should have the same number of rows (try autoSpan)");
        detail::check_pieces(ps, { desc() }, {});
        fullSizeEntry_ = (M_.size() && M_.cols() == ps.fullXDim && ps.fullXDim != ps.xDim)
            || (N_.size() && N_.cols() == ps.fullUDim && ps.fullUDim != ps.uDim);
        CostFunction::initializeCost(ps);
        Q_.setZero(), c_.setZero(), E_.setZero(), f_.setZero(); // costFunctions.cpp:52-55, 184-187
    }
    // Q_, c_, E_, f_ as the reference's update() leaves them -- evaluated by the device, on request only (LMPC::solve
    // never needs them for a built-in class)
    void update(const PreviewSystem& ps) override
    {
        if (!accumulates()) {
            detail::evaluate_cost_on_device(ps, desc(), Q_, c_, E_, f_);
            return;
        }
        if (!detail::solveInProgress()) return; // (an accessor between solves: the members as the last solve left them)
        Eigen::MatrixXd Q1, E1;
        Eigen::VectorXd c1, f1;
        detail::evaluate_cost_on_device(ps, desc(), Q1, c1, E1, f1); // this solve's sums over the steps
        const Eigen::Index n = ps.fullUDim, nx = ps.xDim;
        for (Eigen::Index e = 0; e < n * n; ++e) Q_.data()[e] += Q1.data()[e]; // Q_ += ...   (:75 / :208)
        for (Eigen::Index e = 0; e < nx * n; ++e) E_.data()[e] += E1.data()[e]; // E_ += ...   (:77 / :210)
        for (Eigen::Index e = 0; e < n; ++e) f_.data()[e] += f1.data()[e]; // f_ += ...   (:78 / :211)
        for (Eigen::Index j = 0; j < n; ++j) { // E_' x0 + f_ with the ACCUMULATED E_, f_
            double v = f_(j);
            for (Eigen::Index a = 0; a < nx; ++a) v += E_(a, j) * ps.x0(a);
            if (kind_ == COPRA_COST_MIXED)
                c_(j) += v; // c_ += ...  (:213)
            else
                c_(j) = v; // c_ = ...   (:80)
        }
    }

protected:
    int kind_;
    bool accumulate_ = false;
    Eigen::MatrixXd M_, N_;
    Eigen::VectorXd p_;
};

class TrajectoryCost : public BuiltinCost { // costFunctions.h:103-126
public:
    TrajectoryCost(const Eigen::MatrixXd& M, const Eigen::VectorXd& p)
        : BuiltinCost("TrajectoryCost", COPRA_COST_TRAJECTORY)
    {
        M_ = M;
        p_ = p;
        weights_ = Eigen::VectorXd::Ones(p_.rows());
    }
    void autoSpan() override // costFunctions.cpp:36-42
    {
        const auto m = std::max(M_.rows(), std::max(weights_.rows(), p_.rows()));
        AutoSpan::spanMatrix(M_, m);
        AutoSpan::spanVector(p_, m);
        AutoSpan::spanVector(weights_, m);
    }
};
class TargetCost : public BuiltinCost { // costFunctions.h:134-155
public:
    TargetCost(const Eigen::MatrixXd& M, const Eigen::VectorXd& p)
        : BuiltinCost("TargetCost", COPRA_COST_TARGET)
    {
        M_ = M;
        p_ = p;
        weights_ = Eigen::VectorXd::Ones(p_.rows());
    }
};
class ControlCost : public BuiltinCost { // costFunctions.h:163-186
public:
    ControlCost(const Eigen::MatrixXd& N, const Eigen::VectorXd& p)
        : BuiltinCost("ControlCost", COPRA_COST_CONTROL)
    {
        N_ = N;
        p_ = p;
        weights_ = Eigen::VectorXd::Ones(p_.rows());
    }
    void autoSpan() override // costFunctions.cpp:114-120
    {
        const auto m = std::max(N_.rows(), std::max(weights_.rows(), p_.rows()));
        AutoSpan::spanMatrix(N_, m);
        AutoSpan::spanVector(p_, m);
        AutoSpan::spanVector(weights_, m);
    }
};
class MixedCost : public BuiltinCost { // costFunctions.h:194-219
public:
    MixedCost(const Eigen::MatrixXd& M, const Eigen::MatrixXd& N, const Eigen::VectorXd& p)
        : BuiltinCost("MixedCost", COPRA_COST_MIXED)
    {
        M_ = M;
        N_ = N;
        p_ = p;
        weights_ = Eigen::VectorXd::Ones(p_.rows());
    }
    void autoSpan() override // costFunctions.cpp:164-171
    {
        const auto m = std::max(M_.rows(), std::max(N_.rows(), std::max(weights_.rows(), p_.rows())));
        AutoSpan::spanMatrix(M_, m, 1);
        AutoSpan::spanMatrix(N_, m);
        AutoSpan::spanVector(p_, m);
        AutoSpan::spanVector(weights_, m);
    }
};

// ---------------------------------------------------------------------------------------------- constraints.h
enum class ConstraintFlag { Constraint, EqualityConstraint, InequalityConstraint, BoundConstraint }; // constraints.h:28-33

class Constraint { // constraints.h:42-79
public:
    explicit Constraint(std::string&& name)
        : name_(std::move(name))
    {
    }
    virtual ~Constraint() = default;
    virtual void autoSpan() = 0;
    virtual void initializeConstraint(const PreviewSystem& ps) = 0;
    virtual void update(const PreviewSystem& ps) = 0;
    virtual ConstraintFlag constraintType() const noexcept = 0;
    // engine hook: true + the constructor arguments when the device evaluates this constraint itself (built-in classes)
    virtual bool deviceDescriptor(copra_cstr_desc_t&) const { return false; }
    const std::string& name() const noexcept { return name_; }
    int nrConstr() noexcept { return nrConstr_; }

protected:
    std::string name_;
    int nrConstr_ = 0;
    bool fullSizeEntry_ = false;
    bool hasBeenInitialized_ = false;
};

class EqIneqConstraint : public Constraint { // constraints.h:84-107
public:
    EqIneqConstraint(const std::string& name, bool isInequalityConstraint)
        : Constraint(name + (isInequalityConstraint ? " inequality constraint" : " equality constraint")) // constraints.cpp:22-32
        , isIneq_(isInequalityConstraint)
    {
    }
    const Eigen::MatrixXd& A() const noexcept { return A_; }
    const Eigen::VectorXd& b() const noexcept { return b_; }
    const Eigen::MatrixXd& Y() const noexcept { return Y_; }
    const Eigen::VectorXd& z() const noexcept { return z_; }
    ConstraintFlag constraintType() const noexcept override
    {
        return isIneq_ ? ConstraintFlag::InequalityConstraint : ConstraintFlag::EqualityConstraint;
    }

protected:
    Eigen::MatrixXd A_, Y_;
    Eigen::VectorXd b_, z_;
    bool isIneq_;
};

// the three built-in equality / inequality classes: E / G / f are the descriptor the kernels read
class BuiltinEqIneq : public EqIneqConstraint {
public:
    BuiltinEqIneq(const std::string& name, int kind, bool ineq)
        : EqIneqConstraint(name, ineq)
        , kind_(kind)
    {
    }
    copra_cstr_desc_t desc() const
    {
        copra_cstr_desc_t d {};
        d.kind = kind_;
        d.is_inequality = isIneq_ ? 1 : 0;
        d.rows = (int)f_.rows();
        d.e_cols = (int)E_.cols();
        d.g_cols = (int)G_.cols();
        d.E = E_.size() ? E_.data() : nullptr;
        d.G = G_.size() ? G_.data() : nullptr;
        d.f = f_.data();
        return d;
    }
    bool deviceDescriptor(copra_cstr_desc_t& d) const override
    {
        d = desc();
        return true;
    }
    void initializeConstraint(const PreviewSystem& ps) override // constraints.cpp:45-64, 106-135, 171-195
    {
        if (kind_ == COPRA_CSTR_CONTROL && hasBeenInitialized_) // move semantics of the reference (constraints.cpp:108-110)
            COPRA_RUNTIME_ERROR("You have initialized a " + name_ + " twice. As move semantics are used, you can't do so.");
        if (E_.size() && E_.rows() != f_.rows()) COPRA_DOMAIN_ERROR("E and f should have the same number of rows (try autoSpan)");
        if (G_.size() && G_.rows() != f_.rows()) COPRA_DOMAIN_ERROR("G and f should have the same number of rows (try autoSpan)");
        copra_hip::HostPlan hp;
        detail::check_pieces(ps, {}, { desc() }, &hp);
        nrConstr_ = hp.plan.meq + hp.plan.mineq;
        fullSizeEntry_ = (E_.size() && E_.cols() == ps.fullXDim && ps.fullXDim != ps.xDim)
            || (G_.size() && G_.cols() == ps.fullUDim && ps.fullUDim != ps.uDim);
        hasBeenInitialized_ = true;
    }
    // A_, b_, Y_, z_ as the reference's update() leaves them -- evaluated by the device, on request only
    void update(const PreviewSystem& ps) override { detail::evaluate_constraint_on_device(ps, desc(), isIneq_, A_, b_, Y_, z_); }

protected:
    int kind_;
    Eigen::MatrixXd E_, G_;
    Eigen::VectorXd f_;
};

class TrajectoryConstraint : public BuiltinEqIneq { // constraints.h:114-145
public:
    TrajectoryConstraint(const Eigen::MatrixXd& E, const Eigen::VectorXd& f, bool isInequalityConstraint = true)
        : BuiltinEqIneq("Trajectory", COPRA_CSTR_TRAJECTORY, isInequalityConstraint)
    {
        E_ = E;
        f_ = f;
    }
    void autoSpan() override // constraints.cpp:38-43
    {
        const auto m = std::max(E_.rows(), f_.rows());
        AutoSpan::spanMatrix(E_, m);
        AutoSpan::spanVector(f_, m);
    }
};
class ControlConstraint : public BuiltinEqIneq { // constraints.h:153-185
public:
    ControlConstraint(const Eigen::MatrixXd& G, const Eigen::VectorXd& f, bool isInequalityConstraint = true)
        : BuiltinEqIneq("Control", COPRA_CSTR_CONTROL, isInequalityConstraint)
    {
        G_ = G;
        f_ = f;
    }
    void autoSpan() override // constraints.cpp:99-104
    {
        const auto m = std::max(G_.rows(), f_.rows());
        AutoSpan::spanMatrix(G_, m);
        AutoSpan::spanVector(f_, m);
    }
};
class MixedConstraint : public BuiltinEqIneq { // constraints.h:193-226
public:
    MixedConstraint(const Eigen::MatrixXd& E, const Eigen::MatrixXd& G, const Eigen::VectorXd& f,
        bool isInequalityConstraint = true)
        : BuiltinEqIneq("Control", COPRA_CSTR_MIXED, isInequalityConstraint) // name quirk Q4 kept (constraints.h:204)
    {
        E_ = E;
        G_ = G;
        f_ = f;
    }
    void autoSpan() override // constraints.cpp:161-169
    {
        const auto m = std::max(f_.rows(), std::max(E_.rows(), G_.rows()));
        AutoSpan::spanMatrix(E_, m, 1);
        AutoSpan::spanMatrix(G_, m);
        AutoSpan::spanVector(f_, m);
    }
};
class TrajectoryBoundConstraint : public EqIneqConstraint { // constraints.h:234-276
public:
    TrajectoryBoundConstraint(const Eigen::VectorXd& lower, const Eigen::VectorXd& upper)
        : EqIneqConstraint("Trajectory bound", true)
        , lower_(lower)
        , upper_(upper)
    {
    }
    void autoSpan() override // constraints.cpp:232-261
    {
        const auto m = std::max(lower_.rows(), upper_.rows());
        AutoSpan::spanVector(lower_, m);
        AutoSpan::spanVector(upper_, m);
    }
    copra_cstr_desc_t desc() const
    {
        copra_cstr_desc_t d {};
        d.kind = COPRA_CSTR_TRAJECTORY_BOUND;
        d.is_inequality = 1;
        d.rows = (int)lower_.rows();
        d.lower = lower_.data();
        d.upper = upper_.data();
        return d;
    }
    bool deviceDescriptor(copra_cstr_desc_t& d) const override
    {
        d = desc();
        return true;
    }
    void initializeConstraint(const PreviewSystem& ps) override // constraints.cpp:263-282
    {
        if (lower_.rows() != upper_.rows()) COPRA_DOMAIN_ERROR("lower and upper should have the same number of rows (try autoSpan)");
        copra_hip::HostPlan hp;
        detail::check_pieces(ps, {}, { desc() }, &hp);
        nrConstr_ = hp.plan.mineq;
        hasBeenInitialized_ = true;
    }
    void update(const PreviewSystem& ps) override { detail::evaluate_constraint_on_device(ps, desc(), true, A_, b_, Y_, z_); }

protected:
    Eigen::VectorXd lower_, upper_;
};
class ControlBoundConstraint : public Constraint { // constraints.h:284-308
public:
    ControlBoundConstraint(const Eigen::VectorXd& lower, const Eigen::VectorXd& upper)
        : Constraint("Control bound constraint")
        , lower_(lower)
        , upper_(upper)
    {
    }
    void autoSpan() override // constraints.cpp:326-331
    {
        const auto m = std::max(lower_.rows(), upper_.rows());
        AutoSpan::spanVector(lower_, m);
        AutoSpan::spanVector(upper_, m);
    }
    copra_cstr_desc_t desc() const
    {
        copra_cstr_desc_t d {};
        d.kind = COPRA_CSTR_CONTROL_BOUND;
        d.is_inequality = 1;
        d.rows = (int)lower_.rows();
        d.lower = lower_.data();
        d.upper = upper_.data();
        return d;
    }
    bool deviceDescriptor(copra_cstr_desc_t& d) const override
    {
        d = desc();
        return true;
    }
    void initializeConstraint(const PreviewSystem& ps) override // constraints.cpp:333-357
    {
        if (hasBeenInitialized_) // constraints.cpp:335-337
            COPRA_RUNTIME_ERROR("You have initialized a " + name_ + " twice. As move semantics are used, you can't do so.");
        if (lower_.rows() != upper_.rows()) COPRA_DOMAIN_ERROR("lower and upper should have the same number of rows (try autoSpan)");
        detail::check_pieces(ps, {}, { desc() });
        nrConstr_ = ps.fullUDim;
        hasBeenInitialized_ = true;
        lb_.resize(ps.fullUDim), ub_.resize(ps.fullUDim);
        update(ps);
    }
    void update(const PreviewSystem& ps) override // constraints.cpp:359-367: tile the per-step bounds over the horizon
    {
        lb_.resize(ps.fullUDim), ub_.resize(ps.fullUDim);
        for (int i = 0; i < ps.fullUDim; ++i) {
            const Eigen::Index src = (lower_.rows() == ps.uDim) ? (i % ps.uDim) : i;
            lb_(i) = lower_(src);
            ub_(i) = upper_(src);
        }
    }
    ConstraintFlag constraintType() const noexcept override { return ConstraintFlag::BoundConstraint; }
    const Eigen::VectorXd& lower() noexcept { return lb_; } // constraints.h:298-300
    const Eigen::VectorXd& upper() noexcept { return ub_; }

protected:
    Eigen::VectorXd lower_, upper_, lb_, ub_;
};

// ---------------------------------------------------------------------------------------------- SolverInterface.h
class SolverInterface { // include/SolverInterface.h:19-81; defaults of the optional members: src/SolverInterface.cpp:15-56
public:
    SolverInterface() = default;
    virtual ~SolverInterface() = default;
    virtual int SI_fail() const = 0;
    virtual void SI_inform() const = 0;
    virtual int SI_iter() const
    {
        std::printf("No iter() function for this qp\n");
        return 0;
    }
    virtual int SI_maxIter() const
    {
        std::printf("No maxIter() function for this qp\n");
        return 0;
    }
    virtual void SI_maxIter(int) { std::printf("No maxIter(int) function for this qp\n"); }
    virtual void SI_printLevel(int) { std::printf("No printLevel(int) function for this qp\n"); }
    virtual double SI_feasibilityTolerance() const
    {
        std::printf("No feasibilityTolerance() function for this qp\n");
        return 0.;
    }
    virtual void SI_feasibilityTolerance(double) { std::printf("No feasibilityTolerance(double) function for this qp\n"); }
    virtual bool SI_warmStart() const
    {
        std::printf("No warmStart() function for this qp\n");
        return false;
    }
    virtual void SI_warmStart(bool) { std::printf("No warmStart(bool) function for this qp\n"); }
    virtual const Eigen::VectorXd& SI_result() const = 0;
    virtual void SI_problem(int nrVar, int nrEq, int nrInEq) = 0;
    virtual bool SI_solve(const Eigen::MatrixXd& Q, const Eigen::VectorXd& c, const Eigen::MatrixXd& Aeq,
        const Eigen::VectorXd& beq, const Eigen::MatrixXd& Aineq, const Eigen::VectorXd& bineq,
        const Eigen::VectorXd& XL, const Eigen::VectorXd& XU)
        = 0;
};

// Plug-in point 1 on the GPU: copra::QuadProgDenseSolver (include/QuadProgSolver.h:16-42, src/QuadProgSolver.cpp:14-72) under the
// reference's own name -- code that instantiates it (tests/TestSolvers.cpp:27) compiles unchanged; the arithmetic is the batched
// Goldfarb-Idnani kernel behind copra_qp_solve_dense_batch.  HipQuadProgSolver is the name earlier rounds of this mirror used.
class QuadProgDenseSolver : public SolverInterface {
public:
    QuadProgDenseSolver() = default;
    int SI_fail() const override { return fail_; }
    int SI_iter() const override { return iter_[0]; }
    void SI_inform() const override
    { // src/QuadProgSolver.cpp:24-38
        switch (fail_) {
        case 0: std::printf("No problems\n"); break;
        case 1: std::printf("The minimization problem has no solution\n"); break;
        case 2: std::printf("Problems with the decomposition of Q (Is it symmetric?)\n"); break;
        default: break;
        }
    }
    const Eigen::VectorXd& SI_result() const override { return x_; }
    void SI_problem(int nrVar, int nrEq, int nrInEq) override
    {
        n_ = nrVar;
        neq_ = nrEq;
        nin_ = nrInEq;
        x_.resize(nrVar);
    }
    bool SI_solve(const Eigen::MatrixXd& Q, const Eigen::VectorXd& c, const Eigen::MatrixXd& Aeq,
        const Eigen::VectorXd& beq, const Eigen::MatrixXd& Aineq, const Eigen::VectorXd& bineq,
        const Eigen::VectorXd& XL, const Eigen::VectorXd& XU) override
    {
        throw_status(copra_qp_solve_dense_batch(1, n_, neq_, nin_, Q.data(), c.data(), Aeq.data(), beq.data(),
            Aineq.data(), bineq.data(), XL.data(), XU.data(), x_.data(), &fail_, iter_, 0, nullptr));
        return fail_ == 0;
    }

private:
    int n_ = 0, neq_ = 0, nin_ = 0, fail_ = 0, iter_[2] = { 0, 0 };
    Eigen::VectorXd x_;
};

// include/solverUtils.h:34-50.  DEFAULT means what it means in the reference (src/solverUtils.cpp:9-34: DEFAULT -> QuadProgDense):
// the Goldfarb-Idnani kernels at every size -- the reference's QuadProgDense arithmetic, SI_iter() / iterations = active-set
// iterations.  HipQuadProg: the same through plug-in point 1.  HipRiccati (no counterpart in the reference): "the engine decides" --
// the Goldfarb-Idnani kernels up to 64 variables and the stage-wise Riccati interior-point kernel for long stage-wise horizons
// (COPRA_SOLVER_DEFAULT of the C ABI; its iteration counter counts Newton steps there, and at an ill-conditioned Hessian its result
// is the optimum to 1e-10 where the QuadProgDense arithmetic carries the conditioning: INTEGRATION.md 1).
using HipQuadProgSolver = QuadProgDenseSolver;
enum class SolverFlag { DEFAULT, QuadProgDense, HipQuadProg, HipRiccati };
inline int engineSolver(SolverFlag f) { return f == SolverFlag::HipRiccati ? COPRA_SOLVER_DEFAULT : COPRA_SOLVER_QUADPROG_DENSE; }
inline std::unique_ptr<SolverInterface> solverFactory(SolverFlag) // src/solverUtils.cpp:9-34
{
    return std::unique_ptr<SolverInterface>(new QuadProgDenseSolver());
}
inline SolverInterface* pythonSolverFactory(SolverFlag) // src/solverUtils.cpp:36-59 (a raw pointer for bindings; the caller owns it)
{
    return new QuadProgDenseSolver();
}

// ---------------------------------------------------------------------------------------------- LMPC.h
class LMPC { // include/LMPC.h:36-191, src/LMPC.cpp
public:
    LMPC(SolverFlag flag = SolverFlag::DEFAULT) { selectQPSolver(flag); }
    LMPC(const std::shared_ptr<PreviewSystem>& ps, SolverFlag flag = SolverFlag::DEFAULT)
    {
        selectQPSolver(flag);
        initializeController(ps);
    }
    virtual ~LMPC() { release(); }
    LMPC(LMPC&& o) noexcept { *this = std::move(o); }
    LMPC& operator=(LMPC&& o) noexcept
    {
        release();
        ps_ = std::move(o.ps_);
        spCost_ = std::move(o.spCost_);
        spConstr_ = std::move(o.spConstr_);
        sol_ = std::move(o.sol_);
        flag_ = o.flag_;
        h_ = o.h_;
        o.h_ = nullptr;
        dirty_ = o.dirty_;
        costsDirty_ = o.costsDirty_;
        builtCosts_ = std::move(o.builtCosts_);
        builtCostsKnown_ = o.builtCostsKnown_;
        control_ = std::move(o.control_);
        trajectory_ = std::move(o.trajectory_);
        return *this;
    }
    // LMPC.cpp:62-65.  The fused device solve is the solver behind every flag; the flag picks its algorithm
    // (copra_batch_select_solver) and drops a solver installed by useSolver.
    void selectQPSolver(SolverFlag flag)
    {
        flag_ = flag;
        sol_.reset();
        if (h_) throw_status(copra_batch_select_solver(h_, engineSolver(flag_)));
    }
    // LMPC.cpp:67-70: plug-in point 1.  With a user solver the QP is condensed on the device, copied out and handed to
    // SI_problem / SI_solve / SI_result in the reference's order (LMPC.cpp:88-97, 284).
    void useSolver(std::unique_ptr<SolverInterface>&& solver) { sol_ = std::move(solver); }
    void initializeController(const std::shared_ptr<PreviewSystem>& ps)
    {
        ps_ = ps;
        dirty_ = true;
    }
    void addCost(const std::shared_ptr<CostFunction>& cost) // LMPC.cpp:118-122
    {
        cost->initializeCost(*ps_);
        if (auto b = std::dynamic_pointer_cast<BuiltinCost>(cost)) b->referenceAccumulation(refAccumulation_);
        spCost_.push_back(cost);
        costsDirty_ = true; // (the list of costs has changed: rebuild() looks at what has changed in it)
    }
    // Opt-in compatibility with reference quirk Q2 (no counterpart in the reference's API -- it is how its cost classes behave,
    // src/costFunctions.cpp:73-80, 205-213): with `on`, the per-step TrajectoryCost / MixedCost objects of this controller accumulate
    // Q, E, f (and MixedCost c) from solve to solve as the reference's do, so that the k-th solve() returns the reference's k-th
    // answer.  Off (default): every solve is a fresh controller's first one.  Costs added later follow the switch.
    void referenceAccumulation(bool on)
    {
        refAccumulation_ = on;
        for (auto& c : spCost_)
            if (auto b = std::dynamic_pointer_cast<BuiltinCost>(c)) b->referenceAccumulation(on);
        dirty_ = true;
    }
    bool referenceAccumulation() const noexcept { return refAccumulation_; }
    void addConstraint(const std::shared_ptr<Constraint>& c) // LMPC.cpp:124-128, 173-197
    {
        c->initializeConstraint(*ps_);
        if (c->constraintType() == ConstraintFlag::Constraint) return; // (LMPC.cpp:192-193: unknown type, not stored)
        spConstr_.push_back(c);
        dirty_ = true;
    }
    void clearCosts() noexcept
    {
        spCost_.clear();
        costsDirty_ = true;
    }
    void clearConstraints() noexcept
    {
        spConstr_.clear();
        dirty_ = true;
    }
    void removeCost(const std::shared_ptr<CostFunction>& c)
    {
        auto it = std::find(spCost_.begin(), spCost_.end(), c);
        if (it != spCost_.end()) spCost_.erase(it), costsDirty_ = true;
    }
    void removeConstraint(const std::shared_ptr<Constraint>& c)
    {
        auto it = std::find(spConstr_.begin(), spConstr_.end(), c);
        if (it != spConstr_.end()) spConstr_.erase(it), dirty_ = true;
    }

    bool solve() // LMPC.cpp:79-101
    {
        using clock = std::chrono::high_resolution_clock;
        const auto t0 = clock::now();
        detail::solveInProgress() = true; // (costs in reference-accumulation mode accumulate in THIS evaluation only)
        try {
            prepare();
        } catch (...) {
            detail::solveInProgress() = false;
            throw;
        }
        detail::solveInProgress() = false;
        bool ok;
        if (sol_) {
            // ---- user SolverInterface: the device condenses, the user's solver solves (LMPC.cpp:88-97) ----
            qp_ = fetchQP();
            qpValid_ = true;
            const int nvar = (int)qp_.c.rows();
            sol_->SI_problem(nvar, (int)qp_.beq.rows(), (int)qp_.bineq.rows());
            const auto t1 = clock::now();
            ok = sol_->SI_solve(qp_.Q, qp_.c, qp_.Aeq, qp_.beq, qp_.Aineq, qp_.bineq, qp_.lb, qp_.ub);
            solveTime_ = std::chrono::duration<double>(clock::now() - t1).count();
            fail_ = sol_->SI_fail();
            checkDeleteCostsAndConstraints();
            if (ok) updateResultsFrom(sol_->SI_result());
        } else {
            qpValid_ = false;
            throw_status(copra_batch_solve(h_, nullptr));
            Eigen::VectorXd u(ps_->fullUDim), x(ps_->fullXDim);
            int it[2];
            throw_status(copra_batch_get_results(h_, u.data(), x.data(), &fail_, it));
            ok = fail_ == 0;
            afterSolve(ok);
            double dev = 0.0;
            copra_batch_last_solve_seconds(h_, &dev);
            solveTime_ = dev; // device time of the launch
            iter_ = it[0];
            checkDeleteCostsAndConstraints();
            if (ok) { // LMPC.cpp:95-97: outputs only updated on success
                control_ = u;
                trajectory_ = x;
            }
        }
        ps_->isUpdated = true;
        solveAndBuildTime_ = std::chrono::duration<double>(clock::now() - t0).count();
        return ok;
    }
    void inform() const noexcept
    {
        if (sol_) {
            sol_->SI_inform();
            return;
        }
        std::printf("%s\n", fail_ == 0 ? "No problems" : fail_ == 1 ? "The minimization problem has no solution"
                                                                    : "Problems with the decomposition of Q (Is it symmetric?)");
    }
    double solveTime() const noexcept { return solveTime_; }
    double solveAndBuildTime() const noexcept { return solveAndBuildTime_; }
    const Eigen::VectorXd& control() const noexcept { return control_; }
    const Eigen::VectorXd& trajectory() const noexcept { return trajectory_; }
    int fail() const noexcept { return fail_; }
    int iter() const noexcept { return iter_; }
    int handleBuilds() const noexcept { return handleBuilds_; } // (not in the reference: how often the device-side controller was built)
    // (not in the reference) the algorithm the next solve() runs on the device: COPRA_SOLVER_QUADPROG_DENSE -- Goldfarb-Idnani, what every
    // flag of the reference means -- or COPRA_SOLVER_RICCATI_IPM, which only SolverFlag::HipRiccati can select
    int solverKind()
    {
        prepare();
        return copra_batch_solver_info(h_);
    }
    int nrEqConstr()
    {
        prepare();
        int n, e, i;
        copra_batch_qp_sizes(h_, &n, &e, &i);
        return e;
    }
    int nrIneqConstr()
    {
        prepare();
        int n, e, i;
        copra_batch_qp_sizes(h_, &n, &e, &i);
        return i;
    }
    // LMPC.h:112-127 -- the dense QP as condensed ON THE DEVICE (parity hook)
    struct DenseQP {
        Eigen::MatrixXd Q, Aeq, Aineq;
        Eigen::VectorXd c, beq, bineq, lb, ub;
    };
    DenseQP denseQP()
    {
        prepare();
        return fetchQP();
    }

    // the accessors of LMPC.h:112-127 (matrices of the last problem handed to the solver), fetched from the device on
    // first use after a solve
    const Eigen::MatrixXd& Q() { return qp().Q; }
    const Eigen::VectorXd& c() { return qp().c; }
    const Eigen::MatrixXd& Aeq() { return qp().Aeq; }
    const Eigen::VectorXd& beq() { return qp().beq; }
    const Eigen::MatrixXd& Aineq() { return qp().Aineq; }
    const Eigen::VectorXd& bineq() { return qp().bineq; }
    const Eigen::VectorXd& lb() { return qp().lb; }
    const Eigen::VectorXd& ub() { return qp().ub; }

protected:
    const DenseQP& qp()
    {
        if (!qpValid_) {
            qp_ = denseQP();
            qpValid_ = true;
        }
        return qp_;
    }
    // hooks of the InitialStateLMPC variant
    virtual copra_status_t createHandle(const copra_dims_t& dims, const std::vector<copra_cost_desc_t>& cd,
        const std::vector<copra_cstr_desc_t>& kd)
    {
        return copra_batch_create(&h_, &dims, (int)cd.size(), cd.data(), (int)kd.size(), kd.data());
    }
    virtual void beforeSolve() {}
    virtual void afterSolve(bool) {}
    // LMPC.cpp:282-286 with a user solver's result: control = U, trajectory = Phi x0 + Psi U + xi on the host
    virtual void updateResultsFrom(const Eigen::VectorXd& result)
    {
        control_ = result;
        hostTrajectory(ps_->x0, control_);
    }
    void hostTrajectory(const Eigen::VectorXd& x0, const Eigen::VectorXd& U)
    {
        needPreview();
        trajectory_.resize(ps_->fullXDim);
        for (int i = 0; i < ps_->fullXDim; ++i) {
            double acc = ps_->xi(i);
            for (int a = 0; a < ps_->xDim; ++a) acc += ps_->Phi(i, a) * x0(a);
            for (int j = 0; j < ps_->fullUDim; ++j) acc += ps_->Psi(i, j) * U(j);
            trajectory_(i) = acc;
        }
    }
    void needPreview()
    {
        if (!ps_->previewOnHost) ps_->updateSystem(); // LMPC.cpp:233
        if (!ps_->previewOnHost) throw std::runtime_error(std::string("PreviewSystem::updateSystem: ") + copra_last_error());
    }
    // everything of LMPC::updateSystem + makeQPForm that is not the device's job: host-evaluated user pieces are updated,
    // the controller handle is (re)built from the descriptors, the system and x0 go to the device
    void prepare()
    {
        bool custom = false;
        copra_cost_desc_t cdesc;
        copra_cstr_desc_t kdesc;
        for (auto& c : spCost_) custom = custom || !c->deviceDescriptor(cdesc);
        for (auto& c : spConstr_) custom = custom || !c->deviceDescriptor(kdesc);
        if (custom) { // LMPC.cpp:233-247: constraints first, then costs
            needPreview();
            for (auto& c : spConstr_)
                if (!c->deviceDescriptor(kdesc)) c->update(*ps_);
            for (auto& c : spCost_)
                if (!c->deviceDescriptor(cdesc)) c->update(*ps_);
            dirty_ = true; // their matrices are part of the plan
        }
        const bool fresh = rebuild(); // (a new handle: the system has to be sent to it)
        // The receding-horizon tick of the reference is ps->xInit(x) between solves (PreviewSystem.h:52): only x0 has changed then,
        // and only x0 crosses PCIe again (one 48-byte copy instead of four blocking ones).  The fields of PreviewSystem are public, so
        // what was sent last is compared rather than trusted.
        auto same = [](const double* a, Eigen::Index na, const double* b, Eigen::Index nb) { return na == nb && std::equal(a, a + na, b); };
        const bool same_system = !fresh && same(sentA_.data(), sentA_.rows() * sentA_.cols(), ps_->A.data(), ps_->A.rows() * ps_->A.cols())
            && same(sentB_.data(), sentB_.rows() * sentB_.cols(), ps_->B.data(), ps_->B.rows() * ps_->B.cols())
            && same(sentd_.data(), sentd_.rows(), ps_->d.data(), ps_->d.rows());
        if (same_system) {
            throw_status(copra_batch_set_x0(h_, ps_->x0.data(), 0));
        } else {
            throw_status(copra_batch_set_system(h_, ps_->A.data(), ps_->B.data(), ps_->d.data(), ps_->x0.data(), 0));
            sentA_ = ps_->A, sentB_ = ps_->B, sentd_ = ps_->d;
        }
        beforeSolve();
    }
    DenseQP fetchQP()
    {
        int n, e, i;
        copra_batch_qp_sizes(h_, &n, &e, &i);
        DenseQP q;
        q.Q.resize(n, n);
        q.c.resize(n);
        q.Aeq.resize(e, n);
        q.beq.resize(e);
        q.Aineq.resize(i, n);
        q.bineq.resize(i);
        q.lb.resize(n);
        q.ub.resize(n);
        throw_status(copra_batch_dump_qp(h_, 0, q.Q.data(), q.c.data(), q.Aeq.data(), q.beq.data(), q.Aineq.data(),
            q.bineq.data(), q.lb.data(), q.ub.data()));
        return q;
    }
    // What the handle was built from, cost by cost (built-in classes): the reference evaluates every cost anew in every solve
    // (LMPC.cpp:233-247), so a caller may change weights between solves, or -- the only way its API has to move a reference -- replace a
    // cost by a new one that differs in p alone (costFunctions.h: M, N, p are constructor arguments).  The first needs a new plan; the
    // second is copra_batch_set_cost_reference on the handle that exists: a tracking controller's tick costs a copy of p, not a new handle.
    struct CostSnapshot {
        int kind = 0, rows = 0, m_cols = 0, n_cols = 0;
        std::vector<double> M, N, p, w;
    };
    static CostSnapshot snapshot(const copra_cost_desc_t& d)
    {
        CostSnapshot c;
        c.kind = d.kind, c.rows = d.rows, c.m_cols = d.m_cols, c.n_cols = d.n_cols;
        if (d.M) c.M.assign(d.M, d.M + (size_t)d.rows * d.m_cols);
        if (d.N) c.N.assign(d.N, d.N + (size_t)d.rows * d.n_cols);
        if (d.p) c.p.assign(d.p, d.p + d.rows);
        if (d.weights) c.w.assign(d.weights, d.weights + d.rows);
        return c;
    }
public:
    // (measurement aid of this mirror, no reference counterpart: rebuild the device-side controller whenever a cost object was
    //  replaced, as the mirror did before it learnt to send a changed reference to the handle that exists)
    static bool& newHandlePerCostChange()
    {
        static bool on = false;
        return on;
    }

protected:
    // 0: the costs are what the handle was built from; 1: so they are up to the references p (pushed to the handle); 2: anything else
    int costsAgainstHandle()
    {
        if (!builtCostsKnown_ || spCost_.size() != builtCosts_.size()) return 2;
        if (costsDirty_ && newHandlePerCostChange()) return 2; // (measurements: what a swapped cost cost before)
        auto same = [](const double* a, const std::vector<double>& b, size_t n) { return (a ? n : 0) == b.size() && (!a || std::equal(a, a + n, b.begin())); };
        std::vector<copra_cost_desc_t> now(spCost_.size());
        for (size_t t = 0; t < spCost_.size(); ++t) {
            const CostSnapshot& b = builtCosts_[t];
            copra_cost_desc_t& d = now[t];
            if (!spCost_[t]->deviceDescriptor(d)) return 2;
            if (d.kind != b.kind || d.rows != b.rows || d.m_cols != b.m_cols || d.n_cols != b.n_cols) return 2;
            if (!same(d.weights, b.w, (size_t)d.rows)) return 2;
            // (M and N are constructor arguments: the same object still has the ones the handle was built from)
            if (costsDirty_ && (!same(d.M, b.M, (size_t)d.rows * d.m_cols) || !same(d.N, b.N, (size_t)d.rows * d.n_cols))) return 2;
        }
        int rc = 0;
        for (size_t t = 0; t < spCost_.size(); ++t) {
            if (same(now[t].p, builtCosts_[t].p, (size_t)now[t].rows)) continue;
            // (a controller past the one-wave kernels would leave its fast kernels in per-instance-reference mode -- the LDS-resident
            //  interior-point kernel above all, include/copra_hip.h --: there a new handle is worth more than the handle build)
            if (copra_batch_lanes_per_instance(h_) > 64 || copra_batch_solver_info(h_) == COPRA_SOLVER_RICCATI_IPM) return 2;
            if (copra_batch_set_cost_reference(h_, (int)t, now[t].p, 0) != COPRA_OK) return 2; // (a kernel that cannot: a new handle can)
            builtCosts_[t].p.assign(now[t].p, now[t].p + now[t].rows);
            rc = 1;
        }
        return rc;
    }
    bool rebuild() // true: the handle is a new one
    {
        if (h_ && !dirty_) {
            const int c = costsAgainstHandle();
            if (c < 2) {
                costsDirty_ = false;
                if (c == 1) qpValid_ = false;
                return false;
            }
        }
        release();
        ++handleBuilds_;
        builtCosts_.clear();
        builtCostsKnown_ = true;
        std::vector<copra_cost_desc_t> cd;
        std::vector<copra_cstr_desc_t> kd;
        for (auto& c : spCost_) {
            copra_cost_desc_t d {};
            if (!c->deviceDescriptor(d)) { // host-evaluated user cost (LMPC.cpp:252-255 / InitialStateLMPC.cpp:80-84)
                d = copra_cost_desc_t {};
                d.kind = COPRA_COST_DENSE;
                if (c->Q().rows() != ps_->fullUDim || c->Q().cols() != ps_->fullUDim || c->c().rows() != ps_->fullUDim
                    || c->E().rows() != ps_->xDim || c->E().cols() != ps_->fullUDim || c->f().rows() != ps_->fullUDim)
                    COPRA_DOMAIN_ERROR("cost '" + c->name() + "': Q / c / E / f do not have the sizes initializeCost gives them");
                d.Q = c->Q().data(), d.c = c->c().data(), d.E = c->E().data(), d.f = c->f().data();
                builtCostsKnown_ = false; // (a host-evaluated cost: the handle is rebuilt for every solve anyway, prepare())
            } else {
                builtCosts_.push_back(snapshot(d));
            }
            cd.push_back(d);
        }
        for (auto& c : spConstr_) {
            copra_cstr_desc_t d {};
            if (!c->deviceDescriptor(d)) {
                d = copra_cstr_desc_t {};
                if (c->constraintType() == ConstraintFlag::BoundConstraint) { // LMPC.cpp:274-279: lower() / upper()
                    auto b = std::static_pointer_cast<ControlBoundConstraint>(c);
                    d.kind = COPRA_CSTR_CONTROL_BOUND;
                    d.rows = (int)b->lower().rows();
                    d.lower = b->lower().data();
                    d.upper = b->upper().data();
                } else { // LMPC.cpp:257-271 / InitialStateLMPC.cpp:88-102: A b Y z
                    auto q = std::static_pointer_cast<EqIneqConstraint>(c);
                    d.kind = COPRA_CSTR_DENSE;
                    d.is_inequality = c->constraintType() == ConstraintFlag::InequalityConstraint ? 1 : 0;
                    d.rows = (int)q->A().rows();
                    if (q->A().cols() != ps_->fullUDim || q->b().rows() != d.rows || q->Y().rows() != d.rows
                        || q->Y().cols() != ps_->xDim || q->z().rows() != d.rows)
                        COPRA_DOMAIN_ERROR("constraint '" + c->name() + "': A / b / Y / z sizes do not match");
                    d.A = q->A().data(), d.b = q->b().data(), d.Y = q->Y().data(), d.z = q->z().data();
                }
            }
            kd.push_back(d);
        }
        const copra_dims_t dims = detail::dims_of(*ps_);
        throw_status(createHandle(dims, cd, kd));
        throw_status(copra_batch_select_solver(h_, engineSolver(flag_)));
        dirty_ = false;
        costsDirty_ = false;
        qpValid_ = false;
        return true;
    }
    // LMPC.cpp:288-307: a cost / constraint the caller has released is dropped after the solve (the controller's own
    // reference is then the only one: one list here, where the reference keeps a master and a typed list)
    void checkDeleteCostsAndConstraints()
    {
        for (auto it = spConstr_.begin(); it != spConstr_.end();) {
            if (it->use_count() <= 1) {
                std::fprintf(stderr, "A '%s' has been destroyed.\nIt has been removed from the controller\n", (*it)->name().c_str());
                it = spConstr_.erase(it);
                dirty_ = true;
            } else {
                ++it;
            }
        }
        for (auto it = spCost_.begin(); it != spCost_.end();) {
            if (it->use_count() <= 1) {
                std::fprintf(stderr, "A '%s' has been destroyed.\nIt has been removed from the controller\n", (*it)->name().c_str());
                it = spCost_.erase(it);
                costsDirty_ = true;
            } else {
                ++it;
            }
        }
    }
    void release()
    {
        if (h_) copra_batch_destroy(h_);
        h_ = nullptr;
    }

    std::shared_ptr<PreviewSystem> ps_;
    std::vector<std::shared_ptr<CostFunction>> spCost_;
    std::vector<std::shared_ptr<Constraint>> spConstr_;
    std::unique_ptr<SolverInterface> sol_; // a user solver (useSolver); null: the fused device solve
    SolverFlag flag_ = SolverFlag::DEFAULT;
    copra_batch_t* h_ = nullptr;
    bool refAccumulation_ = false; // reference quirk Q2 reproduced on request (referenceAccumulation)
    bool dirty_ = true; // the handle has to be built anew
    bool costsDirty_ = false; // the list of costs has changed since the handle was built (rebuild() decides what that needs)
    std::vector<CostSnapshot> builtCosts_;
    bool builtCostsKnown_ = false;
    int handleBuilds_ = 0;
    Eigen::VectorXd control_, trajectory_;
    int fail_ = 0, iter_ = 0;
    double solveTime_ = 0.0, solveAndBuildTime_ = 0.0;
    DenseQP qp_;
    bool qpValid_ = false;
    Eigen::MatrixXd sentA_, sentB_; // the system the device holds (prepare)
    Eigen::VectorXd sentd_;
};

// include/InitialStateLMPC.h:18-42, src/InitialStateLMPC.cpp: the initial state is a decision variable too
class InitialStateLMPC : public LMPC {
public:
    InitialStateLMPC(SolverFlag f = SolverFlag::DEFAULT)
        : LMPC(f)
    {
    }
    InitialStateLMPC(const std::shared_ptr<PreviewSystem>& ps, SolverFlag f = SolverFlag::DEFAULT)
        : LMPC(ps, f)
    {
        // InitialStateLMPC.cpp:20-28: R = 0, r = 0, both bounds = ps->x0
        R_ = Eigen::MatrixXd::Zero(ps->xDim, ps->xDim);
        r_ = Eigen::VectorXd::Zero(ps->xDim);
        x0lb_ = ps->x0;
        x0ub_ = ps->x0;
    }
    Eigen::VectorXd initialState() const noexcept { return x0opt_; } // :30-33
    void resetInitialStateCost(const Eigen::MatrixXd& R, const Eigen::VectorXd& r) // :35-40
    {
        R_ = R;
        r_ = r;
        dirty_ = true;
    }
    void resetInitialStateBounds(const Eigen::VectorXd& l, const Eigen::VectorXd& u) // :42-46
    {
        x0lb_ = l;
        x0ub_ = u;
    }

protected:
    copra_status_t createHandle(const copra_dims_t& dims, const std::vector<copra_cost_desc_t>& cd,
        const std::vector<copra_cstr_desc_t>& kd) override
    {
        copra_initial_state_desc_t is { R_.data(), r_.data() };
        return copra_batch_create_initial_state(&h_, &dims, (int)cd.size(), cd.data(), (int)kd.size(), kd.data(), &is);
    }
    void beforeSolve() override
    {
        throw_status(copra_batch_set_initial_state_bounds(h_, x0lb_.data(), x0ub_.data(), 0));
    }
    void afterSolve(bool ok) override
    {
        if (!ok) return;
        x0opt_.resize(ps_->xDim);
        throw_status(copra_batch_get_initial_state(h_, x0opt_.data()));
    }
    void updateResultsFrom(const Eigen::VectorXd& result) override // InitialStateLMPC.cpp:124-128
    {
        x0opt_ = result.head(ps_->xDim);
        control_ = result.tail(ps_->fullUDim);
        hostTrajectory(x0opt_, control_);
    }
    Eigen::MatrixXd R_;
    Eigen::VectorXd r_, x0lb_, x0ub_, x0opt_;
};

} // namespace copra
