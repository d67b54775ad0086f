// copra.h -- host-side C++ mirror of copra's controller API on top of the C ABI (include/copra_hip.h).
//
// Same names, constructor arguments and error behaviour as the reference headers
//   include/PreviewSystem.h, costFunctions.h, constraints.h, LMPC.h, SolverInterface.h, solverUtils.h, AutoSpan.h
// so that code (and tests) written against copra compile against this header; all arithmetic of LMPC::solve runs on
// the GPU (one fused launch), nothing is computed on the host.  Differences, all documented in DESIGN.md:
//   * the cost / constraint classes are DESCRIPTORS (the built-in nine); user subclasses of CostFunction / Constraint
//     with their own update() are not supported on this path -- plug a custom QP pipeline in through
//     SolverInterface (HipQuadProgSolver) instead;
//   * every solve() has fresh-controller semantics (reference quirk Q2 -- accumulation across solves -- is not kept);
//   * PreviewSystem has no public Phi / Psi / xi: the preview matrices only ever exist in LDS.
#pragma once

#include "../../../../include/copra_hip.h"
#include "../../../csrc/plan_builder.hpp" // dimension checks identical to copra_batch_create (pure host code)
#include "EigenCompat.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <limits>
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
    void updateSystem() noexcept { isUpdated = true; } // Phi / Psi / xi are built on the device inside LMPC::solve
    void xInit(const Eigen::VectorXd& xInit) { x0 = xInit; }

    bool isUpdated = false;
    int nrUStep = 0, nrXStep = 0, xDim = 0, uDim = 0, fullXDim = 0, fullUDim = 0;
    Eigen::VectorXd x0;
    Eigen::MatrixXd A, B;
    Eigen::VectorXd d;
};

// ---------------------------------------------------------------------------------------------- costFunctions.h
class CostFunction {
public:
    explicit CostFunction(std::string&& name, int kind)
        : name_(std::move(name))
        , kind_(kind)
    {
    }
    virtual ~CostFunction() = default;
    virtual void autoSpan() {}
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
    void checkRows() const
    { // costFunctions.cpp:47-49, 91-93, 125-127, 176-181
        if (M_.size() && M_.rows() != p_.rows()) COPRA_DOMAIN_ERROR("M and p should have the same number of rows (try autoSpan)");
        if (N_.size() && N_.rows() != p_.rows()) COPRA_DOMAIN_ERROR("N and p should have the same number of rows (try autoSpan)");
    }

protected:
    std::string name_;
    int kind_;
    Eigen::MatrixXd M_, N_;
    Eigen::VectorXd p_, weights_;
};

class TrajectoryCost final : public CostFunction { // costFunctions.h:103-126
public:
    TrajectoryCost(const Eigen::MatrixXd& M, const Eigen::VectorXd& p)
        : CostFunction("TrajectoryCost", COPRA_COST_TRAJECTORY)
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
class TargetCost final : public CostFunction { // costFunctions.h:134-155
public:
    TargetCost(const Eigen::MatrixXd& M, const Eigen::VectorXd& p)
        : CostFunction("TargetCost", COPRA_COST_TARGET)
    {
        M_ = M;
        p_ = p;
        weights_ = Eigen::VectorXd::Ones(p_.rows());
    }
};
class ControlCost final : public CostFunction { // costFunctions.h:163-186
public:
    ControlCost(const Eigen::MatrixXd& N, const Eigen::VectorXd& p)
        : CostFunction("ControlCost", COPRA_COST_CONTROL)
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
class MixedCost final : public CostFunction { // costFunctions.h:194-219
public:
    MixedCost(const Eigen::MatrixXd& M, const Eigen::MatrixXd& N, const Eigen::VectorXd& p)
        : CostFunction("MixedCost", COPRA_COST_MIXED)
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
enum class ConstraintFlag { Constraint, EqualityConstraint, InequalityConstraint, BoundConstraint };

class Constraint {
public:
    Constraint(std::string&& name, int kind, bool ineq)
        : name_(std::move(name))
        , kind_(kind)
        , isIneq_(ineq)
    {
    }
    virtual ~Constraint() = default;
    virtual void autoSpan() = 0;
    virtual ConstraintFlag constraintType() const noexcept
    {
        return isIneq_ ? ConstraintFlag::InequalityConstraint : ConstraintFlag::EqualityConstraint;
    }
    const std::string& name() const noexcept { return name_; }
    int nrConstr() noexcept { return nrConstr_; }
    copra_cstr_desc_t desc() const
    {
        copra_cstr_desc_t d {};
        d.kind = kind_;
        d.is_inequality = isIneq_ ? 1 : 0;
        if (kind_ == COPRA_CSTR_TRAJECTORY_BOUND || kind_ == COPRA_CSTR_CONTROL_BOUND) {
            d.rows = (int)lower_.rows();
            d.lower = lower_.data();
            d.upper = upper_.data();
        } else {
            d.rows = (int)f_.rows();
            d.e_cols = (int)E_.cols();
            d.g_cols = (int)G_.cols();
            d.E = E_.size() ? E_.data() : nullptr;
            d.G = G_.size() ? G_.data() : nullptr;
            d.f = f_.data();
        }
        return d;
    }
    void checkRows() const
    {
        if (kind_ == COPRA_CSTR_TRAJECTORY_BOUND || kind_ == COPRA_CSTR_CONTROL_BOUND) {
            if (lower_.rows() != upper_.rows()) COPRA_DOMAIN_ERROR("lower and upper should have the same number of rows (try autoSpan)");
        } else {
            if (E_.size() && E_.rows() != f_.rows()) COPRA_DOMAIN_ERROR("E and f should have the same number of rows (try autoSpan)");
            if (G_.size() && G_.rows() != f_.rows()) COPRA_DOMAIN_ERROR("G and f should have the same number of rows (try autoSpan)");
        }
    }
    // move-consuming classes may be initialised once (constraints.cpp:108-110, 335-337)
    bool consumesOnInit() const { return kind_ == COPRA_CSTR_CONTROL || kind_ == COPRA_CSTR_CONTROL_BOUND; }
    bool hasBeenInitialized_ = false;
    void setNrConstr(int n) { nrConstr_ = n; }

protected:
    std::string name_;
    int kind_;
    bool isIneq_;
    int nrConstr_ = 0;
    Eigen::MatrixXd E_, G_;
    Eigen::VectorXd f_, lower_, upper_;
};

class TrajectoryConstraint final : public Constraint { // constraints.h:114-145
public:
    TrajectoryConstraint(const Eigen::MatrixXd& E, const Eigen::VectorXd& f, bool isInequalityConstraint = true)
        : Constraint(std::string("Trajectory") + (isInequalityConstraint ? " inequality constraint" : " equality constraint"),
            COPRA_CSTR_TRAJECTORY, isInequalityConstraint)
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
class ControlConstraint final : public Constraint { // constraints.h:153-185
public:
    ControlConstraint(const Eigen::MatrixXd& G, const Eigen::VectorXd& f, bool isInequalityConstraint = true)
        : Constraint(std::string("Control") + (isInequalityConstraint ? " inequality constraint" : " equality constraint"),
            COPRA_CSTR_CONTROL, isInequalityConstraint)
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
class MixedConstraint final : public Constraint { // constraints.h:193-226
public:
    MixedConstraint(const Eigen::MatrixXd& E, const Eigen::MatrixXd& G, const Eigen::VectorXd& f,
        bool isInequalityConstraint = true)
        : Constraint(std::string("Control") + (isInequalityConstraint ? " inequality constraint" : " equality constraint"),
            COPRA_CSTR_MIXED, isInequalityConstraint) // name quirk Q4 kept (constraints.h:204)
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
class TrajectoryBoundConstraint final : public Constraint { // constraints.h:234-276
public:
    TrajectoryBoundConstraint(const Eigen::VectorXd& lower, const Eigen::VectorXd& upper)
        : Constraint("Trajectory bound inequality constraint", COPRA_CSTR_TRAJECTORY_BOUND, true)
    {
        lower_ = lower;
        upper_ = upper;
    }
    void autoSpan() override // constraints.cpp:232-261
    {
        const auto m = std::max(lower_.rows(), upper_.rows());
        AutoSpan::spanVector(lower_, m);
        AutoSpan::spanVector(upper_, m);
    }
    ConstraintFlag constraintType() const noexcept override { return ConstraintFlag::InequalityConstraint; }
};
class ControlBoundConstraint final : public Constraint { // constraints.h:284-308
public:
    ControlBoundConstraint(const Eigen::VectorXd& lower, const Eigen::VectorXd& upper)
        : Constraint("Control bound constraint", COPRA_CSTR_CONTROL_BOUND, true)
    {
        lower_ = lower;
        upper_ = upper;
    }
    void autoSpan() override // constraints.cpp:326-331
    {
        const auto m = std::max(lower_.rows(), upper_.rows());
        AutoSpan::spanVector(lower_, m);
        AutoSpan::spanVector(upper_, m);
    }
    ConstraintFlag constraintType() const noexcept override { return ConstraintFlag::BoundConstraint; }
};

// ---------------------------------------------------------------------------------------------- SolverInterface.h
class SolverInterface { // include/SolverInterface.h:19-81
public:
    SolverInterface() = default;
    virtual ~SolverInterface() = default;
    virtual int SI_fail() const = 0;
    virtual void SI_inform() const = 0;
    virtual int SI_iter() const
    {
        std::printf("No iter() function for this qp\n"); // SolverInterface.cpp:15-19
        return 0;
    }
    virtual const Eigen::VectorXd& SI_result() const = 0;
    virtual void SI_problem(int nrVar, int nrEq, int nrInEq) = 0;
    virtual bool SI_solve(const Eigen::MatrixXd& Q, const Eigen::VectorXd& c, const Eigen::MatrixXd& Aeq,
        const Eigen::VectorXd& beq, const Eigen::MatrixXd& Aineq, const Eigen::VectorXd& bineq,
        const Eigen::VectorXd& XL, const Eigen::VectorXd& XU)
        = 0;
};

// Plug-in point 1 on the GPU: replaces QuadProgDenseSolver (include/QuadProgSolver.h:16-42, src/QuadProgSolver.cpp:14-72)
class HipQuadProgSolver : public SolverInterface {
public:
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

enum class SolverFlag { DEFAULT, QuadProgDense, HipQuadProg }; // include/solverUtils.h:34-50 (+ the HIP back-end)
inline std::unique_ptr<SolverInterface> solverFactory(SolverFlag) // src/solverUtils.cpp:9-34
{
    return std::unique_ptr<SolverInterface>(new HipQuadProgSolver());
}

// ---------------------------------------------------------------------------------------------- LMPC.h
class LMPC { // include/LMPC.h:36-191, src/LMPC.cpp
public:
    LMPC(SolverFlag = SolverFlag::DEFAULT) {}
    LMPC(const std::shared_ptr<PreviewSystem>& ps, SolverFlag = SolverFlag::DEFAULT) { initializeController(ps); }
    virtual ~LMPC() { release(); }
    LMPC(LMPC&& o) noexcept { *this = std::move(o); }
    LMPC& operator=(LMPC&& o) noexcept
    {
        release();
        ps_ = std::move(o.ps_);
        spCost_ = std::move(o.spCost_);
        spConstr_ = std::move(o.spConstr_);
        h_ = o.h_;
        o.h_ = nullptr;
        dirty_ = o.dirty_;
        control_ = std::move(o.control_);
        trajectory_ = std::move(o.trajectory_);
        return *this;
    }
    void selectQPSolver(SolverFlag) {} // the fused kernel is the solver on this path
    void initializeController(const std::shared_ptr<PreviewSystem>& ps)
    {
        ps_ = ps;
        dirty_ = true;
    }
    void addCost(const std::shared_ptr<CostFunction>& cost) // LMPC.cpp:118-122 -> initializeCost
    {
        cost->checkRows();
        validate({ cost->desc() }, {});
        spCost_.push_back(cost);
        dirty_ = true;
    }
    void addConstraint(const std::shared_ptr<Constraint>& c) // LMPC.cpp:124-128 -> initializeConstraint
    {
        if (c->consumesOnInit() && c->hasBeenInitialized_)
            COPRA_RUNTIME_ERROR("You have initialized a " + c->name() + " twice. As move semantics are used, you can't do so.");
        c->checkRows();
        validate({}, { c->desc() });
        c->hasBeenInitialized_ = true;
        spConstr_.push_back(c);
        dirty_ = true;
    }
    void clearCosts() noexcept
    {
        spCost_.clear();
        dirty_ = true;
    }
    void clearConstraints() noexcept
    {
        spConstr_.clear();
        dirty_ = true;
    }
    void removeCost(const std::shared_ptr<CostFunction>& c)
    {
        auto it = std::find(spCost_.begin(), spCost_.end(), c);
        if (it != spCost_.end()) spCost_.erase(it), dirty_ = true;
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
        rebuild();
        throw_status(copra_batch_set_system(h_, ps_->A.data(), ps_->B.data(), ps_->d.data(), ps_->x0.data(), 0));
        beforeSolve();
        ps_->isUpdated = true;
        qpValid_ = false;
        const auto t1 = clock::now();
        throw_status(copra_batch_solve(h_, nullptr));
        Eigen::VectorXd u(ps_->fullUDim), x(ps_->fullXDim);
        int it[2];
        throw_status(copra_batch_get_results(h_, u.data(), x.data(), &fail_, it));
        afterSolve(fail_ == 0);
        double dev = 0.0;
        copra_batch_last_solve_seconds(h_, &dev);
        solveTime_ = dev;
        iter_ = it[0];
        if (fail_ == 0) { // LMPC.cpp:95-97: outputs only updated on success
            control_ = u;
            trajectory_ = x;
        }
        (void)t1;
        solveAndBuildTime_ = std::chrono::duration<double>(clock::now() - t0).count();
        return fail_ == 0;
    }
    void inform() const noexcept
    {
        HipQuadProgSolver s;
        (void)s;
        std::printf("%s\n", fail_ == 0 ? "No problems" : fail_ == 1 ? "The minimization problem has no solution"
                                                                    : "Problems with the decomposition of Q (Is it symmetric?)");
    }
    double solveTime() const noexcept { return solveTime_; } // device time of the fused launch
    double solveAndBuildTime() const noexcept { return solveAndBuildTime_; }
    const Eigen::VectorXd& control() const noexcept { return control_; }
    const Eigen::VectorXd& trajectory() const noexcept { return trajectory_; }
    int fail() const noexcept { return fail_; }
    int iter() const noexcept { return iter_; }
    int nrEqConstr()
    {
        rebuild();
        int n, e, i;
        copra_batch_qp_sizes(h_, &n, &e, &i);
        return e;
    }
    int nrIneqConstr()
    {
        rebuild();
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
        rebuild();
        throw_status(copra_batch_set_system(h_, ps_->A.data(), ps_->B.data(), ps_->d.data(), ps_->x0.data(), 0));
        beforeSolve();
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
    void validate(const std::vector<copra_cost_desc_t>& costs, const std::vector<copra_cstr_desc_t>& cstrs) const
    {
        // the same host-side checks copra_batch_create runs (plan_builder.hpp), without touching the device
        copra_hip::HostPlan hp;
        copra_dims_t dims { ps_->xDim, ps_->uDim, ps_->nrUStep, 1 };
        const copra_status_t rc = copra_hip::build_plan(hp, dims, (int)costs.size(), costs.data(), (int)cstrs.size(), cstrs.data());
        if (rc == COPRA_ERR_DOMAIN) throw std::domain_error(hp.error);
        if (rc == COPRA_ERR_RUNTIME) throw std::runtime_error(hp.error);
    }
    void rebuild()
    {
        if (!dirty_ && h_) return;
        release();
        std::vector<copra_cost_desc_t> cd;
        std::vector<copra_cstr_desc_t> kd;
        for (auto& c : spCost_) cd.push_back(c->desc());
        for (auto& c : spConstr_) kd.push_back(c->desc());
        copra_dims_t dims { ps_->xDim, ps_->uDim, ps_->nrUStep, 1 };
        throw_status(createHandle(dims, cd, kd));
        dirty_ = false;
        qpValid_ = false;
    }
    void release()
    {
        if (h_) copra_batch_destroy(h_);
        h_ = nullptr;
    }

    std::shared_ptr<PreviewSystem> ps_;
    std::vector<std::shared_ptr<CostFunction>> spCost_;
    std::vector<std::shared_ptr<Constraint>> spConstr_;
    copra_batch_t* h_ = nullptr;
    bool dirty_ = true;
    Eigen::VectorXd control_, trajectory_;
    int fail_ = 0, iter_ = 0;
    double solveTime_ = 0.0, solveAndBuildTime_ = 0.0;
    DenseQP qp_;
    bool qpValid_ = false;
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
    Eigen::MatrixXd R_;
    Eigen::VectorXd r_, x0lb_, x0ub_, x0opt_;
};

} // namespace copra
