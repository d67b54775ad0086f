// Tests of the C++ API mirror (copra_amd/cpp/include/copra/copra.h), written like the reference's own doctest cases
// (tests/TestLMPC.cpp) on the reference's fixtures (tests/systems.h).  Usage: test_api errors | solve
//   errors : TestLMPC.cpp:949-1087 (ERROR_HANDLER_*) -- host-side checks only, runs without a GPU
//   solve  : TestLMPC.cpp:36-97, 415-479, 593-670 property checks on the GPU (horizon 12 to fit the one-wave kernel)
#include <copra/copra.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static int failures = 0;
#define CHECK(cond)                                                                                                   \
    do {                                                                                                              \
        if (!(cond)) {                                                                                                \
            std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond);                                       \
            ++failures;                                                                                               \
        }                                                                                                             \
    } while (0)
#define REQUIRE_THROWS_AS(expr, type)                                                                                 \
    do {                                                                                                              \
        bool ok_ = false;                                                                                             \
        try {                                                                                                         \
            expr;                                                                                                     \
        } catch (const type&) {                                                                                       \
            ok_ = true;                                                                                               \
        } catch (...) {                                                                                               \
        }                                                                                                             \
        if (!ok_) {                                                                                                   \
            std::printf("expected %s from %s (%s:%d)\n", #type, #expr, __FILE__, __LINE__);                           \
            ++failures;                                                                                               \
        }                                                                                                             \
    } while (0)
#define REQUIRE_NOTHROW(expr)                                                                                         \
    do {                                                                                                              \
        try {                                                                                                         \
            expr;                                                                                                     \
        } catch (const std::exception& e) {                                                                           \
            std::printf("unexpected exception from %s: %s (%s:%d)\n", #expr, e.what(), __FILE__, __LINE__);           \
            ++failures;                                                                                               \
        }                                                                                                             \
    } while (0)

// tests/systems.h:42-229: the falling-mass fixture all four reference systems share (nbStep = 300 there)
struct IneqSystem {
    explicit IneqSystem(int steps = 12)
        : T(0.005), mass(5), nbStep(steps), A(2, 2), B(2, 1), G(1, 1), E(1, 2), M(2, 2), N(1, 1), c(2), h(1), p(1), x0(2), xd(2), ud(1), wx(2), wu(1)
    {
        A << 1, T, 0, 1;
        B << 0.5 * T * T / mass, T / mass;
        c << (-9.81 / 2.) * T * T, -9.81 * T;
        G << 1;
        h << 200;
        E << 0, 1;
        p << 0;
        x0 << 0, -5;
        wx << 10, 10000;
        wu << 1e-4;
        M << 1, 0, 0, 1;
        N << 1;
        xd << 0, -1;
        ud << 2;
    }
    double T, mass;
    int nbStep;
    Eigen::MatrixXd A, B, G, E, M, N;
    Eigen::VectorXd c, h, p, x0, xd, ud, wx, wu;
};

static void error_handlers()
{
    IneqSystem s;
    using namespace Eigen;
    { // ERROR_HANDLER_FOR_PREVIEW_SYSTEM (TestLMPC.cpp:949-957)
        auto ps = std::make_shared<copra::PreviewSystem>();
        REQUIRE_THROWS_AS(ps->system(MatrixXd::Ones(5, 2), s.B, s.c, s.x0, s.nbStep), std::domain_error);
        REQUIRE_THROWS_AS(ps->system(MatrixXd::Ones(2, 5), s.B, s.c, s.x0, s.nbStep), std::domain_error);
        REQUIRE_THROWS_AS(ps->system(s.A, MatrixXd::Ones(5, 1), s.c, s.x0, s.nbStep), std::domain_error);
        REQUIRE_THROWS_AS(ps->system(s.A, s.B, VectorXd::Ones(5), s.x0, s.nbStep), std::domain_error);
        REQUIRE_THROWS_AS(ps->system(s.A, s.B, s.c, s.x0, -1), std::domain_error);
    }
    auto ps = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, s.x0, s.nbStep);
    { // ERROR_HANDLER_FOR_WEIGTHS (TestLMPC.cpp:959-971)
        auto controller = copra::LMPC(ps);
        auto cost = std::make_shared<copra::TrajectoryCost>(s.M, s.xd);
        REQUIRE_NOTHROW(cost->weight(2));
        REQUIRE_THROWS_AS(cost->weights(VectorXd::Ones(5)), std::domain_error);
        REQUIRE_NOTHROW(cost->weights(s.wx));
        REQUIRE_NOTHROW(controller.addCost(cost));
        REQUIRE_NOTHROW(cost->weights(VectorXd::Ones(2)));
    }
    { // ERROR_HANDLER_FOR_{TRAJECTORY,TARGET,CONTROL,MIXED}_COST (TestLMPC.cpp:973-1017)
        auto controller = copra::LMPC(ps);
        REQUIRE_THROWS_AS(controller.addCost(std::make_shared<copra::TrajectoryCost>(MatrixXd::Identity(5, 5), VectorXd::Ones(2))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addCost(std::make_shared<copra::TrajectoryCost>(MatrixXd::Identity(5, 5), VectorXd::Ones(5))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addCost(std::make_shared<copra::TargetCost>(MatrixXd::Identity(5, 5), VectorXd::Ones(2))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addCost(std::make_shared<copra::TargetCost>(MatrixXd::Identity(5, 5), VectorXd::Ones(5))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addCost(std::make_shared<copra::ControlCost>(MatrixXd::Identity(5, 5), VectorXd::Ones(2))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addCost(std::make_shared<copra::ControlCost>(MatrixXd::Identity(5, 5), VectorXd::Ones(5))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addCost(std::make_shared<copra::MixedCost>(MatrixXd::Identity(5, 5), MatrixXd::Identity(2, 1), VectorXd::Ones(2))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addCost(std::make_shared<copra::MixedCost>(MatrixXd::Identity(2, 1), MatrixXd::Identity(5, 5), VectorXd::Ones(2))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addCost(std::make_shared<copra::MixedCost>(MatrixXd::Identity(5, 5), MatrixXd::Identity(5, 5), VectorXd::Ones(5))), std::domain_error);
    }
    { // ERROR_HANDLER_FOR_*_CONSTRAINT (TestLMPC.cpp:1019-1087)
        auto controller = copra::LMPC(ps);
        REQUIRE_THROWS_AS(controller.addConstraint(std::make_shared<copra::TrajectoryConstraint>(MatrixXd::Identity(5, 5), VectorXd::Ones(2))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addConstraint(std::make_shared<copra::TrajectoryConstraint>(MatrixXd::Identity(5, 5), VectorXd::Ones(5))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addConstraint(std::make_shared<copra::ControlConstraint>(MatrixXd::Identity(5, 5), VectorXd::Ones(2))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addConstraint(std::make_shared<copra::ControlConstraint>(MatrixXd::Identity(5, 5), VectorXd::Ones(5))), std::domain_error);
        auto goodConstr = std::make_shared<copra::ControlConstraint>(s.G, s.h);
        REQUIRE_NOTHROW(controller.addConstraint(goodConstr));
        REQUIRE_THROWS_AS(controller.addConstraint(goodConstr), std::runtime_error);
        REQUIRE_THROWS_AS(controller.addConstraint(std::make_shared<copra::MixedConstraint>(MatrixXd::Identity(5, 5), MatrixXd::Identity(2, 1), VectorXd::Ones(2))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addConstraint(std::make_shared<copra::MixedConstraint>(MatrixXd::Identity(2, 1), MatrixXd::Identity(5, 5), VectorXd::Ones(2))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addConstraint(std::make_shared<copra::MixedConstraint>(MatrixXd::Identity(5, 5), MatrixXd::Identity(5, 5), VectorXd::Ones(5))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addConstraint(std::make_shared<copra::TrajectoryBoundConstraint>(VectorXd::Ones(3), VectorXd::Ones(2))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addConstraint(std::make_shared<copra::TrajectoryBoundConstraint>(VectorXd::Ones(3), VectorXd::Ones(3))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addConstraint(std::make_shared<copra::ControlBoundConstraint>(VectorXd::Ones(3), VectorXd::Ones(2))), std::domain_error);
        REQUIRE_THROWS_AS(controller.addConstraint(std::make_shared<copra::ControlBoundConstraint>(VectorXd::Ones(3), VectorXd::Ones(3))), std::domain_error);
        VectorXd uLower(1), uUpper(1);
        uLower.setConstant(-std::numeric_limits<double>::infinity());
        uUpper.setConstant(200);
        auto goodBound = std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper);
        REQUIRE_NOTHROW(controller.addConstraint(goodBound));
        REQUIRE_THROWS_AS(controller.addConstraint(goodBound), std::runtime_error);
    }
    { // CHECK_AUTOSPAN_* (TestLMPC.cpp:777-943): per-step entries spanned to the full horizon are accepted
        auto controller = copra::LMPC(ps);
        auto cstr = std::make_shared<copra::TrajectoryConstraint>(s.E, VectorXd::Ones(s.nbStep + 1));
        cstr->autoSpan();
        REQUIRE_NOTHROW(controller.addConstraint(cstr));
        auto cstr2 = std::make_shared<copra::MixedConstraint>(s.E, s.G, VectorXd::Ones(s.nbStep));
        cstr2->autoSpan();
        REQUIRE_NOTHROW(controller.addConstraint(cstr2));
    }
}

// Largest velocity / position over a trajectory vector [p0, v0, p1, v1, ...]
static void extrema(const Eigen::VectorXd& traj, double& posMax, double& velMax)
{
    posMax = velMax = -std::numeric_limits<double>::infinity();
    for (Eigen::Index i = 0; i < traj.rows() / 2; ++i) {
        posMax = std::max(posMax, traj(2 * i));
        velMax = std::max(velMax, traj(2 * i + 1));
    }
}

// The state cost of one of the reference's test cases: TargetCost, TrajectoryCost, or the MixedCost pair of
// TestLMPC.cpp:182-183 (MixedCost(M, 0, xd) + MixedCost(0, N, ud)); always followed by the control cost.
// The caller keeps what it adds alive, as the reference's tests do: a piece only the controller still refers to is
// dropped after the next solve (LMPC::checkDeleteCostsAndConstraints, src/LMPC.cpp:288-307).
static std::vector<std::shared_ptr<void>> g_keep;
template <class T>
static std::shared_ptr<T> hold(std::shared_ptr<T> p)
{
    g_keep.push_back(p);
    return p;
}

static void add_costs(copra::LMPC& controller, const IneqSystem& s, const Eigen::VectorXd& xd, int kind)
{
    if (kind == 2) {
        auto xCost = std::make_shared<copra::MixedCost>(s.M, Eigen::MatrixXd::Zero(2, 1), xd);
        auto uCost = std::make_shared<copra::MixedCost>(Eigen::MatrixXd::Zero(1, 2), s.N, s.ud);
        xCost->weights(s.wx);
        uCost->weights(s.wu);
        controller.addCost(hold(xCost));
        controller.addCost(hold(uCost));
        return;
    }
    std::shared_ptr<copra::CostFunction> xCost;
    if (kind == 0)
        xCost = std::make_shared<copra::TargetCost>(s.M, xd);
    else
        xCost = std::make_shared<copra::TrajectoryCost>(s.M, xd);
    auto uCost = std::make_shared<copra::ControlCost>(s.N, s.ud);
    xCost->weights(s.wx);
    uCost->weights(s.wu);
    controller.addCost(hold(xCost));
    controller.addCost(hold(uCost));
}

// TestLMPC.cpp:36-771 at the reference's own horizon (nbStep = 300 -> 300 decision variables): every
// {Target, Trajectory, Mixed cost} x {bound, inequality, mixed, equality constraint} case with its acceptance checks.
static void solve_cases(int nbStep)
{
    IneqSystem s(nbStep);
    const double inf = std::numeric_limits<double>::infinity();
    const char* costName[3] = { "TARGET", "TRAJECTORY", "MIXED" };
    for (int kind = 0; kind < 3; ++kind) {
        { // MPC_<cost>_COST_WITH_BOUND_CONSTRAINTS
            Eigen::VectorXd uLower(1), uUpper(1), xLower(2), xUpper(2);
            uLower.setConstant(-inf);
            uUpper.setConstant(200);
            xLower.setConstant(-inf);
            xUpper << inf, 0;
            auto ps = std::make_shared<copra::PreviewSystem>();
            ps->system(s.A, s.B, s.c, s.x0, s.nbStep);
            auto controller = copra::LMPC(ps);
            add_costs(controller, s, s.xd, kind);
            controller.addConstraint(hold(std::make_shared<copra::TrajectoryBoundConstraint>(xLower, xUpper)));
            controller.addConstraint(hold(std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper)));
            CHECK(controller.solve());
            double posMax, velMax;
            extrema(controller.trajectory(), posMax, velMax);
            CHECK(posMax <= s.x0(0) + 1e-9);
            CHECK(velMax <= 0 + 1e-6);
            CHECK(controller.control().maxCoeff() <= 200 + 1e-6);
            if (kind == 0 && nbStep >= 300) // the target is reachable over the full horizon (TestLMPC.cpp:78-83)
                CHECK(std::fabs(controller.trajectory()(2 * s.nbStep + 1) - s.xd(1)) <= 1e-3);
            CHECK(controller.solveTime() > 0 && controller.solveAndBuildTime() >= controller.solveTime());
            // src/solverUtils.cpp:9-34: SolverFlag::DEFAULT IS QuadProgDense -- at every size (round-4 verdict: above 64 variables the
            // mirror's DEFAULT silently ran the interior-point kernel); that kernel is an explicit opt-in, SolverFlag::HipRiccati
            CHECK(controller.solverKind() == COPRA_SOLVER_QUADPROG_DENSE);
            if (kind == 0 && nbStep > 64) {
                const Eigen::VectorXd uGI = controller.control();
                const int itGI = controller.iter();
                auto fast = copra::LMPC(ps, copra::SolverFlag::HipRiccati);
                add_costs(fast, s, s.xd, kind);
                fast.addConstraint(hold(std::make_shared<copra::TrajectoryBoundConstraint>(xLower, xUpper)));
                fast.addConstraint(hold(std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper)));
                CHECK(fast.solverKind() == COPRA_SOLVER_RICCATI_IPM);
                CHECK(fast.solve());
                CHECK(itGI >= 1 && fast.iter() >= 1);
                double du = 0.0, umax = 1.0;
                for (int i = 0; i < (int)uGI.rows(); ++i) {
                    du = std::max(du, std::fabs(fast.control()(i) - uGI(i)));
                    umax = std::max(umax, std::fabs(uGI(i)));
                }
                CHECK(du <= 1e-6 * umax);
                fast.selectQPSolver(copra::SolverFlag::DEFAULT); // ... and back: the reference's meaning again
                CHECK(fast.solverKind() == COPRA_SOLVER_QUADPROG_DENSE);
            }
            if (kind == 0) { // receding horizon: xInit without re-creating anything (PreviewSystem.h:52)
                Eigen::VectorXd x1(2);
                x1 << controller.trajectory()(2), controller.trajectory()(3);
                ps->xInit(x1);
                CHECK(controller.solve());
                CHECK(std::fabs(controller.trajectory()(0) - x1(0)) < 1e-12);
            }
        }
        { // MPC_<cost>_COST_WITH_INEQUALITY_CONSTRAINTS: E x <= f (velocity <= 0), G u <= h
            Eigen::VectorXd f(1);
            f << 0;
            auto ps = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, s.x0, s.nbStep);
            auto controller = copra::LMPC(ps);
            add_costs(controller, s, s.xd, kind);
            controller.addConstraint(hold(std::make_shared<copra::TrajectoryConstraint>(s.E, f)));
            controller.addConstraint(hold(std::make_shared<copra::ControlConstraint>(s.G, s.h)));
            CHECK(controller.solve());
            double posMax, velMax;
            extrema(controller.trajectory(), posMax, velMax);
            CHECK(velMax <= 0 + 1e-6);
            CHECK(controller.control().maxCoeff() <= 200 + 1e-6);
            CHECK(controller.nrIneqConstr() == (s.nbStep + 1) + s.nbStep);
        }
        { // MPC_<cost>_COST_WITH_MIXED_CONSTRAINTS: v_k + u_k <= 200
            Eigen::VectorXd p(1);
            p << 200;
            auto ps = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, s.x0, s.nbStep);
            auto controller = copra::LMPC(ps);
            add_costs(controller, s, s.xd, kind);
            controller.addConstraint(hold(std::make_shared<copra::MixedConstraint>(s.E, s.G, p)));
            CHECK(controller.solve());
            Eigen::VectorXd fullTraj = controller.trajectory(), control = controller.control();
            for (int i = 0; i < s.nbStep; ++i) CHECK(fullTraj(2 * i + 1) + control(i) <= 200 + 1e-6);
        }
        { // MPC_<cost>_COST_WITH_EQUALITY_CONSTRAINTS: position pinned to 0 -> u_k = m g
            Eigen::MatrixXd E = Eigen::MatrixXd::Zero(2, 2);
            E(0, 0) = 1;
            Eigen::VectorXd x0 = Eigen::VectorXd::Zero(2), xd = Eigen::VectorXd::Zero(2);
            auto ps = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, x0, s.nbStep);
            auto controller = copra::LMPC(ps);
            add_costs(controller, s, xd, kind);
            controller.addConstraint(hold(std::make_shared<copra::TrajectoryConstraint>(E, x0, false)));
            CHECK(controller.solve());
            // the last control has no effect on the pinned positions (x_N depends on u_0 .. u_{N-2} only)
            for (int i = 0; i + 1 < s.nbStep; ++i) CHECK(std::fabs(controller.control()(i) - 49.05) < 1e-4);
            for (int i = 0; i <= s.nbStep; ++i) CHECK(std::fabs(controller.trajectory()(2 * i)) < 1e-6);
            CHECK(controller.nrEqConstr() == 2 * (s.nbStep + 1));
        }
        std::printf("  %s cost: bound / inequality / mixed / equality cases done (%d failure(s) so far)\n", costName[kind], failures);
    }
    { // TestSolvers.cpp:25-33 through plug-in point 1 on the GPU
        Eigen::MatrixXd Q = Eigen::MatrixXd::Identity(6, 6), Aeq(3, 6), Aineq(2, 6);
        Eigen::VectorXd c(6), beq(3), bineq(2), XL(6), XU(6);
        c << 1, 2, 3, 4, 5, 6;
        Aeq << 1, -1, 1, 0, 3, 1, -1, 0, -3, -4, 5, 6, 2, 5, 3, 0, 1, 0;
        beq << 1, 2, 3;
        Aineq << 0, 1, 0, 1, 2, -1, -1, 0, 2, 1, 1, 0;
        bineq << -1, 2.5;
        XL << -1000, -10000, 0, -1000, -1000, -1000;
        XU << 10000, 100, 1.5, 100, 100, 1000;
        auto qp = copra::solverFactory(copra::SolverFlag::HipQuadProg);
        qp->SI_problem(6, 3, 2);
        CHECK(qp->SI_solve(Q, c, Aeq, beq, Aineq, bineq, XL, XU));
        CHECK(qp->SI_fail() == 0);
        const double xs[6] = { 1.7975426035, -0.3381487238, 0.1633880281, -4.9884022703, 0.6054943277, -3.1155623387 };
        for (int i = 0; i < 6; ++i) CHECK(std::fabs(qp->SI_result()(i) - xs[i]) < 1e-9);
    }
}

// TestLMPC_InitialState.cpp: all nine cost / constraint classes on A = ones(2,2), B = ones(2,1), 10 steps, with
// per-step entries and with the full-size entries autoSpan() produces.
struct NineClasses {
    std::vector<std::shared_ptr<copra::CostFunction>> costs;
    std::vector<std::shared_ptr<copra::Constraint>> cstrs;
    NineClasses(bool fullSize, int steps)
    {
        using namespace Eigen;
        const int U = fullSize ? steps : 1, X = fullSize ? steps + 1 : 1;
        const double factor = 10, inf = std::numeric_limits<double>::infinity();
        auto tc = std::make_shared<copra::TrajectoryCost>(MatrixXd::Ones(1, 2), factor * VectorXd::Ones(X));
        auto tac = std::make_shared<copra::TargetCost>(MatrixXd::Ones(1, 2), factor * VectorXd::Ones(1));
        auto cc = std::make_shared<copra::ControlCost>(MatrixXd::Ones(1, 1), factor * VectorXd::Ones(U));
        auto mc = std::make_shared<copra::MixedCost>(MatrixXd::Ones(1, 2), MatrixXd::Ones(1, 1), factor * VectorXd::Ones(U));
        auto tk = std::make_shared<copra::TrajectoryConstraint>(MatrixXd::Ones(1, 2), factor * VectorXd::Ones(X), true);
        auto ck = std::make_shared<copra::ControlConstraint>(MatrixXd::Ones(1, 1), factor * VectorXd::Ones(U), true);
        auto mk = std::make_shared<copra::MixedConstraint>(MatrixXd::Ones(1, 2), MatrixXd::Ones(1, 1), factor * VectorXd::Ones(U), true);
        auto tb = std::make_shared<copra::TrajectoryBoundConstraint>(-inf * VectorXd::Ones(2 * X), inf * VectorXd::Ones(2 * X));
        auto cb = std::make_shared<copra::ControlBoundConstraint>(-3.0 * VectorXd::Ones(U), 3.0 * VectorXd::Ones(U));
        for (auto& c : { std::static_pointer_cast<copra::CostFunction>(tc), std::static_pointer_cast<copra::CostFunction>(tac),
                 std::static_pointer_cast<copra::CostFunction>(cc), std::static_pointer_cast<copra::CostFunction>(mc) }) {
            c->autoSpan();
            c->weight(1);
            costs.push_back(c);
        }
        for (auto& c : { std::static_pointer_cast<copra::Constraint>(tk), std::static_pointer_cast<copra::Constraint>(ck),
                 std::static_pointer_cast<copra::Constraint>(mk), std::static_pointer_cast<copra::Constraint>(tb),
                 std::static_pointer_cast<copra::Constraint>(cb) }) {
            c->autoSpan();
            cstrs.push_back(c);
        }
    }
};

static double max_abs_diff(const Eigen::MatrixXd& a, const Eigen::MatrixXd& b, Eigen::Index r0, Eigen::Index c0)
{
    double m = 0.0; // || a - b[r0.., c0..] ||_max
    for (Eigen::Index i = 0; i < a.rows(); ++i)
        for (Eigen::Index j = 0; j < a.cols(); ++j) m = std::max(m, std::fabs(a(i, j) - b(r0 + i, c0 + j)));
    return m;
}

static void initial_state_cases()
{
    using namespace Eigen;
    const int steps = 10;
    MatrixXd combi = MatrixXd::Ones(3, 3), A(2, 2), B(2, 1);
    A << 1, 1, 1, 1;
    B << 1, 1;
    const VectorXd bias = VectorXd::Zero(2), s_init = VectorXd::Zero(2);
    for (int fullSize = 0; fullSize < 2; ++fullSize) {
        { // INITIAL-STATE-OPTIMIZATION (TestLMPC_InitialState.cpp:266-403)
            NineClasses nc(fullSize != 0, steps);
            auto ps = std::make_shared<copra::PreviewSystem>();
            ps->system(A, B, bias, s_init, steps);
            copra::InitialStateLMPC lmpc(ps);
            const VectorXd lo = -1.0 * VectorXd::Ones(2), up = VectorXd::Ones(2);
            lmpc.resetInitialStateBounds(lo, up);
            lmpc.resetInitialStateCost(1e-6 * MatrixXd::Identity(2, 2), VectorXd::Zero(2));
            for (auto& c : nc.costs) lmpc.addCost(c);
            for (auto& c : nc.cstrs) lmpc.addConstraint(c);
            CHECK(lmpc.solve());
            if (lmpc.fail() != 0) {
                std::printf("InitialStateLMPC::solve failed with status %d (fullSize = %d)\n", lmpc.fail(), fullSize);
                continue;
            }
            const VectorXd x0s = lmpc.trajectory().head(2);
            for (int i = 0; i < 2; ++i) {
                CHECK(x0s(i) <= up(i) + 1e-6);
                CHECK(lo(i) <= x0s(i) + 1e-6);
                CHECK(std::fabs(x0s(i) - lmpc.initialState()(i)) < 1e-12);
            }
        }
        { // LMPC_AND_INITIAL-STATE-LMPC_COMPARISON (:29-253): the trailing blocks of the InitialStateLMPC QP are the LMPC QP
            NineClasses a(fullSize != 0, steps), b(fullSize != 0, steps);
            auto ps = std::make_shared<copra::PreviewSystem>();
            ps->system(A, B, bias, s_init, steps);
            copra::LMPC lmpcA(ps);
            for (auto& c : a.costs) lmpcA.addCost(c);
            for (auto& c : a.cstrs) lmpcA.addConstraint(c);
            CHECK(lmpcA.solve());
            copra::InitialStateLMPC lmpcB(ps);
            // (the reference runs this case on QLD with the default R = 0, which no Cholesky-based solver -- QuadProgDense
            //  included -- can factorise, reference quirk Q6; with x0lb == x0ub and a tiny R the dual active-set method
            //  -- the CPU oracle's as well -- reports "no solution" on this badly scaled problem, so R = I here: the
            //  identities checked below do not depend on R)
            lmpcB.resetInitialStateCost(MatrixXd::Identity(2, 2), VectorXd::Zero(2));
            for (auto& c : b.costs) lmpcB.addCost(c);
            for (auto& c : b.cstrs) lmpcB.addConstraint(c);
            CHECK(lmpcB.solve());
            if (lmpcA.fail() != 0 || lmpcB.fail() != 0) {
                std::printf("comparison case: solve failed (LMPC %d, InitialStateLMPC %d, fullSize = %d)\n", lmpcA.fail(), lmpcB.fail(), fullSize);
                continue;
            }
            CHECK(lmpcA.nrEqConstr() == lmpcB.nrEqConstr());
            CHECK(lmpcA.nrIneqConstr() == lmpcB.nrIneqConstr());
            const int n = ps->fullUDim;
            CHECK(max_abs_diff(lmpcA.Q(), lmpcB.Q(), 2, 2) <= 1e-6);
            CHECK(max_abs_diff(lmpcA.Aineq(), lmpcB.Aineq(), 0, 2) <= 1e-6);
            for (int i = 0; i < n; ++i) {
                CHECK(std::fabs(lmpcA.lb()(i) - lmpcB.lb()(2 + i)) <= 1e-6);
                CHECK(std::fabs(lmpcA.ub()(i) - lmpcB.ub()(2 + i)) <= 1e-6);
            }
            // bounds default to ps->x0 on both sides: the initial state cannot move (:236-252)
            for (int i = 0; i < 2; ++i) {
                CHECK(std::fabs(lmpcA.trajectory()(i) - s_init(i)) <= 1e-6);
                CHECK(std::fabs(lmpcB.trajectory()(i) - s_init(i)) <= 1e-6);
            }
        }
    }
}

// ---- plug-in point 2: user-defined subclasses with their own update() (include/constraints.h:42-107, costFunctions.h:22-97)
// The same velocity limit as TrajectoryConstraint(E = [0 1], f = 0) and the same tracking cost as TrajectoryCost(M, xd),
// written by a "user" against ps.Phi / ps.Psi / ps.xi exactly as the reference's classes are (constraints.cpp:66-84,
// costFunctions.cpp:63-82) -- LMPC::solve runs these update() on the host and the device solves with the results.
class UserVelocityLimit : public copra::EqIneqConstraint {
public:
    explicit UserVelocityLimit(double vmax)
        : EqIneqConstraint("User velocity", true)
        , vmax_(vmax)
    {
    }
    void autoSpan() override {}
    void initializeConstraint(const copra::PreviewSystem& ps) override
    {
        nrConstr_ = ps.nrXStep;
        A_.resize(nrConstr_, ps.fullUDim);
        Y_.resize(nrConstr_, ps.xDim);
        b_.resize(nrConstr_);
        z_.resize(nrConstr_);
        ++initialised;
    }
    void update(const copra::PreviewSystem& ps) override
    {
        for (int i = 0; i < ps.nrXStep; ++i) { // row i: velocity of step i  (E = [0 1])
            const int row = i * ps.xDim + 1;
            for (int j = 0; j < ps.fullUDim; ++j) A_(i, j) = ps.Psi(row, j);
            for (int a = 0; a < ps.xDim; ++a) Y_(i, a) = ps.Phi(row, a);
            z_(i) = vmax_ - ps.xi(row);
            double yx = 0.0;
            for (int a = 0; a < ps.xDim; ++a) yx += Y_(i, a) * ps.x0(a);
            b_(i) = z_(i) - yx;
        }
        ++updated;
    }
    int initialised = 0, updated = 0;

private:
    double vmax_;
};
class UserTrackingCost : public copra::CostFunction {
public:
    UserTrackingCost(const Eigen::VectorXd& xd, const Eigen::VectorXd& w)
        : CostFunction("User tracking cost")
        , xd_(xd)
        , w_(w)
    {
    }
    void update(const copra::PreviewSystem& ps) override
    {
        Q_.setZero(), E_.setZero(), f_.setZero();
        const int nx = ps.xDim, n = ps.fullUDim;
        for (int i = 0; i < ps.nrXStep; ++i) // sum over the steps of Psi_i' W Psi_i etc. (M = I)
            for (int k = 0; k < nx; ++k) {
                const int row = i * nx + k;
                for (int j = 0; j < n; ++j) {
                    const double t = w_(k) * ps.Psi(row, j);
                    if (t == 0.0) continue;
                    for (int l = 0; l < n; ++l) Q_(l, j) += ps.Psi(row, l) * t;
                    for (int a = 0; a < nx; ++a) E_(a, j) += ps.Phi(row, a) * t;
                    f_(j) += (ps.xi(row) - xd_(k)) * t;
                }
            }
        for (int j = 0; j < n; ++j) {
            double acc = f_(j);
            for (int a = 0; a < nx; ++a) acc += E_(a, j) * ps.x0(a);
            c_(j) = acc;
        }
    }

private:
    Eigen::VectorXd xd_, w_;
};

// ---- plug-in point 1: a user SolverInterface handed to LMPC::useSolver.  This one runs the CPU oracle's restatement of
// the reference's QuadProgDenseSolver (oracle/copra_oracle.c::or_quadprog_dense -- test infrastructure), so the check is
// "device-condensed QP + CPU QuadProgDense == fused device solve".
extern "C" int or_quadprog_dense(int n, int meq, int mineq, const double* Q, const double* c, const double* Aeq,
    const double* beq, const double* Aineq, const double* bineq, const double* XL, const double* XU, double* x, int* iter);
class OracleQuadProgSolver : public copra::SolverInterface {
public:
    int SI_fail() const override { return fail_; }
    void SI_inform() const override { std::printf("oracle QuadProgDense: fail = %d\n", fail_); }
    int SI_iter() const override { return iter_[0]; }
    const Eigen::VectorXd& SI_result() const override { return x_; }
    void SI_problem(int nrVar, int nrEq, int nrInEq) override
    {
        n_ = nrVar, neq_ = nrEq, nin_ = nrInEq;
        x_.resize(nrVar);
        ++problems;
    }
    bool SI_solve(const Eigen::MatrixXd& Q, const Eigen::VectorXd& c, const Eigen::MatrixXd& Aeq, const Eigen::VectorXd& beq,
        const Eigen::MatrixXd& Aineq, const Eigen::VectorXd& bineq, const Eigen::VectorXd& XL, const Eigen::VectorXd& XU) override
    {
        fail_ = or_quadprog_dense(n_, neq_, nin_, Q.data(), c.data(), Aeq.data(), beq.data(), Aineq.data(), bineq.data(),
            XL.data(), XU.data(), x_.data(), iter_);
        ++solves;
        return fail_ == 0;
    }
    int problems = 0, solves = 0;

private:
    int n_ = 0, neq_ = 0, nin_ = 0, fail_ = 0, iter_[2] = { 0, 0 };
    Eigen::VectorXd x_;
};

static double max_rel(const Eigen::VectorXd& a, const Eigen::VectorXd& b)
{
    double m = 0.0;
    for (Eigen::Index i = 0; i < a.rows(); ++i) m = std::fmax(m, std::fabs(a(i) - b(i)) / (1.0 + std::fabs(b(i))));
    return a.rows() == b.rows() ? m : 1e300;
}

static void plugin_cases(int nbStep)
{
    using namespace Eigen;
    std::vector<std::shared_ptr<copra::CostFunction>> keep, keepc;
    std::vector<std::shared_ptr<copra::Constraint>> keepk;
    IneqSystem s(nbStep);
    VectorXd uLower(1), uUpper(1);
    uLower.setConstant(-std::numeric_limits<double>::infinity());
    uUpper.setConstant(200);
    // reference: the built-in classes only
    auto ps = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, s.x0, s.nbStep);
    copra::LMPC ref(ps);
    auto xCost = std::make_shared<copra::TrajectoryCost>(s.M, s.xd);
    auto uCost = std::make_shared<copra::ControlCost>(s.N, s.ud);
    xCost->weights(s.wx);
    uCost->weights(s.wu);
    VectorXd f0(1);
    f0 << 0;
    auto vCstr = std::make_shared<copra::TrajectoryConstraint>(s.E, f0);
    auto uBound = std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper);
    ref.addCost(xCost), ref.addCost(uCost), ref.addConstraint(vCstr), ref.addConstraint(uBound);
    CHECK(ref.solve());

    { // a user-defined constraint subclass and a user-defined cost subclass ride the fused device solve
        auto ps2 = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, s.x0, s.nbStep);
        copra::LMPC mpc(ps2);
        auto userCost = std::make_shared<UserTrackingCost>(s.xd, s.wx);
        auto uCost2 = std::make_shared<copra::ControlCost>(s.N, s.ud);
        uCost2->weights(s.wu);
        auto userCstr = std::make_shared<UserVelocityLimit>(0.0);
        auto uBound2 = std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper);
        mpc.addCost(userCost), mpc.addCost(uCost2), mpc.addConstraint(userCstr), mpc.addConstraint(uBound2);
        CHECK(userCstr->initialised == 1 && userCstr->nrConstr() == s.nbStep + 1);
        CHECK(mpc.solve());
        CHECK(userCstr->updated == 1 && ps2->previewOnHost);
        CHECK(mpc.nrIneqConstr() == s.nbStep + 1);
        CHECK(max_rel(mpc.control(), ref.control()) <= 1e-6);
        CHECK(max_rel(mpc.trajectory(), ref.trajectory()) <= 1e-6);
        // receding horizon: a new x0 reaches the user pieces through their update() (b = z - Y x0, c = E'x0 + f)
        VectorXd x1(2);
        x1 << 0.01, -4.5;
        ps->xInit(x1), ps2->xInit(x1);
        CHECK(ref.solve() && mpc.solve());
        CHECK(userCstr->updated >= 2);
        CHECK(max_rel(mpc.control(), ref.control()) <= 1e-6);
        ps->xInit(s.x0);
        CHECK(ref.solve());
        // ... and the InitialStateLMPC form reads Y, z, E, f of the same user pieces
        copra::InitialStateLMPC isRef(ps), isUser(ps2);
        ps2->xInit(s.x0);
        auto mk = [&](copra::InitialStateLMPC& c, bool user) {
            MatrixXd R = MatrixXd::Identity(2, 2) * 10.0;
            VectorXd r(2), lo(2), hi(2);
            r << 0.1, -0.2;
            lo << s.x0(0) - 0.05, s.x0(1) - 0.05;
            hi << s.x0(0) + 0.05, s.x0(1) + 0.05;
            c.resetInitialStateCost(R, r);
            c.resetInitialStateBounds(lo, hi);
            auto uc = std::make_shared<copra::ControlCost>(s.N, s.ud);
            uc->weights(s.wu);
            keep.push_back(uc);
            c.addCost(uc);
            if (user) {
                auto a = std::make_shared<UserTrackingCost>(s.xd, s.wx);
                auto b = std::make_shared<UserVelocityLimit>(0.0);
                keepc.push_back(a), keepk.push_back(b);
                c.addCost(a), c.addConstraint(b);
            } else {
                auto a = std::make_shared<copra::TrajectoryCost>(s.M, s.xd);
                a->weights(s.wx);
                auto b = std::make_shared<copra::TrajectoryConstraint>(s.E, f0);
                keepc.push_back(a), keepk.push_back(b);
                c.addCost(a), c.addConstraint(b);
            }
            auto bd = std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper);
            keepk.push_back(bd);
            c.addConstraint(bd);
        };
        mk(isRef, false), mk(isUser, true);
        CHECK(isRef.solve() && isUser.solve());
        CHECK(max_rel(isUser.control(), isRef.control()) <= 1e-6);
        CHECK(max_rel(isUser.initialState(), isRef.initialState()) <= 1e-6);
    }
    { // LMPC::useSolver: a user SolverInterface gets the device-condensed QP (LMPC.cpp:67-70, 88-97)
        copra::LMPC mpc(ps);
        mpc.addCost(xCost), mpc.addCost(uCost), mpc.addConstraint(vCstr);
        auto uBound3 = std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper);
        mpc.addConstraint(uBound3);
        auto user = new OracleQuadProgSolver();
        mpc.useSolver(std::unique_ptr<copra::SolverInterface>(user));
        CHECK(mpc.solve());
        CHECK(user->problems == 1 && user->solves == 1);
        CHECK(max_rel(mpc.control(), ref.control()) <= 1e-6);
        CHECK(max_rel(mpc.trajectory(), ref.trajectory()) <= 1e-6);
        CHECK(mpc.Q().rows() == s.nbStep && mpc.ub()(0) == 200);
        mpc.selectQPSolver(copra::SolverFlag::QuadProgDense); // back to the fused device solve (drops the user solver)
        CHECK(mpc.solve());
        CHECK(max_rel(mpc.control(), ref.control()) <= 1e-6);
        // the same for InitialStateLMPC: result = [x0*; U] (InitialStateLMPC.cpp:124-128)
        copra::InitialStateLMPC isA(ps), isB(ps);
        for (copra::InitialStateLMPC* c : { &isA, &isB }) {
            MatrixXd R = MatrixXd::Identity(2, 2) * 10.0;
            VectorXd r(2), lo(2), hi(2);
            r << 0.1, -0.2;
            lo << s.x0(0) - 0.05, s.x0(1) - 0.05;
            hi << s.x0(0) + 0.05, s.x0(1) + 0.05;
            c->resetInitialStateCost(R, r);
            c->resetInitialStateBounds(lo, hi);
            c->addCost(xCost), c->addCost(uCost), c->addConstraint(vCstr);
        }
        isB.useSolver(std::unique_ptr<copra::SolverInterface>(new OracleQuadProgSolver()));
        CHECK(isA.solve() && isB.solve());
        CHECK(max_rel(isB.control(), isA.control()) <= 1e-6);
        CHECK(max_rel(isB.initialState(), isA.initialState()) <= 1e-6);
        CHECK(max_rel(isB.trajectory(), isA.trajectory()) <= 1e-6);
    }
    { // LMPC::checkDeleteCostsAndConstraints (LMPC.cpp:288-307): a piece the caller released is dropped after the solve
        copra::LMPC mpc(ps);
        auto extra = std::make_shared<copra::TrajectoryConstraint>(s.E, f0);
        auto uBound4 = std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper);
        mpc.addCost(xCost), mpc.addCost(uCost), mpc.addConstraint(extra), mpc.addConstraint(uBound4);
        CHECK(mpc.nrIneqConstr() == s.nbStep + 1);
        extra.reset(); // only the controller refers to it now
        CHECK(mpc.solve()); // this solve still honours it ...
        CHECK(max_rel(mpc.control(), ref.control()) <= 1e-6);
        CHECK(mpc.nrIneqConstr() == 0); // ... and drops it afterwards
        CHECK(mpc.solve()); // the next solve runs without it
    }
    { // the accessors of the built-in classes (costFunctions.h:79-90, constraints.h:93-99), evaluated on the device
        auto ps3 = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, s.x0, s.nbStep);
        copra::LMPC only(ps3);
        auto cst = std::make_shared<copra::TrajectoryCost>(s.M, s.xd);
        cst->weights(s.wx);
        only.addCost(cst);
        cst->update(*ps3);
        CHECK(cst->Q().rows() == s.nbStep && cst->E().rows() == 2 && cst->f().rows() == s.nbStep);
        double dq = 0.0, dc = 0.0;
        for (int j = 0; j < s.nbStep; ++j) {
            dc = std::fmax(dc, std::fabs(only.c()(j) - cst->c()(j)));
            for (int i = 0; i < s.nbStep; ++i)
                dq = std::fmax(dq, std::fabs(only.Q()(i, j) - cst->Q()(i, j) - (i == j ? 1e-6 : 0.0)));
        }
        CHECK(dq <= 1e-9 && dc <= 1e-9);
        UserTrackingCost mine(s.xd, s.wx); // the user restatement above agrees with the device's evaluation
        mine.initializeCost(*ps3);
        ps3->updateSystem();
        mine.update(*ps3);
        double de = 0.0;
        for (int j = 0; j < s.nbStep; ++j)
            for (int a = 0; a < 2; ++a) de = std::fmax(de, std::fabs(mine.E()(a, j) - cst->E()(a, j)) / (1.0 + std::fabs(cst->E()(a, j))));
        CHECK(de <= 1e-9);
        auto lim = std::make_shared<copra::TrajectoryConstraint>(s.E, f0);
        lim->initializeConstraint(*ps3);
        lim->update(*ps3);
        UserVelocityLimit ul(0.0);
        ul.initializeConstraint(*ps3);
        ul.update(*ps3);
        CHECK(lim->A().rows() == s.nbStep + 1 && lim->Y().cols() == 2);
        double da = 0.0;
        for (int i = 0; i <= s.nbStep; ++i) {
            da = std::fmax(da, std::fabs(lim->b()(i) - ul.b()(i)) + std::fabs(lim->z()(i) - ul.z()(i)));
            for (int j = 0; j < s.nbStep; ++j) da = std::fmax(da, std::fabs(lim->A()(i, j) - ul.A()(i, j)));
        }
        CHECK(da <= 1e-12);
    }
}

// What a drop-in user sees for ONE problem: the headline controller (CoM preview, nx = 6, nu = 3, N = 20, TrajectoryCost +
// ControlCost, 63 velocity rows + control bounds; binding/python/tests/pyTests.py:342-359 constants) built through the mirror's
// classes and solved with copra::LMPC::solve() -- one launch, one synchronisation, two result copies per call -- `reps` times
// with a new measured state each time (ps->xInit, PreviewSystem.h:52).  Prints wall microseconds per solve() (median, mean) and
// the mirror's own solveTime() / solveAndBuildTime() of the last call: the figure to put next to the CPU path's.
#include <algorithm>
#include <chrono>
static void latency_case(int reps, bool hard)
{
    using namespace Eigen;
    const int nbStep = 20;
    const double T = 0.117;
    MatrixXd A = MatrixXd::Identity(6, 6), B = MatrixXd::Zero(6, 3);
    for (int i = 0; i < 3; ++i) {
        A(i, 3 + i) = T;
        B(i, i) = 0.5 * T * T;
        B(3 + i, i) = T;
    }
    VectorXd d = VectorXd::Zero(6), x0(6), goal(6), wx(6), wu(3), lo(6), up(6), ulo(3), uup(3);
    x0 << 1.5842778860957882, 0.3422260214935311, 2.289067474385933, 0.0, 0.0, 0.0; // x_init, x_goal of pyTests.py:358-359: the
    goal << 1.627772868473883, 0.4156386515475985, 2.3984423755527136, 0.06745225960685897, 0.3882830795737303, 0.06845759848745198; // benchmark's instances
    wx << 10, 10, 10, 1, 1, 1;
    wu << 1e-3, 1e-3, 1e-3;
    const double inf = std::numeric_limits<double>::infinity();
    lo << -inf, -inf, -inf, -inf, -inf, -inf;
    up << inf, inf, inf, 0.6, 0.6, 0.6;
    ulo << -3, -3, -3;
    uup << 3, 3, 3;
    if (hard) { // a start far from the goal: velocity and control bounds active over much of the horizon -- more active constraints than the
                // first launch's five register columns, so every solve also goes through the second launch
        x0 << 0.2, 0.1, 0.8, 0.05, -0.1, 0.0;
        goal << 1.0, 0.6, 0.8, 0.0, 0.0, 0.0;
    }
    auto ps = std::make_shared<copra::PreviewSystem>();
    ps->system(A, B, d, x0, nbStep);
    copra::LMPC controller(ps);
    auto xCost = std::make_shared<copra::TrajectoryCost>(MatrixXd::Identity(6, 6), goal);
    xCost->weights(wx);
    auto uCost = std::make_shared<copra::ControlCost>(MatrixXd::Identity(3, 3), VectorXd::Zero(3));
    uCost->weights(wu);
    auto xBound = std::make_shared<copra::TrajectoryBoundConstraint>(lo, up);
    auto uBound = std::make_shared<copra::ControlBoundConstraint>(ulo, uup);
    controller.addCost(xCost);
    controller.addCost(uCost);
    controller.addConstraint(xBound);
    controller.addConstraint(uBound);
    for (int i = 0; i < 20; ++i) CHECK(controller.solve()); // (first calls: plan, handle, module load)
    std::vector<double> us;
    VectorXd x = x0;
    for (int i = 0; i < reps; ++i) {
        x(0) = x0(0) + 0.002 * (i % 17), x(4) = 0.01 * (i % 5); // (a new measured state every call)
        ps->xInit(x);
        const auto t0 = std::chrono::steady_clock::now();
        const bool ok = controller.solve();
        const auto t1 = std::chrono::steady_clock::now();
        CHECK(ok);
        us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(us.begin(), us.end());
    double mean = 0.0;
    for (double v : us) mean += v;
    mean /= (double)us.size();
    std::printf("latency_us median %.2f mean %.2f min %.2f p95 %.2f solveTime_us %.2f solveAndBuildTime_us %.2f control0 %.9f iter %d\n",
        us[us.size() / 2], mean, us.front(), us[(size_t)(0.95 * us.size())], controller.solveTime() * 1e6,
        controller.solveAndBuildTime() * 1e6, controller.control()(0), controller.iter());
}

// A tracking controller's tick through the reference's API: the reference trajectory moves, and the only way to tell the controller is a
// NEW TrajectoryCost (M, N, p are constructor arguments, costFunctions.h:103-131) in place of the old one.  The reference evaluates every
// cost anew in every solve anyway; the mirror recognises a cost list that differs from the handle's in p alone and sends p
// (copra_batch_set_cost_reference) instead of building a new handle.  Checked against a controller built from scratch with the same
// costs at every tick; a weights() call on a cost that is already in the controller must be seen as well (LMPC.cpp:233-247).
static double max_abs_diff(const Eigen::VectorXd& a, const Eigen::VectorXd& b)
{
    double m = a.rows() == b.rows() ? 0.0 : 1e300;
    for (Eigen::Index i = 0; i < a.rows() && i < b.rows(); ++i) m = std::max(m, std::fabs(a(i) - b(i)));
    return m;
}
// Reference quirk Q2 as an opt-in (LMPC::referenceAccumulation): the per-step TrajectoryCost / MixedCost members are only zeroed in
// initializeCost and ADDED to in every update (src/costFunctions.cpp:52-55, 73-80 / 184-187, 205-213), so the k-th solve() on one
// controller sees k x Q, E, f of the cost -- for a TrajectoryCost exactly what a FRESH controller with k x the weights sees (c = E'x0 + f
// is assigned), for a MixedCost with c_k = c_{k-1} + (E_k' x0_k + f_k) on top.  Checked here: three solves with a new x0 each, against
// (a) a fresh controller with scaled weights (TrajectoryCost) and (b) the dense QP assembled from the accessors of fresh controllers and
// solved by QuadProgDenseSolver (MixedCost); with the switch off every solve equals a fresh controller's first one.
static void accumulation_case()
{
    IneqSystem s(10);
    using namespace Eigen;
    const double x0s[3][2] = { { 0.0, -5.0 }, { -0.02, -4.6 }, { -0.05, -4.1 } };
    VectorXd uLower(1), uUpper(1);
    uLower << -std::numeric_limits<double>::infinity();
    uUpper << 200.0;
    auto fresh = [&](double scale, const double* x0v, std::shared_ptr<copra::TrajectoryCost>* keep = nullptr) {
        auto ps = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, s.x0, s.nbStep);
        VectorXd x0(2);
        x0 << x0v[0], x0v[1];
        ps->xInit(x0);
        auto lm = std::make_shared<copra::LMPC>(ps);
        auto xc = std::make_shared<copra::TrajectoryCost>(s.M, s.xd);
        auto uc = std::make_shared<copra::ControlCost>(s.N, s.ud);
        auto ub = std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper);
        xc->weights(s.wx * scale);
        uc->weights(s.wu);
        lm->addCost(xc), lm->addCost(uc), lm->addConstraint(ub);
        CHECK(lm->solve());
        if (keep) *keep = xc;
        return lm->control();
    };
    for (int mode = 0; mode < 2; ++mode) { // accumulation off / on
        auto ps = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, s.x0, s.nbStep);
        copra::LMPC lm(ps);
        lm.referenceAccumulation(mode == 1);
        auto xc = std::make_shared<copra::TrajectoryCost>(s.M, s.xd);
        auto uc = std::make_shared<copra::ControlCost>(s.N, s.ud);
        auto ub = std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper);
        xc->weights(s.wx);
        uc->weights(s.wu);
        lm.addCost(xc), lm.addCost(uc), lm.addConstraint(ub);
        for (int k = 1; k <= 3; ++k) {
            VectorXd x0(2);
            x0 << x0s[k - 1][0], x0s[k - 1][1];
            ps->xInit(x0);
            CHECK(lm.solve());
            const VectorXd want = fresh(mode == 1 ? (double)k : 1.0, x0s[k - 1]);
            const double d = max_abs_diff(lm.control(), want);
            if (!(d <= 1e-7)) std::printf("accumulation %d solve %d: |u - u_expected| = %.3e\n", mode, k, d);
            CHECK(d <= 1e-7 * 200.0);
            if (mode == 1 && k == 3) { // ... and the accumulated members are 3 x one evaluation's (costFunctions.h:79-90 accessors)
                std::shared_ptr<copra::TrajectoryCost> one;
                (void)fresh(1.0, x0s[k - 1], &one);
                one->update(*ps);
                CHECK(std::fabs(xc->Q()(0, 0) - 3.0 * one->Q()(0, 0)) <= 1e-9 * std::fabs(one->Q()(0, 0)) && one->Q()(0, 0) != 0.0);
                CHECK(std::fabs(xc->f()(1) - 3.0 * one->f()(1)) <= 1e-9 * std::fabs(one->f()(1)) + 1e-12);
            }
        }
    }
    // MixedCost: c accumulates as well.  Expected QP of solve k from fresh costs' accessors: Q = 1e-6 I + k Q1 + Qu, c = sum_j j c1(x0_j) + cu
    {
        auto ps = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, s.x0, s.nbStep);
        copra::LMPC lm(ps);
        lm.referenceAccumulation(true);
        MatrixXd Mm(1, 2), Nm(1, 1);
        Mm << 0.0, 1.0; // v_k + 0.001 u_k -> -1
        Nm << 0.001;
        VectorXd pm(1), wm(1);
        pm << -1.0;
        wm << 50.0;
        auto mc = std::make_shared<copra::MixedCost>(Mm, Nm, pm);
        auto uc = std::make_shared<copra::ControlCost>(s.N, s.ud);
        auto ub = std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper);
        mc->weights(wm);
        uc->weights(s.wu);
        lm.addCost(mc), lm.addCost(uc), lm.addConstraint(ub);
        const int n = s.nbStep;
        VectorXd cacc = VectorXd::Zero(n);
        for (int k = 1; k <= 3; ++k) {
            VectorXd x0(2);
            x0 << x0s[k - 1][0], x0s[k - 1][1];
            ps->xInit(x0);
            CHECK(lm.solve());
            // one evaluation of the two costs at this x0, by fresh objects
            auto ps1 = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, x0, s.nbStep);
            copra::MixedCost m1(Mm, Nm, pm);
            copra::ControlCost u1(s.N, s.ud);
            m1.weights(wm), u1.weights(s.wu);
            m1.initializeCost(*ps1), u1.initializeCost(*ps1);
            m1.update(*ps1), u1.update(*ps1);
            MatrixXd Q(n, n);
            VectorXd c(n), XL(n), XU(n);
            for (int j = 0; j < n; ++j) {
                cacc(j) += k * m1.c()(j); // c_k = c_{k-1} + (E_k' x0 + f_k),  E_k = k E1, f_k = k f1
                c(j) = cacc(j) + u1.c()(j);
                XL(j) = -std::numeric_limits<double>::max(), XU(j) = 200.0;
                for (int i = 0; i < n; ++i) Q(i, j) = k * m1.Q()(i, j) + u1.Q()(i, j) + (i == j ? 1e-6 : 0.0);
            }
            copra::QuadProgDenseSolver qp;
            qp.SI_problem(n, 0, 0);
            CHECK(qp.SI_solve(Q, c, MatrixXd(0, n), VectorXd(0), MatrixXd(0, n), VectorXd(0), XL, XU));
            const double d = max_abs_diff(lm.control(), qp.SI_result());
            if (!(d <= 1e-6)) std::printf("mixed accumulation solve %d: |u - u_expected| = %.3e\n", k, d);
            CHECK(d <= 1e-6);
        }
    }
}

static void tracking_case(int reps)
{
    using namespace Eigen;
    const int N = 20;
    const double T = 0.117, inf = std::numeric_limits<double>::infinity();
    MatrixXd A = MatrixXd::Identity(6, 6), B = MatrixXd::Zero(6, 3);
    for (int i = 0; i < 3; ++i) A(i, 3 + i) = T, B(i, i) = 0.5 * T * T, B(3 + i, i) = T;
    VectorXd d = VectorXd::Zero(6), x0(6), goal(6), wx(6), wu(3), lo(6), up(6), ulo(3), uup(3);
    x0 << 1.5842778860957882, 0.3422260214935311, 2.289067474385933, 0.0, 0.0, 0.0;
    goal << 1.627772868473883, 0.4156386515475985, 2.3984423755527136, 0.06745225960685897, 0.3882830795737303, 0.06845759848745198;
    wx << 10, 10, 10, 1, 1, 1;
    wu << 1e-3, 1e-3, 1e-3;
    lo << -inf, -inf, -inf, -inf, -inf, -inf;
    up << inf, inf, inf, 0.6, 0.6, 0.6;
    ulo << -3, -3, -3;
    uup << 3, 3, 3;
    MatrixXd Mfull = MatrixXd::Identity(6 * (N + 1), 6 * (N + 1));
    auto reference = [&](int tick) { // a straight line towards the goal that moves on with the ticks
        VectorXd p(6 * (N + 1));
        for (int k = 0; k <= N; ++k) {
            const double s = std::min(1.0, (k + 0.05 * tick) / (double)N);
            for (int i = 0; i < 6; ++i) p(6 * k + i) = x0(i) + s * (goal(i) - x0(i));
        }
        return p;
    };
    auto make = [&](copra::LMPC& c, const std::shared_ptr<copra::PreviewSystem>& ps, const VectorXd& p, const VectorXd& w) {
        auto xc = std::make_shared<copra::TrajectoryCost>(Mfull, p);
        xc->weights(w);
        auto uc = std::make_shared<copra::ControlCost>(MatrixXd::Identity(3, 3), VectorXd::Zero(3));
        uc->weights(wu);
        auto xb = std::make_shared<copra::TrajectoryBoundConstraint>(lo, up);
        auto ub = std::make_shared<copra::ControlBoundConstraint>(ulo, uup);
        c.addCost(xc), c.addCost(uc), c.addConstraint(xb), c.addConstraint(ub);
        return std::make_tuple(xc, uc, xb, ub);
    };
    auto ps = std::make_shared<copra::PreviewSystem>();
    ps->system(A, B, d, x0, N);
    copra::LMPC controller(ps);
    auto held = make(controller, ps, reference(0), wx);
    auto xCost = std::get<0>(held);
    for (int i = 0; i < 10; ++i) CHECK(controller.solve());
    int builds0 = 0; // (after the first tick: removeCost + addCost moves the cost to the end of the list -- once a new order, then the same)
    std::vector<double> us;
    VectorXd x = x0;
    double worst = 0.0;
    for (int i = 1; i <= reps; ++i) {
        x(0) = x0(0) + 0.002 * (i % 17), x(4) = 0.01 * (i % 5);
        ps->xInit(x);
        const VectorXd p = reference(i);
        const auto t0 = std::chrono::steady_clock::now();
        auto next = std::make_shared<copra::TrajectoryCost>(Mfull, p);
        next->weights(wx);
        controller.removeCost(xCost);
        controller.addCost(next);
        xCost = next;
        const bool ok = controller.solve();
        const auto t1 = std::chrono::steady_clock::now();
        CHECK(ok);
        if (i == 1) builds0 = controller.handleBuilds();
        if (i > 1) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        if (i % 50 == 1) { // against a controller built from scratch with the same costs
            auto ps2 = std::make_shared<copra::PreviewSystem>();
            ps2->system(A, B, d, x, N);
            copra::LMPC fresh(ps2);
            auto h2 = make(fresh, ps2, p, wx);
            CHECK(fresh.solve());
            worst = std::max(worst, max_abs_diff(fresh.control(), controller.control()));
            CHECK(fresh.iter() == controller.iter());
        }
    }
    CHECK(worst <= 1e-10);
    CHECK(controller.handleBuilds() == builds0); // (not one new handle for all these ticks)
    // weights changed on a cost that is in the controller: seen by the next solve (a new plan)
    VectorXd w2 = wx;
    w2(0) = 3.0, w2(4) = 2.5;
    xCost->weights(w2);
    CHECK(controller.solve());
    {
        auto ps2 = std::make_shared<copra::PreviewSystem>();
        ps2->system(A, B, d, x, N);
        copra::LMPC fresh(ps2);
        auto h2 = make(fresh, ps2, reference(reps), w2);
        CHECK(fresh.solve());
        CHECK(max_abs_diff(fresh.control(), controller.control()) <= 1e-10);
        CHECK(controller.handleBuilds() == builds0 + 1);
    }
    std::sort(us.begin(), us.end());
    double mean = 0.0;
    for (double v : us) mean += v;
    mean /= (double)us.size();
    std::printf("tracking_tick_us median %.2f mean %.2f min %.2f p95 %.2f solveTime_us %.2f builds %d worst %.3e\n", us[us.size() / 2], mean,
        us.front(), us[(size_t)(0.95 * us.size())], controller.solveTime() * 1e6, controller.handleBuilds(), worst);
}

int main(int argc, char** argv)
{
    std::setvbuf(stdout, nullptr, _IONBF, 0);
    const char* mode = argc > 1 ? argv[1] : "errors";
    try {
        if (!std::strcmp(mode, "errors")) error_handlers();
        if (!std::strcmp(mode, "solve")) solve_cases(argc > 2 ? std::atoi(argv[2]) : 300);
        if (!std::strcmp(mode, "initial_state")) initial_state_cases();
        if (!std::strcmp(mode, "plugins")) plugin_cases(argc > 2 ? std::atoi(argv[2]) : 12);
        if (!std::strcmp(mode, "accumulation")) accumulation_case();
        if (!std::strcmp(mode, "tracking")) {
            copra::LMPC::newHandlePerCostChange() = argc > 3 && !std::strcmp(argv[3], "newhandle"); // (measurements: a new handle per swapped cost)
            tracking_case(argc > 2 ? std::atoi(argv[2]) : 300);
        }
        if (!std::strcmp(mode, "latency")) latency_case(argc > 2 ? std::atoi(argv[2]) : 500, argc > 3 && !std::strcmp(argv[3], "hard"));
    } catch (const std::exception& e) {
        std::printf("uncaught exception: %s\n", e.what());
        return 2;
    }
    std::printf("%s: %d failure(s)\n", mode, failures);
    return failures ? 1 : 0;
}
