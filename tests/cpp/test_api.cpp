// Tests of the C++ API mirror (copra_amd/cpp/include/copra/copra.h), written like the reference's own doctest cases
// (tests/TestLMPC.cpp) on the reference's fixtures (tests/systems.h).  Usage: test_api errors | solve
//   errors : TestLMPC.cpp:949-1087 (ERROR_HANDLER_*) -- host-side checks only, runs without a GPU
//   solve  : TestLMPC.cpp:36-97, 415-479, 593-670 property checks on the GPU (horizon 12 to fit the one-wave kernel)
#include <copra/copra.h>

#include <cmath>
#include <cstdio>
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

// tests/systems.h:94-137 (IneqSystem) with a shorter horizon
struct IneqSystem {
    IneqSystem()
        : T(0.005), mass(5), nbStep(12), A(2, 2), B(2, 1), G(1, 1), E(1, 2), M(2, 2), N(1, 1), c(2), h(1), p(1), x0(2), xd(2), ud(1), wx(2), wu(1)
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

static void solve_cases()
{
    IneqSystem s;
    const double inf = std::numeric_limits<double>::infinity();
    { // MPC_TARGET_COST_WITH_BOUND_CONSTRAINTS (TestLMPC.cpp:36-97)
        Eigen::VectorXd uLower(1), uUpper(1), xLower(2), xUpper(2);
        uLower.setConstant(-inf);
        uUpper.setConstant(200);
        xLower.setConstant(-inf);
        xUpper << inf, 0;
        auto ps = std::make_shared<copra::PreviewSystem>();
        ps->system(s.A, s.B, s.c, s.x0, s.nbStep);
        auto controller = copra::LMPC(ps);
        auto xCost = std::make_shared<copra::TargetCost>(s.M, s.xd);
        auto uCost = std::make_shared<copra::ControlCost>(s.N, s.ud);
        auto trajConstr = std::make_shared<copra::TrajectoryBoundConstraint>(xLower, xUpper);
        auto contConstr = std::make_shared<copra::ControlBoundConstraint>(uLower, uUpper);
        xCost->weights(s.wx);
        uCost->weights(s.wu);
        controller.addCost(xCost);
        controller.addCost(uCost);
        controller.addConstraint(trajConstr);
        controller.addConstraint(contConstr);
        CHECK(controller.solve());
        Eigen::VectorXd fullTraj = controller.trajectory();
        Eigen::VectorXd control = controller.control();
        double posMax = -inf, velMax = -inf;
        for (Eigen::Index i = 0; i < fullTraj.rows() / 2; ++i) {
            posMax = std::max(posMax, fullTraj(2 * i));
            velMax = std::max(velMax, fullTraj(2 * i + 1));
        }
        CHECK(posMax <= s.x0(0));
        CHECK(velMax <= 0 + 1e-6);
        CHECK(control.maxCoeff() <= 200 + 1e-6);
        CHECK(controller.solveTime() > 0 && controller.solveAndBuildTime() >= controller.solveTime());
        // receding horizon: xInit without re-creating anything (PreviewSystem.h:52)
        Eigen::VectorXd x1(2);
        x1 << fullTraj(2), fullTraj(3);
        ps->xInit(x1);
        CHECK(controller.solve());
        CHECK(std::fabs(controller.trajectory()(0) - x1(0)) < 1e-12);
    }
    { // MPC_TARGET_COST_WITH_MIXED_CONSTRAINTS (TestLMPC.cpp:415-479), p = 200
        Eigen::VectorXd p(1);
        p << 200;
        auto ps = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, s.x0, s.nbStep);
        auto controller = copra::LMPC(ps);
        auto xCost = std::make_shared<copra::TargetCost>(s.M, s.xd);
        auto uCost = std::make_shared<copra::ControlCost>(s.N, s.ud);
        auto mixedConstr = std::make_shared<copra::MixedConstraint>(s.E, s.G, p);
        xCost->weights(s.wx);
        uCost->weights(s.wu);
        controller.addCost(xCost);
        controller.addCost(uCost);
        controller.addConstraint(mixedConstr);
        CHECK(controller.solve());
        Eigen::VectorXd fullTraj = controller.trajectory(), control = controller.control();
        for (int i = 0; i < s.nbStep; ++i) CHECK(fullTraj(2 * i + 1) + control(i) <= 200 + 1e-6);
    }
    { // MPC_TARGET_COST_WITH_EQUALITY_CONSTRAINTS (TestLMPC.cpp:593-670) -> u_k = m g
        Eigen::MatrixXd E = Eigen::MatrixXd::Zero(2, 2);
        E(0, 0) = 1;
        Eigen::VectorXd x0 = Eigen::VectorXd::Zero(2), xd = Eigen::VectorXd::Zero(2);
        auto ps = std::make_shared<copra::PreviewSystem>(s.A, s.B, s.c, x0, s.nbStep);
        auto controller = copra::LMPC(ps);
        auto xCost = std::make_shared<copra::TargetCost>(s.M, xd);
        auto uCost = std::make_shared<copra::ControlCost>(s.N, s.ud);
        auto trajConstr = std::make_shared<copra::TrajectoryConstraint>(E, x0, false);
        xCost->weights(s.wx);
        uCost->weights(s.wu);
        controller.addCost(xCost);
        controller.addCost(uCost);
        controller.addConstraint(trajConstr);
        CHECK(controller.solve());
        for (int i = 0; i < s.nbStep; ++i) CHECK(std::fabs(controller.control()(i) - 49.05) < 1e-5);
        CHECK(controller.nrEqConstr() == 2 * (s.nbStep + 1));
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

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "errors";
    try {
        if (!std::strcmp(mode, "errors")) error_handlers();
        if (!std::strcmp(mode, "solve")) solve_cases();
    } catch (const std::exception& e) {
        std::printf("uncaught exception: %s\n", e.what());
        return 2;
    }
    std::printf("%s: %d failure(s)\n", mode, failures);
    return failures ? 1 : 0;
}
