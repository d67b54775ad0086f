// Source-level drop-in check: this translation unit uses EXACTLY the include lines and class names code written against copra
// uses -- the include block of the reference's tests/TestLMPC.cpp:5-9 and tests/TestSolvers.cpp:5, <Eigen/Core> as at
// TestLMPC.cpp:25, copra::QuadProgDenseSolver as at TestSolvers.cpp:27 -- and nothing of this repository's own naming
// (no <copra/copra.h>, no Hip* class).  Built by tests/test_cpp_api.py with
//     g++ -I copra_amd/cpp/include [-I copra_amd/cpp/include/copra/eigen_shim  when the image has no Eigen3]
// Modes:   compile-time only (no arguments: host-side object construction, runs without a GPU)
//          "solvers"  -- TestSolvers.cpp:25-33 (QuadProgTest) on the device, plus the Scilab known answer of tests/systems.h:11-38
//          "lmpc"     -- the first case of TestLMPC.cpp (target cost + both bound constraints, its acceptance checks at :78-83)
#include "LMPC.h"
#include "PreviewSystem.h"
#include "QuadProgSolver.h"
#include "constraints.h"
#include "costFunctions.h"

#include "AutoSpan.h"
#include "InitialStateLMPC.h"
#include "SolverInterface.h"
#include "api.h"
#include "debugUtils.h"
#include "solverUtils.h"
#include "typedefs.h"

#include <Eigen/Core>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <numeric>
#include <vector>

static int failures = 0;
#define REQUIRE(cond)                                                                  \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            std::printf("REQUIRE failed %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
            ++failures;                                                                \
        }                                                                              \
    } while (0)

static_assert(copra::is_all_arithmetic<int, double, float>::value, "typedefs.h: all arithmetic");
static_assert(!copra::is_all_arithmetic<int, Eigen::VectorXd>::value, "typedefs.h: a vector is not arithmetic");

// a user-side declaration decorated the way copra's own headers are (api.h)
struct COPRA_DLLAPI UserTag {
    int v = 0;
};

// the QP of tests/systems.h:9-38 (the Scilab qld documentation example the reference's solver tests run on)
struct Problem {
    Problem()
        : nrvars(6), nreqs(3), nrineqs(2), Q(6, 6), Aeq(3, 6), Aineq(2, 6), c(6), beq(3), bineq(2), XL(6), XU(6)
    {
        Q.setIdentity();
        c << 1, 2, 3, 4, 5, 6;
        Aeq << 1, -1, 1, 0, 3, 1, -1, 0, -3, -4, 5, 6, 2, 5, 3, 0, 1, 0;
        beq << 1, 2, 3;
        Aineq << 0, 1, 0, 1, 2, -1, -1, 0, 2, 1, 1, 0;
        bineq << -1, 2.5;
        XL << -1000, -10000, 0, -1000, -1000, -1000;
        XU << 10000, 100, 1.5, 100, 100, 1000;
    }
    int nrvars, nreqs, nrineqs;
    Eigen::MatrixXd Q, Aeq, Aineq;
    Eigen::VectorXd c, beq, bineq, XL, XU;
};

static void solver_case()
{
    Problem pb;
    copra::QuadProgDenseSolver qpQuadProg; // TestSolvers.cpp:27
    qpQuadProg.SI_problem(pb.nrvars, pb.nreqs, pb.nrineqs);
    REQUIRE(qpQuadProg.SI_solve(pb.Q, pb.c, pb.Aeq, pb.beq, pb.Aineq, pb.bineq, pb.XL, pb.XU));
    REQUIRE(qpQuadProg.SI_fail() == 0);
    // the published answer of the Scilab example (known answer 1)
    const double xs[6] = { 1.7975426035, -0.3381487238, 0.1633880281, -4.9884022703, 0.6054943277, -3.1155623387 };
    const Eigen::VectorXd& x = qpQuadProg.SI_result();
    for (int i = 0; i < 6; ++i) REQUIRE(std::fabs(x(i) - xs[i]) <= 1e-9);
    // the factory hands out the same class under both flags (src/solverUtils.cpp:9-34)
    std::unique_ptr<copra::SolverInterface> viaFactory = copra::solverFactory(copra::SolverFlag::QuadProgDense);
    viaFactory->SI_problem(pb.nrvars, pb.nreqs, pb.nrineqs);
    REQUIRE(viaFactory->SI_solve(pb.Q, pb.c, pb.Aeq, pb.beq, pb.Aineq, pb.bineq, pb.XL, pb.XU));
    REQUIRE(std::fabs(viaFactory->SI_result()(3) - xs[3]) <= 1e-9);
}

// the falling mass of tests/systems.h:42-83 with both bounds (BoundedSystem), the reference's horizon of 300 steps
struct BoundedSystem {
    BoundedSystem()
        : T(0.005), mass(5), nbStep(300), A(2, 2), B(2, 1), M(2, 2), N(1, 1), c(2), x0(2), xd(2), ud(1), wx(2), wu(1), uLower(1), uUpper(1), xLower(2), xUpper(2)
    {
        A << 1, T, 0, 1;
        B << 0.5 * T * T / mass, T / mass;
        c << (-9.81 / 2.) * T * T, -9.81 * T;
        x0 << 0, -5;
        wx << 10, 10000;
        wu << 1e-4;
        M.setIdentity();
        N.setIdentity();
        xd << 0, -1;
        ud << 2;
        uLower.setConstant(-std::numeric_limits<double>::infinity());
        uUpper.setConstant(200);
        xLower.setConstant(-std::numeric_limits<double>::infinity());
        xUpper << std::numeric_limits<double>::infinity(), 0;
    }
    double T, mass;
    int nbStep;
    Eigen::MatrixXd A, B, M, N;
    Eigen::VectorXd c, x0, xd, ud, wx, wu, uLower, uUpper, xLower, xUpper;
};

// (the controller co-owns its pieces and DROPS the ones the user has released after a solve -- src/LMPC.cpp:288-307 --, so the
//  caller keeps them, as the reference's test bodies do)
struct Pieces {
    std::shared_ptr<copra::PreviewSystem> ps;
    std::shared_ptr<copra::TargetCost> xCost;
    std::shared_ptr<copra::ControlCost> uCost;
    std::shared_ptr<copra::TrajectoryBoundConstraint> trajConstr;
    std::shared_ptr<copra::ControlBoundConstraint> contConstr;
};

static Pieces build_controller(BoundedSystem& s, copra::LMPC& controller)
{
    auto ps = std::make_shared<copra::PreviewSystem>();
    ps->system(s.A, s.B, s.c, s.x0, s.nbStep);
    controller.initializeController(ps);
    auto xCost = std::make_shared<copra::TargetCost>(s.M, s.xd);
    auto uCost = std::make_shared<copra::ControlCost>(s.N, s.ud);
    auto trajConstr = std::make_shared<copra::TrajectoryBoundConstraint>(s.xLower, s.xUpper);
    auto contConstr = std::make_shared<copra::ControlBoundConstraint>(s.uLower, s.uUpper);
    xCost->weights(s.wx);
    uCost->weights(s.wu);
    controller.addCost(xCost);
    controller.addCost(uCost);
    controller.addConstraint(trajConstr);
    controller.addConstraint(contConstr);
    return Pieces { ps, xCost, uCost, trajConstr, contConstr };
}

static void lmpc_case()
{
    BoundedSystem s;
    copra::LMPC controller;
    Pieces keep = build_controller(s, controller);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0)
            controller.selectQPSolver(copra::SolverFlag::QuadProgDense);
        else
            controller.useSolver(std::unique_ptr<copra::SolverInterface>(new copra::QuadProgDenseSolver()));
        REQUIRE(controller.solve());
        Eigen::VectorXd fullTraj = controller.trajectory();
        Eigen::VectorXd control = controller.control();
        REQUIRE(fullTraj.rows() == 2 * (s.nbStep + 1) && control.rows() == s.nbStep);
        // the acceptance checks of the reference's case: terminal velocity reached, bounds held with QuadProg's own slack
        REQUIRE(std::fabs(s.xd(1) - fullTraj(fullTraj.rows() - 1)) <= 0.001);
        REQUIRE(control.maxCoeff() <= s.uUpper(0) + 1e-6);
        double vmax = -1e300;
        for (Eigen::Index i = 1; i < fullTraj.rows(); i += 2) vmax = std::max(vmax, fullTraj(i));
        REQUIRE(vmax <= s.xUpper(1) + 1e-6);
        REQUIRE(controller.solveTime() > 0 && controller.solveAndBuildTime() >= controller.solveTime());
    }
}

static void host_only_case()
{
    BoundedSystem s;
    copra::LMPC controller;
    Pieces keep = build_controller(s, controller);
    auto ps = keep.ps;
    REQUIRE(ps->fullXDim == 2 * (s.nbStep + 1) && ps->fullUDim == s.nbStep);
    copra::InitialStateLMPC isController(ps);
    (void)isController;
    Eigen::MatrixXd spanMe = Eigen::MatrixXd::Identity(2, 2);
    copra::AutoSpan::spanMatrix(spanMe, 6);
    REQUIRE(spanMe.rows() == 6 && spanMe.cols() == 6);
    bool threw = false;
    try {
        DOMAIN_ERROR_EXCEPTION("a user-side check");
    } catch (const std::domain_error& e) {
        threw = std::strstr(e.what(), "In file") != nullptr;
    }
    REQUIRE(threw);
    UserTag tag;
    (void)tag;
}

int main(int argc, char** argv)
{
    try {
        if (argc < 2)
            host_only_case();
        else if (!std::strcmp(argv[1], "solvers"))
            solver_case();
        else if (!std::strcmp(argv[1], "lmpc"))
            lmpc_case();
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    std::printf("%s: %d failure(s)\n", argc > 1 ? argv[1] : "host", failures);
    return failures ? 1 : 0;
}
