// C++ host-mirror test: reads like the reference's own unit tests (neumann.rs:558-649, sparse.rs:905-963).
// Built with g++ against libsublinear_hip.so and run by tests/test_gpu_parity.py::test_cpp_host_mirror.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "sublinear_solver.hpp"

using namespace sublinear;

#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); std::exit(1); } } while (0)

int main()
{
    // sparse.rs:923-933
    auto m = SparseMatrix::from_triplets({{0, 0, 2.0}, {0, 1, 1.0}, {1, 0, 1.0}, {1, 1, 3.0}}, 2, 2);
    std::vector<double> y(2);
    m.multiply_vector({1.0, 2.0}, y);
    EXPECT(y[0] == 4.0 && y[1] == 7.0);
    EXPECT(m.nnz() == 4 && m.is_diagonally_dominant());

    // neumann.rs:572-590 (test_neumann_solver_simple): diagonally dominant 2x2
    auto a = SparseMatrix::from_triplets({{0, 0, 4.0}, {0, 1, 1.0}, {1, 0, 1.0}, {1, 1, 3.0}}, 2, 2, true);
    auto r = NeumannSolver(20, 1e-8).solve(a, {5.0, 4.0}, SolverOptions());
    EXPECT(r.converged && r.iterations == 16);
    EXPECT(std::fabs(r.solution[0] - 1.0) < 1e-7 && std::fabs(r.solution[1] - 1.0) < 1e-7);
    auto q = NeumannSolver(20, 1e-8).with_reference_quirks(true).solve(a, {5.0, 4.0});
    EXPECT(std::fabs(q.solution[0] - 2.25) < 1e-7);          // reference default start: x_true + D^-1 b

    // error mapping: Result::Err(SolverError::...) -> thrown SolverError{kind}
    auto nd = SparseMatrix::from_triplets({{0, 0, 1.0}, {0, 1, 5.0}, {1, 0, 2.0}, {1, 1, 1.0}}, 2, 2);
    try { NeumannSolver().solve(nd, {1.0, 1.0}); EXPECT(false); } catch (const SolverError &e) { EXPECT(e.kind == SL_NOT_DIAGONALLY_DOMINANT && e.is_recoverable()); }
    try { SparseMatrix::from_triplets({{2, 0, 1.0}}, 2, 2); EXPECT(false); } catch (const SolverError &e) { EXPECT(e.kind == SL_INDEX_OUT_OF_BOUNDS); }
    try { NeumannSolver().solve(a, {1.0}); EXPECT(false); } catch (const SolverError &e) { EXPECT(e.kind == SL_DIMENSION_MISMATCH); }
    SolverOptions tight; tight.tolerance = 1e-14; tight.max_iterations = 3;
    try { NeumannSolver(50, 1e-16).solve(a, {5.0, 4.0}, tight); EXPECT(false); } catch (const SolverError &e) { EXPECT(e.kind == SL_CONVERGENCE_FAILURE); }

    // push + single-entry query
    auto p = PushSolver(1e-10).solve(a, {5.0, 4.0});
    EXPECT(p.converged && std::fabs(p.solution[0] - 1.0) < 1e-8 && std::fabs(p.solution[1] - 1.0) < 1e-8);
    double err = 0;
    const double e1 = PushSolver(1e-12).query_single_entry(a, {5.0, 4.0}, 1, &err);
    EXPECT(std::fabs(e1 - 1.0) < 1e-9);
    // the ForwardPushSolver object: setup once, query after query, same answers as the one-shot call and as the solve
    {
        QuerySession session(a, {5.0, 4.0});
        for (int rep = 0; rep < 3; ++rep) {
            size_t touched = 0;
            EXPECT(std::fabs(session.query_single_entry(0, 1e-12, 100000, &err, &touched) - 1.0) < 1e-9 && touched > 0);
            EXPECT(session.query_single_entry(1, 1e-12) == e1);
        }
    }
    // trait SolverAlgorithm on a device state: initialize / step / is_converged / extract_solution / update_rhs / reset
    {
        NeumannSolver ns(60, 1e-14);
        SolverOptions opt; opt.tolerance = 1e-12;
        NeumannState st = ns.initialize(a, {5.0, 4.0}, opt);
        EXPECT(!ns.is_converged(st) && std::isinf(st.residual_norm()));
        EXPECT(ns.step(st) == StepResult::Converged && ns.is_converged(st));
        auto x = ns.extract_solution(st);
        EXPECT(std::fabs(x[0] - 1.0) < 1e-10 && std::fabs(x[1] - 1.0) < 1e-10);
        ns.update_rhs(st, {{0, 4.0}});                           // b = (9, 4): rhs[0] += 4 / 4, solution[0] += the same (neumann.rs:448-453)
        auto x1 = ns.extract_solution(st);
        EXPECT(x1[0] == x[0] + 1.0 && x1[1] == x[1] && !ns.is_converged(st));
        st.reset();                                              // SolverState::reset, then the loop again: the solve of the updated system
        EXPECT(ns.step(st) == StepResult::Converged);
        auto x2 = ns.extract_solution(st);                       // [[4,1],[1,3]] x = (9, 4): x = (23/11, 7/11)
        EXPECT(std::fabs(x2[0] - 23.0 / 11.0) < 1e-10 && std::fabs(x2[1] - 7.0 / 11.0) < 1e-10);
        try { ns.update_rhs(st, {{5, 1.0}}); EXPECT(false); } catch (const SolverError &e) { EXPECT(e.kind == SL_INDEX_OUT_OF_BOUNDS); }
        // a communicator of one: the partitioned state is the same state
        Communicator comm(0, 1, "cpp_mirror_test");
        NeumannState sp = ns.initialize(a, {5.0, 4.0}, opt, &comm);
        EXPECT(ns.step(sp) == StepResult::Converged);
        auto xp = ns.extract_solution(sp);
        EXPECT(xp[0] == x[0] && xp[1] == x[1]);
    }
    // TS solveForwardPush in its own order (solver.ts:437-522): first maximum first
    {
        std::vector<uint32_t> log;
        auto g = GaussSouthwellSolver(1e-12, 10000).solve(a, {5.0, 4.0}, &log);
        EXPECT(g.converged && g.iterations == log.size() && log.size() > 2 && log[0] == 0);      // |5| > |4|: row 0 first
        EXPECT(std::fabs(g.solution[0] - 1.0) < 1e-10 && std::fabs(g.solution[1] - 1.0) < 1e-10);
        try { GaussSouthwellSolver(1e-12, 2).solve(a, {5.0, 4.0}); EXPECT(false); } catch (const SolverError &e) { EXPECT(e.kind == SL_CONVERGENCE_FAILURE); }
    }
    // CG on a symmetric positive definite system (optimized_solver.rs tests: 2x2 SPD)
    auto spd = SparseMatrix::from_triplets({{0, 0, 4.0}, {0, 1, 1.0}, {1, 0, 1.0}, {1, 1, 3.0}}, 2, 2);
    auto c = ConjugateGradientSolver(100, 1e-10).solve(spd, {1.0, 2.0});
    EXPECT(c.converged && std::fabs(c.solution[0] - 1.0 / 11.0) < 1e-9 && std::fabs(c.solution[1] - 7.0 / 11.0) < 1e-9);
    std::printf("cpp host mirror ok\n");
    return 0;
}
