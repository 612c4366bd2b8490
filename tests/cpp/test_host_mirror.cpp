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
    // Matrix::multiply_vector_add (matrix/mod.rs:441-465): result += A x; dominance factor min(2/1, 3/1) and Gershgorin max(3, 4) (:487-514, :83-100)
    std::vector<double> acc = {0.5, -1.0};
    m.multiply_vector_add({1.0, 2.0}, acc);
    EXPECT(acc[0] == 4.5 && acc[1] == 6.0);
    EXPECT(m.diagonal_dominance_factor().has_value() && *m.diagonal_dominance_factor() == 2.0 && m.spectral_radius_estimate() == 4.0);
    EXPECT(!SparseMatrix::from_triplets({{0, 0, 2.0}, {1, 1, 3.0}}, 2, 2).diagonal_dominance_factor().has_value());     // no off-diagonal weight: None

    {   // get / row_iter / col_iter / frobenius_norm / sparsity_info; KAT sparse.rs:910-920 (test_csr_creation)
        auto c = SparseMatrix::from_triplets({{0, 0, 1.0}, {0, 2, 2.0}, {1, 1, 3.0}, {2, 0, 4.0}, {2, 2, 5.0}}, 3, 3);
        EXPECT(c.nnz() == 5 && *c.get(0, 0) == 1.0 && *c.get(0, 2) == 2.0 && *c.get(1, 1) == 3.0 && !c.get(0, 1).has_value() && !c.get(3, 0).has_value());
        auto row2 = c.row_iter(2), col0 = c.col_iter(0);
        EXPECT(row2.size() == 2 && row2[0].first == 0 && row2[0].second == 4.0 && row2[1].first == 2 && row2[1].second == 5.0 && c.row_iter(9).empty());
        EXPECT(col0.size() == 2 && col0[0].first == 0 && col0[0].second == 1.0 && col0[1].first == 2 && col0[1].second == 4.0);
        EXPECT(c.frobenius_norm() == std::sqrt(55.0) && std::string(c.format_name()) == "CSR");
        c.convert_to_format("COO");                                                    // matrix/mod.rs:244-296: the name changes, the products' bits do not
        EXPECT(std::string(c.format_name()) == "COO" && *c.get(2, 2) == 5.0);
        bool bad_format = false;
        try { c.convert_to_format("ELL"); } catch (const SolverError &e) { bad_format = e.kind == SL_UNSUPPORTED_FORMAT; }
        EXPECT(bad_format);
        auto si = c.sparsity_info();
        EXPECT(si.nnz == 5 && si.rows == 3 && si.cols == 3 && si.max_nnz_per_row == 2 && si.bandwidth == 2 && !si.is_banded && si.sparsity_ratio == 5.0 / 9.0);
    }
    {   // the `&mut self` methods: scale / add_diagonal (matrix/mod.rs:346-372 over sparse.rs:229-248) — in place on the device
        auto c = SparseMatrix::from_triplets({{0, 0, 1.0}, {0, 2, 2.0}, {1, 0, 3.0}, {2, 0, 4.0}, {2, 2, 5.0}}, 3, 3);      // row 1 stores no diagonal
        c.scale(-2.0);
        EXPECT(*c.get(0, 0) == -2.0 && *c.get(0, 2) == -4.0 && *c.get(1, 0) == -6.0 && *c.get(2, 2) == -10.0 && c.nnz() == 5);
        c.add_diagonal(0.5);
        auto tr = c.to_triplets();                                                       // matrix/mod.rs:298-305, from the row slices (no raw copy kept)
        EXPECT(tr.size() == 5 && std::get<0>(tr[2]) == 1 && std::get<1>(tr[2]) == 0 && std::get<2>(tr[2]) == -6.0 && std::get<2>(tr[4]) == -9.5);
        EXPECT(*c.get(0, 0) == -1.5 && *c.get(2, 2) == -9.5 && !c.get(1, 1).has_value() && *c.get(1, 0) == -6.0);         // row 1 silently skipped
        std::vector<double> y(3);
        c.multiply_vector({1.0, 1.0, 1.0}, y);
        EXPECT(y[0] == -5.5 && y[1] == -6.0 && y[2] == -17.5);
        bool threw = false;
        try { SparseMatrix::from_triplets({{0, 0, 1.0}, {1, 2, 1.0}}, 2, 3).add_diagonal(1.0); } catch (const SolverError &e) { threw = e.kind == SL_INVALID_INPUT; }
        EXPECT(threw);                                                                                                     // matrix/mod.rs:356-361
    }
    {   // solver::utils (solver/mod.rs:363-461), SolverOptions::streaming (:101-116), meets_quality_criteria (:192-195)
        const std::vector<double> v = {3.0, -4.0, 0.5};
        EXPECT(utils::l1_norm(v) == 7.5 && utils::linf_norm(v) == 4.0 && utils::l2_norm(v) == std::sqrt(25.25));
        EXPECT(utils::compute_norm(v, utils::NormType::L1) == 7.5 && utils::compute_norm(v, utils::NormType::LInfinity) == 4.0
               && utils::compute_norm(v, utils::NormType::Weighted) == utils::l2_norm(v));
        auto a = SparseMatrix::from_triplets({{0, 0, 2.0}, {0, 1, 1.0}, {1, 0, 1.0}, {1, 1, 3.0}}, 2, 2);
        std::vector<double> r;
        utils::compute_residual(a, {1.0, 2.0}, {1.0, 1.0}, r);
        EXPECT(r.size() == 2 && r[0] == 3.0 && r[1] == 6.0);
        const std::vector<double> cur = {1.0, 2.0}, prev = {1.0, 2.5};
        EXPECT(utils::check_convergence(1e-7, 1e-6, utils::ConvergenceMode::ResidualNorm, 0.0, nullptr, cur));
        EXPECT(!utils::check_convergence(1e-3, 1e-6, utils::ConvergenceMode::RelativeResidual, 10.0, nullptr, cur));
        EXPECT(!utils::check_convergence(0.0, 1.0, utils::ConvergenceMode::SolutionChange, 1.0, nullptr, cur));            // no previous solution
        EXPECT(utils::check_convergence(0.0, 0.5, utils::ConvergenceMode::SolutionChange, 1.0, &prev, cur) && !utils::check_convergence(0.0, 0.49, utils::ConvergenceMode::SolutionChange, 1.0, &prev, cur));
        EXPECT(utils::check_convergence(0.0, 0.2, utils::ConvergenceMode::RelativeSolutionChange, 1.0, &prev, cur) && !utils::check_convergence(0.0, 0.18, utils::ConvergenceMode::RelativeSolutionChange, 1.0, &prev, cur));
        EXPECT(utils::check_convergence(1e-7, 1e-6, utils::ConvergenceMode::Combined, 0.0, nullptr, cur) && !utils::check_convergence(1e-7, 1e-6, utils::ConvergenceMode::Combined, 1e-3, nullptr, cur));
        auto so = SolverOptions::streaming(10);
        EXPECT(so.tolerance == 1e-4 && so.max_iterations == 1000 && so.collect_stats && !so.compute_error_bounds && so.streaming_interval == 10);
        SolverResult sr; sr.converged = true; sr.residual_norm = 1e-5;
        EXPECT(sr.meets_quality_criteria(1e-4) && !sr.meets_quality_criteria(1e-6));
    }
    {   // from_dense / identity / diagonal (matrix/mod.rs:204-239)
        auto dn = SparseMatrix::from_dense({2.0, 0.0, 1.0, 3.0}, 2, 2);
        EXPECT(dn.nnz() == 3 && *dn.get(1, 0) == 1.0 && !dn.get(0, 1).has_value());
        auto id = SparseMatrix::identity(3);
        auto dg = SparseMatrix::diagonal({2.0, 0.0, -4.0});
        EXPECT(id.nnz() == 3 && *id.get(2, 2) == 1.0 && dg.nnz() == 2 && *dg.get(2, 2) == -4.0 && !dg.get(1, 1).has_value());
        bool threw = false;
        try { SparseMatrix::from_dense({1.0, 2.0, 3.0}, 2, 2); } catch (const SolverError &) { threw = true; }
        EXPECT(threw);
    }
    {   // simd_ops free functions (lib.rs:83-87) with the reference's own unit-test values: simd_ops.rs:259-286
        std::vector<double> yy(2);
        matrix_vector_multiply_simd({2.0, 1.0, 1.0, 3.0}, {0, 1, 0, 1}, {0, 2, 4}, {1.0, 2.0}, yy);
        EXPECT(yy[0] == 4.0 && yy[1] == 7.0);
        parallel_matrix_vector_multiply({2.0, 1.0, 1.0, 3.0}, {0, 1, 0, 1}, {0, 2, 4}, {1.0, 2.0}, yy);
        EXPECT(yy[0] == 4.0 && yy[1] == 7.0);
        EXPECT(dot_product_simd({1.0, 2.0, 3.0, 4.0}, {5.0, 6.0, 7.0, 8.0}) == 70.0);
        std::vector<double> ay = {1.0, 1.0, 1.0, 1.0};
        axpy_simd(2.0, {1.0, 2.0, 3.0, 4.0}, ay);
        EXPECT(ay[0] == 3.0 && ay[1] == 5.0 && ay[2] == 7.0 && ay[3] == 9.0);
    }
    // neumann.rs:572-590 (test_neumann_solver_simple): diagonally dominant 2x2
    auto a = SparseMatrix::from_triplets({{0, 0, 4.0}, {0, 1, 1.0}, {1, 0, 1.0}, {1, 1, 3.0}}, 2, 2, true);
    auto r = NeumannSolver(20, 1e-8).solve(a, {5.0, 4.0}, SolverOptions());
    EXPECT(r.converged && r.iterations == 16);
    {   // estimate_error_bounds (neumann.rs:321-347): the series ends the solve => Some(bound); one term => Some(0.0)
        SolverOptions eb; eb.tolerance = 1e-30; eb.compute_error_bounds = true;
        auto e = NeumannSolver(20, 1e-8).solve(a, {5.0, 4.0}, eb);
        EXPECT(e.iterations == 17 && e.error_bound.has_value() && std::fabs(*e.error_bound - 1.724974623182487e-09) < 1e-20);
        auto one = NeumannSolver(20, 1e9).solve(a, {5.0, 4.0}, eb);
        EXPECT(one.iterations == 1 && one.error_bound.has_value() && *one.error_bound == 0.0);
    }
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
        const auto batch = session.query_batch({1, 0, 1}, 1e-12, 100000, 2);                     // lanes: same bits as one at a time
        EXPECT(batch.size() == 3 && batch[0] == e1 && batch[2] == e1 && std::fabs(batch[1] - 1.0) < 1e-9);
    }
    // the graph side under the reference's names: PushGraph + ForwardPushSolver / BackwardPushSolver in the spec's own visiting order
    // (tests/rust/push_tests.rs: the 4-node fixture :15-22, 200 pushes at alpha = 0.15 / epsilon = 1e-6 — SURVEY 8(c) G5)
    {
        PushGraph g(4, {0, 2, 4, 6, 7}, {1, 2, 0, 3, 0, 3, 1}, {0.5, 0.5, 0.8, 0.2, 0.6, 0.4, 1.0});
        EXPECT(g.num_nodes() == 4 && g.num_edges() == 7 && g.out_degree(0) == 1.0 && g.out_degree(9) == 0.0);
        ForwardPushSolver fps(g);
        const ForwardPushResult r = fps.solve_single_source(0);
        EXPECT(r.push_count == 200 && r.nodes_visited == 4);
        const double want[4] = {0.431272, 0.276168, 0.183291, 0.109267};
        double mass = 0.0;
        for (int i = 0; i < 4; ++i) { EXPECT(std::fabs(r.estimate[i] - want[i]) < 1e-6 && r.residual[i] >= 0.0); mass += r.estimate[i] + 0.15 * r.residual[i]; }
        EXPECT(std::fabs(mass - 1.0) < 0.01);                                                    // push_tests.rs:107-129
        EXPECT(fps.query_single_entry(0, 2) == r.estimate[2] && fps.query_single_entry(0, 99) == 0.0);
        EXPECT(fps.solve_single_source(10).push_count == 0);                                     // out-of-range source: empty result (:433-495)
        const ForwardPushResult t = fps.solve_with_target(0, 3, 0.05);
        EXPECT(t.push_count > 0 && t.push_count < r.push_count && t.estimate[3] > 0.05);         // forward_push.rs:262-265
        const ForwardPushResult m2 = fps.solve_multi_source({0, 3});
        double mm = 0.0;
        for (int i = 0; i < 4; ++i) mm += m2.estimate[i] + 0.15 * m2.residual[i];
        EXPECT(std::fabs(mm - 1.0) < 0.01);
        BackwardPushSolver bps(g);
        EXPECT(bps.solve_single_target(1).push_count > 0 && bps.query_transition_probability(0, 1) > 0.0);
        // solve_with_source / reachability_probabilities / extrapolated_solution (backward_push.rs:238-311): the oracle's run of the fixture
        // from target 3 takes 305 pushes; stopping once node 1 holds half of its final estimate takes 12
        const BackwardPushResult bfull = bps.solve_single_target(3);
        const BackwardPushResult bsrc = bps.solve_with_source(1, 3, 0.5 * bfull.estimate[1]);
        EXPECT(bfull.push_count == 305 && bsrc.push_count == 12 && bsrc.estimate[1] > 0.5 * bfull.estimate[1]);
        EXPECT(bps.solve_with_source(9, 3, 0.1).push_count == 0 && bps.solve_with_source(1, 9, 0.1).push_count == 0);   // :243-251
        const std::vector<double> reach = bps.reachability_probabilities(3), extra = bps.extrapolated_solution(bfull);
        for (int i = 0; i < 4; ++i) EXPECT(reach[i] == extra[i] && reach[i] == bfull.estimate[i] + 0.15 * bfull.residual[i]);
        EXPECT(bps.solve_multi_target({3, 1}).push_count > 0);
        {   // combine_with_forward / BidirectionalPushSolver (backward_push.rs:314-410): the CPU checker's value for source 0, target 3 of the
            // fixture (200 forward + 305 backward pushes, the reference's order of additions); no degree dominates there: adaptive = bidirectional
            const ForwardPushResult f0 = fps.solve_single_source(0);
            EXPECT(f0.push_count == 200 && bps.combine_with_forward(bfull, f0.estimate, f0.residual) == 0.13230768829166403);
            BidirectionalPushSolver bi(g);
            EXPECT(bi.solve_bidirectional(0, 3) == 0.13230768829166403 && bi.adaptive_solve(0, 3) == 0.13230768829166403 && bi.adaptive_solve(0, 9) == 0.0);
        }
        auto e = PushGraph::from_edges(5, {{0, 1, 1.0}, {1, 2, 1.0}, {2, 3, 1.0}, {7, 1, 1.0}});     // the invalid edge is skipped
        EXPECT(e.num_edges() == 3 && ForwardPushSolver(e).solve_single_source(0).estimate[4] == 0.0);
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
