"""Host-side mirror of g2o's non-linear drivers over the device-resident solver.

`levenberg_solve_iteration` restates OptimizationAlgorithmLevenberg::solve
(/root/reference/g2o/core/optimization_algorithm_levenberg.cpp:57-146) and
`gauss_newton_iteration` OptimizationAlgorithmGaussNewton::solve
(/root/reference/g2o/core/optimization_algorithm_gauss_newton.cpp:50-93), written against a small
"graph" protocol so the same loop drives the GPU solver (estimates, errors and Jacobians stay in
HBM) and, in the tests, the CPU oracle:

    graph.compute_active_errors()   SparseOptimizer::computeActiveErrors   sparse_optimizer.cpp:61-76
    graph.linearize()               errors + Jacobians (what buildSystem's per-edge linearizeOplus needs)
    graph.chi2()                    activeRobustChi2                        sparse_optimizer.cpp:100-114
    graph.update()                  update(solver.x())                      sparse_optimizer.cpp:421-434
    graph.push() / pop() / discard_top()                                    sparse_optimizer.cpp:599-650
    solver.buildSystem/setLambda/solve/restoreDiagonal/maxDiagonal/computeScale   (Solver interface)
"""
import math

OK, TERMINATE, FAIL = 1, 0, -1          # OptimizationAlgorithm::SolverResult (optimization_algorithm.h:49)
DBL_MAX = 1.7976931348623157e308


class LevenbergState:
    """The members of OptimizationAlgorithmLevenberg (optimization_algorithm_levenberg.h:83-96)."""

    def __init__(self, tau=1e-5, max_trials=10, user_lambda_init=0.0):
        self.tau = tau
        self.good_step_upper = 2.0 / 3.0
        self.good_step_lower = 1.0 / 3.0
        self.max_trials = max_trials
        self.user_lambda_init = user_lambda_init
        self.current_lambda = -1.0
        self.ni = 2.0
        self.levenberg_iterations = 0


def levenberg_solve_iteration(graph, solver, st, iteration):
    graph.linearize()                                   # computeActiveErrors + per-edge linearizeOplus
    current_chi = graph.chi2()
    temp_chi = current_chi
    solver.buildSystem()
    if iteration == 0:                                  # computeLambdaInit :149-163
        st.current_lambda = st.user_lambda_init if st.user_lambda_init > 0 else st.tau * solver.maxDiagonal()
        st.ni = 2.0
    rho = 0.0
    qmax = 0
    # device-resident graph + HIP solver: the trial's three host round trips (solve status, chi2, computeScale) collapse
    # into one (g2ohip_solve_async / g2ohip_trial_stats); same values, same decisions
    fused = getattr(graph, "device_resident", False) and hasattr(solver, "solveAsync") and hasattr(solver, "trialStats")
    while True:
        graph.push()
        solver.setLambda(st.current_lambda, True)
        if fused:
            solver.solveAsync()
            graph.update()
            solver.restoreDiagonal()
            graph.compute_active_errors()
            ok2, temp_chi, sc = solver.trialStats(st.current_lambda)
            if ok2 is None:      # the asynchronous solve has to be repeated (g2ohip_trial_stats status 2): redo the trial in
                graph.pop()      # the synchronous form, which retries by itself -- same values, same decisions
                graph.push()
                solver.setLambda(st.current_lambda, True)
                ok2 = solver.solve()
                graph.update()
                solver.restoreDiagonal()
                graph.compute_active_errors()
                temp_chi = graph.chi2()
                sc = solver.computeScale(st.current_lambda) if ok2 else 0.0
        else:
            ok2 = solver.solve()
            graph.update()
            solver.restoreDiagonal()
            graph.compute_active_errors()
            temp_chi = graph.chi2()
            sc = solver.computeScale(st.current_lambda) if ok2 else 0.0
        if not ok2:
            temp_chi = DBL_MAX
        rho = current_chi - temp_chi
        scale = (sc if ok2 else 0.0) + 1e-3             # computeScale :165-172
        rho /= scale
        if rho > 0 and math.isfinite(temp_chi):         # last step was good
            alpha = 1.0 - (2.0 * rho - 1.0) ** 3
            alpha = min(alpha, st.good_step_upper)
            st.current_lambda *= max(st.good_step_lower, alpha)
            st.ni = 2.0
            current_chi = temp_chi
            graph.discard_top()
        else:
            st.current_lambda *= st.ni
            st.ni *= 2.0
            graph.pop()
        qmax += 1
        if not (rho < 0 and qmax < st.max_trials):
            break
    st.levenberg_iterations = qmax
    result = TERMINATE if (qmax == st.max_trials or rho == 0) else OK
    return result, current_chi


def gauss_newton_iteration(graph, solver):
    graph.linearize()
    chi = graph.chi2()
    solver.buildSystem()
    if not solver.solve():
        return FAIL, chi
    graph.update()
    return OK, chi


class DoglegState:
    """The members of OptimizationAlgorithmDogleg (optimization_algorithm_dogleg.cpp:40-50, .h:83-100)."""

    def __init__(self, initial_delta=1e4, max_trials_after_failure=100, initial_lambda=1e-7, lambda_factor=10.0):
        self.delta = initial_delta
        self.max_trials = max_trials_after_failure
        self.current_lambda = initial_lambda
        self.lambda_factor = lambda_factor
        self.was_pd = True
        self.last_step = "Undefined"
        self.last_num_tries = 0


def dogleg_solve_iteration(graph, solver, st):
    """OptimizationAlgorithmDogleg::solve (optimization_algorithm_dogleg.cpp:57-207).  Needs solver.b(), x(), setX(),
    multiplyHessian(v) (dest += H v from a zero dest, block_solver.h:142) besides the LM protocol."""
    import numpy as np
    graph.linearize()                                   # computeActiveErrors + the Jacobians buildSystem needs
    current_chi = graph.chi2()
    solver.buildSystem()
    b = solver.b()
    aux = solver.multiplyHessian(b)
    alpha = float(b @ b) / float(aux @ b)
    hsd = alpha * b
    hsd_norm = float(np.linalg.norm(hsd))
    hgn = None
    hgn_norm = -1.0
    good = False
    tries = 0
    while True:
        tries += 1
        if hgn is None:
            min_lambda, max_lambda = 1e-12, 1e3
            ok = False
            while not ok:                               # damp only once the matrix was seen not positive definite
                if not st.was_pd:
                    solver.setLambda(st.current_lambda, True)
                ok = solver.solve()
                if not st.was_pd:
                    solver.restoreDiagonal()
                st.was_pd = st.was_pd and ok
                if not st.was_pd:
                    if ok:
                        st.current_lambda = max(min_lambda, st.current_lambda / (0.5 * st.lambda_factor))
                    else:
                        st.current_lambda *= st.lambda_factor
                        if st.current_lambda > max_lambda:
                            st.current_lambda = max_lambda
                            return FAIL, current_chi
            hgn = solver.x()
            hgn_norm = float(np.linalg.norm(hgn))
        if hgn_norm < st.delta:
            hdl = hgn
            st.last_step = "GN"
        elif hsd_norm > st.delta:
            hdl = st.delta / hsd_norm * hsd
            st.last_step = "Descent"
        else:
            d = hgn - hsd
            c = float(hsd @ d)
            bma = float(d @ d)
            if c <= 0.0:
                beta = (-c + math.sqrt(c * c + bma * (st.delta * st.delta - float(hsd @ hsd)))) / bma
            else:
                hs2 = float(hsd @ hsd)
                beta = (st.delta * st.delta - hs2) / (c + math.sqrt(c * c + bma * (st.delta * st.delta - hs2)))
            hdl = hsd + beta * d
            st.last_step = "Dogleg"
        aux = solver.multiplyHessian(hdl)
        linear_gain = -1.0 * float(aux @ hdl) + 2.0 * float(b @ hdl)
        graph.push()
        solver.setX(hdl)
        graph.update()
        graph.compute_active_errors()
        new_chi = graph.chi2()
        nonlinear_gain = current_chi - new_chi
        if abs(linear_gain) < 1e-12:
            linear_gain = 1e-12
        rho = nonlinear_gain / linear_gain
        if rho > 0:
            graph.discard_top()
            good = True
        else:
            graph.pop()
        if rho > 0.75:
            st.delta = max(st.delta, 3.0 * float(np.linalg.norm(hdl)))
        elif rho < 0.25:
            st.delta *= 0.5
        if good or tries >= st.max_trials:
            break
    st.last_num_tries = tries
    if tries == st.max_trials or not good:
        return TERMINATE, current_chi
    return OK, current_chi


BATCH_STAT_FIELDS = ("iteration", "numVertices", "numEdges", "chi2", "timeResiduals", "timeLinearize", "timeQuadraticForm",
                     "timeSchurComplement", "timeSymbolicDecomposition", "timeNumericDecomposition", "timeLinearSolution",
                     "iterationsLinearSolver", "timeUpdate", "timeIteration", "levenbergIterations", "timeLinearSolver",
                     "hessianDimension", "hessianPoseDimension", "hessianLandmarkDimension", "choleskyNNZ", "timeMarginals")


def format_batch_stats(st):
    """One line per iteration in the format of `g2o -stats <file>`: operator<<(ostream&, const G2OBatchStatistics&),
    /root/reference/g2o/core/batch_stats.cpp:49-82 (PTHING(s) = "s= value\t ", same field order)."""
    def fmt(v):
        return "%d" % v if isinstance(v, int) else "%g" % v
    return "".join("%s= %s\t " % (k, fmt(st.get(k, 0))) for k in BATCH_STAT_FIELDS)


def optimize(graph, solver, iterations, algorithm="lm", stats=None, num_vertices=0, num_edges=0, times=None, **lm_args):
    """SparseOptimizer::optimize (sparse_optimizer.cpp:354-419): returns (#iterations done, chi2 per
    iteration before its step, lambda per iteration, LM trials per iteration).
    stats: a list that receives one G2OBatchStatistics-like dict per iteration (sparse_optimizer.cpp:379-399; the solver
    must have profiling on for the per-stage times): write them with format_batch_stats for a `-stats` compatible file.
    times: a list that receives the wall-clock seconds of every iteration (each ends with a read-back, so they are exact)."""
    import time
    st = DoglegState(**lm_args) if algorithm == "dogleg" else LevenbergState(**lm_args)
    chis, lams, trials = [], [], []
    done = 0
    for it in range(iterations):
        ts = time.perf_counter()
        if algorithm == "dogleg":
            res, _ = dogleg_solve_iteration(graph, solver, st)
            lams.append(st.delta)
            trials.append(st.last_num_tries)
        elif algorithm == "lm":
            res, _ = levenberg_solve_iteration(graph, solver, st, it)
            lams.append(st.current_lambda)
            trials.append(st.levenberg_iterations)
        else:
            res, _ = gauss_newton_iteration(graph, solver)
        graph.compute_active_errors()
        chis.append(graph.chi2())
        done += 1
        if times is not None:
            times.append(time.perf_counter() - ts)
        if stats is not None:
            d = dict(solver.stats()) if hasattr(solver, "stats") else {}
            d.update(iteration=it, numVertices=int(num_vertices), numEdges=int(num_edges), chi2=chis[-1],
                     timeIteration=time.perf_counter() - ts, levenbergIterations=int(trials[-1]) if algorithm == "lm" else 0)
            for k in ("hessianDimension", "hessianPoseDimension", "hessianLandmarkDimension", "choleskyNNZ", "iterationsLinearSolver"):
                d[k] = int(d.get(k, 0))
            stats.append(d)
        if res != OK:
            break
    return done, chis, lams, trials


class DeviceBAGraph:
    """The graph protocol over HipBlockSolver's device-resident BA front end."""

    device_resident = True

    def __init__(self, solver):
        self.s = solver

    def linearize(self):
        self.s.baLinearize(True)

    def compute_active_errors(self):
        self.s.baLinearize(False)

    def chi2(self):
        return self.s.chi2()

    def update(self):
        self.s.baUpdate()

    def push(self):
        self.s.baPush()

    def pop(self):
        self.s.baPop()

    def discard_top(self):
        self.s.baDiscardTop()


def setup_device_ba(prob, huber_delta=0.0, device=0, options=None):
    """Build a HipBlockSolver for an openslam_g2o_amd.synthetic BA problem with the estimates, errors
    and Jacobians produced on the device.  Returns (solver, graph).  options: g2ohip_set_option pairs that have to be
    in place before buildStructure (ordering / kernel selection knobs)."""
    import numpy as np
    from . import capi
    s = capi.HipBlockSolver(6, 3, device)
    for name, value in (options or {}).items():
        s.setOption(name, value)
    k = s.addEdgeSet(2, prob["v0"], prob["v1"])
    s.buildStructure(prob["nP"], prob["nL"], True)
    s.baSetEdges(k, prob["cam_idx"], prob["pt_idx"], prob["meas"], None, prob["f"], prob["cx"], prob["cy"])
    s.baSetEstimates(prob["cams"], prob["cam_hidx"], prob["pts"], np.arange(prob["L"], dtype=np.int32))
    if huber_delta > 0:
        s.setRobustKernel(k, capi.KERNEL_HUBER, huber_delta)
    return s, DeviceBAGraph(s)


class DevicePoseGraph:
    """The graph protocol over HipBlockSolver's device-resident pose-graph front end (EdgeSE2 / EdgeSE3)."""

    device_resident = True

    def __init__(self, solver):
        self.s = solver

    def linearize(self):
        self.s.pgLinearize(True)

    def compute_active_errors(self):
        self.s.pgLinearize(False)

    def chi2(self):
        return self.s.chi2()

    def update(self):
        self.s.pgUpdate()

    def push(self):
        self.s.pgPush()

    def pop(self):
        self.s.pgPop()

    def discard_top(self):
        self.s.pgDiscardTop()


def setup_device_pose_graph(edge_type, estimates, hidx, num_free, vi, vj, meas, info, landmark_dim=None, device=0):
    """HipBlockSolver for a pose graph with estimates, errors and Jacobians on the device.
    edge_type 1: EdgeSE2 (estimates / measurements (x, y, theta), information [n][9]);
    edge_type 2: EdgeSE3 (isometries [12] = R column-major | t, information [n][36]).
    hidx[v] = hessian index of vertex v or -1 (fixed); BlockSolver_3_2 / BlockSolver_6_3 semantics, no Schur."""
    import numpy as np
    from . import capi
    d = 3 if edge_type == 1 else 6
    s = capi.HipBlockSolver(d, landmark_dim or (2 if edge_type == 1 else 3), device)
    hidx = np.asarray(hidx, np.int32)
    k = s.addEdgeSet(d, hidx[np.asarray(vi)], hidx[np.asarray(vj)])
    s.buildStructure(num_free, 0, False)
    s.pgSetEdges(k, edge_type, vi, vj, meas, info)
    s.pgSetEstimates(estimates, hidx)
    return s, DevicePoseGraph(s)
