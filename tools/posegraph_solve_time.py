#!/usr/bin/env python
"""Wall-clock of solve() on the two pose-graph fixtures (manhattan 3500 / sphere 2200) for a list of analysis options:
python tools/posegraph_solve_time.py max_sn_scalars=24 max_sn_scalars=48"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openslam_g2o_amd import capi
from oracle import oracle as O
from tests.helpers import manhattan_golden, sphere_golden

def run(name, opts):
    if name == "sphere2500":      # create_sphere defaults (config 2 at its stated size), generated
        from openslam_g2o_amd import synthetic as S
        g = S.make_sphere(); p, l, d = 6, 3, 6
        J0, J1, err = O.se3_edges(g["poses"], g["vi"], g["vj"], g["Z"])
    elif name == "sphere":
        g = sphere_golden(); p, l, d = 6, 3, 6
        J0, J1, err = O.se3_edges(g["poses"], g["vi"], g["vj"], g["Z"])
    else:
        g = manhattan_golden(); p, l, d = 3, 2, 3
        J0, J1, err = O.se2_edges(g["estimates"], g["vi"], g["vj"], g["meas"])
    s = capi.HipBlockSolver(p, l, 0)
    for kv in opts:
        k, v = kv.split("=")
        if k != "use_graph":
            s.setOption(k, float(v))
    k = s.addEdgeSet(d, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
    s.buildStructure(g["nP"], 0, False)
    s.setEdgeData(k, J0, J1, g["omega"], err)
    if not any(kv.startswith("use_graph=") for kv in opts):
        s.setOption("use_graph", 1)
    for kv in opts:
        if kv.startswith("use_graph="):
            s.setOption("use_graph", float(kv.split("=")[1]))
    s.buildSystem()
    lam = 1e-5 * s.maxDiagonal()
    for _ in range(3):
        s.setLambda(lam, True); assert s.solve(); s.restoreDiagonal()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        s.setLambda(lam, True); s.solve(); s.restoreDiagonal()
    dt = (time.perf_counter() - t0) / n
    st = s.stats()
    if os.environ.get("POSEGRAPH_CPU"):   # the CPU oracle (single thread) on the same system, for profiles/r1_posegraph.json
        import json
        o = O.OracleSolver(p, l, g["nP"], 0, schur=False)
        ko = o.add_edge_set(d, g["hidx"][g["vi"]], g["hidx"][g["vj"]])
        o.set_dims(ko, p, p)
        o.build_structure()
        o.set_edge_data(ko, J0, J1, g["omega"], err)
        o.build_system()
        o.set_lambda(lam, True)
        assert o.solve()                  # (ordering + symbolic factorisation happen once, like the reference: not timed)
        t0 = time.perf_counter()
        assert o.solve()
        cpu = time.perf_counter() - t0
        xg = s.x(); xo = o.x()
        print(json.dumps({"graph": name, "poses": int(g["nP"]), "edges": int(len(g["vi"])), "gpu_ms_per_solve": 1e3 * dt,
                          "cpu_oracle_ms_per_solve": 1e3 * cpu, "dx_rel_err": float(np.abs(xg - xo).max() / np.abs(xo).max()),
                          "fronts": st["numFronts"], "levels": st["numLevels"], "maxFrontDim": st["maxFrontDim"],
                          "choleskyNNZ": st["choleskyNNZ"]}))
    print("%-10s %-28s %.3f ms/solve  fronts %d levels %d maxdim %d nnz %d" % (name, " ".join(opts), 1e3 * dt, st["numFronts"], st["numLevels"], st["maxFrontDim"], st["choleskyNNZ"]))

for name in (os.environ.get("PG_GRAPHS") or "manhattan,sphere,sphere2500").split(","):
    for o in sys.argv[1:] or [""]:
        run(name, [x for x in o.split(",") if x])
