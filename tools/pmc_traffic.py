#!/usr/bin/env python
"""HBM traffic per bench kernel slot from two rocprofv3 PMC passes (rocpd SQLite DBs).

Recipe (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE do not fit one pass; no
other tracing domains next to --pmc):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/f -o f -- python bench.py --no-cpu-baseline --graph off --steps 2 --warmup 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/w -o w -- python bench.py --no-cpu-baseline --graph off --steps 2 --warmup 1
  python tools/pmc_traffic.py out/f/f_results.db out/w/w_results.db > profiles/rN_pmc_traffic.json
Values: KiB per counter per bench iteration: median over a kernel's launches x its launches per iteration
(iterations = dispatches of a once-per-iteration kernel).  On gfx950 FETCH_SIZE tallies 64 B per 128 B
request for wide streaming reads, so hbm_bytes_per_step = (2*FETCH + WRITE) KiB * 1024; hbm_bytes_raw
leaves FETCH unscaled (gather-type access is uncalibrated: the truth lies between the two)."""
import json
import sqlite3
import sys

SLOTS = [("assemble_vertex(pose)", ("ba_assemble_poses", "assemble_vertex_kernelILi2ELi6")),
         ("assemble_vertex(landmark)", ("ba_assemble_landmarks", "assemble_vertex_kernelILi2ELi3")),
         ("assemble_offdiag(Hpl)", ("assemble_offdiag",)),
         ("landmark_inverse", ("landmark_inverse",)),
         ("schur_tiles", ("schur_tile_kernel",)),
         ("schur_reduce", ("schur_reduce_kernel", "schur_rhs_kernel")),
         ("chol_factor(band chains)", ("band_wave_kernel",)),
         ("chol_factor(all levels)", ("front_factor_kernel", "wave_front_kernel")),
         ("chol_solve(all levels)", ("front_forward_kernel", "front_backward_kernel", "tree_backward_kernel", "permute_in_kernel", "permute_out_kernel")),
         ("back_substitute", ("back_substitute",)),
         ("set_lambda/restore", ("lambda_kernel",))]


def per_kernel(path, iters=None):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    rows = list(db.execute("select s.display_name, d.start, p.value from %s d join %s s on d.kernel_id = s.id "
                           "join %s p on p.event_id = d.id order by d.start" % (kd, ks, pe)))
    by = {}
    for name, start, val in rows:
        by.setdefault(name, []).append(val)
    if iters is None:   # bench iterations in the trace = dispatches of a kernel that runs once per iteration
        once = [len(v) for n, v in by.items() if "back_substitute" in n or "landmark_inverse_kernel" in n]
        iters = max(once) if once else 1
    out = {}
    for name, vals in by.items():
        if len(vals) < iters:   # not part of the iteration (e.g. the stand-alone landmark assembly behind the residual check)
            continue
        n = len(vals) // iters
        # median per launch x launches per step: a stray launch outside the iterations (e.g. the residual check's
        # on-demand Hpl materialisation at the end of bench.py) must not stand in for the per-iteration traffic
        sv = sorted(vals)
        med = sv[len(sv) // 2]
        out[name] = (med * n, n)
    return out


def per_kernel_multi(path):
    """Third pass (optional): several counters per dispatch -> {kernel name: {counter: sum per iteration}}"""
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    rows = list(db.execute("select s.display_name, i.name, d.id, sum(p.value) from %s d join %s s on d.kernel_id = s.id join %s p on p.event_id = d.id "
                           "join %s i on p.pmc_id = i.id group by s.display_name, i.name, d.id" % (kd, ks, pe, pi)))
    by = {}
    for name, ctr, did, val in rows:
        by.setdefault(name, {}).setdefault(ctr, []).append(val)
    once = [len(v) for n, c in by.items() for v in c.values() if "back_substitute" in n or "landmark_inverse_kernel" in n]
    iters = max(once) if once else 1
    out = {}
    for name, ctrs in by.items():
        for ctr, vals in ctrs.items():
            if len(vals) < iters:
                continue
            sv = sorted(vals)
            out.setdefault(name, {})[ctr] = sv[len(sv) // 2] * (len(vals) // iters)
    return out


def main():
    f, w = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    mf = per_kernel_multi(sys.argv[3]) if len(sys.argv) > 3 else {}
    res = {"_note": __doc__.split("Values:")[1].strip().replace("\n", " ")}
    if mf:
        res["_note_mfma"] = ("third pass --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES: sums over the device per bench iteration; "
                             "one v_mfma_f64_16x16x4_f64 wave instruction = 2048 flops; mfma_util in bench.py = flops / kernel time / 78.6 TFLOP/s "
                             "(dense fp64 matrix peak of the MI355X: 256 CUs x 4 SIMDs x 2048 flops / 64 cycles x 2.4 GHz)")
    for slot, keys in SLOTS:
        fk = sum(v for n, (v, c) in f.items() if any(k in n for k in keys))
        wk = sum(v for n, (v, c) in w.items() if any(k in n for k in keys))
        cnt = sum(c for n, (v, c) in f.items() if any(k in n for k in keys))
        if cnt == 0:
            continue
        res[slot] = dict(launches_per_step=cnt, FETCH_SIZE_KiB=fk, WRITE_SIZE_KiB=wk,
                         hbm_bytes_raw=(fk + wk) * 1024.0, hbm_bytes_per_step=(2 * fk + wk) * 1024.0)
        for ctr in ("SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"):
            v = sum(c.get(ctr, 0) for n, c in mf.items() if any(k in n for k in keys))
            if mf:
                res[slot][ctr + "_per_step"] = v
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
