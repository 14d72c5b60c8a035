"""Static instruction mix per basic block of one kernel in a hipcc -S listing (tools/isa_blocks.py file.s kernel-substring).
Loops show up as blocks that a later s_cbranch jumps back to."""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l)
blocks, cur = [], {"name": "entry", "ins": []}
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append(cur)
        cur = {"name": m.group(1), "ins": []}
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur["ins"].append(t.split(";")[0].strip())
blocks.append(cur)
order = {b["name"]: i for i, b in enumerate(blocks)}
def cls(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_pk_fma_f64", "v_pk_mul_f64", "v_pk_add_f64")): return "f64"
    if op.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr", "v_cndmask", "v_readlane", "v_readfirstlane", "v_permlane", "v_bfe", "v_lshl", "v_and_b32", "v_or_b32")) or "_dpp" in op: return "move"
    if op.startswith(("v_add_u32", "v_add_co", "v_addc", "v_mad_u32", "v_mul_lo", "v_mul_u32", "v_sub_u32", "v_mad_i32", "v_lshl_add", "v_add3", "v_mul_hi", "v_min_", "v_max_", "v_ashr", "v_lshr", "v_cmp")): return "int"
    if op.startswith("v_div") or op.startswith("v_rcp") or op.startswith("v_rsq") or op.startswith("v_sqrt"): return "div"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "mem"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
tot = {}
for i, b in enumerate(blocks):
    c = {}
    back = []
    for ins in b["ins"]:
        op = ins.split()[0]
        k = cls(op)
        c[k] = c.get(k, 0) + 1
        tot[k] = tot.get(k, 0) + 1
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = ins.split()[-1]
            if tgt in order and order[tgt] <= i:
                back.append(tgt)
    if sum(c.values()) >= 8 or back:
        print(f"{b['name']:12s} n={sum(c.values()):4d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + (f"  <-loop to {back}" if back else ""))
print("total", tot)
