"""Static instruction mix of the hot loop of a k_tick variant (the innermost loop that contains v_mfma): per 64-pixel step.
usage: isa_count.py kernels.s 'k_tickILi1ELi1ELi4'"""
import re, sys
text = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(_ZN7dvo_amd6" + re.escape(key) + r"[^:\n]*):.*?s_endpgm", text, re.S | re.M)
body = m.group(0).splitlines()
# basic blocks
blocks, cur, name = {}, [], "entry"
order = []
for ln in body:
    t = ln.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks[name] = cur; order.append(name)
        name, cur = t.split(":")[0], []
    elif t and not t.startswith((";", ".")) and not t.endswith(":"):
        cur.append(t.split()[0])
blocks[name] = cur; order.append(name)
# find loops: a block that branches back to an earlier label; report spans containing mfma
lines = "\n".join(body)
def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_",)): return "valu"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"
# spans between a label and the backward branch to it
idx = {n: i for i, n in enumerate(order)}
best = None
for i, n in enumerate(order):
    for j in range(i, len(order)):
        # does block j end with a branch to n?
        pass
# simpler: count over the text between the first and last v_mfma inside loop headers flagged by the compiler comments
spans = []
stack = []
for k, ln in enumerate(body):
    if "Loop Header" in ln or "Inner Loop Header" in ln:
        stack.append(k)
mf = [k for k, ln in enumerate(body) if ln.strip().startswith("v_mfma")]
print("mfma instructions in function:", len(mf))
cnt = {}
for ln in body[mf[0] - 400 if mf[0] > 400 else 0: ]:
    pass
# count per loop: lines tagged 'in Loop: Header=BBx_y Depth=d'
loops = {}
curloop = None
for ln in body:
    t = ln.strip()
    mm = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", t)
    if re.match(r"^\.LBB\d+_\d+:", t):
        if mm: curloop = (mm.group(1), int(mm.group(2)))
        elif "Loop Header" in t:
            h = re.match(r"^\.L(BB\d+_\d+):", t).group(1)
            d = int(re.search(r"Depth=(\d+)", t).group(1))
            curloop = (h, d)
        else: curloop = None
        continue
    if t.startswith("; %bb"):
        mm2 = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", t)
        curloop = (mm2.group(1), int(mm2.group(2))) if mm2 else None
        continue
    if not t or t.startswith((";", ".")) or t.endswith(":"): continue
    if curloop is None: continue
    c = loops.setdefault(curloop, {})
    k = classify(t.split()[0]); c[k] = c.get(k, 0) + 1
for lp, c in loops.items():
    if c.get("mfma"):
        steps = c["mfma"] / (16.0 if c["mfma"] % 16 == 0 and c["mfma"] < 72 else 36.0)
        print(lp, c, "-> per step (", steps, "steps per trip):", {k: round(v / steps, 1) for k, v in c.items()})
