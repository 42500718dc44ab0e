"""Opcode histogram of ONE 64-pixel step of k_tick's main loop, from the gfx950 assembly of a build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -S --cuda-device-only dvo_kernels.hip -o k.s
    python scripts/step_loop_histogram.py LABEL=k.s [LABEL=other.s ...]
The default form of the kernel (k_tick<1> until round 3, k_tick<1, 0> since) is cut at its matrix instructions: the Gram
accumulation of step s - 1 sits in the middle of step s, so the instructions from the first matrix instruction of one step to the
first of the next are exactly one step.  The second step of the two-step loop body is taken (both halves are alike)."""
import collections
import re
import sys


def one_step(path):
    text = open(path).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_ZN7dvo_amd6k_tickILi1E(Li0E)?EEvNS_8TickArgsE:", l))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    body = text[start:end]
    mf = [i for i, l in enumerate(body) if "v_mfma_f32" in l]
    per_step = 36 if "4x4x1" in body[mf[0]] else 16
    groups = [mf[i:i + per_step] for i in range(0, len(mf), per_step)]
    a, b = groups[1][0], groups[2][0]  # (group 0 belongs to the first half of the loop body, whose step-0 guard skips it once)
    h = collections.Counter()
    for l in body[a:b]:
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if m:
            h[m.group(1)] += 1
    return h, per_step


def classify(op):
    if "mfma" in op:
        return "matrix"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "waits / nops"
    if op.startswith("s_"):
        return "SALU / branches"
    if op.startswith("ds_"):
        return "LDS"
    return "memory"


def main():
    cols = []
    for arg in sys.argv[1:]:
        label, path = arg.split("=", 1)
        cols.append((label, *one_step(path)))
    ops = sorted({o for _, h, _ in cols for o in h}, key=lambda o: (classify(o), -max(h[o] for _, h, _ in cols), o))
    w = max(len(o) for o in ops) + 2
    print(" " * w + "".join(f"{label:>24s}" for label, _, _ in cols))
    for cls in ("VALU", "matrix", "LDS", "memory", "SALU / branches", "waits / nops"):
        print(f"{cls + ' (total)':{w}s}" + "".join(f"{sum(v for o, v in h.items() if classify(o) == cls):24d}" for _, h, _ in cols))
    print()
    for o in ops:
        print(f"{o:{w}s}" + "".join(f"{h[o]:24d}" for _, h, _ in cols))


main()
